from .bounding_box import BoxList  # noqa: F401
from .image_list import ImageList, to_image_list  # noqa: F401
from .disparity import DisparityMap  # noqa: F401
