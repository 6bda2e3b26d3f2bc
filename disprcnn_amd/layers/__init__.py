from .roi_align import ROIAlign, roi_align  # noqa: F401
from .nms import nms, nms_pair  # noqa: F401
