from .roi_align import ROIAlign, roi_align  # noqa: F401
from .nms import nms, nms_pair, nms_pair_sorted_joint  # noqa: F401
