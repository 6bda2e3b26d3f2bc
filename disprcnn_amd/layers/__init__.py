from .roi_align import ROIAlign, roi_align  # noqa: F401
