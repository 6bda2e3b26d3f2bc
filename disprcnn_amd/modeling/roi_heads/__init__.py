from .roi_heads import StereoCombinedROIHeads, build_roi_heads  # noqa: F401
