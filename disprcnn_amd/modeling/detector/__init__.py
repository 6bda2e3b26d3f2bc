from .detectors import build_detection_model  # noqa: F401
from .disprcnn3d import DispRCNN3D  # noqa: F401
from .disprcnn import DispRCNN, default_cfg_2d  # noqa: F401
