"""Model factory (reference: disprcnn/modeling/detector/detectors.py:6-14): the instance-disparity stage (DispRCNN3D, the owner of
the hot path) and the stereo 2D stage in front of it (DispRCNN, inference)."""
from .disprcnn import DispRCNN
from .disprcnn3d import DispRCNN3D

_DETECTION_META_ARCHITECTURES = {"DispRCNN3D": DispRCNN3D, "DispRCNN": DispRCNN}


def build_detection_model(cfg):
    name = cfg.MODEL.META_ARCHITECTURE
    if name not in _DETECTION_META_ARCHITECTURES:
        raise NotImplementedError(f"META_ARCHITECTURE {name!r}: only DispRCNN3D and DispRCNN are built on MI355X")
    return _DETECTION_META_ARCHITECTURES[name](cfg)
