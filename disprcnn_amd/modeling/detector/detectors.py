"""Model factory (reference: disprcnn/modeling/detector/detectors.py:6-14).  Only the meta-architecture that owns the
disparity hot path is built here; the 2D detection stage (GeneralizedRCNN / DispRCNN) is out of scope."""
from .disprcnn3d import DispRCNN3D

_DETECTION_META_ARCHITECTURES = {"DispRCNN3D": DispRCNN3D}


def build_detection_model(cfg):
    name = cfg.MODEL.META_ARCHITECTURE
    if name not in _DETECTION_META_ARCHITECTURES:
        raise NotImplementedError(f"META_ARCHITECTURE {name!r}: only DispRCNN3D (the instance-disparity stage) is built on MI355X")
    return _DETECTION_META_ARCHITECTURES[name](cfg)
