from .backbone import build_backbone  # noqa: F401
