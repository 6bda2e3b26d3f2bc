from .stackhourglass import PSMNet, hourglass  # noqa: F401
