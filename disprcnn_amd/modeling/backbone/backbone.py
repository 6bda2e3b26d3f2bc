"""build_backbone(cfg) -- drop-in for disprcnn.modeling.backbone.build_backbone (reference backbone/backbone.py:23-45,74-79).

Returns an ``nn.Sequential(OrderedDict(body=ResNet, fpn=FPN))`` subclass with ``.out_channels`` and the reference's
state-dict keys (``body.*``, ``fpn.*``); calling it runs ResNet-FPN on the HIP engine and returns the tuple of 5 maps."""
from collections import OrderedDict

from torch import nn

from .fpn import FPN
from .resnet import ResNet

_REGISTERED = ("R-50-FPN", "R-101-FPN", "R-152-FPN")


class ResNetFPNBackbone(nn.Sequential):
    def __init__(self, arch, res2_out=256, out_channels=256):
        body = ResNet(arch, res2_out=res2_out)
        super().__init__(OrderedDict([("body", body), ("fpn", FPN(body.stage_channels, out_channels))]))
        self.out_channels = out_channels
        self._rt = None

    def forward(self, x):
        from .runtime import BackboneRuntime
        if self.training:
            raise NotImplementedError("training-mode BatchNorm / backward are not built on the HIP engine yet")
        if self._rt is None or self._rt.device != x.device:
            self._rt = BackboneRuntime(self, x.device)
        return self._rt.forward(x)


def build_backbone(cfg):
    name = cfg.MODEL.BACKBONE.CONV_BODY
    if name not in _REGISTERED:
        raise NotImplementedError(f"CONV_BODY {name!r}: only {_REGISTERED} are built on MI355X")
    return ResNetFPNBackbone(name[:-4], getattr(cfg.MODEL.RESNETS, "RES2_OUT_CHANNELS", 256),
                             cfg.MODEL.RESNETS.BACKBONE_OUT_CHANNELS)
