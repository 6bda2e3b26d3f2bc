"""ResNet trunk parameter holders (reference: disprcnn/modeling/backbone/resnet.py:81-146,229-345).

State-dict compatible with the reference (`stem.conv1/bn1`, `layerL.B.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}}`).
Fork quirk kept: ``FrozenBatchNorm2d`` is a plain ``nn.BatchNorm2d`` (reference layers/batch_norm.py:6-31), so eval uses
running statistics.  No arithmetic here: the trunk runs on the HIP tap-conv engine (backbone/runtime.py).
"""
import math

import torch
from torch import nn

STAGE_BLOCKS = {"R-50": (3, 4, 6, 3), "R-101": (3, 4, 23, 3), "R-152": (3, 8, 36, 3)}


def _kaiming_uniform_a1_(w):
    """nn.init.kaiming_uniform_(w, a=1): bound = sqrt(6 / ((1 + a^2) * fan_in)) = sqrt(3 / fan_in)."""
    fan_in = w.shape[1] * math.prod(w.shape[2:])
    with torch.no_grad():
        w.uniform_(-math.sqrt(3.0 / fan_in), math.sqrt(3.0 / fan_in))


class BottleneckUnit(nn.Module):
    """1x1 (stride here: STRIDE_IN_1X1=True) -> 3x3 -> 1x1, each + BN; projection shortcut when channels change."""

    def __init__(self, cin, mid, cout, stride):
        super().__init__()
        self.downsample = None
        if cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
        self.conv1, self.bn1 = nn.Conv2d(cin, mid, 1, stride, bias=False), nn.BatchNorm2d(mid)
        self.conv2, self.bn2 = nn.Conv2d(mid, mid, 3, 1, 1, bias=False), nn.BatchNorm2d(mid)
        self.conv3, self.bn3 = nn.Conv2d(mid, cout, 1, bias=False), nn.BatchNorm2d(cout)
        self.stride = stride
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                _kaiming_uniform_a1_(m.weight)


class Stem(nn.Module):
    def __init__(self, cout=64):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(3, cout, 7, 2, 3, bias=False), nn.BatchNorm2d(cout)
        _kaiming_uniform_a1_(self.conv1.weight)


class ResNet(nn.Module):
    """Stages 2..5 of R-50/101/152 (all four returned, as the *-FPN stage specs do)."""

    def __init__(self, arch="R-50", stem_out=64, res2_out=256, width=64):
        super().__init__()
        self.stem = Stem(stem_out)
        cin = stem_out
        for i, nblk in enumerate(STAGE_BLOCKS[arch]):
            mid, cout = width * 2 ** i, res2_out * 2 ** i
            units = []
            for b in range(nblk):
                units.append(BottleneckUnit(cin, mid, cout, 2 if (b == 0 and i > 0) else 1))
                cin = cout
            setattr(self, f"layer{i + 1}", nn.Sequential(*units))
        self.stage_channels = [res2_out * 2 ** i for i in range(4)]

    def forward(self, x):
        raise RuntimeError("ResNet is a parameter holder; run it through build_backbone(cfg) (HIP engine)")
