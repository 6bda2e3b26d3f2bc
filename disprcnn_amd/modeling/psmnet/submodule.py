"""Parameter containers of iDispNet's sub-networks (MI355X build).

API/state-dict mirror of the reference ``disprcnn/modeling/psmnet/submodule.py``: the module
tree below exists so that ``state_dict()`` has exactly the reference's keys and shapes
 and reference checkpoints load unchanged.  None of these containers computes
anything in the product path: the arithmetic runs in the HIP engine
(``disprcnn_amd/modeling/psmnet/runtime.py`` -> libdisprcnn_hip.so).

Layer tables (what the reference builds imperatively, written here as data):
  2D trunk   reference submodule.py:60-104     3D trunk   reference stackhourglass.py:7-88
"""
import math

import torch
from torch import nn


def _holder(conv_cls, cin, cout, k, stride=1, pad=0, dilation=1, **kw):
    """conv (bias-free) parameter holder."""
    return conv_cls(cin, cout, kernel_size=k, stride=stride, padding=pad, dilation=dilation, bias=False, **kw)


def conv_bn_2d(cin, cout, k, stride, pad, dilation):
    """Index 0 = Conv2d, index 1 = BatchNorm2d (keys '<p>.0.weight', '<p>.1.*').
    Padding rule of the reference (submodule.py:15): dilation>1 replaces pad by dilation."""
    return nn.Sequential(_holder(nn.Conv2d, cin, cout, k, stride, dilation if dilation > 1 else pad, dilation),
                         nn.BatchNorm2d(cout))


def conv_bn_3d(cin, cout, stride=1):
    """Index 0 = Conv3d(k3,p1), index 1 = BatchNorm3d (reference submodule.py:19-22)."""
    return nn.Sequential(_holder(nn.Conv3d, cin, cout, 3, stride, 1), nn.BatchNorm3d(cout))


def deconv_bn_3d(cin, cout):
    """ConvTranspose3d(k3,s2,p1,op1) + BatchNorm3d (reference stackhourglass.py:22-30)."""
    return nn.Sequential(nn.ConvTranspose3d(cin, cout, kernel_size=3, stride=2, padding=1, output_padding=1, bias=False),
                         nn.BatchNorm3d(cout))


def with_relu_slots(*mods):
    """nn.Sequential whose odd indices are parameter-less activation slots, e.g. (m0, relu, m2, relu)."""
    seq = []
    for m in mods:
        seq.append(m)
        seq.append(nn.ReLU(inplace=True))
    return seq


class ResidualUnit2d(nn.Module):
    """Holder for one BasicBlock (reference submodule.py:25-48): conv1=[convbn,relu], conv2=convbn, downsample."""

    def __init__(self, cin, cout, stride, dilation, project):
        super().__init__()
        self.conv1 = nn.Sequential(conv_bn_2d(cin, cout, 3, stride, 1, dilation), nn.ReLU(inplace=True))
        self.conv2 = conv_bn_2d(cout, cout, 3, 1, 1, dilation)
        self.downsample = (nn.Sequential(_holder(nn.Conv2d, cin, cout, 1, stride), nn.BatchNorm2d(cout))
                           if project else None)
        self.stride, self.dilation = stride, dilation


# (name, width, blocks, stride of first block, dilation) -- reference submodule.py:71-74
TRUNK_STAGES = (("layer1", 32, 3, 1, 1), ("layer2", 64, 16, 2, 1), ("layer3", 128, 3, 1, 1), ("layer4", 128, 3, 1, 2))
# (name, pooling window) -- reference submodule.py:76-90; order of concatenation is :134-135
SPP_BRANCHES = (("branch1", 56), ("branch2", 32), ("branch3", 16), ("branch4", 8))


class feature_extraction(nn.Module):
    """Holder of the siamese 2D CNN + SPP weights (reference submodule.py:60-104)."""

    def __init__(self):
        super().__init__()
        self.firstconv = nn.Sequential(*with_relu_slots(conv_bn_2d(3, 32, 3, 2, 1, 1), conv_bn_2d(32, 32, 3, 1, 1, 1),
                                                        conv_bn_2d(32, 32, 3, 1, 1, 1)))
        width = 32
        for name, planes, nblk, stride, dil in TRUNK_STAGES:
            units = []
            for b in range(nblk):
                s = stride if b == 0 else 1
                units.append(ResidualUnit2d(width, planes, s, dil, project=(b == 0 and (s != 1 or width != planes))))
                width = planes
            setattr(self, name, nn.Sequential(*units))
        for name, k in SPP_BRANCHES:
            setattr(self, name, nn.Sequential(nn.AvgPool2d((k, k), stride=(k, k)), conv_bn_2d(128, 32, 1, 1, 0, 1),
                                              nn.ReLU(inplace=True)))
        self.lastconv = nn.Sequential(conv_bn_2d(320, 128, 3, 1, 1, 1), nn.ReLU(inplace=True),
                                      _holder(nn.Conv2d, 128, 32, 1))

    def forward(self, x):
        raise RuntimeError("feature_extraction is a parameter holder; run it through PSMNet (HIP engine)")


def reference_init_(module):
    """The reference's init loop (stackhourglass.py:90-104): He-normal on Conv2d/Conv3d with
    n = prod(kernel)*Cout, BN (1,0); ConvTranspose3d is not an nn.Conv3d and keeps the torch default."""
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.Conv3d)):
            n = math.prod(m.kernel_size) * m.out_channels
            with torch.no_grad():
                m.weight.normal_(0.0, math.sqrt(2.0 / n))
        elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
            with torch.no_grad():
                m.weight.fill_(1.0)
                m.bias.zero_()
