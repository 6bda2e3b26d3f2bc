"""FPN parameter holders (reference: disprcnn/modeling/backbone/fpn.py:7-82).  Fork quirks kept by the runtime: the top
level is returned WITHOUT its 3x3 layer block (:49-50), the top-down path is bilinear with align_corners=False (:62-64),
and `LastLevelMaxPool` appends max_pool2d(x, 1, 2, 0) (:80-82).  `fpn_layer4` exists (and loads) but is unused, as there."""
import math

import torch
from torch import nn


class FPN(nn.Module):
    def __init__(self, in_channels_list, out_channels):
        super().__init__()
        for i, cin in enumerate(in_channels_list, 1):
            for name, k in ((f"fpn_inner{i}", 1), (f"fpn_layer{i}", 3)):
                conv = nn.Conv2d(cin if k == 1 else out_channels, out_channels, k, 1, k // 2, bias=True)
                fan_in = conv.weight.shape[1] * k * k
                with torch.no_grad():
                    conv.weight.uniform_(-math.sqrt(3.0 / fan_in), math.sqrt(3.0 / fan_in))     # kaiming_uniform_(a=1)
                    conv.bias.zero_()
                self.add_module(name, conv)
        self.levels = len(in_channels_list)

    def forward(self, x):
        raise RuntimeError("FPN is a parameter holder; run it through build_backbone(cfg) (HIP engine)")
