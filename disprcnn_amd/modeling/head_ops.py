"""Engine-backed layers of the 2D stage's heads: nn.Conv2d / nn.Linear parameter holders whose arithmetic runs on the HIP engine
(convolutions: the same MFMA kernels as the backbone, bias in the folded-BN epilogue slot) or, for the plain fully connected layers,
as library GEMMs (torch.addmm -> hipBLASLt).  Inference only; there is no CPU path."""
from collections import OrderedDict

import torch

from .. import engine as E


class EngineConv2d:
    """y = act(conv2d(x, conv.weight) + conv.bias) on dense [N,C,H,W] tensors.  Packed weights are rebuilt when the parameters
    change (version counters); plans and blocked workspaces are cached per input shape (a short LRU: FPN levels, ROI batches)."""

    MAX_SHAPES = 12

    def __init__(self, conv, relu):
        if conv.groups != 1 or conv.stride[0] != conv.stride[1] or conv.padding[0] != conv.padding[1] or conv.dilation[0] != conv.dilation[1]:
            raise NotImplementedError("EngineConv2d: square stride / padding / dilation, groups = 1")
        self.conv, self.relu = conv, bool(relu)
        self._ver, self._plans = None, OrderedDict()
        self._wp = self._sc = self._sh = None

    def _weights(self, dev):
        c = self.conv
        ver = (c.weight._version, c.weight.data_ptr(), None if c.bias is None else c.bias._version, dev)
        if ver != self._ver:
            w = c.weight.detach().to(device=dev, dtype=torch.float32)
            cp = E.cout_pad_of(w.shape[0])
            self._wp = E.pack_conv_weight(w)
            self._sc = torch.ones(cp, device=dev)
            self._sh = torch.zeros(cp, device=dev)
            if c.bias is not None:
                self._sh[: w.shape[0]] = c.bias.detach().to(device=dev, dtype=torch.float32)
            self._w32, self._ver = w, ver
            for ent in self._plans.values():
                ent["w16"] = None
        return self._wp, self._sc, self._sh

    def __call__(self, x):
        E.require_gpu(x, "EngineConv2d")
        c = self.conv
        n, cin, h, w = x.shape
        k, s, p, d = c.kernel_size[0], c.stride[0], c.padding[0], c.dilation[0]
        cout = c.out_channels
        dev = x.device
        if n == 0:
            oh, ow = (h + 2 * p - d * (k - 1) - 1) // s + 1, (w + 2 * p - d * (k - 1) - 1) // s + 1
            return x.new_zeros(0, cout, oh, ow)
        wp, sc, sh = self._weights(dev)
        key = (n, h, w, dev)
        ent = self._plans.get(key)
        if ent is None:
            halo = max(p, 1)
            oh, ow = (h + 2 * p - d * (k - 1) - 1) // s + 1, (w + 2 * p - d * (k - 1) - 1) // s + 1
            xb = E.Blocked(n, cin, 1, h, w, 0, halo, halo, dev)
            yb = E.Blocked(n, cout, 1, oh, ow, 0, 1, 1, dev)
            ent = dict(x=xb, y=yb, plan=E.plan_conv2d(xb, yb, k, s, p, d, cout, self.relu), w16=None)
            self._plans[key] = ent
            while len(self._plans) > self.MAX_SHAPES:
                self._plans.popitem(last=False)
        else:
            self._plans.move_to_end(key)
        if ent["w16"] is None:
            ent["w16"] = ent["plan"].pack16(self._w32)
        ent["x"].from_dense(x.float())
        ent["plan"].run(ent["x"], wp, sc, sh, ent["y"], None, w16=ent["w16"])
        return ent["y"].to_dense()[:, :, 0]


def linear(x, layer, relu=False):
    """x [R, in] @ layer.weight^T + bias (library GEMM); layer is an nn.Linear or an nn.Conv2d that acts on a full window."""
    w = layer.weight.detach().to(device=x.device, dtype=torch.float32).reshape(layer.weight.shape[0], -1)
    y = torch.addmm(layer.bias.detach().to(device=x.device, dtype=torch.float32), x.float(), w.t()) if layer.bias is not None else x.float() @ w.t()
    return torch.relu_(y) if relu else y
