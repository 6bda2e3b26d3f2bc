"""Tensor-level operator wrappers over the C ABI (dense reference layouts in, dense out).

These are the functions a maintainer of the reference would call from
``PSMNet.forward`` (stackhourglass.py:115-128,169-173).  GPU only; errors surface as RuntimeError.
"""
import torch

from . import _lib
from . import engine as E


def cost_volume(left, right, maxdisp, mindisp):
    """[N,C,H',W'] x2 -> [N,2C,(maxdisp-mindisp)//4,H',W'] (bit-exact with stackhourglass.py:115-128)."""
    E.require_gpu(left, "cost_volume"); E.require_gpu(right, "cost_volume")
    if left.shape != right.shape or left.dim() != 4:
        raise ValueError("cost_volume expects two [N,C,H,W] tensors of equal shape")
    left, right = left.contiguous(), right.contiguous()
    n, c, h, w = left.shape
    dp = (maxdisp - mindisp) // 4
    cost = torch.empty(n, 2 * c, dp, h, w, dtype=torch.float32, device=left.device)
    st = _lib.lib().drc_cost_volume_fwd(E._ptr(left), E._ptr(right), E._ptr(cost), n, c, dp, h, w, mindisp // 4, maxdisp // 4,
                                        E._stream_ptr(left.device))
    _lib.check(st, "drc_cost_volume_fwd")
    return cost


def cost_volume_backward(gcost, maxdisp, mindisp):
    E.require_gpu(gcost, "cost_volume_backward")
    gcost = gcost.contiguous()
    n, c2, dp, h, w = gcost.shape
    c = c2 // 2
    gl = torch.empty(n, c, h, w, dtype=torch.float32, device=gcost.device)
    gr = torch.empty_like(gl)
    st = _lib.lib().drc_cost_volume_bwd(E._ptr(gcost), E._ptr(gl), E._ptr(gr), n, c, dp, h, w, mindisp // 4, maxdisp // 4,
                                        E._stream_ptr(gcost.device))
    _lib.check(st, "drc_cost_volume_bwd")
    return gl, gr


def upsample_softargmin(cost, maxdisp, mindisp, H, W):
    """[N,1,D',H',W'] or [N,D',H',W'] -> disparity [N,H,W] (stackhourglass.py:169-173 fused)."""
    E.require_gpu(cost, "upsample_softargmin")
    if cost.dim() == 5:
        cost = cost[:, 0]
    cost = cost.contiguous()
    disp = torch.empty(cost.shape[0], H, W, dtype=torch.float32, device=cost.device)
    E.upsample_softargmin(cost, disp, maxdisp, mindisp)
    return disp


def conv3d_bn(x, weight, scale, shift, stride=1, relu=False, residual=None, transposed=False):
    """Dense-layout convenience wrapper around the tap-conv engine (used by the layer-level parity tests):
    x [N,Cin,D,H,W] -> [N,Cout,D',H',W'];  y = act(scale*conv(x)+shift (+residual))."""
    E.require_gpu(x, "conv3d_bn")
    n, cin, d, h, w = x.shape
    cout = weight.shape[1] if transposed else weight.shape[0]
    dev = x.device
    xb = E.Blocked(n, cin, d, h, w, 1, 1, 1, dev).from_dense(x)
    if transposed:
        od, oh, ow = 2 * d, 2 * h, 2 * w
    else:
        od, oh, ow = (-(-d // stride), -(-h // stride), -(-w // stride))
    yb = E.Blocked(n, cout, od, oh, ow, 1, 1, 1, dev)
    rb = E.Blocked(n, cout, od, oh, ow, 1, 1, 1, dev).from_dense(residual) if residual is not None else None
    plan = E.plan_deconv3d(xb, yb, cout, relu) if transposed else E.plan_conv3d(xb, yb, stride, cout, relu)
    wp = E.pack_weight(weight.to(dev).float(), transposed)
    cp = wp.shape[3]
    sc = torch.ones(cp, device=dev); sh = torch.zeros(cp, device=dev)
    sc[:cout] = scale; sh[:cout] = shift
    plan.run(xb, wp, sc, sh, yb, rb, w16=plan.pack16(weight.to(dev).float(), transposed))
    return yb.to_dense()


def conv2d_bn(x, weight, scale, shift, stride=1, pad=1, dilation=1, relu=False, residual=None, in_halo=None):
    """x [N,Cin,H,W] -> [N,Cout,H',W'] through the same engine (D=1)."""
    E.require_gpu(x, "conv2d_bn")
    n, cin, h, w = x.shape
    cout, _, k, _ = weight.shape
    dev = x.device
    halo = in_halo if in_halo is not None else max(pad, 1)
    xb = E.Blocked(n, cin, 1, h, w, 0, halo, halo, dev).from_dense(x)
    oh = (h + 2 * pad - dilation * (k - 1) - 1) // stride + 1
    ow = (w + 2 * pad - dilation * (k - 1) - 1) // stride + 1
    yb = E.Blocked(n, cout, 1, oh, ow, 0, 1, 1, dev)
    rb = E.Blocked(n, cout, 1, oh, ow, 0, 2, 2, dev).from_dense(residual) if residual is not None else None
    plan = E.plan_conv2d(xb, yb, k, stride, pad, dilation, cout, relu)
    wp = E.pack_conv_weight(weight.to(dev).float())
    cp = E.cout_pad_of(cout)
    sc = torch.ones(cp, device=dev); sh = torch.zeros(cp, device=dev)
    sc[:cout] = scale; sh[:cout] = shift
    plan.run(xb, wp, sc, sh, yb, rb, w16=plan.pack16(weight.to(dev).float()))
    return yb.to_dense()[:, :, 0]
