"""CPU ORACLE (test infrastructure -- never imported by the product path).

A from-scratch restatement, on torch-CPU / numpy primitives, of the instance-disparity
hot path of zju3dv/disprcnn.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module.  Each function cites
the reference lines it follows (paths relative to /root/reference).

Pinning: ``tests/test_oracle_golden.py`` checks every function here against
fixtures produced by the *imported reference itself* (``tests/golden/make_golden.py``,
run once in the authoring container).  Parity status: PINNED for a1-a8, a10.

The oracle is functional: it takes a plain ``state_dict`` (name -> tensor) with the
reference's key layout and fp32 or fp64 tensors.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


# --------------------------------------------------------------------------- a1
def cost_volume(left, right, maxdisp, mindisp):
    """Concat-shift cost volume.  Ref: stackhourglass.py:115-128.

    cost[n,c,j,y,x]   = L[n,c,y,x]      if 0 <= x-i < W' else 0
    cost[n,C+c,j,y,x] = R[n,c,y,x-i]    if 0 <= x-i < W' else 0,  i = mindisp//4 + j
    (Python floor division; the loop is range(mindisp//4, maxdisp//4) while the
    allocated depth is (maxdisp-mindisp)//4 -- restated literally.)
    """
    n, c, h, w = left.shape
    dp = (maxdisp - mindisp) // 4
    cost = torch.zeros(n, 2 * c, dp, h, w, dtype=left.dtype)
    lo = mindisp // 4
    for i in range(mindisp // 4, maxdisp // 4):
        j = i - lo
        x0, x1 = max(i, 0), min(w + i, w)   # valid x: 0 <= x - i < w
        if x1 <= x0:
            continue
        cost[:, :c, j, :, x0:x1] = left[:, :, :, x0:x1]
        cost[:, c:, j, :, x0:x1] = right[:, :, :, x0 - i:x1 - i]
    return cost


def cost_volume_numpy(left, right, maxdisp, mindisp):
    """Pure-numpy gather form of the same closed form (independent second statement)."""
    n, c, h, w = left.shape
    dp = (maxdisp - mindisp) // 4
    lo = mindisp // 4
    out = np.zeros((n, 2 * c, dp, h, w), dtype=left.dtype)
    xs = np.arange(w)
    for j in range(dp):
        i = lo + j
        if i >= maxdisp // 4:
            continue
        src = xs - i
        ok = (src >= 0) & (src < w)
        out[:, :c, j][..., ok] = left[..., ok]
        out[:, c:, j][..., ok] = right[..., src[ok]]
    return out


# ----------------------------------------------------------------------- a2..a6
def _bn(sd, prefix, x, training=False):
    """BatchNorm (eps 1e-5).  eval: running stats; training: batch stats (biased var).
    Ref: submodule.py:19-22 (B3d = nn.BatchNorm3d)."""
    g, b = sd[prefix + ".weight"].to(x.dtype), sd[prefix + ".bias"].to(x.dtype)
    if training:
        dims = [0] + list(range(2, x.dim()))
        mean = x.mean(dims)
        var = x.var(dims, unbiased=False)
    else:
        mean, var = sd[prefix + ".running_mean"].to(x.dtype), sd[prefix + ".running_var"].to(x.dtype)
    shape = [1, -1] + [1] * (x.dim() - 2)
    return (x - mean.view(shape)) / torch.sqrt(var.view(shape) + BN_EPS) * g.view(shape) + b.view(shape)


def convbn3d(sd, prefix, x, stride=1, training=False):
    """convbn_3d: Conv3d(k3, pad1, bias=False) + BN3d.  Ref: submodule.py:19-22."""
    y = F.conv3d(x, sd[prefix + ".0.weight"].to(x.dtype), None, stride, 1)
    return _bn(sd, prefix + ".1", y, training)


def deconvbn3d(sd, prefix, x, training=False):
    """ConvTranspose3d(k3,s2,p1,output_padding=1,bias=False)+BN3d.  Ref: stackhourglass.py:22-30."""
    y = F.conv_transpose3d(x, sd[prefix + ".0.weight"].to(x.dtype), None, stride=2, padding=1, output_padding=1)
    return _bn(sd, prefix + ".1", y, training)


def hourglass(sd, p, x, presqu, postsqu, training=False):
    """Ref: stackhourglass.py:32-51."""
    out = F.relu(convbn3d(sd, p + ".conv1.0", x, 2, training))
    pre = convbn3d(sd, p + ".conv2", out, 1, training)
    pre = F.relu(pre + postsqu) if postsqu is not None else F.relu(pre)
    out = F.relu(convbn3d(sd, p + ".conv3.0", pre, 2, training))
    out = F.relu(convbn3d(sd, p + ".conv4.0", out, 1, training))
    skip = presqu if presqu is not None else pre
    post = F.relu(deconvbn3d(sd, p + ".conv5", out, training) + skip)
    out = deconvbn3d(sd, p + ".conv6", post, training)
    return out, pre, post


def regressor3d(sd, cost, training=False, want_intermediates=False):
    """dres0..dres4 + classif1..3 (cumulative).  Ref: stackhourglass.py:130-144."""
    x = F.relu(convbn3d(sd, "dres0.0", cost, 1, training))
    cost0 = F.relu(convbn3d(sd, "dres0.2", x, 1, training))
    y = F.relu(convbn3d(sd, "dres1.0", cost0, 1, training))
    cost0 = convbn3d(sd, "dres1.2", y, 1, training) + cost0
    out1, pre1, post1 = hourglass(sd, "dres2", cost0, None, None, training)
    out1 = out1 + cost0
    out2, pre2, post2 = hourglass(sd, "dres3", out1, pre1, post1, training)
    out2 = out2 + cost0
    out3, pre3, post3 = hourglass(sd, "dres4", out2, pre1, post2, training)   # pre1 reused: as the reference
    out3 = out3 + cost0

    def classif(p, t):
        t = F.relu(convbn3d(sd, p + ".0", t, 1, training))
        return F.conv3d(t, sd[p + ".2.weight"].to(t.dtype), None, 1, 1)

    cost1 = classif("classif1", out1)
    cost2 = classif("classif2", out2) + cost1
    cost3 = classif("classif3", out3) + cost2
    if want_intermediates:
        return dict(cost0=cost0, out1=out1, out2=out2, out3=out3, pre1=pre1, post1=post1, post2=post2,
                    cost1=cost1, cost2=cost2, cost3=cost3)
    return cost1, cost2, cost3


# --------------------------------------------------------------------------- a7
def _lerp_axis(x, out_len, dim):
    """1-D linear resample along ``dim``, align_corners=True: src = dst*(in-1)/(out-1)."""
    n_in = x.shape[dim]
    if out_len == 1 or n_in == 1:
        idx = torch.zeros(out_len, dtype=torch.long)
        return x.index_select(dim, idx)
    pos = torch.arange(out_len, dtype=torch.float64) * ((n_in - 1) / (out_len - 1))
    i0 = pos.floor().clamp(max=n_in - 1).long()
    i1 = (i0 + 1).clamp(max=n_in - 1)
    t = (pos - i0.to(torch.float64)).to(x.dtype)
    shape = [1] * x.dim()
    shape[dim] = out_len
    t = t.view(shape)
    return x.index_select(dim, i0) * (1 - t) + x.index_select(dim, i1) * t


def upsample_softargmin(cost, maxdisp, mindisp, H, W, separable=False):
    """trilinear(align_corners=True) -> softmax over D (of +cost) -> sum d*p, d=arange(mindisp,maxdisp).

    Ref: stackhourglass.py:169-173 and submodule.py:51-57.  cost: [N,1,D',H',W'] -> [N,H,W].
    ``separable=True`` uses this file's own D->H->W lerps instead of F.interpolate
    (differs from ATen by summation order only, ~1e-4 px)."""
    D = maxdisp - mindisp
    if separable:
        c = _lerp_axis(_lerp_axis(_lerp_axis(cost, D, 2), H, 3), W, 4)
    else:
        c = F.interpolate(cost, [D, H, W], mode="trilinear", align_corners=True)
    c = c.squeeze(1)
    p = F.softmax(c, dim=1)
    d = torch.arange(mindisp, maxdisp, dtype=c.dtype).view(1, D, 1, 1)
    return (p * d).sum(1)


# --------------------------------------------------------------------------- a8
def _convbn2d(sd, p, x, stride, pad, dilation, training=False):
    """convbn: Conv2d(bias=False, padding = dilation if dilation>1 else pad) + BN2d.  Ref: submodule.py:13-16."""
    y = F.conv2d(x, sd[p + ".0.weight"].to(x.dtype), None, stride, dilation if dilation > 1 else pad, dilation)
    return _bn(sd, p + ".1", y, training)


def _basic_block(sd, p, x, stride, pad, dilation, training=False):
    """BasicBlock: conv-bn-relu, conv-bn, (+downsample(x)), add, NO trailing relu.  Ref: submodule.py:25-48."""
    out = F.relu(_convbn2d(sd, p + ".conv1.0", x, stride, pad, dilation, training))
    out = _convbn2d(sd, p + ".conv2", out, 1, pad, dilation, training)
    if (p + ".downsample.0.weight") in sd:
        x = F.conv2d(x, sd[p + ".downsample.0.weight"].to(x.dtype), None, stride)
        x = _bn(sd, p + ".downsample.1", x, training)
    return out + x


def feature_extraction(sd, x, training=False, prefix="feature_extraction"):
    """PSMNet 2D siamese CNN + SPP.  Ref: submodule.py:60-139."""
    P = prefix
    o = F.relu(_convbn2d(sd, P + ".firstconv.0", x, 2, 1, 1, training))
    o = F.relu(_convbn2d(sd, P + ".firstconv.2", o, 1, 1, 1, training))
    o = F.relu(_convbn2d(sd, P + ".firstconv.4", o, 1, 1, 1, training))
    for name, nblk, stride, dil in (("layer1", 3, 1, 1), ("layer2", 16, 2, 1), ("layer3", 3, 1, 1), ("layer4", 3, 1, 2)):
        for b in range(nblk):
            o = _basic_block(sd, f"{P}.{name}.{b}", o, stride if b == 0 else 1, 1, dil, training)
        if name == "layer2":
            raw = o
    skip = o
    hw = skip.shape[2:]
    branches = []
    for name, k in (("branch1", 56), ("branch2", 32), ("branch3", 16), ("branch4", 8)):
        b = F.avg_pool2d(skip, k, k)
        b = F.relu(_convbn2d(sd, f"{P}.{name}.1", b, 1, 0, 1, training))
        branches.append(F.interpolate(b, hw, mode="bilinear", align_corners=True))
    b1, b2, b3, b4 = branches
    feat = torch.cat((raw, skip, b4, b3, b2, b1), 1)
    feat = F.relu(_convbn2d(sd, P + ".lastconv.0", feat, 1, 1, 1, training))
    return F.conv2d(feat, sd[P + ".lastconv.2.weight"].to(feat.dtype), None)


# ---------------------------------------------------------------- whole forward
def psmnet_from_features(sd, fl, fr, maxdisp, mindisp, H, W, training=False):
    """Config-A entry: features -> cost volume -> regressor -> soft-argmin."""
    cost = cost_volume(fl, fr, maxdisp, mindisp)
    c1, c2, c3 = regressor3d(sd, cost, training)
    if training:
        return tuple(upsample_softargmin(c, maxdisp, mindisp, H, W) for c in (c1, c2, c3))
    return upsample_softargmin(c3, maxdisp, mindisp, H, W)


def psmnet_forward(sd, left, right, maxdisp, mindisp, training=False):
    """Ref: PSMNet.forward, stackhourglass.py:106-174."""
    H, W = left.shape[2:]
    fl = feature_extraction(sd, left, training)
    fr = feature_extraction(sd, right, training)
    return psmnet_from_features(sd, fl, fr, maxdisp, mindisp, H, W, training)


# -------------------------------------------------------------------------- a10
def psm_loss(output, target, mask):
    """PSMLoss.  Ref: utils/loss_utils.py:9-32 (== EndPointErrorLoss, stereo_utils.py:185-208)."""
    m = mask.to(target.dtype)
    msum = m.sum()
    if isinstance(output, (tuple, list)) and len(output) == 3:
        ls = []
        for o in output:
            d = (o - target).abs()
            sl1 = torch.where(d < 1.0, 0.5 * d * d, d - 0.5)
            l = (sl1 * m).sum()
            if msum != 0:
                l = l / msum
            ls.append(l)
        return 0.5 * ls[0] + 0.7 * ls[1] + ls[2]
    if msum == 0:
        return torch.zeros((), dtype=target.dtype)
    return ((output - target).abs() * m).sum() / msum


def to_dtype(sd, dtype):
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
