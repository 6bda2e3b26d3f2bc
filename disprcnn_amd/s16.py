"""Host side of the split-f16 ("f16x2") convolution path (csrc/convs16.hip, include/disprcnn_hip.h `drc_s16conv_params`).

An fp32 value v is carried as hi = fp16(v), lo = fp16(v - hi); tensors live in the RS16 layout
    halfs [N][C/32][D+2pd][H+2][8 chunks][W+2][8]        (zero halo stored; pd = 1 for volumes, 0 for 2D maps)
chunk q = p*4 + s*2 + g (p = 0 hi / 1 lo), element e of chunk (s, g) = channel 4g + 8(2s + (e>>2)) + (e&3) of the 32-channel block.
The torch converters below are the layout's definition for tests and weight packing; the product path converts with HIP kernels.
Reference arithmetic: disprcnn/modeling/psmnet/submodule.py:19-22 (convbn_3d), fp32.
"""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import DrcS16ConvParams

F16_MAX = 65504.0


def chunk_channels():
    """[32] long: position (slot = s*2+g, e) -> channel of the 32-channel block."""
    idx = torch.empty(4, 8, dtype=torch.long)
    for s in range(2):
        for g in range(2):
            for e in range(8):
                idx[s * 2 + g, e] = 4 * g + 8 * (2 * s + (e >> 2)) + (e & 3)
    return idx.view(-1)


def split(x):
    """fp32 -> (hi, lo) halfs with hi + lo ~= x to 2^-22 relative (|x| <= 65504)."""
    x = x.float().clamp(-F16_MAX, F16_MAX)
    hi = x.half()
    lo = (x - hi.float()).half()
    return hi, lo


def rs16_from_dense(x, pd=None):
    """x [N,C,D,H,W] (pd = 1) or [N,C,H,W] (pd = 0) fp32 -> RS16 half tensor [N][C/32][D+2pd][H+2][8][W+2][8]."""
    if x.dim() == 4:
        x = x.unsqueeze(2)
        pd = 0 if pd is None else pd
    pd = 1 if pd is None else pd
    N, Cc, D, H, W = x.shape
    assert Cc % 32 == 0
    cb = Cc // 32
    idx = chunk_channels().to(x.device)
    out = torch.zeros(N, cb, D + 2 * pd, H + 2, 8, W + 2, 8, dtype=torch.float16, device=x.device)
    for p_, part in enumerate(split(x)):
        t = part.view(N, cb, 32, D, H, W)[:, :, idx].view(N, cb, 4, 8, D, H, W).permute(0, 1, 4, 5, 2, 6, 3)
        out[:, :, pd:pd + D, 1:H + 1, 4 * p_:4 * p_ + 4, 1:W + 1] = t
    return out


def rs16_to_dense(t, pd=1):
    """RS16 -> fp32 [N,C,D,H,W] (interior)."""
    N, cb, Dp, Hp, _, Wp, _ = t.shape
    D, H, W = Dp - 2 * pd, Hp - 2, Wp - 2
    inv = torch.empty(32, dtype=torch.long)
    inv[chunk_channels()] = torch.arange(32)
    inv = inv.to(t.device)
    tt = t[:, :, pd:pd + D, 1:H + 1, :, 1:W + 1].float()
    v = tt[:, :, :, :, 0:4] + tt[:, :, :, :, 4:8]                      # [N,cb,D,H,4,W,8]
    v = v.permute(0, 1, 4, 6, 2, 3, 5).reshape(N, cb, 32, D, H, W)[:, :, inv]
    return v.reshape(N, cb * 32, D, H, W)


def pack_weight_s16(w):
    """[Cout,Cin,3,3,3] (or [Cout,Cin,3,3]: the 2D layers of convs16r.hip) fp32 -> (packed halfs [Cout/32][Cin/16][taps][2][64][8], wexp):
    the weights scaled by 2^wexp (largest magnitude in [2^13, 2^14): the lo parts stay normal fp16 numbers) and split; the caller folds
    2^-wexp into the epilogue scale."""
    w = w.detach().float()
    cout, cin = w.shape[:2]
    assert cout % 32 == 0 and cin % 16 == 0 and tuple(w.shape[2:]) in ((3, 3, 3), (3, 3))
    taps = 27 if w.dim() == 5 else 9
    amax = float(w.abs().max())
    wexp = int(math.floor(math.log2(16384.0 / amax))) if amax > 0 else 0
    wexp = max(min(wexp, 24), -24)
    hi, lo = split(w * (2.0 ** wexp))
    dev = w.device
    lane = torch.arange(64, device=dev)
    g = lane >> 5
    e = torch.arange(8, device=dev)
    ct_n, kw_n = cout // 32, cin // 16
    co = (torch.arange(ct_n, device=dev)[:, None] * 32 + (lane & 31)[None, :])                    # [ct, 64]
    kk = torch.arange(kw_n, device=dev)
    ci = ((kk >> 1)[:, None, None] * 32 + 4 * g[None, :, None] + 8 * (2 * (kk & 1)[:, None, None] + (e >> 2)[None, None, :]) + (e & 3)[None, None, :])   # [kw, 64, 8]
    out = torch.empty(ct_n, kw_n, taps, 2, 64, 8, dtype=torch.float16, device=dev)
    for p_, part in enumerate((hi, lo)):
        pw = part.reshape(cout, cin, taps)
        # [ct, kw, 64, 8, 27]
        sel = pw[co[:, None, :, None], ci[None, :, :, :]]
        out[:, :, :, p_] = sel.permute(0, 1, 4, 2, 3)
    return out.contiguous(), wexp


def head_rows():
    """[32] (kd, j) or None: which tap of the 32 -> 1 layer sits in MFMA row m of the fused head's A operand (convs16.hip, HEAD form).  The
    product's lane half g' holds rows (r&3) + 8(r>>2) + 4g' in register r; register 3jj + kd is tap (kd, j = 5g' + jj), so that the three depth
    taps of one in-plane tap sit in one lane: g' = 0 takes j 0..4 (15 rows), g' = 1 takes j 5..8 (12 rows), the other rows are zero."""
    rows = [None] * 32
    for g in range(2):
        for r in range(16):
            m = (r & 3) + 8 * (r >> 2) + 4 * g
            jj, kd = divmod(r, 3)
            if jj < (5 if g == 0 else 4):
                rows[m] = (kd, 5 * g + jj)
    return rows


def pack_head_weight_s16(w1):
    """[1,32,3,3,3] fp32 (classif[2], reference stackhourglass.py:78-88) -> (halfs [2 K slices][hi, lo][64 lanes][8], wexp): the A operand
    of the fused head, rows by head_rows(), k element e of lane group g of slice s = channel 4g + 8(2s + (e>>2)) + (e&3) (the RS16 chunk
    a finishing wave of the 32 -> 32 layer holds), scaled by 2^wexp like pack_weight_s16."""
    w1 = w1.detach().float().cpu()
    assert tuple(w1.shape) == (1, 32, 3, 3, 3)
    amax = float(w1.abs().max())
    wexp = int(math.floor(math.log2(16384.0 / amax))) if amax > 0 else 0
    wexp = max(min(wexp, 24), -24)
    ws = w1[0] * (2.0 ** wexp)                       # [32, 3, 3, 3]
    full = torch.zeros(2, 64, 8, dtype=torch.float32)
    rows = head_rows()
    for s_ in range(2):
        for lane in range(64):
            m, g = lane & 31, lane >> 5
            if rows[m] is None:
                continue
            kd, j = rows[m]
            for e in range(8):
                c = 4 * g + 8 * (2 * s_ + (e >> 2)) + (e & 3)
                full[s_, lane, e] = ws[c, kd, j // 3, j % 3]
    hi, lo = split(full)
    return torch.stack((hi, lo), 1).contiguous(), wexp            # [2][2][64][8]


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def conv3d_k3(x16, w_packed, scale, shift, D, H, W, cin, cout, relu, y16=None, y32=None, res=None, left=None, right=None, lo4=0):
    """One launch of drc_conv3d_k3_s16_fwd on RS16 tensors (torch half tensors) / blocked fp32 storage (y32: the flat fp32 storage)."""
    ref = x16 if x16 is not None else left
    N = ref.shape[0]
    p = DrcS16ConvParams(_p(x16), _p(w_packed), _p(scale), _p(shift), _p(res), _p(y16), _p(y32), _p(left), _p(right),
                         N, D, H, W, cin, cout, int(bool(relu)), int(lo4))
    st = _lib.lib().drc_conv3d_k3_s16_fwd(C.byref(p), C.c_void_p(torch.cuda.current_stream(ref.device).cuda_stream))
    _lib.check(st, "drc_conv3d_k3_s16_fwd")


def conv2d_k3(x16, w_packed, scale, shift, N, H, W, cin, cout, relu, y16, res=None, form=0, dil=1):
    """One launch of drc_conv2d_k3_s16_fwd on RS16 2D maps [N][C/32][H+2][8][W+2][8], given as torch half tensors or flat storages (form:
    0 = the dispatcher's choice for cin 64, 1 = one tile per workgroup, 2 = two tiles)."""
    p = DrcS16ConvParams(_p(x16), _p(w_packed), _p(scale), _p(shift), _p(res), _p(y16), _p(None), _p(None), _p(None),
                         N, 1, H, W, cin, cout, int(bool(relu)), int(form), int(dil))
    st = _lib.lib().drc_conv2d_k3_s16_fwd(C.byref(p), C.c_void_p(torch.cuda.current_stream(x16.device).cuda_stream))
    _lib.check(st, "drc_conv2d_k3_s16_fwd")
