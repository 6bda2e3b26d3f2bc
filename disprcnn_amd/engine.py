"""Host side of the HIP engine: blocked tensors, weight packing, tap lists, launch plans.

PyTorch is used for device memory, streams and parameter storage only; all arithmetic of
the hot path runs in libdisprcnn_hip.so through the C ABI (include/disprcnn_hip.h).
"""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import DrcTapconvParams

CB = 16
SLACK_FLOATS = 1 << 16          # over-read room behind every blocked tensor (ragged last row groups)
LDS_PER_WAVE_MAX = 38 * 1024    # 4 waves/block -> 152 KiB of the 160 KiB LDS
MAX_SLOTS = 112                 # 7 voxel tiles of 16
TIMING = None                   # bench.py sets this to a list to collect (kernel name, flops, start_evt, end_evt)
SLIDE = {"enabled": True, "max_slots": 112, "ct": 2, "min_od": 3, "min_share": 1, "min_units": 700, "min_units_ct1": 150}   # sliding-depth-window plans for stride-1 3x3x3 convs (tapdirect.hip and the Winograd kernels; thresholds measured with tools/experiments/exp_conv.py)


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: expected a CUDA/HIP tensor on an MI355X; the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{what}: fp32 only (reference: config/defaults.py:22 DTYPE float32), got {t.dtype}")


class Blocked:
    """Channel-blocked, zero-haloed tensor float[N][CB][D+2pd][H+2ph][W+2pw][16]."""

    def __init__(self, N, C_, D, H, W, pd, ph, pw, device, storage=None):
        self.N, self.C, self.D, self.H, self.W = N, C_, D, H, W
        self.pd, self.ph, self.pw = pd, ph, pw
        self.cb = (C_ + CB - 1) // CB
        self.Dp, self.Hp, self.Wp = D + 2 * pd, H + 2 * ph, W + 2 * pw
        self.h_stride = self.Wp * CB
        self.d_stride = self.Hp * self.h_stride
        self.cb_stride = self.Dp * self.d_stride
        self.n_stride = self.cb * self.cb_stride
        self.numel = N * self.n_stride
        if storage is None:
            self.storage = torch.zeros(self.numel + SLACK_FLOATS, dtype=torch.float32, device=device)
        else:
            # a prefix view of a larger zero-haloed tensor of the same per-unit geometry (WorkspacePool): the first N units
            if storage.numel() < self.numel + SLACK_FLOATS:
                raise ValueError("Blocked: the given storage is too small for this geometry")
            self.storage = storage.narrow(0, 0, self.numel + SLACK_FLOATS)
        self.device = device

    @classmethod
    def geometry(cls, N, C_, D, H, W, pd, ph, pw, device):
        """The shape / stride record of a blocked tensor WITHOUT storage: enough to plan a launch (plans take pointers per call)."""
        g = cls.__new__(cls)
        g.N, g.C, g.D, g.H, g.W, g.pd, g.ph, g.pw = N, C_, D, H, W, pd, ph, pw
        g.cb = (C_ + CB - 1) // CB
        g.Dp, g.Hp, g.Wp = D + 2 * pd, H + 2 * ph, W + 2 * pw
        g.h_stride = g.Wp * CB
        g.d_stride = g.Hp * g.h_stride
        g.cb_stride = g.Dp * g.d_stride
        g.n_stride = g.cb * g.cb_stride
        g.numel = N * g.n_stride
        g.storage, g.device = None, device
        return g

    @property
    def interior_off(self):
        return self.pd * self.d_stride + self.ph * self.h_stride + self.pw * CB

    def view6(self):
        return self.storage[: self.numel].view(self.N, self.cb, self.Dp, self.Hp, self.Wp, CB)

    def from_dense(self, dense):
        """dense [N,C,D,H,W] or [N,C,H,W] -> interior of this blocked tensor (HIP kernel)."""
        require_gpu(dense, "from_dense")
        dense = dense.contiguous()
        if self.N == 0:
            return self
        st = _lib.lib().drc_dense_to_blocked(_ptr(dense), _ptr(self.storage), self.N, self.C, self.D, self.H, self.W,
                                             self.pd, self.ph, self.pw, _stream_ptr(self.device))
        _lib.check(st, "drc_dense_to_blocked")
        return self

    def to_dense(self):
        shape = (self.N, self.C, self.D, self.H, self.W)
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        if self.N:
            st = _lib.lib().drc_blocked_to_dense(_ptr(self.storage), _ptr(out), self.N, self.C, self.D, self.H, self.W,
                                                 self.pd, self.ph, self.pw, _stream_ptr(self.device))
            _lib.check(st, "drc_blocked_to_dense")
        return out


def bucket_units(n):
    """Capacity bucket of a unit count: the smallest of {2^k, 3*2^(k-1)} >= n (at most 1/3 of a pool is idle)."""
    n = max(int(n), 1)
    k = 1
    while k < n:
        if k >= 2 and k + k // 2 >= n:
            return k + k // 2
        k *= 2
    return k


class WorkspacePool:
    """HBM behind the workspaces of ONE geometry (per-unit dims fixed; unit = ROI pair or image).

    Detection output has a different ROI count on every image (BASELINE configs[4]); a workspace per exact count would keep
    a full set of zero-haloed activations alive for every count ever seen.  The pool instead owns each named tensor ONCE,
    sized for `cap` units; the workspace of any N <= cap is a set of prefix views (units are outermost in the blocked layout,
    the halo is zeroed at allocation and never written, so the first N units of a bigger tensor are a valid tensor).  A
    count above `cap` replaces the pool by a bigger one (PSMNetRuntime).  `gen` counts the forward passes that used the
    pool: a backward whose forward's generation is no longer current would read overwritten activations and refuses."""

    def __init__(self, cap, device):
        self.cap, self.device = int(cap), device
        self.full = {}          # name -> Blocked sized for cap units
        self.flat = {}          # name -> dense [cap, ...] tensor
        self.gen = 0

    def blocked(self, name, N, C_, D, H, W, pd, ph, pw):
        if N > self.cap:
            raise ValueError("WorkspacePool: more units than the pool was sized for")
        full = self.full.get(name)
        if full is None:
            full = self.full[name] = Blocked(self.cap, C_, D, H, W, pd, ph, pw, self.device)
        elif (full.C, full.D, full.H, full.W, full.pd, full.ph, full.pw) != (C_, D, H, W, pd, ph, pw):
            raise ValueError(f"WorkspacePool: tensor {name!r} requested with another geometry")
        return Blocked(N, C_, D, H, W, pd, ph, pw, self.device, storage=full.storage)

    def blocked16(self, name, N, C_, D, H, W, pd, ph, pw):
        """fp16-storage tensor (Blocked16) of the pool."""
        if N > self.cap:
            raise ValueError("WorkspacePool: more units than the pool was sized for")
        full = self.full.get(name)
        if full is None:
            full = self.full[name] = Blocked16(self.cap, C_, D, H, W, pd, ph, pw, self.device)
        elif not isinstance(full, Blocked16) or (full.C, full.D, full.H, full.W, full.pd, full.ph, full.pw) != (C_, D, H, W, pd, ph, pw):
            raise ValueError(f"WorkspacePool: tensor {name!r} requested with another geometry")
        return Blocked16(N, C_, D, H, W, pd, ph, pw, self.device, storage=full.storage)

    def rs16(self, name, N, C_, D, H, W, pd=1):
        """split-f16 tensor (RS16) of the pool."""
        if N > self.cap:
            raise ValueError("WorkspacePool: more units than the pool was sized for")
        full = self.full.get(name)
        if full is None:
            full = self.full[name] = RS16(self.cap, C_, D, H, W, pd, self.device)
        elif not isinstance(full, RS16) or (full.C, full.D, full.H, full.W, full.pd) != (C_, D, H, W, pd):
            raise ValueError(f"WorkspacePool: tensor {name!r} requested with another geometry")
        return RS16(N, C_, D, H, W, pd, self.device, storage=full.storage)

    def dense(self, name, N, *shape):
        full = self.flat.get(name)
        if full is None:
            full = self.flat[name] = torch.empty(self.cap, *shape, dtype=torch.float32, device=self.device)
        elif tuple(full.shape[1:]) != tuple(shape):
            raise ValueError(f"WorkspacePool: tensor {name!r} requested with another shape")
        return full[:N]

    def nbytes(self):
        return sum(b.storage.numel() * b.storage.element_size() for b in self.full.values()) + 4 * sum(t.numel() for t in self.flat.values())


class BlockedSlice:
    """A channel-block slice [cb_off, cb_off+cb) of a Blocked tensor (concat without copies)."""

    def __init__(self, base, cb_off, channels):
        self.base, self.cb_off, self.C = base, cb_off, channels
        self.cb = (channels + CB - 1) // CB
        for k in ("N", "D", "H", "W", "pd", "ph", "pw", "Dp", "Hp", "Wp", "h_stride", "d_stride", "cb_stride", "n_stride",
                  "device", "storage", "interior_off"):
            setattr(self, k, getattr(base, k))

    @property
    def ptr_off(self):
        return self.cb_off * self.cb_stride


def _base_ptr(t):
    off = getattr(t, "ptr_off", 0)
    return t.storage.data_ptr() + 4 * off


# ------------------------------------------------------------------------------------------- weights
def pack_layouts(w, transposed=False, flip=False, want_tap=True, want_t16=True):
    """Both packings of a conv weight in ONE launch (drc_pack_weights): (tap [K][cb][2][cout_pad][8], t16 [K][cb][cout_pad][16]).
    w: [Cout,Cin,*k] or ConvTranspose [Cin,Cout,*k] on the GPU; flip reverses the taps (data gradients)."""
    w = w.detach().contiguous().float()
    a, b = w.shape[:2]
    cout, cin = (b, a) if transposed else (a, b)
    K = int(math.prod(w.shape[2:]))
    cb = (cin + CB - 1) // CB
    cout_pad = (cout + CB - 1) // CB * CB
    tap = torch.empty(K, cb, 2, cout_pad, 8, dtype=torch.float32, device=w.device) if want_tap else None
    t16 = torch.empty(K, cb, cout_pad, 16, dtype=torch.float32, device=w.device) if want_t16 else None
    st = _lib.lib().drc_pack_weights(_ptr(w), cout, cin, K, int(transposed), int(flip), _ptr(tap), _ptr(t16), _stream_ptr(w.device))
    _lib.check(st, "drc_pack_weights")
    return tap, t16


def pack_weight(w, transposed=False):
    """[Cout,Cin,*k] (or ConvTranspose [Cin,Cout,*k]) -> [K][cb_in][2 halves][cout_pad][8] fp32 contiguous
    (shape[3] is cout_pad; channel c of block cb sits in half c//8, slot c%8)."""
    if w.is_cuda:
        return pack_layouts(w, transposed, want_t16=False)[0]
    if transposed:
        w = w.transpose(0, 1)
    cout, cin = w.shape[:2]
    K = int(math.prod(w.shape[2:]))
    cb = (cin + CB - 1) // CB
    cout_pad = (cout + CB - 1) // CB * CB
    wp = torch.zeros(K, cb * CB, cout_pad, dtype=torch.float32, device=w.device)
    wp[:, :cin, :cout] = w.reshape(cout, cin, K).permute(2, 1, 0)
    # [K][cb][half][8][cout] -> [K][cb][half][cout][8]
    return wp.view(K, cb, 2, 8, cout_pad).permute(0, 1, 2, 4, 3).contiguous()


POINTWISE = {"enabled": True}      # 1x1 Conv2d through the LDS-free GEMM kernel (pointwise.hip)


def is_pointwise(weight_shape, transposed=False):
    """True for the weights of a 1x1 Conv2d, which the pointwise kernel takes in its own packing."""
    return POINTWISE["enabled"] and len(weight_shape) == 4 and tuple(weight_shape[2:]) == (1, 1) and not transposed


def pack_weight_pw(w):
    """[Cout,Cin,1,1] -> [cb_in][cout_pad][16] fp32 contiguous: lane (cout, g) of the MFMA A operand reads channels 4g..4g+3
    of a 16-channel block as one float4 (1 KiB per 16 couts)."""
    if w.is_cuda:
        return pack_layouts(w, want_tap=False)[1][0]
    cout, cin = w.shape[:2]
    cb = (cin + CB - 1) // CB
    cout_pad = (cout + CB - 1) // CB * CB
    wp = torch.zeros(cb * CB, cout_pad, dtype=torch.float32, device=w.device)
    wp[:cin, :cout] = w.reshape(cout, cin).t()
    return wp.view(cb, CB, cout_pad).permute(0, 2, 1).contiguous()


def pack_weight_t16(w, transposed=False):
    """[Cout,Cin,*k] (or ConvTranspose [Cin,Cout,*k]) -> [K taps][cb_in][cout_pad][16] fp32 contiguous: the packing of the
    LDS-free kernels, whose lane (cout, g) reads channels 4g..4g+3 of a 16-channel block as one float4."""
    if w.is_cuda:
        return pack_layouts(w, transposed, want_tap=False)[1]
    if transposed:
        w = w.transpose(0, 1)
    cout, cin = w.shape[:2]
    K = int(math.prod(w.shape[2:]))
    cb = (cin + CB - 1) // CB
    cout_pad = (cout + CB - 1) // CB * CB
    wp = torch.zeros(K, cb * CB, cout_pad, dtype=torch.float32, device=w.device)
    wp[:, :cin, :cout] = w.reshape(cout, cin, K).permute(2, 1, 0)
    return wp.view(K, cb, CB, cout_pad).permute(0, 1, 3, 2).contiguous()


def pack_weight_wino(w, transposed=False, flip=False):
    """3x3x3 weights [Cout,Cin,3,3,3] (or [Cin,Cout,...] with transposed; flip reverses the taps) -> Winograd F(2,3)^3
    transformed [64 frequency points][cb_in][cout_pad][16] (drc_pack_weights_wino): the packing of wino3d.hip."""
    w = w.detach().contiguous().float()
    require_gpu(w, "pack_weight_wino")
    if w.dim() != 5 or tuple(w.shape[2:]) != (3, 3, 3):
        raise ValueError("pack_weight_wino expects a 3x3x3 kernel")
    a, b = w.shape[:2]
    cout, cin = (b, a) if transposed else (a, b)
    out = torch.empty(64, (cin + CB - 1) // CB, cout_pad_of(cout), 16, dtype=torch.float32, device=w.device)
    st = _lib.lib().drc_pack_weights_wino(_ptr(w), cout, cin, int(transposed), int(flip), _ptr(out), _stream_ptr(w.device))
    _lib.check(st, "drc_pack_weights_wino")
    return out


def pack_weight_wino_rb(w, transposed=False, flip=False):
    """The same transformed weights in the packing of wino3d_rb.hip: [64][cb_in][cout tile][ch / 4][cout % 16][ch % 4] (cout padded
    to 32) -- a (frequency point, channel block, cout tile) unit is the 1 KiB an LDS-DMA instruction copies in lane order."""
    w = w.detach().contiguous().float()
    require_gpu(w, "pack_weight_wino_rb")
    if w.dim() != 5 or tuple(w.shape[2:]) != (3, 3, 3):
        raise ValueError("pack_weight_wino_rb expects a 3x3x3 kernel")
    a, b = w.shape[:2]
    cout, cin = (b, a) if transposed else (a, b)
    out = torch.empty(64, (cin + CB - 1) // CB, (cout + 31) // 32 * 32, 16, dtype=torch.float32, device=w.device)
    st = _lib.lib().drc_pack_weights_wino_rb(_ptr(w), cout, cin, int(transposed), int(flip), _ptr(out), _stream_ptr(w.device))
    _lib.check(st, "drc_pack_weights_wino_rb")
    return out


def pack_weight_wino2d_rb(w, transposed=False, flip=False):
    """The 2D transformed weights in the packing of wino3d_rb.hip's 2D form: [16][cb_in][cout tile][ch / 4][cout % 16][ch % 4]."""
    w = w.detach().contiguous().float()
    require_gpu(w, "pack_weight_wino2d_rb")
    if w.dim() != 4 or tuple(w.shape[2:]) != (3, 3):
        raise ValueError("pack_weight_wino2d_rb expects a 3x3 kernel")
    a, b = w.shape[:2]
    cout, cin = (b, a) if transposed else (a, b)
    out = torch.empty(16, (cin + CB - 1) // CB, (cout + 31) // 32 * 32, 16, dtype=torch.float32, device=w.device)
    st = _lib.lib().drc_pack_weights_wino2d_rb(_ptr(w), cout, cin, int(transposed), int(flip), _ptr(out), _stream_ptr(w.device))
    _lib.check(st, "drc_pack_weights_wino2d_rb")
    return out


def pack_weight_wino2d(w, transposed=False, flip=False):
    """3x3 weights [Cout,Cin,3,3] (or [Cin,Cout,3,3] with transposed; flip reverses the taps) -> Winograd F(2,3)^2 transformed
    [16 frequency points][cb_in][cout_pad][16] (drc_pack_weights_wino2d): the packing of wino2d.hip."""
    w = w.detach().contiguous().float()
    require_gpu(w, "pack_weight_wino2d")
    if w.dim() != 4 or tuple(w.shape[2:]) != (3, 3):
        raise ValueError("pack_weight_wino2d expects a 3x3 kernel")
    a, b = w.shape[:2]
    cout, cin = (b, a) if transposed else (a, b)
    out = torch.empty(16, (cin + CB - 1) // CB, cout_pad_of(cout), 16, dtype=torch.float32, device=w.device)
    st = _lib.lib().drc_pack_weights_wino2d(_ptr(w), cout, cin, int(transposed), int(flip), _ptr(out), _stream_ptr(w.device))
    _lib.check(st, "drc_pack_weights_wino2d")
    return out


def _deconv_tap_order():
    """deconvdirect.hip's use order of the 27 taps: class-major (output-parity class 7 first ... class 0 last); within a class the
    (parity, shift, k) combinations (0,0,1), (1,0,2), (1,1,0) per dimension in lexicographic order."""
    P, K = (0, 1, 1), (1, 2, 0)
    raw = [((P[a] * 2 + P[b]) * 2 + P[c], (K[a] * 3 + K[b]) * 3 + K[c]) for a in range(3) for b in range(3) for c in range(3)]
    return [tap for cls in range(7, -1, -1) for (c_, tap) in raw if c_ == cls]


_DECONV_TAP_ORDER = _deconv_tap_order()
_DECONV_TAP_ORDER_DEV = {}     # per device: a host->device copy is not allowed while a HIP graph is being captured


def pack_weight_deconv_direct(w, transposed=True, flip=False):
    """ConvTranspose3d weight [Cin,Cout,3,3,3] (or, for the data gradient of a stride-2 Conv3d, its weight read as one) ->
    [cb_in][27][cout_pad][16]: the t16 packing re-ordered channel-block-major with the taps in deconvdirect.hip's use order."""
    t16 = pack_layouts(w, transposed, flip, want_tap=False)[1]               # [27][cb][cout_pad][16]
    order = _DECONV_TAP_ORDER_DEV.get(t16.device)
    if order is None:
        order = _DECONV_TAP_ORDER_DEV[t16.device] = torch.tensor(_DECONV_TAP_ORDER, device=t16.device)
    return t16.index_select(0, order).permute(1, 0, 2, 3).contiguous()


SPP_NESTED = {"enabled": True}       # eval: the coarser SPP pools are pooled from the finest pool's cells (one pass over the 128-channel map instead of four)
STEM_DIRECT = {"enabled": True}      # eval: firstconv[0] straight from the dense image (stemconv.hip, round 4) instead of layout conversion + downdirect


def pack_weight_stem(w):
    """Conv2d(3 -> cout, 3x3) weight [cout,3,3,3] -> [7][cout_pad][4]: k = channel * 9 + tap in steps of four, zero past k = 26 (stemconv.hip)."""
    cout = w.shape[0]
    if tuple(w.shape[1:]) != (3, 3, 3):
        raise ValueError("pack_weight_stem expects a [cout,3,3,3] weight")
    cp = cout_pad_of(cout)
    wp = torch.zeros(cp, 28, dtype=torch.float32, device=w.device)
    wp[:cout, :27] = w.detach().float().reshape(cout, 27)
    return wp.view(cp, 7, 4).permute(1, 0, 2).contiguous()


def stem_conv(images, w_packed, scale, shift, y, relu=True, unit0=0):
    """images fp32 NCHW [n,3,H,W] (dense) -> units unit0 .. unit0+n of y: Blocked [N,cout,1,(H+1)//2,(W+1)//2]: Conv2d k3 s2 p1 + folded BN
    (+ReLU), reference submodule.py:65-66.  No layout conversion of the image (and, with unit0, no concatenation of the two views)."""
    require_gpu(images, "stem_conv")
    if images.dim() != 4 or images.shape[1] != 3 or images.dtype != torch.float32:
        raise ValueError("stem_conv expects a float32 [N,3,H,W] image batch")
    n, _, H, W = images.shape
    if (y.D, y.H, y.W) != (1, (H + 1) // 2, (W + 1) // 2) or unit0 < 0 or unit0 + n > y.N:
        raise ValueError("stem_conv: output geometry does not match a 3x3 stride-2 pad-1 convolution of the input")
    cout_pad = w_packed.shape[1]
    if not isinstance(y, Blocked) or y.storage.dtype != torch.float32 or y.cb * CB < cout_pad:
        raise ValueError("stem_conv: y must be a blocked fp32 tensor with at least the packed weights' output channels")
    if scale.numel() < cout_pad or shift.numel() < cout_pad or scale.dtype != torch.float32 or shift.dtype != torch.float32:
        raise ValueError("stem_conv: scale / shift must be fp32 vectors of at least cout_pad entries")
    x = images.contiguous()
    if TIMING is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(y.device))
    st = _lib.lib().drc_conv2d_k3s2_stem_fwd(_ptr(x), n, H, W, _ptr(w_packed), w_packed.shape[1], _ptr(scale), _ptr(shift), _ptr(y.storage),
                                             y.n_stride, y.cb_stride, y.h_stride, y.interior_off + unit0 * y.n_stride, y.H, y.W, int(relu),
                                             _stream_ptr(y.device))
    _lib.check(st, "drc_conv2d_k3s2_stem_fwd")
    if TIMING is not None:
        e1.record(torch.cuda.current_stream(y.device))
        TIMING.append(("stemconv_kernel", 2 * n * y.H * y.W * 27 * cout_pad, e0, e1))


def pack_conv_weight(w, transposed=False):
    """The packing the engine's plan for this convolution expects (pointwise for 1x1 Conv2d, tap layout otherwise)."""
    return pack_weight_pw(w) if is_pointwise(w.shape, transposed) else pack_weight(w, transposed)


def cout_pad_of(cout):
    return (cout + CB - 1) // CB * CB


def pack_weight_cout1(w):
    """Conv3d(Cin->1,k3) weight [1,Cin,3,3,3] -> [27][cb*16]."""
    cin = w.shape[1]
    cb = (cin + CB - 1) // CB
    wp = torch.zeros(27, cb * CB, dtype=torch.float32, device=w.device)
    wp[:, :cin] = w.reshape(cin, 27).t()
    return wp.contiguous()


def fold_bn(gamma, beta, mean, var, eps=1e-5, cout_pad=None):
    """Eval-mode BatchNorm folded to y = scale*conv + shift (reference: submodule.py:19-22 conv->BN)."""
    scale = gamma / torch.sqrt(var + eps)
    shift = beta - mean * scale
    if cout_pad is not None and cout_pad != scale.numel():
        s = torch.ones(cout_pad, dtype=torch.float32, device=scale.device)
        b = torch.zeros(cout_pad, dtype=torch.float32, device=scale.device)
        s[: scale.numel()] = scale
        b[: shift.numel()] = shift
        scale, shift = s, b
    return scale.contiguous().float(), shift.contiguous().float()


# ------------------------------------------------------------------------------------------- tap grids
def taps_conv(kdims, dilation, pad_conv, pad_in):
    """Cross-correlation tap grid of a (kd,kh,kw) kernel: offset = k*dil - pad_conv + pad_in (padded input coords)."""
    kd, kh, kw = kdims
    first = tuple(pad_in[i] - pad_conv[i] for i in range(3))
    if min(first) < 0:
        raise ValueError("input halo too small for this convolution")
    return [dict(n=(kd, kh, kw), first=first, step=tuple(dilation), wbase=0, wstep=(kh * kw, kw, 1), off=(0, 0, 0))]


def taps_deconv3d_k3s2(pad_in=(1, 1, 1)):
    """ConvTranspose3d(k3,s2,p1,op1) as 8 output-parity classes (stackhourglass.py:22-30).
    o = 2i - 1 + k  =>  even o=2j: (i=j,k=1);  odd o=2j+1: (i=j,k=2), (i=j+1,k=0):
    per dim an even class has one tap (offset 0, k=1), an odd class two taps (offsets 0,1 with k=2,0)."""
    classes = []
    kstride = (9, 3, 1)
    for pd_ in (0, 1):
        for ph_ in (0, 1):
            for pw_ in (0, 1):
                par = (pd_, ph_, pw_)
                n = tuple(2 if q else 1 for q in par)
                wbase = sum((2 if q else 1) * ks for q, ks in zip(par, kstride))
                wstep = tuple(-2 * ks if q else 0 for q, ks in zip(par, kstride))
                classes.append(dict(n=n, first=tuple(pad_in), step=(1, 1, 1), wbase=wbase, wstep=wstep, off=par))
    return classes


def taps_deconv2d_s2(k, pad_in=(0, 1, 1)):
    """2D transposed convolution, stride 2, output exactly twice the input -- the data gradient of Conv2d(k, stride 2,
    pad (k-1)//2) on an even-sized map.  k=3: four output-parity classes (same 1-D pattern as the 3D case);
    k=1: only the even positions receive a tap (the caller zero-fills the others)."""
    if k == 1:
        return [dict(n=(1, 1, 1), first=tuple(pad_in), step=(1, 1, 1), wbase=0, wstep=(0, 0, 0), off=(0, 0, 0))]
    assert k == 3
    classes = []
    for ph_ in (0, 1):
        for pw_ in (0, 1):
            par = (ph_, pw_)
            n = (1,) + tuple(2 if q else 1 for q in par)
            wbase = sum((2 if q else 1) * ks for q, ks in zip(par, (3, 1)))
            wstep = (0,) + tuple(-2 * ks if q else 0 for q, ks in zip(par, (3, 1)))
            classes.append(dict(n=n, first=tuple(pad_in), step=(1, 1, 1), wbase=wbase, wstep=wstep, off=(0,) + par))
    return classes


def class_taps(c):
    """Enumerate (dd, dh, dw, widx) of a tap-grid class (host-side mirror of the kernel's arithmetic)."""
    out = []
    for a in range(c["n"][0]):
        for b in range(c["n"][1]):
            for d in range(c["n"][2]):
                out.append((c["first"][0] + a * c["step"][0], c["first"][1] + b * c["step"][1], c["first"][2] + d * c["step"][2],
                            c["wbase"] + a * c["wstep"][0] + b * c["wstep"][1] + d * c["wstep"][2]))
    return out


def choose_tile(OH, OW, in_mul, span_h, span_w, max_slots=MAX_SLOTS):
    """Pick (R, WT): rows x cols of output per wave.  Maximise useful MFMA slots under the LDS budget."""
    best = None
    wts = {-(-OW // parts) for parts in range(1, OW + 1) if -(-OW // parts) <= max_slots}
    for wt in sorted(wts):
        for r in range(1, min(OH, max_slots // wt) + 1):
            rows_in = in_mul * (r - 1) + span_h + 1
            seg = in_mul * (wt - 1) + span_w + 1
            lds = 2 * rows_in * seg * 32      # double-buffered [rows][voxels][8 ch] fp32 tile per wave
            if lds > LDS_PER_WAVE_MAX:
                continue
            n_rt, n_wt = -(-OH // r), -(-OW // wt)
            nvt = -(-(r * wt) // 16)
            useful = OH * OW
            issued = n_rt * n_wt * nvt * 16
            eff = useful / issued
            key = (round(eff, 4), r * wt, wt, -lds)       # ties: more slots, then WIDE tiles (long contiguous rows)
            if best is None or key > best[0]:
                best = (key, r, wt, lds)
    if best is None:
        raise ValueError("no tile fits the LDS budget")
    return best[1], best[2], best[3]


class ConvPlan:
    """A fully resolved tapconv launch: geometry, tap classes, tile choice.  Pointers are patched per call."""

    def __init__(self, x, y, classes, in_mul, out_mul, grid_dhw, cout, relu, slide=False):
        p = DrcTapconvParams()
        self.slide = slide
        self.fused_deconv = False
        self.down = False
        self.pointwise = False
        self.direct = False
        self.wino = False
        self.rb = False            # wino3d_rb.hip (two waves per SIMD, row brick) instead of wino3d.hip
        self.c2d = False
        self.deconv_direct = False
        OD, OH, OW = grid_dhw
        p.x_n_stride, p.x_cb_stride, p.x_d_stride, p.x_h_stride = x.n_stride, x.cb_stride, x.d_stride, x.h_stride
        p.y_n_stride, p.y_cb_stride, p.y_d_stride, p.y_h_stride = y.n_stride, y.cb_stride, y.d_stride, y.h_stride
        p.y_off0 = y.interior_off
        p.N, p.OD, p.OH, p.OW = x.N, OD, OH, OW
        p.in_mul, p.out_mul = in_mul, out_mul
        p.cb_in = x.cb
        p.cout_pad = (cout + CB - 1) // CB * CB
        p.relu = int(relu)
        p.n_classes = len(classes)
        span_h = max((c["n"][1] - 1) * c["step"][1] for c in classes)
        span_w = max((c["n"][2] - 1) * c["step"][2] for c in classes)
        R, WT, lds = choose_tile(OH, OW, in_mul, span_h, span_w, SLIDE["max_slots"] if slide else MAX_SLOTS)
        p.R, p.WT = R, WT
        p.lds_bytes_per_wave = (lds + 1023) // 1024 * 1024
        for ci, c in enumerate(classes):
            k = p.cls[ci]
            k.nd, k.nh, k.nw = c["n"]
            k.dd0, k.dh0, k.dw0 = c["first"]
            k.sd, k.sh, k.sw = c["step"]
            k.wbase = c["wbase"]
            k.wsd, k.wsh, k.wsw = c["wstep"]
            k.out_off_d, k.out_off_h, k.out_off_w = c["off"]
        self.p = p
        self.device = x.device
        ntaps = sum(c["n"][0] * c["n"][1] * c["n"][2] for c in classes)
        self.flops = 2 * x.N * OD * OH * OW * ntaps * x.C * cout       # algorithmic (unpadded channels, valid voxels)
        nvt = -(-(R * WT) // 16)
        ct = p.cout_pad // 16
        groups = x.N * OD * (-(-OH // R)) * (-(-OW // WT)) * len(classes)
        CT = 4 if ct % 4 == 0 else (2 if ct % 2 == 0 else 1)
        while CT > 1 and (groups * (ct // CT) < 2048 or nvt * CT > 16):      # mirrors drc_tapconv_fwd's choice
            CT //= 2
        self.kname = self._generic_kname = "tapconv_kernel<%d,%d>" % (nvt, CT)
        if slide:
            # the sliding-window kernel gives every resident wave an equal share (>= 3 output slices) of the
            # (cout group, column, slice) units: use it only when that still yields enough waves to fill the 1024 SIMDs
            # (measured cross-over, tools/experiments/exp_conv.py)
            self.slide_ct = SLIDE["ct"] if ct % SLIDE["ct"] == 0 else 1
            cols = x.N * (-(-OH // R)) * (-(-OW // WT))
            # small batches: with ONE cout tile per wave the LDS-free direct kernel still beats the generic kernel down to ~150 units
            # (3x7x7 maps, 64->64: 47 vs 87 us at 64 ROIs, 45 vs 84 us at 16; at 128 ROIs two tiles per wave win, 75 vs 93 us)
            self.slide_small_ok = OD >= SLIDE["min_od"] and cols * ct * OD >= SLIDE["min_units_ct1"]
            if OD < SLIDE["min_od"] or cols * (ct // self.slide_ct) * OD // SLIDE["min_share"] < SLIDE["min_units"]:
                self.slide = False
            else:
                self.kname = "tapslide_kernel<%d,%d>" % (nvt, self.slide_ct)

    @property
    def needs_t16(self):
        """True when this plan's kernel reads the [tap][cb][cout][16] packing (the LDS-free kernels) instead of the tap layout."""
        return bool(self.direct and (self.slide or self.down or self.c2d or self.deconv_direct))

    @property
    def pack_kind(self):
        """Name of the weight packing pack16 returns (a layer may run under plans of several kinds: one packing is kept per kind)."""
        if self.wino:
            return ("wino2d_rb" if self.rb else "wino2d") if self.c2d else ("wino_rb" if self.rb else "wino")
        return "deconv_direct" if self.deconv_direct else ("t16" if self.needs_t16 else "tap")

    def pack16(self, w, transposed=False, flip=False, kind=None):
        """The weight packing this plan's LDS-free kernel reads (None when the plan runs an LDS-staged kernel)."""
        if kind == "wino":
            return pack_weight_wino(w, transposed, flip)
        if self.wino and self.rb:
            return pack_weight_wino2d_rb(w, transposed, flip) if self.c2d else pack_weight_wino_rb(w, transposed, flip)
        if self.wino:
            return pack_weight_wino2d(w, transposed, flip) if self.c2d else pack_weight_wino(w, transposed, flip)
        if self.deconv_direct:
            return pack_weight_deconv_direct(w, transposed, flip)
        if self.needs_t16:
            return pack_layouts(w, transposed, flip, want_tap=False)[1]
        return None

    def run(self, x, w, scale, shift, y, res=None, relu=None, w16=None, y16=None):
        """y16 (an RS16 tensor; transposed-conv plans on the LDS-free kernel only): the result is (also) written in the split-f16
        layout for a drc_conv3d_k3_s16_fwd consumer; y may then be None."""
        p = self.p
        if y16 is not None and not self.deconv_direct:
            raise ValueError("y16: only the LDS-free transposed convolution writes RS16")
        if self.needs_t16:
            points = (16 if self.c2d else 64) if self.wino else (9 if self.c2d else 27)
            if w16 is None or w16.shape[1 if self.deconv_direct else 0] != points:
                raise ValueError("this plan runs an LDS-free kernel: pass w16 = plan.pack16(weight)")
            w = w16
        relu_saved = p.relu
        if relu is not None:
            p.relu = int(relu)
        p.x, p.y = _base_ptr(x), (_base_ptr(y) if y is not None else None)
        # strides come from the tensors actually passed (same logical shape as at plan time; e.g. a train-mode raw buffer
        # instead of a concat slice)
        p.x_n_stride, p.x_cb_stride, p.x_d_stride, p.x_h_stride = x.n_stride, x.cb_stride, x.d_stride, x.h_stride
        if y is not None:
            p.y_n_stride, p.y_cb_stride, p.y_d_stride, p.y_h_stride = y.n_stride, y.cb_stride, y.d_stride, y.h_stride
            p.y_off0 = y.interior_off
        p.w, p.scale, p.shift = w.data_ptr(), scale.data_ptr(), shift.data_ptr()
        if res is not None:
            p.res = _base_ptr(res)
            p.r_n_stride, p.r_cb_stride, p.r_d_stride, p.r_h_stride = res.n_stride, res.cb_stride, res.d_stride, res.h_stride
            p.r_off0 = res.interior_off
        else:
            p.res = None
        if TIMING is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        if (w.dim() == 3) != self.pointwise:
            raise ValueError("weights are not in the packing this plan expects (engine.pack_conv_weight)")
        if self.wino and self.c2d and self.rb:
            st = _lib.lib().drc_conv2d_k3_wino_rb_fwd(C.byref(p), _stream_ptr(self.device))
            _lib.check(st, "drc_conv2d_k3_wino_rb_fwd")
        elif self.wino and self.c2d:
            st = _lib.lib().drc_conv2d_k3_wino_fwd(C.byref(p), self.slide_ct, _stream_ptr(self.device))
            _lib.check(st, "drc_conv2d_k3_wino_fwd")
        elif self.direct and self.c2d:
            st = _lib.lib().drc_conv2d_k3_direct_fwd(C.byref(p), self.c2d_ct, _stream_ptr(self.device))
            _lib.check(st, "drc_conv2d_k3_direct_fwd")
        elif self.direct and self.down:
            st = _lib.lib().drc_conv3d_k3s2_direct_fwd(C.byref(p), self.down_ct, _stream_ptr(self.device))
            _lib.check(st, "drc_conv3d_k3s2_direct_fwd")
        elif self.wino and self.rb:
            st = _lib.lib().drc_conv3d_k3_wino_rb_fwd(C.byref(p), _stream_ptr(self.device))
            _lib.check(st, "drc_conv3d_k3_wino_rb_fwd")
        elif self.wino:
            st = _lib.lib().drc_conv3d_k3_wino_fwd(C.byref(p), self.slide_ct, _stream_ptr(self.device))
            _lib.check(st, "drc_conv3d_k3_wino_fwd")
        elif self.direct and self.slide:
            st = _lib.lib().drc_tapconv3d_direct_fwd(C.byref(p), self.slide_ct, _stream_ptr(self.device))
            _lib.check(st, "drc_tapconv3d_direct_fwd")
        elif self.pointwise:
            st = _lib.lib().drc_conv2d_k1_fwd(C.byref(p), _stream_ptr(self.device))
            _lib.check(st, "drc_conv2d_k1_fwd")
        elif self.deconv_direct and y16 is not None:
            st = _lib.lib().drc_deconv3d_k3s2_direct_s16_fwd(C.byref(p), _ptr(y16.storage), _ovf_ptr(), _stream_ptr(self.device))
            _lib.check(st, "drc_deconv3d_k3s2_direct_s16_fwd")
        elif self.deconv_direct:
            st = _lib.lib().drc_deconv3d_k3s2_direct_fwd(C.byref(p), self.deconv_ct, _stream_ptr(self.device))
            _lib.check(st, "drc_deconv3d_k3s2_direct_fwd")
        elif self.fused_deconv:
            st = _lib.lib().drc_deconv3d_k3s2_fwd(C.byref(p), _stream_ptr(self.device))
            _lib.check(st, "drc_deconv3d_k3s2_fwd")
        else:
            st = _lib.lib().drc_tapconv_fwd(C.byref(p), _stream_ptr(self.device))
            _lib.check(st, "drc_tapconv_fwd")
        p.relu = relu_saved
        if TIMING is not None:
            e1.record(torch.cuda.current_stream(self.device))
            TIMING.append((self.kname, self.flops, e0, e1))


    def run_costvol(self, left, right, lo4, w16, scale, shift, y):
        """dres0[0] straight from the feature maps (the cost volume is never materialised): `left`, `right` are blocked 2D feature
        tensors (D = 1, no depth halo, halo >= 1 in y and x) of the same geometry holding this plan's N maps each -- `right` may
        be a (Blocked, first_unit) pair naming a later range of the same tensor.  Only Winograd plans (plan.wino) have this form.
        Reference: stackhourglass.py:115-130."""
        if not (self.wino and not self.c2d):
            raise ValueError("run_costvol needs a 3D Winograd plan")
        r_first = 0
        if isinstance(right, tuple):
            right, r_first = right
        p = self.p
        for f in (left, right):
            if f.D != 1 or f.pd != 0 or f.ph < 1 or f.ph != f.pw or f.W != p.OW or f.H != p.OH or 2 * f.cb != p.cb_in:
                raise ValueError("run_costvol: feature maps do not match the plan's volume")
        if (right.n_stride, right.cb_stride, right.h_stride, right.ph) != (left.n_stride, left.cb_stride, left.h_stride, left.ph):
            raise ValueError("run_costvol: left and right feature maps differ in geometry")
        if left.N < p.N or right.N < r_first + p.N:
            raise ValueError("run_costvol: fewer feature maps than volume units")
        if w16 is None or w16.shape[0] != 64:
            raise ValueError("run_costvol: pass w16 = plan.pack16(weight) (the plan's own packing: wino3d.hip's or wino3d_rb.hip's)")
        cv = _lib.DrcCostvolSrc()
        cv.left = _base_ptr(left)
        cv.right = _base_ptr(right) + 4 * r_first * right.n_stride
        cv.n_stride, cv.cb_stride, cv.h_stride = left.n_stride, left.cb_stride, left.h_stride
        cv.cbi, cv.pad, cv.lo4, cv.Wp = left.cb, left.ph, int(lo4), left.W
        p.x = None
        p.y = _base_ptr(y)
        p.y_n_stride, p.y_cb_stride, p.y_d_stride, p.y_h_stride = y.n_stride, y.cb_stride, y.d_stride, y.h_stride
        p.y_off0 = y.interior_off
        p.w, p.scale, p.shift = w16.data_ptr(), scale.data_ptr(), shift.data_ptr()
        p.res = None
        if TIMING is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        if self.rb:
            st = _lib.lib().drc_conv3d_k3_wino_rb_costvol_fwd(C.byref(p), C.byref(cv), _stream_ptr(self.device))
            _lib.check(st, "drc_conv3d_k3_wino_rb_costvol_fwd")
        else:
            st = _lib.lib().drc_conv3d_k3_wino_costvol_fwd(C.byref(p), C.byref(cv), self.slide_ct, _stream_ptr(self.device))
            _lib.check(st, "drc_conv3d_k3_wino_costvol_fwd")
        if TIMING is not None:
            e1.record(torch.cuda.current_stream(self.device))
            TIMING.append((self.kname.replace("rb_kernel", "rb_cv_kernel") if self.rb else "wino3d_cv_kernel<%d>" % self.slide_ct, self.flops, e0, e1))


# Kernel-selection switches (defaults = the fastest measured path; the tests flip them to keep every variant covered)
WINO = {"enabled": True,      # stride-1 3x3x3 layers with even output dims as Winograd F(2x2x2,3x3x3) (wino3d.hip)
        "small": 48,          # ... also below the sliding kernel's unit threshold, from this many 64-tile chunks (0: never); fewer: tapdirect<.,1>
        "fuse_costvol": True, # eval: dres0[0] reads the feature maps directly, the cost volume is never written (wino3d_cv_kernel)
        "rb": True,           # 28- and 14-wide maps: the two-waves-per-SIMD row-brick kernel (wino3d_rb.hip) ...
        "rb_min_chunks": 256} # ... when every CU gets at least one (64-tile chunk, 32-cout group) unit
WINO2D = {"enabled": True,    # the same for Conv2d 3x3 stride 1 on even maps (wino2d.hip) ...
          "rb": True, "rb_min_chunks": 256, "rb_max_cb": 8,
          # the row-brick kernel's 2D form (widths 14 or multiples of 28; bit-identical) for layers of <= 128 input channels: after the
          # round-3 fix of wino2d's spilled scalars still 70 vs 82 us at 32ch@112^2, 68 vs 75 at 64ch@56^2, 194 vs 201 at 128ch@56^2
          # (32 crops; 265 vs 350 / 218 vs 266 / 693 vs 754 at 128 crops); 320 -> 128 loses (443 vs 405) and stays on wino2d
          "dilated": True,    # dilated layers whose maps divide by 2*dilation: d*d interleaved sub-grids (feature CNN layer4; round 3)
          "odd": True,        # odd maps too (the trunk's 47 x 155 level; the trunk's 3x3 convolutions went 2.02 -> 1.76 ms per pair with the
                              # Winograd layers incl. the odd ones, see "min_chunks"); False: direct kernel as in rounds 1-2
          "min_chunks": 4}    # ... with at least this many rounds-of-four tile groups per cout group (else the direct kernel;
                              #     measured on the R-50-FPN trunk: 64 and 16 give 2.02 ms of 3x3 convs per pair, 4 gives 1.76)
CONV2D_FILL = {"enabled": True}   # 3x3 Conv2d tile choice: trade MFMA padding for waves when a layer cannot fill the 1024 SIMDs
DIRECT = {"enabled": True}    # LDS-free kernels (tapdirect.hip, downdirect.hip, conv2ddirect) and the Winograd kernels built on their plans; False: the generic kernel (tapconv.hip) for every layer
DOWN = {"enabled": True, "tile": None,     # stride-2 Conv3d kernels; "tile" = development override (tools/experiments/exp_conv.py)
        "min_groups": 700}                 # cout tiles per wave grow while >= ~0.7 groups per SIMD remain (measured)


def choose_tile_down(OH, OW):
    """Output tile (R, WT) of the stride-2 kernel: least MFMA padding, then the largest tile (more MFMAs per staged
    byte and per weight load; measured on the 14x14 and 7x7 maps, tools/experiments/exp_conv.py), then the widest rows."""
    if DOWN["tile"]:
        return DOWN["tile"]
    best = None
    for r in range(1, OH + 1):
        for wt in range(1, OW + 1):
            if r * wt > MAX_SLOTS:
                continue
            vox = (r + 1) * (wt + 1) + (r + 1) * wt + r * (wt + 1) + r * wt
            if -(-(vox * 2) // 64) > 18:
                continue
            nvt = -(-(r * wt) // 16)
            waste = (-(-OH // r)) * (-(-OW // wt)) * nvt * 16 / (OH * OW)
            key = (-round(waste, 2), r * wt, wt)
            if best is None or key > best[0]:
                best = (key, r, wt)
    return best[1], best[2]


def plan_conv3d(x, y, stride, cout, relu):
    """Conv3d(k3,pad1,stride) on a blocked tensor with halo 1."""
    assert (x.pd, x.ph, x.pw) == (1, 1, 1)
    classes = taps_conv((3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1))
    pl = ConvPlan(x, y, classes, stride, 1, (y.D, y.H, y.W), cout, relu, slide=(stride == 1 and SLIDE["enabled"]))
    if pl.slide and DIRECT["enabled"]:
        pl.direct = True
        pl.kname = pl.kname.replace("tapslide", "tapdirect")
        if WINO["enabled"] and not (y.D | y.H | y.W) & 1 and x.N * x.n_stride * 4 < 2 ** 32:
            pl.wino = True
            pl.kname = "wino3d_kernel<%d>" % pl.slide_ct
            chunks = x.N * (y.D // 2) * (y.H // 2) * (y.W // 2) // 64
            if (WINO.get("rb") and chunks * (pl.p.cout_pad // 32) >= WINO["rb_min_chunks"]
                    and _lib.lib().drc_conv3d_k3_wino_rb_supported(pl.p.cout_pad, y.D, y.H, y.W)):
                pl.rb = True
                pl.kname = "wino3d_rb_kernel<%d>" % (7 if y.W == 14 else 14)
    elif not pl.slide and stride == 1 and SLIDE["enabled"] and DIRECT["enabled"] and getattr(pl, "slide_small_ok", False):
        wchunks = x.N * (y.D // 2) * (y.H // 2) * (y.W // 2) // 64
        if WINO["enabled"] and wchunks >= WINO["small"] > 0 and not (y.D | y.H | y.W) & 1 and x.N * x.n_stride * 4 < 2 ** 32:
            # even maps at small batch (Config B's quarter-resolution hourglass layers, 16 ROIs: 6 x 14 x 14): Winograd still halves
            # the generic kernel's time (4 ROIs at 12 x 28 x 28: 74 vs 119 us direct; round 3 -- rounds 1-2 sent these to tapconv);
            # below ~48 chunks every kernel sits at its ~65 us latency floor and the direct one is marginally ahead
            pl.slide, pl.direct, pl.wino = True, True, True
            pl.slide_ct = SLIDE["ct"] if (pl.p.cout_pad // 16) % SLIDE["ct"] == 0 else 1
            pl.kname = "wino3d_kernel<%d>" % pl.slide_ct
        else:
            # odd maps (no Winograd) at small batch: the direct kernel with one cout tile per wave
            pl.slide, pl.direct, pl.slide_ct = True, True, 1
            pl.kname = "tapdirect_kernel<%d,1>" % (-(-(pl.p.R * pl.p.WT) // 16))
    elif pl.slide:
        # DIRECT off: the generic kernel (tapconv.hip).  (Rounds 1-2 had an LDS-staged sliding-window kernel here -- tapslide.hip -- which no
        # default plan selected any more; it left the library in round 6, attic/.)
        pl.slide = False
        pl.kname = pl._generic_kname
    if stride == 2 and DOWN["enabled"] and DIRECT["enabled"] and DIRECT.get("down", True):      # (else: the generic kernel)
        pl.down = True
        pl.p.R, pl.p.WT = choose_tile_down(y.H, y.W)
        nvt = -(-(pl.p.R * pl.p.WT) // 16)
        ct = pl.p.cout_pad // 16
        tiles = x.N * y.D * (-(-y.H // pl.p.R)) * (-(-y.W // pl.p.WT))
        CT = 4 if ct % 4 == 0 else (2 if ct % 2 == 0 else 1)
        while CT > 1 and (nvt * CT > 28 or tiles * (ct // CT) < DOWN["min_groups"]):
            CT //= 2
        pl.down_ct = CT
        pl.direct = True
        pl.kname = "downdirect_kernel<%d,%d>" % (nvt, CT)
    return pl


FUSED_DECONV = {"enabled": True}
DECONV_DIRECT = {"enabled": True,     # LDS-free fused transposed conv (deconvdirect.hip) instead of the LDS-staged tapdeconv.hip
                 "ct": 2}             # cout tiles per wave when they pair up (development knob: tools/experiments/exp_conv.py DC_CT)


DECONV_TILE = None    # development override (tools/experiments/exp_conv.py)


def choose_tile_deconv(H, W):
    """Input tile (R, WT) of the fused transposed-conv kernel.  Measured on MI355X (tools/experiments/exp_conv.py, 14x14 and 7x7 maps):
    tiles of 4 voxel-tiles (49..64 slots, one cout tile per wave) beat smaller ones even with more MFMA padding, because a
    tap step then carries 16 MFMAs; among those the least padding wins."""
    if DECONV_TILE:
        return DECONV_TILE
    best = None
    for r in range(1, H + 1):
        for wt in range(1, W + 1):
            if r * wt > 64 or -(-((r + 1) * (wt + 1) * 2) // 64) > 5:      # 8 classes x 4 voxel tiles of accumulators
                continue
            nvt = -(-(r * wt) // 16)
            waste = (-(-H // r)) * (-(-W // wt)) * nvt * 16 / (H * W)
            key = (nvt == 4 and waste <= 1.35, -round(waste, 3), wt, r * wt)
            if best is None or key > best[0]:
                best = (key, r, wt)
    return best[1], best[2]


def plan_deconv3d(x, y, cout, relu):
    assert (x.pd, x.ph, x.pw) == (1, 1, 1) and (y.D, y.H, y.W) == (2 * x.D, 2 * x.H, 2 * x.W)
    pl = ConvPlan(x, y, taps_deconv3d_k3s2(), 1, 2, (x.D, x.H, x.W), cout, relu)
    if FUSED_DECONV["enabled"]:
        pl.fused_deconv = True
        pl.p.R, pl.p.WT = choose_tile_deconv(x.H, x.W)
        nvt = -(-(pl.p.R * pl.p.WT) // 16)
        pl.kname = "tapdeconv_kernel<%d,%d>" % (nvt, 2 if nvt <= 3 and (pl.p.cout_pad // 16) % 2 == 0 else 1)
        if (DIRECT["enabled"] and DECONV_DIRECT["enabled"] and x.N * x.n_stride * 4 < 2 ** 32 and y.N * y.n_stride * 4 < 2 ** 32):
            # LDS-free kernel: 32 consecutive voxels of the flattened (n,d,h,w) index per wave (no tile shape), two cout tiles per wave
            # when the cout tiles pair up (every layer of the regressor)
            ct = pl.p.cout_pad // 16
            CT = 2 if ct % 2 == 0 and DECONV_DIRECT["ct"] == 2 else 1
            pl.deconv_direct, pl.direct, pl.deconv_ct = True, True, CT
            pl.p.R, pl.p.WT = 1, min(x.W, 32)
            pl.kname = "deconvdirect_kernel<2,%d>" % CT
    return pl


def plan_conv2d(x, y, k, stride, pad, dilation, cout, relu):
    """Conv2d(k,stride,pad,dilation) on blocked 2D tensors (D=1, pd=0)."""
    assert x.pd == 0 and x.D == 1
    classes = taps_conv((1, k, k), (1, dilation, dilation), (0, pad, pad), (0, x.ph, x.pw))
    pl = ConvPlan(x, y, classes, stride, 1, (1, y.H, y.W), cout, relu)
    if k == 3 and DIRECT["enabled"] and DIRECT.get("conv2d", True):
        # LDS-free kernel: any stride (1|2) and dilation; the tile only has to give full voxel tiles
        best = None
        ct_ = pl.p.cout_pad // 16
        for r in range(1, min(y.H, MAX_SLOTS) + 1):
            for wt in range(1, min(y.W, MAX_SLOTS // r) + 1):
                nvt_ = -(-(r * wt) // 16)
                ntile = x.N * (-(-y.H // r)) * (-(-y.W // wt))
                waste = (-(-y.H // r)) * (-(-y.W // wt)) * nvt_ * 16 / (y.H * y.W)
                # small maps with many channels (the trunk's 512 -> 512 layers on 12 x 39: 12 whole-row tiles x 32 cout tiles = 384 waves for
                # 1024 SIMDs, each walking K = 4608): a smaller tile that fills the chip beats the one with the least padding (round 3)
                fill = min(1.0, ntile * ct_ / 1024.0) if CONV2D_FILL["enabled"] else 1.0
                key = (round(fill / waste, 2), r * wt, wt)
                if best is None or key > best[0]:
                    best = (key, r, wt)
        pl.c2d, pl.direct = True, True
        pl.p.R, pl.p.WT = best[1], best[2]
        nvt = -(-(best[1] * best[2]) // 16)
        ct = pl.p.cout_pad // 16
        tiles = x.N * (-(-y.H // best[1])) * (-(-y.W // best[2]))
        CT = 4 if ct % 4 == 0 else (2 if ct % 2 == 0 else 1)
        while CT > 1 and (nvt * CT > 28 or tiles * (ct // CT) < 700):
            CT //= 2
        pl.c2d_ct = CT
        pl.kname = "conv2ddirect_kernel<%d,%d>" % (nvt, CT)
        # stride-1, undilated layers on even maps with enough tile groups for every block: Winograd F(2x2,3x3) (wino2d.hip)
        wct = 2 if ct % 2 == 0 else 1
        # (odd maps: the last tile row / column is half used, wino2d.hip; their patch row 3 lies in the slack behind the tensor)
        wtiles = x.N * ((y.H + 1) // 2) * ((y.W + 1) // 2)
        # (drc_conv2d_k3_wino_fwd instantiates dilations 1, 2 and 4 only; any other 3x3 layer keeps the direct kernel)
        dil_ok = dilation == 1 or (WINO2D["dilated"] and dilation in (2, 4) and y.H % (2 * dilation) == 0 and y.W % (2 * dilation) == 0)
        if (WINO2D["enabled"] and stride == 1 and dil_ok and pad == dilation and (WINO2D["odd"] or not (y.H | y.W) & 1) and
                (x.N * x.n_stride + 2 * x.h_stride) * 4 < 2 ** 32 and wtiles // 64 >= WINO2D["min_chunks"]):
            pl.wino = True
            pl.slide_ct = wct
            pl.kname = "wino2d_kernel<%d>" % wct
            chunks = wtiles // 64
            if (WINO2D.get("rb") and dilation == 1 and x.cb <= WINO2D["rb_max_cb"] and not (y.H | y.W) & 1 and chunks * (pl.p.cout_pad // 32) >= WINO2D["rb_min_chunks"]
                    and _lib.lib().drc_conv2d_k3_wino_rb_supported(pl.p.cout_pad, y.H, y.W)):
                pl.rb = True
                pl.kname = "wino2d_rb_kernel<%d>" % (7 if y.W == 14 else 14)
    if k == 1 and POINTWISE["enabled"]:
        pl.pointwise = True
        ct = pl.p.cout_pad // 16
        tiles = -(-(x.N * y.H * y.W) // 16)
        if ct % 4 == 0 and tiles // 4 * (ct // 4) >= 2048:
            vc = (4, 4)
        elif ct % 2 == 0 and tiles // 4 * (ct // 2) >= 2048:
            vc = (4, 2)
        elif ct % 2 == 0 and (tiles // 2 * (ct // 2) >= 1024 or x.cb < 16):        # mirrors drc_conv2d_k1_fwd's choice
            vc = (2, 2)
        elif tiles // 2 * ct >= 1024:
            vc = (2, 1)
        else:
            vc = (1, 1)
        pl.kname = "pointwise_kernel<%d,%d>" % vc
    return pl


def plan_deconv2d(x, y, k, cout, relu=False):
    """2D transposed conv (k in {1,3}, stride 2) on blocked tensors: y is exactly twice x."""
    assert x.pd == 0 and x.D == 1 and (y.H, y.W) == (2 * x.H, 2 * x.W) and x.ph >= 1 and x.pw >= 1
    return ConvPlan(x, y, taps_deconv2d_s2(k, (0, x.ph, x.pw)), 1, 2, (1, x.H, x.W), cout, relu)


# ------------------------------------------------------------------------------------------- other ops
def cost_volume_blocked(left, right, out, lo4, hi4, in_blocked_pad=0):
    """left/right NCHW (or blocked 2D) -> out: Blocked [N,2C,Dp,Hp,Wp] halo 1."""
    st = _lib.lib().drc_cost_volume_blocked_fwd(_ptr(left), _ptr(right), _ptr(out.storage), out.N, out.C // 2, out.D, out.H, out.W,
                                                lo4, hi4, in_blocked_pad, _stream_ptr(out.device))
    _lib.check(st, "drc_cost_volume_blocked_fwd")


def conv3d_cout1(x, w27, res, out):
    """x Blocked (halo 1) -> out dense [N,D,H,W] (+res)."""
    st = _lib.lib().drc_conv3d_cout1_fwd(_ptr(x.storage), _ptr(w27), _ptr(res), _ptr(out), x.N, x.cb, x.D, x.H, x.W,
                                         _stream_ptr(x.device))
    _lib.check(st, "drc_conv3d_cout1_fwd")


# eval: the 3x3 layers of ResNet-FPN / the RPN head on LARGE maps through BridgedConv2dS16.  Measured on the KITTI pair
# (tools/experiments/exp_trunk_s16.py, profiles/r5_exp_trunk_s16_flat.log): with min_tiles 192 -- layer1's 3x3 layers, the FPN blocks of P2 / P3,
# the RPN head's convolution on P2 / P3 -- trunk 3.07 -> 2.78 ms, 2D stage 7.08 -> 6.76 ms; lower thresholds lose (the converters at both ends
# and the chained launches cost more than the fp32 Winograd kernels they replace on smaller maps)
TRUNK_S16 = {"enabled": True, "min_tiles": 192, "min_rows": 24}


class BridgedConv2dS16:
    """A 3x3 stride-1 pad-1 convolution (+BN / bias, +ReLU) between BLOCKED fp32 tensors on the split-f16 kernel (convs16r.hip, any map
    size): the input is converted to RS16 in slices of <= 128 channels, one launch per slice -- a launch adds the previous partial sum as its
    residual; the folded BN scale goes to every launch, the shift to the first, the ReLU to the last (BN is affine) -- and the result is
    converted back.  Reference layers: Bottleneck.conv2 (backbone/resnet.py:274-316), the FPN output blocks (backbone/fpn.py:44-77).
    `cache`: a dict the RS16 maps of equal shape are shared through (layers run one after the other)."""

    @staticmethod
    def slices(cin):
        if cin in (32, 64, 128):
            return ((0, cin),)
        if cin in (256, 384, 512):
            return tuple((a, a + 128) for a in range(0, cin, 128))
        return None

    @staticmethod
    def worth(N, cin, cout, H, W):
        """Large maps only: a column of the kernel is 28 rows x 28 columns x 32 couts, and the converters at both ends are extra launches.
        (No eval gate of its own: its callers -- the ResNet-FPN trunk, the RPN head, the heads of the 2D stage -- are inference-only on this
        engine and raise NotImplementedError in train mode before they get here.)"""
        if not TRUNK_S16["enabled"] or BridgedConv2dS16.slices(cin) is None or cout not in (32, 64, 128, 256, 512) or not S16["enabled"]:
            return False
        if not s16_allowed():               # a guarded pass is being repeated on the fp32 kernels (OverflowGuard)
            return False
        tiles = N * -(-H // 28) * -(-W // 28) * (cout // 32)
        return H >= TRUNK_S16["min_rows"] and tiles >= TRUNK_S16["min_tiles"]

    def __init__(self, N, cin, cout, H, W, relu, device, cache):
        self.bounds = self.slices(cin)
        if self.bounds is None or not s16_supported(self.bounds[0][1] - self.bounds[0][0], cout, 1, H, W, "2d"):
            raise ValueError("BridgedConv2dS16: unsupported shape")
        self.N, self.cin, self.cout, self.H, self.W, self.device = N, cin, cout, H, W, device

        def shared(tag, ch):
            k = (tag, N, ch, H, W)
            if k not in cache:
                cache[k] = RS16(N, ch, 1, H, W, 0, device)
            return cache[k]
        self.x16 = [shared(("x", i), b - a) for i, (a, b) in enumerate(self.bounds)]
        self.y16 = [shared(("y", i), cout) for i in range(min(2, len(self.bounds)))]
        last = len(self.bounds) - 1
        self.plans = [ConvPlanS16(N, b - a, cout, 1, H, W, bool(relu) and i == last, device=device, kind="2d") for i, (a, b) in enumerate(self.bounds)]

    def run(self, x, packs, shift, zero_shift, y):
        """x, y: Blocked fp32 (x: halo >= 0 of any width); packs: [(packed split-f16 weights of the slice, its epilogue scale)]."""
        if (x.C, x.H, x.W, y.C, y.H, y.W) != (self.cin, self.H, self.W, self.cout, self.H, self.W) or x.N < self.N or y.N < self.N:
            raise ValueError("BridgedConv2dS16.run: tensors differ from the plan")
        prev = None
        for i, ((a, b), pl, (w16, sc16)) in enumerate(zip(self.bounds, self.plans, packs)):
            self.x16[i].from_blocked(x if len(self.bounds) == 1 else BlockedSlice(x, a // CB, b - a))
            out = self.y16[i & 1]
            pl.run(self.x16[i], w16, sc16, shift if i == 0 else zero_shift, y16=out, res=prev)
            prev = out
        prev.to_blocked(y)


def head_gather(S, scale, res, out):
    """The second half of a fused cout-1 head (ConvPlanS16.run(head=...)): out dense [N,D,H,W] = (res or 0) + scale * the nine shifted
    in-plane partial sums of S (fp32 [N][D][H][W][12])."""
    N, D, H, W = out.shape
    st = _lib.lib().drc_head_gather_fwd(_ptr(S), _ptr(res), _ptr(out), N, D, H, W, C.c_float(scale), _stream_ptr(out.device))
    _lib.check(st, "drc_head_gather_fwd")


def upsample_softargmin(cost, disp, maxdisp, mindisp):
    N, Dp, Hp, Wp = cost.shape
    _, H, W = disp.shape
    st = _lib.lib().drc_upsample_softargmin_fwd(_ptr(cost), _ptr(disp), N, Dp, Hp, Wp, maxdisp - mindisp, H, W, mindisp,
                                                _stream_ptr(cost.device))
    _lib.check(st, "drc_upsample_softargmin_fwd")


# ------------------------------------------------------------------------------------------- train-mode BatchNorm
def _geom8(t):
    """int[10] geometry {N, CB, D, H, W, pd, ph, pw, cb_total, cb_off} of a Blocked tensor or a BlockedSlice of one."""
    import ctypes
    base = getattr(t, "base", None)
    return (ctypes.c_int * 10)(t.N, t.cb, t.D, t.H, t.W, t.pd, t.ph, t.pw, base.cb if base is not None else t.cb,
                                t.cb_off if base is not None else 0)


_BN_SCRATCH = {}
BN_MAX_CHUNKS = 512


def bn_scratch(dev, cb):
    """Partials + ticket words of the fixed-order BatchNorm reductions, one per (device, stream); tickets start (and end) at zero."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0, cb)   # ticket offset depends on cb
    buf = _BN_SCRATCH.get(key)
    if buf is None:
        buf = _BN_SCRATCH[key] = torch.zeros(BN_MAX_CHUNKS * cb * 32 + cb * 32 * 33, dtype=torch.float32, device=dev)   # DRC_BN_SCRATCH_FLOATS
    return buf


_WGRAD_SCRATCH = {}


def wgrad_scratch(dev):
    """Per-(device, stream) workspace of the weight-gradient kernels' partial sums (DRC_WGRAD_SCRATCH_FLOATS, 61 MB)."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)
    buf = _WGRAD_SCRATCH.get(key)
    if buf is None:
        buf = _WGRAD_SCRATCH[key] = torch.empty(_lib.WGRAD_SCRATCH_FLOATS, dtype=torch.float32, device=dev)
    return buf


_SCRATCH = {}


def scratch(dev, tag, floats):
    """Per-(device, stream, tag) float workspace for the fixed-order two-launch reductions (loss sums, classifier weight gradient,
    soft-argmin adjoint footprints): grows to the largest request and is then reused, so a captured step keeps its pointers."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0, tag)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < floats:
        buf = _SCRATCH[key] = torch.empty(int(floats), dtype=torch.float32, device=dev)
    return buf


def bn_batch_stats(raw):
    """Per-channel batch mean and biased variance of a Blocked tensor's interior (one pass, Chan-combined, fixed order)."""
    dev = raw.device
    C16 = raw.cb * CB
    M = raw.N * raw.D * raw.H * raw.W
    stats = torch.empty(2, C16, dtype=torch.float32, device=dev)
    st = _lib.lib().drc_bn_stats_blocked(_ptr(raw.storage), _geom8(raw), _ptr(stats), _ptr(bn_scratch(dev, raw.cb)), _stream_ptr(dev))
    _lib.check(st, "drc_bn_stats_blocked")
    return stats[0], stats[1] / M, M


def bn_batch_stats_raw(raw):
    """stats [2][C16] = (mean, sum of squared deviations) of a Blocked tensor's interior, and the voxel count."""
    dev = raw.device
    stats = torch.empty(2, raw.cb * CB, dtype=torch.float32, device=dev)
    st = _lib.lib().drc_bn_stats_blocked(_ptr(raw.storage), _geom8(raw), _ptr(stats), _ptr(bn_scratch(dev, raw.cb)), _stream_ptr(dev))
    _lib.check(st, "drc_bn_stats_blocked")
    return stats, raw.N * raw.D * raw.H * raw.W


def bn_finalize(stats, count, bn, cout):
    """invstd from the batch statistics + the module's running-statistics update, one launch (nn.BatchNorm semantics)."""
    dev = stats.device
    invstd = torch.empty(stats.shape[1], dtype=torch.float32, device=dev)
    track = bn.track_running_stats and bn.running_mean is not None
    st = _lib.lib().drc_bn_finalize(_ptr(stats), stats.shape[1], cout, count, float(bn.eps), float(bn.momentum),
                                    _ptr(bn.running_mean) if track else None, _ptr(bn.running_var) if track else None,
                                    _ptr(bn.num_batches_tracked) if track else None, _ptr(invstd), _stream_ptr(dev))
    _lib.check(st, "drc_bn_finalize")
    if track:   # the kernel wrote the buffers behind torch's back: bump their version counters (BN-fold caches key on them)
        for t in (bn.running_mean, bn.running_var, bn.num_batches_tracked):
            torch.autograd.graph.increment_version(t)
    return invstd


def bn_apply(raw, y, res, mean, invstd, gamma, beta, relu):
    st = _lib.lib().drc_bn_apply_blocked(_ptr(raw.storage), _geom8(raw), _ptr(y.storage), _geom8(y),
                                         _ptr(res.storage) if res is not None else None, _geom8(res) if res is not None else None,
                                         _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), int(relu), _stream_ptr(raw.device))
    _lib.check(st, "drc_bn_apply_blocked")


# ------------------------------------------------------------------------------------------- fp16-storage regressor (conv16.hip)
CB16 = 32      # fp16 channels per 64-byte voxel line


class Blocked16:
    """fp16-storage twin of Blocked: half[N][ceil(C/32)][D+2pd][H+2ph][W+2pw][32], zero halo (same byte geometry as the fp32
    layout with 16 channels per line).  Strides are in ELEMENTS (halfs)."""

    def __init__(self, N, C_, D, H, W, pd, ph, pw, device, storage=None):
        self.N, self.C, self.D, self.H, self.W = N, C_, D, H, W
        self.pd, self.ph, self.pw = pd, ph, pw
        self.cb = (C_ + CB16 - 1) // CB16
        self.Dp, self.Hp, self.Wp = D + 2 * pd, H + 2 * ph, W + 2 * pw
        self.h_stride = self.Wp * CB16
        self.d_stride = self.Hp * self.h_stride
        self.cb_stride = self.Dp * self.d_stride
        self.n_stride = self.cb * self.cb_stride
        self.numel = N * self.n_stride
        if storage is None:
            self.storage = torch.zeros(self.numel + 2 * SLACK_FLOATS, dtype=torch.float16, device=device)
        else:
            if storage.numel() < self.numel + 2 * SLACK_FLOATS:
                raise ValueError("Blocked16: the given storage is too small for this geometry")
            self.storage = storage.narrow(0, 0, self.numel + 2 * SLACK_FLOATS)
        self.device = device

    @property
    def interior_off(self):
        return self.pd * self.d_stride + self.ph * self.h_stride + self.pw * CB16

    def view6(self):
        return self.storage[: self.numel].view(self.N, self.cb, self.Dp, self.Hp, self.Wp, CB16)

    def from_dense(self, dense):
        """dense [N,C,D,H,W] fp32/fp16 -> interior (torch ops: test plumbing, not on the product path)."""
        v = self.view6()
        n, c = dense.shape[:2]
        pad_c = self.cb * CB16 - c
        d = torch.nn.functional.pad(dense.to(torch.float16), (0, 0, 0, 0, 0, 0, 0, pad_c)) if pad_c else dense.to(torch.float16)
        d = d.view(n, self.cb, CB16, self.D, self.H, self.W).permute(0, 1, 3, 4, 5, 2)
        v[:, :, self.pd:self.pd + self.D, self.ph:self.ph + self.H, self.pw:self.pw + self.W, :] = d
        return self

    def to_dense(self):
        v = self.view6()[:, :, self.pd:self.pd + self.D, self.ph:self.ph + self.H, self.pw:self.pw + self.W, :]
        return v.permute(0, 1, 5, 2, 3, 4).reshape(self.N, self.cb * CB16, self.D, self.H, self.W)[:, : self.C].float()


class Blocked16Slice:
    """Channel blocks [cb_off, cb_off + ceil(C/32)) of a Blocked16 tensor (a concat operand): same strides, storage view at the slice."""

    def __init__(self, base, cb_off, C_):
        self.base, self.cb_off = base, cb_off
        self.N, self.C, self.D, self.H, self.W = base.N, C_, base.D, base.H, base.W
        self.pd, self.ph, self.pw = base.pd, base.ph, base.pw
        self.cb = (C_ + CB16 - 1) // CB16
        assert cb_off >= 0 and cb_off + self.cb <= base.cb
        self.Dp, self.Hp, self.Wp = base.Dp, base.Hp, base.Wp
        self.h_stride, self.d_stride, self.cb_stride, self.n_stride = base.h_stride, base.d_stride, base.cb_stride, base.n_stride
        self.storage = base.storage[cb_off * base.cb_stride:]
        self.interior_off = base.interior_off
        self.device = base.device


def dense_to_blocked16(dense, out):
    """fp32 [N,C,H,W] -> interior of the 2D Blocked16 `out` (HIP kernel)."""
    require_gpu(dense, "dense_to_blocked16")
    dense = dense.contiguous().float()
    st = _lib.lib().drc_dense_to_blocked16(_ptr(dense), _ptr(out.storage), out.N, out.C, out.H, out.W, out.ph, out.pw, _stream_ptr(out.device))
    _lib.check(st, "drc_dense_to_blocked16")
    return out


def plan_conv2d16(x, y, k, stride, pad, dilation, cout, relu):
    """Conv2d(k, stride, pad, dilation) on 2D Blocked16 tensors (D = 1, pd = 0): the LDS-tiled kernel for stride-1 undilated 3x3 layers,
    conv16.hip's generic tap walk for the rest (stride 2, dilation, 1x1)."""
    assert x.pd == 0 and x.D == 1
    return ConvPlan16(x, y, taps_conv((1, k, k), (1, dilation, dilation), (0, pad, pad), (0, x.ph, x.pw)), stride, 1, (1, y.H, y.W), cout, relu)


def cost_volume16_from16(feat, right_first_unit, out, lo4, hi4):
    """feat: 2D Blocked16 [units,32ch,H',W'] (left units, then right units from `right_first_unit`) -> out: Blocked16 [N,64,D',H',W'] halo 1."""
    st = _lib.lib().drc_cost_volume16_from16(_ptr(feat.storage), _ptr(out.storage), out.N, right_first_unit, feat.C, out.D, out.H, out.W, lo4, hi4,
                                             feat.ph, _stream_ptr(out.device))
    _lib.check(st, "drc_cost_volume16_from16")


def pack_weight16(w, transposed=False):
    """[Cout,Cin,*k] (or ConvTranspose [Cin,Cout,*k]) fp32 -> [K taps][ceil(Cin/32)][cout_pad][32] fp16: lane (cout j, g) of the
    f16 MFMA's A operand reads channels 8g..8g+7 as one 16-byte load."""
    if transposed:
        w = w.transpose(0, 1)
    cout, cin = w.shape[:2]
    K = int(math.prod(w.shape[2:]))
    cb = (cin + CB16 - 1) // CB16
    cout_pad = (cout + CB - 1) // CB * CB
    wp = torch.zeros(K, cb * CB16, cout_pad, dtype=torch.float32, device=w.device)
    wp[:, :cin, :cout] = w.detach().float().reshape(cout, cin, K).permute(2, 1, 0)
    return wp.view(K, cb, CB16, cout_pad).permute(0, 1, 3, 2).contiguous().to(torch.float16)


def choose_tile16(OH, OW):
    """(R, WT) with R*WT <= 64 output voxels per wave: four voxel tiles per weight load whenever the map allows (a 16-voxel tile
    re-loads the weights four times as often: measured 788 vs ~250 us per full-resolution layer), then the least MFMA padding,
    then the widest rows."""
    best = None
    for r in range(1, OH + 1):
        for wt in range(1, min(OW, 64 // r) + 1):
            nvt = -(-(r * wt) // 16)
            waste = (-(-OH // r)) * (-(-OW // wt)) * nvt * 16 / (OH * OW)
            key = (min(nvt, 4) if waste <= 1.35 else 0, -round(waste, 2), r * wt, wt)
            if best is None or key > best[0]:
                best = (key, r, wt)
    return best[1], best[2]


C16_TILE = {"enabled": True}      # 3x3x3 / 3x3 fp16 layers on the LDS-tiled kernels (conv16t.hip: stride 1; conv16x.hip: stride 2 and transposed, round 4); False: conv16.hip as in round 2


def walk2_takes(cb, OD, OH, cw):
    """conv16t.hip's drc_t16_conv3d_walk2_try: two input blocks, depth >= 4, rows filling the 7-row (four cout tiles) / 14-row (two) tiles."""
    return cb == 2 and OD >= 4 and cw in (2, 4) and OH % (7 * (4 // cw)) == 0


def x16_rows(OH, cw, stride, cb=1):
    """Rows per wave of the conv16x.hip kernels (launch_d / launch_u there): the big tile unless it pads the map's rows by more than
    25 % over the small one's.  stride 0 = the transposed kernel (big = seven rows if its stage leaves two blocks per CU)."""
    rg = 4 // cw
    if stride == 0:
        big, small = 7, 2
        if cb * 2 * (7 * rg + 1) * 1024 + 1024 > 80 * 1024:
            return small
    elif stride == 1:
        big, small = 7, 2
    else:
        big, small = {1: 7, 2: 4, 4: 2}[rg], (1 if rg == 4 else 2)
    pad = lambda rw: -(-OH // (rw * rg)) * rw * rg
    return big if pad(big) * 4 <= pad(small) * 5 else small


class ConvPlan16:
    """A resolved conv16 launch (tap-grid classes as in ConvPlan).  dense1: the 32 -> 1 classifier conv, whose single cout is
    written as a dense fp32 [N,D,H,W] volume (+ an optional dense fp32 residual)."""

    def __init__(self, x, y_geom, classes, in_mul, out_mul, grid_dhw, cout, relu, dense1=False):
        p = DrcTapconvParams()
        OD, OH, OW = grid_dhw
        p.N, p.OD, p.OH, p.OW = x.N, OD, OH, OW
        p.in_mul, p.out_mul = in_mul, out_mul
        p.cb_in = x.cb
        p.cout_pad = (cout + CB - 1) // CB * CB
        p.relu = int(relu)
        p.n_classes = len(classes)
        p.R, p.WT = choose_tile16(OH, OW)
        p.reserved = 1 if dense1 else 0
        for ci, c in enumerate(classes):
            k = p.cls[ci]
            k.nd, k.nh, k.nw = c["n"]
            k.dd0, k.dh0, k.dw0 = c["first"]
            k.sd, k.sh, k.sw = c["step"]
            k.wbase = c["wbase"]
            k.wsd, k.wsh, k.wsw = c["wstep"]
            k.out_off_d, k.out_off_h, k.out_off_w = c["off"]
        self.p, self.device, self.dense1 = p, x.device, dense1
        ntaps = sum(c["n"][0] * c["n"][1] * c["n"][2] for c in classes)
        self.flops = 2 * x.N * OD * OH * OW * ntaps * x.C * cout
        self.kname = "conv16_kernel<%d,%d>" % (min(-(-(p.R * p.WT) // 16), 4), 2 if (p.cout_pad // 16) % 2 == 0 else 1)
        self.tile = bool(C16_TILE["enabled"] and _lib.lib().drc_conv16_k3_tile_supported(C.byref(p)))
        ct = p.cout_pad // 16
        cw = 4 if ct % 4 == 0 else (2 if ct % 2 == 0 else 1)                       # cout tiles side by side in a block of conv16x.hip
        if self.tile:
            rw = 4 if (OH % 16 == 0 or OH >= 48) else 2
            if classes[0]["n"][0] == 3 and x.cb == 1 and p.cout_pad <= 32 and OD >= 4:      # conv16t.hip: the depth-sliding walk
                self.kname = "conv16s_kernel<%d,%d>" % (rw, ct)
                if OH % 28 == 0 or OH % 16 == 0:                                            # rows fill the tiles: two slices in flight (round 4)
                    self.kname = "conv16sp_kernel<%d,%d>" % (7 if OH % 28 == 0 else 4, ct)
            elif classes[0]["n"][0] == 3 and not dense1:                                     # conv16x.hip, stride 1 (round 4)
                self.kname = "conv16d_kernel<%d,%d,1>" % (x16_rows(OH, cw, 1), cw)
                if walk2_takes(x.cb, OD, OH, cw):                                            # two input blocks, full row tiles: the depth walk
                    self.kname = "conv16sw_kernel<7,%d>" % cw
            else:
                self.kname = "conv16t_kernel<%d,%d,%d>" % (rw, cw, classes[0]["n"][0])
        self.costvol_lo4 = None          # plan_conv3d16_costvol: x is the feature pair, the cost volume is folded into the stage addresses
        # round 4 (conv16x.hip): the stride-2 and transposed 3x3x3 layers on LDS tiles too
        self.tile_x = None
        if C16_TILE["enabled"] and not self.tile and not dense1:
            if _lib.lib().drc_conv16_k3s2_tile_supported(C.byref(p)):
                self.tile_x = "drc_conv16_k3s2_tile_fwd"
                self.kname = "conv16d_kernel<%d,%d,2>" % (x16_rows(OH, cw, 2), cw)
            elif _lib.lib().drc_deconv16_k3s2_tile_supported(C.byref(p)):
                self.tile_x = "drc_deconv16_k3s2_tile_fwd"
                self.kname = "conv16u_kernel<%d,%d,%d>" % (x16_rows(OH, cw, 0, x.cb), cw, x.cb)

    def run(self, x, w16, scale, shift, y, res=None):
        p = self.p
        p.x, p.w = x.storage.data_ptr(), w16.data_ptr()
        p.x_n_stride, p.x_cb_stride, p.x_d_stride, p.x_h_stride = x.n_stride, x.cb_stride, x.d_stride, x.h_stride
        if self.dense1:
            p.y = y.data_ptr()
            p.res = res.data_ptr() if res is not None else None
            p.scale = p.shift = None
        else:
            p.y = y.storage.data_ptr()
            p.y_n_stride, p.y_cb_stride, p.y_d_stride, p.y_h_stride = y.n_stride, y.cb_stride, y.d_stride, y.h_stride
            p.y_off0 = y.interior_off
            p.scale, p.shift = scale.data_ptr(), shift.data_ptr()
            if res is not None:
                p.res = res.storage.data_ptr()
                p.r_n_stride, p.r_cb_stride, p.r_d_stride, p.r_h_stride = res.n_stride, res.cb_stride, res.d_stride, res.h_stride
                p.r_off0 = res.interior_off
            else:
                p.res = None
        if TIMING is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        if self.costvol_lo4 is not None:
            st = _lib.lib().drc_conv16_k3_costvol_fwd(C.byref(p), self.costvol_lo4, _stream_ptr(self.device))
            _lib.check(st, "drc_conv16_k3_costvol_fwd")
        elif self.tile:
            st = _lib.lib().drc_conv16_k3_tile_fwd(C.byref(p), _stream_ptr(self.device))
            _lib.check(st, "drc_conv16_k3_tile_fwd")
        elif self.tile_x:
            st = getattr(_lib.lib(), self.tile_x)(C.byref(p), _stream_ptr(self.device))
            _lib.check(st, self.tile_x)
        else:
            st = _lib.lib().drc_conv16_fwd(C.byref(p), _stream_ptr(self.device))
            _lib.check(st, "drc_conv16_fwd")
        if TIMING is not None:
            e1.record(torch.cuda.current_stream(self.device))
            TIMING.append((self.kname, self.flops, e0, e1))


def plan_conv3d16(x, y, stride, cout, relu):
    assert (x.pd, x.ph, x.pw) == (1, 1, 1)
    return ConvPlan16(x, y, taps_conv((3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)), stride, 1, (y.D, y.H, y.W), cout, relu)


COSTVOL16_FUSED = {"enabled": True}      # dres0[0] reads the fp16 feature pair instead of a materialised fp16 cost volume (conv16x.hip, round 4)


def plan_conv3d16_costvol(pair, y, lo4, cout, relu):
    """The first 3D layer of the fp16-storage regressor on the cost volume of stackhourglass.py:115-128 without the volume: pair = Blocked16
    [N,64,1,H,W] (channel block 0 = left features, block 1 = right features; cost_volume16_blocked / cost_volume16_from16 with one slice at
    disparity 0), y = Blocked16 [N,cout,D,H,W], lo4 = the volume's first disparity.  Bit-identical to the two-kernel form."""
    assert (pair.pd, pair.ph, pair.pw) == (1, 1, 1) and pair.C == 64 and pair.D == 1 and (pair.H, pair.W) == (y.H, y.W)
    pl = ConvPlan16(pair, y, taps_conv((3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)), 1, 1, (y.D, y.H, y.W), cout, relu)
    pl.costvol_lo4, pl.tile, pl.tile_x = int(lo4), False, None
    ct = pl.p.cout_pad // 16
    cw = 4 if ct % 4 == 0 else (2 if ct % 2 == 0 else 1)
    pl.kname = "conv16d_kernel<%d,%d,1,cv>" % (x16_rows(y.H, cw, 1), cw)
    if walk2_takes(2, y.D, y.H, cw):
        pl.kname = "conv16sw_kernel<7,%d,cv>" % cw
    pl.flops = 2 * y.N * y.D * y.H * y.W * 27 * 64 * cout
    return pl


def plan_deconv3d16(x, y, cout, relu):
    assert (x.pd, x.ph, x.pw) == (1, 1, 1) and (y.D, y.H, y.W) == (2 * x.D, 2 * x.H, 2 * x.W)
    return ConvPlan16(x, y, taps_deconv3d_k3s2(), 1, 2, (x.D, x.H, x.W), cout, relu)


def plan_conv3d16_cout1(x):
    return ConvPlan16(x, None, taps_conv((3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)), 1, 1, (x.D, x.H, x.W), 1, False, dense1=True)


def cost_volume16_blocked(left, right, out, lo4, hi4, in_blocked_pad=-1):
    """left/right fp32 NCHW (in_blocked_pad < 0) or fp32 blocked 2D storage (halo in_blocked_pad) -> out: Blocked16 [N,64,D',H',W'] halo 1."""
    st = _lib.lib().drc_cost_volume16_blocked_fwd(_ptr(left), _ptr(right), _ptr(out.storage), out.N, out.C // 2, out.D, out.H, out.W,
                                                  lo4, hi4, in_blocked_pad, _stream_ptr(out.device))
    _lib.check(st, "drc_cost_volume16_blocked_fwd")


# ------------------------------------------------------------------------------------------- split-f16 ("f16x2") path, round 5
HEAD_FUSED = {"enabled": True}       # eval, split-f16 regressor: classif[0] + the 32 -> 1 layer as one fused launch + a gather (convs16.hip HEAD form) instead of a blocked fp32 tensor + cout1_mfma.hip
LASTCONV_S16 = {"enabled": True}     # eval, split-f16 2D schedule: lastconv[0] (320 -> 128) as three chained split-f16 launches over the concat's parts (runtime._ws2d_s16)
S16 = {"enabled": True}       # eval: the stride-1 3x3x3 layers at full resolution on the f16 matrix cores in split arithmetic (convs16.hip)
CV_WIDE = {"enabled": True}   # eval, large batches: the cost-volume layer on the two-tiles-per-wave kernel (convs16w.hip); False: convs16.hip's one-tile form (A/B switch)



# ---- range guard of the split-f16 path (round 6; csrc/s16_ovf.h, include/disprcnn_hip.h `drc_s16conv_params.ovf`)
# The reference computes in fp32 (config/defaults.py:22; submodule.py:19-22) and has no activation range limit; an RS16 value must stay
# within +-65504.  Every kernel that writes split-f16 values ORs 1 into the device word of the OverflowGuard in scope when it had to clamp
# (or met Inf / NaN); the component that opened the scope reads the word ONCE at the end of its forward pass (one stream synchronisation)
# and -- "auto" -- runs the pass again with every split-f16 decision turned off (s16_allowed() False: the fp32 MFMA kernels), or --
# math = "f16x2" -- raises.  Scopes nest: an inner component (PSMNet inside DispRCNN3D, the trunk inside DispRCNN) reports to the outermost
# scope and leaves the check and the re-run to it.  The scope is process-global state of the HOST layer (the reference drives one Python
# thread per process, engine/inference.py:24-50; autograd's worker threads only run backward passes, which launch no split-f16 kernel); the
# C ABI underneath stays re-entrant -- the word is an explicit argument of every launch.
_GUARD = {"cur": None, "safe": False}


class OverflowPolicy:
    """Host side of the guard: after an overflow the next `2^level` passes go straight to the fp32 kernels (level grows with every further
    overflow up to `cap` passes, a clean split-f16 pass resets it) -- a model whose activations are out of range pays the double pass at
    passes 1, 3, 7, 15, ... instead of at every one, a single outlier input costs two fp32 passes."""

    def __init__(self, cap=256):
        self.cap, self.level, self.skip, self.overflows, self.checks = cap, 0, 0, 0, 0

    def want_fast(self):
        if self.skip > 0:
            self.skip -= 1
            return False
        return True

    def report(self, overflowed):
        self.checks += 1
        if overflowed:
            self.overflows += 1
            self.skip = min(1 << self.level, self.cap)
            self.level = min(self.level + 1, 30)
        else:
            self.level = 0


class OverflowGuard:
    """One int32 device word + its policy.  `used` is set when a launch picked the word up during the scope (no split-f16 launch: no
    synchronisation at the end)."""

    def __init__(self, device):
        self.word = torch.zeros(1, dtype=torch.int32, device=device)
        self.policy = OverflowPolicy()
        self.used = False
        self.warned = False
        self.dirty = False          # launches of a scope that ended in an exception may have set the word without anyone reading it

    def ptr(self):
        self.used = True
        return C.c_void_p(self.word.data_ptr())

    def tripped(self):
        """Read (one device synchronisation) and clear the word."""
        v = int(self.word.item())
        if v:
            self.word.zero_()
        return bool(v)


_SHARED_GUARDS = {}


def one_guard(fn, device, what="guarded call", enabled=True):
    """Run fn() -- several components in a row, e.g. the ResNet-FPN trunk and then DispRCNN3D on the same stereo pair -- under ONE range
    guard of `device`: the components' own scopes nest into it, so the pass costs one read of the guard word (one stream synchronisation)
    instead of one per component, and an overflow anywhere repeats all of fn() on the fp32 kernels."""
    dev = torch.device(device)
    g = _SHARED_GUARDS.get(dev)
    if g is None:
        g = _SHARED_GUARDS[dev] = OverflowGuard(dev)
    return guarded(g, fn, what=what, enabled=enabled)


def s16_allowed():
    """False while a guarded pass is being repeated on the fp32 kernels (or its back-off runs): every split-f16 decision consults this."""
    return not _GUARD["safe"]


def _ovf_ptr():
    g = _GUARD["cur"]
    return g.ptr() if g is not None else None


def guard_in_scope():
    return _GUARD["cur"]


def guarded(guard, fn, strict=False, what="split-f16 path", enabled=True):
    """Run fn() under `guard` (see above).  strict: the caller asked for split-f16 arithmetic explicitly ("f16x2"): an overflow raises.
    A scope that is already open, a stream capture in progress (the word cannot be read inside one) or enabled = False: plain fn()."""
    if _GUARD["cur"] is not None or guard is None or not enabled or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
        return fn()
    import warnings
    pol = guard.policy
    if getattr(guard, "dirty", False):          # a previous scope ended in an exception with launches in flight: its word was never read
        guard.word.zero_()
        guard.dirty = False
    _GUARD["cur"], _GUARD["safe"] = guard, (False if strict else not pol.want_fast())
    guard.used = False
    try:
        try:
            out = fn()
        except BaseException:
            if guard.used:
                guard.dirty = True
            raise
        if not guard.used:
            return out
        over = guard.tripped()
        pol.report(over)
        if not over:
            return out
        if strict:
            raise RuntimeError(f"{what}: a value left the split-f16 range (|v| > 65504, Inf or NaN) and math = 'f16x2' was requested; use "
                               f"'auto' (re-runs on the fp32 kernels) or 'f32'")
        if not guard.warned:
            guard.warned = True
            warnings.warn(f"{what}: a value left the split-f16 range (|v| > 65504, Inf or NaN); this pass and the next {pol.skip} run on the "
                          f"fp32 kernels (reference arithmetic, config/defaults.py:22).  Reported once per model.", RuntimeWarning, stacklevel=3)
        _GUARD["safe"] = True
        guard.used = False
        out = fn()
        if guard.used and guard.tripped():      # (a component that ignored s16_allowed(): a bug, not an input property)
            raise RuntimeError(f"{what}: the fp32 re-run still issued split-f16 launches that overflowed")
        return out
    finally:
        _GUARD["cur"], _GUARD["safe"] = None, False


class RS16:
    """Split-f16 tensor halfs [N][C/32][D+2pd][H+2][8 chunks][W+2][8], zero halo (include/disprcnn_hip.h, drc_s16conv_params)."""

    def __init__(self, N, C_, D, H, W, pd, device, storage=None):
        if C_ % 32:
            raise ValueError("RS16: channels must be a multiple of 32")
        self.N, self.C, self.D, self.H, self.W, self.pd = N, C_, D, H, W, pd
        self.cb = C_ // 32
        self.unit = self.cb * (D + 2 * pd) * (H + 2) * 8 * (W + 2) * 8          # halfs per unit
        self.numel = N * self.unit
        # slack behind the last unit: a masked (ragged) last tile of the kernels stages whole slabs -- up to a tile's rows below the bottom halo
        # and a tile's columns to the right of it; what it over-reads only reaches output lanes that are not stored, but it must be mapped memory
        slack = self.slack = max(4096, 8 * 8 * (W + 2) * 8)
        if storage is None:
            self.storage = torch.zeros(self.numel + slack, dtype=torch.float16, device=device)
        else:
            if storage.numel() < self.numel + slack:
                raise ValueError("RS16: the given storage is too small for this geometry")
            self.storage = storage.narrow(0, 0, self.numel + slack)
        self.device = device

    def view7(self):
        return self.storage[: self.numel].view(self.N, self.cb, self.D + 2 * self.pd, self.H + 2, 8, self.W + 2, 8)

    def from_dense(self, dense):
        """dense [N,C,D,H,W] or [N,C,H,W] fp32 -> interior (HIP kernel)."""
        require_gpu(dense, "RS16.from_dense")
        dense = dense.contiguous()
        if self.N:
            st = _lib.lib().drc_rs16_from_dense(_ptr(dense), _ptr(self.storage), self.N, self.C, self.D, self.H, self.W, self.pd, _ovf_ptr(), _stream_ptr(self.device))
            _lib.check(st, "drc_rs16_from_dense")
        return self

    def from_blocked(self, blk, first_unit=0):
        """units [first_unit, first_unit + N) of a blocked fp32 tensor (or channel slice) of the same logical shape -> interior."""
        base = getattr(blk, "base", blk)
        if (blk.C, blk.D, blk.H, blk.W) != (self.C, self.D, self.H, self.W) or base.N < first_unit + self.N:
            raise ValueError("RS16.from_blocked: shapes differ")
        if self.N:
            src = C.c_void_p(base.storage.data_ptr() + 4 * first_unit * base.n_stride)
            st = _lib.lib().drc_rs16_from_blocked(src, _ptr(self.storage), self.N, self.C, self.D, self.H, self.W, base.pd, base.ph, base.pw, base.cb,
                                                  getattr(blk, "cb_off", 0), self.pd, _ovf_ptr(), _stream_ptr(self.device))
            _lib.check(st, "drc_rs16_from_blocked")
        return self

    def to_blocked(self, blk, first_unit=0):
        """interior -> units [first_unit, first_unit + N) of a blocked fp32 tensor (or channel slice) of the same logical shape."""
        base = getattr(blk, "base", blk)
        if (blk.C, blk.D, blk.H, blk.W) != (self.C, self.D, self.H, self.W) or base.N < first_unit + self.N:
            raise ValueError("RS16.to_blocked: shapes differ")
        if self.N:
            dst = C.c_void_p(base.storage.data_ptr() + 4 * first_unit * base.n_stride)
            st = _lib.lib().drc_rs16_to_blocked(_ptr(self.storage), dst, self.N, self.C, self.D, self.H, self.W, base.pd, base.ph, base.pw, base.cb,
                                                getattr(blk, "cb_off", 0), self.pd, _stream_ptr(self.device))
            _lib.check(st, "drc_rs16_to_blocked")
        return blk

    def to_dense(self):
        shape = (self.N, self.C, self.D, self.H, self.W)
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        if self.N:
            st = _lib.lib().drc_rs16_to_dense(_ptr(self.storage), _ptr(out), self.N, self.C, self.D, self.H, self.W, self.pd, _stream_ptr(self.device))
            _lib.check(st, "drc_rs16_to_dense")
        return out


def s16_supported(cin, cout, D, H, W, kind="s1", dil=1):
    """kind: "s1" stride-1 conv, "s2" stride-2 conv, "up" transposed conv (D, H, W = the layer's INPUT dims); "2d": the stride-1 3x3 conv on
    2D maps (D = 1)."""
    if kind == "2d":
        return bool(S16["enabled"] and D == 1 and _lib.lib().drc_conv2d_k3_s16_supported(cin, cout, H, W, dil))
    fn = {"s1": "drc_conv3d_k3_s16_supported", "s2": "drc_conv3d_k3s2_s16_supported", "up": "drc_deconv3d_k3s2_s16_supported"}[kind]
    return bool(S16["enabled"] and getattr(_lib.lib(), fn)(cin, cout, D, H, W))


class ConvPlanS16:
    """One launch of the split-f16 3x3x3 kernels on RS16 tensors (+BN, +residual, +ReLU): kind "s1" drc_conv3d_k3_s16_fwd (cv: the cost
    volume of the left / right 2D feature maps is the virtual input, reference stackhourglass.py:115-130), "s2" drc_conv3d_k3s2_s16_fwd,
    "up" drc_deconv3d_k3s2_s16_fwd; "2d": drc_conv2d_k3_s16_fwd on RS16 2D maps (D = 1, no depth halo; the 3x3 stride-1 layers of the feature
    CNN, reference submodule.py:9-16).  D, H, W = the INPUT dims."""

    def __init__(self, N, cin, cout, D, H, W, relu, cv=False, device=None, kind="s1", dil=1):
        if not s16_supported(cin, cout, D, H, W, kind, dil) or (cv and (cin != 64 or kind != "s1" or W <= 14)) or (dil != 1 and kind != "2d"):
            raise ValueError("ConvPlanS16: unsupported shape")
        self.dil = dil
        self.N, self.cin, self.cout, self.D, self.H, self.W, self.relu, self.cv, self.device, self.kind = N, cin, cout, D, H, W, bool(relu), cv, device, kind
        self.out_dhw = {"s1": (D, H, W), "s2": (D // 2, H // 2, W // 2), "up": (2 * D, 2 * H, 2 * W), "2d": (1, H, W)}[kind]
        self.pd = 0 if kind == "2d" else 1
        vox = D * H * W if kind != "s2" else (D // 2) * (H // 2) * (W // 2)
        self.flops = 2 * N * vox * (9 if kind == "2d" else 27) * cin * cout
        rt, wt = (1, 28)
        nw = {"s1": W, "s2": W // 2, "up": W, "2d": W}[kind]
        if nw <= 14:                       # (the kernels' dispatch: width <= 7 -> 4 x 7 tiles, <= 14 -> 2 x 14, else 1 x 28; the last tiles masked)
            rt, wt = (2, 14) if nw > 7 else (4, 7)
        if kind == "s1":
            self._kfmt = "convs16_kernel<%d,%s,%d,%d,%%s,%%s,%%s>" % (cin // 16, "true" if cv else "false", rt, wt)
            self.kname = self._kfmt % ("false", "false", "false")  # (the residual / blocked-fp32-output / fused-head template flags follow the call's arguments)
            # round 6: the cost-volume form runs two tiles per wave (convs16w.hip) where the launch has enough whole two-row blocks: the
            # library decides (drc_conv3d_k3_s16_wide), asked here with stand-in pointers
            from ._lib import DrcS16ConvParams
            probe = DrcS16ConvParams(None if cv else 1, 1, 1, 1, None, 1, None, 1 if cv else None, 1 if cv else None, N, D, H, W, cin, cout, int(bool(relu)), 0, 1)
            self._wide = bool(CV_WIDE["enabled"] and _lib.lib().drc_conv3d_k3_s16_wide(C.byref(probe)))
            if cv and not CV_WIDE["enabled"]:
                self.dil = 0x800               # (the library's experiment bit of the 3D layers' unused `dil` field: keep the one-tile kernel)
            self._wname = "convs16w_kernel<%d,%s>" % (cin // 16, "true" if cv else "false")
            if self._wide:
                self.kname = self._wname
        elif kind == "2d":
            # convs16r.hip's dispatch: (waves over K, K slices per wave)
            wide = cin == 64 and W % 56 == 0 and cout == 128 and N * (H // 28) * (W // 56) * (cout // 32) >= 256
            kw, ks = {32: (2, 1), 128: (4, 2)}.get(cin, (2, 2) if wide else (4, 1))
            self._kfmt = "convs16r_kernel<%d,%d,%%s,%d>" % (kw, ks, dil)
            self.kname = self._kfmt % "false"
        elif kind == "s2":
            # convs16d.hip's dispatch: cin 32 -> cout 64 runs the cout-split form (one slab, both cout tiles), slab rows de-interleaved; ring
            # depth by what fits the LDS
            cs = cin == 32 and cout == 64
            ring = 3 if (cs or (cin == 32 and nw == 14)) else 2
            self.kname = "convs16d_kernel<%d,%d,%d,%d,true,%s>" % (cin // 16, rt, wt, ring, "true" if cs else "false")
        else:
            self.kname = "convs16u_kernel<%d,%d>" % (rt, wt)

    def run(self, x16, w16, scale, shift, y16=None, y32=None, res=None, left=None, right=None, lo4=0, head=None):
        """head = (packed 32 -> 1 weights of s16.pack_head_weight_s16, S buffer fp32 of >= N*D*H*W*12 floats): the layer is classif[0] of a
        head, its output is not stored, the partial sums of the cout-1 layer behind it are (head_gather finishes it)."""
        from ._lib import DrcS16ConvParams
        if head is not None:
            if self.kind != "s1" or self.cv or (self.cin, self.cout) != (32, 32) or self.W % 28 or self.D < 6 or self.D % 3 or y16 is not None or y32 is not None or res is not None:
                raise ValueError("ConvPlanS16.run: the fused head is the 32 -> 32 full-resolution layer without another output")
            if head[1].dtype != torch.float32 or head[1].numel() < self.N * self.D * self.H * self.W * 12 or head[0].dtype != torch.float16 or head[0].numel() != 2048:
                raise ValueError("ConvPlanS16.run: head = (halfs [2][2][64][8], fp32 buffer of N*D*H*W*12)")
        if x16 is not None and (x16.N < self.N or (x16.C, x16.D, x16.H, x16.W, x16.pd) != (self.cin, self.D, self.H, self.W, self.pd)):
            raise ValueError("ConvPlanS16.run: input geometry differs from the plan")
        for t_ in (y16, res):
            if t_ is not None and (t_.N < self.N or (t_.C, t_.D, t_.H, t_.W, t_.pd) != (self.cout,) + self.out_dhw + (self.pd,)):
                raise ValueError("ConvPlanS16.run: output / residual geometry differs from the plan")
        if self.kind == "2d" and (y16 is None or y32 is not None or x16 is None):
            raise ValueError("ConvPlanS16.run: the 2D kernel reads and writes RS16 maps")
        if y32 is not None and (self.kind != "s1" or (y32.C, y32.D, y32.H, y32.W, y32.pd, y32.ph, y32.pw) != (self.cout, self.D, self.H, self.W, 1, 1, 1) or getattr(y32, "cb_off", 0)):
            raise ValueError("ConvPlanS16.run: the blocked fp32 output (stride-1 kernel only) must be a whole tensor with halo 1")
        if self.kind != "s1" and (y16 is None or (res is not None and self.kind == "s2")):
            raise ValueError("ConvPlanS16.run: the stride-2 / transposed kernels write RS16 (the stride-2 one takes no residual)")
        if self.cv:
            for f in (left, right):
                if f is None or (f.C, f.D, f.H, f.W, f.pd) != (32, 1, self.H, self.W, 0) or f.N < self.N:
                    raise ValueError("ConvPlanS16.run: the cost-volume variant reads two RS16 2D maps [N,32,H,W]")
        elif x16 is None:
            raise ValueError("ConvPlanS16.run: input missing")
        p = DrcS16ConvParams(_ptr(x16.storage) if x16 is not None else None, _ptr(w16), _ptr(scale), _ptr(shift),
                             _ptr(res.storage) if res is not None else None, _ptr(y16.storage) if y16 is not None else None,
                             _ptr(y32.storage) if y32 is not None else None, _ptr(left.storage) if self.cv else None,
                             _ptr(right.storage) if self.cv else None, self.N, self.D, self.H, self.W, self.cin, self.cout, int(self.relu), int(lo4), int(self.dil),
                             _ptr(head[1]) if head is not None else None, _ptr(head[0]) if head is not None else None, _ovf_ptr())
        dev = self.device
        if TIMING is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(dev))
        fn = {"s1": "drc_conv3d_k3_s16_fwd", "s2": "drc_conv3d_k3s2_s16_fwd", "up": "drc_deconv3d_k3s2_s16_fwd", "2d": "drc_conv2d_k3_s16_fwd"}[self.kind]
        st = getattr(_lib.lib(), fn)(C.byref(p), _stream_ptr(dev))
        _lib.check(st, fn)
        if TIMING is not None:
            e1.record(torch.cuda.current_stream(dev))
            if self.kind == "s1" and self._wide and res is None and y32 is None and head is None:
                kn = self._wname
            elif self.kind == "s1":
                kn = self._kfmt % ("true" if res is not None else "false", "true" if y32 is not None else "false", "true" if head is not None else "false")
            elif self.kind == "2d":
                kn = self._kfmt % ("true" if res is not None else "false")
            else:
                kn = self.kname
            # (a fused-head launch also computes the 32 -> 1 layer behind it: 2 * 27 * 32 flops per voxel on top of the layer's own)
            TIMING.append((kn, self.flops + (2 * 27 * 32 * self.N * self.D * self.H * self.W if head is not None else 0), e0, e1))
