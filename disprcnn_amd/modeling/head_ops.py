"""Engine-backed layers of the 2D stage's heads: nn.Conv2d / nn.Linear parameter holders whose arithmetic runs on the HIP engine
(convolutions: the same MFMA kernels as the backbone, bias in the folded-BN epilogue slot) or, for the fully connected layers, on the
hand-written fp32-MFMA GEMM of csrc/linear.hip (round 3; rounds 1-2: torch.addmm -> hipBLASLt).  Inference only; there is no CPU path."""
from collections import OrderedDict

import torch

from .. import engine as E


class EngineConv2d:
    """y = act(conv2d(x, conv.weight) + conv.bias) on dense [N,C,H,W] tensors.  Packed weights are rebuilt when the parameters
    change (version counters).  Storage is owned ONCE per map geometry (h, w) -- a short LRU: FPN levels, ROI windows -- and sized for a
    capacity bucket of the unit count (engine.bucket_units, like the PSMNet runtime's WorkspacePool): the mask head sees a different
    detection count (0..100) on every image, and a workspace per exact count would allocate and zero-fill two blocked tensors per conv and
    image.  A smaller count runs on prefix views of the same storage (units are outermost, the zero halo is never written);
    launch plans (no memory) are cached per count."""

    MAX_SHAPES = 12
    MAX_PLANS = 32

    def __init__(self, conv, relu):
        if conv.groups != 1 or conv.stride[0] != conv.stride[1] or conv.padding[0] != conv.padding[1] or conv.dilation[0] != conv.dilation[1]:
            raise NotImplementedError("EngineConv2d: square stride / padding / dilation, groups = 1")
        self.conv, self.relu = conv, bool(relu)
        self._ver, self._geo = None, OrderedDict()
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
            for geo in self._geo.values():
                geo["w16"] = {}
        return self._wp, self._sc, self._sh

    def nbytes(self):
        """HBM held by the cached workspaces (tests: flat over varying unit counts)."""
        return sum(4 * (g["x"].storage.numel() + g["y"].storage.numel()) for g in self._geo.values())

    def __call__(self, x):
        E.require_gpu(x, "EngineConv2d")
        c = self.conv
        n, cin, h, w = x.shape
        k, s, p, d = c.kernel_size[0], c.stride[0], c.padding[0], c.dilation[0]
        cout = c.out_channels
        dev = x.device
        oh, ow = (h + 2 * p - d * (k - 1) - 1) // s + 1, (w + 2 * p - d * (k - 1) - 1) // s + 1
        if n == 0:
            return x.new_zeros(0, cout, oh, ow)
        wp, sc, sh = self._weights(dev)
        key = (h, w, dev)
        halo = max(p, 1)
        geo = self._geo.get(key)
        if geo is None or geo["cap"] < n:
            cap = E.bucket_units(n)
            geo = dict(cap=cap, x=E.Blocked(cap, cin, 1, h, w, 0, halo, halo, dev), y=E.Blocked(cap, cout, 1, oh, ow, 0, 1, 1, dev),
                       plans=OrderedDict(), w16={})
            self._geo[key] = geo
            while len(self._geo) > self.MAX_SHAPES:
                self._geo.popitem(last=False)
        self._geo.move_to_end(key)
        xb = E.Blocked(n, cin, 1, h, w, 0, halo, halo, dev, storage=geo["x"].storage)
        yb = E.Blocked(n, cout, 1, oh, ow, 0, 1, 1, dev, storage=geo["y"].storage)
        plan = geo["plans"].get(n)
        if plan is None:
            plan = geo["plans"][n] = E.plan_conv2d(xb, yb, k, s, p, d, cout, self.relu)
            while len(geo["plans"]) > self.MAX_PLANS:
                geo["plans"].popitem(last=False)
        else:
            geo["plans"].move_to_end(n)
        kind = plan.pack_kind                          # plans of different counts may read different packings (Winograd vs direct)
        if kind not in geo["w16"]:
            geo["w16"][kind] = plan.pack16(self._w32)
        xb.from_dense(x.float())
        plan.run(xb, wp, sc, sh, yb, None, w16=geo["w16"][kind])
        return yb.to_dense()[:, :, 0]


class BlockedConv2d:
    """A convolution of the heads that stays in the engine's blocked layout: y_blocked = act(conv2d(x_blocked) + bias).  The weight is
    the channel-wise (dim 0) concatenation of one or more nn.Conv2d layers of equal geometry -- sibling 1x1 predictors read their
    shared input once -- packed when a parameter's version changes; launch plans are cached per geometry (they hold no memory)."""

    MAX_PLANS = 16

    def __init__(self, convs, relu):
        self.convs = list(convs) if isinstance(convs, (list, tuple)) else [convs]
        c = self.convs[0]
        for o in self.convs:
            if (o.kernel_size, o.stride, o.padding, o.dilation, o.in_channels, o.groups) != \
                    (c.kernel_size, c.stride, c.padding, c.dilation, c.in_channels, 1):
                raise NotImplementedError("BlockedConv2d: sibling layers must share their geometry (groups = 1)")
        self.k, self.stride, self.pad, self.dil = c.kernel_size[0], c.stride[0], c.padding[0], c.dilation[0]
        self.cout = sum(o.out_channels for o in self.convs)
        self.relu = bool(relu)
        self._ver, self._plans = None, OrderedDict()

    def _weights(self, dev):
        ver = tuple((o.weight._version, o.weight.data_ptr(), None if o.bias is None else o.bias._version) for o in self.convs) + (dev,)
        if ver != self._ver:
            w = torch.cat([o.weight.detach().to(device=dev, dtype=torch.float32) for o in self.convs], 0)
            cp = E.cout_pad_of(w.shape[0])
            self._wp = E.pack_conv_weight(w)
            self._sc = torch.ones(cp, device=dev)
            self._sh = torch.zeros(cp, device=dev)
            off = 0
            for o in self.convs:
                if o.bias is not None:
                    self._sh[off: off + o.out_channels] = o.bias.detach().to(device=dev, dtype=torch.float32)
                off += o.out_channels
            self._w32, self._ver, self._w16 = w, ver, {}
        return self._wp, self._sc, self._sh

    def out_hw(self, h, w):
        f = lambda n: (n + 2 * self.pad - self.dil * (self.k - 1) - 1) // self.stride + 1
        return f(h), f(w)

    def _bridge(self, xb, yb):
        """Large maps of a 3x3 stride-1 layer (the RPN head's 256 -> 512 convolution on P2..P4: 137 GFLOP of the 2D stage's heads) run on the
        split-f16 kernel between the same blocked tensors (engine.BridgedConv2dS16); None where that does not apply."""
        if (self.k, self.stride, self.pad, self.dil) != (3, 1, 1, 1) or type(xb) is not E.Blocked or type(yb) is not E.Blocked:
            return None
        if xb.n_stride != xb.cb * xb.cb_stride or yb.n_stride != yb.cb * yb.cb_stride or (yb.H, yb.W, xb.D, yb.D) != (xb.H, xb.W, 1, 1) or xb.N != yb.N:
            return None
        if not E.BridgedConv2dS16.worth(xb.N, xb.C, self.cout, xb.H, xb.W):
            return None
        if not hasattr(self, "_s16"):
            self._s16 = dict(cache={}, bridges=OrderedDict(), packs=None)
        key = (xb.N, xb.C, xb.H, xb.W)
        br = self._s16["bridges"].get(key)
        if br is None:
            br = self._s16["bridges"][key] = E.BridgedConv2dS16(xb.N, xb.C, self.cout, xb.H, xb.W, self.relu, xb.device, self._s16["cache"])
            while len(self._s16["bridges"]) > self.MAX_PLANS:
                (n_, c_, h_, w_), _ = self._s16["bridges"].popitem(last=False)
                for k in [k for k in self._s16["cache"] if (k[1], k[3], k[4]) == (n_, h_, w_)]:      # its RS16 maps (keyed (tag, N, ch, H, W)) go with it
                    del self._s16["cache"][k]
        return br

    def __call__(self, xb, yb):
        """(The split-f16 bridge reports to the range guard in scope -- the detector's, engine.guarded -- or, called on its own, to a guard
        of this layer: a value beyond +-65504 repeats the call on the fp32 kernel.)"""
        if E.guard_in_scope() is None and xb.device.type == "cuda":
            if getattr(self, "_guard", None) is None:
                self._guard = E.OverflowGuard(xb.device)
            return E.guarded(self._guard, lambda: self._call_once(xb, yb), what="BlockedConv2d (split-f16 3x3 layer)")
        return self._call_once(xb, yb)

    def _call_once(self, xb, yb):
        wp, sc, sh = self._weights(xb.device)
        br = self._bridge(xb, yb)
        if br is not None:
            from .. import s16 as S
            pk = self._s16["packs"]
            if pk is None or pk[0] is not self._w32 or pk[1] != br.bounds:
                packs = [S.pack_weight_s16(self._w32[:, a:b].contiguous()) for a, b in br.bounds]
                pk = self._s16["packs"] = (self._w32, br.bounds, [(w16, (sc * (2.0 ** -wexp)).contiguous()) for w16, wexp in packs], torch.zeros_like(sh))
            br.run(xb, pk[2], sh, pk[3], yb)
            return yb
        key = (xb.N, xb.C, xb.H, xb.W, xb.ph, xb.pw, xb.n_stride, yb.H, yb.W, yb.ph, yb.pw, yb.n_stride)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = E.plan_conv2d(xb, yb, self.k, self.stride, self.pad, self.dil, self.cout, self.relu)
            while len(self._plans) > self.MAX_PLANS:
                self._plans.popitem(last=False)
        kind = plan.pack_kind
        if kind not in self._w16:
            self._w16[kind] = plan.pack16(self._w32)
        plan.run(xb, wp, sc, sh, yb, None, w16=self._w16[kind])
        return yb


_PACKED_W = {}     # id(parameter) -> (weakref to the parameter, {tag: (version key, packed tensor, 2D shape)}).  The weak reference is the
                   # identity check (ids and allocator addresses are reused once a model is freed) and its callback drops the entry -- and the
                   # packed GPU copies with it -- when the parameter dies; nothing is stored on the parameter, so pickling a model stays lean.


def _packed_cache(prm):
    import weakref
    k = id(prm)
    ent = _PACKED_W.get(k)
    if ent is None or ent[0]() is not prm:
        ent = _PACKED_W[k] = (weakref.ref(prm, lambda _r, k=k: _PACKED_W.pop(k, None)), {})
    return ent[1]


def clear_packed_weights(module):
    """Drop the packed GEMM operands cached for the parameters of `module`.  They are rebuilt on the next call; needed only after an edit
    the version counter cannot see (`param.data.copy_(...)`, `param.data[...] = ...`) -- `load_state_dict`, optimizers and every in-place
    operation on the parameter itself bump `_version` and are picked up automatically."""
    for prm in module.parameters():
        ent = _PACKED_W.get(id(prm))
        if ent is not None and ent[0]() is prm:
            ent[1].clear()


def _pack_rows(a):
    from .. import _lib
    lib = _lib.lib()
    R, K = a.shape
    out = torch.empty(lib.drc_linear_packed_floats(R, K), dtype=torch.float32, device=a.device)
    st = lib.drc_linear_pack_rows(E._ptr(a), R, K, E._ptr(out), E._stream_ptr(a.device))
    _lib.check(st, "drc_linear_pack_rows")
    return out


def gemm_bias_act(x, weight_param, w2d_fn, bias, relu, tag=""):
    """act(x [M,K] @ W^T + bias) with W = w2d_fn(weight_param) [N,K] on drc_linear_fwd.  The packed form of W is cached per
    live parameter object (and tag) and rebuilt when the parameter's version, storage, device or shape changes (see clear_packed_weights)."""
    from .. import _lib
    E.require_gpu(x, "gemm_bias_act")
    dev = x.device
    key = (weight_param._version, weight_param.data_ptr(), dev, tuple(weight_param.shape))
    cache = _packed_cache(weight_param)
    ent = cache.get(tag)
    if ent is None or ent[0] != key:
        w2 = w2d_fn(weight_param.detach().to(device=dev, dtype=torch.float32)).contiguous()
        ent = cache[tag] = (key, _pack_rows(w2), tuple(w2.shape))
    wp, (N, Kw) = ent[1], ent[2]
    x = x.float().contiguous()
    M, K = x.shape
    if Kw != K:
        raise ValueError(f"gemm_bias_act: input has {K} features, the layer expects {Kw}")
    y = torch.empty(M, N, dtype=torch.float32, device=dev)
    if M == 0:
        return y
    b = None if bias is None else bias.detach().to(device=dev, dtype=torch.float32).contiguous()
    lib = _lib.lib()
    xp = _pack_rows(x)
    nscr = lib.drc_linear_scratch_floats(M, N, K)
    scr = E.scratch(dev, "linear", nscr) if nscr else None
    st = lib.drc_linear_fwd(E._ptr(xp), E._ptr(wp), E._ptr(b), E._ptr(y), M, N, K, int(bool(relu)), E._ptr(scr), nscr, E._stream_ptr(dev))
    _lib.check(st, "drc_linear_fwd")
    return y


def linear(x, layer, relu=False):
    """act(x [R, in] @ layer.weight^T + bias) on the hand-written fp32-MFMA GEMM (csrc/linear.hip: drc_linear_fwd); layer is an nn.Linear
    or an nn.Conv2d that acts on a full window (its weight flattened to [out, in], K contiguous).  The weights are packed into the GEMM's
    operand layout once per parameter version, the activations per call.  GPU only."""
    return gemm_bias_act(x, layer.weight, lambda w: w.reshape(w.shape[0], -1), layer.bias, relu)
