"""Execution of PSMNet on the HIP engine (eval mode: BatchNorm folded into the conv epilogues; train mode: per-GPU batch
statistics, a tape of conv+BN sites for the backward in train.py).

Graph restated from the reference (stackhourglass.py:106-174, submodule.py:106-139) as a flat
schedule of launches over blocked, zero-haloed tensors.  Workspaces and launch plans are cached
per input shape; halos are zeroed once at allocation and never written again.
"""
import operator
import weakref
from collections import OrderedDict

import torch

from ... import engine as E
from .submodule import SPP_BRANCHES, TRUNK_STAGES

GRAPH_MAX_UNITS = 96     # eval: batches up to this many ROI pairs replay a captured HIP graph (PSMNet.graph_eval = "auto")
GRAPH_MAX_ENTRIES = 16   # captured graphs kept per runtime (one per unit-count bucket and input geometry), least recently used dropped
GRAPH_CAPTURE_AFTER = 2  # a bucket is captured when it is seen for the second time (a count that never comes back costs one eager pass, not a capture)
MAX_SLOTS = 4            # train-mode forward passes of one geometry that may await their backward at once (each owns a workspace pool)
WS_MAX_PLANS = 48        # launch-plan sets kept per runtime (one per exact unit count; they hold views, not memory)


_UNIT_AFFINE = {}
_TENSOR_VERSION = operator.attrgetter("_version")
_TENSOR_PTR = torch.Tensor.data_ptr


def _unit_affine(n, device):
    """(ones[n], zeros[n]) on `device`, shared by every site that needs them (read-only)."""
    key = (n, device)
    v = _UNIT_AFFINE.get(key)
    if v is None:
        v = _UNIT_AFFINE[key] = (torch.ones(n, dtype=torch.float32, device=device), torch.zeros(n, dtype=torch.float32, device=device))
    return v


class _Conv:
    """Packed weights + folded BN of one conv(+bn) site."""

    def __init__(self, conv, bn, device, transposed=False):
        w = conv.weight.detach().to(device=device, dtype=torch.float32)
        self.conv, self.transposed = conv, transposed
        self.cout = w.shape[1] if transposed else w.shape[0]
        self.cin = w.shape[0] if transposed else w.shape[1]
        # the 3x3x3 Conv3d layers (stride 1 and 2) and the 3x3 Conv2d layers run on the LDS-free kernels, which read weights in
        # their own packing: both packings come out of one launch
        need16 = w.dim() == 5 or (w.dim() == 4 and tuple(w.shape[2:]) == (3, 3))
        if E.is_pointwise(w.shape, transposed):
            self.w, self.w16 = E.pack_weight_pw(w), None
        elif w.is_cuda:
            self.w, self.w16 = E.pack_layouts(w, transposed, want_t16=need16)
        else:
            self.w, self.w16 = E.pack_weight(w, transposed), (E.pack_weight_t16(w, transposed) if need16 else None)
        self._ww = None
        self._s16 = None                    # (split-f16 packing, wexp), built on first use (engine.ConvPlanS16)
        self._s16_scale = None              # (scale tensor it was derived from, scale * 2^-wexp)
        self._stem = None                   # firstconv[0] only: the packing of stemconv.hip, built on first use
        self._s16_parts = self._s16_parts_scale = None      # lastconv[0] only: split-f16 packings per input-channel slice (s16_cin_slices)
        cout_pad = E.cout_pad_of(self.cout)
        self.cout_pad, self.device = cout_pad, device
        self.bn = bn
        # constant vectors are shared per (length, device); the BatchNorm affine is read straight from the module's parameters when no
        # channel padding is needed (every layer of PSMNet): a train step re-builds these objects after each optimizer step, and six
        # tiny fill / copy launches per conv were ~700 launches (3 ms of GPU time, 10 ms of host time) per Config-B step
        self.unit_scale, self.zero_shift = _unit_affine(cout_pad, device)
        if bn is not None:
            direct = (cout_pad == self.cout and bn.weight.device == device and bn.weight.dtype == torch.float32 and bn.weight.is_contiguous()
                      and bn.bias.device == device and bn.bias.dtype == torch.float32 and bn.bias.is_contiguous())
            if direct:
                self.gamma, self.beta = bn.weight.detach(), bn.bias.detach()
            else:
                self.gamma = torch.zeros(cout_pad, dtype=torch.float32, device=device)
                self.beta = torch.zeros(cout_pad, dtype=torch.float32, device=device)
                self.gamma[: self.cout] = bn.weight.detach().to(device).float()
                self.beta[: self.cout] = bn.bias.detach().to(device).float()
            self.scale = self.shift = None          # eval-mode fold: computed on demand (refold), it depends on the running statistics
        else:
            self.scale, self.shift = self.unit_scale, self.zero_shift

    def w16_for(self, plan, kind=None):
        """The LDS-free packing `plan` reads: t16, or the Winograd-transformed weights (built on first use, one per packing kind:
        the same layer runs under different plans at different batch sizes).  kind = "wino": the packing of the fused cost-volume
        launch (wino3d_cv_kernel), whatever the plan's own kernel."""
        if kind is None and not (plan.wino or plan.deconv_direct):
            return self.w16
        kind = kind or plan.pack_kind
        if self._ww is None:
            self._ww = {}
        if kind not in self._ww:   # Winograd-transformed / transposed-conv-ordered packings come from the plan that reads them
            self._ww[kind] = plan.pack16(self.conv.weight.detach().to(device=self.device, dtype=torch.float32), self.transposed, kind=kind)
        return self._ww[kind]

    def s16(self):
        """(packed split-f16 weights, epilogue scale = folded BN scale * 2^-wexp) for drc_conv3d_k3_s16_fwd."""
        from ... import s16 as S
        if self._s16 is None:
            w = self.conv.weight.detach().to(device=self.device, dtype=torch.float32)
            self._s16 = S.pack_weight_s16(w.transpose(0, 1).contiguous() if self.transposed else w)     # ConvTranspose [Cin,Cout,..]: out = 2i - 1 + k, no flip
        if self._s16_scale is None or self._s16_scale[0] is not self.scale:
            self._s16_scale = (self.scale, (self.scale * (2.0 ** -self._s16[1])).contiguous())
        return self._s16[0], self._s16_scale[1]

    def s16_cin_slices(self, bounds):
        """[(packed split-f16 weights of the input channels [a, b), their epilogue scale = folded BN scale * 2^-wexp of the slice)] -- for a
        layer whose cin is not one the kernel takes (lastconv[0]: 320 = 64 + 128 + 128) and that runs as chained partial launches:
        BatchNorm is affine, scale * (c1 + c2 + c3) + shift = (scale * c1 + shift) + scale * c2 + scale * c3."""
        from ... import s16 as S
        key = tuple(bounds)
        if self._s16_parts is None or self._s16_parts[0] != key:
            w = self.conv.weight.detach().to(device=self.device, dtype=torch.float32)
            self._s16_parts = (key, [S.pack_weight_s16(w[:, a:b].contiguous()) for a, b in bounds])
            self._s16_parts_scale = None
        if self._s16_parts_scale is None or self._s16_parts_scale[0] is not self.scale:
            self._s16_parts_scale = (self.scale, [(self.scale * (2.0 ** -wexp)).contiguous() for _, wexp in self._s16_parts[1]])
        return [(wp, sc) for (wp, _), sc in zip(self._s16_parts[1], self._s16_parts_scale[1])]

    def refold(self):
        """Eval-mode BatchNorm folded into per-cout scale/shift from the module's current running statistics."""
        bn, device = self.bn, self.device
        if bn is not None:
            self.scale, self.shift = E.fold_bn(bn.weight.detach().to(device).float(), bn.bias.detach().to(device).float(),
                                               bn.running_mean.detach().to(device).float(),
                                               bn.running_var.detach().to(device).float(), bn.eps, self.cout_pad)


class _LazyTensors(dict):
    """Workspace tensors by name; entries of `lazy` are built (allocated) on first access."""

    def __init__(self):
        super().__init__()
        self.lazy = {}

    def __missing__(self, name):
        v = self[name] = self.lazy.pop(name)()
        return v

    def get(self, name, default=None):
        return self[name] if (name in self or name in self.lazy) else default


class PSMNetRuntime:
    def __init__(self, model, device):
        self.model = model
        self.device = device
        self._weights_version = None
        self._folds_version = None
        self._w = None
        self._ws = OrderedDict()   # workspaces: key (incl. the exact unit count) -> dict of tensor views/plans, LRU-capped
        self._pools = {}           # geometry key (no unit count) -> E.WorkspacePool owning the HBM, sized for the largest count seen
        self._training = False
        self._tape = None       # list of recorded ops while a differentiable train-mode forward runs
        self._held = {}         # workspace slot -> token of the differentiable forward whose saved activations live there
        self._slot = 0          # slot the running forward uses
        self._graphs = OrderedDict()    # eval HIP graphs: (unit-count bucket, key) -> dict(epoch, static inputs, GraphedStep, ...)
        self._graph_seen = {}           # key -> times seen before its capture
        self._epoch = 0         # advanced whenever what a captured graph points at is replaced (packed weights, BN folds, a workspace pool)
        self._guard = None      # engine.OverflowGuard of this model's split-f16 passes (built on the first eval forward)
        self._s16_used = False  # a split-f16 schedule was chosen during the running forward

    # ------------------------------------------------------------------ workspace slots (forwards awaiting their backward)
    def _pick_slot(self):
        """The lowest slot no pending backward depends on.  Slot 0 when nothing is pending -- eval passes and the usual
        forward/backward/step loop never touch another pool; a second forward before the first one's backward (gradient
        accumulation over several batches, an eval pass between forward and backward; reference: plain autograd,
        engine/trainer.py:101-115) gets a pool of its own."""
        for s in range(MAX_SLOTS):
            if s not in self._held:
                return s
        raise RuntimeError(f"PSMNet: {MAX_SLOTS} train-mode forward passes are waiting for their backward; run backward() (or drop "
                           f"their outputs) first, or raise disprcnn_amd.modeling.psmnet.runtime.MAX_SLOTS")

    def _slotted(self, key):
        return key if self._slot == 0 else key + (("slot", self._slot),)

    def _hold(self, slot):
        class _Token:
            pass
        tok = _Token()
        self._held[slot] = id(tok)
        weakref.finalize(tok, PSMNetRuntime._release, weakref.ref(self), slot, id(tok))      # outputs dropped without a backward
        return tok

    @staticmethod
    def _release(rt_ref, slot, tok_id):
        rt = rt_ref() if isinstance(rt_ref, weakref.ref) else rt_ref
        if rt is not None and rt._held.get(slot) == tok_id:
            del rt._held[slot]

    def invalidate(self):
        self._weights_version = None

    # ------------------------------------------------------------------ bounded workspaces
    def _pool_for(self, gkey, n):
        """The pool of one geometry, grown (replaced) when a unit count exceeds its capacity; workspaces that viewed the old
        pool are dropped with it, so HBM use is that of the largest count seen, not the sum over all counts seen."""
        pool = self._pools.get(gkey)
        if pool is None or pool.cap < n:
            if pool is not None:
                for k in [k for k, w in self._ws.items() if w.get("pool") is pool]:
                    del self._ws[k]
            pool = self._pools[gkey] = E.WorkspacePool(E.bucket_units(n), self.device)
            self._epoch += 1            # captured graphs point into the replaced pool
        return pool

    def _ws_get(self, key):
        ws = self._ws.get(key)
        if ws is not None:
            self._ws.move_to_end(key)
        return ws

    def _ws_put(self, key, ws):
        self._ws[key] = ws
        while len(self._ws) > WS_MAX_PLANS:
            self._ws.popitem(last=False)
        return ws

    def workspace_bytes(self):
        return sum(p.nbytes() for p in self._pools.values())

    @staticmethod
    def _stamp(*wss):
        """A forward pass is about to overwrite these workspaces' tensors: advance their pools' generation."""
        for ws in wss:
            ws["pool"].gen += 1

    @staticmethod
    def _generations(*wss):
        return [(ws["pool"], ws["pool"].gen) for ws in wss]

    @staticmethod
    def _check_generations(gens):
        for pool, gen in gens:
            if pool.gen != gen:
                raise RuntimeError(
                    "PSMNet backward: the workspace of this forward pass was reused by a later forward (another batch of the "
                    "same geometry, or an eval pass) before backward() ran, so the saved activations are gone (retain_graph / a "
                    "second backward through the same forward is not supported: run the forward again).")

    # ------------------------------------------------------------------ weights
    def _slots(self, kind):
        """[(owner dict, name, tensor)] of every parameter / buffer: model.parameters() walks ~300 modules (1 ms of host time per call, twice
        per forward); the slots are rebuilt only when a tensor OBJECT was replaced in its owner (checked by identity, ~15 us)."""
        cache = self.__dict__.setdefault("_slot_cache", {})
        sl = cache.get(kind)
        if sl is not None and all(d.get(n) is t for d, n, t in sl):
            return sl
        sl = []
        for mod in self.model.modules():
            d = mod._parameters if kind == "p" else mod._buffers
            sl.extend((d, n, t) for n, t in d.items() if t is not None)
        cache[kind] = sl
        return sl

    def _tensors(self, kind):
        """The tensors of _slots(kind) as a flat list (rebuilt with the slots)."""
        sl = self._slots(kind)
        cache = self.__dict__["_slot_cache"]
        ent = cache.get("t" + kind)
        if ent is None or ent[0] is not sl:
            ent = cache["t" + kind] = (sl, [t for _, _, t in sl])
        return ent[1]

    def _version(self):
        """(versions, storage addresses) of every parameter: an in-place update bumps the first, `.to()` / `.data = ...` changes the second.
        Built with C-level maps: this runs on every forward, in front of the first launch (115 us for the 259 parameters + 255 buffers where
        the comprehension form took 160)."""
        ts = self._tensors("p")
        return (list(map(_TENSOR_VERSION, ts)), list(map(_TENSOR_PTR, ts)))

    def _buffers_version(self):
        ts = self._tensors("b")
        return (list(map(_TENSOR_VERSION, ts)), list(map(_TENSOR_PTR, ts)))

    def _compile(self):
        """Packed weights are rebuilt when a PARAMETER changed; the eval-mode BatchNorm folds additionally when a buffer (running
        statistics) changed -- but only when an eval forward needs them: a train step updates the running statistics every
        time and must not trigger a re-pack."""
        v = self._version()
        if self._w is not None and v == self._weights_version:
            if not self._training:
                vb = self._buffers_version()
                if vb != self._folds_version:
                    for c in self._w.values():
                        if isinstance(c, _Conv):
                            c.refold()
                    self._folds_version = vb
                    self._epoch += 1
            return self._w
        m, dev = self.model, self.device
        W = {}

        def cb3(name, seq, transposed=False):
            W[name] = _Conv(seq[0], seq[1], dev, transposed)

        cb3("dres0.0", m.dres0[0]); cb3("dres0.2", m.dres0[2])
        cb3("dres1.0", m.dres1[0]); cb3("dres1.2", m.dres1[2])
        for hg in ("dres2", "dres3", "dres4"):
            h = getattr(m, hg)
            cb3(hg + ".conv1", h.conv1[0]); cb3(hg + ".conv2", h.conv2); cb3(hg + ".conv3", h.conv3[0])
            cb3(hg + ".conv4", h.conv4[0]); cb3(hg + ".conv5", h.conv5, True); cb3(hg + ".conv6", h.conv6, True)
        for c in ("classif1", "classif2", "classif3"):
            seq = getattr(m, c)
            cb3(c + ".0", seq[0])
            W[c + ".2"] = E.pack_weight_cout1(seq[2].weight.detach().to(device=dev, dtype=torch.float32))
        fe = m.feature_extraction
        for i in (0, 2, 4):
            cb3(f"fe.firstconv.{i}", fe.firstconv[i])
        for name, planes, nblk, stride, dil in TRUNK_STAGES:
            for b, unit in enumerate(getattr(fe, name)):
                cb3(f"fe.{name}.{b}.conv1", unit.conv1[0]); cb3(f"fe.{name}.{b}.conv2", unit.conv2)
                if unit.downsample is not None:
                    cb3(f"fe.{name}.{b}.down", unit.downsample)
        for name, _ in SPP_BRANCHES:
            cb3(f"fe.{name}", getattr(fe, name)[1])
        cb3("fe.lastconv.0", fe.lastconv[0])
        W["fe.lastconv.2"] = _Conv(fe.lastconv[2], None, dev)
        self._w, self._weights_version, self._folds_version = W, v, None
        self._epoch += 1
        if not self._training:
            for c in W.values():
                if isinstance(c, _Conv):
                    c.refold()
            self._folds_version = self._buffers_version()
        return W

    # ------------------------------------------------------------------ one conv(+BN)(+res)(+ReLU) site
    def _site(self, ws, W, plan, wname, x, y, res=None):
        """Eval: BN folded into the conv epilogue.  Train: conv -> per-GPU batch statistics (biased var, eps 1e-5) ->
        normalise + affine (+res) (+ReLU); running statistics updated like nn.BatchNorm (momentum, unbiased var).
        Reference: submodule.py:13-22 with the activations / adds of stackhourglass.py:32-51,130-144."""
        t, p = ws["t"], ws["p"]
        c = W[wname]
        pl = p[plan]
        rs = t[res] if res else None
        if not self._training or c.bn is None:
            pl.run(t[x], c.w, c.scale, c.shift, t[y], rs, w16=c.w16_for(pl))
            if self._training and self._tape is not None:        # BN-less conv (lastconv.2): still a site of the reverse pass
                self._tape.append(("site", ws, plan, wname, x, y, res))
            return
        yt = t[y]
        raw = ws.setdefault("raw", {}).get(plan)
        if raw is None:
            raw = ws["pool"].blocked(("raw", plan), yt.N, yt.C, yt.D, yt.H, yt.W, yt.pd, yt.ph, yt.pw)
            ws["raw"][plan] = raw
        pl.run(t[x], c.w, c.unit_scale, c.zero_shift, raw, None, relu=False, w16=c.w16_for(pl))
        bn = c.bn
        fused = (bn.momentum is not None and bn.running_mean is not None and bn.running_mean.device == raw.device and
                 bn.running_mean.dtype == torch.float32 and bn.running_mean.is_contiguous() and bn.running_var.is_contiguous())
        if fused:
            stats, M = E.bn_batch_stats_raw(raw)
            mean = stats[0]
            invstd = E.bn_finalize(stats, M, bn, c.cout)          # + running statistics (buffers live on the module), one launch
        else:
            mean, var, M = E.bn_batch_stats(raw)
            invstd = torch.rsqrt(var + bn.eps)
            with torch.no_grad():
                bn.num_batches_tracked += 1
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                unb = var[: c.cout] * (M / max(M - 1, 1))
                bn.running_mean.mul_(1 - mom).add_(mean[: c.cout].to(bn.running_mean.device), alpha=mom)
                bn.running_var.mul_(1 - mom).add_(unb.to(bn.running_var.device), alpha=mom)
        relu = bool(pl.p.relu)
        E.bn_apply(raw, yt, rs, mean, invstd, c.gamma, c.beta, relu)     # yt may be a concat slice (geometry carries cb_off)
        ws.setdefault("saved", {})[plan] = (mean, invstd, M)
        if self._tape is not None:
            self._tape.append(("site", ws, plan, wname, x, y, res))

    # ------------------------------------------------------------------ 3D regressor
    def _ws3d(self, N, Dp, Hp, Wp):
        key = self._slotted(("3d", N, Dp, Hp, Wp))
        ws = self._ws_get(key)
        if ws is not None:
            return ws
        pool = self._pool_for(self._slotted(("3d", Dp, Hp, Wp)), N)
        full = (Dp, Hp, Wp)
        half = tuple(-(-s // 2) for s in full)
        quart = tuple(-(-s // 2) for s in half)
        if tuple(2 * s for s in half) != full or tuple(2 * s for s in quart) != half:
            raise ValueError(f"cost-volume dims {full} must be divisible by 4 (D,H,W multiples of 16; SURVEY 8)")
        t = _LazyTensors()

        def B(name, c, d, h, w):
            t[name] = pool.blocked(name, N, c, d, h, w, 1, 1, 1)

        # the 64-channel volume is the largest tensor of the stage and the fused eval path (wino3d_cv_kernel) never touches it: it is
        # allocated (and zero-filled) on first use -- training, the non-fused path -- and planned from its geometry alone
        t.lazy["cost"] = lambda: pool.blocked("cost", N, 64, *full, 1, 1, 1)
        cost_geom = E.Blocked.geometry(N, 64, *full, 1, 1, 1, self.device)
        for n in ("d0a", "cost0a", "d1a", "cost0", "out1", "out2", "out3", "cls_t1", "cls_t2", "cls_t3"):
            B(n, 32, *full)
        for k in (1, 2, 3):
            B(f"hg{k}.c1", 64, *half); B(f"hg{k}.pre", 64, *half); B(f"hg{k}.post", 64, *half)
            B(f"hg{k}.c3", 64, *quart); B(f"hg{k}.c4", 64, *quart)
        for k in (1, 2, 3):
            t[f"costk{k}"] = pool.dense(f"costk{k}", N, *full)
        p = {}
        p["dres0.0"] = E.plan_conv3d(cost_geom, t["d0a"], 1, 32, True)
        p["dres0.2"] = E.plan_conv3d(t["d0a"], t["cost0a"], 1, 32, True)
        p["dres1.0"] = E.plan_conv3d(t["cost0a"], t["d1a"], 1, 32, True)
        p["dres1.2"] = E.plan_conv3d(t["d1a"], t["cost0"], 1, 32, False)
        for k in (1, 2, 3):
            src = t["cost0"] if k == 1 else t[f"out{k - 1}"]
            p[f"hg{k}.conv1"] = E.plan_conv3d(src, t[f"hg{k}.c1"], 2, 64, True)
            p[f"hg{k}.conv2"] = E.plan_conv3d(t[f"hg{k}.c1"], t[f"hg{k}.pre"], 1, 64, True)
            p[f"hg{k}.conv3"] = E.plan_conv3d(t[f"hg{k}.pre"], t[f"hg{k}.c3"], 2, 64, True)
            p[f"hg{k}.conv4"] = E.plan_conv3d(t[f"hg{k}.c3"], t[f"hg{k}.c4"], 1, 64, True)
            p[f"hg{k}.conv5"] = E.plan_deconv3d(t[f"hg{k}.c4"], t[f"hg{k}.post"], 64, True)
            p[f"hg{k}.conv6"] = E.plan_deconv3d(t[f"hg{k}.post"], t[f"out{k}"], 32, False)
            p[f"classif{k}.0"] = E.plan_conv3d(t[f"out{k}"], t[f"cls_t{k}"], 1, 32, True)
        ws = dict(t=t, p=p, pool=pool, flops=sum(pl.flops for pl in p.values()) + 3 * 2 * 27 * 32 * N * Dp * Hp * Wp)
        return self._ws_put(key, ws)

    def _fuse_costvol(self, ws, training):
        """Eval with a Winograd dres0[0]: the cost volume is folded into that layer's input loads (wino3d_cv_kernel)."""
        return bool(not training and self._tape is None and E.WINO["fuse_costvol"] and ws["p"]["dres0.0"].wino)

    def _regress(self, ws, W, cv=None):
        """dres0..dres4 + classif heads on ws['t']['cost'] -> dense cost3 [N,D',H',W'] (reference :130-144).
        cv = (left, right, lo4): dres0[0] reads the blocked feature maps instead of a materialised volume (:115-130)."""
        t, p = ws["t"], ws["p"]

        def run(plan, wname, x, y, res=None):
            self._site(ws, W, plan, wname, x, y, res)

        if cv is not None:
            c, pl = W["dres0.0"], p["dres0.0"]
            pl.run_costvol(cv[0], cv[1], cv[2], c.w16_for(pl), c.scale, c.shift, t["d0a"])
        else:
            run("dres0.0", "dres0.0", "cost", "d0a")
        run("dres0.2", "dres0.2", "d0a", "cost0a")
        run("dres1.0", "dres1.0", "cost0a", "d1a")
        run("dres1.2", "dres1.2", "d1a", "cost0", res="cost0a")           # dres1(cost0)+cost0, no relu
        for k, hg in ((1, "dres2"), (2, "dres3"), (3, "dres4")):
            src = "cost0" if k == 1 else f"out{k - 1}"
            postsqu = None if k == 1 else f"hg{k - 1}.post"                 # dres3(.., post1), dres4(.., post2)
            presqu = f"hg{k}.pre" if k == 1 else "hg1.pre"                  # pre1 reused by dres3 AND dres4 (:136,:139)
            run(f"hg{k}.conv1", hg + ".conv1", src, f"hg{k}.c1")
            run(f"hg{k}.conv2", hg + ".conv2", f"hg{k}.c1", f"hg{k}.pre", res=postsqu)   # relu(conv2 + postsqu)
            run(f"hg{k}.conv3", hg + ".conv3", f"hg{k}.pre", f"hg{k}.c3")
            run(f"hg{k}.conv4", hg + ".conv4", f"hg{k}.c3", f"hg{k}.c4")
            run(f"hg{k}.conv5", hg + ".conv5", f"hg{k}.c4", f"hg{k}.post", res=presqu)   # relu(conv5 + presqu|pre)
            run(f"hg{k}.conv6", hg + ".conv6", f"hg{k}.post", f"out{k}", res="cost0")    # out_k = conv6 + cost0
        prev = None
        for k in (1, 2, 3):
            run(f"classif{k}.0", f"classif{k}.0", f"out{k}", f"cls_t{k}")
            E.conv3d_cout1(t[f"cls_t{k}"], W[f"classif{k}.2"], prev, t[f"costk{k}"])          # cumulative heads
            prev = t[f"costk{k}"]
        return t["costk1"], t["costk2"], t["costk3"]

    # ------------------------------------------------------------------ split-f16 regressor (round 5; eval only)
    @staticmethod
    def _s16_layers(Dp, Hp, Wp):
        """(kind, cin, cout, input dims) of every 3x3x3 layer of the regressor but the three cout-1 heads."""
        full = (Dp, Hp, Wp)
        half = tuple(s // 2 for s in full)
        quart = tuple(s // 2 for s in half)
        return [("s1", 64, 32, full), ("s1", 32, 32, full), ("s2", 32, 64, full), ("s1", 64, 64, half), ("s2", 64, 64, half),
                ("s1", 64, 64, quart), ("up", 64, 64, quart), ("up", 64, 32, half)]

    def _use_s16(self, training, Dp, Hp, Wp):
        """Eval: every 3x3x3 layer of the regressor runs in split-f16 arithmetic on the f16 matrix cores (convs16*.hip: fp32-class results,
        DESIGN 3.7) when the volume's shape is one the kernels take -- the three cout-1 heads fused into the layer in front of them where the
        map is a multiple of 28 columns (_ws3d_s16), else on cout1_mfma.hip; PSMNet.regressor_math = "f32" keeps the fp32 MFMA kernels of
        rounds 1-4."""
        mode = getattr(self.model, "regressor_math", "auto")
        if mode not in ("auto", "f32", "f16x2"):
            raise ValueError("PSMNet.regressor_math must be 'auto', 'f32' or 'f16x2'")
        ok = (not training and self._tape is None and Dp % 4 == 0 and Hp % 4 == 0 and Wp % 4 == 0 and Wp > 14 and      # (Wp > 14: the cost-volume layer's 28-wide tiles)
              all(E.s16_supported(ci, co, *dims, kind=kind) for kind, ci, co, dims in self._s16_layers(Dp, Hp, Wp)))
        if mode == "f16x2" and not ok:
            raise RuntimeError("PSMNet.regressor_math = 'f16x2': eval only; volume dims D', H', W' multiples of 4, W' >= 16 (the cost-volume layer runs on 28-wide tiles)")
        return self._s16_choice(ok and mode != "f32", mode, "regressor_math")

    def _s16_choice(self, use, mode, what):
        """A split-f16 schedule is taken only while the range guard allows it (engine.guarded: the repeat of a pass that overflowed runs on
        the fp32 kernels); an explicit "f16x2" cannot be repeated in fp32 -- it raises."""
        if use and not E.s16_allowed():
            if mode == "f16x2":
                raise RuntimeError(f"PSMNet.{what} = 'f16x2': a value left the split-f16 range (|v| > 65504) in an enclosing guarded pass")
            return False
        if use:
            self._s16_used = True
        return use

    def _ws3d_s16(self, N, Dp, Hp, Wp):
        key = self._slotted(("3ds16", N, Dp, Hp, Wp))
        ws = self._ws_get(key)
        if ws is not None:
            return ws
        pool = self._pool_for(self._slotted(("3ds16", Dp, Hp, Wp)), N)
        full = (Dp, Hp, Wp)
        half = tuple(s // 2 for s in full)
        quart = tuple(s // 2 for s in half)
        t = {}

        def S(name, c, d, h, w, pd=1):
            t[name] = pool.rs16(name, N, c, d, h, w, pd)

        S("featL", 32, 1, Hp, Wp, 0); S("featR", 32, 1, Hp, Wp, 0)
        for n in ("d0a", "cost0a", "d1a", "cost0", "out1", "out2", "out3"):
            S(n, 32, *full)
        for k in (1, 2, 3):
            S(f"hg{k}.c1", 64, *half); S(f"hg{k}.pre", 64, *half); S(f"hg{k}.post", 64, *half)
            S(f"hg{k}.c3", 64, *quart); S(f"hg{k}.c4", 64, *quart)
            t[f"costk{k}"] = pool.dense(f"costk{k}", N, *full)
        # the heads: classif[0] + the 32 -> 1 layer fused (convs16.hip HEAD form: partial sums S, 48 B per voxel, one buffer for the three
        # heads) where the map is a multiple of 28 columns; else classif[0] writes a blocked fp32 tensor that cout1_mfma.hip reads
        fused = E.HEAD_FUSED["enabled"] and Wp % 28 == 0 and Dp >= 6 and Dp % 3 == 0
        if fused:
            t["hs"] = pool.dense("hs", N, Dp, Hp, Wp, 12)
        else:
            for k in (1, 2, 3):
                t[f"cls_t{k}"] = pool.blocked(f"cls_t{k}", N, 32, *full, 1, 1, 1)    # fp32: the cout-1 head reads it
        dev = self.device
        P = lambda kind, ci, co, dims, relu, cv=False: E.ConvPlanS16(N, ci, co, *dims, relu, cv=cv, device=dev, kind=kind)
        p = {}
        p["dres0.0"] = P("s1", 64, 32, full, True, cv=True)
        p["dres0.2"] = P("s1", 32, 32, full, True)
        p["dres1.0"] = P("s1", 32, 32, full, True)
        p["dres1.2"] = P("s1", 32, 32, full, False)
        for k in (1, 2, 3):
            p[f"hg{k}.conv1"] = P("s2", 32, 64, full, True)
            p[f"hg{k}.conv2"] = P("s1", 64, 64, half, True)
            p[f"hg{k}.conv3"] = P("s2", 64, 64, half, True)
            p[f"hg{k}.conv4"] = P("s1", 64, 64, quart, True)
            p[f"hg{k}.conv5"] = P("up", 64, 64, quart, True)
            p[f"hg{k}.conv6"] = P("up", 64, 32, half, False)
            p[f"classif{k}.0"] = P("s1", 32, 32, full, True)
        ws = dict(t=t, p=p, pool=pool, fused_heads=fused, flops=sum(pl.flops for pl in p.values()) + 3 * 2 * 27 * 32 * N * Dp * Hp * Wp)
        return self._ws_put(key, ws)

    def _head_weights_s16(self):
        """{k: (packed 32 -> 1 weights of the fused head, 2^-wexp)} for classif1..3[2], rebuilt with the packed fp32 weights."""
        if getattr(self, "_h16_version", None) == self._weights_version and getattr(self, "_h16", None) is not None:
            return self._h16
        from ... import s16 as S
        out = {}
        for k in (1, 2, 3):
            wp, wexp = S.pack_head_weight_s16(getattr(self.model, f"classif{k}")[2].weight)
            out[k] = (wp.to(self.device), 2.0 ** -wexp)
        self._h16, self._h16_version = out, self._weights_version
        return out

    def _regress_s16(self, ws, W, lo4):
        """The schedule of _regress on RS16 tensors, every layer but the cout-1 heads in split-f16 arithmetic; ws['t']['featL'/'featR'] hold
        the feature maps (RS16 2D).  Reference: stackhourglass.py:115-144."""
        t, p = ws["t"], ws["p"]

        def run(plan, wname, x, y16=None, y32=None, res=None, **kw):
            w16, sc16 = W[wname].s16()
            p[plan].run(t[x] if x else None, w16, sc16, W[wname].shift, y16=t[y16] if y16 else None, y32=t[y32] if y32 else None,
                        res=t[res] if res else None, **kw)

        run("dres0.0", "dres0.0", None, y16="d0a", left=t["featL"], right=t["featR"], lo4=lo4)
        run("dres0.2", "dres0.2", "d0a", y16="cost0a")
        run("dres1.0", "dres1.0", "cost0a", y16="d1a")
        run("dres1.2", "dres1.2", "d1a", y16="cost0", res="cost0a")             # dres1(cost0)+cost0, no relu
        for k, hg in ((1, "dres2"), (2, "dres3"), (3, "dres4")):
            src = "cost0" if k == 1 else f"out{k - 1}"
            postsqu = None if k == 1 else f"hg{k - 1}.post"                      # dres3(.., post1), dres4(.., post2)
            presqu = f"hg{k}.pre" if k == 1 else "hg1.pre"                       # pre1 reused by dres3 AND dres4 (:136,:139)
            run(f"hg{k}.conv1", hg + ".conv1", src, y16=f"hg{k}.c1")
            run(f"hg{k}.conv2", hg + ".conv2", f"hg{k}.c1", y16=f"hg{k}.pre", res=postsqu)      # relu(conv2 + postsqu)
            run(f"hg{k}.conv3", hg + ".conv3", f"hg{k}.pre", y16=f"hg{k}.c3")
            run(f"hg{k}.conv4", hg + ".conv4", f"hg{k}.c3", y16=f"hg{k}.c4")
            run(f"hg{k}.conv5", hg + ".conv5", f"hg{k}.c4", y16=f"hg{k}.post", res=presqu)      # relu(conv5 + presqu|pre)
            run(f"hg{k}.conv6", hg + ".conv6", f"hg{k}.post", y16=f"out{k}", res="cost0")       # out_k = conv6 + cost0
        prev = None
        h16 = self._head_weights_s16() if ws["fused_heads"] else None
        for k in (1, 2, 3):
            if h16 is not None:
                run(f"classif{k}.0", f"classif{k}.0", f"out{k}", head=(h16[k][0], t["hs"]))
                E.head_gather(t["hs"], h16[k][1], prev, t[f"costk{k}"])                         # cumulative heads
            else:
                run(f"classif{k}.0", f"classif{k}.0", f"out{k}", y32=f"cls_t{k}")
                E.conv3d_cout1(t[f"cls_t{k}"], W[f"classif{k}.2"], prev, t[f"costk{k}"])
            prev = t[f"costk{k}"]
        return t["costk1"], t["costk2"], t["costk3"]

    # ------------------------------------------------------------------ fp16-storage regressor (BASELINE configs[3]; eval only)
    def _weights16(self, W):
        """fp16 packings of the regressor's weights (conv16.hip), rebuilt with the packed fp32 weights."""
        if getattr(self, "_w16_version", None) == self._weights_version and getattr(self, "_w16", None) is not None:
            return self._w16
        m, dev = self.model, self.device
        out = {}
        for name, c in W.items():
            if isinstance(c, _Conv):            # (the 2D CNN's weights: used when PSMNet.feature_storage == "f16")
                out[name] = E.pack_weight16(c.conv.weight.detach().to(device=dev, dtype=torch.float32), c.transposed)
        for k in (1, 2, 3):
            out[f"classif{k}.2"] = E.pack_weight16(getattr(m, f"classif{k}")[2].weight.detach().to(device=dev, dtype=torch.float32))
        self._w16, self._w16_version = out, self._weights_version
        return out

    def _ws3d16(self, N, Dp, Hp, Wp):
        key = self._slotted(("3d16", N, Dp, Hp, Wp))
        ws = self._ws_get(key)
        if ws is not None:
            return ws
        pool = self._pool_for(self._slotted(("3d16", Dp, Hp, Wp)), N)
        full = (Dp, Hp, Wp)
        half = tuple(-(-s // 2) for s in full)
        quart = tuple(-(-s // 2) for s in half)
        if tuple(2 * s for s in half) != full or tuple(2 * s for s in quart) != half:
            raise ValueError(f"cost-volume dims {full} must be divisible by 4 (D,H,W multiples of 16; SURVEY 8)")
        t = {}

        def B(name, c, d, h, w):
            t[name] = pool.blocked16(name, N, c, d, h, w, 1, 1, 1)

        fused_cv = E.COSTVOL16_FUSED["enabled"]
        if fused_cv:
            B("pair", 64, 1, Hp, Wp)            # left | right features, one slice: dres0[0] builds the volume's rows from it (conv16x.hip)
        else:
            B("cost", 64, *full)
        for n in ("d0a", "cost0a", "d1a", "cost0", "out1", "out2", "out3", "cls_t1", "cls_t2", "cls_t3"):
            B(n, 32, *full)
        for k in (1, 2, 3):
            B(f"hg{k}.c1", 64, *half); B(f"hg{k}.pre", 64, *half); B(f"hg{k}.post", 64, *half)
            B(f"hg{k}.c3", 64, *quart); B(f"hg{k}.c4", 64, *quart)
            t[f"costk{k}"] = pool.dense(f"costk{k}", N, *full)
        p = {}
        p["dres0.0"] = (E.plan_conv3d16_costvol(t["pair"], t["d0a"], 0, 32, True) if fused_cv        # (first disparity: set per call, _costvol16)
                        else E.plan_conv3d16(t["cost"], t["d0a"], 1, 32, True))
        p["dres0.2"] = E.plan_conv3d16(t["d0a"], t["cost0a"], 1, 32, True)
        p["dres1.0"] = E.plan_conv3d16(t["cost0a"], t["d1a"], 1, 32, True)
        p["dres1.2"] = E.plan_conv3d16(t["d1a"], t["cost0"], 1, 32, False)
        for k in (1, 2, 3):
            src = t["cost0"] if k == 1 else t[f"out{k - 1}"]
            p[f"hg{k}.conv1"] = E.plan_conv3d16(src, t[f"hg{k}.c1"], 2, 64, True)
            p[f"hg{k}.conv2"] = E.plan_conv3d16(t[f"hg{k}.c1"], t[f"hg{k}.pre"], 1, 64, True)
            p[f"hg{k}.conv3"] = E.plan_conv3d16(t[f"hg{k}.pre"], t[f"hg{k}.c3"], 2, 64, True)
            p[f"hg{k}.conv4"] = E.plan_conv3d16(t[f"hg{k}.c3"], t[f"hg{k}.c4"], 1, 64, True)
            p[f"hg{k}.conv5"] = E.plan_deconv3d16(t[f"hg{k}.c4"], t[f"hg{k}.post"], 64, True)
            p[f"hg{k}.conv6"] = E.plan_deconv3d16(t[f"hg{k}.post"], t[f"out{k}"], 32, False)
            p[f"classif{k}.0"] = E.plan_conv3d16(t[f"out{k}"], t[f"cls_t{k}"], 1, 32, True)
            p[f"classif{k}.2"] = E.plan_conv3d16_cout1(t[f"cls_t{k}"])
        ws = dict(t=t, p=p, pool=pool, flops=sum(pl.flops for pl in p.values()))
        return self._ws_put(key, ws)

    def _costvol16(self, ws, mn, mx, left=None, right=None, in_pad=-1, feat16=None, right_first=0):
        """The fp16 cost volume of stackhourglass.py:115-128 from fp32 features (NCHW, or blocked 2D storage with halo in_pad) or from a
        Blocked16 feature tensor -- or, fused (engine.COSTVOL16_FUSED), only the one-slice feature pair dres0[0] builds the volume's rows from."""
        t = ws["t"]
        dst, lo, hi = (t["pair"], 0, 1) if "pair" in t else (t["cost"], mn // 4, mx // 4)
        if "pair" in t:
            ws["p"]["dres0.0"].costvol_lo4 = mn // 4
        if feat16 is not None:
            E.cost_volume16_from16(feat16, right_first, dst, lo, hi)
        else:
            E.cost_volume16_blocked(left, right, dst, lo, hi, in_pad)

    def _regress16(self, ws, W):
        """The schedule of _regress on fp16-storage tensors (eval: BatchNorm folded into the fp32 epilogue)."""
        t, p = ws["t"], ws["p"]
        W16 = self._weights16(W)

        def run(plan, wname, x, y, res=None):
            c = W[wname]
            p[plan].run(t[x], W16[wname], c.scale, c.shift, t[y], t[res] if res else None)

        run("dres0.0", "dres0.0", "pair" if "pair" in t else "cost", "d0a")
        run("dres0.2", "dres0.2", "d0a", "cost0a")
        run("dres1.0", "dres1.0", "cost0a", "d1a")
        run("dres1.2", "dres1.2", "d1a", "cost0", res="cost0a")
        for k, hg in ((1, "dres2"), (2, "dres3"), (3, "dres4")):
            src = "cost0" if k == 1 else f"out{k - 1}"
            postsqu = None if k == 1 else f"hg{k - 1}.post"
            presqu = f"hg{k}.pre" if k == 1 else "hg1.pre"
            run(f"hg{k}.conv1", hg + ".conv1", src, f"hg{k}.c1")
            run(f"hg{k}.conv2", hg + ".conv2", f"hg{k}.c1", f"hg{k}.pre", res=postsqu)
            run(f"hg{k}.conv3", hg + ".conv3", f"hg{k}.pre", f"hg{k}.c3")
            run(f"hg{k}.conv4", hg + ".conv4", f"hg{k}.c3", f"hg{k}.c4")
            run(f"hg{k}.conv5", hg + ".conv5", f"hg{k}.c4", f"hg{k}.post", res=presqu)
            run(f"hg{k}.conv6", hg + ".conv6", f"hg{k}.post", f"out{k}", res="cost0")
        prev = None
        for k in (1, 2, 3):
            run(f"classif{k}.0", f"classif{k}.0", f"out{k}", f"cls_t{k}")
            p[f"classif{k}.2"].run(t[f"cls_t{k}"], W16[f"classif{k}.2"], None, None, t[f"costk{k}"], prev)
            prev = t[f"costk{k}"]
        return t["costk1"], t["costk2"], t["costk3"]

    def _use_f16(self, training):
        mode = getattr(self.model, "regressor_storage", "f32")
        if mode not in ("f32", "f16"):
            raise ValueError("PSMNet.regressor_storage must be 'f32' or 'f16'")
        if mode == "f16" and training:
            raise RuntimeError("the fp16-storage regressor is an inference path (BASELINE configs[3]); train in fp32 like the reference")
        return mode == "f16"

    def _features_f16(self):
        mode = getattr(self.model, "feature_storage", "f32")
        if mode not in ("f32", "f16"):
            raise ValueError("PSMNet.feature_storage must be 'f32' or 'f16'")
        return mode == "f16"

    def _check_disp(self):
        mx, mn = self.model.maxdisp, self.model.mindisp
        if (mx - mn) % 16 != 0 or mx % 4 != 0 or mn % 4 != 0:
            raise ValueError("maxdisp/mindisp must be multiples of 4 with a range divisible by 16 (SURVEY 8, a1)")
        return mx, mn

    def _heads(self, costs, N, H, W, mx, mn, training):
        """eval: disparity from cost3; train: the three heads (stackhourglass.py:145-174)."""
        outs = []
        for c in (costs if training else costs[2:]):
            d = torch.empty(N, H, W, dtype=torch.float32, device=self.device)
            E.upsample_softargmin(c, d, mx, mn)
            outs.append(d)
        return tuple(outs) if training else outs[0]

    # ------------------------------------------------------------------ eval as a replayed HIP graph (small batches are launch-bound)
    def _graph_mode(self, n_units, training):
        """PSMNet.graph_eval: "auto" (default) -- eval batches of at most GRAPH_MAX_UNITS units replay a captured HIP graph (one image's 16 ROIs
        are ~45-150 launches of a few us each: the eager step is bound by the host, 1.47 ms vs 0.70 ms at 16 ROI pairs of Config A);
        True: every eval batch; False: never.  Never while a train-mode forward awaits its backward, under autograd, or inside a capture."""
        mode = getattr(self.model, "graph_eval", "auto")
        if mode not in ("auto", True, False):
            raise ValueError("PSMNet.graph_eval must be 'auto', True or False")
        if training or mode is False or self._held or torch.is_grad_enabled() or n_units == 0 or E.TIMING is not None:
            return False
        if torch.cuda.is_current_stream_capturing():
            return False
        return mode is True or n_units <= GRAPH_MAX_UNITS

    def _replay(self, key, inputs, fn):
        """inputs: the caller's tensors (units along dim 0); fn(*static_inputs) -> output tensor (units along dim 0); key: everything the
        pass depends on but the unit count.  ROI pairs are independent units, so a batch is padded to its capacity bucket
        (engine.bucket_units: {2^k, 3*2^(k-1)}) and ONE graph per bucket serves every count in it -- the ROI count changes from image to image
        (disprcnn3d.py:272-275), a graph per exact count would be captured over and over.  A bucket is captured when it is seen for
        the GRAPH_CAPTURE_AFTER-th time (until then: eager, exact count), after an eager warm-up that builds plans / workspaces; re-captured
        when what it points at was replaced (epoch); replayed otherwise."""
        from ...utils.graph import GraphedStep
        n = inputs[0].shape[0]
        nb = E.bucket_units(n)
        key = (nb,) + tuple(key)
        ent = self._graphs.get(key)
        if ent is not None and ent["epoch"] == self._epoch:
            # OPTIMISTIC replay: launch first, then run the parameter / buffer version check (0.14 ms of host time: ~850 tensors) while the GPU
            # works -- since round 6 every pass ends with the range guard's synchronisation, so host time in front of the launch is no longer
            # hidden behind the previous pass.  A change found afterwards (the epoch advanced) discards this replay and takes the slow path
            # below; the stale launch only read tensors that stay allocated until the stream has passed it (the caching allocator frees in
            # stream order).
            out = self._replay_launch(ent, inputs, n)
            self._compile()
            if ent["epoch"] == self._epoch:
                self._graphs.move_to_end(key)
                return out
            ent = self._graphs.get(key)
        else:
            self._compile()                               # a parameter / buffer change advances the epoch BEFORE the lookup
        if ent is not None and ent["epoch"] != self._epoch:
            del self._graphs[key]
            ent = None
        if ent is None:
            seen = self._graph_seen.get(key, 0) + 1
            if len(self._graph_seen) > 256:
                self._graph_seen.clear()
            self._graph_seen[key] = seen
            if seen < GRAPH_CAPTURE_AFTER:
                return fn(*inputs)
            static = [torch.zeros((nb,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in inputs]
            for s_, t in zip(static, inputs):
                s_[:n].copy_(t)
            self._s16_used = False
            fn(*static)                                    # eager once: plans, workspaces, pools (may advance the epoch)
            step = GraphedStep(lambda: fn(*static), warmup=1)
            ent = dict(epoch=self._epoch, static=static, step=step, filled=n, s16=self._s16_used, guard=E.guard_in_scope())
            self._graphs[key] = ent
            self._graph_seen.pop(key, None)
            while len(self._graphs) > GRAPH_MAX_ENTRIES:
                self._graphs.popitem(last=False)
        else:
            self._graphs.move_to_end(key)
        return self._replay_launch(ent, inputs, n)         # (a capture records the launches, it does not run them: replay also after capturing)

    def _replay_launch(self, ent, inputs, n):
        for s_, t in zip(ent["static"], inputs):
            s_[:n].copy_(t)
            if ent["filled"] > n:                          # rows of an earlier, larger batch: units of their own, but keep the padding inert
                s_[n:ent["filled"]].zero_()
        ent["filled"] = n
        if ent["s16"]:
            self._s16_used = True
            g = E.guard_in_scope()
            if g is not None:
                g.used = True                              # the replayed launches report to the guard word captured with them (key: its id)
        return ent["step"]()[:n].clone()

    # ------------------------------------------------------------------ range guard of the split-f16 schedules (engine.guarded)
    def _guarded(self, training, fn):
        """Eval passes run under this model's OverflowGuard: when a split-f16 kernel had to clamp a value (|v| > 65504, Inf, NaN -- the fp32
        reference has no such limit, config/defaults.py:22) the pass is repeated on the fp32 MFMA kernels ("auto") or raises ("f16x2").
        PSMNet.overflow_check = False skips the check (and its one stream synchronisation per forward)."""
        if training or self.device.type != "cuda":
            return fn()
        if self._guard is None:
            self._guard = E.OverflowGuard(self.device)
        m = self.model
        strict = "f16x2" in (getattr(m, "regressor_math", "auto"), getattr(m, "feature_math", "auto"))
        return E.guarded(self._guard, fn, strict=strict, what="PSMNet (split-f16 regressor / feature CNN)",
                         enabled=bool(getattr(m, "overflow_check", True)))

    def _graph_key_tail(self):
        """What a captured eval graph depends on besides its input shape: the module's arithmetic switches, the engine's schedule switches,
        the range-guard word its launches report to and whether split-f16 schedules are allowed right now."""
        m = self.model
        return (m.maxdisp, m.mindisp, getattr(m, "regressor_math", "auto"), getattr(m, "regressor_storage", "f32"),
                getattr(m, "feature_storage", "f32"), getattr(m, "feature_math", "auto"),
                E.S16["enabled"], E.HEAD_FUSED["enabled"], E.LASTCONV_S16["enabled"], E.TRUNK_S16["enabled"], E.s16_allowed(),
                id(E.guard_in_scope()))

    def forward_features(self, fl, fr, out_hw, training=False):
        return self._guarded(training, lambda: self._forward_features_once(fl, fr, out_hw, training))

    def forward_images(self, left, right, training=False):
        return self._guarded(training, lambda: self._forward_images_once(left, right, training))

    def _forward_features_once(self, fl, fr, out_hw, training=False):
        self._training = bool(training)
        self._slot = self._pick_slot()
        if fl.is_cuda and fl.shape == fr.shape and fl.dim() == 4 and self._graph_mode(fl.shape[0], training):
            key = ("feat", tuple(fl.shape[1:]), tuple(out_hw)) + self._graph_key_tail()
            return self._replay(key, (fl, fr), lambda a, b: self._forward_features_impl(a, b, out_hw, False))
        if training and torch.is_grad_enabled():
            params = [p for _, _, p in self._slots("p") if p.requires_grad and not self._is_fe_param(p)]
            if fl.requires_grad or fr.requires_grad or params:
                return _RegressorTrainFn.apply(self, tuple(out_hw), fl, fr, *params)
        return self._forward_features_impl(fl, fr, out_hw, training)

    def _is_fe_param(self, p):
        ids = getattr(self, "_fe_ids", None)
        if ids is None:
            ids = self._fe_ids = {id(q) for q in self.model.feature_extraction.parameters()}
        return id(p) in ids

    def _forward_features_impl(self, fl, fr, out_hw, training):
        E.require_gpu(fl, "PSMNet features"); E.require_gpu(fr, "PSMNet features")
        mx, mn = self._check_disp()
        N, C, Hp, Wp = fl.shape
        H, W = out_hw
        if C != 32:
            raise ValueError("feature maps must have 32 channels")
        if N == 0:
            z = torch.empty(0, H, W, dtype=torch.float32, device=self.device)
            return (z, z.clone(), z.clone()) if training else z       # empty ROI batch (reference: disprcnn3d.py:272-275)
        if not self._use_f16(training) and self._use_s16(training, (mx - mn) // 4, Hp, Wp):
            ws = self._ws3d_s16(N, (mx - mn) // 4, Hp, Wp)
            self._stamp(ws)
            ws["t"]["featL"].from_dense(fl)             # (the two layout converters need no weights: they are launched BEFORE the parameter-version
            ws["t"]["featR"].from_dense(fr)             #  check -- 0.14 ms of host time that would otherwise sit in front of an idle GPU, §3.8)
            Wt = self._compile()
            return self._heads(self._regress_s16(ws, Wt, mn // 4), N, H, W, mx, mn, False)
        Wt = self._compile()
        if self._use_f16(training):
            ws = self._ws3d16(N, (mx - mn) // 4, Hp, Wp)
            self._stamp(ws)
            self._costvol16(ws, mn, mx, fl.contiguous(), fr.contiguous())
            return self._heads(self._regress16(ws, Wt), N, H, W, mx, mn, False)
        ws = self._ws3d(N, (mx - mn) // 4, Hp, Wp)
        self._stamp(ws)
        cv = None
        if self._fuse_costvol(ws, training):
            pool = ws["pool"]
            cv = (pool.blocked("featL", N, 32, 1, Hp, Wp, 0, 1, 1).from_dense(fl),
                  pool.blocked("featR", N, 32, 1, Hp, Wp, 0, 1, 1).from_dense(fr), mn // 4)
        else:
            E.cost_volume_blocked(fl.contiguous(), fr.contiguous(), ws["t"]["cost"], mn // 4, mx // 4, 0)
        costs = self._regress(ws, Wt, cv)
        self._last_train = (ws, Wt, costs, mx, mn, (H, W))
        self._last_gens = self._generations(ws)
        return self._heads(costs, N, H, W, mx, mn, training)

    # ------------------------------------------------------------------ 2D feature CNN
    def _ws2d(self, N, H, W, side=None):
        key = self._slotted(("2d", N, H, W) if side is None else ("2d", N, H, W, side))
        ws = self._ws_get(key)
        if ws is not None:
            return ws
        if H % 4 or W % 4 or H // 4 < 56 or W // 4 < 56:
            raise ValueError("PSMNet needs H,W multiples of 4 and >= 224 (fixed AvgPool2d(56), reference submodule.py:76)")
        pool = self._pool_for(self._slotted(("2d", H, W, side)), N)
        names = iter(range(1 << 30))
        B2 = lambda c, h, w, pad=1: pool.blocked(("t", next(names)), N, c, 1, h, w, 0, pad, pad)   # allocation order is deterministic
        H2, W2, H4, W4 = H // 2, W // 2, H // 4, W // 4
        t, p = {}, {}
        t["img"] = B2(3, H, W)
        t["f0"], t["f1"], t["f2"] = B2(32, H2, W2), B2(32, H2, W2), B2(32, H2, W2)
        p["fe.firstconv.0"] = E.plan_conv2d(t["img"], t["f0"], 3, 2, 1, 1, 32, True)
        p["fe.firstconv.2"] = E.plan_conv2d(t["f0"], t["f1"], 3, 1, 1, 1, 32, True)
        p["fe.firstconv.4"] = E.plan_conv2d(t["f1"], t["f2"], 3, 1, 1, 1, 32, True)
        t["cat"] = B2(320, H4, W4)
        sched = []   # (plan key, weight key, x, y, res)
        cur, cur_c, cur_hw = "f2", 32, (H2, W2)
        for name, planes, nblk, stride, dil in TRUNK_STAGES:
            pad = 2 if name in ("layer3", "layer4") else 1      # tensors read by the dilated layer4 carry halo 2
            for b in range(nblk):
                s = stride if b == 0 else 1
                hw = (cur_hw[0] // s, cur_hw[1] // s)
                u = f"fe.{name}.{b}"
                mid = u + ".mid"
                t[mid] = B2(planes, hw[0], hw[1], pad)
                p[u + ".conv1"] = E.plan_conv2d(t[cur], t[mid], 3, s, 1 if dil == 1 else dil, dil, planes, True)
                sched.append((u + ".conv1", u + ".conv1", cur, mid, None))
                res = cur
                if (u + ".down") in self._w:
                    t[u + ".sc"] = B2(planes, hw[0], hw[1], pad)
                    p[u + ".down"] = E.plan_conv2d(t[cur], t[u + ".sc"], 1, s, 0, 1, planes, False)
                    sched.append((u + ".down", u + ".down", cur, u + ".sc", None))
                    res = u + ".sc"
                last = b == nblk - 1
                if name == "layer2" and last:
                    t[u + ".out"] = E.BlockedSlice(t["cat"], 0, 64)        # output_raw -> channels 0..63 of the concat
                elif name == "layer4" and last:
                    t[u + ".out"] = E.BlockedSlice(t["cat"], 4, 128)       # output_skip -> channels 64..191
                else:
                    t[u + ".out"] = B2(planes, hw[0], hw[1], pad)
                p[u + ".conv2"] = E.plan_conv2d(t[mid], t[u + ".out"], 3, 1, 1 if dil == 1 else dil, dil, planes, False)
                sched.append((u + ".conv2", u + ".conv2", mid, u + ".out", res))   # out += x, no trailing relu
                cur, cur_c, cur_hw = u + ".out", planes, hw
        skip = cur
        return self._ws_put(key, self._ws2d_tail(t, p, B2, pool, sched, skip, H4, W4))

    @staticmethod
    def _ws2d_tail(t, p, B2, pool, sched, skip, H4, W4):
        """The SPP branches and lastconv of a 2D workspace (shared by the fp32 and the split-f16 schedules)."""
        # SPP: concat order (raw, skip, b4, b3, b2, b1) -> channel blocks 0-3, 4-11, 12-13, 14-15, 16-17, 18-19
        slot = {"branch4": 12, "branch3": 14, "branch2": 16, "branch1": 18}
        spp = []
        for name, k in SPP_BRANCHES:
            oh, ow = H4 // k, W4 // k
            t[name + ".pool"] = B2(128, oh, ow, 0)
            t[name + ".conv"] = B2(32, oh, ow, 0)
            p["fe." + name] = E.plan_conv2d(t[name + ".pool"], t[name + ".conv"], 1, 1, 0, 1, 32, True)
            spp.append((name, k, oh, ow, slot[name]))
        t["last0"] = B2(128, H4, W4, 0)
        t["feat"] = B2(32, H4, W4, 1)
        p["fe.lastconv.0"] = E.plan_conv2d(t["cat"], t["last0"], 3, 1, 1, 1, 128, True)
        p["fe.lastconv.2"] = E.plan_conv2d(t["last0"], t["feat"], 1, 1, 0, 1, 32, False)
        return dict(t=t, p=p, pool=pool, sched=sched, spp=spp, skip=skip, dims=(H4, W4),
                    flops=sum(pl.flops for pl in p.values()) + sum(e[1].flops for e in sched if e[0] == "s16"))

    # ------------------------------------------------------------------ split-f16 2D feature CNN (eval; convs16r.hip)
    def _use_s16_2d(self, training, H, W):
        """Eval: the stride-1 3x3 layers of feature_extraction (firstconv[2], [4], layer1, layer2 but its first conv, layer3, -- maps of a
        multiple of 56 rows -- the dilated layer4, and lastconv[0] as three chained launches: 99 % of the CNN's FLOPs) run in split-f16
        arithmetic on the f16 matrix cores (convs16r.hip: fp32-class results) when the maps are multiples of 28 rows / 56 columns;
        PSMNet.feature_math = "f32" keeps the fp32 MFMA kernels for all of them."""
        mode = getattr(self.model, "feature_math", "auto")
        if mode not in ("auto", "f32", "f16x2"):
            raise ValueError("PSMNet.feature_math must be 'auto', 'f32' or 'f16x2'")
        ok = (not training and self._tape is None and H % 4 == 0 and W % 4 == 0 and H // 4 >= 56 and W // 4 >= 56 and
              # (round 6: any such crop size -- the 2D kernel masks ragged last tiles; whole 28-row / 56-column tiles at the shipped 224 x 224)
              E.s16_supported(32, 32, 1, H // 2, W // 2, "2d") and E.s16_supported(64, 64, 1, H // 4, W // 4, "2d") and
              E.s16_supported(64, 128, 1, H // 4, W // 4, "2d") and E.s16_supported(128, 128, 1, H // 4, W // 4, "2d"))
        if mode == "f16x2" and not ok:
            raise RuntimeError("PSMNet.feature_math = 'f16x2': eval only; H, W multiples of 4 and >= 224 (the reference's AvgPool2d(56), submodule.py:76)")
        return self._s16_choice(ok and mode != "f32", mode, "feature_math")

    def _ws2d_eval(self, N, H, W):
        return self._ws2d_s16(N, H, W) if self._use_s16_2d(False, H, W) else self._ws2d(N, H, W)

    def _ws2d_s16(self, N, H, W):
        """The schedule of _ws2d with the stride-1 undilated 3x3 layers on RS16 maps.  Entries: ("s16", plan, weight key, x, y, res) a
        split-f16 layer; ("to16", blocked, rs16) / ("to32", rs16, blocked) the layout converters at the seams to the fp32 kernels (the
        stride-2 conv and the 1x1 downsamples of layer2 / layer3, the dilated layer4, the concat); 5-tuples: fp32 sites as in _ws2d."""
        key = self._slotted(("2ds16", N, H, W))
        ws = self._ws_get(key)
        if ws is not None:
            return ws
        pool = self._pool_for(self._slotted(("2ds16", H, W)), N)
        names = iter(range(1 << 30))
        B2 = lambda c, h, w, pad=1: pool.blocked(("t", next(names)), N, c, 1, h, w, 0, pad, pad)
        S2 = lambda name, c, h, w: pool.rs16(("s", name), N, c, 1, h, w, 0)
        H2, W2, H4, W4 = H // 2, W // 2, H // 4, W // 4
        t, p = {}, {}
        t["img"] = B2(3, H, W)
        t["f0"] = B2(32, H2, W2)
        p["fe.firstconv.0"] = E.plan_conv2d(t["img"], t["f0"], 3, 2, 1, 1, 32, True)
        t["cat"] = B2(320, H4, W4)
        sched = []
        plans = {}

        def s16(wname, x, y, res, cin, cout, h, w, relu):
            k = (cin, cout, h, w, relu)
            if k not in plans:
                plans[k] = E.ConvPlanS16(N, cin, cout, 1, h, w, relu, device=self.device, kind="2d")
            sched.append(("s16", plans[k], wname, x, y, res))

        # firstconv[2], [4] and layer1: 32 channels at half resolution, three rotating maps
        for i in range(3):
            t[f"a{i}"] = S2(f"a{i}", 32, H2, W2)
        sched.append(("to16", "f0", "a0"))
        s16("fe.firstconv.2", "a0", "a1", None, 32, 32, H2, W2, True)
        s16("fe.firstconv.4", "a1", "a2", None, 32, 32, H2, W2, True)
        cur, free = "a2", ["a0", "a1"]
        for b in range(TRUNK_STAGES[0][2]):
            u = f"fe.layer1.{b}"
            mid, out = free
            s16(u + ".conv1", cur, mid, None, 32, 32, H2, W2, True)
            s16(u + ".conv2", mid, out, cur, 32, 32, H2, W2, False)
            cur, free = out, [mid, cur]
        # layer2: the stride-2 conv and the 1x1 downsample of its first block stay fp32
        t["l2in"] = B2(32, H2, W2)
        t["l2mid"], t["l2sc"] = B2(64, H4, W4), B2(64, H4, W4)
        sched.append(("to32", cur, "l2in"))
        u = "fe.layer2.0"
        p[u + ".conv1"] = E.plan_conv2d(t["l2in"], t["l2mid"], 3, 2, 1, 1, 64, True)
        sched.append((u + ".conv1", u + ".conv1", "l2in", "l2mid", None))
        p[u + ".down"] = E.plan_conv2d(t["l2in"], t["l2sc"], 1, 2, 0, 1, 64, False)
        sched.append((u + ".down", u + ".down", "l2in", "l2sc", None))
        for i in range(3):
            t[f"b{i}"] = S2(f"b{i}", 64, H4, W4)
        sched.append(("to16", "l2mid", "b0"))
        sched.append(("to16", "l2sc", "b1"))
        s16(u + ".conv2", "b0", "b2", "b1", 64, 64, H4, W4, False)
        cur, free = "b2", ["b0", "b1"]
        for b in range(1, TRUNK_STAGES[1][2]):
            u = f"fe.layer2.{b}"
            mid, out = free
            s16(u + ".conv1", cur, mid, None, 64, 64, H4, W4, True)
            s16(u + ".conv2", mid, out, cur, 64, 64, H4, W4, False)
            cur, free = out, [mid, cur]
        # output_raw -> channels 0..63 of the concat (fp32), which is also the input of layer3's 1x1 downsample
        t["raw"] = E.BlockedSlice(t["cat"], 0, 64)
        sched.append(("to32", cur, "raw"))
        raw16 = cur                                            # (its RS16 map stays: layer3 rotates its own three maps)
        t["l3sc"] = B2(128, H4, W4)
        u = "fe.layer3.0"
        p[u + ".down"] = E.plan_conv2d(t["raw"], t["l3sc"], 1, 1, 0, 1, 128, False)
        sched.append((u + ".down", u + ".down", "raw", "l3sc", None))
        for i in range(3):
            t[f"c{i}"] = S2(f"c{i}", 128, H4, W4)
        sched.append(("to16", "l3sc", "c1"))
        s16(u + ".conv1", cur, "c0", None, 64, 128, H4, W4, True)
        s16(u + ".conv2", "c0", "c2", "c1", 128, 128, H4, W4, False)
        cur, free = "c2", ["c0", "c1"]
        for b in range(1, TRUNK_STAGES[2][2]):
            u = f"fe.layer3.{b}"
            mid, out = free
            s16(u + ".conv1", cur, mid, None, 128, 128, H4, W4, True)
            s16(u + ".conv2", mid, out, cur, 128, 128, H4, W4, False)
            cur, free = out, [mid, cur]
        # layer4 (dilation 2): the same kernel's dilated form where the map is a multiple of 56 rows, else fp32 on tensors with halo 2
        name, planes, nblk, stride, dil = TRUNK_STAGES[3]
        if E.s16_supported(128, 128, 1, H4, W4, "2d", dil):
            plans4 = {}

            def s16d(wname, x, y, res, relu):
                if relu not in plans4:
                    plans4[relu] = E.ConvPlanS16(N, 128, 128, 1, H4, W4, relu, device=self.device, kind="2d", dil=dil)
                sched.append(("s16", plans4[relu], wname, x, y, res))
            for b in range(nblk):
                u = f"fe.{name}.{b}"
                mid, out = free
                s16d(u + ".conv1", cur, mid, None, True)
                s16d(u + ".conv2", mid, out, cur, False)
                cur, free = out, [mid, cur]
            t["skip"] = E.BlockedSlice(t["cat"], 4, 128)       # output_skip -> channels 64..191 of the concat
            sched.append(("to32", cur, "skip"))
            last16 = (raw16, cur, free[0], free[1])            # RS16 maps of raw and skip, two free 128-channel maps
            cur = "skip"
        else:
            last16 = None
            t["l4in"] = B2(128, H4, W4, 2)
            sched.append(("to32", cur, "l4in"))
            cur = "l4in"
            for b in range(nblk):
                u = f"fe.{name}.{b}"
                mid = u + ".mid"
                t[mid] = B2(planes, H4, W4, 2)
                p[u + ".conv1"] = E.plan_conv2d(t[cur], t[mid], 3, 1, dil, dil, planes, True)
                sched.append((u + ".conv1", u + ".conv1", cur, mid, None))
                t[u + ".out"] = E.BlockedSlice(t["cat"], 4, 128) if b == nblk - 1 else B2(planes, H4, W4, 2)
                p[u + ".conv2"] = E.plan_conv2d(t[mid], t[u + ".out"], 3, 1, dil, dil, planes, False)
                sched.append((u + ".conv2", u + ".conv2", mid, u + ".out", cur))
                cur = u + ".out"
        ws = self._ws2d_tail(t, p, B2, pool, sched, cur, H4, W4)
        ws["s16"] = True
        if last16 is not None and E.LASTCONV_S16["enabled"]:
            # lastconv[0] (3x3, 320 -> 128, submodule.py:125-128; 10.5 % of the CNN's FLOPs) as three chained split-f16 launches over the
            # concat's parts -- raw (64), skip (128), the four upsampled SPP branches (128) -- each adding the previous partial sum as its
            # residual (BN is affine: its scale goes to every part, its shift to the first, the ReLU to the last).  The parts' RS16 maps exist
            # already (raw, skip) or are converted from the concat's branch slice; the fp32 concat is still what the 1x1 / pooling kernels read.
            t["br32"] = E.BlockedSlice(t["cat"], 12, 128)
            t["br16"] = S2("br16", 128, H4, W4)
            raw16, skip16, lp0, lp1 = last16
            ws["last16"] = dict(
                bounds=((0, 64), (64, 192), (192, 320)),
                steps=((E.ConvPlanS16(N, 64, 128, 1, H4, W4, False, device=self.device, kind="2d"), raw16, lp0, None),
                       (E.ConvPlanS16(N, 128, 128, 1, H4, W4, False, device=self.device, kind="2d"), skip16, lp1, lp0),
                       (E.ConvPlanS16(N, 128, 128, 1, H4, W4, True, device=self.device, kind="2d"), "br16", lp0, lp1)))
        return self._ws_put(key, ws)

    def _features(self, ws, W, images):
        """feature_extraction on a batch of images (left and right stacked) -> blocked [N,32,H/4,W/4] (halo 1)."""
        from ... import _lib
        t, p = ws["t"], ws["p"]
        lib = _lib.lib()
        sp = E._stream_ptr(self.device)

        def run(plan, wname, x, y, res=None):
            self._site(ws, W, plan, wname, x, y, res)

        c0 = W["fe.firstconv.0"]
        views = images if isinstance(images, (tuple, list)) else (images,)      # (left, right): stacked along the batch, left units first
        if (E.STEM_DIRECT["enabled"] and not self._training and c0.cout_pad in (16, 32) and t["f0"].D == 1 and
                all(v.dtype == torch.float32 and v.dim() == 4 and v.shape[1] == 3 for v in views)):
            # eval: the first layer reads the dense image(s) (stemconv.hip): no 16-channel-blocked copy of a 3-channel image, no torch.cat
            if c0._stem is None:
                c0._stem = E.pack_weight_stem(c0.conv.weight.detach().to(device=self.device, dtype=torch.float32))
            unit0 = 0
            for v in views:
                E.stem_conv(v, c0._stem, c0.scale, c0.shift, t["f0"], True, unit0=unit0)
                unit0 += v.shape[0]
        else:
            t["img"].from_dense(images if len(views) == 1 else torch.cat(tuple(views), 0))
            run("fe.firstconv.0", "fe.firstconv.0", "img", "f0")
        if not ws.get("s16"):
            run("fe.firstconv.2", "fe.firstconv.2", "f0", "f1")
            run("fe.firstconv.4", "fe.firstconv.4", "f1", "f2")
        for e in ws["sched"]:
            if e[0] == "s16":
                _, pl, wname, x, y, res = e
                c = W[wname]
                w16, sc16 = c.s16()
                pl.run(t[x], w16, sc16, c.shift, y16=t[y], res=t[res] if res else None)
            elif e[0] == "to16":
                t[e[2]].from_blocked(t[e[1]])
            elif e[0] == "to32":
                t[e[1]].to_blocked(t[e[2]])
            else:
                run(*e)
        skip = t[ws["skip"]]
        H4, W4 = ws["dims"]
        cat = t["cat"]
        # SPP pools.  Eval: every window of the reference's pools (56, 32, 16, 8 on the 56-wide map, floor mode) is a union of the finest
        # pool's 8x8 cells, and a mean of equally sized cells' means is the window's mean -- so only the finest pool reads the 128-channel
        # map (205 MB for the stress shape's 128 crops), the others read its 7x7 result (round 4: four passes over the map were 4 x 53 us).
        # Training keeps one pass per branch (the reverse pass spreads each branch's gradient over its own windows).
        fine = min(ws["spp"], key=lambda e: e[1])
        nested = (not self._training and E.SPP_NESTED["enabled"] and
                  all(e[1] % fine[1] == 0 and e[2] * (e[1] // fine[1]) <= fine[2] and e[3] * (e[1] // fine[1]) <= fine[3] for e in ws["spp"]))
        for name, k, oh, ow, cb_off in ([fine] + [e for e in ws["spp"] if e is not fine] if nested else ws["spp"]):
            pool = t[name + ".pool"]
            if nested and name != fine[0]:
                src = t[fine[0] + ".pool"]
                st = lib.drc_avgpool2d_blocked_slice(E._ptr(src.storage), E._ptr(pool.storage), src.N, src.cb, src.H, src.W, src.ph, k // fine[1], oh, ow, 0,
                                                     src.cb, 0, sp)
            else:
                st = self._avgpool_slice(lib, skip, pool, k, oh, ow, sp)
            _lib.check(st, "drc_avgpool2d_blocked")
        for name, k, oh, ow, cb_off in ws["spp"]:
            conv = t[name + ".conv"]
            run("fe." + name, "fe." + name, name + ".pool", name + ".conv")
            st = lib.drc_bilinear_up_blocked(E._ptr(conv.storage), E._ptr(cat.storage), cat.N, 2, oh, ow, 0, H4, W4, cat.ph,
                                             cat.cb, cb_off, sp)
            _lib.check(st, "drc_bilinear_up_blocked")
        l16 = ws.get("last16")
        if l16 is not None:
            c = W["fe.lastconv.0"]
            t["br16"].from_blocked(t["br32"])
            parts = c.s16_cin_slices(l16["bounds"])
            for i, ((pl, x, y, res), (w16, sc16)) in enumerate(zip(l16["steps"], parts)):
                pl.run(t[x], w16, sc16, c.shift if i == 0 else c.zero_shift, y16=t[y], res=t[res] if res else None)
            t[l16["steps"][-1][2]].to_blocked(t["last0"])
        else:
            run("fe.lastconv.0", "fe.lastconv.0", "cat", "last0")
        run("fe.lastconv.2", "fe.lastconv.2", "last0", "feat")
        return t["feat"]

    # ------------------------------------------------------------------ fp16-storage 2D feature CNN (BASELINE configs[3]; eval only)
    def _ws2d16(self, N, H, W):
        """The schedule of _ws2d on Blocked16 tensors.  The 320-channel concat is 10 blocks of 32: raw (64) -> blocks 0-1, skip (128) ->
        2-5, branch4..branch1 (32 each) -> 6, 7, 8, 9 (concatenation order of submodule.py:134-135)."""
        key = self._slotted(("2d16", N, H, W))
        ws = self._ws_get(key)
        if ws is not None:
            return ws
        if H % 4 or W % 4 or H // 4 < 56 or W // 4 < 56:
            raise ValueError("PSMNet needs H,W multiples of 4 and >= 224 (fixed AvgPool2d(56), reference submodule.py:76)")
        pool = self._pool_for(self._slotted(("2d16", H, W)), N)
        names = iter(range(1 << 30))
        B2 = lambda c, h, w, pad=1: pool.blocked16(("t16", next(names)), N, c, 1, h, w, 0, pad, pad)
        H2, W2, H4, W4 = H // 2, W // 2, H // 4, W // 4
        t, p = {}, {}
        t["img"] = B2(3, H, W)
        t["f0"], t["f1"], t["f2"] = B2(32, H2, W2), B2(32, H2, W2), B2(32, H2, W2)
        p["fe.firstconv.0"] = E.plan_conv2d16(t["img"], t["f0"], 3, 2, 1, 1, 32, True)
        p["fe.firstconv.2"] = E.plan_conv2d16(t["f0"], t["f1"], 3, 1, 1, 1, 32, True)
        p["fe.firstconv.4"] = E.plan_conv2d16(t["f1"], t["f2"], 3, 1, 1, 1, 32, True)
        t["cat"] = B2(320, H4, W4)
        sched = []
        cur, cur_hw = "f2", (H2, W2)
        for name, planes, nblk, stride, dil in TRUNK_STAGES:
            pad = 2 if name in ("layer3", "layer4") else 1      # tensors read by the dilated layer4 carry halo 2
            for b in range(nblk):
                s = stride if b == 0 else 1
                hw = (cur_hw[0] // s, cur_hw[1] // s)
                u = f"fe.{name}.{b}"
                mid = u + ".mid"
                t[mid] = B2(planes, hw[0], hw[1], pad)
                p[u + ".conv1"] = E.plan_conv2d16(t[cur], t[mid], 3, s, 1 if dil == 1 else dil, dil, planes, True)
                sched.append((u + ".conv1", cur, mid, None))
                res = cur
                if (u + ".down") in self._w:
                    t[u + ".sc"] = B2(planes, hw[0], hw[1], pad)
                    p[u + ".down"] = E.plan_conv2d16(t[cur], t[u + ".sc"], 1, s, 0, 1, planes, False)
                    sched.append((u + ".down", cur, u + ".sc", None))
                    res = u + ".sc"
                last = b == nblk - 1
                if name == "layer2" and last:
                    t[u + ".out"] = E.Blocked16Slice(t["cat"], 0, 64)
                elif name == "layer4" and last:
                    t[u + ".out"] = E.Blocked16Slice(t["cat"], 2, 128)
                else:
                    t[u + ".out"] = B2(planes, hw[0], hw[1], pad)
                p[u + ".conv2"] = E.plan_conv2d16(t[mid], t[u + ".out"], 3, 1, 1 if dil == 1 else dil, dil, planes, False)
                sched.append((u + ".conv2", mid, u + ".out", res))
                cur, cur_hw = u + ".out", hw
        slot = {"branch4": 6, "branch3": 7, "branch2": 8, "branch1": 9}
        spp = []
        for name, k in SPP_BRANCHES:
            oh, ow = H4 // k, W4 // k
            t[name + ".pool"] = B2(128, oh, ow, 0)
            t[name + ".conv"] = B2(32, oh, ow, 0)
            p["fe." + name] = E.plan_conv2d16(t[name + ".pool"], t[name + ".conv"], 1, 1, 0, 1, 32, True)
            spp.append((name, k, oh, ow, slot[name]))
        t["last0"] = B2(128, H4, W4, 0)
        t["feat"] = B2(32, H4, W4, 1)
        p["fe.lastconv.0"] = E.plan_conv2d16(t["cat"], t["last0"], 3, 1, 1, 1, 128, True)
        p["fe.lastconv.2"] = E.plan_conv2d16(t["last0"], t["feat"], 1, 1, 0, 1, 32, False)
        ws = dict(t=t, p=p, pool=pool, sched=sched, spp=spp, skip=cur, dims=(H4, W4), flops=sum(pl.flops for pl in p.values()))
        return self._ws_put(key, ws)

    def _features16(self, ws, W, images):
        """feature_extraction with fp16 storage (fp32 accumulation, BatchNorm folded in fp32): -> Blocked16 [N,32,H/4,W/4] (halo 1)."""
        from ... import _lib
        t, p = ws["t"], ws["p"]
        W16 = self._weights16(W)
        lib, sp = _lib.lib(), E._stream_ptr(self.device)
        E.dense_to_blocked16(images, t["img"])

        def run(name, x, y, res=None):
            c = W[name]
            p[name].run(t[x], W16[name], c.scale, c.shift, t[y], t[res] if res else None)

        run("fe.firstconv.0", "img", "f0")
        run("fe.firstconv.2", "f0", "f1")
        run("fe.firstconv.4", "f1", "f2")
        for name, x, y, res in ws["sched"]:
            run(name, x, y, res)
        skip, cat = t[ws["skip"]], t["cat"]
        H4, W4 = ws["dims"]
        for name, k, oh, ow, cb_off in ws["spp"]:
            pool, conv = t[name + ".pool"], t[name + ".conv"]
            st = lib.drc_avgpool2d_blocked16_slice(E._ptr(cat.storage), E._ptr(pool.storage), cat.N, skip.cb, cat.H, cat.W, cat.ph, k, oh, ow, 0,
                                                   cat.cb, skip.cb_off, sp)
            _lib.check(st, "drc_avgpool2d_blocked16_slice")
            run("fe." + name, name + ".pool", name + ".conv")
            st = lib.drc_bilinear_up_blocked16(E._ptr(conv.storage), E._ptr(cat.storage), cat.N, 1, oh, ow, 0, H4, W4, cat.ph, cat.cb, cb_off, sp)
            _lib.check(st, "drc_bilinear_up_blocked16")
        run("fe.lastconv.0", "cat", "last0")
        run("fe.lastconv.2", "last0", "feat")
        return t["feat"]

    def _avgpool_slice(self, lib, skip, pool, k, oh, ow, sp):
        """AvgPool2d(k,k) of output_skip, which lives in channel blocks 4..11 of the 20-block concat tensor."""
        base = skip.base
        return lib.drc_avgpool2d_blocked_slice(E._ptr(base.storage), E._ptr(pool.storage), base.N, skip.cb, base.H, base.W, base.ph, k, oh, ow, 0,
                                               base.cb, skip.cb_off, sp)

    def _forward_images_once(self, left, right, training=False):
        self._training = bool(training)
        self._slot = self._pick_slot()
        if left.is_cuda and left.dim() == 4 and self._graph_mode(left.shape[0], training):
            key = ("img", tuple(left.shape[1:])) + self._graph_key_tail()
            return self._replay(key, (left, right), lambda a, b: self._forward_images_impl(a, b, False))
        if training and torch.is_grad_enabled():
            params = [p for _, _, p in self._slots("p") if p.requires_grad]
            if left.requires_grad or right.requires_grad or params:
                return _PSMNetTrainFn.apply(self, left, right, *params)
        return self._forward_images_impl(left, right, training)

    def _forward_images_impl(self, left, right, training):
        E.require_gpu(left, "PSMNet input"); E.require_gpu(right, "PSMNet input")
        mx, mn = self._check_disp()
        N, _, H, W = left.shape
        disp = torch.empty(N, H, W, dtype=torch.float32, device=self.device)
        if N == 0:
            return disp
        Wt = self._compile()
        if self._use_f16(training):
            # fp32 2D feature CNN (its output is the cost volume's input), fp16-storage cost volume + 3D regressor
            ws3 = self._ws3d16(N, (mx - mn) // 4, H // 4, W // 4)
            if self._features_f16():
                # fp16-storage 2D CNN as well (feature_storage = "f16"): blocked fp16 features straight into the fp16 cost volume
                ws2 = self._ws2d16(2 * N, H, W)
                self._stamp(ws3, ws2)
                feat = self._features16(ws2, Wt, torch.cat((left, right), 0))
                self._costvol16(ws3, mn, mx, feat16=feat, right_first=N)
                return self._heads(self._regress16(ws3, Wt), N, H, W, mx, mn, False)
            ws2 = self._ws2d_eval(2 * N, H, W)
            self._stamp(ws3, ws2)
            feat = self._features(ws2, Wt, (left, right))
            fv = feat.storage
            self._costvol16(ws3, mn, mx, fv, fv[N * feat.n_stride:], feat.ph)
            return self._heads(self._regress16(ws3, Wt), N, H, W, mx, mn, False)
        if self._use_s16(training, (mx - mn) // 4, H // 4, W // 4):
            ws3 = self._ws3d_s16(N, (mx - mn) // 4, H // 4, W // 4)
            ws2 = self._ws2d_eval(2 * N, H, W)
            self._stamp(ws3, ws2)
            feat = self._features(ws2, Wt, (left, right))
            ws3["t"]["featL"].from_blocked(feat, 0)
            ws3["t"]["featR"].from_blocked(feat, N)
            return self._heads(self._regress_s16(ws3, Wt, mn // 4), N, H, W, mx, mn, False)
        ws3 = self._ws3d(N, (mx - mn) // 4, H // 4, W // 4)
        cv = None
        if training:
            # the reference runs feature_extraction(left) and feature_extraction(right) as two calls (stackhourglass.py:112-113):
            # batch statistics and running-stat updates are per call, so the two views must not share a batch here
            wsL, wsR = self._ws2d(N, H, W, "L"), self._ws2d(N, H, W, "R")
            self._stamp(ws3, wsL, wsR)
            featL = self._features(wsL, Wt, left)
            featR = self._features(wsR, Wt, right)
            E.cost_volume_blocked(featL.storage, featR.storage, ws3["t"]["cost"], mn // 4, mx // 4, featL.ph)
            self._last_train_2d = (wsL, wsR)
            self._last_gens = self._generations(ws3, wsL, wsR)
        else:
            ws2 = self._ws2d_eval(2 * N, H, W)
            self._stamp(ws3, ws2)
            feat = self._features(ws2, Wt, (left, right))
            if self._fuse_costvol(ws3, training):
                cv = (feat, (feat, N), mn // 4)
            else:
                fv = feat.storage
                right_view = fv[N * feat.n_stride:]
                E.cost_volume_blocked(fv, right_view, ws3["t"]["cost"], mn // 4, mx // 4, feat.ph)
        costs = self._regress(ws3, Wt, cv)
        self._last_train = (ws3, Wt, costs, mx, mn, (H, W))
        return self._heads(costs, N, H, W, mx, mn, training)


class _RegressorTrainFn(torch.autograd.Function):
    """Differentiable train-mode pass from the feature boundary: forward on the HIP engine (tape recorded), backward by
    modeling/psmnet/train.py.  The forward keeps its workspace slot until its backward has run (or its outputs are dropped), so up to
    MAX_SLOTS forward passes may be in flight -- gradient accumulation, an eval pass in between -- each on a pool of its own; the
    generation check stays as a guard (a backward whose workspace was overwritten raises)."""

    @staticmethod
    def forward(ctx, rt, out_hw, fl, fr, *params):
        rt._tape = []
        rt._need_input_grad = bool(fl.requires_grad or fr.requires_grad)
        try:
            preds = rt._forward_features_impl(fl.detach(), fr.detach(), out_hw, True)
            ctx.tape, ctx.info, ctx.gens = rt._tape, rt._last_train, rt._last_gens
            ctx.slot, ctx.token = rt._slot, rt._hold(rt._slot)
        finally:
            rt._tape = None
        ctx.rt, ctx.params, ctx.need_in = rt, params, rt._need_input_grad
        ctx.feat_shape = tuple(fl.shape)
        return preds

    @staticmethod
    def backward(ctx, g1, g2, g3):
        from .train import RegressorBackward
        rt = ctx.rt
        rt._check_generations(ctx.gens)
        ws, Wt, costs, mx, mn, out_hw = ctx.info
        bw = RegressorBackward(rt, ws, Wt)
        rt._need_input_grad = ctx.need_in
        G = bw.run(ctx.tape, (g1, g2, g3), costs, mx, mn, out_hw)
        gfl = gfr = None
        if ctx.need_in:
            from ... import ops
            gcost = G.get("cost").to_dense()
            gfl, gfr = ops.cost_volume_backward(gcost, mx, mn)
        rt._release(rt, ctx.slot, id(ctx.token))            # the saved activations are spent: the slot is free for the next forward
        return (None, None, gfl, gfr) + tuple(bw.pg.get(id(p)) for p in ctx.params)


class _PSMNetTrainFn(torch.autograd.Function):
    """Differentiable train-mode PSMNet.forward on image crops (reference stackhourglass.py:106-167): the 2D CNN runs once
    per view (two tapes), then the regressor; backward = regressor -> cost-volume adjoint -> both 2D passes."""

    @staticmethod
    def forward(ctx, rt, left, right, *params):
        rt._tape = []
        rt._need_input_grad = True          # the cost volume's gradient feeds the 2D CNN
        try:
            preds = rt._forward_images_impl(left.detach(), right.detach(), True)
            ctx.tape, ctx.info, ctx.ws2, ctx.gens = rt._tape, rt._last_train, rt._last_train_2d, rt._last_gens
            ctx.slot, ctx.token = rt._slot, rt._hold(rt._slot)
        finally:
            rt._tape = None
        ctx.rt, ctx.params = rt, params
        return preds

    @staticmethod
    def backward(ctx, g1, g2, g3):
        from .train import FeaturesBackward, RegressorBackward
        from ... import ops
        rt = ctx.rt
        rt._check_generations(ctx.gens)
        ws3, Wt, costs, mx, mn, out_hw = ctx.info
        rt._need_input_grad = True
        bw = RegressorBackward(rt, ws3, Wt)
        G = bw.run(ctx.tape, (g1, g2, g3), costs, mx, mn, out_hw)
        gfl, gfr = ops.cost_volume_backward(G.get("cost").to_dense(), mx, mn)
        for ws2, gfeat in zip(ctx.ws2, (gfl, gfr)):
            FeaturesBackward(rt, ws2, Wt, bw.pg).run(ctx.tape, gfeat)
        rt._release(rt, ctx.slot, id(ctx.token))
        return (None, None, None) + tuple(bw.pg.get(id(p)) for p in ctx.params)
