"""ResNet-FPN on the HIP tap-conv engine (eval mode: BN folded).  Graph restated from the reference
(resnet.py:137-146,274-316; fpn.py:44-77) as a flat schedule over blocked 2D tensors."""
import torch

from ... import _lib
from ... import engine as E


def _stem_s2d_weight(w):
    """The 7x7 stride-2 pad-3 stem as a 4x4 stride-1 conv over the 2x2 pixel-unshuffled image (resnet.py:274-290 computes the same sums):
    input row 2(Y+ty)+dy with ky-3 = 2ty+dy, ty in -2..1, so w'[o][4c+2dy+dx][ty+2][tx+2] = w[o][c][2(ty+2)+dy-1][2(tx+2)+dx-1]
    (zero where that index is -1).  12 channels x 16 taps = 64 MFMA k-steps a tile instead of 49 taps x 16 padded channels = 196."""
    co, ci = w.shape[:2]
    out = w.new_zeros(co, ci, 2, 2, 4, 4)
    for dy in (0, 1):
        for dx in (0, 1):
            for a in range(4):
                for b in range(4):
                    ky, kx = 2 * a + dy - 1, 2 * b + dx - 1
                    if ky >= 0 and kx >= 0:
                        out[:, :, dy, dx, a, b] = w[:, :, ky, kx]
    return out.reshape(co, ci * 4, 4, 4)


class _Site:
    def __init__(self, conv, bn, device, weight_fn=None):
        w = conv.weight.detach().to(device=device, dtype=torch.float32)
        if weight_fn is not None:
            w = weight_fn(w)
        self.w = E.pack_conv_weight(w)
        self.w16 = E.pack_weight_t16(w) if tuple(w.shape[2:]) == (3, 3) else None      # 3x3 layers: LDS-free kernel's packing
        self._weight, self._ww, self._s16 = w, None, None
        cp = E.cout_pad_of(w.shape[0])
        if bn is not None:
            self.scale, self.shift = E.fold_bn(bn.weight.detach().to(device).float(), bn.bias.detach().to(device).float(),
                                               bn.running_mean.detach().to(device).float(), bn.running_var.detach().to(device).float(),
                                               bn.eps, cp)
        else:
            self.scale = torch.ones(cp, device=device)
            self.shift = torch.zeros(cp, device=device)
            if conv.bias is not None:
                self.shift[: conv.bias.numel()] = conv.bias.detach().to(device).float()


def _s16_slices(site, bounds):
    """[(split-f16 packing of the input channels [a, b), epilogue scale = folded BN scale * 2^-wexp)] of a 3x3 site (engine.BridgedConv2dS16)."""
    from ... import s16 as S
    key = tuple(bounds)
    if site._s16 is None or site._s16[0] != key:
        packs = [S.pack_weight_s16(site._weight[:, a:b].contiguous()) for a, b in bounds]
        site._s16 = (key, [(wp, (site.scale * (2.0 ** -wexp)).contiguous()) for wp, wexp in packs], torch.zeros_like(site.shift))
    return site._s16[1], site._s16[2]


def _w16_for(site, plan):
    """The LDS-free packing `plan` reads: t16, or the Winograd-transformed weights (built on first use)."""
    if not plan.wino:
        return site.w16
    if site._ww is None:
        site._ww = plan.pack16(site._weight)
    return site._ww


def _half(n):       # conv k1/k3 stride 2 (pad k//2) and max_pool2d(1,2,0): floor((n-1)/2)+1
    return (n - 1) // 2 + 1


class BackboneRuntime:
    def __init__(self, model, device):
        self.model, self.device = model, device
        self._w, self._ver, self._ws, self._levels = None, None, {}, None

    def _version(self):
        """(versions of every parameter and buffer, storage addresses of the parameters).  model.parameters() / .buffers() walk ~270 modules
        (0.66 ms of host time per call, in front of the first launch of every forward -- exposed since round 6's range guard synchronises each
        pass); the tensors are cached as (owner dict, name, tensor) slots and the walk is repeated only when a tensor OBJECT was replaced in
        its owner (checked by identity), as in psmnet/runtime.py: 0.1 ms."""
        import operator
        sl = getattr(self, "_slot_cache", None)
        if sl is None or not all(d.get(n) is t for d, n, t in sl[0]):
            slots = []
            for mod in self.model.modules():
                for d in (mod._parameters, mod._buffers):
                    slots.extend((d, n, t) for n, t in d.items() if t is not None)
            params = [t for d, n, t in slots if isinstance(t, torch.nn.Parameter)]
            sl = self._slot_cache = (slots, [t for _, _, t in slots], params)
        return (list(map(operator.attrgetter("_version"), sl[1])), list(map(torch.Tensor.data_ptr, sl[2])))

    def _compile(self):
        v = self._version()
        if self._w is not None and v == self._ver:
            return self._w
        body, fpn, dev = self.model.body, self.model.fpn, self.device
        W = {"stem": _Site(body.stem.conv1, body.stem.bn1, dev, _stem_s2d_weight)}
        for li in range(1, 5):
            for b, u in enumerate(getattr(body, f"layer{li}")):
                p = f"l{li}.{b}"
                W[p + ".c1"], W[p + ".c2"], W[p + ".c3"] = _Site(u.conv1, u.bn1, dev), _Site(u.conv2, u.bn2, dev), _Site(u.conv3, u.bn3, dev)
                if u.downsample is not None:
                    W[p + ".ds"] = _Site(u.downsample[0], u.downsample[1], dev)
        for i in range(1, 5):
            W[f"inner{i}"] = _Site(getattr(fpn, f"fpn_inner{i}"), None, dev)
            W[f"layer{i}"] = _Site(getattr(fpn, f"fpn_layer{i}"), None, dev)
        self._w, self._ver = W, v
        return W

    def _workspace(self, N, H, W_):
        key = (N, H, W_)
        if key in self._ws:
            return self._ws[key]
        dev, body = self.device, self.model.body
        B2 = lambda c, h, w, pad: E.Blocked(N, c, 1, h, w, 0, pad, pad, dev)
        t, p, sched = {}, {}, []
        s16b, s16cache = {}, {}

        def bridge(name, cin, cout, h, w, relu):
            # the 3x3 layers on large maps run on the split-f16 kernel (fp32-class results on the f16 matrix cores) between the same tensors
            if E.BridgedConv2dS16.worth(N, cin, cout, h, w):
                s16b[name] = E.BridgedConv2dS16(N, cin, cout, h, w, relu, dev, s16cache)
        h1, w1 = _half(H), _half(W_)                                    # 7x7 s2 p3
        t["img"] = B2(12, h1, w1, 2)                                    # 2x2 pixel-unshuffled image (see _stem_s2d_weight)
        t["stem"] = B2(64, h1, w1, 0)
        p["stem"] = E.plan_conv2d(t["img"], t["stem"], 4, 1, 2, 1, 64, True)
        # max_pool2d(3, 2, 0, ceil_mode=True): out = ceil((n-3)/2)+1, dropping a last window that would start outside
        ph, pw = -(-(h1 - 3) // 2) + 1, -(-(w1 - 3) // 2) + 1
        if (ph - 1) * 2 >= h1: ph -= 1
        if (pw - 1) * 2 >= w1: pw -= 1
        t["pool"] = B2(64, ph, pw, 0)
        cur, ch, hw = "pool", 64, (ph, pw)
        stage_out = []
        for li in range(1, 5):
            for b, u in enumerate(getattr(body, f"layer{li}")):
                s = u.stride
                mid, cout = u.conv1.out_channels, u.conv3.out_channels
                ohw = (_half(hw[0]), _half(hw[1])) if s == 2 else hw
                q = f"l{li}.{b}"
                t[q + ".a"] = B2(mid, ohw[0], ohw[1], 1)                # feeds the 3x3
                t[q + ".b"] = B2(mid, ohw[0], ohw[1], 0)
                t[q + ".o"] = B2(cout, ohw[0], ohw[1], 0)
                p[q + ".c1"] = E.plan_conv2d(t[cur], t[q + ".a"], 1, s, 0, 1, mid, True)
                p[q + ".c2"] = E.plan_conv2d(t[q + ".a"], t[q + ".b"], 3, 1, 1, 1, mid, True)
                p[q + ".c3"] = E.plan_conv2d(t[q + ".b"], t[q + ".o"], 1, 1, 0, 1, cout, True)   # relu(bn3(conv3) + identity)
                res = cur
                if u.downsample is not None:
                    t[q + ".s"] = B2(cout, ohw[0], ohw[1], 0)
                    p[q + ".ds"] = E.plan_conv2d(t[cur], t[q + ".s"], 1, s, 0, 1, cout, False)
                    sched.append((q + ".ds", cur, q + ".s", None))
                    res = q + ".s"
                sched += [(q + ".c1", cur, q + ".a", None), (q + ".c2", q + ".a", q + ".b", None), (q + ".c3", q + ".b", q + ".o", res)]
                bridge(q + ".c2", mid, mid, ohw[0], ohw[1], True)
                cur, ch, hw = q + ".o", cout, ohw
            stage_out.append((cur, ch, hw))
        # FPN (top-down)
        fsched, outs = [], [None] * 5
        top, tch, thw = stage_out[3]
        t["in4"] = B2(256, thw[0], thw[1], 1)
        p["inner4"] = E.plan_conv2d(t[top], t["in4"], 1, 1, 0, 1, 256, False)
        fsched.append(("conv", "inner4", top, "in4", None))
        outs[3] = "in4"                                                 # top level: no 3x3 block (fork quirk)
        last, lhw = "in4", thw
        for i in (3, 2, 1):
            feat, fch, fhw = stage_out[i - 1]
            t[f"td{i}"] = B2(256, fhw[0], fhw[1], 0)
            t[f"sum{i}"] = B2(256, fhw[0], fhw[1], 1)
            t[f"out{i}"] = B2(256, fhw[0], fhw[1], 1)                 # halo 1: FPN top-down reads out3/out2, the RPN head all levels
            p[f"inner{i}"] = E.plan_conv2d(t[feat], t[f"sum{i}"], 1, 1, 0, 1, 256, False)
            p[f"layer{i}"] = E.plan_conv2d(t[f"sum{i}"], t[f"out{i}"], 3, 1, 1, 1, 256, False)
            fsched.append(("resize", last, f"td{i}", lhw, fhw))
            fsched.append(("conv", f"inner{i}", feat, f"sum{i}", f"td{i}"))      # inner_lateral + inner_top_down
            fsched.append(("conv", f"layer{i}", f"sum{i}", f"out{i}", None))
            bridge(f"layer{i}", 256, 256, fhw[0], fhw[1], False)
            outs[i - 1] = f"out{i}"
            last, lhw = f"out{i}", fhw                                  # reference: last_inner = layer_block(lateral + top_down)
        t["p6"] = B2(256, _half(thw[0]), _half(thw[1]), 1)
        outs[4] = "p6"
        ws = dict(t=t, p=p, sched=sched, fsched=fsched, outs=outs, pool=(h1, w1, ph, pw), thw=thw, s16=s16b,
                  flops=sum(pl.flops for pl in p.values()))
        self._ws[key] = ws
        return ws

    def forward(self, x):
        """The pyramid of backbone/resnet.py:81-146 + fpn.py:44-77.  The 3x3 layers on the largest maps run in split-f16 arithmetic
        (engine.BridgedConv2dS16) under the range guard of engine.guarded: a value beyond +-65504 there (the fp32 reference has no limit)
        repeats the pass on the fp32 kernels.  model.overflow_check = False skips the check."""
        E.require_gpu(x, "backbone input")
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected [N,3,H,W], got {tuple(x.shape)}")
        if getattr(self, "_guard", None) is None:
            self._guard = E.OverflowGuard(self.device)
        return E.guarded(self._guard, lambda: self._forward_once(x), what="ResNet-FPN trunk (split-f16 3x3 layers)",
                         enabled=bool(getattr(self.model, "overflow_check", True)))

    def _forward_once(self, x):
        N, _, H, W_ = x.shape
        Wt = self._compile()
        ws = self._workspace(N, H, W_)
        t, p = ws["t"], ws["p"]
        lib, sp = _lib.lib(), E._stream_ptr(self.device)

        def conv(plan, x_, y_, res=None):
            c = Wt[plan]
            br = ws["s16"].get(plan) if E.s16_allowed() else None
            if br is not None and res is None:
                packs, zero = _s16_slices(c, br.bounds)
                br.run(t[x_], packs, c.shift, zero, t[y_])
            else:
                p[plan].run(t[x_], c.w, c.scale, c.shift, t[y_], t[res] if res else None, w16=_w16_for(c, p[plan]))

        if (H | W_) & 1:
            x = torch.nn.functional.pad(x, (0, W_ & 1, 0, H & 1))
        t["img"].from_dense(torch.nn.functional.pixel_unshuffle(x, 2))
        conv("stem", "img", "stem")
        h1, w1, ph, pw = ws["pool"]
        _lib.check(lib.drc_maxpool2d_blocked(E._ptr(t["stem"].storage), E._ptr(t["pool"].storage), N, 4, h1, w1, 0, 3, 2, ph, pw, 0, sp),
                   "drc_maxpool2d_blocked")
        for plan, x_, y_, res in ws["sched"]:
            conv(plan, x_, y_, res)
        for item in ws["fsched"]:
            if item[0] == "conv":
                conv(item[1], item[2], item[3], item[4])
            else:
                _, src, dst, (ih, iw), (oh, ow) = item
                _lib.check(lib.drc_bilinear_resize_blocked(E._ptr(t[src].storage), E._ptr(t[dst].storage), N, 16, ih, iw, t[src].ph, oh, ow,
                                                           t[dst].ph, 16, 0, 0, sp), "drc_bilinear_resize_blocked")
        th, tw = ws["thw"]
        _lib.check(lib.drc_maxpool2d_blocked(E._ptr(t["in4"].storage), E._ptr(t["p6"].storage), N, 16, th, tw, t["in4"].ph, 1, 2,
                                             t["p6"].H, t["p6"].W, t["p6"].ph, sp), "drc_maxpool2d_blocked")
        self._levels = [t[o] for o in ws["outs"]]
        self._gen = getattr(self, "_gen", 0) + 1
        outs = tuple(t[o].to_dense()[:, :, 0] for o in ws["outs"])
        for o in outs:                                   # the dense maps carry the generation of the blocked workspace they were read from
            o._drc_pyramid_gen = (id(self), self._gen)
        return outs

    def blocked_levels(self, dense_outputs=None):
        """The pyramid in the engine's blocked layout (halo 1), for consumers that run on the engine themselves (the Stereo-RPN head).
        Pass the tuple `forward` returned: the levels are handed out only if that call is still the LAST forward of this runtime (the next
        one overwrites the workspace) -- otherwise None, and the caller uses the dense maps.  Without an argument: the last forward's levels,
        unchecked."""
        if dense_outputs is None:
            return self._levels
        stamp = getattr(dense_outputs[0], "_drc_pyramid_gen", None) if len(dense_outputs) else None
        return self._levels if stamp == (id(self), getattr(self, "_gen", 0)) else None
