"""Backward pass of the 3D regressor on the HIP engine (train mode, from the feature boundary).

The forward (runtime.py) records a tape of conv+BN sites; the backward walks it in reverse:
  BN backward (ReLU mask + residual fan-out fused)   -> drc_bn_bwd_reduce / drc_bn_bwd_apply
  data gradient  = the forward engine with transformed weights (plans cached per site):
        stride-1 conv   -> stride-1 conv, weights transposed + flipped
        stride-2 conv   -> transposed-conv parity classes (weights as they are, ConvTranspose layout)
        transposed conv -> stride-2 conv (weights as they are, Conv layout)
  weight gradient -> drc_tapconv_wgrad (fp32 MFMA, k = voxels)
  classifier conv -> drc_conv3d_cout1_bwd_data/_weight;  soft-argmin -> drc_upsample_softargmin_bwd;
  cost volume     -> drc_cost_volume_bwd.
Reference semantics: torch autograd of stackhourglass.py:115-174 (the oracle's autograd is the checker).
"""
import ctypes as C

import torch

from ... import _lib
from ... import engine as E
from ..._lib import DrcWgradParams



class Grads:
    """Gradient buffers (Blocked, same geometry as the forward tensors) with assign-or-accumulate bookkeeping."""

    def __init__(self, ws, device):
        self.ws, self.device = ws, device
        self.buf = ws.setdefault("gbuf", {})
        self.have = set()

    def get(self, name):
        ft = self.ws["t"][name]
        if isinstance(ft, E.BlockedSlice):                       # gradient of a concat slice = slice of the concat's gradient
            parent = [k for k, v in self.ws["t"].items() if v is ft.base][0]
            return E.BlockedSlice(self.get(parent), ft.cb_off, ft.C)
        b = self.buf.get(name)
        if b is None:
            b = self.ws["pool"].blocked(("g", name), ft.N, ft.C, ft.D, ft.H, ft.W, ft.pd, ft.ph, ft.pw)
            self.buf[name] = b
        return b


# bytes of LDS a wave's (a, b) tile pair may take.  The kernel is bound by each wave's serial LDS-read -> MFMA chain, so waves per SIMD
# matter more than tile size: 12 KiB (3 waves per SIMD) instead of rounds 1-2's 38 KiB (1) took the Config-B train step (8 crops) from
# 50.9 to 44.1 ms; 9 KiB tiles are too ragged (47.4).  Double-buffering the staging inside a wave (tried, round 3) halves the occupancy
# again and lost on every layer.  Layers with no tile under the cap (wide strided taps) fall back to 38 KiB.
WGRAD_TILE = {"lds_cap": 12 * 1024, "lds_max": 38 * 1024}
WGRAD_PARTIALS = {"enabled": True}     # partial sums + fixed-order reduce instead of the atomicAdd flush (tests flip it)


def wgrad(a, b, cls, in_mul, R=None, WT=None):
    """gw[cbA*16][cbB*16][T] for one tap class (see include/disprcnn_hip.h: drc_tapconv_wgrad)."""
    dev = a.device
    nd, nh, nw = cls["n"]
    T = nd * nh * nw
    gw = torch.empty(a.cb * 16, b.cb * 16, T, dtype=torch.float32, device=dev)
    p = DrcWgradParams()
    p.overwrite = 1                       # no zero-fill launch: the kernels store (or clear gw themselves on the atomicAdd path)
    span_h, span_w = (nh - 1) * cls["step"][1], (nw - 1) * cls["step"][2]
    # tile: R rows x WT cols of b with (rows_in*seg + R*WT)*64 B <= 38 KiB per wave
    best = None
    # (strided taps -- in_mul = 2: the stride-2 convolutions and the transposed ones -- keep the large tiles: their a tile is 4x the b tile
    # and small caps leave only ragged tiles; Config-A step 17.19 vs 17.33 ms)
    for cap in ((WGRAD_TILE["lds_cap"], WGRAD_TILE["lds_max"]) if in_mul == 1 else (WGRAD_TILE["lds_max"],)):
        for wt in sorted({-(-b.W // k) for k in range(1, b.W + 1) if -(-b.W // k) <= 112}):
            for r in range(1, min(b.H, 112 // wt) + 1):
                need = ((in_mul * (r - 1) + span_h + 1) * (in_mul * (wt - 1) + span_w + 1) + r * wt) * 64
                if need > cap:
                    continue
                eff = (b.H * b.W) / ((-(-b.H // r)) * (-(-b.W // wt)) * (-(-(r * wt) // 4)) * 4)
                key = (round(eff, 3), r * wt, wt)
                if best is None or key > best[0]:
                    best = (key, r, wt, need)
        if best is not None:
            break
    if best is None:
        raise ValueError("wgrad: no tile fits LDS")
    _, p.R, p.WT, need = best
    p.lds_bytes_per_wave = (need + 1023) // 1024 * 1024
    p.a, p.b, p.gw = E._base_ptr(a), E._base_ptr(b), gw.data_ptr()
    p.a_n_stride, p.a_cb_stride, p.a_d_stride, p.a_h_stride = a.n_stride, a.cb_stride, a.d_stride, a.h_stride
    p.b_n_stride, p.b_cb_stride, p.b_d_stride, p.b_h_stride = b.n_stride, b.cb_stride, b.d_stride, b.h_stride
    p.b_off0 = b.interior_off
    p.N, p.OD, p.OH, p.OW = b.N, b.D, b.H, b.W
    p.in_mul, p.cb_a, p.cb_b = in_mul, a.cb, b.cb
    p.nd, p.nh, p.nw = nd, nh, nw
    p.dd0, p.dh0, p.dw0 = cls["first"]
    p.sd, p.sh, p.sw = cls["step"]
    if WGRAD_PARTIALS["enabled"]:
        scratch = E.wgrad_scratch(dev)
        p.scratch, p.scratch_floats = scratch.data_ptr(), scratch.numel()
    st = _lib.lib().drc_tapconv_wgrad(C.byref(p), E._stream_ptr(dev))
    _lib.check(st, "drc_tapconv_wgrad")
    return gw


class RegressorBackward:
    """Reverse pass over the tape of one train-mode forward_features call."""

    def __init__(self, rt, ws, W):
        self.rt, self.ws, self.W, self.dev = rt, ws, W, rt.device
        self.pg = {}     # id(param) -> grad tensor

    def _padd(self, param, g):
        k = id(param)
        self.pg[k] = g if k not in self.pg else self.pg[k] + g

    # ---------------------------------------------------------------- data-gradient plans (cached in the workspace)
    def _dgrad(self, plan_name, c, g_out, g_in):
        """plan + transformed weights computing d/dx of the site's convolution (see module docstring)."""
        cache = self.ws.setdefault("dplans", {})
        fwd = self.ws["p"][plan_name]
        k0 = fwd.p.cls[0]
        stride, deconv, is3d = fwd.p.in_mul, fwd.p.out_mul == 2, g_out.pd > 0
        wt = c.conv.weight.detach().to(self.dev).float()
        ent = cache.get(plan_name)
        if ent is None:
            if is3d:
                if deconv:         # ConvTranspose [Cin,Cout,k]: dx = conv_s2(dy, W read as Conv[out=Cin,in=Cout])
                    pl = E.plan_conv3d(g_out, g_in, 2, c.cin, False)
                elif stride == 2:   # dx = conv_transpose(dy, W [in=Cout,out=Cin,k])
                    pl = E.plan_deconv3d(g_out, g_in, c.cin, False)
                else:
                    pl = E.plan_conv3d(g_out, g_in, 1, c.cin, False)
            else:
                k, dil = k0.nh, max(k0.sh, 1)
                if stride == 2:
                    pl = E.plan_deconv2d(g_out, g_in, k, c.cin, False)
                else:               # same dilation, flipped kernel, pad' = dil*(k-1) - pad_fwd  (pad_fwd = x.ph - dh0)
                    pad_fwd = self.ws["t"][self._x_of[plan_name]].ph - k0.dh0
                    pl = E.plan_conv2d(g_out, g_in, k, 1, dil * (k - 1) - pad_fwd, dil, c.cin, False)
            ent = dict(plan=pl)
            cache[plan_name] = ent
        # transformed weights, both packings from one launch (drc_pack_weights: in/out swap and tap flip are index arithmetic)
        pl = ent["plan"]
        need16 = pl.needs_t16
        if deconv:                      # ConvTranspose weight [Cin,Cout,k] read as Conv[out=Cin, in=Cout]
            wp, w16 = E.pack_layouts(wt, False, False, want_t16=need16)
        elif stride == 2:               # Conv weight [Cout,Cin,k] as the transposed conv's [in=Cout, out=Cin]
            if pl.deconv_direct:
                wp, w16 = E.pack_layouts(wt, True, False, want_t16=False)[0], pl.pack16(wt, True, False)
            else:
                wp, w16 = E.pack_layouts(wt, True, False, want_t16=need16)
        elif pl.pointwise:              # 1x1: in/out swap only
            wp, w16 = E.pack_layouts(wt, True, False, want_tap=False)[1][0], None
        elif pl.wino:                   # stride 1 as Winograd: in/out swap + flipped taps inside the weight transform
            wp, w16 = E.pack_layouts(wt, True, True, want_t16=False)[0], pl.pack16(wt, True, True)
        else:                           # stride 1: in/out swap + flipped taps
            wp, w16 = E.pack_layouts(wt, True, True, want_t16=need16)
        cp = ent["plan"].p.cout_pad
        unit = self.ws.setdefault("unit_affine", {})
        if cp not in unit:
            unit[cp] = (torch.ones(cp, device=self.dev), torch.zeros(cp, device=self.dev))
        return ent["plan"], wp, unit[cp][0], unit[cp][1], w16

    _x_of = {}

    def site(self, G, plan, wname, x, y, res, need_dx=True):
        t, c = self.ws["t"], self.W[wname]
        fwd = self.ws["p"][plan]
        k0 = fwd.p.cls[0]
        self._x_of[plan] = x
        relu = bool(fwd.p.relu)
        dy = G.get(y)
        assert y in G.have, f"no gradient reached {y}"
        lib, sp = _lib.lib(), E._stream_ptr(self.dev)
        xin, yt = t[x], t[y]
        if c.bn is not None:
            raw = self.ws["raw"][plan]
            mean, invstd, M = self.ws["saved"][plan]
            sums = torch.empty(2, raw.cb * 16, dtype=torch.float32, device=self.dev)
            st = lib.drc_bn_bwd_reduce(E._ptr(dy.storage), E._geom8(dy), E._ptr(yt.storage), E._geom8(yt), E._ptr(raw.storage), E._geom8(raw),
                                       E._ptr(mean), E._ptr(invstd), int(relu), E._ptr(sums), E._ptr(E.bn_scratch(self.dev, raw.cb)), sp)
            _lib.check(st, "drc_bn_bwd_reduce")
            draw = self.ws.setdefault("draw", {}).get(plan)
            if draw is None:
                halo = (1, 1, 1) if raw.pd > 0 else (0, 2, 2)      # 2D: room for the dilated data-gradient taps
                draw = self.ws["pool"].blocked(("draw", plan), raw.N, raw.C, raw.D, raw.H, raw.W, *halo)
                self.ws["draw"][plan] = draw
            dres, acc = None, 0
            if res is not None:
                dres = G.get(res)
                acc = int(res in G.have)
                G.have.add(res)
            st = lib.drc_bn_bwd_apply(E._ptr(dy.storage), E._geom8(dy), E._ptr(yt.storage), E._geom8(yt), E._ptr(raw.storage), E._geom8(raw),
                                      E._ptr(mean), E._ptr(invstd), E._ptr(c.gamma), E._ptr(sums), 1.0 / M, int(relu), E._ptr(draw.storage),
                                      E._geom8(draw), E._ptr(dres.storage) if dres is not None else None,
                                      E._geom8(dres) if dres is not None else None, acc, sp)
            _lib.check(st, "drc_bn_bwd_apply")
            self._padd(c.bn.weight, sums[1, : c.cout])           # `sums` is this site's own tensor: views are safe to keep
            self._padd(c.bn.bias, sums[0, : c.cout])
        else:                       # plain conv (no BN, no activation, no residual): the output gradient is the conv gradient
            assert not relu and res is None
            draw = dy
            if isinstance(draw, E.BlockedSlice) or draw.ph < 1:
                raise NotImplementedError("BN-less site with an unsupported gradient layout")
        # weight gradient
        deconv = fwd.p.out_mul == 2
        if deconv:
            cls = dict(n=(3, 3, 3), first=(0, 0, 0), step=(1, 1, 1))            # a = draw (halo 1): padded index 2i + k
            gw = wgrad(draw, xin, cls, 2)                                        # [co][ci][k]
            self._padd(c.conv.weight, gw[: c.cout, : c.cin].permute(1, 0, 2).reshape(c.conv.weight.shape))
        else:
            cls = dict(n=(k0.nd, k0.nh, k0.nw), first=(k0.dd0, k0.dh0, k0.dw0), step=(max(k0.sd, 1), max(k0.sh, 1), max(k0.sw, 1)))
            gw = wgrad(xin, draw, cls, fwd.p.in_mul)                             # [ci][co][t]
            self._padd(c.conv.weight, gw[: c.cin, : c.cout].permute(1, 0, 2).reshape(c.conv.weight.shape))
        # data gradient into grads[x]
        if need_dx:
            gx = G.get(x)
            pl, wp, ones, zeros, w16 = self._dgrad(plan, c, draw, gx)
            first = x not in G.have
            if first and pl.p.out_mul == 2 and pl.p.n_classes < (8 if gx.pd > 0 else 4):
                gx.storage.zero_()                                               # k=1 stride-2: odd positions get no tap
            pl.run(draw, wp, ones, zeros, gx, None if first else gx, w16=w16)
            G.have.add(x)

    def run(self, tape, gpreds, costs, mx, mn, out_hw):
        ws, t, dev = self.ws, self.ws["t"], self.dev
        G = Grads(ws, dev)
        lib, sp = _lib.lib(), E._stream_ptr(dev)
        N, Dp, Hp, Wp = costs[0].shape
        H, W_ = out_hw
        # heads: d loss / d cost_k (dense), cumulative: cost2 = classif2 + cost1, cost3 = classif3 + cost2
        gc = []
        nfoot = lib.drc_upsample_softargmin_bwd_scratch_floats(N, Dp, Hp, Wp, H, W_)
        for k in range(3):
            if gpreds[k] is not None:
                g = torch.empty_like(costs[k])           # overwritten: tile footprints gathered in a fixed order (no atomics)
                foot = E.scratch(dev, "softargmin_bwd", nfoot)
                gk = gpreds[k].contiguous().float()      # (named: a temporary's block could be reused before the launch)
                st = lib.drc_upsample_softargmin_bwd(E._ptr(costs[k]), E._ptr(gk), E._ptr(g), N, Dp, Hp, Wp,
                                                     mx - mn, H, W_, mn, E._ptr(foot), foot.numel(), sp)
                _lib.check(st, "drc_upsample_softargmin_bwd")
            else:
                g = torch.zeros_like(costs[k])
            gc.append(g)
        gc[1] = gc[1] + gc[2]
        gc[0] = gc[0] + gc[1]
        for k in (1, 2, 3):
            name = f"cls_t{k}"
            gx = G.get(name)
            w27 = self.W[f"classif{k}.2"]
            st = lib.drc_conv3d_cout1_bwd_data(E._ptr(gc[k - 1]), E._ptr(w27), E._ptr(gx.storage), N, gx.cb, Dp, Hp, Wp, 0, sp)
            _lib.check(st, "drc_conv3d_cout1_bwd_data")
            G.have.add(name)
            gw = torch.empty_like(w27)                   # overwritten: per-block partials added in block order (no atomics)
            st = lib.drc_conv3d_cout1_bwd_weight(E._ptr(t[name].storage), E._ptr(gc[k - 1]), E._ptr(gw), N, gx.cb, Dp, Hp, Wp,
                                                 E._ptr(E.scratch(dev, "cout1_wgrad", _lib.cout1_wgrad_scratch_floats(gx.cb))), sp)
            _lib.check(st, "drc_conv3d_cout1_bwd_weight")
            conv = getattr(self.rt.model, f"classif{k}")[2]
            self._padd(conv.weight, gw[:, :32].t().reshape(conv.weight.shape))
        for op in reversed(tape):
            _, ws_, plan, wname, x, y, res = op
            if ws_ is ws:
                self.site(G, plan, wname, x, y, res, need_dx=(x != "cost" or self.rt._need_input_grad))
        return G


class FeaturesBackward(RegressorBackward):
    """Reverse pass over the 2D feature CNN of ONE view (reference submodule.py:106-139): lastconv -> SPP branches
    (bilinear-up adjoint, 1x1 conv+BN site, avg-pool adjoint) -> layer4..layer1 -> firstconv."""

    def __init__(self, rt, ws, W, pg):
        super().__init__(rt, ws, W)
        self.pg = pg                 # shared with the regressor / the other view: weights are siamese

    def run(self, tape, gfeat):
        ws, t, dev = self.ws, self.ws["t"], self.dev
        G = Grads(ws, dev)
        lib, sp = _lib.lib(), E._stream_ptr(dev)
        G.get("feat").from_dense(gfeat)
        G.have.add("feat")
        slot = {name: cb_off for name, k, oh, ow, cb_off in ws["spp"]}
        kk = {name: k for name, k, oh, ow, cb_off in ws["spp"]}
        skip = ws["skip"]
        sites = [op for op in tape if op[1] is ws]
        for op in reversed(sites):
            _, _, plan, wname, x, y, res = op
            if plan.startswith("fe.branch"):
                name = plan[3:]
                gconv = G.get(name + ".conv")
                gconv.storage.zero_()
                gslice = E.BlockedSlice(G.get("cat"), slot[name], 32)
                st = lib.drc_bilinear_up_blocked_bwd(E._ptr(gslice.storage), E._geom8(gslice), E._ptr(gconv.storage), E._geom8(gconv), sp)
                _lib.check(st, "drc_bilinear_up_blocked_bwd")
                G.have.add(name + ".conv")
                self.site(G, plan, wname, x, y, res)
                gpool, gskip = G.get(name + ".pool"), G.get(skip)
                st = lib.drc_avgpool2d_blocked_bwd(E._ptr(gpool.storage), E._geom8(gpool), E._ptr(gskip.storage), E._geom8(gskip), kk[name], sp)
                _lib.check(st, "drc_avgpool2d_blocked_bwd")
                continue
            self.site(G, plan, wname, x, y, res, need_dx=(x != "img"))
            if plan == "fe.lastconv.0":          # its data gradient filled every channel block of the concat
                G.have.update(k for k, v in t.items() if isinstance(v, E.BlockedSlice))
        return G
