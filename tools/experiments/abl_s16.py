"""Round-5 ablation of convs16.hip's step (timing only, results are wrong by construction): which part of a step is NOT hidden behind the
MFMAs.  `python tools/experiments/abl_s16.py build` (here: hipcc, no GPU) writes tools/exp_libs/libs16_<variant>.so; `... run` (GPU box)
times the 32->32 layer at 1024 ROIs with each."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
SRC = os.path.join(ROOT, "disprcnn_amd", "csrc", "convs16.hip")
OUT = os.path.join(ROOT, "tools", "exp_libs")
VARIANTS = {
    "base": [],
    "nofin": [("            float s_;\n            if constexpr (KW == 2) s_ = part", "            float s_; v[e] = 0.f; vh[e] = vl[e] = (_Float16)0.f; return;\n            if constexpr (KW == 2) s_ = part")],
    "nopub": [("            for (int q = 0; q < 4; ++q) *(f32x4*)(xb + q * 1024) = (f32x4){a[q * 4], a[q * 4 + 1], a[q * 4 + 2], a[q * 4 + 3]};", "            for (int q = 0; q < 4; ++q) asm volatile(\"\" :: \"v\"(a[q * 4]), \"v\"(xb));")],
    "nobar": [("        __builtin_amdgcn_s_barrier();\n        asm volatile(\"\" ::: \"memory\");\n        // contexts of the planes finalized", "        asm volatile(\"\" ::: \"memory\");\n        // contexts of the planes finalized")],
    "nostage": [("                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(dst + (cb * 8 + c) * CPB + h * 1024), 16, va[h], so, 0, 0);\n            }", "                asm volatile(\"\" :: \"v\"(va[h]), \"s\"(so), \"s\"(dst));\n            }")],
    "nostore": [("        auto stores = [&]() __attribute__((always_inline)) {\n", "        auto stores = [&]() __attribute__((always_inline)) {\n            return;\n")],
}
VARIANTS["nofin_nopub"] = VARIANTS["nofin"] + VARIANTS["nopub"]
VARIANTS["nofin_nopub_nobar"] = VARIANTS["nofin"] + VARIANTS["nopub"] + VARIANTS["nobar"]
VARIANTS["mfma_only"] = VARIANTS["nofin"] + VARIANTS["nopub"] + VARIANTS["nobar"] + VARIANTS["nostage"] + VARIANTS["nostore"]


# ---- the transposed-conv kernel (convs16u.hip), hourglass conv6 shape (64 -> 32, input 6 x 14 x 14), same method
SRC_U = os.path.join(ROOT, "disprcnn_amd", "csrc", "convs16u.hip")
VARIANTS_U = {
    "u_base": [],
    "u_nostore": [("                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi[s]), y16r, fo[ci] + (unsigned)((s * 2) * o_chunkB), 0, 0);\n"
                   "                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo[s]), y16r, fo[ci] + (unsigned)((4 + s * 2) * o_chunkB), 0, 0);",
                   "                    { const u32x4 h4 = __builtin_bit_cast(u32x4, hi[s]), l4 = __builtin_bit_cast(u32x4, lo[s]);\n"
                   "                      asm volatile(\"\" :: \"v\"(h4.x), \"v\"(h4.y), \"v\"(h4.z), \"v\"(h4.w), \"v\"(l4.x), \"v\"(l4.y), \"v\"(l4.z), \"v\"(l4.w), \"v\"(fo[ci])); }")],   # (values kept alive: without them the MFMAs are dead code)
    "u_nores": [("                    resv[S_][ci][q] = __builtin_amdgcn_raw_buffer_load_b128(resr, fo[ci] + (unsigned)(((q >> 1) * 4 + (q & 1) * 2) * o_chunkB), 0, 0);",
                 "                    resv[S_][ci][q] = (u32x4){0u, 0u, 0u, 0u};")],
    "u_nobar": [("            __builtin_amdgcn_s_barrier();\n            asm volatile(\"\" ::: \"memory\");\n            // output planes finished in this step", "            asm volatile(\"\" ::: \"memory\");\n            // output planes finished in this step")],
    "u_nostage": [("                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + srcoff), LDS_PTR(dst + cc * CPB), 16, 0, 0);", "                asm volatile(\"\" :: \"v\"(src), \"s\"(dst));")],
}
VARIANTS_U["u_nostore_nores"] = VARIANTS_U["u_nostore"] + VARIANTS_U["u_nores"]
VARIANTS_U["u_mfma_only"] = VARIANTS_U["u_nostore"] + VARIANTS_U["u_nores"] + VARIANTS_U["u_nobar"] + VARIANTS_U["u_nostage"]


def build():
    os.makedirs(OUT, exist_ok=True)
    if os.environ.get("WHICH") == "u":
        srcu = open(SRC_U).read().replace('#include "../../include/disprcnn_hip.h"', '#include "%s"' % os.path.join(ROOT, "include", "disprcnn_hip.h"))
        for name, subs in VARIANTS_U.items():
            s = srcu
            for a, b in subs:
                assert a in s, (name, a[:70])
                s = s.replace(a, b)
            f = os.path.join(OUT, f"s16_{name}.hip")
            open(f, "w").write(s)
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-pass-failed", "-shared",
                                   "-o", os.path.join(OUT, f"libs16_{name}.so"), f])
            os.remove(f)
            print("built", name, flush=True)
        return
    src = open(SRC).read().replace('#include "../../include/disprcnn_hip.h"', '#include "%s"' % os.path.join(ROOT, "include", "disprcnn_hip.h"))
    for name, subs in VARIANTS.items():
        s = src
        for a, b in subs:
            assert a in s, (name, a[:60])
            s = s.replace(a, b)
        f = os.path.join(OUT, f"s16_{name}.hip")
        open(f, "w").write(s)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-pass-failed", "-shared",
                               "-I", os.path.dirname(SRC), "-o", os.path.join(OUT, f"libs16_{name}.so"), f])
        os.remove(f)
        print("built", name, flush=True)


def run_u():
    import torch
    sys.path.insert(0, ROOT)
    from disprcnn_amd import s16
    from disprcnn_amd._lib import DrcS16ConvParams
    dev = torch.device("cuda:0")
    N, cin, cout, D, H, W = 1024, 64, 32, 6, 14, 14
    w = torch.randn(cin, cout, 3, 3, 3) * 0.05
    wp, wexp = s16.pack_weight_s16(w.to(dev).transpose(0, 1).contiguous())
    sc = torch.full((cout,), 2.0 ** -wexp, device=dev); sh = torch.zeros(cout, device=dev)

    def rnd(cb, d, h, w_):
        t = torch.zeros(N, cb, d + 2, h + 2, 8, w_ + 2, 8, dtype=torch.float16, device=dev)
        t[:, :, 1:d + 1, 1:h + 1, :, 1:w_ + 1].normal_()
        t[:, :, 1:d + 1, 1:h + 1, 4:, 1:w_ + 1] *= 2.0 ** -11
        return t
    x, res = rnd(2, D, H, W), rnd(1, 2 * D, 2 * H, 2 * W)
    y = torch.zeros_like(res)
    P = lambda t: C.c_void_p(t.data_ptr())
    p = DrcS16ConvParams(P(x), P(wp), P(sc), P(sh), P(res), P(y), None, None, None, N, D, H, W, cin, cout, 0, 0)
    for name in VARIANTS_U:
        lib = C.CDLL(os.path.join(OUT, f"libs16_{name}.so"))
        fn = lib.drc_deconv3d_k3s2_s16_fwd
        fn.restype = C.c_int; fn.argtypes = [C.POINTER(DrcS16ConvParams), C.c_void_p]
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for _ in range(3):
            assert fn(C.byref(p), st) == 0
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record(); fn(C.byref(p), st); b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        print(f"{name:22s} median {ts[5] * 1e3:8.1f} us  min {ts[0] * 1e3:8.1f} us", flush=True)


def run():
    if os.environ.get("WHICH") == "u":
        return run_u()
    import torch
    sys.path.insert(0, ROOT)
    from disprcnn_amd import s16
    from disprcnn_amd._lib import DrcS16ConvParams
    dev = torch.device("cuda:0")
    N, cin, cout, D, H, W = 1024, 32, 32, 12, 28, 28
    w = torch.randn(cout, cin, 3, 3, 3) * 0.05
    wp, wexp = s16.pack_weight_s16(w.to(dev))
    sc = torch.full((cout,), 2.0 ** -wexp, device=dev); sh = torch.zeros(cout, device=dev)
    x = torch.zeros(N, 1, D + 2, H + 2, 8, W + 2, 8, dtype=torch.float16, device=dev)
    x[:, :, 1:D + 1, 1:H + 1, :, 1:W + 1].normal_()
    x[:, :, 1:D + 1, 1:H + 1, 4:, 1:W + 1] *= 2.0 ** -11
    y = torch.zeros_like(x)
    P = lambda t: C.c_void_p(t.data_ptr())
    p = DrcS16ConvParams(P(x), P(wp), P(sc), P(sh), None, P(y), None, None, None, N, D, H, W, cin, cout, 1, 0)
    zero = os.environ.get("ZERO") == "1"
    if zero:
        x.zero_()
    if os.environ.get("CV") == "1":         # the cost-volume form (dres0[0]: 64 -> 32, the volume built from two 2D maps)
        cin = 64
        w = torch.randn(cout, cin, 3, 3, 3) * 0.05
        wp, wexp = s16.pack_weight_s16(w.to(dev))
        sc = torch.full((cout,), 2.0 ** -wexp, device=dev)
        L = torch.zeros(N, 1, 1, H + 2, 8, W + 2, 8, dtype=torch.float16, device=dev)
        L[:, :, :, 1:H + 1, :, 1:W + 1].normal_()
        L[:, :, :, 1:H + 1, 4:, 1:W + 1] *= 2.0 ** -11
        R = L.flip(0).contiguous()
        p = DrcS16ConvParams(None, P(wp), P(sc), P(sh), None, P(y), None, P(L), P(R), N, D, H, W, cin, cout, 1, 0)
    only = os.environ.get("ONLY")
    for name in (only.split(",") if only else (["base", "mfma_only"] if zero else VARIANTS)):
        lib = C.CDLL(os.path.join(OUT, f"libs16_{name}.so"))
        fn = lib.drc_conv3d_k3_s16_fwd
        fn.restype = C.c_int; fn.argtypes = [C.POINTER(DrcS16ConvParams), C.c_void_p]
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for _ in range(3):
            assert fn(C.byref(p), st) == 0
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record(); fn(C.byref(p), st); b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        print(f"{name:22s} median {ts[5] * 1e3:8.1f} us  min {ts[0] * 1e3:8.1f} us", flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
