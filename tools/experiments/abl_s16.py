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
    "nobar": [("        __builtin_amdgcn_s_barrier();\n        asm volatile(\"\" ::: \"memory\");\n        {   // slab t+2", "        asm volatile(\"\" ::: \"memory\");\n        {   // slab t+2")],
    "nostage": [("                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(dst + (cb * 8 + c) * CPB + h * 1024), 16, va[h], so, 0, 0);\n            }", "                asm volatile(\"\" :: \"v\"(va[h]), \"s\"(so), \"s\"(dst));\n            }")],
    "nostore": [("        auto stores = [&]() __attribute__((always_inline)) {\n", "        auto stores = [&]() __attribute__((always_inline)) {\n            return;\n")],
}
# idle lanes (28..31 of a 28-voxel row) read a zeroed LDS region instead of their neighbours' voxels: MFMA array toggling (DVFS) test
VARIANTS["zidle"] = [
    ("    const unsigned bfrag = (unsigned)(((k >> 1) * 8 + (k & 1) * 2 + g) * CPB + ((r * RT + rl) * SX + xl) * 16);",
     "    for (int i_ = threadIdx.x; i_ < RING * SLAB / 16; i_ += 256) ((f32x4*)(lds + RING * SLAB + 2 * 4 * XW))[i_] = (f32x4){0.f, 0.f, 0.f, 0.f};\n"
     "    __syncthreads();\n"
     "    const unsigned bfrag = (n_ >= RT * WT ? (unsigned)(RING * SLAB + 2 * 4 * XW) : 0u) + (unsigned)(((k >> 1) * 8 + (k & 1) * 2 + g) * CPB + ((r * RT + rl) * SX + xl) * 16);"),
    ("    constexpr size_t lds = RING * SLAB + 2 * 4 * XW;", "    constexpr size_t lds = RING * SLAB + 2 * 4 * XW + (KW == 2 ? RING * SLAB : 0);"),
]
VARIANTS["nofin_nopub"] = VARIANTS["nofin"] + VARIANTS["nopub"]
VARIANTS["nofin_nopub_nobar"] = VARIANTS["nofin"] + VARIANTS["nopub"] + VARIANTS["nobar"]
VARIANTS["mfma_only"] = VARIANTS["nofin"] + VARIANTS["nopub"] + VARIANTS["nobar"] + VARIANTS["nostage"] + VARIANTS["nostore"]


def build():
    os.makedirs(OUT, exist_ok=True)
    src = open(SRC).read().replace('#include "../../include/disprcnn_hip.h"', '#include "%s"' % os.path.join(ROOT, "include", "disprcnn_hip.h"))
    for name, subs in VARIANTS.items():
        s = src
        for a, b in subs:
            assert a in s, (name, a[:60])
            s = s.replace(a, b)
        f = os.path.join(OUT, f"s16_{name}.hip")
        open(f, "w").write(s)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-pass-failed", "-shared",
                               "-o", os.path.join(OUT, f"libs16_{name}.so"), f])
        os.remove(f)
        print("built", name, flush=True)


def run():
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
