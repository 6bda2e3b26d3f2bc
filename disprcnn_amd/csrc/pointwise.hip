// pointwise.hip -- 1x1 convolution (+BN/bias, +residual, +ReLU) as a register-blocked GEMM on the fp32 MFMA (gfx950 / CDNA4).
//
//   reference: the 1x1 convolutions of ResNet-50-FPN (backbone/resnet.py Bottleneck conv1/conv3/downsample, backbone/fpn.py
//   inner blocks) and of the PSMNet feature CNN (submodule.py downsample / SPP branches / lastconv[2]).
//
// A 1x1 conv has no tap reuse, so staging tiles through LDS (tapconv.hip) only adds LDS-DMA issue cost: 63 % of the
// R-50-FPN trunk ran at 7-13 TFLOP/s.  Here both MFMA operands come straight from global memory as coalesced float4s:
//   B = x[voxel][16-channel block]: lane (voxel j, g) loads channels 4g..4g+3 -> one 1 KiB transaction per 16 voxels
//   A = w[cb][cout][16]          : lane (cout j, g)  loads channels 4g..4g+3 -> one 1 KiB transaction per 16 couts
// and MFMA k-step s of a block uses channel 4g+s on both sides.  A wave owns VT*16 output voxels x CT*16 output channels and
// walks the input channel blocks with the next block's operands in flight; no LDS, 2-3 waves per SIMD hide the L2 latency.
//
// Wave -> work mapping (round 3): work items are numbered so that each XCD (block b runs on XCD b % 8, each with its own L2) walks
// one contiguous range, with the cout groups of one voxel group adjacent when the activations are the larger operand (M > cout:
// x is then read from HBM once instead of once per cout group -- 256 -> 256 on 94 x 310 moved 4 x 60 MB) and the voxel groups of
// one cout group adjacent when the weights are (the 12 x 39 maps of layer4).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define PW_WAVES 4
#ifndef PW_OCC_BIG
#define PW_OCC_BIG 2
#endif
#ifndef PW_OCC_8
#define PW_OCC_8 3
#endif
#ifndef PW_PF_BIG
#define PW_PF_BIG 2
#endif
#ifndef PW_PF_MID
#define PW_PF_MID 2
#endif
#ifndef PW_PF_SMALL
#define PW_PF_SMALL 4
#endif

namespace {

template <int VT, int CT>
__global__ __launch_bounds__(64 * PW_WAVES) __attribute__((amdgpu_waves_per_eu(VT * CT >= 16 ? PW_OCC_BIG : (VT * CT >= 8 ? PW_OCC_8 : 4))))
void pointwise_kernel(const drc_tapconv_params p, const int nvg, const int ncg,
                                                                     const int cg_fast) {
    // operand sets in flight ahead of the MFMAs: the smaller the tile, the shorter one block's MFMA burst (16 x VT x CT x 8 cycles),
    // so the further ahead the loads have to run to cover an L2/HBM round trip
    constexpr int PF = VT * CT >= 16 ? PW_PF_BIG : (VT * CT >= 4 ? PW_PF_MID : PW_PF_SMALL);
    constexpr int NB = PF + 1;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    const unsigned M = (unsigned)p.N * p.OH * p.OW;               // output voxels (host checks < 2^31)
    const int chunk = gridDim.x >> 3;                             // blocks per XCD
    const unsigned id = ((blockIdx.x & 7) * (unsigned)chunk + (blockIdx.x >> 3)) * PW_WAVES + wave;
    if (id >= (unsigned)nvg * (unsigned)ncg) return;
    const int cg = cg_fast ? (int)(id % (unsigned)ncg) : (int)(id / (unsigned)nvg);
    const unsigned vg = cg_fast ? id / (unsigned)ncg : id % (unsigned)nvg;
    const unsigned vtile0 = vg * VT;                              // first 16-voxel tile of this wave
    const int ct0 = cg * CT;
    const int s_in = p.in_mul;

    // addressing: a wave-uniform 64-bit base (the sample of the wave's first voxel) + 32-bit per-lane byte offsets, so every load is
    // `global_load v, v_off, s[base]` with no 64-bit vector arithmetic (the launcher checks that a wave's voxels stay within 4 GB)
    const unsigned hw = (unsigned)p.OH * (unsigned)p.OW;
    const unsigned n0 = (vtile0 * 16 < M ? vtile0 * 16 : M - 1) / hw;
    const char* xs = (const char*)(p.x + (int64_t)n0 * p.x_n_stride);
    char* ys = (char*)(p.y + p.y_off0 + (int64_t)n0 * p.y_n_stride);
    const char* rs = (const char*)(p.res + p.r_off0 + (int64_t)n0 * p.r_n_stride);
    unsigned xo[VT], yo[VT], ro[VT];
    bool ok[VT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
        unsigned v = (vtile0 + vt) * 16 + j;
        ok[vt] = v < M;
        v = ok[vt] ? v : M - 1;
        const unsigned row = v / (unsigned)p.OW;
        const unsigned ow = v - row * (unsigned)p.OW;
        const unsigned n = row / (unsigned)p.OH;
        const unsigned oh = row - n * (unsigned)p.OH;
        const unsigned dn = n - n0;
        xo[vt] = 4u * (dn * (unsigned)p.x_n_stride + (s_in * oh + p.cls[0].dh0) * (unsigned)p.x_h_stride + (s_in * ow + p.cls[0].dw0) * 16u + g * 4u);
        yo[vt] = 4u * (dn * (unsigned)p.y_n_stride + oh * (unsigned)p.y_h_stride + ow * 16u + g * 4u);
        ro[vt] = 4u * (dn * (unsigned)p.r_n_stride + oh * (unsigned)p.r_h_stride + ow * 16u + g * 4u);
    }
    // weights [cb][cout_pad][16]: lane (cout j, g).  Both operands come through buffer loads: descriptor in SGPRs, the lane offset
    // a loop-invariant VGPR and the channel block a scalar offset, so the K loop holds no address arithmetic at all, and a block past
    // the last one (the prefetch of the final iterations) reads zeros instead of faulting.
    const unsigned wo = 4u * ((ct0 * 16u + j) * 16u + g * 4u);
    const unsigned w_cb = (unsigned)p.cout_pad * 64u;             // bytes per input channel block
    const unsigned x_cb = (unsigned)p.x_cb_stride * 4u;
    const int64_t x_left = ((int64_t)p.N - n0) * p.x_n_stride * 4;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xs, 0, x_left < 0xffffffffLL ? (int)x_left : -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)((unsigned)p.cb_in * w_cb), 0x00020000);

    f32x4 acc[VT][CT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[vt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 a[NB][CT], b[NB][VT];
    auto fetch = [&](int set, int cb) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            a[set][ct] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, wo + ct * 1024, cb * w_cb, 0));
#pragma unroll
        for (int vt = 0; vt < VT; ++vt)
            b[set][vt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, xo[vt], cb * x_cb, 0));
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) fetch(u, u);

    auto mfmas = [&](int u) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[vt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][ct][s], b[u][vt][s], acc[vt][ct], 0, 0, 0);
    };
    // whole groups of NB blocks as one branch-free body (so the compiler's s_waitcnt placement sees the ring: each step waits only
    // for its own set, PF blocks behind the newest request), then the < NB blocks left, which are already in flight
    int cb0 = 0;
    for (; cb0 + NB <= p.cb_in; cb0 += NB) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            fetch((u + PF) % NB, cb0 + u + PF);
            __builtin_amdgcn_sched_barrier(0);                    // hipcc otherwise sinks these loads to their first use
            mfmas(u);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const int rem = p.cb_in - cb0;
#pragma unroll
    for (int u = 0; u < NB - 1; ++u)
        if (u < rem) mfmas(u);

    // epilogue: lane (voxel j, g) holds couts 4g..4g+3 of each tile; one cout tile's residuals are requested together
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const f32x4 sc = *(const f32x4*)(p.scale + (ct0 + ct) * 16 + g * 4), sh = *(const f32x4*)(p.shift + (ct0 + ct) * 16 + g * 4);
        f32x4 r[VT];
#pragma unroll
        for (int vt = 0; vt < VT; ++vt)
            r[vt] = p.res ? *(const f32x4*)(rs + (int64_t)(ct0 + ct) * p.r_cb_stride * 4 + ro[vt]) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) {
            f32x4 v = acc[vt][ct] * sc + sh;
            if (p.res) v += r[vt];
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (ok[vt]) *(f32x4*)(ys + (int64_t)(ct0 + ct) * p.y_cb_stride * 4 + yo[vt]) = v;
        }
    }
}

template <int VT, int CT>
int launch(const drc_tapconv_params& p, hipStream_t s) {
    const long M = (long)p.N * p.OH * p.OW;
    const long tiles = (M + 15) / 16;
    const long nvg = (tiles + VT - 1) / VT;
    const int ncg = p.cout_pad / 16 / CT;
    const long blocks = (nvg * ncg + PW_WAVES - 1) / PW_WAVES;
    if (M >= 0x7fffffffL || (blocks + 7) / 8 * 8 * PW_WAVES > 0x7fffffffL) return -3;
    if ((long)p.cb_in * p.cout_pad * 64 >= (1L << 31) || p.x_n_stride >= (1L << 30)) return -3;
    // 32-bit lane offsets relative to the sample of a wave's first voxel: its VT*16 voxels span at most `span` samples
    const long span = VT * 16 / ((long)p.OH * p.OW) + 2;
    const long big = p.x_n_stride > p.y_n_stride ? p.x_n_stride : p.y_n_stride;
    if (span * (big > p.r_n_stride || !p.res ? big : p.r_n_stride) >= (1L << 30)) return -3;
    dim3 grid((unsigned)((blocks + 7) / 8 * 8), 1, 1);
    hipLaunchKernelGGL((pointwise_kernel<VT, CT>), grid, dim3(64 * PW_WAVES), 0, s, p, (int)nvg, ncg, M > (long)p.cout_pad ? 1 : 0);
    return (int)hipGetLastError();
}

}  // namespace

#ifndef PW_SMALL_WAVES
#define PW_SMALL_WAVES 1024
#endif
extern "C" int drc_conv2d_k1_fwd(const drc_tapconv_params* pp, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD != 1 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (p.cout_pad <= 0 || (p.cout_pad & 15) || p.cb_in <= 0) return -2;
    const drc_tap_class& k = p.cls[0];
    if (p.n_classes != 1 || (p.in_mul != 1 && p.in_mul != 2) || p.out_mul != 1 || k.nd != 1 || k.nh != 1 || k.nw != 1) return -4;
    hipStream_t s = (hipStream_t)stream;
    const int ct = p.cout_pad / 16;
    const long tiles = ((long)p.N * p.OH * p.OW + 15) / 16;
    // 4x4 tiles (64 voxels x 64 couts, 64 MFMAs per 8 loads) when that still gives >= 2 waves per SIMD; smaller otherwise
    if (ct % 4 == 0 && tiles / 4 * (ct / 4) >= 2048) return launch<4, 4>(p, s);
    if (ct % 2 == 0 && tiles / 4 * (ct / 2) >= 2048) return launch<4, 2>(p, s);
    // small maps with many channels (the trunk's 2048 -> 512 / 256 layers on 12 x 39: 30 voxel pairs x 16 cout pairs = 480 waves, each
    // walking K = 2048): one tile per wave fills the 1024 SIMDs (round 3)
    if (ct % 2 == 0 && (tiles / 2 * (ct / 2) >= PW_SMALL_WAVES || (long)p.cb_in < 16)) return launch<2, 2>(p, s);
    if (tiles / 2 * ct >= PW_SMALL_WAVES) return launch<2, 1>(p, s);
    return launch<1, 1>(p, s);
}
