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
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PW_WAVES 4

namespace {

template <int VT, int CT>
__global__ __launch_bounds__(64 * PW_WAVES) void pointwise_kernel(const drc_tapconv_params p) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    const long M = (long)p.N * p.OH * p.OW;                       // output voxels
    const long vtile0 = ((long)blockIdx.x * PW_WAVES + wave) * VT;   // first 16-voxel tile of this wave
    if (vtile0 * 16 >= M) return;
    const int ct0 = blockIdx.y * CT;
    const int s_in = p.in_mul;

    // per-lane input offsets (floats) of the wave's voxels at channel block 0, and output offsets
    int64_t xo[VT], yo[VT], ro[VT];
    bool ok[VT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
        long v = (vtile0 + vt) * 16 + j;
        ok[vt] = v < M;
        v = ok[vt] ? v : M - 1;
        const int ow = (int)(v % p.OW); v /= p.OW;
        const int oh = (int)(v % p.OH);
        const int n = (int)(v / p.OH);
        xo[vt] = (int64_t)n * p.x_n_stride + (int64_t)(s_in * oh + p.cls[0].dh0) * p.x_h_stride + (int64_t)(s_in * ow + p.cls[0].dw0) * 16 + g * 4;
        yo[vt] = p.y_off0 + (int64_t)n * p.y_n_stride + (int64_t)oh * p.y_h_stride + (int64_t)ow * 16 + g * 4;
        ro[vt] = p.r_off0 + (int64_t)n * p.r_n_stride + (int64_t)oh * p.r_h_stride + (int64_t)ow * 16 + g * 4;
    }
    // weights [cb][cout_pad][16]: lane (cout j, g)
    const float* wl = p.w + ((int64_t)(ct0 * 16 + j)) * 16 + g * 4;
    const int64_t w_cb = (int64_t)p.cout_pad * 16;

    f32x4 acc[VT][CT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[vt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 a[CT], b[VT], an[CT], bn[VT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) a[ct] = *(const f32x4*)(wl + ct * 256);
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) b[vt] = *(const f32x4*)(p.x + xo[vt]);

    for (int cb = 0; cb < p.cb_in; ++cb) {
        const int cn = cb + 1 < p.cb_in ? cb + 1 : cb;            // next block (last iteration: harmless reload)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) an[ct] = *(const f32x4*)(wl + (int64_t)cn * w_cb + ct * 256);
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) bn[vt] = *(const f32x4*)(p.x + xo[vt] + (int64_t)cn * p.x_cb_stride);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[vt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ct][s], b[vt][s], acc[vt][ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) a[ct] = an[ct];
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) b[vt] = bn[vt];
    }

    // epilogue: lane (voxel j, g) holds couts 4g..4g+3 of each tile
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const f32x4 sc = *(const f32x4*)(p.scale + (ct0 + ct) * 16 + g * 4), sh = *(const f32x4*)(p.shift + (ct0 + ct) * 16 + g * 4);
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) {
            if (!ok[vt]) continue;
            f32x4 v = acc[vt][ct] * sc + sh;
            if (p.res) v += *(const f32x4*)(p.res + ro[vt] + (int64_t)(ct0 + ct) * p.r_cb_stride);
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *(f32x4*)(p.y + yo[vt] + (int64_t)(ct0 + ct) * p.y_cb_stride) = v;
        }
    }
}

template <int VT, int CT>
int launch(const drc_tapconv_params& p, hipStream_t s) {
    const long M = (long)p.N * p.OH * p.OW;
    const long tiles = (M + 15) / 16;
    const long waves = (tiles + VT - 1) / VT;
    dim3 grid((unsigned)((waves + PW_WAVES - 1) / PW_WAVES), (unsigned)(p.cout_pad / 16 / CT), 1);
    hipLaunchKernelGGL((pointwise_kernel<VT, CT>), grid, dim3(64 * PW_WAVES), 0, s, p);
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
