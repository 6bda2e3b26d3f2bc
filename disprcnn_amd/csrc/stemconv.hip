// stemconv.hip -- the feature CNN's first layer, Conv2d(3 -> cout, 3x3, stride 2, pad 1) + BN + ReLU, straight from the dense image
// (gfx950 / CDNA4), round 4.
//
//   reference: feature_extraction.firstconv[0], submodule.py:65-66 (convbn(3, 32, 3, 2, 1, 1) + ReLU), eval mode (BatchNorm folded)
//
// Why: on the generic path the image is first converted to the channel-blocked layout -- 3 real channels in a block of 16, 64 bytes per
// pixel -- and the stride-2 direct kernel then contracts a 16-channel block per tap: the conversion writes and the convolution reads
// 5.3x the image (411 MB for the 128 crops of the stress shape), 13 of every 16 MFMA k-lanes multiply zeros, and the two launches take
// 126 + 225 us against ~80 us for the bytes that matter (77 MB of image, 205 MB of output).  Here the contraction runs over
// k = channel * 9 + tap = 0..26 (padded to 28: seven k-steps of the 16x16x4 fp32 MFMA instead of 36): lane (pixel j, g) of a wave gathers
// x[n, k / 9, 2 oy + kh - 1, 2 ox + kw - 1] for k = 4 m + g -- scattered 4-byte reads of an image that stays in L1 / L2 -- against
// weights packed [m][cout][4] (host: engine.pack_weight_stem), one wave per 16 consecutive output pixels of a row and all couts.
// Output: the channel-blocked fp32 layout of every other layer, float4 per (pixel, 4 couts), halo untouched (zero from allocation).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kWaves = 4;

template <int CT>
__global__ __launch_bounds__(64 * kWaves) void stemconv_kernel(const float* __restrict__ x, int N, int H, int W, const float* __restrict__ wpk,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              float* __restrict__ y, long y_n_stride, long y_cb_stride, long y_h_stride, long y_off0,
                                                              int OH, int OW, int relu) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    const int n_ct = (OW + 15) >> 4;
    const long tiles = (long)N * OH * n_ct;
    // this lane's seven (channel, kh, kw): k = 4 m + g
    int koff[7], kdy[7], kdx[7];
    bool kreal[7];
#pragma unroll
    for (int m = 0; m < 7; ++m) {
        const int k = 4 * m + g;
        kreal[m] = k < 27;
        const int kk = kreal[m] ? k : 0;
        const int ch = kk / 9, tap = kk - ch * 9;
        kdy[m] = tap / 3 - 1; kdx[m] = tap % 3 - 1;
        koff[m] = ch * H * W;
    }
    float wv[7][CT];
    f32x4 sc[CT], sh[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            // A operand of step m: W[cout ct*16 + j][k = 4 m + g]; the MFMA takes one scalar per lane, the four accumulator rows 4g..4g+3 of
            // lane (j, g) are couts -- so the weight value each lane feeds is the one of "its" A row j and k member g
            wv[m][ct] = wpk[((long)m * (CT * 16) + ct * 16 + j) * 4 + g];
        }
        sc[ct] = *(const f32x4*)(scale + ct * 16 + g * 4);
        sh[ct] = *(const f32x4*)(shift + ct * 16 + g * 4);
    }
    for (long tile = (long)blockIdx.x * kWaves + wave; tile < tiles; tile += (long)gridDim.x * kWaves) {
        long t = tile;
        const int c0 = (int)(t % n_ct) * 16; t /= n_ct;
        const int oy = (int)(t % OH);
        const int n = (int)(t / OH);
        const int ox = c0 + j;
        const float* xn = x + (long)n * 3 * H * W;
        float bv[7];
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            const int iy = 2 * oy + kdy[m], ix = 2 * ox + kdx[m];
            const bool ok = kreal[m] && ox < OW && iy >= 0 && iy < H && ix >= 0 && ix < W;
            bv[m] = ok ? xn[koff[m] + iy * W + ix] : 0.f;
        }
        f32x4 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 7; ++m)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[m][ct], bv[m], acc[ct], 0, 0, 0);
        if (ox < OW) {
            float* yp = y + y_off0 + (long)n * y_n_stride + (long)oy * y_h_stride + (long)ox * 16 + g * 4;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                f32x4 v = acc[ct] * sc[ct] + sh[ct];
                if (relu) v = __builtin_elementwise_max(v, (f32x4){0.f, 0.f, 0.f, 0.f});
                *(f32x4*)(yp + ct * y_cb_stride) = v;
            }
        }
    }
}

}  // namespace

extern "C" int drc_conv2d_k3s2_stem_fwd(const float* x, int N, int H, int W, const float* w_packed, int cout_pad, const float* scale, const float* shift,
                                        float* y, int64_t y_n_stride, int64_t y_cb_stride, int64_t y_h_stride, int64_t y_off0, int OH, int OW, int relu,
                                        void* stream) {
    if (N < 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || cout_pad <= 0) return -2;
    if (OH != (H + 1) / 2 || OW != (W + 1) / 2) return -2;                      // k3 s2 p1
    if (cout_pad != 16 && cout_pad != 32) return -4;                             // one or two cout tiles per wave (the reference's stem has 32)
    if (N == 0) return 0;
    if (!x || !w_packed || !scale || !shift || !y) return -1;
    if ((int64_t)3 * H * W >= (1LL << 31)) return -5;
    const long tiles = (long)N * OH * ((OW + 15) / 16);
    long blocks = (tiles + kWaves - 1) / kWaves;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipStream_t s = (hipStream_t)stream;
    if (cout_pad == 32)
        hipLaunchKernelGGL(stemconv_kernel<2>, dim3((unsigned)blocks), dim3(64 * kWaves), 0, s, x, N, H, W, w_packed, scale, shift, y, (long)y_n_stride,
                           (long)y_cb_stride, (long)y_h_stride, (long)y_off0, OH, OW, relu);
    else
        hipLaunchKernelGGL(stemconv_kernel<1>, dim3((unsigned)blocks), dim3(64 * kWaves), 0, s, x, N, H, W, w_packed, scale, shift, y, (long)y_n_stride,
                           (long)y_cb_stride, (long)y_h_stride, (long)y_off0, OH, OW, relu);
    return (int)hipGetLastError();
}
