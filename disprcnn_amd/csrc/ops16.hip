// ops16.hip -- the non-convolution pieces of the fp16-storage 2D feature CNN (gfx950 / CDNA4), round 3:
// image -> blocked fp16, the SPP average pools and bilinear up-samplings on blocked fp16 tensors, and the concat cost volume built
// from blocked fp16 feature maps.  Layout as in conv16.hip: half[N][ceil(C/32)][H+2p][W+2p][32], zero halo; arithmetic in fp32.
//
//   reference: feature_extraction.forward (submodule.py:106-139: AvgPool2d branches, F.upsample(..., mode='bilinear') with the
//   align_corners=True of the PyTorch 0.4 default the reference was written for, torch.cat), PSMNet.forward's cost volume
//   (stackhourglass.py:115-128).  BASELINE configs[3] only (the reference itself has no fp16 path, config/defaults.py:22).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int kThreads = 256;

inline unsigned grid_for(long total, long cap = 256 * 16) {
    long b = (total + kThreads - 1) / kThreads;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// fp32 NCHW -> blocked fp16 (interior only; channels past C inside the last 32-channel block are written as zeros).
// One thread per (voxel, 8-channel chunk): the reads of one channel are coalesced along w.
__global__ __launch_bounds__(kThreads) void dense_to_blocked16_kernel(const float* __restrict__ x, _Float16* __restrict__ y, int N, int C, int H, int W,
                                                                      int ph, int pw) {
    const int CB = (C + 31) / 32;
    const long total = (long)N * CB * H * W * 4;
    const long Hp = H + 2 * ph, Wp = W + 2 * pw;
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
        long t = idx;
        const int xw = (int)(t % W); t /= W;
        const int ch = (int)(t & 3); t >>= 2;
        const int yh = (int)(t % H); t /= H;
        const int cb = (int)(t % CB);
        const int n = (int)(t / CB);
        f16x8 v;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = cb * 32 + ch * 8 + k;
            v[k] = (_Float16)(c < C ? x[(((long)n * C + c) * H + yh) * W + xw] : 0.f);
        }
        *(f16x8*)(y + ((((long)n * CB + cb) * Hp + (yh + ph)) * Wp + (xw + pw)) * 32 + ch * 8) = v;
    }
}

// AvgPool2d(k, k) of a channel-block slice: one wave per pooled output voxel, 16 window positions x 4 channel chunks per step.
__global__ __launch_bounds__(64) void avgpool2d_blocked16_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y, int N, int CB, int H, int W,
                                                                 int px, int k, int OH, int OW, int py, int x_cb_total, int x_cb_off) {
    long t = blockIdx.x;
    const int ow = (int)(t % OW); t /= OW;
    const int oh = (int)(t % OH); t /= OH;
    const int cb = (int)(t % CB);
    const int n = (int)(t / CB);
    const int lane = threadIdx.x, q = lane & 3, p0 = lane >> 2;
    const long iW = W + 2 * px, iH = H + 2 * px;
    const _Float16* xb = x + (((long)n * x_cb_total + x_cb_off + cb) * iH + (oh * k + px)) * iW * 32 + (long)(ow * k + px) * 32 + q * 8;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p = p0; p < k * k; p += 16) {
        const int r = p / k, c = p - r * k;
        const f16x8 v = *(const f16x8*)(xb + ((long)r * iW + c) * 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] += (float)v[i];
    }
#pragma unroll
    for (int m = 4; m < 64; m <<= 1)
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] += __shfl_xor(s[i], m);
    if (p0 == 0) {
        const float inv = 1.0f / (float)(k * k);
        const long oW = OW + 2 * py, oH = OH + 2 * py;
        f16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (_Float16)(s[i] * inv);
        *(f16x8*)(y + ((((long)n * CB + cb) * oH + (oh + py)) * oW + (ow + py)) * 32 + q * 8) = o;
    }
}

// bilinear up-sampling (align_corners = True, the association of ATen's upsample_bilinear2d) into a channel-block slice
__global__ __launch_bounds__(kThreads) void bilinear_up_blocked16_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y, int N, int CB, int IH,
                                                                         int IW, int px, int OH, int OW, int py, int y_cb_total, int y_cb_off) {
    const long total = (long)N * CB * OH * OW * 4;
    const long iW = IW + 2 * px, iH = IH + 2 * px, oW = OW + 2 * py, oH = OH + 2 * py;
    const float sy = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f;
    const float sx = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
        long t = idx;
        const int q = (int)(t & 3); t >>= 2;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH); t /= OH;
        const int cb = (int)(t % CB);
        const int n = (int)(t / CB);
        const float fy = sy * oy, fx = sx * ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < IH - 1), x1 = x0 + (x0 < IW - 1);
        const float ty = fy - y0, tx = fx - x0;
        const _Float16* xb = x + (((long)n * CB + cb) * iH) * iW * 32 + q * 8;
        const f16x8 v00 = *(const f16x8*)(xb + ((long)(y0 + px) * iW + (x0 + px)) * 32);
        const f16x8 v01 = *(const f16x8*)(xb + ((long)(y0 + px) * iW + (x1 + px)) * 32);
        const f16x8 v10 = *(const f16x8*)(xb + ((long)(y1 + px) * iW + (x0 + px)) * 32);
        const f16x8 v11 = *(const f16x8*)(xb + ((long)(y1 + px) * iW + (x1 + px)) * 32);
        f16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            o[i] = (_Float16)((1.f - ty) * ((1.f - tx) * (float)v00[i] + tx * (float)v01[i]) + ty * ((1.f - tx) * (float)v10[i] + tx * (float)v11[i]));
        *(f16x8*)(y + ((((long)n * y_cb_total + y_cb_off + cb) * oH + (oy + py)) * oW + (ox + py)) * 32 + q * 8) = o;
    }
}

// blocked fp16 feature maps [units][1 block of 32][H+2p][W+2p][32] (left units first, right units `r_first` units later) -> blocked fp16
// cost volume [N][2 blocks][D+2][H+2][W+2][32]: block 0 = left, block 1 = right shifted by the slice's disparity, zero where the
// shifted pixel falls outside (stackhourglass.py:115-128).  One thread per (voxel, side, 8-channel chunk): a 16-byte copy.
__global__ __launch_bounds__(kThreads) void cost_volume16_from16_kernel(const _Float16* __restrict__ f, _Float16* __restrict__ out, int N, int r_first,
                                                                        int D, int H, int W, int lo4, int fp, long n_stride, long cb_stride,
                                                                        long d_stride, long h_stride, long off0) {
    const long total = (long)N * D * H * W * 8;
    const long fW = W + 2 * fp, fH = H + 2 * fp;
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
        long t = idx;
        const int ch = (int)(t & 3); t >>= 2;
        const int side = (int)(t & 1); t >>= 1;
        const int xw = (int)(t % W); t /= W;
        const int yh = (int)(t % H); t /= H;
        const int d = (int)(t % D);
        const int n = (int)(t / D);
        const int xs = xw - (lo4 + d);
        const bool ok = xs >= 0 && xs < W;
        f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ok) {
            const int unit = side == 0 ? n : n + r_first, xx = side == 0 ? xw : xs;
            v = *(const f16x8*)(f + (((long)unit * fH + (yh + fp)) * fW + (xx + fp)) * 32 + ch * 8);
        }
        *(f16x8*)(out + off0 + (long)n * n_stride + (long)side * cb_stride + (long)d * d_stride + (long)yh * h_stride + (long)xw * 32 + ch * 8) = v;
    }
}

}  // namespace

extern "C" {

int drc_dense_to_blocked16(const float* x, void* y16, int N, int C, int H, int W, int ph, int pw, void* stream) {
    if (N < 0 || C <= 0 || H <= 0 || W <= 0 || ph < 0 || pw < 0) return -2;
    if (N == 0) return 0;
    if (!x || !y16) return -1;
    const long total = (long)N * ((C + 31) / 32) * H * W * 4;
    hipLaunchKernelGGL(dense_to_blocked16_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, x, (_Float16*)y16, N, C, H, W, ph, pw);
    return (int)hipGetLastError();
}

int drc_avgpool2d_blocked16_slice(const void* x16, void* y16, int N, int CB, int H, int W, int px, int k, int OH, int OW, int py, int x_cb_total,
                                  int x_cb_off, void* stream) {
    if (N < 0 || CB <= 0 || H <= 0 || W <= 0 || k <= 0 || OH <= 0 || OW <= 0 || OH * k > H || OW * k > W) return -2;
    if (x_cb_off < 0 || x_cb_off + CB > x_cb_total) return -2;
    const long blocks = (long)N * CB * OH * OW;
    if (blocks == 0) return 0;
    if (!x16 || !y16) return -1;
    hipLaunchKernelGGL(avgpool2d_blocked16_kernel, dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, (const _Float16*)x16, (_Float16*)y16, N, CB,
                       H, W, px, k, OH, OW, py, x_cb_total, x_cb_off);
    return (int)hipGetLastError();
}

int drc_bilinear_up_blocked16(const void* x16, void* y16, int N, int CB, int IH, int IW, int px, int OH, int OW, int py, int y_cb_total, int y_cb_off,
                              void* stream) {
    if (N < 0 || CB <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || y_cb_off < 0 || y_cb_off + CB > y_cb_total) return -2;
    const long total = (long)N * CB * OH * OW * 4;
    if (total == 0) return 0;
    if (!x16 || !y16) return -1;
    hipLaunchKernelGGL(bilinear_up_blocked16_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, (const _Float16*)x16, (_Float16*)y16, N,
                       CB, IH, IW, px, OH, OW, py, y_cb_total, y_cb_off);
    return (int)hipGetLastError();
}

int drc_cost_volume16_from16(const void* feat16, void* cost16, int N, int right_first_unit, int C, int Dp, int Hp, int Wp, int mindisp4, int maxdisp4,
                             int feat_pad, void* stream) {
    if (N < 0 || C <= 0 || C > 32 || Dp <= 0 || Hp <= 0 || Wp <= 0 || maxdisp4 - mindisp4 != Dp || feat_pad < 0 || right_first_unit < 0) return -2;
    if (N == 0) return 0;
    if (!feat16 || !cost16) return -1;
    const long h_stride = (long)(Wp + 2) * 32, d_stride = (long)(Hp + 2) * h_stride, cb_stride = (long)(Dp + 2) * d_stride, n_stride = 2 * cb_stride;
    const long off0 = d_stride + h_stride + 32;
    const long total = (long)N * Dp * Hp * Wp * 8;
    hipLaunchKernelGGL(cost_volume16_from16_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, (const _Float16*)feat16,
                       (_Float16*)cost16, N, right_first_unit, Dp, Hp, Wp, mindisp4, feat_pad, n_stride, cb_stride, d_stride, h_stride, off0);
    return (int)hipGetLastError();
}

}  // extern "C"
