// train_ops.hip -- loss kernels of the disparity stage (gfx950).
//   PSMLoss / EndPointErrorLoss: masked smooth-L1 (train, 3 heads, weights 0.5/0.7/1.0) and masked mean |err| (eval).
//   Reference: utils/loss_utils.py:9-32 == utils/stereo_utils.py:185-208.
// One pass over the heads: per-block shuffle/LDS reduction, one atomicAdd per block and quantity.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// sums[0..2] = sum_k mask*smooth_l1(pred_k - tgt), sums[3] = sum mask, sums[4] = sum mask*|pred_0 - tgt|
__global__ __launch_bounds__(kThreads) void psm_loss_sums_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                                                 const float* __restrict__ p2, const float* __restrict__ tgt,
                                                                 const uint8_t* __restrict__ mask, long n, float* __restrict__ sums) {
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
        const float m = mask[i] ? 1.f : 0.f;
        const float t = tgt[i];
        const float* ps[3] = {p0, p1, p2};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (ps[k]) {
                const float d = fabsf(ps[k][i] - t);
                s[k] += m * (d < 1.f ? 0.5f * d * d : d - 0.5f);
                if (k == 0) s[4] += m * d;
            }
        }
        s[3] += m;
    }
    __shared__ float red[5][kThreads / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float v = wave_sum(s[k]);
        if (lane == 0) red[k][w] = v;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        float v = 0.f;
        for (int i = 0; i < kThreads / 64; ++i) v += red[threadIdx.x][i];
        atomicAdd(sums + threadIdx.x, v);
    }
}

// d loss / d pred_k = gscale * w_k * mask * clamp(pred_k - tgt, -1, 1) / msum   (msum==0 -> no division, as the reference)
__global__ __launch_bounds__(kThreads) void psm_loss_grad_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                                 const uint8_t* __restrict__ mask, long n, const float* __restrict__ sums,
                                                                 float weight, const float* __restrict__ gscale, float* __restrict__ gpred) {
    const float msum = sums[3];
    const float k = weight * gscale[0] / (msum != 0.f ? msum : 1.f);
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
        const float d = pred[i] - tgt[i];
        gpred[i] = mask[i] ? k * fminf(fmaxf(d, -1.f), 1.f) : 0.f;
    }
}

inline unsigned grid_for(long work) {
    long b = (work + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return (unsigned)b;
}

}  // namespace

extern "C" {

int drc_psm_loss_sums(const float* pred1, const float* pred2, const float* pred3, const float* target, const uint8_t* mask,
                      int64_t numel, float* sums5, void* stream) {
    if (numel < 0) return -2;
    if (!sums5) return -1;
    if (numel == 0) return 0;           // sums5 must be zeroed by the caller; an empty batch leaves it at zero
    if (!pred1 || !target || !mask) return -1;
    hipLaunchKernelGGL(psm_loss_sums_kernel, dim3(grid_for(numel)), dim3(kThreads), 0, (hipStream_t)stream, pred1, pred2, pred3, target,
                       mask, (long)numel, sums5);
    return (int)hipGetLastError();
}

int drc_psm_loss_grad(const float* pred, const float* target, const uint8_t* mask, int64_t numel, const float* sums5, float weight,
                      const float* grad_scale, float* grad_pred, void* stream) {
    if (numel < 0) return -2;
    if (numel == 0) return 0;
    if (!pred || !target || !mask || !sums5 || !grad_scale || !grad_pred) return -1;
    hipLaunchKernelGGL(psm_loss_grad_kernel, dim3(grid_for(numel)), dim3(kThreads), 0, (hipStream_t)stream, pred, target, mask,
                       (long)numel, sums5, weight, grad_scale, grad_pred);
    return (int)hipGetLastError();
}

}  // extern "C"
