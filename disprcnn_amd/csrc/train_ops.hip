// train_ops.hip -- loss kernels of the disparity stage (gfx950).
//   PSMLoss / EndPointErrorLoss: masked smooth-L1 (train, 3 heads, weights 0.5/0.7/1.0) and masked mean |err| (eval).
//   Reference: utils/loss_utils.py:9-32 == utils/stereo_utils.py:185-208.
// One pass over the heads: per-block shuffle/LDS reduction; every block stores its five partials and a one-block finishing
// launch adds them in block order (no atomicAdd: the loss and everything downstream of it are bit-reproducible run to run).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"
#include "det_reduce.h"
#include "blocked_walk.h"

typedef float f32x4_t __attribute__((ext_vector_type(4)));

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
                                                                 const uint8_t* __restrict__ mask, long n, float* __restrict__ part) {
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
        part[(long)blockIdx.x * 8 + threadIdx.x] = v;
    }
}

// sums[k] = sum over blocks (block order) of part[block][k]: 5 groups of 32 threads take blocks i = lane, lane + 32, ...
// in order, then a fixed xor-tree over the 32 lanes of the group
__global__ __launch_bounds__(kThreads) void psm_loss_finish_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ sums) {
    const int k = threadIdx.x >> 5, l = threadIdx.x & 31;
    float v = 0.f;
    if (k < 5)
        for (int i = l; i < nblocks; i += 32) v += part[(long)i * 8 + k];
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if (k < 5 && l == 0) sums[k] = v;
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
                      int64_t numel, float* sums5, float* scratch, void* stream) {
    if (numel < 0) return -2;
    if (!sums5) return -1;
    if (numel == 0) return (int)hipMemsetAsync(sums5, 0, 5 * sizeof(float), (hipStream_t)stream);   // an empty batch: all sums are zero
    if (!pred1 || !target || !mask || !scratch) return -1;
    const unsigned blocks = grid_for(numel);            // <= 1024 = DRC_LOSS_SCRATCH_FLOATS / 8
    hipLaunchKernelGGL(psm_loss_sums_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, pred1, pred2, pred3, target,
                       mask, (long)numel, scratch);
    hipLaunchKernelGGL(psm_loss_finish_kernel, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, scratch, (int)blocks, sums5);
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

// ------------------------------------------------------------------------------------------------------------------
// Training-mode BatchNorm on blocked tensors (reference: nn.BatchNorm3d/2d inside convbn_3d / convbn, submodule.py:13-22,
// per-GPU batch statistics, eps 1e-5, momentum 0.1 handled by the caller).
//   bn_stats   : per-channel mean and sum of squared deviations over the interior voxels in one pass (Chan-combined partials)
//   bn_apply   : y = act( (x - mean) * invstd * gamma + beta (+ res) ), interior only (halos stay zero)
//   bn_bwd_reduce / bn_bwd_apply : the standard BN backward with the ReLU mask and residual fan-out fused
namespace {

using drc_blk::BlkGeom;
using drc_blk::blk_off;

// grid: (chunks, CB); each block walks a contiguous run of rows of one channel block; thread = float4 quad of a voxel.
// One pass: every thread sums (x - x0) and (x - x0)^2 around its own first value x0, turns that into (count, mean, M2) and the
// statistics are merged pairwise (Chan) lane -> wave -> block -> launch, all in a fixed order.
__global__ __launch_bounds__(kThreads) void bn_stats_kernel(const float* __restrict__ x, BlkGeom g, float* __restrict__ out /* [2][CB*16]: mean, M2 */,
                                                            float* scratch) {
    const int cb = blockIdx.y;
    float n = 0.f;
    f32x4_t x0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    drc_blk::walk_rows<kThreads>(g, [&](int nn, int dd, int yy, int xx, int qq) {
        const f32x4_t v = *(const f32x4_t*)(x + blk_off(g, nn, cb, dd, yy, xx) + qq * 4);
        if (n == 0.f) x0 = v;
        const f32x4_t d = v - x0;
        s1 += d; s2 += d * d; n += 1.f;
    });
    const float inv = n > 0.f ? 1.f / n : 0.f;
    float mean[4] = {x0.x + s1.x * inv, x0.y + s1.y * inv, x0.z + s1.z * inv, x0.w + s1.w * inv};
    float m2[4] = {s2.x - s1.x * s1.x * inv, s2.y - s1.y * s1.y * inv, s2.z - s1.z * s1.z * inv, s2.w - s1.w * s1.w * inv};
    // merge over the 16 voxel-lanes that share this quad inside the wave
#pragma unroll
    for (int m = 4; m < 64; m <<= 1) {
        const float nb = __shfl_xor(n, m);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const drc_det::Stat r = drc_det::merge(drc_det::Stat{n, mean[k], m2[k]}, drc_det::Stat{nb, __shfl_xor(mean[k], m), __shfl_xor(m2[k], m)});
            mean[k] = r.mean; m2[k] = r.m2;
        }
        n += nb;
    }
    __shared__ float red[kThreads / 64][4][9];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane < 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[w][lane][k] = mean[k]; red[w][lane][4 + k] = m2[k]; }
        red[w][lane][8] = n;
    }
    __syncthreads();
    drc_det::Stat st = {0.f, 0.f, 0.f};
    if (threadIdx.x < 16) {
        const int qq = threadIdx.x >> 2, k = threadIdx.x & 3;
        for (int i = 0; i < kThreads / 64; ++i) st = drc_det::merge(st, drc_det::Stat{red[i][qq][8], red[i][qq][k], red[i][qq][4 + k]});
    }
    const BlkGeom gg = g;
    drc_det::finish_stat(st, cb, g.CB, out, out + g.CB * 16, scratch,
                         [&](unsigned i) { return drc_blk::rows_of_block(gg, kThreads, i, gridDim.x); });
}

// grid: (chunks, CB)
__global__ __launch_bounds__(kThreads) void bn_apply_kernel(const float* __restrict__ x, BlkGeom gx, float* __restrict__ y, BlkGeom gy,
                                                            const float* __restrict__ res, BlkGeom gr, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int relu) {
    const int cb = blockIdx.y;
    const int c = cb * 16 + (threadIdx.x & 3) * 4;
    const f32x4_t m = *(const f32x4_t*)(mean + c), is = *(const f32x4_t*)(invstd + c);
    const f32x4_t ga = *(const f32x4_t*)(gamma + c), be = *(const f32x4_t*)(beta + c);
    drc_blk::walk_rows<kThreads>(gx, [&](int n, int dd, int yy, int xx, int q) {
        f32x4_t v = (*(const f32x4_t*)(x + blk_off(gx, n, cb, dd, yy, xx) + q * 4) - m) * is * ga + be;
        if (res) v += *(const f32x4_t*)(res + blk_off(gr, n, cb, dd, yy, xx) + q * 4);
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *(f32x4_t*)(y + blk_off(gy, n, cb, dd, yy, xx) + q * 4) = v;
    });
}

// one launch instead of ~10 tiny torch kernels per site: invstd = rsqrt(M2/M + eps); running statistics updated like nn.BatchNorm
// (momentum, UNBIASED batch variance); num_batches_tracked += 1
__global__ void bn_finalize_kernel(const float* __restrict__ stats, int C16, int C, float M, float eps, float momentum,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, long long* num_batches_tracked,
                                   float* __restrict__ invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
    if (c >= C16) return;
    const float mean = stats[c], var = stats[C16 + c] / M;
    invstd[c] = rsqrtf(var + eps);
    if (c < C && running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (M / fmaxf(M - 1.f, 1.f));
    }
}

inline bool geom_ok(const int* g) { return g && g[0] >= 0 && g[1] > 0 && g[2] > 0 && g[3] > 0 && g[4] > 0 && g[5] >= 0 && g[6] >= 0 && g[7] >= 0 && g[9] >= 0 && g[9] + g[1] <= g[8]; }
inline BlkGeom to_geom(const int* g) { return BlkGeom{g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9]}; }

}  // namespace

extern "C" {

int drc_bn_stats_blocked(const float* x, const int* geom8, float* stats, float* scratch, void* stream) {
    if (!geom_ok(geom8)) return -2;
    if (geom8[0] == 0) return 0;
    if (!x || !stats || !scratch) return -1;
    const BlkGeom g = to_geom(geom8);
    const long nvox = (long)g.N * g.D * g.H * g.W;
    long chunks = (nvox + (kThreads / 4) * 8 - 1) / ((kThreads / 4) * 8);
    if (chunks < 1) chunks = 1;
    if (chunks > DRC_BN_MAX_CHUNKS) chunks = DRC_BN_MAX_CHUNKS;
    if (chunks > (long)g.N * g.D * g.H) chunks = (long)g.N * g.D * g.H;
    hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)chunks, (unsigned)g.CB), dim3(kThreads), 0, (hipStream_t)stream, x, g, stats, scratch);
    return (int)hipGetLastError();
}

int drc_bn_finalize(const float* stats, int C16, int C, long long count, float eps, float momentum, float* running_mean, float* running_var,
                    long long* num_batches_tracked, float* invstd, void* stream) {
    if (C16 <= 0 || C < 0 || C > C16 || count <= 0) return -2;
    if (!stats || !invstd || ((running_mean == nullptr) != (running_var == nullptr))) return -1;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C16 + 127) / 128), dim3(128), 0, (hipStream_t)stream, stats, C16, C, (float)count, eps, momentum,
                       running_mean, running_var, num_batches_tracked, invstd);
    return (int)hipGetLastError();
}

int drc_bn_apply_blocked(const float* x, const int* geom_x, float* y, const int* geom_y, const float* res, const int* geom_r,
                         const float* mean, const float* invstd, const float* gamma, const float* beta, int relu, void* stream) {
    if (!geom_x || !geom_y || !geom_ok(geom_x) || !geom_ok(geom_y)) return -2;
    if (res && (!geom_r || !geom_ok(geom_r))) return -2;
    for (int i = 0; i < 5; ++i)
        if (geom_x[i] != geom_y[i] || (res && geom_x[i] != geom_r[i])) return -2;     // same logical shape
    if (geom_x[0] == 0) return 0;
    if (!x || !y || !mean || !invstd || !gamma || !beta) return -1;
    const BlkGeom gx = to_geom(geom_x), gy = to_geom(geom_y), gr = res ? to_geom(geom_r) : gx;
    const long rows = (long)gx.N * gx.D * gx.H;
    long chunks = ((long)rows * gx.W * 4 + kThreads * 4 - 1) / (kThreads * 4);
    if (chunks > 2048) chunks = 2048;
    if (chunks > rows) chunks = rows;
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)chunks, (unsigned)gx.CB), dim3(kThreads), 0, (hipStream_t)stream, x, gx, y, gy, res, gr,
                       mean, invstd, gamma, beta, relu);
    return (int)hipGetLastError();
}

}  // extern "C"
