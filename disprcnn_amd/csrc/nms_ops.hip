// nms_ops.hip -- greedy non-maximum suppression of score-sorted boxes (gfx950): SURVEY f4, replaces disprcnn._C.nms on the GPU
//   reference: csrc/cuda/nms.cu:23-131 (IoU with the legacy +1 pixel convention, suppression when IoU > threshold),
//              csrc/cpu/nms_cpu.cpp:5-75 (same greedy order; compares with >=).
// The reference computes a 64x64-blocked suppression bitmask on the device, copies it to the HOST and walks it there.  Here the
// bitmask kernel uses one 64-lane wavefront per (row block, column block) -- a mask word is exactly one lane's ballot-free
// 64-bit accumulator -- and the greedy walk stays on the device: one wavefront keeps the running "removed" words in its lanes
// (lane j owns column block j) and ORs a surviving row's words in one step.  No host round trip, no allocation here (the
// caller passes the workspace).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

namespace {

__device__ __forceinline__ float iou_plus1(const float* a, const float* b) {
    const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    const float width = fmaxf(right - left + 1.f, 0.f), height = fmaxf(bottom - top + 1.f, 0.f);
    const float inter = width * height;
    const float sa = (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f);
    const float sb = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
    return inter / (sa + sb - inter);
}

// grid (col_blocks, row_blocks), 64 threads: lane r of row block R tests its box against the 64 boxes of column block C
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, int n, float thresh, int strict, uint64_t* __restrict__ mask) {
    const int rb = blockIdx.y, cb = blockIdx.x;
    const int col_blocks = gridDim.x;
    __shared__ float cbox[64 * 4];
    const int cn = min(n - cb * 64, 64), rn = min(n - rb * 64, 64);
    if ((int)threadIdx.x < cn) {
        const float4 v = *(const float4*)(boxes + (int64_t)(cb * 64 + threadIdx.x) * 4);
        cbox[threadIdx.x * 4 + 0] = v.x; cbox[threadIdx.x * 4 + 1] = v.y; cbox[threadIdx.x * 4 + 2] = v.z; cbox[threadIdx.x * 4 + 3] = v.w;
    }
    __syncthreads();
    if ((int)threadIdx.x >= rn) return;
    const int row = rb * 64 + threadIdx.x;
    uint64_t t = 0;
    if (cb >= rb) {                                        // only later boxes can be suppressed by this one
        const float4 v = *(const float4*)(boxes + (int64_t)row * 4);
        const float me[4] = {v.x, v.y, v.z, v.w};
        const int start = cb == rb ? (int)threadIdx.x + 1 : 0;
        for (int i = start; i < cn; ++i) {
            const float o = iou_plus1(me, cbox + i * 4);
            if (strict ? o > thresh : o >= thresh) t |= 1ULL << i;
        }
    }
    mask[(int64_t)row * col_blocks + cb] = t;
}

// one wavefront: lane j holds the removed-bits of column blocks j, j+64, ...
__global__ __launch_bounds__(64) void nms_walk_kernel(const uint64_t* __restrict__ mask, int n, int col_blocks, uint8_t* __restrict__ keep) {
    constexpr int kMaxWords = 8;                           // up to 64*64*8 = 32768 boxes
    uint64_t remv[kMaxWords];
#pragma unroll
    for (int w = 0; w < kMaxWords; ++w) remv[w] = 0;
    const int lane = threadIdx.x;
    for (int i = 0; i < n; ++i) {
        const int nb = i >> 6, ib = i & 63;
        // the word of block nb lives in lane nb & 63, register nb >> 6
        uint64_t word = 0;
#pragma unroll
        for (int w = 0; w < kMaxWords; ++w)
            if (w == (nb >> 6)) word = remv[w];
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)word, nb & 63), hi = __builtin_amdgcn_readlane((unsigned)(word >> 32), nb & 63);
        const uint64_t cur = ((uint64_t)hi << 32) | lo;
        const bool kept = !((cur >> ib) & 1ULL);           // wave-uniform
        if (lane == 0) keep[i] = kept ? 1 : 0;
        if (kept) {
            const uint64_t* p = mask + (int64_t)i * col_blocks;
#pragma unroll
            for (int w = 0; w < kMaxWords; ++w) {
                const int j = w * 64 + lane;
                if (j >= nb && j < col_blocks) remv[w] |= p[j];
            }
        }
    }
}

}  // namespace

extern "C" int drc_nms_sorted_fwd(const float* boxes_sorted, int n, float thresh, int strict, uint64_t* mask_ws, uint8_t* keep, void* stream) {
    if (n < 0) return -2;
    if (n == 0) return 0;
    if (!boxes_sorted || !mask_ws || !keep) return -1;
    const int col_blocks = (n + 63) / 64;
    if (col_blocks > 64 * 8) return -5;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(col_blocks, col_blocks), dim3(64), 0, s, boxes_sorted, n, thresh, strict, mask_ws);
    hipLaunchKernelGGL(nms_walk_kernel, dim3(1), dim3(64), 0, s, (const uint64_t*)mask_ws, n, col_blocks, keep);
    return (int)hipGetLastError();
}
