// nms_ops.hip -- greedy non-maximum suppression of score-sorted boxes (gfx950): SURVEY f4, replaces disprcnn._C.nms on the GPU
//   reference: csrc/cuda/nms.cu:23-131 (IoU with the legacy +1 pixel convention, suppression when IoU > threshold),
//              csrc/cpu/nms_cpu.cpp:5-75 (same greedy order; compares with >=).
// The reference computes a 64x64-blocked suppression bitmask on the device, copies it to the HOST and walks it there.  Here the
// bitmask kernel uses one 64-lane wavefront per (row block, column block) -- a mask word is exactly one wavefront ballot -- and the
// greedy walk stays on the device: one wavefront keeps the running "removed" words in its lanes
// (lane j owns column block j) and ORs a surviving row's words in one step.  No host round trip, no allocation here (the
// caller passes the workspace).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

namespace {

// Suppression bitmask, one wavefront per (row block R, column block C) of 64 x 64 boxes.  Lane i OWNS column box 64C+i (registers; no LDS,
// no barrier); the row boxes are walked one at a time, row r's corners and area broadcast from lane r through scalar registers
// (v_readlane), every lane tests it against its own column box, and the wavefront-wide ballot of the test IS the row's 64-bit mask word
// (bit i = row suppresses column box 64C+i).  Lane r keeps word r, so the block ends with one coalesced 512-byte store.
// IoU with the legacy +1 pixel convention, as inter / (area_row + area_col - inter) (reference csrc/cuda/nms.cu:13-21, csrc/cpu/nms_cpu.cpp:26-62:
// the operation order is what index-exact parity with the reference's kept sets pins).
// grid (col_blocks, row_blocks, box sets), 64 threads
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, int n, float thresh, int strict, uint64_t* __restrict__ mask) {
    const int rb = blockIdx.y, cb = blockIdx.x;
    const int col_blocks = gridDim.x;
    boxes += (int64_t)blockIdx.z * n * 4;                  // batched launch: box set blockIdx.z
    mask += (int64_t)blockIdx.z * n * col_blocks;
    const int lane = threadIdx.x;
    const int rn = min(n - rb * 64, 64), cn = min(n - cb * 64, 64);
    uint64_t word = 0;
    if (cb >= rb) {                                        // a box only suppresses LATER boxes (the upper triangle; wave-uniform)
        const float4 z4 = {0.f, 0.f, 0.f, 0.f};
        const float4 cq = lane < cn ? *(const float4*)(boxes + (int64_t)(cb * 64 + lane) * 4) : z4;
        const float4 rq = lane < rn ? *(const float4*)(boxes + (int64_t)(rb * 64 + lane) * 4) : z4;
        const float c_area = (cq.z - cq.x + 1.f) * (cq.w - cq.y + 1.f);
        const float r_area = (rq.z - rq.x + 1.f) * (rq.w - rq.y + 1.f);
        auto bcast = [](float v, int r) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), r)); };
        for (int r = 0; r < rn; ++r) {
            const float x1 = bcast(rq.x, r), y1 = bcast(rq.y, r), x2 = bcast(rq.z, r), y2 = bcast(rq.w, r), sa = bcast(r_area, r);
            const float left = fmaxf(x1, cq.x), right = fminf(x2, cq.z);
            const float top = fmaxf(y1, cq.y), bottom = fminf(y2, cq.w);
            const float width = fmaxf(right - left + 1.f, 0.f), height = fmaxf(bottom - top + 1.f, 0.f);
            const float inter = width * height;
            const float o = inter / (sa + c_area - inter);
            const bool hit = lane < cn && (strict ? o > thresh : o >= thresh);
            uint64_t w = __ballot(hit);
            if (cb == rb) w &= r < 63 ? ~0ULL << (r + 1) : 0ULL;     // the diagonal block: only the boxes after row r
            if (lane == r) word = w;
        }
    }
    if (lane < rn) mask[(int64_t)(rb * 64 + lane) * col_blocks + cb] = word;
}

// one wavefront: lane j holds the removed-bits of column blocks j, j+64, ...  The walk goes 64 rows (one column block) at a time:
// (1) lane r fetches the DIAGONAL word of row 64c+r (one parallel load, issued a chunk ahead); (2) the 64 rows of the chunk are
// resolved against each other in registers -- the only truly sequential part, ~20 cycles per row, no memory access; (3) the kept rows'
// mask rows are ORed into the removed-words of the later blocks with independent loads, four rows in flight.  (Round 2, first form:
// one dependent global load per kept row -- 1.27 ms for the Stereo RPN's 6,000 boxes, 27 % of the 2D stage.)
// (Round 3: templated on the removed-words per lane, so that the usual proposal counts -- the RPN's 6,000 boxes per view = 94 column blocks,
// two words per lane -- keep 16 kept rows' mask rows in flight per round instead of 4: the walk is a chain of L2 round trips.)
template <int kMaxWords>
__global__ __launch_bounds__(64) void nms_walk_kernel(const uint64_t* __restrict__ mask, int n, int col_blocks, uint8_t* __restrict__ keep) {
    constexpr int kRows = kMaxWords <= 2 ? 16 : (kMaxWords <= 4 ? 8 : 4);       // kept rows whose mask rows are loaded together
    mask += (int64_t)blockIdx.x * n * col_blocks;          // batched launch: one wavefront per box set
    keep += (int64_t)blockIdx.x * n;
    uint64_t remv[kMaxWords];
#pragma unroll
    for (int w = 0; w < kMaxWords; ++w) remv[w] = 0;
    const int lane = threadIdx.x;
    auto diag_of = [&](int c) {
        const int row = c * 64 + lane;
        return (c < col_blocks && row < n) ? mask[(int64_t)row * col_blocks + c] : 0ULL;
    };
    uint64_t dnext = diag_of(0);
    for (int c = 0; c < col_blocks; ++c) {
        const uint64_t diag = dnext;
        dnext = diag_of(c + 1);
        const int nr = n - c * 64 < 64 ? n - c * 64 : 64;
        // removed-bits of block c: they live in lane c & 63, register c >> 6
        uint64_t word = 0;
#pragma unroll
        for (int w = 0; w < kMaxWords; ++w)
            if (w == (c >> 6)) word = remv[w];
        // (the builtin returns a signed int: widen through unsigned, or bit 31 of the low half smears over the high half)
        uint64_t cur = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((unsigned)(word >> 32), c & 63) << 32) |
                       (uint64_t)(unsigned)__builtin_amdgcn_readlane((unsigned)word, c & 63);
        uint64_t kept = 0;                                 // wave-uniform
        const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
        for (int r = 0; r < nr; ++r) {                     // branch-free: the cross-lane reads stay in uniform control flow
            const uint64_t dr = ((uint64_t)(unsigned)__builtin_amdgcn_readlane(dhi, r) << 32) | (uint64_t)(unsigned)__builtin_amdgcn_readlane(dlo, r);
            const uint64_t k = ((cur >> r) & 1ULL) ^ 1ULL;
            kept |= k << r;
            cur |= dr & (0ULL - k);
        }
        if (lane < nr) keep[c * 64 + lane] = (uint8_t)((kept >> lane) & 1ULL);
        // OR the kept rows into the later blocks' removed-words
        uint64_t km = kept;
        while (km) {
            int rows[kRows];
#pragma unroll
            for (int q = 0; q < kRows; ++q) {
                rows[q] = km ? __builtin_ctzll(km) : -1;
                km &= km - 1;                              // (0 & anything stays 0)
            }
            uint64_t v[kRows][kMaxWords];
#pragma unroll
            for (int q = 0; q < kRows; ++q) {
                const uint64_t* p = mask + (int64_t)(c * 64 + (rows[q] < 0 ? 0 : rows[q])) * col_blocks;
#pragma unroll
                for (int w = 0; w < kMaxWords; ++w) {
                    const int j = w * 64 + lane;
                    v[q][w] = (rows[q] >= 0 && j > c && j < col_blocks) ? p[j] : 0ULL;
                }
            }
#pragma unroll
            for (int q = 0; q < kRows; ++q)
#pragma unroll
                for (int w = 0; w < kMaxWords; ++w) remv[w] |= v[q][w];
        }
    }
}

// The two views of a stereo list walked TOGETHER (round 3): wave w resolves view w exactly as above; after every 64-row chunk the two
// kept-masks meet in LDS, their AND is the joint keep (double_view_boxlist_nms keeps the intersection, boxlist_ops.py:49-79) and the walk
// stops once `max_keep` pairs survive -- the Stereo RPN keeps POST_NMS_TOP_N of its PRE_NMS_TOP_N score-sorted proposals, so most of the
// chain of L2 round trips is never walked.  `keep` (joint flags) must be zero-filled; rows past the stopping chunk stay 0.
template <int kMaxWords>
__global__ __launch_bounds__(128) void nms_walk_joint_kernel(const uint64_t* __restrict__ mask, int n, int col_blocks, int max_keep,
                                                             uint8_t* __restrict__ keep) {
    constexpr int kRows = kMaxWords <= 2 ? 16 : (kMaxWords <= 4 ? 8 : 4);
    __shared__ uint64_t kept_lds[2][2];
    const int view = threadIdx.x >> 6, lane = threadIdx.x & 63;
    mask += (int64_t)view * n * col_blocks;
    uint64_t remv[kMaxWords];
#pragma unroll
    for (int w = 0; w < kMaxWords; ++w) remv[w] = 0;
    auto diag_of = [&](int c) {
        const int row = c * 64 + lane;
        return (c < col_blocks && row < n) ? mask[(int64_t)row * col_blocks + c] : 0ULL;
    };
    uint64_t dnext = diag_of(0);
    int total = 0;
    for (int c = 0; c < col_blocks; ++c) {
        const uint64_t diag = dnext;
        dnext = diag_of(c + 1);
        const int nr = n - c * 64 < 64 ? n - c * 64 : 64;
        uint64_t word = 0;
#pragma unroll
        for (int w = 0; w < kMaxWords; ++w)
            if (w == (c >> 6)) word = remv[w];
        uint64_t cur = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((unsigned)(word >> 32), c & 63) << 32) |
                       (uint64_t)(unsigned)__builtin_amdgcn_readlane((unsigned)word, c & 63);
        uint64_t kept = 0;
        const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
        for (int r = 0; r < nr; ++r) {
            const uint64_t dr = ((uint64_t)(unsigned)__builtin_amdgcn_readlane(dhi, r) << 32) | (uint64_t)(unsigned)__builtin_amdgcn_readlane(dlo, r);
            const uint64_t k = ((cur >> r) & 1ULL) ^ 1ULL;
            kept |= k << r;
            cur |= dr & (0ULL - k);
        }
        if (lane == 0) kept_lds[c & 1][view] = kept;
        __syncthreads();
        const uint64_t both = kept_lds[c & 1][0] & kept_lds[c & 1][1];
        if (view == 0 && lane < nr) keep[c * 64 + lane] = (uint8_t)((both >> lane) & 1ULL);
        total += __builtin_popcountll(both);
        if (total >= max_keep) break;                      // uniform over the block
        uint64_t km = kept;                                // each view propagates ITS kept rows (the views suppress independently)
        while (km) {
            int rows[kRows];
#pragma unroll
            for (int q = 0; q < kRows; ++q) {
                rows[q] = km ? __builtin_ctzll(km) : -1;
                km &= km - 1;
            }
            uint64_t v[kRows][kMaxWords];
#pragma unroll
            for (int q = 0; q < kRows; ++q) {
                const uint64_t* p = mask + (int64_t)(c * 64 + (rows[q] < 0 ? 0 : rows[q])) * col_blocks;
#pragma unroll
                for (int w = 0; w < kMaxWords; ++w) {
                    const int j = w * 64 + lane;
                    v[q][w] = (rows[q] >= 0 && j > c && j < col_blocks) ? p[j] : 0ULL;
                }
            }
#pragma unroll
            for (int q = 0; q < kRows; ++q)
#pragma unroll
                for (int w = 0; w < kMaxWords; ++w) remv[w] |= v[q][w];
        }
    }
}

// The same walk for more than 32,768 boxes (a KITTI pyramid has ~120 k anchors per view when PRE_NMS_TOP_N_TEST is off; the reference's
// host walk takes any n, csrc/cuda/nms.cu:99-124): the removed-words live in LDS (one per column block) instead of registers.
__global__ __launch_bounds__(64) void nms_walk_big_kernel(const uint64_t* __restrict__ mask, int n, int col_blocks, uint8_t* __restrict__ keep) {
    extern __shared__ uint64_t remv_lds[];                 // [col_blocks]
    mask += (int64_t)blockIdx.x * n * col_blocks;
    keep += (int64_t)blockIdx.x * n;
    const int lane = threadIdx.x;
    for (int j = lane; j < col_blocks; j += 64) remv_lds[j] = 0;
    for (int c = 0; c < col_blocks; ++c) {
        const int row = c * 64 + lane;
        const uint64_t diag = row < n ? mask[(int64_t)row * col_blocks + c] : 0ULL;
        const int nr = n - c * 64 < 64 ? n - c * 64 : 64;
        uint64_t cur = remv_lds[c];                        // one wavefront: its own earlier LDS writes are visible in program order
        uint64_t kept = 0;
        const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
        for (int r = 0; r < nr; ++r) {
            const uint64_t dr = ((uint64_t)(unsigned)__builtin_amdgcn_readlane(dhi, r) << 32) | (uint64_t)(unsigned)__builtin_amdgcn_readlane(dlo, r);
            const uint64_t k = ((cur >> r) & 1ULL) ^ 1ULL;
            kept |= k << r;
            cur |= dr & (0ULL - k);
        }
        if (lane < nr) keep[c * 64 + lane] = (uint8_t)((kept >> lane) & 1ULL);
        for (uint64_t km = kept; km; km &= km - 1) {
            const uint64_t* p = mask + (int64_t)(c * 64 + __builtin_ctzll(km)) * col_blocks;
            for (int j = c + 1 + lane; j < col_blocks; j += 64) remv_lds[j] |= p[j];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): the ORs above have landed before the next block's word is read
        __builtin_amdgcn_wave_barrier();                   // (another lane wrote it: do not let the compiler move the read across)
    }
}

}  // namespace

extern "C" int drc_nms_sorted_batch_fwd(const float* boxes_sorted, int sets, int n, float thresh, int strict, uint64_t* mask_ws, uint8_t* keep,
                                        void* stream) {
    if (n < 0 || sets < 0 || sets > 65535) return -2;
    if (n == 0 || sets == 0) return 0;
    if (!boxes_sorted || !mask_ws || !keep) return -1;
    const int col_blocks = (n + 63) / 64;
    if (col_blocks > 8192) return -5;                      // 524,288 boxes: the big walk's removed-words fill 64 KB of LDS (and the mask 34 GB)
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(col_blocks, col_blocks, sets), dim3(64), 0, s, boxes_sorted, n, thresh, strict, mask_ws);
    if (col_blocks <= 64 * 2)
        hipLaunchKernelGGL(nms_walk_kernel<2>, dim3(sets), dim3(64), 0, s, (const uint64_t*)mask_ws, n, col_blocks, keep);
    else if (col_blocks <= 64 * 4)
        hipLaunchKernelGGL(nms_walk_kernel<4>, dim3(sets), dim3(64), 0, s, (const uint64_t*)mask_ws, n, col_blocks, keep);
    else if (col_blocks <= 64 * 8)
        hipLaunchKernelGGL(nms_walk_kernel<8>, dim3(sets), dim3(64), 0, s, (const uint64_t*)mask_ws, n, col_blocks, keep);
    else
        hipLaunchKernelGGL(nms_walk_big_kernel, dim3(sets), dim3(64), (size_t)col_blocks * 8, s, (const uint64_t*)mask_ws, n, col_blocks, keep);
    return (int)hipGetLastError();
}

extern "C" int drc_nms_sorted_pair_joint_fwd(const float* boxes_sorted, int n, float thresh, int strict, int max_keep, uint64_t* mask_ws,
                                             uint8_t* keep_joint, void* stream) {
    if (n < 0) return -2;
    if (n == 0) return 0;
    if (!boxes_sorted || !mask_ws || !keep_joint) return -1;
    const int col_blocks = (n + 63) / 64;
    if (col_blocks > 64 * 8) return -5;                    // beyond 32,768 boxes: drc_nms_sorted_batch_fwd and an AND of its flags
    if (max_keep <= 0 || max_keep > n) max_keep = n;
    hipStream_t s = (hipStream_t)stream;
    if (const hipError_t e = hipMemsetAsync(keep_joint, 0, (size_t)n, s)) return (int)e;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(col_blocks, col_blocks, 2), dim3(64), 0, s, boxes_sorted, n, thresh, strict, mask_ws);
    if (col_blocks <= 64 * 2)
        hipLaunchKernelGGL(nms_walk_joint_kernel<2>, dim3(1), dim3(128), 0, s, (const uint64_t*)mask_ws, n, col_blocks, max_keep, keep_joint);
    else if (col_blocks <= 64 * 4)
        hipLaunchKernelGGL(nms_walk_joint_kernel<4>, dim3(1), dim3(128), 0, s, (const uint64_t*)mask_ws, n, col_blocks, max_keep, keep_joint);
    else
        hipLaunchKernelGGL(nms_walk_joint_kernel<8>, dim3(1), dim3(128), 0, s, (const uint64_t*)mask_ws, n, col_blocks, max_keep, keep_joint);
    return (int)hipGetLastError();
}

extern "C" int drc_nms_sorted_fwd(const float* boxes_sorted, int n, float thresh, int strict, uint64_t* mask_ws, uint8_t* keep, void* stream) {
    return drc_nms_sorted_batch_fwd(boxes_sorted, 1, n, thresh, strict, mask_ws, keep, stream);
}
