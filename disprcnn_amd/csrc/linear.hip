// linear.hip -- fully connected layers of the 2D stage's heads as a hand-written fp32-MFMA GEMM (gfx950 / CDNA4, round 3).
//
//   y[M][N] = act( x[M][K] . w[N][K]^T + bias[N] )
//
//   reference: the stereo box head's feature extractor -- a 7x7 / stride-7 convolution on 7x7 ROI features = a 25088 -> 2048 fully
//   connected layer, then 2048 -> 2048 (roi_heads/box_head/roi_box_feature_extractors.py:85-130) -- its class / box predictors
//   (roi_box_predictors.py) and the mask predictor's 2x2 / stride-2 transposed convolution + 1x1 logits (mask_head/roi_mask_predictors.py),
//   which rounds 1-2 handed to hipBLASLt through torch.addmm ("not a hand-written kernel").
//
// Both MFMA operands come straight from global memory, as in pointwise.hip, from a PACKED form [row / 16][k / 16][k % 16 / 4][row % 16]
// [k % 4] (drc_linear_pack_rows; zero-padded to whole 16 x 16 blocks): lane (row r, k group g) loads the float4 of columns 4g .. 4g+3 of
// its row, the 64 lanes of a wave read one contiguous KiB, and k-step s of a 16-column chunk uses component s on both sides
// (v_mfma_f32_16x16x4_f32: A = w rows, B = x rows, D[n][m]).  (Unpacked, a wave instruction touches 16 rows 100 KB apart -- 16 pages:
// 62 TFLOP/s on the box head's first layer; packed: see DESIGN.)  The weights are packed once per parameter version, the activations
// per call.  A wave owns a 64 x 64 output tile (4 x 4 MFMA tiles, 64 accumulator registers): 8 float4 loads per 64 MFMAs, two chunks
// in flight.  M is the ROI count (a few
// hundred), so M x N tiles alone cannot fill 1024 SIMDs: K is split over `ksplit` waves per tile; each stores its partial tile and a
// second launch adds the partials in split order (no atomics: the result does not depend on the schedule), then bias and ReLU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kLinWaves = 4;
#ifndef LIN_TARGET_WAVES
#define LIN_TARGET_WAVES 1024
#endif

// grid: ceil(mt / 4) * nt * ksplit blocks of 4 waves (mt, nt = 64-row tiles of M, N).  The four waves of a block take four M tiles of the
// same (n tile, K split): they read the same 64 rows of w (the big operand: 205 MB for the box head's first layer) through one L1
__global__ __launch_bounds__(64 * kLinWaves) void linear_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ y, float* __restrict__ part, int M, int N, int K, int ksplit,
                                                               int kchunks_per_split, int relu) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const int mt = (M + 63) >> 6, nt = (N + 63) >> 6;
    // block id -> (K split fastest, n tile, M-tile group): with ksplit a multiple of 8 an XCD (block id % 8) only ever touches ITS K slices
    // of x and w -- x's slice (3.75 MB of the box head's 30 MB) stays in the 4 MB L2 while w streams through once
    int id = blockIdx.x;
    const int ks = id % ksplit; id /= ksplit;
    const int tn = id % nt;
    const int tm = (id / nt) * kLinWaves + wave;
    if (tm >= mt) return;                                   // wave-uniform; no workgroup barrier in this kernel
    const int kchunks = (K + 15) >> 4;
    const int kc0 = ks * kchunks_per_split;
    int kc1 = kc0 + kchunks_per_split;
    kc1 = kc1 < kchunks ? kc1 : kchunks;

    // 16-row block pointers into the packed operands (blocks past M / N are clamped to the last one and never stored)
    const int mb = (M + 15) >> 4, nb = (N + 15) >> 4;
    const float* xp[4];
    const float* wp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int bm = tm * 4 + i, bn = tn * 4 + i;
        bm = bm < mb ? bm : mb - 1;
        bn = bn < nb ? bn : nb - 1;
        xp[i] = x + (int64_t)bm * kchunks * 256 + lane * 4;
        wp[i] = w + (int64_t)bn * kchunks * 256 + lane * 4;
    }
    f32x4 acc[4][4];                                        // [n tile][m tile]: lane holds n = 4g..4g+3 (rows of D) of column m = r
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto load = [&](f32x4 (&xa)[4], f32x4 (&wa)[4], int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xa[i] = *(const f32x4*)(xp[i] + kc * 256);
            wa[i] = *(const f32x4*)(wp[i] + kc * 256);
        }
    };
    // two chunks in flight behind the one being multiplied (three register sets, rotated by unrolling the chunk loop by three)
    f32x4 x0[4], w0[4], x1[4], w1[4], x2[4], w2[4];
    auto mul = [&](const f32x4 (&xa)[4], const f32x4 (&wa)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[a][s], xa[b][s], acc[a][b], 0, 0, 0);
    };
    const int last = kc1 - 1;
    auto clampk = [&](int kc) { return kc < last ? kc : last; };     // past the split's end: a harmless reload of its last chunk
    if (kc0 < kc1) {
        load(x0, w0, kc0);
        load(x1, w1, clampk(kc0 + 1));
        int kc = kc0;
        // (the sched_barriers pin the loads ahead of the MFMAs they overlap: hipcc otherwise sinks every load to its use -- vmcnt(0) in
        // front of each chunk's first MFMA, nothing in flight: 63 instead of the TFLOP/s quoted in DESIGN)
        for (; kc + 2 < kc1; kc += 3) {
            load(x2, w2, clampk(kc + 2)); __builtin_amdgcn_sched_barrier(0); mul(x0, w0); __builtin_amdgcn_sched_barrier(0);
            load(x0, w0, clampk(kc + 3)); __builtin_amdgcn_sched_barrier(0); mul(x1, w1); __builtin_amdgcn_sched_barrier(0);
            load(x1, w1, clampk(kc + 4)); __builtin_amdgcn_sched_barrier(0); mul(x2, w2); __builtin_amdgcn_sched_barrier(0);
        }
        if (kc < kc1) { mul(x0, w0); ++kc; }
        if (kc < kc1) { mul(x1, w1); ++kc; }
    }
    // D[n][m]: lane (r, g) holds rows n = 4g + e of column m = r
    float* dst = ksplit > 1 ? part + (int64_t)ks * M * N : y;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int m = tm * 64 + b * 16 + r;
        if (m >= M) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int n0 = tn * 64 + a * 16 + g * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + e;
                if (n >= N) continue;
                float v = acc[a][b][e];
                if (ksplit == 1) {
                    if (bias) v += bias[n];
                    if (relu) v = fmaxf(v, 0.f);
                }
                dst[(int64_t)m * N + n] = v;
            }
        }
    }
}

// y = act(sum_ks part[ks] + bias), splits added in order
__global__ __launch_bounds__(256) void linear_finish_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ y, long MN, int N,
                                                            int ksplit, int relu) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < MN; i += (long)gridDim.x * 256) {
        float v = part[i];
        for (int k = 1; k < ksplit; ++k) v += part[(long)k * MN + i];
        if (bias) v += bias[i % N];
        if (relu) v = fmaxf(v, 0.f);
        y[i] = v;
    }
}

// out[(rb * KC + kc) * 256 + (g * 16 + r) * 4 + e] = a[rb * 16 + r][kc * 16 + g * 4 + e] (0 outside R x K); thread = one float4 of the output
__global__ __launch_bounds__(256) void linear_pack_kernel(const float* __restrict__ a, int R, int K, float* __restrict__ out) {
    const int KC = (K + 15) >> 4;
    const long total = (long)((R + 15) >> 4) * KC * 64;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        const long blk = i >> 6;
        const int kc = (int)(blk % KC);
        const long rb = blk / KC;
        const long row = rb * 16 + (lane & 15);
        const int k = kc * 16 + (lane >> 4) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < R) {
            const float* src = a + row * K + k;
            if (k + 3 < K && !(K & 3)) v = *(const f32x4*)src;
            else {
                if (k < K) v.x = src[0];
                if (k + 1 < K) v.y = src[1];
                if (k + 2 < K) v.z = src[2];
                if (k + 3 < K) v.w = src[3];
            }
        }
        *(f32x4*)(out + i * 4) = v;
    }
}

int pick_ksplit(int M, int N, int K) {
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    const int kchunks = (K + 15) / 16;
    long ks = (LIN_TARGET_WAVES + tiles - 1) / tiles;       // waves per SIMD x 1024
    if (ks > kchunks / 8) ks = kchunks / 8;                 // at least 8 chunks (128 columns) per split
    if (ks >= 12) ks = 16;                                  // multiples of 8: one set of K slices per XCD (see linear_kernel)
    else if (ks >= 5) ks = 8;
    if (ks > kchunks / 8) ks = kchunks / 8 >= 8 ? 8 : (kchunks / 8 > 0 ? kchunks / 8 : 1);
    if (ks < 1) ks = 1;
    return (int)ks;
}

}  // namespace

extern "C" int64_t drc_linear_scratch_floats(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int ks = pick_ksplit(M, N, K);
    return ks > 1 ? (int64_t)ks * M * N : 0;
}

extern "C" int64_t drc_linear_packed_floats(int R, int K) {
    if (R <= 0 || K <= 0) return 0;
    return (int64_t)((R + 15) / 16) * ((K + 15) / 16) * 256;
}

extern "C" int drc_linear_pack_rows(const float* a, int R, int K, float* out, void* stream) {
    if (R < 0 || K <= 0) return -2;
    if (R == 0) return 0;
    if (!a || !out) return -1;
    const long total = drc_linear_packed_floats(R, K) / 4;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(linear_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, R, K, out);
    return (int)hipGetLastError();
}

extern "C" int drc_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int relu, float* scratch,
                              int64_t scratch_floats, void* stream) {
    if (M < 0 || N <= 0 || K <= 0) return -2;
    if (M == 0) return 0;
    if (!x || !w || !y) return -1;
    const int ks = pick_ksplit(M, N, K);
    if (ks > 1 && (!scratch || scratch_floats < (int64_t)ks * M * N)) return -2;
    const int kchunks = (K + 15) / 16;
    const int per = (kchunks + ks - 1) / ks;
    const long blocks_ = (long)(((M + 63) / 64 + kLinWaves - 1) / kLinWaves) * ((N + 63) / 64) * ks;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(linear_kernel, dim3((unsigned)blocks_), dim3(64 * kLinWaves), 0, s, x, w, bias, y, scratch, M, N, K, ks, per,
                       relu);
    if (ks > 1) {
        const long MN = (long)M * N;
        long blocks = (MN + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(linear_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)scratch, bias, y, MN, N, ks, relu);
    }
    return (int)hipGetLastError();
}
