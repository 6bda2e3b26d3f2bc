// cout1_mfma.hip -- the 32->1 classifier convolution (3x3x3, stackhourglass.py:78-88 `classifN[2]`, + the cumulative head
// add of :142-144) as a 1x1x1 GEMM on the MFMA followed by a shifted sum (gfx950 / CDNA4).
//
//   out[o] = sum_t sum_c w[t][c] * x[c][o + off(t)]  =  sum_t P[t][o + off(t)],   P[t][v] = sum_c w[t][c] * x[c][v]
//
// P is a pointwise 32 -> 27 ("tap channels") product: a plain [27 x 32] x [32 x voxels] GEMM, which the fp32 MFMA does at
// 16 MFMAs per 16 voxels with no padding waste beyond 27 -> 32 rows; the scalar kernel it replaces spends 864 FMAs per voxel
// on the vector pipe.  A wave owns R rows of one ROI (R*W <= 112 voxels) and walks the depth: per input slice it computes P
// for its rows plus one halo row each side (B operand straight from global memory: lane (voxel, g) loads channels 4g..4g+3 of
// a 16-channel block as one coalesced float4), writes P to LDS as [tap][row][col] with zero pad columns, and gathers the 27
// shifted values per output voxel in a fixed order (no atomics: results do not depend on the batch or the schedule).
// Three partial output slices (depth taps 0,1,2 -> od = d+1, d, d-1) live in registers.
// Round 3: with few ROIs (16 crops at 24x56x56: 448 columns for 1024 SIMDs, each walking 24 slices: 257 us for 179 MB) the depth is cut into
// `segs` segments (grid.y): a wave emits the output slices [o0, o1) of its segment from the input slices o0-1 .. o1 (two extra slices
// per segment, no atomics, the same fixed summation order per output).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define C1M_WAVES 4

namespace {

// x: blocked [N][2][D+2][H+2][W+2][16] (halo 1); w: [27][32]; res/out: dense [N][D][H][W]
__global__ __launch_bounds__(64 * C1M_WAVES) void cout1_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                    const float* __restrict__ res, float* __restrict__ out, int N, int D, int H,
                                                                    int W, int R, int segs) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    const int n_rt = (H + R - 1) / R;
    // XCD-aware order: workgroup b runs on XCD b % 8, each XCD has its own L2.  In launch order the row tiles of one ROI -- which share their
    // halo rows -- went to eight different L2s and the halo rows came from HBM once per XCD (fetch 1.37x the tensor, profiles/r5_pmc.md at
    // 5429fbb); here XCD q takes the q-th contiguous eighth of the columns (gridDim.x is a multiple of 8).
    const unsigned bx = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int col = (int)bx * C1M_WAVES + wave;          // (n, row tile)
    if (col >= N * n_rt) return;                         // wave-uniform; no workgroup barrier
    const int n = col / n_rt, oh0 = (col - n * n_rt) * R;
    const int o0 = (int)((long)blockIdx.y * D / segs), o1 = (int)((long)(blockIdx.y + 1) * D / segs);   // output slices of this segment
    const int d_begin = o0 > 0 ? o0 - 1 : 0, d_end = o1 < D ? o1 : D - 1;                              // input slices it reads
    if (o0 >= o1) return;
    const int Wp = W + 2, Hp = H + 2, Dp = D + 2;
    const int prow = W + 2;                              // P row stride (zero pad column each side)
    const int rows_p = R + 2;
    const int tap_stride = rows_p * prow;
    float* P = lds_all + wave * (28 * tap_stride);

    // zero the pad columns of every tap plane once (never written afterwards)
    for (int i = lane; i < 28 * rows_p * 2; i += 64) {
        const int t = i / (rows_p * 2), rr = (i >> 1) % rows_p;
        P[t * tap_stride + rr * prow + ((i & 1) ? prow - 1 : 0)] = 0.f;
    }

    // A operand: weights, tile a = taps 16a..16a+15, k-step s of block cb uses channel cb*16 + 4g + s  (lane: tap j, k member g)
    float wa[2][2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int t = a * 16 + j;
                wa[a][cb][s] = t < 27 ? w[t * 32 + cb * 16 + 4 * g + s] : 0.f;
            }

    const int nvox = rows_p * W;                         // voxels of a slice tile incl. the two halo rows
    const int ntile = (nvox + 15) >> 4;
    const int64_t cbs = (int64_t)Dp * Hp * Wp * 16;      // channel-block stride
    const float* xn = x + (int64_t)n * 2 * cbs;
    // output voxels of this lane: o = lane and lane + 64 (< R*W <= 112)
    const int nout = R * W;
    int oy[2], ox[2];
    bool ov[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int o = lane + 64 * u;
        oy[u] = o / W; ox[u] = o - oy[u] * W;
        ov[u] = o < nout && oh0 + oy[u] < H;
        if (o >= nout) { oy[u] = 0; ox[u] = 0; }
    }
    float a0[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f}, a2[2] = {0.f, 0.f};   // partial outputs of od = d+1, d, d-1

    // B-operand loads run one voxel tile ahead of the MFMAs (across slices too): the only exposed HBM round trip is the first
    auto tile_ptr = [&](int d, int vt, bool& vin, int& r, int& c) __attribute__((always_inline)) -> const float* {
        int v = vt * 16 + j;
        vin = v < nvox;
        v = vin ? v : nvox - 1;
        r = v / W; c = v - r * W;
        int row = oh0 + r;                               // padded input row (real row oh0 - 1 + r)
        row = row < Hp ? row : Hp - 1;                   // ragged last tile: stay inside the tensor (the halo row is zero)
        return xn + ((int64_t)(d + 1) * Hp + row) * Wp * 16 + (int64_t)(c + 1) * 16 + g * 4;
    };
    bool vin_n; int r_n, c_n;
    const float* xp_n = tile_ptr(d_begin, 0, vin_n, r_n, c_n);
    f32x4 nb0 = *(const f32x4*)xp_n, nb1 = *(const f32x4*)(xp_n + cbs);

    for (int d = d_begin; d <= d_end; ++d) {
        // ---- P = W x X over the tile's voxels
        for (int vt = 0; vt < ntile; ++vt) {
            const f32x4 b0 = nb0, b1 = nb1;
            const bool vin = vin_n;
            const int r = r_n, c = c_n;
            {   // next tile (of this slice, or the first of the next slice; past the end: a harmless reload)
                const bool last = vt + 1 == ntile;
                const int dn = last ? (d + 1 <= d_end ? d + 1 : d) : d;
                xp_n = tile_ptr(dn, last ? 0 : vt + 1, vin_n, r_n, c_n);
                nb0 = *(const f32x4*)xp_n; nb1 = *(const f32x4*)(xp_n + cbs);
            }
            f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0][0][s], b0[s], p0, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1][0][s], b0[s], p1, 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0][1][s], b1[s], p0, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1][1][s], b1[s], p1, 0, 0, 0);
            }
            if (vin) {                                   // lane (voxel j, g) holds taps 4g..4g+3 (p0) and 16+4g.. (p1)
                float* pv = P + r * prow + c + 1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pv[(4 * g + e) * tap_stride] = p0[e];
                    if (16 + 4 * g + e < 27) pv[(16 + 4 * g + e) * tap_stride] = p1[e];
                }
            }
        }
        // ---- gather: tap (kd,kh,kw) of this slice feeds output slice d + 1 - kd at (y + kh, x + kw) of the padded P plane
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float* pg = P + oy[u] * prow + ox[u];
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int off = kh * prow + kw, t = kh * 3 + kw;
                    s0 += pg[t * tap_stride + off];
                    s1 += pg[(9 + t) * tap_stride + off];
                    s2 += pg[(18 + t) * tap_stride + off];
                }
            a0[u] += s0; a1[u] += s1; a2[u] += s2;
        }
        // output slice d-1 is complete (emitted when it belongs to this segment)
        if (d - 1 >= o0 && d - 1 < o1) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (ov[u]) {
                    const int64_t oi = (((int64_t)n * D + (d - 1)) * H + oh0 + oy[u]) * W + ox[u];
                    out[oi] = a2[u] + (res ? res[oi] : 0.f);
                }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) { a2[u] = a1[u]; a1[u] = a0[u]; a0[u] = 0.f; }
    }
    // last slice: od = D-1 has no contribution from a slice D (zero halo)
    if (o1 == D)
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (ov[u]) {
            const int64_t oi = (((int64_t)n * D + (D - 1)) * H + oh0 + oy[u]) * W + ox[u];
            out[oi] = a2[u] + (res ? res[oi] : 0.f);
        }
}

}  // namespace

// returns 1 if the shape is not handled here (caller falls back to the scalar kernels), 0 on launch, or a hipError_t
extern "C" int drc_conv3d_cout1_mfma_try(const float* x, const float* w, const float* res, float* out, int N, int cb_in, int D, int H, int W,
                                         void* stream) {
    if (cb_in != 2 || W > 112 || W < 1) return 1;
    int R = 112 / W;
    if (R > H) R = H;
    while (R > 1 && H % R && (H + R - 1) / R * R - H > R / 2) --R;     // avoid a mostly empty last row tile
    const size_t lds = (size_t)C1M_WAVES * 28 * (R + 2) * (W + 2) * 4;
    if (lds > 160 * 1024) return 1;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)cout1_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const long cols = (long)N * ((H + R - 1) / R);
    // depth segments when the columns alone leave SIMDs idle: enough waves for two per SIMD, at least four output slices each (two extra
    // input slices per segment)
    long segs = cols >= 1024 ? 1 : (2048 + cols - 1) / cols;          // (one wave per SIMD or more: the extra slices cost more than they fill)
    if (segs > D / 4) segs = D / 4;
    if (segs < 1) segs = 1;
    const long gx = ((cols + C1M_WAVES - 1) / C1M_WAVES + 7) / 8 * 8;   // a multiple of 8 (XCD-aware order; surplus waves return at once)
    hipLaunchKernelGGL(cout1_mfma_kernel, dim3((unsigned)gx, (unsigned)segs), dim3(64 * C1M_WAVES), lds,
                       (hipStream_t)stream, x, w, res, out, N, D, H, W, R, (int)segs);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
