// wgrad.hip -- weight gradients of the tap-grid convolutions on v_mfma_f32_16x16x4_f32 (gfx950 / CDNA4).
//
//   gw[ca][cb][t] += sum_{n,o} a[n, ca, in_mul*o + tap_t] * b[n, cb, o]
//
// `a` is the tensor the taps slide over, `b` the tensor sampled at the plain positions o:
//   Conv3d/Conv2d (any stride)      : a = x (layer input),  b = dy   -> gw[ci][co][t]     (weight.grad[co][ci][t])
//   ConvTranspose3d k3 s2 p1 op1    : a = dy (in_mul = 2),  b = x    -> gw[co][ci][k]     (weight.grad[ci][co][k])
// Reference: autograd of nn.Conv3d / nn.ConvTranspose3d / nn.Conv2d inside submodule.py:13-22, stackhourglass.py:22-30.
//
// GEMM view per tap: [16 a-channels x V] . [V x 16 b-channels], the voxel index is the MFMA k dimension.
// A wave owns one (depth tap, a-channel block, b-channel block) job and walks its share of the (n, od, row tile, col tile)
// groups, keeping the nh*nw accumulators of the job in registers; one atomicAdd flush at the end.  Per group it LDS-DMAs
// the a tile (rows with halo, 16 ch) and the b tile (R x WT voxels, 16 ch) into its own LDS region; 2 waves per SIMD
// overlap one wave's staging with the other's MFMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

#define WG_WAVES 4

namespace {

typedef const __attribute__((address_space(3))) float wg_lds_f;

template <int NT>   // taps per depth tap (nh*nw): 1, 2, 4, 9 (49 handled by NT=49 instantiation)
__global__ __launch_bounds__(64 * WG_WAVES) void wgrad_kernel(const drc_wgrad_params p) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;   // a-channel (A operand row) / b-channel (B operand column)
    const int g = lane >> 4;   // k member: voxel 4m+g

    // job = (depth tap di, a block ca, b block cbb)
    int job = blockIdx.y;
    const int cbb = job % p.cb_b; job /= p.cb_b;
    const int ca = job % p.cb_a;
    const int di = job / p.cb_a;

    const int n_wt = (p.OW + p.WT - 1) / p.WT;
    const int n_rt = (p.OH + p.R - 1) / p.R;
    const int groups = p.N * p.OD * n_rt * n_wt;
    const int workers = gridDim.x * WG_WAVES;
    const int rows_in = p.in_mul * (p.R - 1) + (p.nh - 1) * p.sh + 1;
    const int seg_vox = p.in_mul * (p.WT - 1) + (p.nw - 1) * p.sw + 1;
    const int a_floats = rows_in * seg_vox * 16;
    const int nslots = p.R * p.WT;
    const int nm = (nslots + 3) >> 2;
    float* lds_a = lds_all + wave * (p.lds_bytes_per_wave >> 2);
    float* lds_b = lds_a + a_floats;

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int gid = blockIdx.x * WG_WAVES + wave; gid < groups; gid += workers) {
        int q = gid;
        const int wt = q % n_wt; q /= n_wt;
        const int rt = q % n_rt; q /= n_rt;
        const int od = q % p.OD;
        const int n = q / p.OD;
        const int oh0 = rt * p.R, ow0 = wt * p.WT;
        // ---- stage the a tile (rows_in x seg_vox x 16 ch) and the b tile (R x WT x 16 ch)
        const float* asrc = p.a + (int64_t)n * p.a_n_stride + (int64_t)ca * p.a_cb_stride +
                            (int64_t)(p.in_mul * od + p.dd0 + di * p.sd) * p.a_d_stride +
                            (int64_t)(p.in_mul * oh0 + p.dh0) * p.a_h_stride + (int64_t)(p.in_mul * ow0 + p.dw0) * 16;
        const int a_units = seg_vox * 4;
        for (int r = 0; r < rows_in; ++r)
            for (int u0 = 0; u0 < a_units; u0 += 64) {
                const int u = u0 + lane;
                if (u < a_units)
                    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(asrc + (int64_t)r * p.a_h_stride + u * 4), LDS_PTR(lds_a + r * seg_vox * 16 + u0 * 4), 16, 0, 0);
            }
        const float* bsrc = p.b + p.b_off0 + (int64_t)n * p.b_n_stride + (int64_t)cbb * p.b_cb_stride + (int64_t)od * p.b_d_stride +
                            (int64_t)oh0 * p.b_h_stride + (int64_t)ow0 * 16;
        const int b_units = p.WT * 4;
        for (int r = 0; r < p.R; ++r)
            for (int u0 = 0; u0 < b_units; u0 += 64) {
                const int u = u0 + lane;
                if (u < b_units)
                    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(bsrc + (int64_t)r * p.b_h_stride + u * 4), LDS_PTR(lds_b + r * p.WT * 16 + u0 * 4), 16, 0, 0);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ---- accumulate: per k-step (4 voxel slots, this lane's is 4m+g) read B once, then one MFMA per tap with the tap-shifted A
        // value.  The k loop is a runtime loop, software-pipelined by one step in two register sets (round 3: the fully unrolled form
        // guarded every step by `m < nm`, and the branches kept the reads of a step from being issued under the previous step's MFMAs:
        // 10 ds_reads, lgkmcnt(0), 9 MFMAs, repeat).  Slots outside the tile or the output grid (ragged last tiles) get b = 0.
        int sr = g / p.WT, sc = g - sr * p.WT;           // (row, col) of this lane's slot, advanced by 4 slots per step
        // Round 4: for NT <= 9 the NT + 1 LDS reads of a step are written out as ds_read_b32 and the wait in front of a step's MFMAs is
        // COUNTED -- lgkmcnt(NT + 1): this step's reads have landed (LDS returns in order), the next step's stay in flight.  The compiler's
        // own wait there was lgkmcnt(0), which also waited for the reads issued a moment earlier: the "pipeline" overlapped nothing.  The
        // b value is masked at use, by a multiplication (the b tile is staged from valid addresses: finite).
        constexpr bool COUNTED = NT <= 9;
        constexpr int WAIT_FETCHED = 0xC07F | ((COUNTED ? NT + 1 : 0) << 8), WAIT_ALL = 0xC07F;
        auto fetch = [&](float (&av)[NT], float& bv, float& okf, int m) __attribute__((always_inline)) {
            const int slot = 4 * m + g;
            const bool ok = slot < nslots && oh0 + sr < p.OH && ow0 + sc < p.OW;
            const int r = slot < nslots ? sr : 0, c = slot < nslots ? sc : 0;
            const float* ap = lds_a + (p.in_mul * r * seg_vox + p.in_mul * c) * 16 + j;
            if constexpr (COUNTED) {
                okf = ok ? 1.f : 0.f;
                const unsigned bb_ = (unsigned)(uintptr_t)(wg_lds_f*)(lds_b + (slot < nslots ? slot : 0) * 16 + j);
                const unsigned ab = (unsigned)(uintptr_t)(wg_lds_f*)ap;
                asm volatile("ds_read_b32 %0, %1" : "=v"(bv) : "v"(bb_));
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int tb = t / p.nw, tc = t - tb * p.nw;
                    const unsigned aa = ab + (unsigned)((tb * p.sh * seg_vox + tc * p.sw) * 64);
                    asm volatile("ds_read_b32 %0, %1" : "=v"(av[t]) : "v"(aa));
                }
            } else {
                okf = 1.f;
                const float b_ = lds_b[(slot < nslots ? slot : 0) * 16 + j];
                bv = ok ? b_ : 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int tb = t / p.nw, tc = t - tb * p.nw;
                    av[t] = ap[(tb * p.sh * seg_vox + tc * p.sw) * 16];
                }
            }
            sc += 4;
            while (sc >= p.WT) { sc -= p.WT; ++sr; }
        };
        auto mfmas = [&](float (&av)[NT], float bv, float okf, int wait_imm) __attribute__((always_inline)) {
            if constexpr (COUNTED) {
                if (wait_imm == WAIT_ALL) __builtin_amdgcn_s_waitcnt(WAIT_ALL); else __builtin_amdgcn_s_waitcnt(WAIT_FETCHED);
                __builtin_amdgcn_sched_barrier(0);
                // the operands were written by `asm volatile` LDS reads the compiler's wait-count pass does not see: tie every use to this point
                // (an empty asm with the register as in/out operand), so that no later pass can hoist a use above the hand-placed wait
                asm volatile("" : "+v"(bv));
#pragma unroll
                for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(av[t]));
                bv = okf != 0.f ? bv : 0.f;        // a select, not a multiply: an Inf / NaN read from a ragged (over-read) slot must not become NaN * 0
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv, acc[t], 0, 0, 0);
        };
        float avA[NT], avB[NT], bvA, bvB, okA, okB;
        fetch(avA, bvA, okA, 0);
        int m = 0;
        for (; m + 2 < nm; m += 2) {
            fetch(avB, bvB, okB, m + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(avA, bvA, okA, WAIT_FETCHED);
            __builtin_amdgcn_sched_barrier(0);
            fetch(avA, bvA, okA, m + 2);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(avB, bvB, okB, WAIT_FETCHED);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (m + 2 == nm) {                              // two steps left
            fetch(avB, bvB, okB, m + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(avA, bvA, okA, WAIT_FETCHED);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(avB, bvB, okB, WAIT_ALL);
        } else {                                        // one step left
            mfmas(avA, bvA, okA, WAIT_ALL);
        }
        // all LDS reads of this group are consumed by the MFMAs above before the next group's DMA is issued (in-order wave)
    }

    // ---- flush: D[i = a channel][j = b channel]: lane holds rows 4g..4g+3 of column j
    if (p.scratch) {                               // partial sums [job][worker][t][lane] for wgrad_reduce_kernel
        float* dst = p.scratch + (((int64_t)blockIdx.y * workers + blockIdx.x * WG_WAVES + wave) * NT) * 256 + lane * 4;
#pragma unroll
        for (int t = 0; t < NT; ++t) *(f32x4*)(dst + t * 256) = acc[t];
        return;
    }
    const int ntaps = p.nd * p.nh * p.nw;
    const int cbt = p.cb_b * 16;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tg = di * (p.nh * p.nw) + t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ia = ca * 16 + g * 4 + r, ib = cbb * 16 + j;
            atomicAdd(p.gw + ((int64_t)ia * cbt + ib) * ntaps + tg, acc[t][r]);
        }
    }
}

// gw[...] += sum over the workers' partials, in worker order.  Block = (job, tap): 64 lanes x 4 worker segments.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ scratch, int workers, int NT, int cb_a, int cb_b, int ntaps,
                                                           int overwrite, float* __restrict__ gw) {
    const int job = blockIdx.x / NT, t = blockIdx.x - job * NT;
    const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int per = (workers + 3) >> 2;
    const int w0 = seg * per, w1 = w0 + per < workers ? w0 + per : workers;
    const int64_t stride = (int64_t)NT * 256;
    const float* src = scratch + (((int64_t)job * workers + w0) * NT + t) * 256 + lane * 4;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    int w = w0;
    for (; w + 8 <= w1; w += 8) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *(const f32x4*)(src + k * stride);
#pragma unroll
        for (int k = 0; k < 8; ++k) sum += v[k];
        src += 8 * stride;
    }
    for (; w < w1; ++w, src += stride) sum += *(const f32x4*)src;
    __shared__ f32x4 red[4][64];
    red[seg][lane] = sum;
    __syncthreads();
    if (seg) return;
    sum = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    const int cbb = job % cb_b, ca = (job / cb_b) % cb_a, di = job / (cb_b * cb_a);
    const int j = lane & 15, g = lane >> 4;
    const int cbt = cb_b * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ia = ca * 16 + g * 4 + r, ib = cbb * 16 + j;
        float* dst = gw + ((int64_t)ia * cbt + ib) * ntaps + di * NT + t;
        *dst = overwrite ? sum[r] : *dst + sum[r];
    }
}

}  // namespace

// shared with wgrad_slide.hip: true if the partial-sum path can be used for `waves` waves of NT accumulators each
bool drc_wgrad_scratch_fits(const drc_wgrad_params& p, long waves, int NT) {
    return p.scratch && p.scratch_floats >= waves * (long)NT * 256;
}
// the atomicAdd flush adds: an `overwrite` launch that cannot use the partial sums clears gw first
int drc_wgrad_clear_for_atomics(const drc_wgrad_params& p, hipStream_t s) {
    if (!p.overwrite) return 0;
    return (int)hipMemsetAsync(p.gw, 0, (size_t)p.cb_a * 16 * p.cb_b * 16 * p.nd * p.nh * p.nw * sizeof(float), s);
}
int drc_wgrad_reduce(const drc_wgrad_params& p, int workers, long jobs, int NT, hipStream_t s) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(jobs * NT)), dim3(256), 0, s, p.scratch, workers, NT, p.cb_a, p.cb_b,
                       p.nd * p.nh * p.nw, p.overwrite, p.gw);
    return (int)hipGetLastError();
}

namespace {

template <int NT>
int launch(const drc_wgrad_params& p0, hipStream_t s) {
    drc_wgrad_params p = p0;
    const long groups = (long)p.N * p.OD * ((p.OH + p.R - 1) / p.R) * ((p.OW + p.WT - 1) / p.WT);
    const long jobs = (long)p.nd * p.cb_a * p.cb_b;
    // Waves across all jobs.  With the atomicAdd flush every extra wave added a full set of atomics onto the same few addresses (Config-A
    // train step: 8192 waves 60 ms, 1024 waves 41 ms), hence one wave per SIMD.  The partial-sum flush (round 2) costs one coalesced
    // store per accumulator tile and wave, so the kernel is launched with three waves per SIMD (round 4:
    // 1024 -> 3072 waves: 92 -> 64 us on Config B's 2D layers, 172 -> 90 us on Config A's strided / transposed ones, wgrad_reduce 5 -> 7
    // us; four were slower again).  Falls back to one per SIMD when the partial sums of that many waves do not fit the scratch.
    long per_simd = 3;        // (also where the LDS tiles admit fewer resident waves: the shorter work lists balance better -- measured)
    long workers = 1;
    for (; per_simd >= 1; --per_simd) {
        workers = per_simd * 1024 / (jobs > 0 ? jobs : 1);
        if (workers > groups) workers = groups;
        if (workers < 1) workers = 1;
        if (per_simd == 1 || drc_wgrad_scratch_fits(p, ((workers + WG_WAVES - 1) / WG_WAVES) * WG_WAVES * jobs, NT)) break;
    }
    const size_t lds = (size_t)p.lds_bytes_per_wave * WG_WAVES;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    dim3 grid((unsigned)((workers + WG_WAVES - 1) / WG_WAVES), (unsigned)jobs, 1);
    const bool partial = drc_wgrad_scratch_fits(p, (long)grid.x * WG_WAVES * jobs, NT);
    if (!partial) {
        p.scratch = nullptr;
        if (const int st = drc_wgrad_clear_for_atomics(p, s)) return st;
    }
    hipLaunchKernelGGL((wgrad_kernel<NT>), grid, dim3(64 * WG_WAVES), lds, s, p);
    const int st = (int)hipGetLastError();
    if (st || !partial) return st;
    return drc_wgrad_reduce(p, (int)grid.x * WG_WAVES, jobs, NT, s);
}

}  // namespace

extern "C" int drc_tapconv_wgrad_slide_try(const drc_wgrad_params* pp, void* stream);   // wgrad_slide.hip

extern "C" int drc_tapconv_wgrad(const drc_wgrad_params* pp, void* stream) {
    if (!pp) return -1;
    const drc_wgrad_params& p = *pp;
    if (!p.a || !p.b || !p.gw) return -1;
    if (p.N < 0 || p.OD <= 0 || p.OH <= 0 || p.OW <= 0 || p.cb_a <= 0 || p.cb_b <= 0) return -2;
    if (p.N == 0) return 0;
    if ((p.in_mul != 1 && p.in_mul != 2) || p.nd < 1 || p.nh < 1 || p.nw < 1 || p.dd0 < 0 || p.dh0 < 0 || p.dw0 < 0) return -2;
    {   // stride-1 3x3x3: sliding depth window, all 27 taps per wave
        const int st = drc_tapconv_wgrad_slide_try(pp, stream);
        if (st != 1) return st;
    }
    if (p.R <= 0 || p.WT <= 0 || p.R * p.WT > 112) return -3;
    const int rows_in = p.in_mul * (p.R - 1) + (p.nh - 1) * p.sh + 1;
    const int seg_vox = p.in_mul * (p.WT - 1) + (p.nw - 1) * p.sw + 1;
    const int need = (rows_in * seg_vox + p.R * p.WT) * 64;
    if (p.lds_bytes_per_wave < need || (p.lds_bytes_per_wave & 15) || (size_t)p.lds_bytes_per_wave * WG_WAVES > 160 * 1024) return -5;
    hipStream_t s = (hipStream_t)stream;
    switch (p.nh * p.nw) {
        case 1: return launch<1>(p, s);
        case 2: return launch<2>(p, s);
        case 4: return launch<4>(p, s);
        case 9: return launch<9>(p, s);
        case 49: return launch<49>(p, s);
    }
    return -4;
}
