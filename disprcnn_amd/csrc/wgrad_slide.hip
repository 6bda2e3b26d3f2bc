// wgrad_slide.hip -- weight gradient of the stride-1 3x3x3 convolutions with a sliding depth window (gfx950 / CDNA4).
//
//   gw[ca][cb][t] += sum_{n,o} a[n, ca, o + tap_t] * b[n, cb, o]        a = layer input x, b = gradient of the conv output
//   reference: autograd of the nn.Conv3d layers of stackhourglass.py:63-88 / :7-20 (dres0/1, classifN[0], hourglass conv2/conv4)
//
// wgrad.hip gives a wave one depth tap and 9 accumulators and re-stages the a and b tiles for every (n, od) group with the
// latency exposed: 13 MFMAs per staged KiB.  Here a wave owns one 16x16 channel-block pair and ALL 27 taps (27 accumulator
// tiles) and walks the depth of a column (n, row tile): per output slice it stages ONE new a slice (ring of three) and the b
// tile, both one slice ahead of the MFMAs, and runs 27 * (R*WT/4) MFMAs on them -- 3x the MFMAs per staged byte, no exposed
// staging inside a column, one atomicAdd flush per wave at the very end.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

#define WS_WAVES 4
#define WS_PA 10     /* LDS-DMA pieces of an a slice tile: (R+2)*(WT+2)*4 units of 16 B */
#define WS_PB 6      /* pieces of a b tile: R*WT*4 units */
#define WS_MAXK 16   /* k-steps (4 voxels each): R*WT <= 64 */

namespace {

typedef const __attribute__((address_space(3))) float ws_lds_f;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// NK: the tile's k-steps (R*WT/4, rounded up) as a compile-time constant, 0 = run-time (any tile).  With a run-time count every stage of
// the software pipeline sat behind a wave-uniform branch, the b value behind an exec-masked read, and the compiler's LDS wait in front of
// each stage's MFMAs became lgkmcnt(0): it also waited for the NEXT stage's reads, issued a moment earlier -- the pipeline overlapped
// nothing (MFMA busy 49 %; the staging DMAs, ablated, were 3 % of the time).  Straight-line stages get counted waits (LDS returns in order).
template <int NK>
__global__ __launch_bounds__(64 * WS_WAVES) void wgrad_slide_kernel(const drc_wgrad_params p, int R, int WT, int nseg, int seglen) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;   // a-channel (A operand row) / b-channel (B operand column)
    const int g = lane >> 4;   // k member: voxel 4m+g

    const int cbb = blockIdx.y % p.cb_b, ca = blockIdx.y / p.cb_b;
    const int D = p.OD, H = p.OH, W = p.OW;
    const int n_wt = (W + WT - 1) / WT, n_rt = (H + R - 1) / R;
    const int cols = p.N * n_rt * n_wt;
    const int workers = gridDim.x * WS_WAVES;
    const int seg = WT + 2, rows_a = R + 2;
    const int units_a = rows_a * seg * 4, units_b = R * WT * 4;
    const int pieces_a = (units_a + 63) >> 6, pieces_b = (units_b + 63) >> 6;
    const int a_floats = pieces_a * 256, b_floats = pieces_b * 256;
    float* lds_a = lds_all + wave * (3 * a_floats + 2 * b_floats);     // ring of three a slices
    float* lds_b = lds_a + 3 * a_floats;                               // two b tiles
    const int nslots = R * WT, nk = NK ? NK : (nslots + 3) >> 2;
    const unsigned mag_a = ((1u << 20) + seg * 4 - 1) / (seg * 4), mag_b = ((1u << 20) + WT * 4 - 1) / (WT * 4);

    // per-lane offsets of voxel slot 4m+g inside an a slice tile (tap (0,0)), and its (row, col)
    int a_off[WS_MAXK];
#pragma unroll
    for (int m = 0; m < WS_MAXK; ++m) {
        const int s = 4 * m + g;
        int r = s / WT, c = s - r * WT;
        if (s >= nslots) { r = 0; c = 0; }
        a_off[m] = (r * seg + c) * 16 + j;
    }

    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    unsigned poff_a[WS_PA], poff_b[WS_PB];
    // work unit = (column, depth segment): a column's D output slices are cut into nseg runs of seglen so that the units divide
    // evenly among the waves (32->32 at 64 ROIs: 896 columns over 256 waves = 3.5 per wave -> 1,792 half columns = 7 per wave)
    for (int unit = blockIdx.x * WS_WAVES + wave; unit < cols * nseg; unit += workers) {
        const int col = unit / nseg, sgi = unit - col * nseg;
        const int od_lo = sgi * seglen, od_hi = od_lo + seglen < D ? od_lo + seglen : D;
        int q = col;
        const int wt = q % n_wt; q /= n_wt;
        const int rt = q % n_rt;
        const int n = q / n_rt;
        const int oh0 = rt * R, ow0 = wt * WT;
        const int nr = H - oh0 < R ? H - oh0 : R, nc = W - ow0 < WT ? W - ow0 : WT;   // valid rows / columns (ragged tiles)
        // per-lane global byte offsets of the pieces (all lanes active; outside a ragged tile's needed rows / columns: a valid neighbour)
#pragma unroll
        for (int k = 0; k < WS_PA; ++k) {
            int u = k * 64 + lane;
            u = u < units_a ? u : units_a - 1;
            int r = (int)(((unsigned)u * mag_a) >> 20);
            int cu = u - r * seg * 4;
            int vx = cu >> 2;
            r = r < nr + 1 ? r : nr + 1;
            vx = vx < nc + 1 ? vx : nc + 1;
            poff_a[k] = (unsigned)((r * (int)p.a_h_stride + vx * 16 + (cu & 3) * 4) * 4);
        }
#pragma unroll
        for (int k = 0; k < WS_PB; ++k) {
            int u = k * 64 + lane;
            u = u < units_b ? u : units_b - 1;
            int r = (int)(((unsigned)u * mag_b) >> 20);
            int cu = u - r * WT * 4;
            int vx = cu >> 2;
            r = r < nr - 1 ? r : nr - 1;
            vx = vx < nc - 1 ? vx : nc - 1;
            poff_b[k] = (unsigned)((r * (int)p.b_h_stride + vx * 16 + (cu & 3) * 4) * 4);
        }
        // voxels of the tile outside the output grid contribute nothing: their b value is forced to zero
        unsigned okmask = 0;
#pragma unroll
        for (int m = 0; m < WS_MAXK; ++m) {
            const int s = 4 * m + g;
            const int r = s / WT, c = s - r * WT;
            if (s < nslots && r < nr && c < nc) okmask |= 1u << m;
        }
        // (the b tile is staged from clamped, i.e. valid and finite, addresses: a multiplication by 0 / 1 masks it without a branch)
        float okf[WS_MAXK];
#pragma unroll
        for (int m = 0; m < WS_MAXK; ++m) okf[m] = (float)((okmask >> m) & 1u);
        const char* abase = (const char*)(p.a + (int64_t)n * p.a_n_stride + (int64_t)ca * p.a_cb_stride + (int64_t)p.dd0 * p.a_d_stride +
                                          (int64_t)(oh0 + p.dh0) * p.a_h_stride + (int64_t)(ow0 + p.dw0) * 16);
        const char* bbase = (const char*)(p.b + p.b_off0 + (int64_t)n * p.b_n_stride + (int64_t)cbb * p.b_cb_stride + (int64_t)oh0 * p.b_h_stride +
                                          (int64_t)ow0 * 16);
        const int64_t a_ds = p.a_d_stride * 4, b_ds = p.b_d_stride * 4;
        auto stage_a = [&](int ps) __attribute__((always_inline)) {       // padded depth slice ps of a -> ring slot ps % 3
            float* dst = lds_a + (ps % 3) * a_floats;
            const char* sb = abase + (int64_t)ps * a_ds;
#pragma unroll
            for (int k = 0; k < WS_PA; ++k)
                if (k < pieces_a) __builtin_amdgcn_global_load_lds(GLOBAL_PTR(sb + poff_a[k]), LDS_PTR(dst + k * 256), 16, 0, 0);
        };
        auto stage_b = [&](int od) __attribute__((always_inline)) {
            float* dst = lds_b + (od & 1) * b_floats;
            const char* sb = bbase + (int64_t)od * b_ds;
#pragma unroll
            for (int k = 0; k < WS_PB; ++k)
                if (k < pieces_b) __builtin_amdgcn_global_load_lds(GLOBAL_PTR(sb + poff_b[k]), LDS_PTR(dst + k * 256), 16, 0, 0);
        };

        // all LDS reads of the previous column are consumed (in-order wave) before its tiles are overwritten
        stage_a(od_lo); stage_a(od_lo + 1); stage_a(od_lo + 2); stage_b(od_lo);
        for (int od = od_lo; od < od_hi; ++od) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // a[od+2] and b[od] (issued one slice ago) have landed
            const float* bt = lds_b + (od & 1) * b_floats;
            // Software pipeline over the (depth tap, k-step) stages of the slice: the ten operand reads of the NEXT stage are issued
            // before the nine MFMAs of the current one (two register sets, alternating statically).  Without it every MFMA pair
            // waited for its own ds_read (MFMA busy ~30 %: 47 TFLOP/s).
            float avs[2][9], bvs[2];
            auto fetch = [&](int set, int kd, int m, int aoff) __attribute__((always_inline)) {
                const float* at = lds_a + ((od + kd) % 3) * a_floats;
                if constexpr (NK > 0) {
                    // exactly SEVEN LDS instructions, written out: the stage's wait below counts them (14 in flight at most: lgkmcnt is a
                    // 4-bit counter).  Row r of the 3x3 taps: columns 0, 1 as one ds_read2_b32 (offsets in dwords), column 2 at +128 B; the
                    // b value is masked at use (a multiply here would wait for the read it follows).
                    const unsigned ab = (unsigned)(uintptr_t)(ws_lds_f*)(at + aoff), bb_ = (unsigned)(uintptr_t)(ws_lds_f*)(bt + (4 * m + g) * 16 + j);
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        f32x2 pr;
                        const unsigned ar = ab + (unsigned)(r * seg * 64);
                        asm volatile("ds_read2_b32 %0, %1 offset1:16" : "=v"(pr) : "v"(ar));
                        asm volatile("ds_read_b32 %0, %1 offset:128" : "=v"(avs[set][r * 3 + 2]) : "v"(ar));
                        avs[set][r * 3] = pr.x; avs[set][r * 3 + 1] = pr.y;
                    }
                    asm volatile("ds_read_b32 %0, %1" : "=v"(bvs[set]) : "v"(bb_));
                } else {
                    bvs[set] = ((okmask >> m) & 1u) ? bt[(4 * m + g) * 16 + j] : 0.f;
#pragma unroll
                    for (int t = 0; t < 9; ++t) avs[set][t] = at[aoff + ((t / 3) * seg + (t % 3)) * 16];
                }
            };
            fetch(0, 0, 0, a_off[0]);
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
#pragma unroll
                for (int m = 0; m < WS_MAXK; ++m) {
                    if (NK > 0 ? m < NK : m < nk) {                    // compile-time (NK) or wave-uniform
                        const int cur = m & 1;                         // every depth tap starts in set 0
                        // next stage: (kd, m+1) into the other set, or (kd+1, 0) into set 0 after the last k-step of this tap -- before
                        // the MFMAs when this stage runs out of set 1, after them when it occupies set 0 itself (odd nk)
                        const bool last_m = !(m + 1 < nk);
                        if (!last_m) {
                            if (m + 1 < WS_MAXK) fetch(cur ^ 1, kd, m + 1, a_off[m + 1 < WS_MAXK ? m + 1 : 0]);
                        } else if (kd < 2 && cur == 1) {
                            fetch(0, kd + 1, 0, a_off[0]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (NK > 0) {
                            // LDS returns in order: this stage's seven reads are older than the next stage's seven, which stay in flight.  (The
                            // compiler's own wait here was lgkmcnt(0).)  gfx9 s_waitcnt: vmcnt[3:0] | expcnt << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14
                            const bool fetched = !last_m || (kd < 2 && cur == 1);
                            if (fetched) __builtin_amdgcn_s_waitcnt(0xC07F | (7 << 8));
                            else __builtin_amdgcn_s_waitcnt(0xC07F);
                            __builtin_amdgcn_sched_barrier(0);
                            // (operands of `asm volatile` LDS reads: every use is tied behind the hand-placed wait)
                            asm volatile("" : "+v"(bvs[cur]));
#pragma unroll
                            for (int t = 0; t < 9; ++t) asm volatile("" : "+v"(avs[cur][t]));
                        }
                        const float bm = NK > 0 ? (okf[m] != 0.f ? bvs[cur] : 0.f) : bvs[cur];      // select, not multiply (Inf / NaN in a ragged slot)
#pragma unroll
                        for (int t = 0; t < 9; ++t)
                            acc[kd * 9 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(avs[cur][t], bm, acc[kd * 9 + t], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (last_m && kd < 2 && cur == 0) fetch(0, kd + 1, 0, a_off[0]);
                    }
                }
                if (kd == 0 && od + 1 < od_hi) {   // slot od % 3 is free now: stage the slice after next into it, and the next b tile
                    __builtin_amdgcn_sched_barrier(0);
                    stage_a(od + 3);
                    stage_b(od + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    // ---- flush: D[i = a channel][j = b channel]: lane holds rows 4g..4g+3 of column j
    if (p.scratch) {                               // partial sums [job][worker][t][lane] for wgrad.hip's wgrad_reduce_kernel
        float* dst = p.scratch + (((int64_t)blockIdx.y * workers + blockIdx.x * WS_WAVES + wave) * 27) * 256 + lane * 4;
#pragma unroll
        for (int t = 0; t < 27; ++t) *(f32x4*)(dst + t * 256) = acc[t];
        return;
    }
    const int cbt = p.cb_b * 16;
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ia = ca * 16 + g * 4 + r, ib = cbb * 16 + j;
            atomicAdd(p.gw + ((int64_t)ia * cbt + ib) * 27 + t, acc[t][r]);
        }
}

}  // namespace

bool drc_wgrad_scratch_fits(const drc_wgrad_params& p, long waves, int NT);                    // wgrad.hip
int drc_wgrad_reduce(const drc_wgrad_params& p, int workers, long jobs, int NT, hipStream_t s);
int drc_wgrad_clear_for_atomics(const drc_wgrad_params& p, hipStream_t s);

// returns 1 if the shape is not handled here (caller uses the generic kernel), 0 on launch, or a hipError_t
extern "C" int drc_tapconv_wgrad_slide_try(const drc_wgrad_params* pp, void* stream) {
    drc_wgrad_params p = *pp;
    if (p.in_mul != 1 || p.nd != 3 || p.nh != 3 || p.nw != 3 || p.sd != 1 || p.sh != 1 || p.sw != 1 || p.OD < 2) return 1;
    // tile: rows as wide as the map up to 32 columns, R*WT <= 64 voxels (16 k-steps)
    const int parts = (p.OW + 31) / 32;
    const int WT = (p.OW + parts - 1) / parts;
    int R = 64 / WT;
    if (R > p.OH) R = p.OH;
    if (R < 1) return 1;
    while (R > 1 && p.OH % R && (p.OH + R - 1) / R * R - p.OH > R / 2) --R;
    const int pieces_a = ((R + 2) * (WT + 2) * 4 + 63) / 64, pieces_b = (R * WT * 4 + 63) / 64;
    if (pieces_a > WS_PA || pieces_b > WS_PB || R * WT > 4 * WS_MAXK) return 1;
    const size_t lds = (size_t)WS_WAVES * (3 * pieces_a + 2 * pieces_b) * 1024;
    if (lds > 160 * 1024) return 1;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)wgrad_slide_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_slide_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_slide_kernel<14>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_slide_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const long cols = (long)p.N * ((p.OH + R - 1) / R) * ((p.OW + WT - 1) / WT);
    const long jobs = (long)p.cb_a * p.cb_b;
    long workers = 1024 / (jobs > 0 ? jobs : 1);            // one wave per SIMD across all jobs (atomicAdd flushes per wave)
    if (workers > cols * 4) workers = cols * 4;              // up to four depth segments per column (below)
    if (workers < 1) workers = 1;
    dim3 grid((unsigned)((workers + WS_WAVES - 1) / WS_WAVES), (unsigned)jobs, 1);
    const bool partial = drc_wgrad_scratch_fits(p, (long)grid.x * WS_WAVES * jobs, 27);
    if (!partial) {
        p.scratch = nullptr;
        if (const int st = drc_wgrad_clear_for_atomics(p, (hipStream_t)stream)) return st;
    }
    // depth segments: the split (<= 4 runs of >= 3 slices) with the shortest critical path, rounds x (slices per unit + staging prologue)
    const long nworkers = (long)grid.x * WS_WAVES;
    int nseg = 1, seglen = p.OD;
    double best = 1e30;
    for (int s = 1; s <= 4; ++s) {
        const int len = (p.OD + s - 1) / s;
        if (s > 1 && len < 3) break;
        const int ns = (p.OD + len - 1) / len;
        const double cost = (double)((cols * ns + nworkers - 1) / nworkers) * (len + 0.75);
        if (cost < best - 1e-9) { best = cost; nseg = ns; seglen = len; }
    }
    // k-steps of the regressor's maps: 28- and 14-wide -> 56 slots (14), 7-wide -> 49 (13), full 64-slot tiles (16); anything else: run-time count
    switch ((R * WT + 3) / 4) {
        case 13: hipLaunchKernelGGL(wgrad_slide_kernel<13>, grid, dim3(64 * WS_WAVES), lds, (hipStream_t)stream, p, R, WT, nseg, seglen); break;
        case 14: hipLaunchKernelGGL(wgrad_slide_kernel<14>, grid, dim3(64 * WS_WAVES), lds, (hipStream_t)stream, p, R, WT, nseg, seglen); break;
        case 16: hipLaunchKernelGGL(wgrad_slide_kernel<16>, grid, dim3(64 * WS_WAVES), lds, (hipStream_t)stream, p, R, WT, nseg, seglen); break;
        default: hipLaunchKernelGGL(wgrad_slide_kernel<0>, grid, dim3(64 * WS_WAVES), lds, (hipStream_t)stream, p, R, WT, nseg, seglen);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess || !partial) return e == hipSuccess ? 0 : (int)e;
    return drc_wgrad_reduce(p, (int)grid.x * WS_WAVES, jobs, 27, (hipStream_t)stream);
}
