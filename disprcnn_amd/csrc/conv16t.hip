// conv16t.hip -- fp16-storage 3x3x3 / 3x3 stride-1 convolutions with the input tile staged in LDS (gfx950 / CDNA4), round 3.
//
//   reference arithmetic: convbn_3d k3 s1 p1 of stackhourglass.py:7-51,63-88 (dres0/dres1, hourglass conv2/conv4, classif[0]) and the
//   stride-1 undilated convbn 3x3 layers of feature_extraction (submodule.py:13-17,24-49,62-88); fp16 storage, fp32 accumulation as in
//   conv16.hip (the reference is fp32-only; this path is held to a stated bound against the fp32 oracle).
//
// Why: conv16.hip reads both MFMA operands straight from global memory.  With `v_mfma_f32_16x16x32_f16` at 16x the fp32 rate a step
// is 8 MFMAs = 128 cycles against 6 KiB of loads, and a wave's 27-tap footprint (3 slices x 10 x 10 voxel lines = 19 KiB) times the
// 8 waves of a CU does not fit the 32 KiB L1: every tap went to L2 (234 TF = 9 % of the f16 peak on the stress shape, unchanged by a
// deeper prefetch ring).  Here a block of four waves stages the (TR+2) x 16-voxel input rows of a 14-column output tile ONCE per
// (channel block, depth tap) with LDS-DMA and all 27 / 9 taps read their B fragments from LDS; only the weights (shared by every
// block, L1-resident) still come through the vector-memory path.
//
//   tile      : TR = 4*RW output rows x 14 columns of one (n, od); wave w owns rows w*RW .. w*RW+RW-1, lane (j, g) column j.
//               Lanes j = 14, 15 compute two columns nobody stores (12.5 % of the MFMA work) so that a staged row is exactly one
//               16-voxel line group: 1 KiB, one global_load_lds per row, no bank conflicts on the reads.
//   LDS row   : [g = 0..3][voxel 0..15][8 halfs]: lane l of the DMA (g = l/16, v = l%16) fetches channels 8g..8g+7 of voxel v and lands at
//               l*16 B; a B fragment of tap kw is ds_read_b128 at g*256 + (j+kw)*16 (+ row*1024): 16 consecutive 16-byte words per g.
//               (j + kw > 15 only for the unused lanes; they read the next plane's first words.)
//   stage     : one (tile, 32-channel block): ND*(TR+2) rows into the block's single buffer, then its 27 / 9 taps.  The DMA is NOT
//               overlapped with the block's own MFMAs: vmcnt retires in order, so a weight load issued behind the next stage's DMA could
//               not be consumed before that DMA had landed.  Instead several blocks are resident per CU (19-56 KiB each) and one
//               block's staging wait is another's MFMA phase.
//   weights   : [tap][cb32][cout_pad][32] fp16 (engine.pack_weight16), buffer loads a few taps ahead (L1-resident: every block of the
//               launch reads the same <= 27 x CT KiB).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

#define T16_WAVES 4
#define T16_COLS 14
// B fragments are re-read from LDS for every tap (volatile): left to itself the compiler keeps each distinct (slice, row, kw) fragment
// of a stage live (18 - 54 x 4 registers), which costs the second and third wave per SIMD that hide the staging waits.
#ifndef T16_OCC_MID
#define T16_OCC_MID 3
#endif
#ifndef T16_BVOL
#define T16_BVOL volatile
#endif
#ifndef T16_WPF_NARROW
#define T16_WPF_NARROW 5   /* weight sets (taps) in flight ahead of the MFMAs, CT <= 2: a tap is only RW*CT 16-cycle MFMAs and the */
#endif                     /* 27 x CT KiB of a stage's weights miss the L1 next to the DMA traffic (L2 round trip ~700+ cycles)    */
#ifndef T16_WPF_WIDE
#define T16_WPF_WIDE 2     /* CT = 4 (16 registers a set) */
#endif

namespace {

template <int RW, int CT, int ND>
__global__ __launch_bounds__(64 * T16_WAVES) __attribute__((amdgpu_waves_per_eu(RW * CT >= 16 ? 2 : (RW * CT >= 8 ? T16_OCC_MID : 4))))
void conv16t_kernel(const drc_tapconv_params p) {
    constexpr int TR = RW * T16_WAVES;                 // output rows per block tile
    constexpr int ROWS = ND * (TR + 2);                // staged rows per stage
    constexpr int NT = ND * 9;
    constexpr int WPF = CT <= 2 ? T16_WPF_NARROW : T16_WPF_WIDE;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    const _Float16* x = (const _Float16*)p.x;
    const drc_tap_class cls = p.cls[0];
    const bool dense1 = p.reserved == 1;

    const int n_ct = (p.OW + T16_COLS - 1) / T16_COLS, n_rt = (p.OH + TR - 1) / TR;
    const int n_cg = p.cout_pad / 16 / CT;
    const unsigned tiles = (unsigned)p.N * p.OD * n_rt * n_ct * n_cg;   // cout group fastest: neighbouring blocks share the input tile in L2
    // 32-bit strides (halfs): the host checks that one unit of x / y / res stays below 2^31 bytes
    const int xh = (int)p.x_h_stride, xd = (int)p.x_d_stride, xc = (int)p.x_cb_stride;
    const int yh = (int)p.y_h_stride, yd_ = (int)p.y_d_stride, yc = (int)p.y_cb_stride;
    const int rh = (int)p.r_h_stride, rd_ = (int)p.r_d_stride, rc = (int)p.r_cb_stride;
    const long w_cb = (long)p.cout_pad * 32, w_tap = w_cb * p.cb_in;   // halfs
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, -1, 0x00020000);

    struct Tile { int n, od, r0, c0, cg; };
    auto tile_of = [&](unsigned t) __attribute__((always_inline)) {
        Tile q;
        unsigned u = t / (unsigned)n_cg; q.cg = (int)(t - u * (unsigned)n_cg); t = u;
        u = t / (unsigned)n_ct; q.c0 = (int)(t - u * (unsigned)n_ct) * T16_COLS; t = u;
        u = t / (unsigned)n_rt; q.r0 = (int)(t - u * (unsigned)n_rt) * TR; t = u;
        u = t / (unsigned)p.OD; q.od = (int)(t - u * (unsigned)p.OD);
        q.n = (int)u;
        return q;
    };
    // DMA the rows of stage (tile, cb): row i = (depth tap i / (TR+2), tile row i % (TR+2)), waves take rows round robin
    auto stage = [&](const Tile& q, int cb) __attribute__((always_inline)) {
        const _Float16* src = x + (long)q.n * p.x_n_stride +
                              (cb * xc + (q.od + cls.dd0) * xd + (q.r0 + cls.dh0) * xh + (q.c0 + cls.dw0 + j) * 32 + g * 8);
        char* dst = lds;
#pragma unroll
        for (int i0 = 0; i0 < ROWS; i0 += T16_WAVES) {
            const int i = i0 + wave;
            if (i < ROWS) {
                const int kd = i / (TR + 2), rr = i - kd * (TR + 2);
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + (kd * xd + rr * xh)), LDS_PTR(dst + i * 1024), 16, 0, 0);
            }
        }
    };

    f32x4 acc[RW][CT];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[r][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const unsigned lane_b = (unsigned)(g * 256 + j * 16 + wave * RW * 1024);        // this lane's B-fragment offset inside a buffer (tap 0)
    // a tile's channel blocks stay on one block: block b walks tiles b, b + gridDim.x, ...
    for (unsigned tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
      const Tile q = tile_of(tile);
      for (int cb = 0; cb < p.cb_in; ++cb) {
        // ---- the stage's taps: weights WPF taps ahead through buffer loads (the first WPF requested ahead of the DMA, so they land
        // with it), B fragments from LDS
        const unsigned wlo = 2u * (unsigned)(((q.cg * CT) * 16 + j) * 32 + g * 8);
        const __attribute__((address_space(3))) char* bb = (const __attribute__((address_space(3))) char*)lds + lane_b;
        f16x8 wt[WPF + 1][CT];
        // taps are requested in order: tap t's weights sit at t * w_tap (canonical [tap][cb32][cout][32] packing, checked by the host)
        unsigned wo = 2u * (unsigned)(cb * (int)w_cb);
        const unsigned wstep = 2u * (unsigned)w_tap;
        auto wfetch = [&](int set) __attribute__((always_inline)) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                wt[set][ct] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, wlo + ct * 1024, wo, 0));
            wo += wstep;
        };
#pragma unroll
        for (int t = 0; t < WPF; ++t) wfetch(t);
        stage(q, cb);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        // software pipeline, one tap deep: tap t+1's B fragments (and tap t+WPF's weights) are requested, then tap t's MFMAs issue;
        // the scheduling fences keep the compiler from hoisting every read of the unrolled stage to its top (registers, spills)
        f16x8 bA[RW], bB[RW];
        auto bfetch = [&](f16x8 (&bv)[RW], int t) __attribute__((always_inline)) {
            const int kd = t / 9, kh = (t - kd * 9) / 3, kw = t - kd * 9 - kh * 3;
#pragma unroll
            for (int r = 0; r < RW; ++r)
                bv[r] = *(const __attribute__((address_space(3))) T16_BVOL f16x8*)(bb + (kd * (TR + 2) + r + kh) * 1024 + kw * 16);
        };
        auto mfmas = [&](const f16x8 (&bv)[RW], int set) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[r][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wt[set][ct], bv[r], acc[r][ct], 0, 0, 0);
        };
        bfetch(bA, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t + 1 < NT) { if (t & 1) bfetch(bA, t + 1); else bfetch(bB, t + 1); }
            if (t + WPF < NT) wfetch((t + WPF) % (WPF + 1));
            __builtin_amdgcn_sched_barrier(0);
            if (t & 1) mfmas(bB, t % (WPF + 1)); else mfmas(bA, t % (WPF + 1));
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- after a tile's last channel block: epilogue (fp32 BN / residual / ReLU, fp16 store) and clear
        if (cb + 1 == p.cb_in) {
            const int col = q.c0 + j;
            const bool col_ok = j < T16_COLS && col < p.OW;
            if (dense1) {
                float* yd = (float*)p.y;
                const float* rd = (const float*)p.res;
                if (g == 0 && q.cg == 0 && col_ok) {
#pragma unroll
                    for (int r = 0; r < RW; ++r) {
                        const int row = q.r0 + wave * RW + r;
                        if (row < p.OH) {
                            const long o = (((long)q.n * p.OD + q.od) * p.OH + row) * p.OW + col;
                            float v = acc[r][0].x;
                            if (rd) v += rd[o];
                            yd[o] = v;
                        }
                    }
                }
            } else {
                _Float16* y = (_Float16*)p.y + p.y_off0 + (long)q.n * p.y_n_stride;
                const _Float16* res = p.res ? (const _Float16*)p.res + p.r_off0 + (long)q.n * p.r_n_stride : nullptr;
                const int row0 = q.r0 + wave * RW;
                const int yl = q.od * yd_ + row0 * yh + col * 32 + g * 4, rl = q.od * rd_ + row0 * rh + col * 32 + g * 4;
                // the residual tile as one batch of loads, then the arithmetic and the stores (not a load -> wait -> store chain per output)
                f16x4 rv[RW][CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int r = 0; r < RW; ++r) {
                        const int cot = q.cg * CT + ct;
                        rv[r][ct] = (f16x4){0, 0, 0, 0};
                        if (res && col_ok && row0 + r < p.OH) rv[r][ct] = *(const f16x4*)(res + rl + (cot >> 1) * rc + r * rh + (cot & 1) * 16);
                    }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int cot = q.cg * CT + ct;
                    const f32x4 sc = *(const f32x4*)(p.scale + cot * 16 + g * 4);
                    const f32x4 sh = *(const f32x4*)(p.shift + cot * 16 + g * 4);
#pragma unroll
                    for (int r = 0; r < RW; ++r) {
                        f32x4 v = acc[r][ct] * sc + sh;
                        v.x += (float)rv[r][ct].x; v.y += (float)rv[r][ct].y; v.z += (float)rv[r][ct].z; v.w += (float)rv[r][ct].w;
                        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        f16x4 hv;
                        hv.x = (_Float16)v.x; hv.y = (_Float16)v.y; hv.z = (_Float16)v.z; hv.w = (_Float16)v.w;
                        if (col_ok && row0 + r < p.OH) *(f16x4*)(y + yl + (cot >> 1) * yc + r * yh + (cot & 1) * 16) = hv;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[r][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // every wave is done reading the buffer before the next stage's DMA overwrites it
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv16s_kernel<RW,CT>: the 3x3x3 layers with ONE 32-channel input block and <= 32 couts (dres0[1], dres1, classif[0], the 32 -> 1
// heads: six of the seven full-resolution layers) as a DEPTH-SLIDING walk.  conv16t re-stages three depth slices per output slice
// (3.7x the input through the DMA path) and cannot overlap that DMA with its MFMAs because its weights share the in-order vmcnt queue.
// Here the layer's whole weight set (27 x CT fragments) is loaded into each wave's REGISTERS once, a block owns a (n, row tile, column tile) COLUMN
// and walks od = 0..OD-1 with a ring of four slice slots: slices od-1, od, od+1 resident, slice od+2 landing while the 27 taps of od
// run -- the MFMA phase issues no vector-memory instruction, so one `vmcnt(0)` + barrier per output slice is all the synchronisation.
// Per (kd, kw) the RW+2 distinct input rows are read once and serve all three kh (6 ds_read_b128 per 24 MFMAs at RW=4, CT=2).
template <int RW, int CT>
__global__ __launch_bounds__(64 * T16_WAVES) void conv16s_kernel(const drc_tapconv_params p) {
    constexpr int TR = RW * T16_WAVES;
    constexpr int SLOT = (TR + 2) * 1024;              // bytes of one staged slice
    extern __shared__ __attribute__((aligned(16))) char lds[];
    typedef const __attribute__((address_space(3))) volatile f16x8 lds_frag;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    const _Float16* x = (const _Float16*)p.x;
    const drc_tap_class cls = p.cls[0];
    const bool dense1 = p.reserved == 1;
    const int xh = (int)p.x_h_stride, xd = (int)p.x_d_stride;
    const int yh = (int)p.y_h_stride, yd_ = (int)p.y_d_stride, yc = (int)p.y_cb_stride;
    const int rh = (int)p.r_h_stride, rd_ = (int)p.r_d_stride, rc = (int)p.r_cb_stride;
    const int n_ct = (p.OW + T16_COLS - 1) / T16_COLS, n_rt = (p.OH + TR - 1) / TR;
    const unsigned columns = (unsigned)p.N * n_rt * n_ct;

    // ---- weights -> REGISTERS, once per wave: 27 x CT fragments (216 VGPRs at CT = 2; the kernel runs one wave per SIMD, 512 are
    // there).  Through LDS they doubled the ds_read_b128 traffic of a (kd, kw) group (12 KiB per wave for 24 MFMAs), and LDS -- not the
    // matrix cores -- bounded the step.
    f16x8 wreg[27][CT];
#pragma unroll
    for (int tap = 0; tap < 27; ++tap)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            wreg[tap][ct] = *(const f16x8*)((const _Float16*)p.w + ((long)tap * p.cout_pad + ct * 16 + j) * 32 + g * 8);
    const __attribute__((address_space(3))) char* ring = (const __attribute__((address_space(3))) char*)lds;
    const unsigned lane_b = (unsigned)(g * 256 + j * 16 + wave * RW * 1024);
    f32x4 sc_[CT], sh_[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        sc_[ct] = dense1 ? (f32x4){1.f, 1.f, 1.f, 1.f} : *(const f32x4*)(p.scale + ct * 16 + g * 4);
        sh_[ct] = dense1 ? (f32x4){0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(p.shift + ct * 16 + g * 4);
    }

    f32x4 acc[RW][CT];
    for (unsigned col = blockIdx.x; col < columns; col += gridDim.x) {
        unsigned t = col;
        unsigned u = t / (unsigned)n_ct; const int c0 = (int)(t - u * (unsigned)n_ct) * T16_COLS; t = u;
        u = t / (unsigned)n_rt; const int r0 = (int)(t - u * (unsigned)n_rt) * TR;
        const int n = (int)u;
        const _Float16* src0 = x + (long)n * p.x_n_stride + (cls.dd0 * xd + (r0 + cls.dh0) * xh + (c0 + cls.dw0 + j) * 32 + g * 8);
        auto stage = [&](int ps) __attribute__((always_inline)) {           // padded slice ps -> ring slot ps & 3
            char* dst = lds + (ps & 3) * SLOT;
#pragma unroll
            for (int i0 = 0; i0 < TR + 2; i0 += T16_WAVES) {
                const int rr = i0 + wave;
                if (rr < TR + 2) __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src0 + (ps * xd + rr * xh)), LDS_PTR(dst + rr * 1024), 16, 0, 0);
            }
        };
        // every wave is done with the previous column's slots before they are overwritten
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        stage(0); stage(1); stage(2);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        for (int od = 0; od < p.OD; ++od) {
            // slices od .. od+2 are resident and slot (od+3)&3 is free (the barrier that ended the previous step)
            if (od + 1 < p.OD) stage(od + 3);
            // this slice's residual tile is requested now and consumed after the 27 taps (one load -> wait -> store chain per output
            // made the epilogue the longest phase of a step)
            const int col_ = c0 + j;
            const bool col_ok = j < T16_COLS && col_ < p.OW;
            const int row0 = r0 + wave * RW;
            const _Float16* res = (p.res && !dense1) ? (const _Float16*)p.res + p.r_off0 + (long)n * p.r_n_stride : nullptr;
            const int rl = od * rd_ + row0 * rh + col_ * 32 + g * 4;
            f16x4 rv[RW][CT];
            float rdv[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                rdv[r] = 0.f;
                if (dense1 && p.res && g == 0 && col_ok && row0 + r < p.OH)
                    rdv[r] = ((const float*)p.res)[(((long)n * p.OD + od) * p.OH + row0 + r) * p.OW + col_];
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    rv[r][ct] = (f16x4){0, 0, 0, 0};
                    if (res && col_ok && row0 + r < p.OH) rv[r][ct] = *(const f16x4*)(res + rl + (ct >> 1) * rc + r * rh + (ct & 1) * 16);
                }
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[r][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // (kd, kw) groups, one ahead: the RW+2 rows of column shift kw in slice od+kd, and the 3 x CT weight fragments of (kd, *, kw)
            f16x8 rowA[RW + 2], rowB[RW + 2];
            auto gfetch = [&](f16x8 (&rows)[RW + 2], int grp) __attribute__((always_inline)) {
                const int kd = grp / 3, kw = grp - kd * 3;
                const __attribute__((address_space(3))) char* sb = ring + ((od + kd) & 3) * SLOT + lane_b + kw * 16;
#pragma unroll
                for (int rr = 0; rr < RW + 2; ++rr) rows[rr] = *(lds_frag*)(sb + rr * 1024);
            };
            auto gmfma = [&](const f16x8 (&rows)[RW + 2], int grp) __attribute__((always_inline)) {
                const int kd = grp / 3, kw = grp - kd * 3;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int r = 0; r < RW; ++r)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            acc[r][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[(kd * 3 + kh) * 3 + kw][ct], rows[r + kh], acc[r][ct], 0, 0, 0);
            };
            gfetch(rowA, 0);
#pragma unroll
            for (int grp = 0; grp < 9; ++grp) {
                if (grp + 1 < 9) { if (grp & 1) gfetch(rowA, grp + 1); else gfetch(rowB, grp + 1); }
                __builtin_amdgcn_sched_barrier(0);
                if (grp & 1) gmfma(rowB, grp); else gmfma(rowA, grp);
                __builtin_amdgcn_sched_barrier(0);
            }
            // the loads of this step (next slice's rows, residuals) are waited for HERE, before the stores are issued: the stores then
            // stay in flight through the barrier and the next step's MFMAs (vmcnt counts them too; waiting at the top of the next step
            // exposed their write latency once per output slice)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // ---- epilogue of output slice od
            if (dense1) {
                float* yd = (float*)p.y;
                if (g == 0 && col_ok) {
#pragma unroll
                    for (int r = 0; r < RW; ++r)
                        if (row0 + r < p.OH) yd[(((long)n * p.OD + od) * p.OH + row0 + r) * p.OW + col_] = acc[r][0].x + rdv[r];
                }
            } else {
                _Float16* y = (_Float16*)p.y + p.y_off0 + (long)n * p.y_n_stride;
                const int yl = od * yd_ + row0 * yh + col_ * 32 + g * 4;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                    for (int r = 0; r < RW; ++r) {
                        f32x4 v = acc[r][ct] * sc_[ct] + sh_[ct];
                        v.x += (float)rv[r][ct].x; v.y += (float)rv[r][ct].y; v.z += (float)rv[r][ct].z; v.w += (float)rv[r][ct].w;
                        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        f16x4 hv;
                        hv.x = (_Float16)v.x; hv.y = (_Float16)v.y; hv.z = (_Float16)v.z; hv.w = (_Float16)v.w;
                        if (col_ok && row0 + r < p.OH) *(f16x4*)(y + yl + (ct >> 1) * yc + r * yh + (ct & 1) * 16) = hv;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave's rows landed; everyone is done reading slice od-1's slot
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv16sp_kernel<RW,CT>: conv16s_kernel with the next-but-one slice in flight (round 4).  conv16s issues slice od+3's DMA at the top of
// step od and drains the queue (vmcnt(0)) at its end: one MFMA phase (1.5 us at RW = 4) to cover an HBM round trip under load, with one
// wave per SIMD and nothing else to run -- 8.7 K cycles per step against 3.5 K of MFMA.  Here
//   * the ring has FIVE slots and step od requests slice od+4; the wait at its end is COUNTED: everything up to slice od+3 and this
//     step's residual tile must have landed, while slice od+4's DMAs, the NEXT step's residual tile (requested one step ahead, second
//     register set) and the previous step's stores stay in flight.  vmcnt retires in order, so the count must be exact: every wave
//     issues the same number of DMAs per stage (the clamped duplicate rows of waves 2, 3 rewrite a row with the same bytes), slices
//     past the volume's end are still requested (clamped, into slots nobody reads), and the kernel only takes maps whose rows fill the
//     tiles (OH % TR == 0: no row-masked -- possibly skipped -- store or residual load);
//   * RW = 7 rows per wave on 28- and 56-row maps: 378 MFMAs per step and weight set instead of 216, no padded rows (16-row tiles padded
//     56 rows to 64).
// MODE: 0 = blocked fp16 output, no residual; 1 = ... with a blocked fp16 residual; 2 = the dense fp32 single-cout head (reserved == 1),
// no residual; 3 = ... with a dense fp32 residual.  (Template argument, not a run-time branch: with branches around the residual requests
// the compiler's wait-count pass gave up on counting and drained the queue right behind the DMAs.)
// s_waitcnt as the BUILTIN (gfx9 encoding: vmcnt[3:0] | expcnt << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14): the compiler's own wait-count pass
// reads it.  Behind inline-asm waits it still believed the kernel's first loads (the weights) pending at the head of the od loop and
// drained the queue before every step's first MFMA.
#define T16_WAITCNT(vm, lgkm) (((vm) & 15) | (7 << 4) | ((lgkm) << 8) | (((vm) >> 4) << 14))
#define T16_WAIT_BARRIER(imm) do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_waitcnt(imm); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

template <int RW, int CT, int MODE>
__global__ __launch_bounds__(64 * T16_WAVES) void conv16sp_kernel(const drc_tapconv_params p) {
    constexpr int TR = RW * T16_WAVES;
    constexpr int NR = TR + 2;                         // staged rows per slice
    constexpr int SLOT = NR * 1024;
    constexpr int SLOTS = 5;
    constexpr int NDMA = (NR + T16_WAVES - 1) / T16_WAVES;                // per wave and stage, the same for every wave
    constexpr bool DENSE = MODE >= 2, RES = MODE & 1;
    constexpr int NST = DENSE ? RW : RW * CT;                             // stores per step = residual loads per step (when RES)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    typedef const __attribute__((address_space(3))) volatile f16x8 lds_frag;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    const _Float16* x = (const _Float16*)p.x;
    const drc_tap_class cls = p.cls[0];
    const int xh = (int)p.x_h_stride, xd = (int)p.x_d_stride;
    const int yh = (int)p.y_h_stride, yd_ = (int)p.y_d_stride, yc = (int)p.y_cb_stride;
    const int rh = (int)p.r_h_stride, rd_ = (int)p.r_d_stride, rc = (int)p.r_cb_stride;
    const int n_ct = (p.OW + T16_COLS - 1) / T16_COLS, n_rt = p.OH / TR;
    const unsigned columns = (unsigned)p.N * n_rt * n_ct;

    f16x8 wreg[27][CT];
#pragma unroll
    for (int tap = 0; tap < 27; ++tap)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            wreg[tap][ct] = *(const f16x8*)((const _Float16*)p.w + ((long)tap * p.cout_pad + ct * 16 + j) * 32 + g * 8);
    const __attribute__((address_space(3))) char* ring = (const __attribute__((address_space(3))) char*)lds;
    const unsigned lane_b = (unsigned)(g * 256 + j * 16 + wave * RW * 1024);
    f32x4 sc_[CT], sh_[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        sc_[ct] = DENSE ? (f32x4){1.f, 1.f, 1.f, 1.f} : *(const f32x4*)(p.scale + ct * 16 + g * 4);
        sh_[ct] = DENSE ? (f32x4){0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(p.shift + ct * 16 + g * 4);
    }

    f32x4 acc[RW][CT];
    for (unsigned col = blockIdx.x; col < columns; col += gridDim.x) {
        unsigned t = col;
        unsigned u = t / (unsigned)n_ct; const int c0 = (int)(t - u * (unsigned)n_ct) * T16_COLS; t = u;
        u = t / (unsigned)n_rt; const int r0 = (int)(t - u * (unsigned)n_rt) * TR;
        const int n = (int)u;
        const _Float16* src0 = x + (long)n * p.x_n_stride + (cls.dd0 * xd + (r0 + cls.dh0) * xh + (c0 + cls.dw0 + j) * 32 + g * 8);
        auto stage = [&](int ps, int slot) __attribute__((always_inline)) {            // padded slice ps (clamped to the last one) -> ring slot
            const int psc = ps < p.OD + 2 ? ps : p.OD + 1;
            char* dst = lds + slot * SLOT;
#pragma unroll
            for (int k = 0; k < NDMA; ++k) {
                int rr = k * T16_WAVES + wave;
                rr = rr < NR ? rr : NR - 1;
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src0 + (psc * xd + rr * xh)), LDS_PTR(dst + rr * 1024), 16, 0, 0);
            }
        };
        const int col_ = c0 + j;
        const bool col_ok = j < T16_COLS && col_ < p.OW;
        const int col_c = col_ < p.OW ? col_ : p.OW - 1;                                // (residual requests of the unused lanes stay inside the map)
        const int row0 = r0 + wave * RW;
        const _Float16* res = (const _Float16*)p.res + p.r_off0 + (long)n * p.r_n_stride;
        const float* resd = (const float*)p.res;
        // this unit's output as a buffer: [base, base + 2 GiB) (the launcher checks the per-unit sizes)
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
            DENSE ? (void*)((float*)p.y + (long)n * p.OD * p.OH * p.OW) : (void*)((_Float16*)p.y + p.y_off0 + (long)n * p.y_n_stride), 0, 0x7FFFFF00, 0x00020000);
        f16x4 rv[2][RW][CT];                           // residual tiles of this step and the next (static rotation: the od loop runs in pairs)
        float rdv[2][RW];
        auto res_request = [&](int od, int set) __attribute__((always_inline)) {        // always NST loads (RES), whatever od
            if constexpr (RES) {
                const int odc = od < p.OD ? od : p.OD - 1;
                const int rl = odc * rd_ + row0 * rh + col_c * 32 + g * 4;
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    if constexpr (DENSE) {
                        rdv[set][r] = resd[(((long)n * p.OD + odc) * p.OH + row0 + r) * p.OW + col_c];
                    } else {
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) rv[set][r][ct] = *(const f16x4*)(res + rl + (ct >> 1) * rc + r * rh + (ct & 1) * 16);
                    }
                }
            }
        };
        // every wave is done with the previous column's slots before they are overwritten
        T16_WAIT_BARRIER(T16_WAITCNT(63, 0));
        stage(0, 0); stage(1, 1); stage(2, 2); stage(3, 3);
        res_request(0, 0);
        T16_WAIT_BARRIER(T16_WAITCNT(0, 15));
        int s0 = 0, s4 = 4;                            // ring slots of padded slices od and od + 4
        auto step = [&](int od, auto SET) __attribute__((always_inline)) {
            constexpr int set = decltype(SET)::value;
            // slices od .. od+3 are resident or landing; slot s4 is free (slice od-1: the barrier that ended the previous step)
            stage(od + 4, s4);
            res_request(od + 1, set ^ 1);
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[r][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int s1 = s0 + 1 < SLOTS ? s0 + 1 : s0 + 1 - SLOTS, s2 = s0 + 2 < SLOTS ? s0 + 2 : s0 + 2 - SLOTS;
            const __attribute__((address_space(3))) char* sl[3] = {ring + s0 * SLOT + lane_b, ring + s1 * SLOT + lane_b, ring + s2 * SLOT + lane_b};
            f16x8 rowA[RW + 2], rowB[RW + 2];
            auto gfetch = [&](f16x8 (&rows)[RW + 2], int grp) __attribute__((always_inline)) {
                const int kd = grp / 3, kw = grp - kd * 3;
#pragma unroll
                for (int rr = 0; rr < RW + 2; ++rr) rows[rr] = *(lds_frag*)(sl[kd] + kw * 16 + rr * 1024);
            };
            auto gmfma = [&](const f16x8 (&rows)[RW + 2], int grp) __attribute__((always_inline)) {
                const int kd = grp / 3, kw = grp - kd * 3;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int r = 0; r < RW; ++r)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            acc[r][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[(kd * 3 + kh) * 3 + kw][ct], rows[r + kh], acc[r][ct], 0, 0, 0);
            };
            gfetch(rowA, 0);
#pragma unroll
            for (int grp = 0; grp < 9; ++grp) {
                if (grp + 1 < 9) { if (grp & 1) gfetch(rowA, grp + 1); else gfetch(rowB, grp + 1); }
                __builtin_amdgcn_sched_barrier(0);
                if (grp & 1) gmfma(rowB, grp); else gmfma(rowA, grp);
                __builtin_amdgcn_sched_barrier(0);
            }
            // in flight past this point: slice od+4 (NDMA), the next step's residual tile (NST if RES), the previous step's stores (NST)
            // -- all younger than slice od+3 and this step's residual tile.  (At od = 0 there are no stores yet and the prologue has
            // drained everything older: the wait is then simply weaker than it may be.)
            __builtin_amdgcn_s_waitcnt(T16_WAITCNT(NDMA + (RES ? 2 : 1) * NST, 15));
            __builtin_amdgcn_sched_barrier(0);         // (the epilogue's register reads must not be scheduled above the wait)
            static_assert(NDMA + 2 * NST <= 63, "vmcnt is a 6-bit counter");
            // ---- epilogue of output slice od.  Buffer stores, the lanes without an output (j = 14, 15, columns past the map, g != 0 of the
            // dense head) pointed past the descriptor's range: the hardware drops them, the instruction count stays the same on every path
            // (behind exec-masked branches the compiler no longer trusted its own count and drained the queue before the epilogue).
            if constexpr (DENSE) {
                const unsigned yo = (g == 0 && col_ok) ? (unsigned)(((od * p.OH + row0) * p.OW + col_) * 4) : 0x80000000u;   // (+ the per-store offsets below: still past num_records, no 32-bit wrap)
#pragma unroll
                for (int r = 0; r < RW; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[r][0].x + (RES ? rdv[set][r] : 0.f)), yr, yo + (unsigned)(r * p.OW * 4), 0, 0);
            } else {
                const unsigned yo = col_ok ? (unsigned)((od * yd_ + row0 * yh + col_ * 32 + g * 4) * 2) : 0x80000000u;   // (+ the per-store offsets below: still past num_records, no 32-bit wrap)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                    for (int r = 0; r < RW; ++r) {
                        f32x4 v = acc[r][ct] * sc_[ct] + sh_[ct];
                        if constexpr (RES) {
                            v.x += (float)rv[set][r][ct].x; v.y += (float)rv[set][r][ct].y; v.z += (float)rv[set][r][ct].z; v.w += (float)rv[set][r][ct].w;
                        }
                        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        f16x4 hv;
                        hv.x = (_Float16)v.x; hv.y = (_Float16)v.y; hv.z = (_Float16)v.z; hv.w = (_Float16)v.w;
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hv), yr, yo + (unsigned)(((ct >> 1) * yc + r * yh + (ct & 1) * 16) * 2), 0, 0);
                    }
                }
            }
            T16_WAIT_BARRIER(T16_WAITCNT(63, 0));     // every wave's rows of slice od+3 landed; everyone is done reading slice od's slot
            s0 = s1;
            s4 = s4 + 1 < SLOTS ? s4 + 1 : 0;
        };
        for (int od = 0; od < p.OD; od += 2) {
            step(od, std::integral_constant<int, 0>{});
            if (od + 1 < p.OD) step(od + 1, std::integral_constant<int, 1>{});
        }
    }
}

template <int RW, int CT, int MODE>
int launch_slide_deep_mode(const drc_tapconv_params& p, hipStream_t stream) {
    constexpr int TR = RW * T16_WAVES;
    constexpr size_t lds = 5 * (size_t)(TR + 2) * 1024 + 1024;
    static_assert(lds <= 160 * 1024, "ring exceeds the LDS");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv16sp_kernel<RW, CT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const long columns = (long)p.N * (p.OH / TR) * ((p.OW + T16_COLS - 1) / T16_COLS);
    if (columns >= (1L << 31)) return -5;
    long blocks = 256;                                   // one block per CU: the weights take the wave's register file
    if (blocks > columns) blocks = columns;
    hipLaunchKernelGGL((conv16sp_kernel<RW, CT, MODE>), dim3((unsigned)blocks), dim3(64 * T16_WAVES), lds, stream, p);
    return (int)hipGetLastError();
}

template <int RW>
int launch_slide_deep(const drc_tapconv_params& p, hipStream_t stream) {
    if (p.reserved == 1) return p.res ? launch_slide_deep_mode<RW, 1, 3>(p, stream) : launch_slide_deep_mode<RW, 1, 2>(p, stream);
    if (p.cout_pad == 32) return p.res ? launch_slide_deep_mode<RW, 2, 1>(p, stream) : launch_slide_deep_mode<RW, 2, 0>(p, stream);
    return p.res ? launch_slide_deep_mode<RW, 1, 1>(p, stream) : launch_slide_deep_mode<RW, 1, 0>(p, stream);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv16sw_kernel<RW,CW,SLOTS,RES,CV>: the depth walk for TWO input blocks (64 input channels: dres0[0] on the cost volume, hourglass
// conv2 / conv4), one cout tile per wave (round 4).  conv16x.hip's conv16d_kernel re-stages three depth slices per output slice in six
// (channel block, depth tap) stages -- six barriers and 63 MFMAs per stage and wave, bound by the DMA round trips (584 us for dres0[0],
// MFMA floor 243).  Here, as in conv16sp_kernel: the block's four waves are CW cout tiles x RG = 4/CW row groups, a wave holds the 54
// weight fragments of ITS cout tile in registers (216 VGPRs, one wave per SIMD), the block owns a (n, row tile, column tile) column and
// walks od with a ring of SLOTS slices ([cb 2][TR + 2 rows][1 KiB] each): one new slice and one barrier per output slice, 378 MFMAs
// per step and wave at RW = 7, no vector-memory instruction in the MFMA phase.  SLOTS = 5: slice od+4 is requested at step od and the
// wait at its end is counted (see conv16sp_kernel); SLOTS = 4 (the 14-row tiles of dres0[0]: 32 KiB per slice): slice od+3, the wait
// leaves only the next residual tile in flight -- enough for CV, whose source rows are L2-resident.
// CV: p.x is the feature pair (see conv16x.hip's conv16d_kernel): slice d of the cost volume = the left rows where the shifted pixel
// exists | the right rows moved by lo4 + d columns; lanes whose voxel is zero in the volume fetch column 0 of the row (the zero halo).
// Rows must fill the tiles (OH % TR == 0), as for every counted-wait kernel.
template <int RW, int CW, int SLOTS, bool RES, bool CV>
__global__ __launch_bounds__(64 * T16_WAVES) void conv16sw_kernel(const drc_tapconv_params p, const int lo4) {
    constexpr int RG = T16_WAVES / CW;
    constexpr int TR = RW * RG;
    constexpr int NR = TR + 2;                         // staged rows per slice and channel block
    constexpr int SLOT = 2 * NR * 1024;
    constexpr int AHEAD = SLOTS - 3;                   // slices in flight beyond the three a step reads
    constexpr int NDMA = (2 * NR + T16_WAVES - 1) / T16_WAVES;
    constexpr int NST = RW;                            // stores per step = residual loads per step (RES)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    typedef const __attribute__((address_space(3))) volatile f16x8 lds_frag;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cw = wave % CW, rg = wave / CW;
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    const _Float16* x = (const _Float16*)p.x;
    const drc_tap_class cls = p.cls[0];
    const int xh = (int)p.x_h_stride, xd = (int)p.x_d_stride, xc = (int)p.x_cb_stride;
    const int yh = (int)p.y_h_stride, yd_ = (int)p.y_d_stride, yc = (int)p.y_cb_stride;
    const int rh = (int)p.r_h_stride, rd_ = (int)p.r_d_stride, rc = (int)p.r_cb_stride;
    const int n_ct = (p.OW + T16_COLS - 1) / T16_COLS, n_rt = p.OH / TR, n_cg = p.cout_pad / 16 / CW;
    const unsigned columns = (unsigned)p.N * n_rt * n_ct * n_cg;
    const int Wp = xh / 32;
    const unsigned lane_b = (unsigned)(g * 256 + j * 16 + rg * RW * 1024);
    const __attribute__((address_space(3))) char* ring = (const __attribute__((address_space(3))) char*)lds;

    f32x4 acc[RW];
    for (unsigned col = blockIdx.x; col < columns; col += gridDim.x) {
        unsigned t = col, u;
        u = t / (unsigned)n_cg; const int cg = (int)(t - u * (unsigned)n_cg); t = u;       // cout group fastest: neighbours share the input column in L2
        u = t / (unsigned)n_ct; const int c0 = (int)(t - u * (unsigned)n_ct) * T16_COLS; t = u;
        u = t / (unsigned)n_rt; const int r0 = (int)(t - u * (unsigned)n_rt) * TR;
        const int n = (int)u;
        const int cot = cg * CW + cw;
        // this wave's cout tile: all 2 x 27 weight fragments -> registers (once per column; the columns of a block mostly share cg)
        f16x8 wreg[2][27];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int tap = 0; tap < 27; ++tap)
                wreg[cb][tap] = *(const f16x8*)((const _Float16*)p.w + (((long)tap * 2 + cb) * p.cout_pad + cot * 16 + j) * 32 + g * 8);
        const f32x4 sc_ = *(const f32x4*)(p.scale + cot * 16 + g * 4), sh_ = *(const f32x4*)(p.shift + cot * 16 + g * 4);
        int colE = c0 + cls.dw0 + j;
        colE = colE < Wp ? colE : Wp - 1;
        const _Float16* src0 = x + (long)n * p.x_n_stride + ((r0 + cls.dh0) * xh + g * 8);
        auto stage = [&](int ps, int slot) __attribute__((always_inline)) {            // padded slice ps (clamped to the last one) -> ring slot
            const int psc = ps < p.OD + 2 ? ps : p.OD + 1;
            char* dst = lds + slot * SLOT;
            int col0 = colE, col1 = colE;              // source column of this lane in channel block 0 / 1
            if constexpr (CV) {
                const int d = psc + cls.dd0 - 1, sft = lo4 + d, xw = colE - 1, xs = xw - sft, Wr = Wp - 2;
                const bool ok = d >= 0 && d < p.OD && xw >= 0 && xw < Wr && xs >= 0 && xs < Wr;
                col0 = ok ? colE : 0;
                col1 = ok ? xs + 1 : 0;
            }
            const _Float16* sd = src0 + (CV ? 1 : psc + cls.dd0) * xd;
#pragma unroll
            for (int k = 0; k < NDMA; ++k) {
                int i = k * T16_WAVES + wave;
                i = i < 2 * NR ? i : 2 * NR - 1;
                const int cb = i >= NR ? 1 : 0, rr = i - cb * NR;
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(sd + (cb * xc + rr * xh + (cb ? col1 : col0) * 32)), LDS_PTR(dst + i * 1024), 16, 0, 0);
            }
        };
        const int col_ = c0 + j;
        const bool col_ok = j < T16_COLS && col_ < p.OW;
        const int col_c = col_ < p.OW ? col_ : p.OW - 1;
        const int row0 = r0 + rg * RW;
        const _Float16* res = (const _Float16*)p.res + p.r_off0 + (long)n * p.r_n_stride + ((cot >> 1) * rc + (cot & 1) * 16 + g * 4);
        const __amdgpu_buffer_rsrc_t yr =
            __builtin_amdgcn_make_buffer_rsrc((void*)((_Float16*)p.y + p.y_off0 + (long)n * p.y_n_stride), 0, 0x7FFFFF00, 0x00020000);
        f16x4 rv[2][RW];
        auto res_request = [&](int od, int set) __attribute__((always_inline)) {        // always NST loads (RES), whatever od
            if constexpr (RES) {
                const int odc = od < p.OD ? od : p.OD - 1;
                const int rl = odc * rd_ + row0 * rh + col_c * 32;
#pragma unroll
                for (int r = 0; r < RW; ++r) rv[set][r] = *(const f16x4*)(res + rl + r * rh);
            }
        };
        // every wave is done with the previous column's slots before they are overwritten
        T16_WAIT_BARRIER(T16_WAITCNT(63, 0));
#pragma unroll
        for (int s_ = 0; s_ < SLOTS - 1; ++s_) stage(s_, s_);
        res_request(0, 0);
        T16_WAIT_BARRIER(T16_WAITCNT(0, 15));
        int s0 = 0, sn = SLOTS - 1;                    // ring slots of padded slice od and of the slice requested at step od
        auto step = [&](int od, auto SET) __attribute__((always_inline)) {
            constexpr int set = decltype(SET)::value;
            stage(od + SLOTS - 1, sn);
            res_request(od + 1, set ^ 1);
#pragma unroll
            for (int r = 0; r < RW; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int s1 = s0 + 1 < SLOTS ? s0 + 1 : s0 + 1 - SLOTS, s2 = s0 + 2 < SLOTS ? s0 + 2 : s0 + 2 - SLOTS;
            const __attribute__((address_space(3))) char* sl[3] = {ring + s0 * SLOT + lane_b, ring + s1 * SLOT + lane_b, ring + s2 * SLOT + lane_b};
            f16x8 rowA[RW + 2], rowB[RW + 2];
            // group = (channel block, depth tap, column tap): RW + 2 rows read once, 3 RW MFMAs
            auto gfetch = [&](f16x8 (&rows)[RW + 2], int grp) __attribute__((always_inline)) {
                const int cb = grp / 9, kd = (grp / 3) % 3, kw = grp % 3;
#pragma unroll
                for (int rr = 0; rr < RW + 2; ++rr) rows[rr] = *(lds_frag*)(sl[kd] + cb * NR * 1024 + kw * 16 + rr * 1024);
            };
            auto gmfma = [&](const f16x8 (&rows)[RW + 2], int grp) __attribute__((always_inline)) {
                const int cb = grp / 9, kd = (grp / 3) % 3, kw = grp % 3;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int r = 0; r < RW; ++r)
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[cb][(kd * 3 + kh) * 3 + kw], rows[r + kh], acc[r], 0, 0, 0);
            };
            gfetch(rowA, 0);
#pragma unroll
            for (int grp = 0; grp < 18; ++grp) {
                if (grp + 1 < 18) { if (grp & 1) gfetch(rowA, grp + 1); else gfetch(rowB, grp + 1); }
                __builtin_amdgcn_sched_barrier(0);
                if (grp & 1) gmfma(rowB, grp); else gmfma(rowA, grp);
                __builtin_amdgcn_sched_barrier(0);
            }
            // younger than the slice the next step needs and this step's residual tile: AHEAD - 1 stage requests (only the one of this
            // step when AHEAD = 2), the next step's residual tile (RES), the previous step's stores when the requests of this step are
            // allowed to stay (AHEAD = 2)
            __builtin_amdgcn_s_waitcnt(T16_WAITCNT(AHEAD == 2 ? NDMA + (RES ? 2 : 1) * NST : (RES ? NST : 0), 15));
            __builtin_amdgcn_sched_barrier(0);
            const unsigned yo = col_ok ? (unsigned)((od * yd_ + row0 * yh + col_ * 32 + g * 4 + (cot >> 1) * yc + (cot & 1) * 16) * 2) : 0x80000000u;
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                f32x4 v = acc[r] * sc_ + sh_;
                if constexpr (RES) { v.x += (float)rv[set][r].x; v.y += (float)rv[set][r].y; v.z += (float)rv[set][r].z; v.w += (float)rv[set][r].w; }
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                f16x4 hv;
                hv.x = (_Float16)v.x; hv.y = (_Float16)v.y; hv.z = (_Float16)v.z; hv.w = (_Float16)v.w;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hv), yr, yo + (unsigned)(r * yh * 2), 0, 0);
            }
            T16_WAIT_BARRIER(T16_WAITCNT(63, 0));      // every wave's share of the awaited slice landed; everyone is done reading slice od's slot
            s0 = s1;
            sn = sn + 1 < SLOTS ? sn + 1 : 0;
        };
        for (int od = 0; od < p.OD; od += 2) {
            step(od, std::integral_constant<int, 0>{});
            if (od + 1 < p.OD) step(od + 1, std::integral_constant<int, 1>{});
        }
    }
}

template <int RW, int CW, bool RES, bool CV>
int launch_walk2_mode(const drc_tapconv_params& p, hipStream_t stream, int lo4) {
    constexpr int TR = RW * (T16_WAVES / CW);
    constexpr int SLOTS = 5 * 2 * (TR + 2) * 1024 + 1024 <= 160 * 1024 ? 5 : 4;
    constexpr size_t lds = (size_t)SLOTS * 2 * (TR + 2) * 1024 + 1024;
    static_assert(lds <= 160 * 1024, "ring exceeds the LDS");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv16sw_kernel<RW, CW, SLOTS, RES, CV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const long columns = (long)p.N * (p.OH / TR) * ((p.OW + T16_COLS - 1) / T16_COLS) * (p.cout_pad / 16 / CW);
    if (columns >= (1L << 31)) return -5;
    long blocks = 256;                                   // one block per CU: the weights take the wave's register file
    if (blocks > columns) blocks = columns;
    hipLaunchKernelGGL((conv16sw_kernel<RW, CW, SLOTS, RES, CV>), dim3((unsigned)blocks), dim3(64 * T16_WAVES), lds, stream, p, lo4);
    return (int)hipGetLastError();
}

}  // namespace

// The two-block depth walk for a stride-1 3x3x3 layer (cv: on the feature pair, see drc_conv16_k3_costvol_fwd).  Returns 1 when the
// shape is not one it takes (the caller keeps conv16x.hip's staged kernel), else 0 / a hipError_t.  Callers have validated p.
int drc_t16_conv3d_walk2_try(const drc_tapconv_params& p, hipStream_t s, bool cv, int lo4) {
#ifndef T16_WALK2
#define T16_WALK2 1
#endif
    if (!T16_WALK2 || p.cb_in != 2 || p.cls[0].nd != 3 || p.reserved == 1 || p.OD < 4) return 1;
    if (cv && p.res) return 1;
    const int ct = p.cout_pad / 16;
    const bool res = p.res != nullptr;
    if (ct % 4 == 0) {                                   // four cout tiles side by side, seven rows
        if (p.OH % 7) return 1;
        if (cv) return launch_walk2_mode<7, 4, false, true>(p, s, lo4);
        return res ? launch_walk2_mode<7, 4, true, false>(p, s, 0) : launch_walk2_mode<7, 4, false, false>(p, s, 0);
    }
    if (ct % 2 == 0) {                                   // two cout tiles x two row groups: 14-row tiles
        if (p.OH % 14) return 1;
        if (cv) return launch_walk2_mode<7, 2, false, true>(p, s, lo4);
        return res ? launch_walk2_mode<7, 2, true, false>(p, s, 0) : launch_walk2_mode<7, 2, false, false>(p, s, 0);
    }
    return 1;
}

namespace {

template <int RW, int CT>
int launch_slide(const drc_tapconv_params& p, hipStream_t stream) {
    constexpr int TR = RW * T16_WAVES;
    constexpr size_t lds = 4 * (size_t)(TR + 2) * 1024 + 1024;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv16s_kernel<RW, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const long columns = (long)p.N * ((p.OH + TR - 1) / TR) * ((p.OW + T16_COLS - 1) / T16_COLS);
    if (columns >= (1L << 31)) return -5;
    long blocks = 256;                                   // one block per CU: the weights take the wave's register file
    if (blocks > columns) blocks = columns;
    hipLaunchKernelGGL((conv16s_kernel<RW, CT>), dim3((unsigned)blocks), dim3(64 * T16_WAVES), lds, stream, p);
    return (int)hipGetLastError();
}

template <int RW, int CT, int ND>
int launch(const drc_tapconv_params& p, hipStream_t stream) {
    constexpr int TR = RW * T16_WAVES;
    constexpr size_t lds = (size_t)(ND * (TR + 2) * 1024 + 1024);            // + one row of slack for the unused lanes' over-read
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv16t_kernel<RW, CT, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const long tiles = (long)p.N * p.OD * ((p.OH + TR - 1) / TR) * ((p.OW + T16_COLS - 1) / T16_COLS) * (p.cout_pad / 16 / CT);
    if (tiles >= (1L << 31)) return -5;
    long per_cu = (160 * 1024) / (long)lds;                           // resident blocks per CU (LDS); at most 4 (16 waves)
    if (per_cu > 4) per_cu = 4;
    long blocks = 256 * per_cu;
    if (blocks > tiles) blocks = tiles;
    hipLaunchKernelGGL((conv16t_kernel<RW, CT, ND>), dim3((unsigned)blocks), dim3(64 * T16_WAVES), lds, stream, p);
    return (int)hipGetLastError();
}

template <int ND>
int pick(const drc_tapconv_params& p, hipStream_t s) {
    const int ct = p.cout_pad / 16;
    // 4 rows per wave (16-row tiles) when the map has them; 2 otherwise (less padding on 8 / 24-row maps)
    const bool tall = p.OH % 16 == 0 || p.OH >= 48;
    if (ct % 4 == 0) return tall ? launch<4, 4, ND>(p, s) : launch<2, 4, ND>(p, s);
    if (ct % 2 == 0) return tall ? launch<4, 2, ND>(p, s) : launch<2, 2, ND>(p, s);
    return tall ? launch<4, 1, ND>(p, s) : launch<2, 1, ND>(p, s);
}

}  // namespace

#ifndef T16_X3D
#define T16_X3D 1
#endif
int drc_x16_conv3d_s1_launch(const drc_tapconv_params& p, hipStream_t s);       // conv16x.hip

extern "C" int drc_conv16_k3_tile_supported(const drc_tapconv_params* pp) {
    if (!pp) return 0;
    const drc_tapconv_params& p = *pp;
    const drc_tap_class& k = p.cls[0];
    if (p.n_classes != 1 || p.in_mul != 1 || p.out_mul != 1) return 0;
    if ((k.nd != 1 && k.nd != 3) || k.nh != 3 || k.nw != 3 || k.sd != 1 || k.sh != 1 || k.sw != 1) return 0;
    if (k.out_off_d || k.out_off_h || k.out_off_w) return 0;
    if (k.wbase != 0 || k.wsw != 1 || k.wsh != 3 || (k.nd == 3 && k.wsd != 9)) return 0;          // weights in tap order
    if (p.cout_pad <= 0 || (p.cout_pad & 15) || p.cb_in <= 0) return 0;
    return 1;
}

extern "C" int drc_conv16_k3_tile_fwd(const drc_tapconv_params* pp, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y) return -1;
    if (p.reserved != 1 && (!p.scale || !p.shift)) return -1;
    if (p.N < 0 || p.OD <= 0 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (!drc_conv16_k3_tile_supported(pp)) return -4;
    if ((int64_t)p.cb_in * p.cout_pad * 32 * 27 * 2 >= (1LL << 31)) return -5;          // 32-bit byte offsets inside the weights
    if (p.x_n_stride * 2 >= (1LL << 31) || (p.reserved != 1 && p.y_n_stride * 2 >= (1LL << 31)) ||
        (p.res && p.reserved != 1 && p.r_n_stride * 2 >= (1LL << 31)))
        return -5;                                                                       // ... and inside one unit of x / y / res
    hipStream_t s = (hipStream_t)stream;
#ifndef T16_SLIDE
#define T16_SLIDE 1
#endif
    if (T16_SLIDE && p.cls[0].nd == 3 && p.cb_in == 1 && p.cout_pad <= 32 && p.OD >= 4) {      // one input block, <= 32 couts: depth-sliding walk
#ifndef T16_DEEP
#define T16_DEEP 1
#endif
        if (T16_DEEP && p.OH % 28 == 0) return launch_slide_deep<7>(p, s);        // full 28-row tiles (the regressor's 28- and 56-row maps)
        if (T16_DEEP && p.OH % 16 == 0) return launch_slide_deep<4>(p, s);
        const bool tall = p.OH % 16 == 0 || p.OH >= 48;
        if (p.cout_pad == 32) return tall ? launch_slide<4, 2>(p, s) : launch_slide<2, 2>(p, s);
        return tall ? launch_slide<4, 1>(p, s) : launch_slide<2, 1>(p, s);
    }
    if (T16_X3D && p.cls[0].nd == 3 && p.reserved != 1) {
        const int st = drc_t16_conv3d_walk2_try(p, s, false, 0);                                   // two input blocks, full row tiles: the depth walk (round 4)
        if (st != 1) return st;
        return drc_x16_conv3d_s1_launch(p, s);      // conv16x.hip: cout-split waves, double-buffered stages (round 4)
    }      // conv16x.hip: cout-split waves, double-buffered stages (round 4)
    return p.cls[0].nd == 3 ? pick<3>(p, s) : pick<1>(p, s);
}
