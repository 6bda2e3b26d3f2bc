// tapslide.hip -- stride-1 3x3x3 convolution (+BN, +residual, +ReLU) with a SLIDING DEPTH WINDOW (gfx950 / CDNA4).
//
// Same arithmetic, layouts and MFMA mapping as tapconv.hip (v_mfma_f32_16x16x4_f32, blocked zero-haloed tensors,
// LDS-DMA staged [rows][voxels][8 ch] tiles), specialised for the layers that carry ~70 % of the regressor's FLOPs:
//   dres0/dres1, classifN[0], hourglass conv2/conv4            reference: stackhourglass.py:63-88, :14-20
//
// A wave owns a COLUMN: R output rows x WT columns of ALL depth slices of one ROI.  It walks the input slices once;
// every staged tile is applied to the three output slices it touches (depth taps dd = 0,1,2 -> od = d_in+1-dd), which
// keeps three accumulator sets live and rotates them without runtime indexing (the slice loop is unrolled by 3).
// Versus one-output-slice-per-wave this stages 3x fewer bytes per MFMA and reads each B fragment from LDS once for
// three depth taps; the all-zero halo slices are never staged or multiplied.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

#define SLIDE_WAVES 4

namespace {

template <int VT, int CT>
__global__ __launch_bounds__(64 * SLIDE_WAVES) void tapslide_kernel(const drc_tapconv_params p) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const int g = lane >> 4;

    const drc_tap_class cls = p.cls[0];
    const int n_wt = (p.OW + p.WT - 1) / p.WT;
    const int n_rt = (p.OH + p.R - 1) / p.R;
    const int cols = p.N * n_rt * n_wt;            // columns = (n, row tile, col tile) of one cout group
    const int D = p.OD;
    // Work units are output slices, linearised as (cout group, column, od).  Every wave takes an equal contiguous share
    // [ucur, u1) of them, so the SIMDs finish together whatever N is; a share that crosses a column boundary is walked as
    // two depth segments (each re-stages its neighbouring input slices).
    const long units = (long)(p.cout_pad / 16 / CT) * cols * D;
    const long workers = (long)gridDim.x * SLIDE_WAVES;
    const long wid = (long)blockIdx.x * SLIDE_WAVES + wave;
    long ucur = units * wid / workers;
    const long u1 = units * (wid + 1) / workers;
    const int rows_in = p.R + 2;
    const int seg_vox = p.WT + 2;
    const int seg_units = seg_vox * 2;             // 16-byte units per tile row
    const int ppr = (seg_units + 63) >> 6;         // LDS-DMA pieces (64 lanes x 16 B) per tile row
    const int seg_floats = ppr * 256;              // LDS row stride: whole pieces, the tail of the last one is padding
    const int pieces = rows_in * ppr;
    const int buf_floats = rows_in * seg_floats;
    float* lds = lds_all + wave * (2 * buf_floats);
    const int nslots = p.R * p.WT;

    int lane_off[VT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
        const int s = vt * 16 + j;
        int r = s / p.WT, c = s - r * p.WT;
        if (s >= nslots) { r = 0; c = 0; }
        lane_off[vt] = r * seg_floats + c * 8 + g * 2;
    }

    const int n_pc = p.cb_in * 2;                  // (channel block, half) phases per input slice

    // per-segment state (set at the top of the segment loop)
    const float* xcol;     // input origin of the column at padded depth index 0 (row oh0+dh0, col ow0+dw0)
    unsigned wlane_off;    // per-lane byte offset into the packed weights [widx = (kd*3+kh)*3+kw][cb*2+half][cout_pad][8]
    int n, oh0, ow0, ct0, od_lo, od_hi, din_lo, din_hi;

    // LDS-DMA of piece q of the tile (input slice d_in, phase pc) into tile buffer bufi.  Always exactly one instruction
    // with all 64 lanes active (tail lanes re-read the row's last unit into the row padding): the count of pieces issued
    // after a step's weight loads is static, so the step that uses those weights waits with vmcnt(#pieces) and a piece
    // has two tap steps to land.  Address = wave-uniform row origin (SALU) + per-lane byte offset (two precomputed VGPRs).
    const unsigned lane_src = (unsigned)(((lane >> 1) * 16 + (lane & 1) * 4) * 4);          // lane's 16-byte unit inside a full piece
    unsigned lane_src_last;                                                                 // same for the last (clamped) piece of a row
    {
        int u = (ppr - 1) * 64 + lane;
        u = u < seg_units ? u : seg_units - 1;
        lane_src_last = (unsigned)((((u >> 1) * 16 + (u & 1) * 4) - (ppr - 1) * 512) * 4);
    }
    const unsigned ppr_magic = (65536u + ppr - 1) / ppr;                                    // q / ppr for q < 64
    auto stage_piece = [&](int d_in, int pc, int bufi, int q) __attribute__((always_inline)) {
        const int r = (int)(((unsigned)q * ppr_magic) >> 16), part = q - r * ppr;
        const char* sb = (const char*)(xcol + (int64_t)(pc >> 1) * p.x_cb_stride + (int64_t)(d_in + cls.dd0 + 1) * p.x_d_stride + (pc & 1) * 8 +   // real slice d_in sits at padded depth d_in + dd0 + 1
                                       (int64_t)r * p.x_h_stride + part * 512);
        const unsigned vo = part == ppr - 1 ? lane_src_last : lane_src;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(sb + vo), LDS_PTR(lds + bufi * buf_floats + r * seg_floats + part * 256), 16, 0, 0);
    };
    const bool two_pieces = pieces > 9;            // 9 tap steps per phase, one or two pieces per step

    const int64_t w_half_stride = (int64_t)p.cout_pad * 8;
    const int64_t w_tap_stride = w_half_stride * n_pc;

    f32x4 bn_sc[CT], bn_sh[CT];
    f32x4 acc0[VT][CT], acc1[VT][CT], acc2[VT][CT];   // three output slices in flight

    // epilogue of output slice od from accumulator set ACC, then clear the set
#define SLIDE_EPILOGUE(OD, ACC)                                                                        \
    {                                                                                                  \
        _Pragma("unroll") for (int vt = 0; vt < VT; ++vt) {                                            \
            const int s_ = vt * 16 + j;                                                                \
            const int r_ = s_ / p.WT, c_ = s_ - r_ * p.WT;                                             \
            const bool valid_ = (s_ < nslots) && (oh0 + r_ < p.OH) && (ow0 + c_ < p.OW);               \
            if (valid_) {                                                                              \
                const int64_t yo_ = p.y_off0 + (int64_t)n * p.y_n_stride + (int64_t)(OD) * p.y_d_stride + \
                                    (int64_t)(oh0 + r_) * p.y_h_stride + (int64_t)(ow0 + c_) * 16 + g * 4; \
                const int64_t ro_ = p.r_off0 + (int64_t)n * p.r_n_stride + (int64_t)(OD) * p.r_d_stride + \
                                    (int64_t)(oh0 + r_) * p.r_h_stride + (int64_t)(ow0 + c_) * 16 + g * 4; \
                _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) {                                    \
                    f32x4 v_ = ACC[vt][ct] * bn_sc[ct] + bn_sh[ct];                                    \
                    if (p.res) v_ += *(const f32x4*)(p.res + ro_ + (int64_t)(ct0 + ct) * p.r_cb_stride); \
                    if (p.relu) { v_.x = fmaxf(v_.x, 0.f); v_.y = fmaxf(v_.y, 0.f); v_.z = fmaxf(v_.z, 0.f); v_.w = fmaxf(v_.w, 0.f); } \
                    *(f32x4*)(p.y + yo_ + (int64_t)(ct0 + ct) * p.y_cb_stride) = v_;                  \
                }                                                                                      \
            }                                                                                          \
            _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) ACC[vt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; \
        }                                                                                              \
    }

#define SLIDE_MFMA(ACC, W, B, KK)                                                                      \
    _Pragma("unroll") for (int vt = 0; vt < VT; ++vt)                                                  \
        _Pragma("unroll") for (int ct = 0; ct < CT; ++ct)                                              \
            ACC[vt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[ct][KK], B[vt][KK], ACC[vt][ct], 0, 0, 0);

    // weights of step (pc, tap t) for the three depth taps: widx = (dd*3+kh)*3+kw = dd*9 + t
    f32x2 wA[3][CT], wB[3][CT], bA[VT], bB[VT];
    // The weight loads are raw asm so that the compiler does not count them: it would otherwise put s_waitcnt vmcnt(0) in
    // front of the first MFMA of every tap step, which also waits for the LDS-DMA piece issued one step earlier (HBM
    // latency > one step).  Issue order inside a step is weights, then the piece(s); the step that uses the weights
    // waits with vmcnt(#pieces), so a piece has two steps to land.  Nothing may read W between load_w and SLIDE_WWAIT.
    const unsigned ts32 = (unsigned)(w_tap_stride * 4), hs32 = (unsigned)(w_half_stride * 4);
    auto load_w = [&](f32x2 (&W)[3][CT], int pc, int t) __attribute__((always_inline)) {
        const unsigned so = (unsigned)t * ts32 + (unsigned)pc * hs32;
#pragma unroll
        for (int dd = 0; dd < 3; ++dd) {
            const unsigned vo = wlane_off + (so + (unsigned)dd * 9u * ts32);
            asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(W[dd][0]) : "v"(vo), "s"(p.w));
            if (CT == 2) asm volatile("global_load_dwordx2 %0, %1, %2 offset:512" : "=v"(W[dd][CT - 1]) : "v"(vo), "s"(p.w));
        }
    };
#define SLIDE_WWAIT()                                                                                  \
    {                                                                                                  \
        if (two_pieces) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                               \
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");                                          \
    }

    int bufsel = 0;

// one tap step of input slice d_in: accumulate into (dd=0 -> A0 if v0), (dd=1 -> A1 if v1), (dd=2 -> A2 if v2).
// The memory instructions of the step (next step's weights, one or two LDS-DMA pieces, next step's B fragments) are pinned
// between MFMA runs of the dd=1 block so that they issue in the MFMA shadow (the wave issues in order: an instruction
// placed behind the last MFMA of a run costs its issue time, one placed between two MFMAs is free).  Edge slices of a depth
// segment (v1 false) issue them back to back.
#define SLIDE_CH_W(T, W_LD)                                                                            \
        {                                                                                              \
            const bool in_ph_ = (T) + 1 < 9;                                                           \
            load_w(W_LD, in_ph_ ? pc : nx_pc, in_ph_ ? (T) + 1 : 0);                                   \
        }
#define SLIDE_CH_DMA(T)                                                                                \
        {   /* steps beyond the last piece re-stage an earlier one (same bytes): the count stays static */ \
            int q_ = two_pieces ? 2 * (T) : (T), q1_ = q_ + 1;                                         \
            while (q_ >= pieces) q_ -= pieces;       /* tiny tiles: wrap as often as needed */          \
            while (q1_ >= pieces) q1_ -= pieces;                                                       \
            stage_piece(st_d, st_pc, bufsel ^ 1, q_);                                                  \
            if (two_pieces) stage_piece(st_d, st_pc, bufsel ^ 1, q1_);                                 \
        }
#define SLIDE_CH_B(T, B_LD)                                                                            \
        {                                                                                              \
            const bool in_ph_ = (T) + 1 < 9;                                                           \
            const int tn_ = in_ph_ ? (T) + 1 : 0;                                                      \
            const int to_ = in_ph_ ? (tn_ / 3) * seg_floats + (tn_ % 3) * 8 : 0;                       \
            _Pragma("unroll") for (int vt = 0; vt < VT; ++vt) B_LD[vt] = *(const f32x2*)(buf + lane_off[vt] + to_); \
        }
#define SLIDE_RUN(ACC, W, B, V0, V1, KK)                                                               \
    _Pragma("unroll") for (int vt = (V0); vt < (V1); ++vt)                                             \
        _Pragma("unroll") for (int ct = 0; ct < CT; ++ct)                                              \
            ACC[vt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[ct][KK], B[vt][KK], ACC[vt][ct], 0, 0, 0);

#define SLIDE_STEP(T, W_USE, B_USE, W_LD, B_LD, A0, A1, A2)                                            \
    {                                                                                                  \
        /* first tile row of the dd=1 block is unconditional: it carries the operand wait; on the two edge slices  \
           of a depth segment (v1 false) it feeds a set that is cleared before its next use */          \
        if ((T) > 0) SLIDE_WWAIT()   /* step 0's weights landed behind the vmcnt(0) that closed the previous phase */ \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        SLIDE_RUN(A1, W_USE[1], B_USE, 0, 1, 0)                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (v1) {                                                                                      \
            SLIDE_RUN(A1, W_USE[1], B_USE, 1, 2, 0)                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                         \
            SLIDE_CH_W(T, W_LD)                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                         \
            SLIDE_RUN(A1, W_USE[1], B_USE, 2, VT > 4 ? 4 : VT, 0)                                      \
            __builtin_amdgcn_sched_barrier(0);                                                         \
            SLIDE_CH_DMA(T)                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                         \
            SLIDE_RUN(A1, W_USE[1], B_USE, VT > 4 ? 4 : VT, VT, 0)                                     \
            SLIDE_RUN(A1, W_USE[1], B_USE, 0, 1, 1)                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                         \
            SLIDE_CH_B(T, B_LD)                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                         \
            SLIDE_RUN(A1, W_USE[1], B_USE, 1, VT, 1)                                                   \
        } else {                                                                                       \
            SLIDE_CH_W(T, W_LD)                                                                        \
            SLIDE_CH_DMA(T)                                                                            \
            SLIDE_CH_B(T, B_LD)                                                                        \
        }                                                                                              \
        if (v0) { SLIDE_MFMA(A0, W_USE[0], B_USE, 0) SLIDE_MFMA(A0, W_USE[0], B_USE, 1) }              \
        if (v2) { SLIDE_MFMA(A2, W_USE[2], B_USE, 0) SLIDE_MFMA(A2, W_USE[2], B_USE, 1) }              \
    }

// all phases of input slice d_in; afterwards output slice d_in-1 (set A2) is complete
#define SLIDE_SLICE(A0, A1, A2)                                                                        \
    {                                                                                                  \
        const bool v0 = d_in + 1 >= od_lo && d_in + 1 < od_hi;                                         \
        const bool v1 = d_in >= od_lo && d_in < od_hi;                                                 \
        const bool v2 = d_in - 1 >= od_lo && d_in - 1 < od_hi;                                         \
        for (int pc = 0; pc < n_pc; ++pc) {                                                            \
            const float* buf = lds + bufsel * buf_floats;                                              \
            const bool last_pc = pc + 1 == n_pc;                                                       \
            const int st_d = last_pc && d_in + 1 <= din_hi ? d_in + 1 : d_in;   /* after the last slice: harmless re-stage */ \
            const int st_pc = last_pc ? 0 : pc + 1;                                                    \
            const int nx_pc = st_pc;                                                                   \
            _Pragma("unroll") for (int vt = 0; vt < VT; ++vt) bA[vt] = *(const f32x2*)(buf + lane_off[vt]); \
            for (int t = 0; t < 8; t += 2) {                                                           \
                SLIDE_STEP(t, wA, bA, wB, bB, A0, A1, A2)                                              \
                SLIDE_STEP(t + 1, wB, bB, wA, bA, A0, A1, A2)                                          \
            }                                                                                          \
            SLIDE_STEP(8, wA, bA, wB, bB, A0, A1, A2)                                                  \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
            __builtin_amdgcn_sched_barrier(0);                                                         \
            _Pragma("unroll") for (int dd = 0; dd < 3; ++dd)                                           \
                _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) wA[dd][ct] = wB[dd][ct];             \
            bufsel ^= 1;                                                                               \
        }                                                                                              \
        if (v2) SLIDE_EPILOGUE(d_in - 1, A2)                                                           \
        else { _Pragma("unroll") for (int vt = 0; vt < VT; ++vt) _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) A2[vt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; } \
        if (v1 && d_in + 1 == D) SLIDE_EPILOGUE(d_in, A1)                                              \
    }

    // output od lives in set (od mod 3): slice d_in feeds dd0 -> set (d_in+1)%3, dd1 -> d_in%3, dd2 -> (d_in+2)%3.
    // The loop is unrolled by 3 so the sets rotate without runtime indexing; it starts at the right phase for din_lo.
#pragma unroll 1
    while (ucur < u1) {
        // ---- segment setup: column, cout group and depth range [od_lo, od_hi); input slices od_lo-1 .. od_hi clipped to the volume
        {
            const long colid = ucur / D;
            od_lo = (int)(ucur - colid * D);
            od_hi = (long)od_lo + (u1 - ucur) < D ? od_lo + (int)(u1 - ucur) : D;
            ucur += od_hi - od_lo;
            int cid = (int)(colid % cols);
            ct0 = (int)(colid / cols) * CT;
            const int wt = cid % n_wt; cid /= n_wt;
            const int rt = cid % n_rt;
            n = cid / n_rt;
            oh0 = rt * p.R; ow0 = wt * p.WT;
            din_lo = od_lo > 0 ? od_lo - 1 : 0;
            din_hi = od_hi < D ? od_hi : D - 1;      // inclusive
            xcol = p.x + (int64_t)n * p.x_n_stride + (int64_t)(oh0 + cls.dh0) * p.x_h_stride + (int64_t)(ow0 + cls.dw0) * 16;
            wlane_off = (unsigned)(((ct0 * 16 + j) * 8 + g * 2) * 4);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                bn_sc[ct] = *(const f32x4*)(p.scale + (ct0 + ct) * 16 + g * 4);
                bn_sh[ct] = *(const f32x4*)(p.shift + (ct0 + ct) * 16 + g * 4);
            }
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) { acc0[vt][ct] = acc1[vt][ct] = acc2[vt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            for (int q = 0; q < pieces; ++q) stage_piece(din_lo, 0, bufsel, q);
            load_w(wA, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        for (int d_in = din_lo - din_lo % 3;;) {
            if (d_in >= din_lo) SLIDE_SLICE(acc1, acc0, acc2)
            if (++d_in > din_hi) break;
            if (d_in >= din_lo) SLIDE_SLICE(acc2, acc1, acc0)
            if (++d_in > din_hi) break;
            if (d_in >= din_lo) SLIDE_SLICE(acc0, acc2, acc1)
            if (++d_in > din_hi) break;
        }
    }
#undef SLIDE_SLICE
#undef SLIDE_STEP
#undef SLIDE_CH_W
#undef SLIDE_CH_DMA
#undef SLIDE_CH_B
#undef SLIDE_RUN
#undef SLIDE_WWAIT
#undef SLIDE_MFMA
#undef SLIDE_EPILOGUE
}

template <int VT, int CT>
int launch(const drc_tapconv_params& p, hipStream_t stream) {
    const long cols = (long)p.N * ((p.OH + p.R - 1) / p.R) * ((p.OW + p.WT - 1) / p.WT);
    const int ppr = (2 * (p.WT + 2) + 63) / 64;
    const size_t lds = (size_t)2 * (p.R + 2) * ppr * 1024 * SLIDE_WAVES;      // two tile buffers per wave, rows padded to whole 1 KiB pieces
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)tapslide_kernel<VT, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    // equal shares of output slices for every resident wave slot; shares stay >= 3 slices so the re-staged seam slices
    // (two per segment) remain a small part of a wave's work
    static int occ_blocks = 0;   // per-instantiation, idempotent
    if (!occ_blocks) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, tapslide_kernel<VT, CT>, 64 * SLIDE_WAVES, lds) != hipSuccess || nb < 1) nb = 1;
        occ_blocks = nb;
    }
    const long units = cols * (p.cout_pad / 16 / CT) * p.OD;
    long workers = 256L * SLIDE_WAVES * occ_blocks;
    if (workers > units / 3) workers = units / 3 > workers / 2 ? units / 3 : (units < workers ? units : workers);   // small volumes: shares down to one slice
    if (workers < SLIDE_WAVES) workers = SLIDE_WAVES;
    dim3 grid((unsigned)((workers + SLIDE_WAVES - 1) / SLIDE_WAVES), 1, 1);
    hipLaunchKernelGGL((tapslide_kernel<VT, CT>), grid, dim3(64 * SLIDE_WAVES), lds, stream, p);
    return (int)hipGetLastError();
}

template <int CT>
int launch_vt(int nvt, const drc_tapconv_params& p, hipStream_t s) {
    switch (nvt) {
        case 1: return launch<1, CT>(p, s);
        case 2: return launch<2, CT>(p, s);
        case 3: return launch<3, CT>(p, s);
        case 4: return launch<4, CT>(p, s);
        case 5: return launch<5, CT>(p, s);
        case 6: return launch<6, CT>(p, s);
        case 7: return launch<7, CT>(p, s);
    }
    return -3;
}

}  // namespace

extern "C" int drc_tapconv3d_slide_fwd(const drc_tapconv_params* pp, int cout_tiles_per_wave, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD <= 0 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (p.cout_pad <= 0 || (p.cout_pad & 15) || p.cb_in <= 0) return -2;
    if (p.R <= 0 || p.WT <= 0 || p.R * p.WT > 112) return -3;
    const drc_tap_class& k = p.cls[0];
    // only the stride-1 3x3x3 class with unit tap spacing and halo 1 in depth
    if (p.n_classes != 1 || p.in_mul != 1 || p.out_mul != 1 || k.nd != 3 || k.nh != 3 || k.nw != 3 || k.sd != 1 || k.sh != 1 ||
        k.sw != 1 || k.wbase != 0 || k.wsd != 9 || k.wsh != 3 || k.wsw != 1)
        return -4;
    const int ppr = (2 * (p.WT + 2) + 63) / 64;
    if ((p.R + 2) * ppr > 18 || (size_t)2 * (p.R + 2) * ppr * 1024 * SLIDE_WAVES > 160 * 1024) return -5;   // <= 2 pieces per tap step
    const int ct = p.cout_pad / 16;
    const int CT = cout_tiles_per_wave;
    if ((CT != 1 && CT != 2) || ct % CT) return -2;
    const int nvt = (p.R * p.WT + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
    return CT == 2 ? launch_vt<2>(nvt, p, s) : launch_vt<1>(nvt, p, s);
}
