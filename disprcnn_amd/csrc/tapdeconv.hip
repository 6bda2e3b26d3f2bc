// tapdeconv.hip -- ConvTranspose3d(k3, s2, p1, op1) (+BN, +residual, +ReLU) with the 8 output-parity classes FUSED (gfx950).
//
//   reference: hourglass.conv5 / conv6 (stackhourglass.py:22-30) and, transposed, the data gradient of the stride-2
//   Conv3d layers (hourglass.conv1 / conv3) in training.
//
// o = 2i - 1 + k per dimension: an even output (o = 2j) has one tap (i = j, k = 1), an odd output (o = 2j+1) two taps
// ((i = j, k = 2), (i = j+1, k = 0)).  tapconv.hip runs the 8 parity classes as 8 separate convolutions with 1..8 taps each,
// so the 1- and 2-tap classes stage a whole input tile for a handful of MFMAs.  Here a wave owns an INPUT tile (R rows x WT
// columns of input slice i, CT*16 output channels) and produces all 8 classes of the 2x2x2-upsampled output block from it:
// per 8-channel phase it stages the (R+1) x (WT+1) tiles of slices i and i+1 once and runs all 27 taps on them
// (27 * VT * CT * 2 MFMAs per staged pair), accumulating into 8 * VT * CT accumulator tiles.
// MFMA mapping, blocked layout, weight packing and the LDS-DMA / uncounted-weight-load protocol are those of tapslide.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

#define DC_WAVES 4

namespace {

// ---- the tap steps of a phase, ordered so that consecutive steps share a B fragment (tile offset (dh, dw)) ----
// combo c (0..8): B offset (dh, dw) and kernel index (kh, kw); output parity ph = (kh != 1), pw = (kw != 1)
//   c: 0..3 -> (dh,dw) = (0,0);  4,5 -> (0,1);  6,7 -> (1,0);  8 -> (1,1)
__device__ constexpr int c_kh(int c) { return c >= 6 ? 0 : ((c == 0 || c == 1 || c == 4) ? 1 : 2); }
__device__ constexpr int c_kw(int c) { return (c == 4 || c == 5 || c == 8) ? 0 : ((c == 0 || c == 2 || c == 6) ? 1 : 2); }
__device__ constexpr int c_set(int c) { return c < 4 ? 0 : (c < 6 ? 1 : (c < 8 ? 2 : 3)); }
__device__ constexpr int c_hw(int c) { return (c_kh(c) != 1 ? 2 : 0) + (c_kw(c) != 1 ? 1 : 0); }       // (ph, pw) part of the class index
// step S (0..17): S < 9 -> slice i (dz = 0), combo S, depth taps kd = 1 (pd = 0) and kd = 2 (pd = 1);
//                 S >= 9 -> slice i+1 (dz = 1), combo S-9, depth tap kd = 0 (pd = 1)
__device__ constexpr int s_dz(int S) { return S >= 9 ? 1 : 0; }
__device__ constexpr int s_combo(int S) { return S >= 9 ? S - 9 : S; }
__device__ constexpr int s_widx(int S, int slot) { return ((S >= 9 ? 0 : 1 + slot) * 3 + c_kh(s_combo(S))) * 3 + c_kw(s_combo(S)); }
__device__ constexpr int s_cls(int S, int slot) { return (S >= 9 ? 1 : slot) * 4 + c_hw(s_combo(S)); }
__device__ constexpr int s_set(int S) { return s_dz(S) * 4 + c_set(s_combo(S)); }                       // B fragment set 0..7 of the phase
__device__ constexpr bool s_set_first(int S) { return S == 0 || s_set(S) != s_set(S - 1); }
__device__ constexpr int set_dz(int s) { return s >> 2; }
__device__ constexpr int set_dh(int s) { return (s & 3) >= 2 ? 1 : 0; }
__device__ constexpr int set_dw(int s) { return (s & 1); }

#define DC_MAXP 5   /* LDS-DMA pieces per slice tile */

template <int VT, int CT>
__global__ __launch_bounds__(64 * DC_WAVES, 2) void tapdeconv_kernel(const drc_tapconv_params p) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const int g = lane >> 4;

    const int D = p.OD, H = p.OH, W = p.OW;          // INPUT grid; the output is (2D, 2H, 2W)
    const int n_wt = (W + p.WT - 1) / p.WT;
    const int n_rt = (H + p.R - 1) / p.R;
    const int per_cg = p.N * D * n_rt * n_wt;        // groups of one cout group: (n, i, row tile, col tile)
    const long groups = (long)(p.cout_pad / 16 / CT) * per_cg;
    const long workers = (long)gridDim.x * DC_WAVES;
    const long wid = (long)blockIdx.x * DC_WAVES + wave;
    long gcur = groups * wid / workers;              // equal contiguous shares
    const long gend = groups * (wid + 1) / workers;
    if (gcur >= gend) return;                        // wave-uniform; no workgroup barrier in this kernel

    const int cols_in = p.WT + 1;
    const int upr = cols_in * 2;                     // 16-byte units per tile row (8 channels x 4 B = 2 units per voxel)
    const int units = (p.R + 1) * upr;
    const int pieces = (units + 63) >> 6;            // LDS-DMA pieces (64 lanes x 16 B) per slice tile; rows are packed densely
    const int tile_floats = pieces * 256;
    const int buf_floats = 2 * tile_floats;          // [dz]
    float* lds = lds_all + wave * (2 * buf_floats);  // double-buffered
    const int nslots = p.R * p.WT;
    const unsigned magic = ((1u << 20) + upr - 1) / upr;   // u / upr == (u * magic) >> 20 for u * upr < 2^20

    int lane_off[VT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
        const int s = vt * 16 + j;
        int r = s / p.WT, c = s - r * p.WT;
        if (s >= nslots) { r = 0; c = 0; }
        lane_off[vt] = (r * cols_in + c) * 8 + g * 2;
    }

    const int n_pc = p.cb_in * 2;                    // (channel block, half) phases
    const int64_t w_half_stride = (int64_t)p.cout_pad * 8;
    const int64_t w_tap_stride = w_half_stride * n_pc;

    struct Group { int n, i, oh0, ow0, ct0, rmax, cmax, dz1; const float* base; };
    auto decode = [&](long gidx) __attribute__((always_inline)) -> Group {
        Group q;
        int r = (int)(gidx % per_cg);
        q.ct0 = (int)(gidx / per_cg) * CT;
        const int wt = r % n_wt; r /= n_wt;
        const int rt = r % n_rt; r /= n_rt;
        q.i = r % D; q.n = r / D;
        q.oh0 = rt * p.R; q.ow0 = wt * p.WT;
        q.rmax = H - q.oh0 < p.R ? H - q.oh0 : p.R;          // tile row r reads input row oh0 + r <= H (the zero halo row)
        q.cmax = W - q.ow0 < p.WT ? W - q.ow0 : p.WT;
        q.dz1 = q.i + 1 < D;                                  // slice i+1 == D is the all-zero halo: skipped
        q.base = p.x + (int64_t)q.n * p.x_n_stride + (int64_t)(q.i + p.cls[0].dd0) * p.x_d_stride +
                 (int64_t)(q.oh0 + p.cls[0].dh0) * p.x_h_stride + (int64_t)(q.ow0 + p.cls[0].dw0) * 16;
        return q;
    };

    // per-lane byte offsets of the LDS-DMA pieces of a tile (all lanes active: lanes past the tile, and rows / columns past the
    // tensor's halo on ragged tiles, re-read a valid neighbour); recomputed only when the staged group's raggedness changes
    unsigned poff[DC_MAXP];
    int poff_rmax = -1, poff_cmax = -1;
    auto set_poff = [&](int rmax, int cmax) __attribute__((always_inline)) {
        poff_rmax = rmax; poff_cmax = cmax;
#pragma unroll
        for (int q = 0; q < DC_MAXP; ++q) {
            int u = q * 64 + lane;
            u = u < units ? u : units - 1;
            int r = (int)(((unsigned)u * magic) >> 20);
            const int c = u - r * upr;
            int vox = c >> 1;
            r = r < rmax ? r : rmax;
            vox = vox < cmax ? vox : cmax;
            poff[q] = (unsigned)((r * (int)p.x_h_stride + vox * 16 + (c & 1) * 4) * 4);
        }
    };
    // LDS-DMA piece Q (compile-time) of the slice-dz tile whose channel-half origin is sbase, into tile buffer bufi
#define DC_STAGE(SBASE, DZ, Q, BUFI)                                                                   \
    __builtin_amdgcn_global_load_lds(GLOBAL_PTR((const char*)((SBASE) + (int64_t)(DZ) * p.x_d_stride) + poff[Q]),   \
                                     LDS_PTR(lds + (BUFI) * buf_floats + (DZ) * tile_floats + (Q) * 256), 16, 0, 0)
    auto phase_base = [&](const Group& G, int pc) __attribute__((always_inline)) -> const float* {
        return G.base + (int64_t)(pc >> 1) * p.x_cb_stride + (pc & 1) * 8;
    };

    // weights: packed [widx = (kd*3+kh)*3+kw][cb*2+half][cout_pad][8]; raw asm loads (not counted by the compiler, see tapslide.hip)
    unsigned wlane_off;                              // per-lane byte offset inside a (widx, pc) block
    const unsigned ts32 = (unsigned)(w_tap_stride * 4), hs32 = (unsigned)(w_half_stride * 4);
    auto load_w = [&](f32x2 (&Wd)[CT], int widx, unsigned wph) __attribute__((always_inline)) {
        unsigned ts = ts32;
        asm volatile("" : "+s"(ts));                 // keep the 27 tap offsets out of loop-invariant registers
        const unsigned vo = wlane_off + ((unsigned)widx * ts + wph);
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(Wd[0]) : "v"(vo), "s"(p.w));
        if (CT == 2) asm volatile("global_load_dwordx2 %0, %1, %2 offset:512" : "=v"(Wd[CT - 1]) : "v"(vo), "s"(p.w));
    };

    // accumulators are cleared from a zero produced by a volatile asm at the point of use: a literal zero vector per tile would
    // be hoisted out of the group loop and parked in 4 AGPRs per tile
#define DC_CLEAR_ACC()                                                                                 \
    {                                                                                                  \
        float z_;                                                                                      \
        asm volatile("v_mov_b32 %0, 0" : "=v"(z_));                                                    \
        const f32x4 z4_ = {z_, z_, z_, z_};                                                            \
        _Pragma("unroll") for (int c = 0; c < 8; ++c)                                                  \
            _Pragma("unroll") for (int vt = 0; vt < VT; ++vt)                                          \
                _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) acc[c][vt][ct] = z4_;                \
    }
    f32x4 acc[8][VT][CT];
    DC_CLEAR_ACC()

    f32x2 wbuf[2][2][CT], bfr[2][VT];                // [step parity][depth-tap slot][ct], [set parity][vt]
    int bufsel = 0;

    Group cur = decode(gcur);
    set_poff(cur.rmax, cur.cmax);
    // prologue: first tile pair of the wave
    {
        const float* sb0 = phase_base(cur, 0);
#pragma unroll
        for (int q = 0; q < DC_MAXP; ++q)
            if (q < pieces) {
                DC_STAGE(sb0, 0, q, 0);
                if (cur.dz1) DC_STAGE(sb0, 1, q, 0);
            }
    }
    wlane_off = (unsigned)(((cur.ct0 * 16 + j) * 8 + g * 2) * 4);
    load_w(wbuf[0][0], s_widx(0, 0), 0u);
    load_w(wbuf[0][1], s_widx(0, 1), 0u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#define DC_MFMA(CLS, SLOT, VT0, VT1, K)                                                                \
    _Pragma("unroll") for (int vt = (VT0); vt < (VT1); ++vt)                                           \
        _Pragma("unroll") for (int ct = 0; ct < CT; ++ct)                                              \
            acc[CLS][vt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wbuf[wsel_][SLOT][ct][K], bfr[bsel_][vt][K], acc[CLS][vt][ct], 0, 0, 0);

    // one tap step; the memory instructions of the step are pinned between MFMA runs so that they issue in the MFMA shadow.
    // nx / nx_pc: what the phase after this one reads (same group next pc, or the next group's pc 0)
#define DC_STEP(S)                                                                                     \
    {                                                                                                  \
        constexpr int wsel_ = (S) & 1, bsel_ = s_set(S) & 1, c0_ = s_cls(S, 0), c1_ = s_cls(S, 1);     \
        constexpr bool two_ = (S) < 9;                                                                 \
        if ((S) > 0) {   /* weights of this step: issued one step ago, followed by at most one LDS-DMA piece */ \
            constexpr int sp_ = (S) - 1;                                                               \
            const bool dma_ = sp_ < DC_MAXP ? sp_ < np0 : (sp_ < 2 * DC_MAXP && sp_ - DC_MAXP < np1);  \
            if (dma_) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");                                 \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                      \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        DC_MFMA(c0_, 0, 0, 1, 0)                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        {   /* weights of the next step (or of the next phase's step 0) */                             \
            constexpr bool last_dz0_ = (S) == 8, last_ = (S) == 17;                                    \
            if (last_ || (last_dz0_ && !cur.dz1)) { load_w(wbuf[wsel_ ^ 1][0], s_widx(0, 0), wph_nx); load_w(wbuf[wsel_ ^ 1][1], s_widx(0, 1), wph_nx); } \
            else { load_w(wbuf[wsel_ ^ 1][0], s_widx((S) + 1, 0), wph); if ((S) + 1 < 9) load_w(wbuf[wsel_ ^ 1][1], s_widx((S) + 1, 1), wph); } \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        DC_MFMA(c0_, 0, 1, VT, 0)                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if ((S) < DC_MAXP) { if ((S) < np0) DC_STAGE(sb_nx, 0, (S) < DC_MAXP ? (S) : 0, bufsel ^ 1); }   /* steps 0..4: slice i */ \
        else if ((S) < 2 * DC_MAXP) { if ((S) - DC_MAXP < np1) DC_STAGE(sb_nx, 1, (S) >= DC_MAXP && (S) < 2 * DC_MAXP ? (S) - DC_MAXP : 0, bufsel ^ 1); } \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        DC_MFMA(c0_, 0, 0, VT, 1)                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (s_set_first(S) && s_set(S) < 7 && (s_set(S) != 3 || cur.dz1)) {   /* B fragments of the next set */ \
            constexpr int sn_ = s_set(S) + 1;                                                          \
            int so_ = set_dz(sn_) * tile_floats + (set_dh(sn_) * cols_in + set_dw(sn_)) * 8;           \
            asm volatile("" : "+s"(so_));                                                              \
            const float* bp_ = buf + so_;                                                              \
            _Pragma("unroll") for (int vt = 0; vt < VT; ++vt) bfr[sn_ & 1][vt] = *(const f32x2*)(bp_ + lane_off[vt]); \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (two_) { DC_MFMA(c1_, 1, 0, VT, 0) DC_MFMA(c1_, 1, 0, VT, 1) }                              \
    }

#pragma unroll 1
    for (;;) {
        const bool has_next_group = gcur + 1 < gend;
        Group nxg = cur;
        for (int pc = 0; pc < n_pc; ++pc) {
            const bool last_pc = pc + 1 == n_pc;
            if (last_pc && has_next_group) {
                nxg = decode(gcur + 1);
                if (nxg.rmax != poff_rmax || nxg.cmax != poff_cmax) set_poff(nxg.rmax, nxg.cmax);
            }
            const Group& nx = last_pc ? nxg : cur;
            const int nx_pc = last_pc ? 0 : pc + 1;
            const bool stage_on = !last_pc || has_next_group;
            const int np0 = stage_on ? pieces : 0, np1 = stage_on && nx.dz1 ? pieces : 0;   // pieces to stage for slices i / i+1
            const float* sb_nx = phase_base(nx, nx_pc);
            const unsigned wph = (unsigned)pc * hs32, wph_nx = (unsigned)nx_pc * hs32;
            const float* buf = lds + bufsel * buf_floats;
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) bfr[0][vt] = *(const f32x2*)(buf + lane_off[vt]);
            DC_STEP(0) DC_STEP(1) DC_STEP(2) DC_STEP(3) DC_STEP(4) DC_STEP(5) DC_STEP(6) DC_STEP(7) DC_STEP(8)
            if (cur.dz1) {
                DC_STEP(9) DC_STEP(10) DC_STEP(11) DC_STEP(12) DC_STEP(13) DC_STEP(14) DC_STEP(15) DC_STEP(16) DC_STEP(17)
            }
            if (!cur.dz1 && np1 == DC_MAXP) DC_STAGE(sb_nx, 1, DC_MAXP - 1, bufsel ^ 1);   /* its step (9) did not run */
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (!cur.dz1) {   /* 9 steps: the next phase's first weights sit in parity 1 */
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) { wbuf[0][0][ct] = wbuf[1][0][ct]; wbuf[0][1][ct] = wbuf[1][1][ct]; }
            }
            bufsel ^= 1;
        }

        // ---- epilogue: 8 parity classes x VT tiles: folded BN, residual, ReLU, store; clear the accumulators
        {
            f32x4 bn_sc[CT], bn_sh[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                bn_sc[ct] = *(const f32x4*)(p.scale + (cur.ct0 + ct) * 16 + g * 4);
                bn_sh[ct] = *(const f32x4*)(p.shift + (cur.ct0 + ct) * 16 + g * 4);
            }
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) {
                const int s = vt * 16 + j;
                const int r = s / p.WT, c = s - r * p.WT;
                const bool valid = (s < nslots) && (cur.oh0 + r < H) && (cur.ow0 + c < W);
                if (valid) {
                    const int64_t yo = p.y_off0 + (int64_t)cur.n * p.y_n_stride + (int64_t)(2 * cur.i) * p.y_d_stride +
                                       (int64_t)(2 * (cur.oh0 + r)) * p.y_h_stride + (int64_t)(2 * (cur.ow0 + c)) * 16 + g * 4;
                    const int64_t ro = p.r_off0 + (int64_t)cur.n * p.r_n_stride + (int64_t)(2 * cur.i) * p.r_d_stride +
                                       (int64_t)(2 * (cur.oh0 + r)) * p.r_h_stride + (int64_t)(2 * (cur.ow0 + c)) * 16 + g * 4;
#pragma unroll
                    for (int c8 = 0; c8 < 8; ++c8) {
                        const int pd = c8 >> 2, ph = (c8 >> 1) & 1, pw = c8 & 1;
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) {
                            f32x4 v = acc[c8][vt][ct] * bn_sc[ct] + bn_sh[ct];
                            if (p.res)
                                v += *(const f32x4*)(p.res + ro + (int64_t)pd * p.r_d_stride + (int64_t)ph * p.r_h_stride + pw * 16 +
                                                     (int64_t)(cur.ct0 + ct) * p.r_cb_stride);
                            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                            *(f32x4*)(p.y + yo + (int64_t)pd * p.y_d_stride + (int64_t)ph * p.y_h_stride + pw * 16 +
                                      (int64_t)(cur.ct0 + ct) * p.y_cb_stride) = v;
                        }
                    }
                }
            }
            DC_CLEAR_ACC()
        }
        if (!has_next_group) break;
        ++gcur;
        cur = nxg;
        // the next group may belong to another cout group: its step-0 weights were loaded with the old lane offset
        const unsigned wl = (unsigned)(((cur.ct0 * 16 + j) * 8 + g * 2) * 4);
        if (wl != wlane_off) {
            wlane_off = wl;
            load_w(wbuf[0][0], s_widx(0, 0), 0u);
            load_w(wbuf[0][1], s_widx(0, 1), 0u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
#undef DC_STEP
#undef DC_MFMA
#undef DC_STAGE
#undef DC_CLEAR_ACC
}

template <int VT, int CT>
int launch(const drc_tapconv_params& p, hipStream_t stream) {
    const int pieces = ((p.R + 1) * (p.WT + 1) * 2 + 63) / 64;
    const size_t lds = (size_t)2 * 2 * pieces * 1024 * DC_WAVES;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)tapdeconv_kernel<VT, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    static int occ_blocks = 0;
    if (!occ_blocks) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, tapdeconv_kernel<VT, CT>, 64 * DC_WAVES, lds) != hipSuccess || nb < 1) nb = 1;
        occ_blocks = nb;
    }
    const long groups = (long)(p.cout_pad / 16 / CT) * p.N * p.OD * ((p.OH + p.R - 1) / p.R) * ((p.OW + p.WT - 1) / p.WT);
    long workers = 256L * DC_WAVES * occ_blocks;
    if (workers > groups) workers = groups;
    dim3 grid((unsigned)((workers + DC_WAVES - 1) / DC_WAVES), 1, 1);
    hipLaunchKernelGGL((tapdeconv_kernel<VT, CT>), grid, dim3(64 * DC_WAVES), lds, stream, p);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int drc_deconv3d_k3s2_fwd(const drc_tapconv_params* pp, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD <= 0 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (p.cout_pad <= 0 || (p.cout_pad & 15) || p.cb_in <= 0) return -2;
    if (p.in_mul != 1 || p.out_mul != 2) return -4;
    if (p.R <= 0 || p.WT <= 0 || p.R * p.WT > 64) return -3;
    const int pieces = ((p.R + 1) * (p.WT + 1) * 2 + 63) / 64;
    if (pieces > DC_MAXP || (long)(p.R + 1) * (p.WT + 1) * 2 * (p.WT + 1) * 2 >= (1L << 20)) return -5;
    if ((size_t)4 * pieces * 1024 * DC_WAVES > 160 * 1024) return -5;
    const int nvt = (p.R * p.WT + 15) / 16;
    const int ct = p.cout_pad / 16;
    hipStream_t s = (hipStream_t)stream;
    if (nvt <= 2 && ct % 2 == 0) return nvt == 1 ? launch<1, 2>(p, s) : launch<2, 2>(p, s);
    switch (nvt) {
        case 1: return launch<1, 1>(p, s);
        case 2: return launch<2, 1>(p, s);
        case 3: return launch<3, 1>(p, s);
        case 4: return launch<4, 1>(p, s);
    }
    return -3;   // 8 * VT * CT accumulator tiles must fit the register file: VT <= 4
}
