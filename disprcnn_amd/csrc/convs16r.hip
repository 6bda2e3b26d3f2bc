// convs16r.hip -- 3x3 stride-1 2D convolution (+BN, +residual, +ReLU) in SPLIT-f16 arithmetic on the f16 matrix cores (gfx950 / CDNA4), round 5.
//
//   reference arithmetic: convbn k3 s1 p1 d1 of submodule.py:9-16 -- firstconv[2], firstconv[4], the BasicBlocks of layer1 / layer2 / layer3
//   (submodule.py:40-60, 68-95) of PSMNet's feature_extraction; fp32 (config/defaults.py:22).
//
// The 2D member of the convs16 family (convs16.hip has the arithmetic, the RS16 layout -- here halfs [N][C/32][H+2][8 chunks][W+2][8] -- and the
// error model).  The 3D kernel walks a column's DEPTH with all 27 taps in registers; a 2D map has no depth, so this one walks the ROWS of a
// 28-row block: a workgroup of four waves owns 4/KW x-adjacent tiles of 28 columns, input row yi of the block (one row of 28*TPW + 2 voxels
// x all input channels: 8-16 KB) is staged once through a three-slot LDS ring and feeds the three output rows it touches,
//     kh = 0 -> row yi+1, kh = 1 -> row yi, kh = 2 -> row yi-1        (three live accumulators, rotated by the row index mod 3)
// so every wave holds its 9 taps x KS K-slices in registers (72 KS VGPRs) and issues 27 KS MFMAs per step.  K is split over KW waves, each
// taking KS 16-channel slices (cin = 16 KW KS); a finished row is exchanged through LDS, BN / residual / ReLU / hi-lo split run in the
// shadow of the next row's MFMAs, exactly as in convs16.hip.  A column = (image, x group, block of 28 rows) takes 30 steps (the rows above
// and below the block are the neighbouring blocks' or the stored zero halo); 30 % 3 == 0, so the accumulator rotation is static across
// columns and the pipeline never drains between them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/disprcnn_hip.h"
#include "s16_ovf.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define S16_WAITCNT(vm, lgkm) (((vm) & 15) | (7 << 4) | ((lgkm) << 8) | (((vm) >> 4) << 14))

namespace {

constexpr int RING = 3;
constexpr int RB = 28;            // output rows per column
constexpr int STEPS = RB + 2;     // input rows per column
constexpr int XW = 4096;          // bytes a wave publishes per row: its 16 accumulator registers

// DIL = 2: the dilated layers (layer4, submodule.py:73): the taps of an output row are the rows 2 above / below, so the even and the odd rows of
// a 56-row block are two independent columns that walk every other row; the staged row carries two halo voxels on either side (fetched
// from the stored zero halo where they fall outside the map: the RS16 layout keeps one halo voxel).
template <int KW, int KS, bool RES, int DIL = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, KS == 1 ? 2 : 1))) void convs16r_kernel(const drc_s16conv_params p) {
    constexpr int TPW = 4 / KW;                 // x-adjacent MFMA tiles (28 columns each) per workgroup
    constexpr int SX = 28 * TPW + 2 * DIL;      // staged voxels per row
    constexpr int PV = TPW == 2 ? 64 : 32;      // voxels per chunk plane of a slab (lanes 28..31 of a tile over-read up to voxel 28 TPW + 5: the next plane's data,
                                                // finite garbage that only reaches the idle columns of the accumulator)
    constexpr int CPB = PV * 16;
    constexpr int PPI = 64 / PV;                // chunk planes one LDS-DMA instruction fills
    constexpr int CBI = KW * KS / 2;            // 32-channel input blocks
    constexpr int PLANES = CBI * 8;
    constexpr int SLAB = PLANES * CPB;
    constexpr int NL = PLANES / PPI / 4;        // LDS-DMA instructions per wave and slab
    constexpr int OWN = 16 / KW;
    constexpr int NG = 3 * KS;                  // tap groups per step: (kw, K slice), 9 MFMAs each
    constexpr int FPG = (OWN + NG - 2) / (NG - 1);
    constexpr int NR = RES ? 2 : 0;
    constexpr int NS = 2;
    static_assert(SX <= PV && PLANES % (PPI * 4) == 0, "slab geometry");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* ring = lds;
    char* xchg = lds + RING * SLAB;             // [2 parities][4 waves][XW]

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int n_ = lane & 31, g = lane >> 5;
    const int r = wave / KW, k = wave % KW;     // tile of the workgroup, K share
    const int n_ct = p.cout / 32;
    const int ct = (int)((blockIdx.x >> 3) % n_ct);   // the cout tiles of one column set side by side on one XCD (second tile's input: L2 hits)

    const int H = p.H, W = p.W;
    const int Wp = W + 2;
    const long rowB = (long)Wp * 128;
    const long cbB = (long)(H + 2) * rowB;
    const long xnB = (long)CBI * cbB;
    const long ynB = (long)n_ct * cbB;

    f16x8 wh[KS][9], wl[KS][9];
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        const char* wb = (const char*)p.w + ((long)(ct * (KW * KS) + k * KS + j) * 18) * 1024 + lane * 16;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            wh[j][t] = *(const f16x8*)(wb + (t * 2) * 1024);
            wl[j][t] = *(const f16x8*)(wb + (t * 2 + 1) * 1024);
        }
    }
    float sc[OWN], sh[OWN];
#pragma unroll
    for (int e = 0; e < OWN; ++e) {
        const int reg = k * OWN + e;
        const int co = ct * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * g;
        sc[e] = p.scale[co];
        sh[e] = p.shift[co];
    }
    // LDS-DMA lane geometry: lane -> (chunk plane of the instruction, voxel)
    const int dps = lane / PV, dv = lane - dps * PV;
    const unsigned dplane = (unsigned)(dps * (Wp * 16));
    const int dvox = dv < SX ? dv : 0;          // idle lanes re-read voxel 0 (their LDS cells are only over-read)
    unsigned bfrag[KS];
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        const int sl = k * KS + j;
        bfrag[j] = (unsigned)(((sl >> 1) * 8 + (sl & 1) * 2 + g) * CPB + (r * 28 + n_) * 16);     // hi; lo at + 4 * CPB; tap kw at + kw * DIL * 16
    }
    const __attribute__((address_space(3))) char* ringl = (const __attribute__((address_space(3))) char*)ring;
    typedef const __attribute__((address_space(3))) f16x8 lds_frag;

    // x groups and row blocks (DIL 2: row blocks x parities, H / (RB * DIL) * DIL).  DIL 1 takes ANY map size (round 5b: the trunk's and the
    // FPN's maps, 94 x 310 ...): the last x group / row block is ragged -- its staged voxels beyond the row's end are the next chunk plane's
    // data (finite, they only reach output columns that are not stored), its rows beyond the bottom halo are clamped onto the halo row, and
    // the stores / residual loads of lanes and rows outside the map are dropped
    const int n_xg = (W + 28 * TPW - 1) / (28 * TPW), n_rb = (H + RB - 1) / RB;
    const unsigned xcd = blockIdx.x & 7, qx = (blockIdx.x >> 3) / n_ct, per_xcd = (gridDim.x >> 3) / n_ct;
    const unsigned cols_unit = (unsigned)n_rb * n_xg;
    if (per_xcd == 0) return;                              // (a grid that is not 8 x workers x cout tiles: nothing to walk -- never an endless loop)

    struct Col { unsigned n; int row0, x0; bool valid; };      // row0: first output row (logical)
    auto col_of = [&](unsigned it) __attribute__((always_inline)) {
        // many units (ROI crops): XCD q walks the units n % 8 == q, so that the row blocks and x groups of a unit (shared halo rows / columns)
        // meet in one L2.  Few units (the two images of a stereo pair in the trunk: only two of eight XCDs would get work): the columns
        // are dealt to all workers in turn
        const bool flat = (unsigned)p.N < 16u;
        const unsigned j = flat ? (it * per_xcd + qx) * 8 + xcd : it * per_xcd + qx;
        const unsigned nl = j / cols_unit;
        const unsigned rem = j - nl * cols_unit;
        Col c;
        const int rb = (int)(rem / n_xg);                      // DIL = 2: (56-row block, parity)
        c.row0 = (rb / DIL) * RB * DIL + rb % DIL;
        c.x0 = (int)(rem - (unsigned)rb * n_xg) * 28 * TPW;
        c.n = flat ? nl : nl * 8 + xcd;
        c.valid = c.n < (unsigned)p.N;
        return c;
    };
    struct Src { const char* a; unsigned v; int prow0; };
    auto src_of = [&](const Col& c) __attribute__((always_inline)) {
        Src q;
        q.a = (const char*)p.x + (long)c.n * xnB;
        int pc = c.x0 + 1 - DIL + dvox;                // padded column of this lane's voxel (logical x0 - DIL + voxel)
        if constexpr (DIL > 1) pc = pc < 0 ? 0 : (pc > W + 1 ? W + 1 : pc);       // outside the stored halo: any zero column
        q.v = dplane + (unsigned)(pc * 16);
        q.prow0 = c.row0 + 1 - DIL;                    // padded row of step 0 = logical row0 - DIL
        return q;
    };
    auto stage = [&](const Src& q, int prow, int slot) __attribute__((always_inline)) {
        char* dst = ring + slot * SLAB;
        prow = prow < 0 ? 0 : (prow > H + 1 ? H + 1 : prow);       // rows beyond the stored halo (dilated taps; a ragged last row block): any zero row
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)q.a, 0, 0x7FFFFF00, 0x00020000);
        const unsigned vo = q.v;        // a local: with the member access as the builtin's operand the host pass drops the kernel's stub (clang 19, ROCm 7.2)
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int pl0 = (wave * NL + i) * PPI;
            const int so = (int)((long)(pl0 >> 3) * cbB + (long)prow * rowB + (long)(pl0 & 7) * (Wp * 16));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(dst + pl0 * CPB), 16, vo, so, 0, 0);
        }
    };
    struct Ctx { char* y16b; const char* resb; unsigned o16; bool ok; int rows; };
    auto ctx_of = [&](const Col& c) __attribute__((always_inline)) {
        Ctx q;
        q.y16b = (char*)p.y16 + (long)c.n * ynB;
        q.resb = p.res ? (const char*)p.res + (long)c.n * ynB : (const char*)p.w;
        q.ok = n_ < 28 && c.x0 + r * 28 + n_ < W;
        q.rows = DIL == 1 ? (H - c.row0 < RB ? H - c.row0 : RB) : RB;          // output rows of this column inside the map
        const long base = (long)ct * cbB + (long)(c.row0 + 1) * rowB + (long)(c.x0 + r * 28 + n_ + 1) * 16;
        if constexpr (KW == 2) q.o16 = (unsigned)(base + (long)(k * 2 + g) * (Wp * 16));
        else q.o16 = (unsigned)(base + (long)((k >> 1) * 2 + g) * (Wp * 16) + (k & 1) * 8);
        return q;
    };
    const unsigned lo_off = (unsigned)(4 * Wp * 16);
    const float relu_lo = p.relu ? 0.f : -65504.f;
    const unsigned nres = p.res ? 0x7FFFFF00u : 0u;

    Col ccur = col_of(0);
    if (!ccur.valid) return;
    S16Ovf og;                                              // range guard (s16_ovf.h)
#pragma unroll
    for (int e = 0; e < OWN; ++e) { og.see_raw(sc[e], 3.0e38f); og.see_raw(sh[e], 3.0e38f); }      // a NaN / Inf folded BN parameter
    Src s_cur = src_of(ccur), s_next = s_cur;
    Ctx cx_cur = ctx_of(ccur), cx_prev = cx_cur;
    cx_prev.ok = false;

    f32x16 acc[3];
    u32x4 resv[3][2];
#pragma unroll
    for (int a_ = 0; a_ < 3; ++a_)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a_][e] = 0.f;
#pragma unroll
    for (int a_ = 0; a_ < 3; ++a_) resv[a_][0] = resv[a_][1] = (u32x4){0u, 0u, 0u, 0u};
    stage(s_cur, s_cur.prow0, 0);
    stage(s_cur, s_cur.prow0 + DIL, 1);
    __builtin_amdgcn_s_waitcnt(S16_WAITCNT(NL, 15));
    unsigned gs = 0;

    // one step: input row t of the column (padded row prow0 + t) into the accumulators of the local output rows t (kh 0), t-1 (kh 1),
    // t-2 (kh 2, complete after this step: published); the row published in the previous step (t-3; t = 0: the previous column's row 27) is
    // finalized.  K0 / K1 / K2: which of the three output rows exist (t <= 27, 1 <= t <= 28, t >= 2).
    auto step = [&](int t, auto JT, auto K0T, auto K1T, auto K2T) __attribute__((always_inline)) {
        constexpr int J = decltype(JT)::value;
        constexpr bool K0 = decltype(K0T)::value, K1 = decltype(K1T)::value, K2 = decltype(K2T)::value;
        constexpr bool COMPUTE = K0 || K1 || K2;
        constexpr int A0 = J, A1 = (J + 2) % 3, A2 = (J + 1) % 3;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(S16_WAITCNT(NL + NS + NR, 0));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool fcur = t >= 3;
        const int qf = fcur ? t - 3 : RB - 1, qp = t - 2;
        const bool p_ok = cx_cur.ok && qp >= 0 && qp < cx_cur.rows;
        if constexpr (RES) {
            const __amdgpu_buffer_rsrc_t resr = __builtin_amdgcn_make_buffer_rsrc((void*)cx_cur.resb, 0, nres, 0x00020000);
            const unsigned po = p_ok ? (unsigned)((long)qp * DIL * rowB) : 0x80000000u;
            if constexpr (KW == 2) {
                resv[J][0] = __builtin_amdgcn_raw_buffer_load_b128(resr, cx_cur.o16 + po, 0, 0);
                resv[J][1] = __builtin_amdgcn_raw_buffer_load_b128(resr, cx_cur.o16 + lo_off + po, 0, 0);
            } else {
                const u32x2 a_ = __builtin_amdgcn_raw_buffer_load_b64(resr, cx_cur.o16 + po, 0, 0);
                const u32x2 b_ = __builtin_amdgcn_raw_buffer_load_b64(resr, cx_cur.o16 + lo_off + po, 0, 0);
                resv[J][0] = (u32x4){a_.x, a_.y, 0u, 0u};
                resv[J][1] = (u32x4){b_.x, b_.y, 0u, 0u};
            }
        }
        {   // row t+2 (of the next column behind this one's last row) into the slot of row t-1: free since the barrier
            const int tp = t + 2;
            const bool nxt = tp >= STEPS;
            Src q;
            q.a = nxt ? s_next.a : s_cur.a;
            q.v = nxt ? s_next.v : s_cur.v;
            int pr = nxt ? s_next.prow0 + (tp - STEPS) * DIL : s_cur.prow0 + tp * DIL;
            stage(q, pr, (J + 2) % 3);
        }
        const __amdgpu_buffer_rsrc_t y16r = __builtin_amdgcn_make_buffer_rsrc(fcur ? cx_cur.y16b : cx_prev.y16b, 0, 0x7FFFFF00, 0x00020000);
        const unsigned f_o16 = fcur ? cx_cur.o16 : cx_prev.o16;
        const bool f_ok = (fcur ? cx_cur.ok : cx_prev.ok && t == 0) && qf < (fcur ? cx_cur.rows : cx_prev.rows);
        f32x4 part[4];
        {
            const char* xb = xchg + ((gs - 1) & 1) * (4 * XW) + (r * KW) * XW + lane * 16;
            if constexpr (KW == 2) {
                part[0] = *(const f32x4*)(xb + (k * 2) * 1024);
                part[1] = *(const f32x4*)(xb + XW + (k * 2) * 1024);
                part[2] = *(const f32x4*)(xb + (k * 2 + 1) * 1024);
                part[3] = *(const f32x4*)(xb + XW + (k * 2 + 1) * 1024);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) part[q] = *(const f32x4*)(xb + q * XW + k * 1024);
            }
        }
        if constexpr (RES) {
            __builtin_amdgcn_s_waitcnt(S16_WAITCNT(2 * NL + NS + NR, 15));
            __builtin_amdgcn_sched_barrier(0);
        }
        _Float16 vh[OWN], vl[OWN];
        const unsigned long long og_keep = S16Ovf::lanes(f_ok);       // idle lanes / rows outside the map hold over-read data
        auto fin = [&](int e) __attribute__((always_inline)) {
            float s_;
            if constexpr (KW == 2) s_ = part[(e >> 2) * 2][e & 3] + part[(e >> 2) * 2 + 1][e & 3];
            else s_ = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
            float x_ = s_ * sc[e] + sh[e];
            if constexpr (RES) {
                const _Float16 rh = __builtin_bit_cast(f16x8, resv[(J + 2) % 3][0])[e], rl_ = __builtin_bit_cast(f16x8, resv[(J + 2) % 3][1])[e];
                x_ += (float)rh + (float)rl_;
            }
            x_ = __builtin_amdgcn_fmed3f(x_, relu_lo, 65504.f);
            og.see(x_, og_keep);
            vh[e] = (_Float16)x_;
            vl[e] = (_Float16)(x_ - (float)vh[e]);
        };
        auto stores = [&]() __attribute__((always_inline)) {
            const unsigned po = f_ok ? (unsigned)((long)qf * DIL * rowB) : 0x80000000u;
            if constexpr (KW == 2) {
                f16x8 hi, lo;
#pragma unroll
                for (int e = 0; e < 8; ++e) { hi[e] = vh[e]; lo[e] = vl[e]; }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), y16r, f_o16 + po, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), y16r, f_o16 + lo_off + po, 0, 0);
            } else {
                f16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) { hi[e] = vh[e]; lo[e] = vl[e]; }
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), y16r, f_o16 + po, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), y16r, f_o16 + lo_off + po, 0, 0);
            }
        };
        auto publish = [&]() __attribute__((always_inline)) {
            const f32x16 a = acc[A2];
            char* xb = xchg + (gs & 1) * (4 * XW) + wave * XW + lane * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) *(f32x4*)(xb + q * 1024) = (f32x4){a[q * 4], a[q * 4 + 1], a[q * 4 + 2], a[q * 4 + 3]};
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[A2][e] = 0.f;
        };
        if constexpr (COMPUTE) {
            const __attribute__((address_space(3))) char* sb = ringl + J * SLAB;
            f16x8 bh[2], bl[2];
            bh[0] = *(lds_frag*)(sb + bfrag[0]);
            bl[0] = *(lds_frag*)(sb + bfrag[0] + 4 * CPB);
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                const int kw = q / KS, j = q - kw * KS;
                if (q + 1 < NG) {
                    const int kw1 = (q + 1) / KS, j1 = (q + 1) - kw1 * KS;
                    bh[(q + 1) & 1] = *(lds_frag*)(sb + bfrag[j1] + kw1 * DIL * 16);
                    bl[(q + 1) & 1] = *(lds_frag*)(sb + bfrag[j1] + 4 * CPB + kw1 * DIL * 16);
                }
                const f16x8 h_ = bh[q & 1], l_ = bl[q & 1];
                const f16x8 w0h = wh[j][kw], w1h = wh[j][3 + kw], w2h = wh[j][6 + kw];
                const f16x8 w0l = wl[j][kw], w1l = wl[j][3 + kw], w2l = wl[j][6 + kw];
                if (q < NG - 1) {
                    if (K0) acc[A0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h, h_, acc[A0], 0, 0, 0);
                    if (K1) acc[A1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h, h_, acc[A1], 0, 0, 0);
                    if (K2) acc[A2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2h, h_, acc[A2], 0, 0, 0);
                    if (K0) acc[A0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h, l_, acc[A0], 0, 0, 0);
                    if (K1) acc[A1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h, l_, acc[A1], 0, 0, 0);
                    if (K2) acc[A2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2h, l_, acc[A2], 0, 0, 0);
                    if (K0) acc[A0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0l, h_, acc[A0], 0, 0, 0);
                    if (K1) acc[A1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1l, h_, acc[A1], 0, 0, 0);
                    if (K2) acc[A2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2l, h_, acc[A2], 0, 0, 0);
#pragma unroll
                    for (int e = q * FPG; e < (q + 1) * FPG && e < OWN; ++e) fin(e);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    stores();
                    if (K2) acc[A2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2h, h_, acc[A2], 0, 0, 0);
                    if (K0) acc[A0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h, h_, acc[A0], 0, 0, 0);
                    if (K2) acc[A2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2h, l_, acc[A2], 0, 0, 0);
                    if (K1) acc[A1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h, h_, acc[A1], 0, 0, 0);
                    if (K2) acc[A2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2l, h_, acc[A2], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (K0) acc[A0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h, l_, acc[A0], 0, 0, 0);
                    if (K1) acc[A1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h, l_, acc[A1], 0, 0, 0);
                    if (K0) acc[A0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0l, h_, acc[A0], 0, 0, 0);
                    if (K1) acc[A1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1l, h_, acc[A1], 0, 0, 0);
                    publish();
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < OWN; ++e) fin(e);
            stores();
            publish();
        }
        ++gs;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using T_ = std::true_type;
    using F_ = std::false_type;
#pragma unroll 1
    for (unsigned it = 0;; ++it) {
        const Col cnext = col_of(it + 1);
        s_next = cnext.valid ? src_of(cnext) : s_cur;
        step(0, I0{}, T_{}, F_{}, F_{});
        step(1, I1{}, T_{}, T_{}, F_{});
        step(2, I2{}, T_{}, T_{}, T_{});
#pragma unroll 1
        for (int t0 = 3; t0 < STEPS - 3; t0 += 3) {
            step(t0, I0{}, T_{}, T_{}, T_{});
            step(t0 + 1, I1{}, T_{}, T_{}, T_{});
            step(t0 + 2, I2{}, T_{}, T_{}, T_{});
        }
        step(STEPS - 3, I0{}, T_{}, T_{}, T_{});
        step(STEPS - 2, I1{}, F_{}, T_{}, T_{});
        step(STEPS - 1, I2{}, F_{}, F_{}, T_{});
        if (!cnext.valid) break;
        cx_prev = cx_cur;
        cx_cur = ctx_of(cnext);
        s_cur = s_next;
    }
    step(STEPS, I0{}, F_{}, F_{}, F_{});       // drain: the last row (published in the step before) is finalized
    og.flush(p.ovf);
}

template <int KW, int KS, bool RES, int DIL = 1>
int launch2(const drc_s16conv_params& p, hipStream_t stream) {
    constexpr int TPW = 4 / KW;
    constexpr int PV = TPW == 2 ? 64 : 32;
    constexpr int SLAB = (KW * KS / 2) * 8 * PV * 16;
    constexpr size_t lds = RING * SLAB + 2 * 4 * XW;
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)convs16r_kernel<KW, KS, RES, DIL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const long columns = DIL == 1 ? (long)p.N * ((p.H + RB - 1) / RB) * ((p.W + 28 * TPW - 1) / (28 * TPW)) : (long)p.N * (p.H / RB) * (p.W / (28 * TPW));
    const int n_ct = p.cout / 32;
    // persistent grid: two workgroups per CU where the registers allow (KS == 1: one hides the other's barriers and waits), else one;
    // 8 XCDs x wpx column workers x n_ct cout tiles (the kernel derives wpx = gridDim / 8 / n_ct: it must divide), halved while there are
    // twice as many workers as columns
    long wpx = ((KS == 1 && !(p.lo4 & 4)) ? 512 : 256) / (8 * n_ct);
    if (wpx < 1) wpx = 1;
    while (wpx > 1 && 4 * wpx >= columns) wpx /= 2;
    const long blocks = 8 * n_ct * wpx;
    hipLaunchKernelGGL((convs16r_kernel<KW, KS, RES, DIL>), dim3((unsigned)blocks), dim3(256), lds, stream, p);
    return (int)hipGetLastError();
}

template <int KW, int KS, int DIL = 1>
int launch(const drc_s16conv_params& p, hipStream_t stream) {
    return p.res ? launch2<KW, KS, true, DIL>(p, stream) : launch2<KW, KS, false, DIL>(p, stream);
}

}  // namespace

extern "C" int drc_conv2d_k3_s16_supported(int cin, int cout, int H, int W, int dil) {
    if (dil == 2) return cin == 128 && (cout == 32 || cout == 64 || cout == 128) && H > 0 && H % (2 * RB) == 0 && W > 0 && W % 28 == 0;
    if (dil != 1) return 0;
    if (cin != 32 && cin != 64 && cin != 128) return 0;            // (a wider layer runs as chained launches over 128-channel input slices: the
    if (cout != 32 && cout != 64 && cout != 128 && cout != 256 && cout != 512) return 0;      //  previous partial sum is the next launch's residual)
    if (H <= 0 || W <= 0) return 0;
    return 1;
}

extern "C" int drc_conv2d_k3_s16_fwd(const drc_s16conv_params* pp, void* stream) {
    if (!pp) return -1;
    const drc_s16conv_params& p = *pp;
    if (!p.x || !p.w || !p.scale || !p.shift || !p.y16 || p.y32 || p.left || p.right) return -1;
    if (p.N < 0) return -2;
    const int dil = p.dil ? p.dil : 1;
    if (!drc_conv2d_k3_s16_supported(p.cin, p.cout, p.H, p.W, dil)) return -4;
    if (p.N == 0) return 0;
    const long unit16 = (long)((p.cin > p.cout ? p.cin : p.cout) / 32) * (p.H + 2) * (p.W + 2) * 128;
    if (unit16 >= 0x7FFFFF00L / 2) return -5;
    hipStream_t s = (hipStream_t)stream;
    if (dil == 2) return launch<4, 2, 2>(p, s);
    if (p.cin == 32) return launch<2, 1>(p, s);
    if (p.cin == 128) return launch<4, 2>(p, s);
    // 64 input channels: one tile per workgroup and K over the four waves (27 MFMAs per step, twice the workgroups), or two tiles with two K
    // slices per wave (54 MFMAs per step).  Measured (tools/experiments/exp_s16_2d.py, 56 x 56 maps): 64 -> 64: 29.0 vs 42.8 us at 32 images,
    // 114 vs 130 us at 128; 64 -> 128: 50.2 vs 49.3 us at 32, 232 vs 202 us at 128.  lo4 = 1 / 2 force one form (experiments).
    const bool wide = p.W % 56 == 0;
    const long cols_wide = (long)p.N * (p.H / RB) * (p.W / 56) * (p.cout / 32);
    const bool use_wide = (p.lo4 & 3) == 1 ? false : ((p.lo4 & 3) == 2 ? wide : wide && p.cout == 128 && cols_wide >= 256);
    return use_wide ? launch<2, 2>(p, s) : launch<4, 1>(p, s);
}
