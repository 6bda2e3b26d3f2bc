// convs16w.hip -- the cost-volume layer (dres0[0]) of convs16.hip with TWO MFMA tiles per wave (round 6).
//
//   reference arithmetic: the concat cost volume of stackhourglass.py:115-128 folded into convbn_3d 64 -> 32 k3 s1 p1 + ReLU (dres0[0],
//   :63-66,130); fp32 (config/defaults.py:22).  Arithmetic, RS16 layout, K split, depth walk, exchange and counted waits: convs16.hip (read
//   its header first); results are bit-identical to it (same products, same summation order per output value).
//
// Why.  convs16.hip's workgroup owns 4/KW image rows of a column; per step (one input plane) a wave runs 81 MFMAs between two barriers, and
// the slab it shares is (rows + 2) x 30 voxels for `rows` output rows: two halo rows per two (KW = 2) or per ONE (KW = 4: the cost-volume
// layer) output row.  Round 5's ablation priced the barrier at ~9 % and the staging at ~16 % of the plain layer.  Here a wave owns TWO
// vertically adjacent tiles (two accumulator sets: 96 registers next to the 216 weight registers), so a workgroup owns 8/KW rows: 162 MFMAs
// per wave between barriers, the slab (rows + 2) x 30 for twice the rows -- staged rows per output row 2.0 -> 1.5 (KW = 2), 3.0 -> 2.0 (KW = 4).
// Measured (tools/experiments/exp_s16_wide.py, interleaved rounds, bit-identical outputs, us per launch, two tiles | one tile): the
// cost-volume layer 2438 | 2592 at 1024 Config-A ROIs, 607 | 661 at 256, 1214 | 1295 at 64 Config-B ROIs (6-8 %: it stages three rows per
// output row with one tile per wave pair); the plain 32 -> 32 layer 1277 | 1278 -- no gain (two rows per two already; the barrier it saves
// was hidden), so only the cost-volume form is instantiated.  The residual and fused-head forms have no registers left for a second
// accumulator set (498 / 502 of 512) and stay on convs16.hip like the plain one.
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
constexpr int TPW = 2;          // tiles per wave
constexpr int TX = 28, SX = 30; // output / staged columns of a tile
constexpr int NPMAX = 3;        // LDS-DMA pieces per chunk plane, at most (array bounds inside the lambdas must not depend on a function-local constant:
                                // with `unsigned va[NP]` the HOST pass silently dropped the kernel's stub -- undefined symbol at load time)

template <int KW>
struct WGeom {
    static constexpr int RPW = 4 / KW;                        // wave pairs (K-split groups) per workgroup
    static constexpr int ROWS = RPW * TPW;                    // output rows per workgroup
    static constexpr int SROWS = ROWS + 2;
    static constexpr int PV = (SROWS * SX + 4 + 63) / 64 * 64;   // voxels per chunk plane (incl. the columns lanes 28..31 over-read)
    static constexpr int NP = PV / 64;                        // LDS-DMA pieces per chunk plane
    static constexpr int CPB = PV * 16;
    static constexpr int CBI = KW / 2;
    static constexpr int SLAB = CBI * 8 * CPB;
    static constexpr int NL = CBI * 8 * NP / 4;               // LDS-DMA instructions per wave and slab
    static constexpr int XW = 4096;
    static constexpr size_t LDS = (size_t)RING * SLAB + 2 * 4 * TPW * XW;
};

template <int KW, bool CV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void convs16w_kernel(const drc_s16conv_params p) {
    using G = WGeom<KW>;
    constexpr int ROWS = G::ROWS, SROWS = G::SROWS, NP = G::NP, CPB = G::CPB, CBI = G::CBI, SLAB = G::SLAB, NL = G::NL, XW = G::XW;
    constexpr int OWN = 16 / KW;                // accumulator registers (couts per lane) a wave finishes per tile
    constexpr int NS = 2 * TPW;                 // stores per step (RS16 hi, lo per tile)
    static_assert(CV ? KW == 4 : KW == 2, "forms: the cost-volume layer (instantiated); the plain 32-channel layer (measured: no gain)");
    static_assert((CBI * 8 * NP) % 4 == 0, "DMA split over the four waves");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* ring = lds;
    char* xchg = lds + RING * SLAB;             // [2 parities][RPW * TPW tiles][KW slices][XW]

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int xl = lane & 31, g = lane >> 5;
    const bool lane_x = xl < TX;
    const int r = wave / KW, k = wave % KW;     // wave pair (its two tiles: rows r*2, r*2+1 of the workgroup), K slice
    const int n_ct = p.cout / 32;
    const int ct = (int)((blockIdx.x >> 3) % n_ct);

    const int D = p.D, H = p.H, W = p.W;
    const int Wp = W + 2, Hp = H + 2;
    const long rowB = (long)Wp * 128;
    const long planeB = (long)Hp * rowB;
    const long xcbB = (long)(D + 2) * planeB;
    const long xnB = (long)CBI * xcbB;
    const int cbo = p.cout / 32;
    const long ynB = (long)cbo * xcbB;
    const long mapnB = planeB;

    f16x8 wh[27], wl[27];
    {
        const char* wb = (const char*)p.w + ((long)(ct * KW + k) * 54) * 1024 + lane * 16;
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            wh[t] = *(const f16x8*)(wb + (t * 2) * 1024);
            wl[t] = *(const f16x8*)(wb + (t * 2 + 1) * 1024);
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
    // LDS-DMA geometry: piece h of a chunk plane holds staged voxels h*64 + lane -> (row, column) of the slab
    int srcrow[NPMAX], srcx[NPMAX];
    bool srcok[NPMAX];
    static_assert(NP <= NPMAX, "pieces");
#pragma unroll
    for (int h = 0; h < NP; ++h) {
        const int v = h * 64 + lane;
        srcok[h] = v < SROWS * SX;
        const int rr = srcok[h] ? v / SX : 0;
        srcrow[h] = rr;
        srcx[h] = srcok[h] ? v - rr * SX : 0;
    }
    // B fragment of tile j: + j * SX * 16
    const unsigned bfrag = (unsigned)(((k >> 1) * 8 + (k & 1) * 2 + g) * CPB + ((r * TPW) * SX + xl) * 16);     // hi; lo at + 4*CPB
    const __attribute__((address_space(3))) char* ringl = (const __attribute__((address_space(3))) char*)ring;
    typedef const __attribute__((address_space(3))) f16x8 lds_frag;

    const int n_xt = (W + TX - 1) / TX, n_yt = (H + ROWS - 1) / ROWS;
    const int Dw = (D + 2) / 3 * 3;             // depth walk padded with phantom zero planes (convs16.hip)
    const unsigned xcd = blockIdx.x & 7, qx = (blockIdx.x >> 3) / n_ct, per_xcd = (gridDim.x >> 3) / n_ct;
    const unsigned cols_unit = (unsigned)n_yt * n_xt;

    struct Col { unsigned n; int y0, x0; bool valid; };
    auto col_of = [&](unsigned it) __attribute__((always_inline)) {
        const unsigned j = it * per_xcd + qx;
        const unsigned nl = j / cols_unit;
        const unsigned rem = j - nl * cols_unit;
        const int yb = (int)(rem / n_xt), xt = (int)(rem - (unsigned)yb * n_xt);
        Col c;
        c.n = nl * 8 + xcd;
        c.valid = c.n < (unsigned)p.N;
        c.y0 = yb * ROWS;
        c.x0 = xt * TX;
        return c;
    };
    struct Src { const char* a; const char* b; unsigned v[NPMAX]; int x0; };
    auto src_of = [&](const Col& c) __attribute__((always_inline)) {
        Src q;
        if constexpr (CV) {
            q.a = (const char*)p.left + (long)c.n * mapnB;
            q.b = (const char*)p.right + (long)c.n * mapnB;
#pragma unroll
            for (int h = 0; h < NP; ++h) q.v[h] = (unsigned)((long)(c.y0 + srcrow[h]) * rowB);
        } else {
            q.a = (const char*)p.x + (long)c.n * xnB;
            q.b = nullptr;
#pragma unroll
            for (int h = 0; h < NP; ++h) q.v[h] = (unsigned)((long)(c.y0 + srcrow[h]) * rowB + (long)(c.x0 + srcx[h]) * 16);
        }
        q.x0 = c.x0;
        return q;
    };
    auto stage = [&](const Src& q, int pl, int slot) __attribute__((always_inline)) {
        char* dst = ring + slot * SLAB;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)q.a, 0, 0x7FFFFF00, 0x00020000);
        unsigned va[NPMAX], vb[NPMAX];
#pragma unroll
        for (int h = 0; h < NP; ++h) va[h] = vb[h] = q.v[h];
        if constexpr (CV) {
            const int ish = p.lo4 + pl;
            const bool real = pl < D;
#pragma unroll
            for (int h = 0; h < NP; ++h) {
                const int xlog = q.x0 + srcx[h] - 1;
                const bool ok = real && srcok[h] && xlog >= 0 && xlog < W && xlog - ish >= 0 && xlog - ish < W;
                va[h] += ok ? (unsigned)((q.x0 + srcx[h]) * 16) : 0u;                  // column 0 = the zero halo
                vb[h] += ok ? (unsigned)((q.x0 + srcx[h] - ish) * 16) : 0u;
            }
        }
        const __amdgpu_buffer_rsrc_t rb = CV ? __builtin_amdgcn_make_buffer_rsrc((void*)q.b, 0, 0x7FFFFF00, 0x00020000) : ra;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int id = wave * NL + i;
            const int cc = id / NP, h = id - cc * NP;       // chunk plane (cb * 8 + c), piece
            const int cb = cc >> 3, c = cc & 7;
            if constexpr (CV) {
                const int so = c * (Wp * 16);
                if (cb) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(dst + cc * CPB + h * 1024), 16, vb[h], so, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(dst + cc * CPB + h * 1024), 16, va[h], so, 0, 0);
            } else {
                const int so = (int)((long)cb * xcbB + (long)(pl + 1) * planeB + (long)c * (Wp * 16));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(dst + cc * CPB + h * 1024), 16, va[h], so, 0, 0);
            }
        }
    };
    // output context of a column: unit base + this lane's voxel offset of tile 0 (plane 0, padded coordinates + 1); tile 1: + rowB
    struct Ctx { char* y16b; unsigned o16; bool ok[TPW]; };
    auto ctx_of = [&](const Col& c) __attribute__((always_inline)) {
        Ctx q;
        q.y16b = (char*)p.y16 + (long)c.n * ynB;
        const int yl = c.y0 + r * TPW;
#pragma unroll
        for (int j = 0; j < TPW; ++j) q.ok[j] = lane_x && yl + j < H && c.x0 + xl < W;
        if constexpr (KW == 2)
            q.o16 = (unsigned)((long)ct * xcbB + planeB + (long)(yl + 1) * rowB + (long)(k * 2 + g) * (Wp * 16) + (long)(c.x0 + xl + 1) * 16);
        else
            q.o16 = (unsigned)((long)ct * xcbB + planeB + (long)(yl + 1) * rowB + (long)((k >> 1) * 2 + g) * (Wp * 16) + (long)(c.x0 + xl + 1) * 16 + (k & 1) * 8);
        return q;
    };
    const unsigned lo_off = (unsigned)(4 * Wp * 16);
    const float relu_lo = p.relu ? 0.f : -65504.f;

    Col ccur = col_of(0);
    if (!ccur.valid) return;
    S16Ovf og;
#pragma unroll
    for (int e = 0; e < OWN; ++e) { og.see_raw(sc[e], 3.0e38f); og.see_raw(sh[e], 3.0e38f); }
    Src s_cur = src_of(ccur), s_next = s_cur;
    Ctx cx_cur = ctx_of(ccur), cx_prev = cx_cur;
#pragma unroll
    for (int j = 0; j < TPW; ++j) cx_prev.ok[j] = false;

    f32x16 acc[TPW][3];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int a_ = 0; a_ < 3; ++a_)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][a_][e] = 0.f;
    stage(s_cur, 0, 0);
    stage(s_cur, 1 < D ? 1 : D, 1);
    __builtin_amdgcn_s_waitcnt(S16_WAITCNT(NL, 15));
    unsigned gs = 0;

    // one step (input plane t): the structure of convs16.hip's step, the tap groups run once per tile
    auto step = [&](int t, auto JT, auto KT) __attribute__((always_inline)) {
        constexpr int J = decltype(JT)::value;
        constexpr int KIND = decltype(KT)::value;
        constexpr bool COMPUTE = KIND != 3;
        constexpr int A0 = (J + 1) % 3, A1 = J, A2 = (J + 2) % 3;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(S16_WAITCNT(NL + NS, 0));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool fcur = t >= 2;
        const int qf = fcur ? t - 2 : Dw - 2 + t;
        {   // slab t+2 (of the next column behind this one's last plane) into the slot of plane t-1: free since the barrier
            const int tp = t + 2;
            const bool nxt = tp >= Dw;
            Src q;
            q.a = nxt ? s_next.a : s_cur.a;
            q.b = nxt ? s_next.b : s_cur.b;
            q.x0 = nxt ? s_next.x0 : s_cur.x0;
#pragma unroll
            for (int h = 0; h < NP; ++h) q.v[h] = nxt ? s_next.v[h] : s_cur.v[h];
            int pl = nxt ? tp - Dw : tp;
            pl = pl < D ? pl : D;
            stage(q, pl, (J + 2) % 3);
        }
        const __amdgpu_buffer_rsrc_t y16r = __builtin_amdgcn_make_buffer_rsrc(fcur ? cx_cur.y16b : cx_prev.y16b, 0, 0x7FFFFF00, 0x00020000);
        const unsigned f_o16 = fcur ? cx_cur.o16 : cx_prev.o16;
        const bool plane_ok = qf >= 0 && qf < D;
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            const bool f_ok = (fcur ? cx_cur.ok[j] : cx_prev.ok[j]) && plane_ok;
            // the K-split partial sums of this tile's plane to finalize (published in the previous step)
            f32x4 part[4];
            {
                const char* xb = xchg + ((gs - 1) & 1) * (4 * TPW * XW) + ((r * TPW + j) * KW) * XW + lane * 16;
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
            _Float16 vh[OWN], vl[OWN];
            const unsigned long long og_keep = S16Ovf::lanes(f_ok);
            auto fin = [&](int e) __attribute__((always_inline)) {
                float s_;
                if constexpr (KW == 2) s_ = part[(e >> 2) * 2][e & 3] + part[(e >> 2) * 2 + 1][e & 3];
                else s_ = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
                float x_ = s_ * sc[e] + sh[e];
                x_ = __builtin_amdgcn_fmed3f(x_, relu_lo, 65504.f);
                og.see(x_, og_keep);
                vh[e] = (_Float16)x_;
                vl[e] = (_Float16)(x_ - (float)vh[e]);
            };
            auto stores = [&]() __attribute__((always_inline)) {
                const unsigned po = f_ok ? (unsigned)((long)qf * planeB + (long)j * rowB) : 0x80000000u;
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
                const f32x16 a = acc[j][A2];
                char* xb = xchg + (gs & 1) * (4 * TPW * XW) + ((r * TPW + j) * KW + k) * XW + lane * 16;
#pragma unroll
                for (int q = 0; q < 4; ++q) *(f32x4*)(xb + q * 1024) = (f32x4){a[q * 4], a[q * 4 + 1], a[q * 4 + 2], a[q * 4 + 3]};
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][A2][e] = 0.f;
            };
            if constexpr (COMPUTE) {
                constexpr bool K0 = KIND != 2, K2 = KIND != 1;
                const __attribute__((address_space(3))) char* sb = ringl + J * SLAB + bfrag + j * (SX * 16);
                f16x8 bh[2], bl[2];
                bh[0] = *(lds_frag*)(sb);
                bl[0] = *(lds_frag*)(sb + 4 * CPB);
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    const int kh = q / 3, kw = q - kh * 3;
                    if (q + 1 < 9) {
                        const int kh1 = (q + 1) / 3, kw1 = (q + 1) - kh1 * 3;
                        bh[(q + 1) & 1] = *(lds_frag*)(sb + (kh1 * SX + kw1) * 16);
                        bl[(q + 1) & 1] = *(lds_frag*)(sb + 4 * CPB + (kh1 * SX + kw1) * 16);
                    }
                    const f16x8 h_ = bh[q & 1], l_ = bl[q & 1];
                    const int t0 = kh * 3 + kw, t1 = 9 + t0, t2 = 18 + t0;
                    if (q == OWN) stores();
                    if (q < 8) {
                        if (K0) acc[j][A0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t0], h_, acc[j][A0], 0, 0, 0);
                        acc[j][A1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t1], h_, acc[j][A1], 0, 0, 0);
                        if (K2) acc[j][A2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t2], h_, acc[j][A2], 0, 0, 0);
                        if (K0) acc[j][A0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t0], l_, acc[j][A0], 0, 0, 0);
                        acc[j][A1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t1], l_, acc[j][A1], 0, 0, 0);
                        if (K2) acc[j][A2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t2], l_, acc[j][A2], 0, 0, 0);
                        if (K0) acc[j][A0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[t0], h_, acc[j][A0], 0, 0, 0);
                        acc[j][A1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[t1], h_, acc[j][A1], 0, 0, 0);
                        if (K2) acc[j][A2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[t2], h_, acc[j][A2], 0, 0, 0);
                        if (q < OWN) fin(q);
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        if (K2) acc[j][A2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t2], h_, acc[j][A2], 0, 0, 0);
                        if (K0) acc[j][A0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t0], h_, acc[j][A0], 0, 0, 0);
                        if (K2) acc[j][A2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t2], l_, acc[j][A2], 0, 0, 0);
                        acc[j][A1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t1], h_, acc[j][A1], 0, 0, 0);
                        if (K2) acc[j][A2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[t2], h_, acc[j][A2], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (K0) acc[j][A0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t0], l_, acc[j][A0], 0, 0, 0);
                        acc[j][A1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t1], l_, acc[j][A1], 0, 0, 0);
                        if (K0) acc[j][A0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[t0], h_, acc[j][A0], 0, 0, 0);
                        acc[j][A1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[t1], h_, acc[j][A1], 0, 0, 0);
                        publish();
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < OWN; ++e) fin(e);
                stores();
                publish();
            }
        }
        ++gs;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
#pragma unroll 1
    for (unsigned it = 0;; ++it) {
        const Col cnext = col_of(it + 1);
        s_next = cnext.valid ? src_of(cnext) : s_cur;
        step(0, I0{}, I1{});
        step(1, I1{}, I0{});
        if (Dw == 3) {
            step(2, I2{}, I2{});
        } else {
            step(2, I2{}, I0{});
#pragma unroll 1
            for (int t0 = 3; t0 + 3 < Dw; t0 += 3) {
                step(t0, I0{}, I0{});
                step(t0 + 1, I1{}, I0{});
                step(t0 + 2, I2{}, I0{});
            }
            step(Dw - 3, I0{}, I0{});
            step(Dw - 2, I1{}, I0{});
            step(Dw - 1, I2{}, I2{});
        }
        if (!cnext.valid) break;
        cx_prev = cx_cur;
        cx_cur = ctx_of(cnext);
        s_cur = s_next;
    }
    step(Dw, I0{}, I3{});
    step(Dw + 1, I1{}, I3{});
    og.flush(p.ovf);
}

template <int KW, bool CV>
int launch(const drc_s16conv_params& p, hipStream_t stream) {
    using G = WGeom<KW>;
    static_assert(G::LDS <= 160 * 1024, "LDS");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)convs16w_kernel<KW, CV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const long columns = (long)p.N * ((p.H + G::ROWS - 1) / G::ROWS) * ((p.W + TX - 1) / TX);
    const int n_ct = p.cout / 32;
    long blocks = 256;
    while (blocks > 8 * n_ct && blocks / (2 * n_ct) >= columns) blocks /= 2;
    hipLaunchKernelGGL((convs16w_kernel<KW, CV>), dim3((unsigned)blocks), dim3(256), G::LDS, stream, p);
    return (int)hipGetLastError();
}

}  // namespace

// 1 when drc_conv3d_k3_s16_fwd sends this parameter block to the two-tiles-per-wave kernel: the cost-volume layer on 1 x 28 tiles with whole
// two-row blocks and enough of them to give every CU several columns (smaller launches keep convs16.hip's finer columns).  Host code: no launch.
extern "C" int drc_conv3d_k3_s16_wide(const drc_s16conv_params* pp) {
    if (!pp) return 0;
    const drc_s16conv_params& p = *pp;
    const bool cv = p.left || p.right;
    if (!cv || p.res || p.y32 || p.head || !p.y16 || p.W <= 14 || p.cout != 32 || p.cin != 64) return 0;
    if (p.dil & 0x800) return 0;                              // experiment bit (`dil` is otherwise unused by the 3D layers): force the one-tile kernel (A/B runs)
    if (p.H % 2) return 0;                                     // (whole row blocks only: a half-empty wave would give the gain back)
    const long columns = (long)p.N * (p.H / 2) * ((p.W + TX - 1) / TX);
    return columns >= 4 * 256;
}

extern "C" int drc_conv3d_k3_s16_wide_fwd(const drc_s16conv_params* pp, void* stream) {
    if (!pp) return -1;
    const drc_s16conv_params& p = *pp;
    const bool cv = p.left || p.right;
    if (!p.w || !p.scale || !p.shift || !p.y16) return -1;
    if (!cv || !p.left || !p.right || p.cin != 64) return -4;
    if (p.res || p.y32 || p.head || p.W <= 14 || p.cout != 32 || p.N < 0 || p.D <= 0 || p.H <= 0) return -4;
    if (p.N == 0) return 0;
    const long unit16 = (long)(p.cout / 32) * (p.D + 2) * (p.H + 2) * (p.W + 2) * 128;
    if (unit16 >= 0x7FFFFF00L / 2) return -5;
    return launch<4, true>(p, (hipStream_t)stream);
}
