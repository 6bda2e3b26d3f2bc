// wino3d_pc.hip -- stride-1 3x3x3 convolution (+BN, +residual, +ReLU) as Winograd F(2x2x2, 3x3x3) on the fp32 matrix cores
// (gfx950 / CDNA4), producer / consumer form.
//
//   reference: dres0/dres1, classifN[0], hourglass conv2/conv4 (stackhourglass.py:63-88, :14-20)
//
//   Y = A^T [ (G g G^T) . (B^T d B) ] A   in each of the three dimensions, with
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1]:
// a 2x2x2 output tile costs 64 multiplies per (cin, cout) pair instead of 216.
//
// Round 1's kernel (one wave per SIMD doing everything) sat at 0.37 of the MFMA peak: input loads, three butterflies, weight
// staging, barriers and the inverse transform all serialise with the MFMAs in an in-order wave.  Here a block is EIGHT
// waves, two per SIMD, with separate roles:
//   * consumers (waves 0-3, one per SIMD): nothing but `ds_read_b128` + MFMA.  A consumer owns 16 tiles (the N dimension of
//     the 16x16x4 MFMA) x CT*16 couts.  Per half step (8 of the 64 frequency points of one 16-channel block) it reads its B
//     fragments and the transformed weights from LDS and issues 32*CT MFMAs; after the last channel block of a depth
//     frequency it inverse-transforms its 16 x CT accumulators in-plane (4x4 -> 2x2) and folds them along depth incrementally
//     (P0 = z0 + z1 + z2 in LDS, P1 = z1 - z2 - z3 in registers), so the two output slices leave after frequencies 2 and 3.
//   * producers (waves 4-7, the second wave on each SIMD): global loads of the 4x4x4 input patches (lane (tile j, g) owns
//     channels 4g..4g+3 -- one float4 per patch voxel in the blocked layout), the depth / w / h butterflies on the VALU (which
//     runs beside the partner's MFMAs), B fragments -> LDS, and the weight slab of the next half step -> LDS.
// One s_barrier per half step hands a double-buffered LDS slot (B fragments + weights) from producers to consumers: the
// producers fill slot (h+1)&1 while the consumers run half step h out of slot h&1.
// Needs even OD, OH, OW (the engine sends other shapes to tapdirect).  Results differ from the direct kernels' by fp32
// rounding only.  Weights: drc_pack_weights_wino, [xi][cb][cout tile][g*16 + j][4] (lane order, conflict-free ds_read_b128).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/disprcnn_hip.h"   // (tools/experiments is two levels below the repo root, like disprcnn_amd/csrc)

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WN_CONS 4

namespace {

template <int CT>
__global__ __launch_bounds__(512) void wino3d_kernel(const drc_tapconv_params p) {
    __shared__ __attribute__((aligned(16))) f32x4 b_lds[2][WN_CONS][8][64];        // B fragments of a half step, per consumer
    __shared__ __attribute__((aligned(16))) f32x4 w_lds[2][8][CT][64];            // transformed weights of a half step
    __shared__ __attribute__((aligned(16))) f32x4 z_lds[WN_CONS][4 * CT][64];     // P0 = z0 (+ z1) per consumer
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const int g = lane >> 4;
    const int cw = wave & (WN_CONS - 1);          // the tile-group slot this wave serves (consumer cw <-> producer cw + 4)

    const drc_tap_class cls = p.cls[0];
    const int TD = p.OD >> 1, TH = p.OH >> 1, TW = p.OW >> 1;
    const int tiles = p.N * TD * TH * TW;
    const int groups = (tiles + 15) >> 4;
    // Block -> (cout group, position); round r of position pos takes tile groups 4*(r*nbk + pos) .. +3.  Positions are numbered
    // XCD by XCD (workgroups are dealt round-robin over the 8 XCDs) so that the blocks sharing an L2 sweep adjacent tiles.
    const int n_cg = p.cout_pad / 16 / CT;
    int cg, pos;
    const int nbk = (int)gridDim.x / n_cg;
    if (gridDim.x % (8 * n_cg) == 0) {
        const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
        cg = l % n_cg;
        pos = xcd * (nbk / 8) + l / n_cg;
    } else {
        cg = blockIdx.x % n_cg;
        pos = blockIdx.x / n_cg;
    }
    const int chunks = (groups + WN_CONS - 1) / WN_CONS;
    const int rounds = (chunks + nbk - 1) / nbk;
    const int ct0 = cg * CT;
    const int steps = rounds * 4 * p.cb_in;        // (round, depth frequency, channel block); two half steps each

    struct Geo { unsigned xo; int n, dt, ht, wt; bool valid; };
    auto geo_of = [&](int round) __attribute__((always_inline)) {
        Geo q;
        int grp = (round * nbk + pos) * WN_CONS + cw;
        const bool active = grp < groups;
        if (!active) grp = groups - 1;
        int tile = grp * 16 + j;
        q.valid = active && tile < tiles;
        if (tile >= tiles) tile = tiles - 1;
        q.wt = tile % TW; tile /= TW;
        q.ht = tile % TH; tile /= TH;
        q.dt = tile % TD;
        q.n = tile / TD;
        q.xo = (unsigned)((q.n * p.x_n_stride + (2 * q.dt + cls.dd0) * p.x_d_stride + (2 * q.ht + cls.dh0) * p.x_h_stride +
                           (int64_t)(2 * q.wt + cls.dw0) * 16 + g * 4) * 4);
        return q;
    };

    if (wave >= WN_CONS) {
        // =========================================================================================== producer
        const int tp = (int)threadIdx.x - 64 * WN_CONS;           // 0..255 among the producers
        struct Cursor { int round, xd, cb; };
        auto advance = [&](Cursor c) __attribute__((always_inline)) {
            if (++c.cb == p.cb_in) { c.cb = 0; if (++c.xd == 4) { c.xd = 0; ++c.round; } }
            return c;
        };
        // depth butterfly of frequency xd: slice a + sgn * slice b  (d0-d2, d1+d2, d2-d1, d1-d3)
        auto slice_a = [](int xd) { return xd == 0 ? 0 : (xd == 2 ? 2 : 1); };
        auto slice_b = [](int xd) { return xd == 2 ? 1 : (xd == 3 ? 3 : 2); };
        f32x4 ra[4][4], rb[4][4];                                  // the raw rows of the step being loaded (two slices)
        // rows h0, h0+1 of a step's two slices: SGPR slice base + per-lane patch offset + immediate
        auto issue_loads = [&](const Cursor& c, unsigned xo, int h0) __attribute__((always_inline)) {
#ifdef WN_COAL
            const unsigned xo_other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)xo, 0x128, 0xf, 0xf, false);     // row_ror:8 -> tile j ^ 8
            const uint64_t xo2 = j < 8 ? ((uint64_t)xo_other << 32 | xo) : ((uint64_t)(xo + 64) << 32 | (xo_other + 64));
#endif
            const char* sa = (const char*)(p.x + (int64_t)c.cb * p.x_cb_stride + (int64_t)slice_a(c.xd) * p.x_d_stride);
            const char* sb = (const char*)(p.x + (int64_t)c.cb * p.x_cb_stride + (int64_t)slice_b(c.xd) * p.x_d_stride);
#pragma unroll
            for (int h = h0; h < h0 + 2; ++h)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
#ifdef WN_ABL_NOLOAD
                    ra[h][w] = (f32x4){(float)xo, 1.f, 2.f, (float)(size_t)sa}; rb[h][w] = (f32x4){(float)h, (float)w, 2.f, (float)(size_t)sb};
#elif defined(WN_ABL_HALFLOAD)
                    if (w < 2) {
                        ra[h][w] = *(const f32x4*)(sa + ((int64_t)h * p.x_h_stride + w * 16) * 4 + xo);
                        rb[h][w] = *(const f32x4*)(sb + ((int64_t)h * p.x_h_stride + w * 16) * 4 + xo);
                    }
#elif defined(WN_COAL)
                    // whole 128-byte lines per instruction: k = 2*pair + half-wave role.  Instruction (pair, A) reads voxels w = 2*pair
                    // (lanes j < 8, tile j) and w = 2*pair + 1 (lanes j >= 8, tile j - 8): 8 tiles x one full line; (pair, B) the other
                    // 8 tiles.  `xo` carries xoA in its low and xoB in its high 32 bits.
                    const unsigned xk = (w & 1) ? (unsigned)(xo2 >> 32) : (unsigned)xo2;
                    ra[h][w] = *(const f32x4*)(sa + ((int64_t)h * p.x_h_stride + (w >> 1) * 32) * 4 + xk);
                    rb[h][w] = *(const f32x4*)(sb + ((int64_t)h * p.x_h_stride + (w >> 1) * 32) * 4 + xk);
#else
                    ra[h][w] = *(const f32x4*)(sa + ((int64_t)h * p.x_h_stride + w * 16) * 4 + xo);
                    rb[h][w] = *(const f32x4*)(sb + ((int64_t)h * p.x_h_stride + w * 16) * 4 + xo);
#endif
                }
        };
        f32x4 tn[4][4];                                            // depth + w butterflies of the step about to be handed over
        auto transform = [&](float sgn, int h0) __attribute__((always_inline)) {
#pragma unroll
            for (int h = h0; h < h0 + 2; ++h) {
                f32x4 d[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
#ifdef WN_ABL_HALFLOAD
                    if (w >= 2) { ra[h][w] = ra[h][w - 2] * 1.5f; rb[h][w] = rb[h][w - 2] * 0.5f; }
#endif
                    d[w].x = __builtin_fmaf(sgn, rb[h][w].x, ra[h][w].x); d[w].y = __builtin_fmaf(sgn, rb[h][w].y, ra[h][w].y);
                    d[w].z = __builtin_fmaf(sgn, rb[h][w].z, ra[h][w].z); d[w].w = __builtin_fmaf(sgn, rb[h][w].w, ra[h][w].w);
                }
#ifdef WN_COAL
                // back to the MFMA lane layout (lane j = tile j): w = 2*pair lives in d[2*pair] for lanes j < 8 and, moved by 8 lanes,
                // in d[2*pair + 1] for lanes j >= 8; w = 2*pair + 1 the other way round (DPP row_ror:8, written under a bank mask)
                auto dppf = [](float old, float src, int bank_mask_lo) __attribute__((always_inline)) {
                    return bank_mask_lo ? __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x128, 0xf, 0x3, false))
                                        : __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x128, 0xf, 0xc, false));
                };
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    f32x4 e0, e1;
                    const f32x4 dA = d[2 * pr], dB = d[2 * pr + 1];
                    e0.x = dppf(dA.x, dB.x, 0); e0.y = dppf(dA.y, dB.y, 0); e0.z = dppf(dA.z, dB.z, 0); e0.w = dppf(dA.w, dB.w, 0);
                    e1.x = dppf(dB.x, dA.x, 1); e1.y = dppf(dB.y, dA.y, 1); e1.z = dppf(dB.z, dA.z, 1); e1.w = dppf(dB.w, dA.w, 1);
                    d[2 * pr] = e0; d[2 * pr + 1] = e1;
                }
#endif
                tn[h][0] = d[0] - d[2]; tn[h][1] = d[1] + d[2]; tn[h][2] = d[2] - d[1]; tn[h][3] = d[1] - d[3];
            }
        };
        // h butterfly of frequency rows 2*half, 2*half+1 -> the consumer's slot of half step `half`
        auto hand_over = [&](int half) __attribute__((always_inline)) {
#ifdef WN_ABL_NOHAND
            if (steps > 0) { asm volatile("" :: "v"(tn[0][0]), "v"(tn[1][1]), "v"(tn[2][2]), "v"(tn[3][3])); return; }
#endif
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (half == 0) {
                    b_lds[0][cw][0 + w][lane] = tn[0][w] - tn[2][w];
                    b_lds[0][cw][4 + w][lane] = tn[1][w] + tn[2][w];
                } else {
                    b_lds[1][cw][0 + w][lane] = tn[2][w] - tn[1][w];
                    b_lds[1][cw][4 + w][lane] = tn[1][w] - tn[3][w];
                }
            }
        };
        // ---- weights: half step (xd, cb, half) uses frequency points xd*16 + half*8 + 0..7 of block cb; the sequence repeats
        // every 8*cb_in half steps.  Producer thread tp copies float4 e = q*256 + tp (q < 2*CT) of the 8 x CT x 64 float4 slab.
        constexpr int kFill = 2 * CT;
        const int n_ct = p.cout_pad / 16;
        const int64_t w_pt = (int64_t)p.cb_in * n_ct * 256;        // floats per frequency point
        int fill_off[kFill];
#pragma unroll
        for (int q = 0; q < kFill; ++q) {
            const int e = q * 256 + tp;
            fill_off[q] = (int)((e / (64 * CT)) * w_pt) + (e % (64 * CT)) * 4;
        }
        const float* wbase = p.w + (int64_t)ct0 * 256;
        f32x4 fill[kFill];
        int f_xd = 0, f_cb = 0, f_hf = 0;
        auto fill_load = [&]() __attribute__((always_inline)) {
            const float* src = wbase + (int64_t)(f_xd * 16 + f_hf * 8) * w_pt + (int64_t)f_cb * n_ct * 256;
#pragma unroll
            for (int q = 0; q < kFill; ++q) fill[q] = *(const f32x4*)(src + fill_off[q]);
            if (++f_hf == 2) { f_hf = 0; if (++f_cb == p.cb_in) { f_cb = 0; f_xd = (f_xd + 1) & 3; } }
        };
        auto fill_store = [&](int slot) __attribute__((always_inline)) {
#ifdef WN_ABL_NOHAND
            if (steps > 0) { asm volatile("" :: "v"(fill[0])); return; }
#endif
            f32x4* dst = &w_lds[slot][0][0][0];
#pragma unroll
            for (int q = 0; q < kFill; ++q) dst[q * 256 + tp] = fill[q];
        };
#ifdef WN_ABL_NOBAR
        auto sync = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
#else
        auto sync = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
#endif

        // ---- prologue: slot 0 <- half step 0; the rows of step 1 in flight
        Cursor cT = {0, 0, 0};                 // the step whose butterflies are in tn
        Geo gT = geo_of(0);
        fill_load();
        issue_loads(cT, gT.xo, 0);
        issue_loads(cT, gT.xo, 2);
        fill_store(0);
        fill_load();
        transform(-1.f, 0);
        transform(-1.f, 2);
        hand_over(0);
        Cursor cL = advance(cT);               // the step whose raw rows are in flight
        Geo gL = cL.round < rounds ? geo_of(cL.round) : gT;
        issue_loads(cL, gL.xo, 0);
        issue_loads(cL, gL.xo, 2);
        // Each of the two windows of a step (one per consumer half step) carries half of the next step's transform and half of the
        // loads of the step after it, so neither window outlasts the consumers' 32*CT MFMAs.
#pragma unroll 1
        for (int s = 0; s < steps; ++s) {
            const float sgn = cL.xd == 1 ? 1.f : -1.f;
            Cursor cN = advance(cL);                               // step s+2
            Geo gN = gL;
            if (cN.round != cL.round && cN.round < rounds) gN = geo_of(cN.round);
            sync();                                                // consumers start (s, half 0) out of slot 0
#ifdef WN_ABL_NOPROD
            sync();
            continue;
#endif
            hand_over(1);                                          // (s, half 1) -> slot 1   (tn of step s is dead after this)
            fill_store(1);
            fill_load();
            transform(sgn, 0);                                     // tn rows 0,1 <- step s+1
            issue_loads(cN, gN.xo, 0);                             // rows 0,1 of step s+2 (past the end: a harmless repeat)
            sync();                                                // consumers start (s, half 1) out of slot 1
            transform(sgn, 2);                                     // tn rows 2,3 <- step s+1
            hand_over(0);                                          // (s+1, half 0) -> slot 0
            fill_store(0);
            fill_load();
            issue_loads(cN, gN.xo, 2);
            cT = cL; gT = gL; cL = cN; gL = gN;
        }
        return;
    }

    // =============================================================================================== consumer
#ifdef WN_PRIO
    __builtin_amdgcn_s_setprio(WN_PRIO);      // the MFMA wave wins issue arbitration against its producer partner
#endif
    f32x4 acc[4][4][CT];
    f32x4 p1[2][2][CT];                       // P1 = z1 - z2 (- z3): the second output slice, folded along depth in registers
    // one half step: 8 frequency points (two xh rows) x 4 k-steps x CT cout tiles = 32*CT MFMAs; operands one point ahead.
    // FIRST (first channel block of a depth frequency): C = 0 instead of clearing the accumulators.
    auto half_step = [&](auto first_tag, auto half_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int HALF = decltype(half_tag)::value;
#ifndef WN_ABL_NOBAR
        asm volatile("s_barrier" ::: "memory");
#endif
        // PP points share a stage so that MFMAs on one accumulator are >= 2 issues apart (40-cycle dependent latency vs 32-cycle issue)
        constexpr int PP = CT == 1 ? 2 : 1;
        constexpr int NS = 8 / PP;
        f32x4 bq[2][PP], wq[2][PP][CT];
        auto fetch = [&](int buf, int st) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < PP; ++q) {
#ifdef WN_ABL_NOLDSR
                bq[buf][q] = (f32x4){(float)st, 1.f, (float)lane, 3.f};
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) wq[buf][q][ct] = (f32x4){(float)ct, (float)lane, 2.f, (float)st};
                asm volatile("" : "+v"(bq[buf][q]));
#else
                bq[buf][q] = b_lds[HALF][cw][st * PP + q][lane];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) wq[buf][q][ct] = w_lds[HALF][st * PP + q][ct][lane];
#endif
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int cur = st & 1;
            if (st + 1 < NS) fetch(cur ^ 1, st + 1);
            __builtin_amdgcn_sched_barrier(0);       // the next stage's reads are issued BEFORE this stage's MFMAs (a full stage of cover)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int q = 0; q < PP; ++q)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const int pt = st * PP + q;
                        const int xh = HALF * 2 + (pt >> 2), xw = pt & 3;
                        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#ifdef WN_ABL_NOMFMA
                        if (s == 0) acc[xh][xw][ct] = (FIRST ? z4 : acc[xh][xw][ct]) + wq[cur][q][ct] * bq[cur][q];
#else
                        acc[xh][xw][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[cur][q][ct][s], bq[cur][q][s], FIRST && s == 0 ? z4 : acc[xh][xw][ct], 0, 0, 0);
#endif
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // end of a depth frequency: in-plane inverse (4x4 -> 2x2) of its accumulators, folded along depth
    // (A^T columns [1 1 1 0] for od 0, [0 1 -1 -1] for od 1); frequency 2 completes output slice 0, frequency 3 slice 1.
    auto phase_end = [&](int xd, const Geo& geo) __attribute__((always_inline)) {
        f32x4 inv[2][2][CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            f32x4 hh[2][4];
#pragma unroll
            for (int xw = 0; xw < 4; ++xw) {
                hh[0][xw] = acc[0][xw][ct] + acc[1][xw][ct] + acc[2][xw][ct];
                hh[1][xw] = acc[1][xw][ct] - acc[2][xw][ct] - acc[3][xw][ct];
            }
            inv[0][0][ct] = hh[0][0] + hh[0][1] + hh[0][2]; inv[0][1][ct] = hh[0][1] - hh[0][2] - hh[0][3];
            inv[1][0][ct] = hh[1][0] + hh[1][1] + hh[1][2]; inv[1][1][ct] = hh[1][1] - hh[1][2] - hh[1][3];
        }
        if (xd == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) z_lds[cw][i * CT + ct][lane] = inv[i >> 1][i & 1][ct];
            return;
        }
        if (xd == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    z_lds[cw][i * CT + ct][lane] += inv[i >> 1][i & 1][ct];
                    p1[i >> 1][i & 1][ct] = inv[i >> 1][i & 1][ct];
                }
            return;
        }
        const int od = xd - 2;
        f32x4 outv[2][2][CT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if (xd == 2) {
                    outv[i >> 1][i & 1][ct] = z_lds[cw][i * CT + ct][lane] + inv[i >> 1][i & 1][ct];
                    p1[i >> 1][i & 1][ct] -= inv[i >> 1][i & 1][ct];
                } else {
                    outv[i >> 1][i & 1][ct] = p1[i >> 1][i & 1][ct] - inv[i >> 1][i & 1][ct];
                }
            }
        if (!geo.valid) return;
        const int64_t yo = p.y_off0 + (int64_t)geo.n * p.y_n_stride + (int64_t)(2 * geo.dt + od) * p.y_d_stride + (int64_t)(2 * geo.ht) * p.y_h_stride +
                           (int64_t)(2 * geo.wt) * 16 + g * 4;
        const int64_t ro = p.r_off0 + (int64_t)geo.n * p.r_n_stride + (int64_t)(2 * geo.dt + od) * p.r_d_stride + (int64_t)(2 * geo.ht) * p.r_h_stride +
                           (int64_t)(2 * geo.wt) * 16 + g * 4;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const f32x4 bn_sc = *(const f32x4*)(p.scale + (ct0 + ct) * 16 + g * 4);
            const f32x4 bn_sh = *(const f32x4*)(p.shift + (ct0 + ct) * 16 + g * 4);
#pragma unroll
            for (int oh = 0; oh < 2; ++oh)
#pragma unroll
                for (int ow = 0; ow < 2; ++ow) {
                    f32x4 v_ = outv[oh][ow][ct] * bn_sc + bn_sh;
                    if (p.res) v_ += *(const f32x4*)(p.res + ro + oh * p.r_h_stride + ow * 16 + (int64_t)(ct0 + ct) * p.r_cb_stride);
                    if (p.relu) { v_.x = fmaxf(v_.x, 0.f); v_.y = fmaxf(v_.y, 0.f); v_.z = fmaxf(v_.z, 0.f); v_.w = fmaxf(v_.w, 0.f); }
                    *(f32x4*)(p.y + yo + oh * p.y_h_stride + ow * 16 + (int64_t)(ct0 + ct) * p.y_cb_stride) = v_;
                }
        }
    };
#pragma unroll 1
    for (int round = 0; round < rounds; ++round) {
        const Geo geo = geo_of(round);
#pragma unroll 1
        for (int xd = 0; xd < 4; ++xd) {
            half_step(std::true_type{}, std::integral_constant<int, 0>{});
            half_step(std::true_type{}, std::integral_constant<int, 1>{});
#pragma unroll 1
            for (int c = 1; c < p.cb_in; ++c) {
                half_step(std::false_type{}, std::integral_constant<int, 0>{});
                half_step(std::false_type{}, std::integral_constant<int, 1>{});
            }
#ifdef WN_ABL_NOPHASE
            if (xd == 3 && round == rounds - 1) phase_end(xd, geo);
#else
            phase_end(xd, geo);
#endif
        }
    }
}

template <int CT>
int launch(const drc_tapconv_params& p, hipStream_t stream) {
    const long tiles = (long)p.N * (p.OD / 2) * (p.OH / 2) * (p.OW / 2);
    const long groups = (tiles + 15) / 16;
    const int n_cg = p.cout_pad / 16 / CT;
    // one block per CU (its LDS slots and two waves per SIMD fill it); every cout group gets the same number of blocks
    long per_cg = 256 / n_cg;
    const long need = (groups + WN_CONS - 1) / WN_CONS;
    if (per_cg > need) per_cg = need;
    if (per_cg < 1) per_cg = 1;
    dim3 grid((unsigned)(per_cg * n_cg), 1, 1);
    hipLaunchKernelGGL((wino3d_kernel<CT>), grid, dim3(512), 0, stream, p);
    return (int)hipGetLastError();
}

// U = (G x G x G) g per (cout, cin) pair in the order the consumers read it: [xi = (xd*4 + xh)*4 + xw][cb][cout tile][g*16 + j][4]
// (cout = tile*16 + j, channel = cb*16 + g*4 + 0..3), zero-padded to whole channel blocks / cout tiles.
__global__ __launch_bounds__(256) void wino_weights_kernel(const float* __restrict__ w, int cout, int cin, int transposed, int flip,
                                                           float* __restrict__ out) {
    const int cb_n = (cin + 15) / 16, cout_pad = (cout + 15) / 16 * 16;
    const long pairs = (long)cb_n * cout_pad * 16;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < pairs; idx += (long)gridDim.x * 256) {
        long t = idx;
        const int c = (int)(t & 15); t >>= 4;
        const int co = (int)(t % cout_pad);
        const int cb = (int)(t / cout_pad);
        const int ci = cb * 16 + c;
        const long dst = ((long)cb * (cout_pad / 16) + co / 16) * 256 + ((c >> 2) * 16 + (co & 15)) * 4 + (c & 3);
        float a[3][3][3];
        const bool live = co < cout && ci < cin;
        const float* src = w + (transposed ? ((long)ci * cout + co) : ((long)co * cin + ci)) * 27;
#pragma unroll
        for (int k = 0; k < 27; ++k) a[k / 9][(k / 3) % 3][k % 3] = live ? src[flip ? 26 - k : k] : 0.f;
        float b[3][3][4], d[3][4][4];
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const float g0 = a[kd][kh][0], g1 = a[kd][kh][1], g2 = a[kd][kh][2];
                b[kd][kh][0] = g0; b[kd][kh][1] = 0.5f * (g0 + g1 + g2); b[kd][kh][2] = 0.5f * (g0 - g1 + g2); b[kd][kh][3] = g2;
            }
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int xw = 0; xw < 4; ++xw) {
                const float g0 = b[kd][0][xw], g1 = b[kd][1][xw], g2 = b[kd][2][xw];
                d[kd][0][xw] = g0; d[kd][1][xw] = 0.5f * (g0 + g1 + g2); d[kd][2][xw] = 0.5f * (g0 - g1 + g2); d[kd][3][xw] = g2;
            }
#pragma unroll
        for (int xh = 0; xh < 4; ++xh)
#pragma unroll
            for (int xw = 0; xw < 4; ++xw) {
                const float g0 = d[0][xh][xw], g1 = d[1][xh][xw], g2 = d[2][xh][xw];
                const float u[4] = {g0, 0.5f * (g0 + g1 + g2), 0.5f * (g0 - g1 + g2), g2};
#pragma unroll
                for (int xd = 0; xd < 4; ++xd) out[(long)((xd * 4 + xh) * 4 + xw) * pairs + dst] = u[xd];
            }
    }
}

}  // namespace

extern "C" int drc_conv3d_k3_wino_fwd(const drc_tapconv_params* pp, int cout_tiles_per_wave, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD <= 0 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (p.cout_pad <= 0 || (p.cout_pad & 15) || p.cb_in <= 0) return -2;
    const drc_tap_class& k = p.cls[0];
    if (p.n_classes != 1 || p.in_mul != 1 || p.out_mul != 1 || k.nd != 3 || k.nh != 3 || k.nw != 3 || k.sd != 1 || k.sh != 1 || k.sw != 1)
        return -4;
    if ((p.OD | p.OH | p.OW) & 1) return -4;                                   // whole 2x2x2 tiles only
    if ((int64_t)p.N * p.x_n_stride * 4 >= (1LL << 32)) return -5;             // 32-bit lane offsets over the whole batch
    if ((int64_t)p.N * p.OD * p.OH * p.OW / 8 >= (1LL << 31) - 16 || (int64_t)64 * p.cb_in * p.cout_pad * 16 >= (1LL << 31)) return -5;
    const int ct = p.cout_pad / 16, CT = cout_tiles_per_wave;
    if ((CT != 1 && CT != 2) || ct % CT) return -2;
    hipStream_t s = (hipStream_t)stream;
    return CT == 2 ? launch<2>(p, s) : launch<1>(p, s);
}

extern "C" int drc_pack_weights_wino(const float* w, int cout, int cin, int transposed, int flip, float* out, void* stream) {
    if (cout <= 0 || cin <= 0) return -2;
    if (!w || !out) return -1;
    const long pairs = (long)((cin + 15) / 16) * ((cout + 15) / 16 * 16) * 16;
    long blocks = (pairs + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, cout, cin, transposed, flip, out);
    return (int)hipGetLastError();
}
