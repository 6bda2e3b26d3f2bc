// wino3d.hip -- stride-1 3x3x3 convolution (+BN, +residual, +ReLU) as Winograd F(2x2x2, 3x3x3) on the fp32 matrix cores
// (gfx950 / CDNA4).
//
//   reference: dres0/dres1, classifN[0], hourglass conv2/conv4 (stackhourglass.py:63-88, :14-20)
//
// fp32 MFMA runs at the same 64 flop/clk/SIMD as the packed fp32 VALU, so the classic trade pays here: a 2x2x2 output tile
// costs 64 multiplies per (cin, cout) pair instead of 8*27 = 216 (3.4x fewer MFMAs) for a few VALU adds per tile.
//   Y = A^T [ (G g G^T) . (B^T d B) ] A   in each of the three dimensions, with
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1].
// A wave owns 16 tiles (the N dimension of the 16x16x4 MFMA) x CT*16 couts; lane (tile j, g) transforms channels 4g..4g+3 of
// its tile's 4x4x4 input patch in registers -- the blocked layout makes every patch voxel one float4.  Work is cut into steps
// (depth frequency xd, channel block cb): the two input slices of xd (2 x 16 float4 loads per lane, issued two row phases
// ahead), the depth butterfly, the two in-plane butterflies -> 16 B fragments, each feeding CT*4 MFMAs against the
// transformed weights U[xi][cb][cout][16] (drc_pack_weights_wino).  The weights of a half step (8 frequency points, 16 KB
// at CT = 2) are the same for every wave: the block's four waves run in lock step (one s_barrier per half step) and share
// them through a three-slab LDS ring filled one half step ahead.  After the last channel block of a depth frequency its
// 16 x CT accumulators are inverse-transformed in-plane to 2x2 and parked in LDS; the fourth frequency combines the four
// along depth (A^T) into the tile's 2x2x2 outputs and runs tapdirect's epilogue.  Needs even OD, OH, OW (the engine sends
// other shapes to tapdirect).  Results differ from the direct kernels' by fp32 rounding only.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WN_WAVES 4

namespace {

// CV = the cost volume fused into the input loads (dres0[0], stackhourglass.py:115-130): instead of a materialised
// [N][2C/16][D'+2][H'+2][W'+2][16] volume the patch loads read the blocked 2D feature maps directly -- channel blocks
// < cbi from the left map at (y, x), the others from the right map at (y, x - i), i = lo4 + slice -- and a load whose voxel
// is outside the volume or fails the validity test 0 <= x - i < W' is pointed at a halo voxel of the map (zero).
template <int CT, bool CV>
__device__ __forceinline__ void wino3d_body(const drc_tapconv_params& p, const drc_costvol_src& cv) {
    // per wave: the in-plane inverse (2x2 x CT float4 per lane) of depth frequencies 0..2 of the tile group in flight; the
    // last frequency combines them along depth and runs the epilogue.  Written once, read once.
    f32x4 o0[4 * CT], o1[4 * CT];      // running depth inverse: od 0 = z0 + z1 + z2, od 1 = z1 - z2 - z3
    // transformed weights of a half step (8 frequency points x CT*16 couts x 16 channels), ring of three, shared by the block
    __shared__ __attribute__((aligned(16))) float w_ring[3][8][CT][256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const int g = lane >> 4;

    const drc_tap_class cls = p.cls[0];
    const int TD = p.OD >> 1, TH = p.OH >> 1, TW = p.OW >> 1;
    const int tiles = p.N * TD * TH * TW;
    const int groups = (tiles + 15) >> 4;
    // A block keeps one cout group (its weights are what the ring holds); its four waves walk the tile groups in rounds of
    // four, in lock step (one s_barrier per half step).  Round r of block position p takes groups 4*(r*nbk + p) .. +3, so at
    // any time the blocks work on one contiguous stretch of tiles -- and positions are numbered XCD by XCD (the dispatcher
    // deals workgroups round-robin over the 8 XCDs): the 32 blocks that share an L2 sweep ~2,000 adjacent tiles (1.7 ROIs of
    // Config A, 2.8 MB of input) together, so the slices and rows a tile shares with its neighbours are fetched from HBM
    // once per XCD instead of once per tile.  (With contiguous per-block ranges each of the 32 streams through its own ROI
    // and the 4 MB L2 turns over before any reuse: 1.28 GB fetched per launch for 0.35 GB of input.)
    const int n_cg = p.cout_pad / 16 / CT;
    int cg, pos;
    const int nbk = (int)gridDim.x / n_cg;         // blocks per cout group (the launcher makes gridDim.x a multiple of n_cg)
    if (gridDim.x % (8 * n_cg) == 0) {
        const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
        cg = l % n_cg;
        pos = xcd * (nbk / 8) + l / n_cg;
    } else {
        cg = blockIdx.x % n_cg;
        pos = blockIdx.x / n_cg;
    }
    const int chunks = (groups + WN_WAVES - 1) / WN_WAVES;
    const int rounds = (chunks + nbk - 1) / nbk;    // the same for every block; a block without a chunk in the last round idles through it
    const int ct0 = cg * CT;
    const int w_cb = p.cout_pad * 16;              // floats per (xi, cb)
    const int w_xi = w_cb * p.cb_in;               // floats per frequency point

    // lane geometry of a round: byte offset of the 4x4x4 patch origin (logical voxel 2t-1 = padded 2t + first), channels 4g..4g+3
    struct Geo { unsigned xo; int n, dt, ht, wt; bool valid; };
    auto geo_of = [&](int round) __attribute__((always_inline)) {
        Geo q;
        int grp = (round * nbk + pos) * WN_WAVES + wave;
        const bool active = grp < groups;
        if (!active) grp = groups - 1;
        int tile = grp * 16 + j;
        q.valid = active && tile < tiles;
        if (tile >= tiles) tile = tiles - 1;
        q.wt = tile % TW; tile /= TW;
        q.ht = tile % TH; tile /= TH;
        q.dt = tile % TD;
        q.n = tile / TD;
        if constexpr (CV)
            q.xo = (unsigned)((q.n * cv.n_stride + (2 * q.ht - 1 + cv.pad) * cv.h_stride + (int64_t)(2 * q.wt - 1 + cv.pad) * 16 + g * 4) * 4);
        else
            q.xo = (unsigned)((q.n * p.x_n_stride + (2 * q.dt + cls.dd0) * p.x_d_stride + (2 * q.ht + cls.dh0) * p.x_h_stride +
                               (int64_t)(2 * q.wt + cls.dw0) * 16 + g * 4) * 4);
        return q;
    };
    // depth butterfly of frequency xd: slice a + sgn * slice b  (d0-d2, d1+d2, d2-d1, d1-d3)
    auto slice_a = [](int xd) { return xd == 0 ? 0 : (xd == 2 ? 2 : 1); };
    auto slice_b = [](int xd) { return xd == 2 ? 1 : (xd == 3 ? 3 : 2); };

    // one h-row of the two slices of a step (8 float4), and its depth + w butterflies
    auto load_row = [&](f32x4 (&ra)[4], f32x4 (&rb)[4], const char* sa, const char* sb, int h, unsigned xo) __attribute__((always_inline)) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            ra[w] = *(const f32x4*)(sa + ((int64_t)h * p.x_h_stride + w * 16) * 4 + xo);
            rb[w] = *(const f32x4*)(sb + ((int64_t)h * p.x_h_stride + w * 16) * 4 + xo);
        }
    };
    // the same row of the two slices, read from the feature maps (CV): patch voxel (slice s, row h, column w) of tile
    // (dt, ht, wt) is volume voxel (d, y, x) = (2dt-1+s, 2ht-1+h, 2wt-1+w); disparity i = lo4 + d, xs = x - i.  The eight
    // column offsets of a step (two slices x four columns; a voxel outside the volume or failing 0 <= xs < W' is pointed at
    // halo column 0 of its row) are worked out once per step, ahead of its first row load; the row term is wave-uniform.
    unsigned cvo[2][4];
    auto cv_offsets = [&](int cb, int xd, const Geo& q) __attribute__((always_inline)) {
        const bool right = cb >= cv.cbi;                                  // wave-uniform
        const int d0 = 2 * q.dt - 1, x0 = 2 * q.wt - 1;
        const int k = x0 - cv.lo4 - d0;                                   // xs = k - s + w
        const unsigned zrow = q.xo - (unsigned)((x0 + cv.pad) * 64);
#pragma unroll
        for (int ab = 0; ab < 2; ++ab) {
            const int s = ab == 0 ? slice_a(xd) : slice_b(xd);
            const int d = d0 + s;
            const bool ind = (unsigned)d < (unsigned)p.OD;
            const unsigned rs = q.xo - (right ? (unsigned)((cv.lo4 + d) * 64) : 0u);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const bool ok = ind && (unsigned)(x0 + w) < (unsigned)cv.Wp && (unsigned)(k - s + w) < (unsigned)cv.Wp;
                cvo[ab][w] = ok ? rs + (unsigned)(w * 64) : zrow;
                asm volatile("" : "+v"(cvo[ab][w]));
            }
        }
    };
    auto load_row_cv = [&](f32x4 (&ra)[4], f32x4 (&rb)[4], int cb, int h) __attribute__((always_inline)) {
        const bool right = cb >= cv.cbi;
        const char* base = (const char*)((right ? cv.right : cv.left) + (int64_t)(right ? cb - cv.cbi : cb) * cv.cb_stride + (int64_t)h * cv.h_stride);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            ra[w] = *(const f32x4*)(base + cvo[0][w]);
            rb[w] = *(const f32x4*)(base + cvo[1][w]);
        }
    };
    auto bfly_row = [&](f32x4 (&t)[4], const f32x4 (&ra)[4], const f32x4 (&rb)[4], float sgn) __attribute__((always_inline)) {
        f32x4 d[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            d[w].x = __builtin_fmaf(sgn, rb[w].x, ra[w].x); d[w].y = __builtin_fmaf(sgn, rb[w].y, ra[w].y);
            d[w].z = __builtin_fmaf(sgn, rb[w].z, ra[w].z); d[w].w = __builtin_fmaf(sgn, rb[w].w, ra[w].w);
        }
        t[0] = d[0] - d[2]; t[1] = d[1] + d[2]; t[2] = d[2] - d[1]; t[3] = d[1] - d[3];
        // pinned here: LLVM otherwise sinks the butterflies to their use after the MFMA phases and parks the raw rows in AGPRs
#pragma unroll
        for (int w = 0; w < 4; ++w) asm volatile("" : "+v"(t[w]));
    };

    // ---- weight ring.  Half step hs = (xd, cb, half) uses frequency points xd*16 + half*8 + 0..7 of block cb; the sequence
    // repeats every 8*cb_in half steps whatever the round.  Thread t copies float4 e = q*256 + t of the slab, q < 2*CT:
    // point i = e / (64*CT), float4 `e % (64*CT)` of that point's CT*16 x 16 chunk.
    constexpr int kFill = 2 * CT;
    int fill_off[kFill];
#pragma unroll
    for (int q = 0; q < kFill; ++q) {
        const int e = q * 256 + (int)threadIdx.x;
        fill_off[q] = (e / (64 * CT)) * w_xi + (e % (64 * CT)) * 4;
    }
    const float* wbase = p.w + ct0 * 256;
    f32x4 fill[kFill];
    int f_xd = 0, f_cb = 0, f_hf = 0;              // half step the next fill_load fetches
    auto fill_load = [&]() __attribute__((always_inline)) {
        const float* src = wbase + (f_xd * 16 + f_hf * 8) * w_xi + f_cb * w_cb;
#pragma unroll
        for (int q = 0; q < kFill; ++q) fill[q] = *(const f32x4*)(src + fill_off[q]);
        if (++f_hf == 2) { f_hf = 0; if (++f_cb == p.cb_in) { f_cb = 0; f_xd = (f_xd + 1) & 3; } }
    };
    auto fill_store = [&](int slab) __attribute__((always_inline)) {
        float* dst = &w_ring[slab][0][0][0];
#pragma unroll
        for (int q = 0; q < kFill; ++q) *(f32x4*)(dst + (q * 256 + (int)threadIdx.x) * 4) = fill[q];
    };
    // half-step boundary hs: publish the weights of hs+1, wait for everyone, fetch the weights of hs+2.  The barrier makes
    // slab hs+1 visible for the next half step and guarantees that nobody still reads slab hs+2 = hs-1 when it is overwritten at
    // the next boundary (slab hs itself became visible at the previous barrier; reading it ahead of this one measured slower).
    // Only the LDS counter is drained: global loads stay in flight across the barrier.
    auto boundary = [&](int hs) __attribute__((always_inline)) {
        fill_store((hs + 1) % 3);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        fill_load();
    };
    auto load_w = [&](f32x4 (&wf)[4][CT], int slab, int row) __attribute__((always_inline)) {
#pragma unroll
        for (int xw = 0; xw < 4; ++xw)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) wf[xw][ct] = *(const f32x4*)&w_ring[slab][(row & 1) * 4 + xw][ct][j * 16 + g * 4];
    };

    f32x4 acc[4][4][CT];
    // first channel block of a depth frequency (FIRST): C = 0, no accumulator clearing; afterwards accumulate.  The two forms
    // sit in two copies of the step body, not behind a branch inside one: merging accumulators from two paths makes the
    // register allocator shuttle them through VGPRs.  s outermost: 4*CT independent accumulators between dependent MFMAs.
#define WN_MFMA_ROW(XH, WF)                                                                            \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                      \
        _Pragma("unroll") for (int xw = 0; xw < 4; ++xw)                                               \
            _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) {                                        \
                const f32x4 z4_ = {0.f, 0.f, 0.f, 0.f};                                                \
                acc[XH][xw][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(WF[xw][ct][s], v[XH][xw][s], FIRST && s == 0 ? z4_ : acc[XH][xw][ct], 0, 0, 0); \
            }

    // ---- prologue: weights of half steps 0 (published) and 1 (in registers); B fragments of step 0; rows 0, 1 of step 1
    fill_load();
    fill_store(0);
    fill_load();
    struct Cursor { int round, xd, cb; };
    auto advance = [&](Cursor c) __attribute__((always_inline)) {
        if (++c.cb == p.cb_in) { c.cb = 0; if (++c.xd == 4) { c.xd = 0; ++c.round; } }
        return c;
    };
    auto slices_of = [&](const Cursor& c, const char*& sa, const char*& sb) __attribute__((always_inline)) {
        sa = (const char*)(p.x + (int64_t)c.cb * p.x_cb_stride + (int64_t)slice_a(c.xd) * p.x_d_stride);
        sb = (const char*)(p.x + (int64_t)c.cb * p.x_cb_stride + (int64_t)slice_b(c.xd) * p.x_d_stride);
    };
    auto ld = [&](f32x4 (&ra)[4], f32x4 (&rb)[4], const char* sa, const char* sb, const Cursor& c, int h, const Geo& q) __attribute__((always_inline)) {
        if constexpr (CV) load_row_cv(ra, rb, c.cb, h);
        else load_row(ra, rb, sa, sb, h, q.xo);
    };
    Cursor c0 = {0, 0, 0};                 // the step whose MFMAs run
    Cursor c1 = advance(c0);               // the step whose rows are being transformed (rows 2, 3 still loading)
    Geo geo0 = geo_of(0);
    Geo geo1 = c1.round < rounds ? geo_of(c1.round) : geo0;
    f32x4 v[4][4], tn[4][4], wfA[4][CT], wfB[4][CT];
    f32x4 r0a[4], r0b[4], r1a[4], r1b[4], r2a[4], r2b[4], r3a[4], r3b[4];      // the four h-rows in flight (two slices each)
    {
        const char *sa, *sb;
        slices_of(c0, sa, sb);
        if constexpr (CV) cv_offsets(c0.cb, c0.xd, geo0);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            ld(r0a, r0b, sa, sb, c0, h, geo0);
            bfly_row(tn[h], r0a, r0b, -1.f);
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            v[0][w] = tn[0][w] - tn[2][w]; v[1][w] = tn[1][w] + tn[2][w]; v[2][w] = tn[2][w] - tn[1][w]; v[3][w] = tn[1][w] - tn[3][w];
        }
        slices_of(c1, sa, sb);
        if constexpr (CV) cv_offsets(c1.cb, c1.xd, geo1);
        ld(r0a, r0b, sa, sb, c1, 0, geo1);
        ld(r1a, r1b, sa, sb, c1, 1, geo1);
    }

    int slab = 0;
    // one (depth frequency, channel block) step: its MFMAs in four row phases of 16*CT MFMAs, with -- in their shadow --
    // the depth/w butterflies of the next step's rows (loaded two phases earlier) and the loads of the rows two phases ahead
    // (rows 2, 3 of the next step, then rows 0, 1 of the one after).  Past the last step the loads are harmless repeats.
    auto do_step = [&](auto first_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const int cb = c0.cb; (void)cb;
        Cursor c2 = advance(c1);
        Geo geo2 = geo1;
        if (c2.round != c1.round && c2.round < rounds) geo2 = geo_of(c2.round);
        const char *sa1, *sb1, *sa2, *sb2;
        slices_of(c1, sa1, sb1);
        slices_of(c2, sa2, sb2);
        const float sgn = c1.xd == 1 ? 1.f : -1.f;
        const int slab_a = slab, slab_b = slab == 2 ? 0 : slab + 1;
        slab = slab_b == 2 ? 0 : slab_b + 1;

        boundary(slab_a);
        load_w(wfA, slab_a, 0);
        ld(r2a, r2b, sa1, sb1, c1, 2, geo1);
        load_w(wfB, slab_a, 1);
        // h butterfly, rows 1..3 of this step's B fragments (row 0 was finished at the end of the previous step): in the shadow
        // of row 0's MFMAs, before tn[1..3] are overwritten by the next step's rows
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            v[1][w] = tn[1][w] + tn[2][w]; v[2][w] = tn[2][w] - tn[1][w]; v[3][w] = tn[1][w] - tn[3][w];
            asm volatile("" : "+v"(v[1][w]), "+v"(v[2][w]), "+v"(v[3][w]));
        }
        bfly_row(tn[0], r0a, r0b, sgn);
        WN_MFMA_ROW(0, wfA)
        __builtin_amdgcn_sched_barrier(0);
        ld(r3a, r3b, sa1, sb1, c1, 3, geo1);
        bfly_row(tn[1], r1a, r1b, sgn);
        WN_MFMA_ROW(1, wfB)
        __builtin_amdgcn_sched_barrier(0);
        boundary(slab_b);
        load_w(wfA, slab_b, 2);
        if constexpr (CV) cv_offsets(c2.cb, c2.xd, geo2);
        ld(r0a, r0b, sa2, sb2, c2, 0, geo2);
        load_w(wfB, slab_b, 3);
        bfly_row(tn[2], r2a, r2b, sgn);
        WN_MFMA_ROW(2, wfA)
        __builtin_amdgcn_sched_barrier(0);
        ld(r1a, r1b, sa2, sb2, c2, 1, geo2);
        bfly_row(tn[3], r3a, r3b, sgn);
        WN_MFMA_ROW(3, wfB)
        __builtin_amdgcn_sched_barrier(0);
        // h butterfly, row 0 of the next step's B fragments (its MFMAs come first); rows 1..3 follow inside that step
#pragma unroll
        for (int w = 0; w < 4; ++w) v[0][w] = tn[0][w] - tn[2][w];
        c0 = c1; c1 = c2; geo0 = geo1; geo1 = geo2;
    };
    // end of a depth frequency: its in-plane inverse is parked in LDS; the last one combines the four along depth
    // (A^T columns [1 1 1 0] for od 0, [0 1 -1 -1] for od 1) and runs the epilogue.
    // (xd_, geo: the frequency and tile geometry of the phase that just ended -- do_step has already advanced the cursors)
    auto phase_end = [&](int xd_, const Geo& geo) __attribute__((always_inline)) {
        f32x4 inv0[2][CT], inv1[2][CT];        // A^T . A in-plane (4x4 -> 2x2): [oh = 0 | 1][ow][ct]
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            f32x4 hh[2][4];
#pragma unroll
            for (int xw = 0; xw < 4; ++xw) {
                hh[0][xw] = acc[0][xw][ct] + acc[1][xw][ct] + acc[2][xw][ct];
                hh[1][xw] = acc[1][xw][ct] - acc[2][xw][ct] - acc[3][xw][ct];
            }
            inv0[0][ct] = hh[0][0] + hh[0][1] + hh[0][2]; inv0[1][ct] = hh[0][1] - hh[0][2] - hh[0][3];
            inv1[0][ct] = hh[1][0] + hh[1][1] + hh[1][2]; inv1[1][ct] = hh[1][1] - hh[1][2] - hh[1][3];
        }
        if (xd_ < 3) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int ow = 0; ow < 2; ++ow) {
                    const int i0 = (0 * 2 + ow) * CT + ct, i1 = (1 * 2 + ow) * CT + ct;
                    if (xd_ == 0) { o0[i0] = inv0[ow][ct]; o0[i1] = inv1[ow][ct]; }
                    else if (xd_ == 1) { o0[i0] += inv0[ow][ct]; o0[i1] += inv1[ow][ct]; o1[i0] = inv0[ow][ct]; o1[i1] = inv1[ow][ct]; }
                    else { o0[i0] += inv0[ow][ct]; o0[i1] += inv1[ow][ct]; o1[i0] -= inv0[ow][ct]; o1[i1] -= inv1[ow][ct]; }
                }
            return;
        }
        if (!geo.valid) return;
        f32x4 bn_sc[CT], bn_sh[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            bn_sc[ct] = *(const f32x4*)(p.scale + (ct0 + ct) * 16 + g * 4);
            bn_sh[ct] = *(const f32x4*)(p.shift + (ct0 + ct) * 16 + g * 4);
        }
        const int64_t yo = p.y_off0 + (int64_t)geo.n * p.y_n_stride + (int64_t)(2 * geo.dt) * p.y_d_stride + (int64_t)(2 * geo.ht) * p.y_h_stride +
                           (int64_t)(2 * geo.wt) * 16 + g * 4;
        const int64_t ro = p.r_off0 + (int64_t)geo.n * p.r_n_stride + (int64_t)(2 * geo.dt) * p.r_d_stride + (int64_t)(2 * geo.ht) * p.r_h_stride +
                           (int64_t)(2 * geo.wt) * 16 + g * 4;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int oh = 0; oh < 2; ++oh)
#pragma unroll
                for (int ow = 0; ow < 2; ++ow) {
                    const int i = (oh * 2 + ow) * CT + ct;
                    const f32x4 z3 = oh == 0 ? inv0[ow][ct] : inv1[ow][ct];
#pragma unroll
                    for (int od = 0; od < 2; ++od) {
                        f32x4 v_ = (od == 0 ? o0[i] : o1[i] - z3) * bn_sc[ct] + bn_sh[ct];
                        if (p.res)
                            v_ += *(const f32x4*)(p.res + ro + od * p.r_d_stride + oh * p.r_h_stride + ow * 16 + (int64_t)(ct0 + ct) * p.r_cb_stride);
                        if (p.relu) { v_.x = fmaxf(v_.x, 0.f); v_.y = fmaxf(v_.y, 0.f); v_.z = fmaxf(v_.z, 0.f); v_.w = fmaxf(v_.w, 0.f); }
                        *(f32x4*)(p.y + yo + od * p.y_d_stride + oh * p.y_h_stride + ow * 16 + (int64_t)(ct0 + ct) * p.y_cb_stride) = v_;
                    }
                }
    };
#pragma unroll 1
    for (int ph = 0; ph < rounds * 4; ++ph) {
        const int xd_ = c0.xd;
        const Geo geo = geo0;
        do_step(std::true_type{});
#pragma unroll 1
        for (int c = 1; c < p.cb_in; ++c) do_step(std::false_type{});
        phase_end(xd_, geo);
    }
#undef WN_MFMA_ROW
}

template <int CT>
__global__ __launch_bounds__(64 * WN_WAVES) void wino3d_kernel(const drc_tapconv_params p) {
    wino3d_body<CT, false>(p, drc_costvol_src{});
}

template <int CT>
__global__ __launch_bounds__(64 * WN_WAVES) void wino3d_cv_kernel(const drc_tapconv_params p, const drc_costvol_src cv) {
    wino3d_body<CT, true>(p, cv);
}

template <int CT>
int launch(const drc_tapconv_params& p, const drc_costvol_src* cv, hipStream_t stream) {
    const long tiles = (long)p.N * (p.OD / 2) * (p.OH / 2) * (p.OW / 2);
    const long groups = (tiles + 15) / 16;
    const int n_cg = p.cout_pad / 16 / CT;
    // one block per CU (the LDS ring and the register file allow no more); every cout group gets the same number of blocks
    long per_cg = 256 / n_cg;
    const long need = (groups + WN_WAVES - 1) / WN_WAVES;
    if (per_cg > need) per_cg = need;
    if (per_cg < 1) per_cg = 1;
    dim3 grid((unsigned)(per_cg * n_cg), 1, 1);
    if (cv)
        hipLaunchKernelGGL((wino3d_cv_kernel<CT>), grid, dim3(64 * WN_WAVES), 0, stream, p, *cv);
    else
        hipLaunchKernelGGL((wino3d_kernel<CT>), grid, dim3(64 * WN_WAVES), 0, stream, p);
    return (int)hipGetLastError();
}

// U = (G x G x G) g per (cout, cin) pair, written in the t16 packing with the 64 frequency points in place of the 27 taps:
// [xi = (xd*4 + xh)*4 + xw][cb][cout_pad][16], zero-padded to whole channel blocks.
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
                for (int xd = 0; xd < 4; ++xd) out[(long)((xd * 4 + xh) * 4 + xw) * pairs + idx] = u[xd];
            }
    }
}

}  // namespace

static int wino_fwd(const drc_tapconv_params* pp, const drc_costvol_src* cv, int cout_tiles_per_wave, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if ((!cv && !p.x) || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD <= 0 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (p.cout_pad <= 0 || (p.cout_pad & 15) || p.cb_in <= 0) return -2;
    const drc_tap_class& k = p.cls[0];
    if (p.n_classes != 1 || p.in_mul != 1 || p.out_mul != 1 || k.nd != 3 || k.nh != 3 || k.nw != 3 || k.sd != 1 || k.sh != 1 || k.sw != 1)
        return -4;
    if ((p.OD | p.OH | p.OW) & 1) return -4;                                   // whole 2x2x2 tiles only
    if (!cv && (int64_t)p.N * p.x_n_stride * 4 >= (1LL << 32)) return -5;      // 32-bit lane offsets over the whole batch
    if (cv) {
        if (!cv->left || !cv->right) return -1;
        if (cv->pad < 1 || cv->cbi <= 0 || p.cb_in != 2 * cv->cbi || cv->Wp != p.OW) return -2;
        if ((int64_t)p.N * cv->n_stride * 4 >= (1LL << 32)) return -5;
    }
    if ((int64_t)p.N * p.OD * p.OH * p.OW / 8 >= (1LL << 31) - 16 || (int64_t)64 * p.cb_in * p.cout_pad * 16 >= (1LL << 31)) return -5;
    const int ct = p.cout_pad / 16, CT = cout_tiles_per_wave;
    if ((CT != 1 && CT != 2) || ct % CT) return -2;
    hipStream_t s = (hipStream_t)stream;
    return CT == 2 ? launch<2>(p, cv, s) : launch<1>(p, cv, s);
}

extern "C" int drc_conv3d_k3_wino_fwd(const drc_tapconv_params* pp, int cout_tiles_per_wave, void* stream) {
    return wino_fwd(pp, nullptr, cout_tiles_per_wave, stream);
}

extern "C" int drc_conv3d_k3_wino_costvol_fwd(const drc_tapconv_params* pp, const drc_costvol_src* cv, int cout_tiles_per_wave, void* stream) {
    if (!cv) return -1;
    return wino_fwd(pp, cv, cout_tiles_per_wave, stream);
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
