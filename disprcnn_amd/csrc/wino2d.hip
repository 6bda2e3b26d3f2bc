// wino2d.hip -- Conv2d 3x3, stride 1, dilation 1, pad 1 (+BN, +residual, +ReLU) as Winograd F(2x2, 3x3) on the fp32 matrix cores
// (gfx950 / CDNA4): the 2D sibling of wino3d.hip, same machinery without the depth dimension.
//
//   reference: feature_extraction's convbn 3x3 layers (submodule.py:13-17,24-49,62-88) and the 3x3 convolutions of the
//   ResNet bottlenecks / FPN outputs (backbone/resnet.py, backbone/fpn.py)
//
// A 2x2 output tile costs 16 multiplies per (cin, cout) pair instead of 4*9 = 36 (2.25x fewer MFMAs).
//   Y = A^T [ (G g G^T) . (B^T d B) ] A,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],
//   A^T = [1 1 1 0; 0 1 -1 -1].
// A wave owns 16 tiles (the N dimension of the 16x16x4 MFMA) x CT*16 couts; lane (tile j, g) transforms channels 4g..4g+3 of
// its tile's 4x4 input patch in registers.  A step is one 16-channel block: 16 float4 loads per lane (rows issued two row
// phases ahead), the w and h butterflies -> 16 B fragments, each feeding CT*4 MFMAs against U[xi][cb][cout][16]
// (drc_pack_weights_wino2d).  The block's four waves run in lock step and share the weights of a half step (8 frequency
// points) through a three-slab LDS ring, positions are numbered XCD by XCD, the first channel block of a tile group issues its
// MFMAs with C = 0 -- all as in wino3d.hip.  After the last channel block the 16 x CT accumulators are inverse-transformed to
// the 2x2 outputs and go through the usual epilogue.
// Odd OH / OW (round 3): the last tile row / column keeps output row / column 0 only.  Its patch row 3 (column 3) lies one past the
// zero halo -- the first row of the next plane, or the slack behind the tensor -- and may hold anything: in F(2,3) input row 3 enters
// frequency point 3 only (v3 = d1 - d3), which enters output row 1 only (y1 = m1 - m2 - m3), the one that is not stored.  The caller
// guarantees (Wp + 2) * 64 readable bytes behind the last plane (engine.Blocked's slack).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define W2_WAVES 4

namespace {

template <int CT, int DIL>      // DIL: dilation as a compile-time constant (as a run-time scalar it cost the undilated layers 36 spilled SGPRs, 5-10 %)
__global__ __launch_bounds__(64 * W2_WAVES) void wino2d_kernel(const drc_tapconv_params p) {
    // transformed weights of a half step (8 frequency points x CT*16 couts x 16 channels), ring of three, shared by the block
    __shared__ __attribute__((aligned(16))) float w_ring[3][8][CT][256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const int g = lane >> 4;

    const drc_tap_class cls = p.cls[0];
    // Dilation d (round 3; the feature CNN's layer4, d = 2): a dilated 3x3 convolution is d*d independent undilated ones on the
    // sub-grids (a, b) + d*(i, j), so a tile is (n, a, b, ht, wt): patch rows / columns d apart, the 2x2 outputs d apart.  The host
    // admits d > 1 only when 2d divides OH and OW (every sub-grid has whole tiles).
    constexpr int dil = DIL;
    const int TH = dil > 1 ? p.OH / (2 * dil) : (p.OH + 1) >> 1;   // d = 1, odd maps: the last tile row / column is half used (see unit_end)
    const int TW = dil > 1 ? p.OW / (2 * dil) : (p.OW + 1) >> 1;
    const int tiles = p.N * dil * dil * TH * TW;
    const int groups = (tiles + 15) >> 4;
    // block -> (cout group, position), rounds of four tile groups: see wino3d.hip
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
    const int chunks = (groups + W2_WAVES - 1) / W2_WAVES;
    const int rounds = (chunks + nbk - 1) / nbk;
    if (rounds == 0) return;
    const int ct0 = cg * CT;
    const int w_cb = p.cout_pad * 16;              // floats per (xi, cb)
    const int w_xi = w_cb * p.cb_in;               // floats per frequency point

    // lane geometry of a round: byte offset of the 4x4 patch origin (logical voxel 2t-1 = padded 2t + first), channels 4g..4g+3
    struct Geo { unsigned xo; int n, ht, wt; bool valid; };      // ht, wt: first output row / column of the tile, in units of 2 rows (d = 1)
    auto geo_of = [&](int round) __attribute__((always_inline)) {
        Geo q;
        int grp = (round * nbk + pos) * W2_WAVES + wave;
        const bool active = grp < groups;
        if (!active) grp = groups - 1;
        int tile = grp * 16 + j;
        q.valid = active && tile < tiles;
        if (tile >= tiles) tile = tiles - 1;
        const int wt = tile % TW; tile /= TW;
        const int ht = tile % TH; tile /= TH;
        const int sub = tile % (dil * dil);                 // sub-grid (a, b) = (sub / dil, sub % dil); 0 for d = 1
        q.n = tile / (dil * dil);
        const int a = sub / dil, b = sub - a * dil;
        q.ht = 2 * dil * ht + a;                            // first output row / column of the tile
        q.wt = 2 * dil * wt + b;
        q.xo = (unsigned)((q.n * p.x_n_stride + (int64_t)cls.dd0 * p.x_d_stride + (q.ht + cls.dh0) * p.x_h_stride +
                           (int64_t)(q.wt + cls.dw0) * 16 + g * 4) * 4);
        return q;
    };

    // one h-row of a step's patch (4 float4) and its w butterfly
    auto load_row = [&](f32x4 (&r)[4], const char* s, int h, unsigned xo) __attribute__((always_inline)) {
#pragma unroll
        for (int w = 0; w < 4; ++w) r[w] = *(const f32x4*)(s + ((int64_t)(h * dil) * p.x_h_stride + w * dil * 16) * 4 + xo);
    };
    auto bfly_row = [&](f32x4 (&t)[4], const f32x4 (&d)[4]) __attribute__((always_inline)) {
        t[0] = d[0] - d[2]; t[1] = d[1] + d[2]; t[2] = d[2] - d[1]; t[3] = d[1] - d[3];
        // pinned here (see wino3d.hip): LLVM otherwise sinks the butterflies behind the MFMA phases
#pragma unroll
        for (int w = 0; w < 4; ++w) asm volatile("" : "+v"(t[w]));
    };

    // ---- weight ring.  Half step hs = (cb, half) uses frequency points half*8 + 0..7 of block cb; the sequence repeats every
    // 2*cb_in half steps whatever the round.  Thread t copies float4 e = q*256 + t of the slab, q < 2*CT.
    constexpr int kFill = 2 * CT;
    int fill_off[kFill];
#pragma unroll
    for (int q = 0; q < kFill; ++q) {
        const int e = q * 256 + (int)threadIdx.x;
        // Inside a [cout 16][16 channels] tile the ring holds float4 (cout j, channel group g) at position g*16 + j -- the order the MFMA
        // fragments are read in (lanes j = 0..15 of a group: consecutive 16-byte words, no bank conflict; in the packed [cout][16] order
        // their ds_read_b128 hit four banks four ways: SQ_LDS_BANK_CONFLICT 40 % of the LDS-active cycles, profiles/r4_configB_pmc.md).
        // The permutation costs nothing: it is the global address this thread fetches from (the weights are cache-resident).
        const int r_ = e % (64 * CT), c_ = r_ & 63;
        fill_off[q] = (e / (64 * CT)) * w_xi + (r_ >> 6) * 256 + (c_ & 15) * 16 + (c_ >> 4) * 4;
    }
    const float* wbase = p.w + ct0 * 256;
    f32x4 fill[kFill];
    int f_cb = 0, f_hf = 0;                        // half step the next fill_load fetches
    auto fill_load = [&]() __attribute__((always_inline)) {
        const float* src = wbase + (f_hf * 8) * w_xi + f_cb * w_cb;
#pragma unroll
        for (int q = 0; q < kFill; ++q) fill[q] = *(const f32x4*)(src + fill_off[q]);
        if (++f_hf == 2) { f_hf = 0; if (++f_cb == p.cb_in) f_cb = 0; }
    };
    auto fill_store = [&](int slab) __attribute__((always_inline)) {
        float* dst = &w_ring[slab][0][0][0];
#pragma unroll
        for (int q = 0; q < kFill; ++q) *(f32x4*)(dst + (q * 256 + (int)threadIdx.x) * 4) = fill[q];
    };
    // half-step boundary hs: publish the weights of hs+1, wait for everyone, fetch the weights of hs+2 (see wino3d.hip)
    auto boundary = [&](int hs) __attribute__((always_inline)) {
        fill_store((hs + 1) % 3);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        fill_load();
    };
    auto load_w = [&](f32x4 (&wf)[4][CT], int slab, int row) __attribute__((always_inline)) {
#pragma unroll
        for (int xw = 0; xw < 4; ++xw)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) wf[xw][ct] = *(const f32x4*)&w_ring[slab][(row & 1) * 4 + xw][ct][(g * 16 + j) * 4];
    };

    f32x4 acc[4][4][CT];
    // FIRST (channel block 0): C = 0; two copies of the step body, not a branch (see wino3d.hip).  s outermost: 4*CT independent
    // accumulators between dependent MFMAs.
#define W2_MFMA_ROW(XH, WF)                                                                            \
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
    struct Cursor { int round, cb; };
    auto advance = [&](Cursor c) __attribute__((always_inline)) {
        if (++c.cb == p.cb_in) { c.cb = 0; ++c.round; }
        return c;
    };
    auto block_of = [&](const Cursor& c) __attribute__((always_inline)) {
        return (const char*)(p.x + (int64_t)c.cb * p.x_cb_stride);
    };
    Cursor c0 = {0, 0};                    // the step whose MFMAs run
    Cursor c1 = advance(c0);               // the step whose rows are being transformed (rows 2, 3 still loading)
    Geo geo0 = geo_of(0);
    Geo geo1 = c1.round != c0.round && c1.round < rounds ? geo_of(c1.round) : geo0;
    f32x4 v[4][4], tn[4][4], wfA[4][CT], wfB[4][CT];
    f32x4 r0[4], r1[4], r2[4], r3[4];      // the four h-rows in flight
    {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            load_row(r0, block_of(c0), h, geo0.xo);
            bfly_row(tn[h], r0);
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            v[0][w] = tn[0][w] - tn[2][w]; v[1][w] = tn[1][w] + tn[2][w]; v[2][w] = tn[2][w] - tn[1][w]; v[3][w] = tn[1][w] - tn[3][w];
        }
        load_row(r0, block_of(c1), 0, geo1.xo);
        load_row(r1, block_of(c1), 1, geo1.xo);
    }

    int slab = 0;
    // one channel-block step: its MFMAs in four row phases of 16*CT MFMAs, with -- in their shadow -- the w butterflies of the next
    // step's rows (loaded two phases earlier), the h butterfly of this step's rows 1..3 and the loads of the rows two phases ahead
    auto do_step = [&](auto first_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        Cursor c2 = advance(c1);
        Geo geo2 = geo1;
        if (c2.round != c1.round && c2.round < rounds) geo2 = geo_of(c2.round);
        const char* s1 = block_of(c1);
        const char* s2 = block_of(c2);
        const int slab_a = slab, slab_b = slab == 2 ? 0 : slab + 1;
        slab = slab_b == 2 ? 0 : slab_b + 1;

        boundary(slab_a);
        load_w(wfA, slab_a, 0);
        load_row(r2, s1, 2, geo1.xo);
        load_w(wfB, slab_a, 1);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            v[1][w] = tn[1][w] + tn[2][w]; v[2][w] = tn[2][w] - tn[1][w]; v[3][w] = tn[1][w] - tn[3][w];
            asm volatile("" : "+v"(v[1][w]), "+v"(v[2][w]), "+v"(v[3][w]));
        }
        bfly_row(tn[0], r0);
        W2_MFMA_ROW(0, wfA)
        __builtin_amdgcn_sched_barrier(0);
        load_row(r3, s1, 3, geo1.xo);
        bfly_row(tn[1], r1);
        W2_MFMA_ROW(1, wfB)
        __builtin_amdgcn_sched_barrier(0);
        boundary(slab_b);
        load_w(wfA, slab_b, 2);
        load_row(r0, s2, 0, geo2.xo);
        load_w(wfB, slab_b, 3);
        bfly_row(tn[2], r2);
        W2_MFMA_ROW(2, wfA)
        __builtin_amdgcn_sched_barrier(0);
        load_row(r1, s2, 1, geo2.xo);
        bfly_row(tn[3], r3);
        W2_MFMA_ROW(3, wfB)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int w = 0; w < 4; ++w) v[0][w] = tn[0][w] - tn[2][w];
        c0 = c1; c1 = c2; geo0 = geo1; geo1 = geo2;
    };
    // end of a tile group: A^T . A (4x4 -> 2x2) of its accumulators, then the epilogue
    auto unit_end = [&](const Geo& geo) __attribute__((always_inline)) {
        if (!geo.valid) return;
        const int64_t yo = p.y_off0 + (int64_t)geo.n * p.y_n_stride + (int64_t)geo.ht * p.y_h_stride + (int64_t)geo.wt * 16 + g * 4;
        const int64_t ro = p.r_off0 + (int64_t)geo.n * p.r_n_stride + (int64_t)geo.ht * p.r_h_stride + (int64_t)geo.wt * 16 + g * 4;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const f32x4 bn_sc = *(const f32x4*)(p.scale + (ct0 + ct) * 16 + g * 4);
            const f32x4 bn_sh = *(const f32x4*)(p.shift + (ct0 + ct) * 16 + g * 4);
            f32x4 hh[2][4];
#pragma unroll
            for (int xw = 0; xw < 4; ++xw) {
                hh[0][xw] = acc[0][xw][ct] + acc[1][xw][ct] + acc[2][xw][ct];
                hh[1][xw] = acc[1][xw][ct] - acc[2][xw][ct] - acc[3][xw][ct];
            }
#pragma unroll
            for (int oh = 0; oh < 2; ++oh)
#pragma unroll
                for (int ow = 0; ow < 2; ++ow) {
                    if (geo.ht + oh * dil >= p.OH || geo.wt + ow * dil >= p.OW) continue;   // the unused half of an odd map's last tiles
                    f32x4 v_ = (ow == 0 ? hh[oh][0] + hh[oh][1] + hh[oh][2] : hh[oh][1] - hh[oh][2] - hh[oh][3]) * bn_sc + bn_sh;
                    if (p.res) v_ += *(const f32x4*)(p.res + ro + (oh * dil) * p.r_h_stride + ow * dil * 16 + (int64_t)(ct0 + ct) * p.r_cb_stride);
                    if (p.relu) { v_.x = fmaxf(v_.x, 0.f); v_.y = fmaxf(v_.y, 0.f); v_.z = fmaxf(v_.z, 0.f); v_.w = fmaxf(v_.w, 0.f); }
                    *(f32x4*)(p.y + yo + (oh * dil) * p.y_h_stride + ow * dil * 16 + (int64_t)(ct0 + ct) * p.y_cb_stride) = v_;
                }
        }
    };
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        const Geo geo = geo0;
        do_step(std::true_type{});
#pragma unroll 1
        for (int c = 1; c < p.cb_in; ++c) do_step(std::false_type{});
        unit_end(geo);
    }
#undef W2_MFMA_ROW
}

template <int CT, int DIL>
int launch(const drc_tapconv_params& p, hipStream_t stream) {
    constexpr int dil = DIL;
    const long tiles = dil > 1 ? (long)p.N * (p.OH / 2) * (p.OW / 2) : (long)p.N * ((p.OH + 1) / 2) * ((p.OW + 1) / 2);
    const long groups = (tiles + 15) / 16;
    const int n_cg = p.cout_pad / 16 / CT;
    // one block per CU; every cout group gets the same number of blocks
    long per_cg = 256 / n_cg;
    const long need = (groups + W2_WAVES - 1) / W2_WAVES;
    if (per_cg > need) per_cg = need;
    if (per_cg < 1) per_cg = 1;
    dim3 grid((unsigned)(per_cg * n_cg), 1, 1);
    hipLaunchKernelGGL((wino2d_kernel<CT, DIL>), grid, dim3(64 * W2_WAVES), 0, stream, p);
    return (int)hipGetLastError();
}

// U = (G x G) g per (cout, cin) pair in the t16 packing with the 16 frequency points in place of the 9 taps:
// [xi = xh*4 + xw][cb][cout_pad][16], zero-padded to whole channel blocks.
__global__ __launch_bounds__(256) void wino2d_weights_kernel(const float* __restrict__ w, int cout, int cin, int transposed, int flip,
                                                             float* __restrict__ out) {
    const int cb_n = (cin + 15) / 16, cout_pad = (cout + 15) / 16 * 16;
    const long pairs = (long)cb_n * cout_pad * 16;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < pairs; idx += (long)gridDim.x * 256) {
        long t = idx;
        const int c = (int)(t & 15); t >>= 4;
        const int co = (int)(t % cout_pad);
        const int cb = (int)(t / cout_pad);
        const int ci = cb * 16 + c;
        const bool live = co < cout && ci < cin;
        const float* src = w + (transposed ? ((long)ci * cout + co) : ((long)co * cin + ci)) * 9;
        float a[3][3], b[3][4];
#pragma unroll
        for (int k = 0; k < 9; ++k) a[k / 3][k % 3] = live ? src[flip ? 8 - k : k] : 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const float g0 = a[kh][0], g1 = a[kh][1], g2 = a[kh][2];
            b[kh][0] = g0; b[kh][1] = 0.5f * (g0 + g1 + g2); b[kh][2] = 0.5f * (g0 - g1 + g2); b[kh][3] = g2;
        }
#pragma unroll
        for (int xw = 0; xw < 4; ++xw) {
            const float g0 = b[0][xw], g1 = b[1][xw], g2 = b[2][xw];
            const float u[4] = {g0, 0.5f * (g0 + g1 + g2), 0.5f * (g0 - g1 + g2), g2};
#pragma unroll
            for (int xh = 0; xh < 4; ++xh) out[(long)(xh * 4 + xw) * pairs + idx] = u[xh];
        }
    }
}

}  // namespace

extern "C" int drc_conv2d_k3_wino_fwd(const drc_tapconv_params* pp, int cout_tiles_per_wave, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD != 1 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (p.cout_pad <= 0 || (p.cout_pad & 15) || p.cb_in <= 0) return -2;
    const drc_tap_class& k = p.cls[0];
    if (p.n_classes != 1 || p.in_mul != 1 || p.out_mul != 1 || k.nd != 1 || k.nh != 3 || k.nw != 3 || k.sh < 1 || k.sh != k.sw) return -4;
    if (k.sh > 1 && (p.OH % (2 * k.sh) || p.OW % (2 * k.sh))) return -4;      // dilated: whole tiles on every sub-grid
    if (((int64_t)p.N * p.x_n_stride + 2 * p.x_h_stride) * 4 >= (1LL << 32)) return -5;   // 32-bit lane offsets over the whole batch
    if ((int64_t)p.N * ((p.OH + 1) / 2) * ((p.OW + 1) / 2) >= (1LL << 31) - 16 || (int64_t)16 * p.cb_in * p.cout_pad * 16 >= (1LL << 31)) return -5;
    const int ct = p.cout_pad / 16, CT = cout_tiles_per_wave;
    if ((CT != 1 && CT != 2) || ct % CT) return -2;
    hipStream_t s = (hipStream_t)stream;
    if (k.sh == 1) return CT == 2 ? launch<2, 1>(p, s) : launch<1, 1>(p, s);
    if (k.sh == 2) return CT == 2 ? launch<2, 2>(p, s) : launch<1, 2>(p, s);
    if (k.sh == 4) return CT == 2 ? launch<2, 4>(p, s) : launch<1, 4>(p, s);
    return -4;
}

extern "C" int drc_pack_weights_wino2d(const float* w, int cout, int cin, int transposed, int flip, float* out, void* stream) {
    if (cout <= 0 || cin <= 0) return -2;
    if (!w || !out) return -1;
    const long pairs = (long)((cin + 15) / 16) * ((cout + 15) / 16 * 16) * 16;
    long blocks = (pairs + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wino2d_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, cout, cin, transposed, flip, out);
    return (int)hipGetLastError();
}
