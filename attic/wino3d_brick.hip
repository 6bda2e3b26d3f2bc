// wino3d_brick.hip -- EXPERIMENT (round 2): the Winograd F(2x2x2,3x3x3) kernel of csrc/wino3d.hip with its input staged per BLOCK.
//
// csrc/wino3d.hip is bound by the vector-memory path: every lane loads its tile's 4x4x4 patch itself (32 float4 per step, two row
// phases ahead, 128 registers of landing space), each input voxel is fetched ~8x per CU and 5-9x from L2.  Here the 64 tiles of a
// block (4 waves x 16 tiles, consecutive in (n, dt, ht, wt) order) share one "brick" per step (depth frequency, channel block): the
// padded input rows their patches touch -- 2 slices x (2 rows per tile row + 2 per slab touched) x (W+2) voxels -- are copied ONCE by
// LDS-DMA (global_load_lds, no registers, issued a whole step ahead into the other of two buffers) and the patch rows are read from
// LDS just before their butterflies.  Layout of a slice buffer: [quad g][slot][x parity][x / 2] float4, so that the 16 lanes of a
// quad group read 16 consecutive float4 (tile j -> x = 2 wt + w) and wave g stages plane g with lane-contiguous DMA writes.
// The partial depth inverses are kept as two running sums in registers (no 96 KB parking slab), the weight ring has two slabs.
//   LDS: ring 2 x 16 KB + bricks 2 x 2 x SLOTS x (W+2) x 64 B  (28x28 maps: 32 + 120 KB; 14x14: 32 + 104 KB).
// Built into a variant library with tools/build_variant2.sh; falls back to the register-patch kernel for shapes it does not take.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/disprcnn_hip.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define WB_WAVES 4

namespace brick {

template <int CT, int VPR>
__global__ __launch_bounds__(64 * WB_WAVES) void wino3d_brick_kernel(const drc_tapconv_params p, int SLOTS) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kSlabF = 8 * CT * 256;                          // floats per weight slab (8 frequency points x CT*16 couts x 16 ch)
    float* w_ring = (float*)smem;                                 // [2][8][CT][256]
    // a slice buffer is a sequence of 1 KB chunks [quad g][16 voxels] -- one LDS-DMA instruction each (lanes (g, vv): the four quads
    // of a voxel sit in one 64-byte line, so an instruction touches 16 lines) -- per (slot, x parity) for 28-wide maps (15 voxels per
    // parity) and per slot for 14-wide maps (8 + 8)
    constexpr int CPS = VPR / 2 > 8 ? 2 : 1;                      // chunks per slot
    char* brick0 = smem + 2 * kSlabF * 4;                         // [buf 2][slice 2][SLOTS * CPS][g 4][16] float4
    const unsigned slice_bytes = (unsigned)(SLOTS * CPS * 1024);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const int g = lane >> 4;

    const int TD = p.OD >> 1, TH = p.OH >> 1, TW = p.OW >> 1;
    const int tiles = p.N * TD * TH * TW;
    const int rows_total = p.N * TD * TH;
    const int groups = (tiles + 15) >> 4;
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
    const int chunks = (groups + WB_WAVES - 1) / WB_WAVES;
    const int rounds = (chunks + nbk - 1) / nbk;
    const int ct0 = cg * CT;
    const int w_cb = p.cout_pad * 16;
    const int w_xi = w_cb * p.cb_in;

    // ---- per-round geometry.  Tile of this lane; LDS byte offset of its patch origin inside a slice buffer (plane g, slot of its
    // tile row, column wt); byte offsets (relative to a slice's (cb, depth) base) of the eight 64-voxel DMA chunks of plane `wave`.
    struct Geo { int n, dt, ht, wt; bool valid; unsigned lds; };
    auto geo_of = [&](int round) __attribute__((always_inline)) {
        Geo q;
        const int chunk = round * nbk + pos;
        int grp = chunk * WB_WAVES + wave;
        const bool active = grp < groups;
        if (!active) grp = groups - 1;
        int tile = grp * 16 + j;
        q.valid = active && tile < tiles;
        if (tile >= tiles) tile = tiles - 1;
        const int R0 = (chunk * 64 < tiles ? chunk * 64 : tiles - 1) / TW;
        const int R = tile / TW;
        q.wt = tile - R * TW;
        int t = R;
        q.ht = t % TH; t /= TH;
        q.dt = t % TD;
        q.n = t / TD;
        int slot = 2 * (R - R0) + 2 * (R / TH - R0 / TH);
        slot = slot < 0 ? 0 : (slot > SLOTS - 4 ? SLOTS - 4 : slot);
        q.lds = (unsigned)(slot * CPS * 1024 + g * 256 + q.wt * 16);
        return q;
    };
    constexpr int kChunks = 8;                                     // DMA chunks per wave and slice (SLOTS * CPS <= 32)
    auto dma_offsets = [&](int round, unsigned (&off)[kChunks]) __attribute__((always_inline)) {
        const int chunk = round * nbk + pos;
        const int R0 = (chunk * 64 < tiles ? chunk * 64 : tiles - 1) / TW;
        const int vv = lane & 15;
#pragma unroll
        for (int k = 0; k < kChunks; ++k) {
            const int c = wave + 4 * k;                            // chunk of this wave
            int s = CPS == 2 ? c >> 1 : c;
            const int par = CPS == 2 ? (c & 1) : (vv >> 3);
            int i = CPS == 2 ? vv : (vv & 7);
            i = i > VPR / 2 - 1 ? VPR / 2 - 1 : i;
            const int x = 2 * i + par;
            if (s > SLOTS - 1) s = SLOTS - 1;
            // slot -> (tile row relative to R0, patch row h): two slots per tile row plus two per slab touched
            int left = TH - R0 % TH, base = 0, rel = 0, h = 0;
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int span = 2 * left + 2;
                if (s < span || it == 2) {
                    int rr = s >> 1;
                    rr = rr > left - 1 ? left - 1 : rr;
                    rel = base + rr;
                    h = s - 2 * rr;
                    h = h > 3 ? 3 : h;
                    break;
                }
                s -= span; base += left; left = TH;
            }
            int R = R0 + rel;
            R = R > rows_total - 1 ? rows_total - 1 : R;
            int t = R;
            const int ht = t % TH; t /= TH;
            const int dt = t % TD;
            const int n = t / TD;
            off[k] = (unsigned)((n * p.x_n_stride + (int64_t)(2 * dt) * p.x_d_stride + (int64_t)(2 * ht + h) * p.x_h_stride + x * 16 + g * 4) * 4);
        }
    };
    auto slice_a = [](int xd) { return xd == 0 ? 0 : (xd == 2 ? 2 : 1); };
    auto slice_b = [](int xd) { return xd == 2 ? 1 : (xd == 3 ? 3 : 2); };
    // stage the brick of step (xd, cb) into buffer `buf`: wave `wave` copies plane g = wave of both slices
    auto dma_step = [&](int buf, int xd, int cb, const unsigned (&off)[kChunks], int nchunks) __attribute__((always_inline)) {
        const char* sa = (const char*)(p.x + (int64_t)cb * p.x_cb_stride + (int64_t)slice_a(xd) * p.x_d_stride);
        const char* sb = (const char*)(p.x + (int64_t)cb * p.x_cb_stride + (int64_t)slice_b(xd) * p.x_d_stride);
        char* da = brick0 + (unsigned)(buf * 2) * slice_bytes + (unsigned)(wave * 1024);
        char* db = da + slice_bytes;
#pragma unroll
        for (int k = 0; k < kChunks; ++k)
            if (wave + 4 * k < nchunks) {                          // wave-uniform
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(sa + off[k]), LDS_PTR(da + k * 4096), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(sb + off[k]), LDS_PTR(db + k * 4096), 16, 0, 0);
            }
    };
    // one h-row of the two slices of a step from the brick (8 float4): compile-time offsets for (h, w)
    auto read_row = [&](f32x4 (&ra)[4], f32x4 (&rb)[4], const char* ba, int h) __attribute__((always_inline)) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int o = CPS == 2 ? h * 2048 + (w & 1) * 1024 + (w >> 1) * 16 : h * 1024 + ((w & 1) * 8 + (w >> 1)) * 16;
            ra[w] = *(const f32x4*)(ba + o);
            rb[w] = *(const f32x4*)(ba + slice_bytes + o);
        }
    };
    auto load_row_global = [&](f32x4 (&ra)[4], f32x4 (&rb)[4], const char* sa, const char* sb, int h, unsigned xo) __attribute__((always_inline)) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            ra[w] = *(const f32x4*)(sa + ((int64_t)h * p.x_h_stride + w * 16) * 4 + xo);
            rb[w] = *(const f32x4*)(sb + ((int64_t)h * p.x_h_stride + w * 16) * 4 + xo);
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
#pragma unroll
        for (int w = 0; w < 4; ++w) asm volatile("" : "+v"(t[w]));
    };

    // ---- weight ring, two slabs: a slab is rewritten AFTER the barrier that ends the half step which read it
    constexpr int kFill = 2 * CT;
    int fill_off[kFill];
#pragma unroll
    for (int q = 0; q < kFill; ++q) {
        const int e = q * 256 + (int)threadIdx.x;
        fill_off[q] = (e / (64 * CT)) * w_xi + (e % (64 * CT)) * 4;
    }
    const float* wbase = p.w + ct0 * 256;
    f32x4 fill[kFill];
    int f_xd = 0, f_cb = 0, f_hf = 0;
    auto fill_load = [&]() __attribute__((always_inline)) {
        const float* src = wbase + (f_xd * 16 + f_hf * 8) * w_xi + f_cb * w_cb;
#pragma unroll
        for (int q = 0; q < kFill; ++q) fill[q] = *(const f32x4*)(src + fill_off[q]);
        if (++f_hf == 2) { f_hf = 0; if (++f_cb == p.cb_in) { f_cb = 0; f_xd = (f_xd + 1) & 3; } }
    };
    auto fill_store = [&](int slab) __attribute__((always_inline)) {
        float* dst = w_ring + slab * kSlabF;
#pragma unroll
        for (int q = 0; q < kFill; ++q) *(f32x4*)(dst + (q * 256 + (int)threadIdx.x) * 4) = fill[q];
    };
    auto load_w = [&](f32x4 (&wf)[4][CT], int slab, int row) __attribute__((always_inline)) {
#pragma unroll
        for (int xw = 0; xw < 4; ++xw)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) wf[xw][ct] = *(const f32x4*)(w_ring + slab * kSlabF + (((row & 1) * 4 + xw) * CT + ct) * 256 + j * 16 + g * 4);
    };

    f32x4 acc[4][4][CT];
    f32x4 o0[4 * CT], o1[4 * CT];
#define WB_MFMA_ROW(XH, WF)                                                                            \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                      \
        _Pragma("unroll") for (int xw = 0; xw < 4; ++xw)                                               \
            _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) {                                        \
                const f32x4 z4_ = {0.f, 0.f, 0.f, 0.f};                                                \
                acc[XH][xw][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(WF[xw][ct][s], v[XH][xw][s], FIRST && s == 0 ? z4_ : acc[XH][xw][ct], 0, 0, 0); \
            }

    struct Cursor { int round, xd, cb; };
    auto advance = [&](Cursor c) __attribute__((always_inline)) {
        if (++c.cb == p.cb_in) { c.cb = 0; if (++c.xd == 4) { c.xd = 0; ++c.round; } }
        return c;
    };
    const int nchunks = SLOTS * CPS;

    // ---- prologue: B fragments of step 0 straight from global memory; brick of step 1 in flight; weights of half steps 0, 1
    fill_load();
    fill_store(0);
    fill_load();
    Cursor c0 = {0, 0, 0};
    Cursor c1 = advance(c0);
    Geo geo0 = geo_of(0);
    Geo geo1 = c1.round < rounds ? geo_of(c1.round) : geo0;
    unsigned dma_off[kChunks];
    int dma_round = c1.round < rounds ? c1.round : 0;
    dma_offsets(dma_round, dma_off);
    dma_step(1, c1.xd, c1.cb, dma_off, nchunks);                   // step s uses buffer s & 1
    f32x4 v[4][4], tn[4][4], wfA[4][CT], wfB[4][CT];
    f32x4 r0a[4], r0b[4], r1a[4], r1b[4];
    {
        const unsigned xo = (unsigned)((geo0.n * p.x_n_stride + (int64_t)(2 * geo0.dt) * p.x_d_stride + (int64_t)(2 * geo0.ht) * p.x_h_stride +
                                        (int64_t)(2 * geo0.wt) * 16 + g * 4) * 4);
        const char* sa = (const char*)(p.x + (int64_t)slice_a(0) * p.x_d_stride);
        const char* sb = (const char*)(p.x + (int64_t)slice_b(0) * p.x_d_stride);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            load_row_global(r0a, r0b, sa, sb, h, xo);
            bfly_row(tn[h], r0a, r0b, -1.f);
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            v[0][w] = tn[0][w] - tn[2][w]; v[1][w] = tn[1][w] + tn[2][w]; v[2][w] = tn[2][w] - tn[1][w]; v[3][w] = tn[1][w] - tn[3][w];
        }
    }
    int stepno = 0;                                                // parity = brick buffer of the step whose MFMAs run
    int hs = 0;                                                    // half step counter (weight slab = hs & 1)

    // one step: MFMAs of c0; B fragments of c1 from its brick (landed: waited for at the step boundary); DMA of c2's brick
    auto do_step = [&](auto first_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        Cursor c2 = advance(c1);
        Geo geo2 = geo1;
        const float sgn = c1.xd == 1 ? 1.f : -1.f;
        // ---- step boundary: own DMA chunks and weight fills landed, everyone's LDS reads of the previous step done
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        fill_store((hs + 1) & 1);
        fill_load();
        if (c2.round != c1.round && c2.round < rounds) { geo2 = geo_of(c2.round); dma_offsets(c2.round, dma_off); }
        dma_step(stepno & 1, c2.xd, c2.cb, dma_off, nchunks);      // c2 = step stepno + 2: same parity as the step that just ended... (see below)
        const char* ba = brick0 + (unsigned)(((stepno + 1) & 1) * 2) * slice_bytes + geo1.lds;
        load_w(wfA, hs & 1, 0);
        read_row(r0a, r0b, ba, 0);
        load_w(wfB, hs & 1, 1);
        read_row(r1a, r1b, ba, 1);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            v[1][w] = tn[1][w] + tn[2][w]; v[2][w] = tn[2][w] - tn[1][w]; v[3][w] = tn[1][w] - tn[3][w];
            asm volatile("" : "+v"(v[1][w]), "+v"(v[2][w]), "+v"(v[3][w]));
        }
        WB_MFMA_ROW(0, wfA)
        bfly_row(tn[0], r0a, r0b, sgn);
        __builtin_amdgcn_sched_barrier(0);
        read_row(r0a, r0b, ba, 2);
        WB_MFMA_ROW(1, wfB)
        bfly_row(tn[1], r1a, r1b, sgn);
        __builtin_amdgcn_sched_barrier(0);
        ++hs;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        fill_store((hs + 1) & 1);
        fill_load();
        load_w(wfA, hs & 1, 2);
        read_row(r1a, r1b, ba, 3);
        load_w(wfB, hs & 1, 3);
        WB_MFMA_ROW(2, wfA)
        bfly_row(tn[2], r0a, r0b, sgn);
        __builtin_amdgcn_sched_barrier(0);
        WB_MFMA_ROW(3, wfB)
        bfly_row(tn[3], r1a, r1b, sgn);
        __builtin_amdgcn_sched_barrier(0);
        ++hs;
#pragma unroll
        for (int w = 0; w < 4; ++w) v[0][w] = tn[0][w] - tn[2][w];
        c0 = c1; c1 = c2; geo0 = geo1; geo1 = geo2;
        ++stepno;
    };
    auto phase_end = [&](int xd_, const Geo& geo) __attribute__((always_inline)) {
        f32x4 inv0[2][CT], inv1[2][CT];
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
#undef WB_MFMA_ROW
}

}  // namespace brick
