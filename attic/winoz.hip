// winoz.hip -- stride-1 3x3x3 convolution (+BN, +residual, +ReLU): Winograd F(2x2, 3x3) IN-PLANE, direct taps with a sliding
// window in DEPTH, on the fp32 matrix cores (gfx950 / CDNA4).
//
//   reference: dres0/dres1, classifN[0], hourglass conv2/conv4 (stackhourglass.py:63-88, :14-20)
//
// Why not the full 3D transform (wino3d.hip: 64 multiplies per 2x2x2 tile, 3.4x fewer MFMAs than direct)?  Measured in round 2
// (tools/experiments/README.md): that kernel is bound by the vector-memory path, not by the matrix cores -- every 2x2x2 tile
// re-reads its 4x4x4 patch twice per slice pair (16 float4 patch loads per lane and output voxel-tile, served mostly by L2
// because 128 KB per step stream through a 32 KB L1), and no amount of hiding VALU / LDS work behind MFMAs helps.
// Here only the 3x3 in-plane part is transformed (16 multiplies per 2x2 tile and depth tap instead of 36: 2.25x fewer MFMAs than
// direct), and depth is handled like tapdirect.hip: a wave walks the input slices of a column of 16 in-plane tiles ONCE; the 16
// B fragments of a slice (one 4x4 patch per tile and 16-channel block = 16 float4 loads per lane) feed the three output slices
// they touch -- three accumulator sets (16 points x 16 couts each) rotated by unrolling the slice loop by 3.  Per step (slice,
// channel block): 16 patch loads for 192 MFMAs (wino3d: 32 for 128), no depth butterfly, no inverse-transform parking in LDS.
//   * transformed weights U[point][depth tap] of a half step (8 points x 3 taps x 16 couts x 16 channels = 24 KB) are shared by
//     the block's four waves through a three-slab LDS ring, waves in lock step (one s_barrier per half step) as in wino2d.hip;
//   * the next step's patch is loaded one step ahead and transformed (w then h butterflies, 128 VALU ops) a few operations per
//     frequency point in the shadow of the MFMAs; points run column by column so that the h butterfly overwrites B fragments in
//     place as soon as a column has been consumed;
//   * blocks take equal contiguous shares of the (4 tile groups, cout tile, output slice) units; a share that ends inside a column
//     costs that column one extra input slice.
// Needs even OH, OW (any OD).  Results differ from the direct kernels' by fp32 rounding only.
// Weights: drc_pack_weights_winoz, [cout tile][cb][half][8 points][3 depth taps][lane g*16+j][4]; point p = xw*4 + xh.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WZ_WAVES 4

namespace {

__global__ __launch_bounds__(64 * WZ_WAVES) void winoz_kernel(const drc_tapconv_params p) {
    __shared__ __attribute__((aligned(16))) f32x4 w_ring[3][8][3][64];           // [slab][point of the half step][depth tap][lane]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const int g = lane >> 4;

    const drc_tap_class cls = p.cls[0];
    const int TH = p.OH >> 1, TW = p.OW >> 1, D = p.OD;
    const int tiles = p.N * TH * TW;
    const int groups = (tiles + 15) >> 4;
    const int chunks = (groups + WZ_WAVES - 1) / WZ_WAVES;
    const int n_ct = p.cout_pad / 16;
    // block share of the (chunk, cout tile, output slice) units; cout tile fastest so that neighbouring blocks read the same input
    const long total = (long)chunks * n_ct * D;
    long cur = total * blockIdx.x / gridDim.x;
    const long end = total * (blockIdx.x + 1) / gridDim.x;

    // ---- weight ring: half step (cb, half) -> slab of 8 points x 3 taps x 64 lanes float4 = 1536 float4; 6 per thread
    // Staged through 3 registers at a time: part 0 (float4 0..767 of the slab) is requested at a boundary and stored in the middle
    // of the half step, part 1 requested there and stored at the next boundary (12 registers instead of 24).
    f32x4 fill[3];
    int f_cb = 0, f_hf = 0, f_ct = 0;
    auto fill_load = [&](int part) __attribute__((always_inline)) {
        const f32x4* src = (const f32x4*)p.w + ((long)(f_ct * p.cb_in + f_cb) * 2 + f_hf) * 1536 + part * 768;
#pragma unroll
        for (int q = 0; q < 3; ++q) fill[q] = src[q * 256 + (int)threadIdx.x];
        if (part == 1 && ++f_hf == 2) { f_hf = 0; if (++f_cb == p.cb_in) f_cb = 0; }
    };
    auto fill_store = [&](int slab, int part) __attribute__((always_inline)) {
        f32x4* dst = &w_ring[slab][0][0][0] + part * 768;
#pragma unroll
        for (int q = 0; q < 3; ++q) dst[q * 256 + (int)threadIdx.x] = fill[q];
    };
    int slab = 0;                                  // slab holding the weights of the half step about to run
    // half-step boundary: publish the next half step's weights, wait for everyone (which also guarantees that nobody still reads
    // the slab overwritten at the NEXT boundary), fetch the one after
    auto boundary = [&]() __attribute__((always_inline)) {
#ifdef WZ_ABL_NOBAR
        return;
#endif
        fill_store(slab == 2 ? 0 : slab + 1, 1);           // second part of the NEXT half step's slab (first part: mid_fill of the last half step)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        fill_load(0);                                       // first part of the slab after the next
    };
    auto mid_fill = [&]() __attribute__((always_inline)) { // in the middle
#ifdef WZ_ABL_NOBAR
        return;
#endif
        //  of a half step: slab two ahead, part 0 -> LDS, part 1 requested
        fill_store(slab == 0 ? 2 : slab - 1, 0);
        fill_load(1);
    };

    f32x4 acc[3][16];                              // [output slice mod 3][point p = xw*4 + xh]
    f32x4 v[16];                                   // B fragments of the current step, point-major
    f32x4 rt[4][4];                                // the next step's patch [h][w]; the w butterfly overwrites it in place ([h][xw])
    f32x4 c3[4];                                   // column 3 of the transformed patch, parked while the following patch lands in rt

#pragma unroll 1
    while (cur < end) {
        // ---- one segment: output slices [od_lo, od_hi) of one (chunk, cout tile)
        const long unit = cur / D;
        const int od_lo = (int)(cur - unit * D);
        const int od_hi = (long)od_lo + (end - cur) < D ? od_lo + (int)(end - cur) : D;
        cur += od_hi - od_lo;
        const int ct = (int)(unit % n_ct);
        const int chunk = (int)(unit / n_ct);
        const int din_lo = od_lo > 0 ? od_lo - 1 : 0, din_hi = od_hi < D ? od_hi : D - 1;        // inclusive
        // lane geometry: patch origin of tile j of this wave's group (logical voxel 2t-1 = padded 2t + first), channels 4g..4g+3
        int grp = chunk * WZ_WAVES + wave;
        const bool active = grp < groups;
        if (!active) grp = groups - 1;
        int tile = grp * 16 + j;
        const bool valid = active && tile < tiles;
        if (tile >= tiles) tile = tiles - 1;
        const int wt = tile % TW; tile /= TW;
        const int ht = tile % TH;
        const int n = tile / TH;
        const unsigned xo = (unsigned)((n * p.x_n_stride + (2 * ht + cls.dh0) * p.x_h_stride + (int64_t)(2 * wt + cls.dw0) * 16 + g * 4) * 4);
        // the weight sequence restarts at (cb 0, half 0) of this segment's cout tile: refill the ring's head
        f_ct = ct; f_cb = 0; f_hf = 0;
        asm volatile("s_barrier" ::: "memory");            // everyone has finished reading the previous segment's slabs
        fill_load(0); fill_store(slab, 0);                 // half step 0 -> the current slab
        fill_load(1); fill_store(slab, 1);
        fill_load(0); fill_store(slab == 2 ? 0 : slab + 1, 0);     // half step 1, part 0 -> the next slab; part 1 travels to the first boundary
        fill_load(1);

        const f32x4 bn_sc = *(const f32x4*)(p.scale + ct * 16 + g * 4), bn_sh = *(const f32x4*)(p.shift + ct * 16 + g * 4);
        {
            float z_;
            asm volatile("v_mov_b32 %0, 0" : "=v"(z_));
            const f32x4 z4 = {z_, z_, z_, z_};
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[s][q] = z4;
        }
        auto slice_ptr = [&](int d_in, int cb) __attribute__((always_inline)) {       // real slice d_in sits at padded depth d_in + dd0 + 1
            return (const char*)(p.x + (int64_t)cb * p.x_cb_stride + (int64_t)(d_in + cls.dd0 + 1) * p.x_d_stride);
        };
        auto load_rows = [&](const char* s, int h0) __attribute__((always_inline)) {
#pragma unroll
            for (int h = h0; h < h0 + 2; ++h)
#pragma unroll
                for (int w = 0; w < 4; ++w) rt[h][w] = *(const f32x4*)(s + ((int64_t)h * p.x_h_stride + w * 16) * 4 + xo);
        };
        // butterflies, one float4 statement (4 VALU ops) per piece so that a piece can sit between two MFMAs; pinned where written
        // (LLVM otherwise sinks them to their use behind the MFMA runs)
        auto wbfly_piece = [&](int h, int k, const f32x4 (&d)[4]) __attribute__((always_inline)) {
            if (k == 0) rt[h][0] = d[0] - d[2];
            if (k == 1) rt[h][1] = d[1] + d[2];
            if (k == 2) rt[h][2] = d[2] - d[1];
            if (k == 3) rt[h][3] = d[1] - d[3];
            asm volatile("" : "+v"(rt[h][k]));
        };
        auto wbfly = [&](int h) __attribute__((always_inline)) {
            const f32x4 d[4] = {rt[h][0], rt[h][1], rt[h][2], rt[h][3]};
#pragma unroll
            for (int k = 0; k < 4; ++k) wbfly_piece(h, k, d);
        };
        auto hbfly_piece = [&](int xw, int k) __attribute__((always_inline)) {       // column xw (< 3) of the B fragments, in place
            if (k == 0) v[xw * 4 + 0] = rt[0][xw] - rt[2][xw];
            if (k == 1) v[xw * 4 + 1] = rt[1][xw] + rt[2][xw];
            if (k == 2) v[xw * 4 + 2] = rt[2][xw] - rt[1][xw];
            if (k == 3) v[xw * 4 + 3] = rt[1][xw] - rt[3][xw];
            asm volatile("" : "+v"(v[xw * 4 + k]));
        };
        auto hbfly = [&](int xw) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) hbfly_piece(xw, k);
        };
        auto park3 = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int h = 0; h < 4; ++h) c3[h] = rt[h][3];
        };
        auto hbfly3 = [&]() __attribute__((always_inline)) {
            v[12] = c3[0] - c3[2]; v[13] = c3[1] + c3[2]; v[14] = c3[2] - c3[1]; v[15] = c3[1] - c3[3];
        };
        // ---- segment prologue: B fragments of step (din_lo, cb 0); the patch of the following step in flight
        load_rows(slice_ptr(din_lo, 0), 0);
        load_rows(slice_ptr(din_lo, 0), 2);
#pragma unroll
        for (int h = 0; h < 4; ++h) wbfly(h);
#pragma unroll
        for (int xw = 0; xw < 3; ++xw) hbfly(xw);
        park3();
        hbfly3();
        int nd = din_lo, ncb = 0;                  // step cursor for the patch loads (clamped at the segment's last step)
        auto advance = [&]() __attribute__((always_inline)) {
            if (++ncb == p.cb_in) { ncb = 0; if (nd < din_hi) ++nd; else ncb = p.cb_in - 1; }
        };
        advance();
        load_rows(slice_ptr(nd, ncb), 0);
        load_rows(slice_ptr(nd, ncb), 2);
        advance();                                 // (nd, ncb) = the step whose patch is requested next: two steps ahead of the MFMAs
        bool pending_col3 = false;                 // h butterfly of column 3 of the NEXT fragments still to do (needs column 3 consumed)

        // epilogue of output slice od out of accumulator set A: in-plane inverse (4x4 -> 2x2), BN, residual, ReLU, stores.  Addresses =
        // uniform base of (cout tile, output slice) + the lane's 32-bit tile offset (+ row stride / 64-byte immediates).  The set is NOT
        // cleared: its next use is as the fresh set of a later slice, whose first channel block accumulates onto C = 0.
        const unsigned yv = (unsigned)((n * p.y_n_stride + (int64_t)(2 * ht) * p.y_h_stride + (int64_t)(2 * wt) * 16 + g * 4) * 4);
        const unsigned rv = (unsigned)((n * p.r_n_stride + (int64_t)(2 * ht) * p.r_h_stride + (int64_t)(2 * wt) * 16 + g * 4) * 4);
        auto finish = [&](int od, f32x4 (&A)[16]) __attribute__((always_inline)) {
#ifdef WZ_ABL_NOFINISH
            if (od != D + 7) return;
#endif
            if (!(valid && od >= od_lo && od < od_hi)) return;
            char* yb = (char*)(p.y + p.y_off0 + (int64_t)ct * p.y_cb_stride + (int64_t)od * p.y_d_stride);
            const char* rb = p.res ? (const char*)(p.res + p.r_off0 + (int64_t)ct * p.r_cb_stride + (int64_t)od * p.r_d_stride) : nullptr;
#pragma unroll
            for (int oh = 0; oh < 2; ++oh) {
                f32x4 hh[4];
#pragma unroll
                for (int xw = 0; xw < 4; ++xw)
                    hh[xw] = oh == 0 ? A[xw * 4 + 0] + A[xw * 4 + 1] + A[xw * 4 + 2] : A[xw * 4 + 1] - A[xw * 4 + 2] - A[xw * 4 + 3];
#pragma unroll
                for (int ow = 0; ow < 2; ++ow) {
                    f32x4 o_ = ow == 0 ? hh[0] + hh[1] + hh[2] : hh[1] - hh[2] - hh[3];
                    o_ = o_ * bn_sc + bn_sh;
                    if (rb) o_ += *(const f32x4*)(rb + rv + (unsigned)(oh * (int)p.r_h_stride * 4 + ow * 64));
                    if (p.relu) { o_.x = fmaxf(o_.x, 0.f); o_.y = fmaxf(o_.y, 0.f); o_.z = fmaxf(o_.z, 0.f); o_.w = fmaxf(o_.w, 0.f); }
                    *(f32x4*)(yb + yv + (unsigned)(oh * (int)p.y_h_stride * 4 + ow * 64)) = o_;
                }
            }
        };

        // one step = (input slice, channel block): 16 points x 3 depth taps x 4 k-steps = 192 MFMAs in two half steps of 8 points
        // (two columns each).  A0 / A1 / A2 = accumulator sets of output slices d+1 / d / d-1 (depth taps 0 / 1 / 2); FIRST (the
        // slice's first channel block): A0 is a fresh set and starts from C = 0.  Between the MFMAs of a point sit, one float4
        // statement per k-step: the w butterflies of the next step's patch (first half step), the h butterflies that overwrite the
        // consumed columns of the B fragments in place and the requests for the patch of the step after the next (second half).
        auto step = [&](auto first_tag, f32x4 (&A0)[16], f32x4 (&A1)[16], f32x4 (&A2)[16]) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const char* sn = slice_ptr(nd, ncb);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                boundary();
                f32x4 wq[2][3];
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) wq[0][kd] = w_ring[slab][0][kd][lane];
                if (half == 0 && pending_col3) hbfly3();
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int pt = half * 8 + q, cbuf = q & 1;
                    if (q + 1 < 8) {
#pragma unroll
                        for (int kd = 0; kd < 3; ++kd) wq[cbuf ^ 1][kd] = w_ring[slab][q + 1][kd][lane];
                    }
                    f32x4 dsrc[4];
                    const int row = q >> 1;                      // first half step: row `row` of the patch is butterflied at odd q
                    if (half == 0 && (q & 1)) { dsrc[0] = rt[row][0]; dsrc[1] = rt[row][1]; dsrc[2] = rt[row][2]; dsrc[3] = rt[row][3]; }
                    if (half == 1 && q == 4) park3();
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                        A0[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[cbuf][0][s], v[pt][s], FIRST && s == 0 ? z4 : A0[pt], 0, 0, 0);
                        A1[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[cbuf][1][s], v[pt][s], A1[pt], 0, 0, 0);
                        A2[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[cbuf][2][s], v[pt][s], A2[pt], 0, 0, 0);
#ifndef WZ_ABL_NOXFORM
                        if (half == 0 && (q & 1)) wbfly_piece(row, s, dsrc);
                        if (half == 1 && (q == 0 || q == 2 || q == 4)) hbfly_piece(q >> 1, s);
#endif
                        if (half == 1 && (q == 5 || q == 7) && s < 2) {
                            const int h = (q == 5 ? 0 : 2) + s;
#pragma unroll
#ifndef WZ_ABL_NOLOAD
                            for (int w = 0; w < 4; ++w) rt[h][w] = *(const f32x4*)(sn + ((int64_t)h * p.x_h_stride + w * 16) * 4 + xo);
#else
                            for (int w = 0; w < 4; ++w) rt[h][w] = (f32x4){(float)h, (float)w, (float)xo, 1.f};
#endif
                        }
                    }
                    if (q == 3) mid_fill();
                    __builtin_amdgcn_sched_barrier(0);
                }
                slab = slab == 2 ? 0 : slab + 1;
            }
        };

        // ---- walk the input slices; the three accumulator sets rotate statically (set index = output slice mod 3): input slice d
        // feeds output d+1 (depth tap 0), d (tap 1) and d-1 (tap 2); after it, output d-1 is complete
        auto slice = [&](int d, f32x4 (&A0)[16], f32x4 (&A1)[16], f32x4 (&A2)[16]) __attribute__((always_inline)) {
            step(std::true_type{}, A0, A1, A2);
            pending_col3 = true;
            advance();
#pragma unroll 1
            for (int cb = 1; cb < p.cb_in; ++cb) {
                step(std::false_type{}, A0, A1, A2);
                advance();
            }
            finish(d - 1, A2);
            if (d == D - 1) finish(d, A1);
        };
#pragma unroll 1
        for (int d = din_lo - din_lo % 3;;) {
            if (d >= din_lo) slice(d, acc[1], acc[0], acc[2]);
            if (++d > din_hi) break;
            if (d >= din_lo) slice(d, acc[2], acc[1], acc[0]);
            if (++d > din_hi) break;
            if (d >= din_lo) slice(d, acc[0], acc[2], acc[1]);
            if (++d > din_hi) break;
        }
    }
}

// U[point][depth tap] = (G x G) applied in-plane to every depth tap of the 3x3x3 kernel, in the order the kernel reads it:
// [cout tile][cb][half][8 points][3 depth taps][lane g*16 + j][4], point p = half*8 + p8 = xw*4 + xh, cout = tile*16 + j,
// channel = cb*16 + g*4 + 0..3; zero-padded to whole channel blocks / cout tiles.
__global__ __launch_bounds__(256) void winoz_weights_kernel(const float* __restrict__ w, int cout, int cin, int transposed, int flip,
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
        const float* src = w + (transposed ? ((long)ci * cout + co) : ((long)co * cin + ci)) * 27;
        const int ctile = co >> 4, jj = co & 15, gg = c >> 2, e = c & 3;
        float* dst = out + ((long)(ctile * cb_n + cb) * 2) * 1536 * 4 + (gg * 16 + jj) * 4 + e;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            float a[3][3];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int kk = kd * 9 + k;
                a[k / 3][k % 3] = live ? src[flip ? 26 - kk : kk] : 0.f;
            }
            float b[3][4];
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
                for (int xh = 0; xh < 4; ++xh) {
                    const int pt = xw * 4 + xh;
                    dst[(((long)(pt >> 3) * 8 + (pt & 7)) * 3 + kd) * 256] = u[xh];
                }
            }
        }
    }
}

}  // namespace

extern "C" int drc_conv3d_k3_winoz_fwd(const drc_tapconv_params* pp, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD <= 0 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (p.cout_pad <= 0 || (p.cout_pad & 15) || p.cb_in <= 0) return -2;
    const drc_tap_class& k = p.cls[0];
    if (p.n_classes != 1 || p.in_mul != 1 || p.out_mul != 1 || k.nd != 3 || k.nh != 3 || k.nw != 3 || k.sd != 1 || k.sh != 1 || k.sw != 1)
        return -4;
    if ((p.OH | p.OW) & 1) return -4;                                          // whole 2x2 tiles only
    if ((int64_t)p.N * p.x_n_stride * 4 >= (1LL << 32) || (int64_t)p.N * p.y_n_stride * 4 >= (1LL << 32) ||
        (p.res && (int64_t)p.N * p.r_n_stride * 4 >= (1LL << 32)))
        return -5;                                                             // 32-bit lane offsets over the whole batch
    const long tiles = (long)p.N * (p.OH / 2) * (p.OW / 2);
    if (tiles >= (1L << 31) - 16) return -5;
    const long chunks = ((tiles + 15) / 16 + WZ_WAVES - 1) / WZ_WAVES;
    const long total = chunks * (p.cout_pad / 16) * p.OD;
    long blocks = 256;                                      // one block per CU (three accumulator sets fill the register file)
    if (blocks > total) blocks = total;
    hipLaunchKernelGGL(winoz_kernel, dim3((unsigned)blocks), dim3(64 * WZ_WAVES), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

extern "C" int drc_pack_weights_winoz(const float* w, int cout, int cin, int transposed, int flip, float* out, void* stream) {
    if (cout <= 0 || cin <= 0) return -2;
    if (!w || !out) return -1;
    const long pairs = (long)((cin + 15) / 16) * ((cout + 15) / 16 * 16) * 16;
    long blocks = (pairs + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(winoz_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, cout, cin, transposed, flip, out);
    return (int)hipGetLastError();
}
