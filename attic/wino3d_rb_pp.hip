// wino3d_rb.hip -- stride-1 3x3x3 convolution (+BN, +residual, +ReLU) as Winograd F(2x2x2, 3x3x3) on the fp32 matrix cores,
// TWO waves per SIMD running one barrier interval apart ("ping-pong"), input staged per BLOCK as depth- and w-transformed ROWS
// ("row brick") in LDS (gfx950 / CDNA4; row brick: round 3, ping-pong: round 4).
//
//   reference: dres0/dres1, classifN[0], hourglass conv2 (stackhourglass.py:63-88, :14-20)
//
//   * a block of 8 waves (2 per SIMD, <= 256 registers each) owns 64 consecutive tiles: waves (tg, ct) -- tile group tg of 16
//     tiles (the N dimension of the 16x16x4 MFMA) x cout tile ct of 16 couts; waves w and w + 4 share a SIMD and a tile group;
//   * per step (depth frequency xd, channel block cb) the block stages the input ROWS its tiles touch ONCE: work item
//     (row slot, tile column wt, channel quad g) loads the row's four columns 2wt..2wt+3 of the two slices of xd (8 float4),
//     applies the depth butterfly (slice a +/- slice b) and the w butterfly and writes the four w-frequencies to LDS --
//     rows are shared by the two tile rows that overlap them, so these two butterflies run once per ROW instead of once
//     per tile, and each input voxel is loaded ~2.3x per CU and step instead of 8x;
//   * a step is two half steps of 32 MFMAs per wave: half hf covers the w-frequencies xw = 2hf, 2hf+1 of all four h-frequencies.
//     Its operands are read from LDS in an L segment (8 transformed-row reads + h butterfly, 8 weight reads from a two-slab ring
//     the block fills by LDS-DMA), its MFMAs run in the following M segment;
//   * PING-PONG: waves 0-3 (group A) and their SIMD partners 4-7 (group B) execute the same stream, B one barrier interval
//     ("slot") behind A, so in every slot each SIMD has one wave in M (matrix pipe) and one in L (LDS pipe + VALU):
//         slot   2i    2i+1   2i+2   2i+3
//         A      L_i   M_i    L_i+1  M_i+1
//         B      M_i-1 L_i    M_i    L_i+1
//     Everything that is not an operand read rides in the issue slots between the MFMAs of an M segment: the ring's LDS-DMA, the
//     transform + LDS write of the staging item whose loads were issued one M earlier (two slots of flight), the next item's loads;
//   * the phase end (in-plane inverse 4x4 -> 2x2 of the 16 accumulators of a depth frequency) is split over the two L segments of
//     the next phase's first step: columns xw 0,1 before the first M overwrites them, columns 2,3 + the depth sums before the second;
//     the partial depth inverses are two running sums in registers.
// The arithmetic and its order are exactly wino3d.hip's: results are BIT-IDENTICAL to drc_conv3d_k3_wino_fwd.
// LDS: ring 2 x 16 KB + brick 2 x (NS + 1) slots x (256 * TW) B; NS = 16 for 28x28 maps (151 KB), 28 for 14x14 maps (134 KB).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define RB_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define RB_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define RB_WAVES 8
#define RB_RING_BYTES 32768

#ifdef RB_TRACE
// development only: s_memtime stamps of block 8, waves 0 and 4 (the two waves of one SIMD), 16 marks per step, 64 steps.  The stamps of a
// step are collected in one VGPR (lane k = mark k) and stored once per step, so that no trace store sits in front of the kernel's own waits.
__device__ unsigned long long rb_trace_buf[2][64][16];
#define RB_MARK(k) do { if (rb_traced) { const unsigned t_ = __builtin_amdgcn_readfirstlane((unsigned)__builtin_readcyclecounter()); \
                             asm volatile("v_writelane_b32 %0, %1, " #k : "+v"(rb_marks) : "s"(t_)); } } while (0)
#define RB_FLUSH_MARKS() do { if (rb_traced && stepno > 8 && stepno <= 72 && lane < 16) rb_trace_buf[wave >> 2][stepno - 9][lane] = rb_marks; } while (0)
extern "C" int drc_rb_trace_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rb_trace_buf), sizeof(rb_trace_buf));
}
#else
#define RB_MARK(k)
#define RB_FLUSH_MARKS()
#endif

namespace {

// CV = the cost volume folded into the staging loads (dres0[0], stackhourglass.py:115-130; the round-2 wino3d_cv_kernel's job): instead of a
// materialised volume the items read the blocked 2D feature maps -- channel blocks < cbi from the left map at (y, x), the others from the
// right map at (y, x - i), i = lo4 + slice -- and a load whose voxel is outside the volume or fails 0 <= x - i < W' is pointed at halo
// column 0 of its row (zero).
// D2 = the 2D form (Conv2d 3x3, stride 1, pad 1 as Winograd F(2x2, 3x3); reference submodule.py:13-17 `convbn`, backbone/resnet.py): one
// "slice" (no depth butterfly), 16 frequency points, a tile's accumulators go straight from the in-plane inverse to the epilogue.
// Bit-identical to wino2d.hip.
template <int TW, bool CV, bool D2>
__device__ __forceinline__ void wino3d_rb_body(const drc_tapconv_params& p, const drc_costvol_src& cv, int NS) {
    static_assert(!(CV && D2), "the fused cost volume is a 3D input");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SB = 256 * TW;                   // bytes per row slot: [xw 4][g 4][wt TW] float4
    constexpr int XWS = 64 * TW;                   // bytes per w-frequency plane of a slot
    constexpr int GS = 16 * TW;                    // bytes per channel quad
    char* const ring = smem;                       // [slab 2][unit = i * 2 + ct_local : 16][g 4][j 16] float4
    char* const brick = smem + RB_RING_BYTES;      // [buffer 2][NS][SB]
    const unsigned buf_bytes = (unsigned)(NS + 1) * SB;     // NS row slots + one scratch slot (item_of)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const int g = lane >> 4;
    const int tg = wave & 3;                       // tile group of the chunk
    const int ctl = wave >> 2;                     // cout tile within the block's 32 couts (waves w and w+4 share a SIMD)
#ifdef RB_TRACE
    const bool rb_traced = blockIdx.x == 8 && (wave & 3) == 0;
    unsigned rb_marks = 0;
#endif

    const drc_tap_class cls = p.cls[0];
    // Maps wider than 2 TW columns (Config B's 56-wide volume at TW = 14) are walked as WS side-by-side strips of TW tile columns: a
    // "slab" is the (n, depth tile, strip) plane of TH tile rows, tiles run (n, dt, strip, ht, wt) -- everything below sees TW-wide maps
    // whose columns start at strip * 2 TW (the strips share their boundary columns like tiles do).
    const int TD = D2 ? 1 : p.OD >> 1, TH = p.OH >> 1, WS = (p.OW >> 1) / TW;
    const int tiles = p.N * TD * WS * TH * TW;
    const int rows_total = p.N * TD * WS * TH;
    const int chunks = (tiles + 63) >> 6;
    const int n_cg = p.cout_pad / 32;
    int cg, pos;
    const int nbk = (int)gridDim.x / n_cg;          // blocks per cout group (positions are numbered XCD by XCD, see wino3d.hip)
    if (gridDim.x % (8 * n_cg) == 0) {
        const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
        cg = l % n_cg;
        pos = xcd * (nbk / 8) + l / n_cg;
    } else {
        cg = blockIdx.x % n_cg;
        pos = blockIdx.x / n_cg;
    }
    const int rounds = (chunks + nbk - 1) / nbk;
    const int n_ct = p.cout_pad / 16;
    const int ct0 = cg * 2;

    auto slice_a = [](int xd) { return xd == 0 ? 0 : (xd == 2 ? 2 : 1); };      // depth butterfly of frequency xd: slice a + sgn * slice b
    auto slice_b = [](int xd) { return xd == 2 ? 1 : (xd == 3 ? 3 : 2); };      // (d0-d2, d1+d2, d2-d1, d1-d3)

    // ---- geometry of a round.  Lane: its tile, the LDS offset of its rows inside a brick buffer.  Thread: its (at most two)
    // staging items (row slot, tile column, channel quad): global byte offset of the row's first column, LDS byte offset.
    struct Geo { int tile; bool valid; unsigned lds; };        // the lane's tile of the round (clamped), the LDS offset of its first row
    auto geo_of = [&](int round) __attribute__((always_inline)) {
        Geo q;
        const int chunk = round * nbk + pos;
        const int t0 = chunk * 64 < tiles ? chunk * 64 : tiles - 1;
        int tile = chunk * 64 + tg * 16 + j;
        q.valid = tile < tiles;
        if (tile >= tiles) tile = tiles - 1;
        q.tile = tile;
        const int R0 = t0 / TW, R = tile / TW;
        int slot = 2 * (R - R0) + 2 * (R / TH - R0 / TH);
        slot = slot > NS - 4 ? NS - 4 : slot;
        q.lds = (unsigned)(slot * SB + g * GS + (tile - R * TW) * 16);
        return q;
    };
    // tile -> (n, depth tile, tile row, tile column in the whole map); only the epilogue needs them (once per round)
    auto tile_coords = [&](int tile, int& n, int& dt, int& ht, int& wt) __attribute__((always_inline)) {
        const int R = tile / TW;
        wt = tile - R * TW;
        int t = R;
        ht = t % TH; t /= TH;
        wt += (t % WS) * TW; t /= WS;
        dt = t % TD;
        n = t / TD;
    };
    // A staging item = (row slot, tile column wt, channel quad gq), thread t takes items t and t + 512: it loads the row's columns
    // 2wt..2wt+3 in the two slices of the step (8 float4; lanes of a quad group share a 64-byte line), applies the depth butterfly and the
    // w butterfly and writes the four w-frequencies of tile column wt to LDS.
    struct Item { unsigned goff, loff; bool valid; int x0, d0; };   // x0, d0 (CV): volume column / slice of the patch origin
    auto item_of = [&](int round, int k) __attribute__((always_inline)) {
        Item it;
        const int chunk = round * nbk + pos;
        const int t0 = chunk * 64 < tiles ? chunk * 64 : tiles - 1;
        const int t1 = chunk * 64 + 63 < tiles ? chunk * 64 + 63 : tiles - 1;
        const int R0 = t0 / TW, R1 = t1 / TW;
        int nslots = 2 * (R1 - R0) + 2 * (R1 / TH - R0 / TH) + 4;
        nslots = nslots > NS ? NS : nslots;
        const int q = (int)threadIdx.x + 512 * k;
        it.valid = q < nslots * TW * 4;
        const int s0 = q / (4 * TW);
        const int rem = q - s0 * (4 * TW);
        const int wt = rem >> 2, gq = rem & 3;
        // slot -> (tile row relative to R0, patch row h): two slots per tile row plus two per slab touched
        int s = s0 < nslots ? s0 : nslots - 1;
        int left = TH - R0 % TH, base = 0, rel = 0, h = 0;
        for (;;) {
            const int span = 2 * left + 2;
            if (s < span) {
                int rr = s >> 1;
                rr = rr > left - 1 ? left - 1 : rr;
                rel = base + rr;
                h = s - 2 * rr;
                break;
            }
            s -= span; base += left; left = TH;
        }
        int R = R0 + rel;
        R = R > rows_total - 1 ? rows_total - 1 : R;
        int t = R;
        const int ht = t % TH; t /= TH;
        const int wg = (t % WS) * TW + wt; t /= WS;     // the item's tile column in the whole map
        const int dt = t % TD;
        const int n = t / TD;
        if constexpr (CV) {         // byte offset of halo column 0 of feature-map row y = 2ht + h - 1 (+ the channel quad)
            it.goff = (unsigned)((n * cv.n_stride + (int64_t)(2 * ht + h - 1 + cv.pad) * cv.h_stride + gq * 4) * 4);
            it.x0 = 2 * wg - 1; it.d0 = 2 * dt - 1;
        } else {
            it.goff = (unsigned)((n * p.x_n_stride + (int64_t)(2 * dt + cls.dd0) * p.x_d_stride + (int64_t)(2 * ht + h + cls.dh0) * p.x_h_stride +
                                  (int64_t)(2 * wg + cls.dw0) * 16 + gq * 4) * 4);
            it.x0 = it.d0 = 0;
        }
        // a lane without an item transforms a (clamped, in-bounds) row like the others and writes it to the scratch slot NS of the
        // brick buffer: no exec-mask branch inside the M segments, which stay one basic block each
        it.loff = (unsigned)((it.valid ? s0 : NS) * SB + gq * GS + wt * 16);
        return it;
    };
    struct Raw { f32x4 a[4], b[4]; };
    // (all uniform offsets are 32-bit: the launcher checks N * x_n_stride * 4 < 2^32 and the packed weights < 2^31 floats)
    const unsigned xcb4 = (unsigned)p.x_cb_stride * 4u, xd4 = (unsigned)p.x_d_stride * 4u;
    // one load of an item: column w of slice a (ab = 0) or slice b (ab = 1).  Issued by EVERY lane -- the offsets of a lane without an item
    // are those of a clamped slot, in bounds -- so that an M segment always ends with exactly eight vector-memory instructions (RB_WAIT_M).
    auto stage_issue_1 = [&](const Item& it, int xd, int cb, Raw& r, int w, int ab) __attribute__((always_inline)) {
#if defined(RB_ABL_NOSTAGE) || defined(RB_ABL_NOISS)
        return;
#endif
        if constexpr (CV) {
            const bool right = cb >= cv.cbi;                                      // wave-uniform
            const char* base = (const char*)((right ? cv.right : cv.left) + (int64_t)(right ? cb - cv.cbi : cb) * cv.cb_stride);
            const int d = it.d0 + (ab == 0 ? slice_a(xd) : slice_b(xd));
            const bool ind = (unsigned)d < (unsigned)p.OD;
            const int sh = right ? cv.lo4 + d : 0;                              // the right map is read at x - i
            const int x = it.x0 + w;
            const bool ok = ind && (unsigned)x < (unsigned)cv.Wp && (unsigned)(x - cv.lo4 - d) < (unsigned)cv.Wp;
            const unsigned off = it.goff + (ok ? (unsigned)((x - sh + cv.pad) * 64) : 0u);
            (ab == 0 ? r.a[w] : r.b[w]) = *(const f32x4*)(base + off);
        } else {
            const char* sl = (const char*)p.x + ((unsigned)cb * xcb4 + (unsigned)(ab == 0 || D2 ? slice_a(xd) : slice_b(xd)) * xd4);
            (ab == 0 ? r.a[w] : r.b[w]) = *(const f32x4*)(sl + w * 64 + it.goff);      // (2D: one slice; its second load keeps the count at eight)
        }
    };
    auto stage_issue_w = [&](const Item& it, int xd, int cb, Raw& r, int w) __attribute__((always_inline)) {
        stage_issue_1(it, xd, cb, r, w, 0);
        stage_issue_1(it, xd, cb, r, w, 1);
    };
    // depth butterfly, w butterfly, four w-frequencies into brick buffer `bb` -- in nine pieces (pc 0..8) so that an M segment can spread
    // them over the gaps between its MFMAs (dd = the depth-butterflied columns, fdst = the item's LDS address; both live across the pieces)
    f32x4 dd[4];
    char* fdst = nullptr;
    auto stage_finish_pc = [&](int pc, const Item& it, int xd, char* bb, const Raw& r) __attribute__((always_inline)) {
#if defined(RB_ABL_NOSTAGE) || defined(RB_ABL_NOFIN)
        return;
#endif
        if (pc == 0) {
            fdst = bb + it.loff;
        } else if (pc <= 4) {
            const int w = pc - 1;
            if constexpr (D2) {
                dd[w] = r.a[w];
            } else {
                const float sgn = xd == 1 ? 1.f : -1.f;
                dd[w].x = __builtin_fmaf(sgn, r.b[w].x, r.a[w].x); dd[w].y = __builtin_fmaf(sgn, r.b[w].y, r.a[w].y);
                dd[w].z = __builtin_fmaf(sgn, r.b[w].z, r.a[w].z); dd[w].w = __builtin_fmaf(sgn, r.b[w].w, r.a[w].w);
            }
        } else if (pc == 5) {
            *(f32x4*)(fdst + 0 * XWS) = dd[0] - dd[2];
        } else if (pc == 6) {
            *(f32x4*)(fdst + 1 * XWS) = dd[1] + dd[2];
        } else if (pc == 7) {
            *(f32x4*)(fdst + 2 * XWS) = dd[2] - dd[1];
        } else if (pc == 8) {
            *(f32x4*)(fdst + 3 * XWS) = dd[1] - dd[3];
        }
    };
    auto stage_finish = [&](const Item& it, int xd, char* bb, const Raw& r) __attribute__((always_inline)) {
#pragma unroll
        for (int pc = 0; pc < 9; ++pc) stage_finish_pc(pc, it, xd, bb, r);
    };

    // ---- weight ring: half step (xd, cb, hf) uses the eight frequency points (xh, xw = 2 hf + xwl) of block cb for the block's two cout
    // tiles = 16 units of 1 KiB [g][j] float4 in the rb packing (drc_pack_weights_wino_rb: [xi = (xd*4 + xh)*4 + xw][cb][cout tile]);
    // unit u = (xh * 2 + xwl) * 2 + cout tile; wave w copies units w and w + 8 by LDS-DMA (lane l -> byte l*16 of the unit): xh = w >> 2
    // (+ 2 for the second unit), xwl = (w >> 1) & 1, cout tile w & 1; the per-wave part of the source offset is constant
    const unsigned wxi4 = (unsigned)(p.cb_in * n_ct) * 1024u;         // bytes per frequency point
    const unsigned wcb4 = (unsigned)n_ct * 1024u;                     // bytes per channel block
    const char* const wlane = (const char*)(p.w + (int64_t)ct0 * 256 + lane * 4) + ((unsigned)((wave >> 2) * 4 + ((wave >> 1) & 1)) * wxi4 + (unsigned)(wave & 1) * 1024u);
    auto ring_fill = [&](int xd, int cb, int hf) __attribute__((always_inline)) {
#ifdef RB_ABL_NOFILL
        return;
#endif
        const char* src = wlane + ((unsigned)(xd * 16 + hf * 2) * wxi4 + (unsigned)cb * wcb4);
        char* dst = ring + hf * 16384 + wave * 1024;
        __builtin_amdgcn_global_load_lds(RB_GLOBAL_PTR(src), RB_LDS_PTR(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(RB_GLOBAL_PTR(src + 8 * wxi4), RB_LDS_PTR(dst + 8192), 16, 0, 0);
    };
    // in the pipeline the four waves of group A copy all 16 units of the NEXT half step from their L segment (four each: unit w + 4m =
    // (xh = m, xwl = w >> 1, cout tile w & 1)); an L segment has issue slots to spare, an M segment does not
    const char* const wlaneA = (const char*)(p.w + (int64_t)ct0 * 256 + lane * 4) + ((unsigned)((wave >> 1) & 1) * wxi4 + (unsigned)(wave & 1) * 1024u);
    auto ring_fill_A = [&](int xd, int cb, int hf) __attribute__((always_inline)) {
#ifdef RB_ABL_NOFILL
        return;
#endif
        const char* src = wlaneA + ((unsigned)(xd * 16 + hf * 2) * wxi4 + (unsigned)cb * wcb4);
        char* dst = ring + hf * 16384 + (wave & 3) * 1024;
#pragma unroll
        for (int m = 0; m < 4; ++m)
            __builtin_amdgcn_global_load_lds(RB_GLOBAL_PTR(src + (unsigned)(4 * m) * wxi4), RB_LDS_PTR(dst + m * 4096), 16, 0, 0);
    };
    auto ring_fill_1 = [&](int xd, int cb, int hf, int second) __attribute__((always_inline)) {      // one of the two copies
#ifdef RB_ABL_NOFILL
        return;
#endif
        const char* src = wlane + ((unsigned)(xd * 16 + hf * 2 + second * 8) * wxi4 + (unsigned)cb * wcb4);
        char* dst = ring + hf * 16384 + wave * 1024 + second * 8192;
        __builtin_amdgcn_global_load_lds(RB_GLOBAL_PTR(src), RB_LDS_PTR(dst), 16, 0, 0);
    };

    f32x4 acc[4][4];                               // [xh][xw]
    f32x4 o0[4], o1[4];                            // running depth sums of the in-plane inverses: [oh * 2 + ow]
    f32x4 q0[2], q1[2];                            // phase end, part 1: q0[oh] = hh[oh][0] + hh[oh][1], q1[oh] = hh[oh][1]
    f32x4 wf[4][2], t[4][2];                       // the operands of the coming M segment: weight fragments [xh][xwl], transformed rows [h][xwl]
#if defined(RB_ABL_EXTRA_M) || defined(RB_ABL_EXTRA_L)
    float rb_dummy = (float)lane;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float rb_d4[8] = {(float)lane, 1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f};
    f32x2 rb_p4[8];
    for (int i_ = 0; i_ < 8; ++i_) { rb_p4[i_].x = (float)(lane + i_); rb_p4[i_].y = 1.f; }
#ifndef RB_ABL_KIND
#define RB_ABL_KIND 0
#endif
#if RB_ABL_KIND == 0
#define RB_DUMMY_OP(e) asm volatile("v_add_f32 %0, %0, %0" : "+v"(rb_dummy))
#elif RB_ABL_KIND == 1
#define RB_DUMMY_OP(e) asm volatile("v_add_f32 %0, %0, %0" : "+v"(rb_d4[(e) & 7]))
#else
#define RB_DUMMY_OP(e) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(rb_p4[(e) & 7]))
#endif
#endif

    // L: the operand reads of half step HALF of the step whose brick rows start at tb (rs = the half step's ring slab + the wave's cout tile
    // + lane): the four transformed rows t[h] of the two w-frequencies and the eight weight fragments.  No VALU here: beside a partner that is
    // multiplying, every vector instruction of this wave costs the SIMD ~8 cycles, while the multiplying wave's own gaps take ~4 for free --
    // so the h butterfly runs just in time inside M, one instruction per MFMA operand.
    auto load_ops = [&](auto half_tag, const char* tb, const char* rs) __attribute__((always_inline)) {
        constexpr int HALF = decltype(half_tag)::value;
#ifdef RB_ABL_NOLDSR
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int xwl = 0; xwl < 2; ++xwl) {
                const f32x4 c_ = {(float)lane, (float)(lane + xwl), 1.f, (float)h};
                wf[h][xwl] = c_; t[h][xwl] = c_ * 3.f;
                asm volatile("" : "+v"(wf[h][xwl]), "+v"(t[h][xwl]));
            }
        if (false)
#endif
        {
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int xwl = 0; xwl < 2; ++xwl) t[h][xwl] = *(const f32x4*)(tb + h * SB + (2 * HALF + xwl) * XWS);
#pragma unroll
        for (int xh = 0; xh < 4; ++xh)
#pragma unroll
            for (int xwl = 0; xwl < 2; ++xwl) wf[xh][xwl] = *(const f32x4*)(rs + (xh * 2 + xwl) * 2048);
        }
#ifdef RB_ABL_EXTRA_L
#pragma unroll
        for (int e_ = 0; e_ < RB_ABL_EXTRA_L; ++e_) RB_DUMMY_OP(e_);
#endif
    };
#define RB_SB() __builtin_amdgcn_sched_barrier(0)
    // M: the 32 MFMAs of half step HALF in the order (k-step S, column xwl, h-frequency xh).  MFMA i's B operand is one component of the h
    // butterfly (t0 - t2, t1 + t2, t2 - t1, t1 - t3) of its column: computed three gaps ahead, one VALU instruction per MFMA.  fill(i) = the few
    // other instructions that ride in the gap behind MFMA i (an MFMA occupies the matrix pipe for 32 cycles; about four of the wave's own
    // instructions fit beside it, a longer filler delays the next MFMA).
    auto mfma_seg = [&](auto first_tag, auto half_tag, auto&& fill) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int HALF = decltype(half_tag)::value;
        float bq[32];                              // operand of MFMA i (registers are reused as soon as the MFMA has issued)
        auto bfly = [&](int i) __attribute__((always_inline)) {
            const int S = i >> 3, xwl = (i >> 2) & 1, xh = i & 3;
            const float t0 = t[0][xwl][S], t1 = t[1][xwl][S], t2 = t[2][xwl][S], t3 = t[3][xwl][S];
            bq[i] = xh == 0 ? t0 - t2 : xh == 1 ? t1 + t2 : xh == 2 ? t2 - t1 : t1 - t3;
        };
        bfly(0); bfly(1); bfly(2);
        RB_SB();
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int S = i >> 3, xwl = (i >> 2) & 1, xh = i & 3;
#ifdef RB_ABL_NOMFMA
            acc[xh][2 * HALF + xwl] += wf[xh][xwl] * bq[i];
#else
            const f32x4 z4_ = {0.f, 0.f, 0.f, 0.f};
            acc[xh][2 * HALF + xwl] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[xh][xwl][S], bq[i], FIRST && S == 0 ? z4_ : acc[xh][2 * HALF + xwl], 0, 0, 0);
#endif
            RB_SB();
            if (i + 3 < 32) bfly(i + 3);
            fill(i);
#ifdef RB_ABL_EXTRA_M
#pragma unroll
            for (int e_ = 0; e_ < RB_ABL_EXTRA_M; ++e_) RB_DUMMY_OP(e_);
#endif
            RB_SB();
        }
    };
    // end of L: the operands have arrived.  End of M: the brick writes of this segment are done (lgkmcnt) and every vector-memory
    // instruction but the youngest eight -- the item loads, which every M issues last -- has completed, i.e. the ring copies have landed.
#define RB_WAIT_L() do { RB_SB(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); RB_SB(); } while (0)
#if defined(RB_ABL_NOSTAGE) || defined(RB_ABL_NOISS)
#define RB_WAIT_M() do { RB_SB(); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); RB_SB(); } while (0)
#else
#define RB_WAIT_M() do { RB_SB(); asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); RB_SB(); } while (0)
#endif
#define RB_BAR() do { RB_SB(); asm volatile("s_barrier" ::: "memory"); RB_SB(); } while (0)
#ifndef RB_PRIO_M
#define RB_PRIO_M 0
#endif
#ifndef RB_PRIO_L
#define RB_PRIO_L 0
#endif
#define RB_STR2(x) #x
#define RB_STR(x) RB_STR2(x)
#ifndef RB_SKEW
#define RB_SKEW 0
#endif
#if RB_SKEW == 1
// the four M waves of a CU (one per SIMD) leave the barrier together and would hit the vector-memory and LDS-write paths in the same cycles
// all segment long: delay wave w by (w & 3) * 16 cycles
#define RB_ENTER_M() do { if (wave & 1) asm volatile("s_nop 15" ::: "memory"); if (wave & 2) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); RB_SB(); } while (0)
#define RB_ENTER_L()
#elif RB_SKEW == 2
#define RB_ENTER_M() do { if (wave & 1) asm volatile("s_nop 15" ::: "memory"); RB_SB(); } while (0)
#define RB_ENTER_L()
#elif RB_PRIO_M != RB_PRIO_L
#define RB_ENTER_M() asm volatile("s_setprio " RB_STR(RB_PRIO_M) ::: "memory")
#define RB_ENTER_L() asm volatile("s_setprio " RB_STR(RB_PRIO_L) ::: "memory")
#else
#define RB_ENTER_M()
#define RB_ENTER_L()
#endif

    // ---- phase end of a depth frequency: in-plane inverse (A^T . A, 4x4 -> 2x2), folded into the running depth sums (A^T columns
    // [1 1 1 0] for od 0, [0 1 -1 -1] for od 1); the last frequency runs the epilogue.  Part 1 (columns xw 0, 1) runs before the
    // next phase's first M overwrites them, part 2 (columns 2, 3) before its second M.
    auto pe_part1 = [&]() __attribute__((always_inline)) {
#ifdef RB_ABL_NOPE
        return;
#endif
        f32x4 h0[2], h1[2];
#pragma unroll
        for (int xw = 0; xw < 2; ++xw) {
            h0[xw] = acc[0][xw] + acc[1][xw] + acc[2][xw];
            h1[xw] = acc[1][xw] - acc[2][xw] - acc[3][xw];
        }
        q0[0] = h0[0] + h0[1]; q1[0] = h0[1];
        q0[1] = h1[0] + h1[1]; q1[1] = h1[1];
    };
    auto pe_part2 = [&](int xd_, const Geo& eg) __attribute__((always_inline)) {
#ifdef RB_ABL_NOPE
        {   // keep every accumulator alive (one add each), nothing else of the phase end
            f32x4 s_ = acc[0][0];
#pragma unroll
            for (int i_ = 1; i_ < 16; ++i_) s_ += acc[i_ >> 2][i_ & 3];
            o0[0] = xd_ == 0 ? s_ : o0[0] + s_;
            if (xd_ == 3 && eg.valid && lane == 0 && o0[0].x + o0[0].y + o0[0].z + o0[0].w == 1.2345e-30f) p.y[0] = 1.f;
        }
        return;
#endif
        // the last frequency: the residual's eight float4 are requested first, so that they travel under the inverse transform
        const bool last = (D2 || xd_ == 3) && eg.valid;
        const int ct = ct0 + ctl;
        f32x4 rv[8];
        int64_t yo = 0;
        if (last) {
            int en, edt, eht, ewt;
            tile_coords(eg.tile, en, edt, eht, ewt);
            yo = p.y_off0 + (int64_t)en * p.y_n_stride + (int64_t)(2 * edt) * p.y_d_stride + (int64_t)(2 * eht) * p.y_h_stride +
                 (int64_t)(2 * ewt) * 16 + g * 4 + (int64_t)ct * p.y_cb_stride;
            if (p.res) {
                const int64_t ro = p.r_off0 + (int64_t)en * p.r_n_stride + (int64_t)(2 * edt) * p.r_d_stride + (int64_t)(2 * eht) * p.r_h_stride +
                                   (int64_t)(2 * ewt) * 16 + g * 4 + (int64_t)ct * p.r_cb_stride;
#pragma unroll
                for (int i = 0; i < (D2 ? 4 : 8); ++i)
                    rv[i] = *(const f32x4*)(p.res + ro + (i >> 2) * p.r_d_stride + ((i >> 1) & 1) * p.r_h_stride + (i & 1) * 16);
            }
        }
        f32x4 inv[4];
        {
            f32x4 h0[2], h1[2];
#pragma unroll
            for (int xw = 0; xw < 2; ++xw) {
                h0[xw] = acc[0][2 + xw] + acc[1][2 + xw] + acc[2][2 + xw];
                h1[xw] = acc[1][2 + xw] - acc[2][2 + xw] - acc[3][2 + xw];
            }
            inv[0] = q0[0] + h0[0]; inv[1] = q1[0] - h0[0] - h0[1];
            inv[2] = q0[1] + h1[0]; inv[3] = q1[1] - h1[0] - h1[1];
        }
        if constexpr (D2) {
            if (!last) return;
            const f32x4 bn_sc = *(const f32x4*)(p.scale + ct * 16 + g * 4);
            const f32x4 bn_sh = *(const f32x4*)(p.shift + ct * 16 + g * 4);
#pragma unroll
            for (int oh = 0; oh < 2; ++oh)
#pragma unroll
                for (int ow = 0; ow < 2; ++ow) {
                    f32x4 v_ = inv[oh * 2 + ow] * bn_sc + bn_sh;
                    if (p.res) v_ += rv[oh * 2 + ow];
                    if (p.relu) { v_.x = fmaxf(v_.x, 0.f); v_.y = fmaxf(v_.y, 0.f); v_.z = fmaxf(v_.z, 0.f); v_.w = fmaxf(v_.w, 0.f); }
                    *(f32x4*)(p.y + yo + oh * p.y_h_stride + ow * 16) = v_;
                }
            return;
        }
        if (xd_ == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) o0[i] = inv[i];
            return;
        }
        if (xd_ == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { o0[i] += inv[i]; o1[i] = inv[i]; }
            return;
        }
        if (xd_ == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { o0[i] += inv[i]; o1[i] -= inv[i]; }
            return;
        }
        if (!last) return;
        const f32x4 bn_sc = *(const f32x4*)(p.scale + ct * 16 + g * 4);
        const f32x4 bn_sh = *(const f32x4*)(p.shift + ct * 16 + g * 4);
#pragma unroll
        for (int od = 0; od < 2; ++od)
#pragma unroll
            for (int oh = 0; oh < 2; ++oh)
#pragma unroll
                for (int ow = 0; ow < 2; ++ow) {
                    const int i = oh * 2 + ow;
                    f32x4 v_ = (od == 0 ? o0[i] : o1[i] - inv[i]) * bn_sc + bn_sh;
                    if (p.res) v_ += rv[od * 4 + i];
                    if (p.relu) { v_.x = fmaxf(v_.x, 0.f); v_.y = fmaxf(v_.y, 0.f); v_.z = fmaxf(v_.z, 0.f); v_.w = fmaxf(v_.w, 0.f); }
                    *(f32x4*)(p.y + yo + od * p.y_d_stride + oh * p.y_h_stride + ow * 16) = v_;
                }
    };

    // ---- the pipeline.  Half step i = 2 * step + hf; brick buffer = step & 1, ring slab = hf.  LDS hand-offs:
    //   * ring slab of half step h: A's four waves copy its 16 units at the start of L_h-1 (slot 2h-2) and wait for them before the barrier
    //     that ends M_h-1 (slot 2h-1); first read in slot 2h; the slab's previous readers (half step h-2) finished in slot 2h-3.
    //   * brick of step s+1 (buffer (s+1) & 1) must be written in slots 4s..4s+3: its first read is in slot 4s+4 (A's L_2s+2) and that
    //     buffer's previous readers (step s-1) finish in slot 4s-1.  A thread has two items per brick, held in two geometry sets P, Q:
    //         M(hf 0) of step s: finish Q -> brick s+1            | issue loads of P for brick s+1 (A) / s+2 (B)
    //         M(hf 1) of step s: finish P -> brick s+1 (A) / s+2 (B) | issue loads of Q for brick s+2
    //     A writes in slots 4s+1, 4s+3; B in 4s+2 (brick s+1) and 4s+4 (brick s+2: slots 4(s+1)..): B stages half a step ahead of A.
    //     An item's loads fly for two slots (one M to the next).
    struct Cursor { int round, xd, cb; };
    auto advance = [&](Cursor c) __attribute__((always_inline)) {
        if (++c.cb == p.cb_in) { c.cb = 0; if (D2 || ++c.xd == 4) { c.xd = 0; ++c.round; } }
        return c;
    };
    auto clampc = [&](Cursor c) __attribute__((always_inline)) {        // steps past the end: stage the last round's rows again (never read)
        if (c.round >= rounds) { c.round = rounds - 1; c.xd = 0; c.cb = 0; }
        return c;
    };
    const int grp = ctl;                            // 0: group A, 1: group B
    const int kP = 1 - grp, kQ = grp;               // which of the thread's two items the sets P and Q hold

    Cursor c0 = {0, 0, 0};
    Cursor c1 = advance(c0);
    Cursor c2 = advance(c1);
    Geo geo = geo_of(0), ego = geo;
    Raw r;
    Item itP, itQ;
    int rndP, rndQ;
    // the fillers of an M segment: gaps 0..8: the nine pieces of the item finished here (its loads are two slots old; hipcc waits for ALL
    // vector memory before the first piece, so nothing younger may be in flight yet); 10, 12: the two ring copies; 14, 16, .., 28: the eight
    // loads of the item issued here (always the segment's last eight vector-memory instructions)
    auto m_fill = [&](int i, int dxd, int dcb, int dhf, const Item& fi, int fxd, char* fbuf, const Item& ii, const Cursor& ic) __attribute__((always_inline)) {
#ifndef RB_SCHED
#define RB_SCHED 1
#endif
#if RB_SCHED == 0
        if (i <= 8) stage_finish_pc(i, fi, fxd, fbuf, r);
        else if (i >= 14 && i <= 28 && !(i & 1)) stage_issue_1(ii, ic.xd, ic.cb, r, (i - 14) >> 2, ((i - 14) >> 1) & 1);
#elif RB_SCHED == 1
        // LDS writes eight MFMAs apart (the four M waves of a CU write at the same moment; back to back they queue behind each other)
        if (i <= 4) stage_finish_pc(i, fi, fxd, fbuf, r);
        else if (i == 7) stage_finish_pc(5, fi, fxd, fbuf, r);
        else if (i == 15) stage_finish_pc(6, fi, fxd, fbuf, r);
        else if (i == 23) stage_finish_pc(7, fi, fxd, fbuf, r);
        else if (i == 31) stage_finish_pc(8, fi, fxd, fbuf, r);
        else if (i >= 9 && i <= 29 && !(i & 1) ) { }
        else if (i == 9 || i == 11 || i == 13 || i == 17 || i == 19 || i == 21 || i == 25 || i == 27) {
            const int m = i == 9 ? 0 : i == 11 ? 1 : i == 13 ? 2 : i == 17 ? 3 : i == 19 ? 4 : i == 21 ? 5 : i == 25 ? 6 : 7;
            stage_issue_1(ii, ic.xd, ic.cb, r, m >> 1, m & 1);
        }
#else
        // LDS writes in the tail
        if (i <= 4) stage_finish_pc(i, fi, fxd, fbuf, r);
        else if (i >= 8 && i <= 22 && !(i & 1)) stage_issue_1(ii, ic.xd, ic.cb, r, (i - 8) >> 2, ((i - 8) >> 1) & 1);
        else if (i >= 28) stage_finish_pc(i - 23, fi, fxd, fbuf, r);
#endif
    };
    // ---- prologue: brick of step 0 into buffer 0; weights of half step 0 (all waves) and B's share of half step 1; B's item P of
    // brick 1 (its M(hf 1) of "step -1"); the loads of item Q of brick 1
    ring_fill(0, 0, 0);
    {
        const Item i0 = item_of(0, 0), i1 = item_of(0, 1);
#pragma unroll
        for (int w = 0; w < 4; ++w) stage_issue_w(i0, 0, 0, r, w);
        stage_finish(i0, 0, brick, r);
#pragma unroll
        for (int w = 0; w < 4; ++w) stage_issue_w(i1, 0, 0, r, w);
        stage_finish(i1, 0, brick, r);
    }
    {
        const Cursor cc1 = clampc(c1);
        itP = item_of(grp ? cc1.round : 0, kP); rndP = grp ? cc1.round : 0;
        if (grp == 1) {
#pragma unroll
            for (int w = 0; w < 4; ++w) stage_issue_w(itP, cc1.xd, cc1.cb, r, w);
            stage_finish(itP, cc1.xd, brick + buf_bytes, r);
        }
        itQ = item_of(cc1.round, kQ); rndQ = cc1.round;
#pragma unroll
        for (int w = 0; w < 4; ++w) stage_issue_w(itQ, cc1.xd, cc1.cb, r, w);
    }
    RB_SB();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (grp == 1) asm volatile("s_barrier" ::: "memory");       // B starts one slot late
    RB_SB();

    const char* const rs_lane = ring + ctl * 1024 + lane * 16;
    int stepno = 0;
    auto do_step = [&](auto first_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const Cursor cc1 = clampc(c1), cc2 = clampc(c2);
        const Cursor cP = grp ? cc2 : cc1;          // the brick P is staged for
        char* const buf1 = brick + (unsigned)((stepno + 1) & 1) * buf_bytes;      // brick s+1
        char* const buf2 = brick + (unsigned)(stepno & 1) * buf_bytes;            // brick s+2 (= this step's buffer, free after L(hf 1))
        char* const bufP = grp ? buf2 : buf1;
        // ---- L of half 0
        RB_MARK(0);
        if constexpr (FIRST) {
            if (stepno > 0) {                      // the phase that ended with the previous step
                pe_part1();
                if (D2 || c0.xd == 0) { ego = geo; geo = geo_of(c0.round); }
            }
        }
        if (rndP != cP.round) { itP = item_of(cP.round, kP); rndP = cP.round; }       // P: last used in the previous M(hf 1), next in M(hf 0)
        if (grp == 0) ring_fill_A(c0.xd, c0.cb, 1);                                      // the ring slab of this step's second half
        const char* tb = brick + (unsigned)(stepno & 1) * buf_bytes + geo.lds;
        load_ops(std::integral_constant<int, 0>{}, tb, rs_lane);
        RB_WAIT_L();
        RB_MARK(1);
        RB_BAR();
        RB_ENTER_M();
        // ---- M of half 0: A copies its ring units of (this step, half 1), B of (next step, half 0)
        RB_MARK(2);
        const Cursor cd0 = grp ? cc1 : c0;          // ring copies (past the end: a repeat into a slab nobody reads any more)
        mfma_seg(first_tag, std::integral_constant<int, 0>{}, [&](int i) __attribute__((always_inline)) {
            m_fill(i, cd0.xd, cd0.cb, 1 - grp, itQ, cc1.xd, buf1, itP, cP);
            if (i == 3) RB_MARK(11);
            if (i == 9) RB_MARK(12);
            if (i == 13) RB_MARK(13);
            if (i == 21) RB_MARK(14);
            if (i == 29) RB_MARK(15);
        });
        RB_MARK(3);
        RB_WAIT_M();
        RB_MARK(4);
        RB_BAR();
        RB_ENTER_L();
        // ---- L of half 1
        RB_MARK(5);
        if constexpr (FIRST) {
            if (stepno > 0) pe_part2(D2 ? 0 : (c0.xd + 3) & 3, ego);
        }
        if (rndQ != cc2.round) { itQ = item_of(cc2.round, kQ); rndQ = cc2.round; }    // Q: last used in M(hf 0), next in M(hf 1)
        if (grp == 0) ring_fill_A(cc1.xd, cc1.cb, 0);                                    // ... of the next step's first half
        load_ops(std::integral_constant<int, 1>{}, tb, rs_lane + 16384);
        RB_WAIT_L();
        RB_MARK(6);
        RB_BAR();
        RB_ENTER_M();
        // ---- M of half 1: A copies (next step, half 0), B (next step, half 1)
        RB_MARK(7);
        mfma_seg(first_tag, std::integral_constant<int, 1>{}, [&](int i) __attribute__((always_inline)) {
            m_fill(i, cc1.xd, cc1.cb, grp, itP, cP.xd, bufP, itQ, cc2);
        });
        c0 = c1; c1 = c2; c2 = advance(c2);
        ++stepno;
        RB_MARK(8);
        RB_WAIT_M();
        RB_MARK(9);
        RB_BAR();
        RB_ENTER_L();
        RB_MARK(10);
        RB_FLUSH_MARKS();
    };
#pragma unroll 1
    for (int ph = 0; ph < rounds * (D2 ? 1 : 4); ++ph) {
        do_step(std::true_type{});
#pragma unroll 1
        for (int c = 1; c < p.cb_in; ++c) do_step(std::false_type{});
    }
    pe_part1();                                    // the last phase of the last round
    pe_part2(D2 ? 0 : 3, geo);
    if (grp == 0) asm volatile("s_barrier" ::: "memory");       // A's count of barriers = B's
#if defined(RB_ABL_EXTRA_M) || defined(RB_ABL_EXTRA_L)
    { float t_ = rb_dummy; for (int i_ = 0; i_ < 8; ++i_) t_ += rb_d4[i_] + rb_p4[i_].x + rb_p4[i_].y; if (t_ == 1.2345e-30f) p.y[0] = t_; }
#endif
#undef RB_SB
}

template <int TW>
__global__ __launch_bounds__(64 * RB_WAVES, 2) void wino3d_rb_kernel(const drc_tapconv_params p, int NS) {
    wino3d_rb_body<TW, false, false>(p, drc_costvol_src{}, NS);
}

template <int TW>
__global__ __launch_bounds__(64 * RB_WAVES, 2) void wino3d_rb_cv_kernel(const drc_tapconv_params p, const drc_costvol_src cv, int NS) {
    wino3d_rb_body<TW, true, false>(p, cv, NS);
}

template <int TW>
__global__ __launch_bounds__(64 * RB_WAVES, 2) void wino2d_rb_kernel(const drc_tapconv_params p, int NS) {
    wino3d_rb_body<TW, false, true>(p, drc_costvol_src{}, NS);
}

// slots a 64-tile chunk can touch: two per tile row plus two per slab
inline int rb_slots(int TW, int TH) {
    const int rows_max = 63 / TW + 2;
    int slabs_max = (rows_max - 2) / TH + 2;
    if (slabs_max > rows_max) slabs_max = rows_max;
    return 2 * rows_max + 2 * slabs_max;
}

template <int TW>
int launch_rb(const drc_tapconv_params& p, const drc_costvol_src* cv, hipStream_t stream, bool d2 = false) {
    const int NS = rb_slots(TW, p.OH / 2);
    const size_t lds = RB_RING_BYTES + (size_t)2 * (NS + 1) * 256 * TW;
    if (lds > 163840 || NS * TW * 4 > 1024) return -4;
    static bool attr_set = false;                  // idempotent: racing first calls set the same value
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)wino3d_rb_kernel<TW>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wino3d_rb_cv_kernel<TW>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wino2d_rb_kernel<TW>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const long tiles = (long)p.N * (d2 ? 1 : p.OD / 2) * (p.OH / 2) * (p.OW / 2);
    const long chunks = (tiles + 63) / 64;
    const int n_cg = p.cout_pad / 32;
    long per_cg = 256 / n_cg;                      // one block (8 waves, up to 160 KB of LDS) per CU
    if (per_cg > chunks) per_cg = chunks;
    if (per_cg < 1) per_cg = 1;
    if (d2)
        hipLaunchKernelGGL((wino2d_rb_kernel<TW>), dim3((unsigned)(per_cg * n_cg)), dim3(64 * RB_WAVES), lds, stream, p, NS);
    else if (cv)
        hipLaunchKernelGGL((wino3d_rb_cv_kernel<TW>), dim3((unsigned)(per_cg * n_cg)), dim3(64 * RB_WAVES), lds, stream, p, *cv, NS);
    else
        hipLaunchKernelGGL((wino3d_rb_kernel<TW>), dim3((unsigned)(per_cg * n_cg)), dim3(64 * RB_WAVES), lds, stream, p, NS);
    return (int)hipGetLastError();
}

// U = (G x G x G) g per (cout, cin) pair in the rb packing: [xi = (xd*4 + xh)*4 + xw][cb][cout tile][g = ch / 4][j = cout % 16][ch % 4]
__global__ __launch_bounds__(256) void wino_weights_rb_kernel(const float* __restrict__ w, int cout, int cin, int transposed, int flip,
                                                              float* __restrict__ out) {
    const int cb_n = (cin + 15) / 16, cout_pad = (cout + 31) / 32 * 32;
    const long pairs = (long)cb_n * cout_pad * 16;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < pairs; idx += (long)gridDim.x * 256) {
        long t = idx;
        const int s = (int)(t & 3); t >>= 2;
        const int jj = (int)(t & 15); t >>= 4;
        const int gq = (int)(t & 3); t >>= 2;
        const int ctile = (int)(t % (cout_pad / 16));
        const int cb = (int)(t / (cout_pad / 16));
        const int co = ctile * 16 + jj, ci = cb * 16 + gq * 4 + s;
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

// the 2D weights: U = (G x G) g in the rb packing with 16 frequency points [xh*4 + xw][cb][cout tile][ch / 4][cout % 16][ch % 4]
__global__ __launch_bounds__(256) void wino2d_weights_rb_kernel(const float* __restrict__ w, int cout, int cin, int transposed, int flip,
                                                                float* __restrict__ out) {
    const int cb_n = (cin + 15) / 16, cout_pad = (cout + 31) / 32 * 32;
    const long pairs = (long)cb_n * cout_pad * 16;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < pairs; idx += (long)gridDim.x * 256) {
        long t = idx;
        const int s = (int)(t & 3); t >>= 2;
        const int jj = (int)(t & 15); t >>= 4;
        const int gq = (int)(t & 3); t >>= 2;
        const int ctile = (int)(t % (cout_pad / 16));
        const int cb = (int)(t / (cout_pad / 16));
        const int co = ctile * 16 + jj, ci = cb * 16 + gq * 4 + s;
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

extern "C" int drc_conv3d_k3_wino_rb_supported(int cout_pad, int OD, int OH, int OW) {
    if (cout_pad <= 0 || (cout_pad & 31) || OD <= 0 || OH <= 0 || OW <= 0 || ((OD | OH | OW) & 1)) return 0;
    if (OW != 14 && OW % 28) return 0;                      // 14-wide maps: TW = 7; multiples of 28: strips of TW = 14 tile columns
    const int TW = OW == 14 ? 7 : 14;
    const int NS = rb_slots(TW, OH / 2);
    return RB_RING_BYTES + (size_t)2 * (NS + 1) * 256 * TW <= 163840 && NS * TW * 4 <= 1024;
}

static int rb_fwd(const drc_tapconv_params* pp, const drc_costvol_src* cv, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if ((!cv && !p.x) || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD <= 0 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (p.cout_pad <= 0 || (p.cout_pad & 31) || p.cb_in <= 0) return -2;
    const drc_tap_class& k = p.cls[0];
    if (p.n_classes != 1 || p.in_mul != 1 || p.out_mul != 1 || k.nd != 3 || k.nh != 3 || k.nw != 3 || k.sd != 1 || k.sh != 1 || k.sw != 1)
        return -4;
    if (!drc_conv3d_k3_wino_rb_supported(p.cout_pad, p.OD, p.OH, p.OW)) return -4;
    if (cv) {
        if (!cv->left || !cv->right) return -1;
        if (cv->pad < 1 || cv->cbi <= 0 || p.cb_in != 2 * cv->cbi || cv->Wp != p.OW) return -2;
        if ((int64_t)p.N * cv->n_stride * 4 >= (1LL << 32)) return -5;
    } else if ((int64_t)p.N * p.x_n_stride * 4 >= (1LL << 32)) {
        return -5;                                                              // 32-bit item offsets over the whole batch
    }
    if ((int64_t)p.N * p.OD * p.OH * p.OW / 8 >= (1LL << 31) - 64 || (int64_t)64 * p.cb_in * p.cout_pad * 16 >= (1LL << 31)) return -5;
    hipStream_t s = (hipStream_t)stream;
    return p.OW == 14 ? launch_rb<7>(p, cv, s) : launch_rb<14>(p, cv, s);
}

extern "C" int drc_conv3d_k3_wino_rb_fwd(const drc_tapconv_params* pp, void* stream) { return rb_fwd(pp, nullptr, stream); }

extern "C" int drc_conv3d_k3_wino_rb_costvol_fwd(const drc_tapconv_params* pp, const drc_costvol_src* cv, void* stream) {
    if (!cv) return -1;
    return rb_fwd(pp, cv, stream);
}

extern "C" int drc_conv2d_k3_wino_rb_supported(int cout_pad, int OH, int OW) {
    return drc_conv3d_k3_wino_rb_supported(cout_pad, 2, OH, OW);
}

extern "C" int drc_conv2d_k3_wino_rb_fwd(const drc_tapconv_params* pp, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD != 1 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (p.cout_pad <= 0 || (p.cout_pad & 31) || p.cb_in <= 0) return -2;
    const drc_tap_class& k = p.cls[0];
    if (p.n_classes != 1 || p.in_mul != 1 || p.out_mul != 1 || k.nd != 1 || k.nh != 3 || k.nw != 3 || k.sh != 1 || k.sw != 1) return -4;
    if (!drc_conv2d_k3_wino_rb_supported(p.cout_pad, p.OH, p.OW)) return -4;
    if ((int64_t)p.N * p.x_n_stride * 4 >= (1LL << 32)) return -5;
    if ((int64_t)p.N * p.OH * p.OW / 4 >= (1LL << 31) - 64 || (int64_t)16 * p.cb_in * p.cout_pad * 16 >= (1LL << 31)) return -5;
    hipStream_t s = (hipStream_t)stream;
    return p.OW == 14 ? launch_rb<7>(p, nullptr, s, true) : launch_rb<14>(p, nullptr, s, true);
}

extern "C" int drc_pack_weights_wino2d_rb(const float* w, int cout, int cin, int transposed, int flip, float* out, void* stream) {
    if (cout <= 0 || cin <= 0) return -2;
    if (!w || !out) return -1;
    const long pairs = (long)((cin + 15) / 16) * ((cout + 31) / 32 * 32) * 16;
    long blocks = (pairs + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wino2d_weights_rb_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, cout, cin, transposed, flip, out);
    return (int)hipGetLastError();
}

extern "C" int drc_pack_weights_wino_rb(const float* w, int cout, int cin, int transposed, int flip, float* out, void* stream) {
    if (cout <= 0 || cin <= 0) return -2;
    if (!w || !out) return -1;
    const long pairs = (long)((cin + 15) / 16) * ((cout + 31) / 32 * 32) * 16;
    long blocks = (pairs + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wino_weights_rb_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, cout, cin, transposed, flip, out);
    return (int)hipGetLastError();
}
