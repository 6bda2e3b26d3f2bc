// conv16x.hip -- the fp16-storage STRIDE-2, TRANSPOSED and (two input blocks / 64 couts) STRIDE-1 3x3x3 convolutions with the input tile staged
// in LDS and ONE cout tile per wave; the fp16 cost volume folded into the first layer's stage addresses (gfx950 / CDNA4), round 4.
//
//   reference arithmetic: hourglass conv1 / conv3 (convbn_3d k3 s2 p1) and conv5 / conv6 (ConvTranspose3d k3 s2 p1 output_padding 1 + BN),
//   stackhourglass.py:11-30,35-49; fp16 storage, fp32 accumulation on v_mfma_f32_16x16x32_f16 as in conv16.hip / conv16t.hip (the reference
//   is fp32-only: this path is held to a stated bound against the fp32 oracle, tests/test_hip_f16.py).
//
// Why: these twelve launches of the stress shape (BASELINE configs[3]) ran on conv16.hip's generic tap walk, which reads both MFMA operands
// from global memory -- MFMA busy 6 %, 3.5 of the regressor's 8.6 ms (profiles/r4_stress16_*).  conv16t.hip's recipe (a block of four waves
// stages the input rows of its output tile once per channel block with LDS-DMA, every tap reads its B fragment from LDS, weights through
// the vector-memory path a few taps ahead) carries over with the changes below -- and with one it did not have: a wave owns ONE 16-channel cout
// tile and seven rows (CW cout tiles x RG row groups per block).  With every wave on all cout tiles and two rows, the layer's whole weight
// set went through L2 per two rows of output (2.65 GB per launch of conv6) and THAT bounded the kernels (see conv16u_kernel).
//
//   * stride 2 (conv16d_kernel): output column j of tap kw reads input column 2j + kw.  A staged input row is split into its even and odd
//     columns -- two 1-KiB planes, one global_load_lds each (lane (g, v) fetches channels 8g..8g+7 of column 2v + plane: still one 64-byte
//     line per four lanes) -- so that tap kw reads plane kw & 1 at entry j + (kw >> 1): unit stride over the lanes, no bank conflicts.
//     Rows: output row r of tap kh reads staged row 2r + kh (2 TR + 1 rows per depth tap); (channel block, depth tap) stages are
//     double-buffered.  The same kernel at stride 1 (ST = 1) serves the stride-1 layers conv16t.hip's depth-sliding walk does not take, and
//     with CV = true it reads the cost volume's rows straight from the feature pair (drc_conv16_k3_costvol_fwd).
//   * transposed (conv16u_kernel): the eight output-parity classes are stride-1 convolutions over the INPUT grid with 1, 2, 4 or 8 taps
//     (o = 2i - 1 + k: an even output has the single tap k = 1 at i, an odd one k = 2 at i and k = 0 at i + 1).  A block stages the
//     (TR + 1) x 16 input positions of its tile in the two depth slices i, i + 1 for ALL channel blocks once, then walks the classes:
//     accumulate the class's taps over the channel blocks, epilogue, store to the strided outputs -- 27 taps per tile in all, one stage.
//
// Layouts as in conv16t.hip: activations half[N][C/32][D+2][H+2][W+2][32] (zero halo), weights [tap][cb32][cout_pad][32] fp16
// (engine.pack_weight16), LDS row = [g 0..3][voxel 0..15][8 halfs].
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define X16_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define X16_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define X16_WAVES 4
#define X16_COLS 14

namespace {

typedef const __attribute__((address_space(3))) volatile f16x8 x16_lds_frag;       // (volatile: re-read per tap instead of kept live, as conv16t)

// BN scale / shift, residual, ReLU, fp16 store of one accumulator tile row (shared by both kernels)
__device__ __forceinline__ f16x4 x16_finish(const f32x4 acc, const f32x4 sc, const f32x4 sh, const f16x4 rv, int relu) {
    f32x4 v = acc * sc + sh;
    v.x += (float)rv.x; v.y += (float)rv.y; v.z += (float)rv.z; v.w += (float)rv.w;
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    f16x4 hv;
    hv.x = (_Float16)v.x; hv.y = (_Float16)v.y; hv.z = (_Float16)v.z; hv.w = (_Float16)v.w;
    return hv;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv16d_kernel<RW,CW,ST>: Conv3d k3 pad 1, stride ST = 2 or 1.  The four waves are CW cout tiles x RG = 4/CW row groups; a wave owns ONE
// 16-channel cout tile and RW output rows (see conv16u_kernel below for why: weights per output through L2).  Tile = TR = RW RG output
// rows x 14 output columns of one (n, od); lane (j, g) output column j (lanes j = 14, 15 compute two columns nobody stores, as in conv16t).
// A "stage" = the IR = ST TR + 3 - ST input rows of one (channel block, depth tap) -- at stride 2 the even and the odd columns as separate
// planes -- and stages are double-buffered: the DMAs and the nine weight fragments of stage s + 1 are issued before the MFMAs of stage s,
// one barrier per stage.  ST = 1 serves the stride-1 layers with two input blocks or 64 couts (dres0[0], hourglass conv2 / conv4), which
// conv16t.hip ran with every wave fetching all cout tiles' weights for two or four rows.
//
// CV (stride 1 only): the input is the fp16 cost volume of stackhourglass.py:115-128, never materialised.  p.x is the FEATURE PAIR
// half[N][2: left, right][3][H+2][W+2][32] (the cost-volume layout with one depth slice; drc_cost_volume16_blocked_fwd with D = 1 and
// disparity 0 writes it), OD = the volume's depth and `lo4` its first disparity.  Slice d of the volume is the left row where the shifted
// pixel exists and the right row moved by s = lo4 + d columns: both are the SAME feature rows for every d, read through per-lane
// addresses -- a lane whose voxel is zero in the volume (xw - s outside [0, W), the halo, a depth tap outside [0, D)) is pointed at
// column 0 of the row, the zero halo.  No extra instruction in the MFMA phase; the 617 MB volume of the stress shape (and the
// kernel that wrote it) are gone, the features stay L2-resident.
template <int RW, int CW, int ST, bool CV>
__global__ __launch_bounds__(64 * X16_WAVES) void conv16d_kernel(const drc_tapconv_params p, const int lo4) {
    constexpr int RG = X16_WAVES / CW;
    constexpr int TR = RW * RG;
    constexpr int IR = ST * TR + 3 - ST;               // staged input rows per depth tap
    constexpr int BUF = (ST * IR + 1) * 1024;          // [IR][plane ST][1 KiB] + one row of slack (entries past 15 of the last row)
    extern __shared__ __attribute__((aligned(16))) char lds[];        // two stage buffers
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cw = wave % CW, rg = wave / CW;
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    const _Float16* x = (const _Float16*)p.x;
    const drc_tap_class cls = p.cls[0];
    const int n_ct = (p.OW + X16_COLS - 1) / X16_COLS, n_rt = (p.OH + TR - 1) / TR;
    const int n_cg = p.cout_pad / 16 / CW;
    const unsigned tiles = (unsigned)p.N * p.OD * n_rt * n_ct * n_cg;   // cout group fastest: neighbouring blocks share the input tile in L2
    const int xh = (int)p.x_h_stride, xd = (int)p.x_d_stride, xc = (int)p.x_cb_stride;
    const int yh = (int)p.y_h_stride, yd_ = (int)p.y_d_stride, yc = (int)p.y_cb_stride;
    const int rh = (int)p.r_h_stride, rd_ = (int)p.r_d_stride, rc = (int)p.r_cb_stride;
    const int Hp = xd / xh, Wp = xh / 32;              // padded input extents (rows / columns): staging clamps to them on ragged tiles
    const long w_cb = (long)p.cout_pad * 32, w_tap = w_cb * p.cb_in;   // halfs
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, -1, 0x00020000);
    const int S = 3 * p.cb_in;                         // stages per tile, (cb, kd) in this order
    // this lane's B-fragment offset inside a stage buffer: staged row ST*(rg*RW), plane 0, entry j
    const unsigned lane_b = (unsigned)(g * 256 + j * 16 + rg * RW * ST * ST * 1024);

    for (unsigned tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        unsigned t = tile, u;
        u = t / (unsigned)n_cg; const int cg = (int)(t - u * (unsigned)n_cg); t = u;
        u = t / (unsigned)n_ct; const int c0 = (int)(t - u * (unsigned)n_ct) * X16_COLS; t = u;
        u = t / (unsigned)n_rt; const int r0 = (int)(t - u * (unsigned)n_rt) * TR; t = u;
        u = t / (unsigned)p.OD; const int od = (int)(t - u * (unsigned)p.OD);
        const int n = (int)u;
        // padded input coordinate of (output o, tap k) = ST o + first + k; columns of the plane(s), clamped into the row
        int colE = ST * (c0 + j) + cls.dw0, colO = colE + 1;
        colE = colE < Wp ? colE : Wp - 1; colO = colO < Wp ? colO : Wp - 1;
        const _Float16* xn = x + (long)n * p.x_n_stride + g * 8;
        const int cot = cg * CW + cw;
        const unsigned wlo = 2u * (unsigned)((cot * 16 + j) * 32 + g * 8);
        f16x8 wt[2][9];
        f32x4 acc[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // stage s -> buffer `par`: the DMAs (waves take rows round robin) and this wave's nine weight fragments (tap order kd, kh, kw)
        auto request = [&](int s, int par) __attribute__((always_inline)) {
            const int cb = s / 3, kd = s - cb * 3;
            const _Float16* src = xn + (cb * xc + (CV ? 1 : ST * od + cls.dd0 + kd) * xd);
            int colA = colE;
            if constexpr (CV) {
                // volume slice d = od + kd - 1 (cls.dd0 = 0: the padded depth index is od + kd), shift sft = lo4 + d; this lane's voxel
                // xw = colE - 1 is non-zero iff 0 <= d < OD, 0 <= xw < W and 0 <= xw - sft < W; the right block reads column xw - sft
                const int d = od + cls.dd0 + kd - 1, sft = lo4 + d, xw = colE - 1, xs = xw - sft, Wr = Wp - 2;
                const bool ok = d >= 0 && d < p.OD && xw >= 0 && xw < Wr && xs >= 0 && xs < Wr;
                colA = ok ? (cb ? xs + 1 : colE) : 0;
            }
            for (int i = wave; i < ST * IR; i += X16_WAVES) {
                int row = ST * r0 + cls.dh0 + (ST == 2 ? (i >> 1) : i);
                row = row < Hp ? row : Hp - 1;
                __builtin_amdgcn_global_load_lds(X16_GLOBAL_PTR(src + (row * xh + ((ST == 2 && (i & 1)) ? colO : colA) * 32)), X16_LDS_PTR(lds + par * BUF + i * 1024), 16, 0, 0);
            }
            const unsigned wo = 2u * (unsigned)((long)(kd * 9) * w_tap + cb * w_cb);
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9)
                wt[par][t9] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, wlo, wo + 2u * (unsigned)(t9 * (int)w_tap), 0));
        };
        auto half_step = [&](int s, auto PAR) __attribute__((always_inline)) {
            constexpr int par = decltype(PAR)::value;
            // stage s has landed (every wave's share: the barrier), and every wave is done reading the other buffer (stage s - 1)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (s + 1 < S) request(s + 1, par ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            const __attribute__((address_space(3))) char* bb = (const __attribute__((address_space(3))) char*)lds + par * BUF + lane_b;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                // the wave's staged rows 0 .. ST RW + 2 - ST, read once for the three row taps: output row r, tap kh reads row ST r + kh
                // (stride 2: plane kw & 1 at entry j + (kw >> 1); stride 1: entry j + kw)
                constexpr int NB = ST * RW + 3 - ST;
                f16x8 bf[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    bf[i] = *(x16_lds_frag*)(bb + (ST == 2 ? (i * 2 + (kw & 1)) * 1024 + (kw >> 1) * 16 : i * 1024 + kw * 16));
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int r = 0; r < RW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wt[par][kh * 3 + kw], bf[ST * r + kh], acc[r], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        request(0, 0);
        for (int s = 0; s < S; s += 2) {
            half_step(s, std::integral_constant<int, 0>{});
            if (s + 1 < S) half_step(s + 1, std::integral_constant<int, 1>{});
        }
        // ---- epilogue (fp32 BN / residual / ReLU, fp16 store)
        const int col = c0 + j;
        const bool col_ok = j < X16_COLS && col < p.OW;
        _Float16* y = (_Float16*)p.y + p.y_off0 + (long)n * p.y_n_stride + ((cot >> 1) * yc + (cot & 1) * 16 + g * 4);
        const _Float16* res = p.res ? (const _Float16*)p.res + p.r_off0 + (long)n * p.r_n_stride + ((cot >> 1) * rc + (cot & 1) * 16 + g * 4) : nullptr;
        const int row0 = r0 + rg * RW;
        const int yl = od * yd_ + row0 * yh + col * 32, rl = od * rd_ + row0 * rh + col * 32;
        f16x4 rv[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            rv[r] = (f16x4){0, 0, 0, 0};
            if (res && col_ok && row0 + r < p.OH) rv[r] = *(const f16x4*)(res + rl + r * rh);
        }
        const f32x4 sc = *(const f32x4*)(p.scale + cot * 16 + g * 4);
        const f32x4 sh = *(const f32x4*)(p.shift + cot * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const f16x4 hv = x16_finish(acc[r], sc, sh, rv[r], p.relu);
            if (col_ok && row0 + r < p.OH) *(f16x4*)(y + yl + r * yh) = hv;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");         // buffer 0 is free for the next tile's first stage
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv16u_kernel<RW,CW,CBN>: ConvTranspose3d k3 s2 p1 op1 as its eight output-parity classes (p.cls[0..7] as engine.taps_deconv3d_k3s2
// builds them: class (pd, ph, pw) has 1 or 2 taps per dimension, input offsets 0 / 0,1 from `first`, weight index wbase + a wsd + b wsh +
// c wsw, output offset (pd, ph, pw)).  Tile = TR x 14 positions of the INPUT grid of one (n, id); the block stages rows r0..r0+TR and
// entries c0..c0+15 of slices id, id+1 for every channel block once: [cb][slice 2][row TR+1][1 KiB].
//
// Who computes what (second form, round 4): the four waves are CW cout tiles x RG = 4/CW row groups; a wave owns ONE 16-channel cout tile
// and RW rows.  The first form gave every wave RW = 2 rows of ALL cout tiles, so each wave pulled the layer's whole weight set (108 KB for
// 64 -> 32 channels) through L2 per two rows: 2.65 GB of L2 -> CU traffic per launch of the stress shape, and ablations showed the kernel
// bound by exactly that (350 us; 232 us with the weights fetched once; the MFMA + LDS skeleton alone 95 us).  One cout tile x seven rows
// cuts the weight traffic per output 7x (3.5x per launch at CW = 2), and the RW + 1 B fragments of a (slice, column shift) serve both row
// taps, so LDS reads per MFMA do not grow.
template <int RW, int CW, int CBN>
__global__ __launch_bounds__(64 * X16_WAVES) void conv16u_kernel(const drc_tapconv_params p) {
    constexpr int RG = X16_WAVES / CW;
    constexpr int TR = RW * RG;
    constexpr int SR = TR + 1;                         // staged rows per slice
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cw = wave % CW, rg = wave / CW;
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    const _Float16* x = (const _Float16*)p.x;
    const int n_ct = (p.OW + X16_COLS - 1) / X16_COLS, n_rt = (p.OH + TR - 1) / TR;       // (OD, OH, OW = the INPUT grid)
    const int n_cg = p.cout_pad / 16 / CW;
    const unsigned tiles = (unsigned)p.N * p.OD * n_rt * n_ct * n_cg;
    const int xh = (int)p.x_h_stride, xd = (int)p.x_d_stride, xc = (int)p.x_cb_stride;
    const int yh = (int)p.y_h_stride, yd_ = (int)p.y_d_stride, yc = (int)p.y_cb_stride;
    const int rh = (int)p.r_h_stride, rd_ = (int)p.r_d_stride, rc = (int)p.r_cb_stride;
    const int Hp = xd / xh, Wp = xh / 32;
    const long w_cb = (long)p.cout_pad * 32, w_tap = w_cb * p.cb_in;   // halfs
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, -1, 0x00020000);
    const int first_d = p.cls[0].dd0, first_h = p.cls[0].dh0, first_w = p.cls[0].dw0;     // the same for every class (the input halo)
    const unsigned lane_b = (unsigned)(g * 256 + j * 16 + rg * RW * 1024);

    for (unsigned tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        unsigned t = tile, u;
        u = t / (unsigned)n_cg; const int cg = (int)(t - u * (unsigned)n_cg); t = u;
        u = t / (unsigned)n_ct; const int c0 = (int)(t - u * (unsigned)n_ct) * X16_COLS; t = u;
        u = t / (unsigned)n_rt; const int r0 = (int)(t - u * (unsigned)n_rt) * TR; t = u;
        u = t / (unsigned)p.OD; const int id = (int)(t - u * (unsigned)p.OD);
        const int n = (int)u;
        // ---- stage: every wave is done with the previous tile's buffer (the barrier that ended it); rows clamped on ragged tiles
        {
            int colv = c0 + first_w + j;
            colv = colv < Wp ? colv : Wp - 1;
            const _Float16* src = x + (long)n * p.x_n_stride + ((id + first_d) * xd + colv * 32 + g * 8);
            constexpr int rows = CBN * 2 * SR;
            for (int i = wave; i < rows; i += X16_WAVES) {
                const int cb = i / (2 * SR), rem = i - cb * (2 * SR), sl = rem / SR;
                int row = r0 + first_h + (rem - sl * SR);
                row = row < Hp ? row : Hp - 1;
                __builtin_amdgcn_global_load_lds(X16_GLOBAL_PTR(src + (cb * xc + sl * xd + row * xh)), X16_LDS_PTR(lds + i * 1024), 16, 0, 0);
            }
        }
        const int cot = cg * CW + cw;
        const unsigned wlo = 2u * (unsigned)((cot * 16 + j) * 32 + g * 8);
        // A "batch" = (class c, channel block cb): its nd*nh*nw weight fragments, requested ONE BATCH AHEAD (two register sets, static
        // rotation: the loops are fully unrolled), so that a class never starts with an L2 round trip.
        f16x8 wt[2][8];
        auto wfetch = [&](int set, int c, int cb) __attribute__((always_inline)) {
            const int nd = (c >> 2) ? 2 : 1, nh = ((c >> 1) & 1) ? 2 : 1, nw = (c & 1) ? 2 : 1;
            const drc_tap_class k = p.cls[c];
#pragma unroll
            for (int a = 0; a < nd; ++a)
#pragma unroll
                for (int b = 0; b < nh; ++b)
#pragma unroll
                    for (int e = 0; e < nw; ++e) {
                        const int widx = k.wbase + a * k.wsd + b * k.wsh + e * k.wsw;
                        const unsigned wo = 2u * (unsigned)((long)widx * w_tap + cb * w_cb);
                        wt[set][(a * 2 + b) * 2 + e] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, wlo, wo, 0));
                    }
        };
        wfetch(0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        const __attribute__((address_space(3))) char* bb = (const __attribute__((address_space(3))) char*)lds + lane_b;
        const int col = c0 + j;
        const bool col_ok = j < X16_COLS && col < p.OW;
        const int row0 = r0 + rg * RW;
        const f32x4 sc = *(const f32x4*)(p.scale + cot * 16 + g * 4);          // BN scale / shift of the wave's cout tile: once per tile
        const f32x4 sh = *(const f32x4*)(p.shift + cot * 16 + g * 4);
        _Float16* y = (_Float16*)p.y + p.y_off0 + (long)n * p.y_n_stride + ((cot >> 1) * yc + (cot & 1) * 16 + g * 4);
        const _Float16* res = p.res ? (const _Float16*)p.res + p.r_off0 + (long)n * p.r_n_stride + ((cot >> 1) * rc + (cot & 1) * 16 + g * 4) : nullptr;
        // Classes are walked in (pd, ph, pw) order; the two pw classes of a (pd, ph) pair are the even and the odd output columns of the same
        // rows -- the two 64-byte voxels of each 128-byte line -- and are stored together.
        f16x4 ov[RW];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int pd = c >> 2, ph = (c >> 1) & 1, pw = c & 1;      // class order of taps_deconv3d_k3s2: (pd, ph, pw) lexicographic
            const int nd = pd ? 2 : 1, nh = ph ? 2 : 1, nw = pw ? 2 : 1;
            const int yl = (2 * id + pd) * yd_ + (2 * row0 + ph) * yh + (2 * col + pw) * 32;
            const int rl = (2 * id + pd) * rd_ + (2 * row0 + ph) * rh + (2 * col + pw) * 32;
            f16x4 rv[RW];
            f32x4 acc[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                rv[r] = (f16x4){0, 0, 0, 0};
                if (res && col_ok && row0 + r < p.OH) rv[r] = *(const f16x4*)(res + rl + 2 * r * rh);
                acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int cb = 0; cb < CBN; ++cb) {
                const int batch = c * CBN + cb;
                if (batch + 1 < 8 * CBN) wfetch((batch + 1) & 1, (batch + 1) / CBN, (batch + 1) % CBN);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < nd; ++a)
#pragma unroll
                    for (int e = 0; e < nw; ++e) {
                        f16x8 bf[RW + 1];                              // rows r .. r + RW of (slice a, column shift e): both row taps read them
#pragma unroll
                        for (int i = 0; i < RW + nh - 1; ++i)
                            bf[i] = *(x16_lds_frag*)(bb + ((cb * 2 + a) * SR + i) * 1024 + e * 16);
#pragma unroll
                        for (int b = 0; b < nh; ++b)
#pragma unroll
                            for (int r = 0; r < RW; ++r)
                                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wt[batch & 1][(a * 2 + b) * 2 + e], bf[r + b], acc[r], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const f16x4 hv = x16_finish(acc[r], sc, sh, rv[r], p.relu);
                if (pw == 0) {
                    ov[r] = hv;
                } else if (col_ok && row0 + r < p.OH) {
                    *(f16x4*)(y + yl - 32 + 2 * r * yh) = ov[r];
                    *(f16x4*)(y + yl + 2 * r * yh) = hv;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");         // the buffer is free for the next tile's stage
    }
}

template <int RW, int CW, int ST, bool CV = false>
int launch_d_rw(const drc_tapconv_params& p, hipStream_t stream, int lo4 = 0) {
    constexpr int TR = RW * (X16_WAVES / CW);
    constexpr size_t lds = (size_t)2 * (ST * (ST * TR + 3 - ST) + 1) * 1024;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv16d_kernel<RW, CW, ST, CV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const long tiles = (long)p.N * p.OD * ((p.OH + TR - 1) / TR) * ((p.OW + X16_COLS - 1) / X16_COLS) * (p.cout_pad / 16 / CW);
    if (tiles >= (1L << 31)) return -5;
    long per_cu = (160 * 1024) / (long)lds;
    if (per_cu > 4) per_cu = 4;
    long blocks = 256 * per_cu;
    if (blocks > tiles) blocks = tiles;
    hipLaunchKernelGGL((conv16d_kernel<RW, CW, ST, CV>), dim3((unsigned)blocks), dim3(64 * X16_WAVES), lds, stream, p, lo4);
    return (int)hipGetLastError();
}

// Rows per wave (engine.x16_rows mirrors this rule for the plan's kernel name).  Stride 2: the block's two stage buffers stay below 80 KiB
// (two blocks per CU) with TR <= 8 -- seven rows for one row group, four for two, two for four; stride 1: seven.  The small tile (two
// rows, one with four row groups at stride 2) when the big one pads the map's rows by more than 25 % over the small one's.
template <int CW, int ST, bool CV = false>
int launch_d(const drc_tapconv_params& p, hipStream_t stream, int lo4 = 0) {
    constexpr int RG = X16_WAVES / CW;
    constexpr int BIG = ST == 1 ? 7 : (RG == 1 ? 7 : RG == 2 ? 4 : 2), SMALL = (ST == 2 && RG == 4) ? 1 : 2;
    const int trb = BIG * RG, trs = SMALL * RG;
    const long padb = (long)((p.OH + trb - 1) / trb) * trb, pads = (long)((p.OH + trs - 1) / trs) * trs;
    return padb * 4 <= pads * 5 ? launch_d_rw<BIG, CW, ST, CV>(p, stream, lo4) : launch_d_rw<SMALL, CW, ST, CV>(p, stream, lo4);
}

template <int RW, int CW, int CBN>
int launch_u_cb(const drc_tapconv_params& p, hipStream_t stream) {
    constexpr int TR = RW * (X16_WAVES / CW);
    constexpr size_t lds = (size_t)CBN * 2 * (TR + 1) * 1024 + 1024;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv16u_kernel<RW, CW, CBN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const long tiles = (long)p.N * p.OD * ((p.OH + TR - 1) / TR) * ((p.OW + X16_COLS - 1) / X16_COLS) * (p.cout_pad / 16 / CW);
    if (tiles >= (1L << 31)) return -5;
    long per_cu = (160 * 1024) / (long)lds;
    if (per_cu > 4) per_cu = 4;
    long blocks = 256 * per_cu;
    if (blocks > tiles) blocks = tiles;
    hipLaunchKernelGGL((conv16u_kernel<RW, CW, CBN>), dim3((unsigned)blocks), dim3(64 * X16_WAVES), lds, stream, p);
    return (int)hipGetLastError();
}

// rows per wave: seven when the block's TR = 7 RG rows tile the map well and the stage leaves room for two blocks per CU, else two
template <int CW>
int launch_u(const drc_tapconv_params& p, hipStream_t stream) {
    constexpr int RG = X16_WAVES / CW;
    const int tr7 = 7 * RG, tr2 = 2 * RG;
    const bool fits7 = (size_t)p.cb_in * 2 * (tr7 + 1) * 1024 + 1024 <= 80 * 1024;
    const long pad7 = (long)((p.OH + tr7 - 1) / tr7) * tr7, pad2 = (long)((p.OH + tr2 - 1) / tr2) * tr2;
    const bool seven = fits7 && pad7 * 4 <= pad2 * 5;                         // (at most 25 % more padded rows than the small tile)
    switch (p.cb_in) {                                 // channel blocks of 32: the regressor has 1 or 2; 3 and 4 cover the layer tests / wider nets
        case 1: return seven ? launch_u_cb<7, CW, 1>(p, stream) : launch_u_cb<2, CW, 1>(p, stream);
        case 2: return seven ? launch_u_cb<7, CW, 2>(p, stream) : launch_u_cb<2, CW, 2>(p, stream);
        case 3: return seven ? launch_u_cb<7, CW, 3>(p, stream) : launch_u_cb<2, CW, 3>(p, stream);
        case 4: return seven ? launch_u_cb<7, CW, 4>(p, stream) : launch_u_cb<2, CW, 4>(p, stream);
        default: return -4;
    }
}

bool x16_common_ok(const drc_tapconv_params& p) {
    return p.cout_pad > 0 && !(p.cout_pad & 15) && p.cb_in > 0 && p.reserved != 1 && p.x_h_stride > 0 && p.x_d_stride % p.x_h_stride == 0 &&
           p.x_h_stride % 32 == 0;
}

int x16_check_sizes(const drc_tapconv_params& p) {
    if ((int64_t)p.cb_in * p.cout_pad * 32 * 27 * 2 >= (1LL << 31)) return -5;          // 32-bit byte offsets inside the weights
    if (p.x_n_stride * 2 >= (1LL << 31) || p.y_n_stride * 2 >= (1LL << 31) || (p.res && p.r_n_stride * 2 >= (1LL << 31))) return -5;
    return 0;
}

}  // namespace

// stride-1 3x3x3 layers, called by drc_conv16_k3_tile_fwd (conv16t.hip) after its own argument checks
int drc_x16_conv3d_s1_launch(const drc_tapconv_params& p, hipStream_t s) {
    if (!x16_common_ok(p)) return -4;
    if (int e = x16_check_sizes(p)) return e;
    const int ct = p.cout_pad / 16;
    if (ct % 4 == 0) return launch_d<4, 1>(p, s);
    if (ct % 2 == 0) return launch_d<2, 1>(p, s);
    return launch_d<1, 1>(p, s);
}

int drc_t16_conv3d_walk2_try(const drc_tapconv_params& p, hipStream_t s, bool cv, int lo4);       // conv16t.hip

extern "C" int drc_conv16_k3_costvol_fwd(const drc_tapconv_params* pp, int mindisp4, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD <= 0 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    const drc_tap_class& k = p.cls[0];
    if (p.n_classes != 1 || p.in_mul != 1 || p.out_mul != 1 || p.cb_in != 2) return -4;                 // 32 left + 32 right channels
    if (k.nd != 3 || k.nh != 3 || k.nw != 3 || k.sd != 1 || k.sh != 1 || k.sw != 1 || k.dd0 || k.dh0 || k.dw0) return -4;   // pad 1 on a halo-1 layout
    if (k.out_off_d || k.out_off_h || k.out_off_w || k.wbase != 0 || k.wsw != 1 || k.wsh != 3 || k.wsd != 9) return -4;
    if (!x16_common_ok(p)) return -4;
    if (p.x_h_stride != (int64_t)(p.OW + 2) * 32 || p.x_d_stride != (int64_t)(p.OH + 2) * p.x_h_stride) return -4;   // the feature pair: halo 1
    if (int e = x16_check_sizes(p)) return e;
    hipStream_t s = (hipStream_t)stream;
    {   // full row tiles: the depth walk with the weights in registers (conv16t.hip's conv16sw_kernel)
        const int st = drc_t16_conv3d_walk2_try(p, s, true, mindisp4);
        if (st != 1) return st;
    }
    const int ct = p.cout_pad / 16;
    if (ct % 4 == 0) return launch_d<4, 1, true>(p, s, mindisp4);
    if (ct % 2 == 0) return launch_d<2, 1, true>(p, s, mindisp4);
    return launch_d<1, 1, true>(p, s, mindisp4);
}

extern "C" int drc_conv16_k3s2_tile_supported(const drc_tapconv_params* pp) {
    if (!pp) return 0;
    const drc_tapconv_params& p = *pp;
    const drc_tap_class& k = p.cls[0];
    if (p.n_classes != 1 || p.in_mul != 2 || p.out_mul != 1) return 0;
    if (k.nd != 3 || k.nh != 3 || k.nw != 3 || k.sd != 1 || k.sh != 1 || k.sw != 1) return 0;
    if (k.out_off_d || k.out_off_h || k.out_off_w) return 0;
    if (k.wbase != 0 || k.wsw != 1 || k.wsh != 3 || k.wsd != 9) return 0;                 // weights in tap order
    return p.cout_pad > 0 && !(p.cout_pad & 15) && p.cb_in > 0 && p.reserved != 1;
}

extern "C" int drc_conv16_k3s2_tile_fwd(const drc_tapconv_params* pp, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD <= 0 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (!drc_conv16_k3s2_tile_supported(pp) || !x16_common_ok(p)) return -4;
    if (int e = x16_check_sizes(p)) return e;
    hipStream_t s = (hipStream_t)stream;
    const int ct = p.cout_pad / 16;
    if (ct % 4 == 0) return launch_d<4, 2>(p, s);
    if (ct % 2 == 0) return launch_d<2, 2>(p, s);
    return launch_d<1, 2>(p, s);
}

extern "C" int drc_deconv16_k3s2_tile_supported(const drc_tapconv_params* pp) {
    if (!pp) return 0;
    const drc_tapconv_params& p = *pp;
    if (p.n_classes != 8 || p.in_mul != 1 || p.out_mul != 2) return 0;
    for (int c = 0; c < 8; ++c) {
        const drc_tap_class& k = p.cls[c];
        const int pd = c >> 2, ph = (c >> 1) & 1, pw = c & 1;
        if (k.nd != (pd ? 2 : 1) || k.nh != (ph ? 2 : 1) || k.nw != (pw ? 2 : 1) || k.sd != 1 || k.sh != 1 || k.sw != 1) return 0;
        if (k.out_off_d != pd || k.out_off_h != ph || k.out_off_w != pw) return 0;
        if (k.dd0 != p.cls[0].dd0 || k.dh0 != p.cls[0].dh0 || k.dw0 != p.cls[0].dw0) return 0;
        if (k.wbase < 0 || k.wbase > 26) return 0;
    }
    if (p.cb_in > 4) return 0;                          // (instantiated channel-block counts; wider layers keep conv16.hip)
    return p.cout_pad > 0 && !(p.cout_pad & 15) && p.cb_in > 0 && p.reserved != 1;
}

extern "C" int drc_deconv16_k3s2_tile_fwd(const drc_tapconv_params* pp, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD <= 0 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (!drc_deconv16_k3s2_tile_supported(pp) || !x16_common_ok(p)) return -4;
    if (int e = x16_check_sizes(p)) return e;
    hipStream_t s = (hipStream_t)stream;
    const int ct = p.cout_pad / 16;
    if (ct % 4 == 0) return launch_u<4>(p, s);
    if (ct % 2 == 0) return launch_u<2>(p, s);
    return launch_u<1>(p, s);
}
