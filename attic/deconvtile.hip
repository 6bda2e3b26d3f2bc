// deconvtile.hip -- fp32 ConvTranspose3d(k3, s2, p1, output_padding 1) (+BN, +residual, +ReLU) with the input tile staged in LDS and one
// cout tile per wave (gfx950 / CDNA4), round 4.  EXPERIMENT, NOT IN THE LIBRARY: parity-green against the oracle (6 shapes, ragged rows /
// columns, 32 / 64 input channels) but not faster than deconvdirect.hip on hourglass conv6 at 1,024 ROIs:
//     deconvdirect_kernel<2,2>                                   ~1,430 us
//     this kernel, residuals / stores class by class, <7,2,4>     1,455-1,466 us   (1 block per CU; <4,2,4>, 2 blocks per CU: 1,497)
//     ... residuals / stores in two groups per tile (this file)   1,592 us          (a wave cannot have more than 63 vm ops in flight)
//     ablations of the class-by-class form: no stores 1,326, no residual loads 1,362, no stage 1,396, weights fetched once 1,410,
//     all four (MFMA + LDS + VALU skeleton) 1,095; ideal MFMA time 970-1,000.
// To wire it in: add it to csrc/build.py, declare the two entries in include/disprcnn_hip.h and route engine.plan_deconv3d to
// drc_deconv3d_k3s2_tile_fwd (same parameter block and weight packing as drc_deconv3d_k3s2_direct_fwd).
//
//   reference: hourglass conv5 / conv6, stackhourglass.py:22-30,44-49 (fp32, the default path)
//
// Why a second fp32 transposed kernel: deconvdirect.hip keeps the eight shifted B fragments of two voxel tiles in registers (two sets)
// next to 8 x 2 x 2 accumulator tiles -- 312 live vector registers before any address -- and pays for it in VALU work: per 432 MFMAs its
// main loop issues 420 VALU instructions (set copies, AGPR shuttling), its item prologue 750 and its epilogue 950 more, 3,150 per 1,728
// MFMAs.  On gfx950 the fp32 MFMA executes on the SIMD's vector ALUs, so every VALU instruction costs ~6 cycles of MFMA time (DESIGN
// 3.0c): MFMA busy 58 %.  This kernel is the fp32 twin of conv16x.hip's conv16u_kernel:
//
//   * the eight output-parity classes are stride-1 convolutions over the INPUT grid with 1, 2, 4 or 8 taps (o = 2i - 1 + k);
//   * a block of four waves = CW cout tiles x RG = 4/CW row groups stages the (TR + 1) x 16 input positions of its tile in the two depth
//     slices id, id + 1 for ALL 16-channel blocks once ([cb][slice][row][1 KiB], LDS-DMA, row = [g][voxel 0..15][4 floats]: the B fragment
//     of column shift e is one conflict-free ds_read_b128 at g*256 + (j + e)*16);
//   * a wave owns ONE cout tile and RW rows: per class it accumulates the class's taps over the channel blocks -- the RW + 1 row fragments
//     of a (slice, column shift) serve both row taps -- then BN / residual / ReLU and the stores; the two column-parity classes of a row
//     are stored together (both 64-byte voxels of a 128-byte line);
//   * weights: deconvdirect.hip's packing [cb][27 combinations in use order][cout][16] (engine.pack_weight_deconv_direct), one
//     (class, channel block) batch ahead in a second register set.
//   Per tile and wave at RW = 7, 64 input channels: 3,024 MFMAs against ~110 weight loads, ~230 ds_read_b128 and ~450 VALU instructions.
//
// 15 of the 16 lanes of a column group carry an output column (shift 1 needs entry j + 1): the kernel is used where that beats
// deconvdirect.hip's flattened voxel tiles (W = 14: conv6; not the 7-wide conv5), engine.plan_deconv3d.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DT_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define DT_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define DT_WAVES 4
#define DT_COLS 15
#ifndef DT_ABL
#define DT_ABL 0
#endif
#define DT_ABL_ST ((DT_ABL & 1) ? p.relu == 77 : true) &&
#define DT_ABL_RES ((DT_ABL & 2) ? p.relu == 77 : true) &&
#define DT_ABL_DMA if ((DT_ABL & 4) ? p.relu == 77 : true)
#define DT_ABL_W if ((DT_ABL & 8) ? p.relu == 77 : true)

namespace {

typedef const __attribute__((address_space(3))) volatile f32x4 dt_lds_frag;

// position of (class c = (pd, ph, pw), input shifts (a, b, e)) in deconvdirect.hip's use order: classes 7 .. 0, inside a class the
// shifts in lexicographic order
constexpr int dt_popcount(int c) { return (c & 1) + ((c >> 1) & 1) + ((c >> 2) & 1); }
constexpr int dt_use_pos(int c, int a, int b, int e) {
    int off = 0;
    for (int k = 7; k > c; --k) off += 1 << dt_popcount(k);
    const int nh = ((c >> 1) & 1) ? 2 : 1, nw = (c & 1) ? 2 : 1;
    return off + (a * nh + b) * nw + e;
}

template <int N, int I = 0, class F>
__device__ __forceinline__ void dt_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        dt_static_for<N, I + 1>(f);
    }
}

template <int RW, int CW, int CBN>
__global__ __launch_bounds__(64 * DT_WAVES) void deconvtile_kernel(const drc_tapconv_params p) {
    constexpr int RG = DT_WAVES / CW;
    constexpr int TR = RW * RG;
    constexpr int SR = TR + 1;                         // staged rows per slice
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cw = wave % CW, rg = wave / CW;
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    const float* x = p.x;
    const int n_ct = (p.OW + DT_COLS - 1) / DT_COLS, n_rt = (p.OH + TR - 1) / TR;         // (OD, OH, OW = the INPUT grid)
    const int n_cg = p.cout_pad / 16 / CW;
    const unsigned tiles = (unsigned)p.N * p.OD * n_rt * n_ct * n_cg;
    const int xh = (int)p.x_h_stride, xd = (int)p.x_d_stride, xc = (int)p.x_cb_stride;
    const int yh = (int)p.y_h_stride, yd_ = (int)p.y_d_stride;
    const int rh = (int)p.r_h_stride, rd_ = (int)p.r_d_stride;
    const int Hp = xd / xh, Wp = xh / 16;
    const unsigned w_tap_b = (unsigned)p.cout_pad * 64;                                   // bytes per (cb, combination)
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, -1, 0x00020000);
    const int first_d = p.cls[0].dd0, first_h = p.cls[0].dh0, first_w = p.cls[0].dw0;     // the same for every class (the input halo)
    const unsigned lane_b = (unsigned)(g * 256 + j * 16 + rg * RW * 1024);

    for (unsigned tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        unsigned t = tile, u;
        u = t / (unsigned)n_cg; const int cg = (int)(t - u * (unsigned)n_cg); t = u;
        u = t / (unsigned)n_ct; const int c0 = (int)(t - u * (unsigned)n_ct) * DT_COLS; t = u;
        u = t / (unsigned)n_rt; const int r0 = (int)(t - u * (unsigned)n_rt) * TR; t = u;
        u = t / (unsigned)p.OD; const int id = (int)(t - u * (unsigned)p.OD);
        const int n = (int)u;
        // ---- stage: every wave is done with the previous tile's buffer (the barrier that ended it); rows / columns clamped on ragged tiles
        {
            int colv = c0 + first_w + j;
            colv = colv < Wp ? colv : Wp - 1;
            const float* src = x + (long)n * p.x_n_stride + ((id + first_d) * xd + colv * 16 + g * 4);
            constexpr int rows = CBN * 2 * SR;
            DT_ABL_DMA for (int i = wave; i < rows; i += DT_WAVES) {
                const int cb = i / (2 * SR), rem = i - cb * (2 * SR), sl = rem / SR;
                int row = r0 + first_h + (rem - sl * SR);
                row = row < Hp ? row : Hp - 1;
                __builtin_amdgcn_global_load_lds(DT_GLOBAL_PTR(src + (cb * xc + sl * xd + row * xh)), DT_LDS_PTR(lds + i * 1024), 16, 0, 0);
            }
        }
        const int cot = cg * CW + cw;
        const unsigned wlo = (unsigned)(((cot * 16 + j) * 16 + g * 4) * 4);
        // A "batch" = (class c, channel block cb): its nd*nh*nw weight fragments, requested ONE BATCH AHEAD (two register sets, static
        // rotation: the loops are fully unrolled).
        f32x4 wt[2][8];
        // (class and channel block are template arguments all the way down: left as `#pragma unroll` loops the 3,024-MFMA body was only
        // partly unrolled and the weight positions were computed at run time)
        auto wfetch = [&](auto BATCH) __attribute__((always_inline)) {
            constexpr int batch = decltype(BATCH)::value, c = batch / CBN, cb = batch % CBN, set = batch & 1;
            constexpr int nd = (c >> 2) ? 2 : 1, nh = ((c >> 1) & 1) ? 2 : 1, nw = (c & 1) ? 2 : 1;
#pragma unroll
            for (int a = 0; a < nd; ++a)
#pragma unroll
                for (int b = 0; b < nh; ++b)
#pragma unroll
                    for (int e = 0; e < nw; ++e)
                        wt[set][(a * 2 + b) * 2 + e] = __builtin_bit_cast(
                            f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, wlo, (unsigned)(cb * 27 + dt_use_pos(c, a, b, e)) * w_tap_b, 0));
        };
        wfetch(std::integral_constant<int, 0>{});
        const __attribute__((address_space(3))) char* bb = (const __attribute__((address_space(3))) char*)lds + lane_b;
        const int col = c0 + j;
        const bool col_ok = j < DT_COLS && col < p.OW;
        const int row0 = r0 + rg * RW;
        const f32x4 sc = *(const f32x4*)(p.scale + cot * 16 + g * 4);
        const f32x4 sh = *(const f32x4*)(p.shift + cot * 16 + g * 4);
        float* y = p.y + p.y_off0 + (long)n * p.y_n_stride + (long)cot * p.y_cb_stride + g * 4;
        const float* res = p.res ? p.res + p.r_off0 + (long)n * p.r_n_stride + (long)cot * p.r_cb_stride + g * 4 : nullptr;
        // vmcnt retires in order: a wave that waits for its next weight fragments (L2) also waits for every residual load and store
        // acknowledgement (HBM) issued before them.  Class by class -- eight groups of residual loads, four of stores per tile -- that
        // was 250 us of conv6's 1,466 at 1,024 ROIs (ablations: tools/experiments/README.md).  The block runs one wave per SIMD anyway
        // (its stage fills the LDS), so the register file is there: the residuals of the four even-depth classes are requested here, next
        // to the stage (one drain at the stage barrier), the epilogues overwrite them in place, and after class 3 their stores leave
        // together with the residual requests of the four odd-depth classes: two drains per tile instead of twelve.
        f32x4 ov[4][RW], acc[RW];
        auto out_off = [&](int c, int r, int sd, int sh_) __attribute__((always_inline)) {
            return (2 * id + (c >> 2)) * sd + (2 * (row0 + r) + ((c >> 1) & 1)) * sh_ + (2 * col + (c & 1)) * 16;
        };
        auto res_request = [&](int half) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    ov[c][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (DT_ABL_RES res && col_ok && row0 + r < p.OH) ov[c][r] = *(const f32x4*)(res + out_off(half * 4 + c, r, rd_, rh));
                }
        };
        // the two column-parity classes of a row side by side: both 64-byte voxels of each 128-byte line
        auto store_half = [&](int half) __attribute__((always_inline)) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int r = 0; r < RW; ++r)
                    if (DT_ABL_ST col_ok && row0 + r < p.OH) {
                        *(f32x4*)(y + out_off(half * 4 + 2 * c2, r, yd_, yh)) = ov[2 * c2][r];
                        *(f32x4*)(y + out_off(half * 4 + 2 * c2 + 1, r, yd_, yh)) = ov[2 * c2 + 1][r];
                    }
        };
        res_request(0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        auto do_batch = [&](auto BATCH) __attribute__((always_inline)) {
            constexpr int batch = decltype(BATCH)::value, c = batch / CBN, cb = batch % CBN;
            constexpr int pd = c >> 2, ph = (c >> 1) & 1, pw = c & 1;
            constexpr int nd = pd ? 2 : 1, nh = ph ? 2 : 1, nw = pw ? 2 : 1;
            if constexpr (cb == 0) {
#pragma unroll
                for (int r = 0; r < RW; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if constexpr (batch == 4 * CBN) {          // (after this batch's weights were requested one batch ago; before the next request)
                store_half(0);
                res_request(1);
            }
            DT_ABL_W if constexpr (batch + 1 < 8 * CBN) wfetch(std::integral_constant<int, batch + 1>{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < nd; ++a)
#pragma unroll
                for (int e = 0; e < nw; ++e) {
                    f32x4 bf[RW + 1];                                  // rows r .. r + RW of (slice a, column shift e): both row taps read them
#pragma unroll
                    for (int i = 0; i < RW + nh - 1; ++i) bf[i] = *(dt_lds_frag*)(bb + ((cb * 2 + a) * SR + i) * 1024 + e * 16);
#pragma unroll
                    for (int b = 0; b < nh; ++b)
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int r = 0; r < RW; ++r)
                                acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[batch & 1][(a * 2 + b) * 2 + e][s], bf[r + b][s], acc[r], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (cb == CBN - 1) {
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    f32x4 v = acc[r] * sc + sh + ov[c & 3][r];
                    if (p.relu) v = __builtin_elementwise_max(v, (f32x4){0.f, 0.f, 0.f, 0.f});
                    ov[c & 3][r] = v;
                }
            }
        };
        dt_static_for<8 * CBN>(do_batch);
        store_half(1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");         // the buffer is free for the next tile's stage
    }
}

template <int RW, int CW, int CBN>
int launch_rw(const drc_tapconv_params& p, hipStream_t stream) {
    constexpr int TR = RW * (DT_WAVES / CW);
    constexpr size_t lds = (size_t)CBN * 2 * (TR + 1) * 1024 + 1024;
    static_assert(lds <= 160 * 1024, "stage exceeds the LDS");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)deconvtile_kernel<RW, CW, CBN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const long tiles = (long)p.N * p.OD * ((p.OH + TR - 1) / TR) * ((p.OW + DT_COLS - 1) / DT_COLS) * (p.cout_pad / 16 / CW);
    if (tiles >= (1L << 31)) return -5;
    long per_cu = (160 * 1024) / (long)lds;
    if (per_cu > 2) per_cu = 2;
    long blocks = 256 * per_cu;
    if (blocks > tiles) blocks = tiles;
    hipLaunchKernelGGL((deconvtile_kernel<RW, CW, CBN>), dim3((unsigned)blocks), dim3(64 * DT_WAVES), lds, stream, p);
    return (int)hipGetLastError();
}

// rows per wave (engine.deconvtile_rows mirrors this): seven for one or two row groups, four for four (the stage of 28 rows x 64
// channels would not fit); the small tile (four / two rows) when the big one pads the map's rows by more than 25 % over the small one's
template <int CW, int CBN>
int launch_cw(const drc_tapconv_params& p, hipStream_t stream) {
    constexpr int RG = DT_WAVES / CW;
    constexpr int BIG = RG == 4 ? 4 : 7, SMALL = RG == 4 ? 2 : 4;
    const int trb = BIG * RG, trs = SMALL * RG;
    const long padb = (long)((p.OH + trb - 1) / trb) * trb, pads = (long)((p.OH + trs - 1) / trs) * trs;
#ifdef DT_FORCE_SMALL
    return launch_rw<SMALL, CW, CBN>(p, stream);
#endif
    return padb * 4 <= pads * 5 ? launch_rw<BIG, CW, CBN>(p, stream) : launch_rw<SMALL, CW, CBN>(p, stream);
}

template <int CBN>
int launch_cb(const drc_tapconv_params& p, hipStream_t stream) {
    const int ct = p.cout_pad / 16;
    if (ct % 4 == 0) return launch_cw<4, CBN>(p, stream);
    if (ct % 2 == 0) return launch_cw<2, CBN>(p, stream);
    return launch_cw<1, CBN>(p, stream);
}

}  // namespace

extern "C" int drc_deconv3d_k3s2_tile_supported(const drc_tapconv_params* pp) {
    if (!pp) return 0;
    const drc_tapconv_params& p = *pp;
    if (p.n_classes != 8 || p.in_mul != 1 || p.out_mul != 2) return 0;
    for (int c = 0; c < 8; ++c) {
        const drc_tap_class& k = p.cls[c];
        const int pd = c >> 2, ph = (c >> 1) & 1, pw = c & 1;
        if (k.nd != (pd ? 2 : 1) || k.nh != (ph ? 2 : 1) || k.nw != (pw ? 2 : 1) || k.sd != 1 || k.sh != 1 || k.sw != 1) return 0;
        if (k.out_off_d != pd || k.out_off_h != ph || k.out_off_w != pw) return 0;
        if (k.dd0 != p.cls[0].dd0 || k.dh0 != p.cls[0].dh0 || k.dw0 != p.cls[0].dw0) return 0;
    }
    if (p.cb_in != 2 && p.cb_in != 4) return 0;         // 32 or 64 input channels (the instantiated stages); others keep deconvdirect.hip
    return p.cout_pad > 0 && !(p.cout_pad & 15);
}

extern "C" int drc_deconv3d_k3s2_tile_fwd(const drc_tapconv_params* pp, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD <= 0 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (!drc_deconv3d_k3s2_tile_supported(pp)) return -4;
    if (p.x_h_stride <= 0 || p.x_d_stride % p.x_h_stride || p.x_h_stride % 16) return -4;
    if ((int64_t)p.cb_in * 27 * p.cout_pad * 64 >= (1LL << 31)) return -5;               // 32-bit byte offsets inside the weights
    if (p.x_n_stride * 4 >= (1LL << 31) || p.y_cb_stride * 4 >= (1LL << 31) || (p.res && p.r_cb_stride * 4 >= (1LL << 31))) return -5;
    hipStream_t s = (hipStream_t)stream;
    return p.cb_in == 4 ? launch_cb<4>(p, s) : launch_cb<2>(p, s);
}
