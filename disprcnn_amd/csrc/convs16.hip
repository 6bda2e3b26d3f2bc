// convs16.hip -- 3x3x3 stride-1 convolution (+BN, +residual, +ReLU) in SPLIT-f16 arithmetic on the f16 matrix cores (gfx950 / CDNA4), round 5.
//
//   reference arithmetic: convbn_3d k3 s1 p1 of stackhourglass.py:7-51,63-88 (dres0 / dres1 / classif[0], hourglass conv2 / conv4) and, fused
//   into the first layer's loads, the concat cost volume of stackhourglass.py:115-128; fp32 (config/defaults.py:22).
//
// Why.  Rounds 2-4 ran these layers as Winograd F(2^3,3^3) on v_mfma_f32_32x32x2_f32.  That instruction executes at the fp32 VECTOR
// rate (157 TFLOP/s, 1/16 of the f16 MFMA) on the SIMD's vector ALU, where it competes with the Winograd butterflies: 0.48 of that
// peak was the end of the road.  Here every fp32 value v is carried as TWO fp16 numbers, hi = fp16(v) and
// lo = fp16(v - hi), i.e. 22-24 significant bits (the fp32 input itself has 24), and a product is three f16 MFMAs into one fp32
// accumulator:  a*w ~= a_hi*w_hi + a_lo*w_hi + a_hi*w_lo  (the dropped a_lo*w_lo is <= 2^-22 |a*w|).  Each f16 product is exact in
// fp32 (11 x 11 bits) and the accumulation is fp32, so the result carries fp32-class error (measured against fp64 next to the
// fp32 FMA chain: tools/experiments/exp_s16.py) at 3/16 of the fp32-MFMA cost: peak 2.5 PF / 3 = 833 TFLOP/s fp32-equivalent.
// Weights are pre-scaled by a power of two (wexp) so that their lo parts stay normal fp16 numbers; 2^-wexp is folded into `scale`.
// Activations are stored unscaled: a lo part below 2^-14 is a subnormal fp16 (the f16 MFMA does not flush it), absolute error <= 2^-25.
// Range: |activation| <= 65504 (fp16 max); BatchNorm'd PSMNet activations are O(1..100).
//
// Layout "RS16" (row-split-16): halfs [N][C/32][D+2pd][H+2][8 chunks][W+2][8], zero halo of one voxel stored (pd = 1 for volumes, 0 for
// 2D maps).  A 32-channel block of one voxel is 8 chunks of 8 halfs: chunk q = p*4 + s*2 + g, p = 0 hi / 1 lo, (s, g) = the k-step and
// k-group of v_mfma_f32_32x32x16_f16 that consume it, element e <-> channel 4g + 8(2s + (e>>2)) + (e&3).  That is exactly the set of
// couts lane (n, g) of the PRODUCING layer's accumulator holds (C/D map: row = (reg&3) + 8(reg>>2) + 4(lane>>5)), so an epilogue
// stores whole 16-byte chunks and the consumer's B fragment is one 16-byte LDS read: no shuffles on either side.
//
// Kernel.  A = weights (32 couts x 16 k), B = activations (16 k x 32 voxels: one image row of 28, lanes 28..31 idle -- the 32x32x16
// form is 15 % faster than 16x16x32 at peak, which pays for the 12.5 %).  A workgroup of four waves owns 4/KW rows x 28 columns of ALL
// depth planes of one unit ("column") and walks the depth: one slab ((rows + 2) x 30 voxels of one input plane) per step through a
// ring of three LDS slots filled by LDS-DMA two steps ahead.  K is split over KW = cin/16 waves (wave k multiplies 16 input channels),
// so a wave holds ALL 27 taps of its K slice in registers (27 x (hi, lo) x 4 VGPRs = 216) and the input is streamed exactly once:
//   per input plane and wave: 9 B-fragment pairs from LDS (18 ds_read_b128), 81 MFMAs (27 taps x 3 products) into the accumulators of
//   the three output planes the slab touches.  A finished plane's accumulator is exchanged between the K-split waves through LDS
//   (each keeps 16/KW registers = the couts it stores), one barrier per step; BN / residual / ReLU / hi-lo split run in the shadow of
//   the next plane's MFMAs.
// CV = the first layer: the slab rows are built from the left / right feature maps (RS16 2D) -- left where the shifted pixel exists,
// right moved by lo4 + plane columns, everything else fetched from the zero halo -- so the 64-channel volume never exists.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
// s_waitcnt immediate (gfx9 encoding: vmcnt[3:0] | expcnt << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14); the BUILTIN, which the compiler's own
// wait-count pass reads (conv16t.hip)
#define S16_WAITCNT(vm, lgkm) (((vm) & 15) | (7 << 4) | ((lgkm) << 8) | (((vm) >> 4) << 14))

namespace {

constexpr int PV = 128;         // voxels per chunk plane of a slab (>= (rows + 2) * SX + the 4 columns lanes 28..31 over-read)
constexpr int CPB = PV * 16;    // bytes per chunk plane
constexpr int RING = 3;

// RT rows x WT columns of the map make one 32-voxel MFMA tile: 1 x 28 (full-resolution maps, tiles along x), 2 x 14 and 4 x 7 (the
// hourglass' half- and quarter-resolution maps: 28 of 32 lanes busy there as well).
template <int KW, bool CV, int RT = 1, int WT = 28>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void convs16_kernel(const drc_s16conv_params p) {
    constexpr int TX = WT;                      // output columns per tile
    constexpr int SX = WT + 2;                  // staged columns (TX + halo)
    constexpr int RPW = 4 / KW;                 // MFMA tiles per workgroup (RPW * RT output rows)
    constexpr int CBI = KW / 2;                 // 32-channel input blocks
    constexpr int SROWS = RPW * RT + 2;
    static_assert(RT * WT <= 32 && (RT == 1 || !CV), "tile shape");
    constexpr int SLAB = CBI * 8 * CPB;
    constexpr int NL = CBI * 8 * (PV / 64) / 4; // LDS-DMA instructions per wave and slab
    constexpr int OWN = 16 / KW;                // accumulator registers (couts per lane) a wave finishes
    constexpr int XW = 4096;                    // bytes a wave publishes per plane: its 16 accumulator registers
    constexpr int NR = 2;                       // residual loads per step (hi, lo), always issued
    constexpr int NS = 4;                       // stores per step (RS16 hi, lo + blocked fp32 x2), always issued (dropped when unused)
    static_assert(RT == 1 ? SROWS * SX + 4 <= PV : (SROWS + 1) * SX + 4 <= PV, "slab plane too small (incl. the rows / columns the idle lanes over-read)");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* ring = lds;
    char* xchg = lds + RING * SLAB;             // [2 parities][4 waves][XW]

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int n_ = lane & 31, g = lane >> 5;
    const int rl = RT == 1 ? 0 : n_ / WT, xl = RT == 1 ? n_ : n_ - (n_ / WT) * WT;      // this lane's voxel inside the MFMA tile (lanes >= RT*WT idle)
    const int r = wave / KW, k = wave % KW;     // MFMA tile of the workgroup, K slice
    const int ct = blockIdx.y;                  // cout tile of 32

    const int D = p.D, H = p.H, W = p.W;
    const int Wp = W + 2, Hp = H + 2;
    const long rowB = (long)Wp * 128;           // bytes: 8 chunks x Wp x 16
    const long planeB = (long)Hp * rowB;
    const long xcbB = (long)(D + 2) * planeB;
    const long xnB = (long)CBI * xcbB;
    const int cbo = p.cout / 32;
    const long ynB = (long)cbo * xcbB;          // RS16 output / residual: same spatial geometry
    const long mapnB = planeB;                  // 2D feature maps (one 32-channel block, no depth halo)
    // blocked fp32 output: float[N][cout/16][D+2][H+2][W+2][16]
    const long b_rowB = (long)Wp * 64, b_planeB = (long)Hp * b_rowB, b_cbB = (long)(D + 2) * b_planeB, b_nB = (long)(p.cout / 16) * b_cbB;

    // ---- weights of this wave's K slice: all 27 taps, (hi, lo), registers for the lifetime of the workgroup
    f16x8 wh[27], wl[27];
    {
        const char* wb = (const char*)p.w + ((long)(ct * KW + k) * 54) * 1024 + lane * 16;
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            wh[t] = *(const f16x8*)(wb + (t * 2) * 1024);
            wl[t] = *(const f16x8*)(wb + (t * 2 + 1) * 1024);
        }
    }
    // ---- BN scale / shift of the couts this lane finishes: element e <-> cout ct*32 + 4g + 8*(own slot) + ...
    float sc[OWN], sh[OWN];
#pragma unroll
    for (int e = 0; e < OWN; ++e) {
        const int reg = k * OWN + e;
        const int co = ct * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * g;
        sc[e] = p.scale[co];
        sh[e] = p.shift[co];
    }
    // ---- LDS-DMA geometry (column independent): this wave's instructions id = wave*NL + i -> (cb, chunk, half of the plane)
    int srcrow[2], srcx[2];
    bool srcok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int v = h * 64 + lane;
        srcok[h] = v < SROWS * SX;
        const int rr = srcok[h] ? v / SX : 0;
        srcrow[h] = rr;
        srcx[h] = srcok[h] ? v - rr * SX : 0;
    }
    const unsigned bfrag = (unsigned)(((k >> 1) * 8 + (k & 1) * 2 + g) * CPB + ((r * RT + rl) * SX + xl) * 16);     // hi; lo at + 4*CPB
    const __attribute__((address_space(3))) char* ringl = (const __attribute__((address_space(3))) char*)ring;
    typedef const __attribute__((address_space(3))) f16x8 lds_frag;

    const int n_xt = W / TX, n_yt = (H + RPW * RT - 1) / (RPW * RT);
    // XCD-aware order: block b runs on XCD b % 8; the 32 blocks of an XCD take 32 consecutive columns of the units n % 8 == xcd, so that the
    // row tiles sharing halo rows meet in one L2
    const unsigned xcd = blockIdx.x & 7, qx = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const unsigned cols_unit = (unsigned)n_yt * n_xt;

    for (unsigned it = 0;; ++it) {
        const unsigned j = it * per_xcd + qx;
        const unsigned nl = j / cols_unit;
        const unsigned n = nl * 8 + xcd;
        if (n >= (unsigned)p.N) break;
        const unsigned rem = j - nl * cols_unit;
        const int yb = (int)(rem / n_xt), xt = (int)(rem - (unsigned)yb * n_xt);
        const int y0 = yb * RPW * RT, x0 = xt * TX;

        // source bases
        const char* xcol = CV ? nullptr : (const char*)p.x + (long)n * xnB + (long)y0 * rowB + (long)x0 * 16;
        const char* lcol = CV ? (const char*)p.left + (long)n * mapnB + (long)y0 * rowB : nullptr;
        const char* rcol = CV ? (const char*)p.right + (long)n * mapnB + (long)y0 * rowB : nullptr;
        auto stage = [&](int plane, int slot) __attribute__((always_inline)) {       // logical input plane (clamped) -> ring slot
            const int pl = plane < D ? plane : D - 1;
            char* dst = ring + slot * SLAB;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int id = wave * NL + i;
                const int cb = id >> 4, c = (id >> 1) & 7, h = id & 1;
                const char* src;
                if constexpr (CV) {
                    const int ish = p.lo4 + pl;
                    const int xl = x0 + srcx[h] - 1;                                   // logical column
                    const bool ok = srcok[h] && xl >= 0 && xl < W && xl - ish >= 0 && xl - ish < W;
                    const int col = ok ? (cb ? x0 + srcx[h] - ish : x0 + srcx[h]) : 0;   // column 0 = the zero halo
                    src = (cb ? rcol : lcol) + (long)srcrow[h] * rowB + (long)c * (Wp * 16) + (long)col * 16;
                } else {
                    src = xcol + (long)cb * xcbB + (long)(pl + 1) * planeB + (long)srcrow[h] * rowB + (long)c * (Wp * 16) + (long)srcx[h] * 16;
                }
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(dst + (cb * 8 + c) * CPB + h * 1024), 16, 0, 0);
            }
        };
        // outputs / residual of this unit as buffers (dropped lanes point past the range)
        const __amdgpu_buffer_rsrc_t y16r = __builtin_amdgcn_make_buffer_rsrc(p.y16 ? (void*)((char*)p.y16 + (long)n * ynB) : (void*)p.w, 0, p.y16 ? 0x7FFFFF00 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t y32r = __builtin_amdgcn_make_buffer_rsrc(p.y32 ? (void*)((char*)p.y32 + (long)n * b_nB) : (void*)p.w, 0, p.y32 ? 0x7FFFFF00 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t resr = __builtin_amdgcn_make_buffer_rsrc(p.res ? (void*)((const char*)p.res + (long)n * ynB) : (void*)p.w, 0, p.res ? 0x7FFFFF00 : 0, 0x00020000);
        const int yl = y0 + r * RT + rl;                                  // this lane's output row
        const bool lane_ok = n_ < RT * TX && yl < H;
        // byte offsets of this lane's output voxel (row yl, column x0 + xl) in plane 0 (padded coordinates + 1)
        unsigned o16, o32;
        if constexpr (KW == 2) {
            // own registers 8k..8k+7 = chunk (s = k, g) complete: 16 B hi at chunk k*2+g, lo at 4 + k*2 + g
            o16 = (unsigned)((long)ct * xcbB + planeB + (long)(yl + 1) * rowB + (long)(k * 2 + g) * (Wp * 16) + (long)(x0 + xl + 1) * 16);
            // blocked fp32: block 2ct + k, channels 4g..4g+3 and 8+4g..
            o32 = (unsigned)((long)(2 * ct + k) * b_cbB + b_planeB + (long)(yl + 1) * b_rowB + (long)(x0 + xl + 1) * 64 + g * 16);
        } else {
            // own registers 4k..4k+3 = couts 8k + 4g + e: half a chunk: chunk (s = k>>1, g), bytes (k&1)*8..
            o16 = (unsigned)((long)ct * xcbB + planeB + (long)(yl + 1) * rowB + (long)((k >> 1) * 2 + g) * (Wp * 16) + (long)(x0 + xl + 1) * 16 + (k & 1) * 8);
            o32 = (unsigned)((long)(2 * ct + (k >> 1)) * b_cbB + b_planeB + (long)(yl + 1) * b_rowB + (long)(x0 + xl + 1) * 64 + (k & 1) * 32 + g * 16);
        }
        const unsigned lo_off = (unsigned)(4 * Wp * 16);
        const float relu_lo = p.relu ? 0.f : -3.0e38f;

        f32x16 acc[3];
        u32x4 resv[2];           // residual (hi, lo) of the plane finalized next step (KW == 4: low 8 bytes used)

        // every wave is done with the previous column's slots and exchange buffers
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(S16_WAITCNT(63, 0));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage(0, 0);
        stage(1, 1);
        __builtin_amdgcn_s_waitcnt(S16_WAITCNT(NL, 15));      // plane 0 landed (step 0's own wait assumes a full previous step)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
        resv[0] = resv[1] = (u32x4){0u, 0u, 0u, 0u};

        // one step: input plane t (COMPUTE) -> accumulators of output planes t-1, t, t+1; publish plane t-1; finalize plane t-2.
        // Straight-line code (no branch between the barrier and the publish): the finalize of plane t-2 is spread over the tap groups, so its
        // VALU / LDS / store instructions issue in the shadow of the MFMAs.
        auto step = [&](int t, auto JT, auto CT_) __attribute__((always_inline)) {
            constexpr int J = decltype(JT)::value;              // t mod 3
            constexpr bool COMPUTE = decltype(CT_)::value;
            constexpr int A0 = (J + 1) % 3, A1 = J, A2 = (J + 2) % 3;    // accumulators of output planes t+1 (kd 0), t (kd 1), t-1 (kd 2)
            // slab t landed (everything but the previous step's requests may be waited for: vmcnt retires in order); lgkmcnt: my exchange
            // writes are visible before the barrier
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(S16_WAITCNT(NL + NS + NR, 0));
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stage(t + 2, (J + 2) % 3);                           // slot of plane t-1: free since the barrier
            // the K-split partial sums of plane t-2 (published in step t-1), this wave's couts: registers k*OWN .. of every K slice
            f32x4 part[4];
            {
                const char* xb = xchg + ((t - 1) & 1) * (4 * XW) + (r * KW) * XW + lane * 16;
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
            // the residual of plane t-2 was requested at the end of step t-1; only this step's DMAs are younger
            __builtin_amdgcn_s_waitcnt(S16_WAITCNT(NL, 15));
            __builtin_amdgcn_sched_barrier(0);
            float v[OWN];
            _Float16 vh[OWN], vl[OWN];
            auto fin = [&](int e) __attribute__((always_inline)) {
                float s_;
                if constexpr (KW == 2) s_ = part[(e >> 2) * 2][e & 3] + part[(e >> 2) * 2 + 1][e & 3];
                else s_ = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
                float x_ = s_ * sc[e] + sh[e];
                _Float16 rh, rl;
                if constexpr (KW == 2) { rh = __builtin_bit_cast(f16x8, resv[0])[e]; rl = __builtin_bit_cast(f16x8, resv[1])[e]; }
                else { rh = __builtin_bit_cast(f16x8, resv[0])[e]; rl = __builtin_bit_cast(f16x8, resv[1])[e]; }
                x_ += (float)rh + (float)rl;
                x_ = fminf(fmaxf(x_, relu_lo), 65504.f);            // (fp16 range of the hi part; relu_lo = -3e38 without ReLU, then the
                x_ = fmaxf(x_, -65504.f);                              //  lower clamp applies)
                v[e] = x_;
                vh[e] = (_Float16)x_;
                vl[e] = (_Float16)(x_ - (float)vh[e]);
            };
            auto stores = [&]() __attribute__((always_inline)) {
                const int q_ = t - 2;
                const bool ok = lane_ok && q_ >= 0;
                const unsigned po = ok ? (unsigned)((long)q_ * planeB) : 0x80000000u;
                const unsigned po32 = ok ? (unsigned)((long)q_ * b_planeB) : 0x80000000u;
                if constexpr (KW == 2) {
                    f16x8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { hi[e] = vh[e]; lo[e] = vl[e]; }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), y16r, o16 + po, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), y16r, o16 + lo_off + po, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){v[0], v[1], v[2], v[3]}), y32r, o32 + po32, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){v[4], v[5], v[6], v[7]}), y32r, o32 + 32 + po32, 0, 0);
                } else {
                    f16x4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { hi[e] = vh[e]; lo[e] = vl[e]; }
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), y16r, o16 + po, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), y16r, o16 + lo_off + po, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){v[0], v[1], v[2], v[3]}), y32r, o32 + po32, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){v[0], v[1], v[2], v[3]}), y32r, 0x80000000u, 0, 0);   // (count)
                }
            };
            // plane t-1 complete: publish the accumulator (every K slice publishes all 16 registers; the finisher of a cout group sums
            // the slices in a fixed order), clear it for plane t+2, request the residual
            auto publish = [&]() __attribute__((always_inline)) {
                const f32x16 a = acc[A2];
                char* xb = xchg + (t & 1) * (4 * XW) + wave * XW + lane * 16;
#pragma unroll
                for (int q = 0; q < 4; ++q) *(f32x4*)(xb + q * 1024) = (f32x4){a[q * 4], a[q * 4 + 1], a[q * 4 + 2], a[q * 4 + 3]};
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[A2][e] = 0.f;
                const int q_ = t - 1;
                const bool ok = lane_ok && q_ >= 0 && q_ < D;
                const unsigned po = ok ? (unsigned)((long)q_ * planeB) : 0x80000000u;
                if constexpr (KW == 2) {
                    resv[0] = __builtin_amdgcn_raw_buffer_load_b128(resr, o16 + po, 0, 0);
                    resv[1] = __builtin_amdgcn_raw_buffer_load_b128(resr, o16 + lo_off + po, 0, 0);
                } else {
                    const u32x2 a_ = __builtin_amdgcn_raw_buffer_load_b64(resr, o16 + po, 0, 0);
                    const u32x2 b_ = __builtin_amdgcn_raw_buffer_load_b64(resr, o16 + lo_off + po, 0, 0);
                    resv[0] = (u32x4){a_.x, a_.y, 0u, 0u};
                    resv[1] = (u32x4){b_.x, b_.y, 0u, 0u};
                }
            };
            if constexpr (COMPUTE) {
                const __attribute__((address_space(3))) char* sb = ringl + J * SLAB + bfrag;
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
                    const int t0 = kh * 3 + kw, t1 = 9 + t0, t2 = 18 + t0;       // taps kd = 0, 1, 2
                    if (q < 8) {
                        acc[A0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t0], h_, acc[A0], 0, 0, 0);
                        acc[A1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t1], h_, acc[A1], 0, 0, 0);
                        acc[A2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t2], h_, acc[A2], 0, 0, 0);
                        acc[A0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t0], l_, acc[A0], 0, 0, 0);
                        acc[A1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t1], l_, acc[A1], 0, 0, 0);
                        acc[A2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t2], l_, acc[A2], 0, 0, 0);
                        acc[A0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[t0], h_, acc[A0], 0, 0, 0);
                        acc[A1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[t1], h_, acc[A1], 0, 0, 0);
                        acc[A2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[t2], h_, acc[A2], 0, 0, 0);
                        if (q < OWN) fin(q);
                        if (q == OWN) stores();
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        // last group: the finished plane's accumulator first, its publication in the shadow of the rest
                        if (q == OWN) stores();
                        acc[A2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t2], h_, acc[A2], 0, 0, 0);
                        acc[A0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t0], h_, acc[A0], 0, 0, 0);
                        acc[A2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t2], l_, acc[A2], 0, 0, 0);
                        acc[A1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t1], h_, acc[A1], 0, 0, 0);
                        acc[A2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[t2], h_, acc[A2], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        acc[A0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t0], l_, acc[A0], 0, 0, 0);
                        acc[A1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t1], l_, acc[A1], 0, 0, 0);
                        acc[A0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[t0], h_, acc[A0], 0, 0, 0);
                        acc[A1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[t1], h_, acc[A1], 0, 0, 0);
                        publish();
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < OWN; ++e) fin(e);
                stores();
                publish();
            }
        };

        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
#pragma unroll 1
        for (int t0 = 0; t0 < D; t0 += 3) {
            step(t0, I0{}, std::true_type{});
            step(t0 + 1, I1{}, std::true_type{});
            step(t0 + 2, I2{}, std::true_type{});
        }
        step(D, I0{}, std::false_type{});
        step(D + 1, I1{}, std::false_type{});
    }
}

template <int KW, bool CV, int RT = 1, int WT = 28>
int launch(const drc_s16conv_params& p, hipStream_t stream) {
    constexpr int CBI = KW / 2;
    constexpr int SLAB = CBI * 8 * CPB;
    constexpr int XW = 4096;
    constexpr size_t lds = RING * SLAB + 2 * 4 * XW;
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)convs16_kernel<KW, CV, RT, WT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    constexpr int rows = (4 / KW) * RT;
    const long columns = (long)p.N * ((p.H + rows - 1) / rows) * (p.W / WT);
    // one block per CU (the weights take the register file); a multiple of 8 so that every XCD runs the same number
    long blocks = 256;
    while (blocks > 8 && blocks / 2 >= columns) blocks /= 2;
    hipLaunchKernelGGL((convs16_kernel<KW, CV, RT, WT>), dim3((unsigned)blocks, (unsigned)(p.cout / 32)), dim3(256), lds, stream, p);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int drc_conv3d_k3_s16_supported(int cin, int cout, int D, int H, int W) {
    if (cin != 32 && cin != 64) return 0;
    if (cout != 32 && cout != 64) return 0;
    if (D <= 0 || D % 3) return 0;
    if (H <= 0) return 0;
    if (W == 14 || W == 7) return 1;                 // 2 x 14 and 4 x 7 tiles (ragged last row tile masked)
    if (W <= 0 || W % 28) return 0;
    if (H % (cin == 32 ? 2 : 1)) return 0;
    return 1;
}

extern "C" int drc_conv3d_k3_s16_fwd(const drc_s16conv_params* pp, void* stream) {
    if (!pp) return -1;
    const drc_s16conv_params& p = *pp;
    const bool cv = p.left || p.right;
    if (!p.w || !p.scale || !p.shift || (!p.y16 && !p.y32)) return -1;
    if (cv ? (!p.left || !p.right || p.cin != 64) : !p.x) return -1;
    if (p.N < 0) return -2;
    if (!drc_conv3d_k3_s16_supported(p.cin, p.cout, p.D, p.H, p.W)) return -4;
    if (cv && p.W % 28) return -4;
    if (p.N == 0) return 0;
    // 32-bit offsets inside one unit
    const long unit16 = (long)(p.cout / 32) * (p.D + 2) * (p.H + 2) * (p.W + 2) * 128;
    if (unit16 >= 0x7FFFFF00L / 2) return -5;
    hipStream_t s = (hipStream_t)stream;
    if (cv) return launch<4, true>(p, s);
    if (p.W == 14) return p.cin == 32 ? launch<2, false, 2, 14>(p, s) : launch<4, false, 2, 14>(p, s);
    if (p.W == 7) return p.cin == 32 ? launch<2, false, 4, 7>(p, s) : launch<4, false, 4, 7>(p, s);
    return p.cin == 32 ? launch<2, false>(p, s) : launch<4, false>(p, s);
}
