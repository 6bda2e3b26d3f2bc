// EXPERIMENT (round 6), not built into libdisprcnn_hip.so: convs16u.hip with the residual tiles fetched by LDS-DMA into a three-slot ring two
// steps ahead (instead of registers one step ahead).  Correct (tests/test_hip_s16.py, test_hip_overflow.py: 140 passed with it) and 63 VGPRs
// lighter (412 instead of 475) -- and NOT faster: conv6 at 1024 ROIs 809 us in the bench step against 794-839 us for the register form
// (profiles/r6_exp_u_ablation.log).  The ablation of the same run says why: without any MFMA the kernel still takes 779 us of 893 -- it is the
// memory pipeline that sets the pace (no stores -185 us, no residual -228, no slab staging -160, all three -365), and a plain 4-wave copy
// kernel with this access pattern (16-byte slots at a 32-byte stride, the other half written by another wave) reaches 4.5-4.9 TB/s against
// 5.0-5.1 TB/s contiguous (tools/experiments/strided_store_probe.*): the layer's 3.0 GB cannot move in less than ~600 us.  The prefetch
// distance was not the limit; the register form stays the product.
// convs16u.hip -- ConvTranspose3d(k3, s2, p1, output_padding 1) (+BN, +residual, +ReLU) in split-f16 arithmetic on the f16 matrix cores
// (gfx950 / CDNA4), round 5.
//
//   reference: hourglass conv5 / conv6, stackhourglass.py:22-30,44-49; fp32 (config/defaults.py:22).
//
// Arithmetic and the RS16 layout: convs16.hip (hi + lo fp16 pairs, three v_mfma_f32_32x32x16_f16 per fp32 product, fp32 accumulate).
// o = 2i - 1 + k: an even output has one tap per dimension (k = 1, input i), an odd one two (k = 2 at input i, k = 0 at input i + 1):
// 8 output-parity classes (pz, py, px) with (1+pz)(1+py)(1+px) taps -- 27 in all, each of them used exactly once per input voxel.
// The classes are DISJOINT outputs, so the workgroup's four waves split the CLASSES, not K (no partial sums, no exchange):
//     wave 0: (1,1,1) 8 taps | wave 1: (0,1,1) + (1,0,1) 4 + 4 | wave 2: (1,1,0) + (0,0,1) 4 + 2 | wave 3: (1,0,0) + (0,1,0) + (0,0,0) 2 + 2 + 1
// and every wave multiplies all 64 input channels (4 k-steps of 16): its taps' weights stay in registers (<= 8 taps x 4 x (hi, lo) = 256
// VGPRs).  An MFMA tile is RT x WT INPUT voxels (2 x 14, 4 x 7, 1 x 28); the workgroup walks the input planes of one column: plane zi gives
// output plane 2zi (kd = 1: complete at once), closes 2zi - 1 (kd = 0) and opens 2zi + 1 (kd = 2).  The staged slab is (RT + 1) x (WT + 1)
// voxels (the shifts are {0, +1}^2): one LDS-DMA instruction per 8-channel chunk plane.  A step runs the taps that COMPLETE accumulators
// first; their epilogues (BN, residual, ReLU, hi/lo split, 16-byte stores of whole chunks) issue in the shadow of the taps that open the
// next plane's accumulators.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/disprcnn_hip.h"
#include "s16_ovf.h"
#include "s16_tilemap.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define S16_WAITCNT(vm, lgkm) (((vm) & 15) | (7 << 4) | ((lgkm) << 8) | (((vm) >> 4) << 14))

namespace {

constexpr int PV = 64;            // voxels per chunk plane of a slab
constexpr int CPB = PV * 16;
constexpr int NCH = 16;           // chunk planes of a slab: 64 input channels
constexpr int SLAB = NCH * CPB;
constexpr int RING = 3;           // slots of the input-slab ring: a slab is requested two steps ahead
constexpr int RD = 3;             // slots of the residual ring (round 6): a step's residual tiles are requested two steps ahead, by LDS-DMA
constexpr int RESSLOT = 8 * 4096; // bytes of one residual-ring slot: 8 finished accumulators per workgroup and step (1 + 2 + 2 + 3 over the four roles) x 4 KiB
constexpr int NL = NCH / 4;       // LDS-DMA instructions per wave and slab

// ---- roles: the classes (pz*4 + py*2 + px) a wave owns
struct Role { int n; int cls[3]; };
constexpr Role kRoles[4] = {{1, {7, 0, 0}}, {2, {3, 5, 0}}, {2, {6, 1, 0}}, {3, {4, 2, 0}}};
// per dimension and parity: p = 0: (k 1, shift 0); p = 1: (k 2, shift 0), (k 0, shift 1)
struct UTap { int ci, w, kd, sy, sx, open; };        // class slot of the role, weight tap kd*9+kh*3+kw, kd, in-plane shifts, 1: opens the next plane (kd 2)
struct TapList { int n; UTap t[8]; };
constexpr TapList make_taps(int role) {
    TapList L{};
    // completing taps first (kd 1 of the pz = 0 classes, kd 0 of the pz = 1 classes), then the opening ones (kd 2)
    for (int open = 0; open < 2; ++open)
        for (int ci = 0; ci < kRoles[role].n; ++ci) {
            const int c = kRoles[role].cls[ci], pz = c >> 2, py = (c >> 1) & 1, px = c & 1;
            for (int a = 0; a < 1 + pz; ++a) {
                const int kd = pz ? (a ? 0 : 2) : 1;
                if ((kd == 2) != (open == 1)) continue;
                for (int b = 0; b < 1 + py; ++b)
                    for (int d = 0; d < 1 + px; ++d) {
                        const int kh = py ? (b ? 0 : 2) : 1, kw = px ? (d ? 0 : 2) : 1;
                        L.t[L.n++] = UTap{ci, kd * 9 + kh * 3 + kw, kd, py ? b : 0, px ? d : 0, open};
                    }
            }
        }
    return L;
}

template <int ROLE, int RT, int WT>
__device__ __forceinline__ void body(const drc_s16conv_params& p, char* lds, int wave, int lane) {
    constexpr TapList TL = make_taps(ROLE);
    constexpr Role R = kRoles[ROLE];
    constexpr int NT = TL.n;
    constexpr int NF = R.n;                       // accumulators finished (epilogues) per step
    constexpr int NS = 4 * NF;                    // stores (= residual loads) per step
    constexpr int SXI = WT + 1;
    static_assert((RT + 2) * SXI + 4 <= PV || RT == 1, "slab plane");
    char* ring = lds;
    float* bnlds = (float*)(lds + RING * SLAB);   // [2 g][2: scale, shift][16] floats of this cout tile
    // residual ring: [RD slots][8 accumulators of the workgroup: this role's NF start at ROFF][4: hi s0, hi s1, lo s0, lo s1][64 lanes][16 B].
    // Round 5 held a step's residual tiles in registers, requested ONE step ahead (96 VGPRs for the three-class role) and waited for them at
    // the head of the step: with ~1.5 us of MFMAs per step and one wave per SIMD the kernel sat at 3.9 TB/s waiting (wait_any 39 %,
    // profiles/r5c_pmc.md).  Now the same lane -> address map goes through LDS-DMA (buffer_load ... lds) two steps ahead; the epilogue reads
    // its own 16-byte slots back (conflict-free), no vector register is held across steps.
    constexpr int ROFF = (ROLE == 0 ? 0 : ROLE == 1 ? 1 : ROLE == 2 ? 3 : 5) * 4096;
    char* resring = lds + RING * SLAB + 1024;

    const int n_ = lane & 31, g = lane >> 5;
    const S16TileLane tln = s16_tile_lane<RT, WT>(n_, !(p.lo4 & 0x100));    // row-major tile lanes; lo4 bit 8: the conflict-free order of s16_tilemap.h (experiment)
    const int rl = tln.rl, xl = tln.xl;
    const int n_ct = p.cout / 32;
    const int ct = (int)((blockIdx.x >> 3) % n_ct);   // cout tiles side by side on one XCD (see convs16.hip)
    const int Di = p.D, Hi = p.H, Wi = p.W;
    const int Wpi = Wi + 2, Hpi = Hi + 2;
    const long i_rowB = (long)Wpi * 128, i_planeB = (long)Hpi * i_rowB, i_cbB = (long)(Di + 2) * i_planeB, i_nB = 2 * i_cbB;
    const int Wpo = 2 * Wi + 2, Hpo = 2 * Hi + 2;
    const long o_chunkB = (long)Wpo * 16, o_rowB = 8 * o_chunkB, o_planeB = (long)Hpo * o_rowB, o_cbB = (long)(2 * Di + 2) * o_planeB,
               o_nB = (long)(p.cout / 32) * o_cbB;

    // weights of this role's taps: [ct][k-step 4][tap 27][hi, lo][lane][8]
    f16x8 wh[NT][4], wl[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const char* wb = (const char*)p.w + (((long)(ct * 4 + kk) * 27 + TL.t[t].w) * 2) * 1024 + lane * 16;
            wh[t][kk] = *(const f16x8*)wb;
            wl[t][kk] = *(const f16x8*)(wb + 1024);
        }
    // BN scale / shift of the 16 couts a lane holds (register e <-> cout (e&3) + 8(e>>2) + 4g), via LDS (read back per epilogue)
    if (wave == 0 && lane < 32) {
        const int gg = lane >> 4, e = lane & 15;
        const int co = ct * 32 + (e & 3) + 8 * (e >> 2) + 4 * gg;
        bnlds[gg * 32 + e] = p.scale[co];
        bnlds[gg * 32 + 16 + e] = p.shift[co];
    }
    S16Ovf og;                                        // range guard (s16_ovf.h)
    if (wave == 0) {                                  // a NaN / Inf folded BN parameter (lanes 0..31 cover the tile's 32 couts)
        const int co = ct * 32 + (lane & 31);
        og.see_raw(p.scale[co], 3.0e38f);
        og.see_raw(p.shift[co], 3.0e38f);
    }
    unsigned srcoff;
    {
        const bool ok = lane < (RT + 1) * SXI;
        const int rr = ok ? lane / SXI : 0;
        const int xx = ok ? lane - rr * SXI : 0;
        srcoff = (unsigned)(rr * i_rowB + xx * 16);
    }
    const unsigned bbase = (unsigned)(g * CPB + (rl * SXI + xl) * 16);           // chunk plane (kk>>1)*8 + (kk&1)*2 + g; lo at + 4 planes
    const __attribute__((address_space(3))) char* ringl = (const __attribute__((address_space(3))) char*)ring;
    const __attribute__((address_space(3))) float* bnl = (const __attribute__((address_space(3))) float*)bnlds;
    typedef const __attribute__((address_space(3))) f16x8 lds_frag;
    typedef const __attribute__((address_space(3))) f32x4 lds_f4;
    const __attribute__((address_space(3))) char* resl = (const __attribute__((address_space(3))) char*)resring + ROFF + lane * 16;

    const int n_xt = (Wi + WT - 1) / WT, n_yt = (Hi + RT - 1) / RT;          // ragged last tiles: lanes outside the map are masked (lane_ok)
    const unsigned xcd = blockIdx.x & 7, qx = (blockIdx.x >> 3) / n_ct, per_xcd = (gridDim.x >> 3) / n_ct;
    const unsigned cols_unit = (unsigned)n_yt * n_xt;
    const float relu_lo = p.relu ? 0.f : -65504.f;

    for (unsigned it = 0;; ++it) {
        const unsigned j = it * per_xcd + qx;
        const unsigned nl = j / cols_unit;
        const unsigned n = nl * 8 + xcd;
        if (n >= (unsigned)p.N) break;
        const unsigned rem = j - nl * cols_unit;
        const int yb = (int)(rem / n_xt), xt = (int)(rem - (unsigned)yb * n_xt);
        const int y0 = yb * RT, x0 = xt * WT;                    // input tile origin

        // staged rows start at input row y0 = padded row y0 + 1, columns at padded column x0 + 1
        const char* xcol = (const char*)p.x + (long)n * i_nB + (long)(y0 + 1) * i_rowB + (long)(x0 + 1) * 16;
        auto stage = [&](int plane, int slot) __attribute__((always_inline)) {
            const int pl = plane < Di ? plane : Di - 1;
            char* dst = ring + slot * SLAB;
#pragma unroll
            for (int ci = 0; ci < NL; ++ci) {
                const int cc = ci * 4 + wave;
                const char* src = xcol + (long)(cc >> 3) * i_cbB + (long)(pl + 1) * i_planeB + (long)(cc & 7) * (Wpi * 16);
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + srcoff), LDS_PTR(dst + cc * CPB), 16, 0, 0);
            }
        };
        const __amdgpu_buffer_rsrc_t y16r = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)p.y16 + (long)n * o_nB + (long)ct * o_cbB), 0, 0x7FFFFF00, 0x00020000);
        // (no residual, dropped lanes: the DMA reads the output tensor's own zero halo -- plane 0 -- so every request is a valid 16-byte read of zeros)
        const bool has_res = p.res != nullptr;
        const __amdgpu_buffer_rsrc_t resr = __builtin_amdgcn_make_buffer_rsrc(
            has_res ? (void*)((const char*)p.res + (long)n * o_nB + (long)ct * o_cbB) : (void*)((char*)p.y16 + (long)n * o_nB + (long)ct * o_cbB), 0, 0x7FFFFF00, 0x00020000);
        const int yl = y0 + rl;
        const bool lane_ok = tln.ok && yl < Hi && x0 + xl < Wi;
        // this lane's even-corner output voxel (2 yl, 2 (x0 + xl)), chunk (s = 0, g), hi, in output plane 0 (padded + 1)
        const unsigned o_lane = (unsigned)(o_planeB + (long)(2 * yl + 1) * o_rowB + (long)g * o_chunkB + (long)(2 * (x0 + xl) + 1) * 16);

        f32x16 acc[3][2];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

        // residual tiles are requested TWO STEPS before their use, into slot (step mod RD) of the residual ring: output offsets of the planes a
        // step at input plane zi finishes -- pz = 0 classes: 2 zi; pz = 1 classes: 2 zi - 1 -- and the 4 NF LDS-DMA loads
        auto offsets_of = [&](int zi, bool compute, unsigned (&fo)[NF]) __attribute__((always_inline)) {
#pragma unroll
            for (int ci = 0; ci < NF; ++ci) {
                const int c = R.cls[ci], pz = c >> 2, py = (c >> 1) & 1, px = c & 1;
                const int zo = pz ? 2 * zi - 1 : 2 * zi;
                const bool ok = lane_ok && zo >= 0 && zo < 2 * Di && (compute || pz);
                fo[ci] = ok ? o_lane + (unsigned)((long)zo * o_planeB + (long)py * o_rowB + px * 16) : 0x80000000u;
            }
        };
        auto request = [&](int zi, int rslot) __attribute__((always_inline)) {        // the residual tiles of the step at input plane zi -> ring slot rslot
            unsigned fo[NF];
            offsets_of(zi, zi < Di, fo);
            char* dst = resring + rslot * RESSLOT + ROFF;
#pragma unroll
            for (int ci = 0; ci < NF; ++ci)
#pragma unroll
                for (int q = 0; q < 4; ++q) {    // q: (lo?, s): chunks s*2 (+g in o_lane), + 4 for lo; a dropped lane reads the zero halo (offset 0)
                    const unsigned vo = (has_res && fo[ci] != 0x80000000u ? fo[ci] : 0u) + (unsigned)(((q >> 1) * 4 + (q & 1) * 2) * o_chunkB);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(resr, LDS_PTR(dst + (ci * 4 + q) * 1024), 16, vo, 0, 0, 0);
                }
        };
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(S16_WAITCNT(63, 0));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        request(0, 0);
        stage(0, 0);
        request(1, 1);
        stage(1, 1);
        __builtin_amdgcn_s_waitcnt(S16_WAITCNT(NS + NL, 15));              // plane 0 and its residual tiles landed (behind them: the requests of plane 1)
        int slot = 0;                                                        // ring slot of the current step (slabs and residual tiles rotate together)

        // one input plane zi (COMPUTE) of parity P; COMPUTE = false: the drain step (closes the last odd output plane)
        auto step = [&](int zi, auto PT, auto COMPT) __attribute__((always_inline)) {
            constexpr int P = decltype(PT)::value;
            constexpr bool COMPUTE = decltype(COMPT)::value;
            // slab zi and the residual tiles of this step landed: both were requested at the head of step zi - 2 (vmcnt retires in order);
            // younger are that step's stores, the previous step's requests, DMAs and stores: 3 NS + NL -- waited down to 2 NS + NL, which also
            // holds for step 1 of a column (behind its requests: step 0's requests, DMAs and stores)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(S16_WAITCNT(2 * NS + NL, 0));
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // output planes finished in this step; the requests of step zi + 2 go out now, into the slot step zi - 1 used (free since the barrier:
            // every wave is past that step's LDS reads)
            unsigned fo[NF];
            offsets_of(zi, COMPUTE, fo);
            {
                int ns = slot + RING - 1; ns = ns >= RING ? ns - RING : ns;
                request(zi + 2, ns);
                stage(zi + RING - 1, ns);
            }
            const __attribute__((address_space(3))) char* sb = ringl + slot * SLAB + bbase;
            auto run_tap = [&](int t) __attribute__((always_inline)) {
                constexpr int dummy = 0; (void)dummy;
                const UTap u = TL.t[t];
                const int pz = R.cls[u.ci] >> 2;
                const int ai = pz ? (u.open ? (P ^ 1) : P) : 0;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int off = ((kk >> 1) * 8 + (kk & 1) * 2) * CPB + (u.sy * SXI + u.sx) * 16;
                    const f16x8 h_ = *(lds_frag*)(sb + off), l_ = *(lds_frag*)(sb + off + 4 * CPB);
                    acc[u.ci][ai] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t][kk], h_, acc[u.ci][ai], 0, 0, 0);
                    acc[u.ci][ai] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t][kk], l_, acc[u.ci][ai], 0, 0, 0);
                    acc[u.ci][ai] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[t][kk], h_, acc[u.ci][ai], 0, 0, 0);
                }
            };
            if constexpr (COMPUTE) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (!TL.t[t].open) run_tap(t);
            }
            __builtin_amdgcn_sched_barrier(0);
            const __attribute__((address_space(3))) char* rs_ = resl + slot * RESSLOT;       // this step's residual tiles (landed: the wait at the head of the step)
            // ---- epilogues of the finished accumulators (in the shadow of the opening taps below)
#pragma unroll
            for (int ci = 0; ci < NF; ++ci) {
                const int pz = R.cls[ci] >> 2;
                const int ai = pz ? P : 0;
                const f32x16 a = acc[ci][ai];
                const unsigned long long og_keep = S16Ovf::lanes(fo[ci] != 0x80000000u);      // dropped lanes / planes hold over-read data
                f16x8 hi[2], lo[2];
                float og_mx = 0.f;                                   // largest |stored value| of this accumulator (one compare per accumulator)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const f32x4 sc0 = *(lds_f4*)(bnl + g * 32 + s * 8), sc1 = *(lds_f4*)(bnl + g * 32 + s * 8 + 4);
                    const f32x4 sh0 = *(lds_f4*)(bnl + g * 32 + 16 + s * 8), sh1 = *(lds_f4*)(bnl + g * 32 + 16 + s * 8 + 4);
                    const f16x8 rh = *(lds_frag*)(rs_ + (ci * 4 + s) * 1024), rl_ = *(lds_frag*)(rs_ + (ci * 4 + 2 + s) * 1024);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float sc_ = e < 4 ? sc0[e & 3] : sc1[e & 3], sh_ = e < 4 ? sh0[e & 3] : sh1[e & 3];
                        float x_ = a[s * 8 + e] * sc_ + sh_;
                        x_ += (float)rh[e] + (float)rl_[e];
                        x_ = __builtin_amdgcn_fmed3f(x_, relu_lo, 65504.f);
                        og_mx = fmaxf(og_mx, __builtin_fabsf(x_));
                        hi[s][e] = (_Float16)x_;
                        lo[s][e] = (_Float16)(x_ - (float)hi[s][e]);
                    }
                }
                og.see_max(og_mx, og_keep);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi[s]), y16r, fo[ci] + (unsigned)((s * 2) * o_chunkB), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo[s]), y16r, fo[ci] + (unsigned)((4 + s * 2) * o_chunkB), 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ci][ai][e] = 0.f;
            }
            if constexpr (COMPUTE) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (TL.t[t].open) run_tap(t);
            }
            slot = slot + 1 == RING ? 0 : slot + 1;
        };
        using F = std::false_type;
        using T = std::true_type;
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        int zi = 0;
#pragma unroll 1
        for (; zi + 1 < Di; zi += 2) {
            step(zi, I0{}, T{});
            step(zi + 1, I1{}, T{});
        }
        if (zi < Di) {
            step(zi, I0{}, T{});
            step(zi + 1, I1{}, F{});
        } else {
            step(zi, I0{}, F{});
        }
    }
    og.flush(p.ovf);
}

template <int RT, int WT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void convs16u_kernel(const drc_s16conv_params p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wave == 0) body<0, RT, WT>(p, lds, wave, lane);
    else if (wave == 1) body<1, RT, WT>(p, lds, wave, lane);
    else if (wave == 2) body<2, RT, WT>(p, lds, wave, lane);
    else body<3, RT, WT>(p, lds, wave, lane);
}

template <int RT, int WT>
int launch(const drc_s16conv_params& p, hipStream_t stream) {
    constexpr size_t lds = (size_t)RING * SLAB + 1024 + (size_t)RD * RESSLOT;
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)convs16u_kernel<RT, WT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const long columns = (long)p.N * ((p.H + RT - 1) / RT) * ((p.W + WT - 1) / WT);
    const int n_ct = p.cout / 32;
    long blocks = 256;                                   // column workers x cout tiles (the tiles of a worker side by side on its XCD)
    while (blocks > 8 * n_ct && blocks / (2 * n_ct) >= columns) blocks /= 2;
    hipLaunchKernelGGL((convs16u_kernel<RT, WT>), dim3((unsigned)blocks), dim3(256), lds, stream, p);
    return (int)hipGetLastError();
}

}  // namespace

// D, H, W = the INPUT dims; the output is (2D, 2H, 2W); cin = 64
extern "C" int drc_deconv3d_k3s2_s16_supported(int cin, int cout, int D, int H, int W) {
    if (cin != 64 || (cout != 32 && cout != 64)) return 0;
    if (D <= 0 || H <= 0 || W <= 0) return 0;
    return 1;       // round 6: any input dims (W <= 7: 4 x 7 input tiles, <= 14: 2 x 14, else 1 x 28; the last tiles masked)
}

extern "C" int drc_deconv3d_k3s2_s16_fwd(const drc_s16conv_params* pp, void* stream) {
    if (!pp) return -1;
    const drc_s16conv_params& p = *pp;
    if (!p.x || !p.w || !p.scale || !p.shift || !p.y16) return -1;
    if (p.y32 || p.left || p.right) return -4;
    if (p.N < 0) return -2;
    if (!drc_deconv3d_k3s2_s16_supported(p.cin, p.cout, p.D, p.H, p.W)) return -4;
    if (p.N == 0) return 0;
    const long unit_out = (long)(2 * p.D + 2) * (2 * p.H + 2) * (2 * p.W + 2) * 128;      // one 32-channel block
    if (unit_out >= 0x7FFFFF00L / 2) return -5;
    hipStream_t s = (hipStream_t)stream;
    if (p.W <= 7) return launch<4, 7>(p, s);
    if (p.W <= 14) return launch<2, 14>(p, s);
    return launch<1, 28>(p, s);
}
