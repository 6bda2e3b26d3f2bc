// deconvdirect.hip -- ConvTranspose3d(k3, s2, p1, output_padding 1) (+BN, +residual, +ReLU): all 27 taps of all 8 output-parity
// classes from ONE set of B fragments held in registers, both MFMA operands straight from global memory (gfx950 / CDNA4).
//
//   reference: hourglass conv5 / conv6, stackhourglass.py:22-30,44-49
//
// o = 2i - 1 + k: an even output (o = 2i) has one tap per dimension (k = 1), an odd one (o = 2i + 1) two (k = 2 at input i,
// k = 0 at input i + 1).  Per dimension that is three (parity, input shift, k) combinations -- (0,0,1), (1,0,2), (1,1,0) -- and
// 27 in 3D: exactly the 27 taps, each used once, i.e. 27 MACs per (cin, cout) pair and INPUT voxel and no multiply wasted.
// tapdeconv.hip stages the input tile through LDS per 8-channel phase and issues 16 / 8 MFMAs per tap step (62 TFLOP/s, 63 %
// LDS bank conflicts, bound by memory-instruction issue).  Here a wave owns VT*16 consecutive input voxels and CT*16 couts:
//   * the eight shifted B fragments (input shifts {0,1}^3) of a 16-channel block are loaded ONCE -- 8*VT coalesced float4 per
//     lane -- one channel block ahead (second register set), and stay in registers for all 27 taps;
//   * a tap streams its weights (CT float4 per lane from [cb][27 combinations in use order][cout][16], DD_AHEAD taps ahead) against them: 4*VT*CT MFMAs into
//     the accumulators of its parity class (8 classes x VT x CT tiles);
//   * 8*VT + 27*CT loads per 108*VT*CT MFMAs (70 per 432 at VT = 2, CT = 2): the vector-memory path idles, nothing touches LDS.
// Epilogue per class: BN scale/shift, residual, ReLU, one float4 store per (voxel, cout tile) at output (2i + parity); in the last
// channel block the taps run class by class and a finished class's epilogue overlaps the next class's MFMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/disprcnn_hip.h"
#include "s16_ovf.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define DD_WAVES 4
#ifndef DD_AHEAD
#define DD_AHEAD 6          // weights this many taps (16*VT*CT/4 MFMAs each) ahead of their use
#endif

namespace {

// the 27 (class, shift, tap) combinations, class-major so that consecutive taps reuse accumulators late (dependent MFMAs on one
// accumulator are VT*CT*... issues apart anyway); per dimension: parity p, shift s, kernel index k
struct Combo { int cls, pos, tap; bool last_of_class; };
constexpr int kP[3] = {0, 1, 1}, kS[3] = {0, 0, 1}, kK[3] = {1, 2, 0};
constexpr Combo combo_raw(int i) {
    const int a = i / 9, b = (i / 3) % 3, c = i % 3;
    return Combo{(kP[a] * 2 + kP[b]) * 2 + kP[c], (kS[a] * 2 + kS[b]) * 2 + kS[c], (kK[a] * 3 + kK[b]) * 3 + kK[c], false};
}
// use order: class-major (class 7 = odd/odd/odd with 8 taps first ... class 0 with one tap last), so that in the last channel
// block a class is complete -- and its epilogue can run in the shadow of the next classes' MFMAs -- as early as possible
struct ComboTable { Combo c[27]; };
constexpr ComboTable make_table() {
    ComboTable t{};
    int i = 0;
    for (int cls = 7; cls >= 0; --cls) {
        int first = i;
        for (int r = 0; r < 27; ++r)
            if (combo_raw(r).cls == cls) t.c[i++] = combo_raw(r);
        t.c[i - 1].last_of_class = true;
        (void)first;
    }
    return t;
}
constexpr ComboTable kTab = make_table();

// (Measured and rejected: one cout tile per wave, 292 registers: 83-91 TFLOP/s against 91-102 for two; the same forced to two
// waves per SIMD with amdgpu_waves_per_eu(2,2), 29 spills: 79-84.  Occupancy is not what this kernel lacks.)
// S16 (round 5): the result (also) goes out as an RS16 tensor (convs16.hip's input layout): lane (voxel j, g) of cout tile t holds couts
// 16t + 4g .. +3 = half a chunk: chunk (s = t & 1, g' = g & 1) of the 32-channel block t >> 1, bytes (g >> 1)*8 .. -- 8 B hi + 8 B lo.
template <int VT, int CT, bool S16 = false>
__global__ __launch_bounds__(64 * DD_WAVES) void deconvdirect_kernel(const drc_tapconv_params p, char* y16, uint32_t* ovf) {
    S16Ovf og;                                        // range guard of the RS16 epilogue (s16_ovf.h)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const int g = lane >> 4;

    // logical grid = the INPUT grid (OD, OH, OW of the params are the input dims; the output is exactly twice as large).  A tile is
    // VT*16 CONSECUTIVE voxels of the flattened (n, d, h, w) index -- every lane carries its own voxel offset, the eight input
    // shifts and the eight output parities are uniform offsets on top -- so only the very last tile of a launch is ragged
    // (per-slice 2x14 tiles wasted 12.5 % of conv6's MFMA slots, 4x7 tiles 24 % of conv5's).
    const long voxels = (long)p.N * p.OD * p.OH * p.OW;
    const long tiles = (voxels + VT * 16 - 1) / (VT * 16);
    const int n_cg = p.cout_pad / 16 / CT;
    const long items = tiles * n_cg;
    const unsigned w_tap_b = (unsigned)p.cout_pad * 64;    // bytes per (cb, combination): weights are packed [cb][27 combinations][cout][16]
    const long workers = (long)gridDim.x * DD_WAVES;
    const long wid = (long)blockIdx.x * DD_WAVES + wave;

#pragma unroll 1
    for (long it = wid; it < items; it += workers) {
        const int cg = (int)(it % n_cg);
        const long tile = it / n_cg;
        const int ct0 = cg * CT;

        // per-lane byte offsets of input voxel slot (vt, j) at shift (0,0,0) (padded coordinates = logical + 1) and of its even-corner
        // output / residual voxel
        unsigned lane_vo[VT], yv[VT], rv[VT], y16v[S16 ? VT : 1];
        bool valid[VT];
        // RS16 output geometry (bytes): halfs [N][cout/32][2OD+2][2OH+2][8][2OW+2][8]
        const long s_chunkB = (long)(2 * p.OW + 2) * 16, s_rowB = 8 * s_chunkB, s_planeB = (long)(2 * p.OH + 2) * s_rowB,
                   s_cbB = (long)(2 * p.OD + 2) * s_planeB, s_nB = (long)(p.cout_pad / 32) * s_cbB;
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) {
            long q = tile * (VT * 16) + vt * 16 + j;
            valid[vt] = q < voxels;
            if (!valid[vt]) q = 0;
            const int c = (int)(q % p.OW); q /= p.OW;
            const int r = (int)(q % p.OH); q /= p.OH;
            const int id = (int)(q % p.OD);
            const int n = (int)(q / p.OD);
            lane_vo[vt] = (unsigned)((n * p.x_n_stride + (int64_t)(id + 1) * p.x_d_stride + (int64_t)(r + 1) * p.x_h_stride + (c + 1) * 16 + g * 4) * 4);
            yv[vt] = (unsigned)((n * p.y_n_stride + (int64_t)(2 * id) * p.y_d_stride + (int64_t)(2 * r) * p.y_h_stride + 2 * c * 16 + g * 4) * 4);
            rv[vt] = (unsigned)((n * p.r_n_stride + (int64_t)(2 * id) * p.r_d_stride + (int64_t)(2 * r) * p.r_h_stride + 2 * c * 16 + g * 4) * 4);
            if constexpr (S16)
                y16v[vt] = (unsigned)(n * s_nB + (long)(2 * id + 1) * s_planeB + (long)(2 * r + 1) * s_rowB + (long)(g & 1) * s_chunkB + (2 * c + 1) * 16 + (g >> 1) * 8);
        }
        const char* xs = (const char*)p.x;
        const unsigned wlane = (unsigned)(((ct0 * 16 + j) * 16 + g * 4) * 4);      // this lane's byte offset inside a combination's weights

        f32x4 acc[8][VT][CT];
        {
            float z_;
            asm volatile("v_mov_b32 %0, 0" : "=v"(z_));       // a literal zero vector per tile gets hoisted and parked
            const f32x4 z4 = {z_, z_, z_, z_};
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) acc[c][vt][ct] = z4;
        }
        f32x4 bA[8][VT], bB[8][VT];
        auto load_b = [&](f32x4 (&B)[8][VT], int cb) __attribute__((always_inline)) {
            const char* sb = xs + (int64_t)cb * p.x_cb_stride * 4;
#pragma unroll
            for (int pos = 0; pos < 8; ++pos) {
                const char* sp = sb + ((int64_t)(pos >> 2) * p.x_d_stride + (int64_t)((pos >> 1) & 1) * p.x_h_stride + (pos & 1) * 16) * 4;
#pragma unroll
                for (int vt = 0; vt < VT; ++vt) B[pos][vt] = *(const f32x4*)(sp + lane_vo[vt]);
            }
        };
        f32x4 bn_sc[CT], bn_sh[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            bn_sc[ct] = *(const f32x4*)(p.scale + (ct0 + ct) * 16 + g * 4);
            bn_sh[ct] = *(const f32x4*)(p.shift + (ct0 + ct) * 16 + g * 4);
        }
        // class (pd, ph, pw) -> output voxel (2*id + pd, 2*r + ph, 2*c + pw): a uniform 64-bit base per (class, cout tile) plus the
        // lane's 32-bit even-corner offset (no per-store 64-bit lane arithmetic to hoist and spill)
        auto y_base = [&](int c, int ct) __attribute__((always_inline)) {
            return (char*)(p.y + p.y_off0 + (int64_t)(ct0 + ct) * p.y_cb_stride + (int64_t)(c >> 2) * p.y_d_stride + (int64_t)((c >> 1) & 1) * p.y_h_stride +
                           (c & 1) * 16);
        };
        auto r_base = [&](int c, int ct) __attribute__((always_inline)) {
            return (const char*)(p.res + p.r_off0 + (int64_t)(ct0 + ct) * p.r_cb_stride + (int64_t)(c >> 2) * p.r_d_stride + (int64_t)((c >> 1) & 1) * p.r_h_stride +
                                 (c & 1) * 16);
        };
        f32x4 resq[VT][CT];
        // all 27 taps of one channel block against the B fragments in registers; weights DD_AHEAD taps ahead.  LAST (the final
        // channel block): the residual of a class is requested when its first tap starts and its epilogue (BN, residual, ReLU,
        // stores) follows its last tap, i.e. runs in the shadow of the next class's MFMAs.
        auto block = [&](const f32x4 (&B)[8][VT], int cb, auto last_tag) __attribute__((always_inline)) {
            constexpr bool LAST = decltype(last_tag)::value;
            const char* wb = (const char*)p.w + (size_t)cb * 27u * w_tap_b;            // uniform: SGPR base + lane offset + immediate
            f32x4 wq[DD_AHEAD + 1][CT];
#pragma unroll
            for (int a = 0; a < DD_AHEAD; ++a)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) wq[a][ct] = *(const f32x4*)(wb + (unsigned)a * w_tap_b + ct * 1024 + wlane);
#pragma unroll
            for (int i = 0; i < 27; ++i) {
                constexpr int dummy_ = 0; (void)dummy_;
                const Combo q = kTab.c[i];
                if (i + DD_AHEAD < 27) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        wq[(i + DD_AHEAD) % (DD_AHEAD + 1)][ct] = *(const f32x4*)(wb + (unsigned)(i + DD_AHEAD) * w_tap_b + ct * 1024 + wlane);
                }
                if (LAST && p.res && (i == 0 || kTab.c[i > 0 ? i - 1 : 0].last_of_class)) {
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            resq[vt][ct] = *(const f32x4*)(r_base(q.cls, ct) + rv[vt]);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            acc[q.cls][vt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[i % (DD_AHEAD + 1)][ct][s], B[q.pos][vt][s], acc[q.cls][vt][ct], 0, 0, 0);
                // keep the loads where they are written: without the fence the scheduler sinks every weight load to its use (three
                // weight registers in total, a full L2 round trip exposed per tap: 598 us instead of ~410 for conv6 at 256 ROIs)
                __builtin_amdgcn_sched_barrier(0);
                if (LAST && q.last_of_class) {
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt) {
                        if (!valid[vt]) continue;
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) {
                            f32x4 v_ = acc[q.cls][vt][ct] * bn_sc[ct] + bn_sh[ct];
                            if (p.res) v_ += resq[vt][ct];
                            if (p.relu) { v_.x = fmaxf(v_.x, 0.f); v_.y = fmaxf(v_.y, 0.f); v_.z = fmaxf(v_.z, 0.f); v_.w = fmaxf(v_.w, 0.f); }
                            if (!S16 || p.y) *(f32x4*)(y_base(q.cls, ct) + yv[vt]) = v_;
                            if constexpr (S16) {
                                f16x4 hi, lo;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    og.see_raw(v_[e]);
                                    const float x_ = fminf(fmaxf(v_[e], -65504.f), 65504.f);
                                    hi[e] = (_Float16)x_;
                                    lo[e] = (_Float16)(x_ - (float)hi[e]);
                                }
                                const int tile = ct0 + ct;
                                char* b16 = y16 + (long)(tile >> 1) * s_cbB + (long)((tile & 1) * 2) * s_chunkB + (long)(q.cls >> 2) * s_planeB +
                                            (long)((q.cls >> 1) & 1) * s_rowB + (q.cls & 1) * 16;
                                *(f16x4*)(b16 + y16v[vt]) = hi;
                                *(f16x4*)(b16 + 4 * s_chunkB + y16v[vt]) = lo;
                            }
                        }
                    }
                }
            }
        };
        // straight-line control flow (accumulators that merge from several paths get shuttled through VGPRs and spill): every block
        // but the last runs out of set A while set B receives the next block, then B is copied to A (64 moves per 432 MFMAs)
        load_b(bA, 0);
#pragma unroll 1
        for (int cb = 0; cb + 1 < p.cb_in; ++cb) {
            load_b(bB, cb + 1);
            block(bA, cb, std::false_type{});
#pragma unroll
            for (int pos = 0; pos < 8; ++pos)
#pragma unroll
                for (int vt = 0; vt < VT; ++vt) bA[pos][vt] = bB[pos][vt];
        }
        block(bA, p.cb_in - 1, std::true_type{});
    }
    if constexpr (S16) og.flush(ovf);
}

template <int VT, int CT, bool S16 = false>
int launch(const drc_tapconv_params& p, hipStream_t stream, char* y16 = nullptr, uint32_t* ovf = nullptr) {
    const long voxels = (long)p.N * p.OD * p.OH * p.OW;
    const long items = ((voxels + VT * 16 - 1) / (VT * 16)) * (p.cout_pad / 16 / CT);
    long workers = 256L * DD_WAVES;                      // one wave per SIMD (the two B sets + 8 accumulator classes fill the file)
    if (workers > items) workers = items;
    if (workers < 1) workers = 1;
    dim3 grid((unsigned)((workers + DD_WAVES - 1) / DD_WAVES), 1, 1);
    hipLaunchKernelGGL((deconvdirect_kernel<VT, CT, S16>), grid, dim3(64 * DD_WAVES), 0, stream, p, y16, ovf);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int drc_deconv3d_k3s2_direct_fwd(const drc_tapconv_params* pp, int cout_tiles_per_wave, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD <= 0 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (p.cout_pad <= 0 || (p.cout_pad & 15) || p.cb_in <= 0) return -2;
    if (p.n_classes != 8 || p.in_mul != 1 || p.out_mul != 2) return -4;
    if ((int64_t)p.N * p.x_n_stride * 4 >= (1LL << 32) || (int64_t)p.N * p.y_n_stride * 4 >= (1LL << 32) ||
        (p.res && (int64_t)p.N * p.r_n_stride * 4 >= (1LL << 32)))
        return -5;                                                             // 32-bit lane offsets over the whole batch
    const int ct = p.cout_pad / 16, CT = cout_tiles_per_wave;
    hipStream_t s = (hipStream_t)stream;
    if (CT == 2 && ct % 2 == 0) return launch<2, 2>(p, s);
    if (CT == 1) return launch<2, 1>(p, s);
    return -2;
}

extern "C" int drc_deconv3d_k3s2_direct_s16_fwd(const drc_tapconv_params* pp, void* y16, uint32_t* ovf, void* stream) {
    if (!pp || !y16) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD <= 0 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (p.cout_pad <= 0 || (p.cout_pad & 31) || p.cb_in <= 0) return -2;
    if (p.n_classes != 8 || p.in_mul != 1 || p.out_mul != 2) return -4;
    const int64_t unit16 = (int64_t)(p.cout_pad / 32) * (2 * p.OD + 2) * (2 * p.OH + 2) * (2 * p.OW + 2) * 128;
    if ((int64_t)p.N * p.x_n_stride * 4 >= (1LL << 32) || (p.y && (int64_t)p.N * p.y_n_stride * 4 >= (1LL << 32)) ||
        (p.res && (int64_t)p.N * p.r_n_stride * 4 >= (1LL << 32)) || (int64_t)p.N * unit16 >= (1LL << 32))
        return -5;
    return launch<2, 2, true>(p, (hipStream_t)stream, (char*)y16, ovf);
}
