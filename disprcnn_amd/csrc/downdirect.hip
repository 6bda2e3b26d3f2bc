// downdirect.hip -- Conv3d(k3, stride 2, pad 1) and Conv2d(k3, stride 1|2, dilation d) (+BN/bias, +residual, +ReLU) with both
// MFMA operands read straight from global memory (no LDS) (gfx950 / CDNA4).
//
//   reference: hourglass.conv1 / conv3 (stackhourglass.py:9-16) and the data gradient of the ConvTranspose3d layers; the 3x3
//   convolutions of the PSMNet feature CNN (submodule.py:60-139) and of ResNet-50-FPN (backbone/resnet.py, backbone/fpn.py).
//
// Operand scheme of tapdirect.hip: in the blocked layout the B fragment of tap (kd, kh, kw) is one float4 per lane -- lane
// (output voxel j, g) reads channels 4g..4g+3 of input voxel (2od+kd, 2oh+kh, 2ow+kw) -- and covers four MFMA k-steps; the
// stride only changes the per-lane address (every other 64-byte line of a row), which costs nothing here, whereas the LDS
// variants had to de-interleave parity planes (tapdown.hip) or eat 4-way bank conflicts (tapconv.hip).  A wave owns R x WT
// output voxels of one output slice and CT*16 output channels (all 64 channels of the hourglass layers in one wave) and walks
// (channel block, kd, tap) with the next step's VT + CT loads in flight: 11 loads per 112 MFMAs at VT = 7, CT = 4.
// The 2D instantiation (DIM3 = false) drops the depth taps and takes stride and dilation from the parameter block.
// Weights: [27 | 9][cb_in][cout_pad][16] (engine.pack_weight_t16).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DW_WAVES 4

namespace {

template <int VT, int CT, bool DIM3>
__global__ __launch_bounds__(64 * DW_WAVES) void downdirect_kernel(const drc_tapconv_params p) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const int g = lane >> 4;

    const int OD = p.OD, OH = p.OH, OW = p.OW;
    const int n_wt = (OW + p.WT - 1) / p.WT;
    const int n_rt = (OH + p.R - 1) / p.R;
    const int per_cg = p.N * OD * n_rt * n_wt;       // groups of one cout group: (n, od, row tile, col tile)
    const long groups = (long)(p.cout_pad / 16 / CT) * per_cg;
    const long workers = (long)gridDim.x * DW_WAVES;
    // blocks are numbered XCD by XCD (block b runs on XCD b % 8): an XCD then walks one contiguous share of the cout-group-major group
    // list, i.e. 1/8 of the weights -- with consecutive shares spread round-robin every XCD's L2 pulled ALL weights (the trunk's
    // 512 -> 512 layers on 12 x 39: 95 MB fetched per launch for 9.4 MB of weights, round 3)
    const long bid = (gridDim.x & 7) == 0 ? (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : (long)blockIdx.x;
    const long wid = bid * DW_WAVES + wave;
    long gcur = groups * wid / workers;              // equal contiguous shares
    const long gend = groups * (wid + 1) / workers;
    if (gcur >= gend) return;
    const int nslots = p.R * p.WT;

    // per-lane byte offset of the input voxel of output slot (vt, j) at tap (0,0,0), channels 4g..4g+3.  Slots of a ragged tile
    // that fall outside the output grid are clamped onto its last valid row / column (their results are never stored).
    int vr[VT], vc[VT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
        const int s = vt * 16 + j;
        const int r = s / p.WT, c = s - r * p.WT;
        vr[vt] = s < nslots ? r : -1; vc[vt] = c;
    }
    const int64_t w_cb = (int64_t)p.cout_pad * 16, w_tap = w_cb * p.cb_in;
    constexpr int NT = DIM3 ? 27 : 9;                // taps per channel block
    const int steps = p.cb_in * NT;                  // per group: (cb outer, kd, kh, kw)
    const int str = p.in_mul, dil = p.cls[0].sh;     // stride; tap spacing (dilation), same in h and w

    struct Group { int n, od, oh0, ow0, ct0, nr, nc; const float* base; };
    auto decode = [&](long gidx) __attribute__((always_inline)) -> Group {
        Group q;
        int r = (int)(gidx % per_cg);
        q.ct0 = (int)(gidx / per_cg) * CT;
        const int wt = r % n_wt; r /= n_wt;
        const int rt = r % n_rt; r /= n_rt;
        q.od = r % OD; q.n = r / OD;
        q.oh0 = rt * p.R; q.ow0 = wt * p.WT;
        q.nr = OH - q.oh0 < p.R ? OH - q.oh0 : p.R;
        q.nc = OW - q.ow0 < p.WT ? OW - q.ow0 : p.WT;
        q.base = p.x + (int64_t)q.n * p.x_n_stride + (int64_t)(str * q.od + p.cls[0].dd0) * p.x_d_stride +
                 (int64_t)(str * q.oh0 + p.cls[0].dh0) * p.x_h_stride + (int64_t)(str * q.ow0 + p.cls[0].dw0) * 16;
        return q;
    };
    unsigned lane_vo[VT];
    auto set_lane_vo = [&](const Group& G) __attribute__((always_inline)) {
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) {
            int r = vr[vt] < 0 ? 0 : vr[vt], c = vc[vt];
            r = r < G.nr ? r : G.nr - 1;
            c = c < G.nc ? c : G.nc - 1;
            lane_vo[vt] = (unsigned)((str * r * (int)p.x_h_stride + str * c * 16 + g * 4) * 4);
        }
    };
    // operands of step s (cb = s / NT, tap = s % NT) of group G
    auto load_step = [&](f32x4 (&B)[VT], f32x4 (&Wt)[CT], const Group& G, const unsigned (&vo)[VT], int s) __attribute__((always_inline)) {
        const int cb = s / NT, t = s - cb * NT;
        const int kd = DIM3 ? t / 9 : 0, kh = (t - kd * 9) / 3, kw = t - kd * 9 - kh * 3;
        const char* sb = (const char*)(G.base + (int64_t)cb * p.x_cb_stride + (int64_t)kd * p.x_d_stride + (int64_t)(kh * dil) * p.x_h_stride +
                                       kw * dil * 16);
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) B[vt] = *(const f32x4*)(sb + vo[vt]);
        const char* wsb = (const char*)(p.w + (int64_t)t * w_tap + (int64_t)cb * w_cb);
        const unsigned wlo = (unsigned)(((G.ct0 * 16 + j) * 16 + g * 4) * 4);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) Wt[ct] = *(const f32x4*)(wsb + wlo + ct * 1024);
    };

#define DW_CLEAR_ACC()                                                                                 \
    {                                                                                                  \
        float z_;                                                                                      \
        asm volatile("v_mov_b32 %0, 0" : "=v"(z_));                                                    \
        const f32x4 z4_ = {z_, z_, z_, z_};                                                            \
        _Pragma("unroll") for (int vt = 0; vt < VT; ++vt)                                              \
            _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) acc[vt][ct] = z4_;                       \
    }
    f32x4 acc[VT][CT];
    DW_CLEAR_ACC()

#define DW_MFMA(B, Wt)                                                                                 \
    _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4)                                                   \
        _Pragma("unroll") for (int vt = 0; vt < VT; ++vt)                                              \
            _Pragma("unroll") for (int ct = 0; ct < CT; ++ct)                                          \
                acc[vt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wt[ct][s4], B[vt][s4], acc[vt][ct], 0, 0, 0);

    // folded BN, residual, ReLU, store; clears the accumulators
    auto epilogue = [&](const Group& cur) __attribute__((always_inline)) {
            f32x4 bn_sc[CT], bn_sh[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                bn_sc[ct] = *(const f32x4*)(p.scale + (cur.ct0 + ct) * 16 + g * 4);
                bn_sh[ct] = *(const f32x4*)(p.shift + (cur.ct0 + ct) * 16 + g * 4);
            }
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) {
                const int r = vr[vt], c = vc[vt];
                if (r >= 0 && r < cur.nr && c < cur.nc) {
                    const int64_t yo = p.y_off0 + (int64_t)cur.n * p.y_n_stride + (int64_t)cur.od * p.y_d_stride +
                                       (int64_t)(cur.oh0 + r) * p.y_h_stride + (int64_t)(cur.ow0 + c) * 16 + g * 4;
                    const int64_t ro = p.r_off0 + (int64_t)cur.n * p.r_n_stride + (int64_t)cur.od * p.r_d_stride +
                                       (int64_t)(cur.oh0 + r) * p.r_h_stride + (int64_t)(cur.ow0 + c) * 16 + g * 4;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        f32x4 v = acc[vt][ct] * bn_sc[ct] + bn_sh[ct];
                        if (p.res) v += *(const f32x4*)(p.res + ro + (int64_t)(cur.ct0 + ct) * p.r_cb_stride);
                        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        *(f32x4*)(p.y + yo + (int64_t)(cur.ct0 + ct) * p.y_cb_stride) = v;
                    }
                }
            }
            DW_CLEAR_ACC()
    };

    // Small tiles (VT*CT <= 4: <= 16 MFMAs = 512 cycles a step; the trunk's 512-channel layers on 12 x 39 maps walk 288 such steps per
    // group): one step of prefetch left every step waiting out an L2 round trip (MFMA busy 26 %, 108 us for 4.4 GFLOP, round 3).  They
    // run a ring of PF + 1 operand sets, PF steps ahead, pinned in front of the MFMAs; no prefetch across groups (a group is long).
    if constexpr (VT * CT <= 4) {
        constexpr int PF = VT * CT <= 2 ? 6 : 4, NB = PF + 1;
        f32x4 rB[NB][VT], rW[NB][CT];
#pragma unroll 1
        for (; gcur < gend; ++gcur) {
            const Group cur = decode(gcur);
            set_lane_vo(cur);
            const int last = steps - 1;
#define DW_RING_PRO(U) if constexpr (U < PF) load_step(rB[U], rW[U], cur, lane_vo, U < last ? U : last);
            DW_RING_PRO(0) DW_RING_PRO(1) DW_RING_PRO(2) DW_RING_PRO(3) DW_RING_PRO(4) DW_RING_PRO(5)
#undef DW_RING_PRO
            int s = 0;
            // (explicit ring steps: left as a `#pragma unroll` loop the compiler kept the set index dynamic and put the ring in scratch)
#define DW_RING_STEP(U)                                                                                \
            if constexpr (U < NB) {                                                                    \
                const int sf = s + U + PF;                                                             \
                load_step(rB[(U + PF) % NB], rW[(U + PF) % NB], cur, lane_vo, sf < last ? sf : last);  \
                __builtin_amdgcn_sched_barrier(0);                                                     \
                DW_MFMA(rB[U], rW[U])                                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                     \
            }
            for (; s + NB <= steps; s += NB) {
                DW_RING_STEP(0) DW_RING_STEP(1) DW_RING_STEP(2) DW_RING_STEP(3) DW_RING_STEP(4) DW_RING_STEP(5) DW_RING_STEP(6)
            }
#undef DW_RING_STEP
            const int rem = steps - s;
#define DW_RING_TAIL(U)                                                                                \
            if constexpr (U < NB - 1) { if (U < rem) { DW_MFMA(rB[U], rW[U]) } }
            DW_RING_TAIL(0) DW_RING_TAIL(1) DW_RING_TAIL(2) DW_RING_TAIL(3) DW_RING_TAIL(4) DW_RING_TAIL(5)
#undef DW_RING_TAIL
            epilogue(cur);
        }
        return;
    }

    f32x4 bA[VT], bB[VT], wA[CT], wB[CT];
    Group cur = decode(gcur);
    set_lane_vo(cur);
    load_step(bA, wA, cur, lane_vo, 0);

#pragma unroll 1
    for (;;) {
        const bool has_next = gcur + 1 < gend;
        // steps come in pairs (A set, B set); an odd step count ends on a lone A step whose prefetch lands in B and is moved to A
        int s = 0;
        for (; s + 2 <= steps - 1; s += 2) {
            load_step(bB, wB, cur, lane_vo, s + 1);
            DW_MFMA(bA, wA)
            load_step(bA, wA, cur, lane_vo, s + 2);
            DW_MFMA(bB, wB)
        }
        // tail: 1 or 2 steps remain (s == steps-1, or s == steps-2); the last one prefetches the next group's step 0
        Group nxg = cur;
        unsigned nvo[VT];
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) nvo[vt] = lane_vo[vt];
        if (has_next) {
            nxg = decode(gcur + 1);
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) {
                int r = vr[vt] < 0 ? 0 : vr[vt], c = vc[vt];
                r = r < nxg.nr ? r : nxg.nr - 1;
                c = c < nxg.nc ? c : nxg.nc - 1;
                nvo[vt] = (unsigned)((str * r * (int)p.x_h_stride + str * c * 16 + g * 4) * 4);
            }
        }
        if (s == steps - 2) {
            load_step(bB, wB, cur, lane_vo, s + 1);
            DW_MFMA(bA, wA)
            load_step(bA, wA, nxg, nvo, 0);
            DW_MFMA(bB, wB)
        } else {
            load_step(bB, wB, nxg, nvo, 0);
            DW_MFMA(bA, wA)
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) bA[vt] = bB[vt];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) wA[ct] = wB[ct];
        }

        epilogue(cur);
        if (!has_next) break;
        ++gcur;
        cur = nxg;
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) lane_vo[vt] = nvo[vt];
    }
#undef DW_MFMA
#undef DW_CLEAR_ACC
}

template <int VT, int CT, bool DIM3>
int launch(const drc_tapconv_params& p, hipStream_t stream) {
    static int occ_blocks = 0;
    if (!occ_blocks) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, downdirect_kernel<VT, CT, DIM3>, 64 * DW_WAVES, 0) != hipSuccess || nb < 1) nb = 1;
        occ_blocks = nb;
    }
    const long groups = (long)(p.cout_pad / 16 / CT) * p.N * p.OD * ((p.OH + p.R - 1) / p.R) * ((p.OW + p.WT - 1) / p.WT);
    long workers = 256L * DW_WAVES * occ_blocks;
    if (workers > groups) workers = groups;
    dim3 grid((unsigned)((workers + DW_WAVES - 1) / DW_WAVES), 1, 1);
    hipLaunchKernelGGL((downdirect_kernel<VT, CT, DIM3>), grid, dim3(64 * DW_WAVES), 0, stream, p);
    return (int)hipGetLastError();
}

template <int CT, bool DIM3>
int launch_vt(int nvt, const drc_tapconv_params& p, hipStream_t s) {
    switch (nvt) {
        case 1: return launch<1, CT, DIM3>(p, s);
        case 2: return launch<2, CT, DIM3>(p, s);
        case 3: return launch<3, CT, DIM3>(p, s);
        case 4: return launch<4, CT, DIM3>(p, s);
        case 5: return launch<5, CT, DIM3>(p, s);
        case 6: return launch<6, CT, DIM3>(p, s);
        case 7: return launch<7, CT, DIM3>(p, s);
    }
    return -3;
}

}  // namespace

extern "C" int drc_conv3d_k3s2_direct_fwd(const drc_tapconv_params* pp, int cout_tiles_per_wave, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD <= 0 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (p.cout_pad <= 0 || (p.cout_pad & 15) || p.cb_in <= 0) return -2;
    const drc_tap_class& k = p.cls[0];
    if (p.n_classes != 1 || p.in_mul != 2 || p.out_mul != 1 || k.nd != 3 || k.nh != 3 || k.nw != 3 || k.sd != 1 || k.sh != 1 ||
        k.sw != 1 || k.wbase != 0 || k.wsd != 9 || k.wsh != 3 || k.wsw != 1)
        return -4;
    if (p.R <= 0 || p.WT <= 0 || p.R * p.WT > 112) return -3;
    if ((int64_t)(2 * p.R + 2) * p.x_h_stride * 4 >= (1LL << 31)) return -5;   // 32-bit lane offsets
    const int ct = p.cout_pad / 16, CT = cout_tiles_per_wave;
    if ((CT != 1 && CT != 2 && CT != 4) || ct % CT) return -2;
    const int nvt = (p.R * p.WT + 15) / 16;
    if (nvt * CT > 28) return -3;
    hipStream_t s = (hipStream_t)stream;
    return CT == 4 ? launch_vt<4, true>(nvt, p, s) : CT == 2 ? launch_vt<2, true>(nvt, p, s) : launch_vt<1, true>(nvt, p, s);
}

extern "C" int drc_conv2d_k3_direct_fwd(const drc_tapconv_params* pp, int cout_tiles_per_wave, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD != 1 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (p.cout_pad <= 0 || (p.cout_pad & 15) || p.cb_in <= 0) return -2;
    const drc_tap_class& k = p.cls[0];
    if (p.n_classes != 1 || (p.in_mul != 1 && p.in_mul != 2) || p.out_mul != 1 || k.nd != 1 || k.nh != 3 || k.nw != 3 || k.sh < 1 ||
        k.sh != k.sw || k.wbase != 0 || k.wsh != 3 || k.wsw != 1)
        return -4;
    if (p.R <= 0 || p.WT <= 0 || p.R * p.WT > 112) return -3;
    if ((int64_t)(p.in_mul * p.R + 2 * k.sh + 1) * p.x_h_stride * 4 >= (1LL << 31)) return -5;   // 32-bit lane offsets
    const int ct = p.cout_pad / 16, CT = cout_tiles_per_wave;
    if ((CT != 1 && CT != 2 && CT != 4) || ct % CT) return -2;
    const int nvt = (p.R * p.WT + 15) / 16;
    if (nvt * CT > 28) return -3;
    hipStream_t s = (hipStream_t)stream;
    return CT == 4 ? launch_vt<4, false>(nvt, p, s) : CT == 2 ? launch_vt<2, false>(nvt, p, s) : launch_vt<1, false>(nvt, p, s);
}
