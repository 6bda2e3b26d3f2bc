// tapconv.hip -- tap-grid convolution on channel-blocked, zero-haloed tensors (gfx950 / CDNA4).
//
// One engine for every dense convolution of the instance-disparity path:
//   Conv3d k3 s1/s2 + BN3d (+ReLU, +residual)         reference: submodule.py:19-22, stackhourglass.py:63-88
//   ConvTranspose3d k3 s2 p1 op1 + BN3d (8 parity classes)        stackhourglass.py:22-30
//   Conv2d + BN2d of feature_extraction (D=1)                       submodule.py:13-16,60-139
//
// Mapping to the hardware
//   * implicit GEMM on v_mfma_f32_16x16x4_f32 (fp32 in / fp32 accumulate, exact FMA chain):
//       A = weights  [16 cout  x 4 cin]   lane l: cout = l&15, cin quad member k = l>>4
//       B = voxels   [4 cin    x 16 vox]  lane l: voxel slot = l&15, k = l>>4
//       D[cout][voxel]: lane l holds couts 4*(l>>4)..+3 of voxel l&15  -> one float4 store per tile
//   * a WAVE owns a group of R output rows x WT columns (VT*16 voxel slots) of one (n, od) slice and
//     CT*16 output channels: VT*CT accumulators of 4 registers.
//   * per phase (one depth offset, one 8-channel half of a 16-channel input block) the wave stages the input rows
//     it needs into ITS OWN LDS region with global_load_lds_dwordx4 (no VGPR round trip, no block barrier: waves
//     are independent), double-buffered; the (kh,kw) taps of the phase are LDS address offsets.
//   * weights stream from L2/L1 straight into VGPRs (the 4 waves of a block share them through L1);
//     weights and B fragments are register double-buffered one tap ahead (A/B sets, no runtime indexing).
//   * epilogue: folded-BN scale/shift, residual add, ReLU, coalesced float4 stores (16 voxels x 64 B).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef WAVES_PER_BLOCK
#define WAVES_PER_BLOCK 4
#endif

namespace {

// A phase covers one depth tap, one 16-channel input block and one 8-channel HALF of it: the LDS tile of a wave is
// [rows_in][seg_vox][8 floats] (32 B per voxel), double-buffered.  Half-width phases keep a wave's LDS footprint
// near 11 KB, so 8-12 waves (2-3 per SIMD) are resident per CU and one wave's prologue / epilogue / memory waits
// overlap another wave's MFMAs.
template <int VT, int CT>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void tapconv_kernel(const drc_tapconv_params p) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;  // voxel slot within a tile (B operand) / cout within a tile (A operand)
    const int g = lane >> 4;  // k member

    const drc_tap_class cls = p.cls[blockIdx.z];
    const int n_wt = (p.OW + p.WT - 1) / p.WT;
    const int n_rt = (p.OH + p.R - 1) / p.R;
    const int groups = p.N * p.OD * n_rt * n_wt;
    // PERSISTENT waves: wave w of the launch walks groups w, w+G, w+2G, ... (G = resident waves).  While a group's last
    // phase computes, the first tile of the wave's NEXT group is already being staged, so only the very first tile of
    // a wave is an exposed HBM round trip; with 2-3 waves per SIMD the epilogue of one group hides behind a
    // neighbour's MFMAs.
    const int G = gridDim.x * WAVES_PER_BLOCK;
    int gid = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (gid >= groups) return;  // wave-uniform; the kernel has no block-level barrier
    struct GroupPos { int n, od, oh0, ow0; };
    auto decode = [&](int gidx) -> GroupPos {
        GroupPos q;
        const int wt = gidx % n_wt; gidx /= n_wt;
        const int rt = gidx % n_rt; gidx /= n_rt;
        q.od = gidx % p.OD; q.n = gidx / p.OD;
        q.oh0 = rt * p.R; q.ow0 = wt * p.WT;
        return q;
    };
    auto group_base = [&](const GroupPos& q) -> const float* {   // input pointer of the group's tile origin (depth tap 0)
        return p.x + (int64_t)q.n * p.x_n_stride + (int64_t)(p.in_mul * q.od + cls.dd0) * p.x_d_stride +
               (int64_t)(p.in_mul * q.oh0 + cls.dh0) * p.x_h_stride + (int64_t)(p.in_mul * q.ow0 + cls.dw0) * 16;
    };
    const int ct0 = blockIdx.y * CT;

    const int rows_in = p.in_mul * (p.R - 1) + (cls.nh - 1) * cls.sh + 1;
    const int seg_vox = p.in_mul * (p.WT - 1) + (cls.nw - 1) * cls.sw + 1;
    const int seg_floats = seg_vox * 8;          // LDS floats per staged row (8 channels per voxel)
    const int seg_units = seg_vox * 2;           // 16-byte units per staged row
    const int buf_floats = rows_in * seg_floats;
    float* lds = lds_all + wave * (p.lds_bytes_per_wave >> 2);
    const int nslots = p.R * p.WT;

    // per-lane B-operand offsets (floats) inside a staged tile: voxel (r,c), channels 2g,2g+1 of the half
    int lane_off[VT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
        const int s = vt * 16 + j;
        int r = s / p.WT, c = s - r * p.WT;
        if (s >= nslots) { r = 0; c = 0; }
        lane_off[vt] = (p.in_mul * r * seg_vox + p.in_mul * c) * 8 + g * 2;
    }

    GroupPos cur = decode(gid);
    const float* xcur = group_base(cur);     // this group's input origin
    const float* xnext = xcur;               // next group's input origin (valid when gid + G < groups)
    const int nt = cls.nh * cls.nw;            // taps per phase
    const int n_ph = cls.nd * p.cb_in * 2;     // phases: depth tap (outer) x input channel block x half (inner)

    // LDS-DMA rows [r0, r1) of phase ph's input tile into its LDS buffer.  Lane l of a piece moves 16 B: LDS side is
    // lane-linear (piece*1 KiB + l*16), global side is per-lane: voxel (unit>>1), 16-byte part (unit&1) of the half.
    // The rows of the NEXT phase are spread over the tap steps of the current one: a smooth request stream instead of
    // one burst per phase from every wave at once.
    auto stage_rows = [&](const float* base, int ph, int bufi, int r0, int r1) {
        const int h = ph & 1, pc = ph >> 1;
        const int di = pc / p.cb_in, cb = pc - di * p.cb_in;
        const float* src = base + (int64_t)cb * p.x_cb_stride + (int64_t)(di * cls.sd) * p.x_d_stride + h * 8;
        float* dst = lds + bufi * buf_floats;
        for (int r = r0; r < r1; ++r) {
            const float* srow = src + (int64_t)r * p.x_h_stride;
            float* drow = dst + r * seg_floats;
            for (int u0 = 0; u0 < seg_units; u0 += 64) {
                const int u = u0 + lane;
                if (u < seg_units)
                    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(srow + (u >> 1) * 16 + (u & 1) * 4), LDS_PTR(drow + u0 * 4), 16, 0, 0);
            }
        }
    };
    const int rows_per_step = (rows_in + (nt > 1 ? nt - 2 : 0)) / (nt > 1 ? nt - 1 : 1);

    f32x4 acc[VT][CT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[vt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // weight fragment address: packed [widx][cb][half][cout_pad][8]; lane (cout j, k member g) reads channels 2g,2g+1
    const float* wlane = p.w + ((int64_t)(ct0 * 16 + j)) * 8 + g * 2;
    const int64_t w_half_stride = (int64_t)p.cout_pad * 8;
    const int64_t w_tap_stride = w_half_stride * 2 * p.cb_in;

    // "next step" cursor, advanced incrementally (no divisions in the loop); wraps to the first step of the next group
    int n_di = 0, n_pcb = 0, n_tb = 0, n_tc = 0;   // n_pcb = cb*2+h within the depth tap
    auto next_wptr = [&]() -> const float* {
        const int widx = cls.wbase + n_di * cls.wsd + n_tb * cls.wsh + n_tc * cls.wsw;
        return wlane + (int64_t)widx * w_tap_stride + (int64_t)n_pcb * w_half_stride;
    };
    auto next_tap_off = [&]() -> int { return (n_tb * cls.sh * seg_vox + n_tc * cls.sw) * 8; };
    auto advance = [&]() {
        if (++n_tc == cls.nw) {
            n_tc = 0;
            if (++n_tb == cls.nh) {
                n_tb = 0;
                if (++n_pcb == 2 * p.cb_in) { n_pcb = 0; if (++n_di == cls.nd) n_di = 0; }
            }
        }
    };

    f32x2 wA[CT], wB[CT], bA[VT], bB[VT];
    int bufsel = 0;                               // LDS buffer holding the phase being computed
    stage_rows(xcur, 0, 0, 0, rows_in);           // the only exposed tile load of this wave
    {
        const float* wp = next_wptr();
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) wA[ct] = *(const f32x2*)(wp + ct * 128);
    }
    // folded-BN scale/shift of this wave's couts (same for every group)
    f32x4 bn_sc[CT], bn_sh[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        bn_sc[ct] = *(const f32x4*)(p.scale + (ct0 + ct) * 16 + g * 4);
        bn_sh[ct] = *(const f32x4*)(p.shift + (ct0 + ct) * 16 + g * 4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#define DRC_MFMA(VT0, VT1, KK, W_USE, B_USE)                                                           \
    _Pragma("unroll") for (int vt = VT0; vt < VT1; ++vt)                                               \
        _Pragma("unroll") for (int ct = 0; ct < CT; ++ct)                                              \
            acc[vt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(W_USE[ct][KK], B_USE[vt][KK], acc[vt][ct], 0, 0, 0);

// One tap step (2*VT*CT MFMAs).  The first CT MFMAs carry the wait for this step's operands; every prefetch (next
// LDS rows via LDS-DMA, next weights, next B fragments) is issued right after them, i.e. almost a full step before
// it is needed, so the (conservative, vmcnt(0)/lgkmcnt(0)) waits hipcc places never see a young load.
#define DRC_STEP(T, W_USE, B_USE, W_LD, B_LD)                                                          \
    {                                                                                                  \
        DRC_MFMA(0, 1, 0, W_USE, B_USE)                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (st_on) {                                                                                   \
            const int r0_ = (T) * rows_per_step;                                                       \
            if (r0_ < rows_in) stage_rows(st_base, st_ph, bufsel ^ 1, r0_, r0_ + rows_per_step < rows_in ? r0_ + rows_per_step : rows_in); \
        }                                                                                              \
        advance();                                                                                     \
        const bool in_phase_ = (T) + 1 < nt;                                                           \
        const float* wp_ = next_wptr();                                                                \
        _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) W_LD[ct] = *(const f32x2*)(wp_ + ct * 128);  \
        const int to_ = in_phase_ ? next_tap_off() : 0;                                                \
        _Pragma("unroll") for (int vt = 0; vt < VT; ++vt) B_LD[vt] = *(const f32x2*)(buf + lane_off[vt] + to_); \
        DRC_MFMA(1, VT, 0, W_USE, B_USE)                                                               \
        DRC_MFMA(0, VT, 1, W_USE, B_USE)                                                               \
    }

    for (;;) {
        const bool has_next = gid + G < groups;
        if (has_next) xnext = group_base(decode(gid + G));
        for (int ph = 0; ph < n_ph; ++ph) {
            const float* buf = lds + bufsel * buf_floats;
            // what this phase's steps stage into the other buffer: this group's next phase, or the next group's first
            const bool st_on = (ph + 1 < n_ph) || has_next;
            const float* st_base = (ph + 1 < n_ph) ? xcur : xnext;
            const int st_ph = (ph + 1 < n_ph) ? ph + 1 : 0;
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) bA[vt] = *(const f32x2*)(buf + lane_off[vt]);   // tap 0 of this phase
            int t = 0;
            for (; t + 1 < nt; t += 2) {
                DRC_STEP(t, wA, bA, wB, bB);
                DRC_STEP(t + 1, wB, bB, wA, bA);
            }
            if (t < nt) {
                DRC_STEP(t, wA, bA, wB, bB);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) wA[ct] = wB[ct];
            }
            // the staged tile must have landed before the next phase reads it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bufsel ^= 1;
        }

        // epilogue of this group: folded BN, residual, ReLU, store; then clear the accumulators
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) {
            const int s = vt * 16 + j;
            const int r = s / p.WT, c = s - r * p.WT;
            const bool valid = (s < nslots) && (cur.oh0 + r < p.OH) && (cur.ow0 + c < p.OW);
            if (valid) {
                const int zd = cur.od * p.out_mul + cls.out_off_d, zh = (cur.oh0 + r) * p.out_mul + cls.out_off_h,
                          zw = (cur.ow0 + c) * p.out_mul + cls.out_off_w;
                const int64_t yo = p.y_off0 + (int64_t)cur.n * p.y_n_stride + (int64_t)zd * p.y_d_stride +
                                   (int64_t)zh * p.y_h_stride + (int64_t)zw * 16 + g * 4;
                const int64_t ro = p.r_off0 + (int64_t)cur.n * p.r_n_stride + (int64_t)zd * p.r_d_stride +
                                   (int64_t)zh * p.r_h_stride + (int64_t)zw * 16 + g * 4;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    f32x4 v = acc[vt][ct] * bn_sc[ct] + bn_sh[ct];
                    if (p.res) v += *(const f32x4*)(p.res + ro + (int64_t)(ct0 + ct) * p.r_cb_stride);
                    if (p.relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    *(f32x4*)(p.y + yo + (int64_t)(ct0 + ct) * p.y_cb_stride) = v;
                }
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[vt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (!has_next) break;
        gid += G;
        cur = decode(gid);
        xcur = xnext;
    }
#undef DRC_STEP
#undef DRC_MFMA
}

template <int VT, int CT>
int launch(const drc_tapconv_params& p, hipStream_t stream) {
    const int n_wt = (p.OW + p.WT - 1) / p.WT;
    const int n_rt = (p.OH + p.R - 1) / p.R;
    const long groups = (long)p.N * p.OD * n_rt * n_wt;
    // persistent launch: at most as many blocks as stay resident (256 CUs x blocks per CU by LDS and registers);
    // each wave then walks several groups back to back
    const size_t lds_blk = (size_t)p.lds_bytes_per_wave * WAVES_PER_BLOCK;
    static size_t occ_lds = (size_t)-1;   // per-instantiation cache of the occupancy query (benign race: idempotent)
    static int occ_blocks = 1;
    if (occ_lds != lds_blk) {
        int nb = 0;
        (void)hipFuncSetAttribute((const void*)tapconv_kernel<VT, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, tapconv_kernel<VT, CT>, 64 * WAVES_PER_BLOCK, lds_blk) != hipSuccess || nb < 1) nb = 1;
        occ_blocks = nb;
        occ_lds = lds_blk;
    }
    const int blocks_per_cu = occ_blocks;
    const long want = (groups + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    const long yz = (long)(p.cout_pad / 16 / CT) * p.n_classes;
    long cap = (256L * blocks_per_cu + yz - 1) / yz;
    if (cap < 1) cap = 1;
    dim3 grid((unsigned)(want < cap ? want : cap), (unsigned)(p.cout_pad / 16 / CT), (unsigned)p.n_classes);
    const size_t lds = (size_t)p.lds_bytes_per_wave * WAVES_PER_BLOCK;
    static bool attr_done = false;  // idempotent attribute set; benign race
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)tapconv_kernel<VT, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((tapconv_kernel<VT, CT>), grid, dim3(64 * WAVES_PER_BLOCK), lds, stream, p);
    return (int)hipGetLastError();
}

template <int CT>
int launch_vt(int nvt, const drc_tapconv_params& p, hipStream_t s) {
    switch (nvt) {
        case 1: return launch<1, CT>(p, s);
        case 2: return launch<2, CT>(p, s);
        case 3: return launch<3, CT>(p, s);
        case 4: return launch<4, CT>(p, s);
        case 5: return launch<5, CT>(p, s);
        case 6: return launch<6, CT>(p, s);
        case 7: return launch<7, CT>(p, s);
    }
    return -3;
}

}  // namespace

extern "C" int drc_tapconv_fwd(const drc_tapconv_params* pp, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD <= 0 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;  // empty ROI batch: nothing to launch (reference: ROIAlign_cuda.cu:278-281 behaviour)
    if (p.cout_pad <= 0 || (p.cout_pad & 15) || p.cb_in <= 0) return -2;
    if (p.R <= 0 || p.WT <= 0 || p.R * p.WT > 112) return -3;
    if (p.n_classes < 1 || p.n_classes > DRC_MAX_CLASSES) return -4;
    if ((p.in_mul != 1 && p.in_mul != 2) || (p.out_mul != 1 && p.out_mul != 2)) return -2;
    int need = 0;
    for (int c = 0; c < p.n_classes; ++c) {
        const drc_tap_class& k = p.cls[c];
        if (k.nd < 1 || k.nh < 1 || k.nw < 1 || k.sd < 0 || k.sh < 0 || k.sw < 0) return -4;
        if (k.dd0 < 0 || k.dh0 < 0 || k.dw0 < 0) return -4;
        const int rows_in = p.in_mul * (p.R - 1) + (k.nh - 1) * k.sh + 1;
        const int seg_vox = p.in_mul * (p.WT - 1) + (k.nw - 1) * k.sw + 1;
        const int bytes = rows_in * seg_vox * 32 * 2;
        if (bytes > need) need = bytes;
    }
    if (p.lds_bytes_per_wave < need || (p.lds_bytes_per_wave & 15) || (size_t)p.lds_bytes_per_wave * 4 > 160 * 1024) return -5;
    hipStream_t s = (hipStream_t)stream;
    const int ct = p.cout_pad / 16;
    const int nvt = (p.R * p.WT + 15) / 16;
    // cout tiles per wave: as many as divide the layer (A-operand reuse), but split further when the launch would
    // otherwise leave most of the 1024 SIMDs without a wave (small pyramid levels)
    const long groups = (long)p.N * p.OD * ((p.OH + p.R - 1) / p.R) * ((p.OW + p.WT - 1) / p.WT) * p.n_classes;
    int CT = (ct % 4 == 0) ? 4 : (ct % 2 == 0) ? 2 : 1;
    while (CT > 1 && (groups * (ct / CT) < 2048 || nvt * CT > 16)) CT >>= 1;   // <=16 accumulators: 2+ waves per SIMD
    if (CT == 4) return launch_vt<4>(nvt, p, s);
    if (CT == 2) return launch_vt<2>(nvt, p, s);
    return launch_vt<1>(nvt, p, s);
}
