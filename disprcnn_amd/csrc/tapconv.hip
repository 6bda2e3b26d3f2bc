// tapconv.hip -- tap-list convolution on channel-blocked, zero-haloed tensors (gfx950 / CDNA4).
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
//   * a WAVE owns a group of R output rows x WT columns (<= VT*16 voxel slots) of one (n, od) slice and
//     ALL (CT*16) output channels of its cout chunk: VT*CT accumulators of 4 VGPRs.
//   * per phase (one depth offset dd, one 16-channel input block) the wave stages the input rows it needs
//     into ITS OWN LDS region with global_load_lds_dwordx4 (no VGPR round trip, no block barrier: waves are
//     independent), double-buffered; the 3x3 (kh,kw) taps of the phase are LDS address offsets.
//   * weights stream from L2/L1 straight into VGPRs (shared by the 4 waves of a block through L1),
//     prefetched one tap ahead.
//   * epilogue: folded-BN scale/shift, residual add, ReLU, coalesced float4 stores (16 voxels x 64 B).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

namespace {

template <int VT, int CT>
__global__ __launch_bounds__(256) void tapconv_kernel(const drc_tapconv_params p) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;  // voxel slot within a tile / cout within a tile (A operand)
    const int g = lane >> 4;  // k member

    const drc_tap_class& cls = p.cls[blockIdx.z];
    const int n_wt = (p.OW + p.WT - 1) / p.WT;
    const int n_rt = (p.OH + p.R - 1) / p.R;
    const int groups = p.N * p.OD * n_rt * n_wt;
    int gid = blockIdx.x * 4 + wave;
    if (gid >= groups) return;  // wave-uniform; the kernel has no block-level barrier
    const int wt = gid % n_wt; gid /= n_wt;
    const int rt = gid % n_rt; gid /= n_rt;
    const int od = gid % p.OD;
    const int n = gid / p.OD;
    const int oh0 = rt * p.R, ow0 = wt * p.WT;
    const int ct0 = blockIdx.y * CT;

    const int rows_in = p.in_mul * (p.R - 1) + (cls.max_dh - cls.min_dh) + 1;
    const int seg_vox = p.in_mul * (p.WT - 1) + (cls.max_dw - cls.min_dw) + 1;
    const int seg_floats = seg_vox * 16;
    const int buf_floats = rows_in * seg_floats;
    float* lds = lds_all + wave * (p.lds_bytes_per_wave >> 2);

    // per-lane B-operand offsets (floats, inside a staged tile) and output offsets
    const int nslots = p.R * p.WT;
    const int nvt = (nslots + 15) >> 4;
    int lane_off[VT];
    int64_t yoff[VT], roff[VT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
        const int s = vt * 16 + j;
        int r = s / p.WT, c = s - r * p.WT;
        const bool valid = (s < nslots) && (oh0 + r < p.OH) && (ow0 + c < p.OW);
        if (!valid) { r = 0; c = 0; }
        lane_off[vt] = (p.in_mul * r * seg_vox + p.in_mul * c) * 16 + g * 4;
        yoff[vt] = valid ? (p.y_off0 + (int64_t)n * p.y_n_stride +
                            (int64_t)(od * p.out_mul + cls.out_off_d) * p.y_d_stride +
                            (int64_t)((oh0 + r) * p.out_mul + cls.out_off_h) * p.y_h_stride +
                            (int64_t)((ow0 + c) * p.out_mul + cls.out_off_w) * 16 + g * 4)
                         : (int64_t)-1;
        roff[vt] = p.r_off0 + (int64_t)n * p.r_n_stride + (int64_t)(od * p.out_mul + cls.out_off_d) * p.r_d_stride +
                   (int64_t)((oh0 + r) * p.out_mul + cls.out_off_h) * p.r_h_stride +
                   (int64_t)((ow0 + c) * p.out_mul + cls.out_off_w) * 16 + g * 4;
    }

    const float* xbase = p.x + (int64_t)n * p.x_n_stride + (int64_t)(p.in_mul * oh0 + cls.min_dh) * p.x_h_stride +
                         (int64_t)(p.in_mul * ow0 + cls.min_dw) * 16;
    const int n_ph = cls.n_phase * p.cb_in;

    // stage phase `ph` (depth offset index ph / cb_in, channel block ph % cb_in) into LDS buffer `b`
    auto stage = [&](int ph, int b) {
        const int di = ph / p.cb_in, cb = ph - di * p.cb_in;
        const int dd = p.taps[cls.phase_tap_begin[di]].dd;
        const float* src = xbase + (int64_t)cb * p.x_cb_stride + (int64_t)(p.in_mul * od + dd) * p.x_d_stride;
        float* dst = lds + b * buf_floats;
        for (int r = 0; r < rows_in; ++r) {
            const float* srow = src + (int64_t)r * p.x_h_stride;
            float* drow = dst + r * seg_floats;
            for (int piece = 0; piece < seg_floats; piece += 256) {
                const int idx = piece + lane * 4;
                if (idx < seg_floats)
                    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(srow + idx), LDS_PTR(drow + piece), 16, 0, 0);
            }
        }
    };

    f32x4 acc[VT][CT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[vt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // weight pointer of (tap t, channel block cb): slab [cb_in][cout_pad][16]
    const float* wlane = p.w + ((int64_t)(ct0 * 16 + j)) * 16 + g * 4;
    const int64_t w_cb_stride = (int64_t)p.cout_pad * 16;
    const int64_t w_tap_stride = w_cb_stride * p.cb_in;

    stage(0, 0);
    f32x4 wcur[CT], wnext[CT];
    {
        const float* wp = wlane + (int64_t)p.taps[cls.tap_begin].widx * w_tap_stride;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) wcur[ct] = *(const f32x4*)(wp + ct * 256);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    for (int ph = 0; ph < n_ph; ++ph) {
        const int di = ph / p.cb_in, cb = ph - di * p.cb_in;
        const int tb = cls.phase_tap_begin[di], te = cls.phase_tap_begin[di + 1];
        const float* buf = lds + (ph & 1) * buf_floats;
        const int t_stage = (te - 2 > tb) ? te - 2 : tb;
        for (int t = tb; t < te; ++t) {
            if (t == t_stage && ph + 1 < n_ph) stage(ph + 1, (ph + 1) & 1);
            // prefetch the next step's weights (next tap of this phase, or first tap of the next phase)
            {
                int tn = t + 1, phn = ph;
                if (tn == te) { phn = ph + 1; }
                if (phn < n_ph) {
                    const int din = phn / p.cb_in, cbn = phn - din * p.cb_in;
                    if (tn == te) tn = cls.phase_tap_begin[din];
                    const float* wp = wlane + (int64_t)p.taps[tn].widx * w_tap_stride + (int64_t)cbn * w_cb_stride;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) wnext[ct] = *(const f32x4*)(wp + ct * 256);
                }
            }
            const drc_tap tp = p.taps[t];
            const int tap_off = ((tp.dh - cls.min_dh) * seg_vox + (tp.dw - cls.min_dw)) * 16;
            f32x4 bf[VT];
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
                if (vt < nvt) bf[vt] = *(const f32x4*)(buf + lane_off[vt] + tap_off);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int vt = 0; vt < VT; ++vt)
                    if (vt < nvt)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            acc[vt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[ct][kk], bf[vt][kk], acc[vt][ct], 0, 0, 0);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) wcur[ct] = wnext[ct];
        }
        // the next phase's tile (issued >= one tap ago) and the prefetched weights must have landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        (void)cb;
    }

    // epilogue: folded BN, residual, ReLU, store
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const f32x4 sc = *(const f32x4*)(p.scale + (ct0 + ct) * 16 + g * 4);
        const f32x4 sh = *(const f32x4*)(p.shift + (ct0 + ct) * 16 + g * 4);
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) {
            if (vt < nvt && yoff[vt] >= 0) {
                const int64_t o = yoff[vt] + (int64_t)(ct0 + ct) * p.y_cb_stride;
                f32x4 v = acc[vt][ct] * sc + sh;
                if (p.res) v += *(const f32x4*)(p.res + roff[vt] + (int64_t)(ct0 + ct) * p.r_cb_stride);
                if (p.relu) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                *(f32x4*)(p.y + o) = v;
            }
        }
    }
}

template <int VT, int CT>
int launch(const drc_tapconv_params& p, hipStream_t stream) {
    const int n_wt = (p.OW + p.WT - 1) / p.WT;
    const int n_rt = (p.OH + p.R - 1) / p.R;
    const long groups = (long)p.N * p.OD * n_rt * n_wt;
    dim3 grid((unsigned)((groups + 3) / 4), (unsigned)(p.cout_pad / 16 / CT), (unsigned)p.n_classes);
    const size_t lds = (size_t)p.lds_bytes_per_wave * 4;
    if (lds > 160 * 1024) return -5;
    static bool attr_done = false;  // idempotent attribute set; benign race
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)tapconv_kernel<VT, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((tapconv_kernel<VT, CT>), grid, dim3(256), lds, stream, p);
    return (int)hipGetLastError();
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
        if (k.tap_begin < 0 || k.tap_end > DRC_MAX_TAPS || k.tap_end <= k.tap_begin) return -4;
        if (k.n_phase < 1 || k.n_phase > 3) return -4;
        const int rows_in = p.in_mul * (p.R - 1) + (k.max_dh - k.min_dh) + 1;
        const int seg_vox = p.in_mul * (p.WT - 1) + (k.max_dw - k.min_dw) + 1;
        const int bytes = rows_in * seg_vox * 64 * 2;
        if (bytes > need) need = bytes;
    }
    if (p.lds_bytes_per_wave < need || (p.lds_bytes_per_wave & 15)) return -5;
    hipStream_t s = (hipStream_t)stream;
    const int ct = p.cout_pad / 16;
    const int nvt = (p.R * p.WT + 15) / 16;
    // instantiations: voxel tiles per wave {4,7} x cout tiles per wave {1,2,4}
    if (ct % 4 == 0) return nvt <= 4 ? launch<4, 4>(p, s) : launch<7, 4>(p, s);
    if (ct % 2 == 0) return nvt <= 4 ? launch<4, 2>(p, s) : launch<7, 2>(p, s);
    return nvt <= 4 ? launch<4, 1>(p, s) : launch<7, 1>(p, s);
}
