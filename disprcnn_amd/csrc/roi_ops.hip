// roi_ops.hip -- the step in front of the disparity path: ROI pairing and ROIAlign crops (gfx950).
//
//   drc_roi_align_fwd   : ROIAlign forward, semantics of the reference op (csrc/cpu/ROIAlign_cpu.cpp:18-111,113-219,
//                         csrc/cuda/ROIAlign_cuda.cu:64-122) with the ImageNet normalisation of
//                         DispRCNN3D.crop_and_transform_roi_img (disprcnn3d.py:44-50) optionally fused in.
//   drc_roi_align_bwd   : its adjoint (csrc/cuda/ROIAlign_cuda.cu:177-254), atomicAdd scatter.
//   drc_align_roi_pairs : the per-ROI box arithmetic of prepare_psmnet_input_and_target (disprcnn3d.py:118-146) on the
//                         device, so no .tolist() host sync sits between the 2D detections and the crops.
// All HBM-bound and tiny next to the regressor; lanes run along x.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

namespace {

constexpr int kThreads = 256;

struct Bilin { int p1, p2, p3, p4; float w1, w2, w3, w4; };

// one bilinear sample at (y, x): reference pre_calc_for_bilinear_interpolate (ROIAlign_cpu.cpp:41-105)
__device__ __forceinline__ Bilin bilinear_setup(float y, float x, int height, int width) {
    Bilin b;
    if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) {
        b.p1 = b.p2 = b.p3 = b.p4 = 0; b.w1 = b.w2 = b.w3 = b.w4 = 0.f;
        return b;
    }
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else { y_high = y_low + 1; }
    if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else { x_high = x_low + 1; }
    const float ly = y - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
    b.p1 = y_low * width + x_low; b.p2 = y_low * width + x_high;
    b.p3 = y_high * width + x_low; b.p4 = y_high * width + x_high;
    b.w1 = hy * hx; b.w2 = hy * lx; b.w3 = ly * hx; b.w4 = ly * lx;
    return b;
}

// One thread per output PIXEL (roi k, channel group, ph, pw), CC channels at a time: the sample geometry (4 positions + 4
// weights per sample point) is computed once and applied to every channel of the group, lanes run along pw so the CC plane
// stores are coalesced and neighbouring lanes gather neighbouring image columns (bin_w <= 1 px for rois up to 224 px wide).
// Arithmetic per (sample, channel) and the accumulation order are the reference's, and the build uses -ffp-contract=off, so
// the un-normalised output is BIT-IDENTICAL to csrc/cpu/ROIAlign_cpu.cpp (tests/golden/roi_golden.npz).
// FPN = true (round 3): one launch over all pyramid levels -- every roi reads its level's map, size and scale from `pyr` through
// levels[k] -- instead of a nonzero / index_select / launch / index_copy per level (12 host syncs per stereo pair in the 2D stage's
// three poolers).  Same per-sample arithmetic, so the outputs are those of the per-level launches.
template <int CC, bool FPN>
__global__ __launch_bounds__(kThreads) void roi_align_fwd_kernel(const float* __restrict__ in0, const float* __restrict__ rois,
                                                                 float* __restrict__ out, int K, int C, int H0, int W0, int PH, int PW,
                                                                 float spatial_scale0, int sampling_ratio,
                                                                 const float* __restrict__ mean, const float* __restrict__ stdv,
                                                                 const drc_fpn_pyramid pyr, const int32_t* __restrict__ levels) {
    const int CG = C / CC;
    const long total = (long)K * CG * PH * PW;
    const long oplane = (long)PH * PW;
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
        long t = idx;
        const int pw = (int)(t % PW); t /= PW;
        const int ph = (int)(t % PH); t /= PH;
        const int cg = (int)(t % CG);
        const int k = (int)(t / CG);
        const int lv = FPN ? levels[k] : 0;
        const float* in = FPN ? pyr.feat[lv] : in0;
        const int H = FPN ? pyr.H[lv] : H0, W = FPN ? pyr.W[lv] : W0;
        const float spatial_scale = FPN ? pyr.scale[lv] : spatial_scale0;
        const long plane = (long)H * W;
        const float* r = rois + (long)k * 5;
        const int b = (int)r[0];
        // no rounding of the roi (ROIAlign_cpu.cpp:146-150); malformed rois forced to 1x1 (:157-158)
        const float rsw = r[1] * spatial_scale, rsh = r[2] * spatial_scale;
        const float rew = r[3] * spatial_scale, reh = r[4] * spatial_scale;
        const float rw = fmaxf(rew - rsw, 1.f), rh = fmaxf(reh - rsh, 1.f);
        const float bin_h = rh / (float)PH, bin_w = rw / (float)PW;
        const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
        const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
        const float count = (float)(gh * gw);
        const float* src = in + ((long)b * C + cg * CC) * plane;
        float acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = 0.f;
        for (int iy = 0; iy < gh; ++iy) {
            const float yy = rsh + ph * bin_h + (iy + 0.5f) * bin_h / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
                const float xx = rsw + pw * bin_w + (ix + 0.5f) * bin_w / (float)gw;
                const Bilin q = bilinear_setup(yy, xx, H, W);
#pragma unroll
                for (int c = 0; c < CC; ++c) {
                    const float* sc = src + c * plane;
                    acc[c] += q.w1 * sc[q.p1] + q.w2 * sc[q.p2] + q.w3 * sc[q.p3] + q.w4 * sc[q.p4];
                }
            }
        }
        float* dst = out + ((long)k * C + cg * CC) * oplane + (long)ph * PW + pw;
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            float v = acc[c] / count;
            if (mean) v = (v - mean[cg * CC + c]) / stdv[cg * CC + c];
            dst[c * oplane] = v;
        }
    }
}

__global__ __launch_bounds__(kThreads) void roi_align_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ rois,
                                                                 float* __restrict__ gin, int K, int C, int H, int W, int PH, int PW,
                                                                 float spatial_scale, int sampling_ratio) {
    const long total = (long)K * C * PH * PW;
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
        long t = idx;
        const int pw = (int)(t % PW); t /= PW;
        const int ph = (int)(t % PH); t /= PH;
        const int c = (int)(t % C);
        const int k = (int)(t / C);
        const float* r = rois + (long)k * 5;
        const int b = (int)r[0];
        const float rsw = r[1] * spatial_scale, rsh = r[2] * spatial_scale;
        const float rew = r[3] * spatial_scale, reh = r[4] * spatial_scale;
        const float rw = fmaxf(rew - rsw, 1.f), rh = fmaxf(reh - rsh, 1.f);
        const float bin_h = rh / (float)PH, bin_w = rw / (float)PW;
        const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
        const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
        const float gval = gout[idx] / (float)(gh * gw);
        float* dst = gin + ((long)b * C + c) * H * W;
        for (int iy = 0; iy < gh; ++iy) {
            const float yy = rsh + ph * bin_h + (iy + 0.5f) * bin_h / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
                const float xx = rsw + pw * bin_w + (ix + 0.5f) * bin_w / (float)gw;
                const Bilin q = bilinear_setup(yy, xx, H, W);
                if (q.w1 != 0.f || q.w2 != 0.f || q.w3 != 0.f || q.w4 != 0.f) {
                    atomicAdd(dst + q.p1, gval * q.w1); atomicAdd(dst + q.p2, gval * q.w2);
                    atomicAdd(dst + q.p3, gval * q.w3); atomicAdd(dst + q.p4, gval * q.w4);
                }
            }
        }
    }
}

// expand_box_to_integer (stereo_utils.py:219-229): floor(x1), floor(y1), ceil(x2), ceil(y2); then the clamps and the
// common width of disprcnn3d.py:121-131.
__global__ void align_roi_pairs_kernel(const float* __restrict__ lbox, const float* __restrict__ rbox, const int* __restrict__ img_idx,
                                       int R, int img_w, int img_h, float* __restrict__ rois_l, float* __restrict__ rois_r,
                                       int* __restrict__ geom) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    int x1 = (int)floorf(lbox[i * 4 + 0]), y1 = (int)floorf(lbox[i * 4 + 1]);
    int x2 = (int)ceilf(lbox[i * 4 + 2]), y2 = (int)ceilf(lbox[i * 4 + 3]);
    int x1p = (int)floorf(rbox[i * 4 + 0]), x2p = (int)ceilf(rbox[i * 4 + 2]);
    x1 = max(0, x1); x1p = max(0, x1p); y1 = max(0, y1);
    y2 = min(y2, img_h - 1); x2 = min(x2, img_w - 1); x2p = min(x2p, img_w - 1);
    int mw = max(x2 - x1, x2p - x1p);
    mw = min(mw, min(img_w - x1, img_w - x1p));
    const float b = (float)img_idx[i];
    rois_l[i * 5 + 0] = b; rois_l[i * 5 + 1] = (float)x1; rois_l[i * 5 + 2] = (float)y1;
    rois_l[i * 5 + 3] = (float)(x1 + mw); rois_l[i * 5 + 4] = (float)y2;
    rois_r[i * 5 + 0] = b; rois_r[i * 5 + 1] = (float)x1p; rois_r[i * 5 + 2] = (float)y1;
    rois_r[i * 5 + 3] = (float)(x1p + mw); rois_r[i * 5 + 4] = (float)y2;
    geom[i * 4 + 0] = x1; geom[i * 4 + 1] = x1p; geom[i * 4 + 2] = x1 + mw; geom[i * 4 + 3] = x1p + mw;
}


// ---- training targets of the disparity stage (DispRCNN3D.prepare_psmnet_input_and_target, disprcnn3d.py:52-112)
// The reference builds, per ROI and on the host: Masker(0.7, padding 1) paste of the 28x28 mask probabilities into a full-size
// uint8 image (roi_heads/mask_head/inference.py:90-190), AND with the ground-truth mask, slice to the ROI, bilinear resize
// (align_corners=True) to res x res, .byte();  DisparityMap.crop -> minus (x1 - x1p) -> resize (align_corners=True, values times
// res / width; structures/disparity.py:38-77).  Here one thread per target pixel evaluates both directly from the inputs:
// nothing full-size is materialised, no .tolist().
struct PasteBox { int bx0, by0, bw, bh; };

// expand_boxes + .to(int32) of paste_mask_in_image: box grown by (M + 2*pad) / M about its centre, truncated toward zero
__device__ __forceinline__ PasteBox paste_box(const float* __restrict__ b, int M, int pad) {
    const float scale = (float)((double)(M + 2 * pad) / (double)M);
    float w_half = (b[2] - b[0]) * 0.5f, h_half = (b[3] - b[1]) * 0.5f;
    const float xc = (b[2] + b[0]) * 0.5f, yc = (b[3] + b[1]) * 0.5f;
    w_half *= scale; h_half *= scale;
    const int x0 = (int)(xc - w_half), x1 = (int)(xc + w_half), y0 = (int)(yc - h_half), y1 = (int)(yc + h_half);
    PasteBox q;
    q.bx0 = x0; q.by0 = y0;
    q.bw = max(x1 - x0 + 1, 1); q.bh = max(y1 - y0 + 1, 1);
    return q;
}

// Masker value at image pixel (Y, X): F.interpolate(padded mask, (bh, bw), bilinear, align_corners=False) > thresh inside the
// pasted window, 0 elsewhere
__device__ __forceinline__ int masker_at(const float* __restrict__ prob, int M, int pad, const PasteBox& q, int Y, int X, int H, int W, float thresh) {
    const int x_lo = max(q.bx0, 0), x_hi = min(q.bx0 + q.bw, W), y_lo = max(q.by0, 0), y_hi = min(q.by0 + q.bh, H);
    if (X < x_lo || X >= x_hi || Y < y_lo || Y >= y_hi) return 0;
    const int P = M + 2 * pad;
    const float sy = (float)P / (float)q.bh, sx = (float)P / (float)q.bw;
    float fy = sy * ((float)(Y - q.by0) + 0.5f) - 0.5f, fx = sx * ((float)(X - q.bx0) + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < P - 1 ? 1 : 0), x1 = x0 + (x0 < P - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    auto at = [&](int yy, int xx) -> float {          // padded mask: zero border of width `pad`
        yy -= pad; xx -= pad;
        return (yy >= 0 && yy < M && xx >= 0 && xx < M) ? prob[yy * M + xx] : 0.f;
    };
    const float top = (1.f - lx) * at(y0, x0) + lx * at(y0, x1);
    const float bot = (1.f - lx) * at(y1, x0) + lx * at(y1, x1);
    return ((1.f - ly) * top + ly * bot) > thresh ? 1 : 0;
}

__global__ __launch_bounds__(kThreads) void roi_train_targets_kernel(const float* __restrict__ disp_maps, const uint8_t* __restrict__ gt_masks,
                                                                     const float* __restrict__ mask_probs, int M, int pad, float thresh,
                                                                     const float* __restrict__ det_boxes, const float* __restrict__ rois_l,
                                                                     const int32_t* __restrict__ geom, int R, int H, int W, int res,
                                                                     float* __restrict__ targets, uint8_t* __restrict__ masks) {
    const long total = (long)R * res * res;
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
        const int j = (int)(idx % res), i = (int)((idx / res) % res), r = (int)(idx / ((long)res * res));
        const int b = (int)rois_l[r * 5 + 0];
        const int x1 = geom[r * 4 + 0], x1p = geom[r * 4 + 1], mw = geom[r * 4 + 2] - geom[r * 4 + 0];
        const int y1 = (int)rois_l[r * 5 + 2], y2 = (int)rois_l[r * 5 + 4];
        const int hc = y2 - y1;
        // ---- disparity target: crop (zero beyond the image) - (x1 - x1p), resized with align_corners=True, values * res / mw
        float tv = 0.f;
        if (hc > 0 && mw > 0) {
            const float sh = res > 1 ? (float)(hc - 1) / (float)(res - 1) : 0.f, sw = res > 1 ? (float)(mw - 1) / (float)(res - 1) : 0.f;
            const float fy = sh * (float)i, fx = sw * (float)j;
            int y0 = (int)fy, x0 = (int)fx;
            y0 = y0 < hc - 1 ? y0 : hc - 1; x0 = x0 < mw - 1 ? x0 : mw - 1;
            const int yp = y0 < hc - 1 ? 1 : 0, xp = x0 < mw - 1 ? 1 : 0;
            const float ly = fy - (float)y0, lx = fx - (float)x0;
            const float off = (float)(x1 - x1p);
            const float* dm = disp_maps + (long)b * H * W;
            auto at = [&](int yy, int xx) -> float {
                const int Y = y1 + yy, X = x1 + xx;
                return ((Y >= 0 && Y < H && X >= 0 && X < W) ? dm[(long)Y * W + X] : 0.f) - off;
            };
            const float top = (1.f - lx) * at(y0, x0) + lx * at(y0, x0 + xp);
            const float bot = (1.f - lx) * at(y0 + yp, x0) + lx * at(y0 + yp, x0 + xp);
            tv = ((1.f - ly) * top + ly * bot) / (float)mw * (float)res;
        }
        targets[idx] = tv;
        // ---- mask: (Masker paste & ground truth)[y1:y2, x1:x1+mw] (slice clipped to the image), resized (align_corners=True), .byte()
        const int mh = min(y2, H) - max(y1, 0), mwc = min(x1 + mw, W) - max(x1, 0);
        uint8_t mv = 0;
        if (mh > 0 && mwc > 0) {
            const float sh = res > 1 ? (float)(mh - 1) / (float)(res - 1) : 0.f, sw = res > 1 ? (float)(mwc - 1) / (float)(res - 1) : 0.f;
            const float fy = sh * (float)i, fx = sw * (float)j;
            int y0 = (int)fy, x0 = (int)fx;
            y0 = y0 < mh - 1 ? y0 : mh - 1; x0 = x0 < mwc - 1 ? x0 : mwc - 1;
            const int yp = y0 < mh - 1 ? 1 : 0, xp = x0 < mwc - 1 ? 1 : 0;
            const float ly = fy - (float)y0, lx = fx - (float)x0;
            const PasteBox q = paste_box(det_boxes + (long)r * 4, M, pad);
            const float* prob = mask_probs + (long)r * M * M;
            const uint8_t* gm = gt_masks + (long)b * H * W;
            const int Y0 = max(y1, 0) + y0, X0 = max(x1, 0) + x0;
            auto at = [&](int Y, int X) -> float {
                return (gm[(long)Y * W + X] != 0 && masker_at(prob, M, pad, q, Y, X, H, W, thresh)) ? 1.f : 0.f;
            };
            const float top = (1.f - lx) * at(Y0, X0) + lx * at(Y0, X0 + xp);
            const float bot = (1.f - lx) * at(Y0 + yp, X0) + lx * at(Y0 + yp, X0 + xp);
            const float v = (1.f - ly) * top + ly * bot;
            mv = (uint8_t)(int)v;                       // .byte(): truncation -- only an interpolated 1.0 survives
        }
        masks[idx] = mv;
    }
}

inline unsigned grid_for(long work) {
    long b = (work + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    if (b > 256 * 16) b = 256 * 16;
    return (unsigned)b;
}

}  // namespace

extern "C" {

int drc_roi_align_fwd(const float* input, const float* rois, float* out, int K, int C, int H, int W, int PH, int PW,
                      float spatial_scale, int sampling_ratio, const float* mean, const float* stdv, void* stream) {
    if (K < 0 || C <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0 || sampling_ratio < 0) return -2;
    if ((mean == nullptr) != (stdv == nullptr)) return -2;
    if (K == 0) return 0;   // empty rois -> empty output, nothing launched (ROIAlign_cuda.cu:278-281)
    if (!input || !rois || !out) return -1;
    const int cc = C % 4 == 0 ? 4 : (C % 3 == 0 ? 3 : 1);       // channels per thread (images: 3; FPN levels: 4)
    const long total = (long)K * (C / cc) * PH * PW;
    const dim3 grid(grid_for(total)), block(kThreads);
    hipStream_t s = (hipStream_t)stream;
    const drc_fpn_pyramid none = {};
    if (cc == 4)
        hipLaunchKernelGGL((roi_align_fwd_kernel<4, false>), grid, block, 0, s, input, rois, out, K, C, H, W, PH, PW, spatial_scale, sampling_ratio, mean, stdv, none, nullptr);
    else if (cc == 3)
        hipLaunchKernelGGL((roi_align_fwd_kernel<3, false>), grid, block, 0, s, input, rois, out, K, C, H, W, PH, PW, spatial_scale, sampling_ratio, mean, stdv, none, nullptr);
    else
        hipLaunchKernelGGL((roi_align_fwd_kernel<1, false>), grid, block, 0, s, input, rois, out, K, C, H, W, PH, PW, spatial_scale, sampling_ratio, mean, stdv, none, nullptr);
    return (int)hipGetLastError();
}

int drc_roi_align_fpn_fwd(const drc_fpn_pyramid* pyr, const float* rois, const int32_t* levels, float* out, int K, int C, int PH, int PW,
                          int sampling_ratio, void* stream) {
    if (!pyr) return -1;
    if (K < 0 || C <= 0 || PH <= 0 || PW <= 0 || sampling_ratio < 0 || pyr->n_levels < 1 || pyr->n_levels > DRC_FPN_MAX_LEVELS) return -2;
    for (int i = 0; i < pyr->n_levels; ++i)
        if (!pyr->feat[i] || pyr->H[i] <= 0 || pyr->W[i] <= 0) return -2;
    if (K == 0) return 0;
    if (!rois || !levels || !out) return -1;
    const int cc = C % 4 == 0 ? 4 : 1;
    const long total = (long)K * (C / cc) * PH * PW;
    const dim3 grid(grid_for(total)), block(kThreads);
    hipStream_t s = (hipStream_t)stream;
    if (cc == 4)
        hipLaunchKernelGGL((roi_align_fwd_kernel<4, true>), grid, block, 0, s, nullptr, rois, out, K, C, 0, 0, PH, PW, 0.f, sampling_ratio, nullptr, nullptr, *pyr, levels);
    else
        hipLaunchKernelGGL((roi_align_fwd_kernel<1, true>), grid, block, 0, s, nullptr, rois, out, K, C, 0, 0, PH, PW, 0.f, sampling_ratio, nullptr, nullptr, *pyr, levels);
    return (int)hipGetLastError();
}

int drc_roi_align_bwd(const float* grad_out, const float* rois, float* grad_in, int K, int C, int H, int W, int PH, int PW,
                      float spatial_scale, int sampling_ratio, void* stream) {
    if (K < 0 || C <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0 || sampling_ratio < 0) return -2;
    if (K == 0) return 0;
    if (!grad_out || !rois || !grad_in) return -1;   // grad_in must be zero-filled by the caller
    const long total = (long)K * C * PH * PW;
    hipLaunchKernelGGL(roi_align_bwd_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, grad_out, rois, grad_in, K, C,
                       H, W, PH, PW, spatial_scale, sampling_ratio);
    return (int)hipGetLastError();
}

int drc_roi_train_targets_fwd(const float* disp_maps, const uint8_t* gt_masks, const float* mask_probs, int mask_size, int padding,
                              float mask_thresh, const float* det_boxes, const float* rois_left, const int32_t* geom, int R, int H, int W,
                              int res, float* targets, uint8_t* masks, void* stream) {
    if (R < 0 || H <= 0 || W <= 0 || res <= 0 || mask_size <= 0 || padding < 0) return -2;
    if (R == 0) return 0;
    if (!disp_maps || !gt_masks || !mask_probs || !det_boxes || !rois_left || !geom || !targets || !masks) return -1;
    const long total = (long)R * res * res;
    hipLaunchKernelGGL(roi_train_targets_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, disp_maps, gt_masks, mask_probs,
                       mask_size, padding, mask_thresh, det_boxes, rois_left, geom, R, H, W, res, targets, masks);
    return (int)hipGetLastError();
}

int drc_align_roi_pairs(const float* left_boxes, const float* right_boxes, const int32_t* img_idx, int R, int img_w, int img_h,
                        float* rois_left, float* rois_right, int32_t* geom, void* stream) {
    if (R < 0 || img_w <= 0 || img_h <= 0) return -2;
    if (R == 0) return 0;
    if (!left_boxes || !right_boxes || !img_idx || !rois_left || !rois_right || !geom) return -1;
    hipLaunchKernelGGL(align_roi_pairs_kernel, dim3((R + 63) / 64), dim3(64), 0, (hipStream_t)stream, left_boxes, right_boxes, img_idx, R,
                       img_w, img_h, rois_left, rois_right, geom);
    return (int)hipGetLastError();
}

}  // extern "C"
