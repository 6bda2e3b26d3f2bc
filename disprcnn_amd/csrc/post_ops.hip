// post_ops.hip -- the step behind the disparity path: per-ROI disparities -> full-image disparity / depth maps (gfx950).
//
//   reference: DisparityMapProcessor._forward_single_image (modeling/psmnet/inference.py:18-47) with DisparityMap.resize / .crop
//              (structures/disparity.py:38-77); DispRCNN3D.roi_disp_postprocess (modeling/detector/disprcnn3d.py:161-190: the
//              same with a clamp at 0 and the pasted instance masks); PointRCNN.process_input's per-ROI depth maps
//              (pointnet_module/point_rcnn/lib/net/point_rcnn.py:121-133).
//
// The reference resizes every S x S ROI map to its box with F.interpolate(bilinear, align_corners=True), scales the values by
// dst_w / S, crops to the left box's width, adds x1 - x1p, pastes it into a zero image and takes the elementwise max of the
// stack of those images -- R full-size temporaries per image and a host round trip per box (.tolist()).  Here one thread per
// output pixel walks the image's ROIs, resamples the ones that cover it and keeps the max; nothing is materialised and the
// boxes stay on the device.  HBM-bound and tiny next to the regressor (one 375 x 1242 map = 1.9 MB written once).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

namespace {

constexpr int kThreads = 256;

struct RoiBox { int x1, y1, x2, y2, x1p, x2p; };

// value of ROI map `d` (S x S) at output pixel (y, x) inside its left box: upsample_bilinear2d(align_corners=True) to
// (y2-y1) x max(x2-x1, x2p-x1p), times dst_w / S, plus x1 - x1p  (fp32 throughout, like the reference on float tensors)
__device__ __forceinline__ float roi_value(const float* __restrict__ d, int S, const RoiBox& b, int y, int x) {
    const int hr = b.y2 - b.y1;
    const int wl = b.x2 - b.x1, wr = b.x2p - b.x1p;
    const int wd = wl > wr ? wl : wr;
    const float sh = hr > 1 ? (float)(S - 1) / (float)(hr - 1) : 0.f;
    const float sw = wd > 1 ? (float)(S - 1) / (float)(wd - 1) : 0.f;
    const float fy = sh * (float)(y - b.y1), fx = sw * (float)(x - b.x1);
    int y0 = (int)fy, x0 = (int)fx;
    y0 = y0 < S - 1 ? y0 : S - 1;
    x0 = x0 < S - 1 ? x0 : S - 1;
    const int yp = y0 < S - 1 ? 1 : 0, xp = x0 < S - 1 ? 1 : 0;
    float ly = fy - (float)y0, lx = fx - (float)x0;
    ly = fminf(fmaxf(ly, 0.f), 1.f); lx = fminf(fmaxf(lx, 0.f), 1.f);
    const float* r0 = d + (int64_t)y0 * S + x0;
    const float* r1 = r0 + (int64_t)yp * S;
    const float top = (1.f - lx) * r0[0] + lx * r0[xp];
    const float bot = (1.f - lx) * r1[0] + lx * r1[xp];
    const float v = (1.f - ly) * top + ly * bot;
    return v / (float)S * (float)wd + (float)(b.x1 - b.x1p);
}

__device__ __forceinline__ RoiBox load_box(const int32_t* __restrict__ boxes, int r) {
    const int32_t* p = boxes + (int64_t)r * 6;
    return RoiBox{p[0], p[1], p[2], p[3], p[4], p[5]};
}

__global__ __launch_bounds__(kThreads) void disparity_paste_kernel(const float* __restrict__ disp, int S, const int32_t* __restrict__ boxes,
                                                                    const int32_t* __restrict__ roi_offsets, int H, int W, int flags,
                                                                    const float* __restrict__ mask, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int r0 = roi_offsets[b], r1 = roi_offsets[b + 1];
    const int64_t hw = (int64_t)H * W;
    for (int64_t pix = (int64_t)blockIdx.x * kThreads + threadIdx.x; pix < hw; pix += (int64_t)gridDim.x * kThreads) {
        const int y = (int)(pix / W), x = (int)(pix - (int64_t)y * W);
        float m = r1 > r0 ? -INFINITY : 0.f;
        for (int r = r0; r < r1; ++r) {
            const RoiBox bx = load_box(boxes, r);
            float v = 0.f;
            if (y >= bx.y1 && y < bx.y2 && x >= bx.x1 && x < bx.x2) v = roi_value(disp + (int64_t)r * S * S, S, bx, y, x);
            if (flags & 1) v = fmaxf(v, 0.f);
            if (mask) v *= mask[(int64_t)r * hw + pix];
            m = fmaxf(m, v);
        }
        out[(int64_t)b * hw + pix] = m;
    }
}

__global__ __launch_bounds__(kThreads) void roi_depth_maps_kernel(const float* __restrict__ disp, int S, const int32_t* __restrict__ boxes,
                                                                   const float* __restrict__ fuxb, int H, int W, float* __restrict__ out) {
    const int r = blockIdx.y;
    const RoiBox bx = load_box(boxes, r);
    const float k = fuxb[r];
    const int64_t hw = (int64_t)H * W;
    for (int64_t pix = (int64_t)blockIdx.x * kThreads + threadIdx.x; pix < hw; pix += (int64_t)gridDim.x * kThreads) {
        const int y = (int)(pix / W), x = (int)(pix - (int64_t)y * W);
        float v = 0.f;
        if (y >= bx.y1 && y < bx.y2 && x >= bx.x1 && x < bx.x2) v = k / (roi_value(disp + (int64_t)r * S * S, S, bx, y, x) + 1e-6f);
        out[(int64_t)r * hw + pix] = v;
    }
}

// DisparityMap.resize (reference structures/disparity.py:39-62): a whole [IH,IW] map to [OH,OW], values times OW / IW (a disparity is a
// horizontal pixel distance).  mode 0: upsample_bilinear2d(align_corners=True) -- source coordinate = dst * (in-1)/(out-1), the two rows
// blended after the two columns; mode 1: the reference's signed max pooling -- adaptive_max_pool2d of the positive part minus
// adaptive_max_pool2d of the negated negative part (window [floor(i*in/out), ceil((i+1)*in/out)) per axis).
__global__ __launch_bounds__(kThreads) void disparity_resize_kernel(const float* __restrict__ src, int IH, int IW, float* __restrict__ dst, int OH,
                                                                     int OW, int mode) {
    const int64_t n = (int64_t)OH * OW;
    const float sh = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f;
    const float sw = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
    for (int64_t pix = (int64_t)blockIdx.x * kThreads + threadIdx.x; pix < n; pix += (int64_t)gridDim.x * kThreads) {
        const int y = (int)(pix / OW), x = (int)(pix - (int64_t)y * OW);
        float v;
        if (mode == 0) {
            const float fy = sh * (float)y, fx = sw * (float)x;
            const int y0 = (int)fy, x0 = (int)fx;
            const int yp = y0 < IH - 1 ? 1 : 0, xp = x0 < IW - 1 ? 1 : 0;
            const float ly = fy - (float)y0, lx = fx - (float)x0;
            const float* r0 = src + (int64_t)y0 * IW + x0;
            const float* r1 = r0 + (int64_t)yp * IW;
            v = (1.f - ly) * ((1.f - lx) * r0[0] + lx * r0[xp]) + ly * ((1.f - lx) * r1[0] + lx * r1[xp]);
        } else {
            const int ys = (int)(((int64_t)y * IH) / OH), ye = (int)((((int64_t)y + 1) * IH + OH - 1) / OH);
            const int xs = (int)(((int64_t)x * IW) / OW), xe = (int)((((int64_t)x + 1) * IW + OW - 1) / OW);
            float mp = -INFINITY, mn = -INFINITY;
            for (int yy = ys; yy < ye; ++yy)
                for (int xx = xs; xx < xe; ++xx) {
                    const float s = src[(int64_t)yy * IW + xx];
                    mp = fmaxf(mp, s > 0.f ? s : 0.f);
                    mn = fmaxf(mn, s < 0.f ? -s : 0.f);
                }
            v = mp - mn;
        }
        dst[pix] = v / (float)IW * (float)OW;
    }
}

unsigned blocks_for(int64_t n) {
    int64_t b = (n + kThreads - 1) / kThreads;
    return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int drc_disparity_paste_fwd(const float* disp, int S, const int32_t* boxes, const int32_t* roi_offsets, int B, int H, int W,
                                       int flags, const float* mask, float* out, void* stream) {
    if (B < 0 || H <= 0 || W <= 0 || S <= 0) return -2;
    if (B == 0) return 0;
    if (!roi_offsets || !out) return -1;            // disp / boxes may be NULL when no image has a ROI
    hipLaunchKernelGGL(disparity_paste_kernel, dim3(blocks_for((int64_t)H * W), (unsigned)B), dim3(kThreads), 0, (hipStream_t)stream, disp, S,
                       boxes, roi_offsets, H, W, flags, mask, out);
    return (int)hipGetLastError();
}

extern "C" int drc_roi_depth_maps_fwd(const float* disp, int S, const int32_t* boxes, const float* fuxb, int R, int H, int W, float* out,
                                      void* stream) {
    if (R < 0 || H <= 0 || W <= 0 || S <= 0) return -2;
    if (R == 0) return 0;
    if (!disp || !boxes || !fuxb || !out) return -1;
    hipLaunchKernelGGL(roi_depth_maps_kernel, dim3(blocks_for((int64_t)H * W), (unsigned)R), dim3(kThreads), 0, (hipStream_t)stream, disp, S,
                       boxes, fuxb, H, W, out);
    return (int)hipGetLastError();
}

extern "C" int drc_disparity_resize_fwd(const float* src, int IH, int IW, float* dst, int OH, int OW, int mode, void* stream) {
    if (IH <= 0 || IW <= 0 || OH < 0 || OW < 0 || (mode != 0 && mode != 1)) return -2;
    if (OH == 0 || OW == 0) return 0;
    if (!src || !dst) return -1;
    hipLaunchKernelGGL(disparity_resize_kernel, dim3(blocks_for((int64_t)OH * OW)), dim3(kThreads), 0, (hipStream_t)stream, src, IH, IW, dst, OH, OW,
                       mode);
    return (int)hipGetLastError();
}
