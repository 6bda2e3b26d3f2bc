// det_ops.hip -- box arithmetic of the 2D detection stage (gfx950): SURVEY f4.  HBM-bound elementwise kernels.
//   reference: modeling/box_coder.py:161-244 (BoxCoder.decode, 4 and 6 codes per box),
//              modeling/rpn/stereo_rpn/inference.py:121-150,287-299 (score / regression flattening, left / right split, clip_boxes),
//              modeling/rpn/stereo_rpn/srpn.py:41-50 (the pairwise softmax of the objectness map).
// drc_box_decode_fwd     : the decode alone, any number of classes per row, optional clip to the image.
// drc_srpn_proposals_fwd : one pass from the Stereo-RPN head's dense maps to per-anchor (score, left box, right box): the
//                          reference's softmax + permute + view + cat + decode + index + clamp chain (~20 elementwise launches and
//                          four intermediate copies of the maps) in one read of the maps and one write of the 9 outputs per anchor.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

namespace {

constexpr int kThreads = 256;

struct Ref { float w, h, cx, cy; };
__device__ __forceinline__ Ref ref_of(const float* b) {
    Ref r;
    r.w = b[2] - b[0] + 1.f;
    r.h = b[3] - b[1] + 1.f;
    r.cx = b[0] + 0.5f * r.w;
    r.cy = b[1] + 0.5f * r.h;
    return r;
}
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// rows x groups threads; a group is 4 or 6 consecutive codes of a row
__global__ __launch_bounds__(kThreads) void box_decode_kernel(const float* __restrict__ codes, const float* __restrict__ boxes, float* __restrict__ out,
                                                              long rows, int groups, int per, float wx, float wy, float ww, float wh, float xclip,
                                                              float img_w, float img_h) {
    const long total = rows * groups;
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
        const long r = idx / groups;
        const int gi = (int)(idx - r * groups);
        const Ref a = ref_of(boxes + r * 4);
        const float* c = codes + (r * groups + gi) * per;
        float* o = out + (r * groups + gi) * per;
        const float dx = c[0] / wx, dy = c[1] / wy;
        const float dw = fminf(c[2] / ww, xclip), dh = fminf(c[3] / wh, xclip);
        const float pcx = dx * a.w + a.cx, pcy = dy * a.h + a.cy;
        const float pw = expf(dw) * a.w, ph = expf(dh) * a.h;
        float v[6] = {pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph, 0.f, 0.f};
        if (per == 6) {
            const float dxp = c[4] / wx, dwp = fminf(c[5] / ww, xclip);
            const float pcxp = dxp * a.w + a.cx, pwp = expf(dwp) * a.w;
            v[4] = pcxp - 0.5f * pwp; v[5] = pcxp + 0.5f * pwp;
        }
        if (img_w > 0.f) {
            v[0] = clampf(v[0], 0.f, img_w - 1.f); v[2] = clampf(v[2], 0.f, img_w - 1.f);
            v[1] = clampf(v[1], 0.f, img_h - 1.f); v[3] = clampf(v[3], 0.f, img_h - 1.f);
            v[4] = clampf(v[4], 0.f, img_w - 1.f); v[5] = clampf(v[5], 0.f, img_w - 1.f);
        }
        for (int k = 0; k < per; ++k) o[k] = v[k];
    }
}

// thread = (image n, position p = y*W + x, anchor a) of one level; maps are dense NCHW
__global__ __launch_bounds__(kThreads) void srpn_proposals_kernel(const float* __restrict__ logits, const float* __restrict__ reg,
                                                                  const float* __restrict__ anchors, const float* __restrict__ im_wh, int N, int A,
                                                                  int HW, long total_anchors, long level_off, float xclip,
                                                                  float* __restrict__ scores, float* __restrict__ left, float* __restrict__ right) {
    const long per_img = (long)HW * A;
    const long total = (long)N * per_img;
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
        const int n = (int)(idx / per_img);
        const long k = idx - (long)n * per_img;
        const int p = (int)(k / A), a = (int)(k - (long)p * A);
        // objectness of anchor a = channel 2a+1 of the map after the reference's softmax, which pairs channel c with c +- A
        const int c1 = 2 * a + 1;
        const int c2 = c1 < A ? c1 + A : c1 - A;
        const float* lg = logits + (long)n * 2 * A * HW + p;
        const float z1 = lg[(long)c1 * HW], z2 = lg[(long)c2 * HW];
        const float m = fmaxf(z1, z2);
        const float e1 = expf(z1 - m), e2 = expf(z2 - m);
        const long o = (long)n * total_anchors + level_off + k;
        scores[o] = e1 / (e1 + e2);
        const float* rg = reg + ((long)n * 6 * A + 6 * a) * HW + p;
        const Ref b = ref_of(anchors + k * 4);
        const float dx = rg[0], dy = rg[(long)HW], dw = fminf(rg[2L * HW], xclip), dh = fminf(rg[3L * HW], xclip);
        const float dxp = rg[4L * HW], dwp = fminf(rg[5L * HW], xclip);
        const float pcx = dx * b.w + b.cx, pcy = dy * b.h + b.cy, pw = expf(dw) * b.w, ph = expf(dh) * b.h;
        const float pcxp = dxp * b.w + b.cx, pwp = expf(dwp) * b.w;
        const float iw = im_wh[2 * n] - 1.f, ih = im_wh[2 * n + 1] - 1.f;
        const float y1 = clampf(pcy - 0.5f * ph, 0.f, ih), y2 = clampf(pcy + 0.5f * ph, 0.f, ih);
        float4 l = {clampf(pcx - 0.5f * pw, 0.f, iw), y1, clampf(pcx + 0.5f * pw, 0.f, iw), y2};
        float4 r = {clampf(pcxp - 0.5f * pwp, 0.f, iw), y1, clampf(pcxp + 0.5f * pwp, 0.f, iw), y2};
        *(float4*)(left + o * 4) = l;
        *(float4*)(right + o * 4) = r;
    }
}

inline unsigned grid_for(long total) {
    long b = (total + kThreads - 1) / kThreads;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int drc_box_decode_fwd(const float* codes, const float* boxes, float* out, int64_t rows, int groups, int per_group, const float* weights4,
                                  float xform_clip, float img_w, float img_h, void* stream) {
    if (rows < 0 || groups <= 0 || (per_group != 4 && per_group != 6) || !weights4) return -2;
    if (rows == 0) return 0;
    if (!codes || !boxes || !out) return -1;
    if (weights4[0] == 0.f || weights4[1] == 0.f || weights4[2] == 0.f || weights4[3] == 0.f) return -2;
    hipLaunchKernelGGL(box_decode_kernel, dim3(grid_for(rows * groups)), dim3(kThreads), 0, (hipStream_t)stream, codes, boxes, out, (long)rows, groups,
                       per_group, weights4[0], weights4[1], weights4[2], weights4[3], xform_clip, img_w, img_h);
    return (int)hipGetLastError();
}

extern "C" int drc_srpn_proposals_fwd(const float* logits, const float* regression, const float* anchors, const float* image_wh, int N, int A, int H,
                                      int W, int64_t total_anchors, int64_t level_offset, float xform_clip, float* scores, float* left, float* right,
                                      void* stream) {
    if (N < 0 || A <= 0 || H <= 0 || W <= 0 || level_offset < 0 || level_offset + (int64_t)H * W * A > total_anchors) return -2;
    if (N == 0) return 0;
    if (!logits || !regression || !anchors || !image_wh || !scores || !left || !right) return -1;
    hipLaunchKernelGGL(srpn_proposals_kernel, dim3(grid_for((long)N * H * W * A)), dim3(kThreads), 0, (hipStream_t)stream, logits, regression, anchors,
                       image_wh, N, A, H * W, (long)total_anchors, (long)level_offset, xform_clip, scores, left, right);
    return (int)hipGetLastError();
}
