/*
 * disprcnn_hip.h -- C ABI of libdisprcnn_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the instance-disparity hot path of zju3dv/disprcnn.  The
 * reference has no FFI of its own for this path on the GPU: its arithmetic is
 * torch.nn -> cuDNN, plus the pybind11 module disprcnn._C (csrc/vision.cpp:7-15) for
 * ROIAlign.  Each entry point below names the reference code it replaces.
 *
 * Conventions (SURVEY.md 8b "Ownership"/"Errors"):
 *   - the CALLER allocates every buffer; the library never allocates, frees or syncs;
 *   - all pointers are device pointers (HBM) unless stated otherwise, fp32;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it;
 *   - return 0 on success, <0 for a bad argument / unsupported shape,
 *     >0 = hipError_t from hipGetLastError() after the launch;
 *   - re-entrant: no global mutable state (callable from autograd worker threads).
 *
 * "Blocked" tensors: channel-blocked, zero-haloed layout
 *     float[N][CB][D+2pd][H+2ph][W+2pw][16],  CB = ceil(C/16)
 * The halo is written once (zero) by the allocator and never touched by a kernel,
 * so 3x3x3 taps need no bounds checks and every staged tile row is contiguous.
 */
#ifndef DISPRCNN_HIP_H
#define DISPRCNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRC_MAX_CLASSES 8
#define DRC_CB 16 /* channels per block */

/* library / build info: returns a static string "disprcnn_hip gfx950 <abi-version>" */
const char* drc_version(void);

/* ---------------------------------------------------------------------------------------
 * a1. Cost volume, reference layout (NCDHW), bit-exact copy semantics.
 * Replaces the Python slice-assign loop PSMNet.forward, stackhourglass.py:115-128.
 *   left,right : [N,C,Hp,Wp]      cost : [N,2C,Dp,Hp,Wp],  Dp = (maxdisp-mindisp)//4 computed by caller
 *   lo4 = mindisp//4 (floor), hi4 = maxdisp//4 (floor): slices j with lo4+j >= hi4 stay zero.
 */
int drc_cost_volume_fwd(const float* left, const float* right, float* cost,
                        int N, int C, int Dp, int Hp, int Wp, int lo4, int hi4, void* stream);

/* a1 backward: gL[n,c,y,x] = sum_j gcost[n,c,j,y,x]*valid, gR[n,c,y,x'] = sum_j gcost[n,C+c,j,y,x'+i]*valid
 * (adjoint of the copy; the reference gets it from autograd of stackhourglass.py:118-127). */
int drc_cost_volume_bwd(const float* gcost, float* gleft, float* gright,
                        int N, int C, int Dp, int Hp, int Wp, int lo4, int hi4, void* stream);

/* a1, blocked output: same values written as a blocked tensor [N][2C/16][Dp+2][Hp+2][Wp+2][16]
 * (pads 1,1,1), the input layout of the 3D regressor.  C must be a multiple of 16.
 * left/right are NCHW (in_blocked=0) or blocked 2D [N][C/16][Hp+2*fp][Wp+2*fp][16] (in_blocked=fp>0). */
int drc_cost_volume_blocked_fwd(const float* left, const float* right, float* cost_blk,
                                int N, int C, int Dp, int Hp, int Wp, int lo4, int hi4,
                                int in_blocked_pad, void* stream);

/* ---------------------------------------------------------------------------------------
 * Layout converters between the reference layouts and blocked tensors.
 *   dense  : [N,C,D,H,W] (D=1 for 2D)      blocked : [N][CB][D+2pd][H+2ph][W+2pw][16]
 * to_blocked writes the interior only (channels C..16*CB-1 are written as zero). */
int drc_dense_to_blocked(const float* dense, float* blk, int N, int C, int D, int H, int W,
                         int pd, int ph, int pw, void* stream);
int drc_blocked_to_dense(const float* blk, float* dense, int N, int C, int D, int H, int W,
                         int pd, int ph, int pw, void* stream);

/* ---------------------------------------------------------------------------------------
 * a2-a6, a8, a12.  Tap-list convolution on blocked tensors: one engine for
 * Conv3d k3 s1/s2 (submodule.py:19-22), ConvTranspose3d k3 s2 p1 op1 as 8 parity classes
 * (stackhourglass.py:22-30), and Conv2d (submodule.py:13-16; D=1), with the folded-BN scale/shift,
 * residual add and ReLU of the surrounding nn.Sequential fused into the epilogue.
 *
 *   y[n,co,O(o)] = act( scale[co] * sum_t sum_ci w[t][ci][co] * x[n,ci, in_mul*o + tap_t] + shift[co] (+ res[n,co,O(o)]) )
 *   O(o) = out_mul*o + class offset  (out_mul=2 for the transposed conv's parity classes)
 *
 * fp32 in, fp32 accumulate on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain).
 */
/* One tap class = a full Cartesian grid of taps (nd x nh x nw).  A plain convolution has one class
 * (3x3x3, 1xkxk, 1x1x1); the k3/s2 transposed convolution has 8 output-parity classes of {1,2}^3 taps.
 *   tap (a,b,c): input offset (dd0+a*sd, dh0+b*sh, dw0+c*sw) in padded-input coordinates relative to in_mul*o,
 *                weight slab index wbase + a*wsd + b*wsh + c*wsw  (steps may be negative). */
typedef struct drc_tap_class {
    int32_t nd, nh, nw;
    int32_t dd0, dh0, dw0;
    int32_t sd, sh, sw;
    int32_t wbase, wsd, wsh, wsw;
    int32_t out_off_d, out_off_h, out_off_w; /* parity offsets (0/1) */
} drc_tap_class;

typedef struct drc_tapconv_params {
    const float* x;      /* blocked input  */
    const float* w;      /* packed weights [n_w_slabs][cb_in][cout_pad][16] */
    const float* scale;  /* [cout_pad] folded BN scale (1 if none) */
    const float* shift;  /* [cout_pad] folded BN shift (0 if none) */
    const float* res;    /* optional residual: blocked, same logical shape as y, own strides (may be NULL) */
    float* y;            /* blocked output */
    int64_t x_n_stride, x_cb_stride, x_d_stride, x_h_stride; /* floats; voxel stride is 16 */
    int64_t y_n_stride, y_cb_stride, y_d_stride, y_h_stride; /* floats */
    int64_t y_off0;      /* float offset of logical output voxel (0,0,0) (skips the halo) */
    int64_t r_n_stride, r_cb_stride, r_d_stride, r_h_stride; /* residual geometry (floats), used when res != NULL */
    int64_t r_off0;
    int32_t N, OD, OH, OW;   /* logical grid iterated by one class */
    int32_t in_mul, out_mul; /* 1 or 2 */
    int32_t cb_in;           /* input channel blocks */
    int32_t cout_pad;        /* multiple of 16 */
    int32_t R, WT;           /* rows x cols of output handled by one wave (R*WT <= 112) */
    int32_t relu;
    int32_t n_classes;
    int32_t lds_bytes_per_wave; /* >= 2 * staged tile bytes of the largest class */
    int32_t reserved;
    drc_tap_class cls[DRC_MAX_CLASSES];
} drc_tapconv_params;

/* Validates, picks the (voxel-tiles, cout-tiles) instantiation and launches. */
int drc_tapconv_fwd(const drc_tapconv_params* p, void* stream);

/* The stride-1 3x3x3 class (one class, nd=nh=nw=3, unit spacing, in_mul=out_mul=1) with a sliding depth window -- a wave owns R x WT
 * voxels of ALL depth slices of a unit and applies every input slice to the three output slices it touches -- and both MFMA operands read
 * straight from global memory (no LDS): the B fragment of tap (kh,kw) is one coalesced float4 per lane (16 channels = four MFMA k-steps).
 * Weights are packed [27][cb_in][cout_pad][16] (engine.pack_weight_t16), NOT in the tap layout of drc_tapconv_fwd; cout_tiles_per_wave in
 * {1,2} must divide cout_pad/16; lds_bytes_per_wave is ignored.  (The LDS-staged predecessors of this and of the two kernels below --
 * tapslide.hip, tapdown.hip, tap2d.hip, rounds 1-2 -- left the library in round 6: no default plan selected them; attic/.) */
int drc_tapconv3d_direct_fwd(const drc_tapconv_params* p, int cout_tiles_per_wave, void* stream);

/* ConvTranspose3d(k3, s2, p1, op1) (+BN, +residual, +ReLU) -- hourglass.conv5/conv6 (stackhourglass.py:22-30) and the data
 * gradient of the stride-2 Conv3d layers -- with the 8 output-parity classes fused: a wave stages an (R+1) x (WT+1) tile
 * of input slices i and i+1 once per 8-channel phase and runs all 27 taps on it.  Takes the same parameter block as
 * drc_tapconv_fwd with the classes of the transposed convolution (in_mul = 1, out_mul = 2, OD/OH/OW = INPUT grid, cls[0].dd0/
 * dh0/dw0 = input halo); only x/w/scale/shift/res/y, the strides, N, OD, OH, OW, cb_in, cout_pad, R, WT, relu are read.
 * Needs R*WT <= 112 and ceil((R+1)*(WT+1)*2/64) <= 9.  Results equal drc_tapconv_fwd's up to the fp32 summation order. */
int drc_deconv3d_k3s2_fwd(const drc_tapconv_params* p, void* stream);

/* Conv3d(k3, stride 2, pad 1) (+BN, +residual, +ReLU) -- hourglass.conv1/conv3 (stackhourglass.py:9-16) and the data gradient of the
 * transposed convolutions -- with both MFMA operands read straight from global memory (no LDS): the stride only changes the per-lane address
 * of the float4 B fragment.  Parameter block of drc_tapconv_fwd for the single 3x3x3 class with in_mul = 2; cout_tiles_per_wave in {1,2,4}
 * must divide cout_pad/16; weights packed [27][cb_in][cout_pad][16] (engine.pack_weight_t16). */
int drc_conv3d_k3s2_direct_fwd(const drc_tapconv_params* p, int cout_tiles_per_wave, void* stream);

/* ConvTranspose3d(k3, s2, p1, output_padding 1) with all 8 output-parity classes from one register-resident set of B
 * fragments, both MFMA operands straight from global memory (deconvdirect.hip; reference stackhourglass.py:22-30).  Params as
 * for drc_deconv3d_k3s2_fwd (8 classes of engine.taps_deconv3d_k3s2, OD/OH/OW = INPUT dims, y exactly twice as large); R, WT are
 * ignored (a wave takes 32 consecutive voxels of the flattened (n,d,h,w) index); needs N * {x,y,r}_n_stride * 4 < 2^32;
 * cout_tiles_per_wave in {1, 2} dividing cout_pad/16; weights
 * [cb_in][27][cout_pad][16]: the t16 layout of the ConvTranspose weight [Cin,Cout,3,3,3] re-ordered channel-block-major with the
 * taps in the kernel's use order i = (a*3 + b)*3 + c -> tap ((K[a]*3 + K[b])*3 + K[c]), K = {1, 2, 0} (engine.pack_weight_deconv_direct). */
int drc_deconv3d_k3s2_direct_fwd(const drc_tapconv_params* p, int cout_tiles_per_wave, void* stream);
/* The same launch writing its result (also) as an RS16 tensor halfs [N][cout/32][2*OD+2][2*OH+2][8][2*OW+2][8] (round 5: the
 * consumer is the split-f16 convolution drc_conv3d_k3_s16_fwd); p->y may be NULL (RS16 output only); cout_pad % 32 == 0, two cout
 * tiles per wave.  ovf: the range-guard word of drc_s16conv_params.ovf (may be NULL). */
int drc_deconv3d_k3s2_direct_s16_fwd(const drc_tapconv_params* p, void* y16, uint32_t* ovf, void* stream);

/* The stride-1 3x3x3 convolution of drc_tapconv3d_direct_fwd (same parameter block; R, WT and lds_bytes_per_wave ignored) as
 * Winograd F(2x2x2, 3x3x3) on the fp32 matrix cores: 64 instead of 216 multiplies per (cin, cout) pair and 2x2x2 output tile.
 * Needs even OD, OH, OW and N * x_n_stride * 4 < 2^32; weights from drc_pack_weights_wino.  Results differ from the direct
 * kernels' by fp32 rounding only (about twice the direct kernel's own error against fp64). */
int drc_conv3d_k3_wino_fwd(const drc_tapconv_params* p, int cout_tiles_per_wave, void* stream);

/* The same convolution (same parameter block, bit-identical results) with TWO waves per SIMD: a block of 8 waves owns 64 consecutive
 * 2x2x2 tiles and stages, once per (depth frequency, channel block), the depth- and w-transformed input ROWS those tiles touch in LDS
 * (wino3d_rb.hip); waves read their tiles' rows from there.  For 28- and 14-wide maps (drc_conv3d_k3_wino_rb_supported), cout_pad a
 * multiple of 32; weights from drc_pack_weights_wino_rb: [64][ceil(Cin/16)][cout_pad/16][ch/4][cout%16][ch%4], cout padded to 32. */
int drc_conv3d_k3_wino_rb_supported(int cout_pad, int OD, int OH, int OW);
int drc_conv3d_k3_wino_rb_fwd(const drc_tapconv_params* p, void* stream);
int drc_pack_weights_wino_rb(const float* w, int cout, int cin, int transposed, int flip, float* out, void* stream);
/* The 2D form of the row-brick kernel: Conv2d 3x3 / stride 1 / pad 1 as Winograd F(2x2, 3x3) (same parameter block and results as
 * drc_conv2d_k3_wino_fwd: bit-identical), for 14-wide maps and widths that are multiples of 28 (the PSMNet feature CNN's 112- and
 * 56-wide maps); weights from drc_pack_weights_wino2d_rb: [16][ceil(Cin/16)][cout_pad/16][ch/4][cout%16][ch%4], cout padded to 32. */
int drc_conv2d_k3_wino_rb_supported(int cout_pad, int OH, int OW);
int drc_conv2d_k3_wino_rb_fwd(const drc_tapconv_params* p, void* stream);
int drc_pack_weights_wino2d_rb(const float* w, int cout, int cin, int transposed, int flip, float* out, void* stream);

/* Source of a cost volume that is never materialised: the two channel-blocked, zero-haloed 2D feature maps of N ROI pairs /
 * image pairs.  Voxel (n, cb, y, x) of a side lives at side + n*n_stride + cb*cb_stride + (y+pad)*h_stride + (x+pad)*16 (floats);
 * the first voxel of every (n, cb) plane must be a (zero) halo voxel, i.e. pad >= 1. */
typedef struct drc_costvol_src {
    const float* left;
    const float* right;
    int64_t n_stride, cb_stride, h_stride;
    int32_t cbi;   /* channel blocks per side (C/16) */
    int32_t pad;
    int32_t lo4;   /* disparity of volume slice 0 (mindisp/4) */
    int32_t Wp;    /* feature-map width W' */
} drc_costvol_src;

/* dres0[0] with the cost volume fused into its input loads (reference stackhourglass.py:115-130 + :63-66): the convolution of
 * drc_conv3d_k3_wino_fwd applied to cost[n, c, j, y, x] = (c < C ? left[n, c, y, x] : right[n, c-C, y, x-i]) if 0 <= x-i < W'
 * else 0, i = lo4 + j, without the volume ever reaching HBM.  p->x is ignored, p->cb_in = 2*cbi, p->OD = D' slices, p->OW = Wp;
 * the same weights, epilogue and limits as drc_conv3d_k3_wino_fwd (N * n_stride * 4 < 2^32).  Results are bit-identical to
 * drc_cost_volume_blocked_fwd followed by drc_conv3d_k3_wino_fwd. */
int drc_conv3d_k3_wino_costvol_fwd(const drc_tapconv_params* p, const drc_costvol_src* cv, int cout_tiles_per_wave, void* stream);
/* The same fused launch on the row-brick kernel (drc_conv3d_k3_wino_rb_fwd's shapes and weight packing): bit-identical results. */
int drc_conv3d_k3_wino_rb_costvol_fwd(const drc_tapconv_params* p, const drc_costvol_src* cv, void* stream);

/* Conv2d 3x3, stride 1, dilation 1, pad 1 (same parameter block as drc_conv2d_k3_direct_fwd; R, WT ignored) as Winograd
 * F(2x2, 3x3): 16 instead of 36 multiplies per (cin, cout) pair and 2x2 output tile.  Dilation d = cls[0].sh = sw > 1 (pad = d) is taken
 * when 2d divides OH and OW (d*d interleaved sub-grids).  Odd OH / OW (d = 1): the last tile row / column
 * keeps one output row / column and its input patch reaches one row past the zero halo (into the next plane; values there do not
 * reach a stored output), so the tensor must be followed by (W + 2 * pad + 2) * 64 readable bytes.  Needs
 * (N * x_n_stride + 2 * x_h_stride) * 4 < 2^32; weights from drc_pack_weights_wino2d ([16 = xh*4+xw][ceil(Cin/16)][cout_pad][16]). */
int drc_conv2d_k3_wino_fwd(const drc_tapconv_params* p, int cout_tiles_per_wave, void* stream);
int drc_pack_weights_wino2d(const float* w, int cout, int cin, int transposed, int flip, float* out, void* stream);

/* Winograd weight transform U = (G x G x G) g of a 3x3x3 kernel, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]:
 * w [Cout][Cin][27] (transposed: [Cin][Cout][27]; flip reverses the taps -- data gradients) ->
 * out [64 = (xd*4+xh)*4+xw][ceil(Cin/16)][cout_pad][16], zero-padded. */
int drc_pack_weights_wino(const float* w, int cout, int cin, int transposed, int flip, float* out, void* stream);

/* Conv2d(k3, stride 1 or 2, dilation d) (+BN/bias, +residual, +ReLU) with both MFMA operands read straight from global memory
 * (the 2D instantiation of drc_conv3d_k3s2_direct_fwd; stride = in_mul, dilation = tap spacing of the class).  Weights packed
 * [9][cb_in][cout_pad][16] (engine.pack_weight_t16). */
int drc_conv2d_k3_direct_fwd(const drc_tapconv_params* p, int cout_tiles_per_wave, void* stream);
/* The feature CNN's first layer straight from the dense image (stemconv.hip, round 4): x fp32 NCHW [N,3,H,W], Conv2d(3 -> cout, 3x3,
 * stride 2, pad 1) + folded BN (+ReLU) -> y in the channel-blocked fp32 layout (strides in floats, y_off0 = offset of the interior's first
 * element).  w_packed = [7 k-steps][cout_pad][4] floats, k = channel * 9 + tap, zero past k = 26 (engine.pack_weight_stem); cout_pad 16 or 32.
 * Replaces firstconv[0] of disprcnn/modeling/psmnet/submodule.py:65-66 together with the image's layout conversion. */
int drc_conv2d_k3s2_stem_fwd(const float* x, int N, int H, int W, const float* w_packed, int cout_pad, const float* scale, const float* shift,
                             float* y, int64_t y_n_stride, int64_t y_cb_stride, int64_t y_h_stride, int64_t y_off0, int OH, int OW, int relu,
                             void* stream);

/* Conv2d(k1, stride 1 or 2, pad 0) (+BN/bias, +residual, +ReLU) as a register-blocked MFMA GEMM with both operands read
 * straight from global memory (no LDS): the 1x1 convolutions of ResNet-50-FPN (backbone/resnet.py, backbone/fpn.py) and of the
 * PSMNet feature CNN (submodule.py).  Parameter block of drc_tapconv_fwd for the single 1x1 class (OD = 1); the weights are
 * packed [cb_in][cout_pad][16] (engine.pack_weight_pw), NOT in the tap layout. */
int drc_conv2d_k1_fwd(const drc_tapconv_params* p, void* stream);

/* Final classifier conv Conv3d(32->1,k3,p1,bias=False) (stackhourglass.py:78-88 `classifN[2]`)
 * with the cumulative head add (`+ cost_{k-1}`, :142-144) fused.
 *   x : blocked [N][cb_in][D+2][H+2][W+2][16];  w : [27][cb_in*16];  out,res : dense [N,D,H,W] */
int drc_conv3d_cout1_fwd(const float* x, const float* w, const float* res, float* out,
                         int N, int cb_in, int D, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------
 * a7. Fused trilinear(align_corners=True) upsample -> softmax over D -> soft-argmin.
 * Replaces F.interpolate + F.softmax + disparityregression (stackhourglass.py:169-173,
 * submodule.py:51-57) without materialising the [N,D,H,W] volume.
 *   cost : dense [N,Dp,Hp,Wp]   disp : [N,H,W]   D = maxdisp-mindisp, d = mindisp..maxdisp-1 */
int drc_upsample_softargmin_fwd(const float* cost, float* disp, int N, int Dp, int Hp, int Wp,
                                int D, int H, int W, int mindisp, void* stream);

/* Weight packing in one launch.  w: dense [Cout][Cin][K] (transposed = 0) or ConvTranspose [Cin][Cout][K] (transposed = 1); flip
 * reverses the tap order (data gradients).  out_tap: [K][cb_in][2][cout_pad][8] (drc_tapconv_fwd and the LDS-staged kernels);
 * out_t16: [K][cb_in][cout_pad][16] (the LDS-free kernels; K = 1: the 1x1 packing).  Either output may be NULL; channel padding is
 * written as zeros. */
int drc_pack_weights(const float* w, int cout, int cin, int K, int transposed, int flip, float* out_tap, float* out_t16, void* stream);

/* ---------------------------------------------------------------------------------------
 * Small helpers for the 2D feature CNN (submodule.py:76-139): average pool and bilinear
 * (align_corners=True) upsample on blocked 2D tensors, channel-offset aware (writes into a
 * slice of the concatenated 320-channel tensor). */
int drc_avgpool2d_blocked(const float* x, float* y, int N, int CB, int H, int W, int px,
                          int k, int OH, int OW, int py, void* stream);
/* same, reading channel blocks [x_cb_off, x_cb_off+CB) of an input that has x_cb_total blocks (output_skip lives inside
 * the 320-channel concat tensor) */
int drc_avgpool2d_blocked_slice(const float* x, float* y, int N, int CB, int H, int W, int px, int k, int OH, int OW, int py,
                                int x_cb_total, int x_cb_off, void* stream);
int drc_bilinear_up_blocked(const float* x, float* y, int N, int CB, int IH, int IW, int px,
                            int OH, int OW, int py, int y_cb_total, int y_cb_off, void* stream);
/* a12 helpers (ResNet-FPN): bilinear resize with either align_corners convention (the fork's FPN top-down path uses
 * F.interpolate(bilinear, align_corners=False), fpn.py:62-64) and max_pool2d(k, stride, pad 0, ceil_mode) with windows
 * clipped to the valid region (stem pool resnet.py:303; LastLevelMaxPool fpn.py:80-82). */
int drc_bilinear_resize_blocked(const float* x, float* y, int N, int CB, int IH, int IW, int px, int OH, int OW, int py,
                                int y_cb_total, int y_cb_off, int align_corners, void* stream);
int drc_maxpool2d_blocked(const float* x, float* y, int N, int CB, int H, int W, int px, int k, int stride, int OH, int OW, int py,
                          void* stream);
/* copy channel blocks of a blocked tensor into a channel slice of another (same spatial geometry) */
int drc_copy_blocks(const float* x, float* y, int N, int CB, int64_t vox_per_cb,
                    int y_cb_total, int y_cb_off, void* stream);

/* ---------------------------------------------------------------------------------------
 * f1. ROIAlign (the crop in front of the path) -- replaces disprcnn._C.roi_align_forward / roi_align_backward
 * (csrc/vision.cpp:7-15, csrc/ROIAlign.h:11-45, csrc/cpu/ROIAlign_cpu.cpp:113-219, csrc/cuda/ROIAlign_cuda.cu:64-122,177-254).
 *   input [B,C,H,W]; rois [K,5] = (batch index, x1, y1, x2, y2); out [K,C,PH,PW].
 * No rounding of roi coordinates, roi size clamped to >= 1, sampling grid = sampling_ratio or ceil(roi/pooled),
 * samples outside [-1, size] contribute 0.  mean/stdv (per channel, both or neither) fuse the ImageNet normalisation
 * of DispRCNN3D.crop_and_transform_roi_img (disprcnn3d.py:44-50) into the crop.
 * Backward accumulates with atomicAdd into grad_in, which the caller zero-fills. */
int drc_roi_align_fwd(const float* input, const float* rois, float* out, int K, int C, int H, int W, int PH, int PW,
                      float spatial_scale, int sampling_ratio, const float* mean, const float* stdv, void* stream);
/* The same operator over an FPN pyramid in ONE launch (round 3): roi k is pooled from level levels[k] (its [N,C,H,W] map, size and
 * scale in `pyr`), which replaces the per-level nonzero / index_select / roi_align / index_copy loop of modeling/poolers.py:118-149;
 * per-sample arithmetic identical to drc_roi_align_fwd. */
#define DRC_FPN_MAX_LEVELS 8
typedef struct {
    const float* feat[DRC_FPN_MAX_LEVELS];
    int32_t H[DRC_FPN_MAX_LEVELS], W[DRC_FPN_MAX_LEVELS];
    float scale[DRC_FPN_MAX_LEVELS];
    int32_t n_levels;
} drc_fpn_pyramid;
int drc_roi_align_fpn_fwd(const drc_fpn_pyramid* pyr, const float* rois, const int32_t* levels, float* out, int K, int C, int PH, int PW,
                          int sampling_ratio, void* stream);
int drc_roi_align_bwd(const float* grad_out, const float* rois, float* grad_in, int K, int C, int H, int W, int PH, int PW,
                      float spatial_scale, int sampling_ratio, void* stream);

/* Left/right ROI alignment of DispRCNN3D.prepare_psmnet_input_and_target (disprcnn3d.py:118-146) on the device:
 * expand_box_to_integer (stereo_utils.py:219-229), clamps, common width.
 *   left/right_boxes [R,4] xyxy, img_idx [R]  ->  rois_left/right [R,5] for drc_roi_align_fwd,
 *   geom [R,4] = (x1, x1p, x1+mw, x1p+mw)  (the x1s, x1ps, x2s, x2ps of the reference) */
int drc_align_roi_pairs(const float* left_boxes, const float* right_boxes, const int32_t* img_idx, int R, int img_w, int img_h,
                        float* rois_left, float* rois_right, int32_t* geom, void* stream);

/* fp16-STORAGE regressor (BASELINE configs[3]: 64 ROIs at 224x224x96, "fp16"): activations as
 * half[N][ceil(C/32)][D+2][H+2][W+2][32] (zero halo; strides and offsets of the params in ELEMENTS = halfs, voxel stride 32),
 * weights half [taps][ceil(Cin/32)][cout_pad][32], BN scale/shift fp32, fp32 accumulation on v_mfma_f32_16x16x32_f16.
 * Same tap-grid classes as drc_tapconv_fwd (Conv3d k3 stride 1|2, the 8 parity classes of ConvTranspose3d k3 s2); R*WT <= 64.
 * params.reserved = 1 ("dense1"): only cout 0 is produced, as dense fp32 y[N,OD,OH,OW] (+ dense fp32 res) -- the 32 -> 1
 * classifier conv with the cumulative head add (stackhourglass.py:78-88,142-144).  No reference counterpart (fp32-only). */
int drc_conv16_fwd(const drc_tapconv_params* p, void* stream);
/* The stride-1 3x3x3 / 3x3 (pad 1, one tap class) layers of the same parameter block with the input tile staged in LDS
 * (conv16t.hip): 16- or 8-row x 14-column output tiles of one slice per block, all taps read their voxel fragments from LDS; 3x3x3 layers
 * with one 32-channel input block, <= 32 couts and OD >= 4 take the depth-sliding form (weights in registers, each input slice staged once).
 * R, WT are ignored.  Reads up to 15 voxel lines past a plane's last padded row / column on ragged maps (values there do not reach a
 * stored output): the tensor must be followed by that much readable memory (engine.Blocked16's slack).  _supported: 1 if the
 * parameter block describes such a layer. */
int drc_conv16_k3_tile_supported(const drc_tapconv_params* p);
/* The rest of the fp16-storage 2D feature CNN (ops16.hip; BASELINE configs[3], the reference has no fp16 path): fp32 NCHW image ->
 * blocked fp16 [N][ceil(C/32)][H+2ph][W+2pw][32] (interior; the caller zero-fills the halo once); AvgPool2d(k,k) of a channel-block slice
 * and align_corners=True bilinear up-sampling into one (submodule.py:76-90,120-135); and the concat cost volume of
 * stackhourglass.py:115-128 from blocked fp16 feature maps (<= 32 channels = one block; unit n = left view, unit n + right_first_unit =
 * right view; feat_pad = their halo) into the blocked fp16 volume drc_cost_volume16_blocked_fwd writes. */
int drc_dense_to_blocked16(const float* x, void* y16, int N, int C, int H, int W, int ph, int pw, void* stream);
int drc_avgpool2d_blocked16_slice(const void* x16, void* y16, int N, int CB32, int H, int W, int px, int k, int OH, int OW, int py,
                                  int x_cb_total, int x_cb_off, void* stream);
int drc_bilinear_up_blocked16(const void* x16, void* y16, int N, int CB32, int IH, int IW, int px, int OH, int OW, int py, int y_cb_total,
                              int y_cb_off, void* stream);
int drc_cost_volume16_from16(const void* feat16, void* cost16, int N, int right_first_unit, int C, int Dp, int Hp, int Wp, int mindisp4,
                             int maxdisp4, int feat_pad, void* stream);
int drc_conv16_k3_tile_fwd(const drc_tapconv_params* p, void* stream);

/* The first 3D layer on the fp16 cost volume WITHOUT the volume (conv16x.hip, round 4): x = the fp16 feature pair
 * half[N][2: left, right][3][H+2][W+2][32] (drc_cost_volume16_blocked_fwd / drc_cost_volume16_from16 with one depth slice at disparity 0),
 * OD = the volume's depth, mindisp4 = its first disparity; cb_in = 2, weights as drc_conv16_k3_tile_fwd.  Bit-identical to
 * drc_cost_volume16_* followed by drc_conv16_k3_tile_fwd.  Replaces the concat loop of disprcnn/modeling/psmnet/stackhourglass.py:115-128
 * plus dres0[0] (:63-66) in the fp16-storage mode. */
int drc_conv16_k3_costvol_fwd(const drc_tapconv_params* p, int mindisp4, void* stream);

/* ---------------------------------------------------------------------------------------
 * a2-a6 (round 5).  The stride-1 3x3x3 convbn_3d layers in SPLIT-f16 arithmetic (convs16.hip): fp32 values carried as
 * hi + lo fp16 pairs, three v_mfma_f32_32x32x16_f16 products per fp32 product, fp32 accumulate -- fp32-class results
 * (error against fp64 of the size of the fp32 FMA chain's own) on the 2.5 PF f16 matrix cores instead of the 157 TF fp32 MFMA.
 * Replaces nn.Conv3d + BatchNorm3d (+ add, + ReLU) of disprcnn/modeling/psmnet/submodule.py:19-22 at the call sites
 * stackhourglass.py:63-70 (dres0, dres1), :78-88 (classif*[0]), :9-20 (hourglass conv2 / conv4); with left/right set also the
 * concat loop stackhourglass.py:115-128 in front of dres0[0].
 *
 * "RS16" tensors: halfs [N][C/32][D+2pd][H+2][8 chunks][W+2][8], zero halo stored (pd = 1 for volumes, 0 for 2D maps);
 * chunk q = p*4 + s*2 + g (p = 0 hi / 1 lo), element e of chunk (s, g) = channel 4g + 8(2s + (e>>2)) + (e&3) of the 32-channel block.
 * value = (float)hi + (float)lo, hi = fp16(value) (round to nearest), lo = fp16(value - hi); |value| <= 65504.
 * Weights: drc_s16_pack_weights layout [cout/32][cin/16][27 taps kd*9+kh*3+kw][hi, lo][64 lanes][8 halfs], pre-scaled by 2^wexp
 * (the caller folds 2^-wexp into scale).  cin, cout in {32, 64}; any D, H, W > 0 (round 6; the reference's contract at this resolution is
 * D, H, W = 0 mod 4, stackhourglass.py:115-128): W <= 7 runs on 4x7 MFMA tiles, W <= 14 on 2x14, wider maps on 1x28 -- the last x tile and
 * the last row tile are masked -- and a D that is not a multiple of 3 walks phantom zero planes behind the last one.  All 32 MFMA columns of a
 * tile and no phantom work at W = 7 | 14 | 0 mod 28, D = 0 mod 3 (BASELINE's configs).  Range: a stored value is clamped to +-65504 and
 * reported through `ovf` (below). */
typedef struct drc_s16conv_params {
    const void* x;       /* RS16 input [N][cin/32][D+2][H+2][8][W+2][8]; ignored when left/right are given */
    const void* w;       /* packed split weights */
    const float* scale;  /* [cout] folded BN scale * 2^-wexp */
    const float* shift;  /* [cout] folded BN shift */
    const void* res;     /* optional RS16 residual, geometry of y16 (may be NULL) */
    void* y16;           /* RS16 output [N][cout/32][D+2][H+2][8][W+2][8], or NULL when y32 is given */
    float* y32;          /* blocked fp32 output float[N][cout/16][D+2][H+2][W+2][16] INSTEAD of y16 (cin = 32, W % 28 == 0, no residual:
                            the cout-1 head's input); exactly one of y16 / y32 */
    const void* left;    /* cost-volume variant (cin = 64): RS16 2D maps [N][1][H+2][8][W+2][8] */
    const void* right;
    int32_t N, D, H, W;
    int32_t cin, cout, relu;
    int32_t lo4;         /* cost-volume variant: disparity of volume slice 0 (mindisp/4) */
    int32_t dil;         /* drc_conv2d_k3_s16_fwd only: dilation (0 or 1: none; 2) */
    float* head;         /* drc_conv3d_k3_s16_fwd, cin = cout = 32, W % 28 == 0, D >= 6, y16 = y32 = res = NULL: the layer is classif[0] of a head and
                            its output is not stored; the in-plane partial sums of the 32 -> 1 convolution behind it (classif[2],
                            stackhourglass.py:78-88) are: float [N][D][H][W][12], S[kh*3+kw] of the SOURCE voxel summed over the depth taps
                            (j 0..4 at floats 0..4, j 5..8 at 8..11), scaled by 2^wexp of w1.  drc_head_gather_fwd finishes the layer. */
    const void* w1;      /* with head: the 32 -> 1 weights as the MFMA A operand, halfs [2 K slices][hi, lo][64][8] (4 KiB) */
    uint32_t* ovf;       /* optional device word (round 6): OR-ed with 1 when a value this launch stores (or, with head, multiplies) left the
                            split-f16 range before the clamp -- |v| > 65504, Inf or NaN.  The reference is fp32 (config/defaults.py:22) and has no
                            such limit: the caller reads the word once per forward and re-runs on the fp32 kernels or raises.  NULL: no report. */
} drc_s16conv_params;
int drc_conv3d_k3_s16_supported(int cin, int cout, int D, int H, int W);
int drc_conv3d_k3_s16_fwd(const drc_s16conv_params* p, void* stream);
/* Round 6 (convs16w.hip): the cost-volume form (left / right set) with TWO MFMA tiles per wave -- a workgroup owns two rows of a column
 * instead of one: 2.0 instead of 3.0 staged rows per output row, half the barriers per MFMA; 6-8 % faster, bit-identical results.  (Measured
 * for the plain 32 -> 32 layer too: no gain; the residual and fused-head forms have no registers for a second accumulator set.)
 * drc_conv3d_k3_s16_wide (host code, no launch) says whether drc_conv3d_k3_s16_fwd sends a parameter block there: the cost-volume form,
 * W > 14, H even, at least 1024 two-row columns (smaller launches keep the finer columns); drc_conv3d_k3_s16_wide_fwd launches it for any
 * cost-volume block. */
int drc_conv3d_k3_s16_wide(const drc_s16conv_params* p);
int drc_conv3d_k3_s16_wide_fwd(const drc_s16conv_params* p, void* stream);
/* The second half of a fused head (p->head above): cost[n][z][y][x] = (res ? res[..] : 0) + scale * sum_{kh,kw} S[n][z][y+kh-1][x+kw-1][kh*3+kw]
 * (sources outside the volume contribute zero: the convolution's zero padding); cost, res: dense float [N][D][H][W] (res = the previous
 * head's cost, stackhourglass.py:142-144, or NULL); scale = 2^-wexp of the packed 32 -> 1 weights. */
int drc_head_gather_fwd(const float* S, const float* res, float* cost, int N, int D, int H, int W, float scale, void* stream);
/* The hourglass' other 3x3x3 layers in the same arithmetic and layout (x, y16, res: RS16; no blocked fp32 output, no cost-volume form):
 *   drc_conv3d_k3s2_s16_fwd   -- Conv3d k3 s2 p1 + BN + ReLU (convs16d.hip; reference hourglass conv1 / conv3, stackhourglass.py:9-12,17-19).
 *                                D, H, W = the INPUT dims (even); y16 has (D/2, H/2, W/2); no residual.
 *   drc_deconv3d_k3s2_s16_fwd -- ConvTranspose3d k3 s2 p1 output_padding 1 + BN (+ residual, + ReLU) (convs16u.hip; reference hourglass
 *                                conv5 / conv6, stackhourglass.py:22-30,44-49).  D, H, W = the INPUT dims; y16 and res have (2D, 2H, 2W);
 *                                cin = 64; weights: drc_s16 packing of the ConvTranspose weight with its first two axes swapped
 *                                ([Cout, Cin, 3,3,3], tap kd*9+kh*3+kw of o = 2i - 1 + k, no flip).
 * Shapes: any (stride 2: even input dims); the narrower map of the layer picks the tile -- <= 7 voxels wide: 4x7, <= 14: 2x14, else 1x28 -- last tiles masked. */
int drc_conv3d_k3s2_s16_supported(int cin, int cout, int D, int H, int W);
int drc_conv3d_k3s2_s16_fwd(const drc_s16conv_params* p, void* stream);
int drc_deconv3d_k3s2_s16_supported(int cin, int cout, int D, int H, int W);
int drc_deconv3d_k3s2_s16_fwd(const drc_s16conv_params* p, void* stream);
/* The 2D member of the family (convs16r.hip, round 5): Conv2d k3 s1 p1 d1 + BN (+ residual, + ReLU) on RS16 2D maps
 * halfs [N][C/32][H+2][8][W+2][8] -- reference convbn of disprcnn/modeling/psmnet/submodule.py:9-16 at firstconv[2], firstconv[4] and the
 * BasicBlocks of layer1 / layer2 / layer3 (submodule.py:40-60, 68-95).  Uses the fields x, w, scale, shift, res, y16, N, H, W, cin, cout,
 * relu of the drc_s16conv_params block -- D is ignored, y32 / left / right must be NULL, lo4 = 0.  Weights: the drc_s16 packing with 9 taps kh*3+kw:
 * [cout/32][cin/16][9][hi, lo][64][8].  cin in {32, 64, 128} (wider layers: the caller chains launches over 128-channel slices, each adding
 * the previous partial sum as its residual), cout a power of two in 32..512, any H, W > 0 (ragged last x group / row block masked).  dil = 2
 * (padding 2; layer4, submodule.py:73): cin = 128, H % 56 == 0.  Range: as for the 3D kernels, a stored value is clamped to +-65504 and
 * reported through p->ovf -- the callers (PSMNet's 2D CNN, the ResNet-FPN trunk and the RPN head under `auto`) run inside engine.guarded, which
 * reads the word once per forward pass and repeats the pass on the fp32 kernels when it is set. */
int drc_conv2d_k3_s16_supported(int cin, int cout, int H, int W, int dil);
int drc_conv2d_k3_s16_fwd(const drc_s16conv_params* p, void* stream);
/* RS16 converters (s16_ops.hip): interior only, the zero halo is the allocator's.  dense = NCDHW fp32 (D = 1, pd = 0 for 2D maps);
 * blocked = the engine's fp32 blocked tensor, channel blocks [cb16_off, cb16_off + C/16) of a tensor with cb16_total blocks and halos
 * (pd_in, ph_in, pw_in).  They stand where the reference hands fp32 NCHW features to the concat loop, stackhourglass.py:112-128.
 * ovf (the two fp32 -> RS16 converters; may be NULL): the range-guard word of drc_s16conv_params.ovf -- set when an input value is outside
 * [-65504, 65504] (it is stored clamped), Inf or NaN. */
int drc_rs16_from_dense(const float* x, void* y16, int N, int C, int D, int H, int W, int pd, uint32_t* ovf, void* stream);
int drc_rs16_from_blocked(const float* xb, void* y16, int N, int C, int D, int H, int W, int pd_in, int ph_in, int pw_in, int cb16_total,
                          int cb16_off, int pd, uint32_t* ovf, void* stream);
int drc_rs16_to_dense(const void* y16, float* x, int N, int C, int D, int H, int W, int pd, void* stream);
int drc_rs16_to_blocked(const void* y16, float* xb, int N, int C, int D, int H, int W, int pd_out, int ph_out, int pw_out, int cb16_total,
                        int cb16_off, int pd, void* stream);
/* Round 4: the same recipe for the stride-2 and the transposed 3x3x3 layers of the fp16-storage regressor (conv16x.hip; hourglass conv1 /
 * conv3 and conv5 / conv6, stackhourglass.py:11-30): drc_conv16_k3s2_tile_* takes the single-class stride-2 grid (in_mul = 2, canonical
 * weight order), drc_deconv16_k3s2_tile_* the eight output-parity classes of ConvTranspose3d(k3, s2, p1, op1) (out_mul = 2, the classes in
 * (pd, ph, pw) order); *_supported returns 1 when the parameter block is one they take, else the caller uses drc_conv16_fwd. */
int drc_conv16_k3s2_tile_supported(const drc_tapconv_params* p);
int drc_conv16_k3s2_tile_fwd(const drc_tapconv_params* p, void* stream);
int drc_deconv16_k3s2_tile_supported(const drc_tapconv_params* p);
int drc_deconv16_k3s2_tile_fwd(const drc_tapconv_params* p, void* stream);
/* fp32 features (NCHW if in_blocked_pad < 0, else the fp32 blocked 2D layout with that halo) -> fp16 blocked cost volume
 * [N][2][Dp+2][Hp+2][Wp+2][32] (block 0 = left, block 1 = shifted right; stackhourglass.py:115-128); C <= 32. */
int drc_cost_volume16_blocked_fwd(const float* left, const float* right, void* cost16, int N, int C, int Dp, int Hp, int Wp,
                                  int mindisp4, int maxdisp4, int in_blocked_pad, void* stream);

/* f4. Greedy NMS -- replaces disprcnn._C.nms for GPU tensors (reference csrc/nms.h:12-28 -> csrc/cuda/nms.cu:23-131; CPU twin
 * csrc/cpu/nms_cpu.cpp:5-75).  boxes_sorted [n,4] xyxy in DESCENDING score order (the caller sorts: torch.sort is plumbing);
 * IoU uses the legacy +1 pixel convention; a box is suppressed by an earlier kept box when IoU > thresh (strict = 1, the CUDA
 * op's test) or >= thresh (strict = 0, the CPU op's).  mask_ws: n * ceil(n/64) uint64 of scratch; keep [n] u8 out (1 = kept).
 * n <= 32768.  The greedy walk runs on the device (the reference copies the mask to the host). */
int drc_nms_sorted_fwd(const float* boxes_sorted, int n, float thresh, int strict, uint64_t* mask_ws, uint8_t* keep, void* stream);
/* The same for `sets` box sets of n boxes each in one launch pair (boxes_sorted [sets][n][4], mask_ws [sets][n*ceil(n/64)], keep [sets][n]):
 * the two views of double_view_boxlist_nms (reference structures/boxlist_ops.py:49-79) share their scores, hence their order. */
int drc_nms_sorted_batch_fwd(const float* boxes_sorted, int sets, int n, float thresh, int strict, uint64_t* mask_ws, uint8_t* keep, void* stream);
/* The two views of a stereo list (boxes_sorted [2][n][4], shared descending scores) walked together: keep_joint[n] = 1 where BOTH views
 * keep the pair (what double_view_boxlist_nms intersects, structures/boxlist_ops.py:49-79), and the walk stops once max_keep pairs are
 * kept (<= 0: all) -- rows after that chunk are 0, so the first max_keep set flags are exactly the reference's keep[:max_proposals]
 * for score-sorted input.  n <= 32,768; mask_ws: 2 * n * ceil(n/64) words. */
int drc_nms_sorted_pair_joint_fwd(const float* boxes_sorted, int n, float thresh, int strict, int max_keep, uint64_t* mask_ws,
                                  uint8_t* keep_joint, void* stream);

/* f4. Box arithmetic of the 2D detection stage (det_ops.hip).
 * drc_box_decode_fwd -- BoxCoder.decode (reference modeling/box_coder.py:161-244): codes [rows][groups][per_group], per_group = 4
 *   (dx,dy,dw,dh) or 6 (+ dx', dw' of the right view, decoded against the SAME reference box: decode_..._fromboxes4), boxes
 *   [rows][4] xyxy with the legacy +1 widths; weights4 = (wx, wy, ww, wh); dw/dh clamped to xform_clip (log(1000/16)); with
 *   img_w > 0 the result is clipped to [0, img_w-1] x [0, img_h-1] (BoxList.clip_to_image, structures/bounding_box.py:317-325).
 * drc_srpn_proposals_fwd -- one FPN level of the Stereo RPN from the head's dense maps to per-anchor proposals (reference
 *   modeling/rpn/stereo_rpn/srpn.py:41-50 pairwise softmax; stereo_rpn/inference.py:121-150 flattening, decode, left = codes
 *   0..3, right = (4,1,5,3), clip_boxes :287-299): logits [N][2A][H][W] (RAW cls_logits output), regression [N][6A][H][W],
 *   anchors [H*W*A][4] (position-major), image_wh [N][2]; writes scores / left / right at [n][level_offset + (y*W+x)*A + a] of
 *   [N][total_anchors] / [N][total_anchors][4] buffers, so the five levels land concatenated like the reference's torch.cat. */
int drc_box_decode_fwd(const float* codes, const float* boxes, float* out, int64_t rows, int groups, int per_group, const float* weights4,
                       float xform_clip, float img_w, float img_h, void* stream);
int drc_srpn_proposals_fwd(const float* logits, const float* regression, const float* anchors, const float* image_wh, int N, int A, int H,
                           int W, int64_t total_anchors, int64_t level_offset, float xform_clip, float* scores, float* left, float* right,
                           void* stream);

/* Training targets of the disparity stage -- replaces the per-ROI host loop of DispRCNN3D.prepare_psmnet_input_and_target
 * (reference modeling/detector/disprcnn3d.py:52-112) incl. Masker(thresh, padding) (roi_heads/mask_head/inference.py:90-190) and
 * DisparityMap.crop / .resize (structures/disparity.py:38-77).
 *   disp_maps [B,H,W] f32 ground-truth disparity, gt_masks [B,H,W] u8 (union of the instance masks, 0/1),
 *   mask_probs [R,mask_size,mask_size] f32 (the 2D stage's 'mask' field), det_boxes [R,4] the left detections (xyxy, float),
 *   rois_left [R,5] and geom [R,4] as written by drc_align_roi_pairs
 *   -> targets [R,res,res] f32 (ROI-normalised disparity, offset x1-x1p removed, scaled by res/width), masks [R,res,res] u8. */
int drc_roi_train_targets_fwd(const float* disp_maps, const uint8_t* gt_masks, const float* mask_probs, int mask_size, int padding,
                              float mask_thresh, const float* det_boxes, const float* rois_left, const int32_t* geom, int R, int H, int W,
                              int res, float* targets, uint8_t* masks, void* stream);

/* ---------------------------------------------------------------------------------------
 * f2. Post-processing: per-ROI disparities [R][S][S] -> full-image maps, replacing DisparityMapProcessor
 * (modeling/psmnet/inference.py:18-47), DispRCNN3D.roi_disp_postprocess (modeling/detector/disprcnn3d.py:161-190) and the
 * per-ROI depth maps of PointRCNN.process_input (pointnet_module/point_rcnn/lib/net/point_rcnn.py:121-133).
 *   boxes [R][6] int32 = x1, y1, x2, y2 of the left box and x1p, x2p of the right box, all after expand_box_to_integer;
 *   value(r, y, x) inside the left box = bilinear(align_corners) resize of map r to (y2-y1) x max(x2-x1, x2p-x1p),
 *   times that width / S, plus x1 - x1p (structures/disparity.py:38-77); boxes reaching outside the image are clipped
 *   (the reference raises a shape error there).
 * drc_disparity_paste_fwd: roi_offsets [B+1] (CSR: the ROIs of image b), out [B][H][W] = max over the image's ROIs of
 *   (inside ? value : 0), 0 without ROIs -- so one ROI keeps its negative values, several clamp at 0 outside their overlap,
 *   exactly like the reference's max over stacked zero images.  flags bit 0: clamp each value at 0 first; mask (optional,
 *   [R][H][W] float) multiplies after the clamp (roi_disp_postprocess).
 * drc_roi_depth_maps_fwd: out [R][H][W] = inside ? fuxb[r] / (value + 1e-6) : 0. */
int drc_disparity_paste_fwd(const float* disp, int S, const int32_t* boxes, const int32_t* roi_offsets, int B, int H, int W, int flags,
                            const float* mask, float* out, void* stream);
int drc_roi_depth_maps_fwd(const float* disp, int S, const int32_t* boxes, const float* fuxb, int R, int H, int W, float* out, void* stream);

/* DisparityMap.resize of a whole map (reference structures/disparity.py:39-62, called by disprcnn3d.py:89-91,173-174 and
 * tools/kitti_object/generate_psmnet_input_inf.py:99-104): src [IH,IW] -> dst [OH,OW], values times OW / IW.
 *   mode 0: bilinear, align_corners=True (F.interpolate);  mode 1: use_max_pooling=True -- adaptive max pooling of the positive part minus
 *   adaptive max pooling of the negated negative part. */
int drc_disparity_resize_fwd(const float* src, int IH, int IW, float* dst, int OH, int OW, int mode, void* stream);

/* ---------------------------------------------------------------------------------------
 * f4. Fully connected layers of the 2D stage's heads (roi_heads/box_head/roi_box_feature_extractors.py:85-130: the 7x7/stride-7
 *   convolution on 7x7 ROI features = a 25088 -> 2048 FC, then 2048 -> 2048; roi_box_predictors.py; the mask predictor's 2x2/stride-2
 *   transposed convolution and 1x1 logits): y[M][N] = act(x[M][K] . w[N][K]^T + bias[N]) as a hand-written fp32-MFMA GEMM (linear.hip).
 *   x and w are PACKED operands: drc_linear_pack_rows turns a row-major [R][K] matrix (K contiguous) into [R/16][K/16][k%16/4][r%16][k%4],
 *   zero-padded to whole 16 x 16 blocks (drc_linear_packed_floats floats), so that a wavefront's operand load is one contiguous KiB; pack
 *   the weights once per parameter version, the activations per call.  bias may be NULL.  K is split over several waves per output tile
 *   when M x N alone cannot fill the chip: scratch must then hold drc_linear_scratch_floats(M, N, K) floats of partial tiles, which a
 *   second launch adds in split order (no atomics). */
int64_t drc_linear_packed_floats(int R, int K);
int drc_linear_pack_rows(const float* a, int R, int K, float* out, void* stream);
int64_t drc_linear_scratch_floats(int M, int N, int K);
int drc_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int relu, float* scratch,
                   int64_t scratch_floats, void* stream);

/* ---------------------------------------------------------------------------------------
 * a10. PSMLoss / EndPointErrorLoss (utils/loss_utils.py:9-32, utils/stereo_utils.py:185-208).
 *   sums5 = { sum m*smoothl1(p1-t), sum m*smoothl1(p2-t), sum m*smoothl1(p3-t), sum m, sum m*|p1-t| }   (overwritten)
 *   (pred2/pred3 may be NULL for the eval form).  scratch: DRC_LOSS_SCRATCH_FLOATS floats of per-block partials, added in
 *   block order by a finishing launch -- no atomics, bit-reproducible.  The scalar loss is assembled by the caller:
 *   train: 0.5*s0/s3 + 0.7*s1/s3 + s2/s3 (division skipped when s3 == 0); eval: s4/s3 (0 when s3 == 0).
 *   grad:  grad_pred = grad_scale[0] * weight * m * clamp(pred - t, -1, 1) / s3 */
#define DRC_LOSS_SCRATCH_FLOATS ((size_t)1024 * 8)
int drc_psm_loss_sums(const float* pred1, const float* pred2, const float* pred3, const float* target, const uint8_t* mask,
                      int64_t numel, float* sums5, float* scratch, void* stream);
int drc_psm_loss_grad(const float* pred, const float* target, const uint8_t* mask, int64_t numel, const float* sums5, float weight,
                      const float* grad_scale, float* grad_pred, void* stream);

/* ---------------------------------------------------------------------------------------
 * a2 (train mode). BatchNorm with per-GPU batch statistics on blocked tensors (nn.BatchNorm3d/2d of convbn_3d / convbn,
 * submodule.py:13-22).  Every `geom` argument of the BatchNorm / SPP-backward entry points is
 * int[10] = {N, CB, D, H, W, pd, ph, pw, cb_total, cb_off}: channel blocks [cb_off, cb_off+CB) of a blocked tensor that
 * has cb_total blocks (a plain tensor has cb_total = CB, cb_off = 0; a concat slice addresses its parent).
 *   drc_bn_stats_blocked : stats[0][c] = mean_c, stats[1][c] = sum (x - mean_c)^2 over the interior voxels (stats [2][CB*16]),
 *                          in ONE pass: per-thread sums around the thread's first value, merged pairwise (Chan) in a fixed
 *                          order lane -> wave -> block -> launch, so the variance is cancellation-free and the statistics (and
 *                          with them every ReLU mask downstream) are bit-reproducible run to run.  `scratch`:
 *                          DRC_BN_SCRATCH_FLOATS(CB) floats whose last CB*32*33 words (arrival counters, one per 128-byte
 *                          line) are zero on entry; they are zero again on exit.  One scratch per stream.
 *   drc_bn_apply_blocked : y = act((x - mean) * invstd * gamma + beta (+ res)), interior only */
#define DRC_BN_MAX_CHUNKS 512
#define DRC_BN_SCRATCH_FLOATS(CB) ((size_t)DRC_BN_MAX_CHUNKS * (CB) * 32 + (size_t)(CB) * 32 * 33)
int drc_bn_stats_blocked(const float* x, const int* geom8, float* stats, float* scratch, void* stream);
/* invstd[c] = rsqrt(stats[1][c]/count + eps) for all C16 (padded) channels, and, if running_mean != NULL, the nn.BatchNorm
 * running-statistics update of the first C channels (momentum, unbiased batch variance); *num_batches_tracked += 1 if non-NULL */
int drc_bn_finalize(const float* stats, int C16, int C, long long count, float eps, float momentum, float* running_mean, float* running_var,
                    long long* num_batches_tracked, float* invstd, void* stream);
int drc_bn_apply_blocked(const float* x, const int* geom_x, float* y, const int* geom_y, const float* res, const int* geom_r,
                         const float* mean, const float* invstd, const float* gamma, const float* beta, int relu, void* stream);

/* ---------------------------------------------------------------------------------------
 * Backward kernels (autograd of stackhourglass.py:130-174 in the reference).  Data gradients of the MFMA convolutions
 * reuse drc_tapconv_fwd / drc_tapconv3d_direct_fwd / drc_deconv3d_k3s2_direct_fwd with transformed weights
 * (disprcnn_amd/modeling/psmnet/train.py).
 *   drc_upsample_softargmin_bwd : grad_cost [N,Dp,Hp,Wp] = d disp / d cost * grad_disp [N,H,W]  (overwritten).  scratch:
 *        drc_upsample_softargmin_bwd_scratch_floats(...) floats -- every 8 x 16 pixel tile stores the gradient of its coarse
 *        footprint there and a second launch gathers, per coarse cell, the tiles that touch it in tile order (no atomics)
 *   drc_conv3d_cout1_bwd_data   : grad of the 32->1 classifier conv w.r.t. its blocked input (assign or accumulate)
 *   drc_conv3d_cout1_bwd_weight : grad_w [27][cb_in*16] = sum x * grad_out  (overwritten).  scratch:
 *        DRC_COUT1_WGRAD_SCRATCH_FLOATS(cb_in) floats of per-block partials, added in block order by a second launch
 *   drc_bn_bwd_reduce / _apply  : training-mode BatchNorm backward with the ReLU mask and the residual fan-out fused:
 *        dz = dy*[y>0];  sums = {sum dz, sum dz*xhat};  draw = gamma*invstd*(dz - sums0/M - xhat*sums1/M);  dres (=|+=) dz */
int64_t drc_upsample_softargmin_bwd_scratch_floats(int N, int Dp, int Hp, int Wp, int H, int W);
int drc_upsample_softargmin_bwd(const float* cost, const float* grad_disp, float* grad_cost, int N, int Dp, int Hp, int Wp, int D, int H,
                                int W, int mindisp, float* scratch, int64_t scratch_floats, void* stream);
int drc_conv3d_cout1_bwd_data(const float* grad_out, const float* w, float* grad_x_blk, int N, int cb_in, int D, int H, int W,
                              int accumulate, void* stream);
#define DRC_COUT1_WGRAD_SCRATCH_FLOATS(cb_in) ((size_t)1024 * (cb_in) * 27 * 16)
int drc_conv3d_cout1_bwd_weight(const float* x_blk, const float* grad_out, float* grad_w, int N, int cb_in, int D, int H, int W,
                                float* scratch, void* stream);
int drc_bn_bwd_reduce(const float* dy, const int* geom_dy, const float* y, const int* geom_y, const float* raw, const int* geom_raw,
                      const float* mean, const float* invstd, int relu, float* sums, float* scratch /* as drc_bn_stats_blocked */, void* stream);
int drc_bn_bwd_apply(const float* dy, const int* geom_dy, const float* y, const int* geom_y, const float* raw, const int* geom_raw,
                     const float* mean, const float* invstd, const float* gamma, const float* sums, float inv_count, int relu, float* draw,
                     const int* geom_draw, float* dres, const int* geom_dres, int dres_accumulate, void* stream);

/* Weight gradient of a tap-grid convolution (one Cartesian tap class) on the fp32 MFMA:
 *   gw[ca][cb][t] += sum_{n,o} a[n, ca, in_mul*o + tap_t] * b[n, cb, o],   t = (td*nh + th)*nw + tw
 * a: blocked tensor the taps slide over (a_* strides in floats, tap offsets dd0/dh0/dw0 + k*sd/sh/sw in padded coordinates);
 * b: blocked tensor read at the plain positions o (b_off0 = float offset of logical voxel (0,0,0));
 * gw: dense fp32 [cb_a*16][cb_b*16][nd*nh*nw]; the launch ADDS into it (zero-fill it for a plain gradient).
 * Conv: a = layer input, b = grad of the conv output.  ConvTranspose k3 s2: a = grad of the output (in_mul = 2), b = input.
 * scratch (optional): DRC_WGRAD_SCRATCH_FLOATS floats of per-stream workspace.  With it every wave stores its partial sums
 * with plain coalesced stores and a second kernel adds them in wave order (bit-reproducible run to run); without it (NULL or
 * scratch_floats too small) the waves flush with atomicAdd -- measured 5.1 ms of the 27.9 ms 64-ROI Config-A train step. */
#define DRC_WGRAD_SCRATCH_FLOATS ((size_t)(1024 + 3 * 64) * 49 * 256)
typedef struct drc_wgrad_params {
    const float* a;
    const float* b;
    float* gw;
    int64_t a_n_stride, a_cb_stride, a_d_stride, a_h_stride;
    int64_t b_n_stride, b_cb_stride, b_d_stride, b_h_stride, b_off0;
    int32_t N, OD, OH, OW;      /* grid of b (positions o) */
    int32_t in_mul, cb_a, cb_b;
    int32_t nd, nh, nw, dd0, dh0, dw0, sd, sh, sw;
    int32_t R, WT, lds_bytes_per_wave;   /* >= ((rows_in*seg_vox) + R*WT) * 64 */
    int32_t overwrite;          /* 1: gw holds garbage -- the launch stores the sums instead of adding them */
    float* scratch;
    int64_t scratch_floats;
} drc_wgrad_params;
int drc_tapconv_wgrad(const drc_wgrad_params* p, void* stream);

/* Adjoints of the SPP helpers (submodule.py:76-90,120-135): bilinear(align_corners=True) upsampling (atomicAdd scatter
 * into grad_x) and AvgPool2d(k,k) (gather, accumulates into grad_x). */
int drc_bilinear_up_blocked_bwd(const float* grad_y, const int* geom_y, float* grad_x, const int* geom_x, void* stream);
int drc_avgpool2d_blocked_bwd(const float* grad_y, const int* geom_y, float* grad_x, const int* geom_x, int k, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DISPRCNN_HIP_H */
