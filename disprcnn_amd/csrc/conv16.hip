// conv16.hip -- fp16-STORAGE tap convolutions on the f16 matrix cores with fp32 accumulation (gfx950 / CDNA4):
// the 3D regressor of BASELINE configs[3] (64 ROIs at 224x224x96, "fp16").
//
//   reference arithmetic: convbn_3d / ConvTranspose3d of stackhourglass.py:7-51,63-88 (the reference itself is fp32-only,
//   config/defaults.py:22 -- this path trades precision for footprint and is held to a stated bound against the fp32 oracle).
//
// Layout: half[N][CB32 = ceil(C/32)][D+2][H+2][W+2][32] -- the fp32 layout's geometry with 32 fp16 channels in the 64-byte
// voxel line, zero halo.  One `v_mfma_f32_16x16x32_f16` contracts a whole 32-channel block of one tap:
//   A = weights [16 cout x 32 cin]  lane (cout j, g): channels 8g..8g+7   (one 16-byte load from [tap][cb32][cout][32])
//   B = voxels  [32 cin  x 16 vox]  lane (voxel j, g): channels 8g..8g+7  (one 16-byte load of the voxel line)
//   D[cout][voxel]: lane (voxel j, g) holds couts 4g..4g+3 in fp32 -> BN scale/shift (+residual) (+ReLU) in fp32 -> 4 halfs.
// Both operands come straight from global memory (no LDS), one tap step ahead; a wave owns R x WT <= VT*16 output voxels of one
// output slice and CT*16 couts, and walks (channel block, tap).  The same kernel serves Conv3d k3 stride 1 and 2 and the eight
// output-parity classes of ConvTranspose3d k3 s2 p1 op1 (tap-grid classes, engine.taps_*), and -- `dense1` mode -- the 32 -> 1
// classifier conv, whose single cout leaves as a dense fp32 [N,D,H,W] volume with the cumulative head add fused.
// With 16x the fp32 MFMA rate the kernel is bound by the vector-memory path, not the matrix cores; it is sized for the stress
// shape's footprint (half the bytes per activation), not tuned to a roofline.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define C16_WAVES 4
#ifndef C16_PF_BIG
#define C16_PF_BIG 4
#endif
#ifndef C16_PF_SMALL
#define C16_PF_SMALL 6
#endif

namespace {

template <int VT, int CT>
__global__ __launch_bounds__(64 * C16_WAVES) void conv16_kernel(const drc_tapconv_params p) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const int g = lane >> 4;
    const _Float16* x = (const _Float16*)p.x;
    const _Float16* w = (const _Float16*)p.w;
    const bool dense1 = p.reserved == 1;

    const int n_wt = (p.OW + p.WT - 1) / p.WT, n_rt = (p.OH + p.R - 1) / p.R;
    const int n_cg = p.cout_pad / 16 / CT;
    const long tiles = (long)p.N * p.OD * n_rt * n_wt;
    const long groups = tiles * n_cg * p.n_classes;
    const int nslots = p.R * p.WT;
    const long w_cb = (long)p.cout_pad * 32;           // halfs per (tap, cb32)
    const long w_tap = w_cb * p.cb_in;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, -1, 0x00020000);

    for (long grp = (long)blockIdx.x * C16_WAVES + wave; grp < groups; grp += (long)gridDim.x * C16_WAVES) {
        long t = grp;
        const int cg = (int)(t % n_cg); t /= n_cg;                 // cout group fastest: the waves of a block share the voxels
        const int ci = (int)(t % p.n_classes); t /= p.n_classes;
        const int wt_i = (int)(t % n_wt); t /= n_wt;
        const int rt_i = (int)(t % n_rt); t /= n_rt;
        const int od = (int)(t % p.OD);
        const int n = (int)(t / p.OD);
        const drc_tap_class cls = p.cls[ci];
        const int oh0 = rt_i * p.R, ow0 = wt_i * p.WT, ct0 = cg * CT;

        // Operands come through buffer loads (round 3): descriptor in SGPRs, a loop-invariant 32-bit lane offset and the (channel
        // block, tap) step as a scalar offset, so the step loop holds no vector address arithmetic.  The base of the voxel descriptor
        // is the group's first input voxel at tap offset 0; lane offset = slot (vt, j) inside the tile, channels 8g..8g+7.
        unsigned lane_vo[VT];
        bool valid[VT];
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) {
            const int s = vt * 16 + j;
            int r = s / p.WT, c = s - r * p.WT;
            valid[vt] = s < nslots && oh0 + r < p.OH && ow0 + c < p.OW;
            if (!valid[vt]) { r = 0; c = 0; }
            lane_vo[vt] = 2u * (unsigned)(r * p.in_mul * (int)p.x_h_stride + c * p.in_mul * 32 + g * 8);
        }
        const _Float16* xg = x + (long)n * p.x_n_stride + (long)(od * p.in_mul) * p.x_d_stride + (long)(oh0 * p.in_mul) * p.x_h_stride +
                             (long)(ow0 * p.in_mul) * 32;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, -1, 0x00020000);
        const unsigned wlo = 2u * (unsigned)((ct0 * 16 + j) * 32 + g * 8);

        f32x4 acc[VT][CT];
#pragma unroll
        for (int vt = 0; vt < VT; ++vt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[vt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

        const int ntap = cls.nd * cls.nh * cls.nw;
        const int steps = ntap * p.cb_in;
        // ring of NB operand sets, PF steps ahead of the MFMAs: a step is only VT*CT 16-cycle MFMAs, a load round trip ~2000 cycles
        constexpr int PF = VT * CT >= 8 ? C16_PF_BIG : C16_PF_SMALL;
        constexpr int NB = PF + 1;
        f16x8 B[NB][VT], Wt[NB][CT];
        int f_cb = 0, f_a = 0, f_b = 0, f_d = 0;       // (channel block, tap) of the next step to request; past the end: the last step again
        auto fetch = [&](int set) __attribute__((always_inline)) {
            const unsigned xo = 2u * (unsigned)(f_cb * (int)p.x_cb_stride + (cls.dd0 + f_a * cls.sd) * (int)p.x_d_stride +
                                                (cls.dh0 + f_b * cls.sh) * (int)p.x_h_stride + (cls.dw0 + f_d * cls.sw) * 32);
            const unsigned wo = 2u * (unsigned)((cls.wbase + f_a * cls.wsd + f_b * cls.wsh + f_d * cls.wsw) * (int)w_tap + f_cb * (int)w_cb);
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) B[set][vt] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(xr, lane_vo[vt], xo, 0));
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) Wt[set][ct] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, wlo + ct * 1024, wo, 0));
            if (f_cb * ntap + (f_a * cls.nh + f_b) * cls.nw + f_d + 1 < steps) {
                if (++f_d == cls.nw) { f_d = 0; if (++f_b == cls.nh) { f_b = 0; if (++f_a == cls.nd) { f_a = 0; ++f_cb; } } }
            }
        };
        auto mfma_step = [&](int set) __attribute__((always_inline)) {
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[vt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wt[set][ct], B[set][vt], acc[vt][ct], 0, 0, 0);
        };
#pragma unroll
        for (int u = 0; u < PF; ++u) fetch(u);
        int st = 0;
        for (; st + NB <= steps; st += NB) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                fetch((u + PF) % NB);
                __builtin_amdgcn_sched_barrier(0);
                mfma_step(u);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int u = 0; u < NB - 1; ++u)
            if (st + u < steps) mfma_step(u);

        // ---- epilogue
        if (dense1) {
            // 32 -> 1 classifier: cout 0 = register 0 of the lanes with g == 0; dense fp32 out (+ the previous head's cost)
            float* yd = (float*)p.y;
            const float* rd = (const float*)p.res;
            if (g == 0 && cg == 0) {
#pragma unroll
                for (int vt = 0; vt < VT; ++vt)
                    if (valid[vt]) {
                        const int s = vt * 16 + j;
                        const int r = s / p.WT, c = s - r * p.WT;
                        const long o = (((long)n * p.OD + od) * p.OH + (oh0 + r)) * p.OW + (ow0 + c);
                        float v = acc[vt][0].x;
                        if (rd) v += rd[o];
                        yd[o] = v;
                    }
            }
            continue;
        }
        _Float16* y = (_Float16*)p.y;
        const _Float16* res = (const _Float16*)p.res;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int cot = ct0 + ct;                                  // cout tile -> channel block cot >> 1, half (cot & 1) of its 32 channels
            const f32x4 sc = *(const f32x4*)(p.scale + cot * 16 + g * 4);
            const f32x4 sh = *(const f32x4*)(p.shift + cot * 16 + g * 4);
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
                if (valid[vt]) {
                    const int s = vt * 16 + j;
                    const int r = s / p.WT, c = s - r * p.WT;
                    const long vox = (long)(od * p.out_mul + cls.out_off_d) * 1;
                    const long yo = p.y_off0 + (long)n * p.y_n_stride + (long)(cot >> 1) * p.y_cb_stride + vox * p.y_d_stride +
                                    (long)((oh0 + r) * p.out_mul + cls.out_off_h) * p.y_h_stride + (long)((ow0 + c) * p.out_mul + cls.out_off_w) * 32 +
                                    (cot & 1) * 16 + g * 4;
                    f32x4 v = acc[vt][ct] * sc + sh;
                    if (res) {
                        const long ro = p.r_off0 + (long)n * p.r_n_stride + (long)(cot >> 1) * p.r_cb_stride + vox * p.r_d_stride +
                                        (long)((oh0 + r) * p.out_mul + cls.out_off_h) * p.r_h_stride + (long)((ow0 + c) * p.out_mul + cls.out_off_w) * 32 +
                                        (cot & 1) * 16 + g * 4;
                        const f16x4 rv = *(const f16x4*)(res + ro);
                        v.x += (float)rv.x; v.y += (float)rv.y; v.z += (float)rv.z; v.w += (float)rv.w;
                    }
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    f16x4 hv;
                    hv.x = (_Float16)v.x; hv.y = (_Float16)v.y; hv.z = (_Float16)v.z; hv.w = (_Float16)v.w;
                    *(f16x4*)(y + yo) = hv;
                }
        }
    }
}

template <int VT, int CT>
int launch(const drc_tapconv_params& p, hipStream_t stream) {
    const long n_wt = (p.OW + p.WT - 1) / p.WT, n_rt = (p.OH + p.R - 1) / p.R;
    const long groups = (long)p.N * p.OD * n_rt * n_wt * (p.cout_pad / 16 / CT) * p.n_classes;
    long blocks = (groups + C16_WAVES - 1) / C16_WAVES;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((conv16_kernel<VT, CT>), dim3((unsigned)blocks), dim3(64 * C16_WAVES), 0, stream, p);
    return (int)hipGetLastError();
}

// features (fp32) -> fp16 blocked cost volume: channel block 0 = left, block 1 = right shifted by the slice's disparity, both zero
// where the shifted right pixel falls outside (stackhourglass.py:115-128).  One thread per (voxel, 8-channel chunk).
__global__ __launch_bounds__(256) void cost_volume16_kernel(const float* __restrict__ L, const float* __restrict__ Rr, _Float16* __restrict__ out,
                                                            int N, int C, int D, int H, int W, int lo4, int in_pad, long n_stride, long cb_stride,
                                                            long d_stride, long h_stride, long off0) {
    // inputs: NCHW fp32 (in_pad < 0) or the fp32 blocked 2D layout [N][C/16][1][H+2p][W+2p][16] (in_pad = p >= 0)
    const long total = (long)N * D * H * W * 8;                        // 2 blocks x 4 chunks of 8 channels
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        long t = idx;
        const int ch = (int)(t & 3); t >>= 2;
        const int side = (int)(t & 1); t >>= 1;
        const int xw = (int)(t % W); t /= W;
        const int yh = (int)(t % H); t /= H;
        const int d = (int)(t % D);
        const int n = (int)(t / D);
        const int i = lo4 + d;
        const int xs = xw - i;
        const bool ok = xs >= 0 && xs < W;
        f16x8 v;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = ch * 8 + k;
            float f = 0.f;
            if (ok && c < C) {
                const float* src = side == 0 ? L : Rr;
                const int xx = side == 0 ? xw : xs;
                if (in_pad < 0) f = src[(((long)n * C + c) * H + yh) * W + xx];
                else f = src[((((long)n * ((C + 15) / 16) + c / 16) * (H + 2 * in_pad) + (yh + in_pad)) * (W + 2 * in_pad) + (xx + in_pad)) * 16 + (c & 15)];
            }
            v[k] = (_Float16)f;
        }
        *(f16x8*)(out + off0 + (long)n * n_stride + (long)side * cb_stride + (long)d * d_stride + (long)yh * h_stride + (long)xw * 32 + ch * 8) = v;
    }
}

}  // namespace

extern "C" int drc_conv16_fwd(const drc_tapconv_params* pp, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y) return -1;
    if (p.reserved != 1 && (!p.scale || !p.shift)) return -1;
    if (p.N < 0 || p.OD <= 0 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (p.cout_pad <= 0 || (p.cout_pad & 15) || p.cb_in <= 0 || p.n_classes < 1 || p.n_classes > DRC_MAX_CLASSES) return -2;
    if (p.R <= 0 || p.WT <= 0 || p.R * p.WT > 64) return -2;
    if (p.in_mul < 1 || p.in_mul > 2 || p.out_mul < 1 || p.out_mul > 2) return -2;
    // 32-bit byte offsets inside one unit's tensor and inside the weights (buffer loads)
    if (p.x_n_stride * 2 >= (1LL << 31) || (int64_t)p.cb_in * p.cout_pad * 32 * 27 * 2 >= (1LL << 31)) return -5;
    const int vt = (p.R * p.WT + 15) / 16, ct = p.cout_pad / 16;
    hipStream_t s = (hipStream_t)stream;
    const bool two = ct % 2 == 0;
    switch (vt) {
        case 1: return two ? launch<1, 2>(p, s) : launch<1, 1>(p, s);
        case 2: return two ? launch<2, 2>(p, s) : launch<2, 1>(p, s);
        case 3: return two ? launch<3, 2>(p, s) : launch<3, 1>(p, s);
        default: return two ? launch<4, 2>(p, s) : launch<4, 1>(p, s);
    }
}

extern "C" int drc_cost_volume16_blocked_fwd(const float* left, const float* right, void* cost16, int N, int C, int Dp, int Hp, int Wp,
                                             int mindisp4, int maxdisp4, int in_blocked_pad, void* stream) {
    if (N < 0 || C <= 0 || C > 32 || Dp <= 0 || Hp <= 0 || Wp <= 0 || maxdisp4 - mindisp4 != Dp) return -2;
    if (N == 0) return 0;
    if (!left || !right || !cost16) return -1;
    const long h_stride = (long)(Wp + 2) * 32, d_stride = (long)(Hp + 2) * h_stride, cb_stride = (long)(Dp + 2) * d_stride, n_stride = 2 * cb_stride;
    const long off0 = d_stride + h_stride + 32;
    const long total = (long)N * Dp * Hp * Wp * 8;
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(cost_volume16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, left, right, (_Float16*)cost16, N, C, Dp, Hp, Wp,
                       mindisp4, in_blocked_pad, n_stride, cb_stride, d_stride, h_stride, off0);
    return (int)hipGetLastError();
}
