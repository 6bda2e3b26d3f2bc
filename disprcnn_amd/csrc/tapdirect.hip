// tapdirect.hip -- stride-1 3x3x3 convolution (+BN, +residual, +ReLU) with a sliding depth window and BOTH MFMA operands read
// straight from global memory (no LDS) (gfx950 / CDNA4).
//
//   reference: dres0/dres1, classifN[0], hourglass conv2/conv4 (stackhourglass.py:63-88, :14-20)
//
// tapslide.hip stages a private tile per wave with LDS-DMA and pays, per 84-MFMA tap step, one LDS-DMA piece (~190 cycles of
// issue), 7 ds_read_b64 (~125) and 6 weight loads (~85): the 0.15 it loses against its own MFMA-only ceiling.  In the blocked
// layout a voxel's 16 channels are one 64-byte line, so the B fragment of tap (kh, kw) is a plain coalesced float4 load per lane
// (lane (voxel j, g) reads channels 4g..4g+3 of voxel (r+kh, c+kw)): it covers FOUR MFMA k-steps, where a ds_read_b64 of the
// 8-channel LDS tile covers two.  The nine shifted reads of a row hit L1/L2; nothing is staged, nothing waits on an LDS-DMA
// counter, the compiler's own vmcnt accounting is exact.  Per tap step (one (kh, kw), 16 channels, three depth taps):
// VT B loads + 3*CT weight loads for 3 * VT * CT * 4 MFMAs (13 loads per 168 MFMAs at VT=7, CT=2), all one step ahead.
// Work decomposition, accumulator rotation and epilogue are those of tapslide.hip.
// Weights: [27 taps][cb_in][cout_pad][16] (engine.pack_weight_t16) -- lane (cout j, g) reads channels 4g..4g+3; k-step s of a
// block uses channel 4g+s on both operands.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TD_WAVES 4

namespace {

template <int VT, int CT>
__global__ __launch_bounds__(64 * TD_WAVES) void tapdirect_kernel(const drc_tapconv_params p) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const int g = lane >> 4;

    const drc_tap_class cls = p.cls[0];
    const int n_wt = (p.OW + p.WT - 1) / p.WT;
    const int n_rt = (p.OH + p.R - 1) / p.R;
    const int cols = p.N * n_rt * n_wt;            // columns = (n, row tile, col tile) of one cout group
    const int D = p.OD;
    // equal contiguous shares of the (cout group, column, output slice) units, as in tapslide.hip
    const long units = (long)(p.cout_pad / 16 / CT) * cols * D;
    const long workers = (long)gridDim.x * TD_WAVES;
    const long wid = (long)blockIdx.x * TD_WAVES + wave;
    long ucur = units * wid / workers;
    const long u1 = units * (wid + 1) / workers;
    const int nslots = p.R * p.WT;

    // per-lane byte offset of voxel slot (vt, j) at tap (0,0), channels 4g..4g+3 (relative to the column's slice origin)
    unsigned lane_vo[VT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
        const int s = vt * 16 + j;
        int r = s / p.WT, c = s - r * p.WT;
        if (s >= nslots) { r = 0; c = 0; }
        lane_vo[vt] = (unsigned)((r * (int)p.x_h_stride + c * 16 + g * 4) * 4);
    }
    const int64_t w_cb = (int64_t)p.cout_pad * 16;         // floats per (tap, cb)
    const int64_t w_tap = w_cb * p.cb_in;                  // floats per tap

    // per-segment state
    const float* xcol;
    const float* wl;       // weights of this lane: + tap * w_tap + cb * w_cb + ct * 256
    int n, oh0, ow0, ct0, od_lo, od_hi, din_lo, din_hi;

    f32x4 bn_sc[CT], bn_sh[CT];
    f32x4 acc0[VT][CT], acc1[VT][CT], acc2[VT][CT];   // three output slices in flight
    // accumulators are cleared from a zero produced by a volatile asm at the point of use (a literal zero vector per tile gets
    // hoisted and parked in 4 AGPRs per tile)
#define TD_ZERO(ACC)                                                                                   \
    {                                                                                                  \
        float z_;                                                                                      \
        asm volatile("v_mov_b32 %0, 0" : "=v"(z_));                                                    \
        const f32x4 z4_ = {z_, z_, z_, z_};                                                            \
        _Pragma("unroll") for (int vt = 0; vt < VT; ++vt)                                              \
            _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) ACC[vt][ct] = z4_;                       \
    }

#define TD_EPILOGUE(OD, ACC)                                                                           \
    {                                                                                                  \
        _Pragma("unroll") for (int vt = 0; vt < VT; ++vt) {                                            \
            const int s_ = vt * 16 + j;                                                                \
            const int r_ = s_ / p.WT, c_ = s_ - r_ * p.WT;                                             \
            const bool valid_ = (s_ < nslots) && (oh0 + r_ < p.OH) && (ow0 + c_ < p.OW);               \
            if (valid_) {                                                                              \
                const int64_t yo_ = p.y_off0 + (int64_t)n * p.y_n_stride + (int64_t)(OD) * p.y_d_stride + \
                                    (int64_t)(oh0 + r_) * p.y_h_stride + (int64_t)(ow0 + c_) * 16 + g * 4; \
                const int64_t ro_ = p.r_off0 + (int64_t)n * p.r_n_stride + (int64_t)(OD) * p.r_d_stride + \
                                    (int64_t)(oh0 + r_) * p.r_h_stride + (int64_t)(ow0 + c_) * 16 + g * 4; \
                _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) {                                    \
                    f32x4 v_ = ACC[vt][ct] * bn_sc[ct] + bn_sh[ct];                                    \
                    if (p.res) v_ += *(const f32x4*)(p.res + ro_ + (int64_t)(ct0 + ct) * p.r_cb_stride); \
                    if (p.relu) { v_.x = fmaxf(v_.x, 0.f); v_.y = fmaxf(v_.y, 0.f); v_.z = fmaxf(v_.z, 0.f); v_.w = fmaxf(v_.w, 0.f); } \
                    *(f32x4*)(p.y + yo_ + (int64_t)(ct0 + ct) * p.y_cb_stride) = v_;                  \
                }                                                                                      \
            }                                                                                          \
        }                                                                                              \
        TD_ZERO(ACC)                                                                                   \
    }

    // operands of one tap step: B fragments of tap (kh,kw) of (slice d, block cb) and the weights of its three depth taps
    f32x4 bA[VT], bB[VT], wA[3][CT], wB[3][CT];
    auto load_step = [&](f32x4 (&B)[VT], f32x4 (&Wt)[3][CT], int d_in, int cb, int t) __attribute__((always_inline)) {
        const int kh = t / 3, kw = t - kh * 3;
        const char* sb = (const char*)(xcol + (int64_t)cb * p.x_cb_stride + (int64_t)(d_in + cls.dd0 + 1) * p.x_d_stride +   // real slice d_in at padded depth d_in + dd0 + 1
                                       (int64_t)kh * p.x_h_stride + kw * 16);
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) B[vt] = *(const f32x4*)(sb + lane_vo[vt]);
        const float* wp = wl + (int64_t)t * w_tap + (int64_t)cb * w_cb;
#pragma unroll
        for (int dd = 0; dd < 3; ++dd)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) Wt[dd][ct] = *(const f32x4*)(wp + (int64_t)dd * 9 * w_tap + ct * 256);
    };

#define TD_MFMA(ACC, W, B)                                                                             \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                      \
        _Pragma("unroll") for (int vt = 0; vt < VT; ++vt)                                              \
            _Pragma("unroll") for (int ct = 0; ct < CT; ++ct)                                          \
                ACC[vt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[ct][s], B[vt][s], ACC[vt][ct], 0, 0, 0);

    // one tap step: prefetch the next step's operands, then up to three depth taps x 4 k-steps on the current ones
#define TD_STEP(B_USE, W_USE, B_LD, W_LD, A0, A1, A2)                                                  \
    {                                                                                                  \
        {   /* next step: next tap, or tap 0 of the next block / next slice (past the end: a harmless reload) */ \
            int tn_ = t + 1, cbn_ = cb, dn_ = d_in;                                                    \
            if (tn_ == 9) { tn_ = 0; if (++cbn_ == p.cb_in) { cbn_ = 0; dn_ = d_in + 1 <= din_hi ? d_in + 1 : d_in; } } \
            load_step(B_LD, W_LD, dn_, cbn_, tn_);                                                     \
        }                                                                                              \
        if (v1) TD_MFMA(A1, W_USE[1], B_USE)                                                           \
        if (v0) TD_MFMA(A0, W_USE[0], B_USE)                                                           \
        if (v2) TD_MFMA(A2, W_USE[2], B_USE)                                                           \
    }

// all steps of input slice d_in; afterwards output slice d_in-1 (set A2) is complete.  The 9 taps of a block alternate the
// A/B operand sets statically; the ninth prefetches the next block's tap 0 into set B, which is then moved to set A.
#define TD_SLICE(A0, A1, A2)                                                                           \
    {                                                                                                  \
        const bool v0 = d_in + 1 >= od_lo && d_in + 1 < od_hi;                                         \
        const bool v1 = d_in >= od_lo && d_in < od_hi;                                                 \
        const bool v2 = d_in - 1 >= od_lo && d_in - 1 < od_hi;                                         \
        for (int cb = 0; cb < p.cb_in; ++cb) {                                                         \
            for (int t = 0; t < 8; t += 2) {                                                           \
                TD_STEP(bA, wA, bB, wB, A0, A1, A2)                                                    \
                { const int t_ = t; (void)t_; }                                                        \
                { const int t1_ = t + 1; const int t = t1_; TD_STEP(bB, wB, bA, wA, A0, A1, A2) }      \
            }                                                                                          \
            { const int t = 8; TD_STEP(bA, wA, bB, wB, A0, A1, A2) }                                   \
            _Pragma("unroll") for (int vt = 0; vt < VT; ++vt) bA[vt] = bB[vt];                         \
            _Pragma("unroll") for (int dd = 0; dd < 3; ++dd)                                           \
                _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) wA[dd][ct] = wB[dd][ct];             \
        }                                                                                              \
        if (v2) TD_EPILOGUE(d_in - 1, A2)                                                              \
        else TD_ZERO(A2)                                                                               \
        if (v1 && d_in + 1 == D) TD_EPILOGUE(d_in, A1)                                                 \
    }

#pragma unroll 1
    while (ucur < u1) {
        {
            const long colid = ucur / D;
            od_lo = (int)(ucur - colid * D);
            od_hi = (long)od_lo + (u1 - ucur) < D ? od_lo + (int)(u1 - ucur) : D;
            ucur += od_hi - od_lo;
            int cid = (int)(colid % cols);
            ct0 = (int)(colid / cols) * CT;
            const int wt = cid % n_wt; cid /= n_wt;
            const int rt = cid % n_rt;
            n = cid / n_rt;
            oh0 = rt * p.R; ow0 = wt * p.WT;
            din_lo = od_lo > 0 ? od_lo - 1 : 0;
            din_hi = od_hi < D ? od_hi : D - 1;      // inclusive
            xcol = p.x + (int64_t)n * p.x_n_stride + (int64_t)(oh0 + cls.dh0) * p.x_h_stride + (int64_t)(ow0 + cls.dw0) * 16;
            wl = p.w + ((int64_t)(ct0 * 16 + j)) * 16 + g * 4;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                bn_sc[ct] = *(const f32x4*)(p.scale + (ct0 + ct) * 16 + g * 4);
                bn_sh[ct] = *(const f32x4*)(p.shift + (ct0 + ct) * 16 + g * 4);
            }
            TD_ZERO(acc0) TD_ZERO(acc1) TD_ZERO(acc2)
            load_step(bA, wA, din_lo, 0, 0);
        }
        for (int d_in = din_lo - din_lo % 3;;) {
            if (d_in >= din_lo) TD_SLICE(acc1, acc0, acc2)
            if (++d_in > din_hi) break;
            if (d_in >= din_lo) TD_SLICE(acc2, acc1, acc0)
            if (++d_in > din_hi) break;
            if (d_in >= din_lo) TD_SLICE(acc0, acc2, acc1)
            if (++d_in > din_hi) break;
        }
    }
#undef TD_SLICE
#undef TD_STEP
#undef TD_MFMA
#undef TD_EPILOGUE
#undef TD_ZERO
}

template <int VT, int CT>
int launch(const drc_tapconv_params& p, hipStream_t stream) {
    const long cols = (long)p.N * ((p.OH + p.R - 1) / p.R) * ((p.OW + p.WT - 1) / p.WT);
    static int occ_blocks = 0;   // per-instantiation, idempotent
    if (!occ_blocks) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, tapdirect_kernel<VT, CT>, 64 * TD_WAVES, 0) != hipSuccess || nb < 1) nb = 1;
        occ_blocks = nb;
    }
    const long units = cols * (p.cout_pad / 16 / CT) * p.OD;
    long workers = 256L * TD_WAVES * occ_blocks;
    if (workers > units / 3) workers = units / 3 > workers / 2 ? units / 3 : (units < workers ? units : workers);
    if (workers < TD_WAVES) workers = TD_WAVES;
    dim3 grid((unsigned)((workers + TD_WAVES - 1) / TD_WAVES), 1, 1);
    hipLaunchKernelGGL((tapdirect_kernel<VT, CT>), grid, dim3(64 * TD_WAVES), 0, stream, p);
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

extern "C" int drc_tapconv3d_direct_fwd(const drc_tapconv_params* pp, int cout_tiles_per_wave, void* stream) {
    if (!pp) return -1;
    const drc_tapconv_params& p = *pp;
    if (!p.x || !p.w || !p.y || !p.scale || !p.shift) return -1;
    if (p.N < 0 || p.OD <= 0 || p.OH <= 0 || p.OW <= 0) return -2;
    if (p.N == 0) return 0;
    if (p.cout_pad <= 0 || (p.cout_pad & 15) || p.cb_in <= 0) return -2;
    if (p.R <= 0 || p.WT <= 0 || p.R * p.WT > 112) return -3;
    const drc_tap_class& k = p.cls[0];
    if (p.n_classes != 1 || p.in_mul != 1 || p.out_mul != 1 || k.nd != 3 || k.nh != 3 || k.nw != 3 || k.sd != 1 || k.sh != 1 ||
        k.sw != 1 || k.wbase != 0 || k.wsd != 9 || k.wsh != 3 || k.wsw != 1)
        return -4;
    if ((int64_t)(p.R + 2) * p.x_h_stride * 4 >= (1LL << 31)) return -5;       // 32-bit lane offsets
    const int ct = p.cout_pad / 16, CT = cout_tiles_per_wave;
    if ((CT != 1 && CT != 2) || ct % CT) return -2;
    const int nvt = (p.R * p.WT + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
    return CT == 2 ? launch_vt<2>(nvt, p, s) : launch_vt<1>(nvt, p, s);
}
