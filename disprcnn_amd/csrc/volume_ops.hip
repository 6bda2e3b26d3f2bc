// volume_ops.hip -- HBM-bound kernels of the instance-disparity path (gfx950):
//   cost-volume build (dense NCDHW, blocked), its adjoint, layout converters, the 32->1 classifier conv,
//   and the fused trilinear-upsample + softmax + soft-argmin epilogue.
// All are pure data movement / short reductions: coalesced 16-byte accesses, no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kThreads = 256;

inline unsigned grid_for(long work, int cap = 256 * 16) {
    long b = (work + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

// ------------------------------------------------------------------------------------------------
// a1: dense cost volume.  Reference: stackhourglass.py:115-128 (zeros + 2*(D/4) slice copies).
// Thread = 4 consecutive x of one (n, c2, j, y) row; L as aligned float4, R as shifted scalars.
template <bool VEC4>
__global__ __launch_bounds__(kThreads) void cost_volume_dense_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                                     float* __restrict__ cost, int N, int C, int Dp, int Hp, int Wp,
                                                                     int lo4, int hi4) {
    const int XV = VEC4 ? 4 : 1;
    const int wq = Wp / XV;
    const long total = (long)N * 2 * C * Dp * Hp * wq;
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
        long t = idx;
        const int xq = (int)(t % wq); t /= wq;
        const int y = (int)(t % Hp); t /= Hp;
        const int j = (int)(t % Dp); t /= Dp;
        const int c2 = (int)(t % (2 * C));
        const int n = (int)(t / (2 * C));
        const int i = lo4 + j;
        const bool live = i < hi4;
        const int x0 = xq * XV;
        float v[4];
        const bool right = c2 >= C;
        const int c = right ? c2 - C : c2;
        const float* src = (right ? R : L) + (((long)n * C + c) * Hp + y) * Wp;
#pragma unroll
        for (int e = 0; e < XV; ++e) {
            const int x = x0 + e, xs = x - i;
            const bool ok = live && xs >= 0 && xs < Wp;
            v[e] = ok ? src[right ? xs : x] : 0.0f;
        }
        float* dst = cost + ((((long)n * 2 * C + c2) * Dp + j) * Hp + y) * Wp + x0;
        if (VEC4) *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
        else dst[0] = v[0];
    }
}

// adjoint of the copy.  Thread per (half, n, c, y, x): sums the D' slices that copied this element.
__global__ __launch_bounds__(kThreads) void cost_volume_bwd_kernel(const float* __restrict__ g, float* __restrict__ gL,
                                                                   float* __restrict__ gR, int N, int C, int Dp, int Hp, int Wp,
                                                                   int lo4, int hi4) {
    const long per = (long)N * C * Hp * Wp;
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < 2 * per; idx += (long)gridDim.x * kThreads) {
        const bool right = idx >= per;
        long t = right ? idx - per : idx;
        const int x = (int)(t % Wp); t /= Wp;
        const int y = (int)(t % Hp); t /= Hp;
        const int c = (int)(t % C);
        const int n = (int)(t / C);
        float s = 0.f;
        for (int j = 0; j < Dp; ++j) {
            const int i = lo4 + j;
            if (i >= hi4) break;
            const int xo = right ? x + i : x;      // output column that copied this input element
            const int xs = xo - i;                  // == x for right, x - i for left validity test
            const bool ok = right ? (xo >= 0 && xo < Wp) : (xs >= 0 && xs < Wp);
            if (ok) s += g[((((long)n * 2 * C + (right ? C + c : c)) * Dp + j) * Hp + y) * Wp + xo];
        }
        (right ? gR : gL)[(((long)n * C + c) * Hp + y) * Wp + x] = s;
    }
}

// a1, blocked output [N][2C/16][Dp+2][Hp+2][Wp+2][16] (pads 1).  Thread = one float4 (4 channels of a voxel).
__global__ __launch_bounds__(kThreads) void cost_volume_blocked_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                                       float* __restrict__ out, int N, int C, int Dp, int Hp, int Wp,
                                                                       int lo4, int hi4, int fp) {
    const int CBo = 2 * C / 16, CBi = C / 16;
    const long total = (long)N * CBo * Dp * Hp * Wp * 4;
    const long oH = Wp + 2, oD = (long)(Hp + 2) * oH, oC = (long)(Dp + 2) * oD;
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
        long t = idx;
        const int q = (int)(t & 3); t >>= 2;
        const int x = (int)(t % Wp); t /= Wp;
        const int y = (int)(t % Hp); t /= Hp;
        const int j = (int)(t % Dp); t /= Dp;
        const int cbo = (int)(t % CBo);
        const int n = (int)(t / CBo);
        const int i = lo4 + j;
        const int xs = x - i;
        const bool ok = (i < hi4) && xs >= 0 && xs < Wp;
        const bool right = cbo >= CBi;
        const int cbi = right ? cbo - CBi : cbo;
        const int xr = right ? xs : x;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
            const float* src = right ? R : L;
            if (fp > 0) {
                const long iW = Wp + 2 * fp, iH = Hp + 2 * fp;
                v = *(const f32x4*)(src + ((((long)n * CBi + cbi) * iH + (y + fp)) * iW + (xr + fp)) * 16 + q * 4);
            } else {
                const long plane = (long)Hp * Wp;
                const float* s0 = src + ((long)n * C + cbi * 16 + q * 4) * plane + (long)y * Wp + xr;
                v = (f32x4){s0[0], s0[plane], s0[2 * plane], s0[3 * plane]};
            }
        }
        *(f32x4*)(out + (((long)n * CBo + cbo) * oC + (long)(j + 1) * oD + (long)(y + 1) * oH + (x + 1)) * 16 + q * 4) = v;
    }
}


// a1, blocked output, NCHW features: one block per (n, y) feature row.  The thread-per-float4 kernel above gathers four
// plane-strided scalars per lane from NCHW (8.5x over-fetch on the read side, 0.35 of the HBM peak).  Here the 2C channel rows of
// the (n, y) line -- W contiguous floats each -- are read ONCE, coalesced along x, and transposed through LDS (row stride C + 1:
// conflict-free); every one of the 2C/16 x D' output rows of that line (W x 64 contiguous bytes each) is then written from LDS
// with the right half shifted by the slice's disparity and the validity mask applied: reads 2*C*W*4 B, writes (2C/16)*D'*W*64 B.
__global__ __launch_bounds__(kThreads) void cost_volume_blocked_rows_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                                            float* __restrict__ out, int C, int Dp, int Hp, int Wp, int lo4, int hi4) {
    extern __shared__ float cv_lds[];                       // [2][Wp][C + 1]
    const int n = blockIdx.x / Hp, y = blockIdx.x - n * Hp;
    const int cs = C + 1;
    float* lL = cv_lds;
    float* lR = cv_lds + (long)Wp * cs;
    const long plane = (long)Hp * Wp;
    // stage: 32 lanes per channel row (x fastest)
    const int xl = threadIdx.x & 31, sub = threadIdx.x >> 5, nsub = kThreads >> 5;
    for (int c = sub; c < 2 * C; c += nsub) {
        const bool right = c >= C;
        const int cc = right ? c - C : c;
        const float* src = (right ? R : L) + ((long)n * C + cc) * plane + (long)y * Wp;
        float* dst = right ? lR : lL;
        for (int x = xl; x < Wp; x += 32) dst[x * cs + cc] = src[x];
    }
    __syncthreads();
    const int CBi = C / 16, CBo = 2 * CBi;
    const long oH = Wp + 2, oD = (long)(Hp + 2) * oH, oC = (long)(Dp + 2) * oD;
    const int per_row = Wp * 4;                             // float4 per output row
    const int total = CBo * Dp * per_row;
    for (int idx = threadIdx.x; idx < total; idx += kThreads) {
        int t = idx;
        const int q = t & 3; t >>= 2;
        const int x = t % Wp; t /= Wp;
        const int j = t % Dp;
        const int cbo = t / Dp;
        const int i = lo4 + j;
        const int xs = x - i;
        const bool ok = (i < hi4) && xs >= 0 && xs < Wp;
        const bool right = cbo >= CBi;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
            const float* s = (right ? lR + xs * cs + (cbo - CBi) * 16 : lL + x * cs + cbo * 16) + q * 4;
            v = (f32x4){s[0], s[1], s[2], s[3]};
        }
        *(f32x4*)(out + (((long)n * CBo + cbo) * oC + (long)(j + 1) * oD + (long)(y + 1) * oH + (x + 1)) * 16 + q * 4) = v;
    }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void dense_to_blocked_kernel(const float* __restrict__ dense, float* __restrict__ blk, int N, int C,
                                                                    int D, int H, int W, int pd, int ph, int pw) {
    const int CB = (C + 15) / 16;
    const long total = (long)N * CB * D * H * W * 4;
    const long bW = W + 2 * pw, bH = H + 2 * ph, bD = D + 2 * pd;
    const long plane = (long)D * H * W;
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
        long t = idx;
        const int q = (int)(t & 3); t >>= 2;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H); t /= H;
        const int d = (int)(t % D); t /= D;
        const int cb = (int)(t % CB);
        const int n = (int)(t / CB);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = cb * 16 + q * 4 + e;
            v[e] = c < C ? dense[((long)n * C + c) * plane + ((long)d * H + y) * W + x] : 0.f;
        }
        *(f32x4*)(blk + ((((long)n * CB + cb) * bD + (d + pd)) * bH * bW + (long)(y + ph) * bW + (x + pw)) * 16 + q * 4) =
            (f32x4){v[0], v[1], v[2], v[3]};
    }
}

// The same conversion with one block per (n, d, y) line: the C channel rows of the line (W contiguous floats each) are read once,
// coalesced along x, transposed through LDS (row stride 16*CB + 1: conflict-free) and written as CB rows of W x 64 contiguous
// bytes.  The thread-per-float4 form above gathers four plane-strided scalars per lane (8x over-fetch on the read side).
__global__ __launch_bounds__(kThreads) void dense_to_blocked_rows_kernel(const float* __restrict__ dense, float* __restrict__ blk, int C,
                                                                         int D, int H, int W, int pd, int ph, int pw) {
    extern __shared__ float d2b_lds[];                      // [W][16*CB + 1]
    const int CB = (C + 15) / 16, cs = 16 * CB + 1;
    int t = blockIdx.x;
    const int y = t % H; t /= H;
    const int d = t % D;
    const int n = t / D;
    const long plane = (long)D * H * W;
    const int xl = threadIdx.x & 31, sub = threadIdx.x >> 5, nsub = kThreads >> 5;
    for (int c = sub; c < 16 * CB; c += nsub) {
        const float* src = dense + ((long)n * C + c) * plane + ((long)d * H + y) * W;
        for (int x = xl; x < W; x += 32) d2b_lds[x * cs + c] = c < C ? src[x] : 0.f;
    }
    __syncthreads();
    const long bW = W + 2 * pw, bH = H + 2 * ph, bD = D + 2 * pd;
    const int total = CB * W * 4;
    for (int idx = threadIdx.x; idx < total; idx += kThreads) {
        const int q = idx & 3;
        const int x = (idx >> 2) % W;
        const int cb = (idx >> 2) / W;
        const float* s4 = d2b_lds + x * cs + cb * 16 + q * 4;
        *(f32x4*)(blk + ((((long)n * CB + cb) * bD + (d + pd)) * bH * bW + (long)(y + ph) * bW + (x + pw)) * 16 + q * 4) =
            (f32x4){s4[0], s4[1], s4[2], s4[3]};
    }
}

__global__ __launch_bounds__(kThreads) void blocked_to_dense_kernel(const float* __restrict__ blk, float* __restrict__ dense, int N, int C,
                                                                    int D, int H, int W, int pd, int ph, int pw) {
    const int CB = (C + 15) / 16;
    const long total = (long)N * C * D * H * W;
    const long bW = W + 2 * pw, bH = H + 2 * ph, bD = D + 2 * pd;
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
        long t = idx;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H); t /= H;
        const int d = (int)(t % D); t /= D;
        const int c = (int)(t % C);
        const int n = (int)(t / C);
        dense[idx] = blk[((((long)n * CB + (c >> 4)) * bD + (d + pd)) * bH * bW + (long)(y + ph) * bW + (x + pw)) * 16 + (c & 15)];
    }
}

// ------------------------------------------------------------------------------------------------
// classifN[2]: Conv3d(32->1,k3,p1).  4 lanes per voxel (channel quads), 27*cb_in float4 loads, quad-reduce by shuffle.
__global__ __launch_bounds__(kThreads) void conv3d_cout1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ res, float* __restrict__ out, int N, int cb_in,
                                                                int D, int H, int W) {
    const long nvox = (long)N * D * H * W;
    const long total = nvox * 4;
    const long bW = W + 2, bH = H + 2, bD = D + 2;
    const long sH = bW * 16, sD = bH * sH, sC = bD * sD, sN = sC * cb_in;
    const long work = ((total + kThreads - 1) / kThreads) * kThreads;  // keep whole waves alive for the shuffles
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < work; idx += (long)gridDim.x * kThreads) {
        const bool live = idx < total;
        long t = live ? idx : 0;
        const int q = (int)(t & 3); t >>= 2;
        const long vox = t;
        const int xw = (int)(t % W); t /= W;
        const int y = (int)(t % H); t /= H;
        const int d = (int)(t % D);
        const int n = (int)(t / D);
        const float* xb = x + n * sN + d * sD + y * sH + (long)xw * 16 + q * 4;  // tap (0,0,0) in padded coords
        float s = 0.f;
        for (int cb = 0; cb < cb_in; ++cb) {
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const f32x4 xv = *(const f32x4*)(xb + cb * sC + kd * sD + kh * sH + kw * 16);
                        const f32x4 wv = *(const f32x4*)(w + ((kd * 3 + kh) * 3 + kw) * cb_in * 16 + cb * 16 + q * 4);
                        s = fmaf(xv.x, wv.x, s); s = fmaf(xv.y, wv.y, s); s = fmaf(xv.z, wv.z, s); s = fmaf(xv.w, wv.w, s);
                    }
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (live && q == 0) out[vox] = res ? s + res[vox] : s;
    }
}


// classifN[2] with a sliding depth window: a wave owns R rows x W columns of ALL depth slices of one ROI, stages every
// input tile ([rows][W+2][16 ch], LDS-DMA, double-buffered) once and applies it to the three output slices it touches
// (od = d_in+1-dd), so the 32-channel input is fetched ~1.5x (row halo) instead of once per tap.  VALU dot products:
// 864 MACs per output voxel are nothing next to reading 128 B per input voxel.
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr int kC1Waves = 4;
__global__ __launch_bounds__(64 * kC1Waves) void conv3d_cout1_slide_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                           const float* __restrict__ res, float* __restrict__ out, int N,
                                                                           int cb_in, int D, int H, int W, int R, int seg_len) {
    extern __shared__ __attribute__((aligned(16))) float lds_c1[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int n_rt = (H + R - 1) / R;
    int cid = blockIdx.x * kC1Waves + wave;
    if (cid >= N * n_rt) return;                          // wave-uniform, no workgroup barrier below
    const int rt = cid % n_rt, n = cid / n_rt;
    const int oh0 = rt * R;
    const int Wp = W + 2, Hp = H + 2;
    const long sH = (long)Wp * 16, sD = (long)Hp * sH, sC = (long)(D + 2) * sD, sN = sC * cb_in;
    const int rows_in = R + 2;
    const int seg_floats = Wp * 16, seg_units = Wp * 4;    // 64 B per voxel
    const int buf_floats = rows_in * seg_floats;
    float* lds = lds_c1 + wave * 2 * buf_floats;
    const float* xcol = x + n * sN + (long)oh0 * sH;      // padded row oh0 (= real row oh0-1), padded col 0
    // this lane's (up to two) voxels inside the R x W tile
    const int nvox = R * W;
    int voff[2]; bool vok[2]; int vr[2], vc[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int s = lane + 64 * k;
        vok[k] = s < nvox && oh0 + s / W < H;
        vr[k] = vok[k] ? s / W : 0; vc[k] = vok[k] ? s - (s / W) * W : 0;
        voff[k] = (vr[k] * Wp + vc[k]) * 16;
    }
    auto stage = [&](int d_in, int cb, int bufi) {
        const float* src = xcol + cb * sC + (long)(d_in + 1) * sD;
        float* dst = lds + bufi * buf_floats;
        for (int r = 0; r < rows_in; ++r)
            for (int u0 = 0; u0 < seg_units; u0 += 64) {
                const int u = u0 + lane;
                if (u < seg_units) __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + r * sH + u * 4), LDS_PTR(dst + r * seg_floats + u0 * 4), 16, 0, 0);
            }
    };
    float a0[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f}, a2[2] = {0.f, 0.f};   // outputs od = d_in+1, d_in, d_in-1 in flight
    // depth segment of this wave: outputs [od_lo, od_hi), input slices od_lo-1 .. od_hi clipped to the volume
    const int od_lo = blockIdx.y * seg_len;
    const int od_hi = od_lo + seg_len < D ? od_lo + seg_len : D;
    const int din_lo = od_lo > 0 ? od_lo - 1 : 0, din_hi = od_hi < D ? od_hi : D - 1;
    int bufsel = 0;
    stage(din_lo, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int rot = lane & 3;                              // rotate the channel-quad order per lane: conflict-free b128 reads
    for (int d_in = din_lo; d_in <= din_hi; ++d_in) {
        const bool v0 = d_in + 1 < D;
        const bool v2 = d_in - 1 >= od_lo && d_in - 1 < od_hi;          // output d_in-1 belongs to this segment
        const bool v1s = d_in + 1 == D && d_in >= od_lo;                // last slice of the volume: output d_in completes too
        for (int cb = 0; cb < cb_in; ++cb) {
            const bool last = cb + 1 == cb_in;
            if (!last || d_in + 1 <= din_hi) stage(last ? d_in + 1 : d_in, last ? 0 : cb + 1, bufsel ^ 1);
            const float* buf = lds + bufsel * buf_floats;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int t = kh * 3 + kw;
                    const float* w0 = w + ((0 * 9 + t) * cb_in + cb) * 16;   // wave-uniform -> scalar loads
                    const float* w1 = w + ((1 * 9 + t) * cb_in + cb) * 16;
                    const float* w2 = w + ((2 * 9 + t) * cb_in + cb) * 16;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const float* px = buf + voff[k] + (kh * Wp + kw) * 16;
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            const int q = (qq + rot) & 3;
                            const f32x4 v = *(const f32x4*)(px + q * 4);
                            const f32x4 u0 = *(const f32x4*)(w0 + q * 4), u1 = *(const f32x4*)(w1 + q * 4), u2 = *(const f32x4*)(w2 + q * 4);
                            a0[k] = fmaf(v.x, u0.x, fmaf(v.y, u0.y, fmaf(v.z, u0.z, fmaf(v.w, u0.w, a0[k]))));
                            a1[k] = fmaf(v.x, u1.x, fmaf(v.y, u1.y, fmaf(v.z, u1.z, fmaf(v.w, u1.w, a1[k]))));
                            a2[k] = fmaf(v.x, u2.x, fmaf(v.y, u2.y, fmaf(v.z, u2.z, fmaf(v.w, u2.w, a2[k]))));
                        }
                    }
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bufsel ^= 1;
        }
        // output slice d_in-1 is complete (its dd=2 contribution just arrived); the last slice also completes d_in
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const long o = (((long)n * D) * H + (oh0 + vr[k])) * W + vc[k];
            if (vok[k] && v2) { const long oo = o + (long)(d_in - 1) * H * W; out[oo] = res ? a2[k] + res[oo] : a2[k]; }
            if (vok[k] && v1s) { const long oo = o + (long)d_in * H * W; out[oo] = res ? a1[k] + res[oo] : a1[k]; }
            a2[k] = a1[k]; a1[k] = v0 ? a0[k] : 0.f; a0[k] = 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// a7: trilinear(align_corners=True) x softmax over D x soft-argmin, one thread per output pixel.
// Bilinear (y,x) resample of every coarse slice into LDS (column per thread), then one pass over the D fine
// disparities with a linear blend between neighbouring coarse slices.  Never materialises [N,D,H,W].
constexpr int kSAThreads = 128;
__global__ __launch_bounds__(kSAThreads) void upsample_softargmin_kernel(const float* __restrict__ cost, float* __restrict__ disp, int N,
                                                                         int Dp, int Hp, int Wp, int D, int H, int W, int mindisp) {
    extern __shared__ float cz[];  // [Dp][kSAThreads]
    const long total = (long)N * H * W;
    const long idx = (long)blockIdx.x * kSAThreads + threadIdx.x;
    if (idx >= total) return;
    long t = idx;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    const float sy = H > 1 ? (float)(Hp - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(Wp - 1) / (float)(W - 1) : 0.f;
    const float sd = D > 1 ? (float)(Dp - 1) / (float)(D - 1) : 0.f;
    const float fy = sy * y, fx = sx * x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hp - 1), x1 = x0 + (x0 < Wp - 1);
    const float ty = fy - y0, tx = fx - x0;
    const float* c = cost + (long)n * Dp * Hp * Wp;
    for (int k = 0; k < Dp; ++k) {
        const float* s = c + (long)k * Hp * Wp;
        const float a = s[y0 * Wp + x0] * (1.f - tx) + s[y0 * Wp + x1] * tx;
        const float b = s[y1 * Wp + x0] * (1.f - tx) + s[y1 * Wp + x1] * tx;
        cz[k * kSAThreads + threadIdx.x] = a * (1.f - ty) + b * ty;
    }
    // the softmax shift is the maximum over the FINE samples, as in the reference's softmax of the upsampled volume: with the
    // maximum of the coarse slices instead, a sharp cost column (|c| ~ 2000, neighbours of opposite sign) underflows every
    // exp(v - m) -- no fine sample lands on the coarse peak -- and the disparity becomes 0/0
    float m = -INFINITY;
    for (int d = 0; d < D; ++d) {
        const float fd = sd * d;
        const int k0 = (int)fd;
        const int k1 = k0 + (k0 < Dp - 1);
        const float td = fd - k0;
        m = fmaxf(m, cz[k0 * kSAThreads + threadIdx.x] * (1.f - td) + cz[k1 * kSAThreads + threadIdx.x] * td);
    }
    float se = 0.f, sde = 0.f;
    for (int d = 0; d < D; ++d) {
        const float fd = sd * d;
        const int k0 = (int)fd;
        const int k1 = k0 + (k0 < Dp - 1);
        const float td = fd - k0;
        const float v = cz[k0 * kSAThreads + threadIdx.x] * (1.f - td) + cz[k1 * kSAThreads + threadIdx.x] * td;
        const float e = expf(v - m);
        se += e;
        sde = fmaf(e, (float)(mindisp + d), sde);
    }
    disp[idx] = sde / se;
}

// The same head for the shapes the network produces (D = 4 D' fine disparities), round 3.  What the generic kernel above spends its time
// on -- 4 D' scattered global loads per thread (one texture request per wave each), a float -> int conversion and two LDS reads per fine
// sample and pass, expf -- is gone:
//   * a block owns 256 consecutive pixels of ONE ROI; the <= 3 coarse rows they touch are copied to LDS with coalesced row loads and
//     each thread builds its bilinear (y, x) column of D' coarse values in REGISTERS;
//   * the disparity loop is unrolled with D', D as template constants, so the coarse indices and blend weights of a fine sample
//     (k0 = (int)(sd * d), td = sd * d - k0: the generic kernel's float expressions, evaluated by the compiler) are immediates: a fine
//     sample is a multiply and an FMA, kept in a register for the second pass;
//   * exp(v - m) = 2^((v - m) * log2 e): one v_exp_f32;
//   * blocks are numbered XCD by XCD (workgroup b runs on XCD b % 8), so the blocks of a ROI share one L2: the generic kernel had every
//     XCD fetch every ROI's cost volume from HBM (rocprofv3: 308 MB fetched per launch for 38.5 MB of input at 1,024 ROIs).
template <int DP, int D>
__global__ __launch_bounds__(256) void upsample_softargmin_t_kernel(const float* __restrict__ cost, float* __restrict__ disp, int N, int Hp, int Wp,
                                                                    int H, int W, int mindisp, int blocks_per_roi) {
    extern __shared__ float rows[];                // [3 rows][DP][Wp]
    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    const int n = b / blocks_per_roi;
    const int pix0 = (b - n * blocks_per_roi) * 256;
    const int HW = H * W;
    const int pl = pix0 + 255 < HW ? pix0 + 255 : HW - 1;
    const float sy = H > 1 ? (float)(Hp - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(Wp - 1) / (float)(W - 1) : 0.f;
    const int cy0 = (int)(sy * (pix0 / W));                            // first coarse row the block touches
    int cy1 = (int)(sy * (pl / W)) + 1;
    cy1 = cy1 > Hp - 1 ? Hp - 1 : cy1;
    const int nrows = cy1 - cy0 + 1;                                   // <= 3 (the launcher checks 256 pixels span < 2 coarse rows + 1)
    const float* c = cost + (long)n * DP * Hp * Wp;
    // a (row, slice) line of Wp floats per 32 (Wp <= 32) or 64 lanes: no integer division by a run-time value in the copy loop
    const int lpr = Wp <= 32 ? 32 : 64, xl = (int)threadIdx.x & (lpr - 1);
    for (int rk = (int)threadIdx.x / lpr; rk < nrows * DP; rk += 256 / lpr) {
        const int r = rk / DP, k = rk - r * DP;
        if (xl < Wp) rows[rk * Wp + xl] = c[((long)k * Hp + cy0 + r) * Wp + xl];
    }
    __syncthreads();
    const int pix = pix0 + (int)threadIdx.x;
    if (pix >= HW) return;
    const int y = pix / W, x = pix - y * W;
    const float fy = sy * y, fx = sx * x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hp - 1), x1 = x0 + (x0 < Wp - 1);
    const float ty = fy - y0, tx = fx - x0;
    const float* r0 = rows + (y0 - cy0) * DP * Wp;
    const float* r1 = rows + (y1 - cy0) * DP * Wp;
    float cz[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) {
        const float a = r0[k * Wp + x0] * (1.f - tx) + r0[k * Wp + x1] * tx;
        const float bb = r1[k * Wp + x0] * (1.f - tx) + r1[k * Wp + x1] * tx;
        cz[k] = a * (1.f - ty) + bb * ty;
    }
    constexpr float sd = D > 1 ? (float)(DP - 1) / (float)(D - 1) : 0.f;
    float v[D];
    float m = -INFINITY;                           // the maximum over the FINE samples (see the generic kernel)
#pragma unroll
    for (int d = 0; d < D; ++d) {
        constexpr float dummy = 0.f; (void)dummy;
        const float fd = sd * (float)d;
        const int k0 = (int)fd;
        const int k1 = k0 + (k0 < DP - 1);
        const float td = fd - (float)k0;
        v[d] = cz[k0] * (1.f - td) + cz[k1] * td;
        m = fmaxf(m, v[d]);
    }
    float se = 0.f, sde = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float e = __builtin_amdgcn_exp2f((v[d] - m) * 1.44269504088896340736f);
        se += e;
        sde = fmaf(e, (float)(mindisp + d), sde);
    }
    disp[(long)n * HW + pix] = sde / se;
}

template <int DP, int D>
static int launch_softargmin_t(const float* cost, float* disp, int N, int Hp, int Wp, int H, int W, int mindisp, hipStream_t stream) {
    const int bpr = (H * W + 255) / 256;
    const long blocks = (long)N * bpr;
    hipLaunchKernelGGL((upsample_softargmin_t_kernel<DP, D>), dim3((unsigned)blocks), dim3(256), (size_t)3 * DP * Wp * sizeof(float), stream, cost, disp, N,
                       Hp, Wp, H, W, mindisp, bpr);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// SPP helpers on blocked 2D tensors (submodule.py:76-90, :120-135).
// One wave per pooled output voxel: 16 window positions x 4 channel quads per step, shuffle-reduce.
__global__ __launch_bounds__(64) void avgpool2d_blocked_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int CB, int H,
                                                               int W, int px, int k, int OH, int OW, int py, int x_cb_total, int x_cb_off) {
    long t = blockIdx.x;
    const int ow = (int)(t % OW); t /= OW;
    const int oh = (int)(t % OH); t /= OH;
    const int cb = (int)(t % CB);
    const int n = (int)(t / CB);
    const int lane = threadIdx.x, q = lane & 3, p0 = lane >> 2;
    const long iW = W + 2 * px, iH = H + 2 * px;
    const float* xb = x + (((long)n * x_cb_total + x_cb_off + cb) * iH + (oh * k + px)) * iW * 16 + (long)(ow * k + px) * 16 + q * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int p = p0; p < k * k; p += 16) {
        const int r = p / k, c = p - r * k;
        s += *(const f32x4*)(xb + ((long)r * iW + c) * 16);
    }
#pragma unroll
    for (int m = 4; m < 64; m <<= 1) {
        s.x += __shfl_xor(s.x, m); s.y += __shfl_xor(s.y, m); s.z += __shfl_xor(s.z, m); s.w += __shfl_xor(s.w, m);
    }
    if (p0 == 0) {
        const float inv = 1.0f / (float)(k * k);
        const long oW = OW + 2 * py, oH = OH + 2 * py;
        *(f32x4*)(y + ((((long)n * CB + cb) * oH + (oh + py)) * oW + (ow + py)) * 16 + q * 4) = s * inv;
    }
}

__global__ __launch_bounds__(kThreads) void bilinear_up_blocked_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int CB,
                                                                       int IH, int IW, int px, int OH, int OW, int py, int y_cb_total,
                                                                       int y_cb_off, int align_corners) {
    const long total = (long)N * CB * OH * OW * 4;
    const long iW = IW + 2 * px, iH = IH + 2 * px, oW = OW + 2 * py, oH = OH + 2 * py;
    // align_corners=True: src = dst*(in-1)/(out-1);  False: src = max((dst+0.5)*in/out - 0.5, 0)  (ATen area_pixel_compute_source_index)
    const float sy = align_corners ? (OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f) : (float)IH / (float)OH;
    const float sx = align_corners ? (OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f) : (float)IW / (float)OW;
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
        long t = idx;
        const int q = (int)(t & 3); t >>= 2;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH); t /= OH;
        const int cb = (int)(t % CB);
        const int n = (int)(t / CB);
        const float fy = align_corners ? sy * oy : fmaxf(sy * (oy + 0.5f) - 0.5f, 0.f);
        const float fx = align_corners ? sx * ox : fmaxf(sx * (ox + 0.5f) - 0.5f, 0.f);
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < IH - 1), x1 = x0 + (x0 < IW - 1);
        const float ty = fy - y0, tx = fx - x0;
        const float* xb = x + (((long)n * CB + cb) * iH) * iW * 16 + q * 4;
        const f32x4 v00 = *(const f32x4*)(xb + ((long)(y0 + px) * iW + (x0 + px)) * 16);
        const f32x4 v01 = *(const f32x4*)(xb + ((long)(y0 + px) * iW + (x1 + px)) * 16);
        const f32x4 v10 = *(const f32x4*)(xb + ((long)(y1 + px) * iW + (x0 + px)) * 16);
        const f32x4 v11 = *(const f32x4*)(xb + ((long)(y1 + px) * iW + (x1 + px)) * 16);
        // same association as ATen's upsample_bilinear2d: h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11)
        const f32x4 v = (1.f - ty) * ((1.f - tx) * v00 + tx * v01) + ty * ((1.f - tx) * v10 + tx * v11);
        *(f32x4*)(y + ((((long)n * y_cb_total + y_cb_off + cb) * oH + (oy + py)) * oW + (ox + py)) * 16 + q * 4) = v;
    }
}

// max_pool2d(k, stride, padding 0, ceil_mode) on blocked 2D tensors: windows are clipped to the valid region
// (reference: BaseStem max_pool2d(3,2,0,ceil_mode=True), resnet.py:303; LastLevelMaxPool max_pool2d(1,2,0), fpn.py:80-82)
__global__ __launch_bounds__(kThreads) void maxpool2d_blocked_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int CB, int H,
                                                                     int W, int px, int k, int stride, int OH, int OW, int py) {
    const long total = (long)N * CB * OH * OW * 4;
    const long iW = W + 2 * px, iH = H + 2 * px, oW = OW + 2 * py, oH = OH + 2 * py;
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
        long t = idx;
        const int q = (int)(t & 3); t >>= 2;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH); t /= OH;
        const int cb = (int)(t % CB);
        const int n = (int)(t / CB);
        const float* xb = x + (((long)n * CB + cb) * iH) * iW * 16 + q * 4;
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int a = 0; a < k; ++a) {
            const int iy = oy * stride + a;
            if (iy >= H) break;
            for (int b = 0; b < k; ++b) {
                const int ix = ox * stride + b;
                if (ix >= W) break;
                const f32x4 v = *(const f32x4*)(xb + ((long)(iy + px) * iW + (ix + px)) * 16);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        *(f32x4*)(y + ((((long)n * CB + cb) * oH + (oy + py)) * oW + (ox + py)) * 16 + q * 4) = m;
    }
}

__global__ __launch_bounds__(kThreads) void copy_blocks_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int CB,
                                                               long vox_per_cb, int y_cb_total, int y_cb_off) {
    const long per = vox_per_cb * 4;
    const long total = (long)N * CB * per;
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
        const long e = idx % per;
        long t = idx / per;
        const int cb = (int)(t % CB);
        const int n = (int)(t / CB);
        *(f32x4*)(y + (((long)n * y_cb_total + y_cb_off + cb) * per + e) * 4) = *(const f32x4*)(x + idx * 4);
    }
}

inline int done() { return (int)hipGetLastError(); }

}  // namespace

// weights [A][B][K] (Conv: A = Cout, B = Cin; ConvTranspose: A = Cin, B = Cout) -> the engine's two packings in one launch,
// zero-padded to whole channel blocks:  tap [K][cb][2 halves][cout_pad][8]  and  t16 [K][cb][cout_pad][16]
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, int cout, int cin, int K, int transposed, int flip,
                                                           float* __restrict__ out_tap, float* __restrict__ out_t16) {
    const int cb_n = (cin + 15) / 16, cout_pad = (cout + 15) / 16 * 16;
    const long total = (long)K * cb_n * cout_pad * 16;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        long t = idx;
        const int c = (int)(t & 15); t >>= 4;
        const int co = (int)(t % cout_pad); t /= cout_pad;
        const int cb = (int)(t % cb_n);
        const int k = (int)(t / cb_n);
        const int ci = cb * 16 + c;
        float v = 0.f;
        if (co < cout && ci < cin) {
            const int ks = flip ? K - 1 - k : k;
            v = transposed ? w[((long)ci * cout + co) * K + ks] : w[((long)co * cin + ci) * K + ks];
        }
        if (out_t16) out_t16[idx] = v;
        if (out_tap) out_tap[((((long)k * cb_n + cb) * 2 + (c >> 3)) * cout_pad + co) * 8 + (c & 7)] = v;
    }
}

extern "C" int drc_conv3d_cout1_mfma_try(const float* x, const float* w, const float* res, float* out, int N, int cb_in, int D, int H, int W,
                                         void* stream);   // cout1_mfma.hip

extern "C" {

const char* drc_version(void) { return "disprcnn_hip gfx950 abi1"; }

int drc_cost_volume_fwd(const float* left, const float* right, float* cost, int N, int C, int Dp, int Hp, int Wp, int lo4, int hi4,
                        void* stream) {
    if (N < 0 || C <= 0 || Dp < 0 || Hp <= 0 || Wp <= 0) return -2;
    const long total = (long)N * 2 * C * Dp * Hp * Wp;
    if (total == 0) return 0;
    if (!left || !right || !cost) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (Wp % 4 == 0 && ((uintptr_t)cost & 15) == 0)
        hipLaunchKernelGGL(cost_volume_dense_kernel<true>, dim3(grid_for(total / 4)), dim3(kThreads), 0, s, left, right, cost, N, C, Dp, Hp, Wp, lo4, hi4);
    else
        hipLaunchKernelGGL(cost_volume_dense_kernel<false>, dim3(grid_for(total)), dim3(kThreads), 0, s, left, right, cost, N, C, Dp, Hp, Wp, lo4, hi4);
    return done();
}

int drc_cost_volume_bwd(const float* gcost, float* gleft, float* gright, int N, int C, int Dp, int Hp, int Wp, int lo4, int hi4,
                        void* stream) {
    if (N < 0 || C <= 0 || Dp < 0 || Hp <= 0 || Wp <= 0) return -2;
    const long total = 2L * N * C * Hp * Wp;
    if (total == 0) return 0;
    if (!gcost || !gleft || !gright) return -1;
    hipLaunchKernelGGL(cost_volume_bwd_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, gcost, gleft, gright, N, C, Dp, Hp, Wp, lo4, hi4);
    return done();
}

int drc_cost_volume_blocked_fwd(const float* left, const float* right, float* cost_blk, int N, int C, int Dp, int Hp, int Wp, int lo4,
                                int hi4, int in_blocked_pad, void* stream) {
    if (N < 0 || C <= 0 || (C & 15) || Dp < 0 || Hp <= 0 || Wp <= 0 || in_blocked_pad < 0) return -2;
    const long total = (long)N * (2 * C / 16) * Dp * Hp * Wp * 4;
    if (total == 0) return 0;
    if (!left || !right || !cost_blk) return -1;
    const size_t lds = (size_t)2 * Wp * (C + 1) * sizeof(float);
    if (in_blocked_pad == 0 && lds <= 64 * 1024 && (long)N * Hp < (1L << 31)) {        // NCHW features: row-transposing kernel
        hipLaunchKernelGGL(cost_volume_blocked_rows_kernel, dim3((unsigned)(N * Hp)), dim3(kThreads), lds, (hipStream_t)stream, left, right, cost_blk, C,
                           Dp, Hp, Wp, lo4, hi4);
        return done();
    }
    hipLaunchKernelGGL(cost_volume_blocked_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, left, right, cost_blk, N, C, Dp, Hp, Wp, lo4, hi4, in_blocked_pad);
    return done();
}

int drc_dense_to_blocked(const float* dense, float* blk, int N, int C, int D, int H, int W, int pd, int ph, int pw, void* stream) {
    if (N < 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || pd < 0 || ph < 0 || pw < 0) return -2;
    const long total = (long)N * ((C + 15) / 16) * D * H * W * 4;
    if (total == 0) return 0;
    if (!dense || !blk) return -1;
    const size_t lds = (size_t)W * (((C + 15) / 16) * 16 + 1) * sizeof(float);
    const long lines = (long)N * D * H;
    if (W >= 8 && lds <= 64 * 1024 && lines < (1L << 31))
        hipLaunchKernelGGL(dense_to_blocked_rows_kernel, dim3((unsigned)lines), dim3(kThreads), lds, (hipStream_t)stream, dense, blk, C, D, H, W, pd, ph, pw);
    else
        hipLaunchKernelGGL(dense_to_blocked_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, dense, blk, N, C, D, H, W, pd, ph, pw);
    return done();
}

int drc_blocked_to_dense(const float* blk, float* dense, int N, int C, int D, int H, int W, int pd, int ph, int pw, void* stream) {
    if (N < 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || pd < 0 || ph < 0 || pw < 0) return -2;
    const long total = (long)N * C * D * H * W;
    if (total == 0) return 0;
    if (!dense || !blk) return -1;
    hipLaunchKernelGGL(blocked_to_dense_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, blk, dense, N, C, D, H, W, pd, ph, pw);
    return done();
}

int drc_conv3d_cout1_fwd(const float* x, const float* w, const float* res, float* out, int N, int cb_in, int D, int H, int W,
                         void* stream) {
    if (N < 0 || cb_in <= 0 || D <= 0 || H <= 0 || W <= 0) return -2;
    const long total = (long)N * D * H * W * 4;
    if (total == 0) return 0;
    if (!x || !w || !out) return -1;
    {   // 32 input channels (the reference's classifier): 1x1x1 GEMM on the MFMA + shifted sum (cout1_mfma.hip)
        const int st = drc_conv3d_cout1_mfma_try(x, w, res, out, N, cb_in, D, H, W, stream);
        if (st != 1) return st;
    }
    // sliding-window kernel when a wave's R x W tile (<= 128 voxels, 2 per lane) and its two LDS tiles fit; else the direct one
    int R = 128 / (W > 0 ? W : 1);
    if (R > H) R = H;
    while (R > 1 && H % R && (H + R - 1) / R * R - H > R / 2) --R;     // avoid a mostly empty last row tile
    const size_t lds = R >= 1 ? (size_t)kC1Waves * 2 * (R + 2) * (W + 2) * 64 : 0;
    if (W <= 64 && R >= 1 && lds <= 160 * 1024 && D >= 2) {
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)conv3d_cout1_slide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_done = true;
        }
        const long cols = (long)N * ((H + R - 1) / R);
        int nseg = (int)(1024 / (cols > 0 ? cols : 1));                // split D only while SIMDs would otherwise sit idle
        if (nseg > D / 3) nseg = D / 3;
        if (nseg < 1) nseg = 1;
        const int seg_len = (D + nseg - 1) / nseg;
        nseg = (D + seg_len - 1) / seg_len;
        hipLaunchKernelGGL(conv3d_cout1_slide_kernel, dim3((unsigned)((cols + kC1Waves - 1) / kC1Waves), (unsigned)nseg), dim3(64 * kC1Waves), lds,
                           (hipStream_t)stream, x, w, res, out, N, cb_in, D, H, W, R, seg_len);
        return done();
    }
    hipLaunchKernelGGL(conv3d_cout1_kernel, dim3(grid_for(total, 256 * 32)), dim3(kThreads), 0, (hipStream_t)stream, x, w, res, out, N, cb_in, D, H, W);
    return done();
}

int drc_upsample_softargmin_fwd(const float* cost, float* disp, int N, int Dp, int Hp, int Wp, int D, int H, int W, int mindisp,
                                void* stream) {
    if (N < 0 || Dp <= 0 || Hp <= 0 || Wp <= 0 || D <= 0 || H <= 0 || W <= 0) return -2;
    if (Dp > 96) return -3;  // LDS column per thread: Dp * 128 * 4 B
    const long total = (long)N * H * W;
    if (total == 0) return 0;
    if (!cost || !disp) return -1;
    // the network's shapes (D = 4 D'): the register-column kernel, when 256 consecutive pixels touch at most 3 coarse rows and those fit LDS
    if (D == 4 * Dp && H > 1 && W >= 86 && Wp <= 64 && (long)N * ((H * W + 255) / 256) < (1L << 31) && (size_t)3 * Dp * Wp * 4 <= 64 * 1024) {
        // 256 pixels span at most ceil(255 / W) + 1 <= 4 fine rows = a coarse extent of < 3 * sy + 1 < 2 rows + the bilinear neighbour
        const float sy = (float)(Hp - 1) / (float)(H - 1);
        if (sy * (float)((255 + W - 1) / W) < 1.f) {
            hipStream_t s = (hipStream_t)stream;
            if (Dp == 12) return launch_softargmin_t<12, 48>(cost, disp, N, Hp, Wp, H, W, mindisp, s);
            if (Dp == 24) return launch_softargmin_t<24, 96>(cost, disp, N, Hp, Wp, H, W, mindisp, s);
            if (Dp == 6) return launch_softargmin_t<6, 24>(cost, disp, N, Hp, Wp, H, W, mindisp, s);
        }
    }
    const unsigned blocks = (unsigned)((total + kSAThreads - 1) / kSAThreads);
    hipLaunchKernelGGL(upsample_softargmin_kernel, dim3(blocks), dim3(kSAThreads), (size_t)Dp * kSAThreads * 4, (hipStream_t)stream, cost, disp, N, Dp, Hp, Wp, D, H, W, mindisp);
    return done();
}

int drc_avgpool2d_blocked(const float* x, float* y, int N, int CB, int H, int W, int px, int k, int OH, int OW, int py, void* stream) {
    if (N < 0 || CB <= 0 || H <= 0 || W <= 0 || k <= 0 || OH <= 0 || OW <= 0 || OH * k > H || OW * k > W) return -2;
    const long blocks = (long)N * CB * OH * OW;
    if (blocks == 0) return 0;
    if (!x || !y) return -1;
    hipLaunchKernelGGL(avgpool2d_blocked_kernel, dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, x, y, N, CB, H, W, px, k, OH, OW, py, CB, 0);
    return done();
}

int drc_pack_weights(const float* w, int cout, int cin, int K, int transposed, int flip, float* out_tap, float* out_t16, void* stream) {
    if (cout <= 0 || cin <= 0 || K <= 0) return -2;
    if (!w || (!out_tap && !out_t16)) return -1;
    const long total = (long)K * ((cin + 15) / 16) * ((cout + 15) / 16 * 16) * 16;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid_for(total, 2048)), dim3(256), 0, (hipStream_t)stream, w, cout, cin, K, transposed, flip,
                       out_tap, out_t16);
    return done();
}

int drc_avgpool2d_blocked_slice(const float* x, float* y, int N, int CB, int H, int W, int px, int k, int OH, int OW, int py, int x_cb_total,
                                int x_cb_off, void* stream) {
    if (N < 0 || CB <= 0 || H <= 0 || W <= 0 || k <= 0 || OH <= 0 || OW <= 0 || OH * k > H || OW * k > W) return -2;
    if (x_cb_off < 0 || x_cb_off + CB > x_cb_total) return -2;
    const long blocks = (long)N * CB * OH * OW;
    if (blocks == 0) return 0;
    if (!x || !y) return -1;
    hipLaunchKernelGGL(avgpool2d_blocked_kernel, dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, x, y, N, CB, H, W, px, k, OH, OW, py,
                       x_cb_total, x_cb_off);
    return done();
}

int drc_bilinear_up_blocked(const float* x, float* y, int N, int CB, int IH, int IW, int px, int OH, int OW, int py, int y_cb_total,
                            int y_cb_off, void* stream) {
    if (N < 0 || CB <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || y_cb_off < 0 || y_cb_off + CB > y_cb_total) return -2;
    const long total = (long)N * CB * OH * OW * 4;
    if (total == 0) return 0;
    if (!x || !y) return -1;
    hipLaunchKernelGGL(bilinear_up_blocked_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, x, y, N, CB, IH, IW, px, OH, OW, py, y_cb_total, y_cb_off, 1);
    return done();
}

int drc_bilinear_resize_blocked(const float* x, float* y, int N, int CB, int IH, int IW, int px, int OH, int OW, int py, int y_cb_total,
                                int y_cb_off, int align_corners, void* stream) {
    if (N < 0 || CB <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || y_cb_off < 0 || y_cb_off + CB > y_cb_total) return -2;
    const long total = (long)N * CB * OH * OW * 4;
    if (total == 0) return 0;
    if (!x || !y) return -1;
    hipLaunchKernelGGL(bilinear_up_blocked_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, x, y, N, CB, IH, IW, px, OH, OW, py, y_cb_total, y_cb_off, align_corners ? 1 : 0);
    return done();
}

int drc_maxpool2d_blocked(const float* x, float* y, int N, int CB, int H, int W, int px, int k, int stride, int OH, int OW, int py,
                          void* stream) {
    if (N < 0 || CB <= 0 || H <= 0 || W <= 0 || k <= 0 || stride <= 0 || OH <= 0 || OW <= 0) return -2;
    if ((OH - 1) * stride >= H || (OW - 1) * stride >= W) return -2;   // every window must start inside the input
    const long total = (long)N * CB * OH * OW * 4;
    if (total == 0) return 0;
    if (!x || !y) return -1;
    hipLaunchKernelGGL(maxpool2d_blocked_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, x, y, N, CB, H, W, px, k, stride, OH, OW, py);
    return done();
}

int drc_copy_blocks(const float* x, float* y, int N, int CB, int64_t vox_per_cb, int y_cb_total, int y_cb_off, void* stream) {
    if (N < 0 || CB <= 0 || vox_per_cb <= 0 || y_cb_off < 0 || y_cb_off + CB > y_cb_total) return -2;
    const long total = (long)N * CB * vox_per_cb * 4;
    if (total == 0) return 0;
    if (!x || !y) return -1;
    hipLaunchKernelGGL(copy_blocks_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, x, y, N, CB, (long)vox_per_cb, y_cb_total, y_cb_off);
    return done();
}

}  // extern "C"
