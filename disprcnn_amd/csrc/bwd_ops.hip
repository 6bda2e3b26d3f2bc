// bwd_ops.hip -- backward kernels of the disparity path (gfx950): soft-argmin / trilinear adjoint, the 32->1 classifier
// conv, training-mode BatchNorm.  (The data gradients of the MFMA convolutions reuse the forward engine: the dgrad of a
// stride-1 conv is a stride-1 conv with flipped, transposed weights; the dgrad of a stride-2 conv IS the transposed-conv
// parity-class list and vice versa -- see disprcnn_amd/autograd.py.  Weight gradients: wgrad.hip.)
// Reference semantics: autograd of stackhourglass.py:130-174, submodule.py:19-22,51-57.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"
#include "det_reduce.h"
#include "blocked_walk.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kThreads = 256;

inline unsigned grid_for(long work, long cap = 4096) {
    long b = (work + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

using drc_blk::BlkGeom;
using drc_blk::blk_off;
inline bool geom_ok(const int* g) { return g && g[0] >= 0 && g[1] > 0 && g[2] > 0 && g[3] > 0 && g[4] > 0 && g[5] >= 0 && g[6] >= 0 && g[7] >= 0 && g[9] >= 0 && g[9] + g[1] <= g[8]; }
inline BlkGeom to_geom(const int* g) { return BlkGeom{g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9]}; }

// ------------------------------------------------------------------------------------------------ a7 backward
// disp = sum_d p_d * dval_d, p = softmax_d(c_up), c_up = trilinear(cost).  d disp / d c_up[d] = p_d (dval_d - disp).
// One thread per output pixel, one block per 8 x 16 pixel tile of an image: recompute the LDS column of bilinear-resampled
// coarse slices, fold the D fine gradients back onto the coarse slices (lerp weights) -> gz[D'][pixel].  The pixels' D' values are
// then gathered onto the tile's coarse footprint (a few cells x D') SEPARABLY and without atomics -- along x into tmp[D'][8 rows][CW],
// along y into the footprint; bilinear weights come from per-row / per-column tables.  Neighbouring tiles share their border cells,
// so (round 3) the tile STORES its footprint [Dp][CH][CW] to scratch and softargmin_bwd_gather_kernel adds, per coarse cell, the
// footprints of the tiles that contain it in tile order: no atomicAdd anywhere, the gradient is bit-reproducible run to run (it feeds
// every upstream gradient, and through the BatchNorm statistics the ReLU masks).  (Round 2: one global atomicAdd per footprint cell;
// before that 48 LDS atomicAdds per pixel onto ~24 cells per slice, ~20-way contended: 303 us per call at 64 ROIs.)
constexpr int kSAThreads = 128, kSATX = 16, kSATY = 8;
__global__ __launch_bounds__(kSAThreads) void upsample_softargmin_bwd_kernel(const float* __restrict__ cost, const float* __restrict__ gdisp,
                                                                             float* __restrict__ foot, int N, int Dp, int Hp, int Wp,
                                                                             int D, int H, int W, int mindisp, int CH, int CW) {
    extern __shared__ float sm[];                      // cz [Dp][T], gz [Dp][T], tmp [Dp][kSATY][CW]
    float* cz = sm;
    float* gz = sm + Dp * kSAThreads;
    float* tmp = gz + Dp * kSAThreads;
    __shared__ int row_l[kSATY][2], col_l[kSATX][2];
    __shared__ float row_t[kSATY], col_t[kSATX];
    const int tiles_x = (W + kSATX - 1) / kSATX, tiles_y = (H + kSATY - 1) / kSATY;
    int bt = blockIdx.x;
    const int bx = bt % tiles_x; bt /= tiles_x;
    const int by = bt % tiles_y;
    const int n = bt / tiles_y;
    const int txi = threadIdx.x % kSATX, tyi = threadIdx.x / kSATX;
    const int x = bx * kSATX + txi, y = by * kSATY + tyi;
    const bool live = x < W && y < H;
    const float sy = H > 1 ? (float)(Hp - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(Wp - 1) / (float)(W - 1) : 0.f;
    const float sd = D > 1 ? (float)(Dp - 1) / (float)(D - 1) : 0.f;
    const int cy0 = (int)(sy * (by * kSATY)), cx0 = (int)(sx * (bx * kSATX));     // footprint origin (coarse)
    const float fy = sy * y, fx = sx * x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hp - 1), x1 = x0 + (x0 < Wp - 1);
    const float ty = fy - y0, tx = fx - x0;
    if (txi == 0) { row_l[tyi][0] = y < H ? y0 - cy0 : -1000; row_l[tyi][1] = y < H ? y1 - cy0 : -1000; row_t[tyi] = ty; }
    if (tyi == 0) { col_l[txi][0] = x < W ? x0 - cx0 : -1000; col_l[txi][1] = x < W ? x1 - cx0 : -1000; col_t[txi] = tx; }
    if (live) {
        const float* c = cost + (long)n * Dp * Hp * Wp;
        for (int k = 0; k < Dp; ++k) {
            const float* s = c + (long)k * Hp * Wp;
            const float a = s[y0 * Wp + x0] * (1.f - tx) + s[y0 * Wp + x1] * tx;
            const float b = s[y1 * Wp + x0] * (1.f - tx) + s[y1 * Wp + x1] * tx;
            cz[k * kSAThreads + threadIdx.x] = a * (1.f - ty) + b * ty;
            gz[k * kSAThreads + threadIdx.x] = 0.f;
        }
        float m = -INFINITY;                            // maximum over the FINE samples (see upsample_softargmin_kernel)
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
            const float e = expf(cz[k0 * kSAThreads + threadIdx.x] * (1.f - td) + cz[k1 * kSAThreads + threadIdx.x] * td - m);
            se += e; sde = fmaf(e, (float)(mindisp + d), sde);
        }
        const float disp = sde / se, g = gdisp[((long)n * H + y) * W + x] / se;
        for (int d = 0; d < D; ++d) {
            const float fd = sd * d;
            const int k0 = (int)fd;
            const int k1 = k0 + (k0 < Dp - 1);
            const float td = fd - k0;
            const float e = expf(cz[k0 * kSAThreads + threadIdx.x] * (1.f - td) + cz[k1 * kSAThreads + threadIdx.x] * td - m);
            const float gv = g * e * ((float)(mindisp + d) - disp);
            gz[k0 * kSAThreads + threadIdx.x] += gv * (1.f - td);
            gz[k1 * kSAThreads + threadIdx.x] += gv * td;
        }
    } else {
        for (int k = 0; k < Dp; ++k) gz[k * kSAThreads + threadIdx.x] = 0.f;
    }
    __syncthreads();
    // along x: tmp[k][row][cx] = sum over the row's 16 pixels of gz * wx(pixel, cx)
    for (int o = threadIdx.x; o < Dp * kSATY * CW; o += kSAThreads) {
        const int cxl = o % CW, r = (o / CW) % kSATY, k = o / (CW * kSATY);
        const float* gr = gz + k * kSAThreads + r * kSATX;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < kSATX; ++c) {
            const float w = (col_l[c][0] == cxl ? 1.f - col_t[c] : 0.f) + (col_l[c][1] == cxl ? col_t[c] : 0.f);
            acc += gr[c] * w;
        }
        tmp[o] = acc;
    }
    __syncthreads();
    // along y, and store the footprint (every cell, zeros included: the gather reads all of it)
    float* fo = foot + (long)blockIdx.x * Dp * CH * CW;
    for (int o = threadIdx.x; o < Dp * CH * CW; o += kSAThreads) {
        const int cxl = o % CW, cyl = (o / CW) % CH, k = o / (CW * CH);
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < kSATY; ++r) {
            const float w = (row_l[r][0] == cyl ? 1.f - row_t[r] : 0.f) + (row_l[r][1] == cyl ? row_t[r] : 0.f);
            acc += tmp[(k * kSATY + r) * CW + cxl] * w;
        }
        fo[o] = acc;
    }
}

// gcost[n][k][yy][xx] = sum over the tiles (by, bx) whose footprint [cy0, cy0+CH) x [cx0, cx0+CW) contains (yy, xx), by ascending then bx
// ascending; cy0 / cx0 are recomputed with the tile kernel's own float expression.  thread = one coarse cell (all of it: overwrite).
__global__ __launch_bounds__(kThreads) void softargmin_bwd_gather_kernel(const float* __restrict__ foot, float* __restrict__ gcost, int N, int Dp,
                                                                         int Hp, int Wp, int H, int W, int CH, int CW) {
    const int tiles_x = (W + kSATX - 1) / kSATX, tiles_y = (H + kSATY - 1) / kSATY;
    const float sy = H > 1 ? (float)(Hp - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(Wp - 1) / (float)(W - 1) : 0.f;
    const long total = (long)N * Dp * Hp * Wp;
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
        long t = idx;
        const int xx = (int)(t % Wp); t /= Wp;
        const int yy = (int)(t % Hp); t /= Hp;
        const int k = (int)(t % Dp);
        const int n = (int)(t / Dp);
        // candidate tile rows / columns: cy0(by) = (int)(sy * 8 by) is non-decreasing in by, so the tiles containing yy are a range
        int by0 = 0, by1 = tiles_y - 1, bx0 = 0, bx1 = tiles_x - 1;
        if (sy > 0.f) {
            by0 = (int)((float)(yy - CH) / (sy * kSATY)) - 1; by1 = (int)((float)(yy + 1) / (sy * kSATY)) + 1;
            by0 = by0 < 0 ? 0 : by0; by1 = by1 > tiles_y - 1 ? tiles_y - 1 : by1;
        }
        if (sx > 0.f) {
            bx0 = (int)((float)(xx - CW) / (sx * kSATX)) - 1; bx1 = (int)((float)(xx + 1) / (sx * kSATX)) + 1;
            bx0 = bx0 < 0 ? 0 : bx0; bx1 = bx1 > tiles_x - 1 ? tiles_x - 1 : bx1;
        }
        float acc = 0.f;
        for (int by = by0; by <= by1; ++by) {
            const int cyl = yy - (int)(sy * (by * kSATY));
            if (cyl < 0 || cyl >= CH) continue;
            for (int bx = bx0; bx <= bx1; ++bx) {
                const int cxl = xx - (int)(sx * (bx * kSATX));
                if (cxl < 0 || cxl >= CW) continue;
                acc += foot[((((long)n * tiles_y + by) * tiles_x + bx) * Dp + k) * CH * CW + cyl * CW + cxl];
            }
        }
        gcost[idx] = acc;
    }
}

// ------------------------------------------------------------------------------------------------ classifN[2] backward
// data: gx[n,c,u] (=|+=) sum_t w[t][c] * gy[n, u - t + 1]   (gather form, zero outside);  thread = (voxel u, channel quad)
__global__ __launch_bounds__(kThreads) void cout1_bwd_data_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                                  float* __restrict__ gx, int N, int cb_in, int D, int H, int W,
                                                                  int accumulate) {
    const long total = (long)N * cb_in * D * H * W * 4;
    const BlkGeom g{N, cb_in, D, H, W, 1, 1, 1, cb_in, 0};
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
        long t = idx;
        const int q = (int)(t & 3); t >>= 2;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H); t /= H;
        const int d = (int)(t % D); t /= D;
        const int cb = (int)(t % cb_in);
        const int n = (int)(t / cb_in);
        const float* gyn = gy + (long)n * D * H * W;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const int od = d - kd + 1;
            if (od < 0 || od >= D) continue;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int oh = y - kh + 1;
                if (oh < 0 || oh >= H) continue;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int ow = x - kw + 1;
                    if (ow < 0 || ow >= W) continue;
                    const float gv = gyn[((long)od * H + oh) * W + ow];
                    acc += gv * *(const f32x4*)(w + (((kd * 3 + kh) * 3 + kw) * cb_in + cb) * 16 + q * 4);
                }
            }
        }
        float* dst = gx + blk_off(g, n, cb, d, y, x) + q * 4;
        if (accumulate) acc += *(const f32x4*)dst;
        *(f32x4*)dst = acc;
    }
}

// weight: gw[t][c] = sum_{n,v} x[n,c,v+t] * gy[n,v];  thread = (voxel, quad) strided, 27 float4 partials, block reduce; the block's
// 27 x 16 sums go to part[block][cb][27][16] and cout1_bwd_weight_finish_kernel adds the blocks in order (no atomicAdd)
__global__ __launch_bounds__(kThreads) void cout1_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                    float* __restrict__ part, int N, int cb_in, int D, int H, int W) {
    const int cb = blockIdx.y;
    const int q = threadIdx.x & 3;
    const BlkGeom g{N, cb_in, D, H, W, 1, 1, 1, cb_in, 0};
    const long nvox = (long)N * D * H * W;
    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long sH = (long)(W + 2) * 16, sD = (long)(H + 2) * sH;
    for (long v = (long)blockIdx.x * (kThreads / 4) + (threadIdx.x >> 2); v < nvox; v += (long)gridDim.x * (kThreads / 4)) {
        long t = v;
        const int xx = (int)(t % W); t /= W;
        const int yy = (int)(t % H); t /= H;
        const int dd = (int)(t % D);
        const int n = (int)(t / D);
        const float gv = gy[v];
        const float* xb = x + blk_off(g, n, cb, dd, yy, xx) - sD - sH - 16 + q * 4;    // tap (0,0,0) = voxel - 1 in every dim
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    acc[(kd * 3 + kh) * 3 + kw] += gv * *(const f32x4*)(xb + kd * sD + kh * sH + kw * 16);
    }
    __shared__ float red[kThreads / 64][4][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int t = 0; t < 27; ++t) {
        float r[4] = {acc[t].x, acc[t].y, acc[t].z, acc[t].w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int m = 4; m < 64; m <<= 1) r[k] += __shfl_xor(r[k], m);
        if (lane < 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) red[wv][lane][k] = r[k];
        }
        __syncthreads();
        if (threadIdx.x < 16) {
            const int qq = threadIdx.x >> 2, k = threadIdx.x & 3;
            float v = 0.f;
            for (int i = 0; i < kThreads / 64; ++i) v += red[i][qq][k];
            part[(((long)blockIdx.x * cb_in + cb) * 27 + t) * 16 + qq * 4 + k] = v;
        }
        __syncthreads();
    }
}

// gw[t][cb*16 + c] = sum_blocks part[block][cb][t][c]: one 64-lane wave per output, lane l takes blocks l, l+64, ... in order, fixed xor-tree
__global__ __launch_bounds__(kThreads) void cout1_bwd_weight_finish_kernel(const float* __restrict__ part, int nblocks, int cb_in,
                                                                           float* __restrict__ gw) {
    const int o = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= 27 * cb_in * 16) return;
    const int c = o & 15, cb = (o >> 4) % cb_in, t = (o >> 4) / cb_in;
    float v = 0.f;
    for (int i = lane; i < nblocks; i += 64) v += part[(((long)i * cb_in + cb) * 27 + t) * 16 + c];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if (lane == 0) gw[(t * cb_in + cb) * 16 + c] = v;
}

// ------------------------------------------------------------------------------------------------ BatchNorm backward
// dz = dy * [y > 0] (if relu);  sums[0][c] += sum dz,  sums[1][c] += sum dz * xhat,  xhat = (raw - mean) * invstd
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_kernel(const float* __restrict__ dy, BlkGeom gdy, const float* __restrict__ y,
                                                                 BlkGeom gy, const float* __restrict__ raw, BlkGeom graw,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd, int relu,
                                                                 float* __restrict__ sums, float* scratch) {
    const int cb = blockIdx.y;
    const int q = threadIdx.x & 3;
    const f32x4 mu = *(const f32x4*)(mean + cb * 16 + q * 4), is = *(const f32x4*)(invstd + cb * 16 + q * 4);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    drc_blk::walk_rows<kThreads>(gdy, [&](int n, int dd, int yy, int xx, int qq) {
        f32x4 dz = *(const f32x4*)(dy + blk_off(gdy, n, cb, dd, yy, xx) + qq * 4);
        if (relu) {
            const f32x4 yv = *(const f32x4*)(y + blk_off(gy, n, cb, dd, yy, xx) + qq * 4);
            dz.x = yv.x > 0.f ? dz.x : 0.f; dz.y = yv.y > 0.f ? dz.y : 0.f; dz.z = yv.z > 0.f ? dz.z : 0.f; dz.w = yv.w > 0.f ? dz.w : 0.f;
        }
        const f32x4 xh = (*(const f32x4*)(raw + blk_off(graw, n, cb, dd, yy, xx) + qq * 4) - mu) * is;
        s1 += dz; s2 += dz * xh;
    });
    float r[8] = {s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w};
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int m = 4; m < 64; m <<= 1) r[k] += __shfl_xor(r[k], m);
    __shared__ float red[kThreads / 64][4][8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane < 4) {
#pragma unroll
        for (int k = 0; k < 8; ++k) red[w][lane][k] = r[k];
    }
    __syncthreads();
    const int qq = (threadIdx.x >> 3) & 3, k = threadIdx.x & 7;
    float v = 0.f;
    if (threadIdx.x < 32)
        for (int i = 0; i < kThreads / 64; ++i) v += red[i][qq][k];
    drc_det::finish(v, cb, gdy.CB, cb * 16 + qq * 4 + (k & 3), k >> 2, sums, scratch);
}

// draw = gamma * invstd * (dz - sum_dz/M - xhat * sum_dzx/M);  optionally dres (=|+=) dz
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(const float* __restrict__ dy, BlkGeom gdy, const float* __restrict__ y,
                                                                BlkGeom gy, const float* __restrict__ raw, BlkGeom graw,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ sums, float invM,
                                                                int relu, float* __restrict__ draw, BlkGeom gdraw, float* __restrict__ dres,
                                                                BlkGeom gdres, int dres_accumulate) {
    const int cb = blockIdx.y;
    const int c = cb * 16 + (threadIdx.x & 3) * 4;
    const f32x4 is = *(const f32x4*)(invstd + c), mu = *(const f32x4*)(mean + c), ga = *(const f32x4*)(gamma + c);
    const f32x4 s1 = *(const f32x4*)(sums + c), s2 = *(const f32x4*)(sums + gdy.CB * 16 + c);
    drc_blk::walk_rows<kThreads>(gdy, [&](int n, int dd, int yy, int xx, int q) {
        f32x4 dz = *(const f32x4*)(dy + blk_off(gdy, n, cb, dd, yy, xx) + q * 4);
        if (relu) {
            const f32x4 yv = *(const f32x4*)(y + blk_off(gy, n, cb, dd, yy, xx) + q * 4);
            dz.x = yv.x > 0.f ? dz.x : 0.f; dz.y = yv.y > 0.f ? dz.y : 0.f; dz.z = yv.z > 0.f ? dz.z : 0.f; dz.w = yv.w > 0.f ? dz.w : 0.f;
        }
        if (dres) {
            float* dr = dres + blk_off(gdres, n, cb, dd, yy, xx) + q * 4;
            *(f32x4*)dr = dres_accumulate ? *(const f32x4*)dr + dz : dz;
        }
        const f32x4 xh = (*(const f32x4*)(raw + blk_off(graw, n, cb, dd, yy, xx) + q * 4) - mu) * is;
        *(f32x4*)(draw + blk_off(gdraw, n, cb, dd, yy, xx) + q * 4) = ga * is * (dz - s1 * invM - xh * s2 * invM);
    });
}


// ------------------------------------------------------------------------------------------------ SPP backward (2D)
// adjoint of bilinear (align_corners=True) upsampling: every fine pixel scatters to its 4 coarse neighbours
__global__ __launch_bounds__(kThreads) void bilinear_up_bwd_kernel(const float* __restrict__ gy, BlkGeom gg, float* __restrict__ gx, BlkGeom gxg) {
    const long total = (long)gg.N * gg.CB * gg.H * gg.W * 4;
    const int IH = gxg.H, IW = gxg.W, OH = gg.H, OW = gg.W;
    const float sy = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f;
    const float sx = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
        long t = idx;
        const int q = (int)(t & 3); t >>= 2;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH); t /= OH;
        const int cb = (int)(t % gg.CB);
        const int n = (int)(t / gg.CB);
        const float fy = sy * oy, fx = sx * ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < IH - 1), x1 = x0 + (x0 < IW - 1);
        const float ty = fy - y0, tx = fx - x0;
        const f32x4 g = *(const f32x4*)(gy + blk_off(gg, n, cb, 0, oy, ox) + q * 4);
        float* p00 = gx + blk_off(gxg, n, cb, 0, y0, x0) + q * 4;
        float* p01 = gx + blk_off(gxg, n, cb, 0, y0, x1) + q * 4;
        float* p10 = gx + blk_off(gxg, n, cb, 0, y1, x0) + q * 4;
        float* p11 = gx + blk_off(gxg, n, cb, 0, y1, x1) + q * 4;
        const float w00 = (1.f - ty) * (1.f - tx), w01 = (1.f - ty) * tx, w10 = ty * (1.f - tx), w11 = ty * tx;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(p00 + e, g[e] * w00); atomicAdd(p01 + e, g[e] * w01); atomicAdd(p10 + e, g[e] * w10); atomicAdd(p11 + e, g[e] * w11);
        }
    }
}

// The same adjoint as a gather: one block per coarse cell (n, cb, cy, cx) sums the fine pixels whose bilinear support contains
// the cell -- the SPP maps are tiny (1x1 .. 7x7 coarse cells under 56x56 fine pixels), so the scatter form piles thousands of
// atomicAdds onto each address (958 us per call in the Config-B train step).  64 pixel slots x 4 channel quads per block; fixed
// summation order (bit-reproducible).  gx[cell] += sum.
__global__ __launch_bounds__(256) void bilinear_up_bwd_gather_kernel(const float* __restrict__ gy, BlkGeom gg, float* __restrict__ gx, BlkGeom gxg) {
    const int IH = gxg.H, IW = gxg.W, OH = gg.H, OW = gg.W;
    int b = blockIdx.x;
    const int cx = b % IW; b /= IW;
    const int cy = b % IH; b /= IH;
    const int cb = b % gg.CB;
    const int n = b / gg.CB;
    const float sy = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.f;
    const float sx = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.f;
    // conservative bounds of the fine rows / columns with y0 == cy or y1 == cy (fy in (cy-1, cy+1)); the exact test is in the loop
    int oy_lo = 0, oy_hi = OH - 1, ox_lo = 0, ox_hi = OW - 1;
    if (sy > 0.f) { oy_lo = max(0, (int)floorf((cy - 1) / sy) - 1); oy_hi = min(OH - 1, (int)ceilf((cy + 1) / sy) + 1); }
    if (sx > 0.f) { ox_lo = max(0, (int)floorf((cx - 1) / sx) - 1); ox_hi = min(OW - 1, (int)ceilf((cx + 1) / sx) + 1); }
    const int ncol = ox_hi - ox_lo + 1, total = (oy_hi - oy_lo + 1) * ncol;
    const int q = threadIdx.x & 3, slot = threadIdx.x >> 2;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = slot; i < total; i += 64) {
        const int r = i / ncol;
        const int oy = oy_lo + r, ox = ox_lo + (i - r * ncol);
        const float fy = sy * oy, fx = sx * ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < IH - 1), x1 = x0 + (x0 < IW - 1);
        const float ty = fy - y0, tx = fx - x0;
        const float wy = (y0 == cy ? 1.f - ty : 0.f) + (y1 == cy ? ty : 0.f);
        const float wx = (x0 == cx ? 1.f - tx : 0.f) + (x1 == cx ? tx : 0.f);
        const float w = wy * wx;
        if (w != 0.f) acc += w * *(const f32x4*)(gy + blk_off(gg, n, cb, 0, oy, ox) + q * 4);
    }
#pragma unroll
    for (int m = 4; m < 64; m <<= 1) {
        acc.x += __shfl_xor(acc.x, m); acc.y += __shfl_xor(acc.y, m); acc.z += __shfl_xor(acc.z, m); acc.w += __shfl_xor(acc.w, m);
    }
    __shared__ f32x4 red[4][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane < 4) red[wv][lane] = acc;
    __syncthreads();
    if (threadIdx.x < 4) {
        const f32x4 v = ((red[0][q] + red[1][q]) + red[2][q]) + red[3][q];
        float* dst = gx + blk_off(gxg, n, cb, 0, cy, cx) + q * 4;
        *(f32x4*)dst = *(const f32x4*)dst + v;
    }
}

// adjoint of AvgPool2d(k,k): gx[y,x] += gy[y/k, x/k] / k^2 inside the pooled region
__global__ __launch_bounds__(kThreads) void avgpool2d_bwd_kernel(const float* __restrict__ gy, BlkGeom gg, float* __restrict__ gx, BlkGeom gxg, int k) {
    const long total = (long)gxg.N * gxg.CB * gxg.H * gxg.W * 4;
    const float inv = 1.f / (float)(k * k);
    for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
        long t = idx;
        const int q = (int)(t & 3); t >>= 2;
        const int x = (int)(t % gxg.W); t /= gxg.W;
        const int y = (int)(t % gxg.H); t /= gxg.H;
        const int cb = (int)(t % gxg.CB);
        const int n = (int)(t / gxg.CB);
        const int oy = y / k, ox = x / k;
        if (oy >= gg.H || ox >= gg.W) continue;
        float* dst = gx + blk_off(gxg, n, cb, 0, y, x) + q * 4;
        *(f32x4*)dst = *(const f32x4*)dst + *(const f32x4*)(gy + blk_off(gg, n, cb, 0, oy, ox) + q * 4) * inv;
    }
}

// dz only (sites without BN, or the plain residual/ReLU fan-out): dz = dy * [y>0]; used to seed residual gradients
}  // namespace

extern "C" {

static void softargmin_bwd_footprint(int Hp, int Wp, int H, int W, int* CH, int* CW) {
    const float sy = H > 1 ? (float)(Hp - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(Wp - 1) / (float)(W - 1) : 0.f;
    *CH = (int)(sy * (kSATY - 1)) + 3; *CW = (int)(sx * (kSATX - 1)) + 3;    // coarse cells an 8 x 16 tile can touch (+ rounding slack)
}

int64_t drc_upsample_softargmin_bwd_scratch_floats(int N, int Dp, int Hp, int Wp, int H, int W) {
    if (N < 0 || Dp <= 0 || Hp <= 0 || Wp <= 0 || H <= 0 || W <= 0) return -2;
    int CH, CW;
    softargmin_bwd_footprint(Hp, Wp, H, W, &CH, &CW);
    return (int64_t)N * ((H + kSATY - 1) / kSATY) * ((W + kSATX - 1) / kSATX) * Dp * CH * CW;
}

int drc_upsample_softargmin_bwd(const float* cost, const float* grad_disp, float* grad_cost, int N, int Dp, int Hp, int Wp, int D, int H,
                                int W, int mindisp, float* scratch, int64_t scratch_floats, void* stream) {
    if (N < 0 || Dp <= 0 || Hp <= 0 || Wp <= 0 || D <= 0 || H <= 0 || W <= 0 || Dp > 96) return -2;
    const long total = (long)N * H * W;
    if (total == 0) return 0;
    if (!cost || !grad_disp || !grad_cost || !scratch) return -1;
    int CH, CW;
    softargmin_bwd_footprint(Hp, Wp, H, W, &CH, &CW);
    const size_t lds = ((size_t)2 * Dp * kSAThreads + (size_t)Dp * kSATY * CW) * 4;
    if (lds > 64 * 1024) return -2;
    const long blocks = (long)N * ((H + kSATY - 1) / kSATY) * ((W + kSATX - 1) / kSATX);
    if (blocks >= (1L << 31) || scratch_floats < blocks * Dp * CH * CW) return -2;
    hipLaunchKernelGGL(upsample_softargmin_bwd_kernel, dim3((unsigned)blocks), dim3(kSAThreads), lds, (hipStream_t)stream,
                       cost, grad_disp, scratch, N, Dp, Hp, Wp, D, H, W, mindisp, CH, CW);
    hipLaunchKernelGGL(softargmin_bwd_gather_kernel, dim3(grid_for((long)N * Dp * Hp * Wp, 8192)), dim3(kThreads), 0, (hipStream_t)stream,
                       scratch, grad_cost, N, Dp, Hp, Wp, H, W, CH, CW);
    return (int)hipGetLastError();
}

int drc_conv3d_cout1_bwd_data(const float* grad_out, const float* w, float* grad_x_blk, int N, int cb_in, int D, int H, int W,
                              int accumulate, void* stream) {
    if (N < 0 || cb_in <= 0 || D <= 0 || H <= 0 || W <= 0) return -2;
    const long total = (long)N * cb_in * D * H * W * 4;
    if (total == 0) return 0;
    if (!grad_out || !w || !grad_x_blk) return -1;
    hipLaunchKernelGGL(cout1_bwd_data_kernel, dim3(grid_for(total, 8192)), dim3(kThreads), 0, (hipStream_t)stream, grad_out, w, grad_x_blk, N,
                       cb_in, D, H, W, accumulate);
    return (int)hipGetLastError();
}

int drc_conv3d_cout1_bwd_weight(const float* x_blk, const float* grad_out, float* grad_w, int N, int cb_in, int D, int H, int W,
                                float* scratch, void* stream) {
    if (N < 0 || cb_in <= 0 || D <= 0 || H <= 0 || W <= 0) return -2;
    const long nvox = (long)N * D * H * W;
    if (!grad_w) return -1;
    if (nvox == 0) return (int)hipMemsetAsync(grad_w, 0, (size_t)27 * cb_in * 16 * sizeof(float), (hipStream_t)stream);
    if (!x_blk || !grad_out || !scratch) return -1;
    long chunks = (nvox + 64 * 16 - 1) / (64 * 16);
    if (chunks > 1024) chunks = 1024;                           // DRC_COUT1_WGRAD_SCRATCH_FLOATS holds 1024 blocks of partials
    hipLaunchKernelGGL(cout1_bwd_weight_kernel, dim3((unsigned)chunks, (unsigned)cb_in), dim3(kThreads), 0, (hipStream_t)stream, x_blk, grad_out,
                       scratch, N, cb_in, D, H, W);
    const int outs = 27 * cb_in * 16;
    hipLaunchKernelGGL(cout1_bwd_weight_finish_kernel, dim3((unsigned)((outs + kThreads / 64 - 1) / (kThreads / 64))), dim3(kThreads), 0,
                       (hipStream_t)stream, scratch, (int)chunks, cb_in, grad_w);
    return (int)hipGetLastError();
}

int drc_bn_bwd_reduce(const float* dy, const int* geom_dy, const float* y, const int* geom_y, const float* raw, const int* geom_raw,
                      const float* mean, const float* invstd, int relu, float* sums, float* scratch, void* stream) {
    if (!geom_ok(geom_dy) || !geom_ok(geom_raw) || (relu && !geom_ok(geom_y))) return -2;
    if (geom_dy[0] == 0) return 0;
    if (!dy || !raw || !mean || !invstd || !sums || !scratch || (relu && !y)) return -1;
    const BlkGeom g = to_geom(geom_dy);
    const long nvox = (long)g.N * g.D * g.H * g.W;
    long chunks = (nvox + 64 * 8 - 1) / (64 * 8);
    if (chunks > DRC_BN_MAX_CHUNKS) chunks = DRC_BN_MAX_CHUNKS;
    if (chunks > (long)g.N * g.D * g.H) chunks = (long)g.N * g.D * g.H;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)chunks, (unsigned)g.CB), dim3(kThreads), 0, (hipStream_t)stream, dy, g, y,
                       relu ? to_geom(geom_y) : g, raw, to_geom(geom_raw), mean, invstd, relu, sums, scratch);
    return (int)hipGetLastError();
}

int drc_bn_bwd_apply(const float* dy, const int* geom_dy, const float* y, const int* geom_y, const float* raw, const int* geom_raw,
                     const float* mean, const float* invstd, const float* gamma, const float* sums, float inv_count, int relu, float* draw,
                     const int* geom_draw, float* dres, const int* geom_dres, int dres_accumulate, void* stream) {
    if (!geom_ok(geom_dy) || !geom_ok(geom_raw) || !geom_ok(geom_draw) || (relu && !geom_ok(geom_y)) || (dres && !geom_ok(geom_dres))) return -2;
    if (geom_dy[0] == 0) return 0;
    if (!dy || !raw || !mean || !invstd || !gamma || !sums || !draw || (relu && !y)) return -1;
    const BlkGeom g = to_geom(geom_dy);
    const long rows = (long)g.N * g.D * g.H;
    long chunks = (rows * g.W * 4 + kThreads * 4 - 1) / (kThreads * 4);
    if (chunks > 2048) chunks = 2048;
    if (chunks > rows) chunks = rows;
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)chunks, (unsigned)g.CB), dim3(kThreads), 0, (hipStream_t)stream, dy, g, y, relu ? to_geom(geom_y) : g,
                       raw, to_geom(geom_raw), mean, invstd, gamma, sums, inv_count, relu, draw, to_geom(geom_draw), dres,
                       dres ? to_geom(geom_dres) : g, dres_accumulate);
    return (int)hipGetLastError();
}

int drc_bilinear_up_blocked_bwd(const float* grad_y, const int* geom_y, float* grad_x, const int* geom_x, void* stream) {
    if (!geom_ok(geom_y) || !geom_ok(geom_x) || geom_y[0] != geom_x[0] || geom_y[1] != geom_x[1]) return -2;
    if (geom_y[0] == 0) return 0;
    if (!grad_y || !grad_x) return -1;                  // grad_x must be zero-filled (or hold earlier contributions)
    const BlkGeom g = to_geom(geom_y), gxg = to_geom(geom_x);
    const long cells = (long)gxg.N * gxg.CB * gxg.H * gxg.W;
    if ((long)g.H * g.W >= 4L * gxg.H * gxg.W && cells < (1L << 31))          // upsampling: gather per coarse cell, no atomics
        hipLaunchKernelGGL(bilinear_up_bwd_gather_kernel, dim3((unsigned)cells), dim3(256), 0, (hipStream_t)stream, grad_y, g, grad_x, gxg);
    else
        hipLaunchKernelGGL(bilinear_up_bwd_kernel, dim3(grid_for((long)g.N * g.CB * g.H * g.W * 4)), dim3(kThreads), 0, (hipStream_t)stream, grad_y, g,
                           grad_x, gxg);
    return (int)hipGetLastError();
}

int drc_avgpool2d_blocked_bwd(const float* grad_y, const int* geom_y, float* grad_x, const int* geom_x, int k, void* stream) {
    if (!geom_ok(geom_y) || !geom_ok(geom_x) || geom_y[0] != geom_x[0] || geom_y[1] != geom_x[1] || k <= 0) return -2;
    if (geom_y[0] == 0) return 0;
    if (!grad_y || !grad_x) return -1;                  // accumulates into grad_x
    const BlkGeom g = to_geom(geom_x);
    hipLaunchKernelGGL(avgpool2d_bwd_kernel, dim3(grid_for((long)g.N * g.CB * g.H * g.W * 4)), dim3(kThreads), 0, (hipStream_t)stream, grad_y,
                       to_geom(geom_y), grad_x, g, k);
    return (int)hipGetLastError();
}

}  // extern "C"
