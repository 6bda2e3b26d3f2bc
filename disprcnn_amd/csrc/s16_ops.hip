// s16_ops.hip -- layout converters of the split-f16 path (RS16 tensors, see convs16.hip / include/disprcnn_hip.h), gfx950.
//
//   RS16: halfs [N][C/32][D+2pd][H+2][8 chunks][W+2][8]; chunk q = p*4 + s*2 + g (p: 0 hi / 1 lo), element e of chunk (s, g) =
//   channel 4g + 8(2s + (e>>2)) + (e&3) of the 32-channel block; hi = fp16(v), lo = fp16(v - hi).
// Pure data movement (HBM-bound): one thread per (voxel, chunk pair): 8 channels in, 16 B hi + 16 B lo out.
// Replaces nothing in the reference (its tensors are NCHW fp32); these sit where the reference hands features to
// PSMNet.forward's concat loop, stackhourglass.py:112-128.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disprcnn_hip.h"
#include "s16_ovf.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ inline void split8(const float (&v)[8], f16x8& hi, f16x8& lo, S16Ovf& og) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        og.see_raw(v[e]);
        float x = fminf(fmaxf(v[e], -65504.f), 65504.f);
        hi[e] = (_Float16)x;
        lo[e] = (_Float16)(x - (float)hi[e]);
    }
}

// dense [N,C,D,H,W] fp32 -> RS16 (interior only; the halo stays as allocated: zero)
__global__ void rs16_from_dense_kernel(const float* __restrict__ x, _Float16* __restrict__ y, int N, int C, int D, int H, int W, int pd, uint32_t* ovf) {
    S16Ovf og;
    const long total = (long)N * (C / 32) * D * H * 4 * W;
    const int Wp = W + 2, Hp = H + 2, Dp = D + 2 * pd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long t = i;
        const int xw = (int)(t % W); t /= W;
        const int sg = (int)(t % 4); t /= 4;
        const int yh = (int)(t % H); t /= H;
        const int z = (int)(t % D); t /= D;
        const int cb = (int)(t % (C / 32));
        const int n = (int)(t / (C / 32));
        const int s = sg >> 1, g = sg & 1;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cb * 32 + 4 * g + 8 * (2 * s + (e >> 2)) + (e & 3);
            v[e] = x[((((long)n * C + c) * D + z) * H + yh) * W + xw];
        }
        f16x8 hi, lo;
        split8(v, hi, lo, og);
        _Float16* row = y + (((((long)n * (C / 32) + cb) * Dp + z + pd) * Hp + yh + 1) * 8) * (long)Wp * 8;
        *(f16x8*)(row + ((long)sg * Wp + xw + 1) * 8) = hi;
        *(f16x8*)(row + ((long)(4 + sg) * Wp + xw + 1) * 8) = lo;
    }
    og.flush(ovf);
}

// blocked fp32 [units][CB16 total][D+2pdi][H+2phi][W+2pwi][16] (channel blocks cb16_off .. of it) -> RS16
__global__ void rs16_from_blocked_kernel(const float* __restrict__ x, _Float16* __restrict__ y, int N, int C, int D, int H, int W, int pdi, int phi,
                                         int pwi, int cb16_total, int cb16_off, int pd, uint32_t* ovf) {
    S16Ovf og;
    const long total = (long)N * (C / 32) * D * H * 4 * W;
    const int Wp = W + 2, Hp = H + 2, Dp = D + 2 * pd;
    const long xw_ = W + 2 * pwi, xh_ = H + 2 * phi, xd_ = D + 2 * pdi;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long t = i;
        const int sg = (int)(t % 4); t /= 4;                 // consecutive threads: the 4 chunk pairs of one voxel (64 contiguous bytes x 2 blocks)
        const int xw = (int)(t % W); t /= W;
        const int yh = (int)(t % H); t /= H;
        const int z = (int)(t % D); t /= D;
        const int cb = (int)(t % (C / 32));
        const int n = (int)(t / (C / 32));
        const int s = sg >> 1, g = sg & 1;
        // channels 4g + 16s + 8j + i: 16-channel block 2cb + s, floats 4g + 8j .. +3
        const float* src = x + (((((long)n * cb16_total + cb16_off + 2 * cb + s) * xd_ + z + pdi) * xh_ + yh + phi) * xw_ + xw + pwi) * 16 + 4 * g;
        const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 8);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        f16x8 hi, lo;
        split8(v, hi, lo, og);
        _Float16* row = y + (((((long)n * (C / 32) + cb) * Dp + z + pd) * Hp + yh + 1) * 8) * (long)Wp * 8;
        *(f16x8*)(row + ((long)sg * Wp + xw + 1) * 8) = hi;
        *(f16x8*)(row + ((long)(4 + sg) * Wp + xw + 1) * 8) = lo;
    }
    og.flush(ovf);
}

// RS16 -> dense fp32 [N,C,D,H,W]
__global__ void rs16_to_dense_kernel(const _Float16* __restrict__ y, float* __restrict__ x, int N, int C, int D, int H, int W, int pd) {
    const long total = (long)N * (C / 32) * D * H * 4 * W;
    const int Wp = W + 2, Hp = H + 2, Dp = D + 2 * pd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long t = i;
        const int xw = (int)(t % W); t /= W;
        const int sg = (int)(t % 4); t /= 4;
        const int yh = (int)(t % H); t /= H;
        const int z = (int)(t % D); t /= D;
        const int cb = (int)(t % (C / 32));
        const int n = (int)(t / (C / 32));
        const int s = sg >> 1, g = sg & 1;
        const _Float16* row = y + (((((long)n * (C / 32) + cb) * Dp + z + pd) * Hp + yh + 1) * 8) * (long)Wp * 8;
        const f16x8 hi = *(const f16x8*)(row + ((long)sg * Wp + xw + 1) * 8);
        const f16x8 lo = *(const f16x8*)(row + ((long)(4 + sg) * Wp + xw + 1) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cb * 32 + 4 * g + 8 * (2 * s + (e >> 2)) + (e & 3);
            x[((((long)n * C + c) * D + z) * H + yh) * W + xw] = (float)hi[e] + (float)lo[e];
        }
    }
}

// RS16 -> blocked fp32 [units][CB16 total][D+2pdo][H+2pho][W+2pwo][16] (channel blocks cb16_off .. of it; interior only): where a split-f16
// layer hands its result to an fp32 kernel (the stride-2 / 1x1 / dilated layers of the 2D CNN, the concat of submodule.py:134-135)
__global__ void rs16_to_blocked_kernel(const _Float16* __restrict__ y, float* __restrict__ x, int N, int C, int D, int H, int W, int pdo, int pho,
                                       int pwo, int cb16_total, int cb16_off, int pd) {
    const long total = (long)N * (C / 32) * D * H * 4 * W;
    const int Wp = W + 2, Hp = H + 2, Dp = D + 2 * pd;
    const long xw_ = W + 2 * pwo, xh_ = H + 2 * pho, xd_ = D + 2 * pdo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long t = i;
        const int sg = (int)(t % 4); t /= 4;
        const int xw = (int)(t % W); t /= W;
        const int yh = (int)(t % H); t /= H;
        const int z = (int)(t % D); t /= D;
        const int cb = (int)(t % (C / 32));
        const int n = (int)(t / (C / 32));
        const int s = sg >> 1, g = sg & 1;
        const _Float16* row = y + (((((long)n * (C / 32) + cb) * Dp + z + pd) * Hp + yh + 1) * 8) * (long)Wp * 8;
        const f16x8 hi = *(const f16x8*)(row + ((long)sg * Wp + xw + 1) * 8);
        const f16x8 lo = *(const f16x8*)(row + ((long)(4 + sg) * Wp + xw + 1) * 8);
        float* dst = x + (((((long)n * cb16_total + cb16_off + 2 * cb + s) * xd_ + z + pdo) * xh_ + yh + pho) * xw_ + xw + pwo) * 16 + 4 * g;
        *(f32x4*)dst = (f32x4){(float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1], (float)hi[2] + (float)lo[2], (float)hi[3] + (float)lo[3]};
        *(f32x4*)(dst + 8) = (f32x4){(float)hi[4] + (float)lo[4], (float)hi[5] + (float)lo[5], (float)hi[6] + (float)lo[6], (float)hi[7] + (float)lo[7]};
    }
}

inline unsigned grid_for(long total) {
    long b = (total + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 256 * 32 ? 256 * 32 : b));
}

}  // namespace

extern "C" int drc_rs16_from_dense(const float* x, void* y16, int N, int C, int D, int H, int W, int pd, uint32_t* ovf, void* stream) {
    if (!x || !y16) return -1;
    if (N < 0 || C <= 0 || (C & 31) || D <= 0 || H <= 0 || W <= 0 || pd < 0 || pd > 1) return -2;
    if (N == 0) return 0;
    hipLaunchKernelGGL(rs16_from_dense_kernel, dim3(grid_for((long)N * (C / 32) * D * H * 4 * W)), dim3(256), 0, (hipStream_t)stream, x, (_Float16*)y16, N, C, D, H, W, pd, ovf);
    return (int)hipGetLastError();
}

extern "C" int drc_rs16_from_blocked(const float* xb, void* y16, int N, int C, int D, int H, int W, int pd_in, int ph_in, int pw_in, int cb16_total,
                                     int cb16_off, int pd, uint32_t* ovf, void* stream) {
    if (!xb || !y16) return -1;
    if (N < 0 || C <= 0 || (C & 31) || D <= 0 || H <= 0 || W <= 0 || pd < 0 || pd > 1 || pd_in < 0 || ph_in < 0 || pw_in < 0 || cb16_off < 0 ||
        cb16_off + C / 16 > cb16_total)
        return -2;
    if (N == 0) return 0;
    hipLaunchKernelGGL(rs16_from_blocked_kernel, dim3(grid_for((long)N * (C / 32) * D * H * 4 * W)), dim3(256), 0, (hipStream_t)stream, xb, (_Float16*)y16, N, C, D,
                       H, W, pd_in, ph_in, pw_in, cb16_total, cb16_off, pd, ovf);
    return (int)hipGetLastError();
}

extern "C" int drc_rs16_to_dense(const void* y16, float* x, int N, int C, int D, int H, int W, int pd, void* stream) {
    if (!x || !y16) return -1;
    if (N < 0 || C <= 0 || (C & 31) || D <= 0 || H <= 0 || W <= 0 || pd < 0 || pd > 1) return -2;
    if (N == 0) return 0;
    hipLaunchKernelGGL(rs16_to_dense_kernel, dim3(grid_for((long)N * (C / 32) * D * H * 4 * W)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)y16, x, N, C, D, H, W, pd);
    return (int)hipGetLastError();
}

extern "C" int drc_rs16_to_blocked(const void* y16, float* xb, int N, int C, int D, int H, int W, int pd_out, int ph_out, int pw_out, int cb16_total,
                                   int cb16_off, int pd, void* stream) {
    if (!xb || !y16) return -1;
    if (N < 0 || C <= 0 || (C & 31) || D <= 0 || H <= 0 || W <= 0 || pd < 0 || pd > 1 || pd_out < 0 || ph_out < 0 || pw_out < 0 || cb16_off < 0 ||
        cb16_off + C / 16 > cb16_total)
        return -2;
    if (N == 0) return 0;
    hipLaunchKernelGGL(rs16_to_blocked_kernel, dim3(grid_for((long)N * (C / 32) * D * H * 4 * W)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)y16, xb, N, C,
                       D, H, W, pd_out, ph_out, pw_out, cb16_total, cb16_off, pd);
    return (int)hipGetLastError();
}

// ---- the second half of a fused cout-1 head (convs16.hip, HEAD form): nine shifted in-plane partial sums per output voxel
namespace {
__global__ __launch_bounds__(256) void head_gather_kernel(const float* __restrict__ S, const float* __restrict__ res, float* __restrict__ cost,
                                                          long total, int H, int W, float scale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % W);
    const long row = i / W;                 // (n * D + z) * H + y
    const int y = (int)(row % H);
    float a = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int yy = y + kh - 1;
        if (yy < 0 || yy >= H) continue;
        const float* sr = S + ((row + (kh - 1)) * W) * 12;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int xx = x + kw - 1;
            if (xx < 0 || xx >= W) continue;
            const int j = kh * 3 + kw;
            a += sr[(long)xx * 12 + (j < 5 ? j : j + 3)];
        }
    }
    a *= scale;
    cost[i] = res ? a + res[i] : a;
}
}  // namespace

namespace {
// The same sums through LDS: a workgroup takes a band of R rows of one (unit, plane) and loads the S slots of its (R + 2) x (W + 2) source
// voxels as whole 48-byte slots (three 16-byte loads per lane, consecutive lanes = consecutive slots: streaming reads instead of nine
// scattered 4-byte reads per output voxel, which left head_gather_kernel waiting 88 % of its cycles), nine floats per voxel into LDS (zeros
// outside the map: the convolution's padding); an output voxel then adds nine LDS values (voxel stride 9 floats: no bank conflicts).
__global__ __launch_bounds__(256) void head_gather_lds_kernel(const float* __restrict__ S, const float* __restrict__ res, float* __restrict__ cost,
                                                              int H, int W, int R, int bands, float scale) {
    extern __shared__ __attribute__((aligned(16))) float tile[];          // [(R + 2)][(W + 2)][9]
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const long plane = blockIdx.x / bands;                                // n * D + z
    const int y0 = (int)(blockIdx.x % bands) * R;
    const int rows = H - y0 < R ? H - y0 : R;
    const int Wt = W + 2, nt = (rows + 2) * Wt;
    const float* sp = S + plane * (long)H * W * 12;
    for (int i = threadIdx.x; i < nt; i += 256) {
        const int ry = i / Wt, rx = i - ry * Wt;
        const int y = y0 - 1 + ry, x = rx - 1;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a, c = a;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const f32x4* q = (const f32x4*)(sp + ((long)y * W + x) * 12);
            a = q[0]; b = q[1]; c = q[2];
        }
        float* t = tile + i * 9;
        t[0] = a[0]; t[1] = a[1]; t[2] = a[2]; t[3] = a[3]; t[4] = b[0];
        t[5] = c[0]; t[6] = c[1]; t[7] = c[2]; t[8] = c[3];
    }
    __syncthreads();
    const int no = rows * W;
    const long o0 = (plane * H + y0) * (long)W;
    for (int i = threadIdx.x; i < no; i += 256) {
        const int ry = i / W, rx = i - ry * W;
        float v = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) v += tile[((ry + kh) * Wt + rx + kw) * 9 + kh * 3 + kw];
        v *= scale;
        cost[o0 + i] = res ? v + res[o0 + i] : v;
    }
}
}  // namespace

extern "C" int drc_head_gather_fwd(const float* S, const float* res, float* cost, int N, int D, int H, int W, float scale, void* stream) {
    if (!S || !cost) return -1;
    if (N < 0 || D <= 0 || H <= 0 || W <= 0) return -2;
    if (N == 0) return 0;
    const long total = (long)N * D * H * W;
    // band height by what a 40 KiB tile holds (four workgroups per CU): the whole 28 x 28 plane of Config A, 16 of Config B's 56 rows
    const long per_row = (long)(W + 2) * 36;
    long R = 40960 / per_row - 2;
    if (R >= 1) {
        if (R > H) R = H;
        const long bands = (H + R - 1) / R;
        const long blocks = (long)N * D * bands;
        if (blocks > 0x7fffffffL) return -3;
        hipLaunchKernelGGL(head_gather_lds_kernel, dim3((unsigned)blocks), dim3(256), (size_t)((R + 2) * per_row), (hipStream_t)stream, S, res, cost, H, W,
                           (int)R, (int)bands, scale);
        return (int)hipGetLastError();
    }
    const long blocks = (total + 255) / 256;                              // very wide maps: one thread per output voxel, reads through the caches
    if (blocks > 0x7fffffffL) return -3;
    hipLaunchKernelGGL(head_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, S, res, cost, total, H, W, scale);
    return (int)hipGetLastError();
}
