// bf16x3_probe.hip -- what would an fp32-accurate convolution on the bf16 matrix cores cost per K = 32 block? (round 4 probe for DESIGN section 8; not
// part of the product)
//   fp32 path today:   8 x v_mfma_f32_16x16x4_f32 per 16 couts x 16 voxels x 32 channels (the MFMA holds the vector ALUs)
//   split path:        the B operand (two float4 = 8 channels per lane) is split into three bf16 pieces b1 + b2 + b3 (b1 = bf16(b), b2 = bf16(b - b1),
//                      b3 = bf16(b - b1 - b2): 24 significant bits), the weights are split offline; products (1,1) (1,2) (2,1) (1,3) (2,2) (3,1) =
//                      6 x v_mfma_f32_16x16x32_bf16, dropped terms <= 2^-24 relative.  3-product form: (1,1) (1,2) (2,1), error ~2^-16.
// Reported: shader cycles per K = 32 block of one wave (one wave per SIMD), and the numerical error of both split forms against fp64 on random data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int ITER = 2048;

__device__ __forceinline__ void split3(const f32x4 lo, const f32x4 hi, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 a = (__bf16)v[i];
        const float r1 = v[i] - (float)a;
        const __bf16 b = (__bf16)r1;
        const float r2 = r1 - (float)b;
        p1[i] = a; p2[i] = b; p3[i] = (__bf16)r2;
    }
}

template <int MODE>   // 0: fp32 MFMA x8; 1: split + 6 bf16 MFMAs; 2: split (two pieces) + 3 bf16 MFMAs
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, const float* __restrict__ w, float* out, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    // per lane: 8 channels of its voxel (B) and of its cout (A), re-read every iteration from L1-resident memory with a varying offset
    const f32x4* xb = (const f32x4*)x + lane * 2;
    const f32x4* wb = (const f32x4*)w + lane * 2;
    bf16x8 w1, w2, w3;
    split3(wb[0], wb[1], w1, w2, w3);                 // weights: split once (offline in a real kernel)
    f32x4 b0 = xb[0], b1 = xb[1];
    const f32x4 a0 = wb[0], a1 = wb[1];
    const f32x4 step = {1e-3f, 2e-3f, 3e-3f, 4e-3f};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 2
    for (int it = 0; it < ITER; ++it) {
        b0 += step; b1 -= step;                        // operands stay in registers (no memory latency in the loop) but change every iteration
        if constexpr (MODE == 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b0[s], acc[s], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b1[s], acc[s], 0, 0, 0);
        } else {
            bf16x8 p1, p2, p3;
            split3(b0, b1, p1, p2, p3);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, p1, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, p2, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, p1, acc[2], 0, 0, 0);
            if constexpr (MODE == 1) {
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, p3, acc[3], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, p2, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3, p1, acc[1], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    *(f32x4*)(out + (blockIdx.x * blockDim.x + threadIdx.x) * 4) = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// accuracy: one 16x16x32 product per mode against fp64
template <int MODE>
__global__ void acc_k(const float* x, const float* w, float* out) {
    const int lane = threadIdx.x;
    const f32x4* xb = (const f32x4*)x + lane * 2;
    const f32x4* wb = (const f32x4*)w + lane * 2;
    f32x4 c = {0, 0, 0, 0};
    if constexpr (MODE == 0) {
        for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[0][s], xb[0][s], c, 0, 0, 0);
        for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[1][s], xb[1][s], c, 0, 0, 0);
    } else {
        bf16x8 w1, w2, w3, p1, p2, p3;
        split3(wb[0], wb[1], w1, w2, w3); split3(xb[0], xb[1], p1, p2, p3);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, p1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, p2, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, p1, c, 0, 0, 0);
        if constexpr (MODE == 1) {
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, p3, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, p2, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3, p1, c, 0, 0, 0);
        }
    }
    *(f32x4*)(out + lane * 4) = c;
}

int main() {
    const int n = 64 * 8 * 256;
    std::vector<float> hx(n), hw(n);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hx) v = rnd() * 3.0f;
    for (auto& v : hw) v = rnd() * 0.1f;
    float *x, *w, *out; unsigned long long* cyc;
    hipMalloc(&x, n * 4); hipMalloc(&w, n * 4); hipMalloc(&out, 256 * 256 * 16); hipMalloc(&cyc, 8);
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), n * 4, hipMemcpyHostToDevice);
    const char* names[3] = {"fp32: 8 x mfma_f32_16x16x4_f32", "bf16 split, 6 products (24-bit): 6 x mfma_f32_16x16x32_bf16 + split", "bf16 split, 3 products (16-bit): 3 x mfma + split"};
    for (int m = 0; m < 3; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, x, w, out, cyc);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, x, w, out, cyc);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, x, w, out, cyc);
        }
        hipDeviceSynchronize();
        unsigned long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        // accuracy of one 16 x 16 x 32 block: lane l holds cout l&15 / voxel l&15, k = 8*(l>>4) .. +7  (fp32 mode: k-step s uses element s of lane group l>>4)
        if (m == 0) hipLaunchKernelGGL(acc_k<0>, dim3(1), dim3(64), 0, 0, x, w, out);
        if (m == 1) hipLaunchKernelGGL(acc_k<1>, dim3(1), dim3(64), 0, 0, x, w, out);
        if (m == 2) hipLaunchKernelGGL(acc_k<2>, dim3(1), dim3(64), 0, 0, x, w, out);
        hipDeviceSynchronize();
        float ho[256]; hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
        double emax = 0, ymax = 0;
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {           // D[cout 4*(l>>4)+i][voxel l&15]
            const int co = 4 * (l >> 4) + i, vx = l & 15;
            double ref = 0;
            for (int g = 0; g < 4; ++g) for (int e = 0; e < 8; ++e) ref += (double)hw[(co + 16 * g) * 8 + e] * (double)hx[(vx + 16 * g) * 8 + e];
            emax = fmax(emax, fabs(ho[l * 4 + i] - ref)); ymax = fmax(ymax, fabs(ref));
        }
        printf("%-78s %7.1f cycles per K=32 block   max |err| / max |y| = %.2e\n", names[m], (double)c / ITER, emax / ymax);
    }
    return 0;
}
