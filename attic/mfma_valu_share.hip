// mfma_valu_share.hip -- microbenchmark behind DESIGN section 3.0c (round 4): does a stream of fp32 MFMAs leave room for ordinary vector
// instructions on the same SIMD?  (development tool, not part of the product)
//
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_valu_share.hip -o /tmp/mfma_valu_share && /tmp/mfma_valu_share
//
// Every CU runs one block; a wave executes ITER times { one MFMA (four independent accumulators in rotation), NV independent v_add_f32 }.
//   mode 0: 4 waves per CU (one per SIMD), the VALU instructions sit in the MFMA wave's own stream
//   mode 1: 8 waves per CU (two per SIMD): waves 0-3 issue only MFMAs, their SIMD partners 4-7 only the VALU instructions (NV per MFMA of the partner)
// Reported: shader cycles (s_memtime) per MFMA of wave 0, for the fp32 MFMA 16x16x4 (8 passes, 32 cycles alone) and, as the control, the
// bf16 MFMA 16x16x16 (which runs on the matrix cores proper).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 4096;

template <int KIND, int NV, int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)(lane + i);
    const float a = 1.0f + lane * 1e-3f, b = 2.0f - lane * 1e-3f;
    const bf16x4 ah = {(short)(0x3f80 + lane), 0x3f80, 0x3f80, 0x3f80}, bh = {0x3f80, (short)(0x3f80 + lane), 0x3f80, 0x3f80};
    const bool do_mfma = MODE == 0 || wave < 4, do_valu = MODE == 0 || wave >= 4;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int it = 0; it < ITER; ++it) {
        if (do_mfma) {
            if constexpr (KIND == 0) acc[it & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[it & 3], 0, 0, 0);
            else acc[it & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, acc[it & 3], 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int e = 0; e < NV; ++e) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[e & 7]) : "v"(a));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int NV, int MODE>
double run(float* out, unsigned long long* cyc) {
    const int threads = MODE == 0 ? 256 : 512;
    hipLaunchKernelGGL((k<KIND, NV, MODE>), dim3(256), dim3(threads), 0, 0, out, cyc);
    hipLaunchKernelGGL((k<KIND, NV, MODE>), dim3(256), dim3(threads), 0, 0, out, cyc);
    hipDeviceSynchronize();
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    return (double)c / ITER;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&cyc, 8);
    printf("shader cycles per MFMA of wave 0 (ITER %d, 256 blocks)\n", ITER);
    printf("%-58s %8s %8s %8s %8s %8s %8s\n", "NV = vector instructions per MFMA", "0", "1", "2", "4", "8", "16");
#define ROW(KIND, MODE, label) printf("%-58s %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f\n", label, run<KIND, 0, MODE>(out, cyc), run<KIND, 1, MODE>(out, cyc), \
                                      run<KIND, 2, MODE>(out, cyc), run<KIND, 4, MODE>(out, cyc), run<KIND, 8, MODE>(out, cyc), run<KIND, 16, MODE>(out, cyc))
    ROW(0, 0, "fp32 MFMA 16x16x4,  VALU in the same wave (1 wave/SIMD)");
    ROW(0, 1, "fp32 MFMA 16x16x4,  VALU in the SIMD partner (2 waves/SIMD)");
    ROW(1, 0, "bf16 MFMA 16x16x16, VALU in the same wave (1 wave/SIMD)");
    ROW(1, 1, "bf16 MFMA 16x16x16, VALU in the SIMD partner (2 waves/SIMD)");
    return 0;
}
