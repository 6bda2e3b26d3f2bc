// strided_store_probe.hip -- does a 16-B-per-lane access at a 32-byte stride (the transposed convolution's px = 0 / px = 1 output classes, written by
// DIFFERENT waves) cost HBM bandwidth against contiguous 16-B-per-lane accesses?  Copy kernel: each workgroup streams a private 56 KB region
// per step (read 28 KB "residual", write 28 KB "output"), 4 waves.  mode 0: lane l of wave w moves bytes [w][l*16 ..] contiguous (1 KB per
// instruction).  mode 1: wave w moves the even (w & 1 == 0) or odd 16-byte slots of a 2 KB span: the same bytes per instruction, half-dense.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/exp_libs/libprobe.so attic/strided_store_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe_kernel(const char* __restrict__ src, char* __restrict__ dst, long bytes_per_wg, int iters) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const char* s = src + (long)blockIdx.x * bytes_per_wg;
    char* d = dst + (long)blockIdx.x * bytes_per_wg;
    // per iteration the workgroup moves 4 waves x 8 instructions x 1 KB = 32 KB
    for (int it = 0; it < iters; ++it) {
        u32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            long off;
            if (MODE == 0) off = ((long)it * 32 + wave * 8 + i) * 1024 + lane * 16;
            else off = ((long)it * 32 + (wave >> 1) * 16 + i * 2) * 1024 + lane * 32 + (wave & 1) * 16;     // waves 2p, 2p+1 interleave 16-B slots over 2 KB spans
            v[i] = *(const u32x4*)(s + off);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            long off;
            if (MODE == 0) off = ((long)it * 32 + wave * 8 + i) * 1024 + lane * 16;
            else off = ((long)it * 32 + (wave >> 1) * 16 + i * 2) * 1024 + lane * 32 + (wave & 1) * 16;
            *(u32x4*)(d + off) = v[i];
        }
    }
}

extern "C" int probe(const void* src, void* dst, long bytes_per_wg, int iters, int mode, int blocks, void* stream) {
    if (mode == 0) hipLaunchKernelGGL(probe_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const char*)src, (char*)dst, bytes_per_wg, iters);
    else hipLaunchKernelGGL(probe_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const char*)src, (char*)dst, bytes_per_wg, iters);
    return (int)hipGetLastError();
}
