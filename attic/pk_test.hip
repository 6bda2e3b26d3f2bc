// development check: v_pk_add_f32 with neg modifiers == a - b ?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k3(const f32x4* a, const f32x4* b, f32x4* c, f32x4* d) { int i = threadIdx.x; f32x4 x = a[i], y = b[i], z;
  f32x2 lo, hi; f32x2 xl = {x.x, x.y}, yl = {y.x, y.y}, xh = {x.z, x.w}, yh = {y.z, y.w};
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(xl), "v"(yl));
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(xh), "v"(yh));
  z.x = lo.x; z.y = lo.y; z.z = hi.x; z.w = hi.y; c[i] = z; d[i] = x - y; }
int main() {
  f32x4 *a, *b, *c, *d; hipMallocManaged(&a, 64 * 16); hipMallocManaged(&b, 64 * 16); hipMallocManaged(&c, 64 * 16); hipMallocManaged(&d, 64 * 16);
  for (int i = 0; i < 64; ++i) { a[i] = f32x4{1.f * i, 2.f + i, -3.5f * i, 0.25f}; b[i] = f32x4{0.5f, -1.f * i, 7.f, 100.f + i}; }
  hipLaunchKernelGGL(k3, dim3(1), dim3(64), 0, 0, a, b, c, d); hipDeviceSynchronize();
  int bad = 0; for (int i = 0; i < 64; ++i) for (int k = 0; k < 4; ++k) if (c[i][k] != d[i][k]) { if (bad < 4) printf("i=%d k=%d pk=%g ref=%g\n", i, k, c[i][k], d[i][k]); ++bad; }
  printf("bad=%d\n", bad); return 0; }
