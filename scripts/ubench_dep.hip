// micro-benchmark: how often ONE wavefront can issue v_fma_f64 when its instructions form NCH independent dependency chains
// (NCH = 1: every instruction waits for the previous result), alone on its SIMD and next to a second wavefront doing the same.
// Answers: is a single wavefront of the spectrum kernels issue-bound or latency-bound, and what does a second wavefront add?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NCH, int OP>
__global__ __launch_bounds__(512) void k(double* out, int iters, double seed) {
  double a[NCH];
  for (int i = 0; i < NCH; ++i) a[i] = seed + 0.001 * (threadIdx.x + i);
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16 / NCH; ++r)
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        if (OP == 0) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a[i]));
        if (OP == 1) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(a[i]));
        if (OP == 2) asm volatile("v_add_f64 %0, %0, %0" : "+v"(a[i]));
        if (OP == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
        if (OP == 4) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(*(float*)&a[i]));
      }
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < NCH; ++i) s += a[i];
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[1024 + (threadIdx.x >> 6)] = (double)(t1 - t0) / (16.0 * iters);
}

int main() {
  double* d;
  CHECK(hipMalloc(&d, 2048 * sizeof(double)));
  double h[2048];
  const char* ops[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_fma_f32"};
#define RUN(NCH, OP, THREADS) hipLaunchKernelGGL((k<NCH, OP>), dim3(1), dim3(THREADS), 0, 0, d, 2000, 1.37); CHECK(hipDeviceSynchronize()); \
  CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost)); \
  printf("%-10s chains %2d, %d wavefront(s) per SIMD: %.2f cycles per instruction of one wavefront -> SIMD issues one every %.2f cycles\n", ops[OP], NCH, THREADS / 256, h[1024], h[1024] / (THREADS / 256));
#define ALL(OP) RUN(1, OP, 256) RUN(2, OP, 256) RUN(4, OP, 256) RUN(8, OP, 256) RUN(1, OP, 512) RUN(2, OP, 512) RUN(4, OP, 512) RUN(8, OP, 512)
  ALL(0) ALL(1) ALL(2) ALL(3) ALL(4)
  return 0;
}
