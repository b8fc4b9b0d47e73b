// micro-benchmark: issue cost (cycles per wavefront instruction on one SIMD) of the FP64 instructions k_spectrum leans on
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(64) void k(double* out, int iters, double seed) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + 0.001 * (threadIdx.x + i);
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a[i]));
      if (OP == 1) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
      if (OP == 2) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[i]));
      if (OP == 3) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(a[i]));
      if (OP == 4) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[i]));
      if (OP == 5) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a[i]));
      if (OP == 6) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a[i]));
      if (OP == 7) asm volatile("v_add_f64 %0, %0, %0" : "+v"(a[i]));
      if (OP == 8) { float f; asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(a[i])); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f)); }
      if (OP == 9) { float f; asm volatile("v_cvt_f32_f64 %0, %1\n v_rcp_f32 %0, %0" : "=v"(f) : "v"(a[i])); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f)); }
      if (OP == 10) { int n; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n) : "v"(a[i])); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(n)); }
      if (OP == 11) asm volatile("v_max_f64 %0, %0, %0" : "+v"(a[i]));
      if (OP == 12) asm volatile("v_exp_f32 %0, %0" : "+v"(*(float*)&a[i]));
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / (8.0 * iters);
}

int main() {
  double* d;
  CHECK(hipMalloc(&d, 65 * sizeof(double)));
  const char* names[] = {"v_fma_f64", "v_rcp_f64", "v_rsq_f64", "v_mul_f64", "v_rndne_f64", "v_ldexp_f64", "v_sqrt_f64", "v_add_f64",
                         "cvt_f32_f64+cvt_f64_f32 (pair)", "cvt+v_rcp_f32+cvt (triple)", "cvt_i32_f64+cvt_f64_i32 (pair)", "v_max_f64", "v_exp_f32"};
  double h[65];
#define RUN(OP) hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, d, 2000, 1.37); CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost)); printf("%-36s %.2f clock64 ticks per instruction group\n", names[OP], h[64]);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12)
  return 0;
}
