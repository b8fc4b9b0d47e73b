// micro-benchmark: do the FP64 matrix instruction (v_mfma_f64_16x16x4_f64) and the FP64 vector instruction (v_fma_f64) of two
// DIFFERENT wavefronts on the same SIMD execute side by side, or do they share one datapath?
// One workgroup of 512 threads on one CU = two wavefronts per SIMD (wavefront w runs on SIMD w % 4; pair (w, w + 4) shares a SIMD).
// Modes: every wavefront MFMA; every wavefront FMA; wavefronts 0-3 MFMA and 4-7 FMA (one of each per SIMD); and each kind alone
// with one wavefront per SIMD (256 threads).  Each wavefront reports cycles per instruction of its own stream (s_memtime).
// If the pipes were independent, the mixed mode would run both streams at their one-wavefront-per-SIMD rates.
// Answers VERDICT r2 item 5 (overlapping the W-table GEMM with the one-sweep kernel): see DESIGN.md section 4.2.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

// kind 0: 8 independent MFMA accumulators; kind 1: 16 independent FMA chains; kind 2: 16 independent v_fma_f32 chains
template <int MODE>   // 0 all MFMA, 1 all FMA f64, 2 mixed MFMA | FMA f64, 3 mixed MFMA | FMA f32, 4 all FMA f32
__global__ __launch_bounds__(512) void k(double* out, int iters, double seed) {
  const int w = threadIdx.x >> 6;
  const int kind = MODE == 0 ? 0 : MODE == 1 ? 1 : MODE == 4 ? 2 : (w < 4 ? 0 : (MODE == 2 ? 1 : 2));
  d4 acc[8];
  double a[16];
  for (int i = 0; i < 8; ++i) acc[i] = (d4){seed, seed, seed, seed};
  for (int i = 0; i < 16; ++i) a[i] = seed + 0.001 * (threadIdx.x + i);
  const double x = seed * 1e-3, y = 1.0 + seed * 1e-6;
  __syncthreads();
  long long t0 = clock64();
  if (kind == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
    }
  } else if (kind == 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[i]) : "v"(y));
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(*(float*)&a[i]) : "v"(*(const float*)&y));
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += a[i];
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[1024 + w] = (double)(t1 - t0) / ((kind == 0 ? 8.0 : 16.0) * iters);
}

int main() {
  double* d;
  CHECK(hipMalloc(&d, 2048 * sizeof(double)));
  double h[2048];
  const int iters = 4000;
#define RUN(MODE, THREADS, label)                                                                                     \
  hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(THREADS), 0, 0, d, iters, 1.37); CHECK(hipDeviceSynchronize());            \
  hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(THREADS), 0, 0, d, iters, 1.37); CHECK(hipDeviceSynchronize());            \
  CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));                                                           \
  printf("%-58s wavefront 0: %7.2f cycles per instruction", label, h[1024]);                                           \
  if (THREADS == 512) printf(" | wavefront 4 (same SIMD): %7.2f", h[1028]);                                            \
  printf("\n");
  RUN(0, 256, "MFMA f64 16x16x4 alone (1 wavefront per SIMD)")
  RUN(1, 256, "v_fma_f64 alone (1 wavefront per SIMD)")
  RUN(4, 256, "v_fma_f32 alone (1 wavefront per SIMD)")
  RUN(0, 512, "MFMA f64 + MFMA f64 (2 wavefronts per SIMD)")
  RUN(1, 512, "v_fma_f64 + v_fma_f64 (2 wavefronts per SIMD)")
  RUN(2, 512, "MFMA f64 (wavefront 0) + v_fma_f64 (wavefront 4)")
  RUN(3, 512, "MFMA f64 (wavefront 0) + v_fma_f32 (wavefront 4)")
  printf("(an MFMA f64 16x16x4 is 1024 FMAs = 16 wavefront-wide v_fma_f64: at the vector rate of one v_fma_f64 per 4 cycles that is 64 cycles)\n");
  return 0;
}
