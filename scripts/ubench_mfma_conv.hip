// micro-benchmark for moving the IRF convolutions of the spectrum kernels onto the FP64 matrix cores:
//  (1) does v_mfma_f64_16x16x4_f64 accumulate D = C + sum_k A_k B_k as the FMA chain fma(a3,b3,fma(a2,b2,fma(a1,b1,fma(a0,b0,c))))
//      -- i.e. would a Toeplitz-matrix form of the convolution reproduce the explicit FMA chains of the kernels bit for bit?
//  (2) what does a dependent chain of such instructions cost per instruction (operands read from LDS), alone on a SIMD and next to
//      a wavefront that issues FP64 VALU instructions (the co-resident workgroup's sweep)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

// D[16][16] = sum_k A[16][K] B[K][16] by K/4 chained MFMAs (one wavefront), and the same sums as explicit FMA chains
__global__ void k_bits(const double* A, const double* B, int K, double* Dm, double* Df) {
  const int lane = threadIdx.x, fr = lane & 15, fk = lane >> 4;
  d4 acc = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[fr * K + k0 + fk], B[(k0 + fk) * 16 + fr], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) Dm[(fk + 4 * r) * 16 + fr] = acc[r];
  for (int r = 0; r < 4; ++r) {
    const int i = fk + 4 * r, j = fr;
    double s = 0.0;
    for (int k = 0; k < K; ++k) s = __builtin_fma(A[i * K + k], B[k * 16 + j], s);
    Df[i * 16 + j] = s;
  }
}

// waves [0, nm): dependent MFMA chain with both operands read from LDS; waves [nm, nwaves): dependent v_fma_f64 chains (2 chains)
__global__ __launch_bounds__(512) void k_time(double* out, int iters, int nm) {
  __shared__ double sm[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  long long t0 = clock64();
  double res = 0.0;
  if (w < nm) {
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    const double* pa = sm + lane, * pb = sm + 2048 + (lane & 15) * 4 + (lane >> 4) * 67;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int c = 0; c < 16; ++c) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * c], pb[c], acc, 0, 0, 0);
    }
    res = acc[0] + acc[1] + acc[2] + acc[3];
  } else {
    double a = 1.0 + 1e-9 * lane, b = 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int c = 0; c < 8; ++c) { asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a)); asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(b)); }
    }
    res = a + b;
  }
  long long t1 = clock64();
  out[threadIdx.x] = res;
  if (lane == 0) out[1024 + w] = (double)(t1 - t0) / (16.0 * iters);
}

int main() {
  const int K = 128;
  double hA[16 * K], hB[K * 16], hDm[256], hDf[256];
  srand(7);
  for (int i = 0; i < 16 * K; ++i) hA[i] = (rand() / (double)RAND_MAX - 0.5) * exp(8.0 * (rand() / (double)RAND_MAX - 0.5));
  for (int i = 0; i < K * 16; ++i) hB[i] = (rand() / (double)RAND_MAX - 0.3) * exp(8.0 * (rand() / (double)RAND_MAX - 0.5));
  for (int i = 0; i < 16; ++i) for (int k = 0; k < K; ++k) if (k < i || k > i + 100) hA[i * K + k] = 0.0;   // (Toeplitz-like zero padding)
  double *dA, *dB, *dDm, *dDf, *dout;
  CHECK(hipMalloc(&dA, sizeof hA)); CHECK(hipMalloc(&dB, sizeof hB)); CHECK(hipMalloc(&dDm, sizeof hDm)); CHECK(hipMalloc(&dDf, sizeof hDf));
  CHECK(hipMalloc(&dout, 2048 * sizeof(double)));
  CHECK(hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_bits, dim3(1), dim3(64), 0, 0, dA, dB, K, dDm, dDf);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(hDm, dDm, sizeof hDm, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hDf, dDf, sizeof hDf, hipMemcpyDeviceToHost));
  int ndiff = 0; double maxrel = 0.0;
  for (int i = 0; i < 256; ++i) { if (memcmp(&hDm[i], &hDf[i], 8)) { ++ndiff; maxrel = fmax(maxrel, fabs(hDm[i] - hDf[i]) / fabs(hDf[i])); } }
  printf("MFMA chain vs explicit FMA chain (K = %d): %d of 256 elements differ in their bits, max relative difference %.3e\n", K, ndiff, maxrel);
  double h[2048];
  struct { int threads, nm; const char* what; } cases[] = {
      {256, 4, "4 wavefronts (1 per SIMD), all MFMA"}, {512, 8, "8 wavefronts (2 per SIMD), all MFMA"},
      {512, 4, "8 wavefronts: 4 MFMA (waves 0-3) + 4 VALU FP64 chains (waves 4-7)"}, {256, 0, "4 wavefronts (1 per SIMD), all VALU"},
      {512, 0, "8 wavefronts (2 per SIMD), all VALU"}};
  for (auto& c : cases) {
    hipLaunchKernelGGL(k_time, dim3(1), dim3(c.threads), 0, 0, dout, 2000, c.nm);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost));
    printf("%-70s ticks per instruction, per wavefront:", c.what);
    for (int w = 0; w < c.threads / 64; ++w) printf(" %.1f", h[1024 + w]);
    printf("\n");
  }
  return 0;
}
