// LDS bank-conflict rule of ds_read_b64 / ds_read2_b64 / ds_read_b128 on gfx950: cycles per wavefront instruction for
// lane addresses base + stride * lane (in doubles), one wavefront alone on its CU.  Compared with the half-wavefront
// model of k_form_factor_2d's pitch rule (32 lanes x 1 double, collision when the addresses differ and agree mod 32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(64) void k(const int* __restrict__ idx, int iters, long long* out, double* sink) {
  __shared__ double lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = i;
  __syncthreads();
  const double* p = lds + idx[threadIdx.x];
  double acc = 0.0;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (OP == 0) { double v; asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p)); acc += v; }
      if (OP == 1) { double2 v; asm volatile("ds_read2_b64 %0, %1 offset1:1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p)); acc += v.x + v.y; }
      if (OP == 2) { double2 v; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p)); acc += v.x + v.y; }
    }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  sink[threadIdx.x] = acc;
}

int model(const std::vector<int>& idx, int nd) {  // passes of the half-wavefront model for nd consecutive doubles per lane
  int total = 0;
  for (int h = 0; h < 2; ++h)
    for (int d = 0; d < nd; ++d) {
      int worst = 0;
      for (int bank = 0; bank < 32; ++bank) {
        std::vector<int> seen;
        for (int l = 32 * h; l < 32 * h + 32; ++l) {
          int a = idx[l] + d;
          if (a % 32 != bank) continue;
          bool dup = false;
          for (int s : seen) dup = dup || s == a;
          if (!dup) seen.push_back(a);
        }
        worst = worst > (int)seen.size() ? worst : (int)seen.size();
      }
      total += worst;
    }
  return total;
}

int main() {
  int* didx; long long* dout; double* dsink;
  CHECK(hipMalloc(&didx, 64 * sizeof(int))); CHECK(hipMalloc(&dout, sizeof(long long))); CHECK(hipMalloc(&dsink, 64 * sizeof(double)));
  const int iters = 2000;
  auto run = [&](const std::vector<int>& idx, const char* name) {
    CHECK(hipMemcpy(didx, idx.data(), 64 * sizeof(int), hipMemcpyHostToDevice));
    long long t[3];
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, didx, iters, dout, dsink); CHECK(hipMemcpy(&t[0], dout, 8, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, didx, iters, dout, dsink); CHECK(hipMemcpy(&t[1], dout, 8, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, didx, iters, dout, dsink); CHECK(hipMemcpy(&t[2], dout, 8, hipMemcpyDeviceToHost));
    printf("%-28s b64 %6.1f  read2_b64 %6.1f  b128 %6.1f cycles/instr | model passes b64 %d, 2 doubles %d\n", name,
           (double)t[0] / (iters * 8.0), (double)t[1] / (iters * 8.0), (double)t[2] / (iters * 8.0), model(idx, 1), model(idx, 2));
    return 0;
  };
  for (int stride : {0, 1, 2, 3, 4, 5, 8, 16, 17, 31, 32, 33, 64, 65}) {
    std::vector<int> idx(64);
    for (int l = 0; l < 64; ++l) idx[l] = (stride * l) % 4096 * 1 + (stride == 0 ? 0 : 0);
    for (int l = 0; l < 64; ++l) idx[l] = (idx[l] / 2) * 2;   // 16-byte aligned for b128
    if (stride % 2) for (int l = 0; l < 64; ++l) idx[l] = (stride * l) % 4096;
    char nm[64]; snprintf(nm, 64, "stride %d doubles", stride);
    if (stride % 2 == 0 || true) run(idx, nm);
  }
  // the sampler's pattern: digital lines of cells, bank = pitch cx + cy, for two pitches, skews 0 and 1, a few angles
  for (int pitch : {130, 131})
    for (int sk : {0, 1})
      for (double beta : {0.3, 0.7, 1.1, 2.0}) {
        std::vector<int> idx(64);
        for (int l = 0; l < 64; ++l) {
          const double u = 50.3 + (sk * l) * cos(beta) - l * sin(beta), v = 50.7 + (sk * l) * sin(beta) + l * cos(beta);
          idx[l] = ((int)floor(u) * pitch + (int)floor(v)) % 8000;
        }
        char nm[64]; snprintf(nm, 64, "pitch %d skew %d beta %.1f", pitch, sk, beta);
        run(idx, nm);
      }
  return 0;
}
