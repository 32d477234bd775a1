// Microbenchmark of the wavefront-level 32x32 routines of deepcgp_amd/csrc/chol_dev.h (cycles per call).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I deepcgp_amd/csrc tools/potrf32_bench.hip -o /tmp/potrf32_bench
#include "chol_dev.h"
#include <cstdio>
#include <vector>
using namespace chol_dev;
__global__ __launch_bounds__(256) void bench(const double* __restrict__ Ain, double* out, unsigned long long* clk, int reps) {
  __shared__ double D[NB][NB + 1];
  __shared__ double Dr[NB];
  __shared__ double col[NB];
  __shared__ double Xs[NB][NB + 1];
  const int tid = threadIdx.x;
  unsigned long long t_f = 0, t_i = 0;
  for (int rep = 0; rep < reps; ++rep) {
    for (int idx = tid; idx < NB * NB; idx += 256) D[idx / NB][idx % NB] = (idx % NB <= idx / NB) ? Ain[idx] : 0.0;
    __syncthreads();
    if (tid < 64) {
      const int r = tid & 31;
      double a[NB];
#pragma unroll
      for (int c = 0; c < NB; ++c) a[c] = D[r][c];
      unsigned long long c0 = clock64();
      int fail = wave_potrf32(a, r, col);
      unsigned long long c1 = clock64();
      t_f += c1 - c0;
      if (tid < 32) {
#pragma unroll
        for (int c = 0; c < NB; ++c) D[r][c] = (c <= r) ? a[c] : 0.0;
        double diag = a[0];
#pragma unroll
        for (int c = 1; c < NB; ++c) diag = (c == r) ? a[c] : diag;
        Dr[r] = 1.0 / diag + fail;
      }
    }
    __syncthreads();
    if (tid < 32) {
      double x[NB];
      unsigned long long c0 = clock64();
      lane_trtri32(D, Dr, tid, x);
      unsigned long long c1 = clock64();
      t_i += c1 - c0;
#pragma unroll
      for (int r = 0; r < NB; ++r) Xs[r][tid] = x[r];
    }
    __syncthreads();
  }
  if (tid < 32) { out[tid] = D[tid][tid]; out[32 + tid] = Xs[tid][tid]; }
  if (tid == 0) { clk[0] = t_f / reps; clk[1] = t_i / reps; }
}
int main() {
  std::vector<double> h(NB * NB);
  for (int i = 0; i < NB; ++i) for (int j = 0; j < NB; ++j) h[i * NB + j] = (i == j) ? 40.0 : 1.0 / (1 + abs(i - j));
  double *dA, *dout; unsigned long long* clk;
  hipMalloc(&dA, sizeof(double) * NB * NB); hipMalloc(&dout, 64 * 8); hipMalloc(&clk, 16);
  hipMemcpy(dA, h.data(), sizeof(double) * NB * NB, hipMemcpyHostToDevice);
  bench<<<1, 256>>>(dA, dout, clk, 50);
  hipDeviceSynchronize();
  unsigned long long c[2]; double o[64];
  hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost); hipMemcpy(o, dout, 512, hipMemcpyDeviceToHost);
  printf("wave_potrf32: %llu cycles (%.2f us at 2.4 GHz); lane_trtri32: %llu cycles (%.2f us); L00=%.6f Linv00=%.6f\n", c[0], c[0] / 2400.0, c[1], c[1] / 2400.0, o[0], o[32]);
  return 0;
}
