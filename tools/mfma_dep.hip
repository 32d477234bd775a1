// Microbenchmark: v_mfma_f64_16x16x4_f64 throughput vs the number of INDEPENDENT accumulators a wave cycles through
// (dependent-issue distance) and the number of resident waves per SIMD.  VGPR accumulators.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_dep.hip -o /tmp/mfma_dep && /tmp/mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC, int WPS>   // WPS = waves per SIMD (block = 4 * WPS waves, one block per CU)
__global__ __launch_bounds__(WPS * 256, 1) void k(double* out, int iters) {
  d4 acc[NACC];
  for (int x = 0; x < NACC; ++x) acc[x] = d4{0, 0, 0, 0};
  const int lane = threadIdx.x & 63;
  double a = 1.0 + lane * 1e-3, b = 0.5 - lane * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8 / NACC; ++u)
#pragma unroll
      for (int x = 0; x < NACC; ++x) acc[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[x], 0, 0, 0);
  }
  double s = 0;
  for (int x = 0; x < NACC; ++x) s += acc[x][0] + acc[x][1] + acc[x][2] + acc[x][3];
  if (s == 12345.678) out[0] = s;
}
template <int NACC, int WPS>
void run(double* out) {
  const int iters = 4000, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, WPS>), dim3(blocks), dim3(WPS * 256), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, WPS>), dim3(blocks), dim3(WPS * 256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)blocks * WPS * 4 * iters * 8;
  printf("acc=%d waves/SIMD=%d : %7.2f TF/s   (%.0f cycles per MFMA per wave at 2.38 GHz)\n", NACC, WPS, mf * 2048 / ms / 1e9,
         ms * 1e-3 * 2.38e9 / (iters * 8.0));
}
int main() {
  double* out; hipMalloc(&out, 64);
  run<1, 1>(out); run<2, 1>(out); run<4, 1>(out); run<8, 1>(out);
  run<1, 2>(out); run<2, 2>(out); run<4, 2>(out);
  run<1, 4>(out); run<2, 4>(out); run<4, 4>(out);
  return 0;
}
