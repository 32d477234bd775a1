// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate (VGPR accumulator form, 16 waves/CU) with constant vs
// random operand data, plus the shader clock actually held (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(double* out, int iters, int randomize, unsigned long long* clk) {
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  double a[8], b[8];
  unsigned long long s = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1 + blockIdx.x * 977);
  for (int i = 0; i < 8; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    double ra = (double)(s >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    double rb = (double)(s >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    a[i] = randomize ? ra : 1.0;
    b[i] = randomize ? rb : 1.0;
  }
  d4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[(i + it) & 7], acc[i], 0, 0, 0);
  }
  double r = 0;
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}
int main() {
  double* d; hipMalloc(&d, 1 << 26);
  unsigned long long* clk; hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 512, iters = 40000;
  for (int rnd = 0; rnd < 2; ++rnd) for (int rep = 0; rep < 2; ++rep) {
    k<<<blocks, 512>>>(d, 10, rnd, clk);
    hipEventRecord(e0);
    k<<<blocks, 512>>>(d, iters, rnd, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%s operands: %.3f ms  %.2f TFLOP/s   shader clock %.3f GHz\n", rnd ? "random  " : "constant", ms,
           blocks * 8.0 * iters * 8 * 2048.0 / ms / 1e9, (double)h[0] / ((double)h[1] / 1e8) / 1e9);
  }
  return 0;
}
