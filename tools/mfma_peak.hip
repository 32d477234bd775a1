// NOTE (round 4): hipcc keeps the accumulators of this loop in AGPRs, and on gfx950 v_mfma_f64_16x16x4_f64 with AGPR accumulators runs at ~105
// cycles per instruction and SIMD -- 47.7 TFLOP/s over the chip -- where VGPR accumulators give 64 cycles / 77 TFLOP/s.  This tool therefore
// UNDERSTATES the pipe; the ceiling bench.py quotes comes from csrc/peaks.hip (dcgp_debug_mfma_f64_rate), which pins them to VGPRs.
// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate (the ceiling the conditional GEMM is priced against)
// and a float4-copy HBM bandwidth ceiling.   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double* out, int iters, unsigned long long* clk = nullptr) {
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3 + 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}
__global__ void copy4(const double4* __restrict__ in, double4* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
}
__global__ void fill8(double* out, size_t n, double v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = v;
}
template <int NACC>
void run(int blocks, int iters, double* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<NACC><<<blocks, 256>>>(d, 10);
  hipEventRecord(e0);
  unsigned long long* clk; hipMalloc(&clk, 16);
  mfma_loop<NACC><<<blocks, 256>>>(d, iters, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  int wcr = 0; hipDeviceGetAttribute(&wcr, hipDeviceAttributeWallClockRate, 0);
  printf("   shader cycles %llu, wall ticks %llu (wall clock rate %d kHz) -> shader clock %.3f GHz; cycles/MFMA/wave %.1f\n", h[0], h[1], wcr,
         (double)h[0] / ((double)h[1] / (wcr * 1e3)) / 1e9, (double)h[0] / ((double)iters * NACC));
  double flops = (double)blocks * 4 * iters * NACC * 2.0 * 16 * 16 * 4;
  printf("mfma_f64_16x16x4: blocks=%d waves/CU=%d nacc=%d  %.2f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", blocks, blocks * 4 / 256,
         NACC, flops / ms / 1e9, 2.4e9 * ms * 1e-3 / ((double)blocks * 4 * iters * NACC / 1024.0));
}
int main() {
  double* d; hipMalloc(&d, 1 << 24);
  run<4>(256, 20000, d); run<8>(256, 20000, d); run<4>(512, 20000, d); run<8>(1024, 10000, d); run<1>(1024, 40000, d); run<2>(256, 40000, d);
  size_t n = (size_t)1 << 28;   // 2 GiB each
  double4 *a, *b; hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
  hipMemset(a, 0, n * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0); copy4<<<2048, 256>>>(a, b, n / 4); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("copy double4: %.1f GB/s (read+write)\n", 2.0 * n * 8 / ms / 1e6);
    hipEventRecord(e0); fill8<<<2048, 256>>>((double*)b, n, 1.0); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("fill double : %.1f GB/s (write only)\n", 1.0 * n * 8 / ms / 1e6);
  }
  return 0;
}
