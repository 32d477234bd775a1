// peaks.hip -- the two ceilings bench.py prices the kernels against, MEASURED on the device the bench runs on instead of quoted:
// the sustained v_mfma_f64_16x16x4_f64 rate (the conditional's products, conv_gp/conditionals.py:29-65) and the rate of a pure store
// sweep in the K_uf sweep's own tile pattern (conv_gp/layers.py:23-32 writes P x M x N').  Debugging / reporting aids: nothing on the
// hot path calls them.
#include <cstdlib>

#include "common.h"

namespace {

// MODE 0: accumulators where the compiler puts them for __builtin_amdgcn_mfma_f64_16x16x4f64 in a loop like this one -- AGPRs;
// MODE 1: accumulators pinned to VGPRs (inline asm), which is where the product kernels of this library keep theirs.  On gfx950 the two
// differ by a factor: ~105 against ~64 cycles per instruction and SIMD (47.7 against ~75 TFLOP/s over the chip) -- the AGPR form is NOT
// the ceiling, and a rate measured with it (tools/mfma_peak.hip) understates what the pipe does.
template <int MODE>
__global__ __launch_bounds__(256) void mfma_rate_kernel(double* out, int iters) {
  d4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = threadIdx.x * 1e-3 + i; b[i] = threadIdx.x * 2e-3 + 1.0 + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], acc[i], 0, 0, 0);
      else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[i]));
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one wave per (image n, 16-row fragment u): per column fragment j and v = 0..3, rows 16 u + lrow + 4 v, columns n P + 16 j + lcol --
// four 128-byte segments per instruction, rows ld * 8 bytes apart (the MFMA accumulator layout the sweep stores from)
__global__ __launch_bounds__(256) void tile_store_kernel(double* out, int N, int P, int Mp, long ld, double v) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lrow = lane >> 4, lcol = lane & 15;
  const int nfm = Mp / 16, nfp = (P + 15) / 16;
  const long unit = (long)blockIdx.x * 4 + wave;
  const int n = (int)(unit / nfm), u = (int)(unit % nfm);
  if (n >= N) return;
  double* base = out + (long)(16 * u + lrow) * ld + (long)n * P + lcol;
  for (int j = 0; j < nfp; ++j) {
    if (16 * j + lcol < P) {
#pragma unroll
      for (int q = 0; q < 4; ++q) base[(long)4 * q * ld + 16 * j] = v + j;
    }
  }
}

}  // namespace

extern "C" int dcgp_debug_mfma_f64_rate(dcgp_ctx* ctx, double* tflops_out) {
  if (!ctx || !tflops_out) return DCGP_ERR_ARG;
  const int blocks = 1024, iters = 20000;   // 4 waves per SIMD on every CU, ~10-15 ms per form
  double* buf = (double*)ws_get(ctx, "peak_mfma", (size_t)blocks * 256 * sizeof(double));
  if (!buf) return DCGP_ERR_ALLOC;
  hipEvent_t e0, e1;
  HIP_TRY(ctx, hipEventCreate(&e0));
  HIP_TRY(ctx, hipEventCreate(&e1));
  double best = 0.0;
  for (int mode = 0; mode < 2; ++mode) {   // the better of the two accumulator placements (see the kernel)
    auto launch = [&](int n) {
      if (mode == 0) hipLaunchKernelGGL(mfma_rate_kernel<0>, dim3(blocks), dim3(256), 0, ctx->stream, buf, n);
      else hipLaunchKernelGGL(mfma_rate_kernel<1>, dim3(blocks), dim3(256), 0, ctx->stream, buf, n);
    };
    launch(2000);
    HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
    launch(iters);
    HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
    // 16 x 16 x 4 x 2 flop per instruction, 4 per iteration per wave, 4 waves per workgroup
    const double tf = 2048.0 * 4.0 * iters * 4.0 * blocks / (ms * 1e-3) / 1e12;
    if (getenv("DCGP_PEAK_VERBOSE")) fprintf(stderr, "mfma_rate mode %d: %.2f TFLOP/s\n", mode, tf);
    if (tf > best) best = tf;
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *tflops_out = best;
  return DCGP_OK;
}

extern "C" int dcgp_debug_store_rate(dcgp_ctx* ctx, int N, int P, int M, double* gbs_out) {
  if (!ctx || !gbs_out || N <= 0 || P <= 0 || M <= 0) return DCGP_ERR_ARG;
  const int Mp = round_up(M, 16);
  const long ld = col_ld((long)N * P);
  double* buf = (double*)ws_get(ctx, "peak_store", (size_t)Mp * ld * sizeof(double));
  if (!buf) return DCGP_ERR_ALLOC;
  const unsigned grid = (unsigned)(((long)N * (Mp / 16) + 3) / 4);
  hipEvent_t e0, e1;
  HIP_TRY(ctx, hipEventCreate(&e0));
  HIP_TRY(ctx, hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(tile_store_kernel, dim3(grid), dim3(256), 0, ctx->stream, buf, N, P, Mp, ld, 1.0);
  const int reps = 10;
  HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(tile_store_kernel, dim3(grid), dim3(256), 0, ctx->stream, buf, N, P, Mp, ld, 1.0);
  HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
  HIP_TRY(ctx, hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *gbs_out = (double)M * N * P * 8.0 * reps / (ms * 1e-3) / 1e9;
  return DCGP_OK;
}
