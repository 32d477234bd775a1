// Microbenchmark (gfx950): what does a VALU instruction cost beside v_mfma_f64_16x16x4_f64?
// One wave runs `nm` independent MFMAs and `nv` independent VALU ops of a given kind per iteration, interleaved in
// program order; 4 waves per SIMD, every CU busy.  Prints cycles per iteration per SIMD (shader clock from
// wall_clock64 against clock64) so that "shared pipe" (costs add) and "separate pipes" (costs overlap) can be told
// apart for fp64 FMA, fp64 add, fp32 FMA, int32 add, v_ldexp_f64, v_rndne_f64, v_cvt_i32_f64.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, int KIND>
__global__ __launch_bounds__(256) void mixk(double* out, int iters, long long* clk) {
  double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-9;
  d4 acc[NM > 0 ? NM : 1];
  for (int i = 0; i < NM; ++i) acc[i] = d4{0, 0, 0, 0};
  double v[NV > 0 ? NV : 1];
  float f[NV > 0 ? NV : 1];
  int n[NV > 0 ? NV : 1];
  for (int i = 0; i < NV; ++i) { v[i] = i + a; f[i] = i + (float)a; n[i] = i + threadIdx.x; }
  const long long t0 = clock64();
  const long long w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    constexpr int STEPS = NM > NV ? NM : NV;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      // spread both kinds evenly over the iteration
      if (NM > 0 && (s * NM) / STEPS != ((s + 1) * NM) / STEPS) {
        const int i = (s * NM) / STEPS;
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      }
      if (NV > 0 && (s * NV) / STEPS != ((s + 1) * NV) / STEPS) {
        const int i = (s * NV) / STEPS;
        if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(a));
        if (KIND == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[i]) : "v"(b));
        if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"((float)b), "v"((float)a));
        if (KIND == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(it));
        if (KIND == 4) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(v[i]) : "v"(1));
        if (KIND == 5) asm volatile("v_rndne_f64 %0, %0" : "+v"(v[i]));
        if (KIND == 6) { int t; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(t) : "v"(v[i])); n[i] += t; }
        if (KIND == 7) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i]) : "v"(b));
        if (KIND == 8) asm volatile("v_max_f64 %0, %0, %1" : "+v"(v[i]) : "v"(b));
        if (KIND == 9) asm volatile("v_lshl_add_u32 %0, %0, 20, %1" : "+v"(n[i]) : "v"(it));
      }
    }
  }
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  double s = 0;
  for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < NV; ++i) s += v[i] + f[i] + n[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int NM, int NV, int KIND>
void run(const char* name, double* d, long long* clk) {
  const int blocks = 1024, iters = 4000;   // 4 blocks of 4 waves per CU = 4 waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mixk<NM, NV, KIND><<<blocks, 256>>>(d, 10, clk);
  hipEventRecord(e0);
  mixk<NM, NV, KIND><<<blocks, 256>>>(d, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  // wall_clock64 ticks at 100 MHz
  const double ghz = (double)h[0] / ((double)h[1] / 0.1);
  const double cyc_per_iter_simd = ms * 1e-3 * ghz * 1e9 / iters;   // 4 waves per SIMD share it: per-wave-iteration cost = this / 4
  printf("%-44s NM=%2d NV=%2d  %.3f ms  clk %.2f GHz  cycles/iter/wave %.1f  (MFMA alone would be %d, VALU alone at 4 clk %d)\n", name, NM, NV, ms, ghz,
         cyc_per_iter_simd / 4.0, NM * 64, NV * 4);
}

int main() {
  double* d; hipMalloc(&d, 1 << 26);
  long long* clk; hipMalloc(&clk, 64);
  run<8, 0, 0>("MFMA f64 only", d, clk);
  run<0, 32, 0>("v_fma_f64 only", d, clk);
  run<0, 32, 1>("v_add_f64 only", d, clk);
  run<0, 32, 7>("v_mul_f64 only", d, clk);
  run<0, 32, 8>("v_max_f64 only", d, clk);
  run<0, 32, 2>("v_fma_f32 only", d, clk);
  run<0, 32, 3>("v_add_u32 only", d, clk);
  run<0, 32, 9>("v_lshl_add_u32 only", d, clk);
  run<0, 32, 4>("v_ldexp_f64 only", d, clk);
  run<0, 32, 5>("v_rndne_f64 only", d, clk);
  run<0, 32, 6>("v_cvt_i32_f64 only", d, clk);
  run<8, 8, 0>("MFMA + v_fma_f64 (1:1)", d, clk);
  run<8, 32, 0>("MFMA + v_fma_f64 (1:4)", d, clk);
  run<4, 32, 0>("MFMA + v_fma_f64 (1:8)", d, clk);
  run<2, 32, 0>("MFMA + v_fma_f64 (1:16)", d, clk);
  run<8, 32, 1>("MFMA + v_add_f64 (1:4)", d, clk);
  run<8, 32, 2>("MFMA + v_fma_f32 (1:4)", d, clk);
  run<4, 32, 2>("MFMA + v_fma_f32 (1:8)", d, clk);
  run<8, 32, 3>("MFMA + v_add_u32 (1:4)", d, clk);
  run<4, 32, 3>("MFMA + v_add_u32 (1:8)", d, clk);
  run<4, 32, 4>("MFMA + v_ldexp_f64 (1:8)", d, clk);
  run<4, 32, 5>("MFMA + v_rndne_f64 (1:8)", d, clk);
  return 0;
}
