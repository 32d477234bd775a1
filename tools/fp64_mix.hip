// Microbenchmark: do v_mfma_f64_16x16x4_f64 and v_fma_f64 (VALU) share a pipe?  Runs MFMA-only, VALU-only and
// mixed (half the waves of every SIMD each) and prints sustained TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
// mode bit0: waves with even id run MFMA; bit1: waves with odd id run VALU; (mode 4: all MFMA, mode 8: all VALU)
__global__ __launch_bounds__(512) void mix(double* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = (mode == 4) || ((mode & 1) && !(wave & 1));
  const bool do_valu = (mode == 8) || ((mode & 2) && (wave & 1));
  double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-9;
  double s = 0;
  if (do_mfma) {
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else if (do_valu) {
    double v[32];
    for (int i = 0; i < 32; ++i) v[i] = i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i) v[i] = fma(v[i], b, a);
    }
    for (int i = 0; i < 32; ++i) s += v[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double* d; hipMalloc(&d, 1 << 26);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 512, iters = 20000;   // 2 blocks of 8 waves per CU = 4 waves / SIMD
  int modes[] = {4, 8, 1, 2, 3};
  const char* names[] = {"all waves MFMA", "all waves VALU fma_f64", "even waves MFMA, odd idle", "odd waves VALU, even idle", "even MFMA + odd VALU"};
  for (int m = 0; m < 5; ++m) {
    mix<<<blocks, 512>>>(d, 10, modes[m]);
    hipEventRecord(e0);
    mix<<<blocks, 512>>>(d, iters, modes[m]);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves = blocks * 8.0;
    double nm = (modes[m] == 4) ? waves : ((modes[m] & 1) ? waves / 2 : 0);
    double nv = (modes[m] == 8) ? waves : ((modes[m] & 2) ? waves / 2 : 0);
    double fm = nm * iters * 8 * 2048.0, fv = nv * iters * 32 * 128.0;
    printf("%-28s %.3f ms  MFMA %.2f TF  VALU %.2f TF  total %.2f TF\n", names[m], ms, fm / ms / 1e9, fv / ms / 1e9, (fm + fv) / ms / 1e9);
  }
  return 0;
}
