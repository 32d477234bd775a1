// Microbenchmark: v_mfma_f64_16x16x4_f64 rate for operand patterns of a register-tiled GEMM inner loop
// (VGPR accumulators, 16 waves/CU): how much of the 78.6 TF peak survives fresh A/B operands per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
// MODE 0: one A, one B for every MFMA.  MODE 1: 2 A x 4 B register tile (8 accumulators), operands constant.
// MODE 2: 2 x 4 tile, operands re-read from LDS every k-substep (ds_read_b64), no barriers.
// MODE 3: 2 x 2 tile (4 accumulators), operands from LDS every substep.
template <int MODE>
__global__ __launch_bounds__(512, 4) void k(double* out, int iters) {
  __shared__ double sh[16 * 160];
  for (int i = threadIdx.x; i < 16 * 160; i += 512) sh[i] = 1.0 + i * 1e-6;
  __syncthreads();
  const int lane = threadIdx.x & 63, lrow = lane >> 4, lcol = lane & 15;
  d4 acc[2][4];
  for (int x = 0; x < 2; ++x) for (int y = 0; y < 4; ++y) acc[x][y] = d4{0, 0, 0, 0};
  double a[2] = {1.0 + lane * 1e-3, 2.0 - lane * 1e-3}, b[4] = {0.5, 0.25 + lane * 1e-4, 0.125, 1.5};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 16; kk += 4) {
      if (MODE >= 2) {
#pragma unroll
        for (int x = 0; x < 2; ++x) a[x] = sh[(kk + lrow) * 144 + x * 16 + lcol];
#pragma unroll
        for (int y = 0; y < (MODE == 3 ? 2 : 4); ++y) b[y] = sh[(kk + lrow) * 144 + 32 + y * 16 + lcol];
      }
      if (MODE == 0) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc[x][y], 0, 0, 0);
      } else if (MODE == 3) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
      } else {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
      }
    }
  }
  double s = 0;
  for (int x = 0; x < 2; ++x) for (int y = 0; y < 4; ++y) s += acc[x][y][0] + acc[x][y][1] + acc[x][y][2] + acc[x][y][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, double* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 512, iters = 4000;
  k<MODE><<<blocks, 512>>>(d, 10);
  hipEventRecord(e0);
  k<MODE><<<blocks, 512>>>(d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double nm = (MODE == 3 ? 16.0 : 32.0);
  printf("%-44s %.2f TFLOP/s\n", name, blocks * 8.0 * iters * nm * 2048.0 / ms / 1e9);
}
int main() {
  double* d; hipMalloc(&d, 1 << 26);
  run<0>("same A,B every MFMA (8 acc)", d);
  run<1>("2x4 register tile, constant operands", d);
  run<2>("2x4 tile, operands from LDS each substep", d);
  run<3>("2x2 tile, operands from LDS each substep", d);
  return 0;
}
