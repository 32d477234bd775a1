// Microbenchmark of the 32 x 32 diagonal-block factor + inverse variants (one wavefront of a 256-thread workgroup).
#include "diag_variants.h"
#include <cstdio>
#include <cmath>
#include <vector>
using namespace chol_dev;
template <int MODE>
__global__ __launch_bounds__(256, 2) void kb(const double* __restrict__ Ain, double* out, long long* clk, int reps) {
  __shared__ double D[NB][NB + 1];
  __shared__ double Xs[NB][NB + 1];
  __shared__ __attribute__((aligned(16))) double col[96];
  __shared__ double T[16][17];
  __shared__ double sc[64];
  const int tid = threadIdx.x;
  long long tot = 0, wtot = 0; int fail = 0;
  for (int rep = 0; rep < reps; ++rep) {
    for (int idx = tid; idx < NB * NB; idx += 256) D[idx / NB][idx % NB] = (idx % NB <= idx / NB) ? Ain[idx] : ((idx % NB == idx / NB) ? 1.0 : 0.0);
    __syncthreads();
    if constexpr (MODE == 5) {
      const long long c0 = clock64(), w0 = wall_clock64();
      const int f = potrf_inv32_wg<NB + 1>(D, Xs, col, T, sc, tid);
      if (tid < 64) { fail = f; tot += clock64() - c0; wtot += wall_clock64() - w0; }
    } else if (tid < 64) {
      const long long c0 = clock64(), w0 = wall_clock64();
      if constexpr (MODE == 9) fail = wave_potrf_inv32_2x16(D, Xs, reinterpret_cast<double (&)[64]>(col), T, tid);
      else if constexpr (MODE == 2) fail = potrf_inv32_halves<NB + 1>(D, Xs, col, T, tid);
      else fail = potrf_inv32_new<MODE, NB + 1>(D, Xs, reinterpret_cast<double (&)[64]>(col), T, tid);
      const long long c1 = clock64();
      tot += c1 - c0; wtot += wall_clock64() - w0;
    }
    __syncthreads();
  }
  for (int idx = tid; idx < NB * NB; idx += 256) { out[idx] = D[idx / NB][idx % NB]; out[1024 + idx] = Xs[idx / NB][idx % NB]; }
  if (tid == 0) { out[2048] = fail; clk[0] = tot / reps; clk[1] = wtot * 10 / reps; }
}
int main() {
  const int n = NB;
  std::vector<double> h(n * n), L(n * n, 0.0), X(n * n, 0.0);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) h[i * n + j] = 5.0 * exp(-0.5 * (i - j) * (i - j) / 30.0) + (i == j ? 1e-3 : 0.0);
  // host reference
  std::vector<double> a = h;
  for (int c = 0; c < n; ++c) {
    double s = a[c * n + c]; for (int k = 0; k < c; ++k) s -= L[c * n + k] * L[c * n + k];
    L[c * n + c] = sqrt(s);
    for (int i = c + 1; i < n; ++i) { double t = a[i * n + c]; for (int k = 0; k < c; ++k) t -= L[i * n + k] * L[c * n + k]; L[i * n + c] = t / L[c * n + c]; }
  }
  for (int c = 0; c < n; ++c) for (int r = 0; r < n; ++r) { double s = (r == c) ? 1.0 : 0.0; for (int k = 0; k < r; ++k) s -= L[r * n + k] * X[k * n + c]; X[r * n + c] = s / L[r * n + r]; }
  double *dA, *dout; long long* clk;
  hipMalloc(&dA, sizeof(double) * n * n); hipMalloc(&dout, 2049 * 8); hipMalloc(&clk, 16);
  hipMemcpy(dA, h.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
  auto run = [&](const char* name, auto kern) {
    hipMemset(dout, 0, 2049 * 8);
    kern<<<1, 256>>>(dA, dout, clk, 200);
    hipDeviceSynchronize();
    long long c[2]; std::vector<double> o(2049);
    hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost); hipMemcpy(o.data(), dout, 2049 * 8, hipMemcpyDeviceToHost);
    double eL = 0, eX = 0, mX = 0;
    for (int i = 0; i < n * n; ++i) { eL = fmax(eL, fabs(o[i] - L[i])); eX = fmax(eX, fabs(o[1024 + i] - X[i])); mX = fmax(mX, fabs(X[i])); }
    printf("%-28s %6lld cycles, %.3f us  fail=%g  max|dL|=%.2e  max|dX|/max|X|=%.2e\n", name, c[0], c[1] * 0.001, o[2048], eL, eX / mX);
  };
  run("existing 2x16 (LDS line)", kb<9>);
  run("new, LDS line cleaned", kb<1>);
  run("halves layout", kb<2>);
  run("whole workgroup, LDL^T glue", kb<5>);
  return 0;
}
