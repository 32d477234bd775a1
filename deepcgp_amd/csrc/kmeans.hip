// kmeans.hip -- Lloyd's k-means for the inducing-patch initialisation (conv_gp/kernels.py:147-164: sklearn
// cluster.KMeans(n_clusters=M, init='random') on 100 M random patches).  Same algorithm -- random observations as the
// initial centres, assign to the nearest centre, move every centre to the mean of its members, stop when the centres
// stop moving -- as GEMMs: distances from X C^T, the new centres from A^T X with A the one-hot assignment matrix
// (deterministic split-k, no atomics: a given seed gives the same centres every run).  An empty cluster keeps its centre.
#include <vector>

#include "common.h"
#include "gemm_gen.h"

namespace {

__global__ void km_norms_kernel(const double* __restrict__ C, int k, int d, double* __restrict__ cn) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k) return;
  double s = 0.0;
  for (int l = 0; l < d; ++l) { const double v = C[(long)j * d + l]; s += v * v; }
  cn[j] = s;
}
// row i: nearest centre by |c_j|^2 - 2 x_i.c_j (ties -> lowest j); one-hot row of A
__global__ void km_assign_kernel(const double* __restrict__ D, const double* __restrict__ cn, long n, int k, double* __restrict__ A,
                                 int* __restrict__ assign) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double best = 0.0;
  int bj = 0;
  for (int j = 0; j < k; ++j) {
    const double v = cn[j] - 2.0 * D[i * k + j];
    if (j == 0 || v < best) { best = v; bj = j; }
  }
  const int old = assign[i];
  if (old >= 0) A[i * k + old] = 0.0;
  A[i * k + bj] = 1.0;
  assign[i] = bj;
}
// counts[j] = sum_i A[i][j]: one block per cluster
__global__ __launch_bounds__(256) void km_count_kernel(const double* __restrict__ A, long n, int k, double* __restrict__ counts) {
  __shared__ double red[256];
  const int j = blockIdx.x, t = threadIdx.x;
  double s = 0.0;
  for (long i = t; i < n; i += 256) s += A[i * k + j];
  red[t] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  if (t == 0) counts[j] = red[0];
}
// C[j] = sums[j] / counts[j] (kept if the cluster is empty); shift2[j] = |C_new - C_old|^2
__global__ void km_update_kernel(const double* __restrict__ sums, const double* __restrict__ counts, int k, int d, double* __restrict__ C,
                                 double* __restrict__ shift2) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k) return;
  double s = 0.0;
  if (counts[j] > 0.0) {
    for (int l = 0; l < d; ++l) {
      const double v = sums[(long)j * d + l] / counts[j];
      const double dv = v - C[(long)j * d + l];
      s += dv * dv;
      C[(long)j * d + l] = v;
    }
  }
  shift2[j] = s;
}

}  // namespace

extern "C" int dcgp_kmeans(dcgp_ctx* ctx, const double* X, long n, int d, int k, const int32_t* init_rows_host, int max_iter, double tol,
                           double* centers, int* iters_out) {
  if (!ctx || !X || !centers || !init_rows_host || n <= 0 || d <= 0 || k <= 0 || k > n || max_iter <= 0 || n > 0x7fffffffL)
    return ctx ? ctx_fail(ctx, DCGP_ERR_ARG, "kmeans: bad arguments") : DCGP_ERR_ARG;
  double* D = (double*)ws_get(ctx, "km_D", (size_t)n * k * sizeof(double));
  double* A = (double*)ws_get(ctx, "km_A", (size_t)n * k * sizeof(double));
  double* cn = (double*)ws_get(ctx, "km_cn", (size_t)k * sizeof(double));
  double* sums = (double*)ws_get(ctx, "km_sums", (size_t)k * d * sizeof(double));
  double* counts = (double*)ws_get(ctx, "km_counts", (size_t)k * sizeof(double));
  double* shift2 = (double*)ws_get(ctx, "km_shift", (size_t)k * sizeof(double));
  int* assign = (int*)ws_get(ctx, "km_assign", (size_t)n * sizeof(int));
  if (!D || !A || !cn || !sums || !counts || !shift2 || !assign) return DCGP_ERR_ALLOC;
  for (int j = 0; j < k; ++j) {   // init='random': k observations
    const long r = init_rows_host[j];
    if (r < 0 || r >= n) return ctx_fail(ctx, DCGP_ERR_ARG, "kmeans: initial row %ld out of range", r);
    HIP_TRY(ctx, hipMemcpyAsync(centers + (size_t)j * d, X + (size_t)r * d, (size_t)d * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  }
  HIP_TRY(ctx, hipMemsetAsync(A, 0, (size_t)n * k * sizeof(double), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(assign, 0xff, (size_t)n * sizeof(int), ctx->stream));   // -1: no previous assignment
  std::vector<double> h(k);
  int it = 0;
  for (; it < max_iter; ++it) {
    GenGemm g;   // D = X C^T
    g.A = X; g.a_rs = d; g.a_cs = 1; g.B = centers; g.b_rs = 1; g.b_cs = d; g.C = D; g.c_rs = k; g.M = (int)n; g.N = k; g.K = d;
    DCGP_TRY(gemm_gen(ctx, g));
    hipLaunchKernelGGL(km_norms_kernel, dim3((k + 255) / 256), dim3(256), 0, ctx->stream, centers, k, d, cn);
    LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(km_assign_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, D, cn, n, k, A, assign);
    LAUNCH_CHECK(ctx);
    GenGemm s;   // sums = A^T X
    s.A = A; s.a_rs = 1; s.a_cs = k; s.B = X; s.b_rs = d; s.b_cs = 1; s.C = sums; s.c_rs = d; s.M = k; s.N = d; s.K = (int)n;
    DCGP_TRY(gemm_gen(ctx, s));
    hipLaunchKernelGGL(km_count_kernel, dim3(k), dim3(256), 0, ctx->stream, A, n, k, counts);
    LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(km_update_kernel, dim3((k + 255) / 256), dim3(256), 0, ctx->stream, sums, counts, k, d, centers, shift2);
    LAUNCH_CHECK(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), shift2, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double tot = 0.0;
    for (double v : h) tot += v;
    if (tot <= tol) { ++it; break; }
  }
  if (iters_out) *iters_out = it;
  return DCGP_OK;
}
