// Stand-alone timing harness for gemm_tn (libdcgp.so internal entry point): dense vs triangular, batched or not.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I deepcgp_amd/csrc tools/gemm_bench.hip -o /tmp/gemm_bench -L deepcgp_amd -ldcgp -Wl,-rpath,$PWD/deepcgp_amd
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 256, K = argc > 2 ? atoi(argv[2]) : 46080, R = argc > 3 ? atoi(argv[3]) : 10;
  dcgp_ctx* ctx = nullptr;
  if (dcgp_ctx_create(0, &ctx) != 0) return 1;
  double *W, *B, *C, *S;
  const long ldb = (K + 127) / 128 * 128;
  hipMalloc(&W, sizeof(double) * R * M * M); hipMalloc(&B, sizeof(double) * M * ldb); hipMalloc(&C, sizeof(double) * M * ldb);
  hipMalloc(&S, sizeof(double) * R * 64 * ldb);
  std::vector<double> h((size_t)M * ldb);
  for (auto& v : h) v = (rand() / (double)RAND_MAX) - 0.5;
  hipMemcpy(B, h.data(), sizeof(double) * M * ldb, hipMemcpyHostToDevice);
  for (int r = 0; r < R; ++r) hipMemcpy(W + (size_t)r * M * M, h.data() + r * 17, sizeof(double) * M * M, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int tri = 0; tri < 3; ++tri) for (int mode = 0; mode < 2; ++mode) {   // mode 0: colsq only, batched R; mode 1: store C, single
    GemmArgs a;
    a.Wt = W; a.ldw = M; a.B = B; a.ldb = (int)ldb; a.Mi = M; a.Mk = M; a.Kc = K; a.tri = tri;
    int nrb = 0;
    if (mode == 0) { a.wBatch = (long)M * M; a.nW = R; a.colsq = S; a.sBatch = 64 * ldb; a.sRowBlk = ldb; }
    else { a.C = C; a.ldc = (int)ldb; a.colsq = S; a.sRowBlk = ldb; }
    gemm_tn(ctx, a, &nrb);
    hipStreamSynchronize(ctx->stream);
    hipEventRecord(e0, ctx->stream);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) gemm_tn(ctx, a, &nrb);
    hipEventRecord(e1, ctx->stream);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double batch = mode == 0 ? R : 1;
    const double dense = 2.0 * M * M * (double)K * batch, alg = tri ? dense / 2 : dense;
    printf("M=%d K=%d tri=%d %-22s %8.1f us  algorithmic %6.2f TF/s  (dense-equivalent %6.2f TF/s)\n", M, K, tri,
           mode == 0 ? "x R, colsq only" : "x 1, store C + colsq", ms * 1e3, alg / ms / 1e9, dense / ms / 1e9);
  }
  return 0;
}
