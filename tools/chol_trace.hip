// Phase timing of the fused Cholesky + inverse panel kernel (wall_clock64 stamps of one trailing-tile workgroup and
// the launch's last workgroup per panel: the look-ahead workgroup, or an inverse tile in the last launch).  Includes the library source with -DCHOL_TRACE; links libdcgp for the context.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCHOL_TRACE -I deepcgp_amd/csrc tools/chol_trace.hip -o /tmp/chol_trace -L deepcgp_amd -ldcgp -Wl,-rpath,$PWD/deepcgp_amd
#include "../deepcgp_amd/csrc/chol_fused.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 256, batch = argc > 2 ? atoi(argv[2]) : 6;
  dcgp_ctx* ctx = nullptr;
  if (dcgp_ctx_create(0, &ctx) != 0) return 1;
  std::vector<double> h((size_t)M * M);
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < M; ++j) h[(size_t)i * M + j] = exp(-0.5 * (i - j) * (i - j) / 40.0) + (i == j ? 1e-3 : 0.0);
  std::vector<double*> A(batch), Li(batch), Lt(batch);
  for (int b = 0; b < batch; ++b) {
    hipMalloc(&A[b], sizeof(double) * M * M); hipMalloc(&Li[b], sizeof(double) * M * M); hipMalloc(&Lt[b], sizeof(double) * M * M);
  }
  double **dA, **dLi, **dLt; int* info;
  hipMalloc(&dA, 8 * batch); hipMalloc(&dLi, 8 * batch); hipMalloc(&dLt, 8 * batch); hipMalloc(&info, 4 * batch);
  hipMemcpy(dA, A.data(), 8 * batch, hipMemcpyHostToDevice); hipMemcpy(dLi, Li.data(), 8 * batch, hipMemcpyHostToDevice);
  hipMemcpy(dLt, Lt.data(), 8 * batch, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    for (int b = 0; b < batch; ++b) hipMemcpy(A[b], h.data(), sizeof(double) * M * M, hipMemcpyHostToDevice);
    hipEventRecord(e0, ctx->stream);
    factor_inverse_batched(ctx, dA, dLi, dLt, batch, M, M, info);
    hipEventRecord(e1, ctx->stream);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("chain: %.1f us\n", ms * 1e3);
  }
  std::vector<long long> tr(64 * 32);
  hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_chol_trace), sizeof(long long) * 64 * 32);
  for (int p = 0; p + 1 < M / 32; ++p) {   // the last workgroup of a launch is the look-ahead workgroup (except in the last launch): the launch's critical path
    const long long* t = &tr[p * 32];
    printf("panel %2d look-ahead WG (start +%.2f): load %.2f own potrf %.2f P, D %.2f potrf %.2f store %.2f | total %.2f us\n", p, (t[16] - t[0]) * 0.01,
           (t[17] - t[16]) * 0.01, (t[18] - t[17]) * 0.01, (t[19] - t[18]) * 0.01, (t[20] - t[19]) * 0.01, (t[21] - t[20]) * 0.01, (t[21] - t[16]) * 0.01);
  }
  for (int p = 0; p < M / 32; ++p) {
    const long long* t = &tr[p * 32];
    printf("panel %2d trailing WG: load %.2f potrf %.2f trsm %.2f mfma+store %.2f | inverse WG (start +%.2f): load %.2f potrf %.2f trtri/trsm/Yload %.2f Ynew %.2f mfma+store %.2f  [us]\n", p,
           (t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01, (t[16] - t[0]) * 0.01,
           (t[17] - t[16]) * 0.01, (t[18] - t[17]) * 0.01, (t[19] - t[18]) * 0.01, (t[20] - t[19]) * 0.01, (t[21] - t[20]) * 0.01);
  }
  return 0;
}
