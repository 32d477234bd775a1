// natgrad.hip -- the natural-gradient step on every layer's (q_mu, q_sqrt): gpflow.train.NatGradOptimizer(gamma) on
// var_list = [(l.q_mu, l.q_sqrt)] as conv_gp/experiment.py:90-99 sets it up (Salimbeni, Eleftheriadis & Hensman 2018;
// restated from the published algorithm -- gpflow is not in the reference tree).  Per output r, q = N(mu, S = L L^T):
//   natural parameters theta = (S^-1 mu, -S^-1 / 2), expectation parameters eta = (mu, S + mu mu^T),
//   theta <- theta + gamma dELBO/d eta,   dELBO/d eta2 = dELBO/dS =: Sbar,   dELBO/d eta1 = dELBO/dmu - 2 Sbar mu,
//   Sbar = sym(L^-T Phi(L^T dELBO/dL) L^-1)   (the Cholesky adjoint of S = L L^T, as in grad.hip),
//   new precision P = S^-1 - 2 gamma Sbar,  S' = P^-1,  mu' = S' theta1',  L' = chol(S').
// All R outputs of a layer go through the batched Cholesky + inverse chain of the forward pass (chol_fused.hip) three
// times (S -> L^-1;  P -> its factor and inverse;  S' -> L') and through batched gemm_gen products.  Nothing is written to
// the parameters unless every precision matrix of every layer was positive definite (DCGP_ERR_NOT_PD otherwise: the
// caller scales gamma back, conv_gp/experiment.py:36-49).
#include <algorithm>
#include <cstdlib>

#include "model_state.h"
#include "gemm_gen.h"

namespace {

inline unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }

// A_b <- sym(A_b) on the M x M block, identity on the padding (rows / columns M..Mp-1)
__global__ void sym_pad_kernel(double* __restrict__ A, int M, int Mp, long bs, double scale) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
  if (j >= Mp || j > i) return;
  double* P = A + b * bs;
  double v;
  if (i < M && j < M) v = 0.5 * scale * (P[(long)i * Mp + j] + P[(long)j * Mp + i]);
  else v = (i == j) ? 1.0 : 0.0;
  P[(long)i * Mp + j] = v;
  P[(long)j * Mp + i] = v;
}
// Phi on a batch: lower triangle kept, diagonal halved
__global__ void phi_batch_kernel(double* __restrict__ P, int M, int Mp, long bs) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
  if (j >= M) return;
  double* p = P + b * bs + (long)i * Mp + j;
  *p = j < i ? *p : (j == i ? 0.5 * *p : 0.0);
}
// out = a - 2 gamma b  (same padded layout)
__global__ void precision_kernel(const double* __restrict__ a, const double* __restrict__ b, long n, double gamma, double* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] - 2.0 * gamma * b[i];
}
// theta1[r][m] = v1[r][m] + gamma (gmu[m][r] - 2 v2[r][m])
__global__ void theta_kernel(const double* __restrict__ v1, const double* __restrict__ v2, const double* __restrict__ gmu, int M, int R, int ldv,
                             double gamma, double* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * R) return;
  const int r = idx / M, m = idx % M;
  out[(long)r * ldv + m] = v1[(long)r * ldv + m] + gamma * (gmu[(long)m * R + r] - 2.0 * v2[(long)r * ldv + m]);
}
// q_mu[m][r] = munew[r][m];  q_sqrt[r][i][j] = tril(Lnew_r)[i][j]
__global__ void write_qmu_kernel(const double* __restrict__ munew, int M, int R, int ldv, double* __restrict__ q_mu) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * R) return;
  const int r = idx / M, m = idx % M;
  q_mu[(long)m * R + r] = munew[(long)r * ldv + m];
}
__global__ void write_qsqrt_kernel(const double* __restrict__ Lnew, int M, int Mp, long bs, double* __restrict__ q_sqrt) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
  if (j >= M) return;
  q_sqrt[((long)b * M + i) * M + j] = j <= i ? Lnew[b * bs + (long)i * Mp + j] : 0.0;
}

GenGemm mkb(const double* A, long ars, long acs, long abs_, const double* B, long brs, long bcs, long bbs, double* C, long crs, long cbs, int M,
            int N, int K, int batch) {
  GenGemm g;
  g.A = A; g.a_rs = ars; g.a_cs = acs; g.a_bs = abs_; g.B = B; g.b_rs = brs; g.b_cs = bcs; g.b_bs = bbs;
  g.C = C; g.c_rs = crs; g.c_bs = cbs; g.M = M; g.N = N; g.K = K; g.batch = batch;
  return g;
}

// batched factorisation of R padded matrices stored back to back in `A` (overwritten by the factors); inverses to Ainv / AinvT
int factor_batch(dcgp_ctx* ctx, FactorGroup& fg, double* A, double* Ainv, double* AinvT, int R, int Mp) {
  if (fg.K.empty()) {
    fg.Mp = Mp;
    for (int r = 0; r < R; ++r) {
      fg.K.push_back(A + (size_t)r * Mp * Mp);
      fg.Linv.push_back(Ainv + (size_t)r * Mp * Mp);
      fg.LinvT.push_back(AinvT + (size_t)r * Mp * Mp);
    }
  }
  DCGP_TRY(fg.upload(ctx));
  HIP_TRY(ctx, hipMemsetAsync(fg.d_info, 0, (size_t)R * sizeof(int), ctx->stream));
  return factor_inverse_batched(ctx, fg.dK, fg.dLinv, fg.dLinvT, R, Mp, Mp, fg.d_info);
}

}  // namespace

struct NatGradState {   // per layer: fixed buffers + the pointer tables of the three factorisations
  double *Sq = nullptr, *SqInv = nullptr, *SqInvT = nullptr, *Pm = nullptr, *T1 = nullptr, *Sbar = nullptr, *Sinv = nullptr;
  double *Prec = nullptr, *PrecInv = nullptr, *PrecInvT = nullptr, *Snew = nullptr, *SnewInv = nullptr, *SnewInvT = nullptr;
  double *v1 = nullptr, *v2 = nullptr, *theta = nullptr, *munew = nullptr;
  FactorGroup f1, f2, f3;
  std::vector<void*> owned;
  ~NatGradState() {
    f1.release(); f2.release(); f3.release();
    for (void* p : owned) hipFree(p);
  }
  double* alloc(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, n * sizeof(double)) != hipSuccess) return nullptr;
    if (dcgp_poison()) { hipMemset(p, 0xFF, n * sizeof(double)); hipDeviceSynchronize(); }   // debugging aid, see ws_get
    owned.push_back(p);
    return (double*)p;
  }
};

extern "C" int dcgp_model_natgrad_step(dcgp_model* model, double gamma, int* info_host) {
  if (!model || !(gamma > 0)) return model ? ctx_fail(model->ctx, DCGP_ERR_ARG, "natgrad_step: bad arguments") : DCGP_ERR_ARG;
  dcgp_ctx* ctx = model->ctx;
  if (info_host) *info_host = 0;
  ++model->param_version;   // q_mu / q_sqrt change: parameter-only state of earlier steps is not reused (model_state.h)
  const int nl = (int)model->layers.size();
  std::vector<NatGradState*> st(nl, nullptr);
  for (int li = 0; li < nl; ++li) {
    LayerState& L = *model->layers[li];
    if (!L.gZ) return ctx_fail(ctx, DCGP_ERR_ARG, "natgrad_step: call dcgp_elbo_grad first");
    if (!L.has_qsqrt) return ctx_fail(ctx, DCGP_ERR_ARG, "natgrad_step: every layer needs a q_sqrt");
    const int M = L.M, Mp = L.Mp, R = L.R;
    const long mm = (long)Mp * Mp, bsz = (long)R * mm;
    if (!L.natgrad_state) {
      L.natgrad_state = std::shared_ptr<void>(new NatGradState(), [](void* p) { delete static_cast<NatGradState*>(p); });
      NatGradState& s = *static_cast<NatGradState*>(L.natgrad_state.get());
      double** all[] = {&s.Sq, &s.SqInv, &s.SqInvT, &s.Pm, &s.T1, &s.Sbar, &s.Sinv, &s.Prec, &s.PrecInv, &s.PrecInvT, &s.Snew, &s.SnewInv, &s.SnewInvT};
      for (double** p : all)
        if (!(*p = s.alloc((size_t)bsz))) return ctx_fail(ctx, DCGP_ERR_ALLOC, "natgrad_step: allocation failed");
      double** vec[] = {&s.v1, &s.v2, &s.theta, &s.munew};
      for (double** p : vec)
        if (!(*p = s.alloc((size_t)R * Mp))) return ctx_fail(ctx, DCGP_ERR_ALLOC, "natgrad_step: allocation failed");
    }
    NatGradState& s = *static_cast<NatGradState*>(L.natgrad_state.get());
    st[li] = &s;
    const dim3 g3(nblk(Mp), Mp, R), gm3(nblk(M), M, R);
    // S = Lq Lq^T from the current q_sqrt (lower part), padded with the identity
    double* Lq = s.T1;   // reuse as the padded lower-triangular copy for a moment
    DCGP_TRY(pad_copy(ctx, L.q_sqrt, M, M, M, Lq, Mp, Mp, Mp, 1, R, (long)M * M, mm));
    DCGP_TRY(gemm_gen(ctx, mkb(Lq, Mp, 1, mm, Lq, 1, Mp, mm, s.Sq, Mp, mm, M, M, M, R)));
    hipLaunchKernelGGL(sym_pad_kernel, g3, dim3(256), 0, ctx->stream, s.Sq, M, Mp, mm, 1.0);
    LAUNCH_CHECK(ctx);
    // P = Phi(Lq^T dELBO/dLq) before Lq's buffer is reused
    DCGP_TRY(gemm_gen(ctx, mkb(Lq, 1, Mp, mm, L.gq_sqrt, M, 1, (long)M * M, s.Pm, Mp, mm, M, M, M, R)));
    hipLaunchKernelGGL(phi_batch_kernel, gm3, dim3(256), 0, ctx->stream, s.Pm, M, Mp, mm);
    LAUNCH_CHECK(ctx);
    DCGP_TRY(factor_batch(ctx, s.f1, s.Sq, s.SqInv, s.SqInvT, R, Mp));                       // Sq <- Lq, SqInv = Lq^-1
    // Sbar = sym(Lq^-T P Lq^-1),  Sinv = Lq^-T Lq^-1
    DCGP_TRY(gemm_gen(ctx, mkb(s.SqInv, 1, Mp, mm, s.Pm, Mp, 1, mm, s.T1, Mp, mm, M, M, M, R)));
    DCGP_TRY(gemm_gen(ctx, mkb(s.T1, Mp, 1, mm, s.SqInv, Mp, 1, mm, s.Sbar, Mp, mm, M, M, M, R)));
    hipLaunchKernelGGL(sym_pad_kernel, g3, dim3(256), 0, ctx->stream, s.Sbar, M, Mp, mm, 1.0);
    LAUNCH_CHECK(ctx);
    DCGP_TRY(gemm_gen(ctx, mkb(s.SqInv, 1, Mp, mm, s.SqInv, Mp, 1, mm, s.Sinv, Mp, mm, M, M, M, R)));
    hipLaunchKernelGGL(sym_pad_kernel, g3, dim3(256), 0, ctx->stream, s.Sinv, M, Mp, mm, 1.0);
    LAUNCH_CHECK(ctx);
    // theta1 = Sinv mu + gamma (gmu - 2 Sbar mu): column r of q_mu [M][R] is the batch-r vector (row stride R)
    DCGP_TRY(gemm_gen(ctx, mkb(s.Sinv, Mp, 1, mm, L.q_mu, R, 1, 1, s.v1, 1, Mp, M, 1, M, R)));
    DCGP_TRY(gemm_gen(ctx, mkb(s.Sbar, Mp, 1, mm, L.q_mu, R, 1, 1, s.v2, 1, Mp, M, 1, M, R)));
    hipLaunchKernelGGL(theta_kernel, dim3(nblk((long)M * R)), dim3(256), 0, ctx->stream, s.v1, s.v2, L.gq_mu, M, R, Mp, gamma, s.theta);
    LAUNCH_CHECK(ctx);
    // new precision, its inverse = the new covariance, the new factor
    hipLaunchKernelGGL(precision_kernel, dim3(nblk(bsz)), dim3(256), 0, ctx->stream, s.Sinv, s.Sbar, bsz, gamma, s.Prec);
    LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(sym_pad_kernel, g3, dim3(256), 0, ctx->stream, s.Prec, M, Mp, mm, 1.0);   // Sbar's identity padding was subtracted twice
    LAUNCH_CHECK(ctx);
    DCGP_TRY(factor_batch(ctx, s.f2, s.Prec, s.PrecInv, s.PrecInvT, R, Mp));                 // Prec <- its factor Lp, PrecInv = Lp^-1
    DCGP_TRY(gemm_gen(ctx, mkb(s.PrecInv, 1, Mp, mm, s.PrecInv, Mp, 1, mm, s.Snew, Mp, mm, M, M, M, R)));
    hipLaunchKernelGGL(sym_pad_kernel, g3, dim3(256), 0, ctx->stream, s.Snew, M, Mp, mm, 1.0);
    LAUNCH_CHECK(ctx);
    DCGP_TRY(gemm_gen(ctx, mkb(s.Snew, Mp, 1, mm, s.theta, 1, 1, Mp, s.munew, 1, Mp, M, 1, M, R)));
    DCGP_TRY(factor_batch(ctx, s.f3, s.Snew, s.SnewInv, s.SnewInvT, R, Mp));                 // Snew <- L'
  }
  // positive definiteness of every precision / covariance before anything is written
  int bad = 0;
  for (int li = 0; li < nl && !bad; ++li) {
    FactorGroup* fgs[] = {&st[li]->f1, &st[li]->f2, &st[li]->f3};
    for (FactorGroup* fg : fgs) {
      std::vector<int> h(fg->K.size());
      HIP_TRY(ctx, hipMemcpyAsync(h.data(), fg->d_info, h.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      for (int v : h)
        if (v && !bad) bad = v;
    }
  }
  if (bad) {
    if (info_host) *info_host = bad;
    return ctx_fail(ctx, DCGP_ERR_NOT_PD, "natgrad_step: the step leaves the positive-definite cone (column %d); reduce gamma", bad);
  }
  for (int li = 0; li < nl; ++li) {
    LayerState& L = *model->layers[li];
    NatGradState& s = *st[li];
    const int M = L.M, Mp = L.Mp, R = L.R;
    hipLaunchKernelGGL(write_qmu_kernel, dim3(nblk((long)M * R)), dim3(256), 0, ctx->stream, s.munew, M, R, Mp, L.q_mu);
    LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(write_qsqrt_kernel, dim3(nblk(M), M, R), dim3(256), 0, ctx->stream, s.Snew, M, Mp, (long)Mp * Mp, L.q_sqrt);
    LAUNCH_CHECK(ctx);
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}
