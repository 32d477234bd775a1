// ops.hip -- single-operator C-ABI entry points (the Level-1 Python classes call these one by one;
// the model-level path in model.hip strings the same building blocks together without host round trips).
#include "layer_impl.h"

namespace {

// B[m][p*N + n] = Kmn[p][m][n]
__global__ void repack_pmn_kernel(const double* __restrict__ Kmn, int P, int M, int N, double* __restrict__ B, long ldb) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)P * M * N;
  if (idx >= total) return;
  int n = (int)(idx % N);
  long t = idx / N;
  int m = (int)(t % M), p = (int)(t / M);
  B[(long)m * ldb + (long)p * N + n] = Kmn[idx];
}

// reference layouts of conditional(): mean [N,P,R], var [R,P,N]; internal column j = p*N + n
__global__ void finalize_api_kernel(const double* __restrict__ s1p, int nrb1, const double* __restrict__ s2p, int nrb3,
                                    const double* __restrict__ mu, long ldk, const double* __restrict__ knn, int P,
                                    int N, int R, double* __restrict__ out_mean, double* __restrict__ out_var) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long Kc = (long)P * N;
  if (idx >= Kc * R) return;
  int r = (int)(idx / Kc);
  long j = idx % Kc;
  int p = (int)(j / N), n = (int)(j % N);
  double s1 = 0.0, s2 = 0.0;
  for (int b = 0; b < nrb1; ++b) s1 += s1p[(long)b * ldk + j];
  if (s2p)
    for (int b = 0; b < nrb3; ++b) s2 += s2p[((long)r * nrb3 + b) * ldk + j];
  out_var[((long)r * P + p) * N + n] = (knn[j] - s1) + s2;
  out_mean[((long)n * P + p) * R + r] = mu[(long)r * ldk + j];
}

__global__ void additive_kdiag_kernel(int N, int P, double variance, const double* __restrict__ w, double* __restrict__ out) {
  __shared__ double s;
  if (threadIdx.x == 0) {
    double acc = 0.0;
    for (int p = 0; p < P; ++p) acc += w[p] * variance;
    s = acc / P;
  }
  __syncthreads();
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) out[n] = s;
}

// padded operands of ONE GP built from caller (device) arrays, in ctx workspaces
struct TmpGp {
  GpMats g;
  FactorGroup fg;
  ~TmpGp() { fg.release(); }
};

int tmp_gp_build(dcgp_ctx* ctx, TmpGp& t, const std::string& pfx, int M, int R, const double* Kmm, const double* q_mu,
                 const double* q_sqrt, int* info_host, int prep_white = -1) {
  const int Mp = round_up(M, 16);
  const size_t mm = (size_t)Mp * Mp;
  GpMats& g = t.g;
  g.M = M; g.Mp = Mp; g.R = R; g.Rp = round_up(R, 16);
  g.K = (double*)ws_get(ctx, pfx + "K", mm * sizeof(double));
  g.Linv = (double*)ws_get(ctx, pfx + "Linv", mm * sizeof(double));
  g.LinvT = (double*)ws_get(ctx, pfx + "LinvT", mm * sizeof(double));
  g.Lq = (double*)ws_get(ctx, pfx + "Lq", (size_t)R * mm * sizeof(double));
  g.qmu = (double*)ws_get(ctx, pfx + "qmu", (size_t)Mp * g.Rp * sizeof(double));
  g.G = (double*)ws_get(ctx, pfx + "G", (size_t)R * mm * sizeof(double));
  g.alpha = (double*)ws_get(ctx, pfx + "alpha", (size_t)Mp * g.Rp * sizeof(double));
  if (!g.K || !g.Linv || !g.LinvT || !g.Lq || !g.qmu || !g.G || !g.alpha) return DCGP_ERR_ALLOC;
  if (Kmm) DCGP_TRY(pad_copy(ctx, Kmm, M, M, M, g.K, Mp, Mp, Mp, 2, 1, 0, 0));
  if (q_sqrt) DCGP_TRY(pad_copy(ctx, q_sqrt, M, M, M, g.Lq, Mp, Mp, Mp, 1, R, (long)M * M, (long)Mp * Mp));
  if (q_mu) DCGP_TRY(pad_copy(ctx, q_mu, M, R, R, g.qmu, g.Rp, Mp, g.Rp, 0, 1, 0, 0));
  if (Kmm) {
    t.fg.Mp = Mp;
    t.fg.K = {g.K}; t.fg.Linv = {g.Linv}; t.fg.LinvT = {g.LinvT};
    DCGP_TRY(t.fg.run(ctx));
    int info = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&info, t.fg.d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (info_host) *info_host = info;
    if (info) return ctx_fail(ctx, DCGP_ERR_NOT_PD, "Cholesky: matrix not positive definite at column %d", info);
    if (prep_white == 1) { g.G = g.Lq; g.alpha = g.qmu; }
    if (prep_white >= 0) DCGP_TRY(cond_prep(ctx, g, prep_white, q_sqrt != nullptr));
  }
  return DCGP_OK;
}

int d2d(dcgp_ctx* ctx, double* dst, const double* src, size_t n) {
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  return DCGP_OK;
}

#define ARG_CHECK(cond, msg) \
  if (!(cond)) return ctx ? ctx_fail(ctx, DCGP_ERR_ARG, msg) : DCGP_ERR_ARG

}  // namespace

int additive_kdiag_async(dcgp_ctx* ctx, int N, int P, double variance, const double* w, double* out_N) {
  hipLaunchKernelGGL(additive_kdiag_kernel, dim3(std::min(64, (N + 255) / 256)), dim3(256), 0, ctx->stream, N, P, variance, w, out_N);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

extern "C" {

static BaseKernel rbf_bk(double variance, double lengthscale) {
  BaseKernel b;
  b.type = 0; b.variance = variance; b.p1 = 1.0 / (lengthscale * lengthscale); b.p2 = 0.0;
  return b;
}
static BaseKernel acos_bk(double variance, double weight_variance, double bias_variance) {
  BaseKernel b;
  b.type = 1; b.variance = variance; b.p1 = weight_variance; b.p2 = bias_variance;
  return b;
}
static int kuu_impl(dcgp_ctx* ctx, const double* Z, int M, int L, BaseKernel bk, double jitter, double* out_MM) {
  DCGP_TRY(rbf_gram_padded(ctx, Z, M, L, bk, jitter, out_MM, M, M));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}
static int kuf_impl(dcgp_ctx* ctx, const double* X, int N, int H, int W, int C, int f, int stride, const double* Z, int M,
                    BaseKernel bk, double* out, int layout);

int dcgp_kuu_rbf(dcgp_ctx* ctx, const double* Z, int M, int L, double variance, double lengthscale, double jitter,
                 double* out_MM) {
  ARG_CHECK(ctx && Z && out_MM && M > 0 && L > 0 && variance > 0 && lengthscale > 0, "kuu_rbf: bad args");
  return kuu_impl(ctx, Z, M, L, rbf_bk(variance, lengthscale), jitter, out_MM);
}

int dcgp_kuu_acos(dcgp_ctx* ctx, const double* Z, int M, int L, double variance, double weight_variance, double bias_variance,
                  double jitter, double* out_MM) {
  ARG_CHECK(ctx && Z && out_MM && M > 0 && L > 0 && variance > 0 && weight_variance > 0 && bias_variance >= 0, "kuu_acos: bad args");
  return kuu_impl(ctx, Z, M, L, acos_bk(variance, weight_variance, bias_variance), jitter, out_MM);
}

int dcgp_kuf_patches_rbf(dcgp_ctx* ctx, const double* X, int N, int H, int W, int C, int f, int stride, const double* Z,
                         int M, double variance, double lengthscale, double* out, int layout) {
  ARG_CHECK(ctx && X && Z && out && N > 0 && M > 0 && f > 0 && stride > 0 && f <= H && f <= W && C > 0 && variance > 0 &&
                lengthscale > 0 && (layout == 0 || layout == 1),
            "kuf_patches_rbf: bad args");
  return kuf_impl(ctx, X, N, H, W, C, f, stride, Z, M, rbf_bk(variance, lengthscale), out, layout);
}

int dcgp_kuf_patches_acos(dcgp_ctx* ctx, const double* X, int N, int H, int W, int C, int f, int stride, const double* Z,
                          int M, double variance, double weight_variance, double bias_variance, double* out, int layout) {
  ARG_CHECK(ctx && X && Z && out && N > 0 && M > 0 && f > 0 && stride > 0 && f <= H && f <= W && C > 0 && variance > 0 &&
                weight_variance > 0 && bias_variance >= 0 && (layout == 0 || layout == 1),
            "kuf_patches_acos: bad args");
  return kuf_impl(ctx, X, N, H, W, C, f, stride, Z, M, acos_bk(variance, weight_variance, bias_variance), out, layout);
}

static int kuf_impl(dcgp_ctx* ctx, const double* X, int N, int H, int W, int C, int f, int stride, const double* Z, int M,
                    BaseKernel bk, double* out, int layout) {
  ViewGeom v;
  v.set(H, W, C, f, stride);
  const int Mp = round_up(M, 16), Lp = round_up(v.L, 4);
  if (bk.type == 0) {   // RBF: the unit sweep in its storing form (head_units.hip)
    HeadUnitsArgs h;
    h.X = X; h.n_mod = N; h.N = N;
    h.H = H; h.W = W; h.C = C; h.f = f; h.s = stride; h.Wo = v.Wo; h.P = v.P; h.L = v.L; h.Lq = sweep_lq(v.L);
    h.M = M; h.Mp = Mp; h.kzx_rows = M;
    const double ls = 1.0 / sqrt(bk.p1);
    h.csq = sqrt(1.4426950408889634074 * bk.p1); h.log2var = log2(bk.variance);
    h.kuf = out;
    if (layout == 0) { h.sP = (long)M * N; h.sM = N; h.sN = 1; }
    else { h.sM = (long)N * v.P; h.sN = v.P; h.sP = 1; }
    head_units_plan(&h);
    if (head_units_ok(h)) {
      double* ZS = (double*)ws_get(ctx, "op_ZS", (size_t)h.Lq * Mp * sizeof(double));
      if (!ZS) return DCGP_ERR_ALLOC;
      DCGP_TRY(sweep_operand(ctx, Z, nullptr, M, Mp, v.L, bk.variance, ls, ZS));
      h.ZS = ZS;
      DCGP_TRY(head_units(ctx, h));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      return DCGP_OK;
    }
  }
  double* ZT = (double*)ws_get(ctx, "op_ZT", (size_t)Lp * Mp * sizeof(double));
  double* zn = (double*)ws_get(ctx, "op_zn", (size_t)Mp * sizeof(double));
  if (!ZT || !zn) return DCGP_ERR_ALLOC;
  DCGP_TRY(z_transpose_norms(ctx, Z, M, v.L, ZT, Mp, Lp, zn));
  PatchRbfArgs a;
  a.X = X; a.N = N; a.n_mod = N;
  a.H = H; a.W = W; a.C = C; a.f = f; a.s = stride; a.Ho = v.Ho; a.Wo = v.Wo; a.P = v.P; a.L = v.L;
  a.ZT = ZT; a.zn = zn; a.M = M; a.Mp = Mp; a.Lp = Lp;
  a.bk = bk;
  a.out = out;
  if (layout == 0) { a.sP = (long)M * N; a.sM = N; a.sN = 1; }
  else { a.sM = (long)N * v.P; a.sN = v.P; a.sP = 1; }
  DCGP_TRY(patch_rbf(ctx, a, "kuf"));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}

int dcgp_potrf_lower(dcgp_ctx* ctx, double* A_MM, int M, int* info_host) {
  ARG_CHECK(ctx && A_MM && M > 0, "potrf_lower: bad args");
  if (info_host) *info_host = 0;
  TmpGp t;
  const int Mp = round_up(M, 16);
  double* K = (double*)ws_get(ctx, "op_potrf_K", (size_t)Mp * Mp * sizeof(double));
  if (!K) return DCGP_ERR_ALLOC;
  DCGP_TRY(pad_copy(ctx, A_MM, M, M, M, K, Mp, Mp, Mp, 2, 1, 0, 0));
  t.fg.Mp = Mp;
  t.fg.K = {K}; t.fg.Linv = {K}; t.fg.LinvT = {K};
  DCGP_TRY(t.fg.upload(ctx));
  DCGP_TRY(potrf_batched(ctx, t.fg.dK, nullptr, 1, Mp, Mp, t.fg.d_info));
  DCGP_TRY(pad_copy(ctx, K, M, M, Mp, A_MM, M, M, M, 0, 1, 0, 0));
  int info = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&info, t.fg.d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (info_host) *info_host = info;
  if (info) return ctx_fail(ctx, DCGP_ERR_NOT_PD, "Cholesky: matrix not positive definite at column %d", info);
  return DCGP_OK;
}

int dcgp_trtri_lower(dcgp_ctx* ctx, const double* L_MM, int M, double* Linv_MM) {
  ARG_CHECK(ctx && L_MM && Linv_MM && M > 0, "trtri_lower: bad args");
  TmpGp t;
  const int Mp = round_up(M, 16);
  const size_t mm = (size_t)Mp * Mp;
  double* L = (double*)ws_get(ctx, "op_trtri_L", mm * sizeof(double));
  double* X = (double*)ws_get(ctx, "op_trtri_X", mm * sizeof(double));
  if (!L || !X) return DCGP_ERR_ALLOC;
  DCGP_TRY(pad_copy(ctx, L_MM, M, M, M, L, Mp, Mp, Mp, 2, 1, 0, 0));
  t.fg.Mp = Mp;
  t.fg.K = {L}; t.fg.Linv = {X}; t.fg.LinvT = {X};
  DCGP_TRY(t.fg.upload(ctx));
  DCGP_TRY(trtri_batched(ctx, t.fg.dK, t.fg.dLinv, nullptr, 1, Mp, Mp));
  DCGP_TRY(pad_copy(ctx, X, M, M, Mp, Linv_MM, M, M, M, 0, 1, 0, 0));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}

int dcgp_conditional(dcgp_ctx* ctx, const double* Kmn, const double* Kmm, const double* Knn, const double* f,
                     const double* q_sqrt, int white, int P, int M, int N, int R, double* out_mean, double* out_var,
                     int* info_host) {
  ARG_CHECK(ctx && Kmn && Kmm && Knn && f && out_mean && out_var && P > 0 && M > 0 && N > 0 && R > 0, "conditional: bad args");
  if (info_host) *info_host = 0;
  TmpGp t;
  DCGP_TRY(tmp_gp_build(ctx, t, "op_cond_", M, R, Kmm, f, q_sqrt, info_host, white ? 1 : 0));
  const int Mp = t.g.Mp;
  const long Kc = (long)P * N, ldb = round_up_l(Kc, 128);
  double* B = (double*)ws_get(ctx, "op_cond_B", (size_t)Mp * ldb * sizeof(double));
  if (!B) return DCGP_ERR_ALLOC;
  if (Mp > M) HIP_TRY(ctx, hipMemsetAsync(B + (size_t)M * ldb, 0, (size_t)(Mp - M) * ldb * sizeof(double), ctx->stream));
  long total = (long)P * M * N;
  hipLaunchKernelGGL(repack_pmn_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, Kmn, P, M, N, B, ldb);
  LAUNCH_CHECK(ctx);
  CondScratch sc;
  DCGP_TRY(cond_core(ctx, t.g, B, ldb, (int)Kc, white, q_sqrt != nullptr, "op_cond_", &sc));
  long tot = Kc * R;
  hipLaunchKernelGGL(finalize_api_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, sc.s1p, sc.nrb1,
                     sc.s2p, sc.nrb3, sc.mu, ldb, Knn, P, N, R, out_mean, out_var);
  LAUNCH_CHECK(ctx);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}

int dcgp_svgp_conditional(dcgp_ctx* ctx, const double* Kuf, const double* Ku, const double* Kdiag, const double* q_mu,
                          const double* q_sqrt, int white, int M, int N, int R, double* out_mean, double* out_var,
                          int* info_host) {
  ARG_CHECK(ctx && Kuf && Ku && Kdiag && q_mu && out_mean && out_var && M > 0 && N > 0 && R > 0, "svgp_conditional: bad args");
  if (info_host) *info_host = 0;
  TmpGp t;
  DCGP_TRY(tmp_gp_build(ctx, t, "op_svgp_", M, R, Ku, q_mu, q_sqrt, info_host, white ? 1 : 0));
  const int Mp = t.g.Mp;
  const long ldb = round_up_l(N, 128);
  double* B = (double*)ws_get(ctx, "op_svgp_B", (size_t)Mp * ldb * sizeof(double));
  if (!B) return DCGP_ERR_ALLOC;
  DCGP_TRY(pad_copy(ctx, Kuf, M, N, N, B, (int)ldb, Mp, (int)ldb, 0, 1, 0, 0));
  CondScratch sc;
  DCGP_TRY(cond_core(ctx, t.g, B, ldb, N, white, q_sqrt != nullptr, "op_svgp_", &sc));
  FinalizeArgs fa;
  fa.s1p = sc.s1p; fa.nrb1 = sc.nrb1; fa.s2p = sc.s2p; fa.nrb3 = sc.nrb3; fa.mu = sc.mu; fa.ldk = ldb;
  fa.Kc = N; fa.R = R; fa.knn_vec = Kdiag;
  fa.out_mean = out_mean; fa.out_var = out_var;
  DCGP_TRY(finalize_layer(ctx, fa));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}

int dcgp_conv_layer_forward(dcgp_ctx* ctx, const double* X, int N, int H, int W, int C, int f, int stride, const double* Z,
                            int M, int R, double variance, double lengthscale, const double* q_mu, const double* q_sqrt,
                            int white, int identity_mean, const double* z, double jitter, double* out_sample,
                            double* out_mean, double* out_var, int* info_host) {
  ARG_CHECK(ctx && X && Z && q_mu && N > 0, "conv_layer_forward: bad args");
  ARG_CHECK(!(out_sample && !z), "conv_layer_forward: out_sample needs z");
  ARG_CHECK(!(identity_mean && f % 2 == 0), "conv_layer_forward: Conv2dMean supports odd filter sizes only");
  if (info_host) *info_host = 0;
  LayerState L;
  DCGP_TRY(L.init(ctx, false, H, W, C, f, stride, M, R, white, identity_mean, 0, variance, lengthscale,
                  /*need_prior=*/false));   // no KL here: skip the prior factorisation
  L.has_qsqrt = q_sqrt != nullptr;
  DCGP_TRY(d2d(ctx, L.Z, Z, (size_t)M * L.v.L));
  DCGP_TRY(d2d(ctx, L.q_mu, q_mu, (size_t)M * R));
  if (q_sqrt) DCGP_TRY(d2d(ctx, L.q_sqrt, q_sqrt, (size_t)R * M * M));
  DCGP_TRY(L.prepare(jitter));
  FactorGroup fg;
  fg.Mp = L.Mp; fg.K = {L.g.K}; fg.Linv = {L.g.Linv}; fg.LinvT = {L.g.LinvT};
  int rc = fg.run(ctx);
  int info = 0;
  if (rc == DCGP_OK) {
    hipMemcpyAsync(&info, fg.d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    hipStreamSynchronize(ctx->stream);
  }
  fg.release();
  if (rc != DCGP_OK) return rc;
  if (info_host) *info_host = info;
  if (info) return ctx_fail(ctx, DCGP_ERR_NOT_PD, "Cholesky: matrix not positive definite at column %d", info);
  DCGP_TRY(cond_prep(ctx, L.g, white, L.has_qsqrt));
  DCGP_TRY(conv_forward(ctx, L, X, N, N, 1, 0, z, 0, 0, jitter, out_sample, out_mean, out_var, "op_conv_"));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}

// ConvKernel.Kzx / Kdiag through the unit sweep (head_units.hip): what the model-level head runs.  false: shape not covered
static bool convkernel_units(dcgp_ctx* ctx, const double* X, int N, const ViewGeom& v, const double* Z, int M, double variance,
                             double lengthscale, const double* w, double* out_MN, double* out_N, int* rc) {
  HeadUnitsArgs h;
  h.X = X; h.n_mod = N; h.N = N;
  h.H = v.H; h.W = v.W; h.C = v.C; h.f = v.f; h.s = v.s; h.Wo = v.Wo; h.P = v.P; h.L = v.L; h.Lq = sweep_lq(v.L);
  h.M = Z ? M : 16; h.Mp = round_up(h.M, 16);
  h.csq = sqrt(1.4426950408889634074) / lengthscale; h.log2var = log2(variance);
  h.w = w; h.kzx = out_MN; h.ldk = N; h.kzx_rows = M; h.kzx_scale = 1.0 / (double)v.P;
  *rc = DCGP_OK;
  h.want_kd = out_N != nullptr;
  h.tail_mode = (int)ctx->opt.head_tail;
  h.occ_force = (int)ctx->opt.sweep_occ;
  head_units_plan(&h);
  if (!head_units_ok(h)) return false;
  if (out_N) {
    h.kd = (double*)ws_get(ctx, "kdiag_partial", (size_t)N * h.n_kd * sizeof(double));   // [N][n_kd] partial sums (head_units_plan)
    if (!h.kd) { *rc = DCGP_ERR_ALLOC; return true; }
  }
  if (Z) {
    double* ZS = (double*)ws_get(ctx, "op_ZS", (size_t)h.Lq * h.Mp * sizeof(double));
    if (!ZS) { *rc = DCGP_ERR_ALLOC; return true; }
    if ((*rc = sweep_operand(ctx, Z, nullptr, M, h.Mp, v.L, variance, lengthscale, ZS)) != DCGP_OK) return true;
    h.ZS = ZS;
  }
  if ((*rc = head_units(ctx, h)) != DCGP_OK) return true;
  if (out_N) *rc = kdiag_reduce(ctx, h.kd, h.n_kd, N, 1.0 / ((double)v.P * (double)v.P), out_N);
  if (*rc == DCGP_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) *rc = ctx_fail(ctx, DCGP_ERR_HIP, "convkernel: stream synchronisation failed");
  return true;
}

int dcgp_convkernel_kzx(dcgp_ctx* ctx, const double* X, int N, int H, int W, int C, int f, int stride, const double* Z,
                        int M, double variance, double lengthscale, const double* w, double* out_MN) {
  ARG_CHECK(ctx && X && Z && w && out_MN && N > 0 && M > 0 && f > 0 && stride > 0 && f <= H && f <= W && C > 0 &&
                variance > 0 && lengthscale > 0, "convkernel_kzx: bad args");
  ViewGeom v;
  v.set(H, W, C, f, stride);
  int urc = DCGP_OK;
  if (convkernel_units(ctx, X, N, v, Z, M, variance, lengthscale, w, out_MN, nullptr, &urc)) return urc;
  const int Mp = round_up(M, 16), Lp = round_up(v.L, 4);
  double* ZT = (double*)ws_get(ctx, "op_ZT", (size_t)Lp * Mp * sizeof(double));
  double* zn = (double*)ws_get(ctx, "op_zn", (size_t)Mp * sizeof(double));
  if (!ZT || !zn) return DCGP_ERR_ALLOC;
  DCGP_TRY(z_transpose_norms(ctx, Z, M, v.L, ZT, Mp, Lp, zn));
  PatchRbfArgs a;
  a.X = X; a.N = N; a.n_mod = N;
  a.H = H; a.W = W; a.C = C; a.f = f; a.s = stride; a.Ho = v.Ho; a.Wo = v.Wo; a.P = v.P; a.L = v.L;
  a.ZT = ZT; a.zn = zn; a.M = M; a.Mp = Mp; a.Lp = Lp;
  a.bk = rbf_bk(variance, lengthscale);
  a.out = out_MN; a.sM = N; a.sN = 1; a.sP = 0;
  a.w = w; a.scale = 1.0 / (double)v.P; a.reduce = 1;
  DCGP_TRY(patch_rbf(ctx, a, "head_kzx"));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}

int dcgp_convkernel_kdiag(dcgp_ctx* ctx, const double* X, int N, int H, int W, int C, int f, int stride, double variance,
                          double lengthscale, const double* w, double* out_N) {
  ARG_CHECK(ctx && X && w && out_N && N > 0 && f > 0 && stride > 0 && f <= H && f <= W && C > 0 && variance > 0 &&
                lengthscale > 0, "convkernel_kdiag: bad args");
  {
    ViewGeom v;
    v.set(H, W, C, f, stride);
    int urc = DCGP_OK;
    if (convkernel_units(ctx, X, N, v, nullptr, 0, variance, lengthscale, w, nullptr, out_N, &urc)) return urc;
  }
  DCGP_TRY(head_kdiag(ctx, X, N, N, H, W, C, f, stride, rbf_bk(variance, lengthscale), w, out_N));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}

int dcgp_additive_kdiag(dcgp_ctx* ctx, int N, int P, double variance, const double* w, double* out_N) {
  ARG_CHECK(ctx && w && out_N && N > 0 && P > 0, "additive_kdiag: bad args");
  DCGP_TRY(additive_kdiag_async(ctx, N, P, variance, w, out_N));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}

int dcgp_gauss_kl(dcgp_ctx* ctx, const double* q_mu, const double* q_sqrt, const double* K, int M, int R,
                  double* out_host, int* info_host) {
  ARG_CHECK(ctx && q_mu && q_sqrt && out_host && M > 0 && R > 0, "gauss_kl: bad args");
  if (info_host) *info_host = 0;
  TmpGp t;
  DCGP_TRY(tmp_gp_build(ctx, t, "op_kl_", M, R, K, q_mu, q_sqrt, info_host));
  double* kl4 = (double*)ws_get(ctx, "op_kl_4", 4 * sizeof(double));
  if (!kl4) return DCGP_ERR_ALLOC;
  const int white = K == nullptr;
  DCGP_TRY(kl_layer(ctx, t.g, t.g.K, t.g.LinvT, white, "op_kl_", kl4));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->h_scratch, kl4, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  const double* k = ctx->h_scratch;
  double two = k[0] - (double)M * R - k[1] + k[3];
  if (!white) two += R * k[2];
  *out_host = 0.5 * two;
  return DCGP_OK;
}

}  // extern "C"
