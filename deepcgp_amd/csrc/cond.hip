// cond.hip -- the doubly-stochastic SVGP conditional (conv_gp/conditionals.py:6-67) and the scalar
// terms around it (gauss_kl, RobustMax variational expectations, reparameterisation).
//
// With Lm = chol(Kmm) and Linv = inv(Lm) the reference's per-patch triangular solves become k-major matrix-core
// products on the [M x (P*N)] Kuf matrix (gemm.hip); the second solve (A = Linv' A1, :44-47) is folded into the small
// operands G_r = Linv Lq_r and alpha = Linv q_mu (cond_prep / prep_solve_kernel), so the big matrix is passed twice:
//     A1 = Linv Kuf              lower-triangular W      -> s1[j]   = sum_m A1[m,j]^2        (:31-33,:40)
//     T_r = G_r' A1 (no store)   upper-triangular W, x R -> s2[r,j] = sum_m T_r[m,j]^2        (:55-65)
//     mu[r,j] = sum_m alpha[m,r] A1[m,j]                                                      (:50)
// and a fused epilogue forms var = Knn - s1 + s2, the N x (P*R) output layout of
// conv_gp/layers.py:128-131 and the sample mean + z*sqrt(var + jitter).  (A few-column problem -- the head -- takes
// the one-launch route of head_cond.hip instead.)
#include "layer_impl.h"
#include "tail_dev.h"
#include "rng.h"

namespace {

// 64 columns x R per block: phase 1 gathers the per-row-block partial sums (column-contiguous reads),
// phase 2 writes the (column, r) pairs r-fastest == the N x (P*R) layout (contiguous writes).
__global__ __launch_bounds__(256) void finalize_kernel(FinalizeArgs a) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* vm = sm;                 // [R][65]
  double* vv = sm + a.R * 65;      // [R][65]
  const int j0 = blockIdx.x * 64, tid = threadIdx.x;
  for (int idx = tid; idx < 64 * a.R; idx += 256) {
    int jj = idx & 63, r = idx >> 6, j = j0 + jj;
    double m = 0.0, v = 0.0;
    if (j < a.Kc) {
      double s1 = 0.0;
      for (int b = 0; b < a.nrb1; ++b) s1 += a.s1p[(long)b * a.ldk + j];
      double s2 = 0.0;
      if (a.s2p)
        for (int b = 0; b < a.nrb3; ++b) s2 += a.s2p[((long)r * a.nrb3 + b) * a.ldk + j];
      double knn = a.knn_vec ? a.knn_vec[j] : a.knn_scalar;
      v = (knn - s1) + s2;
      m = a.mu[(long)r * a.ldk + j];
      if (a.idm && r == 0) {
        const long jg = a.col0 + j;
        int n = (int)(jg / a.P), p = (int)(jg - (long)n * a.P);
        int oh = p / a.Wo, ow = p - oh * a.Wo;
        int c0 = a.f / 2;
        m += a.X[(((long)(n % a.n_mod) * a.H + oh * a.s + c0) * a.W + ow * a.s + c0) * a.C];
      }
    }
    vm[r * 65 + jj] = m;
    vv[r * 65 + jj] = v;
  }
  __syncthreads();
  const int total = 64 * a.R;
  for (int idx = tid; idx < total; idx += 256) {
    int jj = idx / a.R, r = idx - jj * a.R, j = j0 + jj;
    if (j >= a.Kc) continue;
    double m = vm[r * 65 + jj], v = vv[r * 65 + jj];
    long e = (a.col0 + j) * a.R + r;
    for (int s = 0; s < a.rep; ++s) {
      long o = (long)s * a.rep_stride + e;
      if (a.out_mean) a.out_mean[o] = m;
      if (a.out_var) a.out_var[o] = v;
      if (a.out_sample) {
        double zz = a.z ? a.z[o] : philox_normal(a.seed, a.stream_id, rng_index(a.rmap, o));
        a.out_sample[o] = m + zz * sqrt(v + a.jitter);
      }
    }
  }
}

// ---- KL small terms: one block.  ap / tp are the per-row-block column sums of squares of inv(Lp) q_mu and
// inv(Lp) Lq_r produced by the GEMM epilogue; this kernel adds them up together with the log-determinants.
__global__ __launch_bounds__(1024) void kl_small_kernel(const double* __restrict__ Lp, const double* __restrict__ Lq,
                                                        const double* __restrict__ qmu, int Rp,
                                                        const double* __restrict__ ap, long ap_count,
                                                        const double* __restrict__ tp, long tp_count, int M, int Mp,
                                                        int R, int white, double* __restrict__ kl4) {
  __shared__ double red[4][1024];
  const int tid = threadIdx.x;
  double mah = 0.0, ldq = 0.0, ldp = 0.0, tr = 0.0;
  for (int idx = tid; idx < M * R; idx += 1024) {
    int i = idx % M, r = idx / M;
    if (white) {
      double al = qmu[(long)i * Rp + r];
      mah += al * al;
    }
    double d = Lq[((long)r * Mp + i) * Mp + i];
    ldq += log(d * d);
  }
  if (!white) {
    for (long idx = tid; idx < ap_count; idx += 1024) mah += ap[idx];
    for (int i = tid; i < M; i += 1024) {
      double d = Lp[(long)i * Mp + i];
      ldp += log(d * d);
    }
    for (long idx = tid; idx < tp_count; idx += 1024) tr += tp[idx];
  } else {
    for (long idx = tid; idx < (long)R * Mp * Mp; idx += 1024) {
      double v = Lq[idx];
      tr += v * v;
    }
  }
  red[0][tid] = mah; red[1][tid] = ldq; red[2][tid] = ldp; red[3][tid] = tr;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o)
      for (int q = 0; q < 4; ++q) red[q][tid] += red[q][tid + o];
    __syncthreads();
  }
  if (tid < 4) kl4[tid] = red[tid][0];
}

// ---- RobustMax variational expectations: 32 lanes per row, lane g < 20 owns one Gauss-Hermite node ----
__global__ __launch_bounds__(256) void varexp_kernel(const double* __restrict__ mu, const double* __restrict__ var,
                                                     const int32_t* __restrict__ y, int n_rows, int n_labels, int K,
                                                     double eps, const double* __restrict__ gh,
                                                     double* __restrict__ out, int predict) {
  const int tid = threadIdx.x, g = tid & 31;
  const int slot = blockIdx.x * 8 + (tid >> 5);
  // predict mode: slot = row * K + class, output the class probability
  const int row = predict ? slot / K : slot;
  const int yi_p = predict ? slot % K : 0;
  const bool live = row < n_rows;
  double contrib = 0.0;
  if (live && g < 20) {
    const int yi = predict ? yi_p : y[row % n_labels];
    const double* m = mu + (long)row * K;
    const double* v = var + (long)row * K;
    double t = m[yi] + gh[g] * sqrt(fmax(2.0 * v[yi], 1e-10));
    double prod = 1.0;
    for (int k = 0; k < K; ++k) {
      if (k == yi) continue;
      double dist = (t - m[k]) / sqrt(fmax(v[k], 1e-10));
      double cdf = 0.5 * (1.0 + erf(dist * 0.70710678118654752440));
      prod *= cdf * (1.0 - 2e-4) + 1e-4;
    }
    contrib = prod * gh[20 + g] * 0.56418958354775628695;   // w / sqrt(pi)
  }
  for (int o = 1; o < 32; o <<= 1) contrib += __shfl_xor(contrib, o);
  if (live && g == 0) {
    double p = contrib;
    if (predict)
      out[slot] = p * (1.0 - eps) + (1.0 - p) * (eps / (K - 1.0));
    else
      out[row] = p * log(1.0 - eps) + (1.0 - p) * log(eps / (K - 1.0));
  }
}

// ---- the ELBO tail in one launch: RobustMax expectations of every row (as varexp_kernel), their sum by the LAST block to
// arrive (fixed order: reproducible), and -- with fin.nl > 0 -- the assembly  data * scale - sum_l KL_l  with the status words of
// the factorisations.  Three dependent launches and their gaps otherwise, at the very end of the step where nothing hides them.
__global__ __launch_bounds__(256) void elbo_tail_kernel(TailArgs t, KlTail kl) {
  __shared__ double red[4 * 256];
  __shared__ unsigned last;
  const int tid = threadIdx.x, g = tid & 31;
  const int nb_rows = (t.n_rows + 7) / 8;
  if ((int)blockIdx.x >= nb_rows) {
    // the KL pieces of one layer (parameter-only inputs, ready since the chain): no launches, no stream of their own
    const int l = blockIdx.x - nb_rows;
    kl_pieces_block(kl.l[l], t.scal + 4 + 4 * l, red);
  } else {
    const int row = blockIdx.x * 8 + (tid >> 5);
    const bool live = row < t.n_rows;
    double contrib = live ? robustmax_node(t.mu + (long)row * t.K, t.var + (long)row * t.K, t.y[row % t.n_labels], t.K, t.gh, g) : 0.0;
    for (int o = 1; o < 32; o <<= 1) contrib += __shfl_xor(contrib, o);
    if (live && g == 0) t.ve[row] = robustmax_logp(contrib, t.eps, t.K);
  }
  if (!last_to_arrive(t.ticket, gridDim.x, &last)) return;
  double s = 0.0;
  for (int i = tid; i < t.n_rows; i += 256) s += __hip_atomic_load(t.ve + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) elbo_assemble(t, red[0] * t.inv_s);
}

__global__ __launch_bounds__(1024) void reduce_sum_kernel(const double* __restrict__ in, long n, double scale,
                                                          double* __restrict__ out) {
  __shared__ double red[1024];
  double s = 0.0;
  for (long i = threadIdx.x; i < n; i += 1024) s += in[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] * scale;
}

// up to REDUCE_JOBS_MAX independent sums in one launch.  grid (nb, jobs): block (c, k) sums the c-th of nb equal ranges of job k (a multiple of 1024
// elements each: a fixed partition, so the result is the same from run to run) and the LAST block of a job to arrive adds the nb partial sums up in
// index order.  nb == 1 is the single-block sum of reduce_sum_kernel.  (One block per job whatever its length read a training step's 94 MB of
// lengthscale-gradient partials at 0.4 TB/s: 250 us of the 3.9 ms step.)
__global__ __launch_bounds__(1024) void reduce_sum_multi_kernel(ReduceJobs j, double* __restrict__ part, unsigned* ticket, int nb) {
  __shared__ double red[1024];
  __shared__ unsigned last;
  const int k = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
  const double* __restrict__ in = j.in[k];
  const long n = j.n[k];
  const long chunk = (((n + nb - 1) / nb) + 1023) & ~1023L;
  const long lo = c * chunk, hi = lo + chunk < n ? lo + chunk : n;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  long i = lo + tid;
  for (; i + 3 * 1024 < hi; i += 4 * 1024) { s0 += in[i]; s1 += in[i + 1024]; s2 += in[i + 2048]; s3 += in[i + 3072]; }
  for (; i < hi; i += 1024) s0 += in[i];
  red[tid] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (nb == 1) {
    if (tid == 0) j.out[k][0] = red[0] * j.scale[k];
    return;
  }
  if (tid == 0) part[(long)k * nb + c] = red[0];
  if (!last_to_arrive(ticket + k, (unsigned)nb, &last)) return;
  red[tid] = tid < nb ? __hip_atomic_load(part + (long)k * nb + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) j.out[k][0] = red[0] * j.scale[k];
}

__global__ void reparam_kernel(const double* __restrict__ mean, const double* __restrict__ var,
                               const double* __restrict__ z, size_t n, double jitter, double* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = mean[i] + z[i] * sqrt(var[i] + jitter);
}

// Gauss-Hermite nodes/weights (physicists' convention, numpy.polynomial.hermite.hermgauss) by Newton
// iteration on the orthonormal recurrence.
void hermgauss_host(int n, double* x, double* w) {
  const double PIM4 = 0.7511255444649425;
  const int m = (n + 1) / 2;
  double z = 0.0;
  for (int i = 0; i < m; ++i) {
    if (i == 0) z = sqrt((double)(2 * n + 1)) - 1.85575 * pow((double)(2 * n + 1), -0.16667);
    else if (i == 1) z -= 1.14 * pow((double)n, 0.426) / z;
    else if (i == 2) z = 1.86 * z - 0.86 * x[0];
    else if (i == 3) z = 1.91 * z - 0.91 * x[1];
    else z = 2.0 * z - x[i - 2];
    double pp = 0.0;
    for (int it = 0; it < 100; ++it) {
      double p1 = PIM4, p2 = 0.0;
      for (int j = 0; j < n; ++j) {
        double p3 = p2;
        p2 = p1;
        p1 = z * sqrt(2.0 / (j + 1)) * p2 - sqrt((double)j / (j + 1)) * p3;
      }
      pp = sqrt(2.0 * n) * p2;
      double z1 = z;
      z = z1 - p1 / pp;
      if (fabs(z - z1) <= 1e-15 * fmax(1.0, fabs(z))) break;
    }
    x[i] = z;
    x[n - 1 - i] = -z;
    w[i] = 2.0 / (pp * pp);
    w[n - 1 - i] = w[i];
  }
}

}  // namespace

const double* gauss_hermite_table(dcgp_ctx* ctx) {
  auto it = ctx->ws.find("gh20");
  if (it != ctx->ws.end()) return (const double*)it->second.first;
  double* d = (double*)ws_get(ctx, "gh20", 40 * sizeof(double));
  if (!d) return nullptr;
  double h[40];
  hermgauss_host(20, h, h + 20);
  // ascending node order like numpy (irrelevant for the sum, kept for readability of dumps)
  for (int i = 0; i < 10; ++i) {
    double tx = h[i]; h[i] = h[19 - i]; h[19 - i] = tx;
    double tw = h[20 + i]; h[20 + i] = h[39 - i]; h[39 - i] = tw;
  }
  hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream);
  hipStreamSynchronize(ctx->stream);
  return d;
}

// everything of a TailArgs that does not depend on the rows: Gauss-Hermite table, arrival counters
int elbo_tail_prepare(dcgp_ctx* ctx, TailArgs* t) {
  t->gh = gauss_hermite_table(ctx);
  if (!t->gh) return DCGP_ERR_ALLOC;
  auto it = ctx->ws.find("elbo_ticket");
  unsigned* ticket = it != ctx->ws.end() ? (unsigned*)it->second.first : nullptr;
  if (!ticket) {
    ticket = (unsigned*)ws_get(ctx, "elbo_ticket", 256);
    if (!ticket) return DCGP_ERR_ALLOC;
    HIP_TRY(ctx, hipMemsetAsync(ticket, 0, 256, ctx->stream));   // (a poisoned workspace must not leave a wrong count)
  }
  t->ticket = ticket;
  return DCGP_OK;
}

int elbo_tail(dcgp_ctx* ctx, const double* mu, const double* var, const int32_t* y, int n_rows, int n_labels, int K, double eps,
              double* ve_rows, double inv_s, double* scal, const ElboFinish& fin, const KlTail* kl) {
  TailArgs t;
  DCGP_TRY(elbo_tail_prepare(ctx, &t));
  t.mu = mu; t.var = var; t.y = y; t.n_rows = n_rows; t.n_labels = n_labels; t.K = K; t.eps = eps; t.ve = ve_rows;
  t.inv_s = inv_s; t.scal = scal; t.fin = fin;
  ScopedTimer tm(ctx, "elbo_tail");
  KlTail k;   // nl == 0: the KL pieces are already in scal
  if (kl) k = *kl;
  hipLaunchKernelGGL(elbo_tail_kernel, dim3((unsigned)((n_rows + 7) / 8 + k.nl)), dim3(256), 0, ctx->stream, t, k);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

int reduce_sum(dcgp_ctx* ctx, const double* in, long n, double scale, double* out) {
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, ctx->stream, in, n, scale, out);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

int reduce_sum_multi(dcgp_ctx* ctx, const ReduceJobs& jobs, int count) {
  if (count <= 0) return DCGP_OK;
  if (count > REDUCE_JOBS_MAX) return ctx_fail(ctx, DCGP_ERR_ARG, "reduce_sum_multi: at most %d sums per launch", REDUCE_JOBS_MAX);
  long nmax = 0;
  for (int k = 0; k < count; ++k) nmax = jobs.n[k] > nmax ? jobs.n[k] : nmax;
  // eight elements per thread (two rounds of four loads in flight) before another block pays: a block is bound by its memory latencies -- 45 loads
  // per thread one after the other were ~100 us for a sum of 46 080 values
  int nb = (int)((nmax + 8191) / 8192);
  nb = nb < 1 ? 1 : (nb > 256 ? 256 : nb);
  double* part = nullptr;
  unsigned* ticket = nullptr;
  if (nb > 1) {   // partial sums and arrival counters of THIS stream (the reverse pass sums on three)
    char tag[64];
    snprintf(tag, sizeof tag, "_%p", (void*)ctx->stream);
    part = (double*)ws_get(ctx, std::string("red_part") + tag, (size_t)REDUCE_JOBS_MAX * 256 * sizeof(double));
    const std::string tk = std::string("red_ticket") + tag;
    auto it = ctx->ws.find(tk);
    ticket = it != ctx->ws.end() ? (unsigned*)it->second.first : nullptr;
    if (!ticket) {
      ticket = (unsigned*)ws_get(ctx, tk, 256);
      if (ticket) HIP_TRY(ctx, hipMemsetAsync(ticket, 0, 256, ctx->stream));
    }
    if (!part || !ticket) return DCGP_ERR_ALLOC;
  }
  hipLaunchKernelGGL(reduce_sum_multi_kernel, dim3(nb, count), dim3(1024), 0, ctx->stream, jobs, part, ticket, nb);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

int varexp_rows(dcgp_ctx* ctx, const double* mu, const double* var, const int32_t* y, int n_rows, int n_labels, int K,
                double eps, double* out_rows, int predict) {
  const double* gh = gauss_hermite_table(ctx);
  if (!gh) return DCGP_ERR_ALLOC;
  long slots = predict ? (long)n_rows * K : n_rows;
  hipLaunchKernelGGL(varexp_kernel, dim3((unsigned)((slots + 7) / 8)), dim3(256), 0, ctx->stream, mu, var, y, n_rows,
                     n_labels, K, eps, gh, out_rows, predict);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

// The conditional with the second triangular solve folded into the SMALL operands.  The reference computes
//     A1 = inv(L) Kuf ;  A = inv(L)^T A1 ;  mean = A^T q_mu ;  var += sum_m (Lq_r^T A)^2      (conditionals.py:31-65)
// Since Lq_r^T inv(L)^T A1 = (inv(L) Lq_r)^T A1 and q_mu^T inv(L)^T A1 = (inv(L) q_mu)^T A1, the M x (P*N) product
// with inv(L)^T is replaced by two M x M-sized products (cond_prep: G_r = inv(L) Lq_r, still lower triangular, and
// alpha = inv(L) q_mu) that run next to the factorisation on the side stream.  Whitened case: G = Lq, alpha = q_mu.
int cond_prep(dcgp_ctx* ctx, GpMats& g, int white, bool have_qsqrt) {
  const int Mp = g.Mp, R = g.R;
  g.prep_sums_valid = false;
  if (white) return DCGP_OK;   // G / alpha alias Lq / qmu (set by the owner of g)
  ScopedTimer t(ctx, "gemm_prep");
  // The products' epilogues also leave the column sums of squares of G_r and alpha (per row block): ||inv(L) Lq_r||_F^2 and
  // ||inv(L) q_mu||^2 are the KL's trace and Mahalanobis terms where the KL prior is K itself (the head), and kl_layer used to
  // run both products a second time just for them (gemm_kl: 280 us at M = 1024).
  const int BM = gemm_row_block(Mp, Mp, R), nrb = (Mp + BM - 1) / BM;
  const int BMa = gemm_row_block(Mp, g.Rp, 1), nrba = (Mp + BMa - 1) / BMa;
  // named by the model / bank tag of the step (forward_all's ws_tag; empty for the operator entry points) AND the operand's address:
  // dcgp_model_destroy frees everything carrying its tag, so a later model whose G lands at the same address gets a buffer of its own
  char nm[128];
  snprintf(nm, sizeof nm, "prep_tp%s@%p", ctx->ws_tag.c_str(), (void*)g.G);
  g.prep_tp_count = (long)R * nrb * Mp;
  g.prep_tp = have_qsqrt ? (double*)ws_get(ctx, nm, (size_t)g.prep_tp_count * sizeof(double)) : nullptr;
  snprintf(nm, sizeof nm, "prep_ap%s@%p", ctx->ws_tag.c_str(), (void*)g.alpha);
  g.prep_ap_count = (long)nrba * g.Rp;
  g.prep_ap = (double*)ws_get(ctx, nm, (size_t)g.prep_ap_count * sizeof(double));
  if ((have_qsqrt && !g.prep_tp) || !g.prep_ap) return DCGP_ERR_ALLOC;
  if (have_qsqrt) {
    GemmArgs a;   // G_r = inv(L) Lq_r : lower x lower
    a.Wt = g.LinvT; a.ldw = Mp;
    a.B = g.Lq; a.ldb = Mp; a.bBatch = (long)Mp * Mp; a.nB = R;
    a.C = g.G; a.ldc = Mp; a.cBatch = (long)Mp * Mp;
    a.colsq = g.prep_tp; a.sBatch = (long)nrb * Mp; a.sRowBlk = Mp;
    a.Mi = Mp; a.Mk = Mp; a.Kc = Mp; a.tri = 1; a.b_lower = 1;
    DCGP_TRY(gemm_tn(ctx, a, nullptr));
  }
  GemmArgs b;     // alpha = inv(L) q_mu
  b.Wt = g.LinvT; b.ldw = Mp;
  b.B = g.qmu; b.ldb = g.Rp;
  b.C = g.alpha; b.ldc = g.Rp;
  b.colsq = g.prep_ap; b.sRowBlk = g.Rp;
  b.Mi = Mp; b.Mk = Mp; b.Kc = g.Rp; b.tri = 1;
  DCGP_TRY(gemm_tn(ctx, b, nullptr));
  g.prep_sums_valid = have_qsqrt;
  return DCGP_OK;
}

int cond_core(dcgp_ctx* ctx, const GpMats& g, const double* B, long ldb, int Kc, int white, bool have_qsqrt,
              const char* pfx, CondScratch* out, hipEvent_t prep_done, bool head) {
  const int Mp = g.Mp, R = g.R;
  const int BM1 = gemm_row_block(Mp, Kc, 1), BM3 = gemm_row_block(Mp, Kc, R);
  const int nrb = (Mp + BM1 - 1) / BM1, nrb3 = (Mp + BM3 - 1) / BM3;
  std::string p(pfx);
  CondScratch sc;
  sc.ldb = ldb;
  sc.nrb1 = nrb; sc.nrb3 = nrb3;
  sc.A1 = (double*)ws_get(ctx, p + "A1", (size_t)Mp * ldb * sizeof(double));
  sc.s1p = (double*)ws_get(ctx, p + "s1p", (size_t)nrb * ldb * sizeof(double));
  sc.s2p = have_qsqrt ? (double*)ws_get(ctx, p + "s2p", (size_t)R * nrb3 * ldb * sizeof(double)) : nullptr;
  sc.mu = (double*)ws_get(ctx, p + "mu", (size_t)g.Rp * ldb * sizeof(double));
  if (!sc.A1 || !sc.s1p || !sc.mu || (have_qsqrt && !sc.s2p)) return DCGP_ERR_ALLOC;
  {
    ScopedTimer t(ctx, head ? "gemm_head_s1" : "gemm_cond_s1");   // A1 = inv(L) Kuf, s1 = sum_m A1^2
    GemmArgs a;
    a.Wt = g.LinvT; a.ldw = Mp;
    a.B = B; a.ldb = (int)ldb;
    a.C = sc.A1; a.ldc = (int)ldb;
    a.colsq = sc.s1p; a.sRowBlk = ldb;
    a.Mi = Mp; a.Mk = Mp; a.Kc = Kc; a.tri = 1;
    DCGP_TRY(gemm_tn(ctx, a, nullptr));
  }
  if (prep_done) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, prep_done, 0));   // G / alpha come from the side stream
  if (have_qsqrt) {
    ScopedTimer t(ctx, head ? "gemm_head_s3" : "gemm_cond_s3");   // T_r = G_r^T A1 (upper-triangular product), s2 = sum_m T_r^2, never stored
    GemmArgs a;
    a.Wt = g.G; a.ldw = Mp; a.wBatch = (long)Mp * Mp; a.nW = R;
    a.B = sc.A1; a.ldb = (int)ldb;
    a.colsq = sc.s2p; a.sBatch = (long)nrb3 * ldb; a.sRowBlk = ldb;
    a.Mi = Mp; a.Mk = Mp; a.Kc = Kc; a.tri = 2;
    DCGP_TRY(gemm_tn(ctx, a, nullptr));
  }
  {
    // mu[r][j] = sum_k alpha[k][r] A1[k][j]  (conditionals.py:50): a 16-row dense product on the same kernel
    ScopedTimer t(ctx, head ? "head_mean" : "cond_mean");
    GemmArgs a;
    a.Wt = g.alpha; a.ldw = g.Rp;
    a.B = sc.A1; a.ldb = (int)ldb;
    a.C = sc.mu; a.ldc = (int)ldb;
    a.Mi = g.Rp; a.Mk = Mp; a.Kc = Kc; a.tri = 0;
    DCGP_TRY(gemm_tn(ctx, a, nullptr));
  }
  *out = sc;
  return DCGP_OK;
}

int finalize_layer(dcgp_ctx* ctx, const FinalizeArgs& a) {
  if (a.Kc <= 0) return DCGP_OK;
  ScopedTimer t(ctx, "finalize");
  size_t lds = (size_t)2 * a.R * 65 * sizeof(double);
  if (lds > 150 * 1024) return ctx_fail(ctx, DCGP_ERR_ARG, "finalize: R=%d too large", a.R);
  hipLaunchKernelGGL(finalize_kernel, dim3((a.Kc + 63) / 64), dim3(256), lds, ctx->stream, a);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

int kl_layer(dcgp_ctx* ctx, const GpMats& g, const double* Lp, const double* LpinvT, int white, const char* pfx,
             double* kl4) {
  const int Mp = g.Mp, R = g.R;
  double *tp = nullptr, *ap = nullptr;
  long tp_count = 0, ap_count = 0;
  const double* sums = (g.klp && g.klp_valid && LpinvT == g.LinvT) ? g.klp : ((g.klpp && g.klpp_valid && LpinvT == g.LpinvT) ? g.klpp : nullptr);
  if (!white && sums) {
    // prep_solve left the sums of squares of inv(Lp) Lq_r and inv(Lp) q_mu (Lp = L for a layer without a prior of its own)
    const int ns = g.kl_ns > 0 ? g.kl_ns : Mp / 16;
    tp = const_cast<double*>(sums); tp_count = (long)R * ns;
    ap = tp + (long)R * ns; ap_count = g.kl_ns > 0 ? g.kl_nsa : 1;
  } else if (!white && g.prep_sums_valid && LpinvT == g.LinvT) {
    tp = g.prep_tp; tp_count = g.prep_tp_count;   // cond_prep's products left them (same launch stream: ordered behind them)
    ap = g.prep_ap; ap_count = g.prep_ap_count;
  } else if (!white) {
    const int BM = gemm_row_block(Mp, Mp, R), nrb = (Mp + BM - 1) / BM;
    const int BMa = gemm_row_block(Mp, g.Rp, 1), nrba = (Mp + BMa - 1) / BMa;
    tp_count = (long)R * nrb * Mp;
    ap_count = (long)nrba * g.Rp;
    tp = (double*)ws_get(ctx, std::string(pfx) + "kl_tp" + ctx->ws_tag, (size_t)tp_count * sizeof(double));
    ap = (double*)ws_get(ctx, std::string(pfx) + "kl_ap" + ctx->ws_tag, (size_t)ap_count * sizeof(double));
    if (!tp || !ap) return DCGP_ERR_ALLOC;
    ScopedTimer t(ctx, "gemm_kl");
    GemmArgs a;   // || inv(Lp) Lq_r ||_F^2 for every r
    a.Wt = LpinvT; a.ldw = Mp;
    a.B = g.Lq; a.ldb = Mp; a.bBatch = (long)Mp * Mp; a.nB = R;
    a.colsq = tp; a.sBatch = (long)nrb * Mp; a.sRowBlk = Mp;
    a.Mi = Mp; a.Mk = Mp; a.Kc = Mp; a.tri = 1; a.b_lower = 1;
    DCGP_TRY(gemm_tn(ctx, a, nullptr));
    GemmArgs b;   // || inv(Lp) q_mu ||_F^2 (padded columns of qmu are zero)
    b.Wt = LpinvT; b.ldw = Mp;
    b.B = g.qmu; b.ldb = g.Rp;
    b.colsq = ap; b.sRowBlk = g.Rp;
    b.Mi = Mp; b.Mk = Mp; b.Kc = g.Rp; b.tri = 1;
    DCGP_TRY(gemm_tn(ctx, b, nullptr));
  }
  hipLaunchKernelGGL(kl_small_kernel, dim3(1), dim3(1024), 0, ctx->stream, Lp, g.Lq, g.qmu, g.Rp, ap, ap_count, tp,
                     tp_count, g.M, Mp, R, white, kl4);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

int reparam_async(dcgp_ctx* ctx, const double* mean, const double* var, const double* z, size_t n, double jitter,
                  double* out) {
  if (n == 0) return DCGP_OK;
  hipLaunchKernelGGL(reparam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, mean, var, z, n,
                     jitter, out);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

extern "C" int dcgp_reparam(dcgp_ctx* ctx, const double* mean, const double* var, const double* z, size_t n,
                            double jitter, double* out) {
  if (!ctx || !mean || !var || !z || !out) return ctx ? ctx_fail(ctx, DCGP_ERR_ARG, "reparam: null pointer") : DCGP_ERR_ARG;
  if (n == 0) return DCGP_OK;
  hipLaunchKernelGGL(reparam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, mean, var, z, n,
                     jitter, out);
  LAUNCH_CHECK(ctx);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}

extern "C" int dcgp_robustmax_varexp(dcgp_ctx* ctx, const double* mu, const double* var, const int32_t* y, int n, int K,
                                     double eps, double* out_n) {
  if (!ctx || !mu || !var || !y || !out_n || n <= 0 || K < 2)
    return ctx ? ctx_fail(ctx, DCGP_ERR_ARG, "robustmax_varexp: bad args") : DCGP_ERR_ARG;
  DCGP_TRY(varexp_rows(ctx, mu, var, y, n, n, K, eps, out_n, 0));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}

extern "C" int dcgp_robustmax_predict(dcgp_ctx* ctx, const double* mu, const double* var, int n, int K, double eps,
                                      double* out_p) {
  if (!ctx || !mu || !var || !out_p || n <= 0 || K < 2)
    return ctx ? ctx_fail(ctx, DCGP_ERR_ARG, "robustmax_predict: bad args") : DCGP_ERR_ARG;
  DCGP_TRY(varexp_rows(ctx, mu, var, nullptr, n, 1, K, eps, out_p, 1));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}
