// Internal: device-resident state of one GP layer and the shared forward building blocks.
#pragma once
#include "common.h"
#include "rng.h"

// Geometry of a patch view (FullView, conv_gp/views.py:20-30,56-68)
struct ViewGeom {
  int H = 0, W = 0, C = 0, f = 0, s = 0, Ho = 0, Wo = 0, P = 0, L = 0;
  void set(int H_, int W_, int C_, int f_, int s_) {
    H = H_; W = W_; C = C_; f = f_; s = s_;
    Ho = (H - f) / s + 1; Wo = (W - f) / s + 1; P = Ho * Wo; L = f * f * C;
  }
};

// Padded M x M operands of one layer ([Mp x Mp], ld = Mp) living in device memory.
struct GpMats {
  int M = 0, Mp = 0, R = 0, Rp = 0;   // Rp = R rounded up to 16 (column padding of qmu)
  double* K = nullptr;      // Kuu (live Z) -> overwritten by its Cholesky factor L
  double* Linv = nullptr;   // inv(L)
  double* LinvT = nullptr;  // inv(L)^T
  double* Kp = nullptr;     // prior Kuu(Z0) -> its factor (conv layers, non-white); may alias K
  double* Lpinv = nullptr;
  double* LpinvT = nullptr;
  double* Lq = nullptr;     // [R][Mp][Mp] lower-masked q_sqrt, zero padded
  double* qmu = nullptr;    // [Mp][Rp], zero padded rows and columns
  // derived by cond_prep after the factorisation (alias Lq / qmu in the whitened case):
  double* G = nullptr;      // [R][Mp][Mp]  G_r = inv(L) Lq_r   (lower triangular)
  double* alpha = nullptr;  // [Mp][Rp]     alpha = inv(L) q_mu
  double* klp = nullptr;    // [(R + 1)][Mp / 16] sums of squares of the 16-column strips of G_r and (row R, entry 0) of alpha,
                            // left by prep_solve: the KL's trace and Mahalanobis terms when its prior factor is L itself
  bool klp_valid = false;   // set by prep_solve_all for the launch that filled klp
  double* klpp = nullptr;   // the same sums with the PRIOR factor inv(Lp) in place of inv(L) (layers with a prior Kuu(Z0)): nothing
                            // but the sums is kept of those products
  bool klpp_valid = false;
  int kl_ns = 0, kl_nsa = 0;   // layout of klp / klpp: 0 = prep_solve's 16-column strips ([(R + 1)][Mp / 16], one alpha entry); else [(R + 1)][kl_ns] with
                               // kl_nsa alpha entries (the right-hand sides rode the factorisation chain: chain_rhs_slots)
  // M > 256 (cond_prep's generic GEMMs): the same column sums of squares, per row block, from the G / alpha products' epilogues
  double* prep_tp = nullptr; long prep_tp_count = 0;   // [R][nrb][Mp] sums of squares of G
  double* prep_ap = nullptr; long prep_ap_count = 0;   // [nrb][Rp] of alpha
  bool prep_sums_valid = false;
};

// parameter-only preparation of all layers in one launch (prep.hip)
struct PrepLayerArgs {
  const double *Z = nullptr, *Z0 = nullptr, *q_sqrt = nullptr, *q_mu = nullptr;
  double *K = nullptr, *Kp = nullptr, *ZT = nullptr, *zn = nullptr, *Lq = nullptr, *qmu = nullptr;
  int M = 0, Mp = 0, L = 0, Lp = 0, R = 0, Rp = 0;
  BaseKernel bk; double jitter = 0.0;
  const double* in_scale = nullptr;   // [L] or nullptr: Z is read as Z * in_scale (ARD lengthscales)
  double* ZS = nullptr; int Lz = 0;   // the sweeps' scaled operand (sweep_dev.h); nullptr: not an RBF layer
};
struct PrepArgs {
  int nl = 0;
  PrepLayerArgs l[8];
};
// task_mask: bit t = task t of prep.hip (0 / 1: Kuu of the live / prior Z, 2: Z^T and |z|^2, 3: masked q_sqrt, 4: padded q_mu, 5: the sweeps' scaled Z)
constexpr unsigned kPrepSweepTasks = (1u << 2) | (1u << 5);   // what a patch sweep reads
int prepare_all(dcgp_ctx* ctx, const PrepArgs& a, unsigned task_mask = ~0u);

// A = inv(L) Kuf etc. on a k-major Kuf matrix B [Mp x ldb] with Kc columns.
// Produces partial column sums s1p [nrb1][ldb], s2p [R][nrb3][ldb], and mu [R][ldb].
struct CondScratch {
  double *A1 = nullptr, *s1p = nullptr, *s2p = nullptr, *mu = nullptr;
  int nrb1 = 0, nrb3 = 0;
  long ldb = 0;
};
// G and alpha of a layer (two small GEMMs on the current stream); must run after the factorisation of g.K
int cond_prep(dcgp_ctx* ctx, GpMats& g, int white, bool have_qsqrt);
int cond_core(dcgp_ctx* ctx, const GpMats& g, const double* B, long ldb, int Kc, int white, bool have_qsqrt,
              const char* ws_prefix, CondScratch* out, hipEvent_t prep_done = nullptr, bool head = false);   // head: timer labels only

// head_cond.hip: the whole conditional of a few-column problem in one launch (M <= 256): mean / var [Kc][R]
bool head_cond_fused_ok(const GpMats& g);
int head_cond_fused(dcgp_ctx* ctx, const GpMats& g, const double* B, long ldb, int Kc, bool have_qsqrt, const double* kd,
                    double* out_mean, double* out_var, int kd_n = 1, double kd_scale = 1.0, double* A1_out = nullptr, long lda1 = 0);   // Knn[j] = kd_scale * sum_{i < kd_n} kd[j * kd_n + i]

// head_cond.hip: G / alpha of every layer in one launch; done[i] = false where layer i still needs cond_prep
int prep_solve_all(dcgp_ctx* ctx, GpMats* const* gs, const int* white, const bool* have_qsqrt, int nl, bool* done, const bool* skip = nullptr);

struct FinalizeArgs {
  const double* s1p = nullptr; int nrb1 = 0;
  const double* s2p = nullptr; int nrb3 = 0;    // nullptr -> no q_sqrt term
  const double* mu = nullptr;
  long ldk = 0;                  // leading dimension (padded column count) of the above
  int Kc = 0, R = 0;
  long col0 = 0;                 // first column of a chunk: outputs, noise and the identity mean are indexed by col0 + j
  double knn_scalar = 0.0; const double* knn_vec = nullptr;   // Knn per column (vector wins if set)
  // output: element (j, r) of replica s at  s*rep_stride + j*R + r
  int rep = 1; long rep_stride = 0;
  const double* z = nullptr;     // same indexing as the output; nullptr + want sample -> device RNG
  uint64_t seed = 0; uint32_t stream_id = 0;
  RngMap rmap;                    // device RNG: the element's counter in the un-sharded batch
  double jitter = 0.0;
  double *out_sample = nullptr, *out_mean = nullptr, *out_var = nullptr;
  // Conv2dMean (conv_gp/mean_functions.py:28-41): adds the centre pixel of channel 0 to map r == 0
  const double* X = nullptr; int idm = 0; int n_mod = 0; int H = 0, W = 0, C = 0, f = 0, s = 0, Wo = 0, P = 0;
};
int finalize_layer(dcgp_ctx* ctx, const FinalizeArgs& a);

// conv_fused.hip: the whole conv layer (patch sweep, both triangular products, mean, var, sample) of a column strip in one
// workgroup; K_uf and A1 never leave the chip unless the training step asks for them
struct ConvFusedArgs {
  const double* X = nullptr; int n_mod = 0;               // [n_mod, H, W, C]; image of row n is X[n % n_mod]
  int H = 0, W = 0, C = 0, f = 0, s = 0, Wo = 0, P = 0, L = 0, Lp = 0, HWC = 0;
  const double* ZT = nullptr; const double* zn = nullptr; int M = 0, Mp = 0;
  const double* ZS = nullptr; int Lz = 0; double csq = 1.0;   // RBF: the sweep's scaled operand [Lz][Mp] and image scale sqrt(c) (sweep_dev.h)
  BaseKernel bk;
  const double* LinvT = nullptr;                           // [Mp][Mp]
  const double* G = nullptr;                               // [R][Mp][Mp] or nullptr (no q_sqrt term)
  const double* alpha = nullptr; int R = 0, Rp = 0;        // [Mp][Rp]
  int Kc = 0;                                              // columns = rows * P
  double knn = 0.0;
  int rep = 1; long rep_stride = 0;
  const double* z = nullptr; uint64_t seed = 0; uint32_t stream_id = 0; double jitter = 0.0;
  RngMap rmap;                                             // device RNG: the element's counter in the un-sharded batch
  double *out_sample = nullptr, *out_mean = nullptr, *out_var = nullptr;
  int idm = 0;
  double *Kuf_out = nullptr, *A1_out = nullptr; long ldk = 0;   // training step: k-major [Mp][ldk] copies for the reverse pass
  int lds_main = 0, lds_img = 0;                           // set by the launcher
  float inv_HWC = 0, inv_nmod = 0, inv_P = 0, inv_Wo = 0, inv_R = 0;   // set by the launcher: reciprocals of the kernel's divisors (fdiv)
  int split_first = 1 << 30, split_q = 1;                  // set by the launcher: strips >= split_first are shared by split_q workgroups (outputs r = q, q + split_q, ...)
  long long* trace = nullptr;                              // debugging aid: phase timestamps (dcgp_debug_set_fused_trace)
  // set by the launcher: a persistent launch -- one workgroup per slot of the chip, each walking the strips blockIdx, blockIdx + grid, ... < n_strips
  int persist = 0, n_strips = 0;
  int stagger = 0;                                         // the second workgroup to arrive on a CU starts this many 100 MHz ticks late
  int* dyn = nullptr;                                      // [2] {strips dealt beyond the first of every workgroup, workgroups that have left}: zero between launches
  int* cu_slots = nullptr;                                 // [1024] arrival counters per CU (zero between launches: every workgroup gives its count back)
  // set by the launcher: prologues ahead (conv_fused.hip) -- pre_n strips from pre_first on get phases 0 - 2 from the spare workgroups of the partial FIRST
  // round, A1 handed over through pre_buf ([pre_n][pre_stride]: the strip's LDS image, then the partial sums of A1^2) behind pre_flag[slot] == pre_epoch
  // pre_sq > 1 (a launch of few strips, all of them handed over: pre_first = 0, pre_n = n_strips): a handed-over strip is taken up by pre_sq items, part q
  // running the outputs r = q, q + pre_sq, ... of the second product
  int no_rows = 0;                                         // set by the launcher (ctx option sweep_no_rows): 5 x 5 x 10 patches on the generic in-kernel sweep (A/B)
  int pre_n = 0, pre_first = 0, pre_sq = 1; long pre_stride = 0;
  double* pre_buf = nullptr; unsigned* pre_flag = nullptr; unsigned pre_epoch = 0;
};
// the reverse pass of the same strip (conv_bwd_fused.hip): dK_uf = inv(L)^T [sum_r (S_r A1) o (2 gv_r) + alpha gm^T - 2 A1 o gvs]
struct ConvBwdArgs {
  const double* A1 = nullptr; long ld = 0; int Kc = 0;   // [Mp][ld], k-major (left by the forward's training form)
  const double* S = nullptr;                              // [R][Mp][Mp]  S_r = G_r G_r^T, zero beyond M
  const double* alpha = nullptr; int Rp = 0;              // [Mp][Rp]
  const double* Linv = nullptr;                           // [Mp][Mp] inv(L), row-major
  const double *gv = nullptr, *gm = nullptr, *gvs = nullptr;   // d var [Kc][R], d mean [Kc][R], row sums of d var [Kc]
  int M = 0, Mp = 0, R = 0;
  double* dKuf = nullptr;                                 // [Mp][ld]
};
bool conv_bwd_fused_ok(const dcgp_ctx* ctx, const ConvBwdArgs& a);
int conv_bwd_fused(dcgp_ctx* ctx, const ConvBwdArgs& a);
bool conv_fused_ok(const dcgp_ctx* ctx, const ConvFusedArgs& a);
int conv_fused(dcgp_ctx* ctx, const ConvFusedArgs& a);

// KL pieces of one layer -> kl4[0..3] = {mahalanobis, logdet_q, logdet_p, trace} (device)
int kl_layer(dcgp_ctx* ctx, const GpMats& g, const double* Lp, const double* LpinvT, int white, const char* ws_prefix,
             double* kl4);

// the ELBO assembly the tail kernel performs after the data term (nl == 0: data term only -> scal[0])
struct ElboFinish {
  int nl = 0;
  int M[8], R[8], white[8];
  double scale = 1.0;
  const int* info[16];   // per factor group: potrf status words (0 or the 1-based failing column)
  int ninfo[16];
  int ngroups = 0;
  double* host_out = nullptr;   // pinned host slot (device-visible address): the four result words are also written there,
                                // so no copy command follows the launch
  double host_seq = 0.0;        // written to host_out[4] behind them (system-scope release): the word the host polls for (ticket + 1)
};
// RobustMax expectations of every row -> ve_rows, scal[0] = inv_s * their sum, and (fin.nl > 0) scal[40..43] = ELBO, data term,
// KL, potrf status from the KL pieces at scal[4 + 4 l ..]: one launch (cond.hip)
struct TailArgs;   // tail_dev.h
// The KL pieces of the layers inside the tail launch (one extra workgroup per layer): everything they read is parameter-only
// state the chain left behind -- the strip sums of prep_solve and the factors' diagonals.
struct KlTailLayer {
  const double* Lfac = nullptr; long ldf = 0;   // the KL prior's Cholesky factor (diagonal read: log-determinant)
  const double* Lq = nullptr;                   // [R][Mp][Mp] (diagonal read)
  const double* sums = nullptr;                 // [(R + 1)][ns] partial sums of squares (GpMats::klp / klpp): rows r < R all ns, row R the first nsa
  int ns = 0, nsa = 0;                          // 0: prep_solve's layout (ns = Mp / 16 strips, nsa = 1)
  int M = 0, Mp = 0, R = 0;
};
struct KlTail { int nl = 0; KlTailLayer l[8]; };
int elbo_tail_prepare(dcgp_ctx* ctx, TailArgs* t);   // Gauss-Hermite table and arrival counters of a TailArgs
int elbo_tail(dcgp_ctx* ctx, const double* mu, const double* var, const int32_t* y, int n_rows, int n_labels, int K, double eps,
              double* ve_rows, double inv_s, double* scal, const ElboFinish& fin, const KlTail* kl = nullptr);
int varexp_rows(dcgp_ctx* ctx, const double* mu, const double* var, const int32_t* y, int n_rows, int n_labels, int K,
                double eps, double* out_rows, int predict);
const double* gauss_hermite_table(dcgp_ctx* ctx);   // [40]: 20 nodes then 20 weights (device)

// deterministic single-block sum of n doubles, scaled: out[0] = scale * sum
int reduce_sum(dcgp_ctx* ctx, const double* in, long n, double scale, double* out);
constexpr int REDUCE_JOBS_MAX = 12;
struct ReduceJobs { const double* in[REDUCE_JOBS_MAX]; long n[REDUCE_JOBS_MAX]; double scale[REDUCE_JOBS_MAX]; double* out[REDUCE_JOBS_MAX]; };
int reduce_sum_multi(dcgp_ctx* ctx, const ReduceJobs& jobs, int count);   // out[k][0] = scale[k] * sum(in[k][0..n[k])), one launch
