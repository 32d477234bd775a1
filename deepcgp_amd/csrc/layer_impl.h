// Internal (header-only): device-resident layer state and the per-layer forward building blocks
// shared by model.hip (model-level path) and ops.hip (single-operator C-ABI entry points).
#pragma once
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "layer.h"

int additive_kdiag_async(dcgp_ctx* ctx, int N, int P, double variance, const double* w, double* out_N);
int reparam_async(dcgp_ctx* ctx, const double* mean, const double* var, const double* z, size_t n, double jitter,
                  double* out);
int allreduce_sum_f64_async(dcgp_ctx* ctx, double* buf_dev, int n);

struct LayerState {
  dcgp_ctx* ctx = nullptr;
  bool is_head = false;
  bool a1h_ready = false;     // head: the forward pass of a training step left A1 = inv(L) Kzx in "<pfx>g_A1h" (head_forward, keep_k)
  bool kfull_ready = false;   // head: the forward pass of a training step left every patch response in "<pfx>g_Kfull" (head_forward, keep_k)
  ViewGeom v;
  int M = 0, Mp = 0, R = 0, Lp = 0;
  int white = 0, identity_mean = 0, kernel_type = 0;
  double variance = 1.0, ls = 1.0;
  int base_type = 0;             // 0: RBF(variance, ls); 1: ArcCosine order 0 (variance, weight variance = acos_w, bias variance = acos_b)
  double acos_w = 1.0, acos_b = 1.0;
  BaseKernel base() const {
    BaseKernel b;
    b.type = base_type; b.variance = variance;
    if (base_type == 0) { b.p1 = 1.0 / (ls * ls); b.p2 = 0.0; } else { b.p1 = acos_w; b.p2 = acos_b; }
    return b;
  }
  bool has_qsqrt = true;
  // parameters in the caller's layout (device)
  double *Z = nullptr, *Z0 = nullptr, *q_mu = nullptr, *q_sqrt = nullptr, *w = nullptr;
  double* in_scale = nullptr;   // [L] 1 / ARD lengthscale per input dimension, or nullptr (set_param "ard_lengthscales"; single-patch head only)
  double* ard = nullptr;        // [L] the ARD lengthscales themselves (optimiser state; in_scale is refreshed from it)
  double *gard = nullptr, *aard[2] = {};   // their gradient (inside the gradient block) and Adam moments
  // derived every step.  Two banks of them: a step's parameter-only chain writes the bank of its parity, so that the chain of
  // step i + 1 can run while the data path of step i still reads the other bank (dcgp_elbo_forward_enqueue).  g / ZT / zn are the
  // bank in use (use_bank); bank 1 is allocated on first use.
  GpMats g;
  double *ZT = nullptr, *zn = nullptr, *ZS = nullptr;   // ZS: the sweeps' scaled operand [Lq][Mp] (sweep_dev.h)
  int Lz = 0;   // rows of ZS: patch length + the two norm slots, padded to a multiple of 4
  GpMats gbank[2];
  double *ZTb[2] = {}, *znb[2] = {}, *ZSb[2] = {};
  bool need_prior_ = true;
  // gradients of the ELBO with respect to the (constrained) parameter values, caller's layouts (grad.hip); allocated on
  // first use.  gscal = {d variance, d p1, d p2} (p1 = lengthscale | ArcCosine weight variance, p2 = ArcCosine bias variance);
  // gslots = per-contribution partial sums of those three (3 x 16).
  double *gZ = nullptr, *gq_mu = nullptr, *gq_sqrt = nullptr, *gw = nullptr, *gscal = nullptr, *gslots = nullptr;
  // Adam moments, same layouts (allocated zeroed on the first optimiser step); hyp = {variance, p1, p2} device copy
  double *aZ[2] = {}, *aq_mu[2] = {}, *aq_sqrt[2] = {}, *aw[2] = {}, *ahyp[2] = {}, *hyp = nullptr;
  // optimiser: parameters excluded from dcgp_model_adam_step / dcgp_model_sgd_step (bit 0 Z, 1 q_mu, 2 q_sqrt, 3 w, 4 hyper-parameters)
  unsigned frozen = 0;
  std::shared_ptr<void> natgrad_state;   // scratch of dcgp_model_natgrad_step (natgrad.hip), created on first use
  std::vector<void*> owned;

  ~LayerState() {
    for (void* p : owned) hipFree(p);
  }
  double* dalloc(size_t n_doubles) {
    void* p = nullptr;
    if (hipMalloc(&p, (n_doubles ? n_doubles : 2) * sizeof(double)) != hipSuccess) return nullptr;
    if (dcgp_poison()) { hipMemset(p, 0xFF, (n_doubles ? n_doubles : 2) * sizeof(double)); hipDeviceSynchronize(); }   // debugging aid, see ws_get (ctx.hip)
    owned.push_back(p);
    return (double*)p;
  }
  int init(dcgp_ctx* c, bool head, int H, int W, int C, int f, int s, int M_, int R_, int white_, int idm, int ktype,
           double var, double ls_, bool need_prior = true) {
    ctx = c; is_head = head;
    if (H <= 0 || W <= 0 || C <= 0 || f <= 0 || s <= 0 || f > H || f > W || M_ <= 0 || R_ <= 0)
      return ctx_fail(c, DCGP_ERR_ARG, "layer: bad geometry H=%d W=%d C=%d f=%d s=%d M=%d R=%d", H, W, C, f, s, M_, R_);
    if (!(var > 0.0) || !(ls_ > 0.0)) return ctx_fail(c, DCGP_ERR_ARG, "layer: variance and lengthscale must be > 0");
    v.set(H, W, C, f, s);
    M = M_; R = R_; white = white_; identity_mean = idm; kernel_type = ktype; variance = var; ls = ls_;
    Mp = round_up(M, 16);
    Lp = round_up(v.L, 4);
    Lz = sweep_lq(v.L);
    Z = dalloc((size_t)M * v.L);
    Z0 = head ? nullptr : dalloc((size_t)M * v.L);
    q_mu = dalloc((size_t)M * R);
    q_sqrt = dalloc((size_t)R * M * M);
    w = head ? dalloc(v.P) : nullptr;
    need_prior_ = need_prior;
    alloc_bank(0);
    use_bank(0);
    for (void* p : owned)
      if (!p) return ctx_fail(c, DCGP_ERR_ALLOC, "layer: device allocation failed");
    return DCGP_OK;
  }
  // partial sums of squares per output: 16-column strips (prep_solve) or the chain's (panel, column tile) blocks, whichever is more
  static size_t kl_slots(int Mp) { const int a = Mp / 16 + 1, c = chain_rhs_slots(Mp); return (size_t)(a > c ? a : c); }
  void alloc_bank(int b) {
    GpMats& q = gbank[b];
    q.M = M; q.Mp = Mp; q.R = R; q.Rp = round_up(R, 16);
    const size_t mm = (size_t)Mp * Mp;
    q.K = dalloc(mm); q.Linv = dalloc(mm); q.LinvT = dalloc(mm);
    if (!is_head && !white && need_prior_) { q.Kp = dalloc(mm); q.Lpinv = dalloc(mm); q.LpinvT = dalloc(mm); q.klpp = dalloc((size_t)(R + 1) * kl_slots(Mp)); }
    q.Lq = dalloc((size_t)R * mm);
    q.qmu = dalloc((size_t)Mp * q.Rp);
    if (white) { q.G = q.Lq; q.alpha = q.qmu; }
    else { q.G = dalloc((size_t)R * mm); q.alpha = dalloc((size_t)Mp * q.Rp); q.klp = dalloc((size_t)(R + 1) * kl_slots(Mp)); }
    ZTb[b] = dalloc((size_t)Lp * Mp);
    znb[b] = dalloc(Mp);
    ZSb[b] = dalloc((size_t)Lz * Mp);
  }
  int use_bank(int b) {
    if (!gbank[b].K) {
      const size_t before = owned.size();
      alloc_bank(b);
      for (size_t i = before; i < owned.size(); ++i)
        if (!owned[i]) return ctx_fail(ctx, DCGP_ERR_ALLOC, "layer: device allocation failed");
    }
    g = gbank[b]; ZT = ZTb[b]; zn = znb[b]; ZS = ZSb[b];
    return DCGP_OK;
  }
  // one contiguous block per layer [gZ | gq_mu | gq_sqrt | gw | gscal] so that a single all-reduce covers the layer
  static constexpr size_t kGradBlockPad = 64;   // ranks - 1 at most
  double* pstage = nullptr;                     // [grad_block_count() + pad]: the layer's parameters in the gradient block's layout (sharded optimiser step)
  int ensure_stage() {
    if (pstage) return DCGP_OK;
    pstage = dalloc(grad_block_count() + kGradBlockPad);
    return pstage ? DCGP_OK : ctx_fail(ctx, DCGP_ERR_ALLOC, "layer: staging allocation failed");
  }
  size_t grad_block_count() const { return (size_t)M * v.L + (size_t)M * R + (size_t)R * M * M + (size_t)v.P + 3 + (is_head ? (size_t)v.L : 0); }
  int ensure_grads() {
    if (gZ) return DCGP_OK;
    double* blk = dalloc(grad_block_count() + kGradBlockPad);   // (padding: the sharded exchange rounds the block up to ranks x shard)
    gslots = dalloc(48);
    if (!blk || !gslots) return ctx_fail(ctx, DCGP_ERR_ALLOC, "layer: gradient allocation failed");
    gZ = blk; gq_mu = gZ + (size_t)M * v.L; gq_sqrt = gq_mu + (size_t)M * R; gw = gq_sqrt + (size_t)R * M * M; gscal = gw + v.P;
    gard = is_head ? gscal + 3 : nullptr;   // [L] d / d ARD lengthscales (dense head), zero otherwise
    return DCGP_OK;
  }
  int ensure_adam() {
    if (hyp) return DCGP_OK;
    const size_t nz = (size_t)M * v.L, nm = (size_t)M * R, nq = (size_t)R * M * M, nw = (size_t)v.P;
    for (int k = 0; k < 2; ++k) {
      aZ[k] = dalloc(nz); aq_mu[k] = dalloc(nm); aq_sqrt[k] = dalloc(nq); aw[k] = dalloc(nw); ahyp[k] = dalloc(3);
      if (!aZ[k] || !aq_mu[k] || !aq_sqrt[k] || !aw[k] || !ahyp[k]) return ctx_fail(ctx, DCGP_ERR_ALLOC, "layer: optimiser state allocation failed");
      HIP_TRY(ctx, hipMemsetAsync(aZ[k], 0, nz * sizeof(double), ctx->stream));
      HIP_TRY(ctx, hipMemsetAsync(aq_mu[k], 0, nm * sizeof(double), ctx->stream));
      HIP_TRY(ctx, hipMemsetAsync(aq_sqrt[k], 0, nq * sizeof(double), ctx->stream));
      HIP_TRY(ctx, hipMemsetAsync(aw[k], 0, (nw ? nw : 2) * sizeof(double), ctx->stream));
      HIP_TRY(ctx, hipMemsetAsync(ahyp[k], 0, 3 * sizeof(double), ctx->stream));
    }
    hyp = dalloc(3);
    if (!hyp) return ctx_fail(ctx, DCGP_ERR_ALLOC, "layer: optimiser state allocation failed");
    if (ard)
      for (int k = 0; k < 2; ++k) {
        if (!(aard[k] = dalloc(v.L))) return ctx_fail(ctx, DCGP_ERR_ALLOC, "layer: optimiser state allocation failed");
        HIP_TRY(ctx, hipMemsetAsync(aard[k], 0, (size_t)v.L * sizeof(double), ctx->stream));
      }
    return DCGP_OK;
  }
  int upload(double* dst, const double* src_host, size_t n) {
    if (!src_host) return ctx_fail(ctx, DCGP_ERR_ARG, "layer: null parameter array");
    HIP_TRY(ctx, hipMemcpyAsync(dst, src_host, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DCGP_OK;
  }
  PrepLayerArgs prep_args(double jitter) const {
    PrepLayerArgs p;
    p.Z = Z; p.Z0 = Z0; p.q_sqrt = has_qsqrt ? q_sqrt : nullptr; p.q_mu = q_mu;
    p.K = g.K; p.Kp = g.Kp; p.ZT = ZT; p.zn = zn; p.Lq = g.Lq; p.qmu = g.qmu;
    p.M = M; p.Mp = Mp; p.L = v.L; p.Lp = Lp; p.R = R; p.Rp = g.Rp;
    p.bk = base(); p.jitter = jitter; p.in_scale = in_scale;
    p.ZS = base_type == 0 ? ZS : nullptr; p.Lz = Lz;
    return p;
  }
  // step 1 of the forward: everything that depends only on this layer's parameters
  int prepare(double jitter) {
    DCGP_TRY(rbf_gram_padded(ctx, Z, M, v.L, base(), jitter, g.K, Mp, Mp));
    if (g.Kp) DCGP_TRY(rbf_gram_padded(ctx, Z0, M, v.L, base(), jitter, g.Kp, Mp, Mp));
    DCGP_TRY(z_transpose_norms(ctx, Z, M, v.L, ZT, Mp, Lp, zn));
    if (base_type == 0) DCGP_TRY(sweep_operand(ctx, Z, in_scale, M, Mp, v.L, variance, ls, ZS));
    if (has_qsqrt)
      DCGP_TRY(pad_copy(ctx, q_sqrt, M, M, M, g.Lq, Mp, Mp, Mp, 1, R, (long)M * M, (long)Mp * Mp));
    DCGP_TRY(pad_copy(ctx, q_mu, M, R, R, g.qmu, g.Rp, Mp, g.Rp, 0, 1, 0, 0));
    return DCGP_OK;
  }
};

// batched factorisation of a set of [Mp x Mp] matrices that share Mp
struct FactorGroup {
  int Mp = 0;
  std::vector<double*> K, Linv, LinvT;
  double **dK = nullptr, **dLinv = nullptr, **dLinvT = nullptr;
  int* d_info = nullptr;
  bool uploaded = false;
  // right-hand sides riding the chain (ChainRhs, common.h): one entry per matrix, Yw filled in by upload()
  std::vector<ChainRhs> rhs;
  ChainRhs* d_rhs = nullptr;
  double* d_yw = nullptr;
  int max_R = 0;
  bool ride = false;   // some matrix of the group carries right-hand sides
  void release() {
    if (dK) hipFree(dK);
    if (dLinv) hipFree(dLinv);
    if (dLinvT) hipFree(dLinvT);
    if (d_info) hipFree(d_info);
    if (d_rhs) hipFree(d_rhs);
    if (d_yw) hipFree(d_yw);
    dK = dLinv = dLinvT = nullptr; d_info = nullptr; d_rhs = nullptr; d_yw = nullptr; uploaded = false;
  }
  int upload(dcgp_ctx* ctx) {
    if (uploaded) return DCGP_OK;
    const size_t n = K.size();
    if (hipMalloc((void**)&dK, n * sizeof(double*)) != hipSuccess || hipMalloc((void**)&dLinv, n * sizeof(double*)) != hipSuccess ||
        hipMalloc((void**)&dLinvT, n * sizeof(double*)) != hipSuccess || hipMalloc((void**)&d_info, n * sizeof(int)) != hipSuccess)
      return ctx_fail(ctx, DCGP_ERR_ALLOC, "factor group: allocation failed");
    HIP_TRY(ctx, hipMemcpy(dK, K.data(), n * sizeof(double*), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(dLinv, Linv.data(), n * sizeof(double*), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(dLinvT, LinvT.data(), n * sizeof(double*), hipMemcpyHostToDevice));
    if (ride) {
      size_t total = 0;
      for (auto& r : rhs) total += (r.Lq || r.qmu) ? (size_t)r.R * Mp * Mp + (size_t)Mp * r.Rp : 0;
      if (hipMalloc((void**)&d_yw, (total ? total : 2) * sizeof(double)) != hipSuccess || hipMalloc((void**)&d_rhs, n * sizeof(ChainRhs)) != hipSuccess)
        return ctx_fail(ctx, DCGP_ERR_ALLOC, "factor group: allocation failed");
      if (dcgp_poison()) { hipMemset(d_yw, 0xFF, (total ? total : 2) * sizeof(double)); hipDeviceSynchronize(); }
      size_t off = 0;
      for (auto& r : rhs) {
        r.Yw = d_yw + off;
        off += (r.Lq || r.qmu) ? (size_t)r.R * Mp * Mp + (size_t)Mp * r.Rp : 0;
      }
      HIP_TRY(ctx, hipMemcpy(d_rhs, rhs.data(), n * sizeof(ChainRhs), hipMemcpyHostToDevice));
    }
    uploaded = true;
    return DCGP_OK;
  }
  // defer_finish: inv(L) and its transpose are complete on return, the factor itself is copied back over K by finish() --
  // only the KL terms (and the reverse pass) read it, so the copy need not sit in front of the first layer
  bool deferred = false;
  bool rode = false;   // the most recent run() carried the right-hand sides (the ctx's options may rule it out from one step to the next)
  int run(dcgp_ctx* ctx, bool defer_finish = false) {
    DCGP_TRY(upload(ctx));
    rode = ride && chain_can_ride(ctx, Mp);
    DCGP_TRY(factor_inverse_batched(ctx, dK, dLinv, dLinvT, (int)K.size(), Mp, Mp, d_info, defer_finish, rode ? d_rhs : nullptr, max_R));
    deferred = defer_finish;
    return DCGP_OK;
  }
  int finish(dcgp_ctx* ctx) {
    if (!deferred) return DCGP_OK;
    deferred = false;
    return factor_finish_batched(ctx, dK, (int)K.size(), Mp, Mp);
  }
};

// ConvLayer.conditional_ND (+ sampling) on `rows` input images taken as X[(n % n_mod)]
static inline int conv_forward(dcgp_ctx* ctx, LayerState& L, const double* X, int rows, int n_mod, int rep, long rep_stride,
                 const double* z, uint64_t seed, uint32_t stream_id, double jitter, double* out_sample, double* out_mean,
                 double* out_var, const std::string& pfx, hipEvent_t factor_done = nullptr,
                               hipEvent_t prep_done = nullptr, int phase = 3, bool keep_state = true, const RngMap* rmap = nullptr) {
  // phase bit 0: the K_uf sweep (needs only Z); bit 1: conditional + finalize (needs the factorisation).  The model
  // path enqueues bit 0 of its first layer BEFORE the long side-stream sequence so that the sweep is not held up
  // by the host still enqueueing the factorisation chain.
  // keep_state: K_uf and A1 are left in the "<pfx>Kuf" / "<pfx>A1" workspaces (the reverse pass reads them there).
  const int Mp = L.Mp, P = L.v.P;
  const long Kc = (long)rows * P;
  if (Kc > 0x7fffff00L) return ctx_fail(ctx, DCGP_ERR_ARG, "conv layer: %ld patch columns exceed the 32-bit tile index", Kc);
  const long ldb = col_ld(Kc);
  {
    // one launch for the whole layer where the shape allows (conv_fused.hip): the strip of K_uf / A1 a workgroup owns
    // stays in LDS from the patch gather to the sample
    ConvFusedArgs fa;
    fa.X = X; fa.n_mod = n_mod;
    fa.H = L.v.H; fa.W = L.v.W; fa.C = L.v.C; fa.f = L.v.f; fa.s = L.v.s; fa.Wo = L.v.Wo; fa.P = P; fa.L = L.v.L; fa.Lp = L.Lp;
    fa.HWC = L.v.H * L.v.W * L.v.C;
    fa.ZT = L.ZT; fa.zn = L.zn; fa.M = L.M; fa.Mp = Mp; fa.bk = L.base();
    fa.ZS = L.ZS; fa.Lz = L.Lz; fa.csq = sqrt(1.4426950408889634074) / L.ls;
    fa.LinvT = L.g.LinvT; fa.G = L.has_qsqrt ? L.g.G : nullptr; fa.alpha = L.g.alpha; fa.R = L.R; fa.Rp = L.g.Rp;
    fa.Kc = (int)Kc; fa.knn = L.variance;
    fa.rep = rep; fa.rep_stride = rep_stride; fa.z = z; fa.seed = seed; fa.stream_id = stream_id; fa.jitter = jitter;
    fa.out_sample = out_sample; fa.out_mean = out_mean; fa.out_var = out_var; fa.idm = L.identity_mean;
    if (rmap) fa.rmap = *rmap;
    if (conv_fused_ok(ctx, fa)) {
      if (!(phase & 2)) return DCGP_OK;   // nothing to run ahead of the factorisation: the sweep is part of the one launch
      if (keep_state) {
        fa.Kuf_out = (double*)ws_get(ctx, pfx + "Kuf", (size_t)Mp * ldb * sizeof(double));
        fa.A1_out = (double*)ws_get(ctx, pfx + "A1", (size_t)Mp * ldb * sizeof(double));
        if (!fa.Kuf_out || !fa.A1_out) return DCGP_ERR_ALLOC;
        fa.ldk = ldb;
      }
      // G / alpha are recorded behind the factorisation on the chain's stream: one wait covers both
      if (prep_done) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, prep_done, 0));
      else if (factor_done) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, factor_done, 0));
      return conv_fused(ctx, fa);
    }
  }
  // The sweep + GEMM route materialises K_uf [Mp][columns] and fetches whole k-tiles of it through 32-bit-offset buffer
  // descriptors: an operand slab must stay under 2 GiB.  A forward-only pass takes a larger batch in chunks of whole images
  // (sweep, conditional, finalize per chunk; outputs and noise stay indexed by the global column); the training step keeps
  // K_uf and A1 of the whole batch for its reverse pass and is refused by the GEMM as before.
  int chunk_rows = rows;
  if (!keep_state)
    while (chunk_rows > 1 && (long)Mp * col_ld((long)chunk_rows * P) * 8 >= (1L << 31)) chunk_rows = (chunk_rows + 1) / 2;
  const bool chunked = chunk_rows < rows;
  if (chunked && !(phase & 2)) return DCGP_OK;   // nothing to run ahead: the slab is reused chunk after chunk
  const long ldc = col_ld((long)chunk_rows * P);
  double* B = (double*)ws_get(ctx, pfx + "Kuf", (size_t)Mp * ldc * sizeof(double));
  if (!B) return DCGP_ERR_ALLOC;
  if ((phase & 1 || chunked) && Mp > L.M) HIP_TRY(ctx, hipMemsetAsync(B + (size_t)L.M * ldc, 0, (size_t)(Mp - L.M) * ldc * sizeof(double), ctx->stream));
  for (int r0 = 0; r0 < rows; r0 += chunk_rows) {
    const int nr = rows - r0 < chunk_rows ? rows - r0 : chunk_rows;
    const long kc = (long)nr * P;
    PatchRbfArgs a;
    a.X = X; a.N = nr; a.n_mod = n_mod; a.n0 = r0;
    a.H = L.v.H; a.W = L.v.W; a.C = L.v.C; a.f = L.v.f; a.s = L.v.s; a.Ho = L.v.Ho; a.Wo = L.v.Wo; a.P = P; a.L = L.v.L;
    a.ZT = L.ZT; a.zn = L.zn; a.M = L.M; a.Mp = Mp; a.Lp = L.Lp;
    a.bk = L.base();
    a.out = B; a.sM = ldc; a.sN = P; a.sP = 1;
    a.share_cu = phase == 1;
    if ((phase & 1) || chunked) {
      bool done = false;
      if (a.bk.type == 0) {   // RBF: the unit sweep in its storing form (head_units.hip)
        HeadUnitsArgs h;
        h.X = X; h.n_mod = n_mod; h.N = nr; h.n0 = r0;
        h.H = L.v.H; h.W = L.v.W; h.C = L.v.C; h.f = L.v.f; h.s = L.v.s; h.Wo = L.v.Wo; h.P = P; h.L = L.v.L; h.Lq = L.Lz;
        h.ZS = L.ZS; h.M = L.M; h.Mp = Mp;
        h.csq = sqrt(1.4426950408889634074) / L.ls; h.log2var = log2(L.variance);
        h.kuf = B; h.sM = ldc; h.sN = P; h.sP = 1;
        h.share_cu = phase == 1;
        h.upw_force = (int)ctx->opt.kuf_upw;   // A/B switch (0: head_units_plan chooses)
        h.stream_k = (int)ctx->opt.kuf_stream;
        h.wpg_force = (int)ctx->opt.kuf_wpg;
        h.split_force = (int)ctx->opt.kuf_split;
        h.occ_force = (int)ctx->opt.sweep_occ;
        h.share_kb = (int)ctx->opt.share_kb;
        h.no_rep = (int)ctx->opt.kuf_no_rep;   // A/B switch: every row evaluated even where rows share an image
        h.timer = L.v.L > 64 ? "kuf_long" : "kuf";   // long patches (deeper layers, L = 250) are MFMA-bound, short ones HBM-bound: two families
        head_units_plan(&h);
        if (head_units_ok(h)) { DCGP_TRY(head_units(ctx, h)); done = true; }
      }
      if (!done) DCGP_TRY(patch_rbf(ctx, a, "kuf"));
    }
    if (!(phase & 2)) return DCGP_OK;
    if (factor_done && r0 == 0) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, factor_done, 0));   // inv(L) comes from the side stream
    CondScratch sc;
    DCGP_TRY(cond_core(ctx, L.g, B, ldc, (int)kc, L.white, L.has_qsqrt, pfx.c_str(), &sc, r0 == 0 ? prep_done : nullptr));
    FinalizeArgs fa;
    fa.s1p = sc.s1p; fa.nrb1 = sc.nrb1; fa.s2p = sc.s2p; fa.nrb3 = sc.nrb3; fa.mu = sc.mu; fa.ldk = ldc;
    fa.Kc = (int)kc; fa.R = L.R; fa.knn_scalar = L.variance; fa.col0 = (long)r0 * P;
    fa.rep = rep; fa.rep_stride = rep_stride; fa.z = z; fa.seed = seed; fa.stream_id = stream_id; fa.jitter = jitter;
    fa.out_sample = out_sample; fa.out_mean = out_mean; fa.out_var = out_var;
    if (rmap) fa.rmap = *rmap;
    if (L.identity_mean) {
      fa.X = X; fa.idm = 1; fa.H = L.v.H; fa.W = L.v.W; fa.C = L.v.C; fa.f = L.v.f; fa.s = L.v.s; fa.Wo = L.v.Wo; fa.P = P;
      fa.n_mod = n_mod;
    }
    DCGP_TRY(finalize_layer(ctx, fa));
  }
  return DCGP_OK;
}

// SVGP head: Kzx / Kdiag from the conv kernel, then the shared conditional
static inline int head_forward(dcgp_ctx* ctx, LayerState& L, const double* X, int rows, int n_mod, double* kd, double* out_mean,
                 double* out_var, const std::string& pfx, hipEvent_t factor_done = nullptr,
                               hipEvent_t prep_done = nullptr, int phase = 3, int sweep_mode = 0, bool* early_done = nullptr,
                               bool keep_k = false) {
  // keep_k (a training step): the unit sweep also leaves every patch response in the "<pfx>g_Kfull" workspace [Mp][col_ld(rows P)] and sets
  // L.kfull_ready (the reverse pass would otherwise evaluate them all again)
  // sweep_mode (the unit-sweep route only): 1 = the sweep launch alone (it needs Z only: the model enqueues it in front of the long
  // factorisation chain so that the host still enqueueing the chain does not hold it up; *early_done tells whether anything was launched),
  // 2 = everything behind a sweep launched that way, 0 = both
  const int Mp = L.Mp;
  const long ldb = col_ld(rows);
  if (sweep_mode != 2) L.kfull_ready = false;   // (set below by the one route that keeps the patch responses)
  if (sweep_mode != 1) L.a1h_ready = false;     // (set below where the one-launch conditional leaves A1 behind)
  double* a1_out = nullptr;
  if (keep_k && sweep_mode != 1 && !ctx->opt.grad_no_keep_k && head_cond_fused_ok(L.g) && !ctx->opt.head_unfused) {
    a1_out = (double*)ws_get(ctx, pfx + "g_A1h", (size_t)Mp * ldb * sizeof(double));
    if (!a1_out) return DCGP_ERR_ALLOC;
  }
  double* B = (double*)ws_get(ctx, pfx + "Kzx", (size_t)Mp * ldb * sizeof(double));
  if (!B) return DCGP_ERR_ALLOC;
  if ((phase & 1) && sweep_mode != 2 && Mp > L.M) HIP_TRY(ctx, hipMemsetAsync(B + (size_t)L.M * ldb, 0, (size_t)(Mp - L.M) * ldb * sizeof(double), ctx->stream));
  PatchRbfArgs a;
  a.X = X; a.N = rows; a.n_mod = n_mod;
  a.H = L.v.H; a.W = L.v.W; a.C = L.v.C; a.f = L.v.f; a.s = L.v.s; a.Ho = L.v.Ho; a.Wo = L.v.Wo; a.P = L.v.P; a.L = L.v.L;
  a.ZT = L.ZT; a.zn = L.zn; a.M = L.M; a.Mp = Mp; a.Lp = L.Lp;
  a.bk = L.base();
  a.out = B; a.sM = ldb; a.sN = 1; a.sP = 0;
  a.w = L.w; a.scale = 1.0 / (double)L.v.P; a.reduce = 1;
  a.in_scale = L.in_scale;
  a.share_cu = phase == 1;
  const bool unfused = ctx->opt.head_unfused != 0;   // A/B switch
  if (phase == 3 && L.kernel_type == 0 && a.bk.type == 0 && !L.in_scale) {
    // ConvKernel head: Kzx and Kdiag as wave-sized units of one launch (head_units.hip), any M
    HeadUnitsArgs h;
    h.X = X; h.n_mod = n_mod; h.N = rows;
    h.H = L.v.H; h.W = L.v.W; h.C = L.v.C; h.f = L.v.f; h.s = L.v.s; h.Wo = L.v.Wo; h.P = L.v.P; h.L = L.v.L; h.Lq = L.Lz;
    h.ZS = L.ZS; h.M = L.M; h.Mp = Mp;
    h.csq = sqrt(1.4426950408889634074) / L.ls; h.log2var = log2(L.variance);
    h.w = L.w; h.kzx = B; h.ldk = ldb; h.kzx_scale = 1.0 / (double)L.v.P;
    h.share_cu = factor_done != nullptr;   // first layer of the model with the chain on another stream: the chain runs beside this launch
    h.want_kd = 1;
    h.tail_mode = (int)ctx->opt.head_tail;
    h.occ_force = (int)ctx->opt.sweep_occ;
    h.share_kb = (int)ctx->opt.share_kb;
    h.upw_force = (int)ctx->opt.head_upw;
    // (long patches only: a value of a 5 x 5 x 1 patch costs 7 MFMAs to evaluate again -- less than the sweep loses by storing it; head-only MNIST
    // model 0.64 -> 0.67 ms with the values kept, conv + head at L = 250 1.50 -> 1.43)
    if (keep_k && !ctx->opt.grad_no_keep_k && L.v.L >= 64) {
      const long ldf = col_ld((long)rows * L.v.P);
      h.kfull = (double*)ws_get(ctx, pfx + "g_Kfull", (size_t)Mp * ldf * sizeof(double));
      if (!h.kfull) return DCGP_ERR_ALLOC;
      h.kf_sM = ldf; h.kf_sN = L.v.P;
    }
    head_units_plan(&h);
    if (h.kfull && !head_units_ok(h)) { h.kfull = nullptr; head_units_plan(&h); }   // (offsets past 32 bits: the reverse pass recomputes)
    if (head_units_ok(h)) {
      if (sweep_mode != 2) L.kfull_ready = h.kfull != nullptr;
      h.kd = (double*)ws_get(ctx, "kdiag_partial", (size_t)rows * h.n_kd * sizeof(double));   // [rows][n_kd] partial sums (head_units_plan)
      if (!h.kd) return DCGP_ERR_ALLOC;
      if (sweep_mode != 2) DCGP_TRY(head_units(ctx, h));
      if (sweep_mode == 1) {
        if (early_done) *early_done = true;
        return DCGP_OK;
      }
      const double kd_scale = 1.0 / ((double)L.v.P * (double)L.v.P);
      if (prep_done) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, prep_done, 0));
      else if (factor_done) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, factor_done, 0));
      if (head_cond_fused_ok(L.g) && !unfused) {
        L.a1h_ready = a1_out != nullptr;
        return head_cond_fused(ctx, L.g, B, ldb, rows, L.has_qsqrt, h.kd, out_mean, out_var, h.n_kd, kd_scale, a1_out, ldb);
      }
      DCGP_TRY(kdiag_reduce(ctx, h.kd, h.n_kd, rows, kd_scale, kd));
      CondScratch sc;
      DCGP_TRY(cond_core(ctx, L.g, B, ldb, rows, L.white, L.has_qsqrt, pfx.c_str(), &sc, nullptr, true));
      FinalizeArgs fa;
      fa.s1p = sc.s1p; fa.nrb1 = sc.nrb1; fa.s2p = sc.s2p; fa.nrb3 = sc.nrb3; fa.mu = sc.mu; fa.ldk = ldb;
      fa.Kc = rows; fa.R = L.R; fa.knn_vec = kd;
      fa.out_mean = out_mean; fa.out_var = out_var;
      return finalize_layer(ctx, fa);
    }
  }
  if (sweep_mode == 1) return DCGP_OK;   // not the unit-sweep route: nothing is launched ahead
  if (phase == 3 && L.kernel_type == 0 && a.bk.type == 0 && head_cond_fused_ok(L.g) && !unfused) {
    // ConvKernel head, M <= 256: Kzx and Kdiag in one launch, then the whole conditional in one launch that adds up the Kdiag
    // tile-pair sums itself -- two launches on one stream for the layer
    const double* kdp = nullptr; int kd_n = 1; double kd_scale = 1.0;
    DCGP_TRY(head_sweep(ctx, a, L.w, &kdp, &kd_n, &kd_scale));
    if (prep_done) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, prep_done, 0));
    else if (factor_done) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, factor_done, 0));
    L.a1h_ready = a1_out != nullptr;
    return head_cond_fused(ctx, L.g, B, ldb, rows, L.has_qsqrt, kdp, out_mean, out_var, kd_n, kd_scale, a1_out, ldb);
  }
  bool kd_on_side = false;
  if (phase & 1) {
    // Kdiag (all patch pairs of an image) is needed by finalize only: it runs on the side stream beside the
    // Kzx sweep and the conditional GEMMs instead of in front of them.
    hipStream_t main_s = ctx->stream;
    kd_on_side = !ctx->no_side;
    if (kd_on_side) {
      HIP_TRY(ctx, hipEventRecord(ctx->ev_aux, main_s));   // X is ready at this point of the main stream
      HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream_aux, ctx->ev_aux, 0));
      ctx->stream = ctx->stream_aux;
    }
    int rc;
    if (L.kernel_type == 0) {
      rc = head_kdiag(ctx, X, rows, n_mod, L.v.H, L.v.W, L.v.C, L.v.f, L.v.s, L.base(), L.w, kd);
    } else {
      rc = additive_kdiag_async(ctx, rows, L.v.P, L.variance, L.w, kd);
    }
    if (kd_on_side) {
      if (rc == DCGP_OK && hipEventRecord(ctx->ev_aux2, ctx->stream_aux) != hipSuccess) rc = DCGP_ERR_HIP;
      ctx->stream = main_s;
    }
    DCGP_TRY(rc);
    DCGP_TRY(patch_rbf(ctx, a, "head_kzx"));
  }
  if (!(phase & 2)) return DCGP_OK;
  if (factor_done) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, factor_done, 0));
  if (head_cond_fused_ok(L.g) && !unfused) {
    // few columns (one per image): both triangular products, the mean and mean / var in one launch (head_cond.hip)
    if (prep_done) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, prep_done, 0));
    if (!ctx->no_side) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_aux2, 0));
    return head_cond_fused(ctx, L.g, B, ldb, rows, L.has_qsqrt, kd, out_mean, out_var);
  }
  CondScratch sc;
  DCGP_TRY(cond_core(ctx, L.g, B, ldb, rows, L.white, L.has_qsqrt, pfx.c_str(), &sc, prep_done, true));
  // join: with phase == 2 the excursion was started by the earlier phase-1 call on the same stream pair
  if (!ctx->no_side) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_aux2, 0));
  FinalizeArgs fa;
  fa.s1p = sc.s1p; fa.nrb1 = sc.nrb1; fa.s2p = sc.s2p; fa.nrb3 = sc.nrb3; fa.mu = sc.mu; fa.ldk = ldb;
  fa.Kc = rows; fa.R = L.R; fa.knn_vec = kd;
  fa.out_mean = out_mean; fa.out_var = out_var;
  return finalize_layer(ctx, fa);
}

