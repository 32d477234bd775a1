// model.hip -- device-resident layers and the model-level forward (DGP_Base.propagate /
// _build_likelihood of doubly_stochastic_dgp; ConvLayer.conditional_ND of conv_gp/layers.py:96-135;
// SVGP_Layer.conditional_ND with the ConvKernel head of conv_gp/kernels.py:79-136).
//
// Per forward step, nothing cached across steps (main stream = the data path, side stream = the replicated M x M work):
//   1. main: every layer's Kuu(Z) (+ prior Kuu(Z0)), padded q_sqrt / q_mu, Z^T and |z|^2 -- one launch (prep.hip)
//   2. side: ONE batched Cholesky + inverse chain for all M x M matrices of the model (chol_fused.hip), then G / alpha of
//      every layer in one launch (head_cond.hip), then the KL terms; main meanwhile: the first layer's patch sweep
//   3. main, per conv layer: patch sweep -> stage-1 GEMM -> stage-3 GEMM -> mean -> finalize (+ sample)
//   4. main: head Kzx sweep (Kdiag beside it on the side stream), fused head conditional, RobustMax expectations
//   5. (multi-GPU) all-reduce of the data term, ELBO assembly, one 32-byte read-back, one host sync.
#include <cstdlib>
#include <cstring>
#include <memory>

#include "model_state.h"
#include "tail_dev.h"
#include <chrono>


namespace {

int g_model_counter = 0;

int build_groups(dcgp_model* m, int bank) {   // the layers must be on `bank` (use_bank)
  if (m->groups_built[bank]) return DCGP_OK;
  auto& groups = m->groups[bank];
  for (auto& gr : groups) gr.release();
  groups.clear();
  auto add = [&](int Mp, double* K, double* Linv, double* LinvT, const ChainRhs& r) {
    FactorGroup* g = nullptr;
    for (auto& gr : groups)
      if (gr.Mp == Mp) g = &gr;
    if (!g) { groups.emplace_back(); g = &groups.back(); g->Mp = Mp; }
    g->K.push_back(K); g->Linv.push_back(Linv); g->LinvT.push_back(LinvT); g->rhs.push_back(r);
    if (r.Lq || r.qmu) { g->ride = true; g->max_R = r.R > g->max_R ? r.R : g->max_R; }
  };
  // G = inv(L) Lq and alpha = inv(L) q_mu ride the chain (chol_fused.hip) where a layer is unwhitened and small enough; with a prior
  // Kuu(Z0) the same right-hand sides ride its chain for the KL's sums of squares
  for (auto& l : m->layers) {
    const GpMats& g = l->g;
    const bool rides = !l->white && g.Mp <= kChainRhsMaxMp && g.Rp <= 32 && g.G && g.alpha && g.klp;
    ChainRhs live{}, prior{};
    live.R = prior.R = g.R; live.Rp = prior.Rp = g.Rp;
    if (rides) { live.Lq = l->has_qsqrt ? g.Lq : nullptr; live.qmu = g.qmu; live.G = g.G; live.alpha = g.alpha; live.sums = g.klp; }
    if (rides && g.Kp && g.klpp && l->has_qsqrt) { prior.Lq = g.Lq; prior.qmu = g.qmu; prior.sums = g.klpp; }
    add(l->Mp, g.K, g.Linv, g.LinvT, live);
    if (g.Kp) add(l->Mp, g.Kp, g.Lpinv, g.LpinvT, prior);
  }
  m->groups_built[bank] = true;
  return DCGP_OK;
}

int ensure_events(dcgp_model* m) {
  if (m->events_ok) return DCGP_OK;
  dcgp_ctx* ctx = m->ctx;
  for (int b = 0; b < 2; ++b) {
    HIP_TRY(ctx, hipEventCreateWithFlags(&m->ev_sweep[b], hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&m->ev_factor[b], hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&m->ev_kl[b], hipEventDisableTiming));
    for (auto& e : m->ev_prep[b]) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  m->events_ok = true;
  return DCGP_OK;
}

int ensure(dcgp_ctx* ctx, double** p, size_t* cap, size_t n) {
  if (*cap >= n && *p) return DCGP_OK;
  if (*p) { hipDeviceSynchronize(); hipFree(*p); *p = nullptr; }   // steps in flight on any stream may still use it
  if (hipMalloc((void**)p, (n ? n : 2) * sizeof(double)) != hipSuccess) return ctx_fail(ctx, DCGP_ERR_ALLOC, "model: allocation failed");
  if (dcgp_poison()) { hipMemset(*p, 0xFF, (n ? n : 2) * sizeof(double)); hipDeviceSynchronize(); }   // debugging aid, see ws_get
  *cap = n;
  return DCGP_OK;
}

int ensure_out(dcgp_model* m, int li, int rows, int width, bool need_mv) {
  auto& o = m->outs[li];
  size_t n = (size_t)rows * width;
  if (o.cap < n) {
    hipDeviceSynchronize();   // steps in flight on any stream may still use them
    hipFree(o.sample); hipFree(o.mean); hipFree(o.var);
    o.sample = o.mean = o.var = nullptr;
    if (hipMalloc((void**)&o.sample, n * sizeof(double)) != hipSuccess || hipMalloc((void**)&o.mean, n * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&o.var, n * sizeof(double)) != hipSuccess)
      return ctx_fail(m->ctx, DCGP_ERR_ALLOC, "model: output allocation failed");
    if (dcgp_poison()) {
      hipMemset(o.sample, 0xFF, n * sizeof(double)); hipMemset(o.mean, 0xFF, n * sizeof(double)); hipMemset(o.var, 0xFF, n * sizeof(double));
      hipDeviceSynchronize();
    }
    o.cap = n;
  }
  (void)need_mv;
  o.rows = rows; o.width = width;
  return DCGP_OK;
}

struct CombineArgs {
  int nl;
  int M[8], R[8], white[8];
  double scale;
  const int* info[16];   // per factor group: potrf status words (0 or the 1-based failing column)
  int ninfo[16];
  int ngroups;
  double* host_out;   // pinned host slot (device-visible address) or nullptr
  double host_seq;    // completion word behind the four result words (see elbo_forward_collect_impl)
};
__global__ void combine_kernel(const double* __restrict__ scal_in, double* __restrict__ out, CombineArgs c) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double kl = 0.0;
  for (int l = 0; l < c.nl; ++l) {
    const double* k4 = scal_in + 4 + 4 * l;
    double two = k4[0] - (double)c.M[l] * c.R[l] - k4[1] + k4[3];
    if (!c.white[l]) two += (double)c.R[l] * k4[2];
    kl += 0.5 * two;
  }
  double data = scal_in[0];
  out[0] = data * c.scale - kl;
  out[1] = data;
  out[2] = kl;
  int bad = 0;   // first non-positive pivot of any factorisation: rides back with the result (one D2H, one sync)
  for (int g = 0; g < c.ngroups; ++g)
    for (int i = 0; i < c.ninfo[g]; ++i)
      if (c.info[g][i] && !bad) bad = c.info[g][i];
  out[3] = (double)bad;
  if (c.host_out) {
    for (int i = 0; i < 4; ++i) __hip_atomic_store(c.host_out + i, out[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(c.host_out + 4, c.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int read_info(dcgp_model* m, int* info_host) {
  dcgp_ctx* ctx = m->ctx;
  int bad = 0;
  for (auto& gr : m->groups[m->bank]) {
    std::vector<int> h(gr.K.size());
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), gr.d_info, h.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int v : h)
      if (v && !bad) bad = v;
  }
  if (info_host) *info_host = bad;
  if (bad) return ctx_fail(ctx, DCGP_ERR_NOT_PD, "Cholesky: matrix not positive definite at column %d", bad);
  return DCGP_OK;
}

// Restores ctx->stream when a forward step returns, whichever way
struct StreamGuard {
  dcgp_ctx* ctx; hipStream_t saved;
  explicit StreamGuard(dcgp_ctx* c) : ctx(c), saved(c->stream) {}
  ~StreamGuard() { ctx->stream = saved; ctx->ws_tag.clear(); }
};

// layers 0..n-1 forward; leaves ctx->stream on the main stream the step runs on (the caller holds a StreamGuard).
//   side stream: everything that depends on the parameters only, into the bank of this step's parity -- Kuu / prior Kuu /
//                Z^T / padded q_sqrt, q_mu of every layer (one launch), ONE batched Cholesky + inverse chain for all M x M
//                matrices, G / alpha of every layer (one launch), the KL terms;
//   main stream: the data path (sweeps, conditionals, sampling), gated per layer by the side stream's events.
// pipelined (dcgp_elbo_forward_enqueue): main = 30 CUs of every XCD, side = the other 2, so that the chain of step i + 1 runs
// under the data path of step i without competing for its CUs; otherwise both streams see the whole chip.
// what the assembly at the end of a step needs: layer shapes, the status words of the factor groups, the pinned result slot
void fill_finish(dcgp_model* model, std::vector<FactorGroup>& groups, double scale, int slot, ElboFinish* fin) {
  const int nl = (int)model->layers.size();
  fin->nl = nl; fin->scale = scale;
  for (int l = 0; l < nl; ++l) { fin->M[l] = model->layers[l]->M; fin->R[l] = model->layers[l]->R; fin->white[l] = model->layers[l]->white; }
  fin->ngroups = (int)groups.size();
  for (int g = 0; g < fin->ngroups && g < 16; ++g) { fin->info[g] = groups[g].d_info; fin->ninfo[g] = (int)groups[g].K.size(); }
  fin->host_out = model->h_ring_dev + 8 * slot;   // the last kernel of the step writes the result words into the pinned slot itself
  fin->host_seq = (double)(model->enq_seq + 1);   // ... and this step's ticket + 1 behind them
}

int forward_all(dcgp_model* m, const double* X, int N, int S, const double* const* zs, uint64_t seed, int dedup,
                bool need_kl, bool pipelined, int* rows_last) {
  dcgp_ctx* ctx = m->ctx;
  if (!m->has_head) return ctx_fail(ctx, DCGP_ERR_ARG, "model has no head layer");
  const int nl = (int)m->layers.size();
  if (nl > 8) return ctx_fail(ctx, DCGP_ERR_ARG, "at most 8 layers supported");
  // a local batch that overran its declared shard would draw the noise of the NEXT sample's images (the counters are laid out by the
  // un-sharded batch): correlated samples, not an error anyone would see.  Only the ELBO / gradient paths (need_kl) are bound by the
  // declared training shard: propagate / predict_y take any batch (an AccuracyLogger's batches on rank 3 of 4), their counter layout
  // does not matter
  if (need_kl && m->shard_global > 0 && (long)m->shard_lo + N > m->shard_global)
    return ctx_fail(ctx, DCGP_ERR_ARG, "forward: %d images from image %d on overrun the declared global batch of %d (dcgp_model_set_shard)", N,
                    m->shard_lo, m->shard_global);
  DCGP_TRY(ensure_events(m));
  // The chain of the previous step stands if no parameter was written since and this step may use it (model_state.h: factor_reuse): same bank, no
  // preparation, no factorisation, no G / alpha, no KL launches -- the step is its data path.
  const bool reuse = !pipelined && !m->grad_follows && !m->keep_state && m->chain_version == m->param_version && m->chain_with_kl == need_kl &&
                     m->factor_reuse >= (need_kl ? 2 : 1) && !ctx->opt.no_factor_reuse;
  if (!reuse) m->chain_version = 0;   // (stays 0 if this step fails on the way)
  const int bank = reuse ? m->bank : m->bank ^ 1;
  m->bank = bank;
  for (auto& l : m->layers) DCGP_TRY(l->use_bank(bank));
  DCGP_TRY(build_groups(m, bank));
  if (!m->d_scal && hipMalloc((void**)&m->d_scal, 128 * sizeof(double)) != hipSuccess)
    return ctx_fail(ctx, DCGP_ERR_ALLOC, "model: allocation failed");
  double* scal = m->d_scal + 64 * bank;
  m->outs.resize(nl);
  const std::string mp = "m" + std::to_string(m->id) + "_";
  const int rows0 = dedup ? N : S * N;   // rows entering layer 0; it reads image (row % N): tile(X,[S,1,1]) is never formed

  const bool no_side = ctx->no_side;   // A/B switch: everything on one stream
  const bool part = pipelined && ctx->stream_m && !no_side;
  const hipStream_t main_s = part ? ctx->stream_m : ctx->stream;
  // Where the parameter-only chain runs.  A synchronous step has nothing else to do until the factorisation is there: the chain
  // sits on the main stream itself (no cross-stream hand-off in front of the first layer, ~15 us each) and only the KL terms
  // fork to the side stream.  A step enqueued beside others: the side stream of its bank, so that it overtakes the step in flight.
  const hipStream_t kl_s = no_side ? main_s : (part ? ctx->stream2_m : (pipelined ? (bank ? ctx->stream2b : ctx->stream2) : ctx->stream2));
  // (A first layer on the sweep + GEMM route keeps the side stream: its sweep needs Z only and runs beside the chain.)
  // (A model that opens with the head -- the reference's "1-layer" -- has a first kernel that needs Z only: its sweep runs on the main
  // stream beside the chain on the side stream, and the step is the longer of the two instead of their sum.)
  // (a chain of one or two panels is shorter than the hand-off between streams; option head_no_overlap: A/B switch
  // and how bench.py times the sweep alone on the chip)
  bool first_fused = !(m->layers[0]->is_head && m->layers[0]->Mp >= 96 && !ctx->opt.head_no_overlap);
  if (!m->layers[0]->is_head) {
    const LayerState& L0 = *m->layers[0];
    ConvFusedArgs fa;
    fa.Mp = L0.Mp; fa.M = L0.M; fa.R = L0.R; fa.Rp = L0.g.Rp; fa.P = L0.v.P; fa.HWC = L0.v.H * L0.v.W * L0.v.C; fa.Lp = L0.Lp; fa.Lz = L0.Lz;
    first_fused = conv_fused_ok(ctx, fa);
  }
  const hipStream_t chain_s = reuse ? main_s : ((pipelined || !first_fused) ? kl_s : main_s);
  // a step on the other main stream than the previous one starts behind it
  if (ctx->ev_last_valid && ctx->last_main != main_s) HIP_TRY(ctx, hipStreamWaitEvent(main_s, ctx->ev_last, 0));
  ctx->last_main = main_s;

  // one layer of the data path on ctx->stream.  phase 1: only what needs Z alone (the layer's sweep, where it is a launch of its own),
  // 2: the rest, 3: both
  bool early_head = false;
  auto layer_step = [&](int li, const double* F, int rows, int n_mod, int* out_rows_p, int phase) -> int {
    LayerState& L = *m->layers[li];
    const std::string pfx = mp + std::to_string(li) + "_";
    const double* z = zs ? zs[li] : nullptr;
    hipEvent_t fdone = (li == 0 && chain_s != main_s) ? m->ev_factor[bank] : nullptr;   // later layers are stream-ordered behind layer 0
    hipEvent_t pdone = chain_s != main_s ? m->ev_prep[bank][li] : nullptr;
    if (!L.is_head) {
      const int width = L.v.P * L.R;
      const bool expand = dedup && li == 0;          // N distinct images -> S*N sampled rows
      const int out_rows = expand ? S * N : rows;
      DCGP_TRY(ensure_out(m, li, out_rows, width, true));
      auto& o = m->outs[li];
      // device RNG: with a shard declared (dcgp_model_set_shard) every element draws at its counter in the un-sharded batch, one
      // stream per layer -- the step's value is then independent of the number of ranks; otherwise one stream per (layer, rank)
      RngMap rm;
      const bool sharded = m->shard_global > 0;
      if (sharded && (m->shard_global != N || m->shard_lo != 0)) { rm.W = width; rm.Nl = N; rm.Ng = m->shard_global; rm.lo = m->shard_lo; }
      DCGP_TRY(conv_forward(ctx, L, F, rows, n_mod, expand ? S : 1, (long)N * width, z, seed, (uint32_t)(li + 1 + (sharded ? 0 : 64 * ctx->rank)),
                            m->jitter, o.sample, m->keep_outputs ? o.mean : nullptr, m->keep_outputs ? o.var : nullptr, pfx,
                            fdone, pdone, phase, m->keep_state, &rm));
      *out_rows_p = out_rows;
    } else {
      DCGP_TRY(ensure_out(m, li, rows, L.R, true));
      auto& o = m->outs[li];
      DCGP_TRY(ensure(ctx, &m->d_kd, &m->kd_cap, (size_t)rows));
      DCGP_TRY(head_forward(ctx, L, F, rows, n_mod, m->d_kd, o.mean, o.var, pfx, fdone, pdone, 3,
                            phase == 1 ? 1 : (phase == 2 && early_head ? 2 : 0), phase == 1 ? &early_head : nullptr,
                            m->keep_state && m->grad_follows));
      if (phase == 1) { *out_rows_p = rows; return DCGP_OK; }
      if (m->keep_outputs) {
        // the head's sample is not needed by the ELBO; produce it only on request
        size_t n = (size_t)rows * L.R;
        if (z) {
          DCGP_TRY(reparam_async(ctx, o.mean, o.var, z, n, m->jitter, o.sample));
        } else {
          HIP_TRY(ctx, hipMemcpyAsync(o.sample, o.mean, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        }
      }
      *out_rows_p = rows;
    }
    return DCGP_OK;
  };
  int rc = DCGP_OK;
  const bool xs = chain_s != main_s;
  // (steps kept in flight: the preparation stays on the chain's stream, where it runs under the previous step's data path -- on the main stream it
  // waited for that step: head-only model 4830 -> 4590 steps/s in flight)
  const bool prep_on_main = xs && !first_fused && !pipelined && !ctx->opt.no_early_sweep && !ctx->opt.prep_on_chain;
  const bool prep_split = prep_on_main && !ctx->opt.prep_one_launch;
  const bool chain_first = prep_split && !m->layers[0]->is_head;
  bool early0 = false, side_kl = false;
  if (reuse) ++m->chain_skips;
  if (!reuse) {
  // ---- the parameter-only chain ----
  ctx->stream = chain_s;
  // its scratch per model and bank: the chains / KL terms of two steps in flight may overlap, and with the deferred copy the tail
  // launch reads the prior factor's diagonal out of this scratch at the END of the step -- another model's chain on the same
  // ctx must not have overwritten it by then
  ctx->ws_tag = "~m" + std::to_string(m->id) + "b" + std::to_string(bank);
  if (chain_s != main_s) {
    if (m->done_valid[bank]) HIP_TRY(ctx, hipStreamWaitEvent(chain_s, m->done_ev[bank], 0));   // the bank's previous reader
    else if (ctx->ev_last_valid) HIP_TRY(ctx, hipStreamWaitEvent(chain_s, ctx->ev_last, 0));    // first use: behind whatever ran last
  }
  // (events only where another stream waits for them: each record is a packet in front of the next launch)
  // A first layer whose sweep is a launch of its own (the head-first model) is the step's critical path: the operand preparation sits on the MAIN
  // stream, the sweep directly behind it, and it is the CHAIN that pays the hand-off between streams -- it ends well before the sweep does.  With the
  // preparation on the chain's stream the sweep started 20.8 us into the step (8 us of preparation + the event), now at ~9.
  {
    PrepArgs pa;
    pa.nl = nl;
    for (int li = 0; li < nl; ++li) pa.l[li] = m->layers[li]->prep_args(m->jitter);
    // Round 6, second step: the preparation in TWO launches, each on the stream of its reader -- what a sweep reads (Z^T, |z|^2, the scaled Z) on the
    // main stream, the Gram matrices and the padded q_sqrt / q_mu on the chain's -- and no event between the streams at the head of the step: the
    // record was a packet between the preparation and the sweep (7.6 us from one to the other), the wait held the chain back (option prep_one_launch: A/B)
    if (prep_split) {
      // Which stream's part the host enqueues first is which part gets the chip first.  A model that opens with the head: the sweep is the step's longest
      // path (170 us against the chain's ~150 beside it) -- its part and the sweep, then the chain's.  A conv layer on the sweep + GEMM route (M > 256): the
      // CHAIN is (the first product waits for inv(L) long after the sweep is done) -- enqueued behind the sweep its preparation ran 71 us beside it instead
      // of ~15 and the first product of cfg4 started 53 us later (profiles/the first cut of this split)
      if (chain_first) rc = prepare_all(ctx, pa, ~kPrepSweepTasks);
      ctx->stream = main_s;
      if (rc == DCGP_OK) rc = prepare_all(ctx, pa, kPrepSweepTasks);
      ctx->stream = chain_s;   // (head first: the chain's part is enqueued BEHIND the sweep, below: at the head of a synchronous step the device waits for the host, ~4 us a launch)
    } else {
      if (prep_on_main) ctx->stream = main_s;
      rc = prepare_all(ctx, pa);
    }
  }
  if (rc == DCGP_OK && xs && !first_fused && !prep_split && hipEventRecord(m->ev_sweep[bank], ctx->stream) != hipSuccess) rc = DCGP_ERR_HIP;   // Z^T, |z|^2: what a sweep needs
  if (rc == DCGP_OK && prep_on_main && !prep_split && hipStreamWaitEvent(chain_s, m->ev_sweep[bank], 0) != hipSuccess) rc = DCGP_ERR_HIP;
  // The first layer's sweep needs nothing else: it goes to the main stream NOW, in front of the chain's ~12 launches -- enqueued behind
  // them it started when the host was done with those, 60 us after prepare_all had finished (cfg2 head-only: 0.287 -> 0.24 ms).
  if (rc == DCGP_OK && xs && !first_fused && !ctx->opt.no_early_sweep) {
    ctx->stream = main_s;
    if (!prep_on_main && hipStreamWaitEvent(main_s, m->ev_sweep[bank], 0) != hipSuccess) rc = DCGP_ERR_HIP;
    int out_rows = 0;
    if (rc == DCGP_OK) rc = layer_step(0, X, rows0, N, &out_rows, 1);
    early0 = rc == DCGP_OK;
    ctx->stream = chain_s;
  }
  if (rc == DCGP_OK && prep_split && !chain_first) {
    PrepArgs pa;
    pa.nl = nl;
    for (int li = 0; li < nl; ++li) pa.l[li] = m->layers[li]->prep_args(m->jitter);
    rc = prepare_all(ctx, pa, ~kPrepSweepTasks);
  }
  // with a single factor group its "chol_Lout" scratch stays untouched until the deferred copy runs on the KL stream
  const bool defer = m->groups[bank].size() == 1 && need_kl && !m->keep_state;
  ctx->chain_alone = chain_s == main_s && !pipelined;
  ctx->chain_ride_ok = first_fused;
  for (auto& gr : m->groups[bank])
    if (rc == DCGP_OK) rc = gr.run(ctx, defer);
  ctx->chain_alone = ctx->chain_ride_ok = true;
  if (rc == DCGP_OK && xs && hipEventRecord(m->ev_factor[bank], ctx->stream) != hipSuccess) rc = DCGP_ERR_HIP;
  // G_r = inv(L) Lq_r and alpha = inv(L) q_mu of every layer (cond_prep): gate the second conditional GEMM
  bool prep_done[8] = {}, rode[8] = {};
  int kl_ns[8] = {}, kl_nsa[8] = {};
  for (int li = 0; li < nl && rc == DCGP_OK; ++li) {   // layers whose right-hand sides rode the chain (build_groups): nothing left to do
    LayerState& L = *m->layers[li];
    bool live_sums = false, prior_sums = false;
    for (auto& gr : m->groups[bank]) {
      if (!gr.rode) continue;
      for (size_t i = 0; i < gr.K.size(); ++i) {
        if (gr.K[i] == L.g.K && (gr.rhs[i].Lq || gr.rhs[i].qmu)) { rode[li] = true; live_sums = gr.rhs[i].Lq != nullptr; }
        if (L.g.Kp && gr.K[i] == L.g.Kp && gr.rhs[i].Lq) prior_sums = true;
      }
    }
    if (!rode[li]) continue;
    prep_done[li] = true;
    L.g.klp_valid = live_sums; L.g.klpp_valid = prior_sums;
    kl_ns[li] = chain_rhs_slots(L.Mp); kl_nsa[li] = (L.Mp + 31) / 32;
    L.g.kl_ns = kl_ns[li]; L.g.kl_nsa = kl_nsa[li];
  }
  if (rc == DCGP_OK) {   // every other layer the one-launch route covers (unwhitened, M <= 256, <= 16 outputs): head_cond.hip
    GpMats* gs[8]; int wh[8]; bool hq[8];
    for (int li = 0; li < nl; ++li) { gs[li] = &m->layers[li]->g; wh[li] = m->layers[li]->white; hq[li] = m->layers[li]->has_qsqrt; }
    rc = prep_solve_all(ctx, gs, wh, hq, nl, prep_done, rode);
  }
  // The KL pieces need nothing but parameter-only state.  Where prep_solve left the sums of squares they are made of (every
  // layer unwhitened, M <= 256, with q_sqrt), one extra workgroup per layer of the tail launch adds them up with the factors'
  // log-determinants: no KL launches, no stream of their own, no fork in front of the first layer and no join (tail_dev.h).
  bool kl_tail = need_kl && rc == DCGP_OK && nl <= 8 && !ctx->opt.kl_side;
  for (int li = 0; li < nl && kl_tail; ++li) {
    const LayerState& L = *m->layers[li];
    kl_tail = !L.white && prep_done[li] && L.has_qsqrt && L.g.klp_valid && (!L.g.Kp || L.g.klpp_valid);
  }
  m->kl_in_tail[bank] = kl_tail;
  if (kl_tail) {
    KlTail& kt = m->kl_tail[bank];
    kt.nl = nl;
    const double* lout = nullptr;   // deferred copy: the factors are still in the chain's scratch, [matrix of the group][Mp][Mp]
    if (defer) {
      auto it = ctx->ws.find("chol_Lout" + ctx->ws_tag);
      if (it != ctx->ws.end()) lout = (const double*)it->second.first;
    }
    for (int li = 0; li < nl; ++li) {
      const LayerState& L = *m->layers[li];
      KlTailLayer& q = kt.l[li];
      const double* prior = L.g.Kp ? L.g.Kp : L.g.K;
      q.Lfac = prior; q.ldf = L.Mp;
      if (defer) {
        const auto& gr = m->groups[bank][0];
        long idx = -1;
        for (size_t i = 0; i < gr.K.size(); ++i) if (gr.K[i] == prior) idx = (long)i;
        if (!lout || idx < 0) { rc = ctx_fail(ctx, DCGP_ERR_ARG, "model: factor scratch of layer %d not found", li); break; }
        q.Lfac = lout + idx * (long)L.Mp * L.Mp;
      }
      q.Lq = L.g.Lq; q.sums = L.g.Kp ? L.g.klpp : L.g.klp; q.M = L.M; q.Mp = L.Mp; q.R = L.R;
      q.ns = kl_ns[li]; q.nsa = kl_nsa[li];
    }
  }
  side_kl = need_kl && !kl_tail;
  for (int li = 0; li < nl && rc == DCGP_OK; ++li) {
    if (!prep_done[li]) rc = cond_prep(ctx, m->layers[li]->g, m->layers[li]->white, m->layers[li]->has_qsqrt);   // generic GEMMs
    if (rc == DCGP_OK && (xs || (li == nl - 1 && side_kl && kl_s != chain_s)) && hipEventRecord(m->ev_prep[bank][li], ctx->stream) != hipSuccess)
      rc = DCGP_ERR_HIP;   // per layer: layer 0 does not wait for the others (same stream: only the fork of the KL terms needs one)
  }
  if (rc == DCGP_OK && side_kl) {
    if (kl_s != chain_s) {   // fork: the KL terms need the factors only, the main stream goes on with the layers
      if (hipStreamWaitEvent(kl_s, m->ev_prep[bank][nl - 1], 0) != hipSuccess) rc = DCGP_ERR_HIP;
      ctx->stream = kl_s;
    }
    for (auto& gr : m->groups[bank])
      if (rc == DCGP_OK) rc = gr.finish(ctx);   // the factor back over K (deferred copy): the KL terms read its diagonal
    for (int li = 0; li < nl && rc == DCGP_OK; ++li) {
      LayerState& L = *m->layers[li];
      const double* Lp = L.g.Kp ? L.g.Kp : L.g.K;
      const double* LpinvT = L.g.Kp ? L.g.LpinvT : L.g.LinvT;
      rc = kl_layer(ctx, L.g, Lp, LpinvT, L.white, (mp + std::to_string(li)).c_str(), scal + 4 + 4 * li);
    }
  }
  }   // !reuse
  const bool kl_join = side_kl && kl_s != main_s;
  if (rc == DCGP_OK && kl_join && hipEventRecord(m->ev_kl[bank], ctx->stream) != hipSuccess) rc = DCGP_ERR_HIP;
  ctx->stream = main_s;
  ctx->ws_tag.clear();
  if (rc != DCGP_OK) {
    hipStreamSynchronize(kl_s);
    hipStreamSynchronize(chain_s);
    return rc;
  }
  if (!reuse && !pipelined && !m->grad_follows && !m->keep_state) { m->chain_version = m->param_version; m->chain_with_kl = need_kl; }
  // A training step: the parameter-only part of the reverse pass (grad.hip, grad_kl_early) runs beside the forward pass on the auxiliary
  // stream.  Its start is marked behind the FIRST layer (below): that layer's launch fills the chip at the full batch, and forty short
  // launches squeezed in between its rounds cost it more than they gained.  They are enqueued by dcgp_elbo_grad behind the whole forward
  // pass (in front of the layers below the host kept the first layer waiting for 170 us, in front of the tail launch the end of the
  // forward pass for 60).
  m->gkl_state = (m->grad_follows && !pipelined && grad_kl_early(m, false, false) == 1) ? 2 : 0;
  // (a first layer of a few thousand patch columns -- the de-duplicated batch -- leaves half the chip idle: there the mark is here, behind the chain)
  const bool mark_behind_first = (long)rows0 * m->layers[0]->v.P >= 8192;
  if (m->gkl_state && !mark_behind_first) HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, chain_s));
  // with the chain on a side stream a mark on the MAIN stream orders layer 0's operands only: grad_kl_early also waits for the G / alpha
  // of the other layers (cond_prep of whitened / M > 256 layers runs behind ev_prep[0] on that stream)
  m->gkl_prep_wait = (m->gkl_state && mark_behind_first && chain_s != main_s) ? nl : 0;

  // sweeps read Z^T / |z|^2 of this bank (a one-launch first layer waits for its G / alpha, recorded behind them on the same stream)
  if (chain_s != main_s && !first_fused && !prep_on_main) HIP_TRY(ctx, hipStreamWaitEvent(main_s, m->ev_sweep[bank], 0));
  const double* F = X;
  int rows = rows0, n_mod = N;
  // Join the side stream where the wait is already satisfied when the main stream gets to it: in front of the last layer when
  // other layers precede it (the KL terms finish beside the first of them), behind it otherwise.  In front of the tail kernel
  // the wait packet sat between two short launches at the very end of the step (6 us).
  const bool join_early = nl > 1;
  for (int li = 0; li < nl; ++li) {
    int out_rows = 0;
    if (li == nl - 1 && join_early && kl_join) HIP_TRY(ctx, hipStreamWaitEvent(main_s, m->ev_kl[bank], 0));
    // the KL pieces ride the head's one-launch conditional where there is one (head_cond.hip); m->kl_rode[bank] says whether they did
    if (li == nl - 1 && need_kl && m->kl_in_tail[bank]) { ctx->kl_ride = &m->kl_tail[bank]; ctx->kl_ride_scal = scal; ctx->kl_rode = false; }
    const int rc_l = layer_step(li, F, rows, n_mod, &out_rows, (li == 0 && early0) ? 2 : 3);
    if (li == nl - 1) { m->kl_rode[bank] = need_kl && m->kl_in_tail[bank] && ctx->kl_rode; ctx->kl_ride = nullptr; ctx->kl_rode = false; }
    DCGP_TRY(rc_l);
    if (li == 0 && m->gkl_state && mark_behind_first) HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, main_s));   // (the chain's results are ordered in front of this layer)
    if (!m->layers[li]->is_head) F = m->outs[li].sample;
    rows = out_rows;
    n_mod = rows;
  }
  if (!join_early && kl_join) HIP_TRY(ctx, hipStreamWaitEvent(main_s, m->ev_kl[bank], 0));   // join the side stream
  *rows_last = rows;
  return DCGP_OK;
}

// The end of a step's data path on its main stream: `ev` (already recorded there, behind the step's last command) frees the
// bank for its next writer and lets a step on the other main stream start.  ev == nullptr: the caller synchronises the stream
// itself before anything else is enqueued.
int forward_done(dcgp_model* m, hipEvent_t ev) {
  dcgp_ctx* ctx = m->ctx;
  m->done_ev[m->bank] = ev;
  m->done_valid[m->bank] = ev != nullptr;
  ctx->ev_last = ev;
  ctx->ev_last_valid = ev != nullptr;
  return DCGP_OK;
}

}  // namespace

extern "C" {

int dcgp_model_create(dcgp_ctx* ctx, int num_samples, double jitter, dcgp_model** out) {
  if (!ctx || !out || num_samples <= 0 || !(jitter >= 0.0)) return ctx ? ctx_fail(ctx, DCGP_ERR_ARG, "model_create: bad args") : DCGP_ERR_ARG;
  dcgp_model* m = new dcgp_model();
  m->ctx = ctx; m->S = num_samples; m->jitter = jitter; m->id = ++g_model_counter;
  *out = m;
  return DCGP_OK;
}

int dcgp_model_destroy(dcgp_model* model) {
  if (!model) return DCGP_ERR_ARG;
  dcgp_ctx* ctx = model->ctx;
  hipDeviceSynchronize();   // every stream: the model's workspaces may still be in use
  ctx->ev_last = nullptr;   // (it may be one of this model's result-ring events; nothing is in flight any more)
  ctx->ev_last_valid = false;
  // the per-model workspaces of the forward / reverse pass live in the ctx under "m<id>_..." (the training step's are large:
  // R x M x columns doubles per conv layer); they go with the model
  const std::string pfx = "m" + std::to_string(model->id) + "_";
  const std::string tag = "~m" + std::to_string(model->id) + "b";   // the chain's scratch (forward_all's ws_tag)
  for (auto it = ctx->ws.begin(); it != ctx->ws.end();) {
    if (it->first.compare(0, pfx.size(), pfx) == 0 || it->first.find(tag) != std::string::npos) {
      hipFree(it->second.first);
      it = ctx->ws.erase(it);
    } else {
      ++it;
    }
  }
  for (auto it = ctx->chain_epochs.begin(); it != ctx->chain_epochs.end();)   // sync areas of the one-launch chain: gone with their workspaces
    it = (it->first.find(tag) != std::string::npos) ? ctx->chain_epochs.erase(it) : std::next(it);
  if (ctx->ws_tag.find(tag) != std::string::npos) ctx->ws_tag.clear();   // operator calls behind this model must not name scratch after it
  delete model;
  return DCGP_OK;
}

int dcgp_model_set_shard(dcgp_model* model, int first_image, int global_batch) {
  if (!model) return DCGP_ERR_ARG;
  if (global_batch < 0 || first_image < 0 || (global_batch > 0 && first_image >= global_batch))
    return ctx_fail(model->ctx, DCGP_ERR_ARG, "set_shard: first image %d of a global batch of %d", first_image, global_batch);
  model->shard_lo = first_image; model->shard_global = global_batch;
  return DCGP_OK;
}

int dcgp_model_set_keep_outputs(dcgp_model* model, int on) {
  if (!model) return DCGP_ERR_ARG;
  model->keep_outputs = on != 0;
  return DCGP_OK;
}

int dcgp_model_add_conv_layer(dcgp_model* model, int H, int W, int C, int f, int stride, int M, int R, int white,
                              int identity_mean, double variance, double lengthscale, const double* Z_host,
                              const double* Z0_host, const double* q_mu_host, const double* q_sqrt_host) {
  if (!model) return DCGP_ERR_ARG;
  dcgp_ctx* ctx = model->ctx;
  if (model->has_head) return ctx_fail(ctx, DCGP_ERR_ARG, "conv layers must be added before the head");
  if (identity_mean && (f % 2 == 0)) return ctx_fail(ctx, DCGP_ERR_ARG, "Conv2dMean supports odd filter sizes only");
  std::unique_ptr<LayerState> L(new LayerState());
  DCGP_TRY(L->init(ctx, false, H, W, C, f, stride, M, R, white, identity_mean, 0, variance, lengthscale));
  DCGP_TRY(L->upload(L->Z, Z_host, (size_t)M * L->v.L));
  DCGP_TRY(L->upload(L->Z0, Z0_host ? Z0_host : Z_host, (size_t)M * L->v.L));
  DCGP_TRY(L->upload(L->q_mu, q_mu_host, (size_t)M * R));
  DCGP_TRY(L->upload(L->q_sqrt, q_sqrt_host, (size_t)R * M * M));
  model->layers.push_back(std::move(L));
  model->groups_built[0] = model->groups_built[1] = false;
  return DCGP_OK;
}

int dcgp_model_set_head(dcgp_model* model, int H, int W, int C, int f, int stride, int M, int R, int white,
                        int kernel_type, double variance, double lengthscale, const double* Z_host, const double* w_host,
                        const double* q_mu_host, const double* q_sqrt_host) {
  if (!model) return DCGP_ERR_ARG;
  dcgp_ctx* ctx = model->ctx;
  if (model->has_head) return ctx_fail(ctx, DCGP_ERR_ARG, "head already set");
  if (kernel_type != 0 && kernel_type != 1) return ctx_fail(ctx, DCGP_ERR_ARG, "Invalid last layer kernel");
  std::unique_ptr<LayerState> L(new LayerState());
  DCGP_TRY(L->init(ctx, true, H, W, C, f, stride, M, R, white, 0, kernel_type, variance, lengthscale));
  DCGP_TRY(L->upload(L->Z, Z_host, (size_t)M * L->v.L));
  DCGP_TRY(L->upload(L->w, w_host, (size_t)L->v.P));
  DCGP_TRY(L->upload(L->q_mu, q_mu_host, (size_t)M * R));
  DCGP_TRY(L->upload(L->q_sqrt, q_sqrt_host, (size_t)R * M * M));
  model->layers.push_back(std::move(L));
  model->has_head = true;
  model->groups_built[0] = model->groups_built[1] = false;
  return DCGP_OK;
}

int dcgp_model_set_factor_reuse(dcgp_model* model, int mode) {
  if (!model || mode < 0 || mode > 2) return model ? ctx_fail(model->ctx, DCGP_ERR_ARG, "set_factor_reuse: mode 0, 1 or 2") : DCGP_ERR_ARG;
  model->factor_reuse = mode;
  return DCGP_OK;
}
int dcgp_model_chain_skips(dcgp_model* model, uint64_t* out) {
  if (!model || !out) return DCGP_ERR_ARG;
  *out = model->chain_skips;
  return DCGP_OK;
}

int dcgp_model_set_param(dcgp_model* model, int layer, const char* which, const double* value_host, size_t count) {
  if (model) ++model->param_version;   // (whatever becomes of the call: the parameter-only state of earlier steps is not reused)
  if (!model || !which || !value_host) return DCGP_ERR_ARG;
  dcgp_ctx* ctx = model->ctx;
  if (!strcmp(which, "likelihood_epsilon")) {   // model-wide, `layer` is ignored
    if (count != 1 || !(value_host[0] > 0 && value_host[0] < 1)) return ctx_fail(ctx, DCGP_ERR_ARG, "set_param(likelihood_epsilon): one value in (0, 1)");
    model->eps = value_host[0];
    return DCGP_OK;
  }
  if (layer < 0 || layer >= (int)model->layers.size()) return ctx_fail(ctx, DCGP_ERR_ARG, "set_param: no layer %d", layer);
  LayerState& L = *model->layers[layer];
  auto expect = [&](size_t n) { return count == n ? DCGP_OK : ctx_fail(ctx, DCGP_ERR_ARG, "set_param(%s): expected %zu values, got %zu", which, n, count); };
  if (!strcmp(which, "Z")) { DCGP_TRY(expect((size_t)L.M * L.v.L)); return L.upload(L.Z, value_host, count); }
  if (!strcmp(which, "Z0")) {
    if (!L.Z0) return ctx_fail(ctx, DCGP_ERR_ARG, "set_param: the head has no frozen prior Z");
    DCGP_TRY(expect((size_t)L.M * L.v.L)); return L.upload(L.Z0, value_host, count);
  }
  if (!strcmp(which, "q_mu")) { DCGP_TRY(expect((size_t)L.M * L.R)); return L.upload(L.q_mu, value_host, count); }
  if (!strcmp(which, "q_sqrt")) { DCGP_TRY(expect((size_t)L.R * L.M * L.M)); return L.upload(L.q_sqrt, value_host, count); }
  if (!strcmp(which, "w")) {
    if (!L.w) return ctx_fail(ctx, DCGP_ERR_ARG, "set_param: only the head has patch weights");
    DCGP_TRY(expect((size_t)L.v.P)); return L.upload(L.w, value_host, count);
  }
  if (!strcmp(which, "variance")) { DCGP_TRY(expect(1)); if (!(value_host[0] > 0)) return ctx_fail(ctx, DCGP_ERR_ARG, "variance must be > 0"); L.variance = value_host[0]; return DCGP_OK; }
  if (!strcmp(which, "lengthscale")) { DCGP_TRY(expect(1)); if (!(value_host[0] > 0)) return ctx_fail(ctx, DCGP_ERR_ARG, "lengthscale must be > 0"); L.ls = value_host[0]; return DCGP_OK; }
  if (!strcmp(which, "ard_lengthscales")) {
    // gpflow RBF(D, ARD=True) on the flattened features (--last-kernel rbf, conv_gp/models.py:160-168): a head whose
    // single patch is the whole input (P == 1); x / l and Z / l are formed while the operands are staged
    if (!L.is_head || L.v.P != 1) return ctx_fail(ctx, DCGP_ERR_ARG, "set_param(ard_lengthscales): only a single-patch head takes per-dimension lengthscales");
    DCGP_TRY(expect((size_t)L.v.L));
    std::vector<double> inv(count);
    for (size_t i = 0; i < count; ++i) {
      if (!(value_host[i] > 0)) return ctx_fail(ctx, DCGP_ERR_ARG, "lengthscales must be > 0");
      inv[i] = 1.0 / value_host[i];
    }
    if (!L.in_scale && !(L.in_scale = L.dalloc(count))) return ctx_fail(ctx, DCGP_ERR_ALLOC, "layer: device allocation failed");
    if (!L.ard && !(L.ard = L.dalloc(count))) return ctx_fail(ctx, DCGP_ERR_ALLOC, "layer: device allocation failed");
    L.ls = 1.0;
    DCGP_TRY(L.upload(L.ard, value_host, count));
    return L.upload(L.in_scale, inv.data(), count);
  }
  if (!strcmp(which, "base_kernel")) {   // {type, variance, p1, p2}: 0 = RBF (p1 = lengthscale), 1 = ArcCosine order 0 (p1 = weight, p2 = bias variance)
    DCGP_TRY(expect(4));
    const int type = (int)value_host[0];
    if ((type != 0 && type != 1) || !(value_host[1] > 0) || !(value_host[2] > 0) || (type == 1 && !(value_host[3] >= 0)))
      return ctx_fail(ctx, DCGP_ERR_ARG, "set_param(base_kernel): bad kernel description");
    if (type == 1 && L.is_head) return ctx_fail(ctx, DCGP_ERR_ARG, "set_param(base_kernel): the head kernels are RBF-based (conv_gp/models.py:160-187)");
    L.base_type = type; L.variance = value_host[1];
    if (type == 0) L.ls = value_host[2]; else { L.acos_w = value_host[2]; L.acos_b = value_host[3]; }
    return DCGP_OK;
  }
  return ctx_fail(ctx, DCGP_ERR_ARG, "set_param: unknown parameter '%s'", which);
}

int dcgp_elbo_forward(dcgp_model* model, const double* X, const int32_t* y, int N, double scale,
                      const double* const* z_per_layer_host, uint64_t seed, int dedup_layer0, double* out_host,
                      int* info_host) {
  return elbo_forward_impl(model, X, y, N, scale, z_per_layer_host, seed, dedup_layer0, out_host, info_host);
}

int dcgp_elbo_forward_enqueue(dcgp_model* model, const double* X, const int32_t* y, int N, double scale,
                              const double* const* z_per_layer_host, uint64_t seed, int dedup_layer0, uint64_t* ticket) {
  // the caller means to keep steps in flight: data path and parameter-only chain on disjoint CUs (forward_all)
  return elbo_forward_enqueue_impl(model, X, y, N, scale, z_per_layer_host, seed, dedup_layer0, ticket, true);
}

int dcgp_elbo_forward_collect(dcgp_model* model, uint64_t ticket, double* out_host, int* info_host) {
  return elbo_forward_collect_impl(model, ticket, out_host, info_host);
}

}  // extern "C"

int elbo_forward_enqueue_impl(dcgp_model* model, const double* X, const int32_t* y, int N, double scale,
                              const double* const* z_per_layer_host, uint64_t seed, int dedup_layer0, uint64_t* ticket,
                              bool pipelined) {
  if (!model || !X || !y || N <= 0 || !ticket) return model ? ctx_fail(model->ctx, DCGP_ERR_ARG, "elbo_forward: bad args") : DCGP_ERR_ARG;
  dcgp_ctx* ctx = model->ctx;
  if (model->enq_seq - model->col_seq >= (uint64_t)dcgp_model::RING)
    return ctx_fail(ctx, DCGP_ERR_ARG, "elbo_forward_enqueue: %d steps in flight, collect the oldest first", dcgp_model::RING);
  if (!model->h_ring) {
    if (hipHostMalloc((void**)&model->h_ring, dcgp_model::RING * 8 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void**)&model->h_ring_dev, model->h_ring, 0) != hipSuccess)
      return ctx_fail(ctx, DCGP_ERR_ALLOC, "elbo_forward: pinned result slots");
    memset(model->h_ring, 0, dcgp_model::RING * 8 * sizeof(double));
    for (auto& e : model->ring_ev) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  int rows = 0;
  const int S = model->S;
  const auto host_t0 = std::chrono::steady_clock::now();
  StreamGuard guard(ctx);   // forward_all leaves ctx->stream on the step's main stream
  const int nl = (int)model->layers.size();
  LayerState& H = *model->layers[nl - 1];
  const int slot = (int)(model->enq_seq % dcgp_model::RING);
  DCGP_TRY(forward_all(model, X, N, S, z_per_layer_host, seed, dedup_layer0, true, pipelined, &rows));
  auto& o = model->outs[nl - 1];
  double* scal = model->d_scal + 64 * model->bank;
  auto& groups_now = model->groups[model->bank];
  DCGP_TRY(ensure(ctx, &model->d_ve, &model->ve_cap, (size_t)rows));
  // rows == S*N normally; a head-only model under dedup has rows == N with S identical copies
  const double inv_s = (rows == S * N) ? 1.0 / S : 1.0;
  if (groups_now.size() > 16) return ctx_fail(ctx, DCGP_ERR_ARG, "model: too many factor groups");
  ElboFinish fin;
  fill_finish(model, groups_now, scale, slot, &fin);
  const KlTail* klt = (model->kl_in_tail[model->bank] && !model->kl_rode[model->bank]) ? &model->kl_tail[model->bank] : nullptr;
  // From here on a kernel that writes this slot's completion word (ticket + 1) may be in flight.  If anything below fails the ticket is
  // NOT handed out and the next enqueue reuses slot and ticket: the word is cleared, behind a device sync, so that the stale kernel's
  // write cannot satisfy the retried step's wait early.
  struct SlotGuard {
    dcgp_model* m; int slot; bool armed = true;
    ~SlotGuard() {
      if (!armed) return;
      hipDeviceSynchronize();
      m->h_ring[8 * slot + 4] = 0.0;
    }
  } slot_guard{model, slot};
  if (!ctx->comm) {
    // (Measured and dropped, round 6: this launch's work at the end of the head's one-launch conditional -- the last of the R workgroups of a 16-row
    // strip to arrive runs the rows' expectations, the last workgroup of the launch the sum and the assembly.  cfg2 head-only 0.2210 -> 0.2234 ms, cfg1
    // 0.1416 -> 0.1462, conv + head +2 us (profiles/r06_tail_ride_and_prep_ab.txt): two levels of agent-scope release/acquire at the end of 200
    // workgroups cost more than the 13 us launch they replace.)
    // expectations, their sum, the KL pieces where the chain left their ingredients, and the ELBO assembly in one launch
    DCGP_TRY(elbo_tail(ctx, o.mean, o.var, y, rows, N, H.R, model->eps, model->d_ve, inv_s, scal, fin, klt));
  } else {
    // multi-GPU: the data term is summed over the ranks between the reduction and the assembly
    ElboFinish none;
    DCGP_TRY(elbo_tail(ctx, o.mean, o.var, y, rows, N, H.R, model->eps, model->d_ve, inv_s, scal, none, klt));
    // A step kept in flight: collective and assembly go to the comm stream behind one event, and the main stream is free for the next step's
    // data path at once -- in stream, a 1-double ncclAllReduce (~20 us of latency over xGMI, more when a rank is late) sat in front of the next
    // step's layer kernel.  What it reads (scal of this bank, the chain's status words) stays untouched until the bank's next writer, which
    // waits for this step's ring event (forward_all: done_ev).  (Not for a training step: its gradient collectives follow on the main stream,
    // and one communicator's collectives stay on one stream.)
    const bool side_comm = pipelined && !model->grad_follows && !ctx->no_side && !ctx->opt.comm_inline;
    if (side_comm) {
      if (!ctx->stream_comm) {
        if (hipStreamCreateWithFlags(&ctx->stream_comm, hipStreamNonBlocking) != hipSuccess) return ctx_fail(ctx, DCGP_ERR_HIP, "comm stream");
        for (auto& e : ctx->ev_comm) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
      }
      HIP_TRY(ctx, hipEventRecord(ctx->ev_comm[slot], ctx->stream));
      HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream_comm, ctx->ev_comm[slot], 0));
      ctx->stream = ctx->stream_comm;   // (the StreamGuard above restores the caller's stream)
      if (ctx->comm_gate) DCGP_TRY(comm_gate_wait(ctx));
    }
    DCGP_TRY(allreduce_sum_f64_async(ctx, scal, 1));
    CombineArgs c;
    c.nl = nl; c.scale = scale;
    for (int l = 0; l < nl; ++l) { c.M[l] = fin.M[l]; c.R[l] = fin.R[l]; c.white[l] = fin.white[l]; }
    c.ngroups = fin.ngroups;
    for (int g = 0; g < c.ngroups; ++g) { c.info[g] = fin.info[g]; c.ninfo[g] = fin.ninfo[g]; }
    c.host_out = fin.host_out; c.host_seq = fin.host_seq;
    hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(64), 0, ctx->stream, scal, scal + 40, c);
    LAUNCH_CHECK(ctx);
  }
  HIP_TRY(ctx, hipEventRecord(model->ring_ev[slot], ctx->stream));
  DCGP_TRY(forward_done(model, model->ring_ev[slot]));
  if (ctx->timing) {   // host time to enqueue one step (everything before the wait), reported beside the kernel timers
    auto& acc = ctx->tim["host_enqueue"];
    acc.launches += 1;
    acc.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
  }
  slot_guard.armed = false;
  *ticket = model->enq_seq++;
  return DCGP_OK;
}

int elbo_forward_collect_impl(dcgp_model* model, uint64_t ticket, double* out_host, int* info_host) {
  if (!model || !out_host) return model ? ctx_fail(model->ctx, DCGP_ERR_ARG, "elbo_forward_collect: bad args") : DCGP_ERR_ARG;
  dcgp_ctx* ctx = model->ctx;
  if (info_host) *info_host = 0;
  if (ticket != model->col_seq || ticket >= model->enq_seq)
    return ctx_fail(ctx, DCGP_ERR_ARG, "elbo_forward_collect: tickets are collected in the order they were handed out");
  const int slot = (int)(ticket % dcgp_model::RING);
  // The step's last kernel writes its four result words and then (system-scope release) ticket + 1 into the pinned slot: the host polls
  // that word instead of waiting for the event behind the kernel -- the event's signal is another packet for the command processor and a
  // wake-up through the runtime, several microseconds on a step of 150-800.  The event is still consulted now and then: a failed launch
  // or a lost device must end the wait.
  {
    const volatile double* hv = model->h_ring + 8 * slot;
    const double want = (double)(ticket + 1);
    const bool use_event = ctx->opt.sync_event != 0;   // A/B switch: the event wait this replaced
    if (use_event) {
      HIP_TRY(ctx, hipEventSynchronize(model->ring_ev[slot]));
    } else {
      // bounded spin: a step of this path is 0.15-0.8 ms; one that has not answered after ~65 000 polls (a millisecond or two: the
      // large configurations, a rank waiting for a slower peer's all-reduce) hands the core back and blocks on the event instead --
      // eight ranks of a node must not hold eight cores against RCCL's proxy threads
      for (unsigned spins = 1; hv[4] != want; ++spins) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#elif defined(__aarch64__)
        asm volatile("yield");
#endif
        if ((spins & 0xfff) == 0) {
          const hipError_t q = hipEventQuery(model->ring_ev[slot]);
          if (q == hipSuccess) break;                       // complete: coherent host memory already holds the words
          if (q != hipErrorNotReady) { HIP_TRY(ctx, q); }
          if (spins >= (1u << 16)) {
            HIP_TRY(ctx, hipEventSynchronize(model->ring_ev[slot]));
            break;
          }
        }
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
  }
  ++model->col_seq;
  const double* h = model->h_ring + 8 * slot;
  out_host[0] = h[0]; out_host[1] = h[1]; out_host[2] = h[2];
  // the timers resolve their events lazily, once nothing is in flight any more
  if (ctx->timing && ctx->pending.size() > 512 && model->col_seq == model->enq_seq) timing_flush(ctx);   // synchronises every stream of the ctx
  const int bad = (int)h[3];
  if (info_host) *info_host = bad;
  if (bad) return ctx_fail(ctx, DCGP_ERR_NOT_PD, "Cholesky: matrix not positive definite at column %d", bad);
  return DCGP_OK;
}

int elbo_forward_impl(dcgp_model* model, const double* X, const int32_t* y, int N, double scale,
                      const double* const* z_per_layer_host, uint64_t seed, int dedup_layer0, double* out_host,
                      int* info_host) {
  if (!model || !out_host) return model ? ctx_fail(model->ctx, DCGP_ERR_ARG, "elbo_forward: bad args") : DCGP_ERR_ARG;
  if (model->enq_seq != model->col_seq) return ctx_fail(model->ctx, DCGP_ERR_ARG, "elbo_forward: enqueued steps are still to be collected");
  if (info_host) *info_host = 0;
  uint64_t ticket = 0;
  DCGP_TRY(elbo_forward_enqueue_impl(model, X, y, N, scale, z_per_layer_host, seed, dedup_layer0, &ticket));
  return elbo_forward_collect_impl(model, ticket, out_host, info_host);
}

extern "C" {

int dcgp_model_propagate(dcgp_model* model, const double* X, int N, int S, const double* const* z_per_layer_host,
                         uint64_t seed, double* out_fmean, double* out_fvar, int* info_host) {
  if (!model || !X || N <= 0 || S <= 0) return model ? ctx_fail(model->ctx, DCGP_ERR_ARG, "propagate: bad args") : DCGP_ERR_ARG;
  dcgp_ctx* ctx = model->ctx;
  if (info_host) *info_host = 0;
  int rows = 0;
  StreamGuard guard(ctx);
  DCGP_TRY(forward_all(model, X, N, S, z_per_layer_host, seed, 0, false, false, &rows));
  DCGP_TRY(forward_done(model, nullptr));   // this call synchronises the stream before it returns
  const int nl = (int)model->layers.size();
  auto& o = model->outs[nl - 1];
  size_t n = (size_t)rows * o.width;
  if (out_fmean) HIP_TRY(ctx, hipMemcpyAsync(out_fmean, o.mean, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  if (out_fvar) HIP_TRY(ctx, hipMemcpyAsync(out_fvar, o.var, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return read_info(model, info_host);
}

namespace {
// mean over the S samples of the class probabilities: p_bar[n][k] = 1/S sum_s p[s*N + n][k]
__global__ void sample_mean_kernel(const double* __restrict__ p, int S, long NK, double* __restrict__ out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NK) return;
  double acc = 0.0;
  for (int s = 0; s < S; ++s) acc += p[(long)s * NK + i];
  out[i] = acc / (double)S;
}
}  // namespace

int dcgp_model_predict_y(dcgp_model* model, const double* X, int N, int S, const double* const* z_per_layer_host,
                         uint64_t seed, double* out_p, double* out_p_mean, int* info_host) {
  if (!model || !X || N <= 0 || S <= 0 || (!out_p && !out_p_mean))
    return model ? ctx_fail(model->ctx, DCGP_ERR_ARG, "predict_y: bad args") : DCGP_ERR_ARG;
  dcgp_ctx* ctx = model->ctx;
  if (info_host) *info_host = 0;
  int rows = 0;
  StreamGuard guard(ctx);
  DCGP_TRY(forward_all(model, X, N, S, z_per_layer_host, seed, 0, false, false, &rows));
  DCGP_TRY(forward_done(model, nullptr));   // this call synchronises the stream before it returns
  const int nl = (int)model->layers.size();
  auto& o = model->outs[nl - 1];
  const int K = o.width;
  if (K < 2) return ctx_fail(ctx, DCGP_ERR_ARG, "predict_y: the last layer has %d outputs, RobustMax needs >= 2", K);
  double* p = out_p;
  if (!p) {
    p = (double*)ws_get(ctx, "predict_p", (size_t)rows * K * sizeof(double));
    if (!p) return DCGP_ERR_ALLOC;
  }
  DCGP_TRY(varexp_rows(ctx, o.mean, o.var, nullptr, rows, 1, K, model->eps, p, 1));
  if (out_p_mean) {
    long NK = (long)N * K;
    hipLaunchKernelGGL(sample_mean_kernel, dim3((unsigned)((NK + 255) / 256)), dim3(256), 0, ctx->stream, p, S, NK, out_p_mean);
    LAUNCH_CHECK(ctx);
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return read_info(model, info_host);
}

int dcgp_model_layer_output(dcgp_model* model, int layer, double* out_sample, double* out_mean, double* out_var,
                            int* rows, int* width) {
  if (!model) return DCGP_ERR_ARG;
  dcgp_ctx* ctx = model->ctx;
  if (layer < 0 || layer >= (int)model->outs.size()) return ctx_fail(ctx, DCGP_ERR_ARG, "layer_output: no output for layer %d", layer);
  auto& o = model->outs[layer];
  if (rows) *rows = o.rows;
  if (width) *width = o.width;
  size_t n = (size_t)o.rows * o.width * sizeof(double);
  if ((out_mean || out_var || (out_sample && model->layers[layer]->is_head)) && !model->keep_outputs)
    return ctx_fail(ctx, DCGP_ERR_ARG, "layer_output: enable dcgp_model_set_keep_outputs before the forward pass");
  if (out_sample) HIP_TRY(ctx, hipMemcpyAsync(out_sample, o.sample, n, hipMemcpyDeviceToDevice, ctx->stream));
  if (out_mean) HIP_TRY(ctx, hipMemcpyAsync(out_mean, o.mean, n, hipMemcpyDeviceToDevice, ctx->stream));
  if (out_var) HIP_TRY(ctx, hipMemcpyAsync(out_var, o.var, n, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}

}  // extern "C"
