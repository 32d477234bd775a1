// Internal: the device-resident model (layers + per-step outputs), shared by model.hip (forward) and grad.hip (backward).
#pragma once
#include "layer_impl.h"

struct dcgp_model {
  dcgp_ctx* ctx = nullptr;
  int S = 1;
  double jitter = 1e-3;
  double eps = 1e-3;   // RobustMax epsilon (conv_gp/models.py:67 keeps gpflow's default)
  std::vector<std::unique_ptr<LayerState>> layers;   // conv layers..., head last (once set)
  bool has_head = false;
  bool keep_outputs = false;
  bool keep_state = false;   // the forward leaves K_uf / A1 of every conv layer in HBM (set around the forward of dcgp_elbo_grad)
  bool grad_follows = false; // set around the forward of dcgp_elbo_grad: forward_all hands the parameter-only part of the reverse pass to the side stream
  int gkl_state = 0;         // forward_all of such a step: 0 nothing to hand over, 1 the side stream itself holds the parameter-only chain, 2 it waits for ctx->ev_fork
  int gkl_prep_wait = 0;     // forward_all: grad_kl_early must also wait for ev_prep[bank][0 .. gkl_prep_wait) (chain on a side stream, mark on the main stream)
  int prep_early[8] = {};    // per layer: bit 0 its zero fills, G^T and the factor's lower triangle, bit 1 S_r = G_r G_r^T were enqueued by grad_kl_early
  bool kl_early[8] = {};     // per layer: grad_kl_early enqueued its kl_products for the step in flight (consumed by model_backward)
  int adam_t = 0;        // Adam steps taken on this model's moment buffers (bias correction; dcgp_model_adam_step with t = 0)
  int shard_lo = 0, shard_global = 0;   // this rank's first image and the global batch (dcgp_model_set_shard): device-RNG counters
  int grad_exchange = 0; // multi-rank training step (dcgp_model_train_step_adam): 0 all-reduce of every layer's gradient block + the full update on every rank,
                         // 1 reduce-scatter -> Adam on this rank's shard -> all-gather of the parameters (dcgp_model_set_grad_exchange)
  bool adam_follows = false, grad_scattered = false;   // the step in flight ends in the optimiser update / its gradient blocks hold this rank's shard only
  int grad_shards = 0;   // KL gradient weight 1 / shards; 0 = number of ranks of the ctx's communicator (1 without one)
  // two banks of parameter-only state (LayerState::use_bank): factor groups, events and ELBO scalars follow the bank
  std::vector<FactorGroup> groups[2];
  bool groups_built[2] = {false, false};
  int bank = 0;                                  // bank of the most recent forward
  hipEvent_t ev_sweep[2] = {}, ev_factor[2] = {}, ev_kl[2] = {}, ev_prep[2][8] = {};
  hipEvent_t done_ev[2] = {};                    // not owned: the event that marks the end of the last step on the bank (a result-ring event)
  bool done_valid[2] = {false, false};
  bool events_ok = false;
  // the KL pieces computed inside the tail launch (layers whose sums prep_solve left behind): per bank, set by forward_all
  KlTail kl_tail[2];
  bool kl_in_tail[2] = {false, false};
  bool kl_rode[2] = {false, false};   // ... and the head's one-launch conditional of this step carried them (head_cond.hip): the tail launch has none
  // per-layer outputs of the most recent forward
  struct Out { double *sample = nullptr, *mean = nullptr, *var = nullptr; int rows = 0, width = 0; size_t cap = 0; };
  std::vector<Out> outs;
  double* d_scal = nullptr;   // per bank (64 doubles each): [0]=data, [4 + 4l ..] 4 KL pieces of layer l, [40..43] ELBO, data term, KL, potrf status
  double* d_ve = nullptr; size_t ve_cap = 0;
  double* d_kd = nullptr; size_t kd_cap = 0;
  int id = 0;
  // Parameter-only state kept across steps at unchanged parameters.  Every entry point that writes a parameter (set_param, the optimiser
  // steps, the natural-gradient step) bumps param_version; a step that ran the chain records the version and its bank.  factor_reuse: 0 never, 1 the
  // evaluation entry points (propagate, predict_y) reuse a valid chain, 2 the forward ELBO as well (evaluation sweeps at one parameter state: the
  // reference's LogLikelihoodLogger, conv_gp/utils/log.py:55-68) -- never the default for the ELBO step: the reference's step recomputes it.
  uint64_t param_version = 1, chain_version = 0;
  bool chain_with_kl = false;   // the recorded chain ran for an ELBO step (KL pieces / deferred factor copy in place)
  int factor_reuse = 1;
  uint64_t chain_skips = 0;     // steps that reused it (tests, bench)
  // multi-rank training: steps taken with the sharded update (exchange mode 1) leave every rank with the Adam moments of its own shard only
  uint64_t sharded_steps = 0;
  // throughput mode of the forward (dcgp_elbo_forward_enqueue / _collect): results of up to RING steps in flight land in
  // pinned host slots, one event per slot; tickets are handed out and collected in order
  static constexpr int RING = 4;
  double* h_ring = nullptr;            // RING x 8 pinned doubles: ELBO, data term, KL, potrf status, completion word (ticket + 1)
  double* h_ring_dev = nullptr;        // the same slots as the device addresses them (written by the last kernel of a step)
  hipEvent_t ring_ev[RING] = {};
  uint64_t enq_seq = 0, col_seq = 0;   // tickets handed out / collected

  ~dcgp_model() {
    if (h_ring) hipHostFree(h_ring);
    for (auto& e : ring_ev) if (e) hipEventDestroy(e);
    for (auto& gs : groups) for (auto& gr : gs) gr.release();
    for (int b = 0; b < 2; ++b) {
      if (ev_sweep[b]) hipEventDestroy(ev_sweep[b]);
      if (ev_factor[b]) hipEventDestroy(ev_factor[b]);
      if (ev_kl[b]) hipEventDestroy(ev_kl[b]);
      for (auto& e : ev_prep[b]) if (e) hipEventDestroy(e);
    }
    for (auto& o : outs) { hipFree(o.sample); hipFree(o.mean); hipFree(o.var); }
    hipFree(d_scal); hipFree(d_ve); hipFree(d_kd);
  }
};

// model.hip: the forward ELBO of a minibatch; leaves every layer's outputs in model->outs (keep_outputs) and the
// factorisations / conditional operands in the layers' GpMats.  out_host[0..2] = ELBO, data term, KL.
int elbo_forward_impl(dcgp_model* model, const double* X, const int32_t* y, int N, double scale,
                      const double* const* z_per_layer_host, uint64_t seed, int dedup_layer0, double* out_host,
                      int* info_host);
// the two halves of it: queue the step's launches and the copy of its result into a ring slot / wait for the oldest slot
int elbo_forward_enqueue_impl(dcgp_model* model, const double* X, const int32_t* y, int N, double scale,
                              const double* const* z_per_layer_host, uint64_t seed, int dedup_layer0, uint64_t* ticket,
                              bool pipelined = false);
int elbo_forward_collect_impl(dcgp_model* model, uint64_t ticket, double* out_host, int* info_host);
// grad.hip: reverse pass over the state the forward left behind; fills every layer's gradient buffers
int model_backward(dcgp_model* model, const double* X, const int32_t* y, int N, double scale, int dedup_layer0);
// enqueue == false: 1 if a training step's forward should hand the KL adjoint's products to the side stream, else 0;
// enqueue == true: do it (wait_fork: behind ctx->ev_fork, recorded where the parameter-only chain ended)
int grad_kl_early(dcgp_model* model, bool enqueue, bool wait_fork);
