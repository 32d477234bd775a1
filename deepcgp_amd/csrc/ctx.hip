// Context, memory, error and HIP-event timing plumbing of libdcgp.so.
#include <cctype>
#include <cstdarg>
#include <cstdlib>
#include <cstring>

#include "common.h"

int ctx_fail(dcgp_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

// name -> field of the option block (DcgpOptions, common.h)
namespace {
struct OptSlot { const char* name; long DcgpOptions::*field; };
const OptSlot kOptSlots[] = {
    {"no_fused_layer", &DcgpOptions::no_fused_layer}, {"fused_large", &DcgpOptions::fused_large}, {"fused_shape", &DcgpOptions::fused_shape}, {"fused_split", &DcgpOptions::fused_split},
    {"kl_side", &DcgpOptions::kl_side}, {"kl_no_ride", &DcgpOptions::kl_no_ride}, {"sweep_no_rows", &DcgpOptions::sweep_no_rows}, {"no_fused_bwd", &DcgpOptions::no_fused_bwd}, {"fused_bwd_min_cols", &DcgpOptions::fused_bwd_min_cols},
    {"fused_bwd_frags", &DcgpOptions::fused_bwd_frags}, {"grad_late_kl", &DcgpOptions::grad_late_kl}, {"gemm_tile", &DcgpOptions::gemm_tile}, {"grad_no_keep_k", &DcgpOptions::grad_no_keep_k},
    {"head_unfused", &DcgpOptions::head_unfused}, {"no_side_stream", &DcgpOptions::no_side_stream}, {"cu_partition", &DcgpOptions::cu_partition},
    {"grad_nofork", &DcgpOptions::grad_nofork}, {"chol_one_launch", &DcgpOptions::chol_one_launch},
    {"chol_no_lookahead", &DcgpOptions::chol_no_lookahead}, {"head_no_overlap", &DcgpOptions::head_no_overlap},
    {"no_early_sweep", &DcgpOptions::no_early_sweep}, {"prep_on_chain", &DcgpOptions::prep_on_chain}, {"prep_one_launch", &DcgpOptions::prep_one_launch}, {"no_factor_reuse", &DcgpOptions::no_factor_reuse}, {"sync_event", &DcgpOptions::sync_event}, {"kuf_upw", &DcgpOptions::kuf_upw},
    {"chain_graph", &DcgpOptions::chain_graph}, {"no_rhs_ride", &DcgpOptions::no_rhs_ride}, {"comm_inline", &DcgpOptions::comm_inline}, {"chain_no_iso", &DcgpOptions::chain_no_iso}, {"kuf_no_rep", &DcgpOptions::kuf_no_rep}, {"kuf_stream", &DcgpOptions::kuf_stream},
    {"kuf_wpg", &DcgpOptions::kuf_wpg}, {"kuf_split", &DcgpOptions::kuf_split}, {"head_tail", &DcgpOptions::head_tail},
    {"sweep_occ", &DcgpOptions::sweep_occ}, {"share_kb", &DcgpOptions::share_kb}, {"head_upw", &DcgpOptions::head_upw}, {"no_syrk", &DcgpOptions::no_syrk}, {"grad_dz_main", &DcgpOptions::grad_dz_main},
    {"fused_persist", &DcgpOptions::fused_persist}, {"fused_stagger", &DcgpOptions::fused_stagger}, {"fused_pre", &DcgpOptions::fused_pre}, {"fused_parts", &DcgpOptions::fused_parts},
    {"fused_abl", &DcgpOptions::fused_abl}, {"rb_mixed", &DcgpOptions::rb_mixed},
};
}  // namespace
long* dcgp_option_slot(DcgpOptions* o, const char* name) {
  for (const OptSlot& s : kOptSlots)
    if (strcmp(s.name, name) == 0) return &(o->*s.field);
  return nullptr;
}

// the environment as the switches' initial values: DCGP_<NAME>; a variable that is set but not a number counts as 1
static void options_from_env(DcgpOptions* o) {
  for (const OptSlot& s : kOptSlots) {
    std::string e = "DCGP_";
    for (const char* c = s.name; *c; ++c) e += (char)toupper((unsigned char)*c);
    const char* v = getenv(e.c_str());
    if (!v) continue;
    char* end = nullptr;
    const long x = strtol(v, &end, 10);
    o->*s.field = (end != v) ? x : 1;
  }
}

bool dcgp_poison(const char* name) {
  static const bool poison = getenv("DCGP_POISON_WS") != nullptr;
  static const char* only = getenv("DCGP_POISON_ONLY");
  return poison && (!name || !only || strstr(name, only) != nullptr);
}

void* ws_get(dcgp_ctx* ctx, const std::string& name, size_t bytes) {
  if (bytes == 0) bytes = 16;
  auto it = ctx->ws.find(name);
  if (it != ctx->ws.end() && it->second.second >= bytes) return it->second.first;
  if (it != ctx->ws.end()) {
    // in-flight kernels (either stream) may still use the old buffer
    hipDeviceSynchronize();
    hipFree(it->second.first);
    ctx->ws.erase(it);
  }
  void* p = nullptr;
  size_t cap = (bytes + 255) / 256 * 256;
  if (hipMalloc(&p, cap) != hipSuccess) {
    ctx_fail(ctx, DCGP_ERR_ALLOC, "workspace '%s': hipMalloc(%zu) failed", name.c_str(), cap);
    return nullptr;
  }
  // debugging aid: fresh workspaces filled with NaNs (all-ones bit pattern) so that any read of memory a kernel was
  // supposed to have written first shows up in the results (the GPU suite is run this way once per change of the reverse pass)
  if (dcgp_poison(name.c_str())) {
    hipMemset(p, 0xFF, cap);
    hipDeviceSynchronize();   // the ctx streams are non-blocking: the fill must have landed before any kernel writes the buffer
  }
  ctx->ws[name] = {p, cap};
  return p;
}

ScopedTimer::ScopedTimer(dcgp_ctx* c, const char* name) : ctx(c), on(c->timing) {
  if (on && c->timing_mode == 3) {   // the roofline kernels only, every launch (a sampling loop of its own, outside any timed region)
    if (strcmp(name, "gemm_cond_s3") != 0 && strcmp(name, "kuf") != 0 && strcmp(name, "conv_fused") != 0 && strcmp(name, "head_sweep") != 0) on = false;
    else c->tim[std::string(name) + "#calls"].launches += 1;
  } else if (on && c->timing_mode == 2) {
    if (strcmp(name, "gemm_cond_s3") != 0 && strcmp(name, "kuf") != 0 && strcmp(name, "conv_fused") != 0) on = false;
    // every 7th launch of a roofline kernel (odd: a step with two such launches has both sampled in turn): the two event records are packets in front of and behind the launch (~5 us each of
    // stream time), paid by the very step that is being timed; a sample of the launches gives the same average
    else {
      c->tim[std::string(name) + "#calls"].launches += 1;   // every launch is counted ("<name>#calls"), every 7th timed
      if (c->timing_sample++ % 7 != 0) on = false;
    }
  }
  if (!on) return;
  pe.name = name;
  auto take = [&](hipEvent_t& e) {
    if (!ctx->event_pool.empty()) {
      e = ctx->event_pool.back();
      ctx->event_pool.pop_back();
    } else {
      hipEventCreate(&e);
    }
  };
  take(pe.start);
  take(pe.stop);
  hipEventRecord(pe.start, ctx->stream);
}
ScopedTimer::~ScopedTimer() {
  if (!on) return;
  hipEventRecord(pe.stop, ctx->stream);
  ctx->pending.push_back(pe);
}

void timing_flush(dcgp_ctx* ctx) {
  if (ctx->pending.empty()) return;
  hipStreamSynchronize(ctx->stream);
  hipStreamSynchronize(ctx->stream2);
  hipStreamSynchronize(ctx->stream2b);
  hipStreamSynchronize(ctx->stream_aux);
  if (ctx->stream_m) hipStreamSynchronize(ctx->stream_m);
  if (ctx->stream2_m) hipStreamSynchronize(ctx->stream2_m);
  for (auto& pe : ctx->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pe.start, pe.stop) == hipSuccess) {
      auto& acc = ctx->tim[pe.name];
      acc.launches += 1;
      acc.ms += ms;
    }
    ctx->event_pool.push_back(pe.start);
    ctx->event_pool.push_back(pe.stop);
  }
  ctx->pending.clear();
}

extern "C" {

int dcgp_device_count(int* count) {
  if (!count) return DCGP_ERR_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  *count = n;
  return DCGP_OK;
}

// The side stream carries the latency-bound M x M chain (factorisation, conditional prep, KL) while the main stream
// streams the patch sweep: highest priority, so its few short workgroups are placed ahead of the sweep's backlog.
static hipError_t side_stream_create(hipStream_t* s) {
  int lo = 0, hi = 0;
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  return hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi);
}

int dcgp_ctx_create(int device, dcgp_ctx** out) {
  if (!out) return DCGP_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return DCGP_ERR_HIP;
  if (hipSetDevice(device) != hipSuccess) return DCGP_ERR_HIP;
  dcgp_ctx* c = new dcgp_ctx();
  c->device = device;
  {
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) c->n_cus = ncu;
  }
  options_from_env(&c->opt);
  c->no_side = c->opt.no_side_stream != 0;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      side_stream_create(&c->stream2) != hipSuccess || side_stream_create(&c->stream2b) != hipSuccess ||
      hipStreamCreateWithFlags(&c->stream_aux, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_factor, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_aux, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_kl2, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_kl3, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_g[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_g[1], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_g[2], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_g[3], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_g[4], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_g[5], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_aux2, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_kl, hipEventDisableTiming) != hipSuccess) {
    delete c;
    return DCGP_ERR_HIP;
  }
  for (auto& e : c->ev_prep)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      delete c;
      return DCGP_ERR_HIP;
    }
  {
    // CU-mask bit i lands on XCD i % 8, CU i / 8 (tools/cu_mask_census.hip): bits [0, 240) = 30 CUs of every XCD, [240, 256) = the
    // other two.  OPT-IN (DCGP_CU_PARTITION=1), measured and NOT faster: workgroups are dealt to the four shader engines of an XCD
    // in turn whatever their CU count, so the two engines left with 7 CUs need a fourth round for the 720 strips of the headline
    // layer (800 us instead of 635) -- more than the undisturbed chain buys.  Without it the chain of step i + 1 finds its CUs in
    // the partial last round and the head of step i (1107 vs 994 steps/s).
    hipDeviceProp_t prop;
    if (c->opt.cu_partition && !c->no_side && hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount == 256) {
      uint32_t big[8], small[8];
      for (int w = 0; w < 8; ++w) { big[w] = 0xffffffffu; small[w] = 0u; }
      big[7] = 0x0000ffffu; small[7] = 0xffff0000u;
      if (hipExtStreamCreateWithCUMask(&c->stream_m, 8, big) != hipSuccess) c->stream_m = nullptr;
      if (c->stream_m && hipExtStreamCreateWithCUMask(&c->stream2_m, 8, small) != hipSuccess) {
        hipStreamDestroy(c->stream_m);
        c->stream_m = c->stream2_m = nullptr;
      }
    }
  }
  if (hipHostMalloc((void**)&c->h_scratch, 64 * sizeof(double)) != hipSuccess ||
      hipHostMalloc((void**)&c->h_info, 16 * sizeof(int)) != hipSuccess) {
    hipStreamDestroy(c->stream);
    delete c;
    return DCGP_ERR_ALLOC;
  }
  *out = c;
  return DCGP_OK;
}

int dcgp_ctx_destroy(dcgp_ctx* ctx) {
  if (!ctx) return DCGP_ERR_ARG;
  hipSetDevice(ctx->device);
  hipDeviceSynchronize();
  dcgp_comm_destroy(ctx);
  for (auto& kv : ctx->chain_graphs) hipGraphExecDestroy(kv.second);
  for (auto& kv : ctx->ws) hipFree(kv.second.first);
  for (auto& pe : ctx->pending) {
    hipEventDestroy(pe.start);
    hipEventDestroy(pe.stop);
  }
  for (auto e : ctx->event_pool) hipEventDestroy(e);
  hipHostFree(ctx->h_scratch);
  hipHostFree(ctx->h_info);
  if (ctx->comm_gate) hipHostFree(ctx->comm_gate);
  if (ctx->stream_comm) { hipStreamDestroy(ctx->stream_comm); for (auto e : ctx->ev_comm) if (e) hipEventDestroy(e); }
  hipEventDestroy(ctx->ev_fork);
  hipEventDestroy(ctx->ev_factor);
  hipEventDestroy(ctx->ev_aux);
  hipEventDestroy(ctx->ev_kl2);
  hipEventDestroy(ctx->ev_kl3);
  for (hipEvent_t e : ctx->ev_g) hipEventDestroy(e);
  hipEventDestroy(ctx->ev_aux2);
  for (auto& e : ctx->ev_prep)
    if (e) hipEventDestroy(e);
  hipEventDestroy(ctx->ev_kl);
  if (ctx->stream_m) hipStreamDestroy(ctx->stream_m);
  if (ctx->stream2_m) hipStreamDestroy(ctx->stream2_m);
  hipStreamDestroy(ctx->stream2);
  hipStreamDestroy(ctx->stream2b);
  hipStreamDestroy(ctx->stream_aux);
  hipStreamDestroy(ctx->stream);
  delete ctx;
  return DCGP_OK;
}

const char* dcgp_last_error(dcgp_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int dcgp_malloc(dcgp_ctx* ctx, size_t bytes, void** dptr) {
  if (!ctx || !dptr) return DCGP_ERR_ARG;
  *dptr = nullptr;
  if (bytes == 0) bytes = 16;
  if (hipMalloc(dptr, bytes) != hipSuccess) return ctx_fail(ctx, DCGP_ERR_ALLOC, "hipMalloc(%zu) failed", bytes);
  return DCGP_OK;
}
int dcgp_free(dcgp_ctx* ctx, void* dptr) {
  if (!ctx) return DCGP_ERR_ARG;
  if (!dptr) return DCGP_OK;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipFree(dptr));
  return DCGP_OK;
}
int dcgp_h2d(dcgp_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || (bytes && (!dst || !src))) return DCGP_ERR_ARG;
  if (!bytes) return DCGP_OK;
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}
int dcgp_d2h(dcgp_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || (bytes && (!dst || !src))) return DCGP_ERR_ARG;
  if (!bytes) return DCGP_OK;
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}
int dcgp_memset(dcgp_ctx* ctx, void* dptr, int value, size_t bytes) {
  if (!ctx || (bytes && !dptr)) return DCGP_ERR_ARG;
  if (!bytes) return DCGP_OK;
  HIP_TRY(ctx, hipMemsetAsync(dptr, value, bytes, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}
int dcgp_sync(dcgp_ctx* ctx) {
  if (!ctx) return DCGP_ERR_ARG;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}

int dcgp_workspace_query(dcgp_ctx* ctx, size_t* bytes_out, int* count_out) {
  if (!ctx) return DCGP_ERR_ARG;
  size_t total = 0;
  for (auto& kv : ctx->ws) total += kv.second.second;
  if (bytes_out) *bytes_out = total;
  if (count_out) *count_out = (int)ctx->ws.size();
  return DCGP_OK;
}

int dcgp_ctx_set_option(dcgp_ctx* ctx, const char* name, long value) {
  if (!ctx || !name) return DCGP_ERR_ARG;
  long* slot = dcgp_option_slot(&ctx->opt, name);
  if (!slot) return ctx_fail(ctx, DCGP_ERR_ARG, "unknown option '%s'", name);
  if (strcmp(name, "cu_partition") == 0) return ctx_fail(ctx, DCGP_ERR_ARG, "option 'cu_partition' is read at dcgp_ctx_create only (DCGP_CU_PARTITION)");
  *slot = value;
  if (strcmp(name, "no_side_stream") == 0) {
    hipDeviceSynchronize();   // steps in flight were enqueued under the other stream layout
    ctx->no_side = value != 0;
  }
  return DCGP_OK;
}
int dcgp_ctx_get_option(dcgp_ctx* ctx, const char* name, long* value_out) {
  if (!ctx || !name || !value_out) return DCGP_ERR_ARG;
  long* slot = dcgp_option_slot(&ctx->opt, name);
  if (!slot) return ctx_fail(ctx, DCGP_ERR_ARG, "unknown option '%s'", name);
  *value_out = *slot;
  return DCGP_OK;
}

int dcgp_timing_enable(dcgp_ctx* ctx, int on) {
  if (!ctx) return DCGP_ERR_ARG;
  timing_flush(ctx);
  ctx->timing = on != 0;
  ctx->timing_mode = on;
  // pre-create events so that the timed region never pays hipEventCreate
  while (ctx->timing && ctx->event_pool.size() < 1536) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) break;
    ctx->event_pool.push_back(e);
  }
  return DCGP_OK;
}
int dcgp_timing_reset(dcgp_ctx* ctx) {
  if (!ctx) return DCGP_ERR_ARG;
  timing_flush(ctx);
  ctx->tim.clear();
  ctx->timing_sample = 0;   // the first launch after a reset is one of the sampled ones
  return DCGP_OK;
}
int dcgp_timing_query(dcgp_ctx* ctx, const char* name, int* launches, double* total_ms) {
  if (!ctx || !name) return DCGP_ERR_ARG;
  timing_flush(ctx);
  auto it = ctx->tim.find(name);
  if (launches) *launches = it == ctx->tim.end() ? 0 : it->second.launches;
  if (total_ms) *total_ms = it == ctx->tim.end() ? 0.0 : it->second.ms;
  return DCGP_OK;
}
int dcgp_timing_names(dcgp_ctx* ctx, char* buf, size_t buflen) {
  if (!ctx || !buf || buflen == 0) return DCGP_ERR_ARG;
  timing_flush(ctx);
  std::string s;
  for (auto& kv : ctx->tim) {
    if (!s.empty()) s += ";";
    s += kv.first;
  }
  strncpy(buf, s.c_str(), buflen - 1);
  buf[buflen - 1] = 0;
  return DCGP_OK;
}

}  // extern "C"

long col_ld(long columns) {
#ifdef DCGP_EXPERIMENTS
  static long skew = -1;   // timing build only
  if (skew < 0) { const char* e = getenv("DCGP_LD_SKEW"); skew = e ? atol(e) : 0; if (skew < 0 || (skew & 1)) skew = 0; }
  return round_up_l(columns, 128) + skew;
#else
  return round_up_l(columns, 128);
#endif
}


// 2^(j / 256), j < 256, for exp2_tab_n (common.h): one workspace per ctx, filled on first use
const double* exp2_table(dcgp_ctx* ctx) {
  auto it = ctx->ws.find("exp2_tab256");
  if (it != ctx->ws.end()) return (const double*)it->second.first;
  double* d = (double*)ws_get(ctx, "exp2_tab256", 256 * sizeof(double));
  if (!d) return nullptr;
  double h[256];
  for (int j = 0; j < 256; ++j) h[j] = exp2((double)j / 256.0);
  hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream);
  hipStreamSynchronize(ctx->stream);
  return d;
}
