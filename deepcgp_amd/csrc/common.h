// Internal declarations shared by the libdcgp.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/dcgp.h"

typedef double d4 __attribute__((ext_vector_type(4)));

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline long round_up_l(long x, long m) { return (x + m - 1) / m * m; }
// Leading dimension of the [M x columns] matrices (K_uf, A1, dT, ...): a multiple of 128 columns plus a skew.  Without
// the skew the row stride is a multiple of 4 KB at the bench sizes and every row of a k tile lands on the same HBM
// channel (the backward products read 128 B of each of 128 rows per step).
long col_ld(long columns);

struct TimingAcc {
  int launches = 0;
  double ms = 0.0;
};
struct PendingEvent {
  std::string name;
  hipEvent_t start, stop;
};

// A/B and debugging switches of a ctx.  Filled ONCE, at dcgp_ctx_create, from the environment variable of the same name in upper case with
// a DCGP_ prefix (so that shell-level experiments keep working), and from then on changed only through dcgp_ctx_set_option: nothing
// on the step path calls getenv.  Every switch selects between two routes that are both tested, or turns an aid on; none is needed
// for normal use (DESIGN.md 6a lists them).
struct DcgpOptions {
  long no_fused_layer = 0;       // conv layers by the sweep + GEMM route even where the one-launch layer kernel covers them
  long fused_large = 0;          // the one-launch layer kernel also for M > 256 (narrower strips; measured slower there)
  long fused_shape = -1;         // force one strip shape of the one-launch layer kernel (-1: chosen from the layer)
  long fused_split = -1;         // strips of the layer kernel's partial last round shared by this many workgroups (-1: chosen, 0 / 1: never)
  long fused_persist = -1;       // the one-launch layer kernel as a persistent launch: one workgroup per slot of the chip walking its strips (-1: chosen; 0: off;
                                 // 1: on, strips dealt by a device counter; 2: on, dealt by a fixed stride)
  long fused_pre = -1;           // persistent layer kernel: prologues (sweep + first product) of later strips run by the spare workgroups of a partial first round
                                 // (-1: chosen by a simulated deal; 0: never; k > 0: up to k per spare workgroup)
  long fused_parts = -1;         // layer kernel on few strips (< 1.5 rounds of the CUs): every strip's prologue by one item, its outputs by q parts that fetch A1
                                 // (-1, 0: never -- measured slower at every shard, conv_fused.hip: plan_parts; q > 0: this many parts; -2: q by the simulated deal)
  long fused_stagger = -1;       // persistent layer kernel: microseconds the second workgroup of a CU holds back (-1: default; 0: none)
  long kl_side = 0;              // KL terms by their own launches on the side stream instead of inside the tail launch
  long sweep_no_rows = 0;        // long-patch sweeps (5 x 5 x 10 patches) on the generic streamed loop instead of the patch-row form (A/B)
  long kl_no_ride = 0;           // KL pieces in the tail launch even where the head's one-launch conditional could carry them (A/B)
  long no_fused_bwd = 0;         // reverse pass of the conditional by GEMM launches instead of the strip kernel
  long fused_bwd_min_cols = -1;  // strip kernel of the reverse pass from this many columns on (-1: default 4096)
  long fused_bwd_frags = 0;      // strip width of that kernel in 16-column fragments (0: chosen from the column count; 4, 2, 1)
  long gemm_tile = 0;            // gemm_gen: force the 32 / 64 / 128 output tile (0: chosen from the shape)
  long grad_no_keep_k = 0;       // training step: the head's patch responses evaluated again by the reverse pass instead of kept by the forward sweep
  long grad_late_kl = 0;         // reverse pass: the KL adjoint at the end of each layer instead of beside the forward pass
  long head_unfused = 0;         // the head's conditional by the shared GEMM route instead of its one launch
  long no_side_stream = 0;       // everything on one stream (counter collection: the profiler serialises dispatches)
  long cu_partition = 0;         // CU-masked main / side streams for steps in flight (ctx create only)
  long grad_nofork = 0;          // reverse pass without the side-stream fork of the M x M adjoint chains
  long chol_one_launch = 0;      // the persistent one-launch factorisation chain
  long chol_no_lookahead = 0;    // panel launches without the look-ahead workgroup
  long head_no_overlap = 0;      // head-first model: the factorisation chain in front of the sweep instead of beside it
  long no_factor_reuse = 0;      // evaluation entry points run the parameter-only chain every time, also at unchanged parameters (A/B)
  long prep_on_chain = 0;        // head-first model: the operand preparation on the chain's stream instead of in front of the sweep on the main stream (A/B)
  long prep_one_launch = 0;      // head-first model, synchronous step: the operand preparation as ONE launch on the main stream with an event to the chain's stream
                                 // instead of one launch per stream (A/B)
  long no_early_sweep = 0;       // the first layer's sweep enqueued behind the chain instead of in front of it
  long sync_event = 0;           // wait for the step's event instead of polling its completion word
  long chain_graph = 0;          // the factorisation chain's panel launches replayed from a captured HIP graph (measured slower: see chol_fused.hip)
  long chain_no_iso = 0;         // chain launches that carry right-hand sides: no XCD isolation of the look-ahead workgroups (A/B)
  long comm_inline = 0;          // multi-rank steps in flight: the data term's all-reduce in the main stream instead of the comm stream (A/B)
  long no_rhs_ride = 0;          // G / alpha by their own launch behind the chain (prep_solve) instead of riding its panel launches
  long kuf_upw = 0;              // units per wave of the storing sweep (0: chosen by head_units_plan)
  long kuf_split = -1;           // storing sweep with replicas: column-fragment ranges per row fragment (-1: chosen; 0: one)
  long kuf_wpg = 0;              // storing sweep: waves per workgroup (0: chosen by head_units_plan; 1, 2, 4)
  long kuf_stream = 0;           // storing sweep: the streamed-operand kernel also for the patch lengths with a register-resident one
  long kuf_no_rep = 0;           // storing sweep: evaluate every row, also where rows show the same image (tiled batch)
  long grad_dz_main = 0;         // reverse pass of the head: the patch adjoint's dZ product on the main stream instead of the tail stream (A/B)
  long no_syrk = 0;              // reverse pass: W_r = 2 A1 diag(gv_r) A1^T through the general GEMM instead of its own kernel (A/B)
  long head_upw = 0;             // reducing sweep: Kzx row units per wave (0: chosen by head_units_plan)
  long share_kb = 0;             // patch sweeps beside the factorisation chain: LDS claimed per workgroup in KB (0: default)
  long sweep_occ = -1;           // patch sweeps: waves per SIMD a launch is held to so that its rounds come out whole (-1: chosen; 0: off)
  long head_tail = -1;           // head_units: balance of the launch tail (-1: default; see head_units_plan)
  long fused_abl = 0, rb_mixed = 0;   // timing builds only (make EXPERIMENTS=1)
};
long* dcgp_option_slot(DcgpOptions* o, const char* name);   // nullptr: no such option (ctx.hip)
// debugging aid, process-wide, DCGP_POISON_WS read once: fresh device allocations of the library are filled with NaNs;
// `name` != nullptr additionally applies DCGP_POISON_ONLY (only workspaces whose name contains that string)
bool dcgp_poison(const char* name = nullptr);

struct ChainEpoch { unsigned epoch = 0; int T = 0, np = 0, batch = 0; };   // launches so far of the one-launch factorisation chain on a sync area

struct KlTail;   // layer.h
struct dcgp_ctx {
  int device = 0;
  // KL pieces riding the head's one-launch conditional (model.hip sets kl_ride in front of the head layer of an ELBO step; head_cond.hip carries them as
  // extra workgroups of its launch, clears kl_ride and sets kl_rode; a head on another route leaves them to the tail launch)
  const KlTail* kl_ride = nullptr;
  double* kl_ride_scal = nullptr;
  bool kl_rode = false;
  int n_cus = 256;   // compute units of the device (set at dcgp_ctx_create)
  hipStream_t stream = nullptr;    // the stream launches go to (temporarily swapped to stream2 for the side branch)
  hipStream_t stream2 = nullptr;   // side stream: factorisation chain + KL terms
  // CU partition for steps in flight (dcgp_elbo_forward_enqueue): the data path of step i on 30 CUs of every XCD, the
  // parameter-only chain of step i + 1 on the other 2 (hipExtStreamCreateWithCUMask; nullptr when the device is not 8 x 32 CUs)
  hipStream_t stream2b = nullptr;  // second side stream: the chain of a step enqueued while the previous one is in flight (bank 1)
  hipStream_t stream_aux = nullptr;   // short excursions beside the main stream inside a layer (the head's Kdiag)
  hipStream_t stream_comm = nullptr;  // created on first use: the data term's all-reduce + ELBO assembly of a FORWARD step kept in flight (dcgp_elbo_forward_enqueue
                                      // with no reverse pass behind it), so that the next step's kernels on the main stream do not queue behind the collective.
                                      // A training step keeps its collectives on the main stream: one communicator's collectives stay on one stream (model.hip)
  hipEvent_t ev_comm[4] = {};         // main -> comm stream, one per result-ring slot
  int* comm_gate = nullptr;           // debugging aid (dcgp_debug_comm_gate): pinned word; != null: the comm stream's work waits for it to become non-zero
  hipStream_t stream_m = nullptr, stream2_m = nullptr;
  hipStream_t last_main = nullptr;             // main stream of the most recent forward step ...
  hipEvent_t ev_last = nullptr;                // ... and the event marking the end of that step on it (not owned; a step on the other main stream waits for it)
  bool ev_last_valid = false;
  hipEvent_t ev_fork = nullptr, ev_factor = nullptr, ev_kl = nullptr;
  hipEvent_t ev_prep[8] = {};   // per layer: G / alpha of layer l are ready (side stream)
  DcgpOptions opt;                 // A/B switches: environment at dcgp_ctx_create, then dcgp_ctx_set_option only
  bool no_side = false;            // opt.no_side_stream: everything on the main stream (A/B switch; counter-collection runs, where
                                   // the profiler serialises dispatches and cross-stream waits can deadlock it)
  // training step, the part of the reverse pass that runs beside the forward pass on stream_aux (grad_kl_early): ev_kl2 behind the zero fills and
  // parameter-only operands, ev_kl3 behind the KL adjoint's products
  hipEvent_t ev_kl2 = nullptr, ev_kl3 = nullptr;
  // reverse pass of a layer (grad.hip, Lanes): [0] main -> chain: the conditional's operands are there, [1] main -> chain: dK_uf and dq_mu,
  // [2] chain -> tail: S = d ELBO / dK_uu, [3] main -> tail: the patch adjoint's part of dZ and the partial sums, [4] chain -> tail: dq_sqrt
  hipEvent_t ev_g[6] = {};
  hipEvent_t ev_aux = nullptr, ev_aux2 = nullptr;  // fork / join of a short side-stream excursion inside a layer
  std::string err;
  std::map<std::string, hipGraphExec_t> chain_graphs;   // captured panel-launch sequences of the factorisation chain, by argument set (chol_fused.hip)
  std::map<std::string, unsigned> fused_pre_epochs;   // per hand-over area of the layer kernel's prologues ahead (conv_fused.hip): launches so far
  std::map<std::string, ChainEpoch> chain_epochs;   // per sync workspace of chol_persist_kernel (chol_fused.hip)
  bool chain_alone = true;   // the factorisation chain about to run has the chip to itself (forward_all: synchronous step, chain on the main stream): the
                             // look-ahead workgroups are then confined to one XCD.  Beside a patch sweep or the previous step's layer kernel that
                             // confinement costs more than it saves (head-only model 4250 -> 3190 steps/s without the riding right-hand sides)
  bool chain_ride_ok = true; // right-hand sides may ride the chain about to run: not beside a patch sweep (a head-first model: 4250 -> 4200 steps/s).
                             // A property of the model, not of the step's mode: synchronous and in-flight steps take the same route (bit-identical)
  std::string ws_tag;   // suffix of the chain's / KL terms' scratch names: steps in flight on the two banks must not share them
  // named, grow-only device workspaces owned by the ctx
  std::map<std::string, std::pair<void*, size_t>> ws;
  // timing
  bool timing = false;
  int timing_mode = 0;   // 1: every bracketed kernel family; 2: only the roofline kernels (conv_fused, gemm_cond_s3, kuf), every 7th launch
  unsigned timing_sample = 0;
  std::map<std::string, TimingAcc> tim;
  std::vector<PendingEvent> pending;
  std::vector<hipEvent_t> event_pool;
  // pinned host scratch for small result read-backs
  double* h_scratch = nullptr;   // 64 doubles
  int* h_info = nullptr;         // 16 ints
  long long* fused_trace = nullptr;   // debugging aid (dcgp_debug_set_fused_trace): phase stamps of the one-launch layer kernel
  std::string sweep_trace_family;     // ... of the launches of this timer family only ("kuf", "kuf_long", "head_sweep")
  long long* sweep_trace = nullptr; long sweep_trace_wgs = 0;   // debugging aid (dcgp_debug_set_sweep_trace): stamps of every workgroup of the patch sweeps
  // RCCL
  void* comm = nullptr;
  int nranks = 1, rank = 0;
};

int ctx_fail(dcgp_ctx* ctx, int code, const char* fmt, ...);
void* ws_get(dcgp_ctx* ctx, const std::string& name, size_t bytes);   // nullptr on failure
void timing_flush(dcgp_ctx* ctx);

#define HIP_TRY(ctx, call)                                                               \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess)                                                                \
      return ctx_fail((ctx), DCGP_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                      __FILE__, __LINE__);                                               \
  } while (0)

#define DCGP_TRY(call)        \
  do {                        \
    int s_ = (call);          \
    if (s_ != DCGP_OK) return s_; \
  } while (0)

#define LAUNCH_CHECK(ctx)                                                                \
  do {                                                                                   \
    hipError_t e_ = hipGetLastError();                                                   \
    if (e_ != hipSuccess)                                                                \
      return ctx_fail((ctx), DCGP_ERR_HIP, "kernel launch failed: %s (%s:%d)",           \
                      hipGetErrorString(e_), __FILE__, __LINE__);                        \
  } while (0)

// RAII bracket: records start/stop events around the launches issued in its scope when timing is on.
struct ScopedTimer {
  dcgp_ctx* ctx;
  bool on;
  PendingEvent pe;
  ScopedTimer(dcgp_ctx* c, const char* name);
  ~ScopedTimer();
};

// ------------------------------------------------------------------------------------------------
// internal device-side building blocks (defined in the .hip files)
// ------------------------------------------------------------------------------------------------

// C[i][j] = sum_k Wt[k][i] * B[k][j]  (both operands k-major), optional triangular structure of W,
// optional store of C, optional fused per-column sum of squares (partial per row block).
struct GemmArgs {
  const double* Wt = nullptr; long wBatch = 0; int ldw = 0;
  const double* B = nullptr;  long bBatch = 0; int ldb = 0;
  double* C = nullptr;        long cBatch = 0; int ldc = 0;
  double* colsq = nullptr;    long sBatch = 0; long sRowBlk = 0;   // colsq[batch*sBatch + rb*sRowBlk + j]
  const double* cscale = nullptr; long csCol = 0, csBatch = 0; double calpha = 1.0;   // stored C[i][j] *= calpha * cscale[batch*csBatch + j*csCol] (backward pass)
  int Mi = 0, Mk = 0, Kc = 0;  // rows of C, contraction length, columns
  int nW = 1, nB = 1;          // batch = nW * nB, bz -> (iw = bz / nB, ib = bz % nB)
  int tri = 0;                 // 0 dense, 1: W lower (k <= i), 2: W upper (k >= i)
  int b_lower = 0;             // B[k][j] == 0 for k < j (skip those k tiles)
  int rb_major = 0;            // set by the launcher: workgroup order (see the kernel's decode)
};
int gemm_tn(dcgp_ctx* ctx, const GemmArgs& a, int* n_row_blocks_out);
int gemm_row_block(int Mi, int Kc, int batch);   // BM the dispatcher picks (callers size partial-sum buffers with it)

// exp() for the kernel sweeps.  The device library's exp is the same range reduction + degree-13 polynomial, but hipcc
// turns every Horner step into v_fmac with the coefficient re-materialised into a fresh VGPR pair (two v_mov each):
// ~45 VALU instructions per value where 19 do the work, and the sweeps' epilogues are VALU-bound on it.  Here each
// step is a v_fma_f64 whose addend is the coefficient in an SGPR pair (one scalar operand per VOP3 is free).
__device__ __forceinline__ double exp_sweep(double x) {
  const double kf = rint(x * 1.4426950408889634074);          // x / ln 2
  double r = fma(-kf, 6.93147180369123816490e-01, x);        // Cody-Waite: ln 2 in two pieces
  r = fma(-kf, 1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;                          // 1/13!  (|r| <= ln2/2: remainder r^14/14! < 5e-18)
#define DCGP_EXP_STEP(c) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(p), "v"(r), "s"((double)(c)))
  DCGP_EXP_STEP(2.08767569878681e-09);      // 1/12!
  DCGP_EXP_STEP(2.505210838544172e-08);     // 1/11!
  DCGP_EXP_STEP(2.755731922398589e-07);     // 1/10!
  DCGP_EXP_STEP(2.7557319223985893e-06);    // 1/9!
  DCGP_EXP_STEP(2.48015873015873e-05);      // 1/8!
  DCGP_EXP_STEP(1.984126984126984e-04);     // 1/7!
  DCGP_EXP_STEP(1.388888888888889e-03);     // 1/6!
  DCGP_EXP_STEP(8.333333333333333e-03);     // 1/5!
  DCGP_EXP_STEP(4.1666666666666664e-02);    // 1/4!
  DCGP_EXP_STEP(1.6666666666666666e-01);    // 1/3!
  DCGP_EXP_STEP(0.5);
  DCGP_EXP_STEP(1.0);
  DCGP_EXP_STEP(1.0);
#undef DCGP_EXP_STEP
  // exact scaling; underflows gradually to 0 below ~ -745 (the guard covers -inf, where r would be NaN); NaN stays NaN
  return x < -746.0 ? 0.0 : ldexp(p, (int)kf);
}

// ---- 2^t for the RBF sweeps whose MFMA accumulator is the base-2 exponent of the kernel value (head_units.hip, conv_fused.hip) ----
constexpr double kExp2Magic = 6755399441055744.0;   // 1.5 * 2^52

// 2^t for N values, interleaved step by step.  t is clamped at -1100 (ldexp then returns an exact zero; -inf would
// otherwise leave NaN).  Relative error ~1.3e-16 of the polynomial evaluation plus what the argument carries.
template <int N>
__device__ __forceinline__ void exp2_n(double (&t)[N]) {
  double u[N], r[N], p[N];
  const double lo = -1100.0, magic = kExp2Magic;
#pragma unroll
  for (int i = 0; i < N; ++i) asm("v_max_f64 %0, %1, %2" : "=v"(t[i]) : "v"(t[i]), "s"(lo));
#pragma unroll
  for (int i = 0; i < N; ++i) asm("v_add_f64 %0, %1, %2" : "=v"(u[i]) : "v"(t[i]), "s"(magic));
#pragma unroll
  for (int i = 0; i < N; ++i) asm("v_add_f64 %0, %1, -%2" : "=v"(r[i]) : "v"(u[i]), "s"(magic));   // rint(t)
#pragma unroll
  for (int i = 0; i < N; ++i) asm("v_add_f64 %0, %1, -%2" : "=v"(r[i]) : "v"(t[i]), "v"(r[i]));     // |r| <= 1/2
  double c11 = 4.4549605981865186e-10;
  asm("" : "+v"(c11));   // one VGPR pair for the kernel's lifetime (a scalar operand is already taken by the addend)
#pragma unroll
  for (int i = 0; i < N; ++i) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p[i]) : "v"(c11), "v"(r[i]), "s"(7.0725859492692234e-09));
#define DCGP_E2(c)                          \
  _Pragma("unroll") for (int i = 0; i < N; ++i) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p[i]) : "v"(p[i]), "v"(r[i]), "s"((double)(c)))
  DCGP_E2(1.0178062445845774e-07);
  DCGP_E2(1.3215442587921689e-06);
  DCGP_E2(1.5252733829836119e-05);
  DCGP_E2(0.0001540353044173605);
  DCGP_E2(0.0013333558146416936);
  DCGP_E2(0.0096181291076068882);
  DCGP_E2(0.055504108664821597);
  DCGP_E2(0.24022650695910097);
  DCGP_E2(0.69314718055994529);
  DCGP_E2(1.0);
#undef DCGP_E2
#pragma unroll
  for (int i = 0; i < N; ++i) t[i] = __builtin_amdgcn_ldexp(p[i], __double2loint(u[i]));
}

// 2^t through a table of 2^(j / 256) (256 doubles in LDS, gauss... see exp2_table()): t = n + j / 256 + r with |r| <= 2^-9 by the same magic-number split (the
// low mantissa word of t + 1.5 * 2^44 is round(256 t)), 2^r - 1 by a degree-4 polynomial (|remainder| < 4e-17), 2^t = ldexp(T_j + T_j (2^r - 1), n):
// 13 VALU instructions and one LDS read per value against the 16 of exp2_n -- the patch sweeps of short patches are bound by this epilogue (68 of a tile's
// ~90 VALU instructions beside 7 MFMAs, and VALU instructions issue in the fp64 MFMA's place on this part).  Relative error <= ~1.5 ulp.
constexpr double kExp2Magic8 = 26388279066624.0;   // 1.5 * 2^44
template <int N>
__device__ __forceinline__ void exp2_tab_n(double (&t)[N], const double* tab /* LDS, [256] */) {
  double u[N], r[N], h[N], T[N];
  int k[N];
  const double lo = -1100.0, magic = kExp2Magic8;
#pragma unroll
  for (int i = 0; i < N; ++i) asm("v_max_f64 %0, %1, %2" : "=v"(t[i]) : "v"(t[i]), "s"(lo));
#pragma unroll
  for (int i = 0; i < N; ++i) asm("v_add_f64 %0, %1, %2" : "=v"(u[i]) : "v"(t[i]), "s"(magic));
#pragma unroll
  for (int i = 0; i < N; ++i) k[i] = __double2loint(u[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) T[i] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(tab) + ((k[i] << 3) & 0x7f8));
#pragma unroll
  for (int i = 0; i < N; ++i) asm("v_add_f64 %0, %1, -%2" : "=v"(r[i]) : "v"(u[i]), "s"(magic));   // rint(256 t) / 256
#pragma unroll
  for (int i = 0; i < N; ++i) asm("v_add_f64 %0, %1, -%2" : "=v"(r[i]) : "v"(t[i]), "v"(r[i]));     // |r| <= 2^-9
  double c4 = 0.0096181291076284772;   // ln(2)^4 / 24
  asm("" : "+v"(c4));
#pragma unroll
  for (int i = 0; i < N; ++i) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(h[i]) : "v"(c4), "v"(r[i]), "s"(0.055504108664821580));   // ln(2)^3 / 6
#pragma unroll
  for (int i = 0; i < N; ++i) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(h[i]) : "v"(h[i]), "v"(r[i]), "s"(0.24022650695910072));   // ln(2)^2 / 2
#pragma unroll
  for (int i = 0; i < N; ++i) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(h[i]) : "v"(h[i]), "v"(r[i]), "s"(0.69314718055994531));    // ln(2)
#pragma unroll
  for (int i = 0; i < N; ++i) h[i] *= r[i];
#pragma unroll
  for (int i = 0; i < N; ++i) h[i] = fma(T[i], h[i], T[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) t[i] = __builtin_amdgcn_ldexp(h[i], k[i] >> 8);
}
const double* exp2_table(struct dcgp_ctx* ctx);   // device: 2^(j / 256), j < 256 (ctx.hip)

// The base kernel of a layer, evaluated from (x.z, |x|^2, |z|^2):
//   type 0  gpflow RBF:           variance * exp(-(|x|^2 + |z|^2 - 2 x.z) / (2 l^2))     p1 = 1 / l^2 (square_dist form, no clamp)
//   type 1  gpflow ArcCosine(0):  variance * (pi - theta) / pi,  theta = acos(1e-15 + (1 - 2e-15) cos),
//           cos = (w x.z + b) / sqrt((w |x|^2 + b)(w |z|^2 + b)), argument clamped to <= 1    p1 = w (weight variance), p2 = b (bias variance)
//   (conv_gp/models.py:113-121: --base-kernel rbf | acos; Kdiag = variance for both)
struct BaseKernel {
  int type = 0;
  double variance = 1.0, p1 = 1.0, p2 = 0.0;
  template <int T>   // the hot sweeps are instantiated per type: the acos code must not cost the RBF path registers
  __device__ __forceinline__ double eval_as(double dot, double n1, double n2) const {
    if (T == 0) return variance * exp_sweep(-0.5 * (n1 + n2 - 2.0 * dot) * p1);
    const double c = (p1 * dot + p2) / sqrt((p1 * n1 + p2) * (p1 * n2 + p2));
    // fmin: on a diagonal entry cos can round a few ulp above 1 (dot and norms are accumulated in different orders)
    // and overshoot the reference's 1e-15 guard -- acos() would return NaN there, as the reference formula does
    return variance * (1.0 - acos(fmin(1e-15 + (1.0 - 2e-15) * c, 1.0)) * 0.31830988618379067154);
  }
  // N values (io: x.z in, kernel value out).  Tried for the RBF and measured no better or worse than this plain loop
  // (A/B builds, 1 x MI355X): the N exps with every Horner step issued for all of them before the next one (the dependent
  // chains hide behind each other anyway: the sweeps run four waves per SIMD) -- equal; a 64-entry table of 2^(j/64) with a
  // degree-5 polynomial (13 instead of ~25 double-precision operations per value) -- equal on the conv layer + head model,
  // 20 % SLOWER on the head-only model (1664 vs 2105 steps/s: 100 M per-lane table loads per step through the vector L1).
  template <int T, int N>
  __device__ __forceinline__ void eval_n(double (&io)[N], const double (&n1)[N], const double (&n2)[N]) const {
#pragma unroll
    for (int i = 0; i < N; ++i) io[i] = eval_as<T>(io[i], n1[i], n2[i]);
  }
  __device__ __forceinline__ double eval(double dot, double n1, double n2) const {
    return type == 0 ? eval_as<0>(dot, n1, n2) : eval_as<1>(dot, n1, n2);
  }
};

// patch-RBF sweep (kuf / head Kzx)
struct PatchRbfArgs {
  const double* X = nullptr;   // [n_mod, H, W, C]; image of column block n is X[(n0 + n) % n_mod]
  int n_mod = 0, n0 = 0;       // n0: first image of a chunk of a larger batch (the output is indexed by the local n)
  int N = 0, H = 0, W = 0, C = 0, f = 0, s = 0, Ho = 0, Wo = 0, P = 0, L = 0;
  const double* ZT = nullptr;  // [Lp, Mp] k-major, zero padded
  const double* zn = nullptr;  // [Mp] |z|^2 (unscaled)
  int M = 0, Mp = 0, Lp = 0;
  BaseKernel bk;
  const double* in_scale = nullptr;   // [H*W*C] or nullptr: the image is multiplied elementwise while it is staged (ARD head)
  // write mode: out[m*sM + n*sN + p*sP]
  double* out = nullptr; long sM = 0, sN = 0, sP = 0;
  // reduce mode (head Kzx): out[m*sM + n*sN] = scale * sum_p w[p] k
  const double* w = nullptr; double scale = 1; int reduce = 0;
  // the sweep runs beside the latency-bound side-stream chain: hold it to two workgroups per CU (see patch_rbf)
  int share_cu = 0;
};
int patch_rbf(dcgp_ctx* ctx, const PatchRbfArgs& a, const char* timer_name);
int head_sweep(dcgp_ctx* ctx, const PatchRbfArgs& a, const double* w, const double** kd_partial, int* n_pairs, double* kd_scale);
int head_kdiag(dcgp_ctx* ctx, const double* X, int N, int n_mod, int H, int W, int C, int f, int s, BaseKernel bk,
               const double* w, double* out_N);

// head_units.hip: ConvKernel.Kzx (weighted patch sum) and ConvKernel.Kdiag (partial sums per image) in one launch.
// RBF base kernel; the operands carry the kernel's scales: ZS = sqrt(c) Z^T with c = log2(e) / lengthscale^2, rows L, L + 1 =
// (-c |z|^2 / 2 + log2 variance, 1), zero behind (prepare_all writes it beside Z^T).
// A launch of the unit sweep is a list of segments in dispatch order: workgroups [wg0, next wg0) cover images [img0, ...) with wpi
// workgroups per image.  kind 0: Kzx row units (one per 16-row fragment of Z); 1: Kdiag chunks of T tiles of the image's patch Gram
// matrix (C chunks per image); 2: the storing form's row units.  Long units go first, the Kdiag chunks shrink towards the end of the
// launch so that its tail is a short unit, not a long one (head_units_plan).
struct HuSeg { int wg0 = 0, img0 = 0, wpi = 1, kind = 0, T = 0, C = 0, upw = 1; };   // upw: units a wave runs behind one set-up
struct HeadUnitsArgs {
  const double* X = nullptr; int n_mod = 0, N = 0, n0 = 0; // image of row n is X[(n0 + n) % n_mod] (n0: first image of a chunk; outputs are indexed by the local n)
  int H = 0, W = 0, C = 0, f = 0, s = 0, Wo = 0, P = 0, L = 0, Lq = 0, HWC = 0;
  const double* ZS = nullptr; int M = 0, Mp = 0;           // [Lq][Mp]
  double csq = 1.0, log2var = 0.0;                         // sqrt(c); log2(variance)
  const double* w = nullptr;                               // [P] patch weights
  double* kzx = nullptr; long ldk = 0; double kzx_scale = 1.0;   // kzx[m * ldk + n] = kzx_scale * sum_p w_p k(z_m, x_np), rows M..Mp-1 zeroed
  double* kuf = nullptr; long sM = 0, sN = 0, sP = 0;      // the K_uf sweep instead: kuf[m * sM + n * sN + p * sP] = k(z_m, x_np) (rows M..kzx_rows-1 zeroed)
  int kzx_rows = 0;                                        // rows of kzx / kuf that exist (0: all Mp)
  // reducing form of a training step: every kernel value of the Kzx units is ALSO stored, kfull[m * kf_sM + n * kf_sN + p] = k(z_m, x_np), m < M
  // (the reverse pass needs them all again: conv_gp/kernels.py:117-133 differentiated; recomputing them was a 57 us launch at the headline size)
  double* kfull = nullptr; long kf_sM = 0, kf_sN = 0;
  int share_cu = 0;                                        // leave room on every CU for a workgroup of the factorisation chain (see head_units)
  double* kd = nullptr;                                    // kd[n * n_kd + i]: Kdiag[n] = sum_i kd[..] / P^2; every slot of an image is written (values or zeros)
  int upw_force = 0;                                       // units per wave (0: chosen by head_units_plan)
  int wpg_force = 0;                                       // storing form: waves per workgroup (0: chosen by head_units_plan)
  int stream_k = 0;                                        // storing form: the streamed-operand kernel even where a register-resident one exists (A/B)
  int no_rep = 0;                                          // storing form: evaluate every row even where rows share an image (n_mod < N)
  long long* trace = nullptr; long trace_wgs = 0;          // debugging aid (tools/sweep_trace.py): [workgroup][wave][8] stamps, first trace_wgs workgroups
  const char* timer = nullptr;                             // timer family of the launch (nullptr: "kuf" / "head_sweep")
  int want_kd = 0;                                         // Kdiag partial sums wanted: head_units_plan sizes n_kd, the caller then allocates kd [N][n_kd]
  int share_kb = 0;                                        // A/B: LDS claimed per workgroup beside the factorisation chain, KB (0: 54 = two workgroups per CU)
  int occ_force = -1;                                      // A/B: waves per SIMD the launch is held to through its LDS claim (-1: chosen; 0: no shaping)
  int occ = 0;                                             // chosen by head_units_plan (0: none)
  const double* exp_tab = nullptr;                         // 2^(j / 256) (exp2_table): staged in LDS by every workgroup
  int tail_mode = -1;                                      // balance of the launch's tail (-1: default levels; 0: equal Kdiag chunks throughout)
  int nfm = 0, nfp = 0, n_kd = 0, upw = 1, wpg = 4;        // set by head_units_plan (call it with kzx / want_kd / kuf already set)
  int nseg = 0; HuSeg seg[6]; long n_wgs = 0;              // the launch's segments and its workgroup count (head_units_plan)
  int st_split = 1, st_jn = 0, st_hold = 0;                // storing form with replicas: column-fragment ranges per row fragment, fragments per range, hold + replica-outer stores
  int split_force = -1;                                    // A/B: ranges per row fragment (-1: ceil(nfp / 8); 0: one)
  int st_jb = 0, st_rb = 0;                                // storing form: byte step per patch fragment (16 sP) and per replica (n_mod sN)
  int n_base = 0;                                          // rows the launch evaluates: N, or (storing form) the min(N, n_mod) distinct images -- their
                                                           // values are stored to every row n + r n_mod < N that shows the same image (set by head_units_plan)
  float inv_C = 1.f, inv_f = 1.f, inv_Wo = 1.f, inv_Wr = 1.f;         // reciprocals for the set-up's index splits
};
void head_units_plan(HeadUnitsArgs* a);
bool head_units_ok(const HeadUnitsArgs& a);
int head_units(dcgp_ctx* ctx, const HeadUnitsArgs& a);
int sweep_operand(dcgp_ctx* ctx, const double* Z, const double* in_scale, int M, int Mp, int L, double variance, double lengthscale, double* ZS);
int kdiag_reduce(dcgp_ctx* ctx, const double* partial, int n_parts, int N, double scale, double* out_N);   // out[n] = scale * sum_i partial[n * n_parts + i]
inline int sweep_lq(int L) { return (L + 2 + 3) / 4 * 4; }   // patch length + the two norm slots, in whole sub-steps of 4

// small-matrix helpers (rbf.hip / chol.hip / misc.hip)
int rbf_gram_padded(dcgp_ctx* ctx, const double* Z, int M, int L, BaseKernel bk, double jitter,
                    double* out, int ld, int Mp);                       // out [Mp, ld], pad diag = 1
int z_transpose_norms(dcgp_ctx* ctx, const double* Z, int M, int L, double* ZT, int Mp, int Lp, double* zn);
int potrf_batched(dcgp_ctx* ctx, double* const* d_ptrs, double** h_ptrs, int batch, int Mp, int ld,
                  int* d_info);   // d_info[b] = 0 or 1-based failing column
int trtri_batched(dcgp_ctx* ctx, double* const* d_L, double* const* d_Linv, double* const* d_LinvT,
                  int batch, int Mp, int ld);
// Right-hand sides that ride the factorisation chain of one matrix (chol_fused.hip): G_r = inv(L) Lq_r (r < R; Lq lower triangular,
// [R][Mp][Mp]) and alpha = inv(L) q_mu ([Mp][Rp], Rp = 16) by block forward substitution inside the panel launches.  G / alpha ==
// nullptr: only the sums of squares are kept (the KL's trace / Mahalanobis terms with a prior factor).  sums: [(R + 1)][ns],
// ns = np (np + 1) / 2, np = ceil(Mp / 32): slot p (p + 1) / 2 + ct of row r holds the 32 x 32 block (rows of panel p, column tile
// ct <= p) of G_r, row R slot p the panel's rows of alpha.  Yw: scratch [R][Mp][Mp] + [Mp][Rp] (the rows not yet final).
// Lq == nullptr && qmu == nullptr: nothing rides on this matrix.
struct ChainRhs { const double* Lq; const double* qmu; double* G; double* alpha; double* sums; double* Yw; int R, Rp; };
inline int chain_rhs_slots(int Mp) { const int np = (Mp + 31) / 32; return np * (np + 1) / 2; }
constexpr int kChainRhsMaxMp = 256;   // beyond: the substitution's workgroups outgrow a panel launch (the generic GEMM route takes those layers)
// left-looking fused Cholesky (+ inverse of the factor when d_Linv != nullptr): one launch per 32-wide panel.
// d_rhs != nullptr (device array, one entry per matrix; max_R = the largest R among them): right-hand sides ride the chain.
int factor_inverse_batched(dcgp_ctx* ctx, double* const* d_A, double* const* d_Linv, double* const* d_LinvT, int batch,
                           int Mp, int ld, int* d_info, bool defer_finish = false, const ChainRhs* d_rhs = nullptr, int max_R = 0);
int comm_gate_wait(dcgp_ctx* ctx);   // comm.hip (debugging aid)
int reduce_scatter_sum_f64_async(dcgp_ctx* ctx, double* block_dev, size_t shard);   // comm.hip (in place, this rank's shard)
int all_gather_f64_async(dcgp_ctx* ctx, double* block_dev, size_t shard);
int factor_finish_batched(dcgp_ctx* ctx, double* const* d_A, int batch, int Mp, int ld);
bool chain_can_ride(const dcgp_ctx* ctx, int Mp);   // right-hand sides may ride the chain of a matrix of this size under the ctx's options
int pad_copy(dcgp_ctx* ctx, const double* src, int rows, int cols, int lds, double* dst, int ldd, int rows_p,
             int cols_p, int mode, int batch, long src_batch, long dst_batch);   // mode 0 full, 1 lower-tri, 2: +I on pad diag
