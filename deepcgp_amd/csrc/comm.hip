// comm.hip -- multi-GPU: one process per GPU, one RCCL communicator per ctx, collectives on the ctx stream.
// The forward ELBO needs exactly one collective per step: a 1-element fp64 sum all-reduce of the
// per-GPU data term; KL terms are replicated and computed redundantly.
// librccl is dlopen()ed on first use so that libdcgp.so loads on hosts without it.
#include <dlfcn.h>
#include <mutex>
#include <unistd.h>

#include <cstdio>

#include "layer_impl.h"

namespace {

// the few RCCL entry points we need, with the NCCL ABI (rccl.h): ncclUniqueId is 128 bytes
struct UniqueId { char internal[128]; };
typedef int (*fn_get_unique_id)(UniqueId*);
typedef int (*fn_comm_init_rank)(void** comm, int nranks, UniqueId id, int rank);
typedef int (*fn_comm_destroy)(void* comm);
typedef int (*fn_all_reduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t s);
typedef int (*fn_reduce_scatter)(const void* send, void* recv, size_t recvcount, int dtype, int op, void* comm, hipStream_t s);
typedef int (*fn_all_gather)(const void* send, void* recv, size_t sendcount, int dtype, void* comm, hipStream_t s);
typedef const char* (*fn_get_error_string)(int);
typedef int (*fn_comm_count)(void* comm, int* count);

struct Rccl {
  void* h = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_reduce_scatter reduce_scatter = nullptr;
  fn_all_gather all_gather = nullptr;
  fn_get_error_string get_error_string = nullptr;
  fn_comm_count comm_count = nullptr;
};
Rccl g_rccl;

bool rccl_load() {
  if (g_rccl.h) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return false;
  g_rccl.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
  g_rccl.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
  g_rccl.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
  g_rccl.reduce_scatter = (fn_reduce_scatter)dlsym(h, "ncclReduceScatter");
  g_rccl.all_gather = (fn_all_gather)dlsym(h, "ncclAllGather");
  g_rccl.get_error_string = (fn_get_error_string)dlsym(h, "ncclGetErrorString");
  g_rccl.comm_count = (fn_comm_count)dlsym(h, "ncclCommCount");
  if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.all_reduce) {
    dlclose(h);
    return false;
  }
  g_rccl.h = h;
  return true;
}

// RCCL announces itself with printf ("RCCL version : ...", four lines) the first time a communicator is set up.  A library must not write
// to its host's stdout (bench.py's contract is ONE JSON line there, and C stdio flushes the banner at exit, BEHIND that line): while an
// RCCL set-up call runs, file descriptor 1 points at stderr.
// Side effect, documented in include/dcgp.h: for the duration of the set-up call OTHER host threads' writes to fd 1 land on stderr too.
// The swap is serialised process-wide (two contexts initialising concurrently would otherwise restore each other's descriptor).
std::mutex g_fd_mutex;
struct StdoutToStderr {
  int saved = -1;
  std::lock_guard<std::mutex> lock{g_fd_mutex};
  StdoutToStderr() {
    fflush(stdout);
    saved = dup(1);
    if (saved >= 0 && dup2(2, 1) < 0) { close(saved); saved = -1; }
  }
  ~StdoutToStderr() {
    if (saved < 0) return;
    fflush(stdout);
    dup2(saved, 1);
    close(saved);
  }
};

constexpr int kNcclFloat64 = 8;   // ncclDouble
constexpr int kNcclSum = 0;       // ncclSum

}  // namespace

int allreduce_sum_f64_async(dcgp_ctx* ctx, double* buf_dev, int n) {
  if (!ctx->comm) return ctx_fail(ctx, DCGP_ERR_RCCL, "allreduce: no communicator on this ctx");
  int rc = g_rccl.all_reduce(buf_dev, buf_dev, (size_t)n, kNcclFloat64, kNcclSum, ctx->comm, ctx->stream);
  if (rc != 0)
    return ctx_fail(ctx, DCGP_ERR_RCCL, "ncclAllReduce failed: %s",
                    g_rccl.get_error_string ? g_rccl.get_error_string(rc) : "?");
  return DCGP_OK;
}

// In-place reduce-scatter of a block of nranks * shard doubles: this rank's shard [rank * shard, (rank + 1) * shard) ends up summed over the
// ranks (the rest of the block is left as it was).  In-place all-gather: every rank's shard of the block to every rank.
int reduce_scatter_sum_f64_async(dcgp_ctx* ctx, double* block_dev, size_t shard) {
  if (!ctx->comm || !g_rccl.reduce_scatter) return ctx_fail(ctx, DCGP_ERR_RCCL, "reduce_scatter: no communicator on this ctx (or librccl without ncclReduceScatter)");
  const int rc = g_rccl.reduce_scatter(block_dev, block_dev + (size_t)ctx->rank * shard, shard, kNcclFloat64, kNcclSum, ctx->comm, ctx->stream);
  if (rc != 0) return ctx_fail(ctx, DCGP_ERR_RCCL, "ncclReduceScatter failed: %s", g_rccl.get_error_string ? g_rccl.get_error_string(rc) : "?");
  return DCGP_OK;
}
int all_gather_f64_async(dcgp_ctx* ctx, double* block_dev, size_t shard) {
  if (!ctx->comm || !g_rccl.all_gather) return ctx_fail(ctx, DCGP_ERR_RCCL, "all_gather: no communicator on this ctx (or librccl without ncclAllGather)");
  const int rc = g_rccl.all_gather(block_dev + (size_t)ctx->rank * shard, block_dev, shard, kNcclFloat64, ctx->comm, ctx->stream);
  if (rc != 0) return ctx_fail(ctx, DCGP_ERR_RCCL, "ncclAllGather failed: %s", g_rccl.get_error_string ? g_rccl.get_error_string(rc) : "?");
  return DCGP_OK;
}

// debugging aid (dcgp_debug_comm_gate): holds the comm stream until the host opens the gate -- or ~4 s have passed: a test must not hang a GPU
__global__ void comm_gate_kernel(const int* gate) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 && wall_clock64() - t0 < 400000000LL) __builtin_amdgcn_s_sleep(32);
}
int comm_gate_wait(dcgp_ctx* ctx) {
  int* dgate = nullptr;
  if (hipHostGetDevicePointer((void**)&dgate, ctx->comm_gate, 0) != hipSuccess) return ctx_fail(ctx, DCGP_ERR_HIP, "comm gate not mapped");
  hipLaunchKernelGGL(comm_gate_kernel, dim3(1), dim3(1), 0, ctx->stream, dgate);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

extern "C" {

// Debugging aid for the test that the data term's all-reduce of a step in flight does not hold up the next step (tests/test_gpu_model.py):
// closed != 0: every all-reduce enqueued on the comm stream from now on first waits for the gate; closed == 0: opens it (and removes it).
// main_idle_out (may be NULL): bit 0 -- the ctx's main stream has nothing left to do right now (hipStreamQuery), bit 1 -- nor has the comm stream.
int dcgp_debug_comm_gate(dcgp_ctx* ctx, int closed, int* main_idle_out) {
  if (!ctx) return DCGP_ERR_ARG;
  if (closed && !ctx->comm_gate) {
    if (hipHostMalloc((void**)&ctx->comm_gate, sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return ctx_fail(ctx, DCGP_ERR_ALLOC, "comm gate");
    *ctx->comm_gate = 0;
  } else if (!closed && ctx->comm_gate) {
    __atomic_store_n(ctx->comm_gate, 1, __ATOMIC_SEQ_CST);
    if (ctx->stream_comm) hipStreamSynchronize(ctx->stream_comm);
    hipHostFree(ctx->comm_gate);
    ctx->comm_gate = nullptr;
  }
  if (main_idle_out)
    *main_idle_out = (hipStreamQuery(ctx->stream) == hipSuccess ? 1 : 0) | ((!ctx->stream_comm || hipStreamQuery(ctx->stream_comm) == hipSuccess) ? 2 : 0);
  return DCGP_OK;
}

// Contiguous shards of a block of n values over nranks ranks, every shard the same length ceil(n / nranks) (the collectives want equal
// counts: the block is padded to nranks * shard): rank r holds [r * shard, min((r + 1) * shard, n)).  deepcgp_amd/dist.py: grad_shard_range.
int dcgp_shard_range(long n, int nranks, int rank, long* lo, long* hi, long* shard) {
  if (n < 0 || nranks <= 0 || rank < 0 || rank >= nranks || !lo || !hi) return DCGP_ERR_ARG;
  const long sh = (n + nranks - 1) / nranks;
  *lo = (long)rank * sh < n ? (long)rank * sh : n;
  *hi = (long)(rank + 1) * sh < n ? (long)(rank + 1) * sh : n;
  if (shard) *shard = sh;
  return DCGP_OK;
}

int dcgp_comm_unique_id(unsigned char* out_128bytes) {
  if (!out_128bytes) return DCGP_ERR_ARG;
  if (!rccl_load()) return DCGP_ERR_RCCL;
  UniqueId id;
  StdoutToStderr quiet;
  if (g_rccl.get_unique_id(&id) != 0) return DCGP_ERR_RCCL;
  memcpy(out_128bytes, id.internal, 128);
  return DCGP_OK;
}

int dcgp_comm_init_rank(dcgp_ctx* ctx, int nranks, int rank, const unsigned char* id_128bytes) {
  if (!ctx || !id_128bytes || nranks <= 0 || rank < 0 || rank >= nranks)
    return ctx ? ctx_fail(ctx, DCGP_ERR_ARG, "comm_init_rank: bad args") : DCGP_ERR_ARG;
  if (!rccl_load()) return ctx_fail(ctx, DCGP_ERR_RCCL, "librccl could not be loaded: %s", dlerror());
  if (ctx->comm) dcgp_comm_destroy(ctx);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  UniqueId id;
  memcpy(id.internal, id_128bytes, 128);
  void* comm = nullptr;
  int rc;
  {
    StdoutToStderr quiet;
    rc = g_rccl.comm_init_rank(&comm, nranks, id, rank);
  }
  if (rc != 0)
    return ctx_fail(ctx, DCGP_ERR_RCCL, "ncclCommInitRank failed: %s",
                    g_rccl.get_error_string ? g_rccl.get_error_string(rc) : "?");
  ctx->comm = comm;
  ctx->nranks = nranks;
  ctx->rank = rank;
  return DCGP_OK;
}

int dcgp_comm_destroy(dcgp_ctx* ctx) {
  if (!ctx) return DCGP_ERR_ARG;
  if (ctx->comm && g_rccl.comm_destroy) {
    hipStreamSynchronize(ctx->stream);
    g_rccl.comm_destroy(ctx->comm);
  }
  ctx->comm = nullptr;
  ctx->nranks = 1;
  ctx->rank = 0;
  return DCGP_OK;
}

int dcgp_comm_count(dcgp_ctx* ctx, int* out_ranks) {
  if (!ctx || !out_ranks) return ctx ? ctx_fail(ctx, DCGP_ERR_ARG, "comm_count: bad args") : DCGP_ERR_ARG;
  *out_ranks = 0;
  if (!ctx->comm) return DCGP_OK;
  if (!g_rccl.comm_count || g_rccl.comm_count(ctx->comm, out_ranks) != 0) return ctx_fail(ctx, DCGP_ERR_RCCL, "ncclCommCount failed");
  return DCGP_OK;
}

int dcgp_allreduce_sum_f64(dcgp_ctx* ctx, double* buf_dev, int n) {
  if (!ctx || !buf_dev || n <= 0) return ctx ? ctx_fail(ctx, DCGP_ERR_ARG, "allreduce: bad args") : DCGP_ERR_ARG;
  DCGP_TRY(allreduce_sum_f64_async(ctx, buf_dev, n));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}

}  // extern "C"
