// Microbenchmark (gfx950): which cheap protocols are COHERENT between two workgroups on the same XCD (workgroups 0 and 8)?
//   mode 0: flag polled with a workgroup-scope (sc0) atomic load, payload read with sc0 loads
//   mode 1: flag polled with an agent-scope (sc1) load, then buffer_inv sc0, payload read with plain loads, plain stores + s_waitcnt
//   mode 3: as mode 1 with buffer_inv sc1
//   mode 2: everything agent scope (sc1)                 (reference: known to work across XCDs)
// Every hop checks the payload it receives; polls give up after 2^20 tries (reported as FAIL) so that nothing can hang.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void pingpong(unsigned* flag, double* data, int a, int b, int iters, int payload, int* bad, long long* clk) {
  const int me = blockIdx.x == a ? 0 : (blockIdx.x == b ? 1 : -1);
  if (me < 0) return;
  const int tid = threadIdx.x;
  __shared__ int s_ok;
  int errors = 0;
  const long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    const unsigned want = 2 * it + me;
    if (tid == 0) {
      int ok = 0;
      for (int k = 0; k < (1 << 20); ++k) {
        unsigned v;
        if (MODE == 0) v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v == want) { ok = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) { if (tid == 0) atomicAdd(bad, 1000000); return; }
    if (MODE == 1) asm volatile("buffer_inv sc0" ::: "memory");
    if (MODE == 3) asm volatile("buffer_inv sc1" ::: "memory");
    asm volatile("" ::: "memory");
    // the payload the other side wrote in its previous turn: value = (turn index) + i
    if (want > 0) {
      for (int i = tid; i < payload; i += 256) {
        double v;
        const double* src = data + (1 - me) * payload + i;
        if (MODE == 0) v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 1 || MODE == 3) v = *src;
        else v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != (double)(want - 1) + i) ++errors;
      }
    }
    for (int i = tid; i < payload; i += 256) {
      double* dst = data + me * payload + i;
      const double v = (double)want + i;
      if (MODE == 0) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (MODE == 1 || MODE == 3) *dst = v;
      else __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) {
      if (MODE == 0) __hip_atomic_store(flag, want + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_store(flag, want + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  const long long t1 = wall_clock64();
  if (tid == 0 && me == 0) clk[0] = t1 - t0;
  if (errors) atomicAdd(bad, errors);
}

template <int MODE>
void run(const char* name, unsigned* flag, double* data, int* bad, long long* clk, int a, int b, int payload) {
  const int iters = 2000;
  hipMemset(flag, 0, 64); hipMemset(data, 0, 1 << 20); hipMemset(bad, 0, 4);
  pingpong<MODE><<<16, 256>>>(flag, data, a, b, iters, payload, bad, clk);
  hipDeviceSynchronize();
  long long h; int hb;
  hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost); hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  printf("%-44s wg %d<->%d payload %5d: %s  one-way hop %.3f us (errors %d)\n", name, a, b, payload, hb ? "FAIL" : "ok", (double)h * 0.01 / (2.0 * iters), hb);
}

int main() {
  unsigned* flag; double* data; int* bad; long long* clk;
  hipMalloc(&flag, 64); hipMalloc(&data, 1 << 20); hipMalloc(&bad, 4); hipMalloc(&clk, 64);
  for (int pl : {0, 1024, 4096}) {
    run<0>("sc0 flag + sc0 data", flag, data, bad, clk, 0, 8, pl);
    run<1>("sc1 flag + buffer_inv sc0 + plain data", flag, data, bad, clk, 0, 8, pl);
    run<2>("sc1 flag + sc1 data", flag, data, bad, clk, 0, 8, pl);
    run<3>("sc1 flag + buffer_inv sc1 + plain data", flag, data, bad, clk, 0, 8, pl);
  }
  run<3>("sc1 flag + buffer_inv sc1 + plain (2 XCDs!)", flag, data, bad, clk, 0, 1, 1024);
  run<3>("sc1 flag + buffer_inv sc1 + plain (2 XCDs!)", flag, data, bad, clk, 0, 1, 4096);
  return 0;
}
