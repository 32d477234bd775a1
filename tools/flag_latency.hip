// Microbenchmark (gfx950): one-way latency of a release / acquire hand-off between two workgroups of ONE launch --
// what a persistent factorisation chain would pay per panel instead of a launch boundary (~3 us on one stream).
// Workgroup `a` and workgroup `b` (of a grid of 16; workgroup i runs on XCD i % 8) bounce a counter `iters` times;
// with payload > 0 the producer also writes `payload` doubles before the flag and the consumer reads them after it.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void pingpong(unsigned* flag, double* data, int a, int b, int iters, int payload, double* sink, long long* clk) {
  const int me = blockIdx.x == a ? 0 : (blockIdx.x == b ? 1 : -1);
  if (me < 0) return;
  const int tid = threadIdx.x;
  double acc = 0.0;
  const long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    const unsigned want = 2 * it + me;   // flag value that hands the turn to `me`
    if (tid == 0) {
      while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != want) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (payload) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      for (int i = tid; i < payload; i += 256) acc += data[(1 - me) * payload + i];
      for (int i = tid; i < payload; i += 256) data[me * payload + i] = acc + it;
    }
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flag, want + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  const long long t1 = wall_clock64();
  if (tid == 0 && me == 0) clk[0] = t1 - t0;
  sink[blockIdx.x * 256 + tid] = acc;
}

int main() {
  unsigned* flag; double *data, *sink; long long* clk;
  hipMalloc(&flag, 64); hipMalloc(&data, 1 << 20); hipMalloc(&sink, 1 << 20); hipMalloc(&clk, 64);
  const int iters = 2000;
  const int pairs[][2] = {{0, 8}, {0, 1}, {0, 4}, {0, 9}};
  const int payloads[] = {0, 2048, 8192};
  for (auto& p : pairs)
    for (int pl : payloads) {
      hipMemset(flag, 0, 64); hipMemset(data, 0, 1 << 20);
      pingpong<<<16, 256>>>(flag, data, p[0], p[1], iters, pl, sink, clk);
      hipDeviceSynchronize();
      long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
      // wall_clock64: 100 MHz; 2 hops per iteration
      printf("workgroups %d <-> %d (XCD %d, %d)  payload %5d doubles  one-way hop %.3f us\n", p[0], p[1], p[0] % 8, p[1] % 8, pl,
             (double)h * 0.01 / (2.0 * iters));
    }
  return 0;
}
