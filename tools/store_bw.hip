// Microbenchmark: what a pure STORE kernel reaches on MI355X, in the access patterns the K_uf sweep can use
// (csrc/head_units.hip, storing form; MultiOutputConvKernel.Kuf, conv_gp/layers.py:23-32 writes P x M x N').
//   hipcc --offload-arch=gfx950 -O3 tools/store_bw.hip -o /tmp/store_bw && /tmp/store_bw
// Patterns, all writing a [Mp x K] row-major matrix of doubles, K = N * P columns (column n P + p):
//   fill      : grid-stride, 8 B per lane, fully contiguous (the ceiling of the store path)
//   fill16    : the same with 16 B per lane
//   tile      : the MFMA accumulator layout -- a wave owns (image n, row fragment u) and stores, per column fragment j and v = 0..3,
//               rows 16 u + lrow + 4 v, columns n P + 16 j + lcol: four 128-byte segments per instruction, rows K * 8 bytes apart
//   tile_rep  : the same tile stored to REP images n, n + n_mod, ... (layer 0: propagate() tiles the batch S times, the values repeat)
//   tile_rep_jr: the same, but fragment by fragment: each tile to all its replicas before the next tile (replica-outer above)
//   rowrun    : a wave owns (row m, run of 64 columns): 512 contiguous bytes per instruction (what a transpose through LDS would buy)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void fill8(double* out, size_t n, double v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = v;
}
__global__ void fill16(double2* out, size_t n, double v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = double2{v, v};
}
// one wave per (n, u); 4 waves per block
__global__ __launch_bounds__(256) void tile_store(double* out, int N, int P, int Mp, long ld, int rep, int n_mod, double v) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lrow = lane >> 4, lcol = lane & 15;
  const int nfm = Mp / 16, nfp = (P + 15) / 16;
  const long unit = (long)blockIdx.x * 4 + wave;
  const int n = (int)(unit / nfm), u = (int)(unit % nfm);
  if (n >= n_mod) return;
  for (int r = 0; r < rep; ++r) {
    double* base = out + (long)(16 * u + lrow) * ld + (long)(n + r * n_mod) * P + lcol;
    for (int j = 0; j < nfp; ++j) {
      if (16 * j + lcol < P) {
#pragma unroll
        for (int q = 0; q < 4; ++q) base[(long)4 * q * ld + 16 * j] = v + j;
      }
    }
  }
}
// the same, fragment-outer / replica-inner: what a sweep does that stores each tile to every replica as soon as it has it
__global__ __launch_bounds__(256) void tile_store_jr(double* out, int N, int P, int Mp, long ld, int rep, int n_mod, double v) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lrow = lane >> 4, lcol = lane & 15;
  const int nfm = Mp / 16, nfp = (P + 15) / 16;
  const long unit = (long)blockIdx.x * 4 + wave;
  const int n = (int)(unit / nfm), u = (int)(unit % nfm);
  if (n >= n_mod) return;
  for (int j = 0; j < nfp; ++j) {
    if (16 * j + lcol < P) {
      for (int r = 0; r < rep; ++r) {
        double* base = out + (long)(16 * u + lrow) * ld + (long)(n + r * n_mod) * P + lcol;
#pragma unroll
        for (int q = 0; q < 4; ++q) base[(long)4 * q * ld + 16 * j] = v + j;
      }
    }
  }
}
// one wave per (row m, 64 columns)
__global__ __launch_bounds__(256) void rowrun_store(double* out, int Mp, long K, long ld, double v) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long nrun = (K + 63) / 64;
  long unit = (long)blockIdx.x * 4 + wave;
  const long total = nrun * Mp;
  for (; unit < total; unit += (long)gridDim.x * 4) {
    const long m = unit / nrun, c = (unit % nrun) * 64 + lane;
    if (c < K) out[m * ld + c] = v;
  }
}

static float time_it(void (*launch)(void*), void* ctx, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch(ctx);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch(ctx);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

struct Case { double* out; int N, P, Mp, rep, n_mod; long K, ld; };
static void l_fill8(void* p) { Case& c = *(Case*)p; fill8<<<256 * 8, 256>>>(c.out, (size_t)c.Mp * c.ld, 1.0); }
static void l_fill16(void* p) { Case& c = *(Case*)p; fill16<<<256 * 8, 256>>>((double2*)c.out, (size_t)c.Mp * c.ld / 2, 1.0); }
static void l_tile(void* p) { Case& c = *(Case*)p; tile_store<<<(unsigned)(((long)c.N * (c.Mp / 16) + 3) / 4), 256>>>(c.out, c.N, c.P, c.Mp, c.ld, 1, c.N, 1.0); }
static void l_tile_rep(void* p) { Case& c = *(Case*)p; tile_store<<<(unsigned)(((long)c.n_mod * (c.Mp / 16) + 3) / 4), 256>>>(c.out, c.N, c.P, c.Mp, c.ld, c.rep, c.n_mod, 1.0); }
static void l_tile_jr(void* p) { Case& c = *(Case*)p; tile_store_jr<<<(unsigned)(((long)c.n_mod * (c.Mp / 16) + 3) / 4), 256>>>(c.out, c.N, c.P, c.Mp, c.ld, c.rep, c.n_mod, 1.0); }
static void l_rowrun(void* p) { Case& c = *(Case*)p; rowrun_store<<<256 * 16, 256>>>(c.out, c.Mp, c.K, c.ld, 1.0); }

int main() {
  struct Shape { const char* name; int N, P, Mp, n_mod; } shapes[] = {
      {"cfg2 conv0 (P=144 M=256 N'=320)", 320, 144, 256, 32},   {"cfg3 conv0 (P=169 M=256 N'=640)", 640, 169, 256, 64},
      {"cfg4 conv0 (P=225 M=384 N'=320)", 320, 225, 384, 32},   {"cfg5 conv0 (P=144 M=1024 N'=1280)", 1280, 144, 1024, 128}};
  for (auto& s : shapes) {
    Case c;
    c.N = s.N; c.P = s.P; c.Mp = s.Mp; c.n_mod = s.n_mod; c.rep = s.N / s.n_mod;
    c.K = (long)s.N * s.P; c.ld = (c.K + 127) / 128 * 128;
    const size_t bytes = (size_t)c.Mp * c.ld * 8;
    if (hipMalloc(&c.out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const double mb = (double)c.Mp * c.K * 8 / 1e6;
    printf("%s: %.1f MB\n", s.name, mb);
    struct { const char* n; void (*f)(void*); } ks[] = {{"fill", l_fill8}, {"fill16", l_fill16}, {"tile", l_tile}, {"tile_rep", l_tile_rep}, {"tile_rep_jr", l_tile_jr}, {"rowrun", l_rowrun}};
    for (auto& k : ks) {
      const float ms = time_it(k.f, &c, 20);
      printf("  %-12s %8.1f us  %7.0f GB/s\n", k.n, ms * 1e3, mb / ms);
    }
    hipFree(c.out);
  }
  return 0;
}
