// gemm_chain.hip -- the 128 x 128 tile of gemm.hip for W-batched products (stage 3 of the conditional:
// T_r = G_r^T A1 for r = 0..R-1 against the SAME B strip), with the batch CHAINED inside the workgroup.
//
// At M = 256 a workgroup of the plain kernel lives for only 16 (or 8) k-tiles, so the exposed first-tile latency,
// the epilogue and the workgroup turnover cost ~30 % (dense 128x128 tiles: 54 TF/s at M = 256 vs 64 TF/s at
// M = 1024, tools/gemm_bench.hip).  Here one workgroup walks `rchunk` consecutive batch entries for its
// (column tile, row block): the software pipeline (register-staged double-buffered LDS tiles) runs straight
// through the batch boundaries -- the first tiles of entry r+1 are fetched while the last tile of entry r is
// multiplied -- and the B strip stays hot in L2 for the whole chain.  Per entry only the fused column
// sum-of-squares epilogue (and the accumulator reset) remains.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, WAVES_M = 4, WAVES_N = 4, NT = WAVES_M * WAVES_N * 64;
constexpr int WMR = BM / WAVES_M, WNC = BN / WAVES_N, FM = WMR / 16, FN = WNC / 16;
constexpr int LDW = BM + 16, LDB = BN + 16;
constexpr int CHUNKS = BK * BM / 2;   // 16-byte chunks per operand tile = 1024 = one per thread

__global__ __launch_bounds__(NT, 8) void gemm_chain_kernel(GemmArgs a, int n_col_tiles, int n_row_blocks, int n_chunks) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Ws = smem;                        // [2][BK][LDW]
  double* Bs = smem + 2 * BK * LDW;         // [2][BK][LDB]
  double* red = Bs + 2 * BK * LDB;          // [WAVES_M][BN]  (separate: tiles of the next entry are in flight)

  // XCD-aware decode (see gemm.hip)
  const long nwg = (long)n_col_tiles * n_row_blocks * n_chunks;
  const long orig = blockIdx.x;
  const long q = nwg / 8, rr8 = nwg % 8, xcd = orig % 8;
  const long wgid = (xcd < rr8 ? xcd * (q + 1) : rr8 * (q + 1) + (xcd - rr8) * q) + orig / 8;
  const int inner = n_row_blocks * n_chunks;
  const int ct = (int)(wgid / inner);
  const int rem = (int)(wgid % inner);
  const int chunk = rem / n_row_blocks;
  int rb = rem % n_row_blocks;
  if (a.tri == 1) rb = n_row_blocks - 1 - rb;
  const int r_lo = chunk * a.rchunk, r_hi = min(a.nW, r_lo + a.rchunk);
  const int i0 = rb * BM, j0 = ct * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lrow = lane >> 4, lcol = lane & 15;

  int klo = 0, khi = a.Mk;
  if (a.tri == 1) khi = min(a.Mk, i0 + BM);
  if (a.tri == 2) klo = min(i0, a.Mk);
  const int nk = (khi - klo) / BK;                 // k-tiles per batch entry (Mk is a multiple of 16)
  const int total = nk * (r_hi - r_lo);
  const int wave_i_lo = i0 + wm * WMR, wave_i_hi = wave_i_lo + WMR - 1;

  d4 acc[FM][FN];
#pragma unroll
  for (int x = 0; x < FM; ++x)
#pragma unroll
    for (int y = 0; y < FN; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};

  // this thread's chunk of both operand tiles
  const int row = tid / (BM / 2), col = (tid % (BM / 2)) * 2;
  const bool w_ok = i0 + col < a.Mi, b_ok = j0 + col < a.Kc;
  double2 rw, rbv;
  auto load_tile = [&](int t) {
    const int r = r_lo + t / nk, k = klo + (t % nk) * BK + row;
    rw = double2{0.0, 0.0};
    rbv = double2{0.0, 0.0};
    if (w_ok) rw = *reinterpret_cast<const double2*>(a.Wt + (long)r * a.wBatch + (long)k * a.ldw + i0 + col);
    if (b_ok) rbv = *reinterpret_cast<const double2*>(a.B + (long)k * a.ldb + j0 + col);
  };
  auto store_tile = [&](int buf) {
    *reinterpret_cast<double2*>(Ws + buf * BK * LDW + row * LDW + col) = rw;
    *reinterpret_cast<double2*>(Bs + buf * BK * LDB + row * LDB + col) = rbv;
  };

  if (total > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
  int buf = 0;
  for (int t = 0; t < total; ++t) {
    const bool has_next = t + 1 < total;
    if (has_next) load_tile(t + 1);
    const int kidx = t % nk, k0 = klo + kidx * BK;
    bool need = true;   // per-wave structural-zero skip inside the diagonal region
    if (a.tri == 1 && k0 > wave_i_hi) need = false;
    if (a.tri == 2 && k0 + BK - 1 < wave_i_lo) need = false;
    if (need) {
      const double* w = Ws + buf * BK * LDW + wm * WMR + lcol;
      const double* b = Bs + buf * BK * LDB + wn * WNC + lcol;
#pragma unroll
      for (int kk = 0; kk < BK; kk += 4) {
        double av[FM], bv[FN];
#pragma unroll
        for (int x = 0; x < FM; ++x) av[x] = w[(kk + lrow) * LDW + x * 16];
#pragma unroll
        for (int y = 0; y < FN; ++y) bv[y] = b[(kk + lrow) * LDB + y * 16];
#pragma unroll
        for (int x = 0; x < FM; ++x)
#pragma unroll
          for (int y = 0; y < FN; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[x], bv[y], acc[x][y], 0, 0, 0);
      }
    }
    if (kidx == nk - 1) {
      // ---- end of batch entry r: fused column sum of squares, then reset ----
      const int r = r_lo + t / nk;
#pragma unroll
      for (int y = 0; y < FN; ++y) {
        double s = 0.0;
#pragma unroll
        for (int x = 0; x < FM; ++x)
#pragma unroll
          for (int v = 0; v < 4; ++v) s += acc[x][y][v] * acc[x][y][v];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (lrow == 0) red[wm * BN + wn * WNC + y * 16 + lcol] = s;
      }
      __syncthreads();
      if (tid < BN) {
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < WAVES_M; ++m) s += red[m * BN + tid];
        const int j = j0 + tid;
        if (j < a.Kc) a.colsq[(long)r * a.sBatch + (long)rb * a.sRowBlk + j] = s;
      }
#pragma unroll
      for (int x = 0; x < FM; ++x)
#pragma unroll
        for (int y = 0; y < FN; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};
    }
    if (has_next) store_tile(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
}

}  // namespace

// W-batched, colsq-only products with one shared B (nB == 1, no C store) on 128-row tiles
int gemm_chain(dcgp_ctx* ctx, const GemmArgs& a, int* nrb_out) {
  const int nct = (a.Kc + BN - 1) / BN, nrb = (a.Mi + BM - 1) / BM;
  if (nrb_out) *nrb_out = nrb;
  const int rchunk = a.rchunk > 0 ? a.rchunk : 1;
  const int nch = (a.nW + rchunk - 1) / rchunk;
  const long nwg = (long)nct * nrb * nch;
  if (nwg == 0) return DCGP_OK;
  GemmArgs b = a;
  b.rchunk = rchunk;
  const size_t lds = (size_t)(2 * BK * LDW + 2 * BK * LDB + WAVES_M * BN) * sizeof(double);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL(gemm_chain_kernel, dim3((unsigned)nwg), dim3(NT), lds, ctx->stream, b, nct, nrb, nch);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}
