// General strided fp64 GEMM on v_mfma_f64_16x16x4_f64 for the backward pass (grad.hip): every operand is addressed
// through (row stride, column stride, batch stride), so transposes and the [Kc][R] <-> [R][Kc] views of the
// backward products need no copies.  64 x 64 output tile per 256-thread workgroup (4 waves, 32 x 32 each), BK = 16,
// register prefetch of the next k tile.  Long contractions with a small output (d alpha, d G_r, d L: K = number of
// patch columns) are split along k into a partial buffer and summed in a fixed order -- no atomics, so gradients are
// reproducible run to run.  The forward path's tuned kernel is gemm.hip; this one trades peak rate for generality.
#include "gemm_gen.h"

namespace {

constexpr int GT = 64, GK = 16, GLD = 80;   // GLD: 64 + 16 -> the two 16-lane halves of a ds_read_b64 hit disjoint banks

// AKF / BKF: the operand's contraction index is the contiguous one (fetch 4 consecutive k per thread), otherwise 4
// consecutive rows (columns) per thread.  VEC: every 4-element group is 16-byte aligned and contiguous (two b128 loads).
template <bool AKF, bool BKF, bool VEC>
__global__ __launch_bounds__(256) void gemm_gen_kernel(GenGemm g, int kchunk, double* part) {
  __shared__ double As[GK][GLD];
  __shared__ double Bs[GK][GLD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int bz = blockIdx.z, b = bz % g.batch, sp = bz / g.batch;
  const int i0 = blockIdx.y * GT, j0 = blockIdx.x * GT;
  const bool split = part != nullptr;
  double* C = split ? part + ((long)sp * g.batch + b) * (long)g.M * g.N : g.C + (long)b * g.c_bs;
  const long c_rs = split ? g.N : g.c_rs;
  if (g.lower_only && j0 > i0 + GT - 1) {   // tile strictly above the diagonal: nothing to compute
    if (!split && !g.accumulate)
      for (int e = t; e < GT * GT; e += 256) {
        const int i = i0 + e / GT, j = j0 + e % GT;
        if (i < g.M && j < g.N) C[(long)i * c_rs + j] = 0.0;
      }
    return;
  }
  const int kbeg = sp * kchunk, kend = min(g.K, kbeg + kchunk);
  const int a_m = AKF ? (t >> 2) : ((t & 15) * 4), a_k = AKF ? ((t & 3) * 4) : (t >> 4);
  const int b_n = BKF ? (t >> 2) : ((t & 15) * 4), b_k = BKF ? ((t & 3) * 4) : (t >> 4);
  // per-thread fetch pointers, advanced by one k tile per iteration; row / column validity is loop invariant
  const double* pa = g.A + (long)b * g.a_bs + (long)(i0 + a_m) * g.a_rs + (long)(kbeg + a_k) * g.a_cs;
  const double* pb = g.B + (long)b * g.b_bs + (long)(j0 + b_n) * g.b_cs + (long)(kbeg + b_k) * g.b_rs;
  const long a_step = (long)GK * g.a_cs, b_step = (long)GK * g.b_rs;
  const long a_u = AKF ? g.a_cs : g.a_rs, b_u = BKF ? g.b_rs : g.b_cs;   // stride between the thread's 4 elements
  bool va[4], vb[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    va[u] = (i0 + a_m + (AKF ? 0 : u)) < g.M;
    vb[u] = (j0 + b_n + (BKF ? 0 : u)) < g.N;
  }
  const bool a_all = va[0] && va[3], b_all = vb[0] && vb[3];
  double ra[4], rb[4];
  auto fetch = [&](int k0) {
    const bool full = k0 + GK <= kend;
    if (VEC && full && a_all) {
      const double2 x = *(const double2*)pa, y = *(const double2*)(pa + 2);
      ra[0] = x.x; ra[1] = x.y; ra[2] = y.x; ra[3] = y.y;
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) ra[u] = (va[u] && (full || k0 + a_k + (AKF ? u : 0) < kend)) ? pa[u * a_u] : 0.0;
    }
    if (VEC && full && b_all) {
      const double2 x = *(const double2*)pb, y = *(const double2*)(pb + 2);
      rb[0] = x.x; rb[1] = x.y; rb[2] = y.x; rb[3] = y.y;
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) rb[u] = (vb[u] && (full || k0 + b_k + (BKF ? u : 0) < kend)) ? pb[u * b_u] : 0.0;
    }
    if (g.kscale) {
      const double* ks = g.kscale + (long)b * g.ks_bs;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kq = k0 + b_k + (BKF ? u : 0);
        rb[u] *= kq < kend ? ks[(long)kq * g.ks_s] : 0.0;
      }
    }
    pa += a_step;
    pb += b_step;
  };
  d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  if (kbeg < kend) fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      As[a_k + (AKF ? u : 0)][a_m + (AKF ? 0 : u)] = ra[u];
      Bs[b_k + (BKF ? u : 0)][b_n + (BKF ? 0 : u)] = rb[u];
    }
    __syncthreads();
    if (k0 + GK < kend) fetch(k0 + GK);
#pragma unroll
    for (int kk = 0; kk < GK; kk += 4) {
      const int kr = kk + (lane >> 4), c = lane & 15;
      const double a0 = As[kr][wm + c], a1 = As[kr][wm + 16 + c];
      const double b0 = Bs[kr][wn + c], b1 = Bs[kr][wn + 16 + c];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int fi = 0; fi < 2; ++fi)
#pragma unroll
    for (int fj = 0; fj < 2; ++fj)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + wm + fi * 16 + (lane >> 4) + 4 * q, j = j0 + wn + fj * 16 + (lane & 15);
        if (i >= g.M || j >= g.N) continue;
        double v = acc[fi][fj][q];
        if (!split) {
          v *= g.alpha;
          if (g.colscale) v *= g.colscale[(long)j * g.cs_s + (long)b * g.cs_bs];
          if (g.lower_only && j > i) v = 0.0;
          if (g.accumulate) v += C[(long)i * c_rs + j];
        }
        C[(long)i * c_rs + j] = v;
      }
}

__global__ void splitk_reduce_kernel(GenGemm g, int ksplit, const double* part) {
  const long per = (long)g.M * g.N, total = per * g.batch;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int b = (int)(idx / per);
  const long e = idx % per;
  const int i = (int)(e / g.N), j = (int)(e % g.N);
  double v = 0.0;
  for (int s = 0; s < ksplit; ++s) v += part[(long)s * total + idx];
  v *= g.alpha;
  if (g.colscale) v *= g.colscale[(long)j * g.cs_s + (long)b * g.cs_bs];
  if (g.lower_only && j > i) v = 0.0;
  double* c = g.C + (long)b * g.c_bs + (long)i * g.c_rs + j;
  if (g.accumulate) v += *c;
  *c = v;
}

}  // namespace

int gemm_gen(dcgp_ctx* ctx, const GenGemm& g) {
  if (g.M <= 0 || g.N <= 0 || g.batch <= 0) return DCGP_OK;
  if (!g.A || !g.B || !g.C || g.K < 0) return ctx_fail(ctx, DCGP_ERR_ARG, "gemm_gen: bad arguments");
  int tiles = ((g.M + GT - 1) / GT) * ((g.N + GT - 1) / GT) * g.batch;
  if (g.lower_only) tiles = tiles * 5 / 8 + 1;   // tiles above the diagonal exit at once
  // split long contractions until the launch covers the chip a few times over
  int ksplit = 1;
  if (g.K >= 2048 && tiles < 1536) {
    ksplit = (1536 + tiles - 1) / tiles;
    const int max_split = g.K / 512;
    if (ksplit > max_split) ksplit = max_split;
    if (ksplit < 1) ksplit = 1;
  }
  int kchunk = g.K;
  double* part = nullptr;
  if (ksplit > 1) {
    kchunk = round_up((g.K + ksplit - 1) / ksplit, GK);
    ksplit = (g.K + kchunk - 1) / kchunk;
    part = (double*)ws_get(ctx, "gemm_gen_part", (size_t)ksplit * g.batch * g.M * g.N * sizeof(double));
    if (!part) return DCGP_ERR_ALLOC;
  }
  if ((long)g.batch * ksplit > 65535) return ctx_fail(ctx, DCGP_ERR_ARG, "gemm_gen: batch %d x split %d too large", g.batch, ksplit);
  dim3 grid((g.N + GT - 1) / GT, (g.M + GT - 1) / GT, g.batch * ksplit);
  const bool akf = g.a_cs == 1, bkf = g.b_rs == 1;
  // 16-byte loads: the contiguous stride is 1 and every other stride, the base and the k origin of a split keep 16-byte alignment
  auto even = [](long x) { return (x & 1) == 0; };
  const bool a_vec = (akf ? even(g.a_rs) : (g.a_rs == 1 && even(g.a_cs))) && even(g.a_bs) && ((uintptr_t)g.A % 16 == 0);
  const bool b_vec = (bkf ? even(g.b_cs) : (g.b_cs == 1 && even(g.b_rs))) && even(g.b_bs) && ((uintptr_t)g.B % 16 == 0);
  const bool vec = a_vec && b_vec;
#define GG_LAUNCH(AK, BKK, V) hipLaunchKernelGGL((gemm_gen_kernel<AK, BKK, V>), grid, dim3(256), 0, ctx->stream, g, kchunk, part)
  if (akf && bkf) { if (vec) GG_LAUNCH(true, true, true); else GG_LAUNCH(true, true, false); }
  else if (akf) { if (vec) GG_LAUNCH(true, false, true); else GG_LAUNCH(true, false, false); }
  else if (bkf) { if (vec) GG_LAUNCH(false, true, true); else GG_LAUNCH(false, true, false); }
  else { if (vec) GG_LAUNCH(false, false, true); else GG_LAUNCH(false, false, false); }
#undef GG_LAUNCH
  LAUNCH_CHECK(ctx);
  if (part) {
    const long total = (long)g.M * g.N * g.batch;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, g, ksplit, part);
    LAUNCH_CHECK(ctx);
  }
  return DCGP_OK;
}
