// Internal (device): the patch sweeps' Z operand.  ZS [Lq][Mp], k-major: rows l < L = sqrt(c) * Z[m][l] (times the ARD scale where a layer
// has one), row L = -c |z_m|^2 / 2 + log2(variance), row L + 1 = 1, zero behind; c = log2(e) / lengthscale^2.  Against a patch
// column (sqrt(c) x, 1, -c |x|^2 / 2) the contraction over all Lq rows is log2 of the RBF kernel value (head_units.hip, conv_fused.hip).
#pragma once
#include "common.h"

struct ZsTask {
  const double* Z = nullptr; const double* in_scale = nullptr; double* ZS = nullptr;
  int M = 0, Mp = 0, L = 0, Lq = 0;
  double csq = 1.0, log2var = 0.0;
};

// A guarded load that stays ONE load instruction: address select, load, value select.  Written as `ok ? p[i] : 0.0` the compiler
// branches around every load and waits for it behind the branch -- a batch of 8 "independent" loads became 8 memory latencies in a row
// (prepare_all's Gram tile at the head's L = 250: 15.7 us of which 1.5 us were arithmetic).
__device__ __forceinline__ double ld_guard(const double* __restrict__ p, long i, bool ok) {
  const double v = p[ok ? i : 0];
  return ok ? v : 0.0;
}
// Z[m][l] (SC: * in_scale[l]), 0 outside.  SC is a template flag because a test of the pointer beside every load is a branch beside
// every load, with a wait in front of it.
template <bool SC>
__device__ __forceinline__ double ld_z(const double* __restrict__ Z, const double* __restrict__ in_scale, int m, int l, int M, int L) {
  const bool ok = m >= 0 && m < M && l < L;
  const double v = ld_guard(Z, (long)m * L + l, ok);
  if constexpr (SC) return v * ld_guard(in_scale, l, ok);
  else return v;
}

// |scale * z_m|^2 of one row by 32 lanes (part = 0..31), 8 loads in flight per lane; m < 0: no row (0).  All lanes get the sum.
template <bool SC>
__device__ __forceinline__ double row_sq_norm(const double* __restrict__ Z, const double* __restrict__ in_scale, double scale, int m, int L, int part) {
  double acc = 0.0;
  for (int l0 = 0; l0 < L; l0 += 256) {
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int l = l0 + part + 32 * k;
      v[k] = ld_z<SC>(Z, in_scale, m, l, 1 << 30, L) * scale;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc = fma(v[k], v[k], acc);
  }
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) acc += __shfl_xor(acc, o);
  return acc;
}

// 256-thread blocks; work items bx, bx + nbx, ...: one 32 x 32 tile of the scaled transpose each, then (one item per 8 rows) the
// slots behind the patch -- no item waits for more than one round of loads.  Call with all threads of the block and a [32][33]
// staging tile in LDS.  Items: zs_items(Mp, L).
__host__ __device__ inline int zs_items(int Mp, int L) { return ((Mp + 31) / 32) * ((L + 31) / 32) + (Mp + 7) / 8; }
template <bool SC>
__device__ __forceinline__ void zs_task_sc(const ZsTask& p, int bx, int nbx, double (*zs_t)[33]) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int nmb = (p.Mp + 31) / 32, nlb = (p.L + 31) / 32, ntiles = nmb * nlb, nnb = (p.Mp + 7) / 8;
  for (int w = bx; w < ntiles + nnb; w += nbx) {
    if (w < ntiles) {
      const int m0 = (w / nlb) * 32, l0 = (w % nlb) * 32;
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = ty + 8 * e, m = m0 + r, l = l0 + tx;
        zs_t[r][tx] = ld_z<SC>(p.Z, p.in_scale, m, l, p.M, p.L) * p.csq;
      }
      __syncthreads();
      for (int r = ty; r < 32; r += 8) {
        const int l = l0 + r, m = m0 + tx;
        if (l < p.L && m < p.Mp) p.ZS[(long)l * p.Mp + m] = zs_t[tx][r];
      }
    } else {
      const int m = (w - ntiles) * 8 + ty;
      const double acc = row_sq_norm<SC>(p.Z, p.in_scale, p.csq, m < p.M ? m : -1, p.L, tx);
      if (tx == 0 && m < p.Mp) {
        p.ZS[(long)p.L * p.Mp + m] = -0.5 * acc + p.log2var;
        p.ZS[(long)(p.L + 1) * p.Mp + m] = 1.0;
        for (int l = p.L + 2; l < p.Lq; ++l) p.ZS[(long)l * p.Mp + m] = 0.0;
      }
    }
  }
}
__device__ __forceinline__ void zs_task(const ZsTask& p, int bx, int nbx, double (*zs_t)[33]) {
  if (p.in_scale) zs_task_sc<true>(p, bx, nbx, zs_t);
  else zs_task_sc<false>(p, bx, nbx, zs_t);
}
