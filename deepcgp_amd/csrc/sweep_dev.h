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

// one 256-thread block per 32 rows of Z (bx, bx + nbx, ...); call with all threads of the block
__device__ __forceinline__ void zs_task(const ZsTask& p, int bx, int nbx) {
  __shared__ double zs_t[32][33];
  __shared__ double zs_n[8][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int mb = bx; mb * 32 < p.Mp; mb += nbx) {
    const int m0 = mb * 32;
    double acc = 0.0;
    for (int l0 = 0; l0 < p.L; l0 += 32) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = ty + 8 * e, m = m0 + r, l = l0 + tx;
        zs_t[r][tx] = (m < p.M && l < p.L) ? p.Z[(long)m * p.L + l] * (p.in_scale ? p.in_scale[l] : 1.0) * p.csq : 0.0;
      }
      __syncthreads();
      for (int r = ty; r < 32; r += 8) {
        const int l = l0 + r, m = m0 + tx;
        const double v = zs_t[tx][r];
        if (l < p.L && m < p.Mp) p.ZS[(long)l * p.Mp + m] = v;
        acc = fma(v, v, acc);
      }
    }
    __syncthreads();
    zs_n[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && m0 + tx < p.Mp) {
      double s2 = 0.0;
      for (int q = 0; q < 8; ++q) s2 += zs_n[q][tx];
      const int m = m0 + tx;
      p.ZS[(long)p.L * p.Mp + m] = -0.5 * s2 + p.log2var;
      p.ZS[(long)(p.L + 1) * p.Mp + m] = 1.0;
      for (int l = p.L + 2; l < p.Lq; ++l) p.ZS[(long)l * p.Mp + m] = 0.0;
    }
  }
}
