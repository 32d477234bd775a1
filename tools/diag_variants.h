#pragma once
#include "chol_dev.h"
namespace chol_dev {
// ---- V1: multipliers by v_readlane (SGPR operands), no LDS in the loop ----
// lanes [0, W) hold the pivot rows WITH their symmetric upper part, any other lane a vector that takes the same eliminations.
template <int W>
__device__ __forceinline__ void panel_sgpr(double (&v)[W], int lane) {
#pragma unroll
  for (int c = 0; c < W; ++c) {
    const double piv = bcast_lane(v[c], c);
    const double t = v[c] * rcp_nr(piv);
#pragma unroll
    for (int cc = c + 1; cc < W; ++cc) {
      const double u = bcast_lane(v[cc], c);
      v[cc] = fma(-t, u, v[cc]);
    }
  }
}
// scalings after the loop: lane c < W holds piv_c in v[c]; y_c = 1/sqrt(piv_c) goes round through the LDS line
template <int W>
__device__ __forceinline__ int panel_finish(double (&v)[W], int lane, double (&col)[64]) {
  double mypiv = 1.0;
#pragma unroll
  for (int c = 0; c < W; ++c) mypiv = (lane == c) ? v[c] : mypiv;
  const unsigned long long bad = __ballot(!(mypiv > 0.0));
  const double y = rsqrt_nr(mypiv);
  double d = mypiv * y;
  d = fma(0.5 * y, fma(-d, d, mypiv), d);
  col[lane] = y;
#pragma unroll
  for (int c = 0; c < W; ++c) v[c] = (lane == c) ? d : v[c] * col[c];
  return bad ? __ffsll((long long)bad) : 0;
}
template <int MODE>   // 0: SGPR panel; 1: LDS panel (cleaned)
__device__ __forceinline__ void panel_run(double (&v)[16], int lane, double (&col)[64]);

// ---- V2: LDS line, cleaned: no per-step fail / pivot select ----
template <int W>
__device__ __forceinline__ void panel_lds(double (&v)[W], int lane, double (&col)[64]) {
  double ua[W], ub[W];
  col[lane] = v[0];
#pragma unroll
  for (int cc = 1; cc < W; ++cc) ua[cc] = col[cc];
#pragma unroll
  for (int c = 0; c < W; ++c) {
    double (&u)[W] = (c & 1) ? ub : ua;
    double (&un)[W] = (c & 1) ? ua : ub;
    const double piv = bcast_lane(v[c], c);
    const double t = v[c] * rcp_nr(piv);
    if (c + 1 < W) {
      v[c + 1] = fma(-t, u[c + 1], v[c + 1]);
      col[lane] = v[c + 1];
#pragma unroll
      for (int cc = c + 2; cc < W; ++cc) un[cc] = col[cc];
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int cc = c + 2; cc < W; ++cc) v[cc] = fma(-t, u[cc], v[cc]);
  }
}
template <> __device__ __forceinline__ void panel_run<0>(double (&v)[16], int lane, double (&col)[64]) { panel_sgpr<16>(v, lane); }
template <> __device__ __forceinline__ void panel_run<1>(double (&v)[16], int lane, double (&col)[64]) { panel_lds<16>(v, lane, col); }

// D, Xs: [NB][LD] in LDS.  Same contract as wave_potrf_inv32_2x16.
template <int MODE, int LD>
__device__ __forceinline__ int potrf_inv32_new(double (*D)[LD], double (*Xs)[LD], double (&col)[64], double (*T)[17], int lane) {
  constexpr int W = 16;
  const int lrow = lane >> 4, lcol = lane & 15;
  double v[W];
  // panel 1: lanes 0..31 rows (lanes < 16 with the symmetric upper part), lanes 32..47 unit vectors
  {
    const int rr = lane < NB ? lane : 0;
#pragma unroll
    for (int c = 0; c < W; ++c) {
      const double a = (c <= rr) ? D[rr][c] : D[c][rr];     // one read through a selected address
      v[c] = lane < NB ? a : (lane - NB == c ? 1.0 : 0.0);
    }
  }
  panel_run<MODE>(v, lane, col);
  int fail = panel_finish<W>(v, lane, col);
  if (lane < NB) {
#pragma unroll
    for (int c = 0; c < W; ++c) D[lane][c] = (c <= lane) ? v[c] : 0.0;
  } else if (lane < NB + W) {
#pragma unroll
    for (int r = 0; r < W; ++r) Xs[r][lane - NB] = v[r];
  }
  DCGP_WAVE_LDS_SYNC();
  {
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const double a = D[W + lcol][4 * s + lrow];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) D[W + lrow + 4 * q][W + lcol] -= acc[q];   // full 16 x 16 (symmetric) block
  }
  DCGP_WAVE_LDS_SYNC();
  double w[W];
  {
    const int rr = lane < W ? lane : 0;
#pragma unroll
    for (int c = 0; c < W; ++c) {
      const double a = D[W + rr][W + c];   // the MFMA above left the whole symmetric block
      w[c] = lane < W ? a : (lane - W == c ? 1.0 : 0.0);
    }
  }
  panel_run<MODE>(w, lane, col);
  const int fail2 = panel_finish<W>(w, lane, col);
  if (fail == 0 && fail2) fail = W + fail2;
  if (lane < W) {
#pragma unroll
    for (int c = 0; c < W; ++c) {
      D[W + lane][W + c] = (c <= lane) ? w[c] : 0.0;
      D[lane][W + c] = 0.0;
      Xs[lane][W + c] = 0.0;
    }
  } else if (lane < 2 * W) {
#pragma unroll
    for (int r = 0; r < W; ++r) Xs[W + r][lane] = w[r];
  }
  DCGP_WAVE_LDS_SYNC();
  {
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(D[W + lcol][4 * s + lrow], Xs[4 * s + lrow][lcol], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) T[lrow + 4 * q][lcol] = acc[q];
    DCGP_WAVE_LDS_SYNC();
    d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) x = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[W + lcol][W + 4 * s + lrow], T[4 * s + lrow][lcol], x, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) Xs[W + lrow + 4 * q][lcol] = -x[q];
  }
  DCGP_WAVE_LDS_SYNC();
  return fail;
}
}  // namespace chol_dev

