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

// =====================================================================================================================
// V3 "halves": a 16 x 16 block on all 64 lanes.  lane = h * 32 + i; i < 16: row i of the symmetric block, i >= 16: the unit
// vector e_{i-16} (it takes the same eliminations and ends as column i - 16 of the inverse).  A lane keeps the 8 columns of
// its vector with parity h: slot k <-> column 2k + h.  Both halves keep every column entry of their vector as it became
// final (lc[c]) -- the owner half publishes it through the LDS line, the other half reads it there.
// =====================================================================================================================
namespace chol_dev {
typedef double dbl2 __attribute__((ext_vector_type(2)));

// line: 96 doubles of LDS, 16-byte aligned: [0,16) rows in de-interleaved order ((i & 1) * 8 + (i >> 1)), [16,32) unit vectors, [32,96) sink
__device__ __forceinline__ void block16_halves(double (&x)[8], double (&lc)[16], double& pv, int lane, double* line) {
  const int h = lane >> 5, i = lane & 31;
  const int pi = i < 16 ? ((i & 1) * 8 + (i >> 1)) : i;
  double* const w_even = line + (h == 0 ? pi : 32 + lane);
  double* const w_odd = line + (h == 1 ? pi : 32 + lane);
  const double* const rd_u = line + h * 8;
  const double* const rd_x = line + pi;
  double u[8];
  double xi;
  *w_even = x[0];                       // column 0: slot 0 of half 0
#pragma unroll
  for (int j = 0; j < 4; ++j) { const dbl2 p = *reinterpret_cast<const dbl2*>(rd_u + 2 * j); u[2 * j] = p[0]; u[2 * j + 1] = p[1]; }
  xi = *rd_x;
  pv = 1.0;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int kc = c >> 1, hc = c & 1;
    const int kn = (c + 1) >> 1;         // slot of the next column
    const double piv = bcast_lane(x[kc], hc * 32 + c);
    {   // lane c keeps its pivot for the scalings after the loop
      int lo = __double2loint(pv), hi = __double2hiint(pv);
      const int plo = __builtin_amdgcn_readfirstlane(__double2loint(piv)), phi = __builtin_amdgcn_readfirstlane(__double2hiint(piv));
      asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(lo) : "s"(plo), "n"(c));
      asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"(phi), "n"(c));
      pv = __hiloint2double(hi, lo);
    }
    lc[c] = xi;
    const double t = -xi * rcp_nr(piv);
    double un[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double xin = 0.0;
    if (c + 1 < 16) {
      x[kn] = fma(t, u[kn], x[kn]);
      *((c + 1) & 1 ? w_odd : w_even) = x[kn];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (2 * j + 1 >= ((c + 1) >> 1)) { const dbl2 p = *reinterpret_cast<const dbl2*>(rd_u + 2 * j); un[2 * j] = p[0]; un[2 * j + 1] = p[1]; }
      xin = *rd_x;
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k >= kc + hc && k != kn) x[k] = fma(t, u[k], x[k]);
    if (c + 1 < 16) {
#pragma unroll
      for (int k = 0; k < 8; ++k) u[k] = un[k];
      xi = xin;
    }
  }
}

// Factor + inverse of the 32 x 32 block in D (lower triangle valid), as two 16-blocks in the halves layout + MFMA block work.
// Same contract as wave_potrf_inv32_2x16; line: 96 doubles, 16-byte aligned.
#ifndef DIAG_STAMP
#define DIAG_STAMP(k)
#endif
template <int LD>
__device__ __forceinline__ int potrf_inv32_halves(double (*D)[LD], double (*Xs)[LD], double* line, double (*T)[17], int lane) {
  const int lrow = lane >> 4, lcol = lane & 15;
  const int h = lane >> 5, i = lane & 31;
  int fail = 0;
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int o = 16 * blk;
    if (blk == 1) {
      // L21 = A21 X11^T, then A22 -= L21 L21^T (lower part is what is read back)
      d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(D[16 + lcol][4 * s + lrow], Xs[lcol][4 * s + lrow], acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) D[16 + lrow + 4 * q][lcol] = acc[q];
      DCGP_WAVE_LDS_SYNC();
      d4 a2 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double a = D[16 + lcol][4 * s + lrow];
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, a2, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) D[16 + lrow + 4 * q][16 + lcol] -= a2[q];
      DCGP_WAVE_LDS_SYNC();
    }
    DIAG_STAMP(blk * 4 + 0)
    double x[8], lc[16], pv;
    {
      const int rr = i < 16 ? i : 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = 2 * k + h;
        const double a = (c <= rr) ? D[o + rr][o + c] : D[o + c][o + rr];
        x[k] = i < 16 ? a : (i - 16 == c ? 1.0 : 0.0);
      }
    }
    DIAG_STAMP(blk * 4 + 1)
    block16_halves(x, lc, pv, lane, line);
    DIAG_STAMP(blk * 4 + 2)
    // scalings: lane c < 16 holds piv_c
    const unsigned long long bad = __ballot(lane < 16 && !(pv > 0.0));
    if (bad && fail == 0) fail = o + __ffsll((long long)bad);
    const double y = rsqrt_nr(pv);
    line[lane] = y;                       // lanes >= 16 write junk beyond the 16 entries read back
    double ys[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const dbl2 p = *reinterpret_cast<const dbl2*>(line + 2 * j); ys[2 * j] = p[0]; ys[2 * j + 1] = p[1]; }
    // (static register indices only: a column index that depends on the lane's half would turn every access into a select chain)
    double pr[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) pr[c] = lc[c] * ys[c];        // diagonal: piv / sqrt(piv)
    if (lane < 16) {                                           // row i of L, zero above the diagonal
#pragma unroll
      for (int c = 0; c < 16; ++c) D[o + lane][o + c] = c <= lane ? pr[c] : 0.0;
    } else if (lane < 32) {                                    // column m of the inverse (exact zeros above the diagonal)
      const int m = lane - 16;
#pragma unroll
      for (int r = 0; r < 16; ++r) Xs[o + r][o + m] = pr[r];
    }
    if (blk == 1 && i < 16) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { D[i][16 + 2 * k + h] = 0.0; Xs[i][16 + 2 * k + h] = 0.0; }
    }
    DCGP_WAVE_LDS_SYNC();
    DIAG_STAMP(blk * 4 + 3)
  }
  {   // X21 = -X22 (L21 X11)
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(D[16 + lcol][4 * s + lrow], Xs[4 * s + lrow][lcol], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) T[lrow + 4 * q][lcol] = acc[q];
    DCGP_WAVE_LDS_SYNC();
    d4 xx = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) xx = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[16 + lcol][16 + 4 * s + lrow], T[4 * s + lrow][lcol], xx, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) Xs[16 + lrow + 4 * q][lcol] = -xx[q];
  }
  DCGP_WAVE_LDS_SYNC();
  DIAG_STAMP(8)
  return fail;
}
}  // namespace chol_dev

