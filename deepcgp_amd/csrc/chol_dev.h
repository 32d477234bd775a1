// chol_dev.h -- device-side building blocks shared by chol.hip and chol_fused.hip (32-wide panels).
#pragma once
#include "common.h"

namespace chol_dev {

constexpr int NB = 32;

// ---------------------------------------------------------------------------------------------
// 64x64 output tile, operands staged through LDS in k chunks of 32 (coalesced global reads, conflict-free
// ds_read_b64 operand fetches).  A(i,k) = A[i*sAi + k*sAk], B(k,j) = B[k*sBk + j*sBj].
//   acc[x][y][v] -> row (wm*32 + x*16 + lrow + 4v), col (wn*32 + y*16 + lcol) of the tile
// ---------------------------------------------------------------------------------------------
struct TileLds {
  double As[64][NB + 1];   // [i][k]
  double Bs[NB][64 + 1];   // [k][j]
};

__device__ __forceinline__ void tile64_mfma(TileLds& t, const double* __restrict__ A, long sAi, long sAk, int m_valid,
                                            const double* __restrict__ B, long sBk, long sBj, int n_valid, int kdim,
                                            int tid, d4 (&acc)[2][2]) {
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lrow = lane >> 4, lcol = lane & 15;
  const bool a_kfast = sAk == 1, b_jfast = sBj == 1;
  for (int k0 = 0; k0 < kdim; k0 += NB) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = tid + e * 256;   // 2048 elements each
      const int ai = a_kfast ? idx >> 5 : idx & 63, ak = a_kfast ? idx & 31 : idx >> 6;
      t.As[ai][ak] = (ai < m_valid && k0 + ak < kdim) ? A[ai * sAi + (k0 + ak) * sAk] : 0.0;
      const int bj = b_jfast ? idx & 63 : idx >> 5, bk = b_jfast ? idx >> 6 : idx & 31;
      t.Bs[bk][bj] = (bj < n_valid && k0 + bk < kdim) ? B[(k0 + bk) * sBk + bj * sBj] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < NB; kk += 4) {
      double av[2], bv[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) av[x] = t.As[wm * 32 + x * 16 + lcol][kk + lrow];
#pragma unroll
      for (int y = 0; y < 2; ++y) bv[y] = t.Bs[kk + lrow][wn * 32 + y * 16 + lcol];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[x], bv[y], acc[x][y], 0, 0, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// wavefront-level 32x32 routines (lanes 32..63 mirror lanes 0..31)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double bcast_lane(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rsqrt_nr(double p) {
  double y = __builtin_amdgcn_rsq(p);
  y = y * fma(-0.5 * p * y, y, 1.5);
  y = y * fma(-0.5 * p * y, y, 1.5);
  return y;
}
// In-place Cholesky of the block whose row r sits in a[] of lane r.  Step c: the pivot comes over with one
// v_readlane pair, every lane scales its column-c entry, writes it to the LDS line `col` and reads the 31-c
// multipliers L[cc][c] back as uniform-address (broadcast) loads.  Entries above the diagonal pick up garbage
// that is never read.  Returns 0 or the 1-based column of the first non-positive pivot.
__device__ __forceinline__ int wave_potrf32(double (&a)[NB], int r, double (&col)[NB]) {
  int fail = 0;
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    const double piv = bcast_lane(a[c], c);
    if (!(piv > 0.0) && fail == 0) fail = c + 1;
    const double y = rsqrt_nr(piv);
    double d = piv * y;
    d = fma(0.5 * y, fma(-d, d, piv), d);       // sqrt(piv) to ~1 ulp
    a[c] = (r == c) ? d : a[c] * y;
    col[r] = a[c];
    double m[NB];
#pragma unroll
    for (int cc = c + 1; cc < NB; ++cc) m[cc] = col[cc];
#pragma unroll
    for (int cc = c + 1; cc < NB; ++cc) a[cc] = fma(-a[c], m[cc], a[cc]);
  }
  return fail;
}
// Cholesky AND inverse of the factor in the same 32 steps, on one full wavefront.  Lanes 0..31 hold the rows of the
// block as in wave_potrf32; lane 32 + c holds column c of the inverse, started as the unit vector e_c.  Step c of
// the right-looking forward substitution  x[c] /= L[c][c];  x[cc] -= L[cc][c] x[c]  uses exactly the scale (1/L[c][c])
// and the multipliers L[cc][c] the factorisation step has just put in registers, and is the same instruction stream
// as the row update  a[cc] -= a[c] L[cc][c]  -- so the upper half of the wavefront, idle otherwise, delivers inv(L_jj)
// with no additional instruction, LDS access or latency.  `col` is a 64-entry LDS line (upper half is a write sink).
__device__ __forceinline__ double rcp_nr(double p) {
  double r = __builtin_amdgcn_rcp(p);
  r = fma(r, fma(-p, r, 1.0), r);
  r = fma(r, fma(-p, r, 1.0), r);
  return r;
}
__device__ __forceinline__ int wave_potrf_inv32(double (&v)[NB], int lane, double (&col)[2 * NB]) {
  // Latency shaping.  (1) The 32-step recurrence never needs a square root: with the column kept UNSCALED,
  //     v[cc] -= (v[c] / piv) u[cc]        (u = the unscaled column, piv = its diagonal entry)
  // is the whole step, so the serial chain per step is readlane -> 1/piv -> one multiply -> the FMA that produces the
  // next pivot.  All 1/sqrt(piv) scalings -- L[r][c] = v_r[c] / sqrt(piv_c), the inverse's x[c] / L_cc, the diagonal
  // sqrt(piv) -- are applied once, after the loop, by 32 lanes in parallel.  (2) The column is final one step early
  // and is published -- next pivot first -- while the previous step's other updates are still being issued, so the LDS
  // round trip overlaps them.
  int fail = 0;
  double mypiv = 1.0;
  double ua[NB], ub[NB];   // unscaled column of the current / the next step (ping-pong: the reads for step c + 1 are
                           // issued before the trailing updates of step c have consumed the current one)
  col[lane] = v[0];
#pragma unroll
  for (int cc = 1; cc < NB; ++cc) ua[cc] = col[cc];
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    double (&u)[NB] = (c & 1) ? ub : ua;
    double (&un)[NB] = (c & 1) ? ua : ub;
    const double piv = bcast_lane(v[c], c);
    if (!(piv > 0.0) && fail == 0) fail = c + 1;
    mypiv = (lane == c) ? piv : mypiv;
    const double t = v[c] * rcp_nr(piv);
    if (c + 1 < NB) {
      v[c + 1] = fma(-t, u[c + 1], v[c + 1]);
      col[lane] = v[c + 1];
#pragma unroll
      for (int cc = c + 2; cc < NB; ++cc) un[cc] = col[cc];
      __builtin_amdgcn_sched_barrier(0);   // keep publish + read-back ahead of the trailing updates
    }
#pragma unroll
    for (int cc = c + 2; cc < NB; ++cc) v[cc] = fma(-t, u[cc], v[cc]);
  }
  // the deferred scalings: ys[c] = 1 / sqrt(piv_c) = 1 / L_cc
  const double y = rsqrt_nr(mypiv);
  double d = mypiv * y;
  d = fma(0.5 * y, fma(-d, d, mypiv), d);       // sqrt(piv) to ~1 ulp
  col[lane] = y;                                // lanes >= NB write the sink half
#pragma unroll
  for (int c = 0; c < NB; ++c) v[c] = (lane == c) ? d : v[c] * col[c];
  return fail;
}
// one wavefront hands values to itself through LDS: LDS operations of a wave complete in order; this keeps the compiler from moving
// accesses across the hand-off and drains the queue
#define DCGP_WAVE_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// The same recurrence on a W-column panel of any height: lanes [0, rows) hold the rows (lanes < W the pivot rows), any other lane a
// vector that takes the same eliminations (a unit vector e_c yields column c of the inverse of the W x W factor; a zero vector stays zero).
// Unscaled columns, deferred scalings and the one-step-early publication exactly as in wave_potrf_inv32.
template <int W>
__device__ __forceinline__ int wave_potrf_inv_panel(double (&v)[W], int lane, double (&col)[64]) {
  int fail = 0;
  double mypiv = 1.0;
  double ua[W], ub[W];
  col[lane] = v[0];
#pragma unroll
  for (int cc = 1; cc < W; ++cc) ua[cc] = col[cc];
#pragma unroll
  for (int c = 0; c < W; ++c) {
    double (&u)[W] = (c & 1) ? ub : ua;
    double (&un)[W] = (c & 1) ? ua : ub;
    const double piv = bcast_lane(v[c], c);
    if (!(piv > 0.0) && fail == 0) fail = c + 1;
    mypiv = (lane == c) ? piv : mypiv;
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
  const double y = rsqrt_nr(mypiv);
  double d = mypiv * y;
  d = fma(0.5 * y, fma(-d, d, mypiv), d);
  col[lane] = y;
#pragma unroll
  for (int c = 0; c < W; ++c) v[c] = (lane == c) ? d : v[c] * col[c];
  return fail;
}

// Cholesky factor AND inverse of the 32 x 32 block D (lower triangle; LDS, row stride NB + 1) on one wavefront, as two 16-column
// panels with the block work between them on the matrix cores.  The 32-column recurrence above is issue-bound: step c reads
// 31 - c multipliers back from LDS and applies as many FMAs, ~400 cycles a step.  Here
//   1. panel 1 (32 rows x 16 columns; lanes 32..47 carry e_0..e_15): L11, L21 and X11 = inv(L11) in 16 steps of <= 15 multipliers;
//   2. A22' = A22 - L21 L21^T (4 MFMAs);
//   3. panel 2 (16 rows; lanes 16..31 carry the unit vectors): L22 and X22 = inv(L22) in 16 steps;
//   4. X21 = -X22 (L21 X11) (8 MFMAs)
// -- about half the cycles.  On return D holds L (zero above the diagonal), Xs holds inv(L) (zero above), both [NB][NB + 1].
// col: 64 doubles, T: 16 x 17 doubles of LDS scratch.  Returns 0 or the 1-based column of the first non-positive pivot.
__device__ __forceinline__ int wave_potrf_inv32_2x16(double (*D)[NB + 1], double (*Xs)[NB + 1], double (&col)[64], double (*T)[17], int lane) {
  constexpr int W = 16;
  const int lrow = lane >> 4, lcol = lane & 15;
  // ---- panel 1 ----
  double v[W];
#pragma unroll
  for (int c = 0; c < W; ++c) v[c] = lane < NB ? D[lane][c] : ((lane < NB + W && c == lane - NB) ? 1.0 : 0.0);
  int fail = wave_potrf_inv_panel<W>(v, lane, col);
  if (lane < NB) {
#pragma unroll
    for (int c = 0; c < W; ++c) D[lane][c] = (c <= lane) ? v[c] : 0.0;          // L11 (rows < 16: lower part) and L21 (rows >= 16: all 16)
  } else if (lane < NB + W) {
#pragma unroll
    for (int r = 0; r < W; ++r) Xs[r][lane - NB] = v[r];                          // X11 (column lane - 32)
  }
  DCGP_WAVE_LDS_SYNC();
  // ---- A22' = A22 - L21 L21^T ----
  {
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const double a = D[W + lcol][4 * s + lrow];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) D[W + lrow + 4 * q][W + lcol] -= acc[q];
  }
  DCGP_WAVE_LDS_SYNC();
  // ---- panel 2 ----
  double w[W];
#pragma unroll
  for (int c = 0; c < W; ++c) w[c] = lane < W ? D[W + lane][W + c] : ((lane < 2 * W && c == lane - W) ? 1.0 : 0.0);
  const int fail2 = wave_potrf_inv_panel<W>(w, lane, col);
  if (fail == 0 && fail2) fail = W + fail2;
  if (lane < W) {
#pragma unroll
    for (int c = 0; c < W; ++c) {
      D[W + lane][W + c] = (c <= lane) ? w[c] : 0.0;                              // L22
      D[lane][W + c] = 0.0;                                                        // block above the diagonal
      Xs[lane][W + c] = 0.0;
    }
  } else if (lane < 2 * W) {
#pragma unroll
    for (int r = 0; r < W; ++r) Xs[W + r][lane] = w[r];                            // X22 (column lane - 16 of the block = column lane of Xs)
  }
  DCGP_WAVE_LDS_SYNC();
  // ---- X21 = -X22 (L21 X11) ----
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

// =====================================================================================================================
// Round 5: a 16 x 16 block on all 64 lanes ("halves" layout).  lane = h * 32 + i; i < 16: row i of the symmetric block, i >= 16: the unit
// vector e_{i-16} (it takes the same eliminations and ends as column i - 16 of the inverse).  A lane keeps the 8 columns of
// its vector with parity h: slot k <-> column 2k + h.  Both halves keep every column entry of their vector as it became
// final (lc[c]) -- the owner half publishes it through the LDS line, the other half reads it there.
// =====================================================================================================================
typedef double dbl2 __attribute__((ext_vector_type(2)));

// line: 48 doubles of LDS, 16-byte aligned: [0,16) rows in de-interleaved order ((i & 1) * 8 + (i >> 1)), [16,32) unit vectors, [32,48) sink
__device__ __forceinline__ void block16_halves(double (&x)[8], double (&lc)[16], double& pv, int lane, double* line) {
  const int h = lane >> 5, i = lane & 31;
  const int pi = i < 16 ? ((i & 1) * 8 + (i >> 1)) : i;
  double* const w_even = line + (h == 0 ? pi : 32 + (lane & 15));   // (the sink: 16 slots the non-owner lanes scribble on)
  double* const w_odd = line + (h == 1 ? pi : 32 + (lane & 15));
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
    block16_halves(x, lc, pv, lane, line);
    // scalings: lane c < 16 holds piv_c
    const unsigned long long bad = __ballot(lane < 16 && !(pv > 0.0));
    if (bad && fail == 0) fail = o + __ffsll((long long)bad);
    const double y = rsqrt_nr(pv);
    if (lane < 16) line[lane] = y;
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
  return fail;
}


// The whole workgroup (4 waves) on the 32 x 32 block (round 5).  Wave 0 runs the two 16 x 16 recurrences (halves layout) and the block products
// between them in LDL^T form -- UNSCALED columns: L = Lt diag(y), inv(L) = diag(y) Xt, y = 1 / sqrt(piv), r = 1 / piv = y^2 -- so that no
// square root, no broadcast of y and no scaling pass sits on its path:  Pt = A21 Xt11^T,  A22 -= (Pt diag(r1)) Pt^T,
// Xt21 = -Xt22 ((Pt diag(r1)) Xt11).  The other waves do what is off that path while it runs: zero the upper-right block of the inverse,
// T = (Pt diag(r1)) Xt11, the scalings of L11 / inv(L)11 (then L21, L22), each behind one of two workgroup barriers.  (tools/diag_bench.hip: the
// one-wave form spent 5200 of its 9300 cycles outside the two recurrences -- y through an LDS line, 16 multiplies and 32 stores per block, three
// LDS round trips of block products.)  Call with all 256 threads; ends without a barrier (callers synchronise before reading D / Xs).
// D: lower triangle of the block on entry (zero above), L on return; Xs: inv(L); line: 48 doubles, 16-byte aligned; T: 16 x 17; sc: 64 doubles.
// Returns (wave 0) 0 or the 1-based column of the first non-positive pivot.
template <int LD>
__device__ __forceinline__ int potrf_inv32_wg(double (*D)[LD], double (*Xs)[LD], double* line, double (*T)[17], double* sc, int tid) {
  const int lane = tid & 63, wave = tid >> 6, lrow = lane >> 4, lcol = lane & 15;
  const int h = lane >> 5, i = lane & 31;
  int fail = 0;
  double* r1 = sc, *y1 = sc + 16, *y2 = sc + 48;
  if (wave == 0) {
    double x[8], lc[16], pv;
    {
      const int rr = i < 16 ? i : 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = 2 * k + h;
        const double a = (c <= rr) ? D[rr][c] : D[c][rr];
        x[k] = i < 16 ? a : (i - 16 == c ? 1.0 : 0.0);
      }
    }
    block16_halves(x, lc, pv, lane, line);
    {
      const unsigned long long bad = __ballot(lane < 16 && !(pv > 0.0));
      if (bad) fail = __ffsll((long long)bad);
      if (lane < 16) {
        r1[lane] = rcp_nr(pv);
#pragma unroll
        for (int c = 0; c < 16; ++c) D[lane][c] = lc[c];              // Lt11 raw (the scaling pass zeroes above the diagonal)
      } else if (lane < 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Xs[r][lane - 16] = lc[r];        // Xt11 raw (exact zeros above the diagonal)
      }
    }
    DCGP_WAVE_LDS_SYNC();
    {
      d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(D[16 + lcol][4 * s + lrow], Xs[lcol][4 * s + lrow], acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) D[16 + lrow + 4 * q][lcol] = acc[q];   // Pt
    }
  } else {
    for (int e = tid - 64; e < 256; e += 192) Xs[e >> 4][16 + (e & 15)] = 0.0;   // the inverse's upper-right block
  }
  __syncthreads();   // B1: Lt11, Xt11, r1, Pt are there
  if (wave == 0) {
    {
      d4 a2 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double a = D[16 + lcol][4 * s + lrow];
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a * r1[4 * s + lrow], a, a2, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) D[16 + lrow + 4 * q][16 + lcol] -= a2[q];
    }
    DCGP_WAVE_LDS_SYNC();
    double x[8], lc[16], pv;
    {
      const int rr = i < 16 ? i : 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = 2 * k + h;
        const double a = (c <= rr) ? D[16 + rr][16 + c] : D[16 + c][16 + rr];
        x[k] = i < 16 ? a : (i - 16 == c ? 1.0 : 0.0);
      }
    }
    block16_halves(x, lc, pv, lane, line);
    {
      const unsigned long long bad = __ballot(lane < 16 && !(pv > 0.0));
      if (bad && fail == 0) fail = 16 + __ffsll((long long)bad);
      if (lane < 16) {
        y2[lane] = rsqrt_nr(pv);
#pragma unroll
        for (int c = 0; c < 16; ++c) D[16 + lane][16 + c] = lc[c];
      } else if (lane < 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Xs[16 + r][lane] = lc[r];
      }
    }
  } else if (wave == 1) {
    // T = (Pt diag(r1)) Xt11, then y1 and inv(L)11 = diag(y1) Xt11 in place (nobody reads Xt11 any more)
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(D[16 + lcol][4 * s + lrow] * r1[4 * s + lrow], Xs[4 * s + lrow][lcol], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) T[lrow + 4 * q][lcol] = acc[q];
    if (lane < 16) { const double r = r1[lane]; y1[lane] = r * rsqrt_nr(r); }
    DCGP_WAVE_LDS_SYNC();
#pragma unroll
    for (int q = 0; q < 4; ++q) Xs[lrow + 4 * q][lcol] *= y1[lrow + 4 * q];
  } else if (wave == 2) {
    // L11 = Lt11 diag(y1), zero above the diagonal
    const double r = r1[lcol], yc = r * rsqrt_nr(r);
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int rw = lrow + 4 * q; D[rw][lcol] = lcol <= rw ? D[rw][lcol] * yc : 0.0; }
  }
  __syncthreads();   // B2: block 2's raw results, T, y1, y2
  if (wave == 0) {
    d4 xx = d4{0.0, 0.0, 0.0, 0.0};
    double yv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) yv[q] = y2[lrow + 4 * q];
#pragma unroll
    for (int s = 0; s < 4; ++s) xx = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[16 + lcol][16 + 4 * s + lrow], T[4 * s + lrow][lcol], xx, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      Xs[16 + lrow + 4 * q][lcol] = -xx[q] * yv[q];                     // inv(L)21 = -diag(y2) Xt22 T
      Xs[16 + lrow + 4 * q][16 + lcol] *= yv[q];                        // inv(L)22 = diag(y2) Xt22 (this wave's own reads of Xt22 are done)
    }
  } else if (wave == 1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) D[16 + lrow + 4 * q][lcol] *= y1[lcol];                                  // L21 = Pt diag(y1)
  } else if (wave == 2) {
    const double yc = y2[lcol];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int rw = lrow + 4 * q; D[16 + rw][16 + lcol] = lcol <= rw ? D[16 + rw][16 + lcol] * yc : 0.0; }   // L22
  }
  return fail;
}

// Column c of inv(L) for a 32x32 lower-triangular L held in LDS (D) with its reciprocal diagonal (Dr):
// forward substitution, the row of L being read as broadcast loads once per step.
__device__ __forceinline__ void lane_trtri32(const double (*D)[NB + 1], const double* Dr, int c, double (&x)[NB]) {
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    double row[NB];
#pragma unroll
    for (int q = 0; q < r; ++q) row[q] = D[r][q];
    double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < r; ++q) s = fma(-row[q], x[q], s);   // x[q] == 0 for q < c
    x[r] = s * Dr[r];
  }
}


}  // namespace chol_dev
