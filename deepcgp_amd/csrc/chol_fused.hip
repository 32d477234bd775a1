// chol_fused.hip -- right-looking blocked Cholesky with the inverse of the factor folded into the same launches.
//
// The replicated M x M stage is a serial chain; what it costs is the NUMBER of dependent launches and the latency
// inside each (launch + memory round trips), not flops.  A textbook right-looking potrf (panel, update) followed by
// a recursive-doubling trtri took 23 dependent launches at M = 256.  Here panel j is ONE launch whose workgroups
// are independent of each other because each recomputes the small shared pieces it needs:
//     every workgroup : L_jj = chol(A[j,j])   (one wavefront: registers + LDS broadcast line), and the 32-column
//                       panel rows it touches, L[rows,j] = A[rows,j] L_jj^-T (one row per lane)
//     trailing tile   : A[ri,rc] -= L[ri,j] L[rc,j]^T                      (matrix cores, k = 32)
//     inverse tile    : with Y the running product of block Gauss transforms applied to I (inv(L) row blocks
//                       < j are final):  Ynew = inv(L_jj) Y[j,cols]  ->  final rows of inv(L);
//                       Y[rows,cols] -= L[rows,j] Ynew for the rows below (matrix cores, k = 32)
// so every launch has constant depth (k = 32) and the chain is M/32 launches + 1 (finish) for both
// tf.cholesky (conv_gp/conditionals.py:29, layers.py:151,156) and the inverse the triangular solves
// (conditionals.py:31-33,44-47) are applied with.  Final values go to separate buffers (other workgroups of the
// same launch still read the working copies).  All matrices of a model are batched in grid.y.
#include "chol_dev.h"

using namespace chol_dev;

// phase timestamps for tools/chol_trace.hip (compiled out of the library)
#ifdef CHOL_TRACE
__device__ long long g_chol_trace[64 * 32];
#define TR(k)                                                                                              \
  if (tid == 0 && b == 0 && (bx == 0 || bx == a.gx_trace - 1))                                             \
    g_chol_trace[(a.j / NB) * 32 + (bx == 0 ? 0 : 16) + (k)] = wall_clock64();
#else
#define TR(k)
#endif

namespace {

// factor + inverse of a diagonal block held in LDS (chol_dev.h): two 16 x 16 blocks in the halves layout + MFMA block work (round 5;
// tools/diag_bench.hip: 3.6 us against 5.1 for the two 16-column panels of round 3, DCGP_POTRF_PANELS at compile time).  col: 48 doubles.
// (LDS of chol_rl_kernel is a budget: beside a patch sweep -- which claims 53 KB per workgroup so that a chain workgroup fits the moment one of its
// three per CU ends, head_units.hip -- a kernel of 54144 bytes ran the head-only model's chain at 150 us and its sweep at 178, one of 53760 at 108 and
// 168: 4250 -> 4350 steps/s, 4500 -> 4780 in flight.  The line's sink is therefore 16 slots, not one per lane.)
#ifdef DCGP_POTRF_PANELS
constexpr int COL_N = 64;
#else
constexpr int COL_N = 48;
#endif
__device__ __forceinline__ int potrf_inv32(double (*D)[NB + 1], double (*Xs)[NB + 1], double (&col)[COL_N], double (*Tp)[17], int lane) {
#ifdef DCGP_POTRF_PANELS
  return wave_potrf_inv32_2x16(D, Xs, reinterpret_cast<double (&)[64]>(col), Tp, lane);
#else
  return potrf_inv32_halves<NB + 1>(D, Xs, col, Tp, lane);
#endif
}

struct RlArgs {
  double* const* A;      // working matrices (destroyed): trailing part updated in place
  double* Lout;          // [batch][Mp][ld] final factor (lower), then copied back over A by the finish kernel
  double* Y;             // [batch][Mp][ld] running inverse (scratch) or nullptr
  double* const* Linv;   // final inv(L) or nullptr; every element is written by the chain (zeros included)
  double* const* LinvT;  // its transpose, written in the same pass, or nullptr
  int Mp, ld, j, nt, nT, nct;
  int* info;
  // Look-ahead across the launch boundary.  Xin != nullptr: L_jj and inv(L_jj) of THIS panel were produced by the previous launch
  // ([batch][2][NB][NB]: factor, inverse) and are read instead of recomputed by every workgroup.  Xnext != nullptr: one extra
  // workgroup (blockIdx.x == la_idx) forms the NEXT diagonal block -- A[jn,jn] - P P^T, P = A[jn,j] inv(L_jj)^T, a 32 x 32 solve
  // and update -- factors and inverts it while the other workgroups do the trailing update, and leaves both for the next launch.
  // The 32-step pivot recurrence (5.5 us, the only inherently serial part of a panel) thereby runs BESIDE the trailing tiles of
  // the previous panel instead of in front of every one of them: 14.4 -> ~11 us per panel.
  const double* Xin = nullptr;
  double* Xnext = nullptr;
  int la_idx = -1;
  // Right-hand sides riding the chain (round 5): G_r = inv(L) Lq_r and alpha = inv(L) q_mu -- the conditional's small operands, one launch
  // of their own until now (prep_solve, 13 us behind the last panel) -- by block forward substitution ONE PANEL BEHIND the factorisation:
  // launch jp applies panel p = jp - 1, whose rows of L (Lout) and inverse diagonal block (Xall slot p) the previous launch left in memory;
  // the last launch also applies the last panel.  Workgroups [rhs_base, ...) of a launch: (right-hand side q, 32-column tile ct, row tile rt).
  const ChainRhs* rhs = nullptr;   // per matrix of the batch
  const double* Xall = nullptr;    // [np][batch][2][NB][NB]: factor and inverse of every diagonal block (slot 0 written by launch 0's first workgroup)
  double* Xself = nullptr;         // launch 0: where its first workgroup leaves block 0 (slot 0 of Xall)
  int rhs_base = 0, rhs_p = -1, rhs_last = 0, rhs_nt = 0, rhs_nq = 0, rhs_tpw = 1, batch = 0, np = 0;
  // XCD isolation (iso_per > 0: 1-D grid).  Workgroup i of a launch runs on XCD i % 8.  fp64 MFMAs of ANY wave hold a SIMD for 64 cycles
  // during which no other wave's VALU instruction issues there -- wave priorities do not help -- and the look-ahead workgroup's
  // recurrence is one issue-bound wave: with the right-hand sides' ~700 workgroups in the launch, three to a CU, every panel launch took
  // 1.3 us longer.  So the look-ahead workgroups of all matrices sit alone on one XCD and every other tile on the seven others (right-hand
  // sides first: the longest).  Which XCD: workgroups of XCD 7 start 0.7 us before those of XCD 0 (stamps of the first instruction: a launch's
  // look-ahead workgroups ended +8.2-8.9 us after its first workgroup on XCD 7, +9.0-10.5 on XCD 0 or 3).
  int iso_per = 0;
  int gx_trace = 0;

};

// Accesses to data another workgroup of the SAME launch wrote or will read (chol_persist_kernel) are relaxed atomics of agent
// scope = global_load / global_store with the sc1 bit: coherent across the 8 XCDs one access at a time, served by the memory side.
// What was measured on the way (tools/flag_latency.hip, tools/flag_latency2.hip, profiles/r03_flag_latency*.txt):
//   * ordinary accesses bracketed by agent-scope release / acquire fences: every release writes back all dirty lines of the XCD's L2,
//     and the kernels in front of the chain have just left megabytes of them -- 154 us for the 8 panels of M = 256;
//   * sc1 accesses, no fences (this code): a flag hop is 0.5-0.7 us, 8 KB of payload behind it 1.5 us more: 129 us;
//   * everything confined to one XCD (workgroup i of a launch runs on XCD i % 8) with plain data accesses: sc0 (workgroup-scope)
//     loads and `buffer_inv sc0` both leave stale lines in the vector L1 (the ping-pong test fails); `buffer_inv sc1` + plain
//     accesses is coherent inside an XCD (and, as it must, not across two) at 1.1 us per hop -- no better than a launch boundary.
// A launch boundary on one stream costs ~3 us, so a panel's hand-off between workgroups buys at most 1-2 us, and the look-ahead
// it enables is eaten by the round trip chain -> tiles -> chain (~8 us against a 6 us panel period).  The one-launch chain is
// therefore kept as a tested alternative (DCGP_CHOL_ONE_LAUNCH=1), not the default.
constexpr int SC_NONE = -1;
template <int SC>
__device__ __forceinline__ double ldg(const double* p) {
  if constexpr (SC == SC_NONE) return *p;
  else return __hip_atomic_load(p, __ATOMIC_RELAXED, SC);
}
template <int SC>
__device__ __forceinline__ void stg(double* p, double v) {
  if constexpr (SC == SC_NONE) *p = v;
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, SC);
}

// The prologue loads are written as fully unrolled register batches (all global loads issued, then all LDS stores):
// as rolled loops each iteration waited for its own load -- 4 + 8 + 8 serial memory latencies per workgroup.
template <int SH = SC_NONE>
__device__ __forceinline__ void load_diag(const double* __restrict__ A, int ld, int j, int nb, double (*D)[NB + 1], int tid) {
  double t[NB * NB / 256];
#pragma unroll
  for (int e = 0; e < NB * NB / 256; ++e) {
    const int idx = tid + e * 256, r = idx / NB, c = idx % NB;
    t[e] = (r == c) ? 1.0 : 0.0;
    if (r < nb && c < nb && c <= r) t[e] = ldg<SH>(A + (long)(j + r) * ld + j + c);
  }
#pragma unroll
  for (int e = 0; e < NB * NB / 256; ++e) {
    const int idx = tid + e * 256;
    D[idx / NB][idx % NB] = t[e];
  }
}
// rows [r0, r0+64) of the panel columns -> U[64][33] (zero beyond the matrix)
template <int SH = SC_NONE>
__device__ __forceinline__ void load_panel_rows(const double* __restrict__ A, int ld, int Mp, int j, int nb, int r0,
                                                double (*U)[NB + 1], int tid) {
  double t[64 * NB / 256];
#pragma unroll
  for (int e = 0; e < 64 * NB / 256; ++e) {
    const int idx = tid + e * 256, i = idx / NB, c = idx % NB;
    t[e] = (r0 + i < Mp && c < nb) ? ldg<SH>(A + (long)(r0 + i) * ld + j + c) : 0.0;
  }
#pragma unroll
  for (int e = 0; e < 64 * NB / 256; ++e) {
    const int idx = tid + e * 256;
    U[idx / NB][idx % NB] = t[e];
  }
}
// P = U inv(L_jj)^T on the matrix cores (64 x 32 x 32): P[i][c] = sum_q U[i][q] X[c][q].  Wave (wm, wn) owns rows
// wm*32 .. +31 and columns wn*16 .. +15; the result goes back over U after a barrier (both waves of a row block read
// all of its columns).
__device__ __forceinline__ void panel_solve_mfma(double (*U)[NB + 1], const double (*X)[NB + 1], int wm, int wn, int lrow,
                                                 int lcol, d4 (&p)[2]) {
  p[0] = d4{0.0, 0.0, 0.0, 0.0};
  p[1] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < NB; kk += 4) {
    const double bv = X[wn * 16 + lcol][kk + lrow];
#pragma unroll
    for (int x = 0; x < 2; ++x) p[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(U[wm * 32 + x * 16 + lcol][kk + lrow], bv, p[x], 0, 0, 0);
  }
}
__device__ __forceinline__ void panel_store(double (*U)[NB + 1], int wm, int wn, int lrow, int lcol, const d4 (&p)[2]) {
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int v = 0; v < 4; ++v) U[wm * 32 + x * 16 + lrow + 4 * v][wn * 16 + lcol] = p[x][v];
}


// ---- right-hand sides riding the chain (RlArgs::rhs): workgroup `idx` of the launch's right-hand-side part ----
// Item (q, ct, g): right-hand side q (q < R: Lq_q, lower triangular; q == R: q_mu), columns [32 ct, 32 ct + 32), row group g.  Every item forms
// Ynew = inv(L_pp) Y_p (the 32 x 32 block of the lagged panel p: 8 MFMAs a wave, cheaper than waiting for another workgroup); group 0 stores
// it (final rows of G_q / alpha) and its sum of squares; group g then updates the 64-row tiles 2g and 2g + 1 below, Y -= L[rows, p] Ynew.
// (Two row tiles per workgroup: with one, a launch carried ~700 workgroups against 672-768 resident slots, and the few that had to wait for
// a slot made every panel launch 1.4 us longer; the dispatcher also needs ~10 ns per workgroup.)  In the LAST launch the row tile that holds the
// last panel's rows goes on to make them final with the inverse block this launch was handed, and one more item per right-hand side
// (ct == p + 1: the diagonal block of Lq, first touched by the last panel) does only that.  Values a panel touches first come from the
// right-hand side itself, later ones from the scratch Yw.  The item on the diagonal also writes the structural zeros to the right of its
// final rows, so G needs no clearing.
__device__ __forceinline__ void rhs_tile(const RlArgs& a, const int b, const int idx, double (*D)[NB + 1], double (*Xs)[NB + 1], double (&col)[COL_N],
                                         double (*Tp)[17], double (*Ui)[NB + 1], double* UcTs, double* sc) {
  double (*Ts)[64 + 1] = reinterpret_cast<double (*)[64 + 1]>(UcTs);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane >> 4, lcol = lane & 15;
  const ChainRhs d = a.rhs[b];
  const int Mp = a.Mp, ld = a.ld, np = a.np, p = a.rhs_p;
  // decode: R blocks of rhs_nq items (Lq_q), then alpha's
  const int tpw = a.rhs_tpw, ng = a.rhs_nt > 0 ? (a.rhs_nt + tpw - 1) / tpw : 1;   // row tiles per workgroup (1 or 2), row groups
  int q, ct, g;
  bool final_only = false;
  if (idx < d.R * a.rhs_nq) {
    q = idx / a.rhs_nq;
    const int rem = idx % a.rhs_nq;
    if (p >= 0 && rem < (p + 1) * ng) { ct = rem / ng; g = rem % ng; }
    else { ct = p + 1; g = 0; final_only = true; }                     // (only enumerated in the last launch)
  } else {
    const int rem = idx - d.R * a.rhs_nq;
    q = d.R; ct = 0; g = rem;
    if (p < 0) { if (rem != 0 || !a.rhs_last) return; final_only = true; }
    else if (rem >= ng) return;
  }
  const bool is_alpha = q == d.R;
  const double* __restrict__ B = is_alpha ? d.qmu : (d.Lq ? d.Lq + (long)q * Mp * ld : nullptr);
  if (!B) return;
  const int ldb = is_alpha ? d.Rp : ld, ncol = is_alpha ? d.Rp : Mp, c0 = ct * 32;
  if (c0 >= ncol) return;
  const bool lagged = !final_only;
  const int j = 32 * (p < 0 ? 0 : p);
  double* __restrict__ Yq = d.Yw + (long)(is_alpha ? d.R : q) * Mp * ld;          // (alpha's scratch behind the R matrices; its rows are ldb wide)
  double* __restrict__ Cq = is_alpha ? d.alpha : (d.G ? d.G + (long)q * Mp * ld : nullptr);
  double* __restrict__ sq = d.sums + (long)q * (np * (np + 1) / 2);
  const double* __restrict__ Lout = a.Lout + (long)b * Mp * ld;
  const int jl = 32 * (np - 1), nbl = Mp - jl;                                     // the last panel
  const int r0[2] = {j + 32 + 64 * tpw * g, j + 32 + 64 * tpw * g + 64};            // this group's row tiles
  const bool live[2] = {lagged && r0[0] < Mp, lagged && tpw == 2 && r0[1] < Mp};
  const bool cont = live[0] && a.rhs_last && g == 0;                               // the first row tile holds the last panel's rows (r0 == jl)
  const bool have_x = a.Xin != nullptr;

  // ---- every global read, in one latency ----
  double tx[NB * NB / 256], tl[NB * NB / 256], tt[NB * NB / 256], tu[2][64 * NB / 256], old[2][2][4];
#pragma unroll
  for (int e = 0; e < NB * NB / 256; ++e) { tx[e] = 0.0; tl[e] = 0.0; }
  if (lagged) {
    const double* __restrict__ xp = a.Xall + ((long)p * a.batch + b) * 2 * NB * NB + NB * NB;
#pragma unroll
    for (int e = 0; e < NB * NB / 256; ++e) tx[e] = xp[tid + e * 256];
  }
  if ((cont || final_only) && have_x) {
    const double* __restrict__ xl = a.Xin + (long)b * 2 * NB * NB + NB * NB;
#pragma unroll
    for (int e = 0; e < NB * NB / 256; ++e) tl[e] = xl[tid + e * 256];
  }
  {
    const bool first = final_only || (is_alpha ? p == 0 : ct == p);
    const double* __restrict__ src = first ? B : Yq;
    const int jr = final_only ? jl : j;
#pragma unroll
    for (int e = 0; e < NB * NB / 256; ++e) {
      const int i2 = tid + e * 256, qr = i2 >> 5, c = i2 & 31;
      tt[e] = (jr + qr < Mp && c0 + c < ncol) ? src[(long)(jr + qr) * ldb + c0 + c] : 0.0;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int i = r0[t] + 16 * wave + lrow + 4 * v, gc = c0 + 16 * y + lcol;
          old[t][y][v] = (live[t] && i < Mp && gc < ncol) ? src[(long)i * ldb + gc] : 0.0;
        }
#pragma unroll
      for (int e = 0; e < 64 * NB / 256; ++e) {
        const int i2 = tid + e * 256, i = i2 / NB, c = i2 % NB;
        tu[t][e] = (live[t] && r0[t] + i < Mp) ? Lout[(long)(r0[t] + i) * ld + j + c] : 0.0;
      }
    }
  }
  if (final_only && !have_x) load_diag(a.A[b], ld, 0, min(NB, Mp), D, tid);   // one-panel matrix: this launch factors it, every workgroup for itself
  else {
#pragma unroll
    for (int e = 0; e < NB * NB / 256; ++e) { const int i2 = tid + e * 256; D[i2 / NB][i2 % NB] = tx[e]; }
  }
  if (have_x) {
#pragma unroll
    for (int e = 0; e < NB * NB / 256; ++e) { const int i2 = tid + e * 256; Xs[i2 / NB][i2 % NB] = tl[e]; }
  }
#pragma unroll
  for (int e = 0; e < NB * NB / 256; ++e) { const int i2 = tid + e * 256; Ts[i2 >> 5][i2 & 31] = tt[e]; }
  if (live[0]) {
#pragma unroll
    for (int e = 0; e < 64 * NB / 256; ++e) { const int i2 = tid + e * 256; Ui[i2 / NB][i2 % NB] = tu[0][e]; }
  }
  __syncthreads();
  if (final_only && !have_x) {
    potrf_inv32_wg<NB + 1>(D, Xs, col, Tp, sc, tid);
    __syncthreads();
  }

  // Ts (32 x 32) <- Xm Ts: wave w owns the 16 x 16 block (w >> 1, w & 1)
  auto ynew = [&](const double (*Xm)[NB + 1]) {
    const int bm = wave >> 1, bn = wave & 1;
    d4 y = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < NB; kk += 4) y = __builtin_amdgcn_mfma_f64_16x16x4f64(Xm[bm * 16 + lcol][kk + lrow], Ts[kk + lrow][bn * 16 + lcol], y, 0, 0, 0);
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 4; ++v) Ts[bm * 16 + lrow + 4 * v][bn * 16 + lcol] = y[v];
    __syncthreads();
  };
  // rows [jr, jr + nbr) of the result are final: store, zeros to the right of a diagonal tile, sum of squares -> slot
  auto finish = [&](int jr, int nbr, int pc) {
    double s = 0.0;
#pragma unroll
    for (int e = 0; e < NB * NB / 256; ++e) {
      const int i2 = tid + e * 256, qr = i2 >> 5, c = i2 & 31;
      if (qr < nbr && c0 + c < ncol) {
        const double x = Ts[qr][c];
        s = fma(x, x, s);
        if (Cq) Cq[(long)(jr + qr) * ldb + c0 + c] = x;
      }
    }
    if (Cq && !is_alpha && ct == pc) {
      const int z0 = 32 * (pc + 1), zw = Mp - z0;
      for (int i2 = tid; i2 < nbr * zw; i2 += 256) Cq[(long)(jr + i2 / zw) * ldb + z0 + i2 % zw] = 0.0;
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
    if (lane == 0) col[wave] = s;
    __syncthreads();
    if (tid == 0) sq[is_alpha ? pc : pc * (pc + 1) / 2 + ct] = (col[0] + col[1]) + (col[2] + col[3]);
  };

  if (final_only) {
    ynew(Xs);
    finish(jl, nbl, np - 1);
    return;
  }
  ynew(D);
  if (g == 0) finish(j, NB, p);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (!live[t]) break;
    if (t == 1) {   // the second tile's rows of L waited in registers
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 64 * NB / 256; ++e) { const int i2 = tid + e * 256; Ui[i2 / NB][i2 % NB] = tu[1][e]; }
      __syncthreads();
    }
    d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
#pragma unroll
    for (int kk = 0; kk < NB; kk += 4) {
      const double av = Ui[16 * wave + lcol][kk + lrow];
#pragma unroll
      for (int y = 0; y < 2; ++y) acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Ts[kk + lrow][16 * y + lcol], acc[y], 0, 0, 0);
    }
    if (t == 0 && cont) {
      // the last panel's rows, updated through panel p: final with this launch's inverse block
      __syncthreads();
      if (wave < 2) {
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int v = 0; v < 4; ++v) Ts[16 * wave + lrow + 4 * v][16 * y + lcol] = old[0][y][v] - acc[y][v];
      }
      __syncthreads();
      ynew(Xs);
      finish(jl, nbl, np - 1);
      return;
    }
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int i = r0[t] + 16 * wave + lrow + 4 * v, gc = c0 + 16 * y + lcol;
        if (i < Mp && gc < ncol) Yq[(long)i * ldb + gc] = old[t][y][v] - acc[y][v];
      }
  }
}

__global__ __launch_bounds__(256, 2) void chol_rl_kernel(RlArgs a) {
  __shared__ double D[NB][NB + 1];
  __shared__ __attribute__((aligned(16))) double col[COL_N];
  __shared__ double Ui[64][NB + 1];
  __shared__ double UcTs[64 * (NB + 1)];   // trailing tiles: second panel row block; inverse tiles: the Y tile
  __shared__ double Xs[NB][NB + 1];   // inv(L_jj)
  __shared__ double Tp[16][17];       // scratch of the diagonal block's factorisation
  __shared__ double sc[64];           // ... (reciprocal pivots and scalings)
  double (*Uc)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(UcTs);
  double (*Ts)[64 + 1] = reinterpret_cast<double (*)[64 + 1]>(UcTs);   // [NB][65]: old Y rows, then new
  static_assert(NB * 65 <= 64 * (NB + 1), "Y tile fits the shared slot");
  // the chain is latency: when it shares CUs with a throughput kernel (the head sweep of a head-first model, the previous step's layer
  // kernel with steps in flight) its waves go first in the issue arbitration
  // (the look-ahead workgroup is the launch's critical path and issue-bound: the other workgroups of the chain sharing its SIMDs yield to it)
  int b = blockIdx.y, bx = blockIdx.x;
  if (a.iso_per > 0) {
    const int lin = blockIdx.x, slot = lin >> 3;
    const int xcd = (lin + 1) & 7;   // (XCD 7 reads as 0: see iso_per)
    if (xcd == 0) {
      if (slot >= a.batch || a.la_idx < 0) return;
      b = slot; bx = a.la_idx;
    } else {
      // the right-hand sides first (the longest workgroups beside the look-ahead ones), then the trailing / inverse tiles
      const int item = slot * 7 + xcd - 1, nla = a.la_idx >= 0 ? 1 : 0;
      const int rhs_per = a.rhs ? a.iso_per + nla - a.rhs_base : 0, chain_per = a.iso_per - rhs_per;
      if (item < a.batch * rhs_per) { b = item / rhs_per; bx = a.rhs_base + item % rhs_per; }
      else {
        const int it = item - a.batch * rhs_per;
        if (it >= a.batch * chain_per) return;
        b = it / chain_per; bx = it % chain_per;
        if (nla && bx >= a.la_idx) ++bx;
      }
    }
  }
  if (bx == a.la_idx) __builtin_amdgcn_s_setprio(3);
  else __builtin_amdgcn_s_setprio(2);
  if (a.rhs && bx >= a.rhs_base) {   // a right-hand side riding the chain
    rhs_tile(a, b, bx - a.rhs_base, D, Xs, col, Tp, Ui, UcTs, sc);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lrow = lane >> 4, lcol = lane & 15;
  const int Mp = a.Mp, ld = a.ld, j = a.j, nb = min(NB, Mp - j);
  double* __restrict__ A = a.A[b];
  double* __restrict__ Lout = a.Lout + (long)b * Mp * ld;
  double* __restrict__ Y = a.Y ? a.Y + (long)b * Mp * ld : nullptr;
  const int below0 = j + NB;   // first row below the panel block

  // ---- prologue: every global read of this workgroup is issued here, in one latency ----
  TR(0)
  const bool have_x = a.Xin != nullptr;
  const bool is_la = bx == a.la_idx;
  if (have_x) {   // factor and inverse of this panel's diagonal block: left by the previous launch's look-ahead workgroup
    const double* __restrict__ xin = a.Xin + (long)b * 2 * NB * NB;
    double t[2 * NB * NB / 256];
#pragma unroll
    for (int e = 0; e < 2 * NB * NB / 256; ++e) t[e] = xin[tid + e * 256];
#pragma unroll
    for (int e = 0; e < 2 * NB * NB / 256; ++e) {
      const int idx = tid + e * 256, r = (idx & (NB * NB - 1)) / NB, c = idx % NB;
      if (idx < NB * NB) D[r][c] = t[e]; else Xs[r][c] = t[e];
    }
  } else {
    load_diag(A, ld, j, nb, D, tid);
  }
  const bool only_diag = a.nT == 0 && a.nct == 0 && !is_la;
  const bool is_trailing = bx < a.nT;
  int ti = 0, tc = 0, rt = -1, ct = 0;
  double old[2][2][4];   // values the final read-modify-write subtracts from
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int v = 0; v < 4; ++v) old[x][y][v] = 0.0;
  const int jn = j + NB, nbn = min(NB, Mp - jn);
  double la_u[NB * NB / 256], la_d[NB * NB / 256];   // look-ahead: A[jn.., j..] and the lower triangle of A[jn.., jn..]
  if (only_diag) {
  } else if (is_la) {
#pragma unroll
    for (int e = 0; e < NB * NB / 256; ++e) {
      const int idx = tid + e * 256, r = idx / NB, c = idx % NB;
      la_u[e] = (r < nbn && c < nb) ? A[(long)(jn + r) * ld + j + c] : 0.0;
      la_d[e] = (r == c) ? 1.0 : 0.0;
      if (r < nbn && c < nbn && c <= r) la_d[e] = A[(long)(jn + r) * ld + jn + c];
    }
  } else if (is_trailing) {
    int pair = bx;
    while (pair >= a.nt - tc) {   // column-major enumeration of the lower triangle: tc <= ti
      pair -= a.nt - tc;
      ++tc;
    }
    ti = tc + pair;
    load_panel_rows(A, ld, Mp, j, nb, below0 + ti * 64, Ui, tid);
    if (tc != ti) load_panel_rows(A, ld, Mp, j, nb, below0 + tc * 64, Uc, tid);
    const int ri0 = below0 + ti * 64, rc0 = below0 + tc * 64;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int i = ri0 + wm * 32 + x * 16 + lrow + 4 * v, jj = rc0 + wn * 32 + y * 16 + lcol;
          if (i < Mp && jj < Mp && jj <= i) old[x][y][v] = A[(long)i * ld + jj];
        }
  } else {
    const int yy = bx - a.nT;
    rt = yy / a.nct - 1;   // -1: the panel's own row block, else row tile below
    ct = yy % a.nct;
    const int c0 = ct * 64;
    if (rt >= 0) {
      load_panel_rows(A, ld, Mp, j, nb, below0 + rt * 64, Ui, tid);
      const int r0 = below0 + rt * 64;
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int i = r0 + wm * 32 + x * 16 + lrow + 4 * v, gc = c0 + wn * 32 + y * 16 + lcol;
            if (i < Mp && gc < j) old[x][y][v] = Y[(long)i * ld + gc];   // identity part is zero below the diagonal
          }
    }
    // Y[j + q, c0 + c] before this step: stored values left of column j, identity inside [j, j+32), zero beyond
    double tt[NB * 64 / 256];
#pragma unroll
    for (int e = 0; e < NB * 64 / 256; ++e) {
      const int idx = tid + e * 256, q = idx >> 6, c = idx & 63, gc = c0 + c;
      tt[e] = 0.0;
      if (q < nb && gc < j) tt[e] = Y[(long)(j + q) * ld + gc];
      else if (gc == j + q) tt[e] = 1.0;
    }
#pragma unroll
    for (int e = 0; e < NB * 64 / 256; ++e) {
      const int idx = tid + e * 256;
      Ts[idx >> 6][idx & 63] = tt[e];
    }
  }
  __syncthreads();
  TR(1)
  // ---- L_jj and inv(L_jj) (unless the previous launch left them); the look-ahead workgroup then the NEXT diagonal block too ----
  // ONE call site of the block routine for both (a loop of up to two passes): inlined twice, its ~1500 straight-line instructions made the kernel
  // large enough to evict a co-running sweep's code and its own (head-only model: sweep 168 -> 177 us, chain 141 -> 152 beside each other).
  int fail_j = 0, fail_n = 0;
  double* __restrict__ xn = is_la ? a.Xnext + (long)b * 2 * NB * NB : nullptr;
  for (int pass = have_x ? 1 : 0; pass < (is_la ? 2 : 1); ++pass) {
    if (pass == 1) {
      // ---- the next diagonal block: P = U inv(L_jj)^T, D_next = A[jn,jn] - P P^T ----
      double (*U)[NB + 1] = Ui;
      double (*Pm)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(UcTs);
#pragma unroll
      for (int e = 0; e < NB * NB / 256; ++e) { const int idx = tid + e * 256; U[idx / NB][idx % NB] = la_u[e]; }
      __syncthreads();
      const int bm = wave >> 1, bn = wave & 1;
      {
        d4 pacc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < NB; kk += 4)
          pacc = __builtin_amdgcn_mfma_f64_16x16x4f64(U[bm * 16 + lcol][kk + lrow], Xs[bn * 16 + lcol][kk + lrow], pacc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) Pm[bm * 16 + lrow + 4 * v][bn * 16 + lcol] = pacc[v];
      }
      __syncthreads();   // (also: everyone is done with D = L_jj, which the first workgroup of the launch publishes from its own copy)
#pragma unroll
      for (int e = 0; e < NB * NB / 256; ++e) { const int idx = tid + e * 256; D[idx / NB][idx % NB] = la_d[e]; }
      __syncthreads();
      {
        d4 dacc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < NB; kk += 4)
          dacc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pm[bm * 16 + lcol][kk + lrow], Pm[bn * 16 + lcol][kk + lrow], dacc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = bm * 16 + lrow + 4 * v, c = bn * 16 + lcol;
          if (c <= r && r < nbn) D[r][c] -= dacc[v];
        }
      }
      __syncthreads();
      TR(3)
    }
    const int fail = potrf_inv32_wg<NB + 1>(D, Xs, col, Tp, sc, tid);   // (all four waves; the status is wave 0's)
    __syncthreads();
    if (pass == 0) {
      TR(2)
      if (tid < 64) fail_j = fail;
      // (with a look-ahead workgroup in the launch it is the one writer of the status word: it sees this panel's failure too)
      if (tid == 0 && bx == 0 && a.la_idx < 0 && (j == 0 || (fail && a.info[b] == 0))) a.info[b] = fail ? j + fail : 0;
    } else {
      TR(4)
      if (tid < 64) fail_n = fail;
    }
  }
  if (is_la) {
#pragma unroll
    for (int e = 0; e < NB * NB / 256; ++e) {
      const int idx = tid + e * 256, r = idx / NB, c = idx % NB;
      xn[idx] = D[r][c];
      xn[NB * NB + idx] = Xs[r][c];
      if (r < nbn && c < nbn) Lout[(long)(jn + r) * ld + jn + c] = D[r][c];
    }
    if (tid == 0) {   // status word: first failing column of the chain so far (launch 0 initialises it)
      const int mine = fail_j ? j + fail_j : (fail_n ? jn + fail_n : 0);
      if (j == 0) a.info[b] = mine;
      else if (mine && a.info[b] == 0) a.info[b] = mine;
    }
    TR(5)
    return;
  }
  if (bx == 0 && !have_x) {   // publish L_jj (final; with Xin the previous launch's look-ahead workgroup already has)
    for (int idx = tid; idx < nb * nb; idx += 256) {
      const int r = idx / nb, c = idx % nb;
      Lout[(long)(j + r) * ld + j + c] = D[r][c];
    }
    if (a.Xself) {   // ... and block 0 with its inverse where the right-hand sides of the next launch read them
      double* __restrict__ xs = a.Xself + (long)b * 2 * NB * NB;
      for (int idx = tid; idx < NB * NB; idx += 256) { xs[idx] = D[idx / NB][idx % NB]; xs[NB * NB + idx] = Xs[idx / NB][idx % NB]; }
    }
  }
  if (only_diag) return;

  d4 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};

  if (is_trailing) {
    // ---- trailing tile (ti, tc): panel rows L[r,j] = A[r,j] inv(L_jj)^T, then A[ri, rc] -= L[ri,j] L[rc,j]^T ----
    d4 pi[2], pc[2];
    panel_solve_mfma(Ui, Xs, wm, wn, lrow, lcol, pi);
    if (tc != ti) panel_solve_mfma(Uc, Xs, wm, wn, lrow, lcol, pc);
    __syncthreads();
    panel_store(Ui, wm, wn, lrow, lcol, pi);
    if (tc != ti) panel_store(Uc, wm, wn, lrow, lcol, pc);
    __syncthreads();
    TR(3)
    const int ri0 = below0 + ti * 64, rc0 = below0 + tc * 64;
    double (*Ub)[NB + 1] = (tc == ti) ? Ui : Uc;
    if (tc == ti) {   // the diagonal tiles publish the panel rows of L (final)
      for (int idx = tid; idx < 64 * NB; idx += 256) {
        const int i = idx / NB, c = idx % NB;
        if (ri0 + i < Mp && c < nb) Lout[(long)(ri0 + i) * ld + j + c] = Ui[i][c];
      }
    }
#pragma unroll
    for (int kk = 0; kk < NB; kk += 4) {
      double avv[2], bvv[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) avv[x] = Ui[wm * 32 + x * 16 + lcol][kk + lrow];
#pragma unroll
      for (int y = 0; y < 2; ++y) bvv[y] = Ub[wn * 32 + y * 16 + lcol][kk + lrow];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(avv[x], bvv[y], acc[x][y], 0, 0, 0);
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int i = ri0 + wm * 32 + x * 16 + lrow + 4 * v, jj = rc0 + wn * 32 + y * 16 + lcol;
          // (the next diagonal block is the look-ahead workgroup's: it reads it as the previous panel left it, concurrently with this
          // launch -- nobody may update it in place, and nobody reads it from A afterwards)
          const bool la_block = a.la_idx >= 0 && i < below0 + NB && jj < below0 + NB;
          if (i < Mp && jj < Mp && jj <= i && !la_block) A[(long)i * ld + jj] = old[x][y][v] - acc[x][y][v];
        }
    TR(4)
    return;
  }

  // ---- inverse tile (rt, ct): columns [c0, c0+64) of the running inverse ----
  double* __restrict__ Linv = a.Linv[b];
  const int c0 = ct * 64;
  // Ynew = inv(L_jj) * Yold  (32 x 32 times 32 x 64) on the matrix cores: wave w owns columns w*16 .. +15
  d4 yn[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
#pragma unroll
  for (int kk = 0; kk < NB; kk += 4) {
    const double bv = Ts[kk + lrow][wave * 16 + lcol];
#pragma unroll
    for (int x = 0; x < 2; ++x) yn[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[x * 16 + lcol][kk + lrow], bv, yn[x], 0, 0, 0);
  }
  d4 pi[2];
  if (rt >= 0) panel_solve_mfma(Ui, Xs, wm, wn, lrow, lcol, pi);
  __syncthreads();
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int v = 0; v < 4; ++v) Ts[x * 16 + lrow + 4 * v][wave * 16 + lcol] = yn[x][v];
  if (rt >= 0) panel_store(Ui, wm, wn, lrow, lcol, pi);
  __syncthreads();
  TR(4)
  if (rt < 0) {
    // final rows j .. j+31 of inv(L) -- and the same block of the transpose; the structural zeros to the right of the
    // diagonal block are written too (the last column tile also covers everything beyond the tiles), so neither
    // output needs clearing beforehand
    double* __restrict__ LinvT = a.LinvT ? a.LinvT[b] : nullptr;
    const int cend = (ct == a.nct - 1) ? Mp : c0 + 64;
    for (int cb = c0; cb < cend; cb += 64)
      for (int idx = tid; idx < NB * 64; idx += 256) {
        const int r = idx >> 6, c = idx & 63, gc = cb + c;
        if (r < nb && gc < Mp) Linv[(long)(j + r) * ld + gc] = (cb == c0 && gc < j + nb) ? Ts[r][c] : 0.0;
      }
    if (LinvT)
      for (int cb = c0; cb < cend; cb += 64)
        for (int idx = tid; idx < NB * 64; idx += 256) {
          const int r = idx & 31, c = idx >> 5, gc = cb + c;   // r fastest: rows of the transpose are contiguous in r
          if (r < nb && gc < Mp) LinvT[(long)gc * ld + j + r] = (cb == c0 && gc < j + nb) ? Ts[r][c] : 0.0;
        }
    return;
  }
  // rows below: Y[rows, cols] -= L[rows,j] Ynew
  const int r0 = below0 + rt * 64;
#pragma unroll
  for (int kk = 0; kk < NB; kk += 4) {
    double avv[2], bvv[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) avv[x] = Ui[wm * 32 + x * 16 + lcol][kk + lrow];
#pragma unroll
    for (int y = 0; y < 2; ++y) bvv[y] = Ts[kk + lrow][wn * 32 + y * 16 + lcol];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(avv[x], bvv[y], acc[x][y], 0, 0, 0);
  }
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int i = r0 + wm * 32 + x * 16 + lrow + 4 * v, gc = c0 + wn * 32 + y * 16 + lcol;
        if (i < Mp && gc < j + nb) Y[(long)i * ld + gc] = old[x][y][v] - acc[x][y][v];
      }
  TR(5)
}

// ---------------------------------------------------------------------------------------------------------------
// The same chain in ONE launch (opt-in: DCGP_CHOL_ONE_LAUNCH=1; correct and tested, slower than the launches it replaces -- see the
// note above ldg<>).  Eight dependent launches cost eight stream boundaries (~3 us each) and, worse, put
// everything a panel does -- loads, the trailing update, the write-back -- on the serial path beside the only part that is
// inherently serial, the 32-step pivot recurrence of the diagonal block (5.5 us).  Here, per matrix:
//   * ONE chain workgroup walks the diagonal: factor + invert block j (wave_potrf_inv32), publish L_jj and inv(L_jj),
//     then form the NEXT diagonal block itself -- A[j+1,j+1] - P P^T with P = A[j+1,j] inv(L_jj)^T, 16 MFMAs on operands
//     that waves 1-3 fetched while wave 0 ran the recurrence -- and go straight on.  Its period is recurrence + ~1 us.
//   * T trailing workgroups take the tiles of every panel (the code of chol_rl_kernel, with inv(L_jj) read instead of
//     recomputed).  Panel j's tiles wait for two things: the chain's flag j (a release store, ~0.7 us to be seen: measured
//     by tools/flag_latency.hip) and the counter of panel j-1's tiles (the 64 x 64 tiles shift by 32 from panel to panel, so
//     a tile reads what several tiles of the previous panel wrote).  They run one panel period behind the chain, off its path.
// Flags are monotone across launches (base values from an epoch the host advances), so nothing is cleared between steps.
// Every workgroup of the launch must be resident at once (they wait for each other): the launcher caps the grid.
struct PcArgs {
  RlArgs r;               // A, Lout, Y, Linv, LinvT, Mp, ld, info (j, nt, nT, nct are per panel: computed in the kernel)
  int np, T;
  double* Xpub;           // [batch][np][NB][NB]  inv(L_jj), row-major, identity-padded where the last panel is narrow
  unsigned* sync;         // [batch][1 + np]: chain flag, then one counter of finished trailing workgroups per panel
  unsigned base_chain, base_trail;   // epoch * 4096, epoch * T
};

// Polls until *p has reached target (wrap-safe).  Gives up after ~a second of polling -- a workgroup that never arrives must not
// hang the device: the caller's results are then garbage and the launch's status word says so.
template <int SC>
__device__ __forceinline__ bool spin_until(const unsigned* p, unsigned target) {
  for (int it = 0; it < (1 << 22); ++it) {
    if ((int)(__hip_atomic_load(p, __ATOMIC_RELAXED, SC) - target) >= 0) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

// one tile (index bx of panel j's nT + nY tiles) of the trailing update / running inverse, inv(L_jj) given in Xs.
// Same arithmetic as chol_rl_kernel.  Ends with a barrier-free state: the caller synchronises before the LDS arrays are reused.
template <int SC>
__device__ __forceinline__ void pc_tile(const RlArgs& a, int b, int bx, int j, int nt, int nT, int nct, const double* __restrict__ xpub,
                                        const unsigned* chain_flag, unsigned chain_target, double (*Ui)[NB + 1], double* UcTs, double (*Xs)[NB + 1]) {
  double (*Uc)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(UcTs);
  double (*Ts)[64 + 1] = reinterpret_cast<double (*)[64 + 1]>(UcTs);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lrow = lane >> 4, lcol = lane & 15;
  const int Mp = a.Mp, ld = a.ld, nb = min(NB, Mp - j);
  double* __restrict__ A = a.A[b];
  double* __restrict__ Lout = a.Lout + (long)b * Mp * ld;
  double* __restrict__ Y = a.Y + (long)b * Mp * ld;
  const int below0 = j + NB;
  const bool is_trailing = bx < nT;
  int ti = 0, tc = 0, rt = -1, ct = 0;
  double old[2][2][4];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int v = 0; v < 4; ++v) old[x][y][v] = 0.0;
  if (is_trailing) {
    int pair = bx;
    while (pair >= nt - tc) { pair -= nt - tc; ++tc; }
    ti = tc + pair;
    load_panel_rows<SC>(A, ld, Mp, j, nb, below0 + ti * 64, Ui, tid);
    if (tc != ti) load_panel_rows<SC>(A, ld, Mp, j, nb, below0 + tc * 64, Uc, tid);
    const int ri0 = below0 + ti * 64, rc0 = below0 + tc * 64;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int i = ri0 + wm * 32 + x * 16 + lrow + 4 * v, jj = rc0 + wn * 32 + y * 16 + lcol;
          if (i < Mp && jj < Mp && jj <= i) old[x][y][v] = ldg<SC>(A + (long)i * ld + jj);
        }
  } else {
    const int yy = bx - nT;
    rt = yy / nct - 1;
    ct = yy % nct;
    const int c0 = ct * 64;
    if (rt >= 0) {
      load_panel_rows<SC>(A, ld, Mp, j, nb, below0 + rt * 64, Ui, tid);
      const int r0 = below0 + rt * 64;
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int i = r0 + wm * 32 + x * 16 + lrow + 4 * v, gc = c0 + wn * 32 + y * 16 + lcol;
            if (i < Mp && gc < j) old[x][y][v] = ldg<SC>(Y + (long)i * ld + gc);
          }
    }
    double tt[NB * 64 / 256];
#pragma unroll
    for (int e = 0; e < NB * 64 / 256; ++e) {
      const int idx = tid + e * 256, q = idx >> 6, c = idx & 63, gc = c0 + c;
      tt[e] = 0.0;
      if (q < nb && gc < j) tt[e] = ldg<SC>(Y + (long)(j + q) * ld + gc);
      else if (gc == j + q) tt[e] = 1.0;
    }
#pragma unroll
    for (int e = 0; e < NB * 64 / 256; ++e) {
      const int idx = tid + e * 256;
      Ts[idx >> 6][idx & 63] = tt[e];
    }
  }
  // inv(L_jj): wait for the chain workgroup's flag, then fetch the published block
  if (tid == 0) spin_until<SC>(chain_flag, chain_target);
  __syncthreads();   // (the published block is read with coherent loads: no cache invalidate)
  {
    double t[NB * NB / 256];
#pragma unroll
    for (int e = 0; e < NB * NB / 256; ++e) t[e] = ldg<SC>(xpub + tid + e * 256);
#pragma unroll
    for (int e = 0; e < NB * NB / 256; ++e) { const int idx = tid + e * 256; Xs[idx / NB][idx % NB] = t[e]; }
  }
  __syncthreads();

  d4 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};

  if (is_trailing) {
    d4 pi[2], pc[2];
    panel_solve_mfma(Ui, Xs, wm, wn, lrow, lcol, pi);
    if (tc != ti) panel_solve_mfma(Uc, Xs, wm, wn, lrow, lcol, pc);
    __syncthreads();
    panel_store(Ui, wm, wn, lrow, lcol, pi);
    if (tc != ti) panel_store(Uc, wm, wn, lrow, lcol, pc);
    __syncthreads();
    const int ri0 = below0 + ti * 64, rc0 = below0 + tc * 64;
    double (*Ub)[NB + 1] = (tc == ti) ? Ui : Uc;
    if (tc == ti) {   // the diagonal tiles publish the panel rows of L (final)
      for (int idx = tid; idx < 64 * NB; idx += 256) {
        const int i = idx / NB, c = idx % NB;
        if (ri0 + i < Mp && c < nb) Lout[(long)(ri0 + i) * ld + j + c] = Ui[i][c];
      }
    }
#pragma unroll
    for (int kk = 0; kk < NB; kk += 4) {
      double avv[2], bvv[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) avv[x] = Ui[wm * 32 + x * 16 + lcol][kk + lrow];
#pragma unroll
      for (int y = 0; y < 2; ++y) bvv[y] = Ub[wn * 32 + y * 16 + lcol][kk + lrow];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(avv[x], bvv[y], acc[x][y], 0, 0, 0);
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int i = ri0 + wm * 32 + x * 16 + lrow + 4 * v, jj = rc0 + wn * 32 + y * 16 + lcol;
          if (i < Mp && jj < Mp && jj <= i) stg<SC>(A + (long)i * ld + jj, old[x][y][v] - acc[x][y][v]);
        }
    return;
  }

  double* __restrict__ Linv = a.Linv[b];
  const int c0 = ct * 64;
  d4 yn[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
#pragma unroll
  for (int kk = 0; kk < NB; kk += 4) {
    const double bv = Ts[kk + lrow][wave * 16 + lcol];
#pragma unroll
    for (int x = 0; x < 2; ++x) yn[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[x * 16 + lcol][kk + lrow], bv, yn[x], 0, 0, 0);
  }
  d4 pi[2];
  if (rt >= 0) panel_solve_mfma(Ui, Xs, wm, wn, lrow, lcol, pi);
  __syncthreads();
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int v = 0; v < 4; ++v) Ts[x * 16 + lrow + 4 * v][wave * 16 + lcol] = yn[x][v];
  if (rt >= 0) panel_store(Ui, wm, wn, lrow, lcol, pi);
  __syncthreads();
  if (rt < 0) {
    double* __restrict__ LinvT = a.LinvT ? a.LinvT[b] : nullptr;
    const int cend = (ct == nct - 1) ? Mp : c0 + 64;
    for (int cb = c0; cb < cend; cb += 64)
      for (int idx = tid; idx < NB * 64; idx += 256) {
        const int r = idx >> 6, c = idx & 63, gc = cb + c;
        if (r < nb && gc < Mp) Linv[(long)(j + r) * ld + gc] = (cb == c0 && gc < j + nb) ? Ts[r][c] : 0.0;
      }
    if (LinvT)
      for (int cb = c0; cb < cend; cb += 64)
        for (int idx = tid; idx < NB * 64; idx += 256) {
          const int r = idx & 31, c = idx >> 5, gc = cb + c;
          if (r < nb && gc < Mp) LinvT[(long)gc * ld + j + r] = (cb == c0 && gc < j + nb) ? Ts[r][c] : 0.0;
        }
    return;
  }
  const int r0 = below0 + rt * 64;
#pragma unroll
  for (int kk = 0; kk < NB; kk += 4) {
    double avv[2], bvv[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) avv[x] = Ui[wm * 32 + x * 16 + lcol][kk + lrow];
#pragma unroll
    for (int y = 0; y < 2; ++y) bvv[y] = Ts[kk + lrow][wn * 32 + y * 16 + lcol];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(avv[x], bvv[y], acc[x][y], 0, 0, 0);
  }
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int i = r0 + wm * 32 + x * 16 + lrow + 4 * v, gc = c0 + wn * 32 + y * 16 + lcol;
        if (i < Mp && gc < j + nb) stg<SC>(Y + (long)i * ld + gc, old[x][y][v] - acc[x][y][v]);
      }
}

struct PcLds {
  double D[NB][NB + 1];
  __attribute__((aligned(16))) double col[COL_N];
  double Ui[64][NB + 1];
  double UcTs[64 * (NB + 1)];
  double Xs[NB][NB + 1];
  double Tp[16][17];
};

template <int SC>
__device__ __forceinline__ void pc_run(const PcArgs& p, const int b, const int role, PcLds& sh) {
  double (&D)[NB][NB + 1] = sh.D;
  double (&col)[COL_N] = sh.col;
  double (*Ui)[NB + 1] = sh.Ui;
  double* UcTs = sh.UcTs;
  double (*Xs)[NB + 1] = sh.Xs;
  const RlArgs& a = p.r;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane >> 4, lcol = lane & 15;
  const int Mp = a.Mp, ld = a.ld, np = p.np;
  unsigned* chain_flag = p.sync + (long)b * (1 + np);
  unsigned* trail_cnt = chain_flag + 1;
  double* __restrict__ xpub_b = p.Xpub + (long)b * np * NB * NB;
  const unsigned trail_target = p.base_trail + (unsigned)p.T;

  if (role == 0) {
    // ---------------- the chain workgroup ----------------
    double* __restrict__ A = a.A[b];
    double* __restrict__ Lout = a.Lout + (long)b * Mp * ld;
    double (*U)[NB + 1] = Ui;                                              // block row j + 1, panel columns (32 x 32)
    double (*Pm)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(UcTs);     // P = U inv(L_jj)^T
    load_diag(A, ld, 0, min(NB, Mp), D, tid);   // (written by an earlier launch: ordinary loads)
    __syncthreads();
    for (int jp = 0; jp < np; ++jp) {
      const int j = jp * NB, nb = min(NB, Mp - j);
      const bool more = jp + 1 < np;
      const int jn = j + NB, nbn = more ? min(NB, Mp - jn) : 0;
      // waves 1-3: operands of the look-ahead -- A[jn.., j..] and the lower triangle of A[jn.., jn..] as the trailing tiles of
      // panel jp - 1 left them -- into registers while wave 0 runs the recurrence (elements e*192 + tid-64 of 2 x 1024)
      constexpr int PF = (2 * NB * NB + 191) / 192;
      double pf[PF];
      if (tid < 64) {
        const int fail = potrf_inv32(D, Xs, col, sh.Tp, tid);
        if (tid == 0 && (jp == 0 || (fail && a.info[b] == 0))) a.info[b] = fail ? j + fail : 0;
      } else if (more) {
        if (jp > 0) {   // every wave waits for itself (one lane polls; the acquire's cache invalidate is per wave)
          if (lane == 0) spin_until<SC>(trail_cnt + (jp - 1), trail_target);
        }
#pragma unroll
        for (int e = 0; e < PF; ++e) {
          const int idx = e * 192 + tid - 64;
          pf[e] = 0.0;
          if (idx < NB * NB) {
            const int r = idx / NB, c = idx % NB;
            if (r < nbn && c < nb) pf[e] = ldg<SC>(A + (long)(jn + r) * ld + j + c);
          } else if (idx < 2 * NB * NB) {
            const int r = (idx - NB * NB) / NB, c = (idx - NB * NB) % NB;
            pf[e] = (r == c) ? 1.0 : 0.0;
            if (r < nbn && c < nbn && c <= r) pf[e] = ldg<SC>(A + (long)(jn + r) * ld + jn + c);
          }
        }
      }
      __syncthreads();
      // publish L_jj and inv(L_jj), then the flag
      for (int idx = tid; idx < nb * nb; idx += 256) {
        const int r = idx / nb, c = idx % nb;
        Lout[(long)(j + r) * ld + j + c] = D[r][c];
      }
      {
        double* __restrict__ xp = xpub_b + (long)jp * NB * NB;
#pragma unroll
        for (int e = 0; e < NB * NB / 256; ++e) { const int idx = tid + e * 256; stg<SC>(xp + idx, Xs[idx / NB][idx % NB]); }
      }
      __builtin_amdgcn_s_waitcnt(0);   // the coherent stores have been acknowledged before the barrier in front of the flag
      // look-ahead operands -> LDS (U; the next diagonal block goes to D once L_jj has been published from it)
      if (more && tid >= 64) {
#pragma unroll
        for (int e = 0; e < PF; ++e) {
          const int idx = e * 192 + tid - 64;
          if (idx < NB * NB) U[idx / NB][idx % NB] = pf[e];
        }
      }
      __syncthreads();
      if (tid == 0) __hip_atomic_store(chain_flag, p.base_chain + (unsigned)(jp + 1), __ATOMIC_RELAXED, SC);
      if (!more) break;
      if (tid >= 64) {
#pragma unroll
        for (int e = 0; e < PF; ++e) {
          const int idx = e * 192 + tid - 64;
          if (idx >= NB * NB && idx < 2 * NB * NB) D[(idx - NB * NB) / NB][(idx - NB * NB) % NB] = pf[e];
        }
      }
      // P = U inv(L_jj)^T: wave w owns the 16 x 16 block (w >> 1, w & 1)
      {
        const int bm = wave >> 1, bn = wave & 1;
        d4 pacc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < NB; kk += 4)
          pacc = __builtin_amdgcn_mfma_f64_16x16x4f64(U[bm * 16 + lcol][kk + lrow], Xs[bn * 16 + lcol][kk + lrow], pacc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) Pm[bm * 16 + lrow + 4 * v][bn * 16 + lcol] = pacc[v];
      }
      __syncthreads();
      // D_next -= P P^T
      {
        const int bm = wave >> 1, bn = wave & 1;
        d4 dacc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < NB; kk += 4)
          dacc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pm[bm * 16 + lcol][kk + lrow], Pm[bn * 16 + lcol][kk + lrow], dacc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = bm * 16 + lrow + 4 * v, c = bn * 16 + lcol;
          if (c <= r && r < nbn) D[r][c] -= dacc[v];
        }
      }
      __syncthreads();
    }
    return;
  }

  // ---------------- trailing workgroups ----------------
  const int w = role - 1;
  for (int jp = 0; jp < np; ++jp) {
    const int j = jp * NB;
    const int below = Mp - (j + NB);
    const int nt = below > 0 ? (below + 63) / 64 : 0;
    const int nT = nt * (nt + 1) / 2;
    const int nct = (min(j + NB, Mp) + 63) / 64;
    const int ntiles = nT + (1 + nt) * nct;
    if (jp > 0) {   // what this panel reads was written by the tiles of the previous one
      if (tid == 0) spin_until<SC>(trail_cnt + (jp - 1), trail_target);
      __syncthreads();
    }
    for (int t = w; t < ntiles; t += p.T) {
      pc_tile<SC>(a, b, t, j, nt, nT, nct, xpub_b + (long)jp * NB * NB, chain_flag, p.base_chain + (unsigned)(jp + 1), Ui, UcTs, Xs);
      __syncthreads();
    }
    __builtin_amdgcn_s_waitcnt(0);   // this workgroup's coherent stores have been acknowledged
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(trail_cnt + jp, 1u, __ATOMIC_RELAXED, SC);
  }
}

// grid (1 + T, batch): blockIdx.x == 0 is the matrix's chain workgroup
__global__ __launch_bounds__(256, 2) void chol_persist_kernel(PcArgs p) {
  __shared__ PcLds sh;
  pc_run<__HIP_MEMORY_SCOPE_AGENT>(p, blockIdx.y, blockIdx.x, sh);
}

// final factor back over A (lower triangle from Lout, strict upper triangle zero).  grid (Mp, batch)
__global__ void chol_finish_kernel(double* const* __restrict__ Ap, const double* __restrict__ Lout, int Mp, int ld) {
  double* __restrict__ A = Ap[blockIdx.y];
  const double* __restrict__ L = Lout + (long)blockIdx.y * Mp * ld;
  const int i = blockIdx.x;
  for (int c = threadIdx.x; c < Mp; c += blockDim.x) A[(long)i * ld + c] = (c <= i) ? L[(long)i * ld + c] : 0.0;
}

}  // namespace

// Cholesky factor in place (strict upper triangle zeroed) and, when d_Linv != nullptr, inv(L) (+ its transpose).
// right-hand sides ride the plain panel-per-launch chain with its look-ahead (the A/B routes -- one launch, no look-ahead, captured graph -- do not carry them)
bool chain_can_ride(const dcgp_ctx* ctx, int Mp) {
  return ctx->chain_ride_ok && Mp <= kChainRhsMaxMp && !ctx->opt.chol_one_launch && !ctx->opt.chol_no_lookahead && !ctx->opt.chain_graph && !ctx->opt.no_rhs_ride;
}

int factor_inverse_batched(dcgp_ctx* ctx, double* const* d_A, double* const* d_Linv, double* const* d_LinvT, int batch,
                           int Mp, int ld, int* d_info, bool defer_finish, const ChainRhs* d_rhs, int max_R) {
  if (batch <= 0) return DCGP_OK;
  if (d_rhs && (!d_Linv || !chain_can_ride(ctx, Mp))) return ctx_fail(ctx, DCGP_ERR_ARG, "factorisation chain: right-hand sides cannot ride this route");
  ScopedTimer t(ctx, "factor_chain");
  const size_t mm = (size_t)Mp * ld;
  double* Lout = (double*)ws_get(ctx, "chol_Lout" + ctx->ws_tag, (size_t)batch * mm * sizeof(double));
  double* Y = d_Linv ? (double*)ws_get(ctx, "chol_Y" + ctx->ws_tag, (size_t)batch * mm * sizeof(double)) : nullptr;
  if (!Lout || (d_Linv && !Y)) return DCGP_ERR_ALLOC;
  // one launch for the whole chain (chol_persist_kernel) where its workgroups are sure to be co-resident
  const bool one_launch = ctx->opt.chol_one_launch != 0;   // A/B switch (see the note above ldg<>)
  if (d_Linv && one_launch) {
    const int np = (Mp + NB - 1) / NB;
    int max_tiles = 1;
    for (int j = 0; j < Mp; j += NB) {
      const int below = Mp - (j + NB), nt = below > 0 ? (below + 63) / 64 : 0;
      const int tiles = nt * (nt + 1) / 2 + (1 + nt) * ((min(j + NB, Mp) + 63) / 64);
      max_tiles = tiles > max_tiles ? tiles : max_tiles;
    }
    constexpr int kMaxResident = 192;   // workgroups of 256 threads / ~50 KB LDS: two fit a CU, 256 CUs -- far inside what is resident at once
    int T = kMaxResident / batch - 1;
    if (T > max_tiles) T = max_tiles;
    if (T >= 1 && np < 4096) {
      const std::string sname = "chol_sync" + ctx->ws_tag;
      const size_t sync_bytes = (size_t)batch * (1 + np) * sizeof(unsigned);
      const bool fresh = ctx->ws.find(sname) == ctx->ws.end() || ctx->ws[sname].second < sync_bytes;
      unsigned* sync = (unsigned*)ws_get(ctx, sname, sync_bytes);
      double* Xpub = (double*)ws_get(ctx, "chol_Xpub" + ctx->ws_tag, (size_t)batch * np * NB * NB * sizeof(double));
      if (!sync || !Xpub) return DCGP_ERR_ALLOC;
      ChainEpoch& ep = ctx->chain_epochs[sname];
      if (fresh || ep.T != T || ep.np != np || ep.batch != batch) {   // counters are monotone across launches of ONE shape
        HIP_TRY(ctx, hipMemsetAsync(sync, 0, sync_bytes, ctx->stream));
        ep.epoch = 0; ep.T = T; ep.np = np; ep.batch = batch;
      }
      PcArgs p;
      p.r.A = d_A; p.r.Lout = Lout; p.r.Y = Y; p.r.Linv = d_Linv; p.r.LinvT = d_LinvT; p.r.Mp = Mp; p.r.ld = ld; p.r.info = d_info;
      p.r.j = p.r.nt = p.r.nT = p.r.nct = 0;
      p.np = np; p.T = T; p.Xpub = Xpub; p.sync = sync;
      p.base_chain = ep.epoch * 4096u; p.base_trail = ep.epoch * (unsigned)T;
      ++ep.epoch;
      hipLaunchKernelGGL(chol_persist_kernel, dim3(1 + T, batch), dim3(256), 0, ctx->stream, p);
      LAUNCH_CHECK(ctx);
      if (defer_finish) return DCGP_OK;
      return factor_finish_batched(ctx, d_A, batch, Mp, ld);
    }
  }
  const bool no_la = ctx->opt.chol_no_lookahead != 0;   // A/B switch
  const int np_la = (Mp + NB - 1) / NB;
  double* Xla = nullptr;   // [np][batch][2][NB][NB]: factor and inverse of every diagonal block, handed from launch to launch
  if (d_Linv && !no_la && np_la > 1) {
    Xla = (double*)ws_get(ctx, "chol_Xla" + ctx->ws_tag, (size_t)np_la * batch * 2 * NB * NB * sizeof(double));
    if (!Xla) return DCGP_ERR_ALLOC;
  }
  // The panel launches of one (matrix group, bank) never change: same pointer tables, same scratch, same grid -- Mp / 32 of them (8 at
  // M = 256, 32 at M = 1024), ~3.5 us of host time each.  They are captured once into a HIP graph per argument set and replayed with one
  // call (ctx option chain_graph; the captured kernels and their order are exactly the loop's).  MEASURED AND LEFT OFF: the host's enqueue
  // time per step falls (317 -> 200 us at M = 1024, 250 -> 205 us at cfg4) but a replay costs ~7 us of DEVICE time that the loop does not
  // (chain 89.5 -> 97 us at M = 256, 14.4 -> 21 us at M = 32: the graph's own begin / end packets), and the host was never what a step
  // waits for -- a panel launch is >= 10 us of device time against ~3.5 us to enqueue it.  cfg2 1222 -> 1212 steps/s, cfg1 5800 -> 5500.
  std::string gkey;
  const bool use_graph = ctx->opt.chain_graph != 0;   // (the family timer of this function brackets the replay from outside)
  if (use_graph) {
    char kb[256];
    snprintf(kb, sizeof kb, "%p:%p:%p:%d:%d:%d:%p:%p:%p:%p:%d", (const void*)d_A, (const void*)d_Linv, (const void*)d_LinvT, batch, Mp, ld,
             (void*)d_info, (void*)Lout, (void*)Y, (void*)Xla, (int)no_la);
    gkey = kb;
    auto it = ctx->chain_graphs.find(gkey);
    if (it != ctx->chain_graphs.end()) {
      HIP_TRY(ctx, hipGraphLaunch(it->second, ctx->stream));
      if (defer_finish) return DCGP_OK;
      return factor_finish_batched(ctx, d_A, batch, Mp, ld);
    }
    if (ctx->chain_graphs.size() >= 64) {   // models come and go: start over rather than grow without bound
      for (auto& kv : ctx->chain_graphs) hipGraphExecDestroy(kv.second);
      ctx->chain_graphs.clear();
    }
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) gkey.clear();   // no capture: the plain loop
  }
  const bool capturing = use_graph && !gkey.empty();
  // whatever leaves this function between here and hipStreamEndCapture must not leave the stream capturing (every later launch on it would fail)
  struct CaptureGuard {
    hipStream_t s; bool open;
    ~CaptureGuard() {
      if (!open) return;
      hipGraph_t g = nullptr;
      if (hipStreamEndCapture(s, &g) == hipSuccess && g) hipGraphDestroy(g);
    }
  } guard{ctx->stream, capturing};
  for (int j = 0; j < Mp; j += NB) {
    RlArgs a;
    a.A = d_A; a.Lout = Lout; a.Y = Y; a.Linv = d_Linv; a.LinvT = d_Linv ? d_LinvT : nullptr; a.Mp = Mp; a.ld = ld; a.j = j; a.info = d_info;
    const int below = Mp - (j + NB);
    a.nt = below > 0 ? (below + 63) / 64 : 0;
    a.nT = a.nt * (a.nt + 1) / 2;
    a.nct = d_Linv ? (min(j + NB, Mp) + 63) / 64 : 0;
    const int nY = d_Linv ? (1 + a.nt) * a.nct : 0;
    int gx = a.nT + nY;
    const int jp = j / NB;
    if (Xla) {
      const size_t slot = (size_t)batch * 2 * NB * NB;
      if (jp > 0) a.Xin = Xla + (size_t)jp * slot;
      if (jp + 1 < np_la) { a.Xnext = Xla + (size_t)(jp + 1) * slot; a.la_idx = gx; ++gx; }
    }
    if (d_rhs) {   // the right-hand sides: panel jp - 1 here, and the last panel too in the last launch
      const bool last = jp == np_la - 1;
      a.batch = batch; a.np = np_la; a.Xall = Xla;
      if (jp == 0 && Xla) a.Xself = Xla;
      if (jp > 0 || last) {
        a.rhs = d_rhs; a.rhs_p = jp - 1; a.rhs_last = last ? 1 : 0;
        const int below_p = jp > 0 ? Mp - NB * jp : 0;                 // rows below the lagged panel
        a.rhs_nt = below_p > 0 ? (below_p + 63) / 64 : 0;
        // one 64-row tile per workgroup while the launch stays inside the resident slots (~400 beside the chain's own tiles), else two
        const long w1 = (long)batch * (max_R * jp + 1) * (a.rhs_nt > 0 ? a.rhs_nt : 1);
        a.rhs_tpw = w1 > 400 ? 2 : 1;
        const int ng = a.rhs_nt > 0 ? (a.rhs_nt + a.rhs_tpw - 1) / a.rhs_tpw : 1;   // row groups (rhs_tile)
        a.rhs_nq = jp * ng + (last ? 1 : 0);                           // items per Lq_q: (column tile <= jp - 1, group), + the last panel's diagonal tile
        a.rhs_base = gx > 0 ? gx : 1;
        gx = a.rhs_base + max_R * a.rhs_nq + (jp > 0 ? ng : 1);        // ... + alpha's
      }
    }
    // gx == 0: last panel of a plain potrf -- only L_jj is left; one workgroup factors and publishes it
    a.gx_trace = gx > 0 ? gx : 1;
    if (Xla && !ctx->opt.chain_no_iso && ctx->chain_alone) {   // the look-ahead workgroups alone on one XCD (RlArgs::iso_per)
      a.batch = batch;
      a.iso_per = (gx > 0 ? gx : 1) - (a.la_idx >= 0 ? 1 : 0);
      const int items = batch * a.iso_per, slots = (items + 6) / 7;
      hipLaunchKernelGGL(chol_rl_kernel, dim3(8 * (slots > batch ? slots : batch)), dim3(256), 0, ctx->stream, a);
    } else
    hipLaunchKernelGGL(chol_rl_kernel, dim3(gx > 0 ? gx : 1, batch), dim3(256), 0, ctx->stream, a);
    if (!capturing) LAUNCH_CHECK(ctx);
  }
  if (capturing) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    guard.open = false;
    if (hipStreamEndCapture(ctx->stream, &graph) != hipSuccess || !graph)
      return ctx_fail(ctx, DCGP_ERR_HIP, "factorisation chain: graph capture failed");
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (ei != hipSuccess || !exec) return ctx_fail(ctx, DCGP_ERR_HIP, "factorisation chain: graph instantiation failed: %s", hipGetErrorString(ei));
    ctx->chain_graphs[gkey] = exec;
    HIP_TRY(ctx, hipGraphLaunch(exec, ctx->stream));
  }
  if (defer_finish) return DCGP_OK;   // inv(L) is complete; the caller copies the factor back (factor_finish_batched) off its critical path
  return factor_finish_batched(ctx, d_A, batch, Mp, ld);
}

// final factor from the chain's scratch back over A (lower triangle; strict upper triangle zeroed)
int factor_finish_batched(dcgp_ctx* ctx, double* const* d_A, int batch, int Mp, int ld) {
  if (batch <= 0) return DCGP_OK;
  auto it = ctx->ws.find("chol_Lout" + ctx->ws_tag);
  if (it == ctx->ws.end()) return ctx_fail(ctx, DCGP_ERR_ARG, "factor_finish: no factorisation to finish");
  hipLaunchKernelGGL(chol_finish_kernel, dim3(Mp, batch), dim3(128), 0, ctx->stream, d_A, (const double*)it->second.first, Mp, ld);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}
