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
  if (tid == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                    \
    g_chol_trace[(a.j / NB) * 32 + (blockIdx.x == 0 ? 0 : 16) + (k)] = wall_clock64();
#else
#define TR(k)
#endif

namespace {

struct RlArgs {
  double* const* A;      // working matrices (destroyed): trailing part updated in place
  double* Lout;          // [batch][Mp][ld] final factor (lower), then copied back over A by the finish kernel
  double* Y;             // [batch][Mp][ld] running inverse (scratch) or nullptr
  double* const* Linv;   // final inv(L) or nullptr; every element is written by the chain (zeros included)
  double* const* LinvT;  // its transpose, written in the same pass, or nullptr
  int Mp, ld, j, nt, nT, nct;
  int* info;
};

// The prologue loads are written as fully unrolled register batches (all global loads issued, then all LDS stores):
// as rolled loops each iteration waited for its own load -- 4 + 8 + 8 serial memory latencies per workgroup.
__device__ __forceinline__ void load_diag(const double* __restrict__ A, int ld, int j, int nb, double (*D)[NB + 1], int tid) {
  double t[NB * NB / 256];
#pragma unroll
  for (int e = 0; e < NB * NB / 256; ++e) {
    const int idx = tid + e * 256, r = idx / NB, c = idx % NB;
    t[e] = (r == c) ? 1.0 : 0.0;
    if (r < nb && c < nb && c <= r) t[e] = A[(long)(j + r) * ld + j + c];
  }
#pragma unroll
  for (int e = 0; e < NB * NB / 256; ++e) {
    const int idx = tid + e * 256;
    D[idx / NB][idx % NB] = t[e];
  }
}
// rows [r0, r0+64) of the panel columns -> U[64][33] (zero beyond the matrix)
__device__ __forceinline__ void load_panel_rows(const double* __restrict__ A, int ld, int Mp, int j, int nb, int r0,
                                                double (*U)[NB + 1], int tid) {
  double t[64 * NB / 256];
#pragma unroll
  for (int e = 0; e < 64 * NB / 256; ++e) {
    const int idx = tid + e * 256, i = idx / NB, c = idx % NB;
    t[e] = (r0 + i < Mp && c < nb) ? A[(long)(r0 + i) * ld + j + c] : 0.0;
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

__global__ __launch_bounds__(256, 2) void chol_rl_kernel(RlArgs a) {
  __shared__ double D[NB][NB + 1];
  __shared__ double col[2 * NB];
  __shared__ double Ui[64][NB + 1];
  __shared__ double UcTs[64 * (NB + 1)];   // trailing tiles: second panel row block; inverse tiles: the Y tile
  __shared__ double Xs[NB][NB + 1];   // inv(L_jj)
  double (*Uc)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(UcTs);
  double (*Ts)[64 + 1] = reinterpret_cast<double (*)[64 + 1]>(UcTs);   // [NB][65]: old Y rows, then new
  static_assert(NB * 65 <= 64 * (NB + 1), "Y tile fits the shared slot");
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lrow = lane >> 4, lcol = lane & 15;
  const int Mp = a.Mp, ld = a.ld, j = a.j, nb = min(NB, Mp - j);
  double* __restrict__ A = a.A[b];
  double* __restrict__ Lout = a.Lout + (long)b * Mp * ld;
  double* __restrict__ Y = a.Y ? a.Y + (long)b * Mp * ld : nullptr;
  const int below0 = j + NB;   // first row below the panel block

  // ---- prologue: every global read of this workgroup is issued here, in one latency ----
  TR(0)
  load_diag(A, ld, j, nb, D, tid);
  const bool only_diag = a.nT == 0 && a.nct == 0;
  const bool is_trailing = (int)blockIdx.x < a.nT;
  int ti = 0, tc = 0, rt = -1, ct = 0;
  double old[2][2][4];   // values the final read-modify-write subtracts from
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int v = 0; v < 4; ++v) old[x][y][v] = 0.0;
  if (only_diag) {
  } else if (is_trailing) {
    int pair = blockIdx.x;
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
    const int yy = blockIdx.x - a.nT;
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
  // ---- L_jj and inv(L_jj): one wavefront ----
  if (tid < 64) {
    double v[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) v[c] = (tid < NB) ? D[tid][c] : ((c == tid - NB) ? 1.0 : 0.0);
    const int fail = wave_potrf_inv32(v, tid, col);
    if (tid < NB) {
#pragma unroll
      for (int c = 0; c < NB; ++c) D[tid][c] = (c <= tid) ? v[c] : 0.0;
    } else {
#pragma unroll
      for (int r = 0; r < NB; ++r) Xs[r][tid - NB] = v[r];
    }
    if (tid == 0 && blockIdx.x == 0 && (j == 0 || (fail && a.info[b] == 0))) a.info[b] = fail ? j + fail : 0;
  }
  __syncthreads();
  TR(2)
  if (blockIdx.x == 0) {   // publish L_jj (final)
    for (int idx = tid; idx < nb * nb; idx += 256) {
      const int r = idx / nb, c = idx % nb;
      Lout[(long)(j + r) * ld + j + c] = D[r][c];
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
          if (i < Mp && jj < Mp && jj <= i) A[(long)i * ld + jj] = old[x][y][v] - acc[x][y][v];
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

// final factor back over A (lower triangle from Lout, strict upper triangle zero).  grid (Mp, batch)
__global__ void chol_finish_kernel(double* const* __restrict__ Ap, const double* __restrict__ Lout, int Mp, int ld) {
  double* __restrict__ A = Ap[blockIdx.y];
  const double* __restrict__ L = Lout + (long)blockIdx.y * Mp * ld;
  const int i = blockIdx.x;
  for (int c = threadIdx.x; c < Mp; c += blockDim.x) A[(long)i * ld + c] = (c <= i) ? L[(long)i * ld + c] : 0.0;
}

}  // namespace

// Cholesky factor in place (strict upper triangle zeroed) and, when d_Linv != nullptr, inv(L) (+ its transpose).
int factor_inverse_batched(dcgp_ctx* ctx, double* const* d_A, double* const* d_Linv, double* const* d_LinvT, int batch,
                           int Mp, int ld, int* d_info, bool defer_finish) {
  if (batch <= 0) return DCGP_OK;
  ScopedTimer t(ctx, "factor_chain");
  const size_t mm = (size_t)Mp * ld;
  double* Lout = (double*)ws_get(ctx, "chol_Lout" + ctx->ws_tag, (size_t)batch * mm * sizeof(double));
  double* Y = d_Linv ? (double*)ws_get(ctx, "chol_Y" + ctx->ws_tag, (size_t)batch * mm * sizeof(double)) : nullptr;
  if (!Lout || (d_Linv && !Y)) return DCGP_ERR_ALLOC;
  for (int j = 0; j < Mp; j += NB) {
    RlArgs a;
    a.A = d_A; a.Lout = Lout; a.Y = Y; a.Linv = d_Linv; a.LinvT = d_Linv ? d_LinvT : nullptr; a.Mp = Mp; a.ld = ld; a.j = j; a.info = d_info;
    const int below = Mp - (j + NB);
    a.nt = below > 0 ? (below + 63) / 64 : 0;
    a.nT = a.nt * (a.nt + 1) / 2;
    a.nct = d_Linv ? (min(j + NB, Mp) + 63) / 64 : 0;
    const int nY = d_Linv ? (1 + a.nt) * a.nct : 0;
    const int gx = a.nT + nY;
    // gx == 0: last panel of a plain potrf -- only L_jj is left; one workgroup factors and publishes it
    hipLaunchKernelGGL(chol_rl_kernel, dim3(gx > 0 ? gx : 1, batch), dim3(256), 0, ctx->stream, a);
    LAUNCH_CHECK(ctx);
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
