// chol.hip -- batched dense M x M factorisation stage (fp64), replicated on every GPU.
//
//   * potrf_batched : blocked right-looking Cholesky (tf.cholesky at conv_gp/conditionals.py:29 and
//                     conv_gp/layers.py:151,156).  Panel width 32: the 32x32 diagonal block is factored by
//                     ONE wavefront (lane r owns row r in registers; the scaled pivot column goes through a
//                     32-entry LDS line that every lane reads back as a broadcast -- no barriers, no divisions:
//                     v_rsq_f64 + Newton), the rows below are solved against it one row per lane
//                     (wavefront-level trsm panel), and the trailing matrix gets a rank-32 update on the
//                     matrix cores (v_mfma_f64_16x16x4_f64, operands staged through LDS), lower tiles only.
//   * trtri_batched : inverse of the lower factor by recursive doubling
//                     inv([A 0; C B]) = [inv(A) 0; -inv(B) C inv(A)  inv(B)]   (log2(M/32) levels, every
//                     level a batch of independent MFMA products) -- the triangular solves of
//                     conv_gp/conditionals.py:31-33,44-47 are then applied as products with inv(L).
//   All matrices of a model (every layer's Kuu and KL prior) go through ONE batched call per step so the
//   serial panel chain is paid once.
#include "chol_dev.h"

namespace {

using namespace chol_dev;

// ---------------------------------------------------------------------------------------------
// potrf panel: factor diag block (wave 0) + row-per-lane trsm of the rows below
// grid (max(1, ceil(rows_below / 256)), batch), block 256
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void potrf_panel_kernel(double* const* __restrict__ ptrs, int Mp, int ld, int j,
                                                           int* __restrict__ info) {
  __shared__ double D[NB][NB + 1];
  __shared__ double Dr[NB];    // reciprocal diagonal of the factored block
  __shared__ double col[NB];   // pivot-column broadcast line
  double* __restrict__ A = ptrs[blockIdx.y];
  const int tid = threadIdx.x;
  const int nb = min(NB, Mp - j);
  for (int idx = tid; idx < NB * NB; idx += 256) {
    int r = idx / NB, c = idx % NB;
    double v = (r == c) ? 1.0 : 0.0;
    if (r < nb && c < nb && c <= r) v = A[(long)(j + r) * ld + j + c];
    D[r][c] = v;
  }
  __syncthreads();
  if (tid < 64) {
    const int r = tid & 31;
    double a[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) a[c] = D[r][c];
    const int fail = wave_potrf32(a, r, col);
    if (tid < 32) {
#pragma unroll
      for (int c = 0; c < NB; ++c) D[r][c] = (c <= r) ? a[c] : 0.0;
      double diag = a[0];
#pragma unroll
      for (int c = 1; c < NB; ++c) diag = (c == r) ? a[c] : diag;
      Dr[r] = 1.0 / diag;
    }
    if (tid == 0 && fail && blockIdx.x == 0 && info[blockIdx.y] == 0) info[blockIdx.y] = j + fail;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    // write the factored diagonal block back and zero the strict upper part to its right
    for (int idx = tid; idx < nb * nb; idx += 256) {
      int r = idx / nb, c = idx % nb;
      A[(long)(j + r) * ld + j + c] = D[r][c];
    }
    const int right = Mp - (j + nb);
    for (int idx = tid; idx < nb * right; idx += 256) {
      int r = idx / right, c = idx % right;
      A[(long)(j + r) * ld + j + nb + c] = 0.0;
    }
  }
  // rows below: x * L11^T = a  (right-looking forward substitution, one row per thread)
  const int row = j + NB + blockIdx.x * 256 + tid;
  if (row < Mp) {
    double x[NB];
    double* __restrict__ Ar = A + (long)row * ld + j;
#pragma unroll
    for (int c = 0; c < NB; c += 2) {
      double2 v = *reinterpret_cast<const double2*>(Ar + c);
      x[c] = v.x;
      x[c + 1] = v.y;
    }
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      x[c] = x[c] * Dr[c];
#pragma unroll
      for (int q = c + 1; q < NB; ++q) x[q] = fma(-x[c], D[q][c], x[q]);
    }
#pragma unroll
    for (int c = 0; c < NB; c += 2) *reinterpret_cast<double2*>(Ar + c) = double2{x[c], x[c + 1]};
  }
}

// trailing update A22 -= L21 L21^T, lower 64x64 tiles only. grid (tile pairs, batch)
__global__ __launch_bounds__(256, 4) void potrf_update_kernel(double* const* __restrict__ ptrs, int Mp, int ld, int j,
                                                               int nt) {
  __shared__ TileLds t;
  double* __restrict__ A = ptrs[blockIdx.y];
  int pair = blockIdx.x, tc = 0;
  while (pair >= nt - tc) {   // column-major enumeration of the lower triangle: tc <= tr
    pair -= nt - tc;
    ++tc;
  }
  const int tr = tc + pair;
  const int base = j + NB;
  const int r0 = base + tr * 64, c0 = base + tc * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lrow = lane >> 4, lcol = lane & 15;
  d4 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};
  tile64_mfma(t, A + (long)r0 * ld + j, ld, 1, Mp - r0, A + (long)c0 * ld + j, 1, ld, Mp - c0, NB, tid, acc);
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        int i = r0 + wm * 32 + x * 16 + lrow + 4 * v, jj = c0 + wn * 32 + y * 16 + lcol;
        if (i < Mp && jj < Mp && jj <= i) A[(long)i * ld + jj] -= acc[x][y][v];
      }
}

// ---------------------------------------------------------------------------------------------
// trtri
// ---------------------------------------------------------------------------------------------
// level 0: invert every 32x32 diagonal block (lane c solves column c); the rest of the block row of X is zeroed.
__global__ __launch_bounds__(64) void trtri_diag_kernel(double* const* __restrict__ Lp, double* const* __restrict__ Xp,
                                                         int Mp, int ld) {
  __shared__ double D[NB][NB + 1];
  __shared__ double Xs[NB][NB + 1];
  __shared__ double Dr[NB];
  const double* __restrict__ L = Lp[blockIdx.y];
  double* __restrict__ X = Xp[blockIdx.y];
  const int s = blockIdx.x * NB, nb = min(NB, Mp - s), tid = threadIdx.x;
  for (int idx = tid; idx < NB * NB; idx += 64) {
    int r = idx / NB, c = idx % NB;
    double v = (r == c) ? 1.0 : 0.0;
    if (r < nb && c < nb && c <= r) v = L[(long)(s + r) * ld + s + c];
    D[r][c] = v;
  }
  __syncthreads();
  if (tid < NB) Dr[tid] = 1.0 / D[tid][tid];
  __syncthreads();
  if (tid < NB) {
    double x[NB];
    lane_trtri32(D, Dr, tid, x);
#pragma unroll
    for (int r = 0; r < NB; ++r) Xs[r][tid] = x[r];
  }
  __syncthreads();
  // write the block row of X: zeros left/right of the diagonal block, inverse on it
  for (int idx = tid; idx < nb * Mp; idx += 64) {
    int r = idx / Mp, c = idx % Mp;
    double v = 0.0;
    if (c >= s && c < s + nb) v = Xs[r][c - s];
    X[(long)(s + r) * ld + c] = v;
  }
}

// one merge level: mode 0: T = C * inv(A);  mode 1: X21 = -inv(B) * T.   grid (tiles, pairs, batch)
__global__ __launch_bounds__(256, 4) void trtri_merge_kernel(double* const* __restrict__ Lp, double* const* __restrict__ Xp,
                                                              double* __restrict__ Tbase, long Tstride, int Mp, int ld,
                                                              int h, int mode) {
  __shared__ TileLds t;
  const double* __restrict__ L = Lp[blockIdx.z];
  double* __restrict__ X = Xp[blockIdx.z];
  double* __restrict__ T = Tbase + (long)blockIdx.z * Tstride;
  const int s = blockIdx.y * 2 * h;
  const int hb = min(h, Mp - (s + h));   // rows of the lower block
  if (hb <= 0) return;
  const int tiles_n = (h + 63) / 64;
  const int ti = blockIdx.x / tiles_n, tj = blockIdx.x % tiles_n;
  if (ti * 64 >= hb) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lrow = lane >> 4, lcol = lane & 15;
  d4 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};
  const long o21 = (long)(s + h) * ld + s;
  double* out;
  double alpha;
  if (mode == 0) {
    // T[hb x h] = C[hb x h] * Ainv[h x h]
    tile64_mfma(t, L + o21 + (long)ti * 64 * ld, ld, 1, hb - ti * 64, X + (long)s * ld + s + tj * 64, ld, 1, h - tj * 64, h,
                tid, acc);
    out = T + o21;
    alpha = 1.0;
  } else {
    // X21[hb x h] = -Binv[hb x hb] * T[hb x h]
    tile64_mfma(t, X + (long)(s + h) * ld + (s + h) + (long)ti * 64 * ld, ld, 1, hb - ti * 64, T + o21 + tj * 64, ld, 1,
                h - tj * 64, hb, tid, acc);
    out = X + o21;
    alpha = -1.0;
  }
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        int i = ti * 64 + wm * 32 + x * 16 + lrow + 4 * v, jj = tj * 64 + wn * 32 + y * 16 + lcol;
        if (i < hb && jj < h) out[(long)i * ld + jj] = alpha * acc[x][y][v];
      }
}

__global__ void transpose_kernel(double* const* __restrict__ Sp, double* const* __restrict__ Dp, int Mp, int ld) {
  __shared__ double t[32][33];
  const double* __restrict__ S = Sp[blockIdx.z];
  double* __restrict__ D = Dp[blockIdx.z];
  int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    int i = by + r, j = bx + threadIdx.x;
    t[r][threadIdx.x] = (i < Mp && j < Mp) ? S[(long)i * ld + j] : 0.0;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    int i = bx + r, j = by + threadIdx.x;
    if (i < Mp && j < Mp) D[(long)i * ld + j] = t[threadIdx.x][r];
  }
}

// copy with zero/identity padding and optional lower-triangular masking (matrix_band_part(., -1, 0))
__global__ void pad_copy_kernel(const double* __restrict__ src, int rows, int cols, int lds, double* __restrict__ dst,
                                int ldd, int rows_p, int cols_p, int mode, long src_batch, long dst_batch) {
  int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= cols_p || i >= rows_p) return;
  const double* s = src + (long)blockIdx.z * src_batch;
  double* d = dst + (long)blockIdx.z * dst_batch;
  double v = 0.0;
  if (i < rows && j < cols) {
    v = s[(long)i * lds + j];
    if (mode == 1 && j > i) v = 0.0;
  } else if (mode == 2 && i == j) {
    v = 1.0;
  }
  d[(long)i * ldd + j] = v;
}

}  // namespace

int pad_copy(dcgp_ctx* ctx, const double* src, int rows, int cols, int lds, double* dst, int ldd, int rows_p,
             int cols_p, int mode, int batch, long src_batch, long dst_batch) {
  dim3 grid((cols_p + 127) / 128, rows_p, batch);
  hipLaunchKernelGGL(pad_copy_kernel, grid, dim3(128), 0, ctx->stream, src, rows, cols, lds, dst, ldd, rows_p, cols_p,
                     mode, src_batch, dst_batch);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

int potrf_batched(dcgp_ctx* ctx, double* const* d_ptrs, double** /*h_ptrs*/, int batch, int Mp, int ld, int* d_info) {
  if (batch <= 0) return DCGP_OK;
  ScopedTimer t(ctx, "potrf");
  HIP_TRY(ctx, hipMemsetAsync(d_info, 0, sizeof(int) * batch, ctx->stream));
  for (int j = 0; j < Mp; j += NB) {
    const int below = Mp - (j + NB);
    const int gx = below > 0 ? (below + 255) / 256 : 1;
    hipLaunchKernelGGL(potrf_panel_kernel, dim3(gx, batch), dim3(256), 0, ctx->stream, d_ptrs, Mp, ld, j, d_info);
    LAUNCH_CHECK(ctx);
    if (below > 0) {
      const int nt = (below + 63) / 64;
      hipLaunchKernelGGL(potrf_update_kernel, dim3(nt * (nt + 1) / 2, batch), dim3(256), 0, ctx->stream, d_ptrs, Mp, ld,
                         j, nt);
      LAUNCH_CHECK(ctx);
    }
  }
  return DCGP_OK;
}

int trtri_batched(dcgp_ctx* ctx, double* const* d_L, double* const* d_Linv, double* const* d_LinvT, int batch, int Mp,
                  int ld) {
  if (batch <= 0) return DCGP_OK;
  ScopedTimer t(ctx, "trtri");
  // scratch T: one Mp x ld matrix per batch entry
  double* Tbuf = (double*)ws_get(ctx, "trtri_T", (size_t)batch * Mp * ld * sizeof(double));
  if (!Tbuf) return DCGP_ERR_ALLOC;
  const int nblk = (Mp + NB - 1) / NB;
  hipLaunchKernelGGL(trtri_diag_kernel, dim3(nblk, batch), dim3(64), 0, ctx->stream, d_L, d_Linv, Mp, ld);
  LAUNCH_CHECK(ctx);
  for (int h2 = NB; h2 < Mp; h2 *= 2) {
    const int pairs = (Mp + 2 * h2 - 1) / (2 * h2);
    const int tiles = ((h2 + 63) / 64) * ((h2 + 63) / 64);
    for (int mode = 0; mode < 2; ++mode) {
      hipLaunchKernelGGL(trtri_merge_kernel, dim3(tiles, pairs, batch), dim3(256), 0, ctx->stream, d_L, d_Linv,
                         Tbuf, (long)Mp * ld, Mp, ld, h2, mode);
      LAUNCH_CHECK(ctx);
    }
  }
  if (d_LinvT) {
    dim3 grid((Mp + 31) / 32, (Mp + 31) / 32, batch);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, ctx->stream, d_Linv, d_LinvT, Mp, ld);
    LAUNCH_CHECK(ctx);
  }
  return DCGP_OK;
}
