// Internal: general strided fp64 GEMM (gemm_gen.hip), the workhorse of the backward pass.
#pragma once
#include "common.h"

// C_b(i, j) (+)= alpha * colscale_b[j] * (sum_k A_b(i, k) kscale_b[k] B_b(k, j) - sub_v_b[i] sub_x_b(i, j))
//   A_b(i, k) = A[b * a_bs + i * a_rs + k * a_cs],  B_b(k, j) = B[b * b_bs + k * b_rs + j * b_cs],
//   C_b(i, j) = C[b * c_bs + i * c_rs + j],         colscale_b[j] = colscale[b * cs_bs + j * cs_s] (optional)
struct GenGemm {
  const double* A = nullptr; long a_rs = 0, a_cs = 0, a_bs = 0;
  const double* B = nullptr; long b_rs = 0, b_cs = 0, b_bs = 0;
  double* C = nullptr; long c_rs = 0, c_bs = 0;
  int M = 0, N = 0, K = 0, batch = 1;
  double alpha = 1.0;
  int accumulate = 0;      // C += ... instead of C = ...
  const double* colscale = nullptr; long cs_s = 0, cs_bs = 0;
  const double* kscale = nullptr; long ks_s = 0, ks_bs = 0;   // B_b(k, j) is read as B_b(k, j) * kscale[b * ks_bs + k * ks_s]
  // optional rank-one-per-row correction in the epilogue: the kernel adjoints' "E X - rowsum(E) o Z" (grad.hip) without a second pass
  const double* sub_v = nullptr; long sv_bs = 0;               // sub_v_b[i] = sub_v[b * sv_bs + i]
  const double* sub_x = nullptr; long sx_rs = 0, sx_bs = 0;    // sub_x_b(i, j) = sub_x[b * sx_bs + i * sx_rs + j]
  int lower_only = 0;      // entries with j > i are written as 0 (before accumulation)
  int mirror = 0;          // with lower_only on a square result: the strictly lower entries are also stored transposed (a symmetric product, computed once)
  int phi = 0;             // Murray's Phi on the result (before accumulation): strictly lower part kept, diagonal halved, the rest zero
  int lower_compact = 0;   // set by the launcher: the grid enumerates the tiles on / below the diagonal only
};
int gemm_gen(dcgp_ctx* ctx, const GenGemm& g);
