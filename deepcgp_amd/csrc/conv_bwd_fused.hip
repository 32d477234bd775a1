// conv_bwd_fused.hip -- the reverse pass of the conditional's column-wise part for ONE strip of 16 FN columns in ONE workgroup
// (the adjoint of conv_gp/conditionals.py:31-65 with respect to K_uf, given d mean and d var of layers.py:128-134).
//
// With A1 = inv(L) K_uf, T_r = G_r^T A1, var_r = Knn - sum_m A1^2 + sum_m T_r^2 and mean_r = alpha_r^T A1, the adjoint is
//     dT_r  = 2 T_r o gv_r                                   (o: every column j scaled by gv[j][r])
//     dA1   = sum_r G_r dT_r + alpha gm^T - 2 A1 o gvs       (gvs[j] = sum_r gv[j][r])
//     dK_uf = inv(L)^T dA1.
// The launch-per-product form stores dT (R x the size of A1: 943 MB at the headline size), reads it back for a DENSE
// stacked product (G_r is triangular, the stacked operand is not) and makes three more passes over [M x columns] operands.
// Scaling columns commutes with the products, so
//     dA1 = sum_r (S_r A1) o (2 gv_r) + ...,        S_r = G_r G_r^T  (M x M, symmetric, parameter-only: R small products per step)
// -- ONE dense product per r on the LDS-resident strip of A1, exactly the forward kernel's second product (conv_fused.hip)
// with a scaled accumulation in place of the sum of squares; T and dT are never formed.  Same machinery: the strip
// [Mp][64] in LDS (row quads) as the B operand, the A operand (S_r, then inv(L)) streamed from L2 straight into MFMA A
// registers two k-tiles ahead, wave w owning the 16 rows of fragment w for all r (dense S_r: every wave carries the same
// R * Mp/16 k-tiles), no barrier inside the k loops.  dA1 leaves the registers once, into the strip, as the B operand of
// the closing triangular product; HBM traffic is the strip of A1 in and the strip of dK_uf out.
// Strip width: 64 columns (FN = 4) where that still gives the chip a round of workgroups; the de-duplicated first layer of a
// training step has a tenth of the columns (4608 at the headline size: 72 strips of 64 on 256 CUs, 332 us) and takes 32-column
// strips -- twice the workgroups on half the MFMA chain each, the S_r stream (L2) per workgroup unchanged.
#include "layer.h"

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int CB_NT = 1024, CB_D = 2;
#define CB_SB __builtin_amdgcn_sched_barrier(0)   // the LDS reads stay one sub-step ahead of the MFMAs, as written (see conv_fused.hip)

template <int CB_FN>
__global__ __launch_bounds__(CB_NT) void conv_bwd_fused_kernel(ConvBwdArgs a) {
  constexpr int CB_BN = 16 * CB_FN, CB_SH = CB_FN == 4 ? 6 : (CB_FN == 2 ? 5 : 4);
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int Mp = a.Mp, nf = Mp >> 4, R = a.R, Rk = (R + 3) & ~3;
  const int tid = threadIdx.x, lane = tid & 63, lrow = lane >> 4, lcol = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool live = wave < nf;
  const int fw = live ? wave : nf - 1;                  // idle waves shadow the last fragment (loads in range, results dropped)
  // the strip in LDS in row quads, as in conv_fused.hip: element (row k, column 16 y + i) at (k >> 2) * 4 BN + 64 y + 16 (k & 3) + i -- a B fragment is 512
  // contiguous bytes (lane l at 8 l), every address the lane index + immediates + a scalar: one VALU instruction per k-tile where the XOR-swizzled rows took ten
  constexpr int QS = 4 * CB_BN;
  double* strip = smem;                                 // [Mp / 4][FN][4][16]
  double* gvl = strip + (long)Mp * CB_BN;               // [R][64]   2 gv[j][r]
  double* gml = gvl + R * CB_BN;                        // [Rk][64]  gm[j][r], zero rows beyond R
  double* gsl = gml + Rk * CB_BN;                       // [64]      -2 gvs[j]
  const long j0 = (long)blockIdx.x * CB_BN;

  // ---- the strip of A1 and the strip's upstream gradients -> LDS ----
  for (int i0 = 0; i0 < Mp * CB_BN; i0 += 8 * CB_NT) {
    double t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = i0 + e * CB_NT + tid, m = idx >> CB_SH, c = idx & (CB_BN - 1);
      t[e] = (idx < Mp * CB_BN && j0 + c < a.Kc) ? a.A1[(long)m * a.ld + j0 + c] : 0.0;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = i0 + e * CB_NT + tid, m = idx >> CB_SH, c = idx & (CB_BN - 1);
      if (idx < Mp * CB_BN) strip[(m >> 2) * QS + (c >> 4) * 64 + (m & 3) * 16 + (c & 15)] = t[e];
    }
  }
  for (int idx = tid; idx < Rk * CB_BN; idx += CB_NT) {
    const int c = idx / Rk, r = idx - c * Rk;             // consecutive threads walk the R values of a column (contiguous in gv / gm)
    const bool in = r < R && j0 + c < a.Kc;
    if (r < R) gvl[r * CB_BN + c] = in ? 2.0 * a.gv[(j0 + c) * R + r] : 0.0;
    gml[r * CB_BN + c] = in ? a.gm[(j0 + c) * R + r] : 0.0;
  }
  if (tid < CB_BN) gsl[tid] = (j0 + tid < a.Kc) ? -2.0 * a.gvs[j0 + tid] : 0.0;

  unsigned voff[4];   // lane (lrow, lcol) of k-substep q of a k-tile needs Wt[16 kt + 4q + lrow][16 f + lcol]
#pragma unroll
  for (int q = 0; q < 4; ++q) voff[q] = (unsigned)(((4 * q + lrow) * Mp + lcol) * 8);
  auto ldw = [&](const __amdgpu_buffer_rsrc_t& rs, int soff, double (&dst)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff[q], soff, 0);
      __builtin_memcpy(&dst[q], &v, 8);
    }
  };
  d4 acc[CB_FN], dA[CB_FN];
#pragma unroll
  for (int y = 0; y < CB_FN; ++y) { acc[y] = d4{0.0, 0.0, 0.0, 0.0}; dA[y] = d4{0.0, 0.0, 0.0, 0.0}; }
  auto ldb = [&](int kt, int q, double (&dst)[CB_FN]) {
    const double* b = strip + (kt * 4 + q) * QS + lane;
#pragma unroll
    for (int y = 0; y < CB_FN; ++y) dst[y] = b[y * 64];
  };
  auto mf = [&](double w, const double (&b)[CB_FN]) {
#pragma unroll
    for (int y = 0; y < CB_FN; ++y) acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, b[y], acc[y], 0, 0, 0);
  };
  // one k-tile: b0 holds sub-step 0 of tile kt on entry and of tile kt_next on exit
  auto tile = [&](int kt, int kt_next, const double (&w)[4], double (&b0)[CB_FN]) {
    double b1[CB_FN];
    ldb(kt, 1, b1);
    CB_SB;
    mf(w[0], b0);
    CB_SB;
    ldb(kt, 2, b0);
    CB_SB;
    mf(w[1], b1);
    CB_SB;
    ldb(kt, 3, b1);
    CB_SB;
    mf(w[2], b0);
    CB_SB;
    ldb(kt_next, 0, b0);
    CB_SB;
    mf(w[3], b1);
    CB_SB;
  };
  __syncthreads();   // strip resident

  // ---- dA1 = sum_r (S_r A1) o (2 gv_r): one flat stream over (r, k-tile), the load cursor CB_D tiles ahead ----
  {
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.S), 0, R * Mp * Mp * 8, 0x00020000);
    const int total = R * nf;
    auto soff = [&](int t) { return (((t / nf) * Mp + (t % nf) * 16) * Mp + 16 * fw) * 8; };
    double ring[CB_D + 1][4], b0[CB_FN];
#pragma unroll
    for (int u = 0; u < CB_D; ++u) ldw(srs, soff(min(u, total - 1)), ring[u]);
    ldb(0, 0, b0);
    int kt = 0, r = 0;
    auto step = [&](const double (&w)[4]) {
      const int kcur = kt;
      const bool r_end = kcur == nf - 1;
      kt = r_end ? 0 : kt + 1;
      tile(kcur, kt, w, b0);
      if (r_end) {   // S_r A1 of this wave's rows is complete: scale its columns by 2 gv[j][r] into the running dA1
#pragma unroll
        for (int y = 0; y < CB_FN; ++y) {
          const double sc = gvl[r * CB_BN + y * 16 + lcol];
#pragma unroll
          for (int v = 0; v < 4; ++v) dA[y][v] = fma(acc[y][v], sc, dA[y][v]);
          acc[y] = d4{0.0, 0.0, 0.0, 0.0};
        }
        ++r;
      }
    };
    int t = 0;
    for (; t + CB_D + 1 <= total; t += CB_D + 1) {   // full groups: no conditionals around the loads
#pragma unroll
      for (int u = 0; u <= CB_D; ++u) {
        ldw(srs, soff(min(t + u + CB_D, total - 1)), ring[(u + CB_D) % (CB_D + 1)]);
        step(ring[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < CB_D; ++u)
      if (t + u < total) step(ring[u]);
  }
  // ---- + alpha gm^T (k = the R outputs, zero padded to Rk) - 2 A1 o gvs ----
  {
    const double* __restrict__ al = a.alpha + (long)(16 * fw + lcol) * a.Rp + lrow;
    for (int q = 0; q < Rk / 4; ++q) {
      const double w = (4 * q + lrow < R) ? al[4 * q] : 0.0;
      const double* b = gml + (4 * q + lrow) * CB_BN + lcol;
#pragma unroll
      for (int y = 0; y < CB_FN; ++y) dA[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, b[y * 16], dA[y], 0, 0, 0);
    }
#pragma unroll
    for (int y = 0; y < CB_FN; ++y) {
      const double gs = gsl[y * 16 + lcol];
#pragma unroll
      for (int v = 0; v < 4; ++v) dA[y][v] = fma(gs, strip[(4 * fw + v) * QS + y * 64 + lane], dA[y][v]);
    }
  }
  __syncthreads();   // every wave is done with the strip of A1
  if (live) {
#pragma unroll
    for (int y = 0; y < CB_FN; ++y)
#pragma unroll
      for (int v = 0; v < 4; ++v) strip[(4 * fw + v) * QS + y * 64 + lane] = dA[y][v];
  }
  __syncthreads();   // dA1 published
  // ---- dK_uf = inv(L)^T dA1: Wt[k][i] = inv(L)[k][i], upper-triangular product (k-tiles fw .. nf-1) ----
  {
    const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.Linv), 0, Mp * Mp * 8, 0x00020000);
    const int total = nf - fw;
    auto soff = [&](int t) { return ((fw + t) * 16 * Mp + 16 * fw) * 8; };
    double ring[CB_D + 1][4], b0[CB_FN];
#pragma unroll
    for (int u = 0; u < CB_D; ++u) ldw(lrs, soff(min(u, total - 1)), ring[u]);
    ldb(fw, 0, b0);
    int kt = fw;
    auto step = [&](const double (&w)[4]) {
      const int kcur = kt;
      kt = min(kt + 1, nf - 1);
      tile(kcur, kt, w, b0);
    };
    int t = 0;
    for (; t + CB_D + 1 <= total; t += CB_D + 1) {
#pragma unroll
      for (int u = 0; u <= CB_D; ++u) {
        ldw(lrs, soff(min(t + u + CB_D, total - 1)), ring[(u + CB_D) % (CB_D + 1)]);
        step(ring[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < CB_D; ++u)
      if (t + u < total) step(ring[u]);
  }
  if (live) {
#pragma unroll
    for (int y = 0; y < CB_FN; ++y)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const long j = j0 + y * 16 + lcol;
        if (j < a.Kc) a.dKuf[(long)(16 * fw + lrow + 4 * v) * a.ld + j] = acc[y][v];
      }
  }
}

}  // namespace

// strip width in 16-column fragments: 4 while that fills a round of the chip, 2 for a few thousand columns, 1 for a few hundred (the head)
static int strip_frags(const dcgp_ctx* ctx, const ConvBwdArgs& a) {
  if (ctx->opt.fused_bwd_frags == 4 || ctx->opt.fused_bwd_frags == 2 || ctx->opt.fused_bwd_frags == 1) return ctx->opt.fused_bwd_frags;   // A/B switch
  return (a.Kc + 63) / 64 >= 200 ? 4 : ((a.Kc + 31) / 32 >= 64 ? 2 : 1);
}

bool conv_bwd_fused_ok(const dcgp_ctx* ctx, const ConvBwdArgs& a) {
  if (ctx->opt.no_fused_bwd) return false;
  const int Rk = (a.R + 3) & ~3;
  const size_t lds = ((size_t)a.Mp * 64 + (size_t)(a.R + Rk + 1) * 64) * sizeof(double);
  return a.Mp >= 16 && a.Mp <= 256 && a.Mp % 16 == 0 && a.R >= 1 && a.R <= 16 && lds <= 160 * 1024 &&
         (long)a.R * a.Mp * a.Mp * 8 < (1L << 31);
}

int conv_bwd_fused(dcgp_ctx* ctx, const ConvBwdArgs& a) {
  if (a.Kc <= 0) return DCGP_OK;
  if (!conv_bwd_fused_ok(ctx, a)) return ctx_fail(ctx, DCGP_ERR_ARG, "conv_bwd_fused: layer shape not supported (M = %d, R = %d)", a.M, a.R);
  const int Rk = (a.R + 3) & ~3, fn = strip_frags(ctx, a), bn = 16 * fn;
  const size_t lds = ((size_t)a.Mp * bn + (size_t)(a.R + Rk + 1) * bn) * sizeof(double);
  static bool attr[64] = {};   // per device
  const int dv = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
  if (!attr[dv]) {
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)conv_bwd_fused_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)conv_bwd_fused_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)conv_bwd_fused_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr[dv] = true;
  }
  ScopedTimer t(ctx, "conv_bwd_fused");
  const dim3 grid((unsigned)((a.Kc + bn - 1) / bn));
  if (fn == 4) hipLaunchKernelGGL(conv_bwd_fused_kernel<4>, grid, dim3(CB_NT), lds, ctx->stream, a);
  else if (fn == 2) hipLaunchKernelGGL(conv_bwd_fused_kernel<2>, grid, dim3(CB_NT), lds, ctx->stream, a);
  else hipLaunchKernelGGL(conv_bwd_fused_kernel<1>, grid, dim3(CB_NT), lds, ctx->stream, a);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}
