// head_cond.hip -- the whole conditional of a FEW-column problem (the head: one column per image, a few hundred in
// all) in ONE launch, for M <= 512 (round 5: two row fragments per wave).
//
// The generic route is four dependent launches -- A1 = inv(L) Kzx (+ sum of squares), T_r = G_r^T A1 (+ sum of
// squares), mean = alpha^T A1, finalize -- of 13-17 us each at cfg2, every one of them nothing but a serial chain of
// 16 k-steps that each wait a full memory latency: ~75 us with the gaps, at the very end of the step where nothing
// can hide it.  Here a workgroup owns (16 columns, one output r) and runs both triangular products back to back:
//     * the 16-column strip of Kzx is LDS-resident ([k][16], 32 KB at M = 256), and A1 overwrites it in place when stage 1 ends;
//     * wave w owns the row fragments w and nf - 1 - w (nf = M / 16 <= 32) and streams ONLY the 16 columns of the fragment in hand of the
//       W operands (inv(L)^T, then G_r) from global memory straight into MFMA A registers, three k-tiles ahead, and only its live
//       k-tiles: nf + 1 per stage for every wave -- no LDS staging, no barrier inside a k loop.
//       (A first version staged whole W k-tiles through an LDS ring for all waves: 2 tiles = 64 KB in flight per CU
//       against ~1.7 us of latency is 32 GB/s -- 27 us for the 1 MB a workgroup streams.);
//     * sum_m A1^2, alpha_r^T A1 and sum_m T_r^2 are reduced from the accumulators, and mean / var leave the kernel
//       in the [column][R] layout the likelihood reads (conv_gp/layers.py:128-134 semantics, full_cov = False).
// A1 (and stage 1) is recomputed by the R workgroups of a strip: 8 extra MFMA k-tiles each, cheaper than a launch.
#include <algorithm>
#include "layer.h"
#include "tail_dev.h"

namespace {

constexpr int HC_MP = 512;             // largest padded M handled by head_cond_kernel (16 waves x two 16-row fragments)
constexpr int PS_MP = 256;             // ... by prep_solve_kernel (16 waves x one fragment)
constexpr int HC_BK = 16;
constexpr int HC_BN = 16;
constexpr int HC_D = 3;                // W k-tiles in flight per wave beside the one being multiplied

struct HeadCondArgs {
  const double* B; long ldb; int Kc;       // Kzx, k-major [Mp][ldb]
  const double* LinvT;                     // [Mp][Mp]: Wt of stage 1 (lower-triangular W)
  const double* G;                         // [R][Mp][Mp]: Wt of stage 3 (upper-triangular W), nullptr: no q_sqrt term
  const double* alpha; int Rp;             // [Mp][Rp]
  const double* kd;                        // Knn per column: kd_scale * sum_i kd[j * kd_n + i]  (kd_n = 1, kd_scale = 1: plain vector)
  int kd_n; double kd_scale;
  int Mp, R;
  double *out_mean, *out_var;              // [Kc][R]
  double* A1_out; long lda1;               // != null: A1 = inv(L) Kzx [M][lda1] is left behind as well (a training step's reverse pass reads it)
  int M;
  // the KL pieces of the model's layers as workgroups (x < kl.nl, y == R) of this launch (tail_dev.h: kl_pieces_block): parameter-only work that was the
  // long pole of the tail launch -- a lane's M R / 256 dependent (strided load, log) pairs, ~5 us of its 14 at M = 256 -- and costs nothing beside the
  // conditional's ~200 workgroups
  KlTail kl; double* kl_scal;
};

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Row fragments: nf = Mp / 16 <= 32.  Wave w < ceil(nf / 2) owns fragments w and nf - 1 - w (one where they meet): w + 1 and nf - w live k-tiles in
// the lower-triangular stage, nf - w and w + 1 in the upper-triangular one -- nf + 1 per stage for every wave; the other waves only help with the loads
// and the reductions.  (M <= 256 ran one fragment per wave, wave w's k-tiles graded 1 .. 16; M = 384 went through four launches of the GEMM route.)
__global__ __launch_bounds__(1024, 4) void head_cond_kernel(HeadCondArgs a) {
  extern __shared__ __attribute__((aligned(16))) double hc_smem[];
  double* Bt = hc_smem;                       // [Mp][16]: Kzx strip, then A1
  double* red = hc_smem + a.Mp * HC_BN;       // [4][16][16]

  if ((int)blockIdx.y >= a.R) {   // (only with a.kl.nl > 0)
    if ((int)blockIdx.x < a.kl.nl) kl_pieces_block(a.kl.l[blockIdx.x], a.kl_scal + 4 + 4 * blockIdx.x, hc_smem);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, lrow = lane >> 4, lcol = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j0 = blockIdx.x * HC_BN, r = blockIdx.y;
  const int Mp = a.Mp, nt = Mp / HC_BK;                  // k-tiles per stage = row fragments
  const bool s3 = a.G != nullptr;
  const int fr[2] = {wave, nt - 1 - wave};
  const int nfr = fr[0] < fr[1] ? 2 : (fr[0] == fr[1] ? 1 : 0);   // fragments of this wave

  const __amdgpu_buffer_rsrc_t brs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.B), 0, (int)((long)Mp * a.ldb * 8), 0x00020000);
  const __amdgpu_buffer_rsrc_t lrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.LinvT), 0, Mp * Mp * 8, 0x00020000);
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<double*>(s3 ? a.G + (long)r * Mp * Mp : a.LinvT), 0, Mp * Mp * 8, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;

  // Knn of the strip's 16 columns: the sum of each column's kd_n partial sums (the sweep's Kdiag chunks: up to ~100 per column where the
  // launch's tail was cut fine), 64 threads per column, their loads in flight under the strip's DMA.  (Sixteen threads walking kd_n values
  // each at the very end of the kernel were 12 of its 28 us at the M = 32 head.)
  double knn_part = 0.0;
  {
    const int c = tid & 15, sl = tid >> 4, j = j0 + c;
    if (j < a.Kc)
      for (int i = sl; i < a.kd_n; i += 64) knn_part += a.kd[(long)j * a.kd_n + i];
  }
  double al[2][4];                       // alpha[row][r] of this lane's accumulator rows
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) al[c][v] = c < nfr ? a.alpha[(long)(16 * fr[c] + lrow + 4 * v) * a.Rp + r] : 0.0;
  // Kzx strip by LDS-DMA: wave-instruction q covers rows 8q..8q+7 (8 lanes x 16 B per row); Mp / 128 per wave
  {
    const int rl = lane >> 3, cl = (lane & 7) * 2;
    for (int q = wave; q * 8 < Mp; q += 16) {
      const int row = q * 8 + rl;
      const unsigned off = (row < Mp) ? (unsigned)(((long)row * a.ldb + j0 + cl) * 8) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (lds_ptr)(Bt + q * 8 * HC_BN), 16, (int)off, 0, 0, 0);
    }
  }

  // A operand straight from global memory: lane (lrow, lcol) of k-substep q needs W^T[kt*16 + 4q + lrow][16 f + lcol].
  // Each wave streams ONLY the 16 columns of the fragment in hand and only its live k-tiles, HC_D tiles ahead in registers -- no LDS staging
  // and no barrier in the k loop (the B operand is the resident strip).  Per-lane byte offset woff[q]; fragment and k-tile are a scalar offset.
  unsigned woff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) woff[q] = (unsigned)(((4 * q + lrow) * Mp + lcol) * 8);
  auto ldw = [&](const __amdgpu_buffer_rsrc_t& rs, int f, int kt, double (&dst)[4]) {
    const int so = (kt * HC_BK * Mp + 16 * f) * 8;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)woff[q], so, 0);
      __builtin_memcpy(&dst[q], &v, 8);
    }
  };
  // tiles [lo, hi) of one stage of fragment f into acc, prefetch distance HC_D (loads past the end re-read the last tile: static counts)
  auto stage = [&](const __amdgpu_buffer_rsrc_t& rs, int f, int lo, int hi, d4& acc) {
    double wb[HC_D + 1][4];
#pragma unroll
    for (int u = 0; u < HC_D; ++u) ldw(rs, f, min(lo + u, hi - 1), wb[u]);
    for (int kt = lo; kt < hi; kt += HC_D + 1) {
#pragma unroll
      for (int u = 0; u <= HC_D; ++u) {
        if (kt + u < hi) {
          ldw(rs, f, min(kt + u + HC_D, hi - 1), wb[(u + HC_D) % (HC_D + 1)]);
          const double* b = Bt + (kt + u) * HC_BK * HC_BN + lcol;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wb[u][q], b[(4 * q + lrow) * HC_BN], acc, 0, 0, 0);
        }
      }
    }
  };

  __syncthreads();   // Kzx strip resident (drains the DMA queue)
  double s1 = 0.0, mu = 0.0, s2 = 0.0;   // per-lane partials for column lcol
  d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
  // ---- stage 1: A1 = inv(L) Kzx, lower-triangular W: fragment f needs k-tiles 0 .. f ----
#pragma unroll
  for (int c = 0; c < 2; ++c)
    if (c < nfr) stage(lrs, fr[c], 0, fr[c] + 1, acc[c]);
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const double x = c < nfr ? acc[c][v] : 0.0;
      s1 = fma(x, x, s1);
      mu = fma(al[c][v], x, mu);
    }
  __syncthreads();   // every wave is done reading the Kzx strip
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (c < nfr) {
      const int i0 = 16 * fr[c];
#pragma unroll
      for (int v = 0; v < 4; ++v) Bt[(i0 + lrow + 4 * v) * HC_BN + lcol] = acc[c][v];
      if (a.A1_out && r == 0 && j0 + lcol < a.Kc) {   // (every output's workgroup of the strip holds the same A1: the first one stores it)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (i0 + lrow + 4 * v < a.M) a.A1_out[(long)(i0 + lrow + 4 * v) * a.lda1 + j0 + lcol] = acc[c][v];
      }
    }
  }
  __syncthreads();   // A1 published
  // ---- stage 3: T_r = G_r^T A1, upper-triangular W: fragment f needs k-tiles f .. nt-1 ----
  if (s3) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      acc[c] = d4{0.0, 0.0, 0.0, 0.0};
      if (c < nfr) stage(grs, fr[c], fr[c], nt, acc[c]);
#pragma unroll
      for (int v = 0; v < 4; ++v) s2 = fma(acc[c][v], acc[c][v], s2);
    }
  }
  // ---- reduce over the 4 row groups of a wave, then over the waves ----
  s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
  mu += __shfl_xor(mu, 16); mu += __shfl_xor(mu, 32);
  s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
  knn_part += __shfl_xor(knn_part, 16); knn_part += __shfl_xor(knn_part, 32);   // (lane & 15 is the column: the wave's four slices)
  if (lrow == 0) {
    red[(0 * 16 + wave) * 16 + lcol] = s1;
    red[(1 * 16 + wave) * 16 + lcol] = mu;
    red[(2 * 16 + wave) * 16 + lcol] = s2;
    red[(3 * 16 + wave) * 16 + lcol] = knn_part;
  }
  __syncthreads();
  if (tid < HC_BN) {
    double t1 = 0.0, tm = 0.0, t2 = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      t1 += red[(0 * 16 + w) * 16 + tid];
      tm += red[(1 * 16 + w) * 16 + tid];
      t2 += red[(2 * 16 + w) * 16 + tid];
    }
    const int j = j0 + tid;
    if (j < a.Kc) {
      a.out_mean[(long)j * a.R + r] = tm;
      double knn = 0.0;
#pragma unroll
      for (int w = 0; w < 16; ++w) knn += red[(3 * 16 + w) * 16 + tid];
      a.out_var[(long)j * a.R + r] = knn * a.kd_scale - t1 + t2;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// The conditional's small operands for every layer in ONE launch:  G_r = inv(L) Lq_r  (r < R) and alpha = inv(L) q_mu.
// Same structure as stage 1 above -- a 16-column strip of the right-hand side LDS-resident, each wave streaming its
// own 16 columns of inv(L)^T -- with the product stored instead of reduced.  grid (Mp/16 strips, R + 1, layers):
// y < R: strip x of Lq_r (lower triangular: k-tiles above the strip are structurally zero and skipped),
// y == R: q_mu (x == 0 only).  The generic route was two latency-bound GEMM launches per layer (17 + 31 us at cfg2).
struct PrepSolveLayer {
  const double* LinvT; const double* Lq; const double* qmu; double* G; double* alpha; double* klp;
  int Mp, R, Rp, active;   // active == 0: whitened layer (G / alpha alias Lq / q_mu) or larger than PS_MP
};
struct PrepSolveArgs { PrepSolveLayer l[16]; };   // layers, then the prior-factor entries (G == nullptr: sums only)

__global__ __launch_bounds__(1024, 4) void prep_solve_kernel(PrepSolveArgs args) {
  __shared__ __attribute__((aligned(16))) double Bt[PS_MP * HC_BN];   // [k][16]
  __shared__ double ssq[16];
  // Heaviest strips first.  Strip s of a triangular right-hand side costs (16 - s)(17 - s) / 2 k-tiles -- 136 for strip 0, 1 for strip 15 --
  // and the launch is 1.4 rounds of the resident slots: in grid order (strips fastest) the second round still held strip-0 workgroups,
  // 9 us each, behind a first round of the same length.  Workgroups start in the order of their linear id, so the id is re-read as
  // (strip slowest): every strip-0 workgroup of every (r, layer) starts first and the second round is the 1- to 10-tile strips.
  const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nyz = gridDim.y * gridDim.z;
  const int strip = lin / nyz, y = (lin % nyz) % gridDim.y, z = (lin % nyz) / gridDim.y;
  const PrepSolveLayer& a = args.l[z];
  if (!a.active) return;
  __builtin_amdgcn_s_setprio(3);   // part of the latency-bound parameter-only chain (see chol_rl_kernel)
  const int Mp = a.Mp;
  const bool is_alpha = y == a.R;
  if (y > a.R || strip * HC_BN >= (is_alpha ? a.Rp : Mp) || (!is_alpha && !a.Lq)) return;
  const int tid = threadIdx.x, lane = tid & 63, lrow = lane >> 4, lcol = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Row fragment of this wave.  Fragment f costs f + 1 - kt_lo k-tiles and the four waves of a SIMD (w, w + 4, w + 8, w + 12) share its
  // matrix pipe: dealt in order, SIMD 3 carries 4 + 8 + 12 + 16 = 40 tiles against 28 on SIMD 0, and the stage is as long as the
  // busiest SIMD (traced: 9.3 us for strip 0).  Dealt (s, 7 - s, 8 + s, 15 - s) every SIMD carries 34.
  const int rf = (wave >> 2) == 0 ? (wave & 3) : ((wave >> 2) == 1 ? 7 - (wave & 3) : ((wave >> 2) == 2 ? 8 + (wave & 3) : 15 - (wave & 3)));
  const int i0 = rf * 16;
  const bool live = i0 < Mp;
  const double* __restrict__ B = is_alpha ? a.qmu : a.Lq + (long)y * Mp * Mp;
  const int ldb = is_alpha ? a.Rp : Mp;
  double* __restrict__ C = is_alpha ? a.alpha : a.G + (long)y * Mp * Mp;
  const int c0 = strip * HC_BN;
  const int kt_lo = is_alpha ? 0 : c0 / HC_BK;   // Lq lower triangular: rows k < c0 of the strip are zero

  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(B), 0, Mp * ldb * 8, 0x00020000);
  const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.LinvT), 0, Mp * Mp * 8, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  {
    const int rl = lane >> 3, cl = (lane & 7) * 2;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int q = wave * 2 + h, row = q * 8 + rl;
      const unsigned off = (row < Mp) ? (unsigned)((row * ldb + c0 + cl) * 8) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (lds_ptr)(Bt + q * 8 * HC_BN), 16, (int)off, 0, 0, 0);
    }
  }
  unsigned woff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) woff[q] = live ? (unsigned)(((4 * q + lrow) * Mp + i0 + lcol) * 8) : OOB;
  auto ldw = [&](int kt, double (&dst)[4]) {
    const int so = kt * HC_BK * Mp * 8;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(lrs, (int)woff[q], so, 0);
      __builtin_memcpy(&dst[q], &v, 8);
    }
  };
  d4 acc = d4{0.0, 0.0, 0.0, 0.0};
  __syncthreads();   // strip resident
  const int lo = kt_lo, hi = rf + 1;   // inv(L) lower triangular: k <= i
  if (live && lo < hi) {
    double wb[HC_D + 1][4];
#pragma unroll
    for (int u = 0; u < HC_D; ++u) ldw(min(lo + u, hi - 1), wb[u]);
    for (int kt = lo; kt < hi; kt += HC_D + 1) {
#pragma unroll
      for (int u = 0; u <= HC_D; ++u) {
        if (kt + u < hi) {
          ldw(min(kt + u + HC_D, hi - 1), wb[(u + HC_D) % (HC_D + 1)]);
          const double* b = Bt + (kt + u) * HC_BK * HC_BN + lcol;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wb[u][q], b[(4 * q + lrow) * HC_BN], acc, 0, 0, 0);
        }
      }
    }
  }
  if (live && (is_alpha ? a.alpha : a.G)) {
#pragma unroll
    for (int v = 0; v < 4; ++v) C[(long)(i0 + lrow + 4 * v) * ldb + c0 + lcol] = acc[v];
  }
  // sum of squares of the strip just produced, in a fixed order: ||G_r||_F^2 and ||alpha||^2 are the KL's trace and
  // Mahalanobis terms where the KL prior is K itself (the head) -- its two latency-bound GEMMs recomputed exactly these
  if (a.klp) {
    double s = (acc[0] * acc[0] + acc[1] * acc[1]) + (acc[2] * acc[2] + acc[3] * acc[3]);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
    if (lane == 0) ssq[wave] = s;   // waves beyond Mp hold zeros
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < 16; ++w) t += ssq[w];
      a.klp[(long)y * (Mp / HC_BN) + strip] = t;
    }
  }
}

}  // namespace

bool head_cond_fused_ok(const GpMats& g) { return g.Mp <= HC_MP && g.Mp % HC_BK == 0; }
static bool prep_solve_ok(const GpMats& g) { return g.Mp <= PS_MP && g.Mp % HC_BK == 0; }

// mean / var [Kc][R] of the conditional at Kc columns whose Kzx is B [Mp][ldb]; G / alpha from cond_prep
int head_cond_fused(dcgp_ctx* ctx, const GpMats& g, const double* B, long ldb, int Kc, bool have_qsqrt, const double* kd,
                    double* out_mean, double* out_var, int kd_n, double kd_scale, double* A1_out, long lda1) {
  if (Kc <= 0) return DCGP_OK;
  if (!head_cond_fused_ok(g) || (long)g.Mp * ldb * 8 >= (1L << 31))
    return ctx_fail(ctx, DCGP_ERR_ARG, "head_cond_fused: M = %d not supported", g.Mp);
  ScopedTimer t(ctx, "head_cond");
  HeadCondArgs a;
  a.B = B; a.ldb = ldb; a.Kc = Kc;
  a.LinvT = g.LinvT; a.G = have_qsqrt ? g.G : nullptr; a.alpha = g.alpha; a.Rp = g.Rp;
  a.kd = kd; a.kd_n = kd_n; a.kd_scale = kd_scale; a.Mp = g.Mp; a.R = g.R;
  a.out_mean = out_mean; a.out_var = out_var;
  a.A1_out = A1_out; a.lda1 = lda1; a.M = g.M;
  a.kl.nl = 0; a.kl_scal = nullptr;
  const int strips = (Kc + HC_BN - 1) / HC_BN;
  // (worth it from ~1000 diagonal entries per layer on: M = 256, R = 10 -- head-only model 0.2186 -> 0.2160 ms per step; at M = 32 the extra row of
  // workgroups costs the launch more than the tail saves: 0.1408 -> 0.1418)
  long kl_work = 0;
  if (ctx->kl_ride)
    for (int l = 0; l < ctx->kl_ride->nl; ++l) kl_work = std::max<long>(kl_work, (long)ctx->kl_ride->l[l].M * ctx->kl_ride->l[l].R);
  if (ctx->kl_ride && ctx->kl_ride->nl <= strips && kl_work >= 1024 && !ctx->opt.kl_no_ride) {
    a.kl = *ctx->kl_ride; a.kl_scal = ctx->kl_ride_scal;
    ctx->kl_ride = nullptr; ctx->kl_rode = true;
  }
  const size_t lds = (size_t)(g.Mp * HC_BN + 4 * 16 * 16) * sizeof(double);   // (>= the 8 KB of a KL workgroup: Mp >= 16)
  static bool attr_set[64] = {};   // per device (72 KB at Mp = 512)
  const int dv = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
  if (!attr_set[dv]) {
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)head_cond_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (HC_MP * HC_BN + 4 * 16 * 16) * 8));
    attr_set[dv] = true;
  }
  hipLaunchKernelGGL(head_cond_kernel, dim3(strips, g.R + (a.kl.nl > 0 ? 1 : 0)), dim3(1024), lds, ctx->stream, a);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

// G / alpha of up to 8 layers in one launch (replaces cond_prep for unwhitened layers with M <= 256)
int prep_solve_all(dcgp_ctx* ctx, GpMats* const* gs, const int* white, const bool* have_qsqrt, int nl, bool* done, const bool* skip) {
  PrepSolveArgs a;
  int any = 0, maxMp = 0, maxR = 0, ne = nl < 8 ? nl : 8;
  for (int i = 0; i < nl && i < 8; ++i) {
    const GpMats& g = *gs[i];
    PrepSolveLayer& l = a.l[i];
    if (skip && skip[i]) { l.active = 0; continue; }   // G / alpha of this layer rode the factorisation chain (its flags are the caller's)
    l.active = (!white[i] && prep_solve_ok(g) && g.Rp == HC_BN) ? 1 : 0;
    done[i] = l.active != 0;
    l.LinvT = g.LinvT; l.Lq = have_qsqrt[i] ? g.Lq : nullptr; l.qmu = g.qmu; l.G = g.G; l.alpha = g.alpha;
    l.klp = (l.active && have_qsqrt[i]) ? g.klp : nullptr;
    gs[i]->klp_valid = l.klp != nullptr;
    gs[i]->klpp_valid = false;
    gs[i]->kl_ns = gs[i]->kl_nsa = 0;
    l.Mp = g.Mp; l.R = g.R; l.Rp = g.Rp;
    if (l.active) { any = 1; maxMp = g.Mp > maxMp ? g.Mp : maxMp; maxR = g.R > maxR ? g.R : maxR; }
  }
  // layers whose KL prior is Kuu(Z0): the same products with inv(Lp), kept as sums of squares only (the KL's trace and
  // Mahalanobis terms; two latency-bound GEMM launches per layer otherwise, beside the layer kernel)
  for (int i = 0; i < nl && i < 8; ++i) {
    const GpMats& g = *gs[i];
    if (!a.l[i].active || !have_qsqrt[i] || !g.Kp || !g.klpp || !g.LpinvT) continue;
    PrepSolveLayer& l = a.l[ne++];
    l = a.l[i];
    l.LinvT = g.LpinvT; l.G = nullptr; l.alpha = nullptr; l.klp = g.klpp;
    gs[i]->klpp_valid = true;
  }
  if (!any) return DCGP_OK;
  ScopedTimer t(ctx, "prep_solve");
  hipLaunchKernelGGL(prep_solve_kernel, dim3(maxMp / HC_BN, maxR + 1, ne), dim3(1024), 0, ctx->stream, a);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}
