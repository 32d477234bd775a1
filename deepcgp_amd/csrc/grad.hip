// grad.hip -- reverse pass of the ELBO (the reference gets it from TensorFlow autodiff of the graph it
// builds at conv_gp/experiment.py:84-108).  Runs over the state the forward pass (model.hip) left on the device:
// every layer's K_uf and A1 = inv(L) K_uf, the factors and their inverses, G_r / alpha, and the per-layer
// (sample, mean, var).  Per layer, given d ELBO / d(mean, var) of its outputs:
//
//   conditional   mean = alpha^T A1,  var = Knn - sum_m A1^2 + sum_m (G_r^T A1)^2   (the forward's re-association)
//     dT_r = 2 (G_r^T A1) o dvar_r        d alpha = A1 dmean        dG_r = tril(A1 dT_r^T) = tril(2 A1 diag(dvar_r) A1^T G_r)
//     dA1  = alpha dmean^T - 2 A1 o sum_r dvar_r + sum_r G_r dT_r
//     unwhitened: dq_mu = inv(L)^T d alpha, dq_sqrt_r = tril(inv(L)^T dG_r), dL -= tril(dq_mu alpha^T + sum_r (inv(L)^T dG_r) G_r^T)
//     dKuf = inv(L)^T dA1,  dL -= tril(dKuf A1^T)
//   Cholesky      S = inv(L)^T Phi(L^T dL) inv(L)   (Phi: lower triangle, diagonal halved; Murray 2016), used unsymmetrised
//   kernels       E = dK o K:  dvariance = sum E / variance, dlengthscale = sum E d^2 / l^3,
//                 dZ = (E X - rowsum(E) o Z) / l^2,  dX = (E^T Z - colsum(E) o X) / l^2  on im2col'd patches, col2im gather
//   KL            closed form in inv(K) (no second Cholesky adjoint), sample: dmean += dF, dvar += dF (F - mean) / (2 (var + jitter))
//
// The R M^2 K products that fit the forward's tuned kernel run on it (gemm.hip: dT with a column-scaled store, dA1 as one
// launch with the R blocks stacked along k, dK_uf); everything else goes through gemm_gen (strided MFMA GEMM, deterministic
// split-k, the adjoints' corrections in its epilogue) and elementwise / reduction kernels.
// Scheduling (Lanes, below): the main stream carries the data path only -- the column-wise adjoint of each conditional, the patch-kernel
// adjoint, dX for the layer below; the conditional's M x M chain (W_r -> dG_r -> dq_sqrt -> dL -> Cholesky adjoint -> S) runs on a second
// stream, what needs S (Gram adjoint of K_uu, scalar sums) and the KL pieces on a third; zero fills, parameter-only operands and the KL
// adjoint's products are enqueued beside the FORWARD pass (grad_kl_early).  Every buffer two streams touch is ordered by an event; the main
// stream joins the others once, at the end of the step.  No atomics anywhere: the gradients are reproducible run to run
// (tests: test_gradients_repeat_in_steady_state).  The oracle for this file is oracle/grad.py.
#include <algorithm>
#include <chrono>
#include <initializer_list>

#include "model_state.h"
#include "gemm_gen.h"

namespace {

constexpr int VAR_SLOT = 0, LS_SLOT = 16, P2_SLOT = 32;   // gslots: [0,16) variance, [16,32) lengthscale | acos weight variance, [32,48) acos bias variance

inline unsigned blocks_for(long n, int per = 256) { return (unsigned)((n + per - 1) / per); }

__device__ __forceinline__ double block_sum_256(double v, double* red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// ---- likelihood ------------------------------------------------------------------------------------------------
// d sum_rows weight * ve(row) / d(mu, var).  Same quadrature as varexp_kernel (cond.hip): 20 Gauss-Hermite nodes,
// cdf = (1 - 2e-4) Phi + 1e-4, clips at 1e-10.  One thread per (row, node) evaluates the node's terms for every class
// into LDS; one thread per (row, class) then adds the 20 nodes in a fixed order.  RM_ROWS rows per 256-thread block.
constexpr int RM_ROWS = 12, RM_KMAX = 16;
__global__ __launch_bounds__(256) void robustmax_grad_kernel(const double* __restrict__ mu, const double* __restrict__ var, const int32_t* __restrict__ y,
                                                             int rows, int n_labels, int K, double eps, const double* __restrict__ gh, double weight,
                                                             double* __restrict__ gm, double* __restrict__ gv) {
  __shared__ double tm[RM_ROWS][20][RM_KMAX], tv[RM_ROWS][20][RM_KMAX];
  const int t = threadIdx.x, lr = t / 20, g = t % 20;
  const int row = blockIdx.x * RM_ROWS + lr;
  const double inv_sqrt_pi = 0.56418958354775628695, inv_sqrt_2pi = 0.39894228040143267794;
  if (lr < RM_ROWS && row < rows) {
    const int lab = y[row % n_labels];
    const double* m = mu + (long)row * K;
    const double* v = var + (long)row * K;
    const double vy = v[lab];
    const bool live_y = 2.0 * vy > 1e-10;
    const double sy = sqrt(fmax(2.0 * vy, 1e-10));
    const double xg = gh[g], wg = gh[20 + g] * inv_sqrt_pi;
    const double X = m[lab] + xg * sy;
    double prod = 1.0;
    for (int k = 0; k < K; ++k) {
      if (k == lab) continue;
      const double d = (X - m[k]) / sqrt(fmax(v[k], 1e-10));
      prod *= 0.5 * (1.0 + erf(d * 0.70710678118654752440)) * (1.0 - 2e-4) + 1e-4;
    }
    double qs = 0.0;
    for (int k = 0; k < K; ++k) {
      if (k == lab) continue;
      const double sig = sqrt(fmax(v[k], 1e-10));
      const double d = (X - m[k]) / sig;
      const double cdf = 0.5 * (1.0 + erf(d * 0.70710678118654752440)) * (1.0 - 2e-4) + 1e-4;
      const double q = wg * prod / cdf * exp(-0.5 * d * d) * inv_sqrt_2pi * (1.0 - 2e-4);
      tm[lr][g][k] = -q / sig;
      tv[lr][g][k] = v[k] > 1e-10 ? -q * d / (2.0 * v[k]) : 0.0;
      qs += q / sig;
    }
    tm[lr][g][lab] = qs;
    tv[lr][g][lab] = live_y ? qs * xg / sy : 0.0;
  }
  __syncthreads();
  const double c = log(1.0 - eps) - log(eps / (K - 1.0));
  for (int idx = t; idx < RM_ROWS * K; idx += 256) {
    const int r2 = idx / K, k = idx % K, row2 = blockIdx.x * RM_ROWS + r2;
    if (row2 >= rows) continue;
    double a = 0.0, b = 0.0;
    for (int gg = 0; gg < 20; ++gg) { a += tm[r2][gg][k]; b += tv[r2][gg][k]; }
    gm[(long)row2 * K + k] = c * weight * a;
    gv[(long)row2 * K + k] = c * weight * b;
  }
}

// GT[(r * Mp + k) * Mp + i] = G[r][i][k]: the R transposes stacked along k (pads included)
__global__ void restack_transpose_kernel(const double* __restrict__ G, int Mp, int R, double* __restrict__ GT) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long mm = (long)Mp * Mp;
  if (idx >= mm * R) return;
  const int r = (int)(idx / mm);
  const int k = (int)((idx % mm) / Mp), i = (int)(idx % Mp);
  GT[idx] = G[r * mm + (long)i * Mp + k];
}

// previous layer's (dmean, dvar) from d sample:  sample = mean + z sqrt(var + jitter),  z = (sample - mean) / sqrt(var + jitter)
__global__ void sample_backward_kernel(const double* __restrict__ dF, const double* __restrict__ sample, const double* __restrict__ mean,
                                       const double* __restrict__ var, double jitter, long n, double* __restrict__ gm,
                                       double* __restrict__ gv) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = dF[i];
  gm[i] = d;
  gv[i] = d * (sample[i] - mean[i]) / (2.0 * (var[i] + jitter));
}

// the same for a de-duplicated first layer, whose S samples share one conditional: the S gradients arriving per element add up
// (gm[i] = sum_s of the kernel above at s * n + i, in the order s = 0, 1, ...)
__global__ void sample_backward_dedup_kernel(const double* __restrict__ dF, const double* __restrict__ sample, const double* __restrict__ mean,
                                             const double* __restrict__ var, double jitter, int S, long n, double* __restrict__ gm,
                                             double* __restrict__ gv) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double sa = 0.0, sb = 0.0;
  for (int s = 0; s < S; ++s) {
    const long k = (long)s * n + i;
    const double d = dF[k];
    sa += d;
    sb += d * (sample[k] - mean[k]) / (2.0 * (var[k] + jitter));
  }
  gm[i] = sa;
  gv[i] = sb;
}

// out[i] = sum_s in[s * n + i]: the S samples of a de-duplicated first layer share one conditional
__global__ void reduce_replicas_kernel(const double* __restrict__ a, const double* __restrict__ b, int S, long n, double* __restrict__ oa,
                                       double* __restrict__ ob) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double sa = 0.0, sb = 0.0;
  for (int s = 0; s < S; ++s) { sa += a[(long)s * n + i]; sb += b[(long)s * n + i]; }
  oa[i] = sa;
  ob[i] = sb;
}

// ---- conditional -----------------------------------------------------------------------------------------------
__global__ void rowsum_small_kernel(const double* __restrict__ a, long n, int R, double* __restrict__ out) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  double s = 0.0;
  for (int r = 0; r < R; ++r) s += a[c * R + r];
  out[c] = s;
}

// dA1[m][c] -= 2 A1[m][c] gvs[c]
__global__ void dA1_fix_kernel(double* __restrict__ dA1, const double* __restrict__ A1, const double* __restrict__ gvs, int M, long Kc,
                               long ld) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int m = blockIdx.y;
  if (c >= Kc || m >= M) return;
  dA1[m * ld + c] -= 2.0 * A1[m * ld + c] * gvs[c];
}

// dst_b[i][j] (+)= alpha * (j <= i ? src_b[i][j] : 0)
__global__ void mask_lower_kernel(const double* __restrict__ src, long lds, long sbs, double* __restrict__ dst, long ldd, long dbs, int M,
                                  double alpha, int accumulate) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
  if (j >= M) return;
  const double v = j <= i ? alpha * src[b * sbs + i * lds + j] : 0.0;
  double* d = dst + b * dbs + i * ldd + j;
  *d = accumulate ? *d + v : v;
}

// dst[i][j] (+)= alpha * src[i][j]   (rectangular copy between leading dimensions)
__global__ void copy2d_kernel(const double* __restrict__ src, long lds, double* __restrict__ dst, long ldd, int rows, int cols, double alpha,
                              int accumulate) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= cols || i >= rows) return;
  const double v = alpha * src[i * lds + j];
  double* d = dst + i * ldd + j;
  *d = accumulate ? *d + v : v;
}

// gq_sqrt[r][i][i] += 1 / Lq[r][i][i]   (- d/dLq of -1/2 log det(Lq Lq^T))
__global__ void kl_diag_kernel(double* __restrict__ gq, const double* __restrict__ Lq, int M, int Mp, int R, double kw) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * R) return;
  const int r = idx / M, i = idx % M;
  gq[((long)r * M + i) * M + i] += kw / Lq[((long)r * Mp + i) * Mp + i];
}

// ---- RBF Gram backward ------------------------------------------------------------------------------------------
// One block per row i.  S = d ELBO / dK (not symmetrised).  Writes Es[i][j] = (S_ij + S_ji) k_ij (or nothing if Es is
// null), rs[i] = sum_j Es[i][j], and the per-row partial sums pv[i] = sum_j S_ij k_ij / variance, pl[i] = sum_j S_ij k_ij d_ij^2 / l^3.
__global__ __launch_bounds__(256) void kuu_backward_kernel(const double* __restrict__ Z, const double* __restrict__ ZT, int ldzt, int M, int L,
                                                           const double* __restrict__ S, long lds,
                                                           double variance, double inv_l2, double inv_l3, double* __restrict__ Es, long lde,
                                                           double* __restrict__ rs, double* __restrict__ pv, double* __restrict__ pl) {
  __shared__ double red[256];
  const int i = blockIdx.x, t = threadIdx.x;
  double srow = 0.0, sv = 0.0, sl = 0.0;
  for (int j = t; j < M; j += 256) {
    double d2 = 0.0;
    if (ZT) {   // k-major copy of Z (prepare_all): neighbouring threads read neighbouring j
      for (int l = 0; l < L; ++l) {
        const double d = Z[(long)i * L + l] - ZT[(long)l * ldzt + j];
        d2 += d * d;
      }
    } else {
      for (int l = 0; l < L; ++l) {
        const double d = Z[(long)i * L + l] - Z[(long)j * L + l];
        d2 += d * d;
      }
    }
    const double k = variance * exp(-0.5 * d2 * inv_l2);
    const double e = S[i * lds + j] * k;
    sv += e;
    sl += e * d2;
    if (Es) {
      const double es = e + S[j * lds + i] * k;
      Es[i * lde + j] = es;
      srow += es;
    }
  }
  const double a = block_sum_256(srow, red), b = block_sum_256(sv, red), c = block_sum_256(sl, red);
  if (t == 0) {
    if (rs) rs[i] = a;
    pv[i] = b / variance;
    pl[i] = c * inv_l3;
  }
}

// ---- patch kernels backward -------------------------------------------------------------------------------------
// Xcol[c][l], c = n * P + p, l = (kh * f + kw) * C + ch  (FullView.extract_patches, conv_gp/views.py:46-54)
// A patch row (kh fixed) is ONE contiguous run of f C doubles of the image, and so is its place in Xcol: block (n, oh) copies the
// Wo f runs of its patch row -- thread (run slot, j): no division inside the loop.  (One thread per element with 64-bit index
// arithmetic: 26 us for the head's 41 MB at the headline size; one 250-element patch per block: 28.)
__global__ __launch_bounds__(256) void im2col_kernel(const double* __restrict__ X, int n_mod, int H, int W, int C, int f, int s, int Ho, int Wo,
                                                     int L, double* __restrict__ Xcol) {
  const int fc = f * C, n = blockIdx.x / Ho, oh = blockIdx.x - n * Ho;
  const int per = fc >= 256 ? 1 : 256 / fc;               // runs in flight per pass
  const int slot = fc >= 256 ? 0 : threadIdx.x / fc, j0 = fc >= 256 ? threadIdx.x : threadIdx.x - slot * fc;
  if (slot >= per) return;
  const double* __restrict__ img = X + (long)(n % n_mod) * H * W * C;
  double* __restrict__ out = Xcol + ((long)n * Ho * Wo + (long)oh * Wo) * L;
  const int runs = Wo * f;                                 // run r = ow * f + kh
  for (int r = slot; r < runs; r += per) {
    const int ow = r / f, kh = r - ow * f;
    const double* __restrict__ src = img + ((long)(oh * s + kh) * W + ow * s) * C;
    double* __restrict__ dst = out + (long)ow * L + kh * fc;
    for (int j = j0; j < fc; j += 256) dst[j] = src[j];
  }
}

// dX[n][h][w][ch] = sum over the patches that contain the pixel (adjoint of extract_patches; gather, no atomics)
__global__ void col2im_kernel(const double* __restrict__ dXcol, int N, int H, int W, int C, int f, int s, int Ho, int Wo, int L,
                              double* __restrict__ dX) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * H * W * C) return;
  const int ch = (int)(idx % C);
  const int w = (int)((idx / C) % W), h = (int)((idx / ((long)C * W)) % H);
  const int n = (int)(idx / ((long)C * W * H));
  const int P = Ho * Wo;
  double acc = 0.0;
  for (int kh = 0; kh < f; ++kh) {
    const int hh = h - kh;
    if (hh < 0 || hh % s) continue;
    const int oh = hh / s;
    if (oh >= Ho) continue;
    for (int kw = 0; kw < f; ++kw) {
      const int ww = w - kw;
      if (ww < 0 || ww % s) continue;
      const int ow = ww / s;
      if (ow >= Wo) continue;
      acc += dXcol[((long)n * P + oh * Wo + ow) * L + (kh * f + kw) * C + ch];
    }
  }
  dX[idx] = acc;
}

// Conv2dMean adjoint (conv_gp/mean_functions.py:28-41): map 0 of patch p copied the centre pixel of channel 0, so
// dX[n][oh s + f/2][ow s + f/2][0] += gm[(n P + p) R + 0]; distinct patches have distinct centres (no conflicts)
__global__ void idmean_backward_kernel(const double* __restrict__ gm, long Kc, int R, int P, int Wo, int H, int W, int C, int f, int s,
                                       double* __restrict__ dX) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Kc) return;
  const long n = c / P;
  const int p = (int)(c % P), oh = p / Wo, ow = p % Wo;
  dX[((n * H + oh * s + f / 2) * W + ow * s + f / 2) * C] += gm[c * R];
}

// One thread per column c and row chunk (blockIdx.y), loop over the chunk's rows:
//   E[m][c] = dK[m][c / pdiv] * (w ? w[c % pdiv] * wscale : 1) * K[m][c]   (written over dK when pdiv == 1, else into E),
//   csp[chunk][c] = sum_m E, rawp[chunk][c] = sum_m dK K (head: for d patch_weights), per-block partials
//   pv = sum E, pl = sum E d^2 with d^2 = -2 l^2 log(K / variance).
__global__ __launch_bounds__(256) void e_form_kernel(const double* __restrict__ dK, long lddk, int pdiv, const double* __restrict__ w,
                                                     double wscale, const double* __restrict__ K, long ldk, double* __restrict__ E, long lde,
                                                     int M, int rows_per_chunk, long Kc, double inv_var, double two_l2,
                                                     double* __restrict__ csp, double* __restrict__ rawp, double* __restrict__ pv,
                                                     double* __restrict__ pl) {
  __shared__ double red[256];
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
  double sc = 0.0, sr = 0.0, sd = 0.0;
  if (c < Kc) {
    const long cd = c / pdiv;
    const double f = w ? w[c % pdiv] * wscale : 1.0;
    for (int m = m0; m < m1; ++m) {
      const double k = K[m * ldk + c];
      const double r = dK[m * lddk + cd] * k;
      const double e = r * f;
      E[m * lde + c] = e;
      sr += r;
      sc += e;
      // d^2 = -2 l^2 log(k / variance).  A response so small that k / variance is a denormal or rounds to zero (the unit sweep's 2^t keeps
      // denormals where the launch-per-tile sweep flushed them: log -> -inf against e -> 0 would make the lengthscale gradient NaN) weighs
      // less than 1e-300: skipped
      const double q = k * inv_var;
      if (q >= 2.2250738585072014e-308) sd -= e * two_l2 * log(q);
    }
    csp[(long)blockIdx.y * Kc + c] = sc;
    if (rawp) rawp[(long)blockIdx.y * Kc + c] = sr;
  }
  const double a = block_sum_256(sc, red), b = block_sum_256(sd, red);
  if (threadIdx.x == 0) {
    pv[(long)blockIdx.y * gridDim.x + blockIdx.x] = a;
    pl[(long)blockIdx.y * gridDim.x + blockIdx.x] = b;
  }
}

// the same for two arrays at once (e_form: column sums of E and of dK o K)
__global__ void sum_chunks2_kernel(const double* __restrict__ pa, const double* __restrict__ pb, int chunks, long n, double* __restrict__ oa,
                                   double* __restrict__ ob) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = 0.0, b = 0.0;
  for (int ch = 0; ch < chunks; ++ch) { a += pa[(long)ch * n + i]; b += pb[(long)ch * n + i]; }
  oa[i] = a;
  ob[i] = b;
}
// out[i] = sum_ch part[ch][i]
__global__ void sum_chunks_kernel(const double* __restrict__ part, int chunks, long n, double* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int ch = 0; ch < chunks; ++ch) s += part[(long)ch * n + i];
  out[i] = s;
}

// part[chunk][m] = sum over the chunk's columns of A[m][c]: block (m, chunk); sum_chunks_kernel finishes
__global__ __launch_bounds__(256) void rowsum_big_kernel(const double* __restrict__ A, long ld, long Kc, long cols_per_chunk, int M,
                                                         double* __restrict__ part) {
  __shared__ double red[256];
  const int m = blockIdx.x;
  const long c0 = (long)blockIdx.y * cols_per_chunk, c1 = min(Kc, c0 + cols_per_chunk);
  double s = 0.0;
  for (long c = c0 + threadIdx.x; c < c1; c += 256) s += A[m * ld + c];
  const double r = block_sum_256(s, red);
  if (threadIdx.x == 0) part[(long)blockIdx.y * M + m] = r;
}

// out[p] (+)= scale * sum_n raw[n * P + p]: one block per p
__global__ __launch_bounds__(256) void strided_sum_kernel(const double* __restrict__ raw, int N, int P, double scale, int accumulate,
                                                          double* __restrict__ out) {
  __shared__ double red[256];
  const int p = blockIdx.x;
  double s = 0.0;
  for (int n = threadIdx.x; n < N; n += 256) s += raw[(long)n * P + p];
  const double r = block_sum_256(s, red);
  if (threadIdx.x == 0) out[p] = accumulate ? out[p] + scale * r : scale * r;
}

// ConvKernel.Kdiag backward (conv_gp/kernels.py:106-115): block per (image n, patch p), thread per p'.
// Gm[n][p][p'] = x_p . x_p' in, E[n][p][p'] = g_n w_p w_p' k_pp' / P^2 out (over Gm; the squared norms come from the
// copy `norms`, because other blocks overwrite the diagonal they would otherwise be read from).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// one wave per (image n, patch p): 4 rows per 256-thread block, lanes stride over p'
__global__ __launch_bounds__(256) void kdiag_backward_kernel(double* __restrict__ Gm, const double* __restrict__ norms, const double* __restrict__ gkd,
                                                             const double* __restrict__ w, int P, double variance, double inv_l2,
                                                             double* __restrict__ dwn, double* __restrict__ pv, double* __restrict__ pl) {
  const int n = blockIdx.y, p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= P) return;
  double* G = Gm + ((long)n * P + p) * P;
  const double* nr = norms + (long)n * P;
  const double gpp = nr[p], wp = w[p];
  const double coef = gkd[n] / ((double)P * P);
  double se = 0.0, sd = 0.0, sw = 0.0;
  for (int q = lane; q < P; q += 64) {
    const double d2 = gpp + nr[q] - 2.0 * G[q];
    const double k = variance * exp(-0.5 * d2 * inv_l2);
    const double e = coef * wp * w[q] * k;
    sw += k * w[q];
    se += e;
    sd += e * d2;
    G[q] = e;
  }
  se = wave_sum(se); sd = wave_sum(sd); sw = wave_sum(sw);
  if (lane == 0) {
    const long o = (long)n * P + p;
    pv[o] = se;          // sum E over the row (also the row sum the patch gradient needs)
    pl[o] = sd;
    dwn[o] = 2.0 * coef * sw;
  }
}
// AdditivePatchKernel.Kdiag = variance * mean(w):  t[0] = sum_n g_n, t[1] = mean(w)
__global__ void additive_kdiag_backward_kernel(const double* __restrict__ t, int P, double variance, double* __restrict__ gw,
                                               double* __restrict__ var_slot) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P) gw[p] += t[0] * variance / P;
  if (p == 0) *var_slot = t[0] * t[1];
}
__global__ void kdiag_norms_kernel(const double* __restrict__ Gm, int P, long n_img, double* __restrict__ norms) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_img * P) return;
  const long n = idx / P;
  const int p = (int)(idx % P);
  norms[idx] = Gm[(n * P + p) * P + p];
}

// the layer's three scalar gradients from their 16 slots each: lanes 16 q .. 16 q + 15 hold the slots of sum q (one load per lane -- one thread
// walking the 48 slots took ~50 us of a step's tail stream: a load, a wait and an add 48 times over)
__global__ void scal_finish_kernel(const double* __restrict__ slots, double* __restrict__ gscal) {
  if (blockIdx.x) return;
  const int lane = threadIdx.x, q = lane >> 4, i = lane & 15;
  double v = q < 3 ? slots[(q == 0 ? VAR_SLOT : (q == 1 ? LS_SLOT : P2_SLOT)) + i] : 0.0;
  for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
  if (q < 3 && i == 0) gscal[q] = v;
}

// ---- ArcCosine(order 0) base kernel (--base-kernel acos, conv layers) ------------------------------------------------
// K = variance (pi - theta) / pi, theta = acos(c'), c' = 1e-15 + (1 - 2e-15) c, c = (w x.z + b) / sqrt(Q A), Q = w |z|^2 + b,
// A = w |x|^2 + b.  With F = dLoss/dc = dK variance (1 - 2e-15) / (pi sin theta), F1 = F / sqrt(Q A), F2 = F c:
//   dZ = w (F1 X - (rowsum(F2) / Q) o Z),  dX = w (F1^T Z - (colsum(F2) / A) o X)     -- the RBF adjoint's shape with
//   E -> F1, rowsum(E) -> rowsum(F2) / Q, colsum(E) -> colsum(F2) / A, 1 / l^2 -> w, so patch_backward / kuu GEMMs are shared;
//   dw = sum F1 s - F2 (a / A + q / Q) / 2,  db = sum F1 - F2 (1 / A + 1 / Q) / 2   (s = x.z, a = |x|^2, q = |z|^2).
// theta and c are recovered from the stored K (theta = pi (1 - K / variance)).  Coincident points (sin theta ~ 0) carry no
// gradient except through the variance: c == 1 there identically in z, w and b (oracle/grad.py _acos_backward).
__global__ void rownorm_kernel(const double* __restrict__ X, long rows, int L, double* __restrict__ out) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= rows) return;
  double s = 0.0;
  for (int l = 0; l < L; ++l) { const double v = X[c * L + l]; s += v * v; }
  out[c] = s;
}
// one thread per column c and row chunk (blockIdx.y): F1 (beside dK: the chain stream still reads dK), F2, csp[chunk][c] = sum_m F2 / A_c, block partials
__global__ __launch_bounds__(256) void acos_e_form_kernel(const double* __restrict__ dK, double* __restrict__ F1, long ld, const double* __restrict__ K, double* __restrict__ F2,
                                                          int M, int rows_per_chunk, long Kc, double variance, double w, double b,
                                                          const double* __restrict__ zn, const double* __restrict__ xn, double* __restrict__ csp,
                                                          double* __restrict__ pv, double* __restrict__ pw, double* __restrict__ pb) {
  __shared__ double red[256];
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
  double sv = 0.0, sw = 0.0, sb = 0.0, sc = 0.0;
  if (c < Kc) {
    const double a2 = xn[c], A = w * a2 + b;
    for (int m = m0; m < m1; ++m) {
      const double k = K[m * ld + c], dk = dK[m * ld + c];
      const double theta = 3.14159265358979323846 * (1.0 - k / variance);
      const double sn = sin(theta), cc = (cos(theta) - 1e-15) / (1.0 - 2e-15);
      const double F = sn > 1e-12 ? dk * variance * 0.31830988618379067154 * (1.0 - 2e-15) / sn : 0.0;
      const double q2 = zn[m], Q = w * q2 + b, rt = sqrt(Q * A);
      const double f1 = F / rt, f2 = F * cc;
      F1[m * ld + c] = f1;
      F2[m * ld + c] = f2;
      sv += dk * k;
      sw += f1 * (cc * rt - b) / w - 0.5 * f2 * (a2 / A + q2 / Q);
      sb += f1 - 0.5 * f2 * (1.0 / A + 1.0 / Q);
      sc += f2;
    }
    csp[(long)blockIdx.y * Kc + c] = sc / A;
  }
  const double a = block_sum_256(sv, red), bb = block_sum_256(sw, red), cc2 = block_sum_256(sb, red);
  if (threadIdx.x == 0) {
    const long o = (long)blockIdx.y * gridDim.x + blockIdx.x;
    pv[o] = a; pw[o] = bb; pb[o] = cc2;
  }
}
// v[i] /= (w * n[i] + b)
__global__ void acos_divide_kernel(double* __restrict__ v, const double* __restrict__ n, int M, double w, double b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) v[i] /= (w * n[i] + b);
}
// Gram adjoint, one block per row i (both arguments are Z; the diagonal is skipped): Es[i][j] = (F_ij + F_ji) / sqrt(Q_i Q_j),
// rs[i] = sum_j (F_ij + F_ji) c_ij / Q_i, partials pv (variance), pw, pb
__global__ __launch_bounds__(256) void acos_kuu_backward_kernel(const double* __restrict__ Z, const double* __restrict__ zn, int M, int L,
                                                                const double* __restrict__ S, long lds, double variance, double w, double b,
                                                                double* __restrict__ Es, long lde, double* __restrict__ rs,
                                                                double* __restrict__ pv, double* __restrict__ pw, double* __restrict__ pb) {
  __shared__ double red[256];
  const int i = blockIdx.x, t = threadIdx.x;
  const double qi = zn[i], Qi = w * qi + b;
  double srow = 0.0, sv = 0.0, sw = 0.0, sb = 0.0;
  for (int j = t; j < M; j += 256) {
    double s = 0.0;
    for (int l = 0; l < L; ++l) s += Z[(long)i * L + l] * Z[(long)j * L + l];
    const double qj = zn[j], Qj = w * qj + b, rt = sqrt(Qi * Qj);
    const double c = (w * s + b) / rt;
    const double cp = fmin(1e-15 + (1.0 - 2e-15) * c, 1.0);
    const double theta = acos(cp);
    const double k = variance * (1.0 - theta * 0.31830988618379067154);
    const double Sij = S[i * lds + j];
    sv += Sij * k;
    double es = 0.0;
    if (j != i) {
      const double sn = sin(theta);
      const double g = sn > 1e-12 ? variance * 0.31830988618379067154 * (1.0 - 2e-15) / sn : 0.0;
      const double Fij = Sij * g, Fji = S[j * lds + i] * g;
      const double f1 = Fij / rt;
      sw += f1 * s - 0.5 * Fij * c * (qi / Qi + qj / Qj);
      sb += f1 - 0.5 * Fij * c * (1.0 / Qi + 1.0 / Qj);
      es = (Fij + Fji) / rt;
      srow += (Fij + Fji) * c;
    }
    if (Es) Es[i * lde + j] = es;
  }
  const double a = block_sum_256(srow, red), bv = block_sum_256(sv, red), cw = block_sum_256(sw, red), db = block_sum_256(sb, red);
  if (t == 0) {
    if (rs) rs[i] = a / Qi;
    pv[i] = bv / variance;
    pw[i] = cw;
    pb[i] = db;
  }
}

// ---- dense RBF(ARD) head (--last-kernel rbf) --------------------------------------------------------------------------
// out[i][d] = in[(i % n_mod)][d] * s[d]
__global__ void scale_rows_kernel(const double* __restrict__ in, int n_mod, long rows, int D, const double* __restrict__ s, double* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * D) return;
  const long i = idx / D;
  const int d = (int)(idx % D);
  out[idx] = in[(i % n_mod) * D + d] * s[d];
}
// chain rule back through x / l, z / l: gZ = dZs o s, dX = dXs o s (if wanted), and one block per input dimension d:
// gard[d] = -s_d (sum_m dZs[m][d] Zs[m][d] + sum_n dXs[n][d] Xs[n][d])
__global__ __launch_bounds__(256) void ard_finish_kernel(const double* __restrict__ dZs, const double* __restrict__ Zs, int M, const double* __restrict__ dXs,
                                                         const double* __restrict__ Xs, long N, int D, const double* __restrict__ s,
                                                         double* __restrict__ gZ, double* __restrict__ dX, double* __restrict__ gard) {
  __shared__ double red[256];
  const int d = blockIdx.x, t = threadIdx.x;
  const double sd = s[d];
  double acc = 0.0;
  for (int m = t; m < M; m += 256) {
    const double g = dZs[(long)m * D + d];
    acc += g * Zs[(long)m * D + d];
    gZ[(long)m * D + d] = g * sd;
  }
  for (long n = t; n < N; n += 256) {
    const double g = dXs[n * D + d];
    acc += g * Xs[n * D + d];
    if (dX) dX[n * D + d] = g * sd;
  }
  const double r = block_sum_256(acc, red);
  if (t == 0) gard[d] = -sd * r;
}
// ---- optimiser ---------------------------------------------------------------------------------------------------
// tf.train.AdamOptimizer / GradientDescentOptimizer on gpflow's unconstrained variables, ascending the ELBO.  transform 0: identity;
// 1: gpflow transforms.positive (x = softplus(u) + 1e-6): the parameter is held constrained, moved through u.
// ONE launch for every parameter group of every layer (a launch per group: ten 5 us launches and 2 x layers small copies on a 1.5 ms
// step): block b works on group g with first_block[g] <= b < first_block[g + 1].  A layer's kernel hyper-parameters live on the host
// (they are launch arguments of the forward pass): their current values arrive as arguments, the updated ones go to the device copy
// and straight into a pinned host slot.
struct OptGroup {
  double* p; const double* g; double* m; double* v;   // parameter, gradient, Adam moments (unused by SGD)
  double* recip;                                      // != null: 1 / p is kept here as well (the dense head's staging scale)
  long n;
  int transform, hyp_layer;                           // hyp_layer >= 0: p is that layer's {variance, p1, p2} triple
  // sharded step (OptArgs::sharded): the group's offset in its layer's gradient block, the layer, and whether it only passes through
  long off; int layer, frozen;
};
constexpr int OPT_GROUPS_MAX = 48;
struct OptArgs {
  OptGroup grp[OPT_GROUPS_MAX];
  int first_block[OPT_GROUPS_MAX + 1];
  int ng, sgd;
  double lr, b1, b2, eps;            // lr: Adam's bias-corrected step size, or the plain SGD rate
  double hyp_in[8][3];
  double* host_out;                  // [8][3] pinned
  const double* status;              // != null: a non-zero word there (the step's factorisation failed) leaves every parameter untouched
  // Sharded step (dcgp_model_set_grad_exchange 1): only the elements whose position in the layer's block lies in [sh_lo, sh_hi) are updated --
  // the gradient there is this rank's reduce-scattered sum -- and every value of the range, updated or frozen, also goes to stage[layer]
  // (the block's layout), which the all-gather then completes.  stage_only: the parameter arrays are left alone (the debug entry's other ranks).
  int sharded, stage_only;
  long sh_lo[8], sh_hi[8];
  double* stage[8];
};
__global__ __launch_bounds__(256) void opt_step_kernel(OptArgs a) {
  if (a.status && *a.status != 0.0) return;
  int gi = 0;
  while (gi + 1 < a.ng && (int)blockIdx.x >= a.first_block[gi + 1]) ++gi;
  const OptGroup& G = a.grp[gi];
  const long i = (long)((int)blockIdx.x - a.first_block[gi]) * 256 + threadIdx.x;
  if (i >= G.n) return;
  if (a.sharded && (G.off + i < a.sh_lo[G.layer] || G.off + i >= a.sh_hi[G.layer])) return;
  const double x = G.hyp_layer >= 0 ? a.hyp_in[G.hyp_layer][i] : G.p[i];
  double out;
  if (G.frozen) {
    out = x;
  } else if (a.sgd) {   // plain gradient ascent on the ELBO in the unconstrained space
    const double gr = G.g[i];
    if (G.transform == 1) {
      const double y = x - 1e-6;
      double u = y > 35.0 ? y : log(expm1(y));
      u += a.lr * gr * -expm1(-y);
      out = (u > 35.0 ? u : log1p(exp(u))) + 1e-6;
    } else {
      out = x + a.lr * gr;
    }
  } else {
    double gr = -G.g[i];                               // minimise -ELBO
    double u = x;
    if (G.transform == 1) {
      const double y = x - 1e-6;
      u = y > 35.0 ? y : log(expm1(y));               // softplus^-1
      gr *= -expm1(-y);                                // dx/du = sigmoid(u) = 1 - exp(-y)
    }
    const double mi = a.b1 * G.m[i] + (1.0 - a.b1) * gr;
    const double vi = a.b2 * G.v[i] + (1.0 - a.b2) * gr * gr;
    G.m[i] = mi;
    G.v[i] = vi;
    u -= a.lr * mi / (sqrt(vi) + a.eps);
    out = G.transform == 1 ? (u > 35.0 ? u : log1p(exp(u))) + 1e-6 : u;
  }
  if (a.sharded) a.stage[G.layer][G.off + i] = out;
  if (G.frozen || a.stage_only) return;
  G.p[i] = out;
  if (G.recip) G.recip[i] = 1.0 / out;
  if (G.hyp_layer >= 0) a.host_out[3 * G.hyp_layer + i] = out;
}
// behind the all-gather: the other ranks' shards of the staged block into this rank's parameter arrays
__global__ __launch_bounds__(256) void opt_unstage_kernel(OptArgs a) {
  if (a.status && *a.status != 0.0) return;
  int gi = 0;
  while (gi + 1 < a.ng && (int)blockIdx.x >= a.first_block[gi + 1]) ++gi;
  const OptGroup& G = a.grp[gi];
  const long i = (long)((int)blockIdx.x - a.first_block[gi]) * 256 + threadIdx.x;
  if (i >= G.n || G.frozen) return;
  if (G.off + i >= a.sh_lo[G.layer] && G.off + i < a.sh_hi[G.layer]) return;   // this rank's own shard: already in place
  const double out = a.stage[G.layer][G.off + i];
  G.p[i] = out;
  if (G.recip) G.recip[i] = 1.0 / out;
  if (G.hyp_layer >= 0) a.host_out[3 * G.hyp_layer + i] = out;
}

// ---- host helpers ------------------------------------------------------------------------------------------------
struct Bk {   // per-backward bookkeeping
  dcgp_model* m;
  dcgp_ctx* ctx;
  std::string pfx;   // workspace prefix of the layer being processed
  int slot_v = 0, slot_l = 0, slot_b = 0;
  // hyper-parameter partial sums of the layer being processed: collected here, reduced into the layer's scalar slots by ONE launch at the end of
  // its reverse pass (end_layer; they feed nothing else, and each used to be a 5 us launch in the middle of the data path)
  struct PendingSum { const double* in; long n; double scale; double* out; };
  std::vector<PendingSum> pending;
  bool last_layer = false;     // the layer being processed is the last one of the reverse pass (layer 0)
  int prep = 0;                // the layer being processed: bit 0 fills + GT + Lc, bit 1 S_r done beside the forward pass (grad_kl_early)
  bool kl_early = false;       // the layer being processed had its kl_products beside the forward pass (grad_kl_early)
  bool side_pending = false;   // a layer left the end of its reverse pass on the side stream: model_backward joins once, at the end
  double klw = 1.0;   // weight of the (replicated) KL term on this rank: 1 / number of batch shards
  double* ws(const char* name, size_t n_doubles) { return (double*)ws_get(ctx, pfx + "g_" + name, (n_doubles ? n_doubles : 1) * sizeof(double)); }
};

// The three streams of a layer's reverse pass.  main: the data path (the column-wise adjoint, the patch-kernel adjoint, dX for the layer
// below).  chain: the M x M chain of the conditional (W_r -> dG_r -> dq_sqrt -> dL -> Cholesky adjoint -> S).  tail: what needs S -- the Gram
// adjoint of K_uu, the KL pieces, the scalar sums, dZ.  Nothing on chain or tail feeds the layer below: the main stream never waits for
// them inside a step (model_backward joins once, at the end).  Option grad_nofork / no side stream: all three are the main stream.
struct Lanes { hipStream_t main, chain, tail; bool forked; };
Lanes lanes_of(dcgp_ctx* c) {
  Lanes l;
  l.main = c->stream;
  l.forked = !c->opt.grad_nofork && !c->no_side && c->stream2 && c->stream2 != c->stream;
  l.chain = l.forked ? c->stream2 : c->stream;
  l.tail = l.forked ? (c->stream_aux ? c->stream_aux : c->stream2) : c->stream;
  return l;
}
// launches issued in its lifetime go to `s`, behind `after` (an event already recorded on another stream) when s is not the current stream
struct OnStream {
  dcgp_ctx* ctx;
  hipStream_t saved;
  bool ok = true;
  OnStream(dcgp_ctx* c, hipStream_t s, hipEvent_t after) : ctx(c), saved(c->stream) {
    if (s != saved && after && hipStreamWaitEvent(s, after, 0) != hipSuccess) ok = false;
    c->stream = s;
  }
  ~OnStream() { ctx->stream = saved; }
};

GenGemm mk(const double* A, long ars, long acs, const double* B, long brs, long bcs, double* C, long crs, int M, int N, int K) {
  GenGemm g;
  g.A = A; g.a_rs = ars; g.a_cs = acs; g.B = B; g.b_rs = brs; g.b_cs = bcs; g.C = C; g.c_rs = crs; g.M = M; g.N = N; g.K = K;
  return g;
}

#define NEED(p) do { if (!(p)) return DCGP_ERR_ALLOC; } while (0)

// scale * sum(part[0..n)) goes to the next slot of the layer: which = 0 variance, 1 lengthscale | acos weight variance, 2 acos bias variance
int add_scalar(Bk& bk, LayerState& L, int which, const double* part, long n, double scale) {
  int& s = which == 0 ? bk.slot_v : (which == 1 ? bk.slot_l : bk.slot_b);
  if (s >= 16) return ctx_fail(bk.ctx, DCGP_ERR_ARG, "grad: out of scalar slots");
  bk.pending.push_back({part, n, scale, L.gslots + (which == 0 ? VAR_SLOT : (which == 1 ? LS_SLOT : P2_SLOT)) + s});
  ++s;
  return DCGP_OK;
}
struct ScalarPart { int which; const double* part; long n; double scale; };
int add_scalars(Bk& bk, LayerState& L, std::initializer_list<ScalarPart> parts) {
  for (const ScalarPart& p : parts) DCGP_TRY(add_scalar(bk, L, p.which, p.part, p.n, p.scale));
  return DCGP_OK;
}
int flush_scalars(Bk& bk) {
  for (size_t i = 0; i < bk.pending.size(); i += REDUCE_JOBS_MAX) {
    ReduceJobs j{};
    int k = 0;
    for (; k < REDUCE_JOBS_MAX && i + k < bk.pending.size(); ++k) {
      const Bk::PendingSum& p = bk.pending[i + k];
      j.in[k] = p.in; j.n[k] = p.n; j.scale[k] = p.scale; j.out[k] = p.out;
    }
    DCGP_TRY(reduce_sum_multi(bk.ctx, j, k));
  }
  bk.pending.clear();
  return DCGP_OK;
}

int im2col(dcgp_ctx* ctx, const LayerState& L, const double* Xin, int n_mod, long Kc, double* Xcol) {
  const long rows = Kc / L.v.P;
  if (rows * L.v.Ho > 0x7fffffffL) return ctx_fail(ctx, DCGP_ERR_ARG, "grad: too many patch rows");
  hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)(rows * L.v.Ho)), dim3(256), 0, ctx->stream, Xin, n_mod, L.v.H, L.v.W, L.v.C, L.v.f, L.v.s,
                     L.v.Ho, L.v.Wo, L.v.L, Xcol);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

// the hyper-parameter partial sums a Gram backward left in its scratch (`tag`) -> the layer's scalar slots
int kuu_scalars(Bk& bk, LayerState& L, const char* tag) {
  const std::string t(tag);
  const int M = L.M;
  if (L.base_type == 1) DCGP_TRY(add_scalar(bk, L, 2, bk.ws((t + "_pb").c_str(), M), M, 1.0));
  return add_scalars(bk, L, {{0, bk.ws((t + "_pv").c_str(), M), M, 1.0}, {1, bk.ws((t + "_pl").c_str(), M), M, 1.0}});
}

// RBF Gram backward from S = d ELBO / dK (unsymmetrised).  dZ accumulates into L.gZ when Zsrc is the live Z (want_dz).
// tag: prefix of its scratch; defer_scalars: leave the hyper-parameter partial sums there (kuu_scalars adds them later -- the scalar
// slots of a layer are zeroed at the start of its reverse pass, and the KL half of a frozen prior runs before that)
// dz_after: the product that adds into dZ is enqueued behind this event (the patch adjoint adds into the same dZ on another stream)
int kuu_backward(Bk& bk, LayerState& L, const double* Zsrc, const double* S, long lds, bool want_dz, double* dz_out = nullptr,
                 const char* tag = "kuu", bool defer_scalars = false, hipEvent_t dz_after = nullptr) {
  dcgp_ctx* ctx = bk.ctx;
  const int M = L.M, Ld = L.v.L;
  const double inv_l2 = 1.0 / (L.ls * L.ls), inv_l3 = inv_l2 / L.ls;
  const std::string t(tag);
  double *Es = nullptr, *rs = nullptr;
  double* pv = bk.ws((t + "_pv").c_str(), M);
  double* pl = bk.ws((t + "_pl").c_str(), M);
  NEED(pv); NEED(pl);
  if (want_dz) { Es = bk.ws((t + "_Es").c_str(), (size_t)M * M); rs = bk.ws((t + "_rs").c_str(), M); NEED(Es); NEED(rs); }
  double cz = inv_l2;      // factor of the dZ combination: 1 / l^2 (RBF) or the weight variance (ArcCosine)
  if (L.base_type == 1) {
    double* pb = bk.ws((t + "_pb").c_str(), M);
    double* znv = bk.ws((t + "_zn").c_str(), M);
    NEED(pb); NEED(znv);
    hipLaunchKernelGGL(rownorm_kernel, dim3(blocks_for(M)), dim3(256), 0, ctx->stream, Zsrc, (long)M, Ld, znv);
    LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(acos_kuu_backward_kernel, dim3(M), dim3(256), 0, ctx->stream, Zsrc, znv, M, Ld, S, lds, L.variance, L.acos_w, L.acos_b,
                       Es, (long)M, rs, pv, pl, pb);
    LAUNCH_CHECK(ctx);
    cz = L.acos_w;
  } else {
    hipLaunchKernelGGL(kuu_backward_kernel, dim3(M), dim3(256), 0, ctx->stream, Zsrc, Zsrc == L.Z ? L.ZT : nullptr, L.Mp, M, Ld, S, lds, L.variance,
                       inv_l2, inv_l3, Es, (long)M, rs, pv, pl);
    LAUNCH_CHECK(ctx);
  }
  if (!defer_scalars) DCGP_TRY(kuu_scalars(bk, L, tag));
  if (want_dz) {
    // dZ += cz (Es Z - rs o Z): the correction rides in the product's epilogue
    if (dz_after) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, dz_after, 0));
    GenGemm e = mk(Es, M, 1, Zsrc, Ld, 1, dz_out ? dz_out : L.gZ, Ld, M, Ld, M);
    e.alpha = cz; e.accumulate = 1; e.sub_v = rs; e.sub_x = Zsrc; e.sx_rs = Ld;
    DCGP_TRY(gemm_gen(ctx, e));
  }
  return DCGP_OK;
}

// Patch-kernel backward: E [M x Kc] (ld) = d ELBO / dKfull o Kfull is ready, cs = its column sums.
// dZ += (E Xcol - rowsum(E) o Z) / l^2;  dXcol (=|+=) (E^T Z - cs o Xcol) / l^2 when requested.
// dz_lanes (the head): the dZ product goes to the TAIL stream, behind an event recorded here -- nothing on the main stream reads dZ, the tail stream is
// idle at this point of the head's reverse pass and adds its own term into the same dZ later (kuu_backward: in stream order behind this one), and the two
// long products of the patch adjoint (73 + 86 us at the headline size, neither of them bound by the matrix pipe) then run beside each other
int patch_backward(Bk& bk, LayerState& L, const double* E, long ld, long Kc, const double* cs, const double* Xcol, double* dXcol,
                   int dx_accumulate, const double* Zuse = nullptr, double* dz_out = nullptr, const double* rs_in = nullptr, double cz = 0.0,
                   const Lanes* dz_lanes = nullptr) {
  dcgp_ctx* ctx = bk.ctx;
  const int M = L.M, Ld = L.v.L;
  const double inv_l2 = cz != 0.0 ? cz : 1.0 / (L.ls * L.ls);   // cz: the ArcCosine adjoint passes its weight variance (and its own row vector)
  double* rs = rs_in ? const_cast<double*>(rs_in) : bk.ws("pb_rs", M);
  NEED(rs);
  if (!rs_in) {
    const int chunks = (int)std::min<long>(32, (Kc + 4095) / 4096);
    const long cpc = round_up_l((Kc + chunks - 1) / chunks, 256);
    double* rsp = bk.ws("pb_rsp", (size_t)chunks * M);
    NEED(rsp);
    hipLaunchKernelGGL(rowsum_big_kernel, dim3(M, chunks), dim3(256), 0, ctx->stream, E, ld, Kc, cpc, M, rsp);
    LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(sum_chunks_kernel, dim3(blocks_for(M)), dim3(256), 0, ctx->stream, rsp, chunks, (long)M, rs);
    LAUNCH_CHECK(ctx);
  }
  const double* Zp = Zuse ? Zuse : L.Z;
  {   // dZ += (E Xcol - rs o Z) / l^2: one product over the columns (split along k), the correction in its epilogue
    GenGemm e = mk(E, ld, 1, Xcol, Ld, 1, dz_out ? dz_out : L.gZ, Ld, M, Ld, (int)Kc);
    e.alpha = inv_l2; e.accumulate = 1; e.sub_v = rs; e.sub_x = Zp; e.sx_rs = Ld;
    if (dz_lanes && dz_lanes->forked && !dz_out && !ctx->opt.grad_dz_main) {
      HIP_TRY(ctx, hipEventRecord(ctx->ev_g[5], ctx->stream));   // E, Xcol, rs are there
      OnStream on(ctx, dz_lanes->tail, ctx->ev_g[5]);
      if (!on.ok) return ctx_fail(ctx, DCGP_ERR_HIP, "grad: stream wait failed");
      DCGP_TRY(gemm_gen(ctx, e));
    } else {
      DCGP_TRY(gemm_gen(ctx, e));
    }
  }
  if (dXcol) {   // dXcol (+)= (E^T Z - cs o Xcol) / l^2
    GenGemm e = mk(E, 1, ld, Zp, Ld, 1, dXcol, Ld, (int)Kc, Ld, M);
    e.alpha = inv_l2; e.accumulate = dx_accumulate; e.sub_v = cs; e.sub_x = Xcol; e.sx_rs = Ld;
    DCGP_TRY(gemm_gen(ctx, e));
  }
  return DCGP_OK;
}

// KL backward of one layer (ELBO = ... - KL), in two halves.
// kl_products: everything that is a product of parameter-only matrices -- inv(K) q_mu, inv(K) Lq_r, -dKL/dK_prior -- into scratch.  The last one
// goes into `Sacc` [M x M, ld Mp] (head: the prior shares the live Z; s_first: Sacc is written, otherwise added to) or, Sacc == null, through
// the Gram backward on the frozen Z0 (conv layers), whose partial sums stay in scratch.  Nothing here reads the step's data or writes a
// gradient buffer: grad_kl_early runs it beside the FORWARD pass, where the side stream is idle -- at the end of a layer's reverse pass
// (fourteen short launches) it was the tail of the whole step.
// kl_apply: adds the pieces to dq_mu / dq_sqrt and the scalar slots (a whitened layer has no products: its KL adjoint is q_mu and Lq themselves).
int kl_products(Bk& bk, LayerState& L, double* Sacc, bool s_first) {
  dcgp_ctx* ctx = bk.ctx;
  if (L.white) return DCGP_OK;
  const int M = L.M, Mp = L.Mp, R = L.R, Rp = L.g.Rp;
  const long mm = (long)Mp * Mp;
  const double kw = bk.klw;
  const double* Lpinv = L.g.Kp ? L.g.Lpinv : L.g.Linv;
  double* a1 = bk.ws("kl_a1", (size_t)Mp * Rp);
  double* Kimu = bk.ws("kl_Kimu", (size_t)Mp * Rp);
  double* Wm = bk.ws("kl_W", (size_t)R * mm);
  double* KiL = bk.ws("kl_KiL", (size_t)R * mm);
  NEED(a1); NEED(Kimu); NEED(Wm); NEED(KiL);
  DCGP_TRY(gemm_gen(ctx, mk(Lpinv, Mp, 1, L.q_mu, R, 1, a1, Rp, M, R, M)));
  DCGP_TRY(gemm_gen(ctx, mk(Lpinv, 1, Mp, a1, Rp, 1, Kimu, Rp, M, R, M)));
  GenGemm g1 = mk(Lpinv, Mp, 1, L.g.Lq, Mp, 1, Wm, Mp, M, M, M);
  g1.batch = R; g1.b_bs = mm; g1.c_bs = mm;
  DCGP_TRY(gemm_gen(ctx, g1));
  const long Rm = (long)R * Mp;
  if (Mp > M) HIP_TRY(ctx, hipMemsetAsync(KiL, 0, (size_t)R * mm * sizeof(double), ctx->stream));   // (the padded k of the stacked product below)
  GenGemm g2 = mk(Lpinv, 1, Mp, Wm, Mp, 1, KiL, Rm, M, M, M);        // inv(K) Lq_r, stored [i][r][k]
  g2.batch = R; g2.b_bs = mm; g2.c_bs = Mp;
  DCGP_TRY(gemm_gen(ctx, g2));
  // -dKL/dK = -1/2 [R inv(K) - (inv(K) q_mu)(inv(K) q_mu)^T - sum_r (inv(K) Lq_r)(inv(K) Lq_r)^T]
  double* Sk = Sacc;
  int acc = s_first ? 0 : 1;
  if (!Sk) { Sk = bk.ws("kl_S", (size_t)mm); NEED(Sk); acc = 0; }
  GenGemm g3 = mk(Lpinv, 1, Mp, Lpinv, Mp, 1, Sk, Mp, M, M, M);
  g3.alpha = -0.5 * R * kw; g3.accumulate = acc;
  DCGP_TRY(gemm_gen(ctx, g3));
  GenGemm g4 = mk(Kimu, Rp, 1, Kimu, 1, Rp, Sk, Mp, M, M, R);
  g4.alpha = 0.5 * kw; g4.accumulate = 1;
  DCGP_TRY(gemm_gen(ctx, g4));
  GenGemm g5 = mk(KiL, Rm, 1, KiL, 1, Rm, Sk, Mp, M, M, (int)Rm);     // all r at once, stacked along k
  g5.alpha = 0.5 * kw; g5.accumulate = 1;
  DCGP_TRY(gemm_gen(ctx, g5));
  if (!Sacc) DCGP_TRY(kuu_backward(bk, L, L.Z0, Sk, Mp, false, nullptr, "klz", true));   // frozen Z0: hyper-parameters only
  return DCGP_OK;
}
int kl_apply(Bk& bk, LayerState& L, bool frozen_prior) {
  dcgp_ctx* ctx = bk.ctx;
  if (bk.kl_early) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_kl3, 0));   // the products came from the auxiliary stream
  const int M = L.M, Mp = L.Mp, R = L.R, Rp = L.g.Rp;
  const long mm = (long)Mp * Mp;
  const double kw = bk.klw;
  if (L.white) {
    hipLaunchKernelGGL(copy2d_kernel, dim3(blocks_for(R), M), dim3(256), 0, ctx->stream, L.q_mu, (long)R, L.gq_mu, (long)R, M, R, -kw, 1);
    LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(mask_lower_kernel, dim3(blocks_for(M), M, R), dim3(256), 0, ctx->stream, L.g.Lq, (long)Mp, mm, L.gq_sqrt, (long)M,
                       (long)M * M, M, -kw, 1);
    LAUNCH_CHECK(ctx);
  } else {
    const long Rm = (long)R * Mp;
    const double* Kimu = bk.ws("kl_Kimu", (size_t)Mp * Rp);
    const double* KiL = bk.ws("kl_KiL", (size_t)R * mm);
    NEED(Kimu); NEED(KiL);
    hipLaunchKernelGGL(copy2d_kernel, dim3(blocks_for(R), M), dim3(256), 0, ctx->stream, Kimu, (long)Rp, L.gq_mu, (long)R, M, R, -kw, 1);
    LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(mask_lower_kernel, dim3(blocks_for(M), M, R), dim3(256), 0, ctx->stream, KiL, Rm, (long)Mp, L.gq_sqrt, (long)M,
                       (long)M * M, M, -kw, 1);
    LAUNCH_CHECK(ctx);
    if (frozen_prior) DCGP_TRY(kuu_scalars(bk, L, "klz"));
  }
  hipLaunchKernelGGL(kl_diag_kernel, dim3(blocks_for((long)M * R)), dim3(256), 0, ctx->stream, L.gq_sqrt, L.g.Lq, M, Mp, R, kw);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

// Parameter-only operands of cond_backward: sgg == false: G^T stacked along k (GT) and the lower triangle of the factor (Lc);
// sgg == true: S_r = G_r G_r^T, the strip kernel's operand.
int param_operands(Bk& bk, LayerState& L, bool sgg) {
  dcgp_ctx* ctx = bk.ctx;
  const int M = L.M, Mp = L.Mp, R = L.R;
  const long mm = (long)Mp * Mp;
  const GpMats& g = L.g;
  if (sgg) {
    double* Sgg = bk.ws("Sgg", (size_t)R * mm);
    NEED(Sgg);
    if (Mp > M) HIP_TRY(ctx, hipMemsetAsync(Sgg, 0, (size_t)R * mm * sizeof(double), ctx->stream));
    GenGemm sg = mk(g.G, Mp, 1, g.G, 1, Mp, Sgg, Mp, M, M, M);
    sg.batch = R; sg.a_bs = mm; sg.b_bs = mm; sg.c_bs = mm;
    return gemm_gen(ctx, sg);
  }
  if (L.has_qsqrt) {
    double* GT = bk.ws("GT", (size_t)R * mm);
    NEED(GT);
    hipLaunchKernelGGL(restack_transpose_kernel, dim3(blocks_for(R * mm)), dim3(256), 0, ctx->stream, g.G, Mp, R, GT);
    LAUNCH_CHECK(ctx);
  }
  double* Lc = bk.ws("Lc", (size_t)mm);
  NEED(Lc);
  hipLaunchKernelGGL(mask_lower_kernel, dim3(blocks_for(M), M, 1), dim3(256), 0, ctx->stream, g.K, (long)Mp, 0L, Lc, (long)Mp, 0L, M, 1.0, 0);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

// The terms of dL that do not come out of the chain:  dL = -tril(dq_mu alpha^T) [unwhitened; the first writer of dL],  dL (+)= -tril(dKuf A1^T).
// The first one is ALWAYS formed on the main stream, right behind dq_mu: the tail stream adds the KL part into dq_mu as soon as ev_g[1] is past.
int dl_term_mu(Bk& bk, LayerState& L, double* dL) {
  GenGemm l1 = mk(L.gq_mu, L.R, 1, L.g.alpha, 1, L.g.Rp, dL, L.Mp, L.M, L.M, L.R);
  l1.alpha = -1.0; l1.lower_only = 1;
  return gemm_gen(bk.ctx, l1);
}
int dl_term_kuf(Bk& bk, LayerState& L, const double* A1, long ld, long Kc, const double* dKuf, double* dL) {
  GenGemm l3 = mk(dKuf, ld, 1, A1, 1, ld, dL, L.Mp, L.M, L.M, (int)Kc);
  l3.alpha = -1.0; l3.lower_only = 1; l3.accumulate = L.white ? 0 : 1;   // (whitened: the only term)
  return gemm_gen(bk.ctx, l3);
}

// The conditional's backward shared by conv layers and the head, in two calls.  Kuf, A1: [Mp x ld] with Kc live columns; gm, gv: [Kc][R].
// cond_backward_main (main stream): dKuf [M x ld], gvs [Kc] = sum_r gv (= d ELBO / d Knn), dq_mu; enqueues the chain's first part (W_r -> dG_r
//   -> dq_sqrt and its term of dL) on the chain stream behind ev_g[0] and marks its own end with ev_g[1].
// cond_backward_finish (chain stream, behind ev_g[1]): dL's other terms and the Cholesky adjoint, S = d ELBO / dKuu (data part) [M x M, ld Mp];
//   s_acc: S already holds the KL part (kl_products ran first): add to it.  Marks its end with ev_g[2].
// dl_on_main: dL's term from dK_uf is formed on the main stream as well (the last layer of the reverse pass: nothing follows it there, and the
// chain is what the step ends on).
int cond_backward_main(Bk& bk, const Lanes& ln, LayerState& L, const double* A1, long ld, long Kc, const double* gm, const double* gv, double* dKuf,
                       double* gvs, bool dl_on_main) {
  dcgp_ctx* ctx = bk.ctx;
  const int M = L.M, Mp = L.Mp, R = L.R, Rp = L.g.Rp;
  const long mm = (long)Mp * Mp;
  const GpMats& g = L.g;
  if (ln.forked) HIP_TRY(ctx, hipEventRecord(ctx->ev_g[0], ln.main));   // A1 and gv are there: the chain may start
  hipLaunchKernelGGL(rowsum_small_kernel, dim3(blocks_for(Kc)), dim3(256), 0, ctx->stream, gv, Kc, R, gvs);
  LAUNCH_CHECK(ctx);
  // The column-wise part of the adjoint -- dT, dA1, dK_uf -- in one strip-resident launch where the shape allows and there are
  // enough columns to fill the chip (conv_bwd_fused.hip); its operand S_r = G_r G_r^T is parameter-only.
  ConvBwdArgs fb;
  fb.A1 = A1; fb.ld = ld; fb.Kc = (int)Kc; fb.alpha = L.g.alpha; fb.Rp = Rp; fb.Linv = L.g.Linv; fb.gv = gv; fb.gm = gm; fb.gvs = gvs;
  fb.M = M; fb.Mp = Mp; fb.R = R; fb.dKuf = dKuf;
  const long min_cols = ctx->opt.fused_bwd_min_cols >= 0 ? ctx->opt.fused_bwd_min_cols : 4096;   // (tests: the strip kernel at sizes the oracle checks)
  const bool fused_bwd = L.has_qsqrt && !L.white && Kc >= min_cols && conv_bwd_fused_ok(ctx, fb);
  double* dA1 = fused_bwd ? nullptr : bk.ws("dA1", (size_t)Mp * ld);
  double* dalpha = bk.ws("dalpha", (size_t)Mp * Rp);
  double* dG = bk.ws("dG", (size_t)R * mm);
  double* dL = bk.ws("dL", (size_t)mm);
  if (!fused_bwd) NEED(dA1);
  NEED(dalpha); NEED(dG); NEED(dL);
  const long Rm = (long)R * Mp;
  double* GT = nullptr;
  if (L.has_qsqrt) {
    GT = bk.ws("GT", (size_t)R * mm);
    NEED(GT);
  }
  // (parameter-only operands: G^T stacked, S_r = G_r G_r^T, the factor's lower triangle -- already there when grad_kl_early prepared them)
  if (!(bk.prep & 1)) DCGP_TRY(param_operands(bk, L, false));
  if (fused_bwd) {
    if (!(bk.prep & 2)) DCGP_TRY(param_operands(bk, L, true));
    fb.S = bk.ws("Sgg", (size_t)R * mm);
    NEED(fb.S);
  }
  if (!(bk.prep & 1) && ln.forked) HIP_TRY(ctx, hipEventRecord(ctx->ev_g[0], ln.main));   // (the chain reads G^T: behind the launch that made it)
  // main stream: dT, dA1, dK_uf on the tuned kernels ...
  if (fused_bwd) {
    // (the chain's W_r contraction runs BESIDE this launch also where both fill the chip -- tens of thousands of columns: started behind it
    // instead, the tiled step took 4.07 ms against 3.90)
    DCGP_TRY(conv_bwd_fused(ctx, fb));
  } else {
    int dA1_acc = 0;
    // the padded rows M..Mp-1 of dA1 are operands of the gemm_tn launch that forms dK_uf (times zeros of inv(L)'s padding):
    // they must be finite whichever path writes the live rows
    if (Mp > M) HIP_TRY(ctx, hipMemsetAsync(dA1 + (size_t)M * ld, 0, (size_t)(Mp - M) * ld * sizeof(double), ctx->stream));
    if (L.has_qsqrt) {
      double* dT = bk.ws("dT", (size_t)R * Mp * ld);
      NEED(dT);
      const bool tn_ok = (long)Mp * ld * 8 < (1L << 31);
      if (tn_ok) {   // dT_r = 2 (G_r^T A1) o gv_r: the forward's stage-3 product, stored, columns scaled in the epilogue
        GemmArgs a;
        a.Wt = g.G; a.ldw = Mp; a.wBatch = mm; a.nW = R;
        a.B = A1; a.ldb = (int)ld;
        a.C = dT; a.ldc = (int)ld; a.cBatch = (long)Mp * ld;
        a.cscale = gv; a.csCol = R; a.csBatch = 1; a.calpha = 2.0;
        a.Mi = Mp; a.Mk = Mp; a.Kc = (int)Kc; a.tri = 2;
        DCGP_TRY(gemm_tn(ctx, a, nullptr));
      } else {
        GenGemm t = mk(g.G, 1, Mp, A1, ld, 1, dT, ld, Mp, (int)Kc, Mp);
        t.batch = R; t.a_bs = mm; t.c_bs = (long)Mp * ld; t.alpha = 2.0; t.colscale = gv; t.cs_s = R; t.cs_bs = 1;
        DCGP_TRY(gemm_gen(ctx, t));
      }
      // dA1 = sum_r G_r dT_r: ONE product with the R blocks stacked along k (GT [R Mp x Mp], dT [R Mp x ld])
      // (few columns -- a de-duplicated first layer, the head: the 72-workgroup launch would be one long k chain; gemm_gen splits k)
      if (Rm * ld * 8 < (1L << 31) && Kc >= 16384) {
        GemmArgs a;
        a.Wt = GT; a.ldw = Mp;
        a.B = dT; a.ldb = (int)ld;
        a.C = dA1; a.ldc = (int)ld;
        a.Mi = Mp; a.Mk = (int)Rm; a.Kc = (int)Kc; a.tri = 0;
        DCGP_TRY(gemm_tn(ctx, a, nullptr));
      } else {
        DCGP_TRY(gemm_gen(ctx, mk(GT, 1, Mp, dT, ld, 1, dA1, ld, M, (int)Kc, (int)Rm)));
      }
      dA1_acc = 1;
    }
    // dA1 (+)= alpha gm^T - 2 A1 o gvs
    {
      GenGemm a = mk(g.alpha, Rp, 1, gm, 1, R, dA1, ld, M, (int)Kc, R);
      a.accumulate = dA1_acc;
      DCGP_TRY(gemm_gen(ctx, a));
    }
    hipLaunchKernelGGL(dA1_fix_kernel, dim3(blocks_for(Kc), M), dim3(256), 0, ctx->stream, dA1, A1, gvs, M, Kc, ld);
    LAUNCH_CHECK(ctx);
    // dKuf = inv(L)^T dA1
    if ((long)Mp * ld * 8 < (1L << 31)) {   // inv(L) row-major IS the k-major operand of inv(L)^T; upper-triangular product
      GemmArgs a;
      a.Wt = g.Linv; a.ldw = Mp;
      a.B = dA1; a.ldb = (int)ld;
      a.C = dKuf; a.ldc = (int)ld;
      a.Mi = Mp; a.Mk = Mp; a.Kc = (int)Kc; a.tri = 2;
      DCGP_TRY(gemm_tn(ctx, a, nullptr));
    } else {
      DCGP_TRY(gemm_gen(ctx, mk(g.Linv, 1, Mp, dA1, ld, 1, dKuf, ld, M, (int)Kc, M)));
    }
  }
  // ... then d alpha = A1 gm and dq_mu = d alpha (whitened) or inv(L)^T d alpha.  (d alpha used to open the chain: beside the strip kernel
  // its 288 small workgroups took 108 us instead of 8, in front of W_r.)
  DCGP_TRY(gemm_gen(ctx, mk(A1, ld, 1, gm, R, 1, dalpha, Rp, M, R, (int)Kc)));
  if (L.white) {
    hipLaunchKernelGGL(copy2d_kernel, dim3(blocks_for(R), M), dim3(256), 0, ctx->stream, dalpha, (long)Rp, L.gq_mu, (long)R, M, R, 1.0, 0);
    LAUNCH_CHECK(ctx);
  } else {
    DCGP_TRY(gemm_gen(ctx, mk(g.Linv, 1, Mp, dalpha, Rp, 1, L.gq_mu, R, M, R, M)));
  }
  if (!L.white) DCGP_TRY(dl_term_mu(bk, L, dL));
  if (dl_on_main) DCGP_TRY(dl_term_kuf(bk, L, A1, ld, Kc, dKuf, dL));
  if (ln.forked) HIP_TRY(ctx, hipEventRecord(ctx->ev_g[1], ln.main));   // dK_uf, dq_mu and dL's first term(s) are there
  // The chain's first part, ENQUEUED behind the launches above (one host thread feeds the streams at ~4 us a launch; with few columns -- the
  // head -- the main stream's launches are as short as that: fed second, it sat idle while the host was busy with the chain's).
  if (L.has_qsqrt) {
    OnStream on(ctx, ln.chain, ctx->ev_g[0]);
    if (!on.ok) return ctx_fail(ctx, DCGP_ERR_HIP, "grad: stream wait failed");
    // dG_r = tril(A1 dT_r^T) = tril(W_r G_r),  W_r = 2 A1 diag(gv_r) A1^T (symmetric).  Both operands of the long
    // contraction are then A1 itself (94 MB at the headline size: it stays in the 256 MB Infinity Cache across the R
    // batches, where the R x larger dT would stream from HBM); lower tiles only, stored to both triangles.
    double* Wr = bk.ws("Wr", (size_t)R * mm);
    NEED(Wr);
    GenGemm w = mk(A1, ld, 1, A1, 1, ld, Wr, Mp, M, M, (int)Kc);
    // (the k scaling reads gv [Kc][R] in place, stride R: a transposed copy used to cost 290 us of scattered 8-byte stores per step)
    w.batch = R; w.c_bs = mm; w.lower_only = 1; w.mirror = 1; w.alpha = 2.0; w.kscale = gv; w.ks_s = R; w.ks_bs = 1;
    {
      ScopedTimer t(ctx, L.is_head ? "grad_wr_head" : "grad_wr");   // (bench.py: roofline_train)
      DCGP_TRY(gemm_gen(ctx, w));
    }
    GenGemm d = mk(Wr, Mp, 1, g.G, Mp, 1, dG, Mp, M, M, M);
    d.batch = R; d.a_bs = mm; d.b_bs = mm; d.c_bs = mm; d.lower_only = 1;
    DCGP_TRY(gemm_gen(ctx, d));
    if (L.white) {
      hipLaunchKernelGGL(mask_lower_kernel, dim3(blocks_for(M), M, R), dim3(256), 0, ctx->stream, dG, (long)Mp, mm, L.gq_sqrt, (long)M,
                         (long)M * M, M, 1.0, 0);
      LAUNCH_CHECK(ctx);
    } else {
      double* Bm = bk.ws("Bm", (size_t)R * mm);                      // B_r = inv(L)^T dG_r, stored [i][r][k]
      NEED(Bm);
      if (Mp > M) HIP_TRY(ctx, hipMemsetAsync(Bm, 0, (size_t)R * mm * sizeof(double), ctx->stream));   // (the padded k of the stacked product below)
      GenGemm b = mk(g.Linv, 1, Mp, dG, Mp, 1, Bm, Rm, M, M, M);
      b.batch = R; b.b_bs = mm; b.c_bs = Mp;
      DCGP_TRY(gemm_gen(ctx, b));
      hipLaunchKernelGGL(mask_lower_kernel, dim3(blocks_for(M), M, R), dim3(256), 0, ctx->stream, Bm, Rm, (long)Mp, L.gq_sqrt, (long)M,
                         (long)M * M, M, 1.0, 0);
      LAUNCH_CHECK(ctx);
      if (ln.forked) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_g[1], 0));   // (dL's first writer is on the main stream)
      GenGemm l2 = mk(Bm, Rm, 1, GT, Mp, 1, dL, Mp, M, M, (int)Rm);   // dL -= tril(sum_r B_r G_r^T), stacked along k
      l2.alpha = -1.0; l2.lower_only = 1; l2.accumulate = 1;
      DCGP_TRY(gemm_gen(ctx, l2));
    }
    if (ln.forked) HIP_TRY(ctx, hipEventRecord(ctx->ev_g[4], ctx->stream));   // dq_sqrt is there
  }
  return DCGP_OK;
}

int cond_backward_finish(Bk& bk, const Lanes& ln, LayerState& L, const double* A1, long ld, long Kc, const double* dKuf, double* S, bool s_acc,
                         bool dl_on_main) {
  dcgp_ctx* ctx = bk.ctx;
  const int M = L.M, Mp = L.Mp;
  const long mm = (long)Mp * Mp;
  const GpMats& g = L.g;
  OnStream on(ctx, ln.chain, ctx->ev_g[1]);
  if (!on.ok) return ctx_fail(ctx, DCGP_ERR_HIP, "grad: stream wait failed");
  double* dL = bk.ws("dL", (size_t)mm);
  double* Lc = bk.ws("Lc", (size_t)mm);
  double* Pm = bk.ws("Pm", (size_t)mm);
  double* S1 = bk.ws("S1", (size_t)mm);
  NEED(dL); NEED(Lc); NEED(Pm); NEED(S1);
  // dL = -tril(dq_mu alpha^T [main stream, unwhitened] + sum_r B_r G_r^T [the chain's first part, unwhitened with q_sqrt] + dKuf A1^T)
  if (!dl_on_main) DCGP_TRY(dl_term_kuf(bk, L, A1, ld, Kc, dKuf, dL));
  // Cholesky adjoint: S = inv(L)^T Phi(L^T dL) inv(L)
  GenGemm ph = mk(Lc, 1, Mp, dL, Mp, 1, Pm, Mp, M, M, M);
  ph.phi = 1;   // Phi(L^T dL) in the product's epilogue
  DCGP_TRY(gemm_gen(ctx, ph));
  DCGP_TRY(gemm_gen(ctx, mk(g.Linv, 1, Mp, Pm, Mp, 1, S1, Mp, M, M, M)));
  GenGemm sf = mk(S1, Mp, 1, g.Linv, Mp, 1, S, Mp, M, M, M);
  if (s_acc) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_kl3, 0));   // S holds the KL part, written on the auxiliary stream
  sf.accumulate = s_acc ? 1 : 0;
  DCGP_TRY(gemm_gen(ctx, sf));
  if (ln.forked) HIP_TRY(ctx, hipEventRecord(ctx->ev_g[2], ctx->stream));
  return DCGP_OK;
}

int e_form(Bk& bk, LayerState& L, const double* dK, long lddk, int pdiv, const double* w, double wscale, const double* K, long ldk,
           double* E, long lde, long Kc, double* cs, double* raw) {
  dcgp_ctx* ctx = bk.ctx;
  const unsigned nb = blocks_for(Kc);
  // enough row chunks to put a few thousand blocks on the chip
  int chunks = (int)std::min<long>(16, std::max<long>(1, 2048 / nb));
  chunks = std::min(chunks, (L.M + 15) / 16);
  const int rpc = (L.M + chunks - 1) / chunks;
  chunks = (L.M + rpc - 1) / rpc;
  double* pv = bk.ws("ef_pv", (size_t)nb * chunks);
  double* pl = bk.ws("ef_pl", (size_t)nb * chunks);
  double* csp = chunks > 1 ? bk.ws("ef_csp", (size_t)chunks * Kc) : cs;
  double* rawp = raw ? (chunks > 1 ? bk.ws("ef_rawp", (size_t)chunks * Kc) : raw) : nullptr;
  NEED(pv); NEED(pl); NEED(csp);
  if (raw) NEED(rawp);
  hipLaunchKernelGGL(e_form_kernel, dim3(nb, chunks), dim3(256), 0, ctx->stream, dK, lddk, pdiv, w, wscale, K, ldk, E, lde, L.M, rpc, Kc,
                     1.0 / L.variance, 2.0 * L.ls * L.ls, csp, rawp, pv, pl);
  LAUNCH_CHECK(ctx);
  if (chunks > 1) {
    if (raw) hipLaunchKernelGGL(sum_chunks2_kernel, dim3(blocks_for(Kc)), dim3(256), 0, ctx->stream, csp, rawp, chunks, Kc, cs, raw);
    else hipLaunchKernelGGL(sum_chunks_kernel, dim3(blocks_for(Kc)), dim3(256), 0, ctx->stream, csp, chunks, Kc, cs);
    LAUNCH_CHECK(ctx);
  }
  DCGP_TRY(add_scalars(bk, L, {{0, pv, (long)nb * chunks, 1.0 / L.variance}, {1, pl, (long)nb * chunks, 1.0 / (L.ls * L.ls * L.ls)}}));
  return DCGP_OK;
}

// the zero fills in front of a layer's reverse pass: scalar slots, gradient block
int layer_fills(Bk& bk, LayerState& L) {
  DCGP_TRY(L.ensure_grads());
  HIP_TRY(bk.ctx, hipMemsetAsync(L.gslots, 0, 48 * sizeof(double), bk.ctx->stream));
  if (L.grad_block_count() * sizeof(double) <= (8u << 20)) {   // a small block ([Z | q_mu | q_sqrt | w | gscal | gard], contiguous): one fill
    HIP_TRY(bk.ctx, hipMemsetAsync(L.gZ, 0, L.grad_block_count() * sizeof(double), bk.ctx->stream));
    return DCGP_OK;
  }
  HIP_TRY(bk.ctx, hipMemsetAsync(L.gZ, 0, (size_t)L.M * L.v.L * sizeof(double), bk.ctx->stream));
  HIP_TRY(bk.ctx, hipMemsetAsync(L.gw, 0, ((size_t)L.v.P + 3 + (L.is_head ? (size_t)L.v.L : 0)) * sizeof(double), bk.ctx->stream));   // gw, gscal, gard
  if (!L.has_qsqrt) HIP_TRY(bk.ctx, hipMemsetAsync(L.gq_sqrt, 0, (size_t)L.R * L.M * L.M * sizeof(double), bk.ctx->stream));
  return DCGP_OK;
}
int begin_layer(Bk& bk, LayerState& L) {
  bk.slot_v = bk.slot_l = bk.slot_b = 0;
  bk.pending.clear();
  if (bk.prep & 1) return DCGP_OK;   // grad_kl_early ran the fills beside the forward pass
  return layer_fills(bk, L);
}
int end_layer(Bk& bk, LayerState& L) {
  DCGP_TRY(flush_scalars(bk));
  hipLaunchKernelGGL(scal_finish_kernel, dim3(1), dim3(64), 0, bk.ctx->stream, L.gslots, L.gscal);
  LAUNCH_CHECK(bk.ctx);
  return DCGP_OK;
}

// What is left of a layer, on the tail stream.  Behind the conditional's own dq_mu / dq_sqrt (ev_g[1], ev_g[4]): the KL pieces.  Behind
// S = d ELBO / dKuu (ev_g[2]): the Gram adjoint of K_uu, whose product adds into the dZ the main stream's patch adjoint adds into (behind
// ev_g[3]).  Last the layer's scalar sums.  kd: d ELBO / d Knn per column (conv layers) or null.
int layer_tail(Bk& bk, const Lanes& ln, LayerState& L, const double* S, const double* kd, long n_kd, bool frozen_prior) {
  dcgp_ctx* ctx = bk.ctx;
  OnStream on(ctx, ln.tail, ctx->ev_g[1]);
  if (!on.ok) return ctx_fail(ctx, DCGP_ERR_HIP, "grad: stream wait failed");
  if (kd) DCGP_TRY(add_scalar(bk, L, 0, kd, n_kd, 1.0));          // Knn = variance on every column
  const bool kl_first = bk.kl_early || frozen_prior || L.white;   // (otherwise the KL products add into S: behind it)
  if (ln.forked && L.has_qsqrt) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_g[4], 0));
  if (kl_first) {
    if (!bk.kl_early) DCGP_TRY(kl_products(bk, L, nullptr, false));
    DCGP_TRY(kl_apply(bk, L, frozen_prior));
  }
  if (ln.forked) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_g[2], 0));
  if (!kl_first) {
    DCGP_TRY(kl_products(bk, L, const_cast<double*>(S), false));
    DCGP_TRY(kl_apply(bk, L, frozen_prior));
  }
  DCGP_TRY(kuu_backward(bk, L, L.Z, S, L.Mp, true, nullptr, "kuu", false, ln.forked ? ctx->ev_g[3] : nullptr));
  DCGP_TRY(end_layer(bk, L));
  if (ln.forked) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev_kl, ctx->stream));
    bk.side_pending = true;
  }
  return DCGP_OK;
}

// ConvLayer backward.  Xin: the layer's input images ([n_mod, H, W, C], row n reads image n % n_mod); gm / gv [rows * P][R].
int conv_backward(Bk& bk, LayerState& L, const double* Xin, int rows, int n_mod, const double* gm, const double* gv, double* dXin) {
  dcgp_ctx* ctx = bk.ctx;
  const int M = L.M, Mp = L.Mp, P = L.v.P, Ld = L.v.L;
  const long Kc = (long)rows * P, ld = col_ld(Kc);
  const Lanes ln = lanes_of(ctx);
  DCGP_TRY(begin_layer(bk, L));
  // forward leftovers (conv_forward / cond_core workspaces)
  auto itB = ctx->ws.find(bk.pfx + "Kuf"), itA = ctx->ws.find(bk.pfx + "A1");
  if (itB == ctx->ws.end() || itA == ctx->ws.end() || itA->second.second < (size_t)Mp * ld * sizeof(double))
    return ctx_fail(ctx, DCGP_ERR_ARG, "grad: the forward pass left no K_uf / A1 for this layer");
  const double* Kuf = (const double*)itB->second.first;
  const double* A1 = (const double*)itA->second.first;
  double* dKuf = bk.ws("dKuf", (size_t)Mp * ld);
  double* E = bk.ws("E", (size_t)Mp * ld);       // (not over dK_uf: the chain stream still reads that)
  double* S = bk.ws("S", (size_t)Mp * Mp);
  double* gvs = bk.ws("gvs", Kc);
  double* cs = bk.ws("cs", Kc);
  double* Xcol = bk.ws("Xcol", (size_t)Kc * Ld);
  NEED(dKuf); NEED(E); NEED(S); NEED(gvs); NEED(cs); NEED(Xcol);
  // main stream: the column-wise adjoint of the conditional, then the patch-kernel adjoint -> dX.  The M x M chain behind the conditional
  // runs beside them on the chain stream, what needs its result on the tail stream (Lanes).
  DCGP_TRY(cond_backward_main(bk, ln, L, A1, ld, Kc, gm, gv, dKuf, gvs, bk.last_layer && ln.forked));
  DCGP_TRY(im2col(ctx, L, Xin, n_mod, Kc, Xcol));
  double* dXcol = nullptr;
  if (dXin) { dXcol = bk.ws("dXcol", (size_t)Kc * Ld); NEED(dXcol); }
  if (L.base_type == 1) {   // ArcCosine(order 0): F1, F2 beside it, the RBF machinery on (F1, rowsum(F2) / Q, colsum(F2) / A, w)
    const unsigned nb = blocks_for(Kc);
    int chunks = (int)std::min<long>(16, std::max<long>(1, 2048 / nb));
    chunks = std::min(chunks, (M + 15) / 16);
    const int rpc = (M + chunks - 1) / chunks;
    chunks = (M + rpc - 1) / rpc;
    double* F2 = bk.ws("acos_F2", (size_t)Mp * ld);
    double* xn = bk.ws("acos_xn", Kc);
    double* csp = bk.ws("ef_csp", (size_t)chunks * Kc);
    double* pv = bk.ws("ef_pv", (size_t)nb * chunks);
    double* pw = bk.ws("ef_pl", (size_t)nb * chunks);
    double* pb = bk.ws("ef_pb", (size_t)nb * chunks);
    double* rs2 = bk.ws("acos_rs", M);
    double* rsp = bk.ws("pb_rsp", (size_t)32 * M);
    NEED(F2); NEED(xn); NEED(csp); NEED(pv); NEED(pw); NEED(pb); NEED(rs2); NEED(rsp);
    hipLaunchKernelGGL(rownorm_kernel, dim3(blocks_for(Kc)), dim3(256), 0, ctx->stream, Xcol, Kc, Ld, xn);
    LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(acos_e_form_kernel, dim3(nb, chunks), dim3(256), 0, ctx->stream, dKuf, E, ld, Kuf, F2, M, rpc, Kc, L.variance, L.acos_w, L.acos_b,
                       L.zn, xn, csp, pv, pw, pb);
    LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(sum_chunks_kernel, dim3(blocks_for(Kc)), dim3(256), 0, ctx->stream, csp, chunks, Kc, cs);
    LAUNCH_CHECK(ctx);
    DCGP_TRY(add_scalars(bk, L, {{0, pv, (long)nb * chunks, 1.0 / L.variance}, {1, pw, (long)nb * chunks, 1.0}, {2, pb, (long)nb * chunks, 1.0}}));
    {   // rowsum(F2) / Q
      const int rch = (int)std::min<long>(32, (Kc + 4095) / 4096);
      const long cpc = round_up_l((Kc + rch - 1) / rch, 256);
      hipLaunchKernelGGL(rowsum_big_kernel, dim3(M, rch), dim3(256), 0, ctx->stream, F2, ld, Kc, cpc, M, rsp);
      LAUNCH_CHECK(ctx);
      hipLaunchKernelGGL(sum_chunks_kernel, dim3(blocks_for(M)), dim3(256), 0, ctx->stream, rsp, rch, (long)M, rs2);
      LAUNCH_CHECK(ctx);
      hipLaunchKernelGGL(acos_divide_kernel, dim3(blocks_for(M)), dim3(256), 0, ctx->stream, rs2, L.zn, M, L.acos_w, L.acos_b);
      LAUNCH_CHECK(ctx);
    }
    DCGP_TRY(patch_backward(bk, L, E, ld, Kc, cs, Xcol, dXcol, 0, nullptr, nullptr, rs2, L.acos_w));
  } else {
    DCGP_TRY(e_form(bk, L, dKuf, ld, 1, nullptr, 1.0, Kuf, ld, E, ld, Kc, cs, nullptr));
    DCGP_TRY(patch_backward(bk, L, E, ld, Kc, cs, Xcol, dXcol, 0));
  }
  if (dXin) {
    const long n = (long)rows * L.v.H * L.v.W * L.v.C;
    hipLaunchKernelGGL(col2im_kernel, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, dXcol, rows, L.v.H, L.v.W, L.v.C, L.v.f, L.v.s, L.v.Ho,
                       L.v.Wo, Ld, dXin);
    LAUNCH_CHECK(ctx);
    if (L.identity_mean) {
      hipLaunchKernelGGL(idmean_backward_kernel, dim3(blocks_for(Kc)), dim3(256), 0, ctx->stream, gm, Kc, L.R, P, L.v.Wo, L.v.H, L.v.W, L.v.C,
                         L.v.f, L.v.s, dXin);
      LAUNCH_CHECK(ctx);
    }
  }
  // The main stream's part of this layer ends here (dX is out): it goes straight on to the layer below.
  if (ln.forked) HIP_TRY(ctx, hipEventRecord(ctx->ev_g[3], ctx->stream));
  DCGP_TRY(cond_backward_finish(bk, ln, L, A1, ld, Kc, dKuf, S, false, bk.last_layer && ln.forked));
  return layer_tail(bk, ln, L, S, gvs, Kc, true);
}

// Dense head backward: gpflow RBF(D, ARD=True) on the flattened features (--last-kernel rbf, conv_gp/models.py:160-168).
// The forward divides x and Z by the lengthscales while staging (in_scale) and runs the unit-lengthscale kernel
// (L.ls == 1), so every kernel adjoint below is the generic one on the scaled copies Xs = X o s, Zs = Z o s; the chain
// rule back through the scaling gives dZ, dX and the per-dimension lengthscale gradient (ard_finish_kernel).
int dense_head_backward(Bk& bk, LayerState& L, const double* Xin, int rows, int n_mod, const double* gm, const double* gv, double* dXin) {
  dcgp_ctx* ctx = bk.ctx;
  const int M = L.M, Mp = L.Mp, D = L.v.L;
  const long ld = col_ld(rows);
  if (L.v.P != 1 || L.kernel_type != 0) return ctx_fail(ctx, DCGP_ERR_ARG, "grad: ARD lengthscales need the single-patch head");
  DCGP_TRY(begin_layer(bk, L));
  auto itB = ctx->ws.find(bk.pfx + "Kzx");
  if (itB == ctx->ws.end()) return ctx_fail(ctx, DCGP_ERR_ARG, "grad: the forward pass left no K_zx for the head");
  const double* Kzx = (const double*)itB->second.first;
  double* A1 = bk.ws("A1h", (size_t)Mp * ld);
  double* dKzx = bk.ws("dKzx", (size_t)Mp * ld);
  double* S = bk.ws("S", (size_t)Mp * Mp);
  double* gkd = bk.ws("gkd", rows);
  double* cs = bk.ws("cs", rows);
  double* Zs = bk.ws("Zs", (size_t)M * D);
  double* Xs = bk.ws("Xs", (size_t)rows * D);
  double* dZs = bk.ws("dZs", (size_t)M * D);
  double* dXs = bk.ws("dXs", (size_t)rows * D);
  NEED(A1); NEED(dKzx); NEED(S); NEED(gkd); NEED(cs); NEED(Zs); NEED(Xs); NEED(dZs); NEED(dXs);
  if (Mp > M) HIP_TRY(ctx, hipMemsetAsync(A1 + (size_t)M * ld, 0, (size_t)(Mp - M) * ld * sizeof(double), ctx->stream));
  DCGP_TRY(gemm_gen(ctx, mk(L.g.Linv, Mp, 1, Kzx, ld, 1, A1, ld, M, rows, M)));
  Lanes ln = lanes_of(ctx);
  ln.forked = false; ln.chain = ln.tail = ln.main;   // (a few hundred columns, one patch: everything in line on the main stream)
  DCGP_TRY(cond_backward_main(bk, ln, L, A1, ld, rows, gm, gv, dKzx, gkd, false));
  DCGP_TRY(cond_backward_finish(bk, ln, L, A1, ld, rows, dKzx, S, bk.kl_early && !L.white, false));
  if (!bk.kl_early) DCGP_TRY(kl_products(bk, L, L.white ? nullptr : S, false));
  DCGP_TRY(kl_apply(bk, L, false));
  DCGP_TRY(add_scalar(bk, L, 0, gkd, rows, 1.0));             // Kdiag = variance
  hipLaunchKernelGGL(scale_rows_kernel, dim3(blocks_for((long)M * D)), dim3(256), 0, ctx->stream, L.Z, M, (long)M, D, L.in_scale, Zs);
  LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(scale_rows_kernel, dim3(blocks_for((long)rows * D)), dim3(256), 0, ctx->stream, Xin, n_mod, (long)rows, D, L.in_scale, Xs);
  LAUNCH_CHECK(ctx);
  HIP_TRY(ctx, hipMemsetAsync(dZs, 0, (size_t)M * D * sizeof(double), ctx->stream));
  DCGP_TRY(kuu_backward(bk, L, Zs, S, Mp, true, dZs));
  DCGP_TRY(e_form(bk, L, dKzx, ld, 1, nullptr, 1.0, Kzx, ld, dKzx, ld, rows, cs, nullptr));   // E over dKzx (P == 1: K_zx is the full response)
  DCGP_TRY(patch_backward(bk, L, dKzx, ld, rows, cs, Xs, dXs, 0, Zs, dZs));
  hipLaunchKernelGGL(ard_finish_kernel, dim3(D), dim3(256), 0, ctx->stream, dZs, Zs, M, dXs, Xs, (long)rows, D, L.in_scale, L.gZ, dXin, L.gard);
  LAUNCH_CHECK(ctx);
  DCGP_TRY(end_layer(bk, L));
  // the scalar "lengthscale" slot collected d / d(unit lengthscale): meaningless here, and L.ls must stay 1
  HIP_TRY(ctx, hipMemsetAsync(L.gscal + 1, 0, sizeof(double), ctx->stream));
  return DCGP_OK;
}

// SVGP head backward (ConvKernel / AdditivePatchKernel, conv_gp/kernels.py:15-136).  gm / gv [rows][R].
int head_backward(Bk& bk, LayerState& L, const double* Xin, int rows, int n_mod, const double* gm, const double* gv, double* dXin) {
  dcgp_ctx* ctx = bk.ctx;
  const int M = L.M, Mp = L.Mp, P = L.v.P, Ld = L.v.L;
  const long ld = col_ld(rows), Kc = (long)rows * P, ldf = col_ld(Kc);
  if (L.in_scale) return dense_head_backward(bk, L, Xin, rows, n_mod, gm, gv, dXin);
  DCGP_TRY(begin_layer(bk, L));
  auto itB = ctx->ws.find(bk.pfx + "Kzx");
  if (itB == ctx->ws.end()) return ctx_fail(ctx, DCGP_ERR_ARG, "grad: the forward pass left no K_zx for the head");
  const double* Kzx = (const double*)itB->second.first;
  double* A1 = bk.ws("A1h", (size_t)Mp * ld);
  double* dKzx = bk.ws("dKzx", (size_t)Mp * ld);
  double* S = bk.ws("S", (size_t)Mp * Mp);
  double* gkd = bk.ws("gkd", rows);
  double* Kfull = bk.ws("Kfull", (size_t)Mp * ldf);
  double* E = bk.ws("E", (size_t)Mp * ldf);
  double* cs = bk.ws("cs", Kc);
  double* raw = bk.ws("raw", Kc);
  double* Xcol = bk.ws("Xcol", (size_t)Kc * Ld);
  double* dXcol = bk.ws("dXcol", (size_t)Kc * Ld);
  NEED(A1); NEED(dKzx); NEED(S); NEED(gkd); NEED(Kfull); NEED(E); NEED(cs); NEED(raw); NEED(Xcol); NEED(dXcol);
  if (Mp > M) HIP_TRY(ctx, hipMemsetAsync(A1 + (size_t)M * ld, 0, (size_t)(Mp - M) * ld * sizeof(double), ctx->stream));   // padded rows: operands of gemm_tn
  // A1 = inv(L) Kzx: left behind by the forward pass's one-launch conditional in a training step (head_forward, keep_k), formed here otherwise
  if (L.a1h_ready) L.a1h_ready = false;
  else DCGP_TRY(gemm_gen(ctx, mk(L.g.Linv, Mp, 1, Kzx, ld, 1, A1, ld, M, rows, M)));
  const Lanes ln = lanes_of(ctx);
  // as in conv_backward: the conditional's column-wise adjoint and the patch-kernel adjoints (K_zx, K_diag) on the main stream, the M x M
  // chain and what needs S beside them
  DCGP_TRY(cond_backward_main(bk, ln, L, A1, ld, rows, gm, gv, dKzx, gkd, bk.last_layer && ln.forked));
  // every patch response, Kfull[m][n * P + p] = k(Z_m, x_np): kept by the forward pass's sweep where that was the unit sweep (head_forward,
  // keep_k), evaluated again otherwise
  if (L.kfull_ready) {
    L.kfull_ready = false;
  } else {
    PatchRbfArgs a;
    a.X = Xin; a.N = rows; a.n_mod = n_mod;
    a.H = L.v.H; a.W = L.v.W; a.C = L.v.C; a.f = L.v.f; a.s = L.v.s; a.Ho = L.v.Ho; a.Wo = L.v.Wo; a.P = P; a.L = Ld;
    a.ZT = L.ZT; a.zn = L.zn; a.M = M; a.Mp = Mp; a.Lp = L.Lp;
    a.bk = L.base();
    a.out = Kfull; a.sM = ldf; a.sN = P; a.sP = 1;
    DCGP_TRY(patch_rbf(ctx, a, "grad_head_kfull"));
  }
  DCGP_TRY(im2col(ctx, L, Xin, n_mod, Kc, Xcol));
  // Kzx[m][n] = 1/P sum_p w_p k(Z_m, x_np)
  DCGP_TRY(e_form(bk, L, dKzx, ld, P, L.w, 1.0 / P, Kfull, ldf, E, ldf, Kc, cs, raw));
  hipLaunchKernelGGL(strided_sum_kernel, dim3(P), dim3(256), 0, ctx->stream, raw, rows, P, 1.0 / P, 1, L.gw);
  LAUNCH_CHECK(ctx);
  DCGP_TRY(patch_backward(bk, L, E, ldf, Kc, cs, Xcol, dXin ? dXcol : nullptr, 0, nullptr, nullptr, nullptr, 0.0, &ln));
  // Kdiag
  if (L.kernel_type == 0) {
    const double inv_l2 = 1.0 / (L.ls * L.ls);
    double* Gm = bk.ws("kd_G", (size_t)rows * P * P);
    double* norms = bk.ws("kd_norms", (size_t)rows * P);
    double* dwn = bk.ws("kd_dwn", (size_t)rows * P);
    double* pv = bk.ws("kd_pv", (size_t)rows * P);
    double* pl = bk.ws("kd_pl", (size_t)rows * P);
    NEED(Gm); NEED(norms); NEED(dwn); NEED(pv); NEED(pl);
    GenGemm gg = mk(Xcol, Ld, 1, Xcol, 1, Ld, Gm, P, P, P, Ld);     // per image: X_n X_n^T
    gg.batch = rows; gg.a_bs = (long)P * Ld; gg.b_bs = (long)P * Ld; gg.c_bs = (long)P * P;
    DCGP_TRY(gemm_gen(ctx, gg));
    hipLaunchKernelGGL(kdiag_norms_kernel, dim3(blocks_for((long)rows * P)), dim3(256), 0, ctx->stream, Gm, P, (long)rows, norms);
    LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(kdiag_backward_kernel, dim3((P + 3) / 4, rows), dim3(256), 0, ctx->stream, Gm, norms, gkd, L.w, P, L.variance, inv_l2, dwn, pv, pl);
    LAUNCH_CHECK(ctx);
    DCGP_TRY(add_scalars(bk, L, {{0, pv, (long)rows * P, 1.0 / L.variance}, {1, pl, (long)rows * P, inv_l2 / L.ls}}));
    hipLaunchKernelGGL(strided_sum_kernel, dim3(P), dim3(256), 0, ctx->stream, dwn, rows, P, 1.0, 1, L.gw);
    LAUNCH_CHECK(ctx);
    if (dXin) {
      // d x_p += 2 (E_n X_n - rowsum(E_n) o X_n)_p / l^2   (E symmetric), per image
      GenGemm ge = mk(Gm, P, 1, Xcol, Ld, 1, dXcol, Ld, P, Ld, P);
      ge.batch = rows; ge.a_bs = (long)P * P; ge.b_bs = (long)P * Ld; ge.c_bs = (long)P * Ld;
      ge.alpha = 2.0 * inv_l2; ge.accumulate = 1; ge.sub_v = pv; ge.sv_bs = P; ge.sub_x = Xcol; ge.sx_rs = Ld; ge.sx_bs = (long)P * Ld;
      DCGP_TRY(gemm_gen(ctx, ge));
    }
  } else {
    // AdditivePatchKernel.Kdiag = mean_p w_p variance (conv_gp/kernels.py:53-61)
    double* tmp = bk.ws("kd_tmp", 2);
    NEED(tmp);
    DCGP_TRY(reduce_sum(ctx, gkd, rows, 1.0, tmp));                  // sum_n gkd
    DCGP_TRY(reduce_sum(ctx, L.w, P, 1.0 / P, tmp + 1));             // mean(w)
    hipLaunchKernelGGL(additive_kdiag_backward_kernel, dim3(blocks_for(P)), dim3(256), 0, ctx->stream, tmp, P, L.variance, L.gw,
                       L.gslots + VAR_SLOT + bk.slot_v);
    LAUNCH_CHECK(ctx);
    ++bk.slot_v;
  }
  if (dXin) {
    const long n = (long)rows * L.v.H * L.v.W * L.v.C;
    hipLaunchKernelGGL(col2im_kernel, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, dXcol, rows, L.v.H, L.v.W, L.v.C, L.v.f, L.v.s, L.v.Ho,
                       L.v.Wo, Ld, dXin);
    LAUNCH_CHECK(ctx);
  }
  // the main stream's part of the head ends here (dX is out): it goes on to the layer below (see conv_backward)
  if (ln.forked) HIP_TRY(ctx, hipEventRecord(ctx->ev_g[3], ctx->stream));
  DCGP_TRY(cond_backward_finish(bk, ln, L, A1, ld, rows, dKzx, S, bk.kl_early && !L.white, bk.last_layer && ln.forked));
  return layer_tail(bk, ln, L, S, nullptr, 0, false);
}

}  // namespace

static double kl_weight(const dcgp_model* m) {
  // the data term is summed over the batch shards (ranks); the KL term is replicated, so each shard carries 1 / shards of it
  const int shards = m->grad_shards > 0 ? m->grad_shards : (m->ctx->comm ? m->ctx->nranks : 1);
  return 1.0 / shards;
}

// Called by the forward pass of a training step (forward_all, model.hip): the KL adjoint's products of every unwhitened layer go to the
// side stream, beside the layers of the forward pass (see model_state.h for the two calls).
int grad_kl_early(dcgp_model* m, bool enqueue, bool wait_fork) {
  dcgp_ctx* ctx = m->ctx;
  const int nl = (int)m->layers.size();
  for (bool& f : m->kl_early) f = false;
  for (int& f : m->prep_early) f = 0;
  const bool side = !ctx->opt.grad_nofork && !ctx->no_side && ctx->stream2 && ctx->stream_aux;
  if (ctx->opt.grad_late_kl || !side || nl > 8) return DCGP_OK;
  if (!enqueue) return 1;
  Bk bk;
  bk.m = m; bk.ctx = ctx; bk.klw = kl_weight(m);
  hipStream_t saved = ctx->stream;
  // a stream of its own: on the side stream these ~40 launches (slow beside the forward pass's kernels) were still in front of the head's side
  // chain when the reverse pass got there
  hipStream_t es = ctx->stream_aux;
  if (wait_fork) {
    HIP_TRY(ctx, hipStreamWaitEvent(es, ctx->ev_fork, 0));
    // (the mark may sit on the main stream behind layer 0 while G / alpha of the later layers are still being written on the chain's stream)
    for (int li = 0; li < m->gkl_prep_wait && li < nl; ++li) HIP_TRY(ctx, hipStreamWaitEvent(es, m->ev_prep[m->bank][li], 0));
  }
  ctx->stream = es;
  const std::string mp = "m" + std::to_string(m->id) + "_";
  int rc = DCGP_OK;
  // first what the main stream's part of the reverse pass reads: zero fills and the parameter-only operands of every layer's conditional
  // (model_backward waits for ctx->ev_kl2 -- long past by then), the last layer first
  for (int li = nl - 1; li >= 0 && rc == DCGP_OK; --li) {
    LayerState& L = *m->layers[li];
    bk.pfx = mp + std::to_string(li) + "_";
    rc = layer_fills(bk, L);
    if (rc == DCGP_OK) rc = param_operands(bk, L, false);
    if (rc == DCGP_OK) m->prep_early[li] = 1;
    ConvBwdArgs fb;
    fb.M = L.M; fb.Mp = L.Mp; fb.R = L.R;
    if (rc == DCGP_OK && !L.is_head && L.has_qsqrt && !L.white && conv_bwd_fused_ok(ctx, fb)) {
      rc = param_operands(bk, L, true);
      if (rc == DCGP_OK) m->prep_early[li] |= 2;
    }
  }
  if (rc == DCGP_OK && hipEventRecord(ctx->ev_kl2, es) != hipSuccess) rc = DCGP_ERR_HIP;
  for (int li = 0; li < nl && rc == DCGP_OK; ++li) {
    LayerState& L = *m->layers[li];
    if (L.white) continue;
    bk.pfx = mp + std::to_string(li) + "_";
    double* Sacc = nullptr;
    if (L.is_head) {   // the head's prior shares the live Z: its part of d ELBO / dKuu opens the buffer cond_backward adds to
      Sacc = bk.ws("S", (size_t)L.Mp * L.Mp);
      if (!Sacc) { rc = DCGP_ERR_ALLOC; break; }
    }
    rc = kl_products(bk, L, Sacc, true);
    if (rc == DCGP_OK) m->kl_early[li] = true;
  }
  if (rc == DCGP_OK && hipEventRecord(ctx->ev_kl3, es) != hipSuccess) rc = DCGP_ERR_HIP;
  ctx->stream = saved;
  if (rc != DCGP_OK) { hipStreamSynchronize(es); for (bool& f : m->kl_early) f = false; for (int& f : m->prep_early) f = 0; }
  return rc;
}

int model_backward(dcgp_model* m, const double* X, const int32_t* y, int N, double scale, int dedup_layer0) {
  dcgp_ctx* ctx = m->ctx;
  const int nl = (int)m->layers.size(), S = m->S;
  if (!m->keep_outputs) return ctx_fail(ctx, DCGP_ERR_ARG, "grad: the forward pass must keep the layer outputs");
  for (auto& l : m->layers) {
    if (l->base_type != 0 && l->is_head) return ctx_fail(ctx, DCGP_ERR_ARG, "grad: the head kernels are RBF-based");
  }
  const double* gh = gauss_hermite_table(ctx);
  if (!gh) return DCGP_ERR_ALLOC;
  bool prepped = false;
  for (int f : m->prep_early) prepped = prepped || f != 0;
  if (prepped) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_kl2, 0));   // fills and parameter-only operands from the side stream (grad_kl_early)
  Bk bk;
  bk.m = m; bk.ctx = ctx;
  bk.klw = kl_weight(m);
  const std::string mp = "m" + std::to_string(m->id) + "_";
  LayerState& H = *m->layers[nl - 1];
  auto& oh = m->outs[nl - 1];
  const int rows = oh.rows;
  const double weight = scale * ((rows == S * N) ? 1.0 / S : 1.0);
  double* gm = (double*)ws_get(ctx, mp + "g_gm_head", (size_t)rows * H.R * sizeof(double));
  double* gv = (double*)ws_get(ctx, mp + "g_gv_head", (size_t)rows * H.R * sizeof(double));
  NEED(gm); NEED(gv);
  if (H.R > RM_KMAX) return ctx_fail(ctx, DCGP_ERR_ARG, "grad: at most %d classes", RM_KMAX);
  hipLaunchKernelGGL(robustmax_grad_kernel, dim3((rows + RM_ROWS - 1) / RM_ROWS), dim3(256), 0, ctx->stream, oh.mean, oh.var, y, rows, N, H.R,
                     m->eps, gh, weight, gm, gv);
  LAUNCH_CHECK(ctx);
  bool dedup_done = false;   // layer 0's (dmean, dvar) already summed over the S replicas
  for (int li = nl - 1; li >= 0; --li) {
    LayerState& L = *m->layers[li];
    bk.pfx = mp + std::to_string(li) + "_";
    bk.last_layer = li == 0;
    bk.kl_early = li < 8 && m->kl_early[li];
    bk.prep = li < 8 ? m->prep_early[li] : 0;
    if (li < 8) { m->kl_early[li] = false; m->prep_early[li] = 0; }
    const double* Xin = li == 0 ? X : m->outs[li - 1].sample;
    int rows_l = m->outs[li].rows;                  // rows entering == rows leaving ...
    // ... except for a de-duplicated first conv layer: propagate() tiles the batch S times, so layer 0 saw S identical
    // copies; the forward evaluated its conditional on the N distinct images and drew S samples from it.  The S
    // gradients arriving per image add up, and the conditional's reverse pass runs on N rows instead of S N (exact).
    if (li == 0 && dedup_layer0 && !L.is_head && rows_l == S * N && S > 1) {
      if (!dedup_done) {   // (the head-less case: the gradients came from the likelihood, not from a layer above)
        const long n = (long)N * L.v.P * L.R;
        double* gm0 = (double*)ws_get(ctx, bk.pfx + "g_gm_dedup", (size_t)n * sizeof(double));
        double* gv0 = (double*)ws_get(ctx, bk.pfx + "g_gv_dedup", (size_t)n * sizeof(double));
        NEED(gm0); NEED(gv0);
        hipLaunchKernelGGL(reduce_replicas_kernel, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, gm, gv, S, n, gm0, gv0);
        LAUNCH_CHECK(ctx);
        gm = gm0; gv = gv0;
      }
      rows_l = N;
    }
    const int n_mod = li == 0 ? N : rows_l;
    double* dXin = nullptr;
    if (li > 0) {
      dXin = (double*)ws_get(ctx, bk.pfx + "g_dXin", (size_t)rows_l * L.v.H * L.v.W * L.v.C * sizeof(double));
      NEED(dXin);
    }
    if (L.is_head) DCGP_TRY(head_backward(bk, L, Xin, rows_l, n_mod, gm, gv, dXin));
    else DCGP_TRY(conv_backward(bk, L, Xin, rows_l, n_mod, gm, gv, dXin));
    if (li > 0) {
      auto& o = m->outs[li - 1];
      const long n = (long)o.rows * o.width;
      gm = (double*)ws_get(ctx, mp + std::to_string(li - 1) + "_g_gm", (size_t)n * sizeof(double));
      gv = (double*)ws_get(ctx, mp + std::to_string(li - 1) + "_g_gv", (size_t)n * sizeof(double));
      NEED(gm); NEED(gv);
      LayerState& Lb = *m->layers[li - 1];
      if (li - 1 == 0 && dedup_layer0 && !Lb.is_head && o.rows == S * N && S > 1) {   // S gradients per element of the shared conditional: summed here
        const long n0 = (long)N * o.width;
        hipLaunchKernelGGL(sample_backward_dedup_kernel, dim3(blocks_for(n0)), dim3(256), 0, ctx->stream, dXin, o.sample, o.mean, o.var, m->jitter, S, n0,
                           gm, gv);
        dedup_done = true;
      } else {
        hipLaunchKernelGGL(sample_backward_kernel, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, dXin, o.sample, o.mean, o.var, m->jitter, n, gm, gv);
      }
      LAUNCH_CHECK(ctx);
    }
  }
  if (bk.side_pending) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_kl, 0));   // the one join of the step: every layer's side-stream tail
  m->grad_scattered = false;
  if (ctx->comm) {
    // One in-stream collective per layer over its contiguous gradient block (RCCL over xGMI).  A training step in exchange mode 1: a
    // reduce-scatter -- this rank's shard of the block summed over the ranks, 1 / ranks of an all-reduce's receive volume -- followed (opt_enqueue)
    // by the update of that shard alone and an all-gather of the parameters; otherwise the all-reduce every rank follows with the full update.
    const bool scatter = m->grad_exchange == 1 && m->adam_follows && ctx->nranks > 1;
    for (auto& l : m->layers) {
      const size_t n = l->grad_block_count();
      if (n > 0x7fffffffUL) return ctx_fail(ctx, DCGP_ERR_ARG, "grad: gradient block too large for one collective");
      if (scatter) {
        long lo, hi, sh;
        DCGP_TRY(dcgp_shard_range((long)n, ctx->nranks, ctx->rank, &lo, &hi, &sh));
        if ((size_t)sh * ctx->nranks > n + LayerState::kGradBlockPad) return ctx_fail(ctx, DCGP_ERR_ARG, "grad: more ranks than the gradient block is padded for");
        DCGP_TRY(reduce_scatter_sum_f64_async(ctx, l->gZ, (size_t)sh));
      } else {
        DCGP_TRY(allreduce_sum_f64_async(ctx, l->gZ, (int)n));
      }
    }
    m->grad_scattered = scatter;
  }
  return DCGP_OK;
}

extern "C" {

}  // extern "C"

// Adam step enqueued behind the reverse pass (dcgp_model_train_step_adam): lr is the bias-corrected rate
struct AdamReq { double lr_t, beta1, beta2, eps; };
static int opt_enqueue(dcgp_model* model, const char* who, bool sgd, double lr, double beta1, double beta2, double eps, const double* status, int vranks);
static int opt_readback(dcgp_model* model);

static int elbo_grad_run(dcgp_model* model, const double* X, const int32_t* y, int N, double scale, const double* const* z_per_layer_host,
                         uint64_t seed, int dedup_layer0, double* out_host, int* info_host, const AdamReq* adam) {
  dcgp_ctx* ctx = model->ctx;
  const bool keep = model->keep_outputs;
  model->keep_outputs = true;   // the reverse pass reads every layer's (sample, mean, var)
  model->keep_state = true;     // ... and K_uf / A1 of every conv layer
  model->adam_follows = adam != nullptr;
  model->grad_follows = true;   // ... and marks where the parameter-only part of the reverse pass may start (grad_kl_early)
  // The forward pass is ENQUEUED, the reverse pass behind it, and only then is the forward's result collected: the host does not wait for
  // the ELBO before it feeds the ~110 launches of the reverse pass (a failed factorisation is reported all the same; the reverse pass then
  // ran on NaNs, which nothing reads).
  uint64_t ticket = 0;
  if (info_host) *info_host = 0;
  const auto host_t0 = std::chrono::steady_clock::now();
  int rc = model->enq_seq != model->col_seq ? ctx_fail(ctx, DCGP_ERR_ARG, "elbo_grad: enqueued steps are still to be collected")
                                             : elbo_forward_enqueue_impl(model, X, y, N, scale, z_per_layer_host, seed, dedup_layer0, &ticket);
  const bool enqueued = rc == DCGP_OK;
  if (rc == DCGP_OK && model->gkl_state) rc = grad_kl_early(model, true, model->gkl_state == 2);
  if (rc == DCGP_OK) rc = model_backward(model, X, y, N, scale, dedup_layer0);
  // (the update reads the step's factorisation status word on the device: a failed step leaves the parameters as they were)
  if (rc == DCGP_OK && adam)
    rc = opt_enqueue(model, "train_step_adam", false, adam->lr_t, adam->beta1, adam->beta2, adam->eps, model->d_scal + 64 * model->bank + 43, 0);
  if (ctx->timing) {   // host time to enqueue the whole step, forward pass included (reported beside the kernel timers)
    auto& acc = ctx->tim["grad_host_enqueue"];
    acc.launches += 1;
    acc.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
  }
  if (enqueued) {
    const int rc_f = elbo_forward_collect_impl(model, ticket, out_host, info_host);
    if (rc == DCGP_OK) rc = rc_f;
  }
  model->keep_outputs = keep;
  model->keep_state = false;
  model->grad_follows = false;
  model->gkl_state = 0;
  if (rc != DCGP_OK) {
    for (bool& f : model->kl_early) f = false;
    for (int& f : model->prep_early) f = 0;
    hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) hipStreamSynchronize(ctx->stream2);
    if (ctx->stream_aux) hipStreamSynchronize(ctx->stream_aux);
  }
  DCGP_TRY(rc);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (adam) DCGP_TRY(opt_readback(model));
  return DCGP_OK;
}

extern "C" {

int dcgp_elbo_grad(dcgp_model* model, const double* X, const int32_t* y, int N, double scale, const double* const* z_per_layer_host,
                   uint64_t seed, int dedup_layer0, double* out_host, int* info_host) {
  if (!model || !X || !y || N <= 0 || !out_host) return model ? ctx_fail(model->ctx, DCGP_ERR_ARG, "elbo_grad: bad args") : DCGP_ERR_ARG;
  return elbo_grad_run(model, X, y, N, scale, z_per_layer_host, seed, dedup_layer0, out_host, info_host, nullptr);
}

// One training step in one call: value, gradient and the Adam update (dcgp_model_adam_step's arguments; t == 0: the model's own step
// count), enqueued back to back with one wait at the end -- what session.run(minimise_op) is to the reference (conv_gp/experiment.py:84-108).
// A step whose factorisation fails returns DCGP_ERR_NOT_PD and leaves parameters, moments and step count untouched.
int dcgp_model_train_step_adam(dcgp_model* model, const double* X, const int32_t* y, int N, double scale, const double* const* z_per_layer_host,
                               uint64_t seed, int dedup_layer0, double lr, double beta1, double beta2, double eps, int t, double* out_host,
                               int* info_host) {
  if (!model || !X || !y || N <= 0 || !out_host) return model ? ctx_fail(model->ctx, DCGP_ERR_ARG, "train_step_adam: bad args") : DCGP_ERR_ARG;
  if (t < 0 || !(lr > 0) || !(beta1 >= 0 && beta1 < 1) || !(beta2 >= 0 && beta2 < 1) || !(eps > 0))
    return ctx_fail(model->ctx, DCGP_ERR_ARG, "train_step_adam: bad optimiser arguments");
  const int t_use = t == 0 ? model->adam_t + 1 : t;
  AdamReq a;
  a.lr_t = lr * sqrt(1.0 - pow(beta2, (double)t_use)) / (1.0 - pow(beta1, (double)t_use));
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  for (auto& l : model->layers) DCGP_TRY(l->ensure_grads());   // (the optimiser's group table is built before the first reverse pass has run)
  DCGP_TRY(elbo_grad_run(model, X, y, N, scale, z_per_layer_host, seed, dedup_layer0, out_host, info_host, &a));
  model->adam_t = t_use;
  return DCGP_OK;
}

int dcgp_model_get_grad(dcgp_model* model, int layer, const char* which, double* out_host, size_t count) {
  if (!model || !which || !out_host) return DCGP_ERR_ARG;
  dcgp_ctx* ctx = model->ctx;
  if (layer < 0 || layer >= (int)model->layers.size()) return ctx_fail(ctx, DCGP_ERR_ARG, "get_grad: no layer %d", layer);
  LayerState& L = *model->layers[layer];
  if (!L.gZ) return ctx_fail(ctx, DCGP_ERR_ARG, "get_grad: call dcgp_elbo_grad first");
  const double* src = nullptr;
  size_t n = 0;
  if (!strcmp(which, "Z")) { src = L.gZ; n = (size_t)L.M * L.v.L; }
  else if (!strcmp(which, "q_mu")) { src = L.gq_mu; n = (size_t)L.M * L.R; }
  else if (!strcmp(which, "q_sqrt")) { src = L.gq_sqrt; n = (size_t)L.R * L.M * L.M; }
  else if (!strcmp(which, "variance")) { src = L.gscal; n = 1; }
  else if (!strcmp(which, "lengthscale") || !strcmp(which, "weight_variances")) { src = L.gscal + 1; n = 1; }
  else if (!strcmp(which, "bias_variance")) { src = L.gscal + 2; n = 1; }
  else if (!strcmp(which, "w")) {
    if (!L.is_head) return ctx_fail(ctx, DCGP_ERR_ARG, "get_grad: only the head has patch weights");
    src = L.gw; n = (size_t)L.v.P;
  } else if (!strcmp(which, "ard_lengthscales")) {
    if (!L.ard || !L.gard) return ctx_fail(ctx, DCGP_ERR_ARG, "get_grad: this layer has no ARD lengthscales");
    src = L.gard; n = (size_t)L.v.L;
  } else return ctx_fail(ctx, DCGP_ERR_ARG, "get_grad: unknown parameter '%s'", which);
  if (count != n) return ctx_fail(ctx, DCGP_ERR_ARG, "get_grad(%s): expected %zu values, got %zu", which, n, count);
  HIP_TRY(ctx, hipMemcpyAsync(out_host, src, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}

}  // extern "C"

extern "C" {

int dcgp_model_set_grad_shards(dcgp_model* model, int shards) {
  if (!model || shards < 0) return DCGP_ERR_ARG;
  model->grad_shards = shards;
  return DCGP_OK;
}

int dcgp_model_grad_block(dcgp_model* model, int layer, double** block_dev, size_t* count) {
  if (!model || !block_dev || !count) return DCGP_ERR_ARG;
  if (layer < 0 || layer >= (int)model->layers.size()) return ctx_fail(model->ctx, DCGP_ERR_ARG, "grad_block: no layer %d", layer);
  LayerState& L = *model->layers[layer];
  DCGP_TRY(L.ensure_grads());
  *block_dev = L.gZ;
  *count = L.grad_block_count();
  return DCGP_OK;
}

// one optimiser step over every trainable group of the model (sgd: plain ascent, otherwise Adam with the bias-corrected rate lr)
static int opt_readback(dcgp_model* model);
static int opt_enqueue(dcgp_model* model, const char* who, bool sgd, double lr, double beta1, double beta2, double eps, const double* status, int vranks);
static int opt_step(dcgp_model* model, const char* who, bool sgd, double lr, double beta1, double beta2, double eps, int vranks = 0) {
  DCGP_TRY(opt_enqueue(model, who, sgd, lr, beta1, beta2, eps, nullptr, vranks));
  HIP_TRY(model->ctx, hipStreamSynchronize(model->ctx->stream));
  return opt_readback(model);
}
// the launch alone, behind whatever the stream holds (status: see OptArgs).
// vranks > 0 (debugging aid, dcgp_model_debug_sharded_adam): the sharded step of `vranks` ranks played on this one GPU from rank 0's point of
// view -- its own shard for real, the others' into the staging block only, then the unstage pass.
static int opt_enqueue(dcgp_model* model, const char* who, bool sgd, double lr, double beta1, double beta2, double eps, const double* status, int vranks = 0) {
  dcgp_ctx* ctx = model->ctx;
  const int nl = (int)model->layers.size();
  if (nl > 8) return ctx_fail(ctx, DCGP_ERR_ARG, "%s: at most 8 layers", who);
  const bool comm_sharded = ctx->comm && ctx->nranks > 1 && model->grad_exchange == 1 && model->grad_scattered;
  const bool sharded = comm_sharded || vranks > 0;
  ++model->param_version;   // the parameters change: parameter-only state of earlier steps is not reused (model_state.h)
  // Moments are rank-local in exchange mode 1: a rank holds m / v of its own shard only.  A full update on top of them (exchange mode 0, or an
  // optimiser step behind dcgp_elbo_grad) would move every element outside the shard with stale or zero moments -- differently on every rank.
  if (!sgd && !comm_sharded && vranks == 0 && model->sharded_steps > 0)
    return ctx_fail(ctx, DCGP_ERR_ARG, "%s: %llu sharded Adam steps were taken on this model (exchange mode 1): its moments cover this rank's shard only, "
                    "a full update would let the replicas diverge", who, (unsigned long long)model->sharded_steps);
  if (comm_sharded && !sgd) ++model->sharded_steps;
  const int nranks = vranks > 0 ? vranks : ctx->nranks, rank = vranks > 0 ? 0 : ctx->rank;
  OptArgs a{};
  a.sgd = sgd ? 1 : 0; a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.status = status;
  a.sharded = sharded ? 1 : 0;
  double* h = ctx->h_scratch;   // 64 pinned doubles: {variance, p1, p2} per layer
  if (hipHostGetDevicePointer((void**)&a.host_out, h, 0) != hipSuccess) return ctx_fail(ctx, DCGP_ERR_HIP, "%s: pinned slot not mapped", who);
  int nb = 0, cur_layer = 0;
  const double* blk0 = nullptr;
  auto add = [&](double* p, const double* g, double* const* mv, long n, int transform, int hyp_layer, double* recip, bool frozen) -> int {
    if (n <= 0 || (frozen && !sharded)) return DCGP_OK;
    if (a.ng >= OPT_GROUPS_MAX) return ctx_fail(ctx, DCGP_ERR_ARG, "%s: too many parameter groups", who);
    OptGroup& G = a.grp[a.ng];
    G.p = p; G.g = g; G.m = mv[0]; G.v = mv[1]; G.recip = recip; G.n = n; G.transform = transform; G.hyp_layer = hyp_layer;
    G.off = (long)(g - blk0); G.layer = cur_layer; G.frozen = frozen ? 1 : 0;
    a.first_block[a.ng++] = nb;
    nb += (int)blocks_for(n);
    return DCGP_OK;
  };
  long shard[8] = {};
  for (int li = 0; li < nl; ++li) {
    LayerState& L = *model->layers[li];
    if (!L.gZ) return ctx_fail(ctx, DCGP_ERR_ARG, "%s: call dcgp_elbo_grad first", who);
    DCGP_TRY(L.ensure_adam());   // (SGD: for the device copy of the hyper-parameters)
    cur_layer = li; blk0 = L.gZ;
    if (sharded) {
      DCGP_TRY(L.ensure_stage());
      if (nranks - 1 > (int)LayerState::kGradBlockPad) return ctx_fail(ctx, DCGP_ERR_ARG, "%s: more ranks than the gradient block is padded for", who);
      DCGP_TRY(dcgp_shard_range((long)L.grad_block_count(), nranks, rank, &a.sh_lo[li], &a.sh_hi[li], &shard[li]));
      a.stage[li] = L.pstage;
    }
    a.hyp_in[li][0] = L.variance; a.hyp_in[li][1] = L.base_type == 1 ? L.acos_w : L.ls; a.hyp_in[li][2] = L.base_type == 1 ? L.acos_b : 1.0;
    DCGP_TRY(add(L.Z, L.gZ, L.aZ, (long)L.M * L.v.L, 0, -1, nullptr, (L.frozen & 1u) != 0));
    DCGP_TRY(add(L.q_mu, L.gq_mu, L.aq_mu, (long)L.M * L.R, 0, -1, nullptr, (L.frozen & 2u) != 0));
    if (L.has_qsqrt) DCGP_TRY(add(L.q_sqrt, L.gq_sqrt, L.aq_sqrt, (long)L.R * L.M * L.M, 0, -1, nullptr, (L.frozen & 4u) != 0));   // upper triangle: zero gradient, zero step
    if (L.is_head && L.w) DCGP_TRY(add(L.w, L.gw, L.aw, (long)L.v.P, 0, -1, nullptr, (L.frozen & 8u) != 0));
    DCGP_TRY(add(L.hyp, L.gscal, L.ahyp, 3, 1, li, nullptr, (L.frozen & 16u) != 0));
    if (L.ard) DCGP_TRY(add(L.ard, L.gard, L.aard, (long)L.v.L, 1, -1, L.in_scale, (L.frozen & 16u) != 0));   // dense head: per-dimension lengthscales and the staging scale 1 / l
  }
  a.first_block[a.ng] = nb;
  if (nb <= 0) return DCGP_OK;
  hipLaunchKernelGGL(opt_step_kernel, dim3(nb), dim3(256), 0, ctx->stream, a);
  LAUNCH_CHECK(ctx);
  if (!sharded) return DCGP_OK;
  if (vranks > 0) {   // the other ranks' shards: into the staging block only
    for (int r = 1; r < vranks; ++r) {
      OptArgs b = a;
      b.stage_only = 1;
      for (int li = 0; li < nl; ++li) DCGP_TRY(dcgp_shard_range((long)model->layers[li]->grad_block_count(), vranks, r, &b.sh_lo[li], &b.sh_hi[li], nullptr));
      hipLaunchKernelGGL(opt_step_kernel, dim3(nb), dim3(256), 0, ctx->stream, b);
      LAUNCH_CHECK(ctx);
    }
  } else {
    for (int li = 0; li < nl; ++li) DCGP_TRY(all_gather_f64_async(ctx, model->layers[li]->pstage, (size_t)shard[li]));
  }
  hipLaunchKernelGGL(opt_unstage_kernel, dim3(nb), dim3(256), 0, ctx->stream, a);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}
// the updated kernel hyper-parameters back into the host-side layer state (the stream must have been synchronised)
static int opt_readback(dcgp_model* model) {
  const double* h = model->ctx->h_scratch;
  const int nl = (int)model->layers.size();
  for (int li = 0; li < nl; ++li) {
    LayerState& L = *model->layers[li];
    if (L.frozen & 16u) continue;
    L.variance = h[3 * li];
    if (L.base_type == 1) { L.acos_w = h[3 * li + 1]; L.acos_b = h[3 * li + 2]; } else L.ls = h[3 * li + 1];
  }
  return DCGP_OK;
}

int dcgp_model_adam_step(dcgp_model* model, double lr, double beta1, double beta2, double eps, int t) {
  if (!model || t < 0 || !(lr > 0) || !(beta1 >= 0 && beta1 < 1) || !(beta2 >= 0 && beta2 < 1) || !(eps > 0))
    return model ? ctx_fail(model->ctx, DCGP_ERR_ARG, "adam_step: bad arguments") : DCGP_ERR_ARG;
  // t == 0: the model's own count of Adam steps since its moment buffers were created (they start at zero with it) --
  // what tf.train.AdamOptimizer's beta powers do for a freshly built optimiser, whatever global_step a checkpoint carried
  if (t == 0) t = ++model->adam_t; else model->adam_t = t;
  const double lr_t = lr * sqrt(1.0 - pow(beta2, (double)t)) / (1.0 - pow(beta1, (double)t));
  return opt_step(model, "adam_step", false, lr_t, beta1, beta2, eps);
}

// Multi-rank training step: how the ranks exchange a step's gradient (0: all-reduce, every rank updates everything; 1: reduce-scatter, each rank
// updates its shard of every layer's parameter block, all-gather of the parameters -- SURVEY section 5).  Applies to dcgp_model_train_step_adam;
// dcgp_elbo_grad alone always all-reduces (its caller wants the whole gradient).
int dcgp_model_set_grad_exchange(dcgp_model* model, int mode) {
  if (!model || (mode != 0 && mode != 1)) return model ? ctx_fail(model->ctx, DCGP_ERR_ARG, "set_grad_exchange: mode 0 or 1") : DCGP_ERR_ARG;
  if (mode == 0 && model->grad_exchange == 1 && model->sharded_steps > 0)
    return ctx_fail(model->ctx, DCGP_ERR_ARG, "set_grad_exchange: %llu sharded Adam steps were taken: the moments are rank-local (each rank holds its own shard's), "
                    "the all-reduce route cannot continue from them", (unsigned long long)model->sharded_steps);
  model->grad_exchange = mode;
  return DCGP_OK;
}
// Debugging aid: one Adam step (dcgp_model_adam_step's arguments) taken the way `ranks` ranks would take it in exchange mode 1, played on this one
// GPU from rank 0's point of view -- shard 0 updated in place, the other shards through the staging block and the unstage pass.  The gradient must be
// the complete one (dcgp_elbo_grad).  Parameters and moments come out bit-identical to dcgp_model_adam_step's (tests/test_gpu_model.py).
int dcgp_model_debug_sharded_adam(dcgp_model* model, int ranks, double lr, double beta1, double beta2, double eps, int t) {
  if (!model || ranks < 1 || t < 0 || !(lr > 0) || !(beta1 >= 0 && beta1 < 1) || !(beta2 >= 0 && beta2 < 1) || !(eps > 0))
    return model ? ctx_fail(model->ctx, DCGP_ERR_ARG, "debug_sharded_adam: bad arguments") : DCGP_ERR_ARG;
  if (t == 0) t = ++model->adam_t; else model->adam_t = t;
  const double lr_t = lr * sqrt(1.0 - pow(beta2, (double)t)) / (1.0 - pow(beta1, (double)t));
  return opt_step(model, "debug_sharded_adam", false, lr_t, beta1, beta2, eps, ranks);
}

int dcgp_model_sgd_step(dcgp_model* model, double lr) {
  if (!model || !(lr > 0)) return model ? ctx_fail(model->ctx, DCGP_ERR_ARG, "sgd_step: bad arguments") : DCGP_ERR_ARG;
  return opt_step(model, "sgd_step", true, lr, 0.0, 0.0, 0.0);
}

int dcgp_model_set_trainable(dcgp_model* model, int layer, const char* which, int on) {
  if (!model || !which) return DCGP_ERR_ARG;
  if (layer < 0 || layer >= (int)model->layers.size()) return ctx_fail(model->ctx, DCGP_ERR_ARG, "set_trainable: no layer %d", layer);
  LayerState& L = *model->layers[layer];
  unsigned bit = 0;
  if (!strcmp(which, "Z")) bit = 1u;
  else if (!strcmp(which, "q_mu")) bit = 2u;
  else if (!strcmp(which, "q_sqrt")) bit = 4u;
  else if (!strcmp(which, "w")) bit = 8u;
  else if (!strcmp(which, "variance") || !strcmp(which, "lengthscale") || !strcmp(which, "hyper")) bit = 16u;
  else return ctx_fail(model->ctx, DCGP_ERR_ARG, "set_trainable: unknown parameter '%s'", which);
  if (on) L.frozen &= ~bit; else L.frozen |= bit;
  return DCGP_OK;
}

int dcgp_model_get_param(dcgp_model* model, int layer, const char* which, double* out_host, size_t count) {
  if (!model || !which || !out_host) return DCGP_ERR_ARG;
  dcgp_ctx* ctx = model->ctx;
  if (layer < 0 || layer >= (int)model->layers.size()) return ctx_fail(ctx, DCGP_ERR_ARG, "get_param: no layer %d", layer);
  LayerState& L = *model->layers[layer];
  const double* src = nullptr;
  size_t n = 0;
  if (!strcmp(which, "variance") || !strcmp(which, "lengthscale") || !strcmp(which, "weight_variances") || !strcmp(which, "bias_variance")) {
    if (count != 1) return ctx_fail(ctx, DCGP_ERR_ARG, "get_param(%s): expected 1 value", which);
    out_host[0] = which[0] == 'v' ? L.variance : (which[0] == 'l' ? L.ls : (which[0] == 'w' ? L.acos_w : L.acos_b));
    return DCGP_OK;
  }
  if (!strcmp(which, "Z")) { src = L.Z; n = (size_t)L.M * L.v.L; }
  else if (!strcmp(which, "q_mu")) { src = L.q_mu; n = (size_t)L.M * L.R; }
  else if (!strcmp(which, "q_sqrt")) { src = L.q_sqrt; n = (size_t)L.R * L.M * L.M; }
  else if (!strcmp(which, "w")) {
    if (!L.w) return ctx_fail(ctx, DCGP_ERR_ARG, "get_param: only the head has patch weights");
    src = L.w; n = (size_t)L.v.P;
  } else if (!strcmp(which, "ard_lengthscales")) {
    if (!L.ard) return ctx_fail(ctx, DCGP_ERR_ARG, "get_param: this layer has no ARD lengthscales");
    src = L.ard; n = (size_t)L.v.L;
  } else return ctx_fail(ctx, DCGP_ERR_ARG, "get_param: unknown parameter '%s'", which);
  if (count != n) return ctx_fail(ctx, DCGP_ERR_ARG, "get_param(%s): expected %zu values, got %zu", which, n, count);
  HIP_TRY(ctx, hipMemcpyAsync(out_host, src, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}

}  // extern "C"
