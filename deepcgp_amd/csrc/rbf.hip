// rbf.hip -- patch view + RBF kernel evaluations.
//
//   * rbf_gram_padded      : Kuu = RBF.K(Z) + jitter I                (conv_gp/layers.py:18-21)
//   * patch_rbf            : FullView.extract_patches_PNL + Kuf       (conv_gp/views.py:40-44, layers.py:23-32)
//                            and ConvKernel.Kzx                       (conv_gp/kernels.py:117-133)
//   * head_kdiag           : ConvKernel.Kdiag                         (conv_gp/kernels.py:106-115)
//   * extract_patches      : FullView.extract_patches(_PNL)           (conv_gp/views.py:32-54)
//
// The patch sweep never materialises patches: one workgroup stages ONE image (<= 32x32x3 or
// 15x15x10 doubles) in LDS and gathers MFMA B-operands straight from it with
// addr = patch_base[p] + k_offset[l]  (p = oh*W'+ow, l = (kh*f+kw)*C+c).  The cross term z.x runs on
// v_mfma_f64_16x16x4_f64; |x_p|^2 is accumulated from the same operand registers; the epilogue applies
// exp(-0.5 (|z|^2+|x|^2-2 z.x)/l^2) in fp64 and writes 16-double contiguous row segments.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Kuu
// ---------------------------------------------------------------------------------------------
// 32x32 output tile per 256-thread block; Z rows staged through LDS in chunks of 32 columns.
__global__ __launch_bounds__(256) void rbf_gram_kernel(const double* __restrict__ Z, int M, int L, BaseKernel bk,
                                                       double jitter, double* __restrict__ out, int ld,
                                                       int Mp) {
  __shared__ double Zi[32][33], Zj[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // ty in 0..7
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  double dot[4] = {0, 0, 0, 0}, ni[4] = {0, 0, 0, 0}, nj = 0.0;
  for (int l0 = 0; l0 < L; l0 += 32) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * 32; idx += 256) {
      int r = idx >> 5, c = idx & 31;
      Zi[r][c] = (i0 + r < M && l0 + c < L) ? Z[(long)(i0 + r) * L + l0 + c] : 0.0;
      Zj[r][c] = (j0 + r < M && l0 + c < L) ? Z[(long)(j0 + r) * L + l0 + c] : 0.0;
    }
    __syncthreads();
#pragma unroll 8
    for (int l = 0; l < 32; ++l) {
      const double b = Zj[tx][l];
      nj = fma(b, b, nj);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double a = Zi[ty + 8 * q][l];
        dot[q] = fma(a, b, dot[q]);
        ni[q] = fma(a, a, ni[q]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = i0 + ty + 8 * q, j = j0 + tx;
    if (i >= Mp || j >= ld) continue;
    double v = 0.0;
    if (i < M && j < M) {
      v = bk.eval(dot[q], ni[q], nj);
      if (i == j) v += jitter;
    } else if (i == j) {
      v = 1.0;   // identity on the padding so that factorisations of the padded matrix stay valid
    }
    out[(long)i * ld + j] = v;
  }
}

// ZT[l][m] = Z[m][l] (zero padded to [Lp][Mp]) and zn[m] = |Z[m]|^2; one block per 32 rows of Z
__global__ __launch_bounds__(256) void z_transpose_kernel(const double* __restrict__ Z, int M, int L, double* __restrict__ ZT, int Mp,
                                                          int Lp, double* __restrict__ zn) {
  __shared__ double t[32][33];
  __shared__ double nrm[8][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int m0 = blockIdx.x * 32;
  double acc = 0.0;   // partial |z|^2 of row (m0 + tx) over the l's this thread row visits
  for (int l0 = 0; l0 < Lp; l0 += 32) {
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      int m = m0 + r, l = l0 + tx;
      t[r][tx] = (m < M && l < L) ? Z[(long)m * L + l] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      int l = l0 + r, m = m0 + tx;
      double v = t[tx][r];
      if (l < Lp && m < Mp) ZT[(long)l * Mp + m] = v;
      acc = fma(v, v, acc);
    }
  }
  nrm[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && m0 + tx < Mp) {
    double s2 = 0.0;
    for (int q = 0; q < 8; ++q) s2 += nrm[q][tx];
    zn[m0 + tx] = s2;
  }
}

// ---------------------------------------------------------------------------------------------
// patch sweep
// ---------------------------------------------------------------------------------------------
constexpr int PR_BM = 64;    // inducing patches per workgroup
constexpr int PR_BP = 64;    // image patches per tile
constexpr int PR_BK = 32;    // k chunk of ZT staged in LDS
constexpr int PR_LDZ = PR_BM + 16;

__device__ __forceinline__ int patch_base(int p, int P, int Wo, int s, int W, int C) {
  if (p >= P) p = 0;
  int oh = p / Wo, ow = p - oh * Wo;
  return (oh * s * W + ow * s) * C;
}

// grid: (p tiles [write mode] or 1 [reduce mode], Mp/64, N); block 256 = 4 waves as 2 (m) x 2 (p)
template <int BT>
__global__ __launch_bounds__(256, 4) void patch_rbf_kernel(PatchRbfArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int HWC = a.H * a.W * a.C;
  double* img = smem;                                   // [HWC] (+pad to even)
  double* zt = img + ((HWC + 1) & ~1);                  // [PR_BK][PR_LDZ]
  double* red = zt + PR_BK * PR_LDZ;                    // [2][PR_BM] (reduce mode)
  int* koff = reinterpret_cast<int*>(red + 2 * PR_BM);  // [Lp]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = lane >> 4, lcol = lane & 15;
  const int n = blockIdx.z, m0 = blockIdx.y * PR_BM;

  const double* __restrict__ Xn = a.X + (long)(n % a.n_mod) * HWC;
  // image -> LDS in batches of 8 loads per thread: a rolled loop waits one memory latency per iteration
  for (int i0 = 0; i0 < HWC; i0 += 8 * 256) {
    double t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = i0 + e * 256 + tid;
      t[e] = (i < HWC) ? Xn[i] : 0.0;
      if (a.in_scale && i < HWC) t[e] *= a.in_scale[i];   // ARD: x / lengthscales (gpflow RBF(ARD=True), conv_gp/models.py:160-168)
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = i0 + e * 256 + tid;
      if (i < HWC) img[i] = t[e];
    }
  }
  for (int l = tid; l < a.Lp; l += 256) {
    int ll = l < a.L ? l : 0;
    int c = ll % a.C, t = ll / a.C;
    int kw = t % a.f, kh = t / a.f;
    koff[l] = (kh * a.W + kw) * a.C + c;
  }
  __syncthreads();

  const int p_tiles = (a.P + PR_BP - 1) / PR_BP;
  const int pt_lo = a.reduce ? 0 : blockIdx.x, pt_hi = a.reduce ? p_tiles : blockIdx.x + 1;

  double rsum[2][4];   // reduce mode: per (fm, v) running row sums
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int v = 0; v < 4; ++v) rsum[x][v] = 0.0;

  const bool one_chunk = a.Lp <= PR_BK;
  for (int pt = pt_lo; pt < pt_hi; ++pt) {
    const int p0 = pt * PR_BP + wn * 32;
    int pb[2];
#pragma unroll
    for (int y = 0; y < 2; ++y) pb[y] = patch_base(p0 + y * 16 + lcol, a.P, a.Wo, a.s, a.W, a.C);

    d4 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};
    double xn[2] = {0.0, 0.0};

    for (int k0 = 0; k0 < a.Lp; k0 += PR_BK) {
      // stage ZT[k0 .. k0+BK) x [m0 .. m0+64): 32 rows x 32 double2 chunks = 1024 chunks / 256 threads.  A patch
      // length of one chunk (L <= 32: every first layer) is staged once for all the patch tiles of the workgroup --
      // the reduce mode walks all of them, and re-staging cost two barriers and an L2 round trip per tile.
      if (!(one_chunk && pt > pt_lo)) {
        __syncthreads();   // previous chunk fully consumed
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          int ch = tid + c * 256;
          int row = ch >> 5, col = (ch & 31) * 2;
          double2 v = double2{0.0, 0.0};
          if (k0 + row < a.Lp && m0 + col < a.Mp) v = *reinterpret_cast<const double2*>(a.ZT + (long)(k0 + row) * a.Mp + m0 + col);
          *reinterpret_cast<double2*>(zt + row * PR_LDZ + col) = v;
        }
        __syncthreads();
      }
      const int kmax = min(PR_BK, a.Lp - k0);
      for (int kk = 0; kk < kmax; kk += 4) {
        const int k = k0 + kk + lrow;
        const int ko = koff[k];
        const bool kin = k < a.L;
        double av[2], bv[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) av[x] = zt[(kk + lrow) * PR_LDZ + wm * 32 + x * 16 + lcol];
#pragma unroll
        for (int y = 0; y < 2; ++y) {
          double v = img[pb[y] + ko];
          bv[y] = kin ? v : 0.0;
          xn[y] += bv[y] * bv[y];
        }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[x], bv[y], acc[x][y], 0, 0, 0);
      }
    }
    // |x_p|^2 for column lcol of each fragment: combine the 4 k-groups
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      xn[y] += __shfl_xor(xn[y], 16);
      xn[y] += __shfl_xor(xn[y], 32);
    }
    // epilogue
#pragma unroll
    for (int x = 0; x < 2; ++x) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int m = m0 + wm * 32 + x * 16 + lrow + 4 * v;
        const double znm = (m < a.Mp) ? a.zn[m] : 0.0;
#pragma unroll
        for (int y = 0; y < 2; ++y) {
          const int p = p0 + y * 16 + lcol;
          const double kv = a.bk.template eval_as<BT>(acc[x][y][v], xn[y], znm);
          if (a.reduce) {
            if (p < a.P) rsum[x][v] += a.w[p] * kv;
          } else if (m < a.M && p < a.P) {
            a.out[(long)m * a.sM + (long)n * a.sN + (long)p * a.sP] = kv;
          }
        }
      }
    }
  }

  if (a.reduce) {
    // sum over the 16 lanes of a row group, then over the two p-waves
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        double s = rsum[x][v];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        if (lcol == 0) red[wn * PR_BM + wm * 32 + x * 16 + lrow + 4 * v] = s;
      }
    __syncthreads();
    if (tid < PR_BM) {
      int m = m0 + tid;
      if (m < a.M) a.out[(long)m * a.sM + (long)n * a.sN] = a.scale * (red[tid] + red[PR_BM + tid]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ConvKernel.Kdiag: per image sum_{p,p'} w_p w_p' k(x_p, x_p') / P^2, upper triangle of 64x64 patch
// tile pairs (symmetry: off-diagonal pairs count twice).  grid (pairs, N); partial[n][pair].
// ---------------------------------------------------------------------------------------------
template <int BT>
__global__ __launch_bounds__(256, 4) void head_kdiag_kernel(const double* __restrict__ X, int n_mod, int H, int W, int C, int f,
                                                          int s, int Ho, int Wo, int P, int L, BaseKernel bk,
                                                          const double* __restrict__ w,
                                                          double* __restrict__ partial, int n_pairs, int p_tiles) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int HWC = H * W * C;
  const int Lp = (L + 3) & ~3;
  double* img = smem;                             // [HWC]
  double* xn = img + ((HWC + 1) & ~1);            // [128] norms: rows tile then cols tile
  double* red = xn + 128;                         // [4]
  int* koff = reinterpret_cast<int*>(red + 4);    // [Lp]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = lane >> 4, lcol = lane & 15;
  const int n = blockIdx.y;
  // decode pair index -> (tr <= tc)
  int pair = blockIdx.x, tr = 0;
  while (pair >= p_tiles - tr) {
    pair -= p_tiles - tr;
    ++tr;
  }
  const int tc = tr + pair;

  const double* __restrict__ Xn = X + (long)(n % n_mod) * HWC;
  // image -> LDS in batches of 8 loads per thread: a rolled loop waits one memory latency per iteration
  for (int i0 = 0; i0 < HWC; i0 += 8 * 256) {
    double t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = i0 + e * 256 + tid;
      t[e] = (i < HWC) ? Xn[i] : 0.0;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = i0 + e * 256 + tid;
      if (i < HWC) img[i] = t[e];
    }
  }
  for (int l = tid; l < Lp; l += 256) {
    int ll = l < L ? l : 0;
    int c = ll % C, t = ll / C;
    int kw = t % f, kh = t / f;
    koff[l] = (kh * W + kw) * C + c;
  }
  __syncthreads();
  // patch norms: 2 threads per patch over the 128 patches (rows tile, cols tile)
  {
    int q = tid >> 1, half = tid & 1;
    int p = (q < 64 ? tr * 64 + q : tc * 64 + (q - 64));
    int pbq = patch_base(p, P, Wo, s, W, C);
    double sacc = 0.0;
    for (int l = half; l < L; l += 2) {
      double v = img[pbq + koff[l]];
      sacc += v * v;
    }
    sacc += __shfl_xor(sacc, 1);
    if (half == 0) xn[q] = sacc;
  }
  __syncthreads();

  int pa[2], pbc[2];
#pragma unroll
  for (int x = 0; x < 2; ++x) pa[x] = patch_base(tr * 64 + wm * 32 + x * 16 + lcol, P, Wo, s, W, C);
#pragma unroll
  for (int y = 0; y < 2; ++y) pbc[y] = patch_base(tc * 64 + wn * 32 + y * 16 + lcol, P, Wo, s, W, C);
  d4 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};
  for (int kk = 0; kk < Lp; kk += 4) {
    const int k = kk + lrow;
    const int ko = koff[k];
    const bool kin = k < L;
    double av[2], bv[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      double v = img[pa[x] + ko];
      av[x] = kin ? v : 0.0;
    }
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      double v = img[pbc[y] + ko];
      bv[y] = kin ? v : 0.0;
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
        acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[x], bv[y], acc[x][y], 0, 0, 0);
  }
  double sum = 0.0;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int ql = wm * 32 + x * 16 + lrow + 4 * v;
      const int p = tr * 64 + ql;
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        const int qc = wn * 32 + y * 16 + lcol;
        const int pc = tc * 64 + qc;
        if (p < P && pc < P) {
          sum += w[p] * w[pc] * bk.template eval_as<BT>(acc[x][y][v], xn[ql], xn[64 + qc]);
        }
      }
    }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  if (tid == 0) {
    double t = (red[0] + red[1]) + (red[2] + red[3]);
    partial[(long)n * n_pairs + blockIdx.x] = (tr == tc) ? t : 2.0 * t;
  }
}

__global__ void kdiag_reduce_kernel(const double* __restrict__ partial, int n_pairs, int N, double scale,
                                    double* __restrict__ out) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double s = 0.0;
  for (int i = 0; i < n_pairs; ++i) s += partial[(long)n * n_pairs + i];
  out[n] = s * scale;
}

__global__ void extract_patches_kernel(const double* __restrict__ X, int N, int H, int W, int C, int f, int s,
                                       int Ho, int Wo, double* __restrict__ out, int pnl) {
  const int P = Ho * Wo, L = f * f * C;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)N * P * L;
  if (idx >= total) return;
  int l = (int)(idx % L);
  long t = idx / L;
  int p = (int)(t % P), n = (int)(t / P);
  int c = l % C, kk = l / C, kw = kk % f, kh = kk / f;
  int oh = p / Wo, ow = p % Wo;
  double v = X[(((long)n * H + oh * s + kh) * W + ow * s + kw) * C + c];
  if (pnl)
    out[((long)p * N + n) * L + l] = v;
  else
    out[idx] = v;
}

}  // namespace

int rbf_gram_padded(dcgp_ctx* ctx, const double* Z, int M, int L, BaseKernel bk, double jitter,
                    double* out, int ld, int Mp) {
  dim3 grid((ld + 31) / 32, (Mp + 31) / 32);
  hipLaunchKernelGGL(rbf_gram_kernel, grid, dim3(256), 0, ctx->stream, Z, M, L, bk, jitter, out, ld, Mp);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

int z_transpose_norms(dcgp_ctx* ctx, const double* Z, int M, int L, double* ZT, int Mp, int Lp, double* zn) {
  hipLaunchKernelGGL(z_transpose_kernel, dim3((Mp + 31) / 32), dim3(256), 0, ctx->stream, Z, M, L, ZT, Mp, Lp, zn);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

int patch_rbf(dcgp_ctx* ctx, const PatchRbfArgs& a, const char* timer_name) {
  if (a.N <= 0) return DCGP_OK;
  if (a.n_mod <= 0) return ctx_fail(ctx, DCGP_ERR_ARG, "patch_rbf: n_mod must be positive");
  if (a.Mp % PR_BM != 0 && a.Mp % 16 != 0) return ctx_fail(ctx, DCGP_ERR_ARG, "patch_rbf: Mp must be a multiple of 16");
  const int HWC = a.H * a.W * a.C;
  size_t lds = (size_t)(((HWC + 1) & ~1) + PR_BK * PR_LDZ + 2 * PR_BM) * sizeof(double) + (size_t)a.Lp * sizeof(int);
  if (lds > 160 * 1024) return ctx_fail(ctx, DCGP_ERR_ARG, "patch_rbf: image of %d doubles does not fit LDS", HWC);
  // A sweep that overlaps the factorisation chain would otherwise starve it: its thousands of short 128-VGPR
  // workgroups refill every slot the moment it frees, and a chain workgroup (200 VGPRs, 50 KB LDS) never finds a
  // whole CU's worth of room -- even from a high-priority stream the first panel waited for the entire sweep
  // (56 us instead of 18).  Claiming 54 KB of LDS per workgroup caps the sweep at two per CU (half the register
  // file, 108 KB LDS) and leaves a standing slot for the chain; the sweep gets slower (54 -> 76 us) but stays far
  // shorter than the chain it hides behind.
  if (a.share_cu && lds < 54 * 1024) lds = 54 * 1024;
  const int p_tiles = (a.P + PR_BP - 1) / PR_BP;
  dim3 grid(a.reduce ? 1 : p_tiles, (a.Mp + PR_BM - 1) / PR_BM, a.N);
  ScopedTimer t(ctx, timer_name);
  if (a.bk.type == 0) hipLaunchKernelGGL(patch_rbf_kernel<0>, grid, dim3(256), lds, ctx->stream, a);
  else hipLaunchKernelGGL(patch_rbf_kernel<1>, grid, dim3(256), lds, ctx->stream, a);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

int head_kdiag(dcgp_ctx* ctx, const double* X, int N, int n_mod, int H, int W, int C, int f, int s, BaseKernel bk,
               const double* w, double* out_N) {
  const int Ho = (H - f) / s + 1, Wo = (W - f) / s + 1, P = Ho * Wo, L = f * f * C;
  const int p_tiles = (P + 63) / 64, n_pairs = p_tiles * (p_tiles + 1) / 2;
  const int HWC = H * W * C, Lp = (L + 3) & ~3;
  double* partial = (double*)ws_get(ctx, "kdiag_partial", (size_t)N * n_pairs * sizeof(double));
  if (!partial) return DCGP_ERR_ALLOC;
  size_t lds = (size_t)(((HWC + 1) & ~1) + 128 + 4) * sizeof(double) + (size_t)Lp * sizeof(int);
  if (lds > 160 * 1024) return ctx_fail(ctx, DCGP_ERR_ARG, "head_kdiag: image does not fit LDS");
  ScopedTimer t(ctx, "head_kdiag");
  if (bk.type == 0)
    hipLaunchKernelGGL(head_kdiag_kernel<0>, dim3(n_pairs, N), dim3(256), lds, ctx->stream, X, n_mod, H, W, C, f, s, Ho, Wo, P, L,
                       bk, w, partial, n_pairs, p_tiles);
  else
    hipLaunchKernelGGL(head_kdiag_kernel<1>, dim3(n_pairs, N), dim3(256), lds, ctx->stream, X, n_mod, H, W, C, f, s, Ho, Wo, P, L,
                       bk, w, partial, n_pairs, p_tiles);
  LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(kdiag_reduce_kernel, dim3((N + 127) / 128), dim3(128), 0, ctx->stream, partial, n_pairs, N,
                     1.0 / ((double)P * (double)P), out_N);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

extern "C" int dcgp_extract_patches(dcgp_ctx* ctx, const double* X, int N, int H, int W, int C, int f, int stride,
                                    double* out, int pnl) {
  if (!ctx || !X || !out || N <= 0 || f <= 0 || stride <= 0 || f > H || f > W)
    return ctx ? ctx_fail(ctx, DCGP_ERR_ARG, "extract_patches: bad args") : DCGP_ERR_ARG;
  const int Ho = (H - f) / stride + 1, Wo = (W - f) / stride + 1;
  long total = (long)N * Ho * Wo * f * f * C;
  hipLaunchKernelGGL(extract_patches_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, X, N, H,
                     W, C, f, stride, Ho, Wo, out, pnl);
  LAUNCH_CHECK(ctx);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}
