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
#include "sweep_dev.h"

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
constexpr int PR_D = 3;      // k sub-steps of Z^T in flight per wave beside the one being multiplied

__device__ __forceinline__ int patch_base(int p, int P, int Wo, int s, int W, int C) {
  if (p >= P) p = 0;
  int oh = p / Wo, ow = p - oh * Wo;
  return (oh * s * W + ow * s) * C;
}

// grid: (p tiles [write mode] or 1 [reduce mode], Mp/64, N); block 256 = 4 waves as 2 (m) x 2 (p).
// The A operand (Z^T, k-major, L2-resident: every workgroup reads the same few hundred KB) goes from global memory straight
// into MFMA registers, PR_D sub-steps ahead -- no LDS staging, no barrier in the k loop.  (Staged through LDS in chunks of 32
// rows it cost two barriers and one exposed memory latency per chunk: 8 chunks at the head's L = 250, ~40 us per workgroup
// where the MFMAs need 7.)
template <int BT>
__device__ __forceinline__ void patch_rbf_body(const PatchRbfArgs& a, const int bx, const int by, const int bz) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int HWC = a.H * a.W * a.C;
  double* img = smem;                                   // [HWC] (+pad to even)
  double* red = img + ((HWC + 1) & ~1);                 // [2][PR_BM] (reduce mode)
  int* koff = reinterpret_cast<int*>(red + 3 * PR_BM);  // [Lp]  (red: [2][PR_BM] reduce scratch, then [PR_BM] |z|^2)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = lane >> 4, lcol = lane & 15;
  const int n = bz, m0 = by * PR_BM;

  const double* __restrict__ Xn = a.X + (long)((a.n0 + n) % a.n_mod) * HWC;
  // everything the k loop and the epilogue read from global memory is requested here, in front of the image: one memory
  // latency for all of it instead of one each (|z|^2 after the image, the first Z^T sub-steps after the barrier and the
  // patch weights in the epilogue cost ~1 us apiece per workgroup)
  const double znv = (tid < PR_BM && m0 + tid < a.Mp) ? a.zn[m0 + tid] : 0.0;
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.ZT), 0, a.Lp * a.Mp * 8, 0x00020000);
  unsigned zoff[2];
#pragma unroll
  for (int x = 0; x < 2; ++x) zoff[x] = (unsigned)((lrow * a.Mp + min(m0 + wm * 32 + x * 16 + lcol, a.Mp - 1)) * 8);
  const int nk4 = a.Lp >> 2;
  double ring[PR_D + 1][2];
  auto ldz = [&](int k4, double (&dst)[2]) {
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(zrs, (int)zoff[x], k4 * 4 * a.Mp * 8, 0);
      __builtin_memcpy(&dst[x], &v, 8);
    }
  };
#pragma unroll
  for (int u = 0; u < PR_D; ++u) ldz(min(u, nk4 - 1), ring[u]);
  bool primed = true;
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
  // (zoff: this lane's A-operand columns: rows m0 + wm*32 + x*16 + lcol of Z (columns of Z^T); beyond Mp: re-read the last one, dropped later)
  double* znl = red + 2 * PR_BM;   // behind the reduce scratch: |z|^2 of the workgroup's 64 rows (koff follows)
  if (tid < PR_BM) znl[tid] = znv;
  __syncthreads();

  const int p_tiles = (a.P + PR_BP - 1) / PR_BP;
  const int pt_lo = a.reduce ? 0 : bx, pt_hi = a.reduce ? p_tiles : bx + 1;

  double rsum[2][4];   // reduce mode: per (fm, v) running row sums
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int v = 0; v < 4; ++v) rsum[x][v] = 0.0;

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

    double wp[2];   // reduce mode: the patch weights of this lane's two columns (0 beyond the last patch)
#pragma unroll
    for (int y = 0; y < 2; ++y) wp[y] = (a.reduce && p0 + y * 16 + lcol < a.P) ? a.w[p0 + y * 16 + lcol] : 0.0;
    // B operand: this lane's two patch columns, gathered from the LDS image ONE sub-step ahead of the MFMAs that use them,
    // their image offsets read a whole group of sub-steps ahead (offset -> gather -> MFMA inside one step exposed two
    // dependent LDS round trips per step; with fewer than four waves on the SIMD -- the second round of a 1.5-round grid --
    // nothing covered them).  Only the last sub-step can reach k >= L, so only gathers that may be the last carry the mask.
    constexpr int G = PR_D + 1;
    auto offs = [&](int k4, int (&ko)[G]) {
#pragma unroll
      for (int u = 0; u < G; ++u) ko[u] = koff[4 * min(k4 + u, nk4 - 1) + lrow];
    };
    auto gather = [&](int ko, bool kin, double (&bv)[2]) {
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        const double v = img[pb[y] + ko];
        bv[y] = kin ? v : 0.0;
      }
    };
    auto mma = [&](const double (&w)[2], const double (&bv)[2]) {
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(w[x], bv[y], acc[x][y], 0, 0, 0);
#pragma unroll
      for (int y = 0; y < 2; ++y) xn[y] = fma(bv[y], bv[y], xn[y]);
    };
    if (!primed) {
#pragma unroll
      for (int u = 0; u < PR_D; ++u) ldz(min(u, nk4 - 1), ring[u]);
    }
    primed = false;
    const bool last_in = 4 * (nk4 - 1) + lrow < a.L;
    int kc[G], kn[G];
    double bc[2], bn[2];
    offs(0, kc);
    gather(kc[0], nk4 > 1 || last_in, bc);
    // full groups cover sub-steps [0, T), T the largest multiple of G that leaves the last sub-step to the tail
    const int T = ((nk4 - 1) / G) * G;
    int t = 0;
    for (; t < T; t += G) {   // no conditionals around the loads
      offs(t + G, kn);
#pragma unroll
      for (int u = 0; u < G; ++u) {
        ldz(min(t + u + PR_D, nk4 - 1), ring[(u + PR_D) % G]);
        if (u < PR_D) gather(kc[u + 1], true, bn);
        else gather(kn[0], t + G < nk4 - 1 || last_in, bn);   // t + G <= T <= nk4 - 1: may be the last sub-step
        mma(ring[u], bc);
        bc[0] = bn[0]; bc[1] = bn[1];
      }
#pragma unroll
      for (int u = 0; u < G; ++u) kc[u] = kn[u];
    }
    ldz(min(t + PR_D, nk4 - 1), ring[PR_D]);   // the tail can be G sub-steps long; kc holds their offsets
#pragma unroll
    for (int u = 0; u < G; ++u)
      if (t + u < nk4) {
        if (u < PR_D && t + u + 1 < nk4) gather(kc[u + 1], t + u + 1 < nk4 - 1 || last_in, bn);
        mma(ring[u], bc);
        bc[0] = bn[0]; bc[1] = bn[1];
      }
    // |x_p|^2 for column lcol of each fragment: combine the 4 k-groups
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      xn[y] += __shfl_xor(xn[y], 16);
      xn[y] += __shfl_xor(xn[y], 32);
    }
    // epilogue: the 4 accumulator values of a fragment together (their exps interleaved)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        const int p = p0 + y * 16 + lcol;
        double kv[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // two values at a time: all four interleaved cost 20 spilled registers per lane
          double k2[2], n1[2], n2[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) { k2[e] = acc[x][y][2 * h + e]; n1[e] = xn[y]; n2[e] = znl[wm * 32 + x * 16 + lrow + 4 * (2 * h + e)]; }
          a.bk.template eval_n<BT, 2>(k2, n1, n2);
          kv[2 * h] = k2[0]; kv[2 * h + 1] = k2[1];
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int m = m0 + wm * 32 + x * 16 + lrow + 4 * v;
          if (a.reduce) {
            rsum[x][v] += wp[y] * kv[v];   // wp = 0 beyond the last patch
          } else if (m < a.M && p < a.P) {
            a.out[(long)m * a.sM + (long)n * a.sN + (long)p * a.sP] = kv[v];
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

template <int BT>
__global__ __launch_bounds__(256, BT == 0 ? 4 : 2) void patch_rbf_kernel(PatchRbfArgs a) {   // the acos epilogue needs the registers
  patch_rbf_body<BT>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---------------------------------------------------------------------------------------------
// ConvKernel.Kdiag: per image sum_{p,p'} w_p w_p' k(x_p, x_p') / P^2, upper triangle of 64x64 patch
// tiles (symmetry: off-diagonal tiles count twice).  grid (tile-row pairs, N); partial[n][row pair].
// ---------------------------------------------------------------------------------------------
struct KdiagArgs {
  const double* X; int n_mod, H, W, C, f, s, Ho, Wo, P, L; BaseKernel bk; const double* w; double* partial; int n_pairs, p_tiles;
};
// One workgroup = one image and one PAIR of tile rows (brow, p_tiles - 1 - brow) of the symmetric P x P patch Gram matrix: it
// walks the tiles on and right of the diagonal of both rows -- p_tiles + 1 tile products whatever brow is -- so the image
// load, the offset table and the patch norms are paid once per ~p_tiles tile products and every workgroup is the same length
// (one workgroup per tile PAIR paid them per product: 14400 workgroups of 7 k sub-steps each on the 28x28 head, setup-bound).
template <int BT>
__device__ __forceinline__ void head_kdiag_body(const KdiagArgs& k, const int brow, const int n) {
  const double* __restrict__ X = k.X;
  const int n_mod = k.n_mod, H = k.H, W = k.W, C = k.C, f = k.f, s = k.s, Wo = k.Wo, P = k.P, L = k.L, p_tiles = k.p_tiles;
  const BaseKernel& bk = k.bk;
  const double* __restrict__ w = k.w;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int HWC = H * W * C;
  const int Lp = (L + 3) & ~3;
  double* img = smem;                             // [HWC]
  double* xn = img + ((HWC + 1) & ~1);            // [p_tiles * 64] patch norms
  double* wl = xn + p_tiles * 64;                 // [p_tiles * 64] patch weights (0 beyond P)
  double* red = wl + p_tiles * 64;                // [4]
  int* koff = reinterpret_cast<int*>(red + 4);    // [Lp]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = lane >> 4, lcol = lane & 15;
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
  // patch norms and weights of the whole image: 4 threads per patch, 8 terms in flight per thread (one term at a time each
  // waited for its two dependent LDS reads: 16 us per workgroup at L = 250, as long as the tile products themselves)
  for (int q0 = 0; q0 < p_tiles * 64; q0 += 64) {
    const int p = q0 + (tid >> 2), part = tid & 3;
    const int pbq = patch_base(p, P, Wo, s, W, C);
    double sacc = 0.0;
    int l = part;
    for (; l + 28 < L; l += 32) {
      double v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = img[pbq + koff[l + 4 * e]];
#pragma unroll
      for (int e = 0; e < 8; ++e) sacc = fma(v[e], v[e], sacc);
    }
    for (; l < L; l += 4) {
      const double v = img[pbq + koff[l]];
      sacc = fma(v, v, sacc);
    }
    sacc += __shfl_xor(sacc, 1);
    sacc += __shfl_xor(sacc, 2);
    if (part == 0) { xn[p] = sacc; wl[p] = p < P ? w[p] : 0.0; }
  }
  __syncthreads();

  const int nk4 = Lp >> 2;
  double sum = 0.0;
  for (int pass = 0; pass < 2; ++pass) {
    const int tr = pass == 0 ? brow : p_tiles - 1 - brow;
    if (pass == 1 && tr == brow) break;
    int pa[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) pa[x] = patch_base(tr * 64 + wm * 32 + x * 16 + lcol, P, Wo, s, W, C);
    for (int tc = tr; tc < p_tiles; ++tc) {
      int pbc[2];
#pragma unroll
      for (int y = 0; y < 2; ++y) pbc[y] = patch_base(tc * 64 + wn * 32 + y * 16 + lcol, P, Wo, s, W, C);
      d4 acc[2][2];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};
      // operands gathered one k sub-step ahead of the MFMAs that use them, their image offsets read a group of four sub-steps
      // ahead: with offset -> gather -> MFMA in one step the two dependent LDS round trips cost 500 cycles a step on a SIMD
      // with a single wave (the tail of the grid), twice the four MFMAs.  Only the last sub-step can reach k >= L.
      auto offs = [&](int k4, int (&ko)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) ko[u] = koff[4 * min(k4 + u, nk4 - 1) + lrow];
      };
      auto gather = [&](int ko, bool kin, double (&av)[2], double (&bv)[2]) {
#pragma unroll
        for (int x = 0; x < 2; ++x) { const double v = img[pa[x] + ko]; av[x] = kin ? v : 0.0; }
#pragma unroll
        for (int y = 0; y < 2; ++y) { const double v = img[pbc[y] + ko]; bv[y] = kin ? v : 0.0; }
      };
      auto mma = [&](const double (&av)[2], const double (&bv)[2]) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[x], bv[y], acc[x][y], 0, 0, 0);
      };
      const bool last_in = 4 * (nk4 - 1) + lrow < L;
      int kc[4], kn[4];
      double ac[2], bc[2], an[2], bn[2];
      offs(0, kc);
      gather(kc[0], nk4 > 1 || last_in, ac, bc);
      int k4 = 0;
      for (; k4 + 4 < nk4; k4 += 4) {   // the gathers of sub-steps k4+1 .. k4+4 <= nk4-1: only the last may need the mask
        offs(k4 + 4, kn);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (u < 3) gather(kc[u + 1], true, an, bn);
          else gather(kn[0], k4 + 4 < nk4 - 1 || last_in, an, bn);
          mma(ac, bc);
#pragma unroll
          for (int x = 0; x < 2; ++x) { ac[x] = an[x]; bc[x] = bn[x]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) kc[u] = kn[u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)      // 1..4 sub-steps left; kc holds their offsets
        if (k4 + u < nk4) {
          if (u < 3 && k4 + u + 1 < nk4) gather(kc[u + 1], k4 + u + 1 < nk4 - 1 || last_in, an, bn);
          mma(ac, bc);
#pragma unroll
          for (int x = 0; x < 2; ++x) { ac[x] = an[x]; bc[x] = bn[x]; }
        }
      const double sym = (tr == tc) ? 1.0 : 2.0;
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
          const int pc = tc * 64 + wn * 32 + y * 16 + lcol;
          const double wc = sym * wl[pc], nc = xn[pc];
#pragma unroll
          for (int h = 0; h < 2; ++h) {   // two dependent exp chains interleaved, times the four waves of the SIMD
            double kv[2], n1[2], n2[2], ww[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int v = 2 * h + e;
              const int p = tr * 64 + wm * 32 + x * 16 + lrow + 4 * v;
              kv[e] = acc[x][y][v]; n1[e] = xn[p]; n2[e] = nc;
              ww[e] = wl[p] * wc;
            }
            bk.template eval_n<BT, 2>(kv, n1, n2);
            sum += ww[0] * kv[0];
            sum += ww[1] * kv[1];
          }
        }
    }
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  if (tid == 0) k.partial[(long)n * k.n_pairs + brow] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <int BT>
__global__ __launch_bounds__(256, 4) void head_kdiag_kernel(KdiagArgs k) {
  head_kdiag_body<BT>(k, blockIdx.x, blockIdx.y);
}

// The head's two sweeps in ONE launch (no second stream, no cross-stream join in front of the head's conditional): first the
// 64-row blocks of Kzx of every image -- each walks all p_tiles patch tiles of its image, the long workgroups -- then the Kdiag
// workgroups (a pair of tile rows of the patch Gram matrix each, p_tiles + 1 tile products -- about as long as a Kzx block).
__global__ __launch_bounds__(256, 4) void head_sweep_kernel(PatchRbfArgs a, KdiagArgs k, int ny) {
  const int nz = a.N * ny;
  if ((int)blockIdx.x < nz) {
    patch_rbf_body<0>(a, 0, blockIdx.x % ny, blockIdx.x / ny);
  } else {
    const int b = blockIdx.x - nz;
    head_kdiag_body<0>(k, b % k.n_pairs, b / k.n_pairs);
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

__global__ __launch_bounds__(256) void zs_build_kernel(ZsTask t) {
  __shared__ double tile[32][33];
  zs_task(t, blockIdx.x, gridDim.x, tile);
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

// the sweeps' scaled Z operand (sweep_dev.h) for callers outside the one-launch prepare_all
int sweep_operand(dcgp_ctx* ctx, const double* Z, const double* in_scale, int M, int Mp, int L, double variance, double lengthscale, double* ZS) {
  ZsTask t;
  t.Z = Z; t.in_scale = in_scale; t.ZS = ZS; t.M = M; t.Mp = Mp; t.L = L; t.Lq = sweep_lq(L);
  t.csq = sqrt(1.4426950408889634074) / lengthscale; t.log2var = log2(variance);
  hipLaunchKernelGGL(zs_build_kernel, dim3(zs_items(Mp, L)), dim3(256), 0, ctx->stream, t);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

int kdiag_reduce(dcgp_ctx* ctx, const double* partial, int n_parts, int N, double scale, double* out_N) {
  hipLaunchKernelGGL(kdiag_reduce_kernel, dim3((N + 127) / 128), dim3(128), 0, ctx->stream, partial, n_parts, N, scale, out_N);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

int patch_rbf(dcgp_ctx* ctx, const PatchRbfArgs& a, const char* timer_name) {
  if (a.N <= 0) return DCGP_OK;
  if (a.n_mod <= 0) return ctx_fail(ctx, DCGP_ERR_ARG, "patch_rbf: n_mod must be positive");
  if (a.Mp % PR_BM != 0 && a.Mp % 16 != 0) return ctx_fail(ctx, DCGP_ERR_ARG, "patch_rbf: Mp must be a multiple of 16");
  const int HWC = a.H * a.W * a.C;
  size_t lds = (size_t)(((HWC + 1) & ~1) + 3 * PR_BM) * sizeof(double) + (size_t)a.Lp * sizeof(int);
  if ((long)a.Lp * a.Mp * 8 >= (1L << 31)) return ctx_fail(ctx, DCGP_ERR_ARG, "patch_rbf: Z^T exceeds 2 GiB");
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

static int kdiag_args(dcgp_ctx* ctx, const double* X, int N, int n_mod, int H, int W, int C, int f, int s, BaseKernel bk, const double* w,
                      KdiagArgs* k, size_t* lds) {
  const int Ho = (H - f) / s + 1, Wo = (W - f) / s + 1, P = Ho * Wo, L = f * f * C;
  const int p_tiles = (P + 63) / 64, n_pairs = (p_tiles + 1) / 2;   // workgroups per image: pairs of tile rows
  const int HWC = H * W * C, Lp = (L + 3) & ~3;
  double* partial = (double*)ws_get(ctx, "kdiag_partial", (size_t)N * n_pairs * sizeof(double));
  if (!partial) return DCGP_ERR_ALLOC;
  *lds = (size_t)(((HWC + 1) & ~1) + 2 * p_tiles * 64 + 4) * sizeof(double) + (size_t)Lp * sizeof(int);
  if (*lds > 160 * 1024) return ctx_fail(ctx, DCGP_ERR_ARG, "head_kdiag: image does not fit LDS");
  k->X = X; k->n_mod = n_mod; k->H = H; k->W = W; k->C = C; k->f = f; k->s = s; k->Ho = Ho; k->Wo = Wo; k->P = P; k->L = L;
  k->bk = bk; k->w = w; k->partial = partial; k->n_pairs = n_pairs; k->p_tiles = p_tiles;
  return DCGP_OK;
}

int head_kdiag(dcgp_ctx* ctx, const double* X, int N, int n_mod, int H, int W, int C, int f, int s, BaseKernel bk,
               const double* w, double* out_N) {
  KdiagArgs k;
  size_t lds = 0;
  DCGP_TRY(kdiag_args(ctx, X, N, n_mod, H, W, C, f, s, bk, w, &k, &lds));
  ScopedTimer t(ctx, "head_kdiag");
  if (bk.type == 0) hipLaunchKernelGGL(head_kdiag_kernel<0>, dim3(k.n_pairs, N), dim3(256), lds, ctx->stream, k);
  else hipLaunchKernelGGL(head_kdiag_kernel<1>, dim3(k.n_pairs, N), dim3(256), lds, ctx->stream, k);
  LAUNCH_CHECK(ctx);
  hipLaunchKernelGGL(kdiag_reduce_kernel, dim3((N + 127) / 128), dim3(128), 0, ctx->stream, k.partial, k.n_pairs, N,
                     1.0 / ((double)k.P * (double)k.P), out_N);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

// ConvKernel.Kzx (reduce-mode sweep `a`) and ConvKernel.Kdiag of the same images in one launch (RBF base kernel).  Kdiag is left
// as per-image partial sums: Kdiag[n] = *kd_scale * sum_i (*kd_partial)[n * *n_pairs + i]  (the fused head conditional adds them up).
int head_sweep(dcgp_ctx* ctx, const PatchRbfArgs& a, const double* w, const double** kd_partial, int* n_pairs, double* kd_scale) {
  if (a.N <= 0) return DCGP_OK;
  if (!a.reduce || a.bk.type != 0 || a.n_mod <= 0) return ctx_fail(ctx, DCGP_ERR_ARG, "head_sweep: reduce-mode RBF sweep expected");
  KdiagArgs k;
  size_t lds_k = 0;
  DCGP_TRY(kdiag_args(ctx, a.X, a.N, a.n_mod, a.H, a.W, a.C, a.f, a.s, a.bk, w, &k, &lds_k));
  const int HWC = a.H * a.W * a.C;
  size_t lds = (size_t)(((HWC + 1) & ~1) + 3 * PR_BM) * sizeof(double) + (size_t)a.Lp * sizeof(int);
  if (lds_k > lds) lds = lds_k;
  if (lds > 160 * 1024) return ctx_fail(ctx, DCGP_ERR_ARG, "head_sweep: image of %d doubles does not fit LDS", HWC);
  if ((long)a.Lp * a.Mp * 8 >= (1L << 31)) return ctx_fail(ctx, DCGP_ERR_ARG, "head_sweep: Z^T exceeds 2 GiB");
  const int ny = (a.Mp + PR_BM - 1) / PR_BM;
  const long nwg = (long)a.N * (k.n_pairs + ny);
  if (nwg > 0x7fffffffL) return ctx_fail(ctx, DCGP_ERR_ARG, "head_sweep: too many workgroups");
  ScopedTimer t(ctx, "head_sweep");
  hipLaunchKernelGGL(head_sweep_kernel, dim3((unsigned)nwg), dim3(256), lds, ctx->stream, a, k, ny);
  LAUNCH_CHECK(ctx);
  *kd_partial = k.partial; *n_pairs = k.n_pairs; *kd_scale = 1.0 / ((double)k.P * (double)k.P);
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
