// conv_fused.hip -- ConvLayer.conditional_ND + Layer.sample_from_conditional of ONE column strip in ONE workgroup
// (conv_gp/layers.py:96-135 -> views.py:40-44, layers.py:23-32, conditionals.py:29-65, layers.py:128-134, reparameterize).
//
// The unfused route passes the [M x P*N'] matrix through HBM three times (K_uf written, A1 = inv(L) K_uf written and read
// back R + 2 times by 128 x 128 tiles, each of which re-stages both operands through LDS behind a barrier per k-tile).
// Here a workgroup owns BN = 16 * FN patch columns for the whole layer:
//   0. the strip's images (at most a few: columns are n*P + p) are staged in LDS, |x_p|^2 per column;
//   1. K_uf[:, strip] from the LDS images (patch gather as MFMA B operand, inducing patches Z^T streamed from L2 as A
//      operand, fp64 exp) straight into the LDS-resident strip [Mp][BN] -- never written to HBM;
//   2. A1 = inv(L) K_uf in place (lower-triangular product; every wave keeps its rows in registers until all waves
//      have read the strip), sum_m A1^2 from the accumulators;
//   3. for r < R: T_r = G_r^T A1 (upper-triangular product), sum_m T_r^2 from the accumulators, T never stored;
//      then mean = alpha^T A1;
//   4. var = Knn - s1 + s2, sample = mean + z sqrt(var + jitter), written in the N x (P*R) layout of layers.py:128-131.
// The B operand of every product is the resident strip; the A operand (inv(L)^T, G_r, alpha, Z^T: a few MB that every
// workgroup reads, L2 / Infinity-Cache resident) goes from global memory STRAIGHT into MFMA A registers: each wave streams
// only the 16 columns of its own row fragment and only its live k-tiles, CF_D tiles ahead -- no LDS staging, no barrier
// inside any k loop (5 barriers per workgroup in all), HBM traffic = the images in and the samples out.
// Row fragments are dealt to the waves boustrophedon (w, 2W-1-w, 2W+w, ...) so that every wave carries the same number
// of live k-tiles of the triangular products.
#include "layer.h"
#include "rng.h"

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int CF_D = 3;   // A-operand k-tiles in flight per wave beside the one being multiplied

__device__ __forceinline__ int frag_of(int w, int W, int c) { return (c >> 1) * 2 * W + ((c & 1) ? 2 * W - 1 - w : w); }

template <int FN, int MAXF, int NT, int BT>
__global__ __launch_bounds__(NT) void conv_fused_kernel(ConvFusedArgs a) {
  constexpr int BN = FN * 16;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int Mp = a.Mp, nf = Mp >> 4, R = a.R;
  const int tid = threadIdx.x, lane = tid & 63, lrow = lane >> 4, lcol = lane & 15;
  const int W = NT / 64;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* strip = smem;                                  // [Mp][BN], 16-column groups XOR-swizzled by (row & (FN-1))
  double* aux = smem + a.lds_main;                       // images of the strip; dead after phase 1
  double* xn = aux + a.lds_img;                          // [BN] |x_p|^2
  int* koff = reinterpret_cast<int*>(xn + BN);           // [Lp]
  const int j0 = blockIdx.x * BN;
  const int jmax = a.Kc - 1;

  // fragments of this wave
  int nfw = 0;
#pragma unroll
  for (int c = 0; c < MAXF; ++c) nfw += frag_of(wave, W, c) < nf ? 1 : 0;
  nfw = __builtin_amdgcn_readfirstlane(nfw);

  // ---- phase 0: images of the strip -> LDS, patch-element offsets, |x|^2 per column -----------------------------
  const int n_first = j0 / a.P;
  const int n_last = min(j0 + BN - 1, jmax) / a.P;
  {
    const int total = (n_last - n_first + 1) * a.HWC;
    for (int i0 = 0; i0 < total; i0 += 4 * NT) {
      double t[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = i0 + e * NT + tid;
        const int n = i / a.HWC, o = i - n * a.HWC;
        t[e] = (i < total) ? a.X[(long)((n_first + n) % a.n_mod) * a.HWC + o] : 0.0;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = i0 + e * NT + tid;
        if (i < total) aux[i] = t[e];
      }
    }
    for (int l = tid; l < a.Lp; l += NT) {
      const int ll = l < a.L ? l : 0;
      const int c = ll % a.C, t = ll / a.C;
      const int kw = t % a.f, kh = t / a.f;
      koff[l] = (kh * a.W + kw) * a.C + c;
    }
  }
  // LDS offset of the patch of strip column c (columns beyond the matrix repeat the last one: finite, never written out)
  auto patch_off = [&](int c) {
    const int j = min(j0 + c, jmax);
    const int n = j / a.P, p = j - n * a.P;
    const int oh = p / a.Wo, ow = p - oh * a.Wo;
    return (n - n_first) * a.HWC + (oh * a.s * a.W + ow * a.s) * a.C;
  };
  __syncthreads();
  if (tid < 8 * BN) {   // 8 threads per column
    const int c = tid >> 3, sub = tid & 7;
    const int pb = patch_off(c);
    double s = 0.0;
    for (int l = sub; l < a.L; l += 8) {
      const double v = aux[pb + koff[l]];
      s = fma(v, v, s);
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    if (sub == 0) xn[c] = s;
  }
  __syncthreads();

  // per-lane constants of the strip accesses: element (row k, column y*16 + lcol) with k & 3 == lrow lives at
  // k * BN + ((y ^ (lrow & (FN-1))) * 16 + lcol)
  int bsw[FN];
#pragma unroll
  for (int y = 0; y < FN; ++y) bsw[y] = ((y ^ (lrow & (FN - 1))) << 4) + lcol;

  d4 acc[FN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int y = 0; y < FN; ++y) acc[y] = d4{0.0, 0.0, 0.0, 0.0};
  };

  // ---- phase 1: K_uf[:, strip] -> strip -------------------------------------------------------------------------
  {
    int pb[FN];
#pragma unroll
    for (int y = 0; y < FN; ++y) pb[y] = patch_off(y * 16 + lcol);
    const int nk4 = a.Lp >> 2;
    for (int c = 0; c < nfw; ++c) {
      const int f = frag_of(wave, W, c);
      const double* __restrict__ zt = a.ZT + 16 * f + lcol;
      zero_acc();
      constexpr int D4 = 4;
      double ring[D4 + 1];
#pragma unroll
      for (int u = 0; u < D4; ++u) ring[u] = zt[(long)(4 * min(u, nk4 - 1) + lrow) * Mp];
      for (int t = 0; t < nk4; t += D4 + 1) {
#pragma unroll
        for (int u = 0; u <= D4; ++u) {
          if (t + u < nk4) {
            ring[(u + D4) % (D4 + 1)] = zt[(long)(4 * min(t + u + D4, nk4 - 1) + lrow) * Mp];
            const int k = 4 * (t + u) + lrow;
            const int ko = koff[k];
            const bool kin = k < a.L;
#pragma unroll
            for (int y = 0; y < FN; ++y) {
              const double v = aux[pb[y] + ko];
              acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u], kin ? v : 0.0, acc[y], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int m = 16 * f + lrow + 4 * v;
        const double znm = a.zn[m];
#pragma unroll
        for (int y = 0; y < FN; ++y) {
          const double kv = a.bk.template eval_as<BT>(acc[y][v], xn[y * 16 + lcol], znm);
          strip[m * BN + bsw[y]] = (m < a.M) ? kv : 0.0;
        }
      }
    }
  }
  __syncthreads();
  // training step: the reverse pass reads K_uf and A1 from HBM (k-major [Mp][ldk], column j)
  auto store_strip = [&](double* __restrict__ out) {
    for (int idx = tid; idx < Mp * BN; idx += NT) {
      const int m = idx / BN, c = idx - m * BN;
      const int j = j0 + c;
      if (j <= jmax) out[(long)m * a.ldk + j] = strip[m * BN + ((((c >> 4) ^ (m & (FN - 1))) << 4) | (c & 15))];
    }
  };
  if (a.Kuf_out) store_strip(a.Kuf_out);

  // ---- the A-operand stream ---------------------------------------------------------------------------------------
  // lane (lrow, lcol) of k-substep q of k-tile kt needs Wt[kt*16 + 4q + lrow][16 f + lcol]: per-lane byte offset voff[q],
  // everything else (matrix r, k-tile, fragment) is a scalar byte offset
  constexpr unsigned OOB = 0x80000000u;
  unsigned voff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) voff[q] = (unsigned)(((4 * q + lrow) * Mp + lcol) * 8);
  auto ldw = [&](const __amdgpu_buffer_rsrc_t& rs, int soff, double (&dst)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff[q], soff, 0);
      __builtin_memcpy(&dst[q], &v, 8);
    }
  };
  auto mfma_tile = [&](int kt, const double (&w)[4]) {
    const double* b = strip + kt * 16 * BN;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int y = 0; y < FN; ++y)
        acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(w[q], b[(4 * q + lrow) * BN + bsw[y]], acc[y], 0, 0, 0);
  };
  (void)OOB;

  // ---- phase 2: A1 = inv(L) K_uf (lower-triangular W: fragment f needs k-tiles 0 .. f) --------------------------------
  const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.LinvT), 0, Mp * Mp * 8, 0x00020000);
  d4 a1[MAXF][FN];
  double s1acc[FN];
#pragma unroll
  for (int y = 0; y < FN; ++y) s1acc[y] = 0.0;
#pragma unroll
  for (int c = 0; c < MAXF; ++c) {
    if (c < nfw) {
      const int f = frag_of(wave, W, c);
      const int fo = 16 * f * 8;
      zero_acc();
      double ring[CF_D + 1][4];
#pragma unroll
      for (int u = 0; u < CF_D; ++u) ldw(lrs, fo + min(u, f) * 16 * Mp * 8, ring[u]);
      for (int kt = 0; kt <= f; kt += CF_D + 1) {
#pragma unroll
        for (int u = 0; u <= CF_D; ++u) {
          if (kt + u <= f) {
            ldw(lrs, fo + min(kt + u + CF_D, f) * 16 * Mp * 8, ring[(u + CF_D) % (CF_D + 1)]);
            mfma_tile(kt + u, ring[u]);
          }
        }
      }
#pragma unroll
      for (int y = 0; y < FN; ++y) {
        a1[c][y] = acc[y];
#pragma unroll
        for (int v = 0; v < 4; ++v) s1acc[y] = fma(acc[y][v], acc[y][v], s1acc[y]);
      }
    }
  }
  __syncthreads();   // every wave is done reading K_uf
#pragma unroll
  for (int c = 0; c < MAXF; ++c) {
    if (c < nfw) {
      const int f = frag_of(wave, W, c);
#pragma unroll
      for (int y = 0; y < FN; ++y)
#pragma unroll
        for (int v = 0; v < 4; ++v) strip[(16 * f + lrow + 4 * v) * BN + bsw[y]] = a1[c][y][v];
    }
  }
  __syncthreads();   // A1 published
  if (a.A1_out) store_strip(a.A1_out);

  // ---- phase 3: T_r = G_r^T A1 for every r (upper-triangular W: fragment f needs k-tiles f .. nf-1), one flat stream ----
  // s2 of output r is parked in the lanes with lrow == (r & 3) of keep[r >> 2][.]
  double keep[4][FN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int y = 0; y < FN; ++y) keep[i][y] = 0.0;
  if (a.G && nfw > 0) {
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.G), 0, R * Mp * Mp * 8, 0x00020000);
    int steps_per_r = 0;
    for (int c = 0; c < nfw; ++c) steps_per_r += nf - frag_of(wave, W, c);
    const int total = R * steps_per_r;
    // load cursor / compute cursor: (r, c, kt) with f = frag_of(c)
    int lr = 0, lc = 0, lf = frag_of(wave, W, 0), lk = lf, lleft = total - 1;   // lleft: advances still allowed (clamp at the last tile)
    int cr = 0, cc = 0, cf = lf, ck = lf;
    auto lsoff = [&]() { return ((lr * Mp + lk * 16) * Mp + 16 * lf) * 8; };
    auto ladv = [&]() {
      if (lleft > 0) {
        --lleft;
        if (++lk >= nf) {
          if (++lc == nfw) { lc = 0; ++lr; }
          lf = frag_of(wave, W, lc);
          lk = lf;
        }
      }
    };
    double s2acc[FN];
#pragma unroll
    for (int y = 0; y < FN; ++y) s2acc[y] = 0.0;
    zero_acc();
    double ring[CF_D + 1][4];
#pragma unroll
    for (int u = 0; u < CF_D; ++u) { ldw(grs, lsoff(), ring[u]); ladv(); }
    for (int t = 0; t < total; t += CF_D + 1) {
#pragma unroll
      for (int u = 0; u <= CF_D; ++u) {
        if (t + u < total) {
          ldw(grs, lsoff(), ring[(u + CF_D) % (CF_D + 1)]);
          ladv();
          mfma_tile(ck, ring[u]);
          if (++ck >= nf) {   // fragment done: fold its rows into the column sums of squares
#pragma unroll
            for (int y = 0; y < FN; ++y) {
#pragma unroll
              for (int v = 0; v < 4; ++v) s2acc[y] = fma(acc[y][v], acc[y][v], s2acc[y]);
              acc[y] = d4{0.0, 0.0, 0.0, 0.0};
            }
            if (++cc == nfw) {   // output r done
#pragma unroll
              for (int y = 0; y < FN; ++y) {
                double s = s2acc[y];
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
#pragma unroll
                for (int i = 0; i < 4; ++i) keep[i][y] = ((cr >> 2) == i && (cr & 3) == lrow) ? s : keep[i][y];
                s2acc[y] = 0.0;
              }
              cc = 0;
              ++cr;
            }
            cf = frag_of(wave, W, cc);
            ck = cf;
          }
        }
      }
    }
  }

  // ---- mean = alpha^T A1: k-tiles dealt round-robin to the waves, partial sums joined in the final reduction ----------
  zero_acc();
  {
    const double* __restrict__ al = a.alpha + lcol;
    for (int kt = wave; kt < nf; kt += W) {
      double w[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = al[(long)(kt * 16 + 4 * q + lrow) * a.Rp];
      mfma_tile(kt, w);
    }
  }
  // per-column sums over this wave's rows
#pragma unroll
  for (int y = 0; y < FN; ++y) {
    s1acc[y] += __shfl_xor(s1acc[y], 16);
    s1acc[y] += __shfl_xor(s1acc[y], 32);
  }
  __syncthreads();   // the strip is dead: reuse it as [W][BN] s1 | [W][R][BN] s2 | [W][16][BN] mean partials
  double* s1p = smem;
  double* s2p = s1p + W * BN;
  double* mup = s2p + W * R * BN;
#pragma unroll
  for (int y = 0; y < FN; ++y) {
    const int c = y * 16 + lcol;
    if (lrow == 0) s1p[wave * BN + c] = s1acc[y];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * i + lrow;
      if (r < R) s2p[(wave * R + r) * BN + c] = keep[i][y];
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) mup[(wave * 16 + lrow + 4 * v) * BN + c] = acc[y][v];
  }
  __syncthreads();

  // ---- phase 4: var, mean, sample in the N x (P*R) layout (column j, output r at j*R + r) ----------------------------
  for (int idx = tid; idx < BN * R; idx += NT) {
    const int c = idx / R, r = idx - c * R;
    const int j = j0 + c;
    if (j > jmax) continue;
    double s1 = 0.0, s2 = 0.0, m = 0.0;
    for (int w = 0; w < W; ++w) {
      s1 += s1p[w * BN + c];
      s2 += s2p[(w * R + r) * BN + c];
      m += mup[(w * 16 + r) * BN + c];
    }
    const double v = (a.knn - s1) + s2;
    if (a.idm && r == 0) {   // Conv2dMean (conv_gp/mean_functions.py:28-41): centre pixel of channel 0 onto map 0
      const int n = j / a.P, p = j - n * a.P;
      const int oh = p / a.Wo, ow = p - oh * a.Wo, c0 = a.f / 2;
      m += a.X[(((long)(n % a.n_mod) * a.H + oh * a.s + c0) * a.W + ow * a.s + c0) * a.C];
    }
    const long e = (long)j * R + r;
    for (int s = 0; s < a.rep; ++s) {
      const long o = (long)s * a.rep_stride + e;
      if (a.out_mean) a.out_mean[o] = m;
      if (a.out_var) a.out_var[o] = v;
      if (a.out_sample) {
        const double zz = a.z ? a.z[o] : philox_normal(a.seed, a.stream_id, (uint64_t)o);
        a.out_sample[o] = m + zz * sqrt(v + a.jitter);
      }
    }
  }
}

template <int FN, int MAXF, int NT>
int launch_fused(dcgp_ctx* ctx, const ConvFusedArgs& a, size_t lds) {
  const int BN = FN * 16;
  const unsigned grid = (unsigned)((a.Kc + BN - 1) / BN);
  if (a.bk.type == 0) hipLaunchKernelGGL((conv_fused_kernel<FN, MAXF, NT, 0>), dim3(grid), dim3(NT), lds, ctx->stream, a);
  else hipLaunchKernelGGL((conv_fused_kernel<FN, MAXF, NT, 1>), dim3(grid), dim3(NT), lds, ctx->stream, a);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

struct FusedPlan { int FN, W, MAXF; size_t lds; int lds_main, lds_img; };

// tile shape for a layer: the widest strip whose LDS footprint fits, waves = half the row fragments (one early + one late
// fragment each) up to 16
bool plan_fused(const ConvFusedArgs& a, FusedPlan* p) {
  static const int force_fn = getenv("DCGP_FUSED_FN") ? atoi(getenv("DCGP_FUSED_FN")) : 0;
  const int nf = a.Mp / 16;
  if (a.Rp != 16 || a.R > 16 || a.Mp > 1024) return false;
  for (int FN = 4; FN >= 1; FN >>= 1) {
    if (force_fn && FN != force_fn) continue;
    const int BN = FN * 16;
    int W, MAXF;
    if (nf <= 16) { W = 8; MAXF = 2; }
    else if (nf <= 24) { W = 12; MAXF = 2; }
    else if (nf <= 32) { W = 16; MAXF = 2; }
    else { W = 16; MAXF = 4; }
    if (MAXF == 4 && FN != 1) continue;            // instantiated shapes: <4,2,512> <2,2,512|768|1024> <1,2,512> <1,4,1024>
    if (MAXF == 2 && W != 8 && FN != 2) continue;
    const int nimg = (BN - 1) / a.P + 2;           // images a strip can touch
    const long main_d = (long)a.Mp * BN > (long)(W + W * a.R + W * 16) * BN ? (long)a.Mp * BN : (long)(W + W * a.R + W * 16) * BN;
    const long img_d = ((long)nimg * a.HWC + 1) & ~1L;
    const long bytes = (main_d + img_d + BN) * 8 + (long)a.Lp * 4;
    if (bytes > 160 * 1024) continue;
    p->FN = FN; p->W = W; p->MAXF = MAXF; p->lds = (size_t)bytes; p->lds_main = (int)main_d; p->lds_img = (int)img_d;
    return true;
  }
  return false;
}

}  // namespace

bool conv_fused_ok(const ConvFusedArgs& a) {
  const bool off = getenv("DCGP_NO_FUSED_LAYER") != nullptr;   // A/B switch (read per call: tests flip it): the unfused sweep + GEMM route
  FusedPlan p;
  return !off && plan_fused(a, &p);
}

int conv_fused(dcgp_ctx* ctx, const ConvFusedArgs& a_in) {
  if (a_in.Kc <= 0) return DCGP_OK;
  FusedPlan p;
  if (!plan_fused(a_in, &p)) return ctx_fail(ctx, DCGP_ERR_ARG, "conv_fused: layer shape not supported (M = %d, R = %d)", a_in.M, a_in.R);
  if ((long)a_in.R * a_in.Mp * a_in.Mp * 8 >= (1L << 31)) return ctx_fail(ctx, DCGP_ERR_ARG, "conv_fused: G exceeds 2 GiB");
  ConvFusedArgs a = a_in;
  a.lds_main = p.lds_main; a.lds_img = p.lds_img;
  static bool attr_done = false;
  if (!attr_done) {   // more than 64 KB of dynamic LDS needs the opt-in
#define CF_ATTR(FN, MF, NT)                                                                                                           \
  hipFuncSetAttribute((const void*)conv_fused_kernel<FN, MF, NT, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
  hipFuncSetAttribute((const void*)conv_fused_kernel<FN, MF, NT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    CF_ATTR(4, 2, 512) CF_ATTR(2, 2, 512) CF_ATTR(1, 2, 512) CF_ATTR(2, 2, 768) CF_ATTR(2, 2, 1024) CF_ATTR(1, 4, 1024)
#undef CF_ATTR
    attr_done = true;
  }
  ScopedTimer t(ctx, "conv_fused");
  if (p.MAXF == 4) return launch_fused<1, 4, 1024>(ctx, a, p.lds);
  if (p.W == 16) return launch_fused<2, 2, 1024>(ctx, a, p.lds);
  if (p.W == 12) return launch_fused<2, 2, 768>(ctx, a, p.lds);
  if (p.FN == 4) return launch_fused<4, 2, 512>(ctx, a, p.lds);
  if (p.FN == 2) return launch_fused<2, 2, 512>(ctx, a, p.lds);
  return launch_fused<1, 2, 512>(ctx, a, p.lds);
}
