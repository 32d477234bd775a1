// head_units.hip -- ConvKernel.Kzx and ConvKernel.Kdiag of the head (conv_gp/kernels.py:106-133) in one launch,
// cut into equal WAVE-sized units.
//
// What bounds these sweeps on gfx950 (tools/pipe_mix.hip, profiles/r03_pipe_mix.txt): v_mfma_f64_16x16x4_f64 and every
// VALU instruction of a SIMD issue one after the other -- 64 cycles per MFMA, 4.4 per fp64 FMA, no overlap even across
// waves.  A 16 x 16 tile of kernel values at patch length L costs ceil((L+2)/4) MFMAs plus, per value, whatever the
// epilogue spends in the VALU, so the epilogue's instruction count is as much the kernel as the products are:
//   * both operands arrive scaled by sqrt(log2(e))/lengthscale, and the two free slots behind a patch (L = 25 or 250
//     pads to 28 / 252) carry (-|z|^2/2 + log2 variance, 1) against (1, -|x|^2/2): the MFMA accumulator IS the base-2
//     exponent of the kernel value, no norm / scale arithmetic per value;
//   * 2^t by a magic-number split (t + 1.5*2^52: integer part in the low mantissa word, no v_rndne / v_cvt), a
//     degree-11 minimax polynomial on [-1/2, 1/2] (11 FMAs with scalar coefficient operands; |error| 2e-17) and
//     v_ldexp_f64: 16 VALU instructions per value with the clamp, 17 with its weighted accumulation (the previous
//     epilogue: 29 and 8 hazard nops);
//   * chains of 8 values interleaved, so no dependent fp64 pair is adjacent (no s_nop).
// Work decomposition: per image, one unit per 16-row fragment of Z (its Kzx row sums over all patches) and one per
// PAIR of fragment rows (i, nf-1-i) of the symmetric patch Gram matrix (tiles on and right of the diagonal, off-diagonal
// tiles counted twice) -- nf or nf+1 tile products each.  A workgroup is 4 waves = 4 units of one image behind ONE
// image load and one pass of patch norms; waves never synchronise after that.  ~11 000 units at the headline size, so
// the tail of the launch is one unit (~10 us) whatever the image count is relative to 256 CUs.
#include "common.h"
#include <type_traits>

namespace {

// small-range integer division by a launch-time constant without the ~40-instruction sequence: q = floor((i + 0.5) / d)
__device__ __forceinline__ int fdiv_small(int i, float inv_d) { return (int)(((float)i + 0.5f) * inv_d); }

// NK4 > 0 (with TL = L & 3): the k extent is NK4 sub-steps of 4, known at compile time -- the A operand of a unit is fetched
// once into registers, the patch-element offsets are per-lane constants, the product is straight-line code (L = 25: <7, 1>).
// NK4 == 0: any patch length, both operands streamed (L = 250: 63 sub-steps).
#ifndef HU_SB1
#define HU_SB1 __builtin_amdgcn_sched_barrier(0)   // the gathers of sub-step s + 1 issue BEFORE the MFMAs of sub-step s
#endif
#ifndef HU_WAVES
#define HU_WAVES 4
#endif
#ifndef HU_EXPN
#define HU_EXPN 4
#endif
// WRITE: the K_uf sweep of a conv layer (conv_gp/layers.py:23-32 on views.py:40-44) -- the same row units, every kernel value stored
// (kuf[m * sM + n * sN + p * sP]) instead of reduced; no Kdiag units.
template <int NK4, int TL, bool WRITE, int NT>
__global__ __launch_bounds__(NT, HU_WAVES) void head_units_kernel(HeadUnitsArgs a) {
  constexpr int WPG = NT / 64;   // units (waves) per workgroup
  constexpr bool RES = NK4 > 0;
  constexpr int NKR = RES ? NK4 : 1;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int HWC = a.HWC, L = a.L, nk4 = RES ? NK4 : a.Lq >> 2, nfp = a.nfp, P = a.P, np16 = nfp * 16;
  const int HWCe = (HWC + 1) & ~1;
  double* img = smem;                                   // [HWC] image times sqrt(c)
  double* xb = img + HWCe;                              // [np16]  -c |x_p|^2 / 2
  double* wl = xb + np16;                               // [np16]  patch weights, 0 beyond P
  double* rs = wl + np16;                               // [H * Wr] row sums of squares (set-up only)
  int* pbl = reinterpret_cast<int*>(rs + ((a.H * (a.W - a.f + 1) + 1) & ~1));   // [np16]  byte offset of the patch's first element in img
  int* koff = pbl + np16;                               // [Lq]    byte offset of patch element l
  const char* imgb = reinterpret_cast<const char*>(img);
  const int tid = threadIdx.x, lane = tid & 63, lrow = lane >> 4, lcol = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.x / a.wgs_per_img, bw = blockIdx.x - n * a.wgs_per_img;
  const double* __restrict__ Xn = a.X + (long)((a.n0 + n) % a.n_mod) * HWC;
  auto ldi = [&](int byte_off) { return *reinterpret_cast<const double*>(imgb + byte_off); };

  // ---- set-up, once per workgroup: the scaled image, the offset tables, patch norms from a separable window sum ----
  for (int i0 = 0; i0 < HWC; i0 += 8 * NT) {   // batches of 8 loads per thread: one memory latency for all of them
    double t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = i0 + e * NT + tid;
      t[e] = (i < HWC) ? Xn[i] : 0.0;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = i0 + e * NT + tid;
      if (i < HWC) img[i] = t[e] * a.csq;
    }
  }
  for (int l = tid; l < a.Lq; l += NT) {
    const int ll = l < L ? l : 0;
    const int t = fdiv_small(ll, a.inv_C), c = ll - t * a.C;
    const int kh = fdiv_small(t, a.inv_f), kw = t - kh * a.f;
    koff[l] = ((kh * a.W + kw) * a.C + c) * 8;
  }
  for (int p = tid; p < np16; p += NT) {
    const int q = p < P ? p : 0;                 // patches beyond P repeat the first one (finite values, weight 0)
    const int oh = fdiv_small(q, a.inv_Wo), ow = q - oh * a.Wo;
    pbl[p] = (oh * a.s * a.W + ow * a.s) * a.C * 8;
    wl[p] = (!WRITE && p < P) ? a.w[p] : 0.0;
  }
  __syncthreads();
  {
    // rs[r][x] = sum over the f*C contiguous elements of image row r that a patch starting at column x covers; 4 threads per entry
    const int Wr = a.W - a.f + 1, fC = a.f * a.C;
    for (int i0 = 0; i0 < a.H * Wr; i0 += NT / 4) {
      const int i = i0 + (tid >> 2), part = tid & 3;
      double acc = 0.0;
      if (i < a.H * Wr) {
        const int r = fdiv_small(i, a.inv_Wr), x = i - r * Wr;
        const double* src = img + (r * a.W + x) * a.C;
        for (int j = part; j < fC; j += 4) acc = fma(src[j], src[j], acc);
      }
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      if (part == 0 && i < a.H * Wr) rs[i] = acc;
    }
    __syncthreads();
    for (int p = tid; p < np16; p += NT) {
      const int q = p < P ? p : 0;
      const int oh = fdiv_small(q, a.inv_Wo), ow = q - oh * a.Wo;
      const double* src = rs + oh * a.s * Wr + ow * a.s;
      double acc = 0.0;
      for (int kh = 0; kh < a.f; ++kh) acc += src[kh * Wr];
      xb[p] = -0.5 * acc;
    }
  }
  __syncthreads();

  // this wave's unit: rotated by the image so that the empty slots of the last workgroup of an image (U % 4 != 0) do not
  // always fall on the same SIMDs
  // (a workgroup covers WPG * upw consecutive units, wave w the units w, w + WPG, ...: one set-up for upw units per wave)
  int u = a.u_lo + WPG * a.upw * bw + ((wave + n) & (WPG - 1));
  if (u >= a.U) return;

  // The operand slots k = 4 s + lrow behind the patch (k >= L) sit in the last one or two sub-steps (ts = s - sL): the A side
  // (rows) carries (norm + log2 variance, 1) at k = L, L + 1, the B side (columns) (1, norm).  Per lane and tail sub-step:
  //   operand = v * t_real + norm * t_nrm + t_one       (v: the gathered element; two FMAs where a select chain was ten)
  const int sL = RES ? NK4 - (TL == 3 ? 2 : 1) : L >> 2;
  double tB_real[2], tB_nrm[2], tB_one[2];
#pragma unroll
  for (int ts = 0; ts < 2; ++ts) {
    const int k = 4 * (sL + ts) + lrow;
    tB_real[ts] = k < L ? 1.0 : 0.0;
    tB_one[ts] = k == L ? 1.0 : 0.0;
    tB_nrm[ts] = k == L + 1 ? 1.0 : 0.0;
  }
  int kob[NKR];   // RES: byte offsets of this lane's patch elements, all sub-steps (0 for the slots behind the patch)
  if (RES) {
#pragma unroll
    for (int s = 0; s < NKR; ++s) kob[s] = koff[4 * s + lrow];
  }
  auto fixB = [&](double v, int s, double nrm) {   // s >= sL (wave-uniform test at the call site)
    const int ts = s - sL;
    const double t0 = ts ? tB_real[1] : tB_real[0], t1 = ts ? tB_nrm[1] : tB_nrm[0], t2 = ts ? tB_one[1] : tB_one[0];
    return fma(v, t0, fma(nrm, t1, t2));
  };
  auto fixA = [&](double v, int s, double nrm) {   // the A side carries the two slots the other way round
    const int ts = s - sL;
    const double t0 = ts ? tB_real[1] : tB_real[0], t1 = ts ? tB_one[1] : tB_one[0], t2 = ts ? tB_nrm[1] : tB_nrm[0];
    return fma(v, t0, fma(nrm, t1, t2));
  };

  // NY column fragments starting at fragment j0 against one row fragment: product (operands of the next sub-step requested
  // before the MFMAs of the current one), then 2^t and the weighted row sums.  getA(s): the A operand of sub-step s.
  // `pre`: the caller has already put this group's patch offsets into pb and its sub-step-0 operands into bv (requested
  // before the previous group's epilogue); next_j0 >= 0: do the same for the group that follows.
  auto group = [&](auto ny_tag, auto&& getA, auto&& getA_raw, int j0, int next_j0, int nyn, double* rdiag, double (&rsum)[4], int (&pb)[4], double (&bv)[4]) {
    constexpr int NY = decltype(ny_tag)::value;
    d4 acc[NY];
#pragma unroll
    for (int y = 0; y < NY; ++y) acc[y] = d4{0.0, 0.0, 0.0, 0.0};
    auto xbv = [&](int y) { return xb[16 * (j0 + y) + lcol]; };
    if (RES) {
#pragma unroll
      for (int s = 0; s < NKR; ++s) {
        double bn[NY];
        if (s + 1 < NKR) {
#pragma unroll
          for (int y = 0; y < NY; ++y) bn[y] = ldi(pb[y] + kob[s + 1]);
        }
        HU_SB1;
        const double av = getA(s);
        if (s >= sL) {
#pragma unroll
          for (int y = 0; y < NY; ++y) bv[y] = fixB(bv[y], s, xbv(y));
        }
#pragma unroll
        for (int y = 0; y < NY; ++y) acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[y], acc[y], 0, 0, 0);
        if (s + 1 < NKR) {
#pragma unroll
          for (int y = 0; y < NY; ++y) bv[y] = bn[y];
        }
        __builtin_amdgcn_sched_barrier(0);   // one sub-step of prefetch, not all of them (the scheduler would hoist every gather: 56 registers)
      }
    } else {
      // Sub-steps [0, sL) hold patch elements only.  They go in chunks of 4: the A operands (global memory for Kzx) and the patch-element
      // offsets of chunk c + 1 are requested before the products of chunk c (16 MFMAs = 1024 cycles against ~600 of a global load),
      // and inside a chunk the B gathers run one sub-step ahead of their MFMAs -- no conditional anywhere in the chunk.
      // (the offsets run two chunks ahead, so that a Kdiag row's A operands -- gathered from the image through them -- are requested
      // with offsets that have long arrived: offset -> gather -> MFMA in one chunk stalled the wave for two LDS round trips per sub-step)
      double ac[4], an[4];
      int kc[4], kn[4], kf[4];
      auto ldk = [&](int s0, int (&K)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) K[q] = koff[4 * min(s0 + q, nk4 - 1) + lrow];
      };
      ldk(0, kc);
      ldk(4, kn);
#pragma unroll
      for (int q = 0; q < 4; ++q) ac[q] = getA_raw(min(q, nk4 - 1), kc[q]);
      int s = 0;
      for (; s + 4 <= sL; s += 4) {
        ldk(s + 8, kf);
#pragma unroll
        for (int q = 0; q < 4; ++q) an[q] = getA_raw(min(s + 4 + q, nk4 - 1), kn[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          double bn[NY];
          const int kon = q < 3 ? kc[q + 1] : kn[0];   // sub-step s + q + 1 <= sL exists
#pragma unroll
          for (int y = 0; y < NY; ++y) bn[y] = ldi(pb[y] + kon);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int y = 0; y < NY; ++y) acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[q], bv[y], acc[y], 0, 0, 0);
#pragma unroll
          for (int y = 0; y < NY; ++y) bv[y] = bn[y];
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { ac[q] = an[q]; kc[q] = kn[q]; kn[q] = kf[q]; }
      }
      // what is left: at most three sub-steps of patch elements and the one or two that carry the norm slots
      for (; s < nk4; ++s) {
        const double av = getA(s);
        double bn[NY];
        const int kon = koff[4 * min(s + 1, nk4 - 1) + lrow];
#pragma unroll
        for (int y = 0; y < NY; ++y) bn[y] = ldi(pb[y] + kon);
        if (s >= sL) {
#pragma unroll
          for (int y = 0; y < NY; ++y) bv[y] = fixB(bv[y], s, xbv(y));
        }
#pragma unroll
        for (int y = 0; y < NY; ++y) acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[y], acc[y], 0, 0, 0);
#pragma unroll
        for (int y = 0; y < NY; ++y) bv[y] = bn[y];
      }
    }
    // weights of this group's columns, then the next group's first operands on their way before the VALU-only epilogue
    double wc[NY];
#pragma unroll
    for (int y = 0; y < NY; ++y) wc[y] = WRITE ? 0.0 : wl[16 * (j0 + y) + lcol];
    if (nyn > 0) {
      const int ko0 = RES ? kob[0] : koff[lrow];
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        if (y < nyn) {
          pb[y] = pbl[16 * (next_j0 + y) + lcol];
          bv[y] = ldi(pb[y] + ko0);
        }
      }
    }
    constexpr int YE = (HU_EXPN == 8 && NY % 2 == 0) ? 2 : 1;   // fragments per batch of interleaved chains
#pragma unroll
    for (int y0 = 0; y0 < NY; y0 += YE) {
      double t[4 * YE];
#pragma unroll
      for (int y = 0; y < YE; ++y)
#pragma unroll
        for (int v = 0; v < 4; ++v) t[4 * y + v] = acc[y0 + y][v];
      exp2_n<4 * YE>(t);
      if (WRITE) {   // rows m = 16 u + lrow + 4 v, 16 consecutive patches per row: 128-byte segments when sP == 1
#pragma unroll
        for (int y = 0; y < YE; ++y) {
          const int pp = 16 * (j0 + y0 + y) + lcol;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int m = 16 * u + lrow + 4 * v;
            if (m < a.kzx_rows && pp < P) a.kuf[(long)m * a.sM + (long)n * a.sN + (long)pp * a.sP] = m < a.M ? t[4 * y + v] : 0.0;
          }
        }
        continue;
      }
#pragma unroll
      for (int y = 0; y < YE; ++y) {
#pragma unroll
        for (int v = 0; v < 4; ++v) rsum[v] = fma(wc[y0 + y], t[4 * y + v], rsum[v]);
        if (y0 + y == 0 && rdiag) {   // first fragment of a Kdiag row pass = the diagonal tile: its share, counted once
#pragma unroll
          for (int v = 0; v < 4; ++v) rdiag[v] = rsum[v];
        }
      }
    }
  };
  using T1 = std::integral_constant<int, 1>;
  using T2 = std::integral_constant<int, 2>;
  using T3 = std::integral_constant<int, 3>;
  using T4 = std::integral_constant<int, 4>;

  // one row fragment against column fragments [j_lo, nfp): groups of four, then one group of the remaining 1..3.
  // rdiag != nullptr: receives the share of the first fragment (the diagonal tile of a Kdiag row; rsum must start at zero)
  auto row_pass = [&](auto&& getA, auto&& getA_raw, int j_lo, double* rdiag, double (&rsum)[4]) {
    const int nfull = (nfp - j_lo) >> 2, nrem = (nfp - j_lo) & 3;
    int pb[4];
    double bv[4];
    const int ko0 = RES ? kob[0] : koff[lrow];
    int j0 = j_lo;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      if (y < (nfull ? 4 : nrem)) { pb[y] = pbl[16 * (j0 + y) + lcol]; bv[y] = ldi(pb[y] + ko0); }
    }
    for (int g = 0; g < nfull; ++g, j0 += 4) group(T4{}, getA, getA_raw, j0, j0 + 4, g + 1 < nfull ? 4 : nrem, g == 0 ? rdiag : nullptr, rsum, pb, bv);
    double* rd = nfull == 0 ? rdiag : nullptr;
    if (nrem == 1) group(T1{}, getA, getA_raw, j0, -1, 0, rd, rsum, pb, bv);
    else if (nrem == 2) group(T2{}, getA, getA_raw, j0, -1, 0, rd, rsum, pb, bv);
    else if (nrem == 3) group(T3{}, getA, getA_raw, j0, -1, 0, rd, rsum, pb, bv);
  };

  for (int uu = 0; uu < a.upw && u < a.U; ++uu, u += WPG) {
  if (u < a.nfm) {
    // ---- Kzx rows 16 u .. 16 u + 15: out[m][n] = scale * sum_p w_p k(z_m, x_p) ----
    const double* __restrict__ zs = a.ZS + 16 * u + lcol;
    double rsum[4] = {0.0, 0.0, 0.0, 0.0};
    if (RES) {
      double areg[NKR];
#pragma unroll
      for (int s = 0; s < NKR; ++s) areg[s] = zs[(long)(4 * s + lrow) * a.Mp];
      row_pass([&](int s) { return areg[s]; }, [&](int s, int) { return areg[RES ? s : 0]; }, 0, nullptr, rsum);
    } else {
      row_pass([&](int s) { return zs[(long)(4 * s + lrow) * a.Mp]; }, [&](int s, int) { return zs[(long)(4 * s + lrow) * a.Mp]; }, 0, nullptr, rsum);
    }
    if (WRITE) continue;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      double s = rsum[v];
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      s += __shfl_xor(s, 8);
      const int m = 16 * u + lrow + 4 * v;
      if (lcol == 0 && m < a.kzx_rows) a.kzx[(long)m * a.ldk + n] = m < a.M ? a.kzx_scale * s : 0.0;
    }
  } else {
    // ---- Kdiag: fragment rows i and nfp - 1 - i of the patch Gram matrix, tiles on and right of the diagonal ----
    const int i = u - a.nfm;
    double total = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
      const int fr = pass == 0 ? i : nfp - 1 - i;
      if (pass == 1 && fr <= i) break;
      const int pr = 16 * fr + lcol;
      const int pa = pbl[pr];
      const double xav = xb[pr] + a.log2var;
      double rsum[4] = {0.0, 0.0, 0.0, 0.0}, rdiag[4] = {0.0, 0.0, 0.0, 0.0};
      auto getA_img = [&](int s, int ko) {
        double v = ldi(pa + ko);
        if (s >= sL) v = fixA(v, s, xav);
        return v;
      };
      if (RES) {
        double areg[NKR];
#pragma unroll
        for (int s = 0; s < NKR; ++s) areg[s] = getA_img(s, kob[s]);
        row_pass([&](int s) { return areg[s]; }, [&](int s, int) { return areg[RES ? s : 0]; }, fr, rdiag, rsum);
      } else {
        row_pass([&](int s) { return getA_img(s, koff[4 * s + lrow]); }, [&](int, int ko) { return ldi(pa + ko); }, fr, rdiag, rsum);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) total = fma(wl[16 * fr + lrow + 4 * v], 2.0 * rsum[v] - rdiag[v], total);   // off-diagonal tiles count twice
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) total += __shfl_xor(total, o);
    if (lane == 0) a.kd[(long)n * a.n_kd + i] = total;
  }
  }
}

}  // namespace

static size_t head_units_lds(const HeadUnitsArgs& a) {
  return (size_t)(((a.HWC + 1) & ~1) + 2 * a.nfp * 16 + ((a.H * (a.W - a.f + 1) + 1) & ~1)) * sizeof(double) +
         (size_t)(a.nfp * 16 + a.Lq) * sizeof(int);
}

bool head_units_ok(const HeadUnitsArgs& a) {
  return head_units_lds(a) <= 54 * 1024 && (long)a.Lq * a.Mp * 8 < (1L << 31);
}

// fills the derived fields of `a` (fragment counts, units per image)
void head_units_plan(HeadUnitsArgs* a) {
  a->HWC = a->H * a->W * a->C;
  a->nfm = a->Mp / 16;
  a->nfp = (a->P + 15) / 16;
  a->n_kd = (a->nfp + 1) / 2;
  a->U = (a->kd && !a->kuf) ? a->nfm + a->n_kd : a->nfm;     // no Kdiag output (or the K_uf sweep): the row units only
  a->u_lo = (a->kzx || a->kuf) ? 0 : a->nfm;                 // no Kzx output: the Kdiag units only
  // units per wave: the set-up of a workgroup (image, window sums, tables: ~3 us of latency) is as long as a short unit (a 16-row
  // fragment against the 9 patch fragments of a 12 x 12 view), so such launches put several units behind one set-up -- as many as
  // leave >= 512 workgroups (two per CU), and never a count between one and two rounds of the 1024 resident slots
  // waves per workgroup: 4 where the patch fits registers (one set-up for four long units); 2 for long patches on small views
  // (a 12 x 12 x 10 head input: 18 units of 252 MFMAs per image -- 1600 four-wave workgroups are 1.56 rounds of the resident
  // slots, 2880 two-wave ones 1.4, and with no empty unit slot: 84 -> 79 us)
  a->wpg = (a->L == 25 || a->kuf || a->nfp > 8) ? 4 : 2;
  const int HU_WPG = a->wpg;
  const int nu = a->U - a->u_lo;
  a->upw = 1;
  if (a->upw_force > 0) a->upw = a->upw_force;
  else
    for (int k = 4; k > 1; k >>= 1) {
      const long nwg = (long)a->N * ((nu + HU_WPG * k - 1) / (HU_WPG * k));
      if (a->nfp <= 16 && nwg >= 512 && nu % (HU_WPG * k) == 0) { a->upw = k; break; }
    }
  a->wgs_per_img = (nu + HU_WPG * a->upw - 1) / (HU_WPG * a->upw);
  if (a->kzx_rows <= 0) a->kzx_rows = a->Mp;
  a->inv_C = 1.0f / (float)a->C; a->inv_f = 1.0f / (float)a->f; a->inv_Wo = 1.0f / (float)a->Wo; a->inv_Wr = 1.0f / (float)(a->W - a->f + 1);
}

int head_units(dcgp_ctx* ctx, const HeadUnitsArgs& a) {
  if (a.N <= 0) return DCGP_OK;
  if (!head_units_ok(a) || a.n_mod <= 0 || a.Lq != round_up(a.L + 2, 4) || a.Mp % 16)
    return ctx_fail(ctx, DCGP_ERR_ARG, "head_units: unsupported shape (image %d doubles, L = %d, Mp = %d)", a.HWC, a.L, a.Mp);
  const long nwg = (long)a.N * a.wgs_per_img;
  if (nwg > 0x7fffffffL) return ctx_fail(ctx, DCGP_ERR_ARG, "head_units: too many workgroups");
  size_t lds = head_units_lds(a);
  // Beside the factorisation chain (a head-first model: the sweep needs Z only): a chain workgroup is one wave of 250 VGPRs per SIMD
  // and 50 KB of LDS, and would never find that much free at once on a CU this launch keeps refilling with four 128-register
  // workgroups.  Claiming 54 KB per workgroup holds the sweep to two per CU (half the register file stays free; 2 waves per SIMD
  // cost it ~3 %) -- the chain's workgroups then start the moment they are launched.
  if (a.share_cu && lds < 54 * 1024) lds = 54 * 1024;
  ScopedTimer t(ctx, a.kuf ? "kuf" : "head_sweep");
  if (a.kuf) {
    if (a.L == 25) hipLaunchKernelGGL((head_units_kernel<7, 1, true, 256>), dim3((unsigned)nwg), dim3(256), lds, ctx->stream, a);
    else hipLaunchKernelGGL((head_units_kernel<0, 0, true, 256>), dim3((unsigned)nwg), dim3(256), lds, ctx->stream, a);
  } else if (a.L == 25) {
    hipLaunchKernelGGL((head_units_kernel<7, 1, false, 256>), dim3((unsigned)nwg), dim3(256), lds, ctx->stream, a);   // 5 x 5 x 1 patches
  } else if (a.wpg == 4) {
    hipLaunchKernelGGL((head_units_kernel<0, 0, false, 256>), dim3((unsigned)nwg), dim3(256), lds, ctx->stream, a);
  } else {
    hipLaunchKernelGGL((head_units_kernel<0, 0, false, 128>), dim3((unsigned)nwg), dim3(128), lds, ctx->stream, a);
  }
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}
