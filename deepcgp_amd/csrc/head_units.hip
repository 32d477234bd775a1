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
#include <cmath>
#include <type_traits>

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int kOob = (int)0x80000000u;   // a per-lane buffer offset past every descriptor's num_records (<= 0x7fffffff): the store is dropped

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
// (kuf[m * sM + n * sN + p * sP]) instead of reduced; no Kdiag units.  A pure store kernel reaches 5.3-6.0 TB/s on this part in exactly
// this tile pattern (tools/store_bw.hip: four 128-byte segments per instruction, rows sM apart, misaligned P included), so what the
// sweep must not do is spend issue slots beside its stores:
//   * stores go through ONE buffer descriptor per unit: per-lane offsets (row lrow + 4 v, column lcol) computed once per unit, the
//     fragment / replica part a scalar offset -- no 64-bit address arithmetic per value; lanes outside the matrix (rows >= kzx_rows,
//     patches >= P) carry an out-of-range offset and are dropped by the bounds check;
//   * rows that show the SAME image (propagate() tiles the batch S times: row n shows image (n0 + n) % n_mod) get the same values:
//     a unit evaluates its tiles once and stores them to every such row (a.n_base < a.N).
// WMODE 3: the reducing form that also stores every kernel value of its Kzx units (a.kfull: the head of a training step).
// WMODE 0: the reducing form; 1: the storing form, every tile stored as it is evaluated; 2: the storing form that can also hold a batch
// of tiles for replica-outer stores (row_pass_hold).  The storing forms run at three / two waves per SIMD (168 / 256 registers, no
// spill): their stores and a spill reload share the wave's in-order memory counter, so a single reload in the tile loop waits for
// every store issued before it -- the store queue drained once per tile.
// The arguments are read through the kernarg pointer (constant address space: scalar loads where they are used) instead of as a by-value struct, whose
// ~90 words the compiler loads up front and then spills (10-76 scalar spills per instantiation, 0-44 this way -- each one a v_writelane / v_readlane, VALU
// instructions in the sweep's issue slots): cfg5's storing sweep 286 -> 249 us, the others 0-3 % (profiles/r06_sweep_fetch_ab.txt).
// (Round 6 also tried the reducing form as a persistent launch fetching workgroup indices from a device counter, now that the argument spills of round 2's
// attempt are gone: slower everywhere -- cfg2 conv + head +27 us, cfg1 +26 -- a finished wave of a persistent workgroup idles until its workgroup is done,
// in the plain launch its slot goes to the next workgroup at once.)
template <int NK4, int TL, int WMODE, int NT>
// (aligned(4096): the streamed forms are 40-50 KB of mostly straight-line code and their speed depends on where the code object puts them -- the same binary of the
// 5 x 5 x 10 head's sweep measured 70.7-71.9 us at one 256-byte-aligned address and 67.7-68.2 at another; page-aligned it is the latter wherever it lands)
__global__ __attribute__((aligned(4096))) __launch_bounds__(NT, WMODE == 2 ? 2 : ((WMODE == 1 || WMODE == 3) ? 3 : HU_WAVES)) void head_units_kernel(HeadUnitsArgs a_in) {
  const __attribute__((address_space(4))) HeadUnitsArgs& a = *(const __attribute__((address_space(4))) HeadUnitsArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  constexpr bool WRITE = WMODE == 1 || WMODE == 2;
  constexpr bool KEEP = WMODE == 3;
  constexpr int WPG = NT / 64;   // units (waves) per workgroup
  constexpr bool RES = NK4 > 0;
  constexpr int NKR = RES ? NK4 : 1;
  // Streamed form with TL > 0 (<0, RW, ...>): the patch is f rows of RW = f C contiguous image elements, RW = 4 n + 2 and f odd (5 x 5 x 10: RW = 50, L = 250 --
  // every long patch of the BASELINE configurations).  Lane group lrow's element of sub-step s, k = 4 s + lrow, then walks a patch row at +32 bytes per sub-step:
  // the gathers of a row's n aligned sub-steps are `ds_read_b64 ... offset: 32 i` on ONE per-lane address per fragment and row, two rows and the sub-step that
  // straddles them are a period of 2 n + 1 sub-steps, and the loop is straight-line code per period.  The generic streamed loop spends 33 VALU instructions per 16
  // MFMAs on its operands' way (16 gather addresses, 4 offset-table addresses, 11 ring moves: ISA count) and issues at ~75 % of the MFMA rate with four waves per
  // SIMD (profiles/r06_head_packed_trace.txt); this one ~3 per period of 100.
  constexpr int RW = RES ? 0 : TL;
  // Register-resident forms: the k slots are dealt so that lane group lrow walks RL CONTIGUOUS patch elements RL lrow .. RL lrow + RL - 1 in
  // sub-steps 0 .. RL - 1 (L = 25: RL = 5, f * C is 5 or 25; L = 16: RL = 4; L = 48: RL = 12 -- RL consecutive elements never straddle a patch
  // row), then what is left in the old order (L = 25: elements 20 + lrow in sub-step 5, 24 and the norm slots in sub-step 6).  The gathers of
  // sub-steps 1 .. RL - 1 are then `ds_read_b64 ... offset: 8 s` on the address of sub-step 0 -- no address arithmetic (L = 25: 16 of the ~90
  // VALU instructions of a tile).  The Z operand is read in the same order (ldz below).
  constexpr int RL = (NK4 == 7 && TL == 1) ? 5 : ((NK4 == 5 && TL == 0) ? 4 : ((NK4 == 13 && TL == 0) ? 12 : 0));
  constexpr bool PERM = RL > 0;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int HWC = a.HWC, L = a.L, nk4 = RES ? NK4 : a.Lq >> 2, nfp = a.nfp, P = a.P, np16 = nfp * 16;
  const int HWCe = (HWC + 1) & ~1;
  double* img = smem;                                   // [HWC] image times sqrt(c)
  double* xb = img + HWCe;                              // [np16]  -c |x_p|^2 / 2
  double* wl = xb + np16;                               // [np16]  patch weights, 0 beyond P
  double* rs = wl + np16;                               // [H * Wr] row sums of squares (set-up only)
  int* pbl = reinterpret_cast<int*>(rs + ((a.H * (a.W - a.f + 1) + 1) & ~1));   // [np16]  byte offset of the patch's first element in img
  int* koff = pbl + np16;                               // [Lq]    byte offset of patch element l
  double* etab = reinterpret_cast<double*>(koff + ((a.Lq + 1) & ~1));   // [256]  2^(j / 256) (exp2_tab_n)
  const char* imgb = reinterpret_cast<const char*>(img);
  const int tid = threadIdx.x, lane = tid & 63, lrow = lane >> 4, lcol = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // this workgroup's segment of the launch (HuSeg, common.h), its image and its place among the image's workgroups
  int sg = 0;
#pragma unroll
  for (int q = 1; q < 6; ++q)
    if (q < a.nseg && (int)blockIdx.x >= a.seg[q].wg0) sg = q;
  const int seg_kind = a.seg[sg].kind, seg_T = a.seg[sg].T, seg_C = a.seg[sg].C;
  const int wloc = (int)blockIdx.x - a.seg[sg].wg0, wpi = a.seg[sg].wpi;
  const int nloc = wloc / wpi, bw = wloc - nloc * wpi;
  const int n = a.seg[sg].img0 + nloc;   // storing form: n < a.n_base
  const double* __restrict__ Xn = a.X + (long)((a.n0 + n) % a.n_mod) * HWC;
  auto ldi = [&](int byte_off) { return *reinterpret_cast<const double*>(imgb + byte_off); };
  // stamps for tools/sweep_trace.py (a.trace == nullptr in normal use: one scalar branch each).  [0] wall clock (100 MHz) at entry,
  // [1] shader clock at entry, [2] image in LDS, [3] set-up done, [4] first unit done, [5] last unit done, [6] wall clock at exit, [7] units run
  long long* const tr = (a.trace && blockIdx.x < a.trace_wgs) ? a.trace + ((long)blockIdx.x * WPG + wave) * 8 : nullptr;
  auto stamp = [&](int k) { if (tr && lane == 0) tr[k] = (long long)clock64(); };
  if (tr && lane == 0) {   // [7]: units run in the low word, (XCC_ID << 16 | HW_ID[15:0]) -- where the wave sits -- in the high word
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    tr[0] = (long long)wall_clock64();
    tr[7] = (long long)(((xcc & 0xf) << 16) | (hwid & 0xffff)) << 32;
  }
  stamp(1);

  // ---- set-up, once per workgroup: the scaled image, the offset tables, patch norms from a separable window sum ----
  // The image's loads go out first; the tables, which need no pixel, are built under their latency.  (A trace of every workgroup,
  // tools/sweep_trace.py, showed the set-up at 4.9 us of a 256-thread workgroup's ~20 and 12.5 us of a one-wave workgroup's 26 before
  // this was reordered and the window sums were given one thread per entry: four threads per entry with two shuffles each is a dependent
  // chain per iteration, and a 64-thread workgroup walked 42 of them.)
  {
    // batches of IB loads per thread: one memory latency for all of them (16 for the narrow workgroups, so that a 12 x 12 x 10 image is one
    // batch of a two-wave workgroup as well: its second batch was a second full latency, ~1.5 us of a ~6 us set-up)
    constexpr int IB = NT <= 128 ? 16 : 8;
    double t[IB];
#pragma unroll
    for (int e = 0; e < IB; ++e) {
      const int i = e * NT + tid;
      t[e] = (i < HWC) ? Xn[i] : 0.0;
    }
    if (RES)
      for (int i = tid; i < 256; i += NT) etab[i] = a.exp_tab[i];
    for (int l = tid; l < a.Lq; l += NT) {
      const int ll = l < L ? l : 0;
      const int tq = fdiv_small(ll, a.inv_C), c = ll - tq * a.C;
      const int kh = fdiv_small(tq, a.inv_f), kw = tq - kh * a.f;
      koff[l] = ((kh * a.W + kw) * a.C + c) * 8;
    }
    for (int p = tid; p < np16; p += NT) {
      const int q = p < P ? p : 0;                 // patches beyond P repeat the first one (finite values, weight 0)
      const int oh = fdiv_small(q, a.inv_Wo), ow = q - oh * a.Wo;
      pbl[p] = (oh * a.s * a.W + ow * a.s) * a.C * 8;
      wl[p] = (!WRITE && p < P) ? a.w[p] : 0.0;
    }
#pragma unroll
    for (int e = 0; e < IB; ++e) {
      const int i = e * NT + tid;
      if (i < HWC) img[i] = t[e] * a.csq;
    }
    for (int i0 = IB * NT; i0 < HWC; i0 += IB * NT) {
#pragma unroll
      for (int e = 0; e < IB; ++e) {
        const int i = i0 + e * NT + tid;
        t[e] = (i < HWC) ? Xn[i] : 0.0;
      }
#pragma unroll
      for (int e = 0; e < IB; ++e) {
        const int i = i0 + e * NT + tid;
        if (i < HWC) img[i] = t[e] * a.csq;
      }
    }
  }
  __syncthreads();
  stamp(2);
  {
    // rs[r][x] = sum of squares over the f*C contiguous elements of image row r that a patch starting at column x covers: one thread
    // per entry, its reads independent of each other (four in flight per step)
    const int Wr = a.W - a.f + 1, fC = a.f * a.C;
    for (int i = tid; i < a.H * Wr; i += NT) {
      const int r = fdiv_small(i, a.inv_Wr), x = i - r * Wr;
      const double* src = img + (r * a.W + x) * a.C;
      double acc = 0.0;
      int j = 0;
      for (; j + 4 <= fC; j += 4) {
        const double v0 = src[j], v1 = src[j + 1], v2 = src[j + 2], v3 = src[j + 3];
        acc = fma(v0, v0, acc); acc = fma(v1, v1, acc); acc = fma(v2, v2, acc); acc = fma(v3, v3, acc);
      }
      for (; j < fC; ++j) acc = fma(src[j], src[j], acc);
      rs[i] = acc;
    }
    __syncthreads();
    for (int p = tid; p < np16; p += NT) {
      const int q = p < P ? p : 0;
      const int oh = fdiv_small(q, a.inv_Wo), ow = q - oh * a.Wo;
      const double* src = rs + oh * a.s * Wr + ow * a.s;
      double acc = 0.0;
      int kh = 0;
      for (; kh + 4 <= a.f; kh += 4) {
        const double v0 = src[kh * Wr], v1 = src[(kh + 1) * Wr], v2 = src[(kh + 2) * Wr], v3 = src[(kh + 3) * Wr];
        acc += (v0 + v1) + (v2 + v3);
      }
      for (; kh < a.f; ++kh) acc += src[kh * Wr];
      xb[p] = -0.5 * acc;
    }
  }
  __syncthreads();

  // this wave's unit: rotated by the image so that the empty slots of the last workgroup of an image (U % 4 != 0) do not
  // always fall on the same SIMDs
  // (a workgroup covers WPG * upw consecutive units, wave w the units w, w + WPG, ...: one set-up for upw units per wave)
  stamp(3);
  const int seg_upw = a.seg[sg].upw;
  int u = WPG * seg_upw * bw + ((wave + n) & (WPG - 1));
  const int n_units = seg_kind == 1 ? seg_C : (WRITE ? a.nfm * a.st_split : a.nfm);
  if (u >= n_units) {
    if (tr && lane == 0) tr[6] = (long long)wall_clock64();
    return;
  }

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
    for (int s = 0; s < NKR; ++s) kob[s] = (PERM && s < RL) ? koff[RL * lrow] + 8 * s : koff[4 * s + lrow];
  }
  int kdel[2] = {0, 0};   // PERM: the one or two sub-steps behind the runs, relative to sub-step 0
  if (PERM) {
#pragma unroll
    for (int q = 0; q < 2; ++q) kdel[q] = RL + q < NKR ? kob[RL + q < NKR ? RL + q : 0] - kob[0] : 0;
  }
  static_assert(!PERM || NKR - RL <= 2, "at most two sub-steps behind the runs");
  // pbx(fragment): the LDS byte address of this lane's sub-step-0 element of its patch (PERM: kob[0] folded in); ld0 / ldB: the gathers
  auto pbx = [&](int frag) { const int p0 = pbl[16 * frag + lcol]; return PERM ? p0 + kob[0] : p0; };
  auto ld0 = [&](int pbv) { return PERM ? ldi(pbv) : ldi(pbv + (RES ? kob[0] : koff[lrow])); };
  auto ldB = [&](int pbv, int s) {   // RES only; s is a compile-time constant after unrolling
    if (PERM) return s < RL ? ldi(pbv + 8 * s) : ldi(pbv + kdel[s - RL < 1 ? 0 : 1]);
    return ldi(pbv + kob[RES ? s : 0]);
  };
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

  // storing form: the unit's buffer descriptor (base: row 16 u of base row n), its per-lane offsets and its replica count
  __amdgpu_buffer_rsrc_t st_rs = __builtin_amdgcn_make_buffer_rsrc(a.kuf, 0, 0, 0x00020000);
  int st_voff[4] = {kOob, kOob, kOob, kOob};
  bool st_zero = false;
  const int st_nrep = (WRITE && a.n_base < a.N) ? (a.N - 1 - n) / a.n_mod + 1 : 1;
  const bool st_hold = WMODE == 2 && st_nrep > 1 && a.st_hold;
  int ur = 0;   // the unit's row fragment (storing form: a unit is a row fragment x one of a.st_split ranges of column fragments)

  // NY column fragments starting at fragment j0 against one row fragment: product (operands of the next sub-step requested
  // before the MFMAs of the current one), then 2^t and the weighted row sums.  getA(s): the A operand of sub-step s.
  // `pre`: the caller has already put this group's patch offsets into pb and its sub-step-0 operands into bv (requested
  // before the previous group's epilogue); next_j0 >= 0: do the same for the group that follows.
  int kd_pa = 0;   // a row pass of the patch Gram matrix (kd_tag true): LDS byte offset of the row patch's first element (the A operand is gathered like the B ones)
  auto group = [&](auto kd_tag, auto ny_tag, auto&& getA, auto&& getA_raw, int j0, int next_j0, int nyn, double* rdiag, double (&rsum)[4], int (&pb)[4], double (&bv)[4]) {
    constexpr int NY = decltype(ny_tag)::value;
    constexpr bool KD = decltype(kd_tag)::value;
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
          for (int y = 0; y < NY; ++y) bn[y] = ldB(pb[y], s + 1);
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
    } else if constexpr (RW > 0) {
      static_assert(RW % 4 == 2, "patch rows of 4 n + 2 elements: two rows and the sub-step that straddles them are a period of 2 n + 1 sub-steps");
      constexpr int NA = RW / 4, DA = 5;        // aligned sub-steps per patch row; A-operand sub-steps in flight (Kzx: from global memory)
      static_assert((2 * NA + 1) % DA == 0, "the ring of A operands closes over a period");
      const int RS = a.W * a.C * 8;             // bytes from one patch row to the next in the image
      const int sdl = lrow < 2 ? NA * 32 : RS - 16;   // the straddling sub-step: lane groups 0, 1 end the row, 2, 3 open the next one
      int pA[NY], pS[NY], pB[NY];
#pragma unroll
      for (int y = 0; y < NY; ++y) pA[y] = pb[y] + lrow * 8;
      int qA = KD ? kd_pa + lrow * 8 : 0, qS = 0, qB = 0;
      double aq[DA], akd = 0.0;
      int sc = 0;                               // sub-steps done (the Kzx A operand's scalar offset)
      if (KD) {
        akd = ldi(qA);
      } else {
#pragma unroll
        for (int u = 0; u < DA; ++u) aq[u] = getA_raw(min(u, nk4 - 1), 0);
      }
      // one sub-step: the gathers of the NEXT one (addresses nx / qn, offset off) go out before this one's MFMAs
      auto one = [&](int SL, const int (&nx)[NY], int qn, int off, bool more) __attribute__((always_inline)) {   // (SL, off: constants once the callers' loops are unrolled)
        double bn[NY], an = 0.0;
        if (more) {
#pragma unroll
          for (int y = 0; y < NY; ++y) bn[y] = ldi(nx[y] + off);
          if (KD) an = ldi(qn + off);
        }
        __builtin_amdgcn_sched_barrier(0);
        const double av = KD ? akd : aq[SL];
#pragma unroll
        for (int y = 0; y < NY; ++y) acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[y], acc[y], 0, 0, 0);
        if (more) {
#pragma unroll
          for (int y = 0; y < NY; ++y) bv[y] = bn[y];
        }
        if (KD) akd = an;
        else aq[SL] = getA_raw(min(sc + DA, nk4 - 1), 0);
        ++sc;
        __builtin_amdgcn_sched_barrier(0);
      };
      // the NA aligned sub-steps of a patch row at `base` (BLK: the ring slot of its first sub-step); `after`: what follows the row
      auto row_block = [&](int BLK, const int (&base)[NY], int qbase, const int (&after)[NY], int qafter) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          if (i + 1 < NA) one((BLK + i) % DA, base, qbase, 32 * (i + 1), true);
          else one((BLK + i) % DA, after, qafter, 0, true);
        }
      };
      const int nper = (a.f - 1) >> 1;
      for (int t = 0; t < nper; ++t) {
#pragma unroll
        for (int y = 0; y < NY; ++y) { pS[y] = pA[y] + sdl; pB[y] = pA[y] + (RS + 16); }
        if (KD) { qS = qA + sdl; qB = qA + (RS + 16); }
        row_block(0, pA, qA, pS, qS);
        one(NA % DA, pB, qB, 0, true);               // the straddling sub-step
#pragma unroll
        for (int y = 0; y < NY; ++y) pA[y] += 2 * RS;
        if (KD) qA += 2 * RS;
        row_block((NA + 1) % DA, pB, qB, pA, qA);
      }
      // the last row and the sub-step that carries its last two elements and the two norm slots (lane groups 2, 3 gather the patch's first element: finite, unused)
      // (the last row as a pass of the loop above -- one copy of the row block's code, a third less of it -- measured 66 -> 77-80 us at the 12 x 12 x 10 head: the
      // selects and the early exit cost the schedule more than the instruction cache gains)
#pragma unroll
      for (int y = 0; y < NY; ++y) pS[y] = lrow < 2 ? pA[y] + NA * 32 : pb[y];
      row_block(0, pA, qA, pS, qA);
      {
        const int sl = nk4 - 1;
        const double av = KD ? getA(sl) : aq[NA % DA];
#pragma unroll
        for (int y = 0; y < NY; ++y) bv[y] = fixB(bv[y], sl, xbv(y));
#pragma unroll
        for (int y = 0; y < NY; ++y) acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[y], acc[y], 0, 0, 0);
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
#ifdef HU_ABL_NOB
          for (int y = 0; y < NY; ++y) bn[y] = (double)(lane + y + kon) * 1e-3;   // timing experiment (wrong results): no B gathers
#else
          for (int y = 0; y < NY; ++y) bn[y] = ldi(pb[y] + kon);
#endif
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
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        if (y < nyn) {
          pb[y] = pbx(next_j0 + y);
          bv[y] = ld0(pb[y]);
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
      // short patches (register-resident forms): the epilogue is most of a tile's issue slots -- the table form of 2^t (13 VALU instructions + one LDS read per
      // value against 16: MNIST head 114.6 -> ~110 us at M = 32, cfg5 head-only 1.923 -> 1.878 ms); long patches are bound by their MFMAs and keep the polynomial
      // (the table's reads and its staging cost the 12 x 12 x 10 head's sweep 3 us)
#ifdef HU_EXP_POLY
      exp2_n<4 * YE>(t);
#else
      if constexpr (RES) exp2_tab_n<4 * YE>(t, etab);
      else exp2_n<4 * YE>(t);
#endif
      if (WRITE) {   // rows m = 16 u + lrow + 4 v, 16 consecutive patches per row: 128-byte segments when sP == 1
#pragma unroll
        for (int y = 0; y < YE; ++y) {
          const int j = j0 + y0 + y;
          int vo[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) vo[v] = st_voff[v];
          if (j == nfp - 1) {   // the ragged last fragment: patches >= P are dropped
#pragma unroll
            for (int v = 0; v < 4; ++v) vo[v] = (16 * j + lcol < P) ? vo[v] : kOob;
          }
          if (st_zero) {        // the fragment that holds the padded rows M .. kzx_rows - 1: zeros
#pragma unroll
            for (int v = 0; v < 4; ++v) t[4 * y + v] = (16 * ur + lrow + 4 * v < a.M) ? t[4 * y + v] : 0.0;
          }
          if (st_hold) {        // several replicas: the values wait in `rdiag` (row_pass_hold's batch of tiles) for the replica-outer stores
#pragma unroll
            for (int v = 0; v < 4; ++v) rdiag[4 * (y0 + y) + v] = t[4 * y + v];
            continue;
          }
          int so = j * a.st_jb;
          for (int r = 0; r < st_nrep; ++r, so += a.st_rb) {
#pragma unroll
            for (int v = 0; v < 4; ++v)
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, t[4 * y + v]), st_rs, vo[v], so, 0);
          }
        }
        continue;
      }
      if (KEEP && seg_kind != 1) {   // Kzx unit of a training step: the values go out as well (128-byte segments, a fragment = 16 consecutive patches)
#pragma unroll
        for (int y = 0; y < YE; ++y) {
          const int j = j0 + y0 + y;
          const bool in = j < nfp - 1 || 16 * j + lcol < P;   // the ragged last fragment: patches >= P are dropped
#pragma unroll
          for (int v = 0; v < 4; ++v)
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, t[4 * y + v]), st_rs, in ? st_voff[v] : kOob, j * 128, 0);
        }
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

  // one row fragment against column fragments [j_lo, j_hi): groups of four, then one group of the remaining 1..3.
  // rdiag != nullptr: receives the share of the first fragment (the diagonal tile of a Kdiag row; rsum must start at zero)
  auto row_pass = [&](auto kd_tag, auto&& getA, auto&& getA_raw, int j_lo, int j_hi, double* rdiag, double (&rsum)[4]) {
    const int nfull = (j_hi - j_lo) >> 2, nrem = (j_hi - j_lo) & 3;
    int pb[4];
    double bv[4];
    int j0 = j_lo;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      if (y < (nfull ? 4 : nrem)) { pb[y] = pbx(j0 + y); bv[y] = ld0(pb[y]); }
    }
    for (int g = 0; g < nfull; ++g, j0 += 4) group(kd_tag, T4{}, getA, getA_raw, j0, j0 + 4, g + 1 < nfull ? 4 : nrem, g == 0 ? rdiag : nullptr, rsum, pb, bv);
    double* rd = nfull == 0 ? rdiag : nullptr;
    if (nrem == 1) group(kd_tag, T1{}, getA, getA_raw, j0, -1, 0, rd, rsum, pb, bv);
    else if (nrem == 2) group(kd_tag, T2{}, getA, getA_raw, j0, -1, 0, rd, rsum, pb, bv);
    else if (nrem == 3) group(kd_tag, T3{}, getA, getA_raw, j0, -1, 0, rd, rsum, pb, bv);
  };

  // storing form with replicas: the row fragment's tiles in batches of up to 8, each batch evaluated into registers and then stored
  // replica by replica, fragment by fragment within a replica -- a row receives 1 KB contiguously.  Tile by tile (each tile to all its
  // replicas before the next one) a row's 128-byte segments arrive ~one tile time apart, and where P is not a multiple of 16 (13 x 13,
  // 15 x 15 views) every segment straddles two cache lines whose halves are then written back separately: 4.0 instead of 6.0 TB/s in a
  // pure store kernel (tools/store_bw.hip, tile_rep_jr against tile_rep).
  auto row_pass_hold = [&](auto&& getA, auto&& getA_raw, int j_lo, int j_hi) {
    double dummy[4] = {0.0, 0.0, 0.0, 0.0};
    for (int jb = j_lo; jb < j_hi; jb += 8) {
      const int nb = min(8, j_hi - jb), n0 = min(nb, 4), n1 = nb - n0;
      double keep[32];
      int pb[4];
      double bv[4];
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        if (y < n0) { pb[y] = pbx(jb + y); bv[y] = ld0(pb[y]); }
      }
      if (n0 == 4) group(std::false_type{}, T4{}, getA, getA_raw, jb, jb + 4, n1, keep, dummy, pb, bv);
      else if (n0 == 3) group(std::false_type{}, T3{}, getA, getA_raw, jb, -1, 0, keep, dummy, pb, bv);
      else if (n0 == 2) group(std::false_type{}, T2{}, getA, getA_raw, jb, -1, 0, keep, dummy, pb, bv);
      else group(std::false_type{}, T1{}, getA, getA_raw, jb, -1, 0, keep, dummy, pb, bv);
      if (n1 == 4) group(std::false_type{}, T4{}, getA, getA_raw, jb + 4, -1, 0, keep + 16, dummy, pb, bv);
      else if (n1 == 3) group(std::false_type{}, T3{}, getA, getA_raw, jb + 4, -1, 0, keep + 16, dummy, pb, bv);
      else if (n1 == 2) group(std::false_type{}, T2{}, getA, getA_raw, jb + 4, -1, 0, keep + 16, dummy, pb, bv);
      else if (n1 == 1) group(std::false_type{}, T1{}, getA, getA_raw, jb + 4, -1, 0, keep + 16, dummy, pb, bv);
      int so_r = jb * a.st_jb;
      for (int r = 0; r < st_nrep; ++r, so_r += a.st_rb) {
#pragma unroll
        for (int y = 0; y < 8; ++y) {
          if (y < nb) {
            const int j = jb + y;
            int vo[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) vo[v] = st_voff[v];
            if (j == nfp - 1) {   // the ragged last fragment: patches >= P are dropped
#pragma unroll
              for (int v = 0; v < 4; ++v) vo[v] = (16 * j + lcol < P) ? vo[v] : kOob;
            }
#pragma unroll
            for (int v = 0; v < 4; ++v)
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, keep[4 * y + v]), st_rs, vo[v], so_r + y * a.st_jb, 0);
          }
        }
      }
    }
  };

  for (int uu = 0; uu < seg_upw && u < n_units; ++uu, u += WPG) {
  if (seg_kind != 1) {
    // ---- Kzx rows 16 u .. 16 u + 15: out[m][n] = scale * sum_p w_p k(z_m, x_p) ----
    ur = WRITE ? u / a.st_split : u;
    const int part = WRITE ? u - ur * a.st_split : 0;
    const int uj_lo = WRITE ? part * a.st_jn : 0, uj_hi = WRITE ? min(nfp, uj_lo + a.st_jn) : nfp;
    // the Z operand through a buffer descriptor: a per-lane offset that is constant for the unit, the sub-step a scalar offset -- no
    // address arithmetic in the k loop (as plain pointer arithmetic it was 16 VALU instructions per chunk of 16 MFMAs, four of them
    // quarter-rate 64-bit multiply-adds: ~11 % of the loop's issue slots at L = 250)
    const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.ZS), 0, a.Lq * a.Mp * 8, 0x00020000);
    const int zvo = (lrow * a.Mp + 16 * ur + lcol) * 8, zstep = 4 * a.Mp * 8;
    const int zvp = (RL * lrow * a.Mp + 16 * ur + lcol) * 8;   // PERM: rows RL lrow + s of ZS in sub-steps 0 .. RL - 1
    auto ldz = [&](int sub) {
#ifdef HU_ABL_NOA
      return (double)(lane + sub) * 1e-3;   // timing experiment (tools/r05_abl.sh; wrong results): no A-operand loads
#endif
      if (PERM && sub < RL) return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(zrs, zvp, sub * a.Mp * 8, 0));
      return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(zrs, zvo, sub * zstep, 0));
    };
    double rsum[4] = {0.0, 0.0, 0.0, 0.0};
    if (WRITE) {
      st_rs = __builtin_amdgcn_make_buffer_rsrc(a.kuf + ((long)(16 * ur) * a.sM + (long)n * a.sN), 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int v = 0; v < 4; ++v)
        st_voff[v] = (16 * ur + lrow + 4 * v < a.kzx_rows) ? (int)(((long)(lrow + 4 * v) * a.sM + (long)lcol * a.sP) * 8) : kOob;
      st_zero = 16 * ur + 16 > a.M;
    }
    if (KEEP) {   // the unit's 16 rows of the image's P kernel values each: the descriptor's base is (row 16 u, image n)
      st_rs = __builtin_amdgcn_make_buffer_rsrc(a.kfull + ((long)(16 * ur) * a.kf_sM + (long)n * a.kf_sN), 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int v = 0; v < 4; ++v) st_voff[v] = (16 * ur + lrow + 4 * v < a.M) ? (int)(((long)(lrow + 4 * v) * a.kf_sM + lcol) * 8) : kOob;
    }
    if (RES) {
      double areg[NKR];
#pragma unroll
      for (int s = 0; s < NKR; ++s) areg[s] = ldz(s);
      if (WMODE == 2 && st_hold) row_pass_hold([&](int s) { return areg[s]; }, [&](int s, int) { return areg[RES ? s : 0]; }, uj_lo, uj_hi);
      else row_pass(std::false_type{}, [&](int s) { return areg[s]; }, [&](int s, int) { return areg[RES ? s : 0]; }, uj_lo, uj_hi, nullptr, rsum);
    } else {
      if (WMODE == 2 && st_hold) row_pass_hold([&](int s) { return ldz(s); }, [&](int s, int) { return ldz(s); }, uj_lo, uj_hi);
      else row_pass(std::false_type{}, [&](int s) { return ldz(s); }, [&](int s, int) { return ldz(s); }, uj_lo, uj_hi, nullptr, rsum);
    }
    if (!WRITE) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      double s = rsum[v];
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      s += __shfl_xor(s, 8);
      const int m = 16 * ur + lrow + 4 * v;
      if (lcol == 0 && m < a.kzx_rows) a.kzx[(long)m * a.ldk + n] = m < a.M ? a.kzx_scale * s : 0.0;
    }
    }
  } else {
    // ---- Kdiag: chunk u of the image's patch Gram matrix -- tiles on and right of the diagonal (off-diagonal ones count twice), the
    // fragment rows taken in the order 0, nfp - 1, 1, nfp - 2, ... (a long row, then a short one), the tiles of that list cut into
    // chunks of seg_T: a chunk is one to three row segments [j_lo, j_hi) whatever its size ----
    const int ntot = nfp * (nfp + 1) / 2;
    const int lo = u * seg_T, hi = min(lo + seg_T, ntot);
    auto row_of = [&](int k) { return (k & 1) ? nfp - 1 - (k >> 1) : (k >> 1); };
    int k = 0, off = 0;
    while (k < nfp && off + (nfp - row_of(k)) <= lo) { off += nfp - row_of(k); ++k; }
    double total = 0.0;
    for (; k < nfp && off < hi; ++k) {
      const int fr = row_of(k), len = nfp - fr;
      const int j_lo = fr + max(lo - off, 0), j_hi = fr + min(hi - off, len);
      off += len;
      const int pr = 16 * fr + lcol;
      const int pa = pbl[pr];
      const double xav = xb[pr] + a.log2var;
      double rsum[4] = {0.0, 0.0, 0.0, 0.0}, rdiag[4] = {0.0, 0.0, 0.0, 0.0};
      double* rd = j_lo == fr ? rdiag : nullptr;   // the segment opens with the row's diagonal tile
      auto getA_img = [&](int s, int ko) {
        double v = ldi(pa + ko);
        if (s >= sL) v = fixA(v, s, xav);
        return v;
      };
      if (RES) {
        double areg[NKR];
#pragma unroll
        for (int s = 0; s < NKR; ++s) areg[s] = getA_img(s, kob[s]);
        row_pass(std::true_type{}, [&](int s) { return areg[s]; }, [&](int s, int) { return areg[RES ? s : 0]; }, j_lo, j_hi, rd, rsum);
      } else {
        kd_pa = pa;
        row_pass(std::true_type{}, [&](int s) { return getA_img(s, koff[4 * s + lrow]); }, [&](int, int ko) { return ldi(pa + ko); }, j_lo, j_hi, rd, rsum);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) total = fma(wl[16 * fr + lrow + 4 * v], 2.0 * rsum[v] - rdiag[v], total);   // off-diagonal tiles count twice
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) total += __shfl_xor(total, o);
    // every slot of the image is written by someone: chunk u its own and, where the launch's finest chunks are more numerous than this
    // segment's (n_kd > seg_C), zeros into the slots u + seg_C, u + 2 seg_C, ...
    const int slot = u + lane * seg_C;
    if (slot < a.n_kd) a.kd[(long)n * a.n_kd + slot] = lane == 0 ? total : 0.0;
  }
  if (tr) {
    stamp(uu == 0 ? 4 : 5);
    if (lane == 0) tr[7] += 1;
  }
  }
  if (tr && lane == 0) tr[6] = (long long)wall_clock64();
}

}  // namespace

static size_t head_units_lds(const HeadUnitsArgs& a) {
  return (size_t)(((a.HWC + 1) & ~1) + 2 * a.nfp * 16 + ((a.H * (a.W - a.f + 1) + 1) & ~1)) * sizeof(double) +
         (size_t)(a.nfp * 16 + ((a.Lq + 1) & ~1)) * sizeof(int) +
         (((a.L == 25 || a.L == 16 || a.L == 48) && !a.stream_k) ? 256 * sizeof(double) : 0);   // the register-resident forms' table of 2^(j / 256)
}

bool head_units_ok(const HeadUnitsArgs& a) {
  if (a.kuf) {
    // the stores' 32-bit offsets: per lane (15 rows + 15 patches), per fragment + replica (scalar)
    const long lane_max = (15 * a.sM + 15 * a.sP) * 8, uni_max = ((long)(a.nfp + 1) * 16 * a.sP + (long)a.N * a.sN) * 8;
    if (lane_max >= (1L << 31) || uni_max >= (1L << 31)) return false;
  }
  if (a.kfull && (15 * a.kf_sM + 15) * 8 + (long)(a.nfp + 1) * 128 >= (1L << 31)) return false;
  return head_units_lds(a) <= 54 * 1024 && (long)a.Lq * a.Mp * 8 < (1L << 31);
}

// fills the derived fields of `a` (fragment counts, units per image)
void head_units_plan(HeadUnitsArgs* a) {
  a->HWC = a->H * a->W * a->C;
  a->nfm = a->Mp / 16;
  a->nfp = (a->P + 15) / 16;
  if (a->kzx_rows <= 0) a->kzx_rows = a->Mp;
  a->inv_C = 1.0f / (float)a->C; a->inv_f = 1.0f / (float)a->f; a->inv_Wo = 1.0f / (float)a->Wo; a->inv_Wr = 1.0f / (float)(a->W - a->f + 1);
  a->nseg = 0; a->n_wgs = 0; a->n_kd = 0; a->upw = 1; a->n_base = a->N;
  // waves per workgroup: 4 where the patch fits registers (one set-up for four long units); 2 for long patches on small views
  // (a 12 x 12 x 10 head input: units of 252 MFMAs, short beside the set-up they sit behind)
  a->wpg = (a->L == 25 || a->kuf || a->nfp > 8) ? 4 : 2;
  auto push = [&](int kind, int n_img, int wpi, int T, int C, int upw = 1) {
    if (n_img <= 0 || a->nseg >= 6) return;
    HuSeg& g = a->seg[a->nseg++];
    g.wg0 = (int)a->n_wgs; g.img0 = 0; g.wpi = wpi; g.kind = kind; g.T = T; g.C = C; g.upw = upw;
    a->n_wgs += (long)n_img * wpi;
  };
  // Occupancy shaping.  Workgroups are handed out as slots free up, so a launch of r = waves / resident slots rounds costs ceil(r)
  // rounds' worth of time when r is small: 6016 equal units on 4096 slots (4 waves per SIMD) are a full round and then a round at 47 %
  // occupancy that lasts almost as long (tools/sweep_trace.py: 74 us for 41 us of MFMA work).  Held to 3 waves per SIMD -- by claiming
  // enough LDS that only so many workgroups fit a CU -- the same launch is 1.96 rounds, both full.  `cap`: what the kernel's register
  // budget allows; the pipe reaches ~92 / 96 / 98 % of its rate from 2 / 3 / 4 waves per SIMD.
  auto shape_occupancy = [&](int cap) {
    // Round 5 looked at where the thin second round goes (tools/sweep_trace.py now records every wave's HW_ID; tools/r05_abl.sh): its waves
    // ARE spread evenly (896 SIMDs with two of them, 128 with one at the 12 x 12 x 10 head); a unit of 252 MFMAs = 7.2 us of issue takes a
    // wave 12.9 us alone on its SIMD and 17.9 us beside one other, and exactly as long with its A loads, its B gathers or both removed --
    // one wave issues an fp64 MFMA every ~96 cycles, not 64, so a SIMD needs three to four waves whatever they wait for.  Two plans that
    // keep the launch to ONE round (every image's units dealt evenly to the workgroups that fit the chip at once, two to three units per
    // wave behind one set-up; the Kdiag chunks as second units, or alone on a wave) measured 76 - 78 us against 70: a Kdiag chunk -- two row
    // passes, one of them a single tile wide -- takes a wave 50 - 67 us beside three others and is the launch's last wave either way.
    // MEASURED AND LEFT OFF (tools/head_ab.sh): every shape got slower held to fewer waves -- 79 -> 101 us at the 12 x 12 x 10 head, 110 ->
    // 134 us at the CIFAR head, 163 -> 176 us at the MNIST head.  A wave of these sweeps is bound by its own latencies (LDS gathers, the
    // A operand from L2), not by the pipe, so a SIMD with three waves does less than one with four whatever the round count says.
    a->occ = 0;
    if (a->occ_force <= 0) return;
    if (a->occ_force > 0) { a->occ = a->occ_force < cap ? a->occ_force : 0; return; }
    const double waves = (double)a->n_wgs * a->wpg;
    if (waves < 2048.0 || waves / (1024.0 * cap) >= 6.0) return;
    static const double eff[5] = {0.0, 0.75, 0.92, 0.96, 0.98};
    double best = 0.0;
    int best_s = cap;
    for (int s = cap; s >= 2; --s) {
      const double r = waves / (1024.0 * s);
      const double e = r / ceil(r) * eff[s];
      if (e > best + 0.02) { best = e; best_s = s; }
    }
    if (best_s < cap) a->occ = best_s;
  };
  if (a->kuf) {
    // ---- the storing form: row units only ----
    // rows n, n + n_mod, ... show the same image: evaluated once, stored to each (see the note at the kernel)
    if (!a->no_rep && a->n_mod > 0 && a->n_mod < a->N) a->n_base = a->n_mod;
    a->st_jb = (int)(16 * a->sP * 8);
    a->st_rb = (int)((long)a->n_mod * a->sN * 8);
    // few units (the distinct images of a tiled batch: 32 x 16 at the headline size): narrower workgroups, so that every CU has one
    // with replicas a unit is a row fragment x a range of <= 8 column fragments (one batch of row_pass_hold): more, shorter waves -- the
    // launch ends within a short unit's time of its slowest wave (identical waves finished 27 ... 53 us after their set-up at the CIFAR
    // first layer: the memory system does not serve them evenly) -- and the stores of a range start after <= 8 tiles, not after all
    a->st_split = 1; a->st_jn = a->nfp;
    a->st_hold = a->n_base < a->N && a->sP == 1 && (a->sN % 16) != 0;   // row segments of an image not 128-byte aligned: hold + replica-outer
    if (a->n_base < a->N && a->split_force != 0) {
      a->st_split = a->split_force > 0 ? a->split_force : (a->nfp + 7) / 8;
      if (a->st_split > a->nfp) a->st_split = a->nfp;
      a->st_jn = (a->nfp + a->st_split - 1) / a->st_split;
      a->st_split = (a->nfp + a->st_jn - 1) / a->st_jn;
    }
    const int nuw = a->nfm * a->st_split;   // units per image
    const long units = (long)a->n_base * nuw;
    if (units < 1024) a->wpg = units >= 256 ? 2 : 1;
    if (a->wpg_force == 1 || a->wpg_force == 2 || a->wpg_force == 4) a->wpg = a->wpg_force;
    // units per wave: the set-up of a workgroup (image, window sums, tables: ~2.5 us of latency) is as long as a short unit (a 16-row
    // fragment against the 9 patch fragments of a 12 x 12 view at L = 25: 63 MFMAs), so such launches put several units behind one
    // set-up -- as many as leave >= 512 workgroups (two per CU); units of a few hundred MFMAs (L = 250) stay one per wave
    if (a->upw_force > 0) a->upw = a->upw_force;
    else if (a->st_jn * (a->Lq / 4) < 200 && a->n_base == a->N)
      for (int k = 4; k > 1; k >>= 1) {
        const long nwg = (long)a->n_base * ((nuw + a->wpg * k - 1) / (a->wpg * k));
        if (a->nfp <= 16 && nwg >= 512 && nuw % (a->wpg * k) == 0) { a->upw = k; break; }
      }
    push(2, a->n_base, (nuw + a->wpg * a->upw - 1) / (a->wpg * a->upw), 0, 0, a->upw);
    shape_occupancy(a->st_hold ? 2 : 3);
    return;
  }
  // ---- the reducing form: the Kzx row units of every image first, then the Kdiag chunks, shrinking towards the end of the launch ----
  // A workgroup lives as long as its longest wave, and a wave that shares its SIMD with three others needs ~4 x its own issue time: a
  // unit of 37 tiles is ~50 us of a ~150 us launch, and workgroups that END at random moments of their last 50 us leave a third of the
  // chip idle for the final stretch (tools/sweep_trace.py: busy waves over time).  The dispatcher hands out workgroups in id order as
  // slots free up, so the order of the list is the schedule: long units first, and the last stretch made of chunks of 1/2, 1/4, 1/8 the
  // size -- each level about half a round of the resident slots -- ends within one short chunk.
  const int W = a->wpg;
  // row units of a few dozen MFMAs (a 5 x 5 view of long patches: two column fragments) are shorter than the set-up they sit behind:
  // two to four per wave
  int kz_upw = 1;
  {
    const int unit_mfma = a->nfp * (a->Lq / 4);
    if (unit_mfma < 200) {
      kz_upw = (250 + unit_mfma - 1) / unit_mfma;
      if (kz_upw > 4) kz_upw = 4;
      while (kz_upw > 1 && a->nfm % (W * kz_upw)) --kz_upw;
    }
    // long launches (M = 1024 on long patches: ~20 rounds of the resident slots) halve their set-ups the same way (817 -> 778 us); launches of
    // a round or two must not (the 12 x 12 x 10 head at M = 256: 74 -> 76 us with two units per wave, 90 with four)
    if (kz_upw == 1 && unit_mfma < 400 && (long)a->N * a->nfm / W >= 16 * 1024 && a->nfm % (W * 2) == 0) kz_upw = 2;
    if (a->upw_force > 0) kz_upw = a->upw_force;   // A/B (ctx option head_upw)
  }
  if (a->kzx) push(0, a->N, (a->nfm + W * kz_upw - 1) / (W * kz_upw), 0, 0, kz_upw);
  if (a->want_kd || a->kd) {
    const int ntot = a->nfp * (a->nfp + 1) / 2;
    // coarse chunks: about one Kzx-sized unit (nfp + 1 tiles) each, their count a multiple of the workgroup's waves where that is possible
    int C0 = ntot / (a->nfp + 1);
    if (C0 >= W) C0 = C0 / W * W;
    if (C0 < 1) C0 = 1;
    int T0 = (ntot + C0 - 1) / C0;
    C0 = (ntot + T0 - 1) / T0;
    int lvT[4] = {T0, 0, 0, 0}, lvC[4] = {C0, 0, 0, 0}, lvN[4] = {a->N, 0, 0, 0}, nlv = 1;
    const long slots = 1024;   // wave slots the chip holds of this kernel (4 per SIMD)
    // measured (tools/head_ab.sh): worth 4-5 % on launches of a few rounds of the slots (M = 32 head, the 12 x 12 x 10 heads); from ~10
    // rounds on (MNIST head at M = 256: 10.6) the extra workgroups' set-ups cost what the sharper end saves, so those keep equal chunks
    const long rounds = (long)a->N * ((a->kzx ? a->nfm : 0) + C0) / slots;
    if (a->tail_mode > 0 || (a->tail_mode < 0 && rounds < 8)) {
      int left = a->N;
      for (int lv = 1; lv < 4; ++lv) {
        const int T = (T0 + (1 << lv) - 1) >> lv;
        if (T < 3 || T == lvT[lv - 1]) break;
        const int C = (ntot + T - 1) / T;
        long n = slots / 2 / C;                     // half a round of the slots at this chunk size
        if (a->tail_mode > 0) n = n * a->tail_mode / 4;   // A/B: tail_mode / 4 of that
        if (n < 1) n = 1;
        if (n > left / 2) n = left / 2;             // never more than half of what is left: the coarse levels keep the bulk
        if (n <= 0) break;
        lvT[lv] = T; lvC[lv] = C; lvN[lv] = (int)n; left -= (int)n; nlv = lv + 1;
      }
      lvN[0] = left;
    }
    int img = 0;
    for (int lv = 0; lv < nlv; ++lv) {
      const int before = a->nseg;
      push(1, lvN[lv], (lvC[lv] + W - 1) / W, lvT[lv], lvC[lv]);
      if (a->nseg > before) { a->seg[before].img0 = img; img += lvN[lv]; }
      if (lvC[lv] > a->n_kd) a->n_kd = lvC[lv];
    }
  }
  shape_occupancy(4);
}

extern "C" int dcgp_debug_set_sweep_trace(dcgp_ctx* ctx, long long* buf_dev, long n_workgroups, const char* family) {
  if (!ctx) return DCGP_ERR_ARG;
  ctx->sweep_trace = buf_dev;   // [n_workgroups][waves per workgroup][8] int64; nullptr switches the stamps off
  ctx->sweep_trace_wgs = buf_dev ? n_workgroups : 0;
  ctx->sweep_trace_family = family ? family : "";
  return DCGP_OK;
}

int head_units(dcgp_ctx* ctx, const HeadUnitsArgs& a_in) {
  HeadUnitsArgs a = a_in;
  const char* family = a.timer ? a.timer : (a.kuf ? "kuf" : "head_sweep");
  if (ctx->sweep_trace && (ctx->sweep_trace_family.empty() || ctx->sweep_trace_family == family)) { a.trace = ctx->sweep_trace; a.trace_wgs = ctx->sweep_trace_wgs; }
  if (a.N <= 0) return DCGP_OK;
  a.exp_tab = exp2_table(ctx);
  if (!a.exp_tab) return DCGP_ERR_ALLOC;
  if (a.want_kd && !a.kd) return ctx_fail(ctx, DCGP_ERR_ARG, "head_units: Kdiag partial sums wanted but no buffer (allocate kd [N][n_kd] behind head_units_plan)");
  if (!head_units_ok(a) || a.n_mod <= 0 || a.Lq != round_up(a.L + 2, 4) || a.Mp % 16)
    return ctx_fail(ctx, DCGP_ERR_ARG, "head_units: unsupported shape (image %d doubles, L = %d, Mp = %d)", a.HWC, a.L, a.Mp);
  const long nwg = a.n_wgs;
  if (nwg <= 0) return DCGP_OK;
  if (nwg > 0x7fffffffL) return ctx_fail(ctx, DCGP_ERR_ARG, "head_units: too many workgroups");
  size_t lds = head_units_lds(a);
  // Beside the factorisation chain (a head-first model: the sweep needs Z only): a chain workgroup is one wave of 250 VGPRs per SIMD
  // and 50 KB of LDS, and would never find that much free at once on a CU this launch keeps refilling with four 128-register
  // workgroups.  Claiming 53 KB per workgroup holds the sweep to THREE per CU: whenever one of them ends -- somewhere on the chip
  // every ~0.1 us -- that CU has 53 KB and half of every SIMD's registers free, and the chain's high-priority stream takes the slot
  // before the sweep's backlog does.  (Round 3 claimed 54 KB = two per CU, so that every CU always had room: the sweep ran at two waves
  // per SIMD, 184 us instead of 171, and the chain no faster -- 157 against 134 us; head-only model 4100 -> 4300 steps/s.  52 KB and
  // below measured worse again, and with no claim at all the chain starves: 213 us.  ctx option share_kb.)
  const size_t share_claim = (size_t)(a.share_kb > 0 ? a.share_kb : 53) * 1024;
  if (a.share_cu && lds < share_claim) lds = share_claim;
  if (a.occ > 0) {   // occupancy shaping (head_units_plan): exactly 4 occ / wpg workgroups per CU
    const int per_cu = 4 * a.occ / a.wpg;
    const size_t claim = (size_t)(160 * 1024 / per_cu) & ~(size_t)255;
    if (claim <= 64 * 1024 && lds < claim) lds = claim;
  }
  // patch rows of 50 contiguous elements, an odd number of them (5 x 5 x 10): the patch-row form of the streamed loop (kernel: RW); ctx option sweep_no_rows: A/B
  const bool rw50 = a.f * a.C == 50 && (a.f & 1) && a.L == a.f * a.f * a.C && !ctx->opt.sweep_no_rows;
  ScopedTimer t(ctx, family);
  if (a.kuf) {
    // patch lengths of the first layers (5 x 5 x 1, 4 x 4 x 1, 4 x 4 x 3: MNIST / CIFAR conv0) with the row operand resident in registers
#define HU_STORE_W(NK4, TL, WM)                                                                                                          \
  do {                                                                                                                                     \
    if (a.wpg == 4) hipLaunchKernelGGL((head_units_kernel<NK4, TL, WM, 256>), dim3((unsigned)nwg), dim3(256), lds, ctx->stream, a);        \
    else if (a.wpg == 2) hipLaunchKernelGGL((head_units_kernel<NK4, TL, WM, 128>), dim3((unsigned)nwg), dim3(128), lds, ctx->stream, a);   \
    else hipLaunchKernelGGL((head_units_kernel<NK4, TL, WM, 64>), dim3((unsigned)nwg), dim3(64), lds, ctx->stream, a);                     \
  } while (0)
#define HU_STORE(NK4, TL)                     \
  do {                                        \
    if (a.st_hold) HU_STORE_W(NK4, TL, 2);    \
    else HU_STORE_W(NK4, TL, 1);              \
  } while (0)
    if (a.L == 25 && !a.stream_k) HU_STORE(7, 1);
    else if (a.L == 16 && !a.stream_k) HU_STORE(5, 0);
    else if (a.L == 48 && !a.stream_k) HU_STORE(13, 0);
    else if (rw50) HU_STORE(0, 50);
    else HU_STORE(0, 0);
#undef HU_STORE
#undef HU_STORE_W
  } else if (a.kfull) {   // a training step's head: the reducing form that also leaves every kernel value behind
    if (a.L == 25) hipLaunchKernelGGL((head_units_kernel<7, 1, 3, 256>), dim3((unsigned)nwg), dim3(256), lds, ctx->stream, a);
    else if (rw50 && a.wpg == 4) hipLaunchKernelGGL((head_units_kernel<0, 50, 3, 256>), dim3((unsigned)nwg), dim3(256), lds, ctx->stream, a);
    else if (rw50) hipLaunchKernelGGL((head_units_kernel<0, 50, 3, 128>), dim3((unsigned)nwg), dim3(128), lds, ctx->stream, a);
    else if (a.wpg == 4) hipLaunchKernelGGL((head_units_kernel<0, 0, 3, 256>), dim3((unsigned)nwg), dim3(256), lds, ctx->stream, a);
    else hipLaunchKernelGGL((head_units_kernel<0, 0, 3, 128>), dim3((unsigned)nwg), dim3(128), lds, ctx->stream, a);
  } else if (a.L == 25) {
    hipLaunchKernelGGL((head_units_kernel<7, 1, 0, 256>), dim3((unsigned)nwg), dim3(256), lds, ctx->stream, a);   // 5 x 5 x 1 patches
  } else if (rw50) {   // 5 x 5 x 10 patches (every long patch of the BASELINE configurations): the streamed form walking patch rows
    if (a.wpg == 4) hipLaunchKernelGGL((head_units_kernel<0, 50, 0, 256>), dim3((unsigned)nwg), dim3(256), lds, ctx->stream, a);
    else hipLaunchKernelGGL((head_units_kernel<0, 50, 0, 128>), dim3((unsigned)nwg), dim3(128), lds, ctx->stream, a);
  } else if (a.wpg == 4) {
    hipLaunchKernelGGL((head_units_kernel<0, 0, 0, 256>), dim3((unsigned)nwg), dim3(256), lds, ctx->stream, a);
  } else {
    hipLaunchKernelGGL((head_units_kernel<0, 0, 0, 128>), dim3((unsigned)nwg), dim3(128), lds, ctx->stream, a);
  }
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}
