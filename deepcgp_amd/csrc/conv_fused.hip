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
#include <array>
#include <map>
#include <mutex>
#include <queue>
#include <vector>

#include "layer.h"
#include "rng.h"

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// the scheduler otherwise sinks each sub-step's LDS reads behind the previous sub-step's MFMAs and into the same registers: one buffer, every
// sub-step waits a full LDS latency (the source order -- reads one sub-step ahead -- is what is meant)
#define CF_SB __builtin_amdgcn_sched_barrier(0)
#define CF_DW 2
#define CF_P1 50
#define CF_P2 75
#define CF_P3 90
constexpr int CF_D_DEFAULT = 3;   // A-operand k-tiles in flight per wave beside the one being multiplied

__device__ __forceinline__ void set_prio(int p) {   // s_setprio takes an immediate
  switch (p) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
}
// i / d for 0 <= i, d < 2^23 without the ~40-instruction integer division (quarter-rate multiplies among them): the float quotient of i + 1/2,
// put right by one step where rounding tipped it over (full-rate 24-bit multiply).  A strip's set-up and epilogue held ~560 VALU instructions of
// divisions per wave -- a third of everything the launch issues outside its MFMAs, and VALU cycles are MFMA cycles on this part.
__device__ __forceinline__ int fdiv(int i, int d, float inv_d) {
  int q = (int)(((float)i + 0.5f) * inv_d);
  const int r = i - (int)__umul24((unsigned)q, (unsigned)d);
  return q + (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
}
__device__ __forceinline__ int frag_of(int w, int W, int c) { return (c >> 1) * 2 * W + ((c & 1) ? 2 * W - 1 - w : w); }

// phase timestamps for tools/fused_trace.py: [8 sampled workgroups][4 strips of a persistent workgroup][wave][16] shader-clock ticks
#define CF_TR(k) \
  if (tr_slot >= 0 && lane == 0) a.trace[(tr_slot * 16 + wave) * 16 + (k)] = (long long)__builtin_readcyclecounter();

// FN: 16-column fragments per strip.  NS: the W = NT/64 waves form NS teams of TW = W/NS; the row fragments are dealt to the
// TW members of a team (boustrophedon), and the teams split the rest of the work: the columns of the strip in the sweep and in
// the first product (FN/NS fragments each), the outputs r = team, team + NS, ... in the R-batched second product (all FN
// column fragments, so that an A tile is still fetched by exactly one wave).  NS = 2 puts 16 waves = 4 per SIMD on a
// 64-column strip: the fp64 MFMA pipe reaches 92 % of its rate from two waves per SIMD and 98 % from four.
// MAXF: row fragments per wave at most.  ABL: timing experiments only (wrong results): 1 = second product without its
// A-operand loads, 2 = without the LDS reads of the B operand, 4 = two more A tiles in flight.
// BTP: 0 RBF, 1 ArcCosine, 2 RBF on 5 x 5 x 10 patches (the in-kernel sweep walks patch rows; an instance of its own so that the others do not carry its registers)
template <int FN, int NS, int MAXF, int NT, int BTP, int ABL = 0>
__global__ __launch_bounds__(NT) void conv_fused_kernel(ConvFusedArgs a_in) {
  constexpr int BT = BTP == 1 ? 1 : 0;
  // The arguments are read through the kernarg pointer, which every strip of a persistent workgroup sees as a new value: as a by-value
  // struct the loop-invariant loads of all ~70 words are hoisted out of the strip loop, live across it, and spill (240 VGPRs at 16 waves)
  typedef const __attribute__((address_space(4))) ConvFusedArgs KArgs;
  KArgs* ap = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  KArgs& a = *ap;
  constexpr int BN = FN * 16, W = NT / 64, TW = W / NS, FNS = FN / NS, KG = W / FN;
  static_assert(FN % NS == 0 && W % NS == 0 && W % FN == 0, "team split");
  constexpr int CF_D = ((ABL & 4) ? 2 : 0) + (NT >= 1024 ? CF_DW : CF_D_DEFAULT);   // 128-register budget at 16 waves: two tiles ahead
  extern __shared__ __attribute__((aligned(16))) double smem[];

  // A persistent launch (a.persist: one workgroup per slot of the chip): the workgroup walks the strips blockIdx, blockIdx + grid, ...
  // The second workgroup to arrive on a CU holds back for a.stagger ticks of the 100 MHz clock, so that its sweep / first product / epilogue -- the
  // stretches that leave the matrix pipe thin -- fall into the other's second product and the other's into its own, strip after strip.
  int cu_word = -1;
  if (a.persist && a.cu_slots) {
    int* word = reinterpret_cast<int*>(smem);
    if (threadIdx.x == 0) {
      unsigned xcc, hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      cu_word = (int)(((xcc & 7u) << 7) | ((hwid >> 8) & 0x7fu));   // XCC, SE_ID [15:13], SH_ID [12], CU_ID [11:8]
      word[0] = atomicAdd(a.cu_slots + cu_word, 1);
    }
    __syncthreads();
    const int arrival = word[0];
    __syncthreads();
    if ((arrival & 1) && a.stagger > 0) {
      const long long t0 = wall_clock64();
      while (wall_clock64() - t0 < a.stagger) __builtin_amdgcn_s_sleep(64);
    }
  }
  int strip_next = blockIdx.x;
  if (a.pre_n > 0) {
    // Prologues ahead: EVERY item, a workgroup's first included, comes off the counter -- an item is then held by a workgroup that is running.  (Dealt by
    // blockIdx, a prologue item could belong to a workgroup that is not resident yet -- the chip shared with another kernel, a CU-masked stream -- while a
    // running one already waits for its A1.)
    int* tk = reinterpret_cast<int*>(smem);
    if (threadIdx.x == 0) tk[0] = atomicAdd(a.dyn, 1);
    __syncthreads();
    strip_next = __builtin_amdgcn_readfirstlane(tk[0]);
    __syncthreads();
  }
  // (a workgroup that starts when every item has been dealt -- more workgroups than CUs it may run on -- has nothing to do but sign off)
  const int first_limit = a.pre_n > 0 ? a.n_strips + a.pre_n * a.pre_sq : 0x7fffffff;
  for (int it = 0; strip_next < first_limit; ++it) {
  // every strip sees the kernarg pointer and the thread index as new values: nothing of a strip's set-up (argument words, per-lane offsets of every
  // phase) is then loop-invariant, hoisted and kept live across the whole body -- as plain invariants they cost 240 spilled VGPRs at 16 waves
  int tid = threadIdx.x;
  asm volatile("" : "+s"(ap), "+v"(tid));
  KArgs& a = *ap;
  const int Mp = a.Mp, nf = Mp >> 4, R = a.R;
  const int lane = tid & 63, lrow = lane >> 4, lcol = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % TW, sp = NS == 1 ? 0 : wave / TW;   // member index within the team, team
  double* strip = smem;                                  // [Mp / 4][FN][4][16]: rows in fours (below)
  double* aux = smem + a.lds_main;                       // images of the strip; after phase 1: [TW][BN] partial sums of A1^2
  double* xn = aux + a.lds_img;                          // [BN] |x_p|^2
  int* ticket = reinterpret_cast<int*>(xn + BN);         // persistent launch: the next strip of this workgroup
  int* koff = reinterpret_cast<int*>(xn + BN + 2);       // [Lp]
  const int jmax = a.Kc - 1;
  // row fragments of this wave: frag_of(wm, TW, c), c < nfw
  int nfw = 0;
#pragma unroll
  for (int c = 0; c < MAXF; ++c) nfw += frag_of(wm, TW, c) < nf ? 1 : 0;
  nfw = __builtin_amdgcn_readfirstlane(nfw);
  int fr[MAXF];   // fragments beyond the matrix repeat the first one (their loads stay in range, their results are dropped)
#pragma unroll
  for (int c = 0; c < MAXF; ++c) fr[c] = c < nfw ? frag_of(wm, TW, c) : min(wm, nf - 1);
  // strips of the launch's partial last round are shared by SQ workgroups: each runs phases 0 - 2 and the outputs r = sq, sq + SQ, ...
  int sidx = strip_next, sq = 0, SQ = 1;
  if (sidx >= a.split_first) {
    const int t = sidx - a.split_first;
    SQ = a.split_q;
    sidx = a.split_first + t / SQ;
    sq = t - (sidx - a.split_first) * SQ;
  }
  // Prologues ahead (a.pre_n > 0, a persistent launch whose strips do not fill its last round: DESIGN 4i).  The items of the launch, in the order the
  // counter deals them: the strips [0, pre_first) whole -- one per workgroup of the partial FIRST round --, then pre_n items that run only phases 0 - 2
  // of the strips [pre_first, pre_first + pre_n) and leave A1 (the LDS image of the strip) and the partial sums of A1^2 in a.pre_buf: they fill the
  // first round's spare workgroups; then the remaining strips whole; then the pre_n strips again, which fetch their A1 and go straight to phase 3.
  // Same arithmetic in the same order as a whole strip: bit-identical.
  bool pre_fail = false;
  int mode = 0, pre_slot = 0;   // 0: a whole strip; 1: phases 0 - 2 only, A1 left in a.pre_buf; 2: A1 fetched from a.pre_buf, phases 3 - 4
  if (a.pre_n > 0) {
    if (strip_next >= a.n_strips) {   // (pre_sq > 1: part sq of the strip's SQ, the split of a shared last round applied to a strip whose A1 is fetched)
      mode = 2;
      const int t = strip_next - a.n_strips;
      SQ = a.pre_sq;
      pre_slot = SQ > 1 ? t / SQ : t;
      sq = t - pre_slot * SQ;
      sidx = a.pre_first + pre_slot;
    }
    else if (strip_next >= a.pre_first && strip_next < a.pre_first + a.pre_n) { mode = 1; pre_slot = strip_next - a.pre_first; }
  }
  const int j0 = sidx * BN;
  int tr_slot = -1;
  if (a.trace) {
    const int every = a.persist ? max((int)gridDim.x / 8, 1) : 90;
    if (blockIdx.x % every == 0 && blockIdx.x / every < 8 && it < 4) tr_slot = (blockIdx.x / every) * 4 + it;
  }
  CF_TR(0)
  if (tr_slot >= 0 && lane == 0) a.trace[(tr_slot * 16 + wave) * 16 + 10] = (long long)wall_clock64();
  // Two workgroups share a CU in a persistent launch.  The stretches that cannot fill the matrix pipe by themselves (images, sweep, first product,
  // epilogue: latency- and barrier-bound) issue ahead of the other workgroup's second product, which takes what they leave: at equal or lower
  // priority a sweep beside a second product took 110 us instead of 25 (profiles/r06_fused_persistent_static_trace.txt) while the second product,
  // alone on two waves per SIMD, cannot use more than 81 % of the pipe
  __builtin_amdgcn_s_setprio(3);

  // ---- the A-operand stream ---------------------------------------------------------------------------------------
  // lane (lrow, lcol) of k-substep q of k-tile kt needs Wt[kt*16 + 4q + lrow][16 f + lcol]: per-lane byte offset voff[q],
  // everything else (matrix r, k-tile, fragment) is a scalar byte offset
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
  constexpr int QS = 4 * BN;
  if (mode != 2) {
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
        const int n = fdiv(i, a.HWC, a.inv_HWC), o = i - n * a.HWC;
        const int nn = n_first + n, img = nn - fdiv(nn, a.n_mod, a.inv_nmod) * a.n_mod;
        t[e] = (i < total) ? a.X[(long)img * a.HWC + o] : 0.0;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = i0 + e * NT + tid;
        if (i < total) aux[i] = BT == 0 ? t[e] * a.csq : t[e];   // RBF: the sweep's operands carry the kernel's scales (sweep_dev.h)
      }
    }
    if (it == 0)
    for (int l = tid; l < (BT == 0 ? a.Lz : a.Lp); l += NT) {
      const int ll = l < a.L ? l : 0;
      const int c = ll % a.C, t = ll / a.C;
      const int kw = t % a.f, kh = t / a.f;
      koff[l] = (kh * a.W + kw) * a.C + c;
    }
  }
  // LDS offset of the patch of strip column c (columns beyond the matrix repeat the last one: finite, never written out)
  auto patch_off = [&](int c) {
    const int j = min(j0 + c, jmax);
    const int n = fdiv(j, a.P, a.inv_P), p = j - n * a.P;
    const int oh = fdiv(p, a.Wo, a.inv_Wo), ow = p - oh * a.Wo;
    return (n - n_first) * a.HWC + (oh * a.s * a.W + ow * a.s) * a.C;
  };
  // acos: |z_m|^2 of this lane's accumulator rows: fetched here, needed after the sweep (RBF: the norms ride in the product)
  double znv[BT == 0 ? 1 : MAXF][4];
  if (BT != 0) {
#pragma unroll
    for (int c = 0; c < MAXF; ++c)
#pragma unroll
      for (int v = 0; v < 4; ++v) znv[c][v] = a.zn[16 * fr[c] + lrow + 4 * v];
  }
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
    if (sub == 0) xn[c] = BT == 0 ? -0.5 * s : s;   // RBF: the column's norm slot, -c |x|^2 / 2 (the image is already scaled)
  }
  __syncthreads();
  CF_TR(1)

  // The strip in LDS: rows in fours, element (row k, column 16 y + i) at (k >> 2) * 4 BN + 64 y + 16 (k & 3) + i: a B fragment (rows k0 .. k0 + 3 by lrow,
  // k0 a multiple of 4; columns 16 y + lcol) is 512 contiguous bytes, lane l at l * 8 -- no bank conflicts, and every address of the kernel is the lane
  // index plus compile-time offsets (column group, sub-step, accumulator row: multiples of 512 bytes, the unit of ds_read2st64_b64) plus a scalar
  // (k-tile).  On gfx950 a VALU instruction issues in the fp64 MFMA's place: the row-major strip with its column groups XOR-swizzled by the row spent
  // 10 - 14 of them per k-tile of the second product on addresses (4 - 5 % of its 1024 MFMA cycles), this one spends 1
  // ---- phase 1: K_uf[:, strip] -> strip: this wave's row fragments x its team's FNS column fragments ----------------
  {
    int pb[FNS];
#pragma unroll
    for (int y = 0; y < FNS; ++y) pb[y] = patch_off((sp * FNS + y) * 16 + lcol);
    const int nk4 = (BT == 0 ? a.Lz : a.Lp) >> 2;
    d4 kacc[MAXF][FNS];
#pragma unroll
    for (int c = 0; c < MAXF; ++c)
#pragma unroll
      for (int y = 0; y < FNS; ++y) kacc[c][y] = d4{0.0, 0.0, 0.0, 0.0};
    // RBF: ZS = sqrt(c) Z^T with the rows (-c |z|^2 / 2 + log2 variance, 1) behind the patch, against columns
    // (sqrt(c) x, 1, -c |x|^2 / 2): the accumulator is log2 of the kernel value (sweep_dev.h, head_units.hip)
    // (through a buffer descriptor: one 32-bit per-lane offset per row fragment, the k sub-step a scalar offset -- as a 64-bit pointer per load the address
    // arithmetic was ~8 VALU instructions per sub-step in the MFMA stream, more than the sub-step's own 4 MFMAs at a long patch)
    const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(BT == 0 ? a.ZS : a.ZT), 0, nk4 * 4 * Mp * 8, 0x00020000);
    unsigned zoff[MAXF];
#pragma unroll
    for (int c = 0; c < MAXF; ++c) zoff[c] = (unsigned)((lrow * Mp + 16 * fr[c] + lcol) * 8);
    double xnv[FNS];
#pragma unroll
    for (int y = 0; y < FNS; ++y) xnv[y] = xn[(sp * FNS + y) * 16 + lcol];
    constexpr int D4 = 4;
    double ring[D4 + 1][MAXF];
#pragma unroll
    for (int c = 0; c < MAXF; ++c) ring[D4][c] = 0.0;   // (defined on every path: an undefined slot would be carried from strip to strip of a persistent workgroup)
    auto ldz = [&](int k4, double (&dst)[MAXF]) {
#pragma unroll
      for (int c = 0; c < MAXF; ++c) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(zrs, (int)zoff[c], k4 * 4 * Mp * 8, 0);
        __builtin_memcpy(&dst[c], &v, 8);
      }
    };
    auto kstep = [&](int k4, const double (&w)[MAXF]) {
      const int k = 4 * k4 + lrow;
      const int ko = koff[k];
      double bv[FNS];
      if (BT == 0) {
        // slots behind the patch: k == L carries 1, k == L + 1 the column's norm, anything further 0
        const double f_real = k < a.L ? 1.0 : 0.0, f_one = k == a.L ? 1.0 : 0.0, f_nrm = k == a.L + 1 ? 1.0 : 0.0;
#pragma unroll
        for (int y = 0; y < FNS; ++y) bv[y] = fma(aux[pb[y] + ko], f_real, fma(xnv[y], f_nrm, f_one));
      } else {
        const bool kin = k < a.L;
#pragma unroll
        for (int y = 0; y < FNS; ++y) {
          const double v = aux[pb[y] + ko];
          bv[y] = kin ? v : 0.0;
        }
      }
#pragma unroll
      for (int c = 0; c < MAXF; ++c)
#pragma unroll
        for (int y = 0; y < FNS; ++y) kacc[c][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(w[c], bv[y], kacc[c][y], 0, 0, 0);
    };
    // sub-steps that hold patch elements only (every lane's k < L): the gathered value IS the operand -- the slot selects and the two FMAs per column
    // fragment of the general sub-step are VALU instructions in the MFMA stream (a long patch, L = 250, walks 62 such sub-steps of 63: cfg3's second layer)
    // (the patch-element offsets of a group of sub-steps are read a group ahead: offset -> gather -> MFMA inside one sub-step is two LDS round trips)
    auto kstep_fast = [&](int ko, const double (&w)[MAXF]) {
      double bv[FNS];
#pragma unroll
      for (int y = 0; y < FNS; ++y) bv[y] = aux[pb[y] + ko];
#pragma unroll
      for (int c = 0; c < MAXF; ++c)
#pragma unroll
        for (int y = 0; y < FNS; ++y) kacc[c][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(w[c], bv[y], kacc[c][y], 0, 0, 0);
    };
    const int nfast = min(a.L >> 2, nk4);
#pragma unroll
    for (int u = 0; u < D4; ++u) ldz(min(u, nk4 - 1), ring[u]);
    // 5 x 5 x 10 patches (a layer behind a 10-map conv layer: cfg3's second layer): f = 5 rows of 50 contiguous image elements, and lane group lrow's element
    // of sub-step s, k = 4 s + lrow, walks a row at +4 doubles per sub-step -- the 12 aligned sub-steps of a row gather at ONE per-lane address per column fragment +
    // immediates, two rows and the sub-step that straddles them are a period of 25 (head_units.hip: the patch-row form of the streamed sweep; same operands into
    // the same MFMAs in the same order).  No offset table, no address per gather.
    if constexpr (BTP == 2) {   // (the launcher checks: f C == 50, f odd, L == f f C)
      constexpr int NA = 12;
      static_assert(D4 + 1 == 5, "the ring of Z sub-steps closes over a period of 25");
      const int RSd = a.W * a.C;                       // doubles from one patch row to the next in the image
      const int sdl = lrow < 2 ? NA * 4 : RSd - 2;     // the straddling sub-step: lane groups 0, 1 end the row, 2, 3 open the next
      int pA[FNS], pS[FNS], pB[FNS];
#pragma unroll
      for (int y = 0; y < FNS; ++y) pA[y] = pb[y] + lrow;
      int sc = 0;
      auto sub = [&](int SL, const int (&base)[FNS], int off) __attribute__((always_inline)) {   // (SL, off: constants once the callers' loops are unrolled)
        ldz(min(sc + D4, nk4 - 1), ring[(SL + D4) % (D4 + 1)]);
        double bv[FNS];
#pragma unroll
        for (int y = 0; y < FNS; ++y) bv[y] = aux[base[y] + off];
#pragma unroll
        for (int c = 0; c < MAXF; ++c)
#pragma unroll
          for (int y = 0; y < FNS; ++y) kacc[c][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[SL][c], bv[y], kacc[c][y], 0, 0, 0);
        ++sc;
      };
      const int nper = (a.f - 1) >> 1;
      for (int tp = 0; tp < nper; ++tp) {
#pragma unroll
        for (int y = 0; y < FNS; ++y) { pS[y] = pA[y] + sdl; pB[y] = pA[y] + (RSd + 2); }
#pragma unroll
        for (int i = 0; i < NA; ++i) sub(i % (D4 + 1), pA, 4 * i);
        sub(NA % (D4 + 1), pS, 0);
#pragma unroll
        for (int i = 0; i < NA; ++i) sub((NA + 1 + i) % (D4 + 1), pB, 4 * i);
#pragma unroll
        for (int y = 0; y < FNS; ++y) pA[y] += 2 * RSd;
      }
#pragma unroll
      for (int i = 0; i < NA; ++i) sub(i % (D4 + 1), pA, 4 * i);
      kstep(nk4 - 1, ring[NA % (D4 + 1)]);             // the last two elements and the two norm slots
    } else {
    int t = 0;
    int kc[D4 + 1], kn[D4 + 1];
#pragma unroll
    for (int u = 0; u <= D4; ++u) kc[u] = koff[4 * min(u, nk4 - 1) + lrow];
    for (; t + D4 + 1 <= nfast; t += D4 + 1) {
#pragma unroll
      for (int u = 0; u <= D4; ++u) kn[u] = koff[4 * min(t + D4 + 1 + u, nk4 - 1) + lrow];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u <= D4; ++u) {
        ldz(min(t + u + D4, nk4 - 1), ring[(u + D4) % (D4 + 1)]);
        kstep_fast(kc[u], ring[u]);
      }
#pragma unroll
      for (int u = 0; u <= D4; ++u) kc[u] = kn[u];
    }
    for (; t + D4 + 1 <= nk4; t += D4 + 1) {
#pragma unroll
      for (int u = 0; u <= D4; ++u) {
        ldz(min(t + u + D4, nk4 - 1), ring[(u + D4) % (D4 + 1)]);
        kstep(t + u, ring[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < D4; ++u)
      if (t + u < nk4) kstep(t + u, ring[u]);
    }
#pragma unroll
    for (int c = 0; c < MAXF; ++c) {
      if (c < nfw) {
        double kv[FNS * 4];   // this fragment's accumulator values: their exponentials interleaved
#pragma unroll
        for (int y = 0; y < FNS; ++y)
#pragma unroll
          for (int v = 0; v < 4; ++v) kv[y * 4 + v] = kacc[c][y][v];
        if (BT == 0) {
          exp2_n<FNS * 4>(kv);
        } else {
          double n1[FNS * 4], n2[FNS * 4];
#pragma unroll
          for (int y = 0; y < FNS; ++y)
#pragma unroll
            for (int v = 0; v < 4; ++v) { n1[y * 4 + v] = xnv[y]; n2[y * 4 + v] = znv[BT == 0 ? 0 : c][v]; }
          BaseKernel bk;
          bk.type = a.bk.type; bk.variance = a.bk.variance; bk.p1 = a.bk.p1; bk.p2 = a.bk.p2;
          bk.template eval_n<BT, FNS * 4>(kv, n1, n2);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int m = 16 * fr[c] + lrow + 4 * v;
#pragma unroll
          for (int y = 0; y < FNS; ++y) strip[(4 * fr[c] + v) * QS + (sp * FNS + y) * 64 + lane] = (m < a.M) ? kv[y * 4 + v] : 0.0;
        }
      }
    }
  }
  CF_TR(2)
  __syncthreads();
  CF_TR(3)
  // training step: the reverse pass reads K_uf and A1 from HBM (k-major [Mp][ldk], column j)
  auto store_strip = [&](double* __restrict__ out) {
    for (int idx = tid; idx < Mp * BN; idx += NT) {
      const int m = idx / BN, c = idx - m * BN;
      const int j = j0 + c;
      if (j <= jmax) out[(long)m * a.ldk + j] = strip[(m >> 2) * QS + (c >> 4) * 64 + (m & 3) * 16 + (c & 15)];
    }
  };
  if (a.Kuf_out && sq == 0) store_strip(a.Kuf_out);

  // ---- phase 2: A1 = inv(L) K_uf (lower-triangular W: fragment f needs k-tiles 0 .. f), FNS column fragments per wave ----
  const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.LinvT), 0, Mp * Mp * 8, 0x00020000);
  d4 a1[MAXF][FNS];
#pragma unroll
  for (int c = 0; c < MAXF; ++c)
#pragma unroll
    for (int y = 0; y < FNS; ++y) a1[c][y] = d4{0.0, 0.0, 0.0, 0.0};
  {
    double s1acc[FNS];
#pragma unroll
    for (int y = 0; y < FNS; ++y) s1acc[y] = 0.0;
    d4 acc[FNS];
    auto ldb = [&](int kt, int q, double (&dst)[FNS]) {
      const double* b = strip + (kt * 4 + q) * QS + lane;
#pragma unroll
      for (int y = 0; y < FNS; ++y) dst[y] = b[(sp * FNS + y) * 64];
    };
    auto mf = [&](double w, const double (&b)[FNS]) {
#pragma unroll
      for (int y = 0; y < FNS; ++y) acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, b[y], acc[y], 0, 0, 0);
    };
    auto tile = [&](int kt, int kt_next, const double (&w)[4], double (&b0)[FNS]) {
      double b1[FNS], b2[FNS], b3[FNS];
      ldb(kt, 1, b1);
      mf(w[0], b0);
      ldb(kt, 2, b2);
      mf(w[1], b1);
      ldb(kt, 3, b3);
      mf(w[2], b2);
      ldb(kt_next, 0, b0);
      mf(w[3], b3);
    };
#pragma unroll
    for (int c = 0; c < MAXF; ++c) {
      if (c < nfw) {
        const int f = fr[c];
        const int fo = 16 * f * 8;
        const int n1 = f + 1;   // tiles 0 .. f
#pragma unroll
        for (int y = 0; y < FNS; ++y) acc[y] = d4{0.0, 0.0, 0.0, 0.0};
        double ring[CF_D + 1][4], b0[FNS];
#pragma unroll
        for (int q = 0; q < 4; ++q) ring[CF_D][q] = 0.0;
#pragma unroll
        for (int u = 0; u < CF_D; ++u) ldw(lrs, fo + min(u, f) * 16 * Mp * 8, ring[u]);
        ldb(0, 0, b0);
        int kt = 0;
        for (; kt + CF_D + 1 <= n1; kt += CF_D + 1) {   // full groups: no conditionals, the load counters stay exact
#pragma unroll
          for (int u = 0; u <= CF_D; ++u) {
            ldw(lrs, fo + min(kt + u + CF_D, f) * 16 * Mp * 8, ring[(u + CF_D) % (CF_D + 1)]);
            tile(kt + u, min(kt + u + 1, f), ring[u], b0);
          }
        }
#pragma unroll
        for (int u = 0; u < CF_D; ++u) {
          if (kt + u < n1) tile(kt + u, min(kt + u + 1, f), ring[u], b0);
        }
#pragma unroll
        for (int y = 0; y < FNS; ++y) {
          a1[c][y] = acc[y];
#pragma unroll
          for (int v = 0; v < 4; ++v) s1acc[y] = fma(acc[y][v], acc[y][v], s1acc[y]);
        }
      }
    }
    CF_TR(4)
    __syncthreads();   // every wave is done reading K_uf (and the images)
#pragma unroll
    for (int y = 0; y < FNS; ++y) {   // sum over this wave's rows of A1^2, per column: joined over the team in phase 4
      double s = s1acc[y];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (lrow == 0) aux[wm * BN + (sp * FNS + y) * 16 + lcol] = s;
    }
  }
#pragma unroll
  for (int c = 0; c < MAXF; ++c) {
    if (c < nfw) {
#pragma unroll
      for (int y = 0; y < FNS; ++y)
#pragma unroll
        for (int v = 0; v < 4; ++v) strip[(4 * fr[c] + v) * QS + (sp * FNS + y) * 64 + lane] = a1[c][y][v];
    }
  }
  __syncthreads();   // A1 published
  CF_TR(5)
  if (a.A1_out && sq == 0) store_strip(a.A1_out);
  if (mode == 1) {
    // the strip as it lies in LDS and the [TW][BN] partial sums behind it, by coherent (sc1) stores another XCD's loads see (chol_fused.hip: ldg / stg);
    // acknowledged before the barrier in front of the flag
    double* __restrict__ dst = a.pre_buf + (long)pre_slot * a.pre_stride;
    for (int i0 = 0; i0 < Mp * BN; i0 += 8 * NT) {
      double t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = strip[min(i0 + e * NT + tid, Mp * BN - 1)];
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (i0 + e * NT + tid < Mp * BN) __hip_atomic_store(dst + i0 + e * NT + tid, t[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int i = tid; i < TW * BN; i += NT) __hip_atomic_store(dst + Mp * BN + i, aux[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) __hip_atomic_store(a.pre_flag + pre_slot, a.pre_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  } else {
    // ---- A1 of the strip and its partial sums from the workgroup that ran phases 0 - 2 (dealt earlier: it is running or done) ----
    if (tid == 0) {
      // (its holder is running and waits for nobody, so this ends within a prologue's time; bounded all the same -- a launch must not hang the device --
      // and a strip that gave up says so: its samples come out as NaNs)
      int ok = 0;
      for (int spin = 0; spin < (1 << 24) && !ok; ++spin) {
        ok = __hip_atomic_load(a.pre_flag + pre_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.pre_epoch ? 1 : 0;
        if (!ok) __builtin_amdgcn_s_sleep(2);
      }
      ticket[1] = ok;
    }
    __syncthreads();
    pre_fail = ticket[1] == 0;
    const double* __restrict__ src = a.pre_buf + (long)pre_slot * a.pre_stride;
    for (int i0 = 0; i0 < Mp * BN; i0 += 8 * NT) {
      double t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = __hip_atomic_load(src + min(i0 + e * NT + tid, Mp * BN - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (i0 + e * NT + tid < Mp * BN) strip[i0 + e * NT + tid] = t[e];
    }
    for (int i = tid; i < TW * BN; i += NT) aux[i] = __hip_atomic_load(src + Mp * BN + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    CF_TR(5)
  }
  if (mode != 1) {

  // ---- phase 3: T_r = G_r^T A1 for r = r0, r0 + rstep, ... (r0 = sp, rstep = NS for a whole strip) (upper-triangular W: fragment f needs k-tiles f .. nf-1): one flat
  // stream over (r, fragment, k-tile), all FN column fragments.  s2 of this wave's i-th output is parked in the lanes with
  // lrow == (i & 3) of keep[i >> 2][.]
  constexpr int KEEP = (16 / NS + 3) / 4;
  double keep[KEEP][FN];
#pragma unroll
  for (int i = 0; i < KEEP; ++i)
#pragma unroll
    for (int y = 0; y < FN; ++y) keep[i][y] = 0.0;
  d4 acc[FN];
#pragma unroll
  for (int y = 0; y < FN; ++y) acc[y] = d4{0.0, 0.0, 0.0, 0.0};
  auto ldb = [&](int kt, int q, double (&dst)[FN]) {
    const double* b = strip + (kt * 4 + q) * QS + lane;
#pragma unroll
    for (int y = 0; y < FN; ++y) dst[y] = (ABL & 2) ? (double)(lane + y + q) : b[y * 64];
  };
  auto mf = [&](double w, const double (&b)[FN]) {
#pragma unroll
    for (int y = 0; y < FN; ++y) acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, b[y], acc[y], 0, 0, 0);
  };
  // one k-tile: b0 holds sub-step 0 of tile kt on entry and of tile kt_next on exit -- the LDS reads of every sub-step are
  // issued one sub-step ahead of the MFMAs that consume them, across the scalar bookkeeping between two tiles as well
  auto tile = [&](int kt, int kt_next, const double (&w)[4], double (&b0)[FN]) {
    double b1[FN];
    ldb(kt, 1, b1);
    CF_SB;
    mf(w[0], b0);
    CF_SB;
    ldb(kt, 2, b0);
    CF_SB;
    mf(w[1], b1);
    CF_SB;
    ldb(kt, 3, b1);
    CF_SB;
    mf(w[2], b0);
    CF_SB;
    ldb(kt_next, 0, b0);
    CF_SB;
    mf(w[3], b1);
    CF_SB;
  };
  const int r0 = sq + SQ * sp, rstep = SQ * NS;
  const int nr = r0 < R ? (R - 1 - r0) / rstep + 1 : 0;   // outputs of this team
  if (a.G && nfw > 0 && nr > 0) {
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.G), 0, R * Mp * Mp * 8, 0x00020000);
    int steps_per_r = 0;
    for (int c = 0; c < nfw; ++c) steps_per_r += nf - frag_of(wm, TW, c);
    const int total = nr * steps_per_r;
    // load cursor / compute cursor over the flat sequence of (r, fragment c, k-tile); the load cursor stops on the last tile
    const int f0 = frag_of(wm, TW, 0);
    int lr = r0, lc = 0, lf = f0, lk = f0, lleft = total - 1;
    int ci = 0, cc = 0, ck = f0, cleft = total - 1;
    auto lsoff = [&]() { return ((lr * Mp + lk * 16) * Mp + 16 * lf) * 8; };
    auto ladv = [&]() {
      if (lleft > 0) {
        --lleft;
        if (++lk >= nf) {
          if (++lc == nfw) { lc = 0; lr += rstep; }
          lf = frag_of(wm, TW, lc);
          lk = lf;
        }
      }
    };
    double s2acc[FN];
#pragma unroll
    for (int y = 0; y < FN; ++y) s2acc[y] = 0.0;
    double ring[CF_D + 1][4], b0[FN];
#pragma unroll
    for (int q = 0; q < 4; ++q) ring[CF_D][q] = 0.0;
#pragma unroll
    for (int u = 0; u < CF_D; ++u) { ldw(grs, lsoff(), ring[u]); ladv(); }
    ldb(ck, 0, b0);
    auto step = [&](const double (&w)[4]) {
      const int kcur = ck;
      const bool frag_end = kcur == nf - 1;
      bool r_end = false;
      if (cleft > 0) {   // advance the compute cursor first: the next tile's first B sub-step is fetched under this tile's MFMAs
        --cleft;
        if (frag_end) {
          if (++cc == nfw) { cc = 0; r_end = true; }
          ck = frag_of(wm, TW, cc);
        } else {
          ++ck;
        }
      } else {
        r_end = true;
      }
      tile(kcur, ck, w, b0);
      if (frag_end) {   // fragment done: fold its rows into the column sums of squares
#pragma unroll
        for (int y = 0; y < FN; ++y) {
#pragma unroll
          for (int v = 0; v < 4; ++v) s2acc[y] = fma(acc[y][v], acc[y][v], s2acc[y]);
          // (starting a fragment's first MFMAs from the constant 0 in their C operand instead of these sixteen moves, on a second copy of the
          // k-tile body: +17 us on the cfg2 launch -- profiles/r06_fused_ab.txt)
          acc[y] = d4{0.0, 0.0, 0.0, 0.0};
        }
        if (r_end) {   // output done
#pragma unroll
          for (int y = 0; y < FN; ++y) {
            double s = s2acc[y];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
#pragma unroll
            for (int i = 0; i < KEEP; ++i) keep[i][y] = ((ci >> 2) == i && (ci & 3) == lrow) ? s : keep[i][y];
            s2acc[y] = 0.0;
          }
          ++ci;
        }
      }
    };
    int t = 0;
    for (; t + CF_D + 1 <= total; t += CF_D + 1) {   // full groups: no conditionals around the loads
      // issue priority falls with progress: the arbiter favours the oldest wave of a SIMD, which then finishes this phase 50 us (of 160) ahead of the
      // youngest and leaves it the pipe to itself at the end; a wave ahead yields.  The steps shorten towards the end (1/2, 3/4, 9/10 of the stream): what the
      // waves of a SIMD differ by when they finish is about half the last step.  cfg2 launch: this schedule 576 us, quarters 3..0 579, thirds 2..0 581
      // (profiles/r06_fused_ab.txt); rotating the priorities per group measured like the quarters, pinning the LDS reads with sched_group_barrier +1 %
      set_prio(100 * t < CF_P1 * total ? 3 : (100 * t < CF_P2 * total ? 2 : (100 * t < CF_P3 * total ? 1 : 0)));
#pragma unroll
      for (int u = 0; u <= CF_D; ++u) {
        if (!(ABL & 1)) ldw(grs, lsoff(), ring[(u + CF_D) % (CF_D + 1)]);
        ladv();
        step(ring[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < CF_D; ++u) {
      if (t + u < total) step(ring[u]);
    }
  }
  __builtin_amdgcn_s_setprio(3);
  CF_TR(6)

  // ---- mean = alpha^T A1: wave (y, g) = (wave % FN, wave / FN) takes column fragment y and the k-tiles g, g + KG, ... -------
  d4 macc = d4{0.0, 0.0, 0.0, 0.0};
  {
    const int y = wave % FN, g = wave / FN;
    const double* __restrict__ al = a.alpha + lcol;
    for (int kt = g; kt < nf; kt += KG) {
      double w[4], b[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        w[q] = al[(long)(kt * 16 + 4 * q + lrow) * a.Rp];
        b[q] = strip[(kt * 4 + q) * QS + y * 64 + lane];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) macc = __builtin_amdgcn_mfma_f64_16x16x4f64(w[q], b[q], macc, 0, 0, 0);
    }
  }
  CF_TR(7)
  __syncthreads();   // the strip is dead: reuse it as [TW][R][BN] s2 | [KG][16][BN] mean partials
  double* s1p = aux;
  double* s2p = smem;
  double* mup = s2p + TW * R * BN;
#pragma unroll
  for (int y = 0; y < FN; ++y) {
    const int c = y * 16 + lcol;
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
      const int r = r0 + rstep * (4 * i + lrow);
      if (r < R) s2p[(wm * R + r) * BN + c] = keep[i][y];
    }
  }
  {
    const int y = wave % FN, g = wave / FN;
#pragma unroll
    for (int v = 0; v < 4; ++v) mup[(g * 16 + lrow + 4 * v) * BN + y * 16 + lcol] = macc[v];
  }
  __syncthreads();

  CF_TR(8)
  RngMap rmap;
  rmap.W = a.rmap.W; rmap.Nl = a.rmap.Nl; rmap.Ng = a.rmap.Ng; rmap.lo = a.rmap.lo;
  // ---- phase 4: var, mean, sample in the N x (P*R) layout (column j, output r at j*R + r) ----------------------------
  for (int idx = tid; idx < BN * R; idx += NT) {
    const int c = fdiv(idx, R, a.inv_R), r = idx - c * R;
    const int j = j0 + c;
    if (j > jmax || (SQ > 1 && r % SQ != sq)) continue;
    double s1 = 0.0, s2 = 0.0, m = 0.0;
    for (int w = 0; w < TW; ++w) {
      s1 += s1p[w * BN + c];
      s2 += s2p[(w * R + r) * BN + c];
    }
    for (int g = 0; g < KG; ++g) m += mup[(g * 16 + r) * BN + c];
    const double v = (a.knn - s1) + s2;
    if (pre_fail) m = __builtin_nan("");
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
        const double zz = a.z ? a.z[o] : philox_normal(a.seed, a.stream_id, rng_index(rmap, o));
        a.out_sample[o] = m + zz * sqrt(v + a.jitter);
      }
    }
  }
  CF_TR(9)
  }   // mode != 1
  if (tr_slot >= 0 && lane == 0) a.trace[(tr_slot * 16 + wave) * 16 + 11] = (long long)wall_clock64();
  if (!a.persist) break;
  // the next strip: dealt by arrival (a.dyn: one counter per launch) -- two workgroups share a CU and the one launched first wins every arbitration
  // between them, so a fixed deal leaves the other with a strip and a half to run alone at the end (profiles/r06_fused_persistent_static_trace.txt)
  if (tid == 0) ticket[0] = a.dyn ? (a.pre_n > 0 ? 0 : (int)gridDim.x) + atomicAdd(a.dyn, 1) : strip_next + (int)gridDim.x;
  __syncthreads();   // (and the partial sums are read: the next strip's images may land on them)
  strip_next = __builtin_amdgcn_readfirstlane(ticket[0]);
  if (strip_next >= a.n_strips + a.pre_n * a.pre_sq) break;
  }
  if (threadIdx.x == 0) {
    if (cu_word >= 0) atomicSub(ap->cu_slots + cu_word, 1);
    // the last workgroup to leave puts the counters back for the next launch
    if (ap->persist && ap->dyn && atomicAdd(ap->dyn + 1, 1) == (int)gridDim.x - 1) {
      ap->dyn[1] = 0;
      __threadfence();
      atomicExch(ap->dyn, 0);
    }
  }
}

// the instantiated shapes: <FN, NS, MAXF, NT>
//   0: <4,2,2,1024>  Mp <= 256, 64-column strips, 16 waves in two teams       1: <4,1,2,512>  the same on 8 waves
//   2: <2,1,2,512>   Mp <= 256, 32-column strips (large images)               3: <1,1,2,512>  16-column strips
//   4: <2,1,2,768>   Mp <= 384 (12 waves)     5: <2,1,2,1024>  Mp <= 512      6: <1,1,4,1024>  Mp <= 1024
struct FusedShape { int FN, NS, MAXF, NT, max_nf; };
constexpr FusedShape kShapes[] = {{4, 2, 2, 1024, 16}, {4, 1, 2, 512, 16}, {2, 1, 2, 512, 16}, {1, 1, 2, 512, 16},
                                  {2, 1, 2, 768, 24},  {2, 1, 2, 1024, 32}, {1, 1, 4, 1024, 64},
                                  {2, 2, 2, 1024, 16}};   // 7: a 32-column strip on 16 waves, two teams splitting the outputs (few columns: a rank's shard)
constexpr int kNumShapes = sizeof(kShapes) / sizeof(kShapes[0]);

template <int FN, int NS, int MAXF, int NT>
int launch_fused(dcgp_ctx* ctx, const ConvFusedArgs& a, size_t lds) {
  const int BN = FN * 16;
  const long strips = ((long)a.Kc + BN - 1) / BN;
  const unsigned grid = (unsigned)(a.persist ? a.persist : (a.split_q > 1 ? a.split_first + (strips - a.split_first) * a.split_q : strips));
  static bool attr_done[64] = {};   // per device: a second ctx on another device of this process needs the opt-in too
  const int dv = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
  if (!attr_done[dv]) {   // more than 64 KB of dynamic LDS needs the opt-in
    hipFuncSetAttribute((const void*)conv_fused_kernel<FN, NS, MAXF, NT, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)conv_fused_kernel<FN, NS, MAXF, NT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done[dv] = true;
  }
  const bool rows50 = a.bk.type == 0 && a.f * a.C == 50 && (a.f & 1) && a.L == a.f * a.f * a.C && a.Lz == ((a.L + 2 + 3) & ~3) && !a.no_rows;
  if constexpr (FN == 4 && NS == 2) {   // (the patch-row instance only where such layers run: the 64-column strip on 16 waves)
    if (rows50) {
      static bool attr2[64] = {};
      if (!attr2[dv]) {
        hipFuncSetAttribute((const void*)conv_fused_kernel<FN, NS, MAXF, NT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr2[dv] = true;
      }
      hipLaunchKernelGGL((conv_fused_kernel<FN, NS, MAXF, NT, 2>), dim3(grid), dim3(NT), lds, ctx->stream, a);
      LAUNCH_CHECK(ctx);
      return DCGP_OK;
    }
  }
  if (a.bk.type == 0) hipLaunchKernelGGL((conv_fused_kernel<FN, NS, MAXF, NT, 0>), dim3(grid), dim3(NT), lds, ctx->stream, a);
  else hipLaunchKernelGGL((conv_fused_kernel<FN, NS, MAXF, NT, 1>), dim3(grid), dim3(NT), lds, ctx->stream, a);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

struct FusedPlan { int shape; size_t lds; int lds_main, lds_img; int split_q; };

// The partial last round.  A 1024-thread strip owns its CU, so `strips` workgroups take ceil(strips / CUs) strip times and the last round
// leaves CUs idle.  Where they are enough, its strips are shared by Q workgroups each: every one of them runs the sweep and the first product
// (~0.11 of a strip, `kFront`) and the outputs r = q, q + Q, ... of the R-batched product, whose teams take them in turn -- a part costs
// kFront + (1 - kFront) * (outputs of its busiest team) / (outputs of a whole strip's busiest team).  Returns the last round's cost in strip
// times (1 unshared) and the Q to use.  Measured on shards of the headline batch (tools/shape_try.py): 90 strips of 64 columns 198 -> 118 us
// with Q = 2, 360 strips of 32 columns on 16 waves 215 -> 174 us; the full batch (720 strips, 208 in the last round) has no CUs to share with.
double last_round(const dcgp_ctx* ctx, const FusedShape& sh, long strips, int R, bool has_g, int* q_out) {
  *q_out = 1;
  const long want = ctx->opt.fused_split;
  const int slots = ctx->n_cus > 0 ? ctx->n_cus : 256;
  const long rem = strips % slots;
  if (rem == 0) return 0.0;
  if (want == 0 || want == 1 || sh.NT != 1024 || !has_g || R < 2) return 1.0;
  constexpr double kFront = 0.11;
  const int whole = (R + sh.NS - 1) / sh.NS;
  const long qmax = slots / rem < R ? slots / rem : R;
  double best = 0.9;   // a split must save a tenth of a strip time to be worth its extra sweeps
  for (int q = 2; q <= qmax; ++q) {
    const int part = ((R + q - 1) / q + sh.NS - 1) / sh.NS;
    const double cost = kFront + (1.0 - kFront) * part / whole;
    if ((want > 1 && q <= want) || (want < 0 && cost < best - 1e-9)) { best = cost; *q_out = q; }
  }
  return *q_out > 1 ? best : 1.0;
}

// Prologues ahead.  A persistent launch of `slots` workgroups (one per CU) over `strips` strips runs ceil(strips / slots) strip times, and a partial
// round leaves slots - rem workgroups idle for a whole one (720 strips on 256 CUs: 3 rounds for 2.81 of work).  Sharing a strip's OUTPUTS between workgroups re-pays
// its sweep and first product (4h.6: no split wins).  Sharing its PROLOGUE does not: the partial round goes first, its spare workgroups run phases 0 - 2 of
// later strips (a sixth of a strip each) and leave A1 in memory (132 KB per strip, L2 / Infinity-Cache traffic); those strips then start at the second product.
// The items are dealt in list order to whichever workgroup is free; this returns the makespan of that deal in units of one output of the second product
// (sweep + first product 1.75, epilogue 0.3, hand-over 0.2 on either side: the phase times of profiles/r06_fused_phase_trace.txt at M = 256), for n_pre
// prologues ahead, a hand-over waiting for its prologue where the deal has it so.
double deal_makespan(long strips, int slots, long n_pre, int R) {
  const double pro = 1.75, epi = 0.3, io = 0.2;
  const double F = pro + R + epi, P = pro + io, C = io + R + epi;
  if (n_pre <= 0) return (double)((strips + slots - 1) / slots) * F;
  const long rem = strips % slots;
  std::priority_queue<double, std::vector<double>, std::greater<double>> free_at;
  for (int i = 0; i < slots; ++i) free_at.push(0.0);
  std::vector<double> ready((size_t)n_pre, 0.0);
  double end = 0.0;
  auto give = [&](double cost, double not_before) {
    double t = free_at.top();
    free_at.pop();
    if (t < not_before) t = not_before;
    t += cost;
    free_at.push(t);
    if (t > end) end = t;
    return t;
  };
  for (long i = 0; i < rem; ++i) give(F, 0.0);
  for (long i = 0; i < n_pre; ++i) ready[(size_t)i] = give(P, 0.0);
  for (long i = 0; i < strips - rem - n_pre; ++i) give(F, 0.0);
  for (long i = 0; i < n_pre; ++i) give(C, ready[(size_t)i]);
  return end;
}
// Few strips (a rank's shard of a strongly-scaled batch: strips < 1.5 rounds of the CUs), every one handed over: `strips` prologue items first, then every
// strip's outputs as SQ parts (part q: r = q, q + SQ, ...; a team of a part runs its outputs in turn) -- what sharing a strip between workgroups always wanted,
// without re-paying sweep and first product per part.  Makespan of the counter's deal in the units of deal_makespan.
double deal_makespan_parts(long strips, int slots, int SQ, int R, int NS) {
  const double pro = 1.75, epi = 0.3, io = 0.2, P = pro + io;
  std::priority_queue<double, std::vector<double>, std::greater<double>> free_at;
  for (int i = 0; i < slots; ++i) free_at.push(0.0);
  std::vector<double> ready((size_t)strips, 0.0);
  double end = 0.0;
  auto give = [&](double cost, double not_before) {
    double t = free_at.top();
    free_at.pop();
    if (t < not_before) t = not_before;
    t += cost;
    free_at.push(t);
    if (t > end) end = t;
    return t;
  };
  for (long i = 0; i < strips; ++i) ready[(size_t)i] = give(P, 0.0);
  for (long i = 0; i < strips; ++i)
    for (int q = 0; q < SQ; ++q) {
      const int nr = q < R ? (R - 1 - q) / SQ + 1 : 0;
      const double work = NS == 2 ? 2.0 * ((nr + 1) / 2) : (double)nr;   // (two teams: an output costs its team two units)
      give(io + work + epi, ready[(size_t)i]);
    }
  return end;
}
// SQ for such a launch (0: not worth it / not wanted), given what the launch would cost without (`legacy_units`).
// MEASURED (tools/parts_try.py, the 4 / 8 / 16-image shards of the headline batch): correct and bit-identical, and SLOWER than the launches it would replace at
// every shard and every SQ -- 4 images 109 us (180 strips of 32 columns, one round) against 120-148 us as parts, 8 images 169 against 185-259, 16 images 303-308
// against 302-421.  A part pays its ticket, the flag, the fetch of the strip, the mean product, two barriers of partial sums and the epilogue (~10 us) for 8-15 us of
// second product; the simulated deal prices that at 0.5 of an output.  So the deal below never chooses parts by itself any more (fused_parts = -1 is "off"); the
// form stays reachable through fused_parts = q for tests/test_gpu_ops.py and for a part that is made cheaper one day.
int plan_parts(const dcgp_ctx* ctx, long strips, int slots, int R, int NS, double legacy_units) {
  const long want = ctx->opt.fused_parts;   // -1 / 0: off, q > 0: this SQ; -2: chosen by the simulated deal (A/B)
  if (want == 0 || want == -1 || R < 2 || strips <= 0 || 2 * strips > 3L * slots) return 0;
  if (want > 0) return (int)std::min<long>(want, R);
  double best = legacy_units * 0.9;   // a tenth better, or the plain launch stays
  int best_q = 0;
  for (int q = 2; q <= R; ++q) {
    const double t = deal_makespan_parts(strips, slots, q, R, NS);
    if (t < best - 1e-9) { best = t; best_q = q; }
  }
  return best_q;
}
// the number of prologues ahead for a persistent launch (0: none)
long plan_prologues(const dcgp_ctx* ctx, long strips, int slots, int R) {
  const long want = ctx->opt.fused_pre;
  const long rem = strips % slots, q = strips / slots;
  if (want == 0 || rem == 0 || q < 1 || q > 16 || R < 2) return 0;
  static std::mutex mu;
  static std::map<std::array<long, 4>, long> memo;
  const std::array<long, 4> key = {strips, (long)slots, (long)R, want};
  std::lock_guard<std::mutex> lock(mu);
  auto it = memo.find(key);
  if (it != memo.end()) return it->second;
  const long per = want > 0 ? want : (long)((1.75 + R + 0.3) / 1.95);   // prologues a spare workgroup runs in one strip time
  const long most = std::min<long>((slots - rem) * per, strips - rem);
  const long step = std::max<long>(slots / 8, 1);
  long best_n = 0;
  double best = deal_makespan(strips, slots, 0, R) * (want > 0 ? 2.0 : 0.98);   // chosen: a deal must save 2 %; forced: the best non-zero count
  for (long n = step; n <= most; n += step) {
    const double t = deal_makespan(strips, slots, n, R);
    if (t < best - 1e-9) { best = t; best_n = n; }
  }
  if (memo.size() > 256) memo.clear();
  memo[key] = best_n;
  return best_n;
}

// the first instantiated shape (widest strip, most waves) that covers Mp and whose LDS footprint fits
bool plan_fused(const dcgp_ctx* ctx, const ConvFusedArgs& a, FusedPlan* p) {
  const int force = (int)ctx->opt.fused_shape;   // A/B experiments (-1: none)
  const int nf = a.Mp / 16;
  if (a.Rp != 16 || a.R > 16 || a.Mp > 1024 || a.Mp % 16) return false;
  if (a.Kc >= (1 << 23) || a.HWC >= (1 << 23)) return false;   // the kernel's index arithmetic (fdiv) is exact below 2^23: larger layers take the sweep + GEMM route
  // M > 256: the 32- / 16-column strips LDS leaves room for re-fetch the A operands 2 - 4 x as often per MFMA and measure
  // 2 % (M = 384) to 16 % (M = 1024) behind the sweep + 128 x 128-tile GEMM route (87 % of the MFMA peak there); opt-in
  if (nf > 16 && force < 0 && !ctx->opt.fused_large) return false;
  // Among the shapes that fit, the one whose busiest CU carries the fewest columns: workgroups go round the 256 CUs, a CU works
  // through ceil(strips / 256) strips of BN columns at a rate that does not depend on BN (narrow strips share the CU), so few
  // columns -- a shard of a strongly-scaled batch -- are better cut into narrower strips (4 images x 10 samples x 144 patches:
  // 90 strips of 64 keep 90 CUs busy for a full strip time, 180 strips of 32 keep 180 busy for half of it).  The wider strip
  // wins ties: fewer A-operand fetches per MFMA (measured 3 % / 6 % behind at 32 / 16 columns on the full batch).
  double best = 0.0;
  bool found = false;
  for (int i = 0; i < kNumShapes; ++i) {
    const FusedShape& sh = kShapes[i];
    if (force >= 0 && i != force) continue;
    if (nf > sh.max_nf) continue;
    if (sh.max_nf > 16 && nf <= 16 && force < 0) continue;   // the many-wave shapes are for the large matrices
    if (i == 1 && force < 0) continue;                        // (the 8-wave form of shape 0: A/B experiments only)
    const int BN = sh.FN * 16, W = sh.NT / 64, TW = W / sh.NS, KG = W / sh.FN;
    const int nimg = (BN - 1) / a.P + 2;           // images a strip can touch
    const long fin = (long)(TW * a.R + KG * 16) * BN;
    const long main_d = (long)a.Mp * BN > fin ? (long)a.Mp * BN : fin;
    long img_d = ((long)nimg * a.HWC + 1) & ~1L;
    if (img_d < (long)TW * BN) img_d = (long)TW * BN;
    const long bytes = (main_d + img_d + BN + 2) * 8 + (long)(a.Lz > a.Lp ? a.Lz : a.Lp) * 4;
    if (bytes > 160 * 1024) continue;
    const long strips = a.Kc > 0 ? ((long)a.Kc + BN - 1) / BN : 1;
    // shape 7 (32 columns on 16 waves, the outputs split over two teams): a strip's latency is what a launch of one round costs, and
    // the second team shortens it (a 4-image shard of the headline batch: 0.297 -> 0.290 ms per step); over several rounds the eight-wave
    // form's two strips per CU do better (8 images: 0.398 against 0.384)
    if (i == 7 && force < 0 && strips > 512) continue;
    int q = 1;
    const int slots = ctx->n_cus > 0 ? ctx->n_cus : 256;
    const double rounds = sh.NT == 1024 ? (double)(strips / slots) + last_round(ctx, sh, strips, a.R, a.G != nullptr, &q) : (double)((strips + 255) / 256);
    const double cost = rounds * BN * (i == 7 ? 0.97 : (sh.FN == 4 ? 1.0 : (sh.FN == 2 ? 1.03 : 1.12)));
    if (found && cost >= best) continue;
    found = true; best = cost;
    p->shape = i; p->lds = (size_t)bytes; p->lds_main = (int)main_d; p->lds_img = (int)img_d; p->split_q = q;
  }
  return found;
}

}  // namespace

// debugging aid (tools/fused_trace.py): device buffer of 8 x 16 x 16 int64 that the fused layer kernel stamps its phases into
extern "C" int dcgp_debug_set_fused_trace(dcgp_ctx* ctx, long long* buf_dev) {
  if (!ctx) return DCGP_ERR_ARG;
  ctx->fused_trace = buf_dev;   // per ctx: goes away with it (nullptr switches the stamps off)
  return DCGP_OK;
}

bool conv_fused_ok(const dcgp_ctx* ctx, const ConvFusedArgs& a) {
  const bool off = ctx->opt.no_fused_layer != 0;   // A/B switch (tests flip it through dcgp_ctx_set_option): the unfused sweep + GEMM route
  FusedPlan p;
  return !off && plan_fused(ctx, a, &p);
}

int conv_fused(dcgp_ctx* ctx, const ConvFusedArgs& a_in) {
  if (a_in.Kc <= 0) return DCGP_OK;
  FusedPlan p;
  if (!plan_fused(ctx, a_in, &p)) return ctx_fail(ctx, DCGP_ERR_ARG, "conv_fused: layer shape not supported (M = %d, R = %d)", a_in.M, a_in.R);
  if ((long)a_in.R * a_in.Mp * a_in.Mp * 8 >= (1L << 31)) return ctx_fail(ctx, DCGP_ERR_ARG, "conv_fused: G exceeds 2 GiB");
  ConvFusedArgs a = a_in;
  a.lds_main = p.lds_main; a.lds_img = p.lds_img;
  a.trace = ctx->fused_trace;
  a.no_rows = ctx->opt.sweep_no_rows ? 1 : 0;
  a.inv_HWC = 1.0f / (float)a.HWC; a.inv_nmod = 1.0f / (float)a.n_mod; a.inv_P = 1.0f / (float)a.P; a.inv_Wo = 1.0f / (float)a.Wo; a.inv_R = 1.0f / (float)a.R;
  const long strips = ((long)a.Kc + kShapes[p.shape].FN * 16 - 1) / (kShapes[p.shape].FN * 16);
  const int n_cus = ctx->n_cus > 0 ? ctx->n_cus : 256;
  // workgroups a CU holds: LDS (160 KB) and wave slots (the kernels are held to 128 registers: 16 waves of 64 per CU)
  const long per_cu = std::min<long>(160 * 1024 / (long)p.lds, 1024 / kShapes[p.shape].NT);
  // chosen (-1): where a workgroup owns its CU and no strip of the last round is shared.  What it buys is the deal, not the persistence: strips handed out
  // by a device counter to whichever workgroup is free 572 us at cfg2, dealt by a fixed stride 576 -- as many as one workgroup per strip takes
  // (profiles/r06_fused_ab.txt)
  const long want = ctx->opt.fused_persist;
  bool persist = per_cu >= 1 && strips > per_cu * n_cus && (want > 0 || (want < 0 && per_cu == 1 && p.split_q == 1 && !a.Kuf_out && !a.A1_out));
  // few strips: all of them handed over, their outputs dealt as parts (deal_makespan_parts)
  int parts_sq = 0;
  if (!persist && per_cu == 1 && a.G && !a.Kuf_out && !a.A1_out && want != 0 && want != 2) {
    const FusedShape& sh = kShapes[p.shape];
    int q_legacy = 1;
    const double last = last_round(ctx, sh, strips, a.R, true, &q_legacy);
    const double legacy = ((double)(strips / n_cus) + (strips % n_cus ? last : 0.0)) * (1.75 + a.R + 0.3);
    parts_sq = plan_parts(ctx, strips, n_cus, a.R, sh.NS, legacy);
    if (parts_sq > 1) persist = true;
  }
  if (persist) {
    a.persist = (int)(per_cu * n_cus);
    a.n_strips = (int)strips;
    if (want != 2) {   // (2: the fixed deal blockIdx, blockIdx + grid, ... -- A/B)
      const std::string nm = "fused_dyn" + ctx->ws_tag;   // steps in flight on the two banks run this kernel side by side: a counter pair each
      const bool fresh = ctx->ws.find(nm) == ctx->ws.end();
      a.dyn = static_cast<int*>(ws_get(ctx, nm, 2 * sizeof(int)));
      if (!a.dyn) return DCGP_ERR_ALLOC;
      if (fresh) HIP_TRY(ctx, hipMemsetAsync(a.dyn, 0, 2 * sizeof(int), ctx->stream));
    }
    if (per_cu == 1 && a.dyn && a.G) {
      const int TW = kShapes[p.shape].NT / 64 / kShapes[p.shape].NS, BN = kShapes[p.shape].FN * 16;
      const long n_pre = parts_sq > 1 ? strips : plan_prologues(ctx, strips, a.persist, a.R);
      if (n_pre > 0) {
        a.pre_n = (int)n_pre;
        a.pre_first = parts_sq > 1 ? 0 : (int)(strips % a.persist);
        a.pre_sq = parts_sq > 1 ? parts_sq : 1;
        a.pre_stride = (long)a.Mp * BN + (long)TW * BN;
        const std::string nb = "fused_pre_buf" + ctx->ws_tag, nf = "fused_pre_flag" + ctx->ws_tag;
        const size_t fbytes = (size_t)n_pre * sizeof(unsigned);
        const bool fresh = ctx->ws.find(nf) == ctx->ws.end() || ctx->ws[nf].second < fbytes;
        a.pre_buf = static_cast<double*>(ws_get(ctx, nb, (size_t)n_pre * a.pre_stride * sizeof(double)));
        a.pre_flag = static_cast<unsigned*>(ws_get(ctx, nf, fbytes));
        if (!a.pre_buf || !a.pre_flag) return DCGP_ERR_ALLOC;
        unsigned& epoch = ctx->fused_pre_epochs[nf];
        if (fresh || epoch == 0xffffffffu) {   // flags compare equal to the launch's epoch: a fresh (or wrapped) area starts from zero
          HIP_TRY(ctx, hipMemsetAsync(a.pre_flag, 0, ctx->ws[nf].second, ctx->stream));
          epoch = 0;
        }
        a.pre_epoch = ++epoch;
      }
    }
    if (per_cu > 1) {
      const long us = ctx->opt.fused_stagger >= 0 ? ctx->opt.fused_stagger : 40;
      a.stagger = (int)(us * 100);
      bool fresh = ctx->ws.find("fused_cu_slots") == ctx->ws.end();
      a.cu_slots = static_cast<int*>(ws_get(ctx, "fused_cu_slots", 1024 * sizeof(int)));
      if (!a.cu_slots) return DCGP_ERR_ALLOC;
      if (fresh) HIP_TRY(ctx, hipMemsetAsync(a.cu_slots, 0, 1024 * sizeof(int), ctx->stream));
    }
  } else if (p.split_q > 1 && !a.trace) {
    a.split_q = p.split_q;
    a.split_first = (int)(strips - strips % n_cus);
  }
  ScopedTimer t(ctx, "conv_fused");
#ifdef DCGP_EXPERIMENTS
  const int abl = (int)ctx->opt.fused_abl;   // timing build only (make EXPERIMENTS=1): wrong results
  if (abl && p.shape == 0 && a.bk.type == 0) {
    const unsigned grid = (unsigned)((a.Kc + 63) / 64);
#define CF_ABL(X)                                                                                                                       \
  case X:                                                                                                                               \
    hipFuncSetAttribute((const void*)conv_fused_kernel<4, 2, 2, 1024, 0, X>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((conv_fused_kernel<4, 2, 2, 1024, 0, X>), dim3(grid), dim3(1024), p.lds, ctx->stream, a);                        \
    break;
    switch (abl) { CF_ABL(1) CF_ABL(2) CF_ABL(3) CF_ABL(4) default: return ctx_fail(ctx, DCGP_ERR_ARG, "conv_fused: unknown DCGP_FUSED_ABL value %d", abl); }
#undef CF_ABL
    LAUNCH_CHECK(ctx);
    return DCGP_OK;
  }
#endif
  switch (p.shape) {
    case 0: return launch_fused<4, 2, 2, 1024>(ctx, a, p.lds);
    case 1: return launch_fused<4, 1, 2, 512>(ctx, a, p.lds);
    case 2: return launch_fused<2, 1, 2, 512>(ctx, a, p.lds);
    case 3: return launch_fused<1, 1, 2, 512>(ctx, a, p.lds);
    case 4: return launch_fused<2, 1, 2, 768>(ctx, a, p.lds);
    case 5: return launch_fused<2, 1, 2, 1024>(ctx, a, p.lds);
    case 6: return launch_fused<1, 1, 4, 1024>(ctx, a, p.lds);
    default: return launch_fused<2, 2, 2, 1024>(ctx, a, p.lds);
  }
}
