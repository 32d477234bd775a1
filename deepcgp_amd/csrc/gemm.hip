// gemm.hip -- the matrix-core workhorse of the conditional (conv_gp/conditionals.py:31-65).
//
//   C[i][j] = sum_k Wt[k][i] * B[k][j]        (fp64, v_mfma_f64_16x16x4_f64)
//
// Both operands are k-major so that every LDS fill is a plain row copy and every MFMA operand read
// is a 16-double contiguous ds_read_b64 group.  W may be lower/upper triangular (the triangular
// solves of the reference are applied as products with the inverted factor, and G_r^T A1 is an
// upper-triangular product): k tiles that are structurally zero are skipped per workgroup AND per
// 16-row fragment.  The epilogue optionally stores C and/or reduces sum_i C[i][j]^2 per column (the
// reduce_sum(square(A), 1) / reduce_sum(square(LTA), 1) of conditionals.py:40,65) so the
// R x M x (P*N) intermediate of the reference is never materialised.
//
// Tile: BM x BN output per workgroup, BK = 16; waves in a WAVES_M x WAVES_N grid, each owning
// FM x FN 16x16 accumulator fragments (4 f64 per lane each).  Double-buffered LDS, one barrier per k-tile: the
// 128 x 128 tile of 16 waves fills the other buffer with LDS-DMA loads issued before the MFMAs of the current
// k-tile; the smaller tiles stage through registers.
#include <cstdlib>

#include "common.h"
#include <type_traits>

namespace {

constexpr int BK = 16;

__host__ __device__ constexpr int lds_ld(int b) { return (b % 32 == 0) ? b + 16 : b + 32; }

template <int BM, int BN, int WAVES_M, int WAVES_N, int ABL = 0>
// second launch-bound argument = waves per SIMD the register budget must allow: it caps the kernel at 128 (64 for the
// 16-wave tile) unified registers, which also keeps the MFMA accumulators in VGPRs -- the AGPR form of
// v_mfma_f64_16x16x4_f64 measured ~1.6x slower on gfx950 (tools/mfma_peak.hip vs tools/fp64_mix.hip).
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, (WAVES_M * WAVES_N >= 16) ? 8 : 4) void gemm_tn_kernel(GemmArgs a, int n_col_tiles,
                                                                         int n_row_blocks) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int WMR = BM / WAVES_M, WNC = BN / WAVES_N;   // wave tile
  constexpr int FM = WMR / 16, FN = WNC / 16;
  constexpr int LDW = lds_ld(BM), LDB = lds_ld(BN);
  constexpr int W_CHUNKS = BK * BM / 2, B_CHUNKS = BK * BN / 2;   // 16-byte chunks per tile
  constexpr int NLW = (W_CHUNKS + NT - 1) / NT, NLB = (B_CHUNKS + NT - 1) / NT;
  static_assert(W_CHUNKS % NT == 0 || W_CHUNKS < NT, "W tile / thread mismatch");
  static_assert(B_CHUNKS % NT == 0, "B tile / thread mismatch");
  // One k-row of either operand tile = 1 KiB = one wavefront x 16 B: the 128 x 128 tile of 16 waves is filled by
  // LDS-DMA loads (buffer_load ... lds: wave-uniform LDS row base + lane * 16 B, so the row padding survives), no
  // staging registers and no ds_write pass.  Other shapes stage through registers.
  constexpr bool GLDS = BM == 128 && BN == 128 && NT == 1024 && !(ABL & 3);

  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Ws = smem;                       // [2][BK][LDW]
  double* Bs = smem + 2 * BK * LDW;        // [2][BK][LDB]

  // ---- XCD-aware decode: workgroups that share a B column strip (all batches / row blocks of one
  // column tile) get consecutive ids on ONE XCD (dispatch places linear id b on XCD b % 8).
  const int batch = a.nW * a.nB;
  const long nwg = (long)n_col_tiles * n_row_blocks * batch;
  long orig = blockIdx.x;
  long q = nwg / 8, r = nwg % 8, xcd = orig % 8;
  long wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
  if (a.rb_major) wgid = orig;   // ids stay interleaved over the XCDs: each XCD gets its share of the heavy blocks
  int ct, bz, rb;
  if (a.rb_major) {
    // ALL heavy row blocks first, then the lighter ones (longest-processing-time order): the light blocks fill the tail
    // while the last heavy ones run, where mixed orders end with a heavy block that started late.  No fixed-stride
    // heavy/light alternation either.  First used for launches of at most ~2 rounds; on the 14-round stage-3 launch it
    // beats the Thue-Morse mixing below by 5 % as well (535 -> 509 us), so it is the order of every triangular launch.
    const int per = n_col_tiles * batch;
    const int rbk = (int)(wgid / per), rem = (int)(wgid % per);
    ct = rem / batch;
    bz = rem % batch;
    rb = (a.tri == 1) ? n_row_blocks - 1 - rbk : rbk;
  } else {
    const int inner = n_row_blocks * batch;
    ct = (int)(wgid / inner);
    const int rem = (int)(wgid % inner);
    // heavy (long-k) row blocks first within a column strip
    bz = rem / n_row_blocks;
    rb = rem % n_row_blocks;
    if (a.tri == 1) rb = n_row_blocks - 1 - rb;
    // Row blocks of a triangular product carry graded work.  Workgroups are handed to shader engines / CUs round-robin
    // in id order, so a FIXED heavy,light,heavy,light... sequence parks all heavy blocks on the same engines and the
    // launch lasts as long as a heavy-only queue (measured in MFMA-only mode: 19.6 us lost per workgroup turn vs 7.7 us
    // with equal blocks).  Rotating the row-block order by the bit parity of the global (column tile, batch) index
    // (Thue-Morse) is balanced over every power-of-two stride.
    if (a.tri) rb = (rb + __popc(ct * batch + bz)) % n_row_blocks;
  }
  const int iw = bz / a.nB, ib = bz % a.nB;

  const int i0 = rb * BM, j0 = ct * BN;
  const double* __restrict__ Wt = a.Wt + (long)iw * a.wBatch;
  const double* __restrict__ Bm = a.B + (long)ib * a.bBatch;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lrow = lane >> 4, lcol = lane & 15;

  // k range of this workgroup (multiples of BK)
  int klo = 0, khi = a.Mk;
  if (a.tri == 1) khi = min(a.Mk, i0 + BM);
  if (a.tri == 2) klo = min(i0, a.Mk);
  if (a.b_lower) klo = max(klo, (j0 / BK) * BK);
  static_assert(FM == 1 || FM == 2, "one or two 16-row fragments per wave");
  auto frag_off = [&](int x) { return (x == 0 ? wm : 2 * WAVES_M - 1 - wm) * 16; };   // row offset of fragment x in the block

  d4 acc[FM][FN];
#pragma unroll
  for (int x = 0; x < FM; ++x)
#pragma unroll
    for (int y = 0; y < FN; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};

  // Global fetch through buffer descriptors: the operand slab of this workgroup is a raw buffer (4 SGPRs), the
  // k-tile advance is the scalar offset, and each thread keeps ONE constant 32-bit byte offset per chunk.  No
  // 64-bit pointer lives in VGPRs across the loops (the kernel is held under a 64-VGPR cap to keep 8 waves per
  // SIMD), and chunks outside the matrix (rows of C beyond Mi, columns beyond Kc) carry an out-of-range offset,
  // which the hardware bounds check turns into zeros.
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  constexpr unsigned OOB = 0x80000000u;   // >= any slab size (the host refuses slabs of 2 GiB and more)
  const int nkt = max(khi - klo, 0);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<double*>(Wt + (long)klo * a.ldw), 0, (int)((long)nkt * a.ldw * 8), 0x00020000);
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<double*>(Bm + (long)klo * a.ldb), 0, (int)((long)nkt * a.ldb * 8), 0x00020000);
  double2 rw[NLW], rbv[NLB];
  unsigned woff[NLW], boff[NLB];
#pragma unroll
  for (int c = 0; c < NLW; ++c) {
    const int ch = tid + c * NT, row = ch / (BM / 2), col = (ch % (BM / 2)) * 2;
    woff[c] = (ch < W_CHUNKS && i0 + col < a.Mi) ? (unsigned)(row * a.ldw + i0 + col) * 8u : OOB;
  }
#pragma unroll
  for (int c = 0; c < NLB; ++c) {
    const int ch = tid + c * NT, row = ch / (BN / 2), col = (ch % (BN / 2)) * 2;
    boff[c] = (j0 + col < a.Kc) ? (unsigned)(row * a.ldb + j0 + col) * 8u : OOB;
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably uniform: the LDS-DMA row base goes to M0
  auto load_tile = [&](int k0, int nbuf) {   // Mk is a multiple of BK (callers pad), so every k row of a tile exists
    const int sw = (k0 - klo) * a.ldw * 8, sb = (k0 - klo) * a.ldb * 8;
    if constexpr (GLDS) {
      typedef __attribute__((address_space(3))) void* lds_ptr;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr)(Ws + nbuf * BK * LDW + wave_u * LDW), 16, (int)woff[0], sw, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (lds_ptr)(Bs + nbuf * BK * LDB + wave_u * LDB), 16, (int)boff[0], sb, 0, 0);
    } else {
#pragma unroll
      for (int c = 0; c < NLW; ++c) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)woff[c], sw, 0);
        __builtin_memcpy(&rw[c], &v, 16);
      }
#pragma unroll
      for (int c = 0; c < NLB; ++c) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(brs, (int)boff[c], sb, 0);
        __builtin_memcpy(&rbv[c], &v, 16);
      }
    }
  };
  auto store_tile = [&](int buf) {
    if constexpr (GLDS) return;
    double* w = Ws + buf * BK * LDW;
    double* b = Bs + buf * BK * LDB;
#pragma unroll
    for (int c = 0; c < NLW; ++c) {
      int ch = tid + c * NT;
      if (ch < W_CHUNKS) {
        int row = ch / (BM / 2), col = (ch % (BM / 2)) * 2;
        *reinterpret_cast<double2*>(w + row * LDW + col) = rw[c];
      }
    }
#pragma unroll
    for (int c = 0; c < NLB; ++c) {
      int ch = tid + c * NT;
      int row = ch / (BN / 2), col = (ch % (BN / 2)) * 2;
      *reinterpret_cast<double2*>(b + row * LDB + col) = rbv[c];
    }
  };

  // ---- k loop, split into phases by which of this wave's fragments are structurally non-zero ------------------
  // A wave owns FM <= 2 fragments of 16 rows: fragment 0 = rows (wm) * 16, fragment 1 = rows (2 WAVES_M - 1 - wm) * 16
  // of the row block -- an early and a late one, so that inside a DIAGONAL block of a triangular product every wave
  // has the same number of fragment-steps (9 of 16 at BM = 128) and all waves of a workgroup finish together; with
  // contiguous 32-row wave tiles they finish 4 : 3 : 2 : 1 and the workgroup holds its LDS and wave slots until the
  // last one is done.  The skip is per fragment and, k being monotone, a sequence of at most five loops
  // (none | frag 0 | both | frag 1 | none) with wave-uniform bounds -- no branch inside a loop body.
  // LDS addresses of a k-tile (round 6): one per-lane base per A fragment and one for the B operand, moved to the current buffer once per k-tile; sub-step
  // and column fragment are compile-time offsets, and a wave's column fragments lie WAVES_N * 16 columns (512 bytes at the 16-wave tile) apart, so that
  // every offset is a multiple of ds_read2st64_b64's unit.  With the fragments side by side the body carried 9 v_add per 16 MFMAs (the B reads paired as
  // ds_read2_b64, whose 8-bit offsets wanted a fresh base per sub-step) -- VALU instructions issue in the fp64 MFMA's place.  (The body
  // instantiated per buffer, all offsets immediates: the two copies keep the accumulators in different registers and the 64-register cap spills.)
  const double* wl0[FM];
#pragma unroll
  for (int x = 0; x < FM; ++x) wl0[x] = Ws + lrow * LDW + lcol + frag_off(x);
  const double* bl0 = Bs + lrow * LDB + wn * 16 + lcol;
  int buf = 0;
  // a phase walks its k-tiles in PAIRS (current buffer, other buffer): the bases of both are fixed at its entry, so a k-tile carries no address arithmetic at all
  auto steps = [&](int kb, int ke, auto x0c, auto x1c) {
    constexpr int X0 = decltype(x0c)::value, X1 = decltype(x1c)::value;
    const double* wc[FM];
    const double* wo[FM];
#pragma unroll
    for (int x = 0; x < FM; ++x) { wc[x] = wl0[x] + buf * BK * LDW; wo[x] = wl0[x] + (buf ^ 1) * BK * LDW; }
    const double* bc = bl0 + buf * BK * LDB;
    const double* bo = bl0 + (buf ^ 1) * BK * LDB;
    auto tile = [&](int k0, int nbuf, const double* const (&wl)[FM], const double* bl) {
      const bool has_next = k0 + BK < khi;
      if (has_next && !(ABL & 1)) load_tile(k0 + BK, nbuf);
      if (X0 < X1) {
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
          double av[FM], bv[FN];
#pragma unroll
          for (int x = X0; x < X1; ++x) av[x] = (ABL & 8) ? (double)(kk + x) : wl[x][kk * LDW];
#pragma unroll
          for (int y = 0; y < FN; ++y) bv[y] = (ABL & 8) ? (double)(lane + y) : bl[kk * LDB + y * WAVES_N * 16];
#pragma unroll
          for (int x = X0; x < X1; ++x)
#pragma unroll
            for (int y = 0; y < FN; ++y)
              acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[x], bv[y], acc[x][y], 0, 0, 0);
        }
      }
      if (has_next && !(ABL & 2)) store_tile(nbuf);
      if (!(ABL & 4)) __syncthreads();
    };
    int k0 = kb;
    for (; k0 + BK < ke; k0 += 2 * BK) {
      tile(k0, buf ^ 1, wc, bc);
      tile(k0 + BK, buf, wo, bo);
    }
    if (k0 < ke) {
      tile(k0, buf ^ 1, wc, bc);
      buf ^= 1;
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using IF = std::integral_constant<int, FM>;
  // phase bounds: none [klo,b0) | frag 0 [b0,b1) | all [b1,b2) | frag FM-1 [b2,b3) | none [b3,khi)
  int b0 = klo, b1 = klo, b2 = khi, b3 = khi;
  const int r_first = i0 + frag_off(0), r_last = i0 + frag_off(FM - 1);
  if (a.tri == 2) {   // fragment live from k0 >= its first row on
    b0 = min(max(r_first, klo), khi);
    b1 = min(max(r_last, klo), khi);
  }
  if (a.tri == 1) {   // fragment live while k0 <= its last row
    b2 = max(min(r_first + 16, khi), klo);
    b3 = max(min(r_last + 16, khi), klo);
  }
  b0 = __builtin_amdgcn_readfirstlane(b0);   // wave-uniform by construction: keep the loop control scalar
  b1 = __builtin_amdgcn_readfirstlane(b1);
  b2 = __builtin_amdgcn_readfirstlane(b2);
  b3 = __builtin_amdgcn_readfirstlane(b3);
  if (klo < khi) {
    load_tile(klo, 0);
    store_tile(0);
  }
  __syncthreads();
  steps(klo, b0, I0{}, I0{});
  if (FM == 2) steps(b0, b1, I0{}, I1{});
  steps(b1, b2, I0{}, IF{});
  if (FM == 2) steps(b2, b3, I1{}, IF{});
  steps(b3, khi, I0{}, I0{});

  // ---- epilogue ----------------------------------------------------------------------------
  const int bzl = iw * a.nB + ib;
  if (a.C) {
    double* C = a.C + (long)bzl * a.cBatch;
#pragma unroll
    for (int x = 0; x < FM; ++x)
#pragma unroll
      for (int y = 0; y < FN; ++y) {
        int j = j0 + (wn + y * WAVES_N) * 16 + lcol;
        const double cs = (a.cscale && j < a.Kc) ? a.calpha * a.cscale[(long)bzl * a.csBatch + (long)j * a.csCol] : 1.0;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          int i = i0 + frag_off(x) + lrow + 4 * v;
          if (i < a.Mi && j < a.Kc) C[(long)i * a.ldc + j] = acc[x][y][v] * cs;
        }
      }
  }
  if (a.colsq) {
    // per-lane partial over this wave's rows, then over the 4 lane groups, then over WAVES_M via LDS
    __syncthreads();   // LDS tiles are dead; reuse as [WAVES_M][BN]
    double* red = smem;
#pragma unroll
    for (int y = 0; y < FN; ++y) {
      double s = 0.0;
#pragma unroll
      for (int x = 0; x < FM; ++x)
#pragma unroll
        for (int v = 0; v < 4; ++v) s += acc[x][y][v] * acc[x][y][v];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (lrow == 0) red[wm * BN + (wn + y * WAVES_N) * 16 + lcol] = s;
    }
    __syncthreads();
    if (tid < BN) {
      double s = 0.0;
#pragma unroll
      for (int m = 0; m < WAVES_M; ++m) s += red[m * BN + tid];
      int j = j0 + tid;
      if (j < a.Kc) a.colsq[(long)bzl * a.sBatch + (long)rb * a.sRowBlk + j] = s;
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int ABL = 0>
int launch(dcgp_ctx* ctx, const GemmArgs& a, int* nrb_out) {
  const int nct = (a.Kc + BN - 1) / BN, nrb = (a.Mi + BM - 1) / BM;
  if (nrb_out) *nrb_out = nrb;
  const long nwg = (long)nct * nrb * a.nW * a.nB;
  if (nwg == 0) return DCGP_OK;
  constexpr int NT = WAVES_M * WAVES_N * 64;
  size_t lds = (size_t)(2 * BK * lds_ld(BM) + 2 * BK * lds_ld(BN)) * sizeof(double);
  size_t red = (size_t)WAVES_M * BN * sizeof(double);
  if (red > lds) lds = red;
  GemmArgs k = a;
#ifdef DCGP_EXPERIMENTS
  const bool mixed = ctx->opt.rb_mixed != 0;   // timing build only: Thue-Morse mixed order for launches of many rounds
#else
  constexpr bool mixed = false;
#endif
  k.rb_major = a.tri != 0 && (nwg <= 1024 || !mixed);
  hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, WAVES_M, WAVES_N, ABL>), dim3((unsigned)nwg), dim3(NT), lds, ctx->stream,
                     k, nct, nrb);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

}  // namespace

// Timing builds (make EXPERIMENTS=1) read DCGP_GEMM_ABLATE: tuning / timing experiments (ablations 1..15 give wrong results);
// the shipped library has no such switch.
#ifdef DCGP_EXPERIMENTS
static int gemm_variant() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("DCGP_GEMM_ABLATE"); v = e ? atoi(e) : 0; }
  return v;
}
#else
static constexpr int gemm_variant() { return 0; }
#endif

// Row-block height for an Mi x Kc product with `batch` independent instances.  Large problems use 128-row tiles
// (best reuse); problems that would not even put one workgroup on every CU use shorter tiles -- their run time is
// the serial k-chain of ONE workgroup (Mk/16 steps of BM*128*16*2 flops on one CU), so shorter tiles shorten it.
int gemm_row_block(int Mi, int Kc, int batch) {
  if (Mi < 128) return Mi >= 64 ? 64 : (Mi >= 32 ? 32 : 16);
  const int v = gemm_variant();
  if (v == 128 || v == 64 || v == 32) return v;
  const long nct = (Kc + 127) / 128;
  for (int bm = 128; bm > 32; bm >>= 1)
    if (nct * ((Mi + bm - 1) / bm) * batch >= 512) return bm;
  return 32;
}

int gemm_tn(dcgp_ctx* ctx, const GemmArgs& a, int* nrb_out) {
  if (!a.Wt || !a.B || a.Mi <= 0 || a.Mk <= 0 || a.Kc <= 0) return ctx_fail(ctx, DCGP_ERR_ARG, "gemm_tn: bad args");
  // operands are fetched as 16-byte pairs: rows must be 16-byte aligned, and an odd column count needs
  // one column of (ignored) padding behind it
  if ((a.ldw & 1) || (a.ldb & 1) || (a.Mi & 1) || ((a.Kc & 1) && a.ldb <= a.Kc))
    return ctx_fail(ctx, DCGP_ERR_ARG, "gemm_tn: leading dimensions / extents must be even");
  // k-tiles are fetched whole through 32-bit-offset buffer descriptors
  if ((a.Mk % BK) || (long)a.Mk * a.ldw * 8 >= (1L << 31) || (long)a.Mk * a.ldb * 8 >= (1L << 31))
    return ctx_fail(ctx, DCGP_ERR_ARG, "gemm_tn: Mk must be a multiple of 16 and an operand slab (Mk x ld) under 2 GiB");
  switch (gemm_row_block(a.Mi, a.Kc, a.nW * a.nB)) {
    case 128:
      switch (gemm_variant()) {
#ifdef DCGP_EXPERIMENTS
        case 1: return launch<128, 128, 4, 2, 1>(ctx, a, nrb_out);
        case 7: return launch<128, 128, 4, 2, 7>(ctx, a, nrb_out);
        case 15: return launch<128, 128, 4, 2, 15>(ctx, a, nrb_out);
        case 100: return launch<128, 128, 4, 2>(ctx, a, nrb_out);
        case 101: return launch<128, 128, 4, 4, 1>(ctx, a, nrb_out);
        case 104: return launch<128, 128, 4, 4, 4>(ctx, a, nrb_out);
        case 107: return launch<128, 128, 4, 4, 7>(ctx, a, nrb_out);
        case 115: return launch<128, 128, 4, 4, 15>(ctx, a, nrb_out);
#endif
        default: return launch<128, 128, 4, 4>(ctx, a, nrb_out);   // 16 waves: +6% over 8 waves (more MFMA-phase waves per SIMD)
      }
    // short tiles = small problems whose run time is one workgroup's serial k-chain: spread each tile over 16
    // (8) waves with one or two accumulator fragments each so that the chain is a handful of MFMAs per k-step
    case 64: return launch<64, 128, 4, 4>(ctx, a, nrb_out);
    case 32: return launch<32, 128, 2, 8>(ctx, a, nrb_out);
    default: return launch<16, 128, 1, 8>(ctx, a, nrb_out);
  }
}
