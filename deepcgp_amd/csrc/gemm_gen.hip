// General strided fp64 GEMM on v_mfma_f64_16x16x4_f64 for the backward pass (grad.hip): every operand is addressed
// through (row stride, column stride, batch stride), so transposes and the [Kc][R] <-> [R][Kc] views of the
// backward products need no copies.  64 x 64 (or 128 x 128) output tile, 32 x 32 per wave, BK = 16, register prefetch of
// the next k tile, double-buffered LDS (one barrier per k tile).  Long contractions with a small output (d alpha, d G_r, d L: K = number of
// patch columns) are split along k into a partial buffer and summed in a fixed order -- no atomics, so gradients are
// reproducible run to run.  The forward path's tuned kernel is gemm.hip; this one trades peak rate for generality.
#include <cstdlib>

#include "gemm_gen.h"

namespace {

constexpr int GK = 16;
// LDS tile layouts.  An operand whose rows (columns) are contiguous in memory is staged [k][m] with row stride GT + 16
// (the two 16-lane halves of a ds_read_b64 hit disjoint banks); one whose contraction index is contiguous is staged
// [m][k] with row stride LDK = 17, so that each thread's consecutive k land next to each other and the fragment reads
// (16 rows x 4 k per wave) still spread over all banks.  Two buffers per operand: one barrier per k tile.

// GT x GT output tile on NT threads (NT / 64 waves, 32 x 32 per wave); each thread fetches EPT = GT * GK / NT elements of
// either operand per k tile.  <64, 256> is the general configuration; <128, 1024> halves the operand traffic per flop
// for the long contractions whose operands stream from the Infinity Cache / HBM (8 -> 16 flop per byte).
// AKF / BKF: the operand's contraction index is the contiguous one (fetch EPT consecutive k per thread), otherwise EPT
// consecutive rows (columns) per thread.  VEC: every EPT-element group is 16-byte aligned and contiguous (b128 loads).
template <int GT, int NT, int WT, bool AKF, bool BKF, bool VEC>
__global__ __launch_bounds__(NT, NT == 1024 ? 8 : 4) void gemm_gen_kernel(GenGemm g, int kchunk, double* part) {
  constexpr int LDK = GK + 1, EPT = GT * GK / NT, LDM = GT + 16, WF = WT / 16;
  constexpr int TILE = (GK * LDM > GT * LDK) ? GK * LDM : GT * LDK;
  constexpr int TPR = GK / EPT, TPK = GT / EPT, WAVES_N = GT / WT;
  static_assert(NT == WAVES_N * WAVES_N * 64, "one wave per WT x WT block of the tile");
  static_assert(EPT == 2 || EPT == 4, "fetch width");
  auto lds_a = [](int m, int k) { return AKF ? m * LDK + k : k * LDM + m; };
  auto lds_b = [](int m, int k) { return BKF ? m * LDK + k : k * LDM + m; };
  __shared__ __attribute__((aligned(16))) double As[2][TILE];
  __shared__ __attribute__((aligned(16))) double Bs[2][TILE];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int bz = blockIdx.z, b = bz % g.batch, sp = bz / g.batch;
  // lower_only launches enumerate the tiles on and below the diagonal only (blockIdx.x = bi (bi + 1) / 2 + bj): a
  // rectangular grid with early-exit tiles above the diagonal starves whole XCDs -- workgroups go to XCD (id mod 8),
  // and with 2 x 2 tiles every dead workgroup has id = 1 mod 4, i.e. XCDs 1 and 5 would receive nothing else
  int bi = blockIdx.y, bj = blockIdx.x;
  if (g.lower_compact) {
    bi = (int)((sqrtf(8.0f * blockIdx.x + 1.0f) - 1.0f) * 0.5f);
    while ((bi + 1) * (bi + 2) / 2 <= (int)blockIdx.x) ++bi;
    while (bi * (bi + 1) / 2 > (int)blockIdx.x) --bi;
    bj = blockIdx.x - bi * (bi + 1) / 2;
  }
  const int i0 = bi * GT, j0 = bj * GT;
  const bool split = part != nullptr;
  double* C = split ? part + ((long)sp * g.batch + b) * (long)g.M * g.N : g.C + (long)b * g.c_bs;
  const long c_rs = split ? g.N : g.c_rs;
  if (g.lower_only && j0 > i0 + GT - 1) {   // rectangular grid (direct store): the tile above the diagonal is defined as zero
    if (g.mirror) return;                   // ... or is the mirror image of a tile below it, stored by that tile's workgroup
    for (int e = t; e < GT * GT; e += NT) {
      const int i = i0 + e / GT, j = j0 + e % GT;
      if (i < g.M && j < g.N) C[(long)i * c_rs + j] = 0.0;
    }
    return;
  }
  const int kbeg = sp * kchunk, kend = min(g.K, kbeg + kchunk);
  const int a_m = AKF ? (t / TPR) : ((t % TPK) * EPT), a_k = AKF ? ((t % TPR) * EPT) : (t / TPK);
  const int b_n = BKF ? (t / TPR) : ((t % TPK) * EPT), b_k = BKF ? ((t % TPR) * EPT) : (t / TPK);
  // per-thread fetch pointers, advanced by one k tile per fetch; row / column validity is loop invariant
  const double* pa = g.A + (long)b * g.a_bs + (long)(i0 + a_m) * g.a_rs + (long)(kbeg + a_k) * g.a_cs;
  const double* pb = g.B + (long)b * g.b_bs + (long)(j0 + b_n) * g.b_cs + (long)(kbeg + b_k) * g.b_rs;
  const long a_step = (long)GK * g.a_cs, b_step = (long)GK * g.b_rs;
  const long a_u = AKF ? g.a_cs : g.a_rs, b_u = BKF ? g.b_rs : g.b_cs;   // stride between the thread's elements
  bool va[EPT], vb[EPT];
#pragma unroll
  for (int u = 0; u < EPT; ++u) {
    va[u] = (i0 + a_m + (AKF ? 0 : u)) < g.M;
    vb[u] = (j0 + b_n + (BKF ? 0 : u)) < g.N;
  }
  const bool a_all = va[0] && va[EPT - 1], b_all = vb[0] && vb[EPT - 1];
  double ra[EPT], rb[EPT];
  auto fetch = [&](int k0) {
    const bool full = k0 + GK <= kend;
    if (VEC && full && a_all) {
#pragma unroll
      for (int u = 0; u < EPT; u += 2) {
        const double2 x = *(const double2*)(pa + u);
        ra[u] = x.x; ra[u + 1] = x.y;
      }
    } else {
#pragma unroll
      for (int u = 0; u < EPT; ++u) ra[u] = (va[u] && (full || k0 + a_k + (AKF ? u : 0) < kend)) ? pa[u * a_u] : 0.0;
    }
    if (VEC && full && b_all) {
#pragma unroll
      for (int u = 0; u < EPT; u += 2) {
        const double2 x = *(const double2*)(pb + u);
        rb[u] = x.x; rb[u + 1] = x.y;
      }
    } else {
#pragma unroll
      for (int u = 0; u < EPT; ++u) rb[u] = (vb[u] && (full || k0 + b_k + (BKF ? u : 0) < kend)) ? pb[u * b_u] : 0.0;
    }
    if (g.kscale) {
      const double* ks = g.kscale + (long)b * g.ks_bs;
#pragma unroll
      for (int u = 0; u < EPT; ++u) {
        const int kq = k0 + b_k + (BKF ? u : 0);
        rb[u] *= kq < kend ? ks[(long)kq * g.ks_s] : 0.0;
      }
    }
    pa += a_step;
    pb += b_step;
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
      As[buf][lds_a(a_m + (AKF ? 0 : u), a_k + (AKF ? u : 0))] = ra[u];
      Bs[buf][lds_b(b_n + (BKF ? 0 : u), b_k + (BKF ? u : 0))] = rb[u];
    }
  };
  d4 acc[WF][WF];
#pragma unroll
  for (int i = 0; i < WF; ++i)
#pragma unroll
    for (int j = 0; j < WF; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
  const int wm = (wave / WAVES_N) * WT, wn = (wave % WAVES_N) * WT;
  const int nk = (kend - kbeg + GK - 1) / GK;
  if (nk > 0) {
    fetch(kbeg);
    stage(0);
    __syncthreads();
    if (nk > 1) fetch(kbeg + GK);
  }
  for (int it = 0; it < nk; ++it) {
    const int cur = it & 1;
#pragma unroll
    for (int kk = 0; kk < GK; kk += 4) {
      const int kr = kk + (lane >> 4), c = lane & 15;
      double af[WF], bf[WF];
#pragma unroll
      for (int f = 0; f < WF; ++f) {
        af[f] = As[cur][lds_a(wm + 16 * f + c, kr)];
        bf[f] = Bs[cur][lds_b(wn + 16 * f + c, kr)];
      }
#pragma unroll
      for (int fi = 0; fi < WF; ++fi)
#pragma unroll
        for (int fj = 0; fj < WF; ++fj) acc[fi][fj] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[fi], bf[fj], acc[fi][fj], 0, 0, 0);
    }
    if (it + 1 < nk) stage(cur ^ 1);     // the other buffer was last read before the previous barrier
    __syncthreads();
    if (it + 2 < nk) fetch(kbeg + (it + 2) * GK);
  }
#pragma unroll
  for (int fi = 0; fi < WF; ++fi)
#pragma unroll
    for (int fj = 0; fj < WF; ++fj)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + wm + fi * 16 + (lane >> 4) + 4 * q, j = j0 + wn + fj * 16 + (lane & 15);
        if (i >= g.M || j >= g.N) continue;
        if (g.mirror && j > i) continue;   // written by the thread that holds (j, i)
        double v = acc[fi][fj][q];
        if (!split) {
          if (g.sub_v) v -= g.sub_v[(long)b * g.sv_bs + i] * g.sub_x[(long)b * g.sx_bs + (long)i * g.sx_rs + j];
          v *= g.alpha;
          if (g.colscale) v *= g.colscale[(long)j * g.cs_s + (long)b * g.cs_bs];
          if (g.lower_only && j > i) v = 0.0;
          if (g.phi) v = j < i ? v : (j == i ? 0.5 * v : 0.0);
          if (g.accumulate) v += C[(long)i * c_rs + j];
          if (g.mirror && j < i) C[(long)j * c_rs + i] = v;
        }
        C[(long)i * c_rs + j] = v;
      }
}

__global__ void splitk_reduce_kernel(GenGemm g, int ksplit, const double* part) {
  const long per = (long)g.M * g.N, total = per * g.batch;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int b = (int)(idx / per);
  const long e = idx % per;
  const int i = (int)(e / g.N), j = (int)(e % g.N);
  if (g.mirror && j > i) return;   // written by the thread that holds (j, i)
  double v = 0.0;
  // batches of 8 partials requested together, added in the same fixed order (a rolled loop waited one memory latency per partial:
  // 150-330 us for the 25-50 partials of the W_r contraction)
  int s = 0;
  for (; s + 8 <= ksplit; s += 8) {
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = part[(long)(s + u) * total + idx];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  for (; s < ksplit; ++s) v += part[(long)s * total + idx];
  if (g.sub_v) v -= g.sub_v[(long)b * g.sv_bs + i] * g.sub_x[(long)b * g.sx_bs + (long)i * g.sx_rs + j];
  v *= g.alpha;
  if (g.colscale) v *= g.colscale[(long)j * g.cs_s + (long)b * g.cs_bs];
  if (g.lower_only && j > i) v = 0.0;
  if (g.phi) v = j < i ? v : (j == i ? 0.5 * v : 0.0);
  double* c = g.C + (long)b * g.c_bs + (long)i * g.c_rs + j;
  if (g.accumulate) v += *c;
  *c = v;
  if (g.mirror && j < i) g.C[(long)b * g.c_bs + (long)j * g.c_rs + i] = v;
}


// ---- C_b = alpha A diag(kscale_b) A^T for ONE long, k-contiguous A of at most 256 rows (the reverse pass's W_r = 2 A1 diag(gv_r) A1^T: K = every
// patch column of the batch, R outputs) -------------------------------------------------------------------------------------------------
// Both operands are the same matrix, so a workgroup stages ONE [256][32] chunk of A per step (the general kernel stages two 128-row tiles per 16
// columns and meets at a barrier every 16 MFMAs of a wave) and owns the WHOLE lower triangle of one output for its share of the columns: 36 blocks
// of 32 x 32 on 12 waves, three blocks a wave (the general kernel's three 128 x 128 tiles compute 48 such blocks), 96 MFMAs of a wave between two
// barriers, the column scale applied to the B fragment as it leaves LDS.  grid (k ranges, outputs): the ranges are chosen so that the launch is one
// workgroup per CU; the partial sums meet in splitk_reduce_kernel (fixed order, both triangles stored).
constexpr int SY_ROWS = 256, SY_KC = 32, SY_LD = SY_KC + 1, SY_NT = 768;
// the 36 blocks on and below the diagonal of the 8 x 8 grid, row by row; wave w owns entries 3 w .. 3 w + 2.  A kernel ARGUMENT: as a constant table
// the compiler specialised the whole kernel per wave (twelve copies of the loop, 400 KB of code)
struct SyBlocks { unsigned char b[36][2]; };
static SyBlocks sy_blocks_host() {
  SyBlocks t;
  int n = 0;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j <= i; ++j) { t.b[n][0] = (unsigned char)i; t.b[n][1] = (unsigned char)j; ++n; }
  return t;
}

__global__ __launch_bounds__(SY_NT) void syrk_kscale_kernel(GenGemm g, int kchunk, double* __restrict__ part, SyBlocks tab) {
  extern __shared__ __attribute__((aligned(16))) double sy_smem[];
  double* As = sy_smem;                            // [2][SY_ROWS][SY_LD]
  double* ks = sy_smem + 2 * SY_ROWS * SY_LD;      // [2][SY_KC]
  const int tid = threadIdx.x, lane = tid & 63, lrow = lane >> 4, lcol = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sp = blockIdx.x, b = blockIdx.y;
  const int kbeg = sp * kchunk, kend = min(g.K, kbeg + kchunk);
  const int nk = (kend - kbeg + SY_KC - 1) / SY_KC;
  const double* __restrict__ A = g.A + (long)b * g.a_bs;
  const double* __restrict__ sc = g.kscale + (long)b * g.ks_bs;
  // this thread's 16-byte pieces of a chunk: piece idx = tid + e * SY_NT covers row idx >> 4, columns 2 (idx & 15), 2 (idx & 15) + 1.  Through a buffer
  // descriptor: a 32-bit per-piece offset that never changes, the chunk a scalar offset; rows >= M and whatever lies behind the matrix read as zero
  constexpr int NP = (SY_ROWS * SY_KC / 2 + SY_NT - 1) / SY_NT;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(A), 0, (int)min((long)g.M * g.a_rs * 8, 0x7fffffffL), 0x00020000);
  int poff[NP];
#pragma unroll
  for (int e = 0; e < NP; ++e) {
    const int idx = tid + e * SY_NT, row = idx >> 4;
    poff[e] = (idx < SY_ROWS * SY_KC / 2 && row < g.M) ? (int)(((long)row * g.a_rs + 2 * (idx & 15)) * 8) : (int)0x80000000u;
  }
  // pieces [0, NP / 2) travel in the first half of a step, the rest in the second: half the staging registers (a chunk's 24 of them were what spilled)
  constexpr int NH = NP / 2;
  static_assert(NP % 2 == 0, "two halves");
  double2 ra[NH];
  double rk = 0.0;
  auto fetch = [&](int k0, int half) {
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int e = half * NH + h;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ars, poff[half ? NH + h : h], k0 * 8, 0);
      __builtin_memcpy(&ra[h], &v, 16);
      const int k = k0 + 2 * ((tid + e * SY_NT) & 15);   // columns >= kend belong to the next range (or to nobody): their scale is zero, but 0 x NaN is not
      if (k >= kend) ra[h].x = 0.0;
      if (k + 1 >= kend) ra[h].y = 0.0;
    }
    if (half == 0 && tid < SY_KC) rk = k0 + tid < kend ? sc[(long)(k0 + tid) * g.ks_s] : 0.0;
  };
  auto stage = [&](int buf, int half) {
    double* dst = As + buf * SY_ROWS * SY_LD;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int idx = tid + (half * NH + h) * SY_NT, row = idx >> 4, c = 2 * (idx & 15);
      if (idx < SY_ROWS * SY_KC / 2) { dst[row * SY_LD + c] = ra[h].x; dst[row * SY_LD + c + 1] = ra[h].y; }
    }
    if (half == 0 && tid < SY_KC) ks[buf * SY_KC + tid] = rk;
  };
  int bi[3], bj[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) { bi[t] = tab.b[3 * wave + t][0]; bj[t] = tab.b[3 * wave + t][1]; }
  d4 acc[3][2][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) acc[t][x][y] = d4{0.0, 0.0, 0.0, 0.0};
  if (nk > 0) {
    fetch(kbeg, 0); stage(0, 0);
    fetch(kbeg, 1); stage(0, 1);
    __syncthreads();
  }
  for (int it = 0; it < nk; ++it) {
    const int cur = it & 1;
    const bool more = it + 1 < nk;
    const double* Ac = As + cur * SY_ROWS * SY_LD + lcol * SY_LD + lrow;
    const double* kc = ks + cur * SY_KC + lrow;
    // (one k-step's twelve fragments at a time: left to itself the scheduler hoists every LDS read of the chunk -- 96 of them -- and spills)
    auto ksteps = [&](int k_lo) {
#pragma unroll
      for (int kk = k_lo; kk < k_lo + SY_KC / 2; kk += 4) {
        const double kv = kc[kk];
        double a0[3], a1[3], b0[3], b1[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          a0[t] = Ac[(32 * bi[t]) * SY_LD + kk]; a1[t] = Ac[(32 * bi[t] + 16) * SY_LD + kk];
          b0[t] = Ac[(32 * bj[t]) * SY_LD + kk]; b1[t] = Ac[(32 * bj[t] + 16) * SY_LD + kk];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          b0[t] *= kv; b1[t] *= kv;
          acc[t][0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[t], b0[t], acc[t][0][0], 0, 0, 0);
          acc[t][0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[t], b1[t], acc[t][0][1], 0, 0, 0);
          acc[t][1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[t], b0[t], acc[t][1][0], 0, 0, 0);
          acc[t][1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[t], b1[t], acc[t][1][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // the next chunk goes into the other buffer (last read before the previous barrier) in two halves, each requested before half of this chunk's
    // products and written behind them
    if (more) fetch(kbeg + (it + 1) * SY_KC, 0);
    ksteps(0);
    if (more) { stage(cur ^ 1, 0); fetch(kbeg + (it + 1) * SY_KC, 1); }
    ksteps(SY_KC / 2);
    if (more) stage(cur ^ 1, 1);
    __syncthreads();
  }
  // partial sums in splitk_reduce_kernel's layout: part[(sp * batch + b) * M * N + i * N + j], entries on and below the diagonal only
  double* __restrict__ out = part + ((long)sp * g.batch + b) * (long)g.M * g.N;
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = 32 * bi[t] + 16 * x + lrow + 4 * q, j = 32 * bj[t] + 16 * y + lcol;
          if (i < g.M && j <= i) out[(long)i * g.N + j] = acc[t][x][y][q];
        }
}

}  // namespace

template <int GT, int NT, int WT>
static int gemm_gen_launch(dcgp_ctx* ctx, const GenGemm& g, int slots_per_round) {
  const int nt_m = (g.M + GT - 1) / GT, nt_n = (g.N + GT - 1) / GT;
  long tiles = (long)nt_m * nt_n * g.batch;
  if (g.lower_only) {                 // tiles above the diagonal exit at once
    long live = 0;
    for (int bi = 0; bi < nt_m; ++bi) live += (bi + 1 < nt_n ? bi + 1 : nt_n);
    tiles = live * g.batch;
  }
  // split long contractions so that the launch fills the chip: a whole number of rounds of co-resident workgroups
  // (the 32-tile configuration serves launches that cannot fill the chip: it splits earlier and finer)
  constexpr int SPLIT_MIN_K = GT == 32 ? 1024 : 2048, SPLIT_CHUNK = GT == 32 ? 128 : 512;
  int ksplit = 1;
  if (g.K >= SPLIT_MIN_K && tiles < slots_per_round) {
    ksplit = (int)((slots_per_round + tiles - 1) / tiles);
    if (tiles * ksplit > slots_per_round && ksplit > 1) --ksplit;   // stay within one round rather than spill a few workgroups into a second
    const int max_split = g.K / SPLIT_CHUNK;
    if (ksplit > max_split) ksplit = max_split;
    if (ksplit < 1) ksplit = 1;
  }
  int kchunk = g.K;
  double* part = nullptr;
  if (ksplit > 1) {
    kchunk = round_up((g.K + ksplit - 1) / ksplit, GK);
    ksplit = (g.K + kchunk - 1) / kchunk;
    // one partial buffer per stream: the backward pass runs two chains concurrently
    char pname[48];
    snprintf(pname, sizeof pname, "gemm_gen_part@%p", (void*)ctx->stream);
    part = (double*)ws_get(ctx, pname, (size_t)ksplit * g.batch * g.M * g.N * sizeof(double));
    if (!part) return DCGP_ERR_ALLOC;
  }
  if ((long)g.batch * ksplit > 65535) return ctx_fail(ctx, DCGP_ERR_ARG, "gemm_gen: batch %d x split %d too large", g.batch, ksplit);
  dim3 grid(nt_n, nt_m, g.batch * ksplit);
  GenGemm gk = g;
  if (g.lower_only && (part || g.accumulate)) {   // nothing has to be written above the diagonal: visit the live tiles only
    if (g.M != g.N) return ctx_fail(ctx, DCGP_ERR_ARG, "gemm_gen: lower_only needs a square output");
    gk.lower_compact = 1;
    grid = dim3((unsigned)(tiles / g.batch), 1, g.batch * ksplit);
  }
  const bool akf = g.a_cs == 1, bkf = g.b_rs == 1;
  // 16-byte loads: the contiguous stride is 1 and every other stride, the base and the k origin of a split keep 16-byte alignment
  auto even = [](long x) { return (x & 1) == 0; };
  const bool a_vec = (akf ? even(g.a_rs) : (g.a_rs == 1 && even(g.a_cs))) && even(g.a_bs) && ((uintptr_t)g.A % 16 == 0);
  const bool b_vec = (bkf ? even(g.b_cs) : (g.b_cs == 1 && even(g.b_rs))) && even(g.b_bs) && ((uintptr_t)g.B % 16 == 0);
  const bool vec = a_vec && b_vec;
#define GG_LAUNCH(AK, BKK, V) hipLaunchKernelGGL((gemm_gen_kernel<GT, NT, WT, AK, BKK, V>), grid, dim3(NT), 0, ctx->stream, gk, kchunk, part)
  if (akf && bkf) { if (vec) GG_LAUNCH(true, true, true); else GG_LAUNCH(true, true, false); }
  else if (akf) { if (vec) GG_LAUNCH(true, false, true); else GG_LAUNCH(true, false, false); }
  else if (bkf) { if (vec) GG_LAUNCH(false, true, true); else GG_LAUNCH(false, true, false); }
  else { if (vec) GG_LAUNCH(false, false, true); else GG_LAUNCH(false, false, false); }
#undef GG_LAUNCH
  LAUNCH_CHECK(ctx);
  if (part) {
    const long total = (long)g.M * g.N * g.batch;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, g, ksplit, part);
    LAUNCH_CHECK(ctx);
  }
  return DCGP_OK;
}

// the symmetric long contraction of the reverse pass on its own kernel (syrk_kscale_kernel): A diag(kscale) A^T, k contiguous, <= 256 rows
static bool syrk_applies(const dcgp_ctx* ctx, const GenGemm& g) {
  return !ctx->opt.no_syrk && g.A == g.B && g.a_bs == g.b_bs && g.a_cs == 1 && g.b_rs == 1 && g.a_rs == g.b_cs && g.M == g.N && g.M <= SY_ROWS &&
         g.M > 128 && g.K >= 8192 && g.lower_only && g.mirror && g.kscale && !g.colscale && !g.sub_v && !g.phi && (g.a_rs & 1) == 0 &&
         (g.a_bs & 1) == 0 && (uintptr_t)g.A % 16 == 0 && g.batch <= 64 && (long)g.M * g.a_rs * 8 < 0x7fffffffL;
}
static int syrk_launch(dcgp_ctx* ctx, const GenGemm& g) {
  const int cus = ctx->n_cus > 0 ? ctx->n_cus : 256;
  int ksplit = cus / g.batch;   // one workgroup per CU, no second round
  if (ksplit < 1) ksplit = 1;
  if (ksplit > g.K / 256) ksplit = g.K / 256;
  const int kchunk = round_up((g.K + ksplit - 1) / ksplit, SY_KC);
  ksplit = (g.K + kchunk - 1) / kchunk;
  char pname[48];
  snprintf(pname, sizeof pname, "gemm_gen_part@%p", (void*)ctx->stream);
  double* part = (double*)ws_get(ctx, pname, (size_t)ksplit * g.batch * g.M * g.N * sizeof(double));
  if (!part) return DCGP_ERR_ALLOC;
  const size_t lds = (size_t)(2 * SY_ROWS * SY_LD + 2 * SY_KC) * sizeof(double);
  static bool attr_set[64] = {};   // per device: the attribute is the device's, and a process may hold ctxs on several
  const int dv = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
  if (!attr_set[dv]) {
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)syrk_kscale_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set[dv] = true;
  }
  hipLaunchKernelGGL(syrk_kscale_kernel, dim3(ksplit, g.batch), dim3(SY_NT), lds, ctx->stream, g, kchunk, part, sy_blocks_host());
  LAUNCH_CHECK(ctx);
  const long total = (long)g.M * g.N * g.batch;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, g, ksplit, part);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}

int gemm_gen(dcgp_ctx* ctx, const GenGemm& g) {
  if (g.M <= 0 || g.N <= 0 || g.batch <= 0) return DCGP_OK;
  if (!g.A || !g.B || !g.C || g.K < 0) return ctx_fail(ctx, DCGP_ERR_ARG, "gemm_gen: bad arguments");
  if (syrk_applies(ctx, g)) return syrk_launch(ctx, g);
  // 256 CUs; 4 co-resident 64-tile workgroups per CU (40 KB LDS each), 2 of the 128-tile ones (72 KB, 1024 threads)
  // the 1024-thread 128-tile (twice the flop per operand byte) pays off where it can fill its 512 slots, split included:
  // the tiled batch's contractions (K = 46080 columns at the headline size), M = 1024.  With a few thousand columns and
  // M = 256 (de-duplicated first layer) its ~270 workgroups lose to ~900 64-tiles (training step 2.05 -> 1.93 ms)
#ifdef DCGP_EXPERIMENTS
  static const long big_fill = getenv("DCGP_GEMM_BIG_FILL") ? atol(getenv("DCGP_GEMM_BIG_FILL")) : 512;
  static const long small_wgs = getenv("DCGP_GEMM_SMALL_WGS") ? atol(getenv("DCGP_GEMM_SMALL_WGS")) : 256;
#else
  constexpr long big_fill = 512, small_wgs = 256;
#endif
  if (ctx->opt.gemm_tile == 32) return gemm_gen_launch<32, 256, 16>(ctx, g, 1024);     // A/B switch
  if (ctx->opt.gemm_tile == 64) return gemm_gen_launch<64, 256, 32>(ctx, g, 1024);
  if (ctx->opt.gemm_tile == 128) return gemm_gen_launch<128, 1024, 32>(ctx, g, 512);
  const long tiles128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * g.batch / (g.lower_only ? 2 : 1);
  if (g.M >= 128 && g.N >= 128 && g.K >= 4096 && tiles128 * (g.K / 512) >= big_fill) return gemm_gen_launch<128, 1024, 32>(ctx, g, 512);
  // the M x M x M products of the Cholesky / KL adjoint chains would launch a few dozen 64-tile workgroups on 256 CUs and
  // take as long as one wave needs for its 32 x 32 x K block (K x 64 cycles of fp64 MFMA); 32-tiles with a 16 x 16 block per
  // wave put four times as many CUs to work on a quarter of that each.  Likewise the long contractions with a narrow
  // output (d alpha = A1 gm: M x R): a handful of tiles, split into 128-deep chunks instead of 512-deep ones
  const long wgs = (long)((g.M + 63) / 64) * ((g.N + 63) / 64) * g.batch / (g.lower_only ? 2 : 1);
  const long max_wgs = wgs * (g.K >= 2048 ? g.K / 512 : 1);   // what the 64-tile configuration could launch, split included
  if (max_wgs <= small_wgs) return gemm_gen_launch<32, 256, 16>(ctx, g, 1024);
  return gemm_gen_launch<64, 256, 32>(ctx, g, 1024);
}

// C-ABI view of the same kernel (tests drive every layout / edge case through it)
extern "C" int dcgp_gemm_strided(dcgp_ctx* ctx, const double* A, long a_rs, long a_cs, long a_bs, const double* B, long b_rs, long b_cs,
                                 long b_bs, double* C, long c_rs, long c_bs, int M, int N, int K, int batch, double alpha, int accumulate,
                                 const double* colscale, long cs_s, long cs_bs, const double* kscale, long ks_s, long ks_bs, int lower_only) {
  if (!ctx) return DCGP_ERR_ARG;
  if (!A || !B || !C || M < 0 || N < 0 || K < 0 || batch < 0 || c_rs < N) return ctx_fail(ctx, DCGP_ERR_ARG, "gemm_strided: bad arguments");
  GenGemm g;
  g.A = A; g.a_rs = a_rs; g.a_cs = a_cs; g.a_bs = a_bs;
  g.B = B; g.b_rs = b_rs; g.b_cs = b_cs; g.b_bs = b_bs;
  g.C = C; g.c_rs = c_rs; g.c_bs = c_bs;
  g.M = M; g.N = N; g.K = K; g.batch = batch; g.alpha = alpha; g.accumulate = accumulate;
  g.colscale = colscale; g.cs_s = cs_s; g.cs_bs = cs_bs; g.kscale = kscale; g.ks_s = ks_s; g.ks_bs = ks_bs; g.lower_only = lower_only;
  DCGP_TRY(gemm_gen(ctx, g));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}

// the same with the epilogue hooks of the kernel adjoints: C_b(i, j) (+)= alpha (sum_k ... - sub_v_b[i] sub_x_b(i, j)), flags bit 0: lower_only,
// bit 1: mirror (the strictly lower entries also stored transposed), bit 2: phi (strictly lower part kept, diagonal halved, rest zero)
extern "C" int dcgp_gemm_strided_ex(dcgp_ctx* ctx, const double* A, long a_rs, long a_cs, long a_bs, const double* B, long b_rs, long b_cs,
                                    long b_bs, double* C, long c_rs, long c_bs, int M, int N, int K, int batch, double alpha, int accumulate,
                                    const double* sub_v, long sv_bs, const double* sub_x, long sx_rs, long sx_bs, int flags) {
  if (!ctx) return DCGP_ERR_ARG;
  if (!A || !B || !C || M < 0 || N < 0 || K < 0 || batch < 0 || c_rs < N || ((sub_v != nullptr) != (sub_x != nullptr)))
    return ctx_fail(ctx, DCGP_ERR_ARG, "gemm_strided_ex: bad arguments");
  if ((flags & 3) && M != N) return ctx_fail(ctx, DCGP_ERR_ARG, "gemm_strided_ex: lower_only / mirror need a square result");
  if ((flags & 2) && !(flags & 1)) return ctx_fail(ctx, DCGP_ERR_ARG, "gemm_strided_ex: mirror goes with lower_only");
  GenGemm g;
  g.A = A; g.a_rs = a_rs; g.a_cs = a_cs; g.a_bs = a_bs;
  g.B = B; g.b_rs = b_rs; g.b_cs = b_cs; g.b_bs = b_bs;
  g.C = C; g.c_rs = c_rs; g.c_bs = c_bs;
  g.M = M; g.N = N; g.K = K; g.batch = batch; g.alpha = alpha; g.accumulate = accumulate;
  g.sub_v = sub_v; g.sv_bs = sv_bs; g.sub_x = sub_x; g.sx_rs = sx_rs; g.sx_bs = sx_bs;
  g.lower_only = flags & 1; g.mirror = (flags >> 1) & 1; g.phi = (flags >> 2) & 1;
  DCGP_TRY(gemm_gen(ctx, g));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DCGP_OK;
}
