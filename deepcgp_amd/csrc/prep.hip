// prep.hip -- everything of a forward step that depends only on the parameters, for ALL layers, in ONE launch:
//   task 0/1  Kuu = RBF.K(Z) + jitter I for the live Z and for the frozen prior Z0   (conv_gp/layers.py:18-21,149-152)
//   task 2    Z^T (k-major, zero padded) and |z|^2                                    (operands of the patch sweep)
//   task 3    lower-masked, zero-padded q_sqrt  (matrix_band_part, conv_gp/conditionals.py:55)
//   task 4    zero-padded q_mu
//   task 5    the sweeps' scaled Z operand (sweep_dev.h)
// These are ~5 tiny launches per layer when issued one by one; at ~5 us of launch latency each they cost more
// than the work itself and sit on the critical path in front of the factorisation chain.
// Round 3: the launch is a flat list of small items (tools/prep_trace.py shows every block): at the front of a step every operand is
// cold (HBM, ~3 us per round of loads), so an item is ONE round of branch-free loads and the Gram tiles run on the matrix pipe
// (22 -> 13 us at the headline configuration).
// PREP_TRACE_SWITCH
#include <algorithm>
#include "layer.h"
#include "sweep_dev.h"

namespace {

// One 16 x 16 tile of Kuu on the matrix pipe: the four waves of the block split the patch length (k-steps w, w + 4, ...), each lane loads
// its MFMA operand elements straight from Z (8 k-steps = 16 loads in flight), |z|^2 is accumulated from the same registers, and the four
// partial tiles meet in LDS.  (One output per thread from LDS tiles read 2 LDS words per 3 FMAs: at the head's L = 250 a block spent
// ~7 us on LDS bandwidth alone, and with per-chunk waits for Z on top was the long pole of the launch at 15 us.)
template <bool SC>
__device__ void gram_tile(const PrepLayerArgs& p, const double* __restrict__ Z, double* __restrict__ out, int t, double* lds) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lcol = lane & 15, lrow = lane >> 4;
  const int nt = (p.Mp + 15) / 16;
  const int i0 = (t / nt) * 16, j0 = (t % nt) * 16;
  const int nsteps = (p.L + 3) / 4;
  d4 acc = d4{0.0, 0.0, 0.0, 0.0};
  double na = 0.0, nb = 0.0;
  for (int s0 = w; s0 < nsteps; s0 += 32) {
    double av[8], bv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int l = 4 * (s0 + 4 * k) + lrow;   // beyond the last k-step l >= L: zeros
      av[k] = ld_z<SC>(Z, p.in_scale, i0 + lcol, l, p.M, p.L);
      bv[k] = ld_z<SC>(Z, p.in_scale, j0 + lcol, l, p.M, p.L);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[k], bv[k], acc, 0, 0, 0);
      na = fma(av[k], av[k], na);
      nb = fma(bv[k], bv[k], nb);
    }
  }
  na += __shfl_xor(na, 16); na += __shfl_xor(na, 32);
  nb += __shfl_xor(nb, 16); nb += __shfl_xor(nb, 32);
  double* red = lds + w * 288;
#pragma unroll
  for (int q = 0; q < 4; ++q) red[(lrow + 4 * q) * 16 + lcol] = acc[q];
  if (lane < 16) { red[256 + lane] = na; red[272 + lane] = nb; }
  __syncthreads();
  const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
  double dot = 0.0, ni = 0.0, nj = 0.0;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    dot += lds[v * 288 + ti * 16 + tj];
    ni += lds[v * 288 + 256 + ti];
    nj += lds[v * 288 + 272 + tj];
  }
  const int i = i0 + ti, j = j0 + tj;
  if (i < p.Mp && j < p.Mp) {
    double v = 0.0;
    if (i < p.M && j < p.M) {
      v = p.bk.eval(dot, ni, nj);
      if (i == j) v += p.jitter;
    } else if (i == j) {
      v = 1.0;   // identity on the padding keeps the padded matrix factorisable
    }
    out[(long)i * p.Mp + j] = v;
  }
}

// Z^T (k-major, zero padded) and |z|^2.  One 32 x 32 tile of the transpose per item (a block that walked all of a row block's chunks
// waited one memory latency per chunk), then items of 8 rows of norms: 32 threads per row, 8 loads in flight per thread.
template <bool SC>
__device__ void transpose_task(const PrepLayerArgs& p, int bx, int nbx, double (*t)[33]) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int nmb = (p.Mp + 31) / 32, nlb = (p.Lp + 31) / 32, ntiles = nmb * nlb, nnb = (p.Mp + 7) / 8;
  for (int w = bx; w < ntiles + nnb; w += nbx) {
    if (w < ntiles) {
      const int m0 = (w / nlb) * 32, l0 = (w % nlb) * 32;
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = ty + 8 * e, m = m0 + r, l = l0 + tx;
        t[r][tx] = ld_z<SC>(p.Z, p.in_scale, m, l, p.M, p.L);
      }
      __syncthreads();
      for (int r = ty; r < 32; r += 8) {
        const int l = l0 + r, m = m0 + tx;
        if (l < p.Lp && m < p.Mp) p.ZT[(long)l * p.Mp + m] = t[tx][r];
      }
    } else {
      const int m = (w - ntiles) * 8 + ty;
      const double acc = row_sq_norm<SC>(p.Z, p.in_scale, 1.0, m < p.M ? m : -1, p.L, tx);
      if (tx == 0 && m < p.Mp) p.zn[m] = acc;
    }
  }
}

#ifdef DCGP_PREP_TRACE
// timing aid (tools/ab_build.sh ... 's/^\/\/ PREP_TRACE_SWITCH/#define DCGP_PREP_TRACE 1/'): start / end of every block, 100 MHz
__device__ unsigned long long g_prep_trace[4096][6];   // per block: start, end, layer | task << 8, shader clocks, segment known, layer arguments read
#endif

// Work list of the launch: segment g = items [first[g], first[g + 1]) of one (layer, task), longest items first.  A grid of
// 256 x 6 x layers blocks, most of them with nothing to do, took longer to START than its work takes to run.
struct PrepPlan {
  int nseg = 0;
  int first[48];    // first block of the segment
  int what[48];     // layer | task << 8
};

// launch bounds: all ~1400 blocks of a two-layer model resident at once (5 per CU) -- a block is a chain of a few memory latencies,
// and a second round of blocks starts only when the first has finished its chain
__global__ __launch_bounds__(256, 5) void prepare_all_kernel(PrepArgs a, PrepPlan plan) {
  __shared__ double lds[36][33];   // every task's staging tile
#ifdef DCGP_PREP_TRACE
  const unsigned long long stamp0 = wall_clock64(), cyc0 = clock64();
#endif
  // the block's segment by selects over the whole table: one round of scalar loads, no load that waits for an index
  int first = 0, what = plan.what[0];
#pragma unroll
  for (int k = 1; k < 48; ++k) {
    const bool in = k < plan.nseg && (int)blockIdx.x >= plan.first[k];
    first = in ? plan.first[k] : first;
    what = in ? plan.what[k] : what;
  }
  const int item = blockIdx.x - first, task = what >> 8, layer = what & 255;
  const PrepLayerArgs& p = a.l[layer];
#ifdef DCGP_PREP_TRACE
  asm volatile("s_nop 0" ::"s"(what));
  const unsigned long long stamp1 = wall_clock64();
  asm volatile("s_nop 0" ::"s"(p.M), "s"(p.L), "s"(p.Z));
  const unsigned long long stamp2 = wall_clock64();
#endif
  switch (task) {
    case 0:
    case 1: {
      const double* Z = task ? p.Z0 : p.Z;
      double* out = task ? p.Kp : p.K;
      if (p.in_scale) gram_tile<true>(p, Z, out, item, &lds[0][0]);
      else gram_tile<false>(p, Z, out, item, &lds[0][0]);
    } break;
    case 2:
      if (p.in_scale) transpose_task<true>(p, item, 1 << 30, lds);
      else transpose_task<false>(p, item, 1 << 30, lds);
      break;
    case 3: {
      // item = (r, 16 rows): the lower triangle's part of those rows, all loads of a thread in flight together
      const int nrb = (p.Mp + 15) / 16, r = item / nrb, i0 = (item % nrb) * 16;
      const double* __restrict__ src = p.q_sqrt + (long)r * p.M * p.M;
      double* __restrict__ dst = p.Lq + (long)r * p.Mp * p.Mp;
      for (int j = threadIdx.x; j < p.Mp; j += 256) {
        double t16[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) t16[e] = ld_guard(src, (long)(i0 + e) * p.M + j, i0 + e < p.M && j <= i0 + e);
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (i0 + e < p.Mp) dst[(long)(i0 + e) * p.Mp + j] = t16[e];
      }
    } break;
    case 4: {
      const int idx = item * 256 + threadIdx.x;
      if (idx < p.Mp * p.Rp) {
        const int r = idx % p.Rp, i = idx / p.Rp;
        p.qmu[idx] = ld_guard(p.q_mu, (long)i * p.R + r, i < p.M && r < p.R);
      }
    } break;
    default: {
      ZsTask z;
      z.Z = p.Z; z.in_scale = p.in_scale; z.ZS = p.ZS; z.M = p.M; z.Mp = p.Mp; z.L = p.L; z.Lq = p.Lz;
      z.csq = sqrt(1.4426950408889634074 * p.bk.p1); z.log2var = log2(p.bk.variance);
      zs_task(z, item, 1 << 30, lds);
    }
  }
#ifdef DCGP_PREP_TRACE
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x < 4096) {
    g_prep_trace[blockIdx.x][0] = stamp0;
    g_prep_trace[blockIdx.x][1] = wall_clock64();
    g_prep_trace[blockIdx.x][2] = what;
    g_prep_trace[blockIdx.x][3] = clock64() - cyc0;
    g_prep_trace[blockIdx.x][4] = stamp1;
    g_prep_trace[blockIdx.x][5] = stamp2;
  }
#endif
}

}  // namespace

#ifdef DCGP_PREP_TRACE
extern "C" void dcgp_debug_prep_trace(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prep_trace), sizeof(unsigned long long) * 4096 * 6);
  if (reset) {
    static unsigned long long zero[4096][6];
    hipMemcpyToSymbol(HIP_SYMBOL(g_prep_trace), zero, sizeof(zero));
  }
}
#endif

int prepare_all(dcgp_ctx* ctx, const PrepArgs& a, unsigned task_mask) {
  if (a.nl <= 0) return DCGP_OK;
  ScopedTimer t(ctx, "prepare");
  // (layer, task) segments with their item counts and a per-item cost (chunks of the patch length for the Gram tiles, 1 otherwise)
  struct Seg { int layer, task, n, cost; };
  Seg segs[48];
  int ns = 0;
  for (int i = 0; i < a.nl; ++i) {
    const PrepLayerArgs& p = a.l[i];
    const int nt = (p.Mp + 15) / 16, chunks = (p.L + 31) / 32, nmb = (p.Mp + 31) / 32, nnb = (p.Mp + 7) / 8;
    auto on = [&](int task) { return (task_mask >> task & 1u) != 0; };
    if (on(0)) segs[ns++] = {i, 0, nt * nt, 2 + chunks};
    if (p.Kp && on(1)) segs[ns++] = {i, 1, nt * nt, 2 + chunks};
    if (on(2)) segs[ns++] = {i, 2, nmb * ((p.Lp + 31) / 32) + nnb, 1};
    if (p.q_sqrt && on(3)) segs[ns++] = {i, 3, p.R * ((p.Mp + 15) / 16), 2};
    if (on(4)) segs[ns++] = {i, 4, (p.Mp * p.Rp + 255) / 256, 0};
    if (p.ZS && on(5)) segs[ns++] = {i, 5, nmb * ((p.L + 31) / 32) + nnb, 1};
  }
  if (ns == 0) return DCGP_OK;
  std::stable_sort(segs, segs + ns, [](const Seg& x, const Seg& y) { return x.cost > y.cost; });
  PrepPlan plan;
  plan.nseg = ns;
  int total = 0;
  for (int g = 0; g < ns; ++g) {
    plan.first[g] = total;
    plan.what[g] = segs[g].layer | (segs[g].task << 8);
    total += segs[g].n;
  }
  hipLaunchKernelGGL(prepare_all_kernel, dim3(total), dim3(256), 0, ctx->stream, a, plan);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}
