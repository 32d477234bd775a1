// tail_dev.h -- the end of an ELBO step as device code (the tail kernel of cond.hip):
// RobustMax variational expectations of every row (conv_gp/../likelihoods: RobustMax via 20-node Gauss-Hermite), their sum
// in a fixed order, and the assembly  data * scale - sum_l KL_l  with the status words of the factorisations.
#pragma once
#include "layer.h"

struct TailArgs {
  const double* mu = nullptr; const double* var = nullptr;   // [n_rows][K]
  const int32_t* y = nullptr; int n_rows = 0, n_labels = 0, K = 0;
  double eps = 0.0;
  const double* gh = nullptr;       // [40]: 20 Gauss-Hermite nodes, 20 weights
  double* ve = nullptr;             // [n_rows] expectations (kept: the reverse pass reads them)
  double inv_s = 1.0;
  unsigned* ticket = nullptr;       // arrival counter (zero between launches)
  double* scal = nullptr;           // nullptr: no tail
  ElboFinish fin;
};

// lane g (< 32) of the 32 that share a row whose K means / variances are at m / v (global memory or LDS) and whose label
// is yi: this Gauss-Hermite node's term of the probability that class yi wins; the sum over the lanes is p
__device__ __forceinline__ double robustmax_node(const double* m, const double* v, int yi, int K, const double* gh, int g) {
  if (g >= 20) return 0.0;
  const double x = m[yi] + gh[g] * sqrt(fmax(2.0 * v[yi], 1e-10));
  double prod = 1.0;
  for (int k = 0; k < K; ++k) {
    if (k == yi) continue;
    const double dist = (x - m[k]) / sqrt(fmax(v[k], 1e-10));
    const double cdf = 0.5 * (1.0 + erf(dist * 0.70710678118654752440));
    prod *= cdf * (1.0 - 2e-4) + 1e-4;
  }
  return prod * gh[20 + g] * 0.56418958354775628695;   // w / sqrt(pi)
}
__device__ __forceinline__ double robustmax_logp(double p, double eps, int K) {
  return p * log(1.0 - eps) + (1.0 - p) * log(eps / (K - 1.0));
}

// One workgroup (>= 256 threads, the first 256 work: the sums and their order do not depend on the launch that carries the block): the four KL
// pieces of a layer -- {Mahalanobis, log det q, log det prior, trace}, the layout kl_small_kernel (cond.hip) leaves -- from the strip sums of
// prep_solve and the diagonals of the factors.
__device__ __forceinline__ void kl_pieces_block(const KlTailLayer& L, double* __restrict__ kl4, double* red /* [4][256] LDS */) {
  const int tid = threadIdx.x, ns = L.ns > 0 ? L.ns : L.Mp / 16, nsa = L.ns > 0 ? L.nsa : 1;
  if (tid < 256) {
    double ldq = 0.0, ldp = 0.0, tr = 0.0, mh = 0.0;
    for (int idx = tid; idx < L.M * L.R; idx += 256) {
      const int i = idx % L.M, r = idx / L.M;
      const double d = L.Lq[((long)r * L.Mp + i) * L.Mp + i];
      ldq += log(d * d);
    }
    for (int i = tid; i < L.M; i += 256) {
      const double d = L.Lfac[(long)i * L.ldf + i];
      ldp += log(d * d);
    }
    for (int i = tid; i < L.R * ns; i += 256) tr += L.sums[i];
    for (int i = tid; i < nsa; i += 256) mh += L.sums[(long)L.R * ns + i];
    red[tid] = mh; red[256 + tid] = ldq; red[512 + tid] = ldp; red[768 + tid] = tr;
  }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o)
      for (int q = 0; q < 4; ++q) red[q * 256 + tid] += red[q * 256 + tid + o];
    __syncthreads();
  }
  if (tid < 4) kl4[tid] = red[tid * 256];
}

// one thread: scal[0] = data term; with fin.nl > 0 the four result words (device and pinned host slot)
__device__ __forceinline__ void elbo_assemble(const TailArgs& t, double data) {
  double* scal = t.scal;
  const ElboFinish& fin = t.fin;
  scal[0] = data;
  if (fin.nl <= 0) return;
  double kl = 0.0;
  for (int l = 0; l < fin.nl; ++l) {
    const double* k4 = scal + 4 + 4 * l;
    double two = k4[0] - (double)fin.M[l] * fin.R[l] - k4[1] + k4[3];
    if (!fin.white[l]) two += (double)fin.R[l] * k4[2];
    kl += 0.5 * two;
  }
  int bad = 0;   // first non-positive pivot of any factorisation: rides back with the result (no further copy, one sync)
  for (int q = 0; q < fin.ngroups; ++q)
    for (int i = 0; i < fin.ninfo[q]; ++i)
      if (fin.info[q][i] && !bad) bad = fin.info[q][i];
  const double res[4] = {data * fin.scale - kl, data, kl, (double)bad};
  for (int i = 0; i < 4; ++i) scal[40 + i] = res[i];
  if (fin.host_out) {   // straight into the caller's pinned slot: a 32-byte copy command cost 4 us and a gap behind this kernel
    for (int i = 0; i < 4; ++i) __hip_atomic_store(fin.host_out + i, res[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fin.host_out + 4, fin.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // the host polls this word
  }
}

// Arrival at the end of a workgroup's part of a launch: true in every thread of the LAST workgroup to arrive, whose loads then
// see what the others wrote (agent scope: the workgroups sit on different XCDs, whose L2s are not coherent).  `flag` in LDS.
__device__ __forceinline__ bool last_to_arrive(unsigned* counter, unsigned n_groups, unsigned* flag) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_groups - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      *counter = 0u;   // the next launch on this stream starts from zero
    }
    *flag = last ? 1u : 0u;
  }
  __syncthreads();
  return *flag != 0u;
}
