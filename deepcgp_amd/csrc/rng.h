// Internal: the counter-based device RNG of the sampling steps (Layer.sample_from_conditional draws z inside the graph).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

// ---- counter-based RNG (Philox4x32-10) + Box-Muller, one normal per element -------------------
static __device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
  uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
  uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
static __device__ __forceinline__ double philox_normal(uint64_t seed, uint32_t stream, uint64_t idx) {
  uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), stream, 0x5eed5eedu};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  // 53-bit uniform in (0,1] and a 32-bit angle
  uint64_t bits = ((uint64_t)c[0] << 21) ^ (uint64_t)(c[1] >> 11);
  double u1 = ((double)(bits & ((1ull << 53) - 1)) + 1.0) * (1.0 / 9007199254740992.0);
  double u2 = ((double)c[2] + 0.5) * (1.0 / 4294967296.0);
  return sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
}


// Where a rank's output element sits in the GLOBAL batch: a rank holds images [lo, lo + Nl) of Ng, its rows are (sample s, local
// image n) -> s * Nl + n, W outputs each.  The noise of element o is drawn at the counter of the same (sample, image, output) of the
// un-sharded batch, so a step's value does not depend on the number of ranks (W == 0: not sharded, identity).
struct RngMap {
  int W = 0, Nl = 0, Ng = 0, lo = 0;
};
static __device__ __forceinline__ uint64_t rng_index(const RngMap& m, long o) {
  if (m.W == 0) return (uint64_t)o;
  const long row = o / m.W, w = o - row * m.W;
  const long s = row / m.Nl, n = row - s * m.Nl;
  return (uint64_t)(((s * m.Ng + m.lo + n) * m.W) + w);
}
