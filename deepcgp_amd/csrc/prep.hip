// prep.hip -- everything of a forward step that depends only on the parameters, for ALL layers, in ONE launch:
//   task 0/1  Kuu = RBF.K(Z) + jitter I for the live Z and for the frozen prior Z0   (conv_gp/layers.py:18-21,149-152)
//   task 2    Z^T (k-major, zero padded) and |z|^2                                    (operands of the patch sweep)
//   task 3    lower-masked, zero-padded q_sqrt  (matrix_band_part, conv_gp/conditionals.py:55)
//   task 4    zero-padded q_mu
//   task 5    the sweeps' scaled Z operand (sweep_dev.h)
// These are ~5 tiny launches per layer when issued one by one; at ~5 us of launch latency each they cost more
// than the work itself and sit on the critical path in front of the factorisation chain.
#include "layer.h"
#include "sweep_dev.h"

namespace {

__device__ void gram_task(const PrepLayerArgs& p, const double* __restrict__ Z, double* __restrict__ out, int bx, int nbx) {
  __shared__ double Zi[16][33], Zj[16][33];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int nt = (p.Mp + 15) / 16;
  for (int t = bx; t < nt * nt; t += nbx) {
    const int i0 = (t / nt) * 16, j0 = (t % nt) * 16;
    double dot = 0.0, ni = 0.0, nj = 0.0;
    // chunks of 32 along the patch length, the next chunk's loads in flight while the current one is multiplied
    // (the head's L = 250 means 8 chunks: rolled, each waited a full memory latency at the front of the step)
    double ri[2], rj[2];
    auto fetch = [&](int l0) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int idx = threadIdx.x + e * 256, r = idx >> 5, c = idx & 31;
        const double sc = (p.in_scale && l0 + c < p.L) ? p.in_scale[l0 + c] : 1.0;
        ri[e] = (i0 + r < p.M && l0 + c < p.L) ? Z[(long)(i0 + r) * p.L + l0 + c] * sc : 0.0;
        rj[e] = (j0 + r < p.M && l0 + c < p.L) ? Z[(long)(j0 + r) * p.L + l0 + c] * sc : 0.0;
      }
    };
    fetch(0);
    for (int l0 = 0; l0 < p.L; l0 += 32) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int idx = threadIdx.x + e * 256, r = idx >> 5, c = idx & 31;
        Zi[r][c] = ri[e];
        Zj[r][c] = rj[e];
      }
      __syncthreads();
      if (l0 + 32 < p.L) fetch(l0 + 32);
#pragma unroll 8
      for (int l = 0; l < 32; ++l) {
        const double a = Zi[ty][l], b = Zj[tx][l];
        dot = fma(a, b, dot);
        ni = fma(a, a, ni);
        nj = fma(b, b, nj);
      }
    }
    const int i = i0 + ty, j = j0 + tx;
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
}

__device__ void transpose_task(const PrepLayerArgs& p, int bx, int nbx) {
  __shared__ double t[32][33];
  __shared__ double nrm[8][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int mb = bx; mb * 32 < p.Mp; mb += nbx) {
    const int m0 = mb * 32;
    double acc = 0.0;
    double rz[4];   // this thread's 4 elements of the next 32 x 32 chunk (prefetched while the current one is written)
    auto fetch = [&](int l0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = ty + 8 * e, m = m0 + r, l = l0 + tx;
        rz[e] = (m < p.M && l < p.L) ? p.Z[(long)m * p.L + l] * (p.in_scale ? p.in_scale[l] : 1.0) : 0.0;
      }
    };
    fetch(0);
    for (int l0 = 0; l0 < p.Lp; l0 += 32) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 4; ++e) t[ty + 8 * e][tx] = rz[e];
      __syncthreads();
      if (l0 + 32 < p.Lp) fetch(l0 + 32);
      for (int r = ty; r < 32; r += 8) {
        const int l = l0 + r, m = m0 + tx;
        const double v = t[tx][r];
        if (l < p.Lp && m < p.Mp) p.ZT[(long)l * p.Mp + m] = v;
        acc = fma(v, v, acc);
      }
    }
    nrm[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && m0 + tx < p.Mp) {
      double s2 = 0.0;
      for (int q = 0; q < 8; ++q) s2 += nrm[q][tx];
      p.zn[m0 + tx] = s2;
    }
  }
}

__global__ __launch_bounds__(256) void prepare_all_kernel(PrepArgs a) {
  const PrepLayerArgs& p = a.l[blockIdx.z];
  const int bx = blockIdx.x, nbx = gridDim.x;
  switch (blockIdx.y) {
    case 0: gram_task(p, p.Z, p.K, bx, nbx); break;
    case 1: if (p.Kp) gram_task(p, p.Z0, p.Kp, bx, nbx); break;
    case 2: transpose_task(p, bx, nbx); break;
    case 3:
      if (p.q_sqrt) {
        // batches of 8 loads per thread, then 8 stores: a rolled copy loop waits one memory latency per element
        const long total = (long)p.R * p.Mp * p.Mp, stride = (long)nbx * 256;
        for (long base = (long)bx * 256 + threadIdx.x; base < total; base += 8 * stride) {
          double t8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const long idx = base + e * stride;
            t8[e] = 0.0;
            if (idx < total) {
              const int j = (int)(idx % p.Mp);
              const long t = idx / p.Mp;
              const int i = (int)(t % p.Mp), r = (int)(t / p.Mp);
              if (i < p.M && j <= i) t8[e] = p.q_sqrt[((long)r * p.M + i) * p.M + j];
            }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const long idx = base + e * stride;
            if (idx < total) p.Lq[idx] = t8[e];
          }
        }
      }
      break;
    case 5:
      if (p.ZS) {
        ZsTask z;
        z.Z = p.Z; z.in_scale = p.in_scale; z.ZS = p.ZS; z.M = p.M; z.Mp = p.Mp; z.L = p.L; z.Lq = p.Lz;
        z.csq = sqrt(1.4426950408889634074 * p.bk.p1); z.log2var = log2(p.bk.variance);
        zs_task(z, bx, nbx);
      }
      break;
    default: {
      const long total = (long)p.Mp * p.Rp;
      for (long idx = (long)bx * 256 + threadIdx.x; idx < total; idx += (long)nbx * 256) {
        const int r = (int)(idx % p.Rp), i = (int)(idx / p.Rp);
        p.qmu[idx] = (i < p.M && r < p.R) ? p.q_mu[(long)i * p.R + r] : 0.0;
      }
    }
  }
}

}  // namespace

int prepare_all(dcgp_ctx* ctx, const PrepArgs& a) {
  if (a.nl <= 0) return DCGP_OK;
  ScopedTimer t(ctx, "prepare");
  hipLaunchKernelGGL(prepare_all_kernel, dim3(256, 6, a.nl), dim3(256), 0, ctx->stream, a);
  LAUNCH_CHECK(ctx);
  return DCGP_OK;
}
