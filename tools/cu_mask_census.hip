// Which CUs does a stream created with hipExtStreamCreateWithCUMask run on?  Census of (XCC, SE, CU) per workgroup for a
// mask that enables bits [lo, hi): hipcc --offload-arch=gfx950 -O3 tools/cu_mask_census.hip -o /tmp/cu_mask_census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void census(unsigned* out, int spin) {
  if (threadIdx.x == 0) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hwid;
  }
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
}
int main(int argc, char** argv) {
  int lo = argc > 1 ? atoi(argv[1]) : 0, hi = argc > 2 ? atoi(argv[2]) : 240;
  uint32_t mask[8] = {};
  for (int i = lo; i < hi; ++i) mask[i / 32] |= 1u << (i % 32);
  hipStream_t s;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
  printf("mask bits [%d, %d): create -> %s\n", lo, hi, hipGetErrorString(e));
  if (e != hipSuccess) return 1;
  const int nb = 2048;
  unsigned* d; hipMalloc(&d, nb * 8);
  hipLaunchKernelGGL(census, dim3(nb), dim3(1024), 64 * 1024, s, d, 2000);   // 1024 threads + 64 KB LDS: at most 2 per CU
  hipStreamSynchronize(s);
  std::vector<unsigned> h(2 * nb);
  hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
  int cnt[8][8][16] = {};
  for (int b = 0; b < nb; ++b) {
    unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
    unsigned cu = (hw >> 8) & 0xf, se = (hw >> 13) & 0x7;   // gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    cnt[xcc & 7][se][cu]++;
  }
  int per_xcc[8] = {}, total = 0;
  for (int x = 0; x < 8; ++x) {
    printf("XCC %d:", x);
    for (int se = 0; se < 8; ++se) {
      int n = 0;
      for (int c = 0; c < 16; ++c) n += cnt[x][se][c] > 0;
      if (n) printf(" SE%d:%d", se, n);
      per_xcc[x] += n;
    }
    printf("  -> %d CUs\n", per_xcc[x]);
    total += per_xcc[x];
  }
  printf("distinct CUs used: %d\n", total);
  return 0;
}
