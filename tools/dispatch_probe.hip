// Where and when do the blocks of a VALU-bound grid run?  Records per block the
// XCC id, hardware id (SE/CU/SIMD), and start/end wall ticks (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>
#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2);} } while (0)
struct Rec { unsigned long long r0, r1; unsigned xcc, hwid; };
__global__ __launch_bounds__(256) void spin(uint32_t *out, Rec *rec, int iters, uint32_t seed) {
  uint32_t a = threadIdx.x * 747796405u + seed, b = a ^ 0x5bd1e995u, c = a * 3u, d = a * 7u;
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    a = __builtin_popcount(a ^ seed) + a; b = __builtin_popcount(b ^ seed) + b;
    c = __builtin_popcount(c ^ seed) + c; d = __builtin_popcount(d ^ seed) + d;
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  }
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    rec[blockIdx.x] = Rec{r0, r1, xcc, hw};
  }
}
int main(int argc, char **argv) {
  int per_cu = argc > 1 ? atoi(argv[1]) : 1, iters = 100000;
  hipDeviceProp_t prop; HIPCHECK(hipGetDeviceProperties(&prop, 0));
  int blocks = prop.multiProcessorCount * per_cu;
  uint32_t *out; Rec *rec; HIPCHECK(hipMalloc(&out, (size_t)blocks * 1024)); HIPCHECK(hipMalloc(&rec, blocks * sizeof(Rec)));
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, 0, out, rec, iters, 5u + rep); HIPCHECK(hipDeviceSynchronize()); }
  std::vector<Rec> h(blocks); HIPCHECK(hipMemcpy(h.data(), rec, blocks * sizeof(Rec), hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull; for (auto &r : h) t0 = std::min(t0, r.r0);
  std::map<unsigned, std::vector<int>> by_xcc; std::map<unsigned long long, int> by_cu;
  for (int i = 0; i < blocks; ++i) {
    by_xcc[h[i].xcc & 0xf].push_back(i);
    unsigned hw = h[i].hwid; unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    by_cu[((unsigned long long)(h[i].xcc & 0xf) << 16) | (se << 8) | (sh << 4) | cu]++;
  }
  printf("blocks=%d (%d per CU); distinct (xcc,se,sh,cu) used: %zu\n", blocks, per_cu, by_cu.size());
  for (auto &kv : by_xcc) {
    double smin = 1e9, smax = 0, emax = 0, dsum = 0;
    for (int i : kv.second) { double s = (h[i].r0 - t0) * 1e-2, e = (h[i].r1 - t0) * 1e-2; smin = std::min(smin, s); smax = std::max(smax, s); emax = std::max(emax, e); dsum += e - s; }
    printf(" xcc %u: %3zu blocks  start %.1f..%.1f us  last end %.1f us  mean lifetime %.1f us\n", kv.first, kv.second.size(), smin, smax, emax, dsum / kv.second.size());
  }
  std::vector<double> starts; for (auto &r : h) starts.push_back((r.r0 - t0) * 1e-2); std::sort(starts.begin(), starts.end());
  printf(" start-time percentiles us: p10 %.1f p50 %.1f p60 %.1f p90 %.1f p100 %.1f\n", starts[blocks / 10], starts[blocks / 2], starts[blocks * 6 / 10], starts[blocks * 9 / 10], starts[blocks - 1]);
  std::map<int,int> hist; for (auto &kv : by_cu) hist[kv.second]++;
  for (auto &kv : hist) printf(" CUs that ran %d blocks: %d\n", kv.first, kv.second);
  return 0;
}
