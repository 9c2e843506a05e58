// Stand-alone timing harness (HIP events) for the matcher kernel variants and a
// VALU issue-rate probe for v_xor_b32 + v_bcnt_u32_b32.  Links libslamhip.so.
//   usage: bench_matcher [nq nt desc_bytes reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include "../include/slamhip.h"

#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2);} } while (0)

// 8 independent xor+bcnt chains per lane, ITER iterations: 16 VALU ops per iteration.
__global__ __launch_bounds__(256) void valu_probe(uint32_t *out, uint32_t seed, int iters) {
  uint32_t x0 = threadIdx.x * 2654435761u + seed, x1 = x0 ^ 0x9e3779b9u, x2 = x0 * 3u, x3 = x0 * 5u;
  uint32_t x4 = x0 * 7u, x5 = x0 * 11u, x6 = x0 * 13u, x7 = x0 * 17u;
  uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
  uint32_t k = seed;
  for (int i = 0; i < iters; ++i) {
    asm volatile(
        "v_xor_b32 %8, %16, %8\n v_bcnt_u32_b32 %0, %8, %0\n"
        "v_xor_b32 %9, %16, %9\n v_bcnt_u32_b32 %1, %9, %1\n"
        "v_xor_b32 %10, %16, %10\n v_bcnt_u32_b32 %2, %10, %2\n"
        "v_xor_b32 %11, %16, %11\n v_bcnt_u32_b32 %3, %11, %3\n"
        "v_xor_b32 %12, %16, %12\n v_bcnt_u32_b32 %4, %12, %4\n"
        "v_xor_b32 %13, %16, %13\n v_bcnt_u32_b32 %5, %13, %5\n"
        "v_xor_b32 %14, %16, %14\n v_bcnt_u32_b32 %6, %14, %6\n"
        "v_xor_b32 %15, %16, %15\n v_bcnt_u32_b32 %7, %15, %7\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
          "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
        : "s"(k));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; HIPCHECK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char **argv) {
  int nq = argc > 1 ? atoi(argv[1]) : 8192, nt = argc > 2 ? atoi(argv[2]) : 8192;
  int D = argc > 3 ? atoi(argv[3]) : 61, reps = argc > 4 ? atoi(argv[4]) : 20;
  hipDeviceProp_t prop; HIPCHECK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  clock=%d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
  hipEvent_t e0, e1; HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));

  { // VALU probe
    uint32_t *out; const int blocks = prop.multiProcessorCount * 8, iters = 20000;
    HIPCHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    for (int r = 0; r < 3; ++r) {
      HIPCHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(valu_probe, dim3(blocks), dim3(256), 0, 0, out, 123u + r, iters);
      HIPCHECK(hipEventRecord(e1)); HIPCHECK(hipEventSynchronize(e1));
      double ops = (double)blocks * 256 * iters * 16.0;
      float ms = time_ms(e0, e1);
      printf("valu_probe: %.3f ms  %.2f Tlane-op/s (xor+bcnt mix)  => %.2f lane-ops/clk/CU at %d MHz\n", ms,
             ops / ms * 1e-9, ops / (ms * 1e-3) / prop.multiProcessorCount / (prop.clockRate * 1e3), prop.clockRate / 1000);
    }
    HIPCHECK(hipFree(out));
  }

  std::mt19937 rng(1);
  std::vector<uint8_t> hq((size_t)nq * D), ht((size_t)nt * D);
  for (auto &b : hq) b = rng(); for (auto &b : ht) b = rng();
  uint8_t *dq, *dt; int32_t *di, *dd; void *ws;
  HIPCHECK(hipMalloc(&dq, hq.size())); HIPCHECK(hipMalloc(&dt, ht.size()));
  HIPCHECK(hipMemcpy(dq, hq.data(), hq.size(), hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(dt, ht.data(), ht.size(), hipMemcpyHostToDevice));
  HIPCHECK(hipMalloc(&di, (size_t)nq * 8)); HIPCHECK(hipMalloc(&dd, (size_t)nq * 8));
  size_t wsb = slamhip_hamming_knn2_workspace_bytes(nq, nt, 1, nq, nt, D);
  HIPCHECK(hipMalloc(&ws, wsb));
  std::vector<int32_t> ref((size_t)nq * 2), got((size_t)nq * 2);
  const int variants[] = {0, 1, 4, 5}; const int splits[] = {0, 8, 16, 32, 64};
  bool have_ref = false;
  for (int v : variants) for (int s : splits) {
    slamhip_tune(0, v); slamhip_tune(1, s);
    for (int w = 0; w < 3; ++w) slamhip_hamming_knn2(dq, nq, dt, nt, D, D, di, dd, ws, wsb, nullptr);
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) {
      int rc = slamhip_hamming_knn2(dq, nq, dt, nt, D, D, di, dd, ws, wsb, nullptr);
      if (rc) { fprintf(stderr, "rc=%d\n", rc); return 3; }
    }
    HIPCHECK(hipEventRecord(e1)); HIPCHECK(hipEventSynchronize(e1));
    float ms = time_ms(e0, e1) / reps;
    HIPCHECK(hipMemcpy(got.data(), di, got.size() * 4, hipMemcpyDeviceToHost));
    if (!have_ref) { ref = got; have_ref = true; }
    bool same = memcmp(ref.data(), got.data(), got.size() * 4) == 0;
    printf("variant=%d(%s,Q%d) splits=%2d : %8.2f us/call  %.3e compares/s  %s\n", v, (v & 1) ? "lds " : "sgpr",
           (v >> 1) == 2 ? 2 : 1, s, ms * 1e3, (double)nq * nt / (ms * 1e-3), same ? "same" : "DIFFERENT");
  }
  return 0;
}
