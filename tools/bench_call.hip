// Per-call time of slamhip_hamming_knn2_ratio through the C ABI from a C++ loop (no Python on
// the submitting side): wall time per call over a long run of back-to-back calls on one stream.
//   usage: bench_call [desc_bytes]      build: hipcc -O2 -o tools/bench_call tools/bench_call.hip -Lslam_amd -lslamhip -Wl,-rpath,'$ORIGIN/../slam_amd'
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../include/slamhip.h"

#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2);} } while (0)

int main(int argc, char **argv) {
  const int D = argc > 1 ? atoi(argv[1]) : 61;
  const int sizes[] = {1024, 2048, 4096, 8192};
  std::mt19937 rng(1);
  for (int n : sizes) {
    std::vector<uint8_t> hq((size_t)n * D), ht((size_t)n * D);
    for (auto &b : hq) b = rng();
    for (auto &b : ht) b = rng();
    uint8_t *dq, *dt, *keep; int32_t *di, *dd; void *ws;
    HIPCHECK(hipMalloc(&dq, hq.size())); HIPCHECK(hipMalloc(&dt, ht.size()));
    HIPCHECK(hipMemcpy(dq, hq.data(), hq.size(), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dt, ht.data(), ht.size(), hipMemcpyHostToDevice));
    HIPCHECK(hipMalloc(&di, (size_t)n * 8)); HIPCHECK(hipMalloc(&dd, (size_t)n * 8)); HIPCHECK(hipMalloc(&keep, n));
    for (int variant : {-1}) {
      for (int fold : {2}) {
        
        for (int splits : {0}) {
          slamhip_tune(0, variant); slamhip_tune(1, splits); slamhip_tune(8, fold);
          const size_t wsb = slamhip_hamming_knn2_workspace_bytes(n, n, 1, n, n, D);
          HIPCHECK(hipMalloc(&ws, wsb));
          for (int w = 0; w < 200; ++w) slamhip_hamming_knn2_ratio(dq, n, dt, n, D, D, 0.8, di, dd, keep, ws, wsb, nullptr);
          HIPCHECK(hipDeviceSynchronize());
          const int reps = 2000;
          const auto t0 = std::chrono::steady_clock::now();
          for (int r = 0; r < reps; ++r) {
            const int rc = slamhip_hamming_knn2_ratio(dq, n, dt, n, D, D, 0.8, di, dd, keep, ws, wsb, nullptr);
            if (rc) { fprintf(stderr, "rc=%d\n", rc); return 3; }
          }
          const auto t1 = std::chrono::steady_clock::now();
          HIPCHECK(hipDeviceSynchronize());
          const auto t2 = std::chrono::steady_clock::now();
          const double submit = std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
          const double total = std::chrono::duration<double, std::micro>(t2 - t0).count() / reps;
          printf("n=%5d D=%d variant %2d fold %d splits %2d: %6.2f us per call (submission alone %5.2f us)\n", n, D, variant,
                 fold, splits, total, submit);
          HIPCHECK(hipFree(ws));
        }
      }
    }
    HIPCHECK(hipFree(dq)); HIPCHECK(hipFree(dt)); HIPCHECK(hipFree(di)); HIPCHECK(hipFree(dd)); HIPCHECK(hipFree(keep));
  }
  return 0;
}
