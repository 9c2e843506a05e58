// Shader clock under sustained load on gfx950: cycles of s_memtime (shader clock) per tick of
// s_memrealtime (constant 100 MHz) while the whole chip runs (a) packed-fp32 arithmetic,
// (b) v_xor / v_bcnt integer work, (c) fp64 FMAs, for about a millisecond per launch.
//   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/clock_probe && tools/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void load_kernel(unsigned long long *stamps, float *sink, int iters) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f2 a[8];
  double d[4];
  uint32_t u[8];
  for (int i = 0; i < 8; ++i) { a[i] = f2{1.0f + threadIdx.x * 1e-3f + i, 0.5f + i}; u[i] = threadIdx.x * 2654435761u + i; }
  for (int i = 0; i < 4; ++i) d[i] = 1.0 + threadIdx.x * 1e-6 + i;
  const f2 m = {1.0000001f, 0.9999999f}, c = {1e-7f, -1e-7f};
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
    } else if (KIND == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) { u[i] ^= u[(i + 1) & 7]; u[i] = __builtin_popcount(u[i]) + u[(i + 3) & 7]; }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = fma(d[i], 1.0000000001, 1e-12);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.0f;
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y + (float)u[i];
  for (int i = 0; i < 4; ++i) s += (float)d[i];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <int KIND>
static void run(const char *name, int iters, int instr_per_iter) {
  const int blocks = 256 * 8;
  unsigned long long *st;
  float *sink;
  hipMalloc(&st, blocks * 16);
  hipMalloc(&sink, 4);
  for (int rep = 0; rep < 6; ++rep) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(load_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, st, sink, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2 * 256 * 8];
    hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int b = 0; b < blocks; ++b) { cyc += (double)h[2 * b]; real += (double)h[2 * b + 1]; }
    printf("%-22s launch %d: %.0f MHz (block time %.0f us, launch %.0f us, %.2f cycles per instruction and SIMD)\n",
           name, rep, cyc / real * 100.0, real / blocks / 100.0, ms * 1e3,
           ms * 1e-3 * (cyc / real * 1e8) / ((double)blocks * 4 / 1024 * iters * instr_per_iter));
  }
  hipFree(st);
  hipFree(sink);
}

int main() {
  run<0>("packed fp32 fma", 6000, 32);      // 32 v_pk_fma_f32 per iteration
  run<1>("xor + popcount", 6000, 64);       // 32 v_xor_b32 + 32 v_bcnt_u32_b32
  run<2>("fp64 fma", 6000, 16);             // 16 v_fma_f64
  return 0;
}
