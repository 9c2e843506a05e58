// Issue-rate probe for the floating-point VALU instructions of the consensus
// kernels: v_fma_f64, v_fma_f32 and v_pk_fma_f32.  Each kernel runs independent
// FMA chains (16 accumulators per lane) in a loop; the table reports wave-level
// instructions per second per SIMD at several occupancies.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp_probe.hip -o tools/fp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <typename T> struct One;
template <> struct One<double> { static __device__ double v(float a) { return (double)a; } };
template <> struct One<float> { static __device__ float v(float a) { return a; } };
template <> struct One<f2> { static __device__ f2 v(float a) { return f2{a, a * 1.5f}; } };

template <typename T>
__global__ __launch_bounds__(256) void probe(T *out, float seed, int iters) {
  T acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = One<T>::v(seed + (float)(i + threadIdx.x));
  const T a = One<T>::v(seed * 0.999f), b = One<T>::v(seed * 1e-3f);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = acc[i] * a + b;
  }
  T s = acc[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the compiler pairs adjacent f32 accumulators into v_pk_fma_f32, so the scalar
// f32 instruction is written out
__global__ __launch_bounds__(256) void probe_f32(float *out, float seed, int iters) {
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = seed + (float)(i + threadIdx.x);
  const float a = seed * 0.999f, b = seed * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  float s = acc[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename T>
void launch(T *out, int blocks, int iters) {
  hipLaunchKernelGGL(probe<T>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, iters);
}
template <>
void launch<float>(float *out, int blocks, int iters) {
  hipLaunchKernelGGL(probe_f32, dim3(blocks), dim3(256), 0, 0, out, 1.0f, iters);
}

template <typename T>
double run(const char *name, int blocks, int iters, double lanes_per_inst) {
  T *out;
  if (hipMalloc(&out, sizeof(T) * blocks * 256) != hipSuccess) return -1.0;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch<T>(out, blocks, iters);
  (void)hipEventRecord(e0, 0);
  launch<T>(out, blocks, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double wave_insts = (double)blocks * 4 * iters * 16;      // 4 waves per block
  const double per_simd = wave_insts / (256.0 * 4);               // 256 CUs x 4 SIMDs
  const double cyc = ms * 1e-3 * 2.4e9 / per_simd;
  std::printf("  %-14s blocks %5d  %.3f ms  %.2f cycles (at 2.4 GHz) per wave64 instruction per SIMD, "
              "%.1f T FMA/s\n", name, blocks, ms, cyc, wave_insts * 64 * lanes_per_inst / (ms * 1e-3) / 1e12);
  (void)hipFree(out);
  return cyc;
}

int main() {
  for (int waves_per_simd : {1, 2, 4, 8}) {
    const int blocks = 256 * waves_per_simd;   // 4 waves per block -> waves_per_simd per SIMD
    std::printf("%d wave(s) per SIMD\n", waves_per_simd);
    run<double>("v_fma_f64", blocks, 4000, 1);
    run<float>("v_fma_f32", blocks, 4000, 1);
    run<f2>("v_pk_fma_f32", blocks, 4000, 2);
  }
  return 0;
}
