// Issue-mix probe for gfx950: how many vector instructions (v_med3_f32 / v_max3_i32, the matcher's
// top-2 fold) and LDS reads fit beside back-to-back v_mfma_scale_f32_32x32x64_f8f6f4 (FP4) on one
// SIMD, with one and two waves per SIMD.  Everything in the loop is inline asm, in source order.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_mix_probe tools/mfma_mix_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

// per iteration: 2 matrix instructions (two accumulators), each followed by NFILL vector
// instructions and NLDS ds_read_b128
template <int NFILL, int NLDS, bool MFMA>
__global__ __launch_bounds__(256) void mix(int iters, float *sink, unsigned *cycles, int operands) {
  __shared__ v4i lds[1024];
  lds[threadIdx.x] = v4i{(int)threadIdx.x, 1, 2, 3};
  __syncthreads();
  v16f acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = (float)(threadIdx.x + i); acc1[i] = (float)(threadIdx.x * 3 + i); }
  v4i fa = {(int)(0x2A2A2A2A ^ (threadIdx.x * 0x08080808u)), 0x22AA22AA, 0x2A2AA2A2, 0x22222222};
  v4i fb = {0x2AA22AA2, (int)(0xA2A2A2A2 ^ (threadIdx.x * 0x80808080u)), 0x2222AAAA, 0x2A2A2A2A};
  // operands: 0 = +-1 in every nibble (what the matcher feeds), 1 = all zero, 2 = 0 / +1 on one
  // side and 0 / -1 on the other (half of the products are zero)
  if (operands == 1) { fa = v4i{0, 0, 0, 0}; fb = v4i{0, 0, 0, 0}; }
  if (operands == 2) { for (int j = 0; j < 4; ++j) { fa[j] &= 0x22222222; fb[j] &= (int)0xAAAAAAAA; } }
  int sc = 127;
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = (float)(threadIdx.x + j);
  float y = 3.0f, z = 5.0f;
  v4i ld[4];
  const int laddr = (threadIdx.x & 63) * 16;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (MFMA) {
        if (half == 0)
          asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4"
                       : "+v"(acc0) : "v"(fa), "v"(fb), "v"(sc));
        else
          asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4"
                       : "+v"(acc1) : "v"(fa), "v"(fb), "v"(sc));
      }
#pragma unroll
      for (int l = 0; l < NLDS; ++l)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[(half * NLDS + l) & 3]) : "v"(laddr), "n"(1024 * l));
#pragma unroll
      for (int f = 0; f < NFILL; ++f) {
        if (f & 1) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x[f & 7]) : "v"(y), "v"(z));
        else asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[f & 7]) : "v"(y), "v"(z));
      }
    }
    if (NLDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  for (int j = 0; j < 8; ++j) s += x[j];
  if (NLDS) for (int l = 0; l < 4; ++l) s += (float)ld[l][0];
  if (s == 1.2345f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = (unsigned)(t1 - t0);
}

// the 16 x 16 x 128 shape of the same instruction family: 4 accumulator registers per lane, half the
// multiply-adds per instruction; two accumulators, NFILL vector instructions behind each
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NFILL>
__global__ __launch_bounds__(256) void mix16(int iters, float *sink, unsigned *cycles) {
  v4f acc0 = {1, 2, 3, 4}, acc1 = {5, 6, 7, 8};
  v4i fa = {(int)(0x2A2A2A2A ^ (threadIdx.x * 0x08080808u)), 0x22AA22AA, 0x2A2AA2A2, 0x22222222};
  v4i fb = {0x2AA22AA2, (int)(0xA2A2A2A2 ^ (threadIdx.x * 0x80808080u)), 0x2222AAAA, 0x2A2A2A2A};
  int sc = 127;
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = (float)(threadIdx.x + j);
  float y = 3.0f, z = 5.0f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (half == 0)
        asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4"
                     : "+v"(acc0) : "v"(fa), "v"(fb), "v"(sc));
      else
        asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4"
                     : "+v"(acc1) : "v"(fa), "v"(fb), "v"(sc));
#pragma unroll
      for (int f = 0; f < NFILL; ++f) {
        if (f & 1) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x[f & 7]) : "v"(y), "v"(z));
        else asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[f & 7]) : "v"(y), "v"(z));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc0[i] + acc1[i];
  for (int j = 0; j < 8; ++j) s += x[j];
  if (s == 1.2345f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = (unsigned)(t1 - t0);
}

template <int NFILL>
void run16(int blocks, int iters, float *sink, unsigned *cyc) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((mix16<NFILL>), dim3(blocks), dim3(256), 0, 0, iters, sink, cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((mix16<NFILL>), dim3(blocks), dim3(256), 0, 0, iters, sink, cyc);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double wps = blocks / 256.0;
  printf("16x16x128  fillers/mfma %2d  waves/SIMD %.0f: %.3f ms = %.2f ns per instruction per SIMD = %.2f ns per 32x32x64 "
         "equivalent\n", NFILL, wps, ms, ms * 1e6 / (2.0 * iters * wps), 2.0 * ms * 1e6 / (2.0 * iters * wps));
}

static int g_operands = 0;

template <int NFILL, int NLDS, bool MFMA>
void run(int blocks, int iters, float *sink, unsigned *cyc) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((mix<NFILL, NLDS, MFMA>), dim3(blocks), dim3(256), 0, 0, iters, sink, cyc, g_operands);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((mix<NFILL, NLDS, MFMA>), dim3(blocks), dim3(256), 0, 0, iters, sink, cyc, g_operands);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned c;
  CK(hipMemcpy(&c, cyc, 4, hipMemcpyDeviceToHost));
  const double wps = blocks / 256.0;
  printf("mfma %d  fillers/mfma %2d  ds_read_b128/mfma %d  waves/SIMD %.0f: %.3f ms  wave: %.1f cycles per (mfma + extras), "
         "SIMD: %.1f cycles per mfma slot, clock %.2f GHz\n", (int)MFMA, NFILL, NLDS, wps, ms,
         (double)c / (2.0 * iters), (double)c / (2.0 * iters * wps), c / (ms * 1e6));
}

int main(int argc, char **argv) {
  float *sink; unsigned *cyc;
  CK(hipMalloc(&sink, 64)); CK(hipMalloc(&cyc, 64));
  const int iters = 20000;
  if (argc > 1) {   // operand study: the bare instruction and the matcher's mix, per operand kind
    for (int kind = 0; kind < 3; ++kind) {
      g_operands = kind;
      printf("operands %d (0: +-1 everywhere, 1: all zero, 2: half of the products zero)\n", kind);
      for (int rep = 0; rep < atoi(argv[1]); ++rep) {
        run<0, 0, true>(512, 4 * iters, sink, cyc);
        run<4, 1, true>(512, 4 * iters, sink, cyc);
      }
    }
    return 0;
  }
  for (int blocks : {256, 512}) {
    run<0, 0, true>(blocks, iters, sink, cyc);
    run<2, 0, true>(blocks, iters, sink, cyc);
    run<3, 0, true>(blocks, iters, sink, cyc);
    run<4, 0, true>(blocks, iters, sink, cyc);
    run<5, 0, true>(blocks, iters, sink, cyc);
    run<6, 0, true>(blocks, iters, sink, cyc);
    run<8, 0, true>(blocks, iters, sink, cyc);
    run<3, 1, true>(blocks, iters, sink, cyc);
    run<4, 1, true>(blocks, iters, sink, cyc);
    run<4, 0, false>(blocks, iters, sink, cyc);
    run<8, 0, false>(blocks, iters, sink, cyc);
    run<4, 1, false>(blocks, iters, sink, cyc);
    run16<0>(blocks, 2 * iters, sink, cyc);
    run16<1>(blocks, 2 * iters, sink, cyc);
    run16<2>(blocks, 2 * iters, sink, cyc);
    run16<3>(blocks, 2 * iters, sink, cyc);
  }
  return 0;
}
