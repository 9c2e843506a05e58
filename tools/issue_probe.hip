// VALU issue-rate map for gfx950: how many shader cycles one SIMD needs per
// wave64 instruction, by instruction kind, operand source and waves per SIMD.
// The chip is kept busy for >100 ms before anything is timed (clock ramp), and
// every figure is the wall time of many back-to-back launches plus the median
// in-kernel s_memtime span of a wave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/issue_probe tools/issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2);} } while (0)

#define R16 "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
            "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
// operands: %0..%15 accumulators, %16 sgpr, %17 vgpr a, %18 vgpr b
#define REP16(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)

#define XOR_S(i)   "v_xor_b32 %" #i ", %16, %" #i "\n"
#define XOR_V(i)   "v_xor_b32 %" #i ", %17, %" #i "\n"
#define XOR_C(i)   "v_xor_b32 %" #i ", 21, %" #i "\n"
#define MOV_V(i)   "v_mov_b32 %" #i ", %17\n"
#define BCNT_V(i)  "v_bcnt_u32_b32 %" #i ", %17, %" #i "\n"
#define BCNT_S(i)  "v_bcnt_u32_b32 %" #i ", %16, %" #i "\n"
#define FMA_V(i)   "v_fma_f32 %" #i ", %17, %18, %" #i "\n"
#define FMAC_V(i)  "v_fmac_f32 %" #i ", %17, %18\n"
#define ADD_V(i)   "v_add_u32 %" #i ", %17, %" #i "\n"
#define MED3_(i)   "v_med3_u32 %" #i ", %" #i ", %17, %18\n"
#define PKFMA_(i)  ""
// the matcher's chain: 8 xor into temporaries (%8..%15) then 8 dependent bcnt into ONE accumulator (%0)
#define CHAIN_X(t) "v_xor_b32 %" #t ", %17, %" #t "\n"
#define CHAIN_B(t) "v_bcnt_u32_b32 %0, %" #t ", %0\n"
#define CHAIN  CHAIN_X(8) CHAIN_X(9) CHAIN_X(10) CHAIN_X(11) CHAIN_X(12) CHAIN_X(13) CHAIN_X(14) CHAIN_X(15) \
               CHAIN_B(8) CHAIN_B(9) CHAIN_B(10) CHAIN_B(11) CHAIN_B(12) CHAIN_B(13) CHAIN_B(14) CHAIN_B(15)
// interleaved as a compiler would schedule it
#define CHAIN2 CHAIN_X(8) CHAIN_X(9) CHAIN_B(8) CHAIN_X(10) CHAIN_B(9) CHAIN_X(11) CHAIN_B(10) CHAIN_X(12) \
               CHAIN_B(11) CHAIN_X(13) CHAIN_B(12) CHAIN_X(14) CHAIN_B(13) CHAIN_X(15) CHAIN_B(14) CHAIN_B(15)
// two accumulators (two train rows in flight)
#define CH2_X(t) "v_xor_b32 %" #t ", %17, %" #t "\n"
#define CH2_B0(t) "v_bcnt_u32_b32 %0, %" #t ", %0\n"
#define CH2_B1(t) "v_bcnt_u32_b32 %1, %" #t ", %1\n"
#define CHAIN3 CH2_X(8) CH2_X(9) CH2_B0(8) CH2_B1(9) CH2_X(10) CH2_X(11) CH2_B0(10) CH2_B1(11) \
               CH2_X(12) CH2_X(13) CH2_B0(12) CH2_B1(13) CH2_X(14) CH2_X(15) CH2_B0(14) CH2_B1(15)

#define PROBE(NAME, BODY)                                                                     \
  __global__ __launch_bounds__(256) void NAME(uint32_t *out, unsigned long long *st, uint32_t seed, int iters) { \
    uint32_t r[16];                                                                           \
    for (int i = 0; i < 16; ++i) r[i] = (threadIdx.x + 1) * 2654435761u * (i + 1) + seed;    \
    uint32_t k = seed, a = threadIdx.x * 40503u + seed, b = threadIdx.x * 7919u + seed;       \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                     \
    for (int i = 0; i < iters; ++i) { asm volatile(BODY BODY : R16 : "s"(k), "v"(a), "v"(b)); } \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                     \
    uint32_t s = 0;                                                                           \
    for (int i = 0; i < 16; ++i) s += r[i];                                                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                           \
    if (threadIdx.x == 0) st[blockIdx.x] = t1 - t0;                                           \
  }

PROBE(p_xor_s, REP16(XOR_S))
PROBE(p_xor_v, REP16(XOR_V))
PROBE(p_xor_c, REP16(XOR_C))
PROBE(p_mov_v, REP16(MOV_V))
PROBE(p_bcnt_v, REP16(BCNT_V))
PROBE(p_bcnt_s, REP16(BCNT_S))
PROBE(p_fma_v, REP16(FMA_V))
PROBE(p_fmac_v, REP16(FMAC_V))
PROBE(p_add_v, REP16(ADD_V))
PROBE(p_med3, REP16(MED3_))
PROBE(p_chain, CHAIN)
PROBE(p_chain2, CHAIN2)
PROBE(p_chain3, CHAIN3)

typedef void (*kern_t)(uint32_t *, unsigned long long *, uint32_t, int);

int main(int argc, char **argv) {
  hipDeviceProp_t prop; HIPCHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int iters = 4000;            // x 32 instructions per iteration
  uint32_t *out; unsigned long long *st;
  HIPCHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
  HIPCHECK(hipMalloc(&st, (size_t)cus * 8 * 8));
  hipEvent_t e0, e1; HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
  struct { const char *name; kern_t k; } ks[] = {
    {"v_xor_b32 sgpr,vgpr", p_xor_s}, {"v_xor_b32 vgpr,vgpr", p_xor_v}, {"v_xor_b32 const,vgpr", p_xor_c},
    {"v_mov_b32 vgpr", p_mov_v}, {"v_add_u32 vgpr,vgpr", p_add_v},
    {"v_bcnt vgpr,vgpr", p_bcnt_v}, {"v_bcnt sgpr,vgpr", p_bcnt_s}, {"v_med3_u32", p_med3},
    {"v_fma_f32", p_fma_v}, {"v_fmac_f32", p_fmac_v},
    {"8 xor + 8 bcnt (1 chain)", p_chain}, {"same, interleaved", p_chain2}, {"same, 2 chains", p_chain3}};
  printf("device CUs=%d nominal=%d MHz; 32 instructions x %d iterations per wave\n", cus, prop.clockRate / 1000, iters);
  // ramp the clocks
  for (int i = 0; i < 60; ++i) hipLaunchKernelGGL(p_xor_v, dim3(cus * 8), dim3(256), 0, 0, out, st, 1u, iters);
  HIPCHECK(hipDeviceSynchronize());
  const int ws[] = {1, 2, 3, 4, 6, 8};
  for (auto &k : ks) {
    printf("%-26s", k.name);
    for (int w : ws) {
      const int blocks = cus * w, reps = 12;
      hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, st, 3u, iters);
      HIPCHECK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, st, 7u + r, iters);
      HIPCHECK(hipEventRecord(e1)); HIPCHECK(hipEventSynchronize(e1));
      float ms; HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> h(blocks);
      HIPCHECK(hipMemcpy(h.data(), st, blocks * 8, hipMemcpyDeviceToHost));
      std::sort(h.begin(), h.end());
      const double insts = (double)iters * 32.0;
      const double wave_cyc = (double)h[blocks / 2] / insts;          // cycles a wave needs per own instruction
      const double tlops = (double)blocks * 256 * insts * reps / (ms * 1e-3) * 1e-12;
      printf(" | w%d %5.2f cyc/SIMD-inst %5.1f T", w, wave_cyc / w, tlops);
    }
    printf("\n");
  }
  return 0;
}
