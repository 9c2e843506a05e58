// Issue-rate probe for the VALU instructions the Hamming matcher is made of.
// Each kernel runs 16 independent instructions of one kind per loop iteration
// on 8 waves per SIMD and reports lane-ops per clock per CU, with the shader
// clock measured in-kernel (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2);} } while (0)

#define REGS16 "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
               "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])

#define BODY16(INSN_FMT) \
  INSN_FMT(0) INSN_FMT(1) INSN_FMT(2) INSN_FMT(3) INSN_FMT(4) INSN_FMT(5) INSN_FMT(6) INSN_FMT(7) \
  INSN_FMT(8) INSN_FMT(9) INSN_FMT(10) INSN_FMT(11) INSN_FMT(12) INSN_FMT(13) INSN_FMT(14) INSN_FMT(15)

#define XOR_(i)  "v_xor_b32 %" #i ", %16, %" #i "\n"
#define BCNT_(i) "v_bcnt_u32_b32 %" #i ", %17, %" #i "\n"
#define ADD_(i)  "v_add_u32 %" #i ", %16, %" #i "\n"
#define ANDOR_(i) "v_and_or_b32 %" #i ", %" #i ", %16, %17\n"
#define MED3_(i) "v_med3_u32 %" #i ", %" #i ", %17, %16\n"
#define BFI_(i)  "v_bfi_b32 %" #i ", %17, %" #i ", %16\n"
#define LSHLOR_(i) "v_lshl_or_b32 %" #i ", %" #i ", 1, %17\n"
#define MIN_(i)  "v_min_u32 %" #i ", %16, %" #i "\n"
// xor into a temp then bcnt-accumulate (the matcher's pair), 8 pairs interleaved
#define PAIRX_(i, t) "v_xor_b32 %" #t ", %16, %" #t "\n"
#define PAIRB_(i, t) "v_bcnt_u32_b32 %" #i ", %" #t ", %" #i "\n"

struct Stamp { unsigned long long t0, t1, r0, r1; };

#define PROBE_KERNEL(NAME, ASM_BODY)                                                          \
  __global__ __launch_bounds__(256) void NAME(uint32_t *out, Stamp *st, uint32_t seed, int iters) { \
    uint32_t r[16];                                                                           \
    for (int i = 0; i < 16; ++i) r[i] = (threadIdx.x + 1) * 2654435761u * (i + 1) + seed;    \
    uint32_t k = seed, v = threadIdx.x * 40503u + seed;                                       \
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
    for (int i = 0; i < iters; ++i) { asm volatile(ASM_BODY : REGS16 : "s"(k), "v"(v)); }     \
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
    uint32_t s = 0;                                                                           \
    for (int i = 0; i < 16; ++i) s += r[i];                                                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                           \
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{t0, t1, r0, r1};                              \
  }

PROBE_KERNEL(probe_xor, BODY16(XOR_))
PROBE_KERNEL(probe_bcnt, BODY16(BCNT_))
PROBE_KERNEL(probe_add, BODY16(ADD_))
PROBE_KERNEL(probe_andor, BODY16(ANDOR_))
PROBE_KERNEL(probe_med3, BODY16(MED3_))
PROBE_KERNEL(probe_bfi, BODY16(BFI_))
PROBE_KERNEL(probe_lshlor, BODY16(LSHLOR_))
PROBE_KERNEL(probe_min, BODY16(MIN_))
PROBE_KERNEL(probe_pair,
  PAIRX_(0,8) PAIRX_(1,9) PAIRX_(2,10) PAIRX_(3,11) PAIRX_(4,12) PAIRX_(5,13) PAIRX_(6,14) PAIRX_(7,15)
  PAIRB_(0,8) PAIRB_(1,9) PAIRB_(2,10) PAIRB_(3,11) PAIRB_(4,12) PAIRB_(5,13) PAIRB_(6,14) PAIRB_(7,15))

typedef void (*kern_t)(uint32_t *, Stamp *, uint32_t, int);

int main(int argc, char **argv) {
  const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 8;
  hipDeviceProp_t prop; HIPCHECK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * waves_per_simd, iters = 20000;
  uint32_t *out; Stamp *st;
  HIPCHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  HIPCHECK(hipMalloc(&st, (size_t)blocks * sizeof(Stamp)));
  hipEvent_t e0, e1; HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
  struct { const char *name; kern_t k; } ks[] = {
    {"v_xor_b32 (VOP2, sgpr src)", probe_xor}, {"v_bcnt_u32_b32", probe_bcnt},
    {"v_med3_u32", probe_med3}, {"xor+bcnt pairs", probe_pair}};
  printf("device CUs=%d nominal=%d MHz, %d waves/SIMD, %d blocks\n", prop.multiProcessorCount, prop.clockRate / 1000, waves_per_simd, blocks);
  for (auto &k : ks) {
    for (int rep = 0; rep < 2; ++rep) {
      HIPCHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, st, 7u + rep, iters);
      HIPCHECK(hipEventRecord(e1)); HIPCHECK(hipEventSynchronize(e1));
      float ms; HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<Stamp> h(blocks);
      HIPCHECK(hipMemcpy(h.data(), st, blocks * sizeof(Stamp), hipMemcpyDeviceToHost));
      std::vector<double> mhz, cyc;
      for (auto &s : h) { mhz.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 100.0); cyc.push_back((double)(s.t1 - s.t0)); }
      std::sort(mhz.begin(), mhz.end()); std::sort(cyc.begin(), cyc.end());
      unsigned long long rmin = ~0ull, rmax = 0; std::vector<double> dur;
      for (auto &x : h) { rmin = std::min(rmin, x.r0); rmax = std::max(rmax, x.r1); dur.push_back((double)(x.r1 - x.r0) * 1e-5); }
      std::sort(dur.begin(), dur.end());
      const double span_ms = (double)(rmax - rmin) * 1e-5, blk_ms = dur[dur.size() / 2];
      const double ops = (double)blocks * 256 * iters * 16.0;
      // cycles one wave spent per own instruction, and the per-SIMD issue interval with W waves sharing it
      const double cyc_per_inst_wave = cyc[cyc.size() / 2] / (iters * 16.0);
      if (rep == 1)
        printf("%-28s %7.3f ms  %6.2f Tlane-op/s  clock(median) %4.0f MHz  %.2f cyc/wave-inst/SIMD  %.1f lane-ops/clk/CU\n",
               k.name, ms, ops / ms * 1e-9, mhz[mhz.size() / 2], cyc_per_inst_wave / waves_per_simd,
               64.0 * 4 / (cyc_per_inst_wave / waves_per_simd));
      if (rep == 1) printf("      span %.3f ms, median block lifetime %.3f ms -> %.2f rounds\n", span_ms, blk_ms, span_ms / blk_ms);
    }
  }
  return 0;
}
