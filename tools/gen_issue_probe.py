#!/usr/bin/env python3
"""Writes tools/issue_probe2.hip: a second VALU issue-rate map for gfx950.

Part 1 times single instruction kinds (which ones issue at the fast rate that
v_xor_b32/v_add_u32/v_fma_f32 reach and which at the slow rate of
v_bcnt_u32_b32/v_med3_u32).  Part 2 times candidate schedules of the Hamming
inner loop (16 xor + 16 popcount-accumulate + 3 top-2 instructions per train
row) so the matcher can be written in the order the hardware likes.

Every kernel is one asm block with explicit registers: v0-v15 accumulators,
v16-v31 "query" words, v32-v63 temporaries, v64-v95 "train" words.
"""
import os
import re

OPS = {
    # name: (template, note)      d = dst/accumulator, a/b = sources
    "v_xor_b32":        "v_xor_b32 {d}, {a}, {d}",
    "v_and_b32":        "v_and_b32 {d}, {a}, {d}",
    "v_sub_u32":        "v_sub_u32 {d}, {a}, {d}",
    "v_lshlrev_b32":    "v_lshlrev_b32 {d}, 1, {d}",
    "v_min_u32":        "v_min_u32 {d}, {a}, {d}",
    "v_max_u32":        "v_max_u32 {d}, {a}, {d}",
    "v_cndmask_b32":    "v_cndmask_b32 {d}, {a}, {d}, vcc",
    "v_bcnt_u32_b32":   "v_bcnt_u32_b32 {d}, {a}, {d}",
    "v_med3_u32":       "v_med3_u32 {d}, {d}, {a}, {b}",
    "v_min3_u32":       "v_min3_u32 {d}, {d}, {a}, {b}",
    "v_lshl_or_b32":    "v_lshl_or_b32 {d}, {d}, 1, {a}",
    "v_and_or_b32":     "v_and_or_b32 {d}, {d}, {a}, {b}",
    "v_or3_b32":        "v_or3_b32 {d}, {d}, {a}, {b}",
    "v_add3_u32":       "v_add3_u32 {d}, {d}, {a}, {b}",
    "v_lshl_add_u32":   "v_lshl_add_u32 {d}, {d}, 1, {a}",
    "v_add_lshl_u32":   "v_add_lshl_u32 {d}, {d}, {a}, 1",
    "v_xad_u32":        "v_xad_u32 {d}, {d}, {a}, {b}",
    "v_bfi_b32":        "v_bfi_b32 {d}, {a}, {d}, {b}",
    "v_bitop3_b32":     "v_bitop3_b32 {d}, {d}, {a}, {b} bitop3:0x96",
    "v_sad_u8":         "v_sad_u8 {d}, {a}, {b}, {d}",
    "v_sad_u32":        "v_sad_u32 {d}, {a}, {b}, {d}",
    "v_mad_u32_u24":    "v_mad_u32_u24 {d}, {a}, {b}, {d}",
    "v_dot4_u32_u8":    "v_dot4_u32_u8 {d}, {a}, {b}, {d}",
    "v_dot8_u32_u4":    "v_dot8_u32_u4 {d}, {a}, {b}, {d}",
    "v_perm_b32":       "v_perm_b32 {d}, {d}, {a}, {b}",
    "v_alignbit_b32":   "v_alignbit_b32 {d}, {d}, {a}, {b}",
    "v_bfe_u32":        "v_bfe_u32 {d}, {d}, {a}, {b}",
    "v_pk_add_u16":     "v_pk_add_u16 {d}, {d}, {a}",
    "v_pk_fma_f32":     None,   # needs register pairs, written out below
    "v_cmp_lt_u32":     "v_cmp_lt_u32 vcc, {a}, {d}",
    "v_mbcnt_lo":       "v_mbcnt_lo_u32_b32 {d}, {a}, {d}",
    "v_mul_u32_u24":    "v_mul_u32_u24 {d}, {a}, {d}",
    "v_add_f32":        "v_add_f32 {d}, {a}, {d}",
    "v_ffbh_u32":       "v_ffbh_u32 {d}, {a}",
    "v_bfrev_b32":      "v_bfrev_b32 {d}, {a}",
    "v_cvt_f32_ubyte0": "v_cvt_f32_ubyte0 {d}, {a}",
    # lane-mask consumers and producers (the consensus-counting loop)
    "v_cndmask_sgpr":   "v_cndmask_b32_e64 {d}, {a}, {d}, s[22:23]",
    "v_cndmask_vcc_e64": "v_cndmask_b32_e64 {d}, {a}, {d}, vcc",
    "v_cndmask_2src":   "v_cndmask_b32_e32 {d}, {a}, {b}, vcc",
    "v_cmp_then_cnd":   "v_cmp_lt_f32 vcc, {a}, {b}\\nv_cndmask_b32_e32 {d}, {a}, {d}, vcc",
    "v_cndmask_imm":    "v_cndmask_b32_e64 {d}, 0, 1, s[22:23]",
    "v_addc_vcc":       "v_addc_co_u32 {d}, vcc, 0, {d}, vcc",
    "v_addc_sgpr":      "v_addc_co_u32_e64 {d}, s[24:25], 0, {d}, s[22:23]",
    "v_cmp_lt_f32_vcc": "v_cmp_lt_f32 vcc, {a}, {d}",
    "v_cmp_lt_f32_sgpr": "v_cmp_lt_f32_e64 s[24:25], {a}, {d}",
    "v_cmp_abs_f32":    "v_cmp_lt_f32_e64 s[24:25], |{a}|, {d}",
    "v_max_f32":        "v_max_f32 {d}, {a}, {d}",
    "v_min_f32":        "v_min_f32 {d}, {a}, {d}",
    "v_med3_f32":       "v_med3_f32 {d}, {d}, {a}, {b}",
    "v_mul_f32":        "v_mul_f32 {d}, {a}, {d}",
    "v_pk_mul_f32":     None,
    "v_pk_add_f32":     None,
    "v_fma_f64":        None,
    "v_or_b32":         "v_or_b32 {d}, {a}, {d}",
    "v_xor_neg":        "v_fma_f32 {d}, {a}, {b}, -{d}",
    "v_subrev_u32":     "v_subrev_u32 {d}, {a}, {d}",
    "v_ashrrev_i32":    "v_ashrrev_i32 {d}, 31, {d}",
    "v_and_sgpr":       "v_and_b32 {d}, s22, {d}",
    "v_xor_lit":        "v_xor_b32 {d}, 0x80000000, {d}",
}


def single(op):
    t = OPS[op]
    lines = []
    if t is None:   # packed / 64-bit operations on register pairs
        for i in range(0, 32, 2):
            d = f"v[{i % 16 // 2 * 2}:{i % 16 // 2 * 2 + 1}]"
            if op == "v_pk_fma_f32":
                lines.append(f"v_pk_fma_f32 {d}, v[16:17], v[32:33], {d}")
            elif op == "v_pk_mul_f32":
                lines.append(f"v_pk_mul_f32 {d}, v[16:17], {d}")
            elif op == "v_pk_add_f32":
                lines.append(f"v_pk_add_f32 {d}, v[16:17], {d}")
            else:
                lines.append(f"v_fma_f64 {d}, v[16:17], v[32:33], {d}")
        return lines
    for i in range(32):
        lines.append(t.format(d=f"v{i % 16}", a=f"v{16 + i % 16}", b=f"v{32 + i % 16}"))
    return lines


def row_xor(tmp0, train0, n=16):
    return [f"v_xor_b32 v{tmp0 + i}, v{16 + i}, v{train0 + i}" for i in range(n)]


def row_bcnt(acc, tmp0, n=16, first_zero=False):
    out = []
    for i in range(n):
        src1 = "0" if (first_zero and i == 0) else f"v{acc}"
        out.append(f"v_bcnt_u32_b32 v{acc}, v{tmp0 + i}, {src1}")
    return out


def top2(acc, b1=1, b2=2):
    return [f"v_lshl_or_b32 v{acc}, v{acc}, 21, v3", f"v_med3_u32 v{b2}, v{b1}, v{b2}, v{acc}",
            f"v_min_u32 v{b1}, v{b1}, v{acc}"]


def sched(kind):
    """Two train rows per body (so every body is 70 VALU instructions)."""
    L = []
    if kind == "x16b16":            # row: 16 xor, 16 bcnt, top2
        for r in range(2):
            L += row_xor(32, 64 + 16 * r) + row_bcnt(0, 32, first_zero=True) + top2(0)
    elif kind == "x8b8":
        for r in range(2):
            for h in range(2):
                L += [f"v_xor_b32 v{32 + i}, v{16 + 8 * h + i}, v{64 + 16 * r + 8 * h + i}" for i in range(8)]
                L += [f"v_bcnt_u32_b32 v0, v{32 + i}, " + ("0" if (h == 0 and i == 0) else "v0") for i in range(8)]
            L += top2(0)
    elif kind == "x4b4":
        for r in range(2):
            for h in range(4):
                L += [f"v_xor_b32 v{32 + i}, v{16 + 4 * h + i}, v{64 + 16 * r + 4 * h + i}" for i in range(4)]
                L += [f"v_bcnt_u32_b32 v0, v{32 + i}, " + ("0" if (h == 0 and i == 0) else "v0") for i in range(4)]
            L += top2(0)
    elif kind == "x1b1":            # fully interleaved, bcnt one behind
        for r in range(2):
            x = row_xor(32, 64 + 16 * r)
            b = row_bcnt(0, 32, first_zero=True)
            L.append(x[0])
            for i in range(1, 16):
                L += [x[i], b[i - 1]]
            L.append(b[15])
            L += top2(0)
    elif kind == "x32b32":          # both rows' xors, then both rows' bcnts (two chains interleaved), then top2 x2
        L += row_xor(32, 64) + row_xor(48, 80)
        b0 = row_bcnt(0, 32, first_zero=True)
        b1 = row_bcnt(4, 48, first_zero=True)
        for i in range(16):
            L += [b0[i], b1[i]]
        L += top2(0) + top2(4)
    elif kind == "x32b32seq":       # both rows' xors, then chain 0, then chain 1
        L += row_xor(32, 64) + row_xor(48, 80)
        L += row_bcnt(0, 32, first_zero=True) + row_bcnt(4, 48, first_zero=True)
        L += top2(0) + top2(4)
    elif kind == "x16b16_2acc":     # one row, bcnt alternates two accumulators, one add joins them
        for r in range(2):
            L += row_xor(32, 64 + 16 * r)
            for i in range(16):
                acc = 0 if i % 2 == 0 else 4
                src1 = "0" if i < 2 else f"v{acc}"
                L.append(f"v_bcnt_u32_b32 v{acc}, v{32 + i}, {src1}")
            L += ["v_add_u32 v0, v0, v4"] + top2(0)
    elif kind == "x16b16_lds":      # x16b16 with the next row's 4 broadcast ds_read_b128 in front
        for r in range(2):
            L += [f"ds_read_b128 v[{64 + 16 * r + 4 * j}:{64 + 16 * r + 4 * j + 3}], v96 offset:{64 * r + 16 * j}"
                  for j in range(4)]
            L += row_xor(32, 64 + 16 * (1 - r)) + row_bcnt(0, 32, first_zero=True) + top2(0)
            L += ["s_waitcnt lgkmcnt(0)"]
    elif kind == "bcnt_only":       # the 32 bcnt + top2, no xor: the floor set by the slow class
        for r in range(2):
            L += row_bcnt(0, 32, first_zero=True) + top2(0)
    elif kind == "xor_only":
        for r in range(2):
            L += row_xor(32, 64 + 16 * r)
    return L


SCHEDS = ["x16b16", "x8b8", "x4b4", "x1b1", "x32b32", "x32b32seq", "x16b16_2acc", "x16b16_lds",
          "bcnt_only", "xor_only"]


def kernel(name, body, valu_count, uses_lds=False):
    used = [96]
    for l in body:
        used += [int(x) for x in re.findall(r"v(\d+)", l)]
        used += [int(x) for pair in re.findall(r"v\[(\d+):(\d+)\]", l) for x in pair]
    top = max(used) if uses_lds else max(u for u in used if u != 96 or uses_lds)
    addr = f"v{top + 1}" if uses_lds else "v0"
    body = [l.replace("v96", addr) for l in body]
    clob = ", ".join(f'"v{i}"' for i in range(0, top + 2))
    asm = "\\n\"\n      \"".join(body)
    return f"""
__global__ __launch_bounds__(256) void {name}(uint32_t *out, unsigned long long *st, int iters) {{
  {'__shared__ uint32_t lds[1024]; lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 256] = 1; lds[threadIdx.x + 512] = 2; lds[threadIdx.x + 768] = 3; __syncthreads();' if uses_lds else ''}
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  uint32_t res;
  asm volatile(
      "v_mov_b32 {addr}, 0\\n"
      "s_mov_b32 s20, %1\\n"
      "s_mov_b32 s22, 0x55555555\\n"
      "s_mov_b64 vcc, 0x33\\n"
      "s_mov_b32 s23, 0x33333333\\n"
      "PROBE_LOOP_%=:\\n"
      "{asm}\\n"
      "s_sub_u32 s20, s20, 1\\n"
      "s_cmp_lg_u32 s20, 0\\n"
      "s_cbranch_scc1 PROBE_LOOP_%=\\n"
      "v_add_u32 %0, v0, v1\\n"
      : "=v"(res) : "s"(iters) : {clob}, "s20", "s22", "s23", "s24", "s25", "scc", "vcc", "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = res;
  if (threadIdx.x == 0) st[blockIdx.x] = t1 - t0;
}}
"""


def main():
    ks = []
    src = ["// Generated by tools/gen_issue_probe.py; do not edit.",
           "#include <hip/hip_runtime.h>", "#include <cstdio>", "#include <cstdlib>", "#include <vector>",
           "#include <algorithm>",
           "#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, \"HIP error %s at %s:%d\\n\", hipGetErrorString(e), __FILE__, __LINE__); exit(2);} } while (0)"]
    for op in OPS:
        body = single(op)
        name = "k_" + op
        src.append(kernel(name, body, len(body)))
        ks.append((op, name, len(body), 0))
    for s in SCHEDS:
        body = sched(s)
        nv = sum(1 for l in body if l.startswith("v_"))
        rows = 0 if s == "xor_only" else 2
        src.append(kernel("s_" + s, body, nv, uses_lds=(s == "x16b16_lds")))
        ks.append((s, "s_" + s, nv, rows))
    table = ",\n    ".join(f'{{"{n}", {k}, {nv}, {rows}}}' for n, k, nv, rows in ks)
    src.append(f"""
typedef void (*kern_t)(uint32_t *, unsigned long long *, int);
struct Entry {{ const char *name; kern_t k; int valu; int rows; }};
int main() {{
  hipDeviceProp_t prop; HIPCHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, iters = 3000;
  uint32_t *out; unsigned long long *st;
  HIPCHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
  HIPCHECK(hipMalloc(&st, (size_t)cus * 8 * 8));
  hipEvent_t e0, e1; HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
  Entry ks[] = {{
    {table}}};
  printf("device CUs=%d; T = 1e12 lane-instructions per second chip-wide (a packed or 64-bit instruction counts once per lane) (wall time of 10 launches); rows/s in 1e9 (64 queries x 1 train row = 64 compares)\\n", cus);
  for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(ks[0].k, dim3(cus * 8), dim3(256), 0, 0, out, st, iters);
  HIPCHECK(hipDeviceSynchronize());
  const int ws[] = {{2, 4, 5, 8}};   // waves per SIMD resident (100 VGPRs allow 5; 8 only with fewer)
  for (auto &k : ks) {{
    printf("%-18s", k.name);
    for (int w : ws) {{
      const int blocks = cus * w, reps = 10;
      hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, st, iters);
      HIPCHECK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, st, iters);
      HIPCHECK(hipEventRecord(e1)); HIPCHECK(hipEventSynchronize(e1));
      float ms; HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
      const double lane_ops = (double)blocks * 256 * (double)iters * k.valu * reps;
      printf(" | w%d %5.1f T", w, lane_ops / (ms * 1e-3) * 1e-12);
      if (k.rows) printf(" %5.2fe12 cmp/s", (double)blocks * 256 * (double)iters * k.rows * reps / (ms * 1e-3) * 1e-12);
    }}
    printf("\\n");
  }}
  return 0;
}}
""")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "issue_probe2.hip")
    with open(path, "w") as f:
        f.write("\n".join(src))
    print("wrote", path)


if __name__ == "__main__":
    main()
