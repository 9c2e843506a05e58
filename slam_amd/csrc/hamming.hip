// Hamming brute-force top-2 matcher for gfx950 (MI355X).
//
// Work decomposition
//   lane      = Q query descriptors held in registers (W dwords each)
//   wave      = 64*Q queries; walks a slice of the train rows, one row at a
//               time, the row being wave-uniform (scalar loads -> SGPR operands,
//               or an LDS-staged tile read with a broadcast address)
//   block     = 256 threads = 256*Q consecutive queries of one pair
//   grid      = (query blocks, train splits, pairs)
// Per (query, train row): W x (v_xor_b32 + v_bcnt_u32_b32 accumulate), then the
// distance and the train index are packed into one 32-bit key
//   key = dist << 21 | train_index
// so that "smaller distance, then lower train index" is plain unsigned order and
// the running top-2 update is v_med3_u32 + v_min_u32.  Every train split writes
// its (best, second) key pair; a merge kernel folds the splits in index order.
//
// Tie rule = cv2.BFMatcher / batchDistance: scanning train rows in ascending
// order with a strict `<` insertion keeps the lowest train index among equal
// distances, for the best and for the second best
// (call site: frontend/vision/descriptor_matcher.py:33 of the reference).
#include "common.h"
#include <hip/hip_ext.h>
#include "sort_device.h"
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>

namespace {

constexpr int kKeyShift = slamhip_dev::kSortKeyShift;   // one key layout for every sort kernel
constexpr uint32_t kIdxMask = (1u << kKeyShift) - 1u;
constexpr uint32_t kNoKey = 0xFFFFFFFFu;
constexpr int kThreads = 256;
constexpr int kTileRows = 128;  // LDS variant: train rows staged per tile
constexpr int kMaxSplits = 64;

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// Running top-2 of unsigned keys; invariant b1 <= b2.
__device__ __forceinline__ void top2_push(uint32_t &b1, uint32_t &b2, uint32_t key) {
  b2 = umed3(b1, b2, key);
  b1 = min(b1, key);
}

// ---------------------------------------------------------------------------
// Scan of one LDS tile of 16-word train rows by a wave whose lanes each hold one
// query, written out instruction by instruction.
//
// Why: measured on gfx950 (tools/issue_probe2), v_xor_b32 with two VGPR sources
// issues every ~2.2 cycles per wave, v_bcnt_u32_b32 (like the shifts, min/max/
// med3 and every other VOP3 integer operation tried) every ~4, and a stream that
// alternates the two kinds pays for every switch: 16 xors in a row followed by
// the 16 dependent popcount-accumulates, with the next row's LDS reads already
// in flight, reaches 1.27e12 row compares/s chip-wide against 1.07e12 for
// xor/bcnt/xor/bcnt, which is what the compiler schedules.  Tying compiler-
// allocated scalars to the 4-register tuples of ds_read_b128 costs copies and
// early waits, so the row buffers are fixed registers: A = v[30:45],
// B = v[46:61], v62 = distance, v63 = LDS address.
//
// Row r of the tile has key index key0 + r; the running top-2 keys are updated
// in train-index order.  The read of "row + 2" at the end of the tile may touch
// up to two rows past `rows`: the tile is allocated two rows longer for that.
// ---------------------------------------------------------------------------
#define SLAMHIP_LD_ROW(v0, v1, v2, v3, o0, o1, o2, o3)                                          \
  "ds_read_b128 v[" #v0 ":" #v0 "+3], v63 offset:" #o0 "\n"                                     \
  "ds_read_b128 v[" #v1 ":" #v1 "+3], v63 offset:" #o1 "\n"                                     \
  "ds_read_b128 v[" #v2 ":" #v2 "+3], v63 offset:" #o2 "\n"                                     \
  "ds_read_b128 v[" #v3 ":" #v3 "+3], v63 offset:" #o3 "\n"
#define SLAMHIP_LD_A(off0, off1, off2, off3) SLAMHIP_LD_ROW(30, 34, 38, 42, off0, off1, off2, off3)
#define SLAMHIP_LD_B(off0, off1, off2, off3) SLAMHIP_LD_ROW(46, 50, 54, 58, off0, off1, off2, off3)
#define SLAMHIP_XR(r, qi) "v_xor_b32 v" #r ", %" #qi ", v" #r "\n"
#define SLAMHIP_BR(r) "v_bcnt_u32_b32 v62, v" #r ", v62\n"
// operands %5..%20 are the query words
#define SLAMHIP_DIST_A                                                                          \
  SLAMHIP_XR(30, 5) SLAMHIP_XR(31, 6) SLAMHIP_XR(32, 7) SLAMHIP_XR(33, 8) SLAMHIP_XR(34, 9)      \
  SLAMHIP_XR(35, 10) SLAMHIP_XR(36, 11) SLAMHIP_XR(37, 12) SLAMHIP_XR(38, 13) SLAMHIP_XR(39, 14) \
  SLAMHIP_XR(40, 15) SLAMHIP_XR(41, 16) SLAMHIP_XR(42, 17) SLAMHIP_XR(43, 18) SLAMHIP_XR(44, 19) \
  SLAMHIP_XR(45, 20)                                                                            \
  "v_bcnt_u32_b32 v62, v30, 0\n"                                                                \
  SLAMHIP_BR(31) SLAMHIP_BR(32) SLAMHIP_BR(33) SLAMHIP_BR(34) SLAMHIP_BR(35) SLAMHIP_BR(36)     \
  SLAMHIP_BR(37) SLAMHIP_BR(38) SLAMHIP_BR(39) SLAMHIP_BR(40) SLAMHIP_BR(41) SLAMHIP_BR(42)     \
  SLAMHIP_BR(43) SLAMHIP_BR(44) SLAMHIP_BR(45)
#define SLAMHIP_DIST_B                                                                          \
  SLAMHIP_XR(46, 5) SLAMHIP_XR(47, 6) SLAMHIP_XR(48, 7) SLAMHIP_XR(49, 8) SLAMHIP_XR(50, 9)      \
  SLAMHIP_XR(51, 10) SLAMHIP_XR(52, 11) SLAMHIP_XR(53, 12) SLAMHIP_XR(54, 13) SLAMHIP_XR(55, 14) \
  SLAMHIP_XR(56, 15) SLAMHIP_XR(57, 16) SLAMHIP_XR(58, 17) SLAMHIP_XR(59, 18) SLAMHIP_XR(60, 19) \
  SLAMHIP_XR(61, 20)                                                                            \
  "v_bcnt_u32_b32 v62, v46, 0\n"                                                                \
  SLAMHIP_BR(47) SLAMHIP_BR(48) SLAMHIP_BR(49) SLAMHIP_BR(50) SLAMHIP_BR(51) SLAMHIP_BR(52)     \
  SLAMHIP_BR(53) SLAMHIP_BR(54) SLAMHIP_BR(55) SLAMHIP_BR(56) SLAMHIP_BR(57) SLAMHIP_BR(58)     \
  SLAMHIP_BR(59) SLAMHIP_BR(60) SLAMHIP_BR(61)
// key = dist << 21 | index; b2 = med3(b1, b2, key); b1 = min(b1, key)   (%0 = b1, %1 = b2, %3 = index)
#define SLAMHIP_PUSH_R(r)                                                                       \
  "v_lshl_or_b32 v" #r ", v" #r ", 21, %3\n"                                                    \
  "v_med3_u32 %1, %0, %1, v" #r "\n"                                                            \
  "v_min_u32 %0, %0, v" #r "\n"
#define SLAMHIP_PUSH SLAMHIP_PUSH_R(62)
// best match only (k = 1 callers): no second-best update
#define SLAMHIP_PUSH1_R(r)                                                                      \
  "v_lshl_or_b32 v" #r ", v" #r ", 21, %3\n"                                                    \
  "v_min_u32 %0, %0, v" #r "\n"
#define SLAMHIP_PUSH1 SLAMHIP_PUSH1_R(62)
#define SLAMHIP_SCAN_CLOBBERS                                                                   \
  "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42",     \
  "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55",     \
  "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "scc", "memory"

#define SLAMHIP_DEFINE_SCAN16(NAME, PUSHM) \
__device__ __forceinline__ void NAME(uint32_t lds_addr, int rows, uint32_t key0, \
                                            const uint32_t (&q)[16], uint32_t &b1, uint32_t &b2) { \
  asm volatile( \
      "v_mov_b32 v63, %4\n" \
      SLAMHIP_LD_A(0, 16, 32, 48) \
      "s_cmp_lt_u32 %2, 2\n" \
      "s_cbranch_scc1 SCAN_TAIL_%=\n" \
      "SCAN_LOOP_%=:\n" \
      SLAMHIP_LD_B(64, 80, 96, 112) \
      "s_waitcnt lgkmcnt(4)\n" \
      SLAMHIP_DIST_A \
      PUSHM \
      "v_add_u32 v63, 0x80, v63\n" \
      SLAMHIP_LD_A(0, 16, 32, 48) \
      "s_add_u32 %3, %3, 1\n" \
      "s_waitcnt lgkmcnt(4)\n" \
      SLAMHIP_DIST_B \
      PUSHM \
      "s_add_u32 %3, %3, 1\n" \
      "s_sub_u32 %2, %2, 2\n" \
      "s_cmp_lt_u32 %2, 2\n" \
      "s_cbranch_scc0 SCAN_LOOP_%=\n" \
      "SCAN_TAIL_%=:\n" \
      "s_waitcnt lgkmcnt(0)\n" \
      "s_cmp_eq_u32 %2, 0\n" \
      "s_cbranch_scc1 SCAN_DONE_%=\n" \
      SLAMHIP_DIST_A \
      PUSHM \
      "SCAN_DONE_%=:\n" \
      : "+v"(b1), "+v"(b2), "+s"(rows), "+s"(key0) \
      : "v"(lds_addr), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), \
        "v"(q[7]), "v"(q[8]), "v"(q[9]), "v"(q[10]), "v"(q[11]), "v"(q[12]), "v"(q[13]), "v"(q[14]), \
        "v"(q[15]) \
      : SLAMHIP_SCAN_CLOBBERS); \
}
SLAMHIP_DEFINE_SCAN16(scan_tile16, SLAMHIP_PUSH)
SLAMHIP_DEFINE_SCAN16(scan_tile16_best, SLAMHIP_PUSH1)

// The same for 8-word rows (descriptors of at most 32 bytes): a buffer holds two
// consecutive rows, compared as 16 xors and two popcount chains (the second one
// accumulates in its own first word), four rows per loop trip.  Operands %5..%12 are
// the query words.  The reads past the last row stay within two rows.
#define SLAMHIP_XR8(base)                                                                       \
  SLAMHIP_XR_(base, 0, 5) SLAMHIP_XR_(base, 1, 6) SLAMHIP_XR_(base, 2, 7) SLAMHIP_XR_(base, 3, 8) \
  SLAMHIP_XR_(base, 4, 9) SLAMHIP_XR_(base, 5, 10) SLAMHIP_XR_(base, 6, 11) SLAMHIP_XR_(base, 7, 12)
#define SLAMHIP_XR_(base, i, qi) "v_xor_b32 v[" #base "+" #i "], %" #qi ", v[" #base "+" #i "]\n"
#define SLAMHIP_BC_(acc, base, i) "v_bcnt_u32_b32 v" #acc ", v[" #base "+" #i "], v" #acc "\n"
#define SLAMHIP_CHAIN8(acc, base)                                                               \
  "v_bcnt_u32_b32 v" #acc ", v[" #base "+0], 0\n"                                               \
  SLAMHIP_BC_(acc, base, 1) SLAMHIP_BC_(acc, base, 2) SLAMHIP_BC_(acc, base, 3)                 \
  SLAMHIP_BC_(acc, base, 4) SLAMHIP_BC_(acc, base, 5) SLAMHIP_BC_(acc, base, 6) SLAMHIP_BC_(acc, base, 7)
// two rows of a buffer starting at register `lo` (second row at `hi` = lo + 8)
#define SLAMHIP_DIST8X2(lo, hi, P)                                                              \
  SLAMHIP_XR8(lo) SLAMHIP_XR8(hi) SLAMHIP_CHAIN8(62, lo) SLAMHIP_CHAIN8(hi, hi)                 \
  P(62) "s_add_u32 %3, %3, 1\n" P(hi) "s_add_u32 %3, %3, 1\n"
#define SLAMHIP_DIST8X1(lo, P)                                                                  \
  SLAMHIP_XR8(lo) SLAMHIP_CHAIN8(62, lo) P(62) "s_add_u32 %3, %3, 1\n"

#define SLAMHIP_DEFINE_SCAN8(NAME, P) \
__device__ __forceinline__ void NAME(uint32_t lds_addr, int rows, uint32_t key0, \
                                           const uint32_t (&q)[8], uint32_t &b1, uint32_t &b2) { \
  asm volatile( \
      "v_mov_b32 v63, %4\n" \
      SLAMHIP_LD_A(0, 16, 32, 48) \
      "s_cmp_lt_u32 %2, 4\n" \
      "s_cbranch_scc1 SCAN_TAIL_%=\n" \
      "SCAN_LOOP_%=:\n" \
      SLAMHIP_LD_B(64, 80, 96, 112) \
      "s_waitcnt lgkmcnt(4)\n" \
      SLAMHIP_DIST8X2(30, 38, P) \
      "v_add_u32 v63, 0x80, v63\n" \
      SLAMHIP_LD_A(0, 16, 32, 48) \
      "s_waitcnt lgkmcnt(4)\n" \
      SLAMHIP_DIST8X2(46, 54, P) \
      "s_sub_u32 %2, %2, 4\n" \
      "s_cmp_lt_u32 %2, 4\n" \
      "s_cbranch_scc0 SCAN_LOOP_%=\n" \
      "SCAN_TAIL_%=:\n" \
      "s_waitcnt lgkmcnt(0)\n" \
      "s_cmp_eq_u32 %2, 0\n" \
      "s_cbranch_scc1 SCAN_DONE_%=\n" \
      SLAMHIP_DIST8X1(30, P) \
      "s_cmp_eq_u32 %2, 1\n" \
      "s_cbranch_scc1 SCAN_DONE_%=\n" \
      SLAMHIP_DIST8X1(38, P) \
      "s_cmp_eq_u32 %2, 2\n" \
      "s_cbranch_scc1 SCAN_DONE_%=\n" \
      "ds_read_b128 v[46:49], v63 offset:64\n" \
      "ds_read_b128 v[50:53], v63 offset:80\n" \
      "s_waitcnt lgkmcnt(0)\n" \
      SLAMHIP_DIST8X1(46, P) \
      "SCAN_DONE_%=:\n" \
      : "+v"(b1), "+v"(b2), "+s"(rows), "+s"(key0) \
      : "v"(lds_addr), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), \
        "v"(q[7]) \
      : SLAMHIP_SCAN_CLOBBERS); \
}
SLAMHIP_DEFINE_SCAN8(scan_tile8, SLAMHIP_PUSH_R)
SLAMHIP_DEFINE_SCAN8(scan_tile8_best, SLAMHIP_PUSH1_R)

// ---------------------------------------------------------------------------
// Row packing: (n, desc_bytes) rows `stride` bytes apart -> (n, W) dwords,
// zero padded.  Handles both descriptor sets of a call in one launch.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_rows_kernel(
    const uint8_t *__restrict__ src_a, int n_a, uint32_t *__restrict__ dst_a,
    const uint8_t *__restrict__ src_b, int n_b, uint32_t *__restrict__ dst_b, int desc_bytes,
    int stride, int log2W, unsigned *__restrict__ zero = nullptr, int n_zero = 0) {
  // one thread per four packed dwords (one 16-byte store); W is a power of two >= 8,
  // so row / word group come from shifts
  const int lq = log2W - 2;
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  // the matcher's arrival counters start every call at zero
  for (unsigned i = gid; i < (unsigned)n_zero; i += gridDim.x * blockDim.x) zero[i] = 0u;
  const unsigned total_a = (unsigned)n_a << lq;
  const unsigned total = total_a + ((unsigned)n_b << lq);
  if (gid >= total) return;
  const bool first = gid < total_a;
  const uint8_t *src = first ? src_a : src_b;
  uint32_t *dst = first ? dst_a : dst_b;
  const unsigned g = first ? gid : gid - total_a;
  const unsigned r = g >> lq;
  const int w0 = 4 * (int)(g & ((1u << lq) - 1u));
  // Rows of 61 bytes start at any byte offset: fetch the aligned dwords that hold the
  // wanted 16 bytes and funnel-shift.  A dword is fetched only if it holds a byte of
  // the row, so no access can fall on a page the row does not touch.
  const int remaining = desc_bytes - 4 * w0;          // bytes of the row from this group on
  const uintptr_t a = (uintptr_t)(src + (size_t)r * (size_t)stride + 4 * w0);
  const unsigned sh = (unsigned)(a & 3u);
  const uint32_t *pa = reinterpret_cast<const uint32_t *>(a - sh);
  uint32_t d[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) d[k] = (4 * k < remaining + (int)sh) ? pa[k] : 0u;
  uint32_t v[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const uint32_t x = __builtin_amdgcn_alignbyte(d[w + 1], d[w], sh);
    const int valid = remaining - 4 * w;
    v[w] = valid >= 4 ? x : (valid > 0 ? (x & ((1u << (8 * valid)) - 1u)) : 0u);
  }
  reinterpret_cast<uint4 *>(dst)[g] = make_uint4(v[0], v[1], v[2], v[3]);
}

// ---------------------------------------------------------------------------
// Main kernel.
// ---------------------------------------------------------------------------
// BEST: only the best match per query is wanted (k = 1 callers); the hand-scheduled
// scans then skip the second-best update and the second key stays "none".
template <int W, int Q, bool USE_LDS, int T = kThreads, bool BEST = false>
__global__ __launch_bounds__(T) void hamming_top2_kernel(
    const uint32_t *__restrict__ qp, const uint32_t *__restrict__ tp,
    const int32_t *__restrict__ q_off, const int32_t *__restrict__ q_cnt,
    const int32_t *__restrict__ t_off, const int32_t *__restrict__ t_cnt,
    const int32_t *__restrict__ out_off, int nq_single, int nt_single, int nq_total,
    uint2 *__restrict__ partial, int32_t *__restrict__ idx2, int32_t *__restrict__ dist2) {
  const int pair = blockIdx.z;
  int q0 = 0, nq = nq_single, t0 = 0, nt = nt_single;
  if (q_off != nullptr) {
    q0 = q_off[pair];
    nq = q_cnt ? q_cnt[pair] : q_off[pair + 1] - q0;
    t0 = t_off[pair];
    nt = t_cnt ? t_cnt[pair] : t_off[pair + 1] - t0;
  }
  // results of query row r of this pair go to output row o0 + r (default: the
  // query's own global row); nq_total is the number of output rows
  const int o0 = out_off ? out_off[pair] : q0;
  const int qblk = blockIdx.x * (T * Q);
  if (qblk >= nq) return;  // block-uniform

  const int n_splits = gridDim.y;
  const int split = blockIdx.y;
  const int rps = (nt + n_splits - 1) / n_splits;
  const int r_begin = min(nt, split * rps);
  const int r_end = min(nt, r_begin + rps);

  // Queries -> registers.  Lanes past the end re-read the last row (their
  // results are never stored).
  uint32_t qv[Q][W];
  int qrow[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    qrow[q] = qblk + q * T + (int)threadIdx.x;
    const int r = min(qrow[q], nq - 1);
    const uint4 *src = reinterpret_cast<const uint4 *>(qp + (size_t)(q0 + r) * W);
#pragma unroll
    for (int w4 = 0; w4 < W / 4; ++w4) {
      const uint4 v = src[w4];
      qv[q][4 * w4 + 0] = v.x;
      qv[q][4 * w4 + 1] = v.y;
      qv[q][4 * w4 + 2] = v.z;
      qv[q][4 * w4 + 3] = v.w;
    }
  }

  uint32_t b1[Q], b2[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) b1[q] = b2[q] = kNoKey;

  if constexpr (USE_LDS) {
    __shared__ uint4 tile[(kTileRows + 2) * W / 4];   // + 2: see scan_tile16
    for (int tile_begin = r_begin; tile_begin < r_end; tile_begin += kTileRows) {
      const int rows = min(kTileRows, r_end - tile_begin);
      __syncthreads();
      const uint4 *src = reinterpret_cast<const uint4 *>(tp + (size_t)(t0 + tile_begin) * W);
      for (int i = threadIdx.x; i < rows * (W / 4); i += T) tile[i] = src[i];
      __syncthreads();
      // wave-uniform trip count kept in an SGPR so the loop control stays off the VALU
      const int rows_u = __builtin_amdgcn_readfirstlane(rows);
      if constexpr (Q == 1 && W <= 16) {
        const uint32_t lds_addr =
            (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)tile;
        const uint32_t key0 = (uint32_t)__builtin_amdgcn_readfirstlane(tile_begin);
        if constexpr (W == 16 && BEST) scan_tile16_best(lds_addr, rows_u, key0, qv[0], b1[0], b2[0]);
        else if constexpr (W == 16) scan_tile16(lds_addr, rows_u, key0, qv[0], b1[0], b2[0]);
        else if constexpr (BEST) scan_tile8_best(lds_addr, rows_u, key0, qv[0], b1[0], b2[0]);
        else scan_tile8(lds_addr, rows_u, key0, qv[0], b1[0], b2[0]);
      } else {
      int jj = 0;
      for (; jj + 2 <= rows_u; jj += 2) {
        uint32_t ta[W], tb[W];
#pragma unroll
        for (int w4 = 0; w4 < W / 4; ++w4) {
          const uint4 v = tile[jj * (W / 4) + w4];
          ta[4 * w4 + 0] = v.x; ta[4 * w4 + 1] = v.y; ta[4 * w4 + 2] = v.z; ta[4 * w4 + 3] = v.w;
        }
#pragma unroll
        for (int w4 = 0; w4 < W / 4; ++w4) {
          const uint4 v = tile[(jj + 1) * (W / 4) + w4];
          tb[4 * w4 + 0] = v.x; tb[4 * w4 + 1] = v.y; tb[4 * w4 + 2] = v.z; tb[4 * w4 + 3] = v.w;
        }
        const uint32_t jidx = (uint32_t)(tile_begin + jj);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          uint32_t acc = 0, acc2 = 0;
#pragma unroll
          for (int w = 0; w < W; ++w) {
            acc += __builtin_popcount(ta[w] ^ qv[q][w]);
            acc2 += __builtin_popcount(tb[w] ^ qv[q][w]);
          }
          top2_push(b1[q], b2[q], (acc << kKeyShift) | jidx);
          top2_push(b1[q], b2[q], (acc2 << kKeyShift) | (jidx + 1u));
        }
      }
      if (jj < rows_u) {
        uint32_t tw[W];
#pragma unroll
        for (int w4 = 0; w4 < W / 4; ++w4) {
          const uint4 v = tile[jj * (W / 4) + w4];
          tw[4 * w4 + 0] = v.x; tw[4 * w4 + 1] = v.y; tw[4 * w4 + 2] = v.z; tw[4 * w4 + 3] = v.w;
        }
        const uint32_t jidx = (uint32_t)(tile_begin + jj);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          uint32_t acc = 0;
#pragma unroll
          for (int w = 0; w < W; ++w) acc += __builtin_popcount(tw[w] ^ qv[q][w]);
          top2_push(b1[q], b2[q], (acc << kKeyShift) | jidx);
        }
      }
      }
    }
  } else {
    // Wave-uniform train row: the compiler turns these loads into s_load_dwordx*
    // and feeds the SGPRs straight into v_xor_b32.
    // The next row is fetched into a second SGPR set before the current one is
    // consumed, so the scalar-load latency overlaps the 2*W*Q VALU ops of a row.
    if (r_begin < r_end) {
      const uint32_t *trow = tp + (size_t)(t0 + r_begin) * W;
      const uint32_t *tlast = tp + (size_t)(t0 + r_end - 1) * W;
      uint32_t cur[W];
#pragma unroll
      for (int w = 0; w < W; ++w) cur[w] = trow[w];
      for (int j = r_begin; j < r_end; ++j) {
        trow = (trow == tlast) ? trow : trow + W;
        uint32_t nxt[W];
#pragma unroll
        for (int w = 0; w < W; ++w) nxt[w] = trow[w];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          uint32_t acc = 0;
#pragma unroll
          for (int w = 0; w < W; ++w) acc += __builtin_popcount(cur[w] ^ qv[q][w]);
          top2_push(b1[q], b2[q], (acc << kKeyShift) | (uint32_t)j);
        }
#pragma unroll
        for (int w = 0; w < W; ++w) cur[w] = nxt[w];
      }
    }
  }

#pragma unroll
  for (int q = 0; q < Q; ++q) {
    if (qrow[q] >= nq) continue;
    const size_t row = (size_t)(o0 + qrow[q]);
    if (n_splits == 1) {
      idx2[2 * row + 0] = b1[q] == kNoKey ? -1 : (int32_t)(b1[q] & kIdxMask);
      idx2[2 * row + 1] = b2[q] == kNoKey ? -1 : (int32_t)(b2[q] & kIdxMask);
      dist2[2 * row + 0] = b1[q] == kNoKey ? -1 : (int32_t)(b1[q] >> kKeyShift);
      dist2[2 * row + 1] = b2[q] == kNoKey ? -1 : (int32_t)(b2[q] >> kKeyShift);
    } else {
      partial[(size_t)split * nq_total + row] = make_uint2(b1[q], b2[q]);
    }
  }
}

// Folds the per-split key pairs.  Block = ROWS query rows x GROUPS split groups
// (ROWS * GROUPS = 256): lane (row, g) folds splits g, g+GROUPS, ... (consecutive
// rows: coalesced 8-byte entries), the groups meet in LDS and lane (row, 0) folds
// them.  The keys are unique, so the order of the fold does not matter.  Many splits
// (a single large pair: 64 splits of 8192 rows) take 16 x 16 - the fold is latency-
// bound and 64-row blocks left half of the CUs without work; few splits over many
// rows (a batch of frames) take 64 x 4 so that no group idles.  Optionally applies
// the ratio test to the final pair.
template <int ROWS, int GROUPS>
__global__ __launch_bounds__(256) void hamming_merge_kernel(const uint2 *__restrict__ partial,
                                                            int nq_total, int n_splits,
                                                            int32_t *__restrict__ idx2,
                                                            int32_t *__restrict__ dist2,
                                                            double ratio,
                                                            uint8_t *__restrict__ keep) {
  static_assert(ROWS * GROUPS == 256, "one thread per (row, group)");
  __shared__ uint2 part[GROUPS][ROWS];
  const int lane = threadIdx.x % ROWS, g = threadIdx.x / ROWS;
  const int row = blockIdx.x * ROWS + lane;
  uint32_t b1 = kNoKey, b2 = kNoKey;
  if (row < nq_total) {
    for (int s = g; s < n_splits; s += GROUPS) {
      const uint2 p = partial[(size_t)s * nq_total + row];
      top2_push(b1, b2, p.x);
      top2_push(b1, b2, p.y);
    }
  }
  part[g][lane] = make_uint2(b1, b2);
  __syncthreads();
  if (g != 0 || row >= nq_total) return;
#pragma unroll
  for (int k = 1; k < GROUPS; ++k) {
    const uint2 p = part[k][lane];
    top2_push(b1, b2, p.x);
    top2_push(b1, b2, p.y);
  }
  int2 oi, od;
  oi.x = b1 == kNoKey ? -1 : (int32_t)(b1 & kIdxMask);
  oi.y = b2 == kNoKey ? -1 : (int32_t)(b2 & kIdxMask);
  od.x = b1 == kNoKey ? -1 : (int32_t)(b1 >> kKeyShift);
  od.y = b2 == kNoKey ? -1 : (int32_t)(b2 >> kKeyShift);
  reinterpret_cast<int2 *>(idx2)[row] = oi;
  reinterpret_cast<int2 *>(dist2)[row] = od;
  if (keep != nullptr)
    keep[row] = (od.x >= 0 && od.y >= 0 && (double)od.x < ratio * (double)od.y) ? 1 : 0;
}

// The same fold with one thread per query row, for a split count known at compile time: the row's
// S entries are asked for at once (consecutive rows: coalesced 8-byte loads), no LDS, no barrier,
// a quarter of the workgroups.  Used where S is 2, 4, 8, 16 or 32; on 8192 x 8192 (16 splits)
// 0.25-0.5 us per call ahead of the kernel above, whose lanes take one entry each.
template <int S>
__global__ __launch_bounds__(64) void hamming_merge_rows_kernel(const uint2 *__restrict__ partial, int nq_total,
                                                                 int32_t *__restrict__ idx2,
                                                                 int32_t *__restrict__ dist2, double ratio,
                                                                 uint8_t *__restrict__ keep) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= nq_total) return;
  uint2 p[S];
#pragma unroll
  for (int s = 0; s < S; ++s) p[s] = partial[(size_t)s * nq_total + row];
  uint32_t b1 = kNoKey, b2 = kNoKey;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    top2_push(b1, b2, p[s].x);
    top2_push(b1, b2, p[s].y);
  }
  int2 oi, od;
  oi.x = b1 == kNoKey ? -1 : (int32_t)(b1 & kIdxMask);
  oi.y = b2 == kNoKey ? -1 : (int32_t)(b2 & kIdxMask);
  od.x = b1 == kNoKey ? -1 : (int32_t)(b1 >> kKeyShift);
  od.y = b2 == kNoKey ? -1 : (int32_t)(b2 >> kKeyShift);
  reinterpret_cast<int2 *>(idx2)[row] = oi;
  reinterpret_cast<int2 *>(dist2)[row] = od;
  if (keep != nullptr)
    keep[row] = (od.x >= 0 && od.y >= 0 && (double)od.x < ratio * (double)od.y) ? 1 : 0;
}

__global__ __launch_bounds__(256) void ratio_test_kernel(const int32_t *__restrict__ dist2, int nq,
                                                         double ratio,
                                                         uint8_t *__restrict__ keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const int2 d = reinterpret_cast<const int2 *>(dist2)[i];
  keep[i] = (d.x >= 0 && d.y >= 0 && (double)d.x < ratio * (double)d.y) ? 1 : 0;
}

// crossCheck=True fold (single pair).  Every train row bids (distance, its own index)
// for the query row it chose with one atomic minimum on a key per query row: a query
// keeps the closest of the train rows that chose it, lowest train index on ties (the
// fold of cv2's batchDistance with crossCheck) in O(nt + nq).  The keys live in
// match_idx until they are decoded in place.
__global__ __launch_bounds__(256) void cross_bid_one_kernel(const int32_t *__restrict__ t2q_idx2,
                                                            const int32_t *__restrict__ t2q_dist2,
                                                            int nt, int nq, uint32_t *__restrict__ bid) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nt) return;
  const int who = t2q_idx2[2 * (size_t)t];
  if (who < 0 || who >= nq) return;
  const uint32_t d = (uint32_t)t2q_dist2[2 * (size_t)t];
  atomicMin(&bid[who], (d << kKeyShift) | (uint32_t)t);
}

__global__ __launch_bounds__(256) void cross_decode_one_kernel(int nq, int32_t *__restrict__ match_idx,
                                                               int32_t *__restrict__ match_dist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const uint32_t b = (uint32_t)match_idx[i];
  match_idx[i] = b == kNoKey ? -1 : (int32_t)(b & kIdxMask);
  match_dist[i] = b == kNoKey ? -1 : (int32_t)(b >> kKeyShift);
}

// ---------------------------------------------------------------------------
// Stable order by (distance, query index) for segments of at most
// kBitonicMax rows: one block sorts the segment's keys (dist << 21 | index,
// unique, so an unstable network gives the stable order) in LDS with a bitonic
// network and writes the index part.  O(n log^2 n) against the O(n^2) of rank
// counting below, which remains for longer segments.
// ---------------------------------------------------------------------------
constexpr int kBitonicMax = 8192;

// T threads per block: 256 when there are segments enough to fill the chip with such blocks,
// 1024 when there are few (a block's time is what the call takes then: 16 waves per segment
// instead of 4).
template <int T>
__global__ __launch_bounds__(T) void bitonic_sort_kernel(const int32_t *__restrict__ dist,
                                                         int dist_stride, int n_single,
                                                         const int32_t *__restrict__ off,
                                                         const int32_t *__restrict__ cnt,
                                                         int32_t *__restrict__ order, int lds_keys) {
  extern __shared__ uint32_t skeys[];
  const int seg = blockIdx.x;
  int s0 = 0, n = n_single;
  if (off != nullptr) {
    s0 = off[seg];
    n = cnt ? cnt[seg] : off[seg + 1] - s0;
  }
  // block-uniform; a segment longer than the caller's stated maximum does not fit the
  // LDS that was sized from it and is left alone rather than sorted out of bounds
  if (n <= 0 || n > lds_keys) return;
  int N = 1;
  while (N < n) N <<= 1;
  for (int i = threadIdx.x; i < N; i += T) {
    uint32_t key = kNoKey;                  // padding sorts last
    if (i < n) {
      const int d = dist[(size_t)(s0 + i) * dist_stride];
      key = ((d < 0 ? 2047u : (uint32_t)d) << kKeyShift) | (uint32_t)i;
    }
    skeys[i] = key;
  }
  __syncthreads();
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (N >> 1); t += T) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // t with a 0 inserted at bit log2(j)
        const int hi = lo | j;
        const uint32_t a = skeys[lo], b = skeys[hi];
        const bool ascending = (lo & k) == 0;
        if ((a > b) == ascending) {
          skeys[lo] = b;
          skeys[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int r = threadIdx.x; r < n; r += T) order[s0 + r] = (int32_t)(skeys[r] & kIdxMask);
}

// ---------------------------------------------------------------------------
// The same order by a stable least-significant-digit radix sort in LDS (sort_device.h): one
// block per segment, CAP = keys the block's LDS holds.
// ---------------------------------------------------------------------------
template <int T, int CAP>
__global__ __launch_bounds__(T) void radix_sort_kernel(const int32_t *__restrict__ dist, int dist_stride,
                                                       int n_single, const int32_t *__restrict__ off,
                                                       const int32_t *__restrict__ cnt,
                                                       int32_t *__restrict__ order) {
  __shared__ slamhip_dev::RadixSortLds<T, CAP> lds;
  const int seg = blockIdx.x;
  int s0 = 0, n = n_single;
  if (off != nullptr) {
    s0 = off[seg];
    n = cnt ? cnt[seg] : off[seg + 1] - s0;
  }
  if (n <= 0 || n > CAP) return;   // block-uniform; longer segments are not this kernel's (launch_sort)
  for (int i = threadIdx.x; i < n; i += T)
    lds.keys[0][i] = slamhip_dev::sort_key(dist[(size_t)(s0 + i) * dist_stride], i);
  slamhip_dev::radix_sort_lds<T, CAP>(lds, n);
  for (int r = threadIdx.x; r < n; r += T) order[s0 + r] = (int32_t)(lds.keys[0][r] & kIdxMask);
}

// ---------------------------------------------------------------------------
// Stable order by (distance, query index) through rank counting: the keys
// (dist << 21 | index) are unique, so rank(i) = #{j : key_j < key_i} and the
// sorted sequence is order[rank(i)] = i.  2 VALU ops per (i, j) pair, all j
// wave-uniform.  One block covers 256 i's of one segment and all j of that
// segment.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rank_sort_kernel(const int32_t *__restrict__ dist,
                                                        int dist_stride, int n_single,
                                                        const int32_t *__restrict__ off,
                                                        const int32_t *__restrict__ cnt,
                                                        int32_t *__restrict__ order) {
  __shared__ __align__(16) uint32_t keys[1024];
  const int seg = blockIdx.y;
  int s0 = 0, n = n_single;
  if (off != nullptr) {
    s0 = off[seg];
    n = cnt ? cnt[seg] : off[seg + 1] - s0;
  }
  const int i0 = blockIdx.x * 256;
  if (i0 >= n) return;
  const int i = i0 + (int)threadIdx.x;
  uint32_t my = kNoKey;
  if (i < n) {
    const int d = dist[(size_t)(s0 + i) * dist_stride];
    my = ((d < 0 ? 2047u : (uint32_t)d) << kKeyShift) | (uint32_t)i;
  }
  int rank = 0;
  for (int j0 = 0; j0 < n; j0 += 1024) {
    __syncthreads();
    for (int k = threadIdx.x; k < 1024; k += 256) {
      const int j = j0 + k;
      uint32_t key = kNoKey;
      if (j < n) {
        const int d = dist[(size_t)(s0 + j) * dist_stride];
        key = ((d < 0 ? 2047u : (uint32_t)d) << kKeyShift) | (uint32_t)j;
      }
      keys[k] = key;
    }
    __syncthreads();
    const int lim = min(1024, n - j0);
    const int lim4 = lim & ~3;
    int k = 0;
    for (; k < lim4; k += 4) {
      const uint4 v = *reinterpret_cast<const uint4 *>(&keys[k]);
      rank += (v.x < my) + (v.y < my) + (v.z < my) + (v.w < my);
    }
    for (; k < lim; ++k) rank += keys[k] < my;
  }
  if (i < n) order[s0 + rank] = i;
}

constexpr int kRowsPerSplit = 384;   // train rows per block the plan aims at when there are blocks enough

struct MatchPlan {
  int W, Q, qblocks, splits, threads;
  bool use_lds;
  bool mfma;   // matrix-core kernel (hamming_mfma.hip); G = groups of 32 queries per wave,
  int G, NW;   // NW = waves per block
  size_t off_qp, off_tp, off_part, off_cnt, total;
  bool fold_in_kernel;   // matrix-core kernel folds its splits itself (arrival counters at off_cnt)
  bool raw;              // matrix-core kernel reads the caller's rows as they are: no packing launch
};

// Optional per-launch timing of the dominant (top-2) kernel with HIP events
// recorded on the launch stream; used by the benchmark to get the kernel's
// average duration inside its timed region.
struct ProfEvents {
  bool enabled = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
  size_t used = 0;
};
ProfEvents g_prof;

// diagnostics: -1 auto; 0..7 popcount kernel variants (bit0 = LDS, bits 1.. = Q);
// 8 = matrix-core kernel, 9 / 10 = the same with 1 / 2 query groups per wave
int g_force_variant = -1;

// Kernel family when nothing is forced: the environment variable SLAMHIP_MATCHER
// ("popcount" or "mfma") read once, else the library default.
bool default_mfma() {
  static const int v = [] {
    const char *e = getenv("SLAMHIP_MATCHER");
    if (e && e[0] == 'p') return 0;
    if (e && e[0] == 'm') return 1;
    return SLAMHIP_DEFAULT_MFMA;
  }();
  return v != 0;
}
int g_force_splits = 0;
int g_mfma_target_blocks = 512;    // matrix-core kernel: blocks a launch aims at (knob 6)
int g_mfma_rows_per_split = 4096;  // and train rows per block it does not exceed when blocks abound (knob 7)
int g_prof_repeat = 1;     // knob 10: launches of the top-2 kernel per probed call; 0 = one launch timed by its own dispatch (see run_knn2)
int g_raw_rows = 1;        // knob 9: matrix-core kernel reads unpacked rows itself (0 = packing launch first)
int g_fold_in_kernel = 2;  // knob 8: 0 = always the separate merge launch, 1 = never, 2 = by size
int g_force_threads = 0;   // 0 auto, else 256 or 128 threads per matcher block
int g_force_sort = 0;      // 0 auto (radix passes where the segment fits LDS), 1 rank counting, 2 bitonic network

MatchPlan make_plan(int nq_total, int nt_total, int n_pairs, int max_nq, int max_nt,
                    int desc_bytes, long long n_out = -1, bool prepacked = false,
                    int pinned_splits = 0) {
  if (n_out < 0) n_out = nq_total;
  MatchPlan p;
  p.W = slamhip_packed_words(desc_bytes);
  // Measured on MI355X (8192x8192x61 B): LDS-staged tile, one query per lane is
  // the fastest of the four variants; the SGPR-fed variant exposes scalar-load
  // latency, two queries per lane cost occupancy.
  p.use_lds = true;
  p.Q = 1;
  p.mfma = default_mfma();
  p.G = 0;
  p.NW = 4;
  if (g_force_variant >= 16) {          // 18: eight waves per block, two query groups per wave
    p.mfma = true;
    p.G = g_force_variant - 16;
    p.NW = 8;
  } else if (g_force_variant >= 8) {
    p.mfma = true;
    p.G = g_force_variant - 8;
  } else if (g_force_variant >= 0) {
    p.mfma = false;
    p.use_lds = (g_force_variant & 1) != 0;
    p.Q = (g_force_variant >> 1) == 2 && p.W <= 16 ? 2 : 1;
  }
  // the matrix-core kernel indexes train rows inside a split with SH bits
  const int mfma_rows = slamhip_internal::mfma_max_split_rows(p.W);
  if (p.mfma && (long long)max_nt > (long long)kMaxSplits * mfma_rows) p.mfma = false;
  if (p.mfma) {
    // 4 waves x G groups of 32 queries per block; two groups unless one fills the last
    // block of a pair much better, or the registers do not allow more (32-word rows)
    int G = p.G;
    const bool four = G == 4 && p.W == 16;   // forced only: four groups per wave, one wave per SIMD
    if (G != 1 && G != 2 && !four) {
      G = 2;
      const int idle2 = slamhip_ceil_div(max_nq > 0 ? max_nq : 1, 256) * 256 - max_nq;
      const int idle1 = slamhip_ceil_div(max_nq > 0 ? max_nq : 1, 128) * 128 - max_nq;
      if (idle2 - idle1 >= 128 && idle2 * 8 > max_nq) G = 1;
    }
    if (p.W == 32) G = 1;   // the query fragments of two groups of 32-word rows do not fit the registers
    // One block of eight waves per CU (its waves share every staged train row and the two waves
    // of a SIMD advance together) where one pair alone gives every CU such a block within 16
    // splits: 8192 x 8192 x 61 B 19.0 us per call against 19.6; batches of pairs, whose blocks
    // differ in length, run faster with two independent 4-wave blocks per CU (60 frames of
    // 2000-8000 features: 0.90 ms against 1.11).
    if (g_force_variant < 0 && n_pairs == 1 && G == 2 && (long long)slamhip_ceil_div(max_nq, 512) * 16 >= 256)
      p.NW = 8;
    if (p.NW == 8 && (G != 2 || p.W == 32)) p.NW = 4;
    p.G = G;
    p.threads = 64 * p.NW;
    p.qblocks = slamhip_ceil_div(max_nq > 0 ? max_nq : 1, 32 * p.NW * G);
    const long long base_blocks = (long long)p.qblocks * (n_pairs > 0 ? n_pairs : 1);
    // Two blocks per CU when the queries alone do not give them, at most 16 splits and at
    // least one 128-row stage per split (measured, one n x n pair of 61-byte rows: n = 2048
    // 16 splits, 4096 16, 8192 16, 16384 8 are the fastest or within 2 % of it); and no
    // block longer than 4096 train rows (measured on batches, whose query blocks alone fill the
    // chip: 500 frames of 1500 features 0.82 ms per pass with at most 1024 rows per block,
    // 0.79 with 2048 or more: no split, no merge launch; 60 frames of 2000-8000 features
    // 0.91 / 0.87 / 0.86 / 0.89 ms with 1024 / 2048 / 4096 / 8192).
    const int target_blocks = p.NW == 8 ? g_mfma_target_blocks / 2 : g_mfma_target_blocks;   // one block per CU
    long long s = (target_blocks + base_blocks - 1) / base_blocks;
    if (s > 16) s = 16;
    const int by_rows = slamhip_ceil_div(max_nt > 0 ? max_nt : 1, 128);
    if (s > by_rows) s = by_rows;
    const long long by_grain = slamhip_ceil_div(max_nt > 0 ? max_nt : 1, g_mfma_rows_per_split);
    if (s < by_grain) s = by_grain;
    if (pinned_splits > 0) s = pinned_splits;
    if (g_force_splits > 0) s = g_force_splits;
    const long long need = slamhip_ceil_div(max_nt > 0 ? max_nt : 1, mfma_rows);
    if (s < need) s = need;
    if (s > kMaxSplits) s = kMaxSplits;
    if (s < 1) s = 1;
    p.splits = (int)s;
  } else {
  // Half-size blocks when that fills the last block of a pair much better (e.g. 610
  // queries: 3 x 256 leaves 158 lanes idle, 5 x 128 leaves 30).  Default variant only.
  p.threads = kThreads;
  if (p.use_lds && p.Q == 1 && max_nq > 0) {
    const int idle_full = slamhip_ceil_div(max_nq, kThreads) * kThreads - max_nq;
    const int idle_half = slamhip_ceil_div(max_nq, kThreads / 2) * (kThreads / 2) - max_nq;
    if (idle_full - idle_half >= kThreads / 2 && idle_full * 8 > max_nq) p.threads = kThreads / 2;
  }
  if (g_force_threads == kThreads || g_force_threads == kThreads / 2)
    p.threads = (p.use_lds && p.Q == 1) ? g_force_threads : kThreads;
  p.qblocks = slamhip_ceil_div(max_nq > 0 ? max_nq : 1, p.threads * p.Q);
  const long long base_blocks = (long long)p.qblocks * (n_pairs > 0 ? n_pairs : 1);
  const long long target = 2048;  // 8 blocks of 4 waves per CU
  long long s = (target + base_blocks - 1) / base_blocks;
  // and blocks short enough that the last ones of a launch do not leave the chip idle
  // for long, which matters most when the pairs differ in size: a few hundred train
  // rows per block (measured: 60 frames of 2000-8000 features 2.9 -> 2.3 ms, 500 frames
  // of 1500 features 1.87 -> 1.80 ms)
  const long long by_grain = slamhip_ceil_div(max_nt > 0 ? max_nt : 1, kRowsPerSplit);
  if (s < by_grain) s = by_grain;
  const int by_rows = slamhip_ceil_div(max_nt > 0 ? max_nt : 1, 32);
  if (s > by_rows) s = by_rows;
  if (s > kMaxSplits) s = kMaxSplits;
  if (s < 1) s = 1;
  if (pinned_splits > 0) s = pinned_splits > kMaxSplits ? kMaxSplits : pinned_splits;
  if (g_force_splits > 0) s = g_force_splits > kMaxSplits ? kMaxSplits : g_force_splits;
  p.splits = (int)s;
  }
  p.raw = p.mfma && !prepacked && g_raw_rows != 0 && desc_bytes >= 16;   // see below
  const bool packs = !prepacked && !p.raw;
  size_t o = 0;
  p.off_qp = o;
  if (packs) o = slamhip_align_up(o + (size_t)nq_total * p.W * 4, 256);
  p.off_tp = o;
  if (packs) o = slamhip_align_up(o + (size_t)nt_total * p.W * 4, 256);
  p.off_part = o;
  if (p.splits > 1) o = slamhip_align_up(o + (size_t)p.splits * (size_t)n_out * sizeof(uint2), 256);
  // The matrix-core kernel folds its splits inside the launch when the call has a row
  // packing launch in front of it (which clears the arrival counters): one launch less.
  // Measured from a C++ loop (8192 x 8192 x 61 B, 16 splits): 20.2 us per call folded in the
  // kernel against 19.5 with the merge launch (the fold is three dependent memory round trips
  // at the kernel's tail, the merge launch overlaps with the next call's packing), but 9.6
  // against 11.2 at 2048 x 2048, where the call is bound by its launches: small calls fold
  // in the kernel (knob 8: 1 = always, 0 = never, 2 = by size).
  const long long compares = (long long)(max_nq > 0 ? max_nq : 1) * (max_nt > 0 ? max_nt : 1) * (n_pairs > 0 ? n_pairs : 1);
  // The matrix-core kernel reads unpacked rows itself (unaligned 16-byte loads that stay inside
  // the row, bytes of the neighbouring chunk masked) unless a row is shorter than one such load.  There is then no packing launch to
  // clear the arrival counters, and a fill launch in its place costs what the merge launch
  // costs (2048 x 2048: 7.9 us per call with the merge launch, 10.0 folded behind a fill), so
  // the fold in the kernel is left to calls that pack (knob 9 = 0) or ask for it (knob 8 = 1).
  const bool fold_wanted = g_fold_in_kernel == 1 || (g_fold_in_kernel == 2 && !p.raw && compares <= (8ll << 20));
  p.fold_in_kernel = p.mfma && p.splits > 1 && !prepacked && fold_wanted;
  p.off_cnt = o;
  if (p.fold_in_kernel)
    o = slamhip_align_up(o + (size_t)p.qblocks * (size_t)(n_pairs > 0 ? n_pairs : 1) * sizeof(unsigned), 256);
  p.total = o + 256;
  return p;
}

template <int W, int Q, bool L, int T = kThreads, bool BEST = false>
void launch_top2(dim3 grid, hipStream_t st, const uint32_t *qp, const uint32_t *tp,
                 const int32_t *q_off, const int32_t *q_cnt, const int32_t *t_off,
                 const int32_t *t_cnt, const int32_t *out_off, int nq, int nt, int nq_total,
                 uint2 *part, int32_t *idx2, int32_t *dist2) {
  hipLaunchKernelGGL((hamming_top2_kernel<W, Q, L, T, BEST>), grid, dim3(T), 0, st, qp, tp, q_off, q_cnt,
                     t_off, t_cnt, out_off, nq, nt, nq_total, part, idx2, dist2);
}

int run_knn2(const uint8_t *d_query, int nq_total, const uint8_t *d_train, int nt_total,
             int desc_bytes, int row_stride, const int32_t *d_q_off, const int32_t *d_t_off,
             int n_pairs, int max_nq, int max_nt, int32_t *d_idx2, int32_t *d_dist2,
             void *d_workspace, size_t workspace_bytes, void *stream, bool prepacked = false,
             const int32_t *d_q_cnt = nullptr, const int32_t *d_t_cnt = nullptr,
             uint8_t *d_keep = nullptr, double ratio = 0.0, const int32_t *d_out_off = nullptr,
             int n_out_total = -1, bool best_only = false, int pinned_splits = 0,
             slamhip_internal::Knn2Splits *unfolded = nullptr) {
  if (n_out_total < 0) n_out_total = nq_total;
  if (unfolded != nullptr) { unfolded->part = nullptr; unfolded->splits = 1; unfolded->rows = n_out_total; }
  if (nq_total < 0 || nt_total < 0 || n_pairs < 1 || max_nq < 0 || max_nt < 0)
    return SLAMHIP_EINVAL;
  if (desc_bytes < 1 || row_stride < desc_bytes) return SLAMHIP_EINVAL;
  if (desc_bytes > SLAMHIP_MAX_DESC_BYTES || max_nt > SLAMHIP_MAX_TRAIN_ROWS ||
      max_nq > SLAMHIP_MAX_TRAIN_ROWS)
    return SLAMHIP_EUNSUPPORTED;
  if (n_pairs > 65535) return SLAMHIP_EUNSUPPORTED;
  if (nq_total == 0) return SLAMHIP_OK;
  if (!d_query || !d_idx2 || !d_dist2 || (nt_total > 0 && !d_train)) return SLAMHIP_EINVAL;
  if (n_pairs > 1 && (!d_q_off || !d_t_off)) return SLAMHIP_EINVAL;
  if (max_nq == 0) return SLAMHIP_OK;
  const MatchPlan p =
      make_plan(nq_total, nt_total, n_pairs, max_nq, max_nt, desc_bytes, n_out_total, prepacked,
                pinned_splits);
  if (!d_workspace || workspace_bytes < p.total) return SLAMHIP_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;

  // 256-byte align the workspace base ourselves.
  uintptr_t base = ((uintptr_t)d_workspace + 255) & ~(uintptr_t)255;
  uint32_t *qp = (uint32_t *)(base + p.off_qp);
  uint32_t *tp = (uint32_t *)(base + p.off_tp);
  uint2 *part = (uint2 *)(base + p.off_part);
  unsigned *arrivals = (unsigned *)(base + p.off_cnt);

  // The matrix-core kernel reads unpacked rows itself (unaligned 16-byte loads inside the row):
  // no packing launch.  Rows shorter than one 16-byte load keep the packing path.
  slamhip_internal::RawRows raw_rows;
  const bool raw = p.raw;
  if (raw) {
    raw_rows.row_stride = row_stride;
    raw_rows.desc_bytes = desc_bytes;
    qp = (uint32_t *)d_query;
    tp = (uint32_t *)d_train;
    if (p.fold_in_kernel &&
        hipMemsetAsync(arrivals, 0, (size_t)p.qblocks * n_pairs * sizeof(unsigned), st) != hipSuccess)
      return SLAMHIP_ELAUNCH;
  } else if (prepacked) {
    // rows are already W dwords each, 16-byte aligned
    if (row_stride != p.W * 4 || ((uintptr_t)d_query & 15) || ((uintptr_t)d_train & 15))
      return SLAMHIP_EINVAL;
    qp = (uint32_t *)d_query;
    tp = (uint32_t *)d_train;
  } else {
    const long long total = ((long long)nq_total + nt_total) * (p.W / 4);
    if (total >= (1ll << 31)) return SLAMHIP_EUNSUPPORTED;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(pack_rows_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, d_query, nq_total, qp,
                       d_train, nt_total, tp, desc_bytes, row_stride, p.W == 8 ? 3 : (p.W == 16 ? 4 : 5),
                       p.fold_in_kernel ? arrivals : nullptr, p.fold_in_kernel ? p.qblocks * n_pairs : 0);
    SLAMHIP_CHECK_LAUNCH();
  }

  const dim3 grid(p.qblocks, p.splits, n_pairs);
  std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
  // knob 10 = 0: the two events are handed to the launch itself and hold the dispatch's own start
  // and end (matrix-core kernel only; the popcount family keeps the records around the launch)
  const bool by_dispatch = g_prof_repeat == 0 && p.mfma;
  if (g_prof.enabled && g_prof.used < g_prof.pool.size()) {
    ev = &g_prof.pool[g_prof.used++];
    if (by_dispatch) {
      slamhip_internal::g_dispatch_start = ev->first;
      slamhip_internal::g_dispatch_stop = ev->second;
    } else {
      (void)hipEventRecord(ev->first, st);
    }
  }
  const int nqs = d_q_off ? 0 : nq_total, nts = d_q_off ? 0 : nt_total;
#define SLAMHIP_DISPATCH(WW, QQ, LL)                                                        \
  launch_top2<WW, QQ, LL>(grid, st, qp, tp, d_q_off, d_q_cnt, d_t_off, d_t_cnt, d_out_off, nqs, nts, \
                          n_out_total, part, d_idx2, d_dist2)
#define SLAMHIP_DISPATCH_HALF(WW)                                                                  \
  launch_top2<WW, 1, true, kThreads / 2>(grid, st, qp, tp, d_q_off, d_q_cnt, d_t_off, d_t_cnt,     \
                                         d_out_off, nqs, nts, n_out_total, part, d_idx2, d_dist2)
#define SLAMHIP_DISPATCH_BEST(WW, TT)                                                              \
  launch_top2<WW, 1, true, TT, true>(grid, st, qp, tp, d_q_off, d_q_cnt, d_t_off, d_t_cnt, d_out_off, \
                                     nqs, nts, n_out_total, part, d_idx2, d_dist2)
  // Under the duration probe the top-2 kernel may be launched several times back to back between
  // the two events (knob 10; the launches are identical and idempotent), so that the events'
  // own cost is spread over them.
  // (knob 10 < 0: that many launches in every call, no events: for timing the step with one more
  // launch of the kernel in it)
  const int launches = p.fold_in_kernel ? 1
                       : (g_prof_repeat < 0 ? -g_prof_repeat : (ev != nullptr && g_prof_repeat > 1 ? g_prof_repeat : 1));
  for (int rep = 0; rep < launches; ++rep) {
  if (p.mfma) {
    slamhip_internal::launch_hamming_mfma(p.W, p.G, p.NW, best_only, grid, st, qp, tp, d_q_off, d_q_cnt, d_t_off,
                                          d_t_cnt, d_out_off, nqs, nts, n_out_total, part, d_idx2, d_dist2,
                                          p.fold_in_kernel ? arrivals : nullptr, ratio, d_keep,
                                          raw ? &raw_rows : nullptr);
  } else if (best_only && p.use_lds && p.Q == 1 && p.W <= 16) {
    if (p.threads == kThreads / 2) { if (p.W == 8) SLAMHIP_DISPATCH_BEST(8, kThreads / 2); else SLAMHIP_DISPATCH_BEST(16, kThreads / 2); }
    else                           { if (p.W == 8) SLAMHIP_DISPATCH_BEST(8, kThreads); else SLAMHIP_DISPATCH_BEST(16, kThreads); }
  } else if (p.threads == kThreads / 2) {
    if (p.W == 8) SLAMHIP_DISPATCH_HALF(8);
    else if (p.W == 16) SLAMHIP_DISPATCH_HALF(16);
    else SLAMHIP_DISPATCH_HALF(32);
  } else if (p.W == 8) {
    if (p.Q == 2) { if (p.use_lds) SLAMHIP_DISPATCH(8, 2, true); else SLAMHIP_DISPATCH(8, 2, false); }
    else          { if (p.use_lds) SLAMHIP_DISPATCH(8, 1, true); else SLAMHIP_DISPATCH(8, 1, false); }
  } else if (p.W == 16) {
    if (p.Q == 2) { if (p.use_lds) SLAMHIP_DISPATCH(16, 2, true); else SLAMHIP_DISPATCH(16, 2, false); }
    else          { if (p.use_lds) SLAMHIP_DISPATCH(16, 1, true); else SLAMHIP_DISPATCH(16, 1, false); }
  } else {
    if (p.use_lds) SLAMHIP_DISPATCH(32, 1, true); else SLAMHIP_DISPATCH(32, 1, false);
  }
  }
#undef SLAMHIP_DISPATCH
#undef SLAMHIP_DISPATCH_HALF
#undef SLAMHIP_DISPATCH_BEST
  if (ev && by_dispatch) slamhip_internal::g_dispatch_start = slamhip_internal::g_dispatch_stop = nullptr;
  else if (ev) (void)hipEventRecord(ev->second, st);
  SLAMHIP_CHECK_LAUNCH();

  if (p.fold_in_kernel) {
    // folded and ratio-tested by the kernel itself
  } else if (p.splits > 1 && unfolded != nullptr && d_keep == nullptr) {
    unfolded->part = part;     // the caller's next kernel folds the splits
    unfolded->splits = p.splits;
  } else if (p.splits > 1) {
    // under the dispatch probe the merge launch carries events of its own too (never read): every
    // dispatch of the probed steps then has a profiling signal, as under a kernel trace
    static hipEvent_t m0 = nullptr, m1 = nullptr;
    const bool signalled = ev != nullptr && by_dispatch;
    if (signalled && m0 == nullptr && (hipEventCreate(&m0) != hipSuccess || hipEventCreate(&m1) != hipSuccess))
      return SLAMHIP_ELAUNCH;
    // one wavefront per workgroup: measured on ten GPUs of one pool, 0.25-0.5 us per 8192 x 8192 call
    // ahead of the 16-lanes-per-row kernel on every one of them; 256-thread workgroups were another
    // 0.1-0.2 us ahead on six and 0.1-0.7 us BEHIND that kernel on four
    constexpr int row_bs = 64;
    const dim3 row_grid(slamhip_ceil_div(n_out_total, row_bs));
#define SLAMHIP_MERGE_ROWS(SS)                                                                              \
  if (signalled)                                                                                            \
    hipExtLaunchKernelGGL(hamming_merge_rows_kernel<SS>, row_grid, dim3(row_bs), 0, st, m0, m1, 0, part,       \
                          n_out_total, d_idx2, d_dist2, ratio, d_keep);                                     \
  else                                                                                                      \
    hipLaunchKernelGGL(hamming_merge_rows_kernel<SS>, row_grid, dim3(row_bs), 0, st, part, n_out_total, d_idx2, \
                       d_dist2, ratio, d_keep)
    if (p.splits == 2) { SLAMHIP_MERGE_ROWS(2); }
    else if (p.splits == 4) { SLAMHIP_MERGE_ROWS(4); }
    else if (p.splits == 8) { SLAMHIP_MERGE_ROWS(8); }
    else if (p.splits == 16) { SLAMHIP_MERGE_ROWS(16); }
    else if (p.splits == 32) { SLAMHIP_MERGE_ROWS(32); }
    else if (p.splits >= 16)
      hipLaunchKernelGGL((hamming_merge_kernel<16, 16>), dim3(slamhip_ceil_div(n_out_total, 16)), dim3(256),
                         0, st, part, n_out_total, p.splits, d_idx2, d_dist2, ratio, d_keep);
    else
      hipLaunchKernelGGL((hamming_merge_kernel<64, 4>), dim3(slamhip_ceil_div(n_out_total, 64)), dim3(256),
                         0, st, part, n_out_total, p.splits, d_idx2, d_dist2, ratio, d_keep);
#undef SLAMHIP_MERGE_ROWS
    SLAMHIP_CHECK_LAUNCH();
  } else if (d_keep != nullptr) {
    hipLaunchKernelGGL(ratio_test_kernel, dim3(slamhip_ceil_div(n_out_total, 256)), dim3(256), 0, st,
                       d_dist2, n_out_total, ratio, d_keep);
    SLAMHIP_CHECK_LAUNCH();
  }
  return SLAMHIP_OK;
}

// Stable sort of every segment by (distance, index): radix passes in LDS (one block per
// segment; 1024 threads when the segments are few and long) when the segments fit, rank
// counting (many blocks per segment) for longer ones; knob 5 = 2 keeps the bitonic network.
void launch_sort(const int32_t *d_dist, int dist_stride, int n_single, const int32_t *d_off,
                 const int32_t *d_cnt, int n_segments, int max_n, int32_t *d_order, hipStream_t st) {
  // Measured: 500 segments of 1500 rows 16 us by the radix passes, 40 by the network, 60 by rank
  // counting (700 rows: 9 / 20); two frames of 8000 features, the whole front-end call: 0.21 ms
  // (radix) against 0.27 (network); 60 segments of 2000-8000 rows: the front-end pass 0.91 ms
  // with the 1024-thread network against 1.03 ms with rank counting.  The radix kernel takes
  // every segment that fits its LDS unless knob 5 says otherwise.
  bool bitonic = max_n <= kBitonicMax;   // the segment fits one block's LDS (either LDS kernel)
  if (g_force_sort == 1) bitonic = false;
  if (bitonic && g_force_sort != 2) {
    if (max_n <= 2048) {
      hipLaunchKernelGGL((radix_sort_kernel<256, 2048>), dim3(n_segments), dim3(256), 0, st, d_dist, dist_stride,
                         n_single, d_off, d_cnt, d_order);
    } else if (n_segments >= 128) {
      hipLaunchKernelGGL((radix_sort_kernel<256, kBitonicMax>), dim3(n_segments), dim3(256), 0, st, d_dist,
                         dist_stride, n_single, d_off, d_cnt, d_order);
    } else {
      hipLaunchKernelGGL((radix_sort_kernel<1024, kBitonicMax>), dim3(n_segments), dim3(1024), 0, st, d_dist,
                         dist_stride, n_single, d_off, d_cnt, d_order);
    }
  } else if (bitonic) {
    int N = 1;
    while (N < max_n) N <<= 1;
    if (n_segments >= 128 || N <= 1024)
      hipLaunchKernelGGL(bitonic_sort_kernel<256>, dim3(n_segments), dim3(256), (size_t)N * sizeof(uint32_t),
                         st, d_dist, dist_stride, n_single, d_off, d_cnt, d_order, N);
    else
      hipLaunchKernelGGL(bitonic_sort_kernel<1024>, dim3(n_segments), dim3(1024), (size_t)N * sizeof(uint32_t),
                         st, d_dist, dist_stride, n_single, d_off, d_cnt, d_order, N);
  } else {
    const dim3 grid(slamhip_ceil_div(max_n, 256), n_segments);
    hipLaunchKernelGGL(rank_sort_kernel, grid, dim3(256), 0, st, d_dist, dist_stride, n_single, d_off,
                       d_cnt, d_order);
  }
}

}  // namespace

extern "C" {

// Diagnostics knob used by the kernel-variant benchmark: knob 0 = kernel
// variant (-1 auto; bit0 LDS-staged train tile, value>>1 = queries per lane),
// knob 1 = forced number of train splits (0 auto), knob 2 = packed-fp32
// screening in the consensus counting kernels (1 on, 0 fp64 only; same counts),
// knob 3 = matcher block size (0 auto, 256, 128), knob 4 = pair RANSAC launch scheme
// (0 auto, 1 always rounds; set it before sizing the workspace), knob 5 = sort
// (0 auto, 1 rank counting, 2 bitonic network where the segments fit it).
int slamhip_tune(int knob, int value) {
  if (knob == 0) g_force_variant = value;
  else if (knob == 1) g_force_splits = value;
  else if (knob == 2) slamhip_internal::g_ransac_screen = (value >= 0 && value <= 3) ? value : 1;
  else if (knob == 3) g_force_threads = value;
  else if (knob == 4) slamhip_internal::g_ransac_mode = value;
  else if (knob == 5) g_force_sort = value;
  else if (knob == 6) g_mfma_target_blocks = value > 0 ? value : 512;
  else if (knob == 7) g_mfma_rows_per_split = value > 0 ? value : 4096;
  else if (knob == 9) g_raw_rows = value ? 1 : 0;
  else if (knob == 11) slamhip_internal::g_xcd_remap = value ? 1 : 0;
  else if (knob == 13) slamhip_internal::g_wide_rounds = (value >= 1 && value <= 3) ? value : 2;
  else if (knob == 14) slamhip_internal::g_small_call_fusion = value ? 1 : 0;
  else if (knob == 15) slamhip_internal::g_small_call_span = value > 0 ? (value + 1) & ~1 : 0;
  else if (knob == 16) slamhip_internal::g_count_split = value ? 1 : 0;
  else if (knob == 10) g_prof_repeat = value >= -8 && value <= 64 ? value : 1;
  else if (knob == 8) g_fold_in_kernel = (value >= 0 && value <= 2) ? value : 2;
  else return SLAMHIP_EINVAL;
  return SLAMHIP_OK;
}

int slamhip_matcher_kind(void) { return default_mfma() ? 1 : 0; }

// Kernel-duration probe for benchmarks.  begin(max_launches): from now on the
// matcher brackets its top-2 kernel with HIP events on the launch stream (at
// most max_launches of them).  end(): synchronises those events, returns the
// number of bracketed launches and their summed duration in milliseconds, and
// switches the probe off.
int slamhip_profile_begin(int max_launches) {
  if (max_launches < 1 || max_launches > (1 << 20)) return SLAMHIP_EINVAL;
  while (g_prof.pool.size() < (size_t)max_launches) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return SLAMHIP_ELAUNCH;
    g_prof.pool.emplace_back(a, b);
  }
  g_prof.used = 0;
  g_prof.enabled = true;
  return SLAMHIP_OK;
}

int slamhip_profile_end(double *total_ms) {
  g_prof.enabled = false;
  double sum = 0.0;
  for (size_t i = 0; i < g_prof.used; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(g_prof.pool[i].second) != hipSuccess) return SLAMHIP_ELAUNCH;
    if (hipEventElapsedTime(&ms, g_prof.pool[i].first, g_prof.pool[i].second) != hipSuccess)
      return SLAMHIP_ELAUNCH;
    sum += ms;
  }
  if (total_ms) *total_ms = sum;
  const int n = (int)g_prof.used;
  g_prof.used = 0;
  return n;
}

size_t slamhip_hamming_knn2_workspace_bytes(int nq_total, int nt_total, int n_pairs, int max_nq,
                                            int max_nt, int desc_bytes) {
  if (nq_total < 0 || nt_total < 0 || n_pairs < 1 || desc_bytes < 1 ||
      desc_bytes > SLAMHIP_MAX_DESC_BYTES)
    return 0;
  return make_plan(nq_total, nt_total, n_pairs, max_nq, max_nt, desc_bytes).total;
}

int slamhip_hamming_knn2(const uint8_t *d_query, int nq, const uint8_t *d_train, int nt,
                         int desc_bytes, int row_stride, int32_t *d_idx2, int32_t *d_dist2,
                         void *d_workspace, size_t workspace_bytes, void *stream) {
  SLAMHIP_BEGIN();
  return run_knn2(d_query, nq, d_train, nt, desc_bytes, row_stride, nullptr, nullptr, 1, nq, nt,
                  d_idx2, d_dist2, d_workspace, workspace_bytes, stream);
}

int slamhip_hamming_knn2_ratio(const uint8_t *d_query, int nq, const uint8_t *d_train, int nt,
                               int desc_bytes, int row_stride, double ratio, int32_t *d_idx2,
                               int32_t *d_dist2, uint8_t *d_keep, void *d_workspace,
                               size_t workspace_bytes, void *stream) {
  SLAMHIP_BEGIN();
  if (!d_keep && nq > 0) return SLAMHIP_EINVAL;
  return run_knn2(d_query, nq, d_train, nt, desc_bytes, row_stride, nullptr, nullptr, 1, nq, nt,
                  d_idx2, d_dist2, d_workspace, workspace_bytes, stream, false, nullptr, nullptr,
                  d_keep, ratio);
}

int slamhip_hamming_knn2_batched(const uint8_t *d_query, int nq_total, const uint8_t *d_train,
                                 int nt_total, int desc_bytes, int row_stride,
                                 const int32_t *d_q_off, const int32_t *d_t_off, int n_pairs,
                                 int max_nq, int max_nt, int32_t *d_idx2, int32_t *d_dist2,
                                 void *d_workspace, size_t workspace_bytes, void *stream) {
  SLAMHIP_BEGIN();
  if (!d_q_off || !d_t_off) return SLAMHIP_EINVAL;
  return run_knn2(d_query, nq_total, d_train, nt_total, desc_bytes, row_stride, d_q_off, d_t_off,
                  n_pairs, max_nq, max_nt, d_idx2, d_dist2, d_workspace, workspace_bytes, stream);
}

int slamhip_ratio_test(const int32_t *d_dist2, int nq, double ratio, uint8_t *d_keep,
                       void *stream) {
  SLAMHIP_BEGIN();
  if (nq < 0) return SLAMHIP_EINVAL;
  if (nq == 0) return SLAMHIP_OK;
  if (!d_dist2 || !d_keep) return SLAMHIP_EINVAL;
  hipLaunchKernelGGL(ratio_test_kernel, dim3(slamhip_ceil_div(nq, 256)), dim3(256), 0,
                     (hipStream_t)stream, d_dist2, nq, ratio, d_keep);
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}

int slamhip_cross_check(const int32_t *d_t2q_idx2, const int32_t *d_t2q_dist2, int nt, int nq,
                        int32_t *d_match_idx, int32_t *d_match_dist, void *stream) {
  SLAMHIP_BEGIN();
  if (nt < 0 || nq < 0) return SLAMHIP_EINVAL;
  if (nq == 0) return SLAMHIP_OK;
  if (!d_match_idx || !d_match_dist || (nt > 0 && (!d_t2q_idx2 || !d_t2q_dist2)))
    return SLAMHIP_EINVAL;
  if (nt > SLAMHIP_MAX_TRAIN_ROWS) return SLAMHIP_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(d_match_idx, 0xFF, (size_t)nq * sizeof(int32_t), st) != hipSuccess) return SLAMHIP_ELAUNCH;
  if (nt > 0) {
    hipLaunchKernelGGL(cross_bid_one_kernel, dim3(slamhip_ceil_div(nt, 256)), dim3(256), 0, st, d_t2q_idx2,
                       d_t2q_dist2, nt, nq, (uint32_t *)d_match_idx);
    SLAMHIP_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(cross_decode_one_kernel, dim3(slamhip_ceil_div(nq, 256)), dim3(256), 0, st, nq,
                     d_match_idx, d_match_dist);
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}

int slamhip_sort_by_distance(const int32_t *d_dist, int dist_stride, int n_total,
                             const int32_t *d_off, int n_segments, int max_n, int32_t *d_order,
                             void *stream) {
  SLAMHIP_BEGIN();
  if (n_total < 0 || dist_stride < 1 || n_segments < 1 || max_n < 0) return SLAMHIP_EINVAL;
  if (max_n > SLAMHIP_MAX_TRAIN_ROWS || n_segments > 65535) return SLAMHIP_EUNSUPPORTED;
  if (n_total == 0 || max_n == 0) return SLAMHIP_OK;
  if (!d_dist || !d_order) return SLAMHIP_EINVAL;
  if (n_segments > 1 && !d_off) return SLAMHIP_EINVAL;
  launch_sort(d_dist, dist_stride, d_off ? 0 : n_total, d_off, nullptr, n_segments, max_n, d_order,
              (hipStream_t)stream);
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}

// ---------------------------------------------------------------------------
// The k=2 + ratio-test call for callers that cannot allocate: no workspace
// argument, the matcher's scratch (packed rows, per-split partial results) lives
// in a file-scope device array.  Every GPU the library runs on holds its own copy
// of the array at its own address; the address is looked up once per device, with
// that device current, and only that pointer is ever handed to kernels (the
// array's name is a host-side stand-in, not a device address).
// ---------------------------------------------------------------------------
}  // extern "C"

namespace {
constexpr size_t kStaticScratchBytes = (size_t)SLAMHIP_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

constexpr int kMaxDevices = 64;
std::atomic<void *> g_scratch_ptr[kMaxDevices];   // per device; nullptr until looked up
std::mutex g_scratch_mutex;

// Device address of this device's copy of `scratch` (nullptr on error).
void *static_scratch_for_current_device() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  void *p = g_scratch_ptr[dev].load(std::memory_order_acquire);
  if (p != nullptr) return p;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  p = g_scratch_ptr[dev].load(std::memory_order_relaxed);
  if (p != nullptr) return p;
  void *addr = nullptr;
  if (hipGetSymbolAddress(&addr, HIP_SYMBOL(scratch)) != hipSuccess || addr == nullptr) {
    (void)hipGetLastError();
    return nullptr;
  }
  g_scratch_ptr[dev].store(addr, std::memory_order_release);
  return addr;
}
}  // namespace

extern "C" {

int slamhip_hamming_knn2_ratio_static(const uint8_t *d_query, int nq, const uint8_t *d_train, int nt,
                                      int desc_bytes, int row_stride, double ratio, int32_t *d_idx2,
                                      int32_t *d_dist2, uint8_t *d_keep, void *stream) {
  SLAMHIP_BEGIN();
  if (nq < 0 || nt < 0) return SLAMHIP_EINVAL;
  if (!d_keep && nq > 0) return SLAMHIP_EINVAL;
  if (nq > SLAMHIP_STATIC_MAX_ROWS || nt > SLAMHIP_STATIC_MAX_ROWS) return SLAMHIP_EUNSUPPORTED;
  if (desc_bytes < 1) return SLAMHIP_EINVAL;
  if (desc_bytes > SLAMHIP_MAX_DESC_BYTES) return SLAMHIP_EUNSUPPORTED;   // as the workspace call answers
  // the plan of any call within the row limit fits the array (checked again inside run_knn2)
  if (make_plan(nq, nt, 1, nq, nt, desc_bytes).total > kStaticScratchBytes) return SLAMHIP_EUNSUPPORTED;
  void *ws = static_scratch_for_current_device();
  if (ws == nullptr) return SLAMHIP_ELAUNCH;
  // Nothing in the scratch has to start at zero: the packing kernel writes every packed
  // row the matcher reads and the matcher writes every (split, row) entry the fold reads.
  return run_knn2(d_query, nq, d_train, nt, desc_bytes, row_stride, nullptr, nullptr, 1, nq, nt,
                  d_idx2, d_dist2, ws, kStaticScratchBytes, stream, false, nullptr, nullptr, d_keep,
                  ratio);
}

}  // extern "C"

// Entry points for the other translation units of the library.
namespace slamhip_internal {

size_t knn2_workspace_bytes(int nq_total, int nt_total, int n_pairs, int max_nq, int max_nt,
                            int desc_bytes, long long n_out, bool prepacked, int splits) {
  return make_plan(nq_total, nt_total, n_pairs, max_nq, max_nt, desc_bytes, n_out, prepacked, splits).total;
}

bool knn2_reads_raw(int nq_total, int nt_total, int n_pairs, int max_nq, int max_nt, int desc_bytes) {
  return make_plan(nq_total, nt_total, n_pairs, max_nq, max_nt, desc_bytes, 0, false).raw;
}

int sort_choice() { return g_force_sort; }

int knn2_plan_splits(int n_pairs, int max_nq, int max_nt, int desc_bytes) {
  return make_plan(0, 0, n_pairs, max_nq, max_nt, desc_bytes, 0, true).splits;
}

int knn2(const uint8_t *d_query, int nq_total, const uint8_t *d_train, int nt_total, int desc_bytes,
         int row_stride, bool prepacked, const int32_t *d_q_off, const int32_t *d_q_cnt,
         const int32_t *d_t_off, const int32_t *d_t_cnt, int n_pairs, int max_nq, int max_nt,
         int32_t *d_idx2, int32_t *d_dist2, void *d_workspace, size_t workspace_bytes,
         hipStream_t st, const int32_t *d_out_off, int n_out_total, bool best_only, int splits,
         Knn2Splits *unfolded) {
  return run_knn2(d_query, nq_total, d_train, nt_total, desc_bytes, row_stride, d_q_off, d_t_off,
                  n_pairs, max_nq, max_nt, d_idx2, d_dist2, d_workspace, workspace_bytes, (void *)st,
                  prepacked, d_q_cnt, d_t_cnt, nullptr, 0.0, d_out_off, n_out_total, best_only, splits,
                  unfolded);
}

int pack_rows(const uint8_t *d_a, int n_a, uint32_t *d_dst_a, const uint8_t *d_b, int n_b,
              uint32_t *d_dst_b, int desc_bytes, int row_stride, hipStream_t st) {
  const int W = slamhip_packed_words(desc_bytes);
  const long long total = ((long long)n_a + n_b) * (W / 4);   // threads: one per four dwords
  if (total == 0) return SLAMHIP_OK;
  if (total >= (1ll << 31)) return SLAMHIP_EUNSUPPORTED;
  hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_a,
                     n_a, d_dst_a, d_b, n_b, d_dst_b, desc_bytes, row_stride,
                     W == 8 ? 3 : (W == 16 ? 4 : 5));
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}

int sort_by_distance(const int32_t *d_dist, int dist_stride, const int32_t *d_off,
                     const int32_t *d_cnt, int n_segments, int max_n, int32_t *d_order,
                     hipStream_t st) {
  if (max_n <= 0 || n_segments <= 0) return SLAMHIP_OK;
  launch_sort(d_dist, dist_stride, 0, d_off, d_cnt, n_segments, max_n, d_order, st);
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}

}  // namespace slamhip_internal
