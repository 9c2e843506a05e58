// Hamming top-2 matcher on the matrix cores of gfx950 (MI355X).
//
// Over D = 32 W bits (rows are zero padded to W dwords, padding bits are equal on
// both sides), with every bit written as the FP4 value +1 (bit 0) or -1 (bit 1):
//     <q', t'> = D - 2 * hamming(q, t)
// so a 32 x 32 tile of distances is a 32 x 32 x D contraction, which
// v_mfma_scale_f32_32x32x64_f8f6f4 does 64 bit pairs per instruction.  The
// accumulator is preloaded and the train operand scaled (E8M0 block scale, a power
// of two, exact) so that the instruction's result IS the matcher's sort key:
//     C_in  = 2^(SH-1) * D + (MASK - local_row),      scale = 2^(SH-1)
//     D_out = C_in + 2^(SH-1) * (D - 2 d) = (D - d) << SH | (MASK - local_row)
// an integer below 2^24, exact in float32.  Larger key = smaller distance, then lower
// train row: the order of a sequential scan with strict `<` insertion
// (cv2.BFMatcher / batchDistance, the tie rule behind
// frontend/vision/descriptor_matcher.py:33 of the reference).  What is left for
// the vector ALU per (query, train row) is the running top-2 update (v_med3 / v_max3, 20
// instructions per 16 keys).
//
// Work decomposition
//   block  = NW waves (4, or 8 for one large pair); wave = G groups of 32 queries whose
//            expanded rows stay in registers (B operand) for the whole scan
//   grid   = (query blocks, train splits, pairs), as the popcount kernel
//   queries = the block's query rows are copied into LDS as they lie in memory (consecutive
//            lanes, consecutive 16 bytes) and each lane picks its row's chunks from there; the
//            fragments of the first k-step are expanded in front of the first tile, the others
//            inside it (it has no earlier tile to fold)
//   stage  = R train rows: loaded 16 bytes per thread, either from rows packed to W dwords or
//            from the caller's rows as they are (RAW: unaligned loads that stay inside the row,
//            bytes of the neighbouring chunk masked off), expanded to FP4 on the fly (one dword -> four dwords: bit plane p of
//            the dword, (x << (3 - p)) & 0x88888888 | 0x22222222), written to LDS in a
//            16-byte-slot XOR swizzle that makes the A-fragment ds_read_b128 of 32 rows
//            conflict-free; two buffers, one barrier per stage.  A table of the stage's C_in
//            values sits beside it (rows past the end hold a large negative number, so they
//            can never win).
//   loop   = software pipeline: every k-step issues its matrix instructions, re-fills the A
//            fragment it has just used with the next tile's, folds a slice of the previous
//            tile's keys into the running top-2 and expands one piece of the next stage.
//   tile   = 32 train rows x 32 queries: KS = W / 2 matrix instructions per group.
// Accumulator layout (32 x 32): lane l holds query column l & 31 and train rows
// (i & 3) + 8 (i >> 2) + 4 (l >> 5), i = 0..15; lanes l and l ^ 32 meet at the end.
// Both operands are fetched the same way (lane l: row l & 31, 16-byte chunk l >> 5 of the
// k-step), so whatever order the instruction gives the 64 k positions is the same for A and B.
#include "common.h"
#include <hip/hip_ext.h>

#ifndef SLAMHIP_ABLATE
#define SLAMHIP_ABLATE 0
#endif
#ifndef SLAMHIP_MFMA_STAGE_ROWS
#define SLAMHIP_MFMA_STAGE_ROWS 128   // train rows per stage for 16-word rows (8-word rows: twice, 32-word rows: half)
#endif

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <bool B> struct IntC { static constexpr bool value = B; };

constexpr int kKeyShift = 21;
constexpr uint32_t kNoKey = 0xFFFFFFFFu;

// NW = waves per block: 4 (two blocks share a CU) or 8 (one block per CU: its waves share every
// staged train row, and the two waves of a SIMD advance together, stage by stage)
template <int W, int NW> struct Geo {
  static constexpr int NT = 64 * NW;                  // threads per block
  static constexpr int KS = W / 2;                    // matrix instructions per 32 x 32 tile
  static constexpr int CPR = W;                       // 16-byte chunks per expanded row
  static constexpr int R = SLAMHIP_MFMA_STAGE_ROWS * (W == 8 ? 2 : 1) / (W == 32 ? 2 : 1);   // train rows per stage
  static constexpr int SH = (W == 32) ? 12 : 13;      // key = (D - d) << SH | (MASK - row)
  static constexpr int MASK = (1 << SH) - 1;
  static constexpr int UNITS = R * W / 4 / NT;        // 16-byte packed units per thread and stage
  static_assert(R * W / 4 % NT == 0 && R <= NT, "stage units divide over the block");
};

// Running top-2 of float keys (larger is better), invariant b1 >= b2 among valid keys.
// Compiler builtins, not inline asm: the first read of a matrix-instruction result needs
// wait states that the compiler only inserts for instructions it knows (with an asm
// v_med3 as the first reader the second-best came out stale).
__device__ __forceinline__ float fmed3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
// Larger of two keys through their bit patterns: valid keys are non-negative floats, whose
// order is the order of their bits as integers, and every negative value ("no key", rows
// past the end) stays below all of them either way; unlike fmaxf this needs no
// canonicalising instruction in front.
__device__ __forceinline__ float kmax(float a, float b) {
  return __int_as_float(max(__float_as_int(a), __float_as_int(b)));
}
__device__ __forceinline__ float kmax3(float a, float b, float c) {   // v_max3_i32
  return __int_as_float(max(max(__float_as_int(a), __float_as_int(b)), __float_as_int(c)));
}
// Folds a tile's 16 keys of one query into the running (best, second best), two keys at a
// time, in 20 instructions (8 for the best alone) instead of 32:
//   t_j   = med3(b1, x, y)      the middle of the running best and the two new keys
//   b1    = max3(b1, x, y)
//   b2    = max(b2, t_0 .. t_7)
// Every t_j is the median of three keys seen so far, so it never exceeds the second largest
// key S; and S is caught by the pair it arrives in (if the largest key came earlier or sits in
// the same pair) or by the pair the largest key arrives in (S is the running best by then).
template <bool BEST>
__device__ __forceinline__ void fold16(const v16f &k, float &b1, float &b2) {
  float t[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (!BEST) t[j] = fmed3(b1, k[2 * j], k[2 * j + 1]);
    b1 = kmax3(b1, k[2 * j], k[2 * j + 1]);
  }
  if (!BEST) {
    b2 = kmax3(b2, t[0], t[1]);
    b2 = kmax3(b2, t[2], t[3]);
    b2 = kmax3(b2, t[4], t[5]);
    b2 = kmax3(b2, t[6], t[7]);
  }
}

// one packed dword -> 32 FP4 values (16 bytes): nibble j of word p = bit 4 j + p
// (v_and_or_b32 takes no literal operands on gfx9, and the compiler splits the expression into
// an and and an or when the masks are literals: the masks are passed in registers instead.)
__device__ __forceinline__ uint32_t and_or(uint32_t x, uint32_t m, uint32_t o) {
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(m), "v"(o));   // one scalar operand at most
  return r;
}
__device__ __forceinline__ v4i expand_dword(uint32_t x, uint32_t m8, uint32_t m2) {
  v4i o;
  o[0] = (int)and_or(x << 3, m8, m2);
  o[1] = (int)and_or(x << 2, m8, m2);
  o[2] = (int)and_or(x << 1, m8, m2);
  o[3] = (int)and_or(x, m8, m2);
  return o;
}

// 16 bytes at any alignment (global memory or LDS)
__device__ __forceinline__ uint4 load16(const uint8_t *p) {
  uint4 x;
  __builtin_memcpy(&x, p, 16);
  return x;
}
__device__ __forceinline__ void store16(uint8_t *p, uint4 v) { __builtin_memcpy(p, &v, 16); }

// Chunk `off / 16` of a descriptor row stored as it came (any alignment, `nbytes` >= 16 bytes
// long): bytes [off, off + 16) of the row, the ones behind its end reading as zero.  The 16-byte
// load is moved back to start at min(off, nbytes - 16), so it never leaves the row (and so never
// the array), and the bytes it then brings along from the chunk before are masked off: the
// chunk's own bytes sit at the high end of the result.  Query and train rows are fetched by this
// one function, so a descriptor bit lands on the same k position of both operands.
struct RawChunk {
  int start;          // byte offset of the load inside the row
  uint32_t m[4];      // the chunk's own bytes, per dword of the load
  __device__ __forceinline__ RawChunk(int off, int nbytes) {
    start = min(off, nbytes - 16);
    const int skip = min(off - start, 16);
#pragma unroll
    for (int j = 0; j < 4; ++j)   // dword j loses its low min(max(skip - 4 j, 0), 4) bytes
      m[j] = (uint32_t)(0xFFFFFFFFull << (8 * min(max(skip - 4 * j, 0), 4)));
  }
  __device__ __forceinline__ uint4 load(const uint8_t *row) const {   // global memory or LDS
    uint4 x;
    __builtin_memcpy(&x, row + start, 16);
    return make_uint4(x.x & m[0], x.y & m[1], x.z & m[2], x.w & m[3]);
  }
};

// swizzle term of a row (XORed into the 16-byte chunk index)
template <int W> __device__ __forceinline__ int row_swizzle(int row) {
  return W == 8 ? ((row >> 1) & 7) : (row & 15);
}

// RAW: qp / tp are the caller's descriptor rows as they are (desc_bytes >= 16 long, row_stride
// apart, any alignment) instead of rows packed to W dwords: no packing launch in front.
template <int W, int G, bool BEST, int NW, bool RAW>
__global__ __launch_bounds__(64 * NW, G == 4 ? 1 : 8 / NW) void hamming_mfma_kernel(
    const uint32_t *__restrict__ qp, const uint32_t *__restrict__ tp,
    const int32_t *__restrict__ q_off, const int32_t *__restrict__ q_cnt,
    const int32_t *__restrict__ t_off, const int32_t *__restrict__ t_cnt,
    const int32_t *__restrict__ out_off, int nq_single, int nt_single, int nq_total,
    uint2 *__restrict__ partial, int32_t *__restrict__ idx2, int32_t *__restrict__ dist2,
    unsigned *__restrict__ arrivals, double ratio, uint8_t *__restrict__ keep, int row_stride, int desc_bytes,
    int xq, int xs, int xp) {
  // Every argument is asked for at once: left alone, the compiler fetches them in three or four
  // dependent groups behind the early exits, one scalar-memory round trip each, in front of the
  // first descriptor load.
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" ::"s"(qp), "s"(tp), "s"(q_off), "s"(q_cnt), "s"(t_off), "s"(t_cnt), "s"(out_off), "s"(nq_single),
               "s"(nt_single), "s"(nq_total), "s"(partial), "s"(idx2), "s"(dist2), "s"(arrivals), "s"(keep),
               "s"(row_stride), "s"(desc_bytes), "s"(xq), "s"(xs), "s"(xp), "s"(gridDim.x), "s"(gridDim.y));
#endif
  using GEO = Geo<W, NW>;
  constexpr int kThreads = GEO::NT;
  constexpr int KS = GEO::KS, CPR = GEO::CPR, R = GEO::R, SH = GEO::SH, MASK = GEO::MASK;
  constexpr int D = 32 * W;
  constexpr int kStageSlots = R * CPR;                 // 16-byte slots per stage buffer
  // one LDS object: two stage buffers, then two C_in tables (one float per thread, the first
  // R of which are the stage's rows: every thread writes, so the loop body has no branch)
  __shared__ uint4 lds[2 * kStageSlots + 2 * kThreads / 4];
  float *ctab = reinterpret_cast<float *>(lds + 2 * kStageSlots);

  // Which (query block, split, pair) this block is.  Plain: the grid's three dimensions.  With
  // xq > 0 the grid is one-dimensional and blocks are dealt to the chip's eight XCDs by hand: the
  // hardware hands consecutive workgroups to consecutive XCDs, each with an L2 of its own, so
  // block id -> XCD id % 8, and every block of a pair (all query blocks, all splits) is given an
  // id of the same residue -- the pair's train rows then come into ONE L2 once instead of into
  // as many as the pair has query blocks (xq query blocks x xs splits x xp pairs; pairs are dealt
  // round-robin to the XCDs, ids whose pair does not exist leave at once).
  int pair = blockIdx.z, bx = blockIdx.x, split = blockIdx.y, n_splits = gridDim.y;
  if (xq > 0) {
    const int id = blockIdx.x, j = id >> 3;
    bx = j % xq;
    split = (j / xq) % xs;
    pair = (j / (xq * xs)) * 8 + (id & 7);
    n_splits = xs;
    if (pair >= xp) return;   // block-uniform
  }
  int q0 = 0, nq = nq_single, t0 = 0, nt = nt_single;
  if (q_off != nullptr) {
    q0 = q_off[pair];
    nq = q_cnt ? q_cnt[pair] : q_off[pair + 1] - q0;
    t0 = t_off[pair];
    nt = t_cnt ? t_cnt[pair] : t_off[pair + 1] - t0;
  }
  const int o0 = out_off ? out_off[pair] : q0;
  const int qblk = bx * (NW * 32 * G);
  if (qblk >= nq) return;  // block-uniform

  // whole tiles per split; the host keeps a split within MASK + 1 rows
  int rps = (nt + n_splits - 1) / n_splits;
  rps = (rps + 31) & ~31;
  const int r_begin = min(nt, split * rps);
  const int r_end = min(nt, r_begin + rps);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
#if (SLAMHIP_ABLATE & 32)  // timing experiment: lifetime of one wave of the first 64 blocks in shader
                           // cycles and in 100 MHz ticks, stored behind the last row of dist2
  int *const stamp_out = dist2 + 2 * (size_t)nq_total + (blockIdx.y * gridDim.x + blockIdx.x) * 40;
  const bool stamp_on = lane == 0 && wave == 1 && blockIdx.z == 0 && (blockIdx.y * gridDim.x + blockIdx.x) < 64;
  const unsigned long long stamp0 = __builtin_amdgcn_s_memtime();
  const unsigned long long real0 = __builtin_amdgcn_s_memrealtime();
#endif
#if (SLAMHIP_ABLATE & 64)  // timing experiment: start and end of every block in 100 MHz ticks, behind dist2
  const unsigned span0 = (unsigned)__builtin_amdgcn_s_memrealtime();
  unsigned span_prologue = span0, span_loop = span0, span_staged = span0;
#endif
#if (SLAMHIP_ABLATE & 64) && defined(__HIP_DEVICE_COMPILE__)
  unsigned span_t2 = span0, span_t3 = span0;
#endif
  const uint32_t m8 = 0x88888888u, m2 = 0x22222222u;

  // ---- queries -> expanded B fragments in registers.  Lane half h takes dwords
  // h * KS .. h * KS + KS - 1 of its row (k-step s <-> dword h * KS + s, on both operands):
  // contiguous, so 16-byte loads.
  v8i bq[G][KS];
  int qrow[G];
  uint4 qraw[G][KS / 4];
#pragma unroll
  for (int g = 0; g < G; ++g) qrow[g] = qblk + (wave * G + g) * 32 + r;
  // The block's query rows reach the lanes in one of two ways.  Directly: lane (r, h) loads its
  // row's chunks itself -- 32 rows a row pitch apart per request, which the memory pipeline of a
  // CU works through at a fifth of its rate for consecutive addresses, and every CU of the launch
  // does this at the same moment.  Or, when the rows fit (kQueryCopies requests per thread), as a
  // plain copy of the rows' bytes into LDS (consecutive lanes, consecutive 16 bytes; the image
  // lies in the second stage buffer, which nobody writes before the second barrier below), from
  // where each lane then picks its chunks.
  constexpr int kQueryCopies = 4;
  const int q_pitch = RAW ? row_stride : 4 * W, q_bytes = RAW ? desc_bytes : 4 * W;
  const int q_rows = min(NW * 32 * G, nq - qblk);
  const int q_len = (q_rows - 1) * q_pitch + q_bytes;             // bytes from the block's first row to its last row's end
  const bool q_via_lds = q_len <= kQueryCopies * kThreads * 16;   // block-uniform
  const uint8_t *const q_base = reinterpret_cast<const uint8_t *>(qp) + (size_t)(q0 + qblk) * q_pitch;
  uint8_t *const q_image = reinterpret_cast<uint8_t *>(lds + kStageSlots);
  uint4 qcopy[kQueryCopies];
  auto copy_offset = [&](int j) { return min((j * kThreads + (int)threadIdx.x) * 16, q_len - 16); };   // pieces past the end: the last one again
  auto load_query_copy = [&]() {
#pragma unroll
    for (int j = 0; j < kQueryCopies; ++j) qcopy[j] = load16(q_base + copy_offset(j));
  };
  auto store_query_copy = [&]() {
#pragma unroll
    for (int j = 0; j < kQueryCopies; ++j) store16(q_image + copy_offset(j), qcopy[j]);
  };
  // chunk k (16 bytes: four k-steps) of every group's query rows, from the image or from memory
  auto load_queries = [&](int k, const uint8_t *rows) {
#pragma unroll
    for (int g = 0; g < G; ++g)
      qraw[g][k] = RawChunk(4 * h * KS + 16 * k, q_bytes).load(rows + (size_t)(min(qrow[g], nq - 1) - qblk) * q_pitch);
  };
  const int scale_t = 127 + SH - 1, scale_q = 127;   // E8M0: 2^(SH-1) and 1

  float b1[G], b2[G];
#pragma unroll
  for (int g = 0; g < G; ++g) b1[g] = b2[g] = -1.0f;             // negative: no key yet
  // k-step s of every group's B fragments from the raw query dwords
  auto expand_query_step = [&](int s) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const uint4 v = qraw[g][s / 4];
      const uint32_t x = s % 4 == 0 ? v.x : (s % 4 == 1 ? v.y : (s % 4 == 2 ? v.z : v.w));
      const v4i e = expand_dword(x, m8, m2);
      bq[g][s] = v8i{e[0], e[1], e[2], e[3], 0, 0, 0, 0};
    }
  };

  if (r_begin < r_end) {
    // ---- per-lane LDS byte offsets: A fragment of k-step s inside a tile, C_in quads
    const int fr = row_swizzle<W>(r);
    int aoff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) aoff[s] = (r * CPR + ((h * KS + s) ^ fr)) * 16;
    const int coff = 2 * kStageSlots * 16 + 16 * h;

    // ---- stage staging: a thread owns UNITS packed 16-byte units (4 dwords of one row) per stage
    constexpr int UNITS = GEO::UNITS, PIECES = 4 * UNITS;   // a piece = one dword -> one 16-byte LDS slot
    int u_row[UNITS], u_dst[UNITS][4];
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
      const int unit = tid + k * kThreads;
      const int row = unit / (W / 4), c4 = unit % (W / 4);
      u_row[k] = row;
      const int f = row_swizzle<W>(row);
#pragma unroll
      for (int j = 0; j < 4; ++j) u_dst[k][j] = (row * CPR + ((4 * c4 + j) ^ f)) * 16;
    }
    const uint32_t *tbase = tp + (size_t)t0 * W + 4 * ((tid % (W / 4)));
    const uint8_t *traw = reinterpret_cast<const uint8_t *>(tp) + (size_t)t0 * row_stride;
    const RawChunk tchunk(16 * (tid % (W / 4)), RAW ? desc_bytes : 16);   // this thread's chunk of every row it stages
    char *const lbase = reinterpret_cast<char *>(lds);
    uint32_t ux[PIECES];
    auto load_stage = [&](int stage_begin) {
#pragma unroll
      for (int k = 0; k < UNITS; ++k) {
        const int grow = min(stage_begin + u_row[k], r_end - 1);   // rows past the end: the last row again
        const uint4 v = RAW ? tchunk.load(traw + (size_t)grow * row_stride)
                            : *reinterpret_cast<const uint4 *>(tbase + (size_t)grow * W);
        ux[4 * k + 0] = v.x; ux[4 * k + 1] = v.y; ux[4 * k + 2] = v.z; ux[4 * k + 3] = v.w;
      }
    };
    auto store_piece = [&](char *dst, int p) {
      *reinterpret_cast<v4i *>(dst + u_dst[p / 4][p % 4]) = expand_dword(ux[p], m8, m2);
    };
    // C_in of the stage's rows (one float per thread, the first R of them are rows: every thread
    // writes, so there is no branch); rows past the end get a value that can never win
    auto store_ctab = [&](int buf, int stage_begin) {
      const int lrow = stage_begin - r_begin + tid;                 // row index inside the split
      ctab[buf * kThreads + tid] = (stage_begin + tid < r_end)
                                       ? (float)((D << (SH - 1)) + (MASK - lrow)) : -1.0e30f;
    };
    auto read_cin = [&](const char *cbase, int tile) {
      v16f c;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 q4 = *reinterpret_cast<const float4 *>(cbase + (tile * 32 + 8 * j) * 4);
        c[4 * j + 0] = q4.x; c[4 * j + 1] = q4.y; c[4 * j + 2] = q4.z; c[4 * j + 3] = q4.w;
      }
      return c;
    };

    // ---- software pipeline.  The wave always holds the A fragments `a` and the C_in `cin` of
    // the tile it is about to multiply; each k-step issues its G matrix instructions, then
    // re-fills the fragment register it has just used with the same k-step of the NEXT tile,
    // then does a slice of the vector work: the top-2 fold of the PREVIOUS tile's keys `old`,
    // and one piece of the next stage's expansion.  Nothing in the loop waits for LDS or
    // memory it has only just asked for.
    constexpr int T = R / 32;                      // tiles per stage
    constexpr int SLOTS = (T - 1) * KS;            // k-steps before the stage's barrier
    static_assert(T >= 2 && SLOTS >= PIECES, "a stage has room for its staging pieces");
    v4i a[KS];
    v16f cin;
    float tprev[G];
#pragma unroll
    for (int g = 0; g < G; ++g) tprev[g] = -1.0f;
    constexpr bool PIPE = W < 32;   // 32-word rows: the fragments alone are 64 registers, no room to run ahead
    // `first` (a compile-time flag): the very first tile of the block.  It has no tile before it to
    // fold; its k-steps expand the next k-step's query fragments instead, so the block's first
    // matrix instruction waits for one k-step of the queries and not for all of them.
    auto tile_step = [&](int tile, const char *cur_a, const char *cur_c, const char *pf_a, const char *pf_c,
                         int pf_tile, char *st_dst, v16f (&acc)[G], const v16f (&old)[G], int ct_buf,
                         int ct_begin, auto first) {
      constexpr bool FIRST = decltype(first)::value;
      if (!PIPE) {
#pragma unroll
        for (int s = 0; s < KS; ++s) a[s] = *reinterpret_cast<const v4i *>(cur_a + tile * 32 * CPR * 16 + aoff[s]);
        cin = read_cin(cur_c, tile);
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const v8i a8 = v8i{a[s][0], a[s][1], a[s][2], a[s][3], 0, 0, 0, 0};
#pragma unroll
        for (int g = 0; g < G; ++g) {
#if (SLAMHIP_ABLATE & 2) && defined(__HIP_DEVICE_COMPILE__)   // timing experiment: no matrix instructions
          if (s == 0) acc[g] = cin;
          asm volatile("" ::"v"(a[s]), "v"(bq[g][s]));
#else
          acc[g] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, bq[g][s], s == 0 ? cin : acc[g], 4, 4, 0,
                                                                  scale_t, 0, scale_q);
#endif
        }
        if (PIPE) {
          a[s] = *reinterpret_cast<const v4i *>(pf_a + pf_tile * 32 * CPR * 16 + aoff[s]);
          if (s == 0) cin = read_cin(pf_c, pf_tile);
        }
        if (FIRST) {
          if (s + 1 < KS) expand_query_step(s + 1);
        } else {
        // fold: the 8 key pairs of `old` spread over the KS steps
#if !(SLAMHIP_ABLATE & 1)
#pragma unroll
        for (int j = s * 8 / KS; j < (s + 1) * 8 / KS; ++j) {
#pragma unroll
          for (int g = 0; g < G; ++g) {
            const float x = old[g][2 * j], y = old[g][2 * j + 1];
            if (!BEST) {
              const float t = fmed3(b1[g], x, y);
              if (j & 1) b2[g] = kmax3(b2[g], tprev[g], t);
              else tprev[g] = t;
            }
            b1[g] = kmax3(b1[g], x, y);
          }
        }
        // the slice's results are pinned to this k-step (left to itself the compiler gathers the
        // folds of several k-steps behind the stage's barrier, where no matrix instruction
        // runs beside them)
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (BEST) asm volatile("" : "+v"(b1[g]));
          else asm volatile("" : "+v"(b1[g]), "+v"(b2[g]), "+v"(tprev[g]));
        }
#else
        if (s == 0)
#pragma unroll
          for (int g = 0; g < G; ++g) asm volatile("" ::"v"(old[g]));
#endif
        }
        // staging: piece p of the next stage in k-step p * SLOTS / PIECES of this stage
        const int slot = tile * KS + s;
        if (slot == 1) store_ctab(ct_buf, ct_begin);   // the next stage's C_in table (read behind the barrier)
        if (tile < T - 1) {
#pragma unroll
          for (int p = 0; p < PIECES; ++p)
            if (p * SLOTS / PIECES == slot) store_piece(st_dst, p);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    v16f accA[G], accB[G];
    // Order of the block's first requests (they come back in this order, and every block of the
    // launch asks at the same moment): the first stage's rows, which all waves wait for at the
    // first barrier, then the query chunks in the order the first tile uses them.
#if (SLAMHIP_ABLATE & 64) && defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(aoff[KS - 1]), "v"(u_dst[0][3]));
    span_t2 = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
    load_stage(r_begin);
    if (q_via_lds) {
      load_query_copy();
    } else {
#pragma unroll
      for (int k = 0; k < KS / 4; ++k) load_queries(k, q_base);
    }
#pragma unroll
    for (int p = 0; p < PIECES; ++p) store_piece(lbase, p);
#if (SLAMHIP_ABLATE & 64) && defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    span_t3 = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
    store_ctab(0, r_begin);
    if (q_via_lds) {
      store_query_copy();
      __syncthreads();                         // the image is whole
#pragma unroll
      for (int k = 0; k < KS / 4; ++k) load_queries(k, q_image);
    }
    load_stage(r_begin + R);
    expand_query_step(0);                      // the other k-steps' fragments are expanded inside the first tile
#if (SLAMHIP_ABLATE & 64) && defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    span_staged = (unsigned)__builtin_amdgcn_s_memrealtime();
    if (lane == 0 && blockIdx.z == 0 && gridDim.x * gridDim.y <= 256) {   // every wave: its start and its arrival at the barrier
      int *o = dist2 + 2 * (size_t)nq_total + 8 * 256 + 16 * (blockIdx.y * gridDim.x + blockIdx.x) + 2 * wave;
      o[0] = (int)span0;
      o[1] = (int)span_staged;
      if ((wave & 3) == 0) {      // waves 0 and 4 (the two waves of one SIMD): the steps in between
        int *o2 = dist2 + 2 * (size_t)nq_total + 8 * 256 + 16 * 256 + 8 * (blockIdx.y * gridDim.x + blockIdx.x) + (wave & 4);
        o2[0] = (int)span0;
        o2[1] = (int)span_t2;
        o2[2] = (int)span_t3;
        o2[3] = (int)span_staged;
      }
    }
#endif
    __syncthreads();
#if (SLAMHIP_ABLATE & 64)
    span_prologue = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
    if (PIPE) {
#pragma unroll
      for (int s = 0; s < KS; ++s) a[s] = *reinterpret_cast<const v4i *>(lbase + aoff[s]);
      cin = read_cin(lbase + coff, 0);
    }
    // Every stage runs whole: rows past the end carry a C_in that can never win.  The first
    // stage is written out on its own (its first tile differs), the others loop.
    auto run_stage = [&](int sb, int buf, auto first) {
      const char *sbase = lbase + buf * (kStageSlots * 16);
      const char *cbase = lbase + coff + buf * (kThreads * 4);
      char *nbase = lbase + (buf ^ 1) * (kStageSlots * 16);
      const char *ncbase = lbase + coff + (buf ^ 1) * (kThreads * 4);
      // the next stage (in registers since the stage before) goes to the other buffer piece by
      // piece during tiles 0 .. T-2; nobody reads that buffer before the barrier
#pragma unroll
      for (int tile = 0; tile < T; tile += 2) {
        if (tile == 0)
          tile_step(tile, sbase, cbase, sbase, cbase, tile + 1, nbase, accA, accB, buf ^ 1, sb + R, first);
        else
          tile_step(tile, sbase, cbase, sbase, cbase, tile + 1, nbase, accA, accB, buf ^ 1, sb + R, IntC<false>{});
        if (tile + 2 == T) {
          // every piece of the next stage is on its way to LDS: meet the other waves (after
          // the barrier the next stage may be read and this one is no longer needed), and
          // ask for the stage after it (rows clamped when there is none)
          __syncthreads();
          load_stage(sb + 2 * R);
          tile_step(tile + 1, sbase, cbase, nbase, ncbase, 0, nbase, accB, accA, buf ^ 1, sb + R, IntC<false>{});
        } else {
          tile_step(tile + 1, sbase, cbase, sbase, cbase, tile + 2, nbase, accB, accA, buf ^ 1, sb + R, IntC<false>{});
        }
      }
    };
    run_stage(r_begin, 0, IntC<true>{});
    int buf = 1;
    for (int sb = r_begin + R; sb < r_end; sb += R, buf ^= 1) run_stage(sb, buf, IntC<false>{});
#if (SLAMHIP_ABLATE & 64)
    span_loop = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
    for (int g = 0; g < G; ++g) fold16<BEST>(accB[g], b1[g], b2[g]);
  }

  // ---- the two half-waves of a query meet; lane h == 0 writes
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float o1 = __shfl_xor(b1[g], 32, 64), o2 = __shfl_xor(b2[g], 32, 64);
    float m1 = b1[g], m2 = b2[g];
    if (!BEST) { m2 = fmed3(m1, m2, o1); }
    m1 = kmax(m1, o1);
    if (!BEST) { m2 = fmed3(m1, m2, o2); }     // o2 <= o1 <= m1: this keeps the larger of m2, o2
    if (h != 0 || qrow[g] >= nq) continue;
    // key -> (distance, global train row); negative = none
    uint32_t k1 = kNoKey, k2 = kNoKey;
    if (m1 >= 0.0f) {
      const uint32_t v = (uint32_t)m1;
      k1 = ((uint32_t)(D - (int)(v >> SH)) << kKeyShift) | (uint32_t)(r_begin + (MASK - (int)(v & MASK)));
    }
    if (!BEST && m2 >= 0.0f) {
      const uint32_t v = (uint32_t)m2;
      k2 = ((uint32_t)(D - (int)(v >> SH)) << kKeyShift) | (uint32_t)(r_begin + (MASK - (int)(v & MASK)));
    }
    const size_t row = (size_t)(o0 + qrow[g]);
    if (n_splits == 1) {
      const uint32_t im = (1u << kKeyShift) - 1u;
      idx2[2 * row + 0] = k1 == kNoKey ? -1 : (int32_t)(k1 & im);
      idx2[2 * row + 1] = k2 == kNoKey ? -1 : (int32_t)(k2 & im);
      dist2[2 * row + 0] = k1 == kNoKey ? -1 : (int32_t)(k1 >> kKeyShift);
      dist2[2 * row + 1] = k2 == kNoKey ? -1 : (int32_t)(k2 >> kKeyShift);
    } else if (arrivals == nullptr) {
      partial[(size_t)split * nq_total + row] = make_uint2(k1, k2);
    } else {
      // write-through store (sc1): visible to the block that folds the splits without a
      // release of this compute die's L2
      __hip_atomic_store(reinterpret_cast<unsigned long long *>(partial + (size_t)split * nq_total + row),
                         (unsigned long long)k1 | ((unsigned long long)k2 << 32), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  // ---- fold of the splits inside the launch: every block of a query block adds one to the
  // query block's arrival count once its key pairs are written; the block whose add comes
  // last reads all of them back (sc1 loads: past the vector L1) and writes the final result.
  // The count is zero at launch (the row packing launch clears it) and the last block puts
  // it back to zero.
  if (arrivals != nullptr && n_splits > 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's key pairs have left
    __syncthreads();                                      // ... and every other wave's of the block
    int *const last_flag = reinterpret_cast<int *>(lds);  // the stage buffers are no longer read
    unsigned *const cnt = arrivals + (size_t)pair * gridDim.x + blockIdx.x;   // plain grid only (launch())
    if (tid == 0) {
      const unsigned before = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last_flag[0] = before == (unsigned)(n_splits - 1);
    }
    __syncthreads();
    if (last_flag[0]) {
      if (tid == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int q = tid; q < NW * 32 * G; q += kThreads) {
        if (qblk + q >= nq) continue;
        const size_t row = (size_t)(o0 + qblk + q);
        uint32_t m1 = kNoKey, m2 = kNoKey;                // smallest, second smallest key
        for (int sp0 = 0; sp0 < n_splits; sp0 += 8) {     // eight loads in flight at a time
          unsigned long long v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j)
            v[j] = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(
                                         partial + (size_t)min(sp0 + j, n_splits - 1) * nq_total + row),
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int j = 0; j < 8; ++j) {                   // a split read twice changes nothing
            const uint32_t a = (uint32_t)v[j], b = (uint32_t)(v[j] >> 32);   // a <= b within a split
            // keys are unique (distance << 21 | train row), "none" is the largest value
            if (a < m1) { m2 = min(m1, b); m1 = a; }
            else if (a < m2 && a != m1) m2 = a;
          }
        }
        const uint32_t im = (1u << kKeyShift) - 1u;
        int2 oi, od;
        oi.x = m1 == kNoKey ? -1 : (int32_t)(m1 & im);
        oi.y = m2 == kNoKey ? -1 : (int32_t)(m2 & im);
        od.x = m1 == kNoKey ? -1 : (int32_t)(m1 >> kKeyShift);
        od.y = m2 == kNoKey ? -1 : (int32_t)(m2 >> kKeyShift);
        reinterpret_cast<int2 *>(idx2)[row] = oi;
        reinterpret_cast<int2 *>(dist2)[row] = od;
        if (keep != nullptr)
          keep[row] = (od.x >= 0 && od.y >= 0 && (double)od.x < ratio * (double)od.y) ? 1 : 0;
      }
    }
  }
#if (SLAMHIP_ABLATE & 64)
  if (tid == 0 && blockIdx.z == 0) {
    int *o = dist2 + 2 * (size_t)nq_total + 8 * (blockIdx.y * gridDim.x + blockIdx.x);
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    o[0] = (int)span0;
    o[1] = (int)(unsigned)__builtin_amdgcn_s_memrealtime();
    o[2] = (int)xcc;
    o[3] = (int)hw;
    o[4] = (int)span_prologue;
    o[5] = (int)span_loop;
    o[6] = 0;
    o[7] = (int)span_staged;
  }
#endif
#if (SLAMHIP_ABLATE & 32)  // slots 38 / 39: the wave's lifetime in shader cycles and in 100 MHz ticks
  if (stamp_on) {
    stamp_out[38] = (int)(__builtin_amdgcn_s_memtime() - stamp0);
    stamp_out[39] = (int)(__builtin_amdgcn_s_memrealtime() - real0);
  }
#endif
}

template <int W, int G, bool BEST, int NW>
void launch(dim3 grid, hipStream_t st, const uint32_t *qp, const uint32_t *tp, const int32_t *q_off,
            const int32_t *q_cnt, const int32_t *t_off, const int32_t *t_cnt, const int32_t *out_off,
            int nq, int nt, int nq_total, uint2 *part, int32_t *idx2, int32_t *dist2, unsigned *arrivals,
            double ratio, uint8_t *keep, const slamhip_internal::RawRows *raw) {
  // Batches of pairs with several blocks each: blocks dealt to the XCDs pair by pair (see the
  // kernel).  A single pair keeps the plain grid (its query blocks, grid.x, are a multiple of
  // eight where it matters, so blocks that share queries already share an XCD), and so does the
  // fold inside the kernel, whose counters are indexed by the plain grid.
  int xq = 0, xs = 0, xp = 0;
  if (grid.z >= 64 && grid.x * grid.y > 1 && arrivals == nullptr && slamhip_internal::g_xcd_remap != 0) {
    xq = (int)grid.x; xs = (int)grid.y; xp = (int)grid.z;
    grid = dim3(grid.x * grid.y * ((grid.z + 7) / 8 * 8), 1, 1);
  }
  hipEvent_t t0 = slamhip_internal::g_dispatch_start, t1 = slamhip_internal::g_dispatch_stop;
  if (t0 != nullptr && t1 != nullptr) {   // duration probe: the dispatch's own start and end
    if (raw != nullptr)
      hipExtLaunchKernelGGL((hamming_mfma_kernel<W, G, BEST, NW, true>), grid, dim3(64 * NW), 0, st, t0, t1, 0, qp, tp,
                            q_off, q_cnt, t_off, t_cnt, out_off, nq, nt, nq_total, part, idx2, dist2, arrivals, ratio,
                            keep, raw->row_stride, raw->desc_bytes, xq, xs, xp);
    else
      hipExtLaunchKernelGGL((hamming_mfma_kernel<W, G, BEST, NW, false>), grid, dim3(64 * NW), 0, st, t0, t1, 0, qp, tp,
                            q_off, q_cnt, t_off, t_cnt, out_off, nq, nt, nq_total, part, idx2, dist2, arrivals, ratio,
                            keep, 0, 0, xq, xs, xp);
    return;
  }
  if (raw != nullptr)
    hipLaunchKernelGGL((hamming_mfma_kernel<W, G, BEST, NW, true>), grid, dim3(64 * NW), 0, st, qp, tp, q_off,
                       q_cnt, t_off, t_cnt, out_off, nq, nt, nq_total, part, idx2, dist2, arrivals, ratio, keep,
                       raw->row_stride, raw->desc_bytes, xq, xs, xp);
  else
    hipLaunchKernelGGL((hamming_mfma_kernel<W, G, BEST, NW, false>), grid, dim3(64 * NW), 0, st, qp, tp, q_off,
                       q_cnt, t_off, t_cnt, out_off, nq, nt, nq_total, part, idx2, dist2, arrivals, ratio, keep, 0,
                       0, xq, xs, xp);
}

}  // namespace

namespace slamhip_internal {

int mfma_max_split_rows(int W) { return W == 32 ? (1 << 12) : (1 << 13); }

void launch_hamming_mfma(int W, int G, int NW, bool best, dim3 grid, hipStream_t st, const uint32_t *qp,
                         const uint32_t *tp, const int32_t *q_off, const int32_t *q_cnt,
                         const int32_t *t_off, const int32_t *t_cnt, const int32_t *out_off, int nq,
                         int nt, int nq_total, void *part, int32_t *idx2, int32_t *dist2,
                         unsigned *arrivals, double ratio, uint8_t *keep, const RawRows *raw) {
#define SLAMHIP_MFMA_CASE(WW, GG, NN)                                                                \
  if (W == WW && G == GG && NW == NN) {                                                              \
    if (best) launch<WW, GG, true, NN>(grid, st, qp, tp, q_off, q_cnt, t_off, t_cnt, out_off, nq, nt, \
                                       nq_total, (uint2 *)part, idx2, dist2, arrivals, ratio, keep,  \
                                       raw);                                                         \
    else launch<WW, GG, false, NN>(grid, st, qp, tp, q_off, q_cnt, t_off, t_cnt, out_off, nq, nt,    \
                                   nq_total, (uint2 *)part, idx2, dist2, arrivals, ratio, keep, raw); \
    return;                                                                                          \
  }
  SLAMHIP_MFMA_CASE(8, 1, 4) SLAMHIP_MFMA_CASE(8, 2, 4) SLAMHIP_MFMA_CASE(8, 2, 8)
  SLAMHIP_MFMA_CASE(16, 1, 4) SLAMHIP_MFMA_CASE(16, 2, 4) SLAMHIP_MFMA_CASE(16, 2, 8) SLAMHIP_MFMA_CASE(16, 4, 4)
  SLAMHIP_MFMA_CASE(32, 1, 4)
#undef SLAMHIP_MFMA_CASE
}

}  // namespace slamhip_internal
