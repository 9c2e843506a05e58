// Device code shared by the kernels that order a segment of matches by (distance, index): a
// stable least-significant-digit radix sort of one segment in one block's LDS.
//
// Keys are dist << 21 | index (absent distances, < 0, take 2047 and sort last); the distance
// field has 11 bits, so two passes over 6-bit digits order the segment, and equal distances stay
// in index order, which is the order they start in.  A pass: every wavefront takes chunks of 64
// keys; the lanes of a chunk that hold the same digit find each other with one ballot per digit
// bit, which gives a key's rank among them and their number -- noted per (digit, chunk); an
// exclusive scan over those counts in (digit, chunk) order turns them into positions.  Nine
// barriers per segment where a bitonic network has log2(N) (log2(N) + 1) / 2 + 2 = 68 for 1500
// keys, and work that grows with n, not n log^2 n.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slamhip_dev {

constexpr int kSortKeyShift = 21;
constexpr uint32_t kSortIdxMask = (1u << kSortKeyShift) - 1u;

__device__ __forceinline__ uint32_t sort_key(int dist, int index) {
  return ((dist < 0 ? 2047u : (uint32_t)dist) << kSortKeyShift) | (uint32_t)index;
}

// Inclusive sum over the lanes of a wavefront in lane order, through DPP (a few cycles a step;
// six shuffles through the LDS hardware take some hundred cycles each, and the scan sits between
// two barriers of every radix pass): shifts by 1, 2, 4, 8 within the rows of 16 lanes (a lane
// whose source falls outside its row adds 0), then row 0's total to row 1 and row 2's to row 3,
// then the first half's total to the second half.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
  return v;
}

// LDS of a block of T threads that sorts up to CAP keys: 2 x CAP keys (ping-pong) + per pass CAP
// 16-bit counts ([digit][chunk], chunks = ceil(n / 64) <= CAP / 64) + the wavefronts' totals.
template <int T, int CAP>
struct RadixSortLds {
  uint32_t keys[2][CAP];
  uint16_t counts[2][CAP];
  uint32_t wave_total[2][T / 64];
};

// Sorts s.keys[0][0 .. n) (written by the block; n <= CAP, block-uniform) in place; every
// thread of the block calls it; the result is visible to all of them when it returns.
template <int T, int CAP>
__device__ __forceinline__ void radix_sort_lds(RadixSortLds<T, CAP> &s, int n) {
  constexpr int W = T / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunks = (n + 63) >> 6, entries = chunks << 6;
  // digit of the key at position c * 64 + lane, its rank among the chunk's keys with that digit,
  // and how many those are
  auto classify = [&](const uint32_t *src, int c, int shift, uint32_t &key, int &digit, int &rank, int &same) {
    const int e = (c << 6) + lane;
    const bool valid = e < n;
    key = valid ? src[e] : 0u;
    digit = (int)((key >> shift) & 63u);
    unsigned long long peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const bool bit = (digit >> b) & 1;
      const unsigned long long has = __builtin_amdgcn_ballot_w64(bit);
      peers &= bit ? has : ~has;
    }
    rank = __builtin_popcountll(peers & ((1ull << lane) - 1ull));
    same = __builtin_popcountll(peers);
    return valid;
  };
  // both passes' count tables are cleared up front: no barrier between a pass's last reads of its
  // table and the clearing for the next
  for (int i = threadIdx.x; i < entries; i += T) { s.counts[0][i] = 0; s.counts[1][i] = 0; }
  __syncthreads();   // also: the caller's keys are all written
  for (int pass = 0; pass < 2; ++pass) {
    const uint32_t *src = s.keys[pass];
    uint32_t *dst = s.keys[pass ^ 1];
    uint16_t *counts = s.counts[pass];
    const int shift = kSortKeyShift + 6 * pass;
    for (int c = wave; c < chunks; c += W) {
      uint32_t key; int digit, rank, same;
      if (classify(src, c, shift, key, digit, rank, same) && rank == 0)
        counts[digit * chunks + c] = (uint16_t)same;
    }
    __syncthreads();
    // exclusive scan of counts[0 .. entries): a run per thread, the runs' totals over the block
    const int per = (entries + T - 1) / T;
    const int b0 = min((int)threadIdx.x * per, entries), b1 = min(b0 + per, entries);
    uint32_t mine = 0;
    for (int i = b0; i < b1; ++i) mine += counts[i];
    const uint32_t incl = wave_inclusive_sum(mine);
    if (lane == 63) s.wave_total[pass][wave] = incl;
    __syncthreads();
    uint32_t running = incl - mine;
#pragma unroll
    for (int w = 0; w < W; ++w) running += w < wave ? s.wave_total[pass][w] : 0u;
    for (int i = b0; i < b1; ++i) {
      const uint32_t v = counts[i];
      counts[i] = (uint16_t)running;
      running += v;
    }
    __syncthreads();
    for (int c = wave; c < chunks; c += W) {
      uint32_t key; int digit, rank, same;
      if (classify(src, c, shift, key, digit, rank, same)) dst[counts[digit * chunks + c] + rank] = key;
    }
    __syncthreads();
  }
}

}  // namespace slamhip_dev
