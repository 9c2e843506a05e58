// Probe for the matrix-core formulation of the Hamming matcher on gfx950:
//   Hamming(a, b) over D bits = (D - <a', b'>) / 2 with a', b' in {+1, -1}^D,
// so a 32 x 32 tile of distances is a 32 x 32 x D contraction.  With the query side
// scaled by -2^12 and the accumulator preloaded with D * 2^12 + row index, the
// matrix instruction's result IS the matcher's sort key  dist << 13 | index.
//
// Checks, with exact integer data:
//   1. v_mfma_i32_32x32x32_i8       (elements +-64 x -+64 = -2^12 per bit pair)
//   2. v_mfma_scale_f32_32x32x64_f8f6f4, FP4 operands +-1, A scale 2^12 (E8M0)
// that every output element equals the key computed on the host, when both operands
// are loaded the same way (lane l: row l & 31, 16-byte chunk l >> 5 of the k-step);
// then times back-to-back issue of both.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_probe tools/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));        \
      exit(2);                                                                         \
    }                                                                                  \
  } while (0)

constexpr int KSTEPS = 8;       // 512 bit pairs: 16 i8 steps of 32, or 8 fp4 steps of 64
constexpr int D_BITS = 486;     // valid bits; the rest of the 512 are zero padding

// a: (32 rows, ROWBYTES), b: same; out (32 x 32) keys, out[row * 32 + col]
__global__ void i8_tile(const uint8_t *a, const uint8_t *b, int *out) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  v16i acc;
  for (int i = 0; i < 16; ++i) acc[i] = (D_BITS << 12) + ((i & 3) + 8 * (i >> 2) + 4 * h);
  for (int s = 0; s < 16; ++s) {
    const v4i fa = *(const v4i *)(a + r * 512 + s * 32 + h * 16);
    const v4i fb = *(const v4i *)(b + r * 512 + s * 32 + h * 16);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) out[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

__global__ void fp4_tile(const uint8_t *a, const uint8_t *b, float *out) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  v16f acc;
  for (int i = 0; i < 16; ++i) acc[i] = (float)((D_BITS << 12) + ((i & 3) + 8 * (i >> 2) + 4 * h));
  const int scale_a = 127 + 12, scale_b = 127;
  for (int s = 0; s < 8; ++s) {
    const v4i qa = *(const v4i *)(a + r * 256 + s * 32 + h * 16);
    const v4i qb = *(const v4i *)(b + r * 256 + s * 32 + h * 16);
    v8i fa = {qa[0], qa[1], qa[2], qa[3], 0, 0, 0, 0};
    v8i fb = {qb[0], qb[1], qb[2], qb[3], 0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, acc, 4, 4, 0, scale_a, 0, scale_b);
  }
  for (int i = 0; i < 16; ++i) out[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

// back-to-back issue, NACC independent accumulators per wave, one wave per SIMD x waves_per_simd
template <int NACC>
__global__ void i8_rate(int iters, int *sink) {
  v16i acc[NACC];
  for (int k = 0; k < NACC; ++k)
    for (int i = 0; i < 16; ++i) acc[k][i] = threadIdx.x + i + k;
  v4i fa = {(int)threadIdx.x * 0x01010101, 0x40C040C0, (int)blockIdx.x, 0x40404040};
  v4i fb = {0x40C0C040, (int)threadIdx.x, 0x40404040, 0x1};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc[k], 0, 0, 0);
  }
  int s = 0;
  for (int k = 0; k < NACC; ++k)
    for (int i = 0; i < 16; ++i) s += acc[k][i];
  if (s == 0x7fffffff) sink[0] = s;
}

template <int NACC>
__global__ void fp4_rate(int iters, float *sink) {
  v16f acc[NACC];
  for (int k = 0; k < NACC; ++k)
    for (int i = 0; i < 16; ++i) acc[k][i] = (float)(threadIdx.x + i + k);
  v8i fa = {(int)(0x2A2A2A2A ^ (threadIdx.x * 0x08080808u)), 0x22AA22AA, 0x2A2AA2A2, 0x22222222, 0, 0, 0, 0};
  v8i fb = {0x2AA22AA2, (int)(0xA2A2A2A2 ^ (threadIdx.x * 0x80808080u)), 0x2222AAAA, 0x2A2A2A2A, 0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NACC; ++k)
      acc[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, acc[k], 4, 4, 0, 127, 0, 127);
  }
  float s = 0;
  for (int k = 0; k < NACC; ++k)
    for (int i = 0; i < 16; ++i) s += acc[k][i];
  if (s == 1.2345f) sink[0] = s;
}

int main() {
  srand(7);
  std::vector<uint8_t> bits_a(32 * 64), bits_b(32 * 64);   // 32 rows x 512 bits, packed
  for (int r = 0; r < 32; ++r)
    for (int k = 0; k < 64; ++k) {
      uint8_t x = rand() & 0xFF, y = (r % 3 == 0) ? (uint8_t)(x ^ (1u << (rand() & 7))) : (uint8_t)(rand() & 0xFF);
      if (k * 8 >= D_BITS) x = y = 0;                       // padding (bit-exact tail handled below)
      bits_a[r * 64 + k] = x;
      bits_b[r * 64 + k] = y;
    }
  auto bit = [](const std::vector<uint8_t> &v, int r, int k) { return (v[r * 64 + (k >> 3)] >> (k & 7)) & 1; };
  // host keys: key[row][col] = hamming(a_row, b_col) << 13 | row
  std::vector<int> ref(32 * 32);
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      int d = 0;
      for (int k = 0; k < D_BITS; ++k) d += bit(bits_a, i, k) ^ bit(bits_b, j, k);
      ref[i * 32 + j] = (d << 13) | i;
    }
  // expanded operands: valid bit -> +-1 (a: +-64 / fp4 +-1; b: -+64 / fp4 -+1), padding -> 0
  std::vector<uint8_t> a8(32 * 512), b8(32 * 512), a4(32 * 256, 0), b4(32 * 256, 0);
  for (int r = 0; r < 32; ++r)
    for (int k = 0; k < 512; ++k) {
      const bool valid = k < D_BITS;
      const int ba = bit(bits_a, r, k), bb = bit(bits_b, r, k);
      a8[r * 512 + k] = valid ? (uint8_t)(int8_t)(ba ? -64 : 64) : 0;
      b8[r * 512 + k] = valid ? (uint8_t)(int8_t)(bb ? 64 : -64) : 0;
      const uint8_t na = valid ? (ba ? 0xA : 0x2) : 0, nb = valid ? (bb ? 0x2 : 0xA) : 0;
      a4[r * 256 + (k >> 1)] |= na << (4 * (k & 1));
      b4[r * 256 + (k >> 1)] |= nb << (4 * (k & 1));
    }
  uint8_t *da8, *db8, *da4, *db4;
  int *oi;
  float *of;
  CK(hipMalloc(&da8, a8.size())); CK(hipMalloc(&db8, b8.size()));
  CK(hipMalloc(&da4, a4.size())); CK(hipMalloc(&db4, b4.size()));
  CK(hipMalloc(&oi, 1024 * 4)); CK(hipMalloc(&of, 1024 * 4));
  CK(hipMemcpy(da8, a8.data(), a8.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(db8, b8.data(), b8.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(da4, a4.data(), a4.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(db4, b4.data(), b4.size(), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(i8_tile, dim3(1), dim3(64), 0, 0, da8, db8, oi);
  hipLaunchKernelGGL(fp4_tile, dim3(1), dim3(64), 0, 0, da4, db4, of);
  CK(hipDeviceSynchronize());
  std::vector<int> hi(1024);
  std::vector<float> hf(1024);
  CK(hipMemcpy(hi.data(), oi, 4096, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hf.data(), of, 4096, hipMemcpyDeviceToHost));
  int bad_i = 0, bad_f = 0;
  for (int e = 0; e < 1024; ++e) {
    bad_i += hi[e] != ref[e];
    bad_f += hf[e] != (float)ref[e];
  }
  printf("i8  32x32x32 keys: %d of 1024 wrong (e.g. got %d want %d)\n", bad_i, hi[37], ref[37]);
  printf("fp4 32x32x64 keys: %d of 1024 wrong (e.g. got %.1f want %d)\n", bad_f, hf[37], ref[37]);

  // issue rate: blocks of 256 threads (one wave per SIMD) x 1 or 2 blocks per CU
  int *si; float *sf;
  CK(hipMalloc(&si, 64)); CK(hipMalloc(&sf, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  auto time_it = [&](auto launch, const char *name, int nacc, int blocks) {
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double waves_per_simd = blocks / 256.0;
    const double mfma_per_simd = (double)iters * nacc * waves_per_simd;
    printf("%-26s nacc %d  %4d blocks: %.3f ms, %.1f ns per MFMA per SIMD (= %.1f cycles at 2.4 GHz)\n", name,
           nacc, blocks, ms, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4);
  };
  for (int blocks : {256, 512}) {
    time_it([&] { hipLaunchKernelGGL(i8_rate<1>, dim3(blocks), dim3(256), 0, 0, iters, si); }, "i8 32x32x32", 1, blocks);
    time_it([&] { hipLaunchKernelGGL(i8_rate<4>, dim3(blocks), dim3(256), 0, 0, iters, si); }, "i8 32x32x32", 4, blocks);
    time_it([&] { hipLaunchKernelGGL(fp4_rate<1>, dim3(blocks), dim3(256), 0, 0, iters, sf); }, "fp4 scaled 32x32x64", 1, blocks);
    time_it([&] { hipLaunchKernelGGL(fp4_rate<4>, dim3(blocks), dim3(256), 0, 0, iters, sf); }, "fp4 scaled 32x32x64", 4, blocks);
  }
  return (bad_i || bad_f) ? 1 : 0;
}
