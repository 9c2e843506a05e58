/*
 * CPU restatement of brute-force Hamming k=2 matching.  TEST INFRASTRUCTURE:
 * the checker for the HIP matcher at sizes NumPy is too slow for, and the
 * "port" CPU baseline timed by bench.py.  Never linked into the product path.
 *
 * Follows cv2.BFMatcher(NORM_HAMMING).knnMatch(k=2) as the reference calls it
 * (frontend/vision/descriptor_matcher.py:26,33): for each query row scan the
 * train rows in ascending index order and insert with a strict `<`
 * (OpenCV modules/core/src/batch_distance.cpp), so equal distances keep the
 * lowest train index.  PARITY UNPINNED against OpenCV itself (not installed).
 */
#include <stdint.h>
#include <string.h>
#include <limits.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline int row_distance(const uint8_t *a, const uint8_t *b, int nbytes) {
  int d = 0, i = 0;
  for (; i + 8 <= nbytes; i += 8) {
    uint64_t x, y;
    memcpy(&x, a + i, 8);
    memcpy(&y, b + i, 8);
    d += __builtin_popcountll(x ^ y);
  }
  if (i + 4 <= nbytes) {
    uint32_t x, y;
    memcpy(&x, a + i, 4);
    memcpy(&y, b + i, 4);
    d += __builtin_popcount(x ^ y);
    i += 4;
  }
  for (; i < nbytes; ++i) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
  return d;
}

/* idx2/dist2: nq*2 int32, -1 where absent.  threads <= 0: OpenMP default. */
void oracle_hamming_knn2(const uint8_t *q, int nq, const uint8_t *t, int nt, int nbytes,
                         int stride, int32_t *idx2, int32_t *dist2, int threads) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(static)
  for (int i = 0; i < nq; ++i) {
    const uint8_t *qi = q + (size_t)i * stride;
    int d0 = INT_MAX, d1 = INT_MAX, i0 = -1, i1 = -1;
    for (int j = 0; j < nt; ++j) {
      const int d = row_distance(qi, t + (size_t)j * stride, nbytes);
      if (d < d1) {
        if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
        else        { d1 = d;  i1 = j; }
      }
    }
    idx2[2 * i] = i0;     dist2[2 * i] = i0 < 0 ? -1 : d0;
    idx2[2 * i + 1] = i1; dist2[2 * i + 1] = i1 < 0 ? -1 : d1;
  }
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
