// Hand-over record of a finished front-end batch: the rows in use of every per-frame and per-pair
// result array of slamhip_frontend_batch, compacted into ONE byte string on the device, so that a
// rank's results cross to the rank that runs the back-end in a single gather.  It carries what
// TrackingDB.add_frame and the bundle adjustment consume per frame (run_slam.py:88-89,
// backend/tracking/fill_database.py:231 of the reference): links' pixels, inlier descriptors,
// temporal matches, inlier masks, poses.
//
// Layout (little endian): int64 head[5] = {frames, pairs, first_frame, first_pair, desc_bytes},
// int64 stereo counts[frames], int64 temporal counts[pairs], then, each flat over the owned
// frames: left kp index, right kp index, distance (int32), xyz (3 float32), left pixel, right
// pixel (2 float32 each), inlier descriptor (desc_bytes uint8); over the owned pairs: query
// index, train index, distance (int32), mask (uint8); then per pair refined pose, RANSAC pose
// (12 float64 each) and the 8 int32 of ransac_info.
#include "common.h"

namespace {

struct RecordLayout {
  long long st_cnt, tm_cnt, st_left, st_right, st_dist, xyz, pix_left, pix_right, des, tm_q, tm_t, tm_d, mask,
      rt, rt_ransac, info, total;
};

RecordLayout make_layout(long long nf, long long np, long long ns, long long nt, int D) {
  RecordLayout L;
  long long o = 5 * 8;
  L.st_cnt = o; o += nf * 8;
  L.tm_cnt = o; o += np * 8;
  L.st_left = o; o += ns * 4;
  L.st_right = o; o += ns * 4;
  L.st_dist = o; o += ns * 4;
  L.xyz = o; o += ns * 12;
  L.pix_left = o; o += ns * 8;
  L.pix_right = o; o += ns * 8;
  L.des = o; o += ns * D;
  L.tm_q = o; o += nt * 4;
  L.tm_t = o; o += nt * 4;
  L.tm_d = o; o += nt * 4;
  L.mask = o; o += nt;
  L.rt = o; o += np * 96;
  L.rt_ransac = o; o += np * 96;
  L.info = o; o += np * 32;
  L.total = o;
  return L;
}

// Fields start at any byte offset (the descriptor field has an odd length): stores go byte by
// byte through memcpy-like helpers; a wave's lanes write consecutive bytes.
__device__ __forceinline__ void put32(uint8_t *dst, uint32_t v) {
  if ((reinterpret_cast<uintptr_t>(dst) & 3) == 0) { *reinterpret_cast<uint32_t *>(dst) = v; return; }
  dst[0] = (uint8_t)v; dst[1] = (uint8_t)(v >> 8); dst[2] = (uint8_t)(v >> 16); dst[3] = (uint8_t)(v >> 24);
}
__device__ __forceinline__ void put64(uint8_t *dst, unsigned long long v) {
  put32(dst, (uint32_t)v);
  put32(dst + 4, (uint32_t)(v >> 32));
}

// Segment y, dealt over grid.y (a launch has at most 65535 rows of blocks; a batch of 65535
// frames has twice as many segments): [0, nf) = owned frame, [nf, nf + np) = owned pair,
// nf + np = head, counts and the per-pair fixed fields
__global__ __launch_bounds__(256) void pack_record_kernel(
    RecordLayout L, int nf, int np, int frame_lo, int pair_lo, long long first_frame, long long first_pair, int D,
    int W, const int32_t *__restrict__ off_left, const long long *__restrict__ st_out_off,
    const long long *__restrict__ tm_out_off, const int32_t *__restrict__ st_left,
    const int32_t *__restrict__ st_right, const int32_t *__restrict__ st_dist, const float *__restrict__ st_xyz,
    const float *__restrict__ st_pix_left, const float *__restrict__ st_pix_right,
    const uint32_t *__restrict__ st_desc, const int32_t *__restrict__ tm_query,
    const int32_t *__restrict__ tm_train, const int32_t *__restrict__ tm_dist, const uint8_t *__restrict__ tm_mask,
    const double *__restrict__ pose_refined, const double *__restrict__ pose_ransac,
    const int32_t *__restrict__ info, uint8_t *__restrict__ out) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  for (int y = blockIdx.y; y < nf + np + 1; y += gridDim.y) {
  if (y < nf) {
    const long long o0 = st_out_off[y];
    const int n = (int)(st_out_off[y + 1] - o0);
    const size_t src0 = (size_t)off_left[frame_lo + y];
    for (int r = tid; r < n; r += stride) {
      const size_t s = src0 + r;
      const long long o = o0 + r;
      put32(out + L.st_left + 4 * o, (uint32_t)st_left[s]);
      put32(out + L.st_right + 4 * o, (uint32_t)st_right[s]);
      put32(out + L.st_dist + 4 * o, (uint32_t)st_dist[s]);
#pragma unroll
      for (int k = 0; k < 3; ++k) put32(out + L.xyz + 12 * o + 4 * k, __float_as_uint(st_xyz[3 * s + k]));
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        put32(out + L.pix_left + 8 * o + 4 * k, __float_as_uint(st_pix_left[2 * s + k]));
        put32(out + L.pix_right + 8 * o + 4 * k, __float_as_uint(st_pix_right[2 * s + k]));
      }
    }
    // descriptors: the frame's n x D bytes are one contiguous run of the record
    const uint8_t *dsrc = reinterpret_cast<const uint8_t *>(st_desc) + src0 * (size_t)(4 * W);
    uint8_t *ddst = out + L.des + o0 * D;
    for (long long b = tid; b < (long long)n * D; b += stride) {
      const int r = (int)(b / D), c = (int)(b - (long long)r * D);
      ddst[b] = dsrc[(size_t)r * (4 * W) + c];
    }
  } else if (y < nf + np) {
    const int p = y - nf;
    const long long o0 = tm_out_off[p];
    const int n = (int)(tm_out_off[p + 1] - o0);
    const size_t src0 = (size_t)off_left[pair_lo + p];
    for (int r = tid; r < n; r += stride) {
      const size_t s = src0 + r;
      const long long o = o0 + r;
      put32(out + L.tm_q + 4 * o, (uint32_t)tm_query[s]);
      put32(out + L.tm_t + 4 * o, (uint32_t)tm_train[s]);
      put32(out + L.tm_d + 4 * o, (uint32_t)tm_dist[s]);
      out[L.mask + o] = tm_mask[s];
    }
  } else {
    if (tid == 0) {
      put64(out + 0, (unsigned long long)nf);
      put64(out + 8, (unsigned long long)np);
      put64(out + 16, (unsigned long long)first_frame);
      put64(out + 24, (unsigned long long)first_pair);
      put64(out + 32, (unsigned long long)D);
    }
    for (int j = tid; j < nf; j += stride) put64(out + L.st_cnt + 8 * (long long)j, (unsigned long long)(st_out_off[j + 1] - st_out_off[j]));
    for (int j = tid; j < np; j += stride) put64(out + L.tm_cnt + 8 * (long long)j, (unsigned long long)(tm_out_off[j + 1] - tm_out_off[j]));
    for (int j = tid; j < np * 12; j += stride) {
      put64(out + L.rt + 8 * (long long)j, (unsigned long long)__double_as_longlong(pose_refined[(size_t)pair_lo * 12 + j]));
      put64(out + L.rt_ransac + 8 * (long long)j, (unsigned long long)__double_as_longlong(pose_ransac[(size_t)pair_lo * 12 + j]));
    }
    for (int j = tid; j < np * 8; j += stride) put32(out + L.info + 4 * (long long)j, (uint32_t)info[(size_t)pair_lo * 8 + j]);
  }
  }
}

/* One new frame's packet -> the rows of the caller's arrays it belongs at.  blockIdx.y names the
 * piece: 0 left pixels, 1 right pixels (both as 32-bit words), 2 left descriptors, 3 right
 * descriptors (bytes: a row of `stride` bytes need not start on a word), 4 the six offsets. */
__global__ __launch_bounds__(256) void unpack_frame_kernel(const uint8_t *__restrict__ packet, const int n_left,
                                                           const int n_right, const int D, const int stride,
                                                           float *__restrict__ kp_left, float *__restrict__ kp_right,
                                                           uint8_t *__restrict__ des_left,
                                                           uint8_t *__restrict__ des_right, int32_t *__restrict__ off) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, step = gridDim.x * blockDim.x;
  const size_t at_kp_left = 24, at_kp_right = at_kp_left + 8 * (size_t)n_left,
               at_des_left = at_kp_right + 8 * (size_t)n_right, at_des_right = at_des_left + (size_t)D * n_left;
  switch (blockIdx.y) {
    case 0:
    case 1: {
      const uint32_t *src = (const uint32_t *)(packet + (blockIdx.y == 0 ? at_kp_left : at_kp_right));
      uint32_t *dst = (uint32_t *)(blockIdx.y == 0 ? kp_left : kp_right);
      const int words = 2 * (blockIdx.y == 0 ? n_left : n_right);
      for (int i = tid; i < words; i += step) dst[i] = src[i];
      break;
    }
    case 2:
    case 3: {
      const uint8_t *src = packet + (blockIdx.y == 2 ? at_des_left : at_des_right);
      uint8_t *dst = blockIdx.y == 2 ? des_left : des_right;
      const long long bytes = (long long)D * (blockIdx.y == 2 ? n_left : n_right);
      if (stride == D) {
        for (long long i = tid; i < bytes; i += step) dst[i] = src[i];
      } else {
        for (long long i = tid; i < bytes; i += step) dst[(i / D) * stride + i % D] = src[i];
      }
      break;
    }
    default:
      if (off && tid < 6) off[tid] = ((const int32_t *)packet)[tid];
  }
}

}  // namespace

extern "C" {

size_t slamhip_frontend_record_bytes(int n_frames, int n_pairs, long long stereo_rows, long long temporal_rows,
                                     int desc_bytes) {
  if (n_frames < 0 || n_pairs < 0 || stereo_rows < 0 || temporal_rows < 0 || desc_bytes < 1) return 0;
  return (size_t)make_layout(n_frames, n_pairs, stereo_rows, temporal_rows, desc_bytes).total;
}

int slamhip_frontend_pack_record(const slamhip_frontend_args *a, int frame_lo, int frame_hi, int pair_lo,
                                 int pair_hi, int64_t first_frame, int64_t first_pair,
                                 const int64_t *d_st_out_off, const int64_t *d_tm_out_off, int64_t stereo_rows,
                                 int64_t temporal_rows, uint8_t *d_record, size_t record_bytes, void *stream) {
  SLAMHIP_BEGIN();
  if (!a || frame_lo < 0 || frame_hi < frame_lo || frame_hi > a->n_frames || pair_lo < 0 || pair_hi < pair_lo ||
      pair_hi > (a->n_frames > 0 ? a->n_frames - 1 : 0) || stereo_rows < 0 || temporal_rows < 0)
    return SLAMHIP_EINVAL;
  const int nf = frame_hi - frame_lo, np = pair_hi - pair_lo;
  if (a->struct_size != sizeof(slamhip_frontend_args)) return SLAMHIP_EINVAL;
  const int D = a->desc_bytes, W = slamhip_packed_words(D);
  const RecordLayout L = make_layout(nf, np, stereo_rows, temporal_rows, D);
  if (!d_record || record_bytes < (size_t)L.total) return SLAMHIP_EWORKSPACE;
  if ((nf > 0 && !d_st_out_off) || (np > 0 && !d_tm_out_off) || !a->d_off_left || !a->d_st_left || !a->d_st_right ||
      !a->d_st_dist || !a->d_st_xyz || !a->d_st_pix_left || !a->d_st_pix_right || !a->d_st_desc || !a->d_tm_query ||
      !a->d_tm_train || !a->d_tm_dist || !a->d_tm_mask || !a->d_pose_refined || !a->d_pose_ransac ||
      !a->d_ransac_info)
    return SLAMHIP_EINVAL;
  const int per_segment = a->max_left > 0 ? a->max_left : 1;
  const dim3 grid(slamhip_ceil_div(per_segment, 256) < 8 ? slamhip_ceil_div(per_segment, 256) : 8,
                  nf + np + 1 < 65535 ? nf + np + 1 : 65535);
  hipLaunchKernelGGL(pack_record_kernel, grid, dim3(256), 0, (hipStream_t)stream, L, nf, np, frame_lo, pair_lo,
                     (long long)first_frame, (long long)first_pair, D, W, a->d_off_left,
                     (const long long *)d_st_out_off, (const long long *)d_tm_out_off, a->d_st_left, a->d_st_right,
                     a->d_st_dist, a->d_st_xyz, a->d_st_pix_left, a->d_st_pix_right, a->d_st_desc, a->d_tm_query,
                     a->d_tm_train, a->d_tm_dist, a->d_tm_mask, a->d_pose_refined, a->d_pose_ransac,
                     a->d_ransac_info, d_record);
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}

int slamhip_frontend_unpack_frame(const uint8_t *d_packet, int n_left, int n_right, int desc_bytes,
                                  int desc_stride, float *d_kp_left, float *d_kp_right, uint8_t *d_des_left,
                                  uint8_t *d_des_right, int32_t *d_off, void *stream) {
  SLAMHIP_BEGIN();
  if (!d_packet || n_left < 0 || n_right < 0 || desc_bytes < 1 || desc_stride < desc_bytes ||
      ((uintptr_t)d_packet & 3) || (n_left > 0 && (!d_kp_left || !d_des_left)) ||
      (n_right > 0 && (!d_kp_right || !d_des_right)))
    return SLAMHIP_EINVAL;
  const long long most = (long long)desc_bytes * (n_left > n_right ? n_left : n_right);
  const int blocks = most > 256 * 64 ? 64 : (most > 256 ? (int)slamhip_ceil_div((int)most, 256) : 1);
  hipLaunchKernelGGL(unpack_frame_kernel, dim3(blocks, 5), dim3(256), 0, (hipStream_t)stream, d_packet, n_left,
                     n_right, desc_bytes, desc_stride, d_kp_left, d_kp_right, d_des_left, d_des_right, d_off);
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}

}  // extern "C"
