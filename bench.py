#!/usr/bin/env python3
"""Benchmark of the front-end hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): brute-force Hamming kNN (k=2) + ratio test
of 8192 x 8192 AKAZE-shaped 61-byte descriptors, inputs resident in HBM.  One
step = one full matcher call (row packing, top-2 kernel, split merge, ratio
test).  With N > 1 every rank (one process per GPU, RCCL only for the barrier
and the max-over-ranks of the time) matches its own independent descriptor
pair: pairs shard with no data-path collective, so scaling is weak.

Prints ONE JSON line on rank 0.  `value` = Hamming compares per second over all
ranks.  `roofline` prices the dominant kernel.  With the matrix-core matcher
(hamming_mfma_kernel, the default) a compare is a dot product of two rows of
+-1 FP4 values, one multiply-add per descriptor bit: 2 x 8 x 61 = 976 flop per
compare (1024 issued: rows are padded to 64 bytes), against the ~10 PFLOP/s
dense FP4 rate of v_mfma_scale_f32_32x32x64_f8f6f4.  With the popcount matcher
(SLAMHIP_MATCHER=popcount, hamming_top2_kernel) a compare is 2*ceil(D/4) = 32
int32 VALU lane-operations (16 v_xor_b32 + 16 v_bcnt_u32_b32), against the
78.6 Tlane-op/s of a SIMD that issues 32 lanes per clock; the measured issue
ceiling of that instruction mix (tools/issue_probe2.hip,
profiles/r1/issue_probe2.log: v_bcnt_u32_b32 issues at 16 lanes per clock) is
52.4 T and reported beside it.
`cpu_baseline` times the C/OpenMP restatement (oracle/hamming_oracle.c) of the
same workload on the host cores of this machine.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NQ = NT = 8192
DESC_BYTES = 61
RATIO = 0.8
CLOCK_HZ = 2.4e9
N_SIMD = 256 * 4
UNIFORM_PEAK_LANE_OPS = N_SIMD * 32 * CLOCK_HZ      # every VALU op at the 2-cycle wave64 rate: 78.6 T
# Issue cost per wave64 instruction measured with tools/issue_probe2.hip: v_xor_b32 (VGPR
# sources) 2 cycles like v_add/v_fma, v_bcnt_u32_b32 4 cycles like shifts, min/max, med3.
CYCLES_XOR, CYCLES_BCNT = 2, 4


def peak_compares_per_sec(desc_bytes):
    words = (desc_bytes + 3) // 4
    return N_SIMD * 64 * CLOCK_HZ / (words * (CYCLES_XOR + CYCLES_BCNT))


PEAK_LANE_OPS = peak_compares_per_sec(DESC_BYTES) * 2 * ((DESC_BYTES + 3) // 4)   # 52.4 T for 61 B
# HBM-side bytes per top-2 launch from the committed PMC passes (profiles/r1/pmc_matcher8k_v.txt):
# FETCH_SIZE 2327 KB x2 (gfx950 wide-read correction) + WRITE_SIZE 4096 KB.
PMC_TRAFFIC_BYTES = (2327 * 2 + 4096) * 1024
OPS_PER_COMPARE = 2 * ((DESC_BYTES + 3) // 4)
# Matrix-core matcher: one multiply-add per descriptor bit; the instruction stream pads rows to
# 8 / 16 / 32 dwords.  Peak: MI355X block-scaled FP4 MFMA, dense (no sparsity).
MFMA_PEAK_FLOPS = 10.0e15
# What the instruction sustains on this chip when nothing stands beside it and its operands are +-1 in every
# element, as descriptor bits are (profiles/r3/mfma_operands_clock.log: one issue per 33.5 cycles and SIMD at the
# 2.17-2.19 GHz the chip holds under it once warm; the clock follows the operand bits, all-zero operands run at 2.37).
MFMA_SUSTAINED_FLOPS = 2 * 32 * 32 * 64 * N_SIMD / (33.5 / 2.19e9)
# HBM-side bytes per matrix-core top-2 launch, same recipe (profiles/r3/pmc_matcher8k.txt, final build:
# FETCH_SIZE 2294 KB x2 + WRITE_SIZE 1024 KB; profiles/r2/pmc_matcher8k_mfma.txt had 2276).  The reads are the
# rows arriving in eight L2s, one per XCD: each holds two query blocks' rows (62 KB) and every train row (500 KB).
PMC_TRAFFIC_BYTES_MFMA = (2294 * 2 + 1024) * 1024


def flop_per_compare(desc_bytes, issued=False):
    if issued:
        words = 8 if desc_bytes <= 32 else (16 if desc_bytes <= 64 else 32)
        return 2 * 32 * words
    return 2 * 8 * desc_bytes


def matcher_roofline(mfma, compares, kernel_s, n_launch, desc_bytes=DESC_BYTES):
    """The `roofline` object of a stretch of top-2 kernel launches that did `compares`
    Hamming compares per launch and took `kernel_s` seconds per launch (HIP events on the
    launch stream)."""
    hbm = {"hbm_algorithmic_GBps": HBM_BYTES_PER_CALL / kernel_s / 1e9, "hbm_peak_GBps": 8000.0,
           "hbm_frac": HBM_BYTES_PER_CALL / kernel_s / 8e12}
    if mfma:
        achieved = compares * flop_per_compare(desc_bytes) / kernel_s
        issued = compares * flop_per_compare(desc_bytes, True) / kernel_s
        return {"bound": "mfma", "kernel": "hamming_mfma_kernel",
                "achieved": achieved / 1e12, "peak": MFMA_PEAK_FLOPS / 1e12, "unit": "TFLOP/s",
                "frac": achieved / MFMA_PEAK_FLOPS,
                "flop_per_compare": flop_per_compare(desc_bytes),
                "peak_source": "MI355X_MICROARCH.md: FP4 block-scaled MFMA ~10 PF dense "
                               "(v_mfma_scale_f32_32x32x64_f8f6f4, 32 cycles per SIMD at 2.4 GHz)",
                "achieved_issued": issued / 1e12, "frac_issued": issued / MFMA_PEAK_FLOPS,
                "sustained_instruction_peak": MFMA_SUSTAINED_FLOPS / 1e12,
                "frac_of_sustained_instruction_peak": achieved / MFMA_SUSTAINED_FLOPS,
                "sustained_source": "profiles/r3/mfma_operands_clock.log (tools/mfma_mix_probe 3): the bare instruction "
                                    "with +-1 operands, 33.5 cycles per issue and SIMD at 2.19 GHz",
                "issued_note": f"{flop_per_compare(desc_bytes, True)} flop per compare pass through "
                               "the matrix cores: rows are zero padded to 64 bytes",
                "kernel_us": kernel_s * 1e6, "launches_timed": n_launch, **hbm,
                "traffic": PMC_TRAFFIC_BYTES_MFMA,
                "traffic_source": "profiles/r3/pmc_matcher8k.txt (separate rocprofv3 --pmc passes of "
                                  "the same call, not read in this run): FETCH_SIZE x2 + WRITE_SIZE; "
                                  "includes the 1 MB of split results the kernel writes for the merge"}
    achieved = compares * OPS_PER_COMPARE / kernel_s
    return {"bound": "valu-int32", "kernel": "hamming_top2_kernel",
            "achieved": achieved / 1e12, "peak": UNIFORM_PEAK_LANE_OPS / 1e12,
            "unit": "Tlane-op/s", "frac": achieved / UNIFORM_PEAK_LANE_OPS,
            "ops_per_compare": OPS_PER_COMPARE,
            "peak_source": "MI355X_MICROARCH.md: 1024 SIMDs x 32 lanes per clock x 2.4 GHz",
            "frac_of_measured_issue_ceiling": achieved / PEAK_LANE_OPS,
            "measured_issue_ceiling": PEAK_LANE_OPS / 1e12,
            "measured_issue_ceiling_source": "profiles/r1/issue_probe2.log: 16 v_xor_b32 at 2 cycles + 16 "
                                             "v_bcnt_u32_b32 at 4 cycles per wave64 and train row",
            "kernel_us": kernel_s * 1e6, "launches_timed": n_launch, **hbm,
            "traffic": PMC_TRAFFIC_BYTES,
            "traffic_source": "profiles/r1/pmc_matcher8k_v.txt (separate rocprofv3 --pmc passes, not read "
                              "in this run); includes the 4 MB of split results the kernel writes"}
HBM_BYTES_PER_CALL = (NQ + NT) * DESC_BYTES + NQ * 16 + NQ   # read descriptors, write idx2/dist2/keep


def cpu_baseline(q: np.ndarray, t: np.ndarray, target_s: float = 10.0):
    """Times the CPU port (C + OpenMP, all host threads OpenMP gives us)."""
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    lib.oracle_hamming_knn2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_int]
    lib.oracle_hamming_knn2.restype = None
    lib.oracle_max_threads.restype = C.c_int
    threads = int(lib.oracle_max_threads())
    idx2 = np.empty((q.shape[0], 2), np.int32)
    dist2 = np.empty((q.shape[0], 2), np.int32)

    def once():
        lib.oracle_hamming_knn2(q.ctypes.data, q.shape[0], t.ctypes.data, t.shape[0], q.shape[1],
                                q.shape[1], idx2.ctypes.data, dist2.ctypes.data, 0)
    once()  # warm
    t0 = time.perf_counter()
    once()
    one = time.perf_counter() - t0
    reps = max(1, min(2000, int(target_s / max(one, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    dt = time.perf_counter() - t0
    return {
        "value": q.shape[0] * t.shape[0] * reps / dt,
        "unit": "hamming_compares/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{reps} x full {q.shape[0]}x{t.shape[0]}x{q.shape[1]}B kNN(k=2), "
                  f"oracle/hamming_oracle.c (popcnt, OpenMP {threads} threads of "
                  f"{os.cpu_count()} host CPUs), {dt:.2f} s wall; cv2 is not installed so the "
                  f"reference's cv2.BFMatcher call itself cannot be timed",
    }, idx2, dist2


def bench_frontend(args, rank, world, dist, red_dev="cuda", deferred=None):
    """Secondary measurement (BASELINE.json configs[2]/[3] shape): the whole per-frame
    front-end (stereo match, filter, triangulate, temporal match, adaptive P3P
    RANSAC, LM refinement) over `--frames` synthetic KITTI-shaped frames per GPU,
    features resident in HBM.  With N > 1 every rank processes its own frame
    range and the per-frame records are gathered once (all_gather over RCCL)."""
    from slam_amd import CameraModel
    from slam_amd.backend.tracking import sharding
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    from slam_amd.utils.synth import StereoSequence

    cam = CameraModel.kitti_like()
    F = args.frames
    seq = StereoSequence(cam, n_frames=F, n_features=1500, seed=100 + rank, clutter=0.6)
    frames = []
    for k in range(F):
        f = seq.frame(k)
        frames.append((f.kp_left, f.kp_right, f.des_left, f.des_right))
    fe = TrackingFrontend(cam, seed=rank)
    inputs = fe.upload(frames)

    def step():
        return fe.run(inputs)

    for _ in range(5):       # a pass is well under a millisecond: a few of them before the clock is read
        res = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.frontend_steps):
        res = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    def max_over_ranks(x):
        if dist is None:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())
    elapsed = max_over_ranks(elapsed)
    # The hand-over, timed on its own: every rank packs its records on the device (one
    # kernel) and ONE gather brings them to rank 0 -- once per sequence in a real run, so it
    # is not part of the per-pass rate above.
    gather = None
    if dist is not None:
        def pack():
            return sharding.pack_result(res, (0, F), (0, F - 1), rank * F, rank * (F - 1))
        blob = pack()
        sharding.gather_records(blob, dst=0)             # warm: communicator, buffers
        reps = 3
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            blob = pack()
        torch.cuda.synchronize()
        pack_s = max_over_ranks((time.perf_counter() - t0) / reps)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            got = sharding.gather_records(blob, dst=0)
        torch.cuda.synchronize()
        dist.barrier()
        gather_s = max_over_ranks((time.perf_counter() - t0) / reps)
        total = None
        if rank == 0:
            total = int(sum(g.numel() for g in got))
            head = got[world - 1][:40].cpu().numpy().view(np.int64)   # the last rank's header, as a check
            assert int(head[0]) == F and int(head[2]) == (world - 1) * F, head
        gather = {"pack_ms": pack_s * 1e3, "gather_ms": gather_s * 1e3, "record_bytes_this_rank": int(blob.numel()),
                  "bytes_on_rank0": total,
                  "GBps_into_rank0": None if not total else total * (world - 1) / world / gather_s / 1e9,
                  "note": "one variable-length byte record per rank (poses, links, inlier descriptors, matches, "
                          "masks), packed by one kernel, sizes by a tiny all_gather, payload by one gather "
                          "onto rank 0's GPU; done once per sequence, outside the timed passes"}
    if rank != 0:
        return None
    h = res.to_host()
    info = h["ransac_info"].reshape(-1, 8)
    # dominant kernel of the pass: the matcher's top-2 kernel (stereo pairs, then temporal
    # pairs); one more pass with its launches bracketed by HIP events on the launch stream
    from slam_amd import _lib
    _lib.check(_lib.lib.slamhip_profile_begin(64), "slamhip_profile_begin")
    fe.run(inputs)
    torch.cuda.synchronize()
    tot_ms = C.c_double(0.0)
    n_launch = _lib.lib.slamhip_profile_end(C.byref(tot_ms))
    stc = h["st_count"][:F].astype(np.float64)
    n_feat = np.array([[len(fr[0]), len(fr[1])] for fr in frames], dtype=np.float64)
    compares_pass = float((n_feat[:, 0] * n_feat[:, 1]).sum() + (stc[1:] * stc[:-1]).sum())
    mfma = _lib.lib.slamhip_matcher_kind() == 1
    roof = matcher_roofline(mfma, compares_pass, max(tot_ms.value, 1e-9) * 1e-3, n_launch)
    for k in ("hbm_algorithmic_GBps", "hbm_peak_GBps", "hbm_frac", "traffic", "traffic_source"):
        roof.pop(k, None)
    roof["kernel"] += "<best only> (k=1)"
    roof["compares_per_pass"] = compares_pass
    roof["kernel_us"] = tot_ms.value * 1e3
    roof["kernel_us_note"] = f"sum over the {n_launch} top-2 launches of one pass (stereo batch, temporal batch)"
    roof["share_of_pass"] = tot_ms.value * 1e-3 / (elapsed / args.frontend_steps)
    out = {
        "metric": "frontend_frames_per_sec",
        "value": F * args.frontend_steps * world / elapsed,
        "unit": "frames/s",
        "ms_per_frame": elapsed / (F * args.frontend_steps) * 1e3,
        "workload": f"{F} synthetic KITTI-shaped stereo frames per GPU (1226x370, 1500 AKAZE-like "
                    f"features per image, 61-byte descriptors), full front-end per frame, features given",
        "steps": args.frontend_steps,
        "ms_per_pass": elapsed / args.frontend_steps * 1e3,
        "hand_over": gather,
        "value_with_one_hand_over_per_pass": None if gather is None else
        F * world / (elapsed / args.frontend_steps + (gather["pack_ms"] + gather["gather_ms"]) * 1e-3),
        "stereo_inliers_mean": float(h["st_count"][:F].mean()),
        "temporal_matches_mean": float(info[:, 5].mean()),
        "ransac_inlier_ratio_mean": float((info[:, 1] / np.maximum(info[:, 5], 1)).mean()),
        "ransac_iterations_mean": float(info[:, 2].mean()),
        "pose_error_vs_truth_max_m": float(max(
            np.abs(res.pair(p)["Rt"][:, 3] - seq.relative_pose(p + 1)[:, 3]).max() for p in range(0, F - 1, 25))),
        "roofline": roof,
    }
    # accuracy of the chained trajectory against the synthetic truth (frame 0 = world), the
    # figure the reference prints beside its trajectories (utils/utils.py:165-169)
    from slam_amd.utils import utils as pu
    est = pu.chain_rel_to_abs_w2c(res.relative_Rts())
    gt = pu.relatives_from_abs_w2c_endpoints(np.stack(seq.poses[:F]), np.arange(F), np.zeros(F, int))[:, :3]
    out["trajectory"] = {"rmse_m": pu.trajectory_rmse(est, gt), "path_m": 0.8 * (F - 1),
                         "rot_err_deg_max": float(pu.rot_err_deg(est, gt).max()),
                         "note": "PnP odometry alone (no bundle adjustment, no loop closure) on synthetic "
                                 "KITTI-shaped frames with 0.5 px pixel noise"}
    # The same pass with the host on both sides: the sequence starts as one page-locked array
    # set in host memory (PackedFrames: what a feature cache hands over) and every result
    # array ends in host memory.  One pass, and the same in overlapped chunks (H2D of chunk
    # k+1 and D2H of chunk k-1 beside the pass over chunk k).
    from slam_amd.backend.tracking.feature_cache import PackedFrames
    packed = PackedFrames.from_frames(frames).pin()
    in_bytes = sum(a.nbytes for a in (packed.kp_left, packed.kp_right, packed.des_left, packed.des_right))

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    variants = {"one_pass": timed(lambda: fe.run(fe.upload(packed)).to_host())}
    for chunk in (64, 128, 256):
        if chunk < F - 1:
            variants[f"chunks_of_{chunk}_pairs"] = timed(lambda: fe.run_chunked(packed, chunk_pairs=chunk))
    for chunk in (128, 167):
        if chunk < F - 1:       # every upload queued back to back, no frame's stereo stage twice
            variants[f"streamed_{chunk}_pairs"] = timed(lambda: fe.run_streamed(packed, chunk_pairs=chunk))
    best = min(variants, key=variants.get)
    e2e = variants[best]
    out_bytes = sum(v.nbytes for v in fe.run(fe.upload(packed)).to_host().values())
    t0 = time.perf_counter()
    fe.run(fe.upload(frames)).to_host()
    tuples_s = time.perf_counter() - t0
    out["end_to_end"] = {"value": F / e2e, "unit": "frames/s", "scheme": best,
                         "includes": f"H2D of keypoints + descriptors from page-locked host arrays "
                                     f"({in_bytes / 1e6:.0f} MB), the pass, D2H of the result arrays "
                                     f"({out_bytes / 1e6:.0f} MB; inlier descriptors and pixels are rows of "
                                     f"the inputs the host gathers itself when asked)",
                         "seconds_per_pass": e2e,
                         "seconds_per_pass_by_scheme": variants,
                         "from_per_frame_tuples": {"value": F / tuples_s, "seconds_per_pass": tuples_s,
                                                   "includes": "host-side concatenation of 4 arrays per frame "
                                                               "first, and D2H of every result array"}}
    if args.envelope_frames >= 2:
        seq2 = StereoSequence(cam, n_frames=args.envelope_frames, n_features=(2000, 8000),
                              seed=500 + rank, clutter=0.6)
        fr2 = []
        for k in range(args.envelope_frames):
            f = seq2.frame(k)
            fr2.append((f.kp_left, f.kp_right, f.des_left, f.des_right))
        in2 = fe.upload(fr2)
        fe.run(in2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            r2 = fe.run(in2)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        i2 = r2.to_host()["ransac_info"].reshape(-1, 8)
        nf = np.array([len(f[0]) for f in fr2])
        out["envelope"] = {"value": args.envelope_frames / dt, "unit": "frames/s",
                           "workload": f"{args.envelope_frames} frames with the feature count per image "
                                       "drawn from U[2000, 8000] (ragged batch), otherwise as above",
                           "features_per_image_mean": float(nf.mean()),
                           "stereo_compares_per_frame_mean": float((nf.astype(np.float64) ** 2).mean()),
                           "temporal_matches_mean": float(i2[:, 5].mean()),
                           "ransac_iterations_mean": float(i2[:, 2].mean())}
    # The same pass on a sequence with the statistics the reference's README reports for its
    # KITTI-05 run (README.md:327 ~550 temporal matches per frame, :345 ~65 % PnP inliers): the
    # adaptive stop then needs ~60 iterations instead of ~400, one round of hypotheses.
    seq3 = StereoSequence(cam, n_frames=F, n_features=1500, seed=900 + rank, clutter=0.63, landmark_density=0.58)
    fr3 = []
    for k in range(F):
        f = seq3.frame(k)
        fr3.append((f.kp_left, f.kp_right, f.des_left, f.des_right))
    in3 = fe.upload(fr3)
    for _ in range(3):
        r3 = fe.run(in3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.frontend_steps):
        r3 = fe.run(in3)
    torch.cuda.synchronize()
    dt3 = (time.perf_counter() - t0) / args.frontend_steps
    h3 = r3.to_host()
    i3 = h3["ransac_info"].reshape(-1, 8)
    readme = {"metric": "frontend_frames_per_sec", "value": F / dt3, "unit": "frames/s", "ms_per_pass": dt3 * 1e3,
              "workload": f"{F} synthetic frames shaped like the reference's published KITTI-05 statistics "
                          "(README.md:327,345): 1500 features per image of which ~565 survive the stereo filter, "
                          "~565 temporal matches, ~65 % of them PnP inliers",
              "stereo_inliers_mean": float(h3["st_count"][:F].mean()),
              "temporal_matches_mean": float(i3[:, 5].mean()),
              "ransac_inlier_ratio_mean": float((i3[:, 1] / np.maximum(i3[:, 5], 1)).mean()),
              "ransac_iterations_mean": float(i3[:, 2].mean()),
              "pose_error_vs_truth_max_m": float(max(
                  np.abs(r3.pair(p)["Rt"][:, 3] - seq3.relative_pose(p + 1)[:, 3]).max() for p in range(0, F - 1, 25)))}
    out["readme_shape"] = readme
    if world == 1 and not args.no_cpu_baseline:
        def cpu_leg3():
            readme["cpu_baseline"] = cpu_baseline_frontend(cam, fr3, fe, target_s=6.0)
            readme["gpu_vs_cpu"] = readme["value"] / readme["cpu_baseline"]["value"]
        deferred.append(cpu_leg3)
    if world == 1 and not args.no_cpu_baseline:
        def cpu_leg():      # run by main() after every GPU section, so that the GPU work is contiguous
            out["cpu_baseline"] = cpu_baseline_frontend(cam, frames, fe)
            out["gpu_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        deferred.append(cpu_leg)
    return out


def bench_sequence(args, rank, world, dist, red_dev="cuda"):
    """BASELINE.json configs[3]: ONE sequence of `--sequence-frames` frames (2761 = KITTI-05) cut
    into contiguous frame ranges, one per rank, each with its halo frame (pair p needs frames p
    and p+1); every rank runs the front-end over its range, packs its records on the device, one
    gather brings them to rank 0's GPU, and rank 0 copies them to the host and merges them into
    the per-frame arrays the back-end's hand-over reads (run_slam.py:88-89).  Strong scaling:
    the job is the same at every N.  Features resident in HBM when the clock starts."""
    from slam_amd import CameraModel
    from slam_amd.backend.tracking import sharding
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    from slam_amd.utils.synth import StereoSequence

    cam = CameraModel.kitti_like()
    F = args.sequence_frames
    lo, hi, (plo, phi), (flo, fhi) = sharding.frames_for_rank(F, world, rank)
    seq = StereoSequence(cam, n_frames=F, n_features=1500, seed=55, clutter=0.63, landmark_density=0.58)
    frames = []
    for k in range(lo, hi):
        f = seq.frame(k)
        frames.append((f.kp_left, f.kp_right, f.des_left, f.des_right))
    fe = TrackingFrontend(cam, seed=5)
    inputs = fe.upload(frames) if frames else None
    phases = {"front_end": 0.0, "pack": 0.0, "gather": 0.0, "to_host_and_merge": 0.0}

    def lap(name, t0):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        phases[name] += t1 - t0
        return t1

    def one():
        t = time.perf_counter()
        if inputs is not None:
            res = fe.run(inputs, first_pair=plo)
            t = lap("front_end", t)
            blob = sharding.pack_result(res, (flo - lo, fhi - lo), (plo - lo, phi - lo), flo, plo)
        else:
            blob = torch.from_numpy(sharding.pack_records([], [], 0, 0, 61)).cuda()
        t = lap("pack", t)
        got = sharding.gather_records(blob, dst=0) if dist is not None else [blob]
        t = lap("gather", t)
        merged = None
        if rank == 0:
            merged = sharding.merge_records(sharding.records_to_host(got, reuse_buffer=True), F)
            lap("to_host_and_merge", t)
        return merged

    one()
    for k in phases:
        phases[k] = 0.0
    reps = 3
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        merged = one()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = (time.perf_counter() - t0) / reps
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank != 0:
        return None
    Rts = merged.relative_Rts
    err = max(float(np.abs(Rts[p][:, 3] - seq.relative_pose(p + 1)[:, 3]).max()) for p in range(0, F - 1, 97))
    return {"metric": "frontend_frames_per_sec", "value": F / elapsed, "unit": "frames/s", "scaling": "strong",
            "frames": F, "seconds": elapsed, "pairs_on_rank0": phi - plo,
            "workload": f"one {F}-frame synthetic sequence (KITTI-05 length, 1500 features per image, the "
                        "reference's published match / inlier statistics) cut into one contiguous frame range "
                        "per rank with a halo frame; front-end, device-side record packing, one gather onto rank "
                        "0, copy to the host and merge into the per-frame arrays of the hand-over",
            "rank0_phase_seconds": {k: v / reps for k, v in phases.items()},
            "record_bytes_total": merged.nbytes,
            "poses": int(len(Rts)), "pose_error_vs_truth_max_m": err}


def bench_online(args):
    """The reference's own calling pattern (fill_database.py:140: one frame at a time).
    online_pair_us: one front-end call per new frame, two frames per call, features resident
    (the device-side chain; online_pair_run_us: the same through TrackingFrontend.run).  online_push_us: OnlineFrontend.push of a new frame held in host
    arrays (that frame alone goes up, as one packet that one library launch spreads over its
    rows; one front-end call that carries the previous frame's stereo stage over) followed by pose(), i.e. features in host memory to relative pose in
    host memory, one frame at a time."""
    from slam_amd import CameraModel
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    from slam_amd.backend.tracking.online import OnlineFrontend
    from slam_amd.utils.synth import StereoSequence
    cam = CameraModel.kitti_like()
    n_seq = 24
    seq = StereoSequence(cam, n_frames=n_seq, n_features=1500, seed=3, clutter=0.63, landmark_density=0.58)
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(n_seq))]
    fe = TrackingFrontend(cam)
    inp = fe.upload(frames[:2])
    for _ in range(20):
        fe.run(inp)
    torch.cuda.synchronize()
    reps = 300
    t0 = time.perf_counter()
    for _ in range(reps):
        fe.run(inp)
    torch.cuda.synchronize()
    dt_run = (time.perf_counter() - t0) / reps
    # the same call with its argument block kept (what OnlineFrontend does from push to push):
    # no allocations, no argument conversion on the host, so the figure is the device's chain
    # and not this box's Python speed
    import ctypes as C
    from slam_amd import _lib
    res = fe.run(inp)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(20):
        _lib.check(_lib.lib.slamhip_frontend_batch(C.byref(res.args), stream), "slamhip_frontend_batch")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        _lib.lib.slamhip_frontend_batch(C.byref(res.args), stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    online = OnlineFrontend(cam)
    for f in frames[:4]:
        r = online.push(*f)
    r.pose()
    online.reset()
    online.push(*frames[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    inl = []
    for f in frames[1:]:
        Rt, info = online.push(*f).pose()
        inl.append(info[6] / max(info[5], 1))
    dt_push = (time.perf_counter() - t0) / (n_seq - 1)
    return {"online_pair_us": dt * 1e6, "frames_per_sec": 1.0 / dt, "online_pair_run_us": dt_run * 1e6,
            "online_push_us": dt_push * 1e6, "push_frames_per_sec": 1.0 / dt_push,
            "push_inlier_ratio_mean": float(np.mean(inl)),
            "note": "online_pair_us: one pair of 1500-feature frames per call (slamhip_frontend_batch on a kept "
                    "argument block; online_pair_run_us: through TrackingFrontend.run, which allocates the "
                    "outputs and builds the block per call), features resident: six "
                    "dependent launches (matcher, fused stereo tail, matcher, fused temporal tail, P3P + "
                    "consensus count, replay + mask + refinement), each one or a few workgroups running a "
                    "dependent chain; online_push_us: OnlineFrontend.push(frame in host arrays).pose(), "
                    "which includes the upload of the new frame, the call and the pose's way back"}


def bench_sweep(kernels, mfma=False):
    """Matcher alone at the other sizes the workload family covers (one pair,
    inputs resident): N in {2048, 4096, 8192} x D in {61 (AKAZE), 32 (ORB)}."""
    from slam_amd.utils.synth import matcher_workload
    rows = []
    for nbytes in (61, 32):
        for n in (2048, 4096, 8192):
            qn, tn = matcher_workload(n, n, nbytes, seed=11)
            q = kernels.to_device(qn, torch.uint8)
            t = kernels.to_device(tn, torch.uint8)
            call = kernels.BoundRatioMatch(q, t, RATIO)
            for _ in range(50):
                call()
            torch.cuda.synchronize()
            reps = 500
            t0 = time.perf_counter()
            for _ in range(reps):
                call()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            if mfma:    # whole call (pack + top-2 + merge) against the FP4 matrix-core peak
                frac = n * n * flop_per_compare(nbytes) / dt / MFMA_PEAK_FLOPS
            else:       # ... against the 2-cycle VALU rate
                frac = n * n * 2 * ((nbytes + 3) // 4) / dt / UNIFORM_PEAK_LANE_OPS
            rows.append({"n": n, "desc_bytes": nbytes, "us_per_call": dt * 1e6,
                         "compares_per_sec": n * n / dt, "frac_of_peak": frac})
    return rows


def bench_l2(kernels):
    """The matcher's other branch (float32 descriptors, L2 norm: what SuperPoint's 256
    floats get with matcher_type="bf"), one 8192 x 8192 pair resident in HBM."""
    n, dim, reps = 8192, 256, 10
    g = torch.Generator().manual_seed(0)
    t = torch.randn((n, dim), generator=g)
    q = (t[torch.randperm(n, generator=g)] + 0.05 * torch.randn((n, dim), generator=g)).cuda()
    t = t.cuda()
    for _ in range(3):
        kernels.l2_knn2(q, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        kernels.l2_knn2(q, t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return {"metric": "l2_compares_per_sec", "value": n * n / dt, "unit": "compares/s", "n": n, "dim": dim,
            "us_per_call": dt * 1e6, "dtype": "f32 (subtract, multiply, add; no FMA)",
            "lane_ops_per_sec": 3.0 * n * n * dim / dt,
            "frac_of_2_cycle_valu_rate": 3.0 * n * n * dim / dt / UNIFORM_PEAK_LANE_OPS}


def bench_loop_verify(args, rank, world, dist, red_dev="cuda"):
    """Secondary measurement (BASELINE.json configs[4] shape): descriptor matching +
    PnP-RANSAC consensus + refinement for ALL pairs of `--lc-keyframes` synthetic
    keyframes (one every ~14 m like the reference's keyframe rule), pair list
    dealt round-robin over the ranks, no collective on the data path."""
    from slam_amd import CameraModel
    from slam_amd.backend.optimizers.loop_verify import PairVerifier
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    from slam_amd.utils.synth import StereoSequence

    cam = CameraModel.kitti_like()
    KF, step_frames = args.lc_keyframes, 17
    seq = StereoSequence(cam, n_frames=KF * step_frames, n_features=1500, seed=7, clutter=0.6)
    frames = []
    for k in range(KF):
        f = seq.frame(k * step_frames)
        frames.append((f.kp_left, f.kp_right, f.des_left, f.des_right))
    st = TrackingFrontend(cam, n_iter_max=1).run(frames)      # stereo stage gives the DB content
    keyframes = []
    for k in range(KF):
        fr = st.frame(k)
        keyframes.append((fr["links"], fr["inlier_des_left"]))
    every = np.array([(i, n) for n in range(KF) for i in range(n)], dtype=np.int64)
    # samples are keyed by the pair's index in the whole list: N ranks together give the results of one
    pairs, ids = every[rank::world], np.arange(len(every))[rank::world]
    ver = PairVerifier(cam, seed=0).set_keyframes(keyframes)
    ver.verify(pairs, pair_ids=ids).verified_matches()          # warm-up: sizes the workspace
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    res = ver.verify(pairs, pair_ids=ids)
    vm = res.verified_matches()        # matches + inlier masks of the verified pairs, on the host
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    n_pairs_total = KF * (KF - 1) // 2
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank != 0:
        return None
    return {"metric": "loop_closure_pairs_verified_per_sec", "value": n_pairs_total / elapsed,
            "unit": "pairs/s", "seconds": elapsed, "keyframes": KF, "pairs": n_pairs_total,
            "features_per_keyframe_mean": float(np.mean([len(k[0]) for k in keyframes])),
            "verified_on_rank0": int(res["ok"].sum()), "pairs_on_rank0": int(len(pairs)),
            "ransac_iterations_mean_rank0": float(res["iterations"].mean()),
            "workload": "all-pairs keyframe verification (match k=1 + sort, P3P RANSAC <=1500 iters at "
                        "1.5 px with stereo consensus, LM refinement); time includes the per-call H2D "
                        "of the pair tables and D2H of every pair's flag / pose / counts and of the "
                        "matches and inlier masks of the verified pairs",
            "verified_match_rows_rank0": int(vm["off"][-1])}


def cpu_baseline_frontend(cam, frames, fe, target_s=10.0):
    """The same pass on the host cores: oracle/frontend_oracle.c, the C/OpenMP restatement of
    fill_database.py:140-240 (one frame, or one frame pair, per thread; popcount matcher, P3P
    + consensus one RANSAC iteration at a time like pnp.py:123-158, Gauss-Newton refinement),
    with the front-end's thresholds and iteration cap."""
    from oracle import frontend_c as fc
    kw = dict(stereo_thresh=fe.pixel_thresh, disparity_min=fe.disparity_min, pnp_thresh=fe.pixel_thresh,
              p_success=fe.p_success, n_iter_max=fe.n_iter_max, seed=1)
    fc.frontend_run(cam, frames[:17], **kw)         # warm: threads, pages
    t0 = time.perf_counter()
    reps = 0
    while reps < 1 or time.perf_counter() - t0 < target_s:
        c = fc.frontend_run(cam, frames, **kw)
        reps += 1
    dt = time.perf_counter() - t0
    n = len(frames)
    return {"value": n * reps / dt, "unit": "frames/s", "cores": int(fc.lib().oracle_max_threads()),
            "kind": "port",
            "sample": f"{reps} x the same {n} frames, oracle/frontend_oracle.c (C + OpenMP, one frame / frame "
                      f"pair per thread: popcount matcher, stereo filter, DLT, P3P RANSAC "
                      f"(mean {float(c['info'][:n - 1, 2].mean()):.0f} iterations, own sampler), Gauss-Newton "
                      f"refinement), {dt:.2f} s wall; cv2 is not installed, so the reference's own cv2 calls "
                      f"cannot be timed",
            "ransac_iterations_mean": float(c["info"][:n - 1, 2].mean()),
            "ransac_inliers_mean": float(c["info"][:n - 1, 1].mean())}


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher around it: start N copies of this
    script, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run
    sets them), before anything in this process has touched the GPU.  The children share
    this process's stdout, and only rank 0 prints, so its JSON line is the output.  Returns
    0 if every rank ended with 0, else the first failing rank's code; ranks still running
    30 s after another has failed are ended (by their own process ids)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", str(port)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(sys.argv[0])] + sys.argv[1:], env=env))
    rc, deadline, pending = 0, None, set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0:
                print(f"bench.py: rank {r} ended with code {code}", file=sys.stderr, flush=True)
                rc = rc or code
                deadline = deadline or time.monotonic() + 30.0
        if deadline is not None and time.monotonic() > deadline:
            for r in pending:
                procs[r].kill()
            deadline = time.monotonic() + 1e9
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--preheat", type=float, default=0.25,
                    help="seconds of untimed steps before the warm-up steps (clock ramp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames", type=int, default=500,
                    help="frames per GPU of the secondary front-end measurement (0 = skip)")
    ap.add_argument("--frontend-steps", type=int, default=20)
    ap.add_argument("--envelope-frames", type=int, default=60,
                    help="frames of the 2000-8000 features/image front-end measurement (0 = skip)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the matcher size sweep")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="skip the legs that issue the step on two / three streams (for kernel-trace runs: "
                         "launches that overlap there would enter the trace's average)")
    ap.add_argument("--section-timeout", type=float, default=300.0,
                    help="seconds the secondary sections (front-end, loop verification) may take together")
    ap.add_argument("--sequence-frames", type=int, default=2761,
                    help="frames of the one sequence sharded over the ranks by frame range (0 = skip)")
    ap.add_argument("--lc-keyframes", type=int, default=270,
                    help="keyframes of the secondary loop-closure verification measurement (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: be the launcher (this process never touches the GPU)
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU")
    # Rehearsal knobs (one-GPU box): SLAMHIP_BENCH_BACKEND=gloo and
    # SLAMHIP_BENCH_ONE_DEVICE=1 run all ranks on cuda:0 with a CPU-side barrier.
    backend = os.environ.get("SLAMHIP_BENCH_BACKEND", "nccl")
    if os.environ.get("SLAMHIP_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    # SLAMHIP_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, barriers,
    # all_reduce of the time, record gather) with a single rank as well, so that the
    # RCCL calls of an N-GPU run can be exercised on a one-GPU box.
    force_dist = os.environ.get("SLAMHIP_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    # how many ranks the collective backend really joined (a sum of ones over the group, on
    # device tensors with nccl = RCCL), which devices they sit on, and the library's version
    ranks_seen, devices, rccl_version = 1, None, None
    if dist is not None:
        ones = torch.ones(1, dtype=torch.int64, device=red_dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        prop = torch.cuda.get_device_properties(local_rank)
        mine = f"rank {rank}: cuda:{local_rank} {prop.name} {getattr(prop, 'uuid', '')}".strip()
        devices = [None] * world
        dist.all_gather_object(devices, mine)
        if backend == "nccl":
            try:
                rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:  # noqa: BLE001
                rccl_version = None

    from slam_amd import _lib, kernels
    from slam_amd.utils.synth import matcher_workload
    mfma = _lib.lib.slamhip_matcher_kind() == 1

    q_np, t_np = matcher_workload(NQ, NT, DESC_BYTES, seed=rank)
    q = kernels.to_device(q_np, torch.uint8)
    t = kernels.to_device(t_np, torch.uint8)

    idx2 = torch.empty((NQ, 2), dtype=torch.int32, device=q.device)
    dist2 = torch.empty((NQ, 2), dtype=torch.int32, device=q.device)
    keep = torch.empty(NQ, dtype=torch.uint8, device=q.device)

    # the call bound once (arguments checked and converted, workspace allocated): what a loop
    # that matches into the same buffers frame after frame does; one foreign call per step
    step = kernels.BoundRatioMatch(q, t, RATIO, out=(idx2, dist2, keep))

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # The GPU raises its clocks over the first tens of milliseconds of load (20 timed steps
    # right after start-up run 15 % slower than 3000): an untimed stretch of the same
    # steps first, so that a short --steps measures the kernel and not the ramp.
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.preheat:
        for _ in range(50):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        out = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # Same K steps again with the top-2 kernel bracketed by HIP events on its
    # launch stream: average duration of the dominant kernel.  Kept out of the
    # timed region above so the event records do not perturb `value`.
    # Two HIP events around a 15 us kernel cost 2-3 us themselves, so under the probe every call
    # launches the kernel PROBE_REPEAT times back to back between its two events (identical,
    # idempotent launches) and the average is taken over all of them; the same with one launch
    # per event pair is reported beside it.
    check = _lib.check
    PROBE_REPEAT = 8

    def probe(repeat):
        check(_lib.lib.slamhip_tune(10, repeat), "slamhip_tune")
        check(_lib.lib.slamhip_profile_begin(args.steps), "slamhip_profile_begin")
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        total_ms = C.c_double(0.0)
        n = _lib.lib.slamhip_profile_end(C.byref(total_ms))
        check(_lib.lib.slamhip_tune(10, 1), "slamhip_tune")
        if n < 0:
            check(n, "slamhip_profile_end")
        return (total_ms.value / max(n * max(repeat, 1), 1)) * 1e-3, n
    kernel_s_single, n_launch = probe(1)
    kernel_s, _ = probe(PROBE_REPEAT)
    # the same steps with the two events handed to the launch itself: they then hold the start and
    # the end of the dispatch, which is the duration a kernel trace reports (matrix-core kernel only)
    kernel_s_dispatch = None
    if mfma:
        try:
            kernel_s_dispatch = probe(0)[0]
        except Exception as e:  # noqa: BLE001  (the events recorded around the launch remain)
            print(f"bench: dispatch-event probe failed ({e}); using the recorded events", file=sys.stderr)
            check(_lib.lib.slamhip_tune(10, 1), "slamhip_tune")
    # Differential figure, no events at all: the same K steps timed by the host clock with the top-2
    # kernel launched twice in every step (identical, idempotent launches), minus the K steps with one
    # launch: what one more launch of the kernel costs inside the step, its launch gap included.
    def timed_steps(launches_per_step):
        check(_lib.lib.slamhip_tune(10, -launches_per_step if launches_per_step > 1 else 1), "slamhip_tune")
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        n = max(args.steps, 200)         # a short --steps would leave this difference of two times to the host's jitter
        t = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        check(_lib.lib.slamhip_tune(10, 1), "slamhip_tune")
        return dt / n
    kernel_s_differential = None
    if mfma:
        one = min(timed_steps(1) for _ in range(3))
        two = min(timed_steps(2) for _ in range(3))
        kernel_s_differential = two - one
    # The same steps issued on two and on three streams in turn (every stream with a bound call,
    # outputs and workspace of its own): what a caller gets who matches independent pairs and does not
    # wait for one before it issues the next -- the merge launch of one step and the start of the next
    # step's top-2 kernel then overlap.  Reported beside `value`, which stays the one-stream rate.
    def pipelined(n_streams):
        streams = [torch.cuda.Stream() for _ in range(n_streams)]
        calls = []
        for st in streams:
            with torch.cuda.stream(st):
                calls.append(kernels.BoundRatioMatch(q, t, RATIO))
        n = max(args.steps, 1000)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.3:          # new streams: their queues settle over the first milliseconds
            for i in range(100):
                calls[i % n_streams]()
            torch.cuda.synchronize()
        dt_p = float("inf")
        for _ in range(3):
            t_p = time.perf_counter()
            for i in range(n):
                calls[i % n_streams]()
            torch.cuda.synchronize()
            dt_p = min(dt_p, (time.perf_counter() - t_p) / n)
        same = all(torch.equal(a, b) for c in calls for a, b in zip(c.out, (idx2, dist2, keep)))
        return {"streams": n_streams, "us_per_step": dt_p * 1e6, "compares_per_sec": float(NQ) * NT / dt_p,
                "steps": n, "best_of": 3, "outputs_equal_the_one_stream_run": bool(same)}
    pipelined_runs = [pipelined(2), pipelined(3)] if dist is None and not args.no_pipelined else None
    # (under rocprofv3 the profiler owns the dispatch signals and these events read ~26 us: the
    # events recorded around the launch are then the headline again, and the line says so)
    by_dispatch = bool(mfma and kernel_s_dispatch is not None and 0.0 < kernel_s_dispatch <= kernel_s_single)

    line = None
    if rank == 0:
        compares = float(NQ) * NT
        line = {
            "metric": "hamming_compares_per_sec",
            "value": compares * args.steps * world / elapsed,
            "unit": "compares/s",
            "n_gpus": world,
            "dist_backend": backend if dist is not None else "none (single process)",
            "ranks_seen": ranks_seen,
            "rank_devices": devices,
            "rccl_version": rccl_version,
            "steps": args.steps,
            "warmup": args.warmup,
            "preheat_s": args.preheat,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("fp4 e2m1 (+-1 per descriptor bit) products, f32 accumulation of exact integers"
                      if mfma else "u32 (xor+popcount)"),
            "matcher": "mfma" if mfma else "popcount",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 8192x8192 AKAZE 61-byte descriptors, "
                                   "brute-force Hamming kNN(k=2) + ratio test, one pair per GPU",
                       "nq": NQ, "nt": NT, "desc_bytes": DESC_BYTES, "ratio": RATIO,
                       "parallelism": f"independent pairs x{world}"},
            # headline figures: the kernel's own duration, from HIP events that the launch itself fills
            # with the start and the end of its dispatch (hipExtLaunchKernelGGL; what rocprofv3
            # --kernel-trace reports for the same launches).  Beside it: events recorded on the stream
            # around ONE launch per call (the two records are inside that interval, an upper bound that
            # can exceed the whole step), and around several identical launches back to back.
            "roofline": dict(matcher_roofline(mfma, compares, kernel_s_dispatch if by_dispatch else kernel_s_single, n_launch),
                             kernel_us_method=(f"HIP events filled by the launch with the start and the end of its own "
                                               f"dispatch, one launch per call, {n_launch} calls in the step loop"
                                               if by_dispatch else
                                               f"HIP events on the launch stream around one launch of the kernel per "
                                               f"call, {n_launch} calls (includes the two event records)"),
                             kernel_us_by_dispatch_events=None if kernel_s_dispatch is None else kernel_s_dispatch * 1e6,
                             kernel_us_events_around_one_launch=kernel_s_single * 1e6,
                             frac_events_around_one_launch=matcher_roofline(mfma, compares, kernel_s_single, n_launch)["frac"],
                             events_around_one_launch_note="two event records on the stream around the launch; their own "
                                                           "cost (2-3 us) is inside the interval",
                             kernel_us_differential=None if kernel_s_differential is None else kernel_s_differential * 1e6,
                             differential_method="host-clock time of the steps with the kernel launched twice in each, "
                                                 "minus the steps as they are (best of three each): one more launch "
                                                 "inside the step, launch gap included, no events",
                             kernel_us_back_to_back=kernel_s * 1e6,
                             frac_back_to_back=matcher_roofline(mfma, compares, kernel_s, n_launch)["frac"],
                             back_to_back_method=f"{PROBE_REPEAT} identical launches between the two events"),
        }
        if pipelined_runs is not None:
            line["pipelined_over_streams"] = {
                "runs": pipelined_runs,
                "note": "the same 8192 x 8192 step issued on two / three HIP streams in turn, each with its own "
                        "bound call, outputs and workspace, timed by the host clock between two device "
                        "synchronisations: the rate for a caller that matches independent pairs without waiting "
                        "for one before issuing the next; `value` above is the one-stream rate"}
        # north_star words its target in popcount units (>= 50 % of the popcount-bound roofline):
        # the step's rate against the two popcount ceilings, whichever kernel family ran
        v = line["value"] / world
        line["vs_popcount_roofline"] = {
            "per_gpu_compares_per_sec": v,
            "vs_uniform_2_cycle_valu": v / (UNIFORM_PEAK_LANE_OPS / OPS_PER_COMPARE),
            "uniform_2_cycle_valu_compares_per_sec": UNIFORM_PEAK_LANE_OPS / OPS_PER_COMPARE,
            "uniform_source": "MI355X_MICROARCH.md: 1024 SIMDs x 32 lanes/clock x 2.4 GHz = 78.6 Tlane-op/s, "
                              "32 lane-ops (16 xor + 16 popcount) per 61-byte compare",
            "vs_survey_estimate": v / 1.2e12,
            "survey_estimate_compares_per_sec": 1.2e12,
            "survey_source": "256 CUs x 64 lanes x 2.4 GHz / 32 lane-ops per compare",
            "note": "above 1 means faster than any xor+popcount kernel can be on this chip: the matrix-core "
                    "matcher does the compare as an FP4 contraction instead" if mfma else
                    "the popcount kernel family"}
    # The secondary sections exchange records between ranks.  The primary measurement is
    # complete at this point: if a section fails on one rank (the others would then wait in
    # a collective for ever) or does not finish in time, rank 0 still prints the line, with
    # what went wrong, and every rank ends.
    # Exit code 3 = the primary line was printed but a secondary section hung or failed
    # (a GPU hang or exception must not reach the launcher as success).
    def abandon():
        if rank == 0:
            line["secondary_sections"] = f"abandoned after {args.section_timeout} s"
            print(json.dumps(line), flush=True)
        os._exit(3)
    watchdog = threading.Timer(args.section_timeout, abandon)
    watchdog.daemon = True
    watchdog.start()
    failed = {}
    frontend = loop_verify = None
    deferred = []           # CPU legs, run last
    try:
        if args.frames >= 2:
            frontend = bench_frontend(args, rank, world, dist, red_dev, deferred)
    except Exception as e:  # noqa: BLE001
        failed["frontend"] = repr(e)
    sequence = None
    try:
        if args.sequence_frames >= 2 and not failed:
            sequence = bench_sequence(args, rank, world, dist, red_dev)
    except Exception as e:  # noqa: BLE001
        failed["sequence"] = repr(e)
    try:
        if args.lc_keyframes >= 2 and not failed:
            loop_verify = bench_loop_verify(args, rank, world, dist, red_dev)
    except Exception as e:  # noqa: BLE001
        failed["loop_verify"] = repr(e)
    if failed and world > 1 and rank != 0:
        # leaving now would make the launcher stop rank 0 before it has printed: stay until
        # the watchdog ends this process
        while True:
            time.sleep(1.0)
    watchdog.cancel()

    if rank == 0:
        if world == 1 and not args.no_sweep:
            line["sweep"] = bench_sweep(kernels, mfma)
            line["l2_matcher"] = bench_l2(kernels)
        if frontend is not None:
            line["frontend"] = frontend
        if sequence is not None:
            line["sequence_sharded"] = sequence
        if world == 1 and args.frames >= 2 and not failed:
            line["online"] = bench_online(args)
        if loop_verify is not None:
            line["loop_verify"] = loop_verify
        if failed:
            line["secondary_sections_failed"] = failed
        for leg in deferred:
            leg()
        if world == 1 and not args.no_cpu_baseline:
            res = cpu_baseline(q_np, t_np)
            if res is not None:
                base, ci, cd = res
                line["cpu_baseline"] = base
                line["gpu_vs_cpu"] = line["value"] / base["value"]
                idx2, dist2, _ = out
                line["parity_vs_cpu_port"] = bool((idx2.cpu().numpy() == ci).all() and
                                                  (dist2.cpu().numpy() == cd).all())
        print(json.dumps(line), flush=True)
    if failed:
        sys.stdout.flush()
        os._exit(3)                          # the other ranks may be inside a collective
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
