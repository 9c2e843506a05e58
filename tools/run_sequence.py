#!/usr/bin/env python3
"""Runs the front-end (batched, or with --online one frame at a time) over a cached feature sequence and writes what the
reference's pose-graph stage loads (run_slam.py:80-89): the relative poses and
the per-frame link / match records.

    python tools/run_sequence.py --features DIR --out OUT [--calib calib.txt]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 \
        tools/run_sequence.py --features DIR --out OUT          # one rank per GPU

`--features DIR` holds NNNNNN.npz files (slam_amd/backend/tracking/feature_cache.py).
`--synthesize F` first fills DIR with F synthetic KITTI-shaped frames (no dataset
ships with this repository).  Rank 0 writes OUT/relative_Rts (pickle of an
(F-1, 3, 4) array, as fill_database.py:252-255 does) and OUT/frontend_records.npz,
from which slam_amd.backend.tracking.db_adapter.replay_into_db fills a TrackingDB
on the host afterwards.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--features", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--calib", default=None, help="KITTI calib.txt (default: KITTI 04-12 values)")
    ap.add_argument("--max-frames", type=int, default=None)
    ap.add_argument("--synthesize", type=int, default=0)
    ap.add_argument("--disp-min", type=float, default=1.2)
    ap.add_argument("--pix-thresh", type=float, default=4.0)
    ap.add_argument("--stereo-cross-check", action="store_true")   # as run_slam.py:34-35
    ap.add_argument("--frame-cross-check", action="store_true")
    ap.add_argument("--online", action="store_true",
                    help="one frame at a time (OnlineFrontend.push, the reference's own loop shape) instead of "
                         "one batched call; single GPU; same results")
    args = ap.parse_args(argv)

    import torch
    from slam_amd import CameraModel
    from slam_amd.backend.tracking import db_adapter, feature_cache, sharding
    from slam_amd.backend.tracking.frontend import TrackingFrontend

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cam = CameraModel.from_kitti(args.calib) if args.calib else CameraModel.kitti_like()

    if args.synthesize and rank == 0:
        from slam_amd.utils.synth import StereoSequence
        seq = StereoSequence(cam, n_frames=args.synthesize, n_features=1500, seed=5, clutter=0.6)
        for k in range(args.synthesize):
            f = seq.frame(k)
            feature_cache.save_frame(args.features, k, f.kp_left, f.kp_right, f.des_left, f.des_right)

    torch.cuda.set_device(local_rank)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
        dist.barrier()          # the synthesised cache is complete before anyone lists it

    frames = feature_cache.FeatureCache(args.features, args.max_frames)
    fe = TrackingFrontend(cam, disparity_min=args.disp_min, pixel_thresh=args.pix_thresh, seed=0,
                          stereo_cross_check=args.stereo_cross_check,
                          frame_cross_check=args.frame_cross_check)
    t0 = time.perf_counter()
    if args.online:
        if world > 1:
            raise SystemExit("--online runs on one GPU")
        from slam_amd.backend.tracking.online import OnlineFrontend, OnlineRecords
        online = OnlineFrontend(cam, disparity_min=args.disp_min, pixel_thresh=args.pix_thresh, seed=0,
                                stereo_cross_check=args.stereo_cross_check,
                                frame_cross_check=args.frame_cross_check)
        res = OnlineRecords(args.frame_cross_check)
        for k in range(len(frames)):
            pair = online.push(*frames[k])
            if pair is not None:
                res.add(pair)
    else:
        res = sharding.run_sharded(fe, frames, rank, world, group)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        os.makedirs(args.out, exist_ok=True)
        Rts = db_adapter.save_relative_Rts(res, os.path.join(args.out, "relative_Rts"))
        db_adapter.save_records(res, os.path.join(args.out, "frontend_records.npz"))
        print(f"{len(frames)} frames on {world} GPU(s) in {dt:.3f} s (file reads included): "
              f"{len(frames) / dt:.0f} frames/s; {len(Rts)} relative poses -> {args.out}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
