#!/usr/bin/env python3
"""Fills a feature cache from a KITTI odometry sequence on a machine that has OpenCV.

    python tools/extract_features.py --data .../sequences/05 --out CACHE [--akaze 2e-3]
                                     [--max-frames N] [--orb]

Detection is outside this package (it produces the hot path's inputs), and OpenCV
is not installed where this repository is built, so this script is the one place
that imports cv2; it does what the reference's frame loop does before matching
(backend/tracking/fill_database.py:102-109,141-145): read image_0/NNNNNN.png and
image_1/NNNNNN.png as grayscale (frontend/io/image_sequence.py:30-69) and call
AKAZE detectAndCompute with the given threshold on both
(frontend/vision/feature_extractor.py:18,39-44; default 2e-3 as run_slam.py:29).
Each frame becomes CACHE/NNNNNN.npz (slam_amd/backend/tracking/feature_cache.py),
which tools/run_sequence.py --features CACHE consumes.
"""
from __future__ import annotations

import argparse
import os
import sys
from pathlib import Path

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", required=True, help="sequence folder holding image_0/ and image_1/")
    ap.add_argument("--out", required=True, help="feature cache directory to write")
    ap.add_argument("--akaze", type=float, default=2e-3, help="AKAZE detector threshold")
    ap.add_argument("--orb", action="store_true", help="ORB (32-byte descriptors) instead of AKAZE")
    ap.add_argument("--max-frames", type=int, default=None)
    ap.add_argument("--extension", default=".png")
    args = ap.parse_args(argv)

    try:
        import cv2
    except ImportError:
        raise SystemExit("extract_features.py needs OpenCV (cv2); run it where the dataset and "
                         "opencv-python are, then copy the cache directory")
    import numpy as np
    from slam_amd.backend.tracking import feature_cache

    base = Path(args.data)
    if not base.is_dir():
        raise FileNotFoundError(f"Base directory not found: {base}")
    left = sorted((base / "image_0").glob(f"*{args.extension}"))
    right = sorted((base / "image_1").glob(f"*{args.extension}"))
    if len(left) != len(right):
        raise ValueError(f"Mismatched number of left ({len(left)}) vs right ({len(right)}) images")
    if args.max_frames is not None:
        left, right = left[:args.max_frames], right[:args.max_frames]
    det = cv2.ORB_create(nfeatures=4000) if args.orb else cv2.AKAZE_create(threshold=args.akaze)
    width = det.descriptorSize()
    for k, (pl, pr) in enumerate(zip(left, right)):
        imgs = [cv2.imread(str(p), cv2.IMREAD_GRAYSCALE) for p in (pl, pr)]
        if imgs[0] is None or imgs[1] is None:
            raise IOError(f"Could not read images at index {k}: {pl, pr}")
        out = []
        for img in imgs:
            kps, des = det.detectAndCompute(img, None)
            if des is None:
                des = np.empty((0, width), dtype=np.uint8)
            out.append((np.array([kp.pt for kp in kps], dtype=np.float32).reshape(-1, 2), des))
        feature_cache.save_frame(args.out, k, out[0][0], out[1][0], out[0][1], out[1][1])
        if k % 100 == 0:
            print(f"frame {k}: {len(out[0][0])} / {len(out[1][0])} features", flush=True)
    print(f"{len(left)} frames -> {args.out}")


if __name__ == "__main__":
    main()
