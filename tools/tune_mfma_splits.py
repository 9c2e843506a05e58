#!/usr/bin/env python3
"""Whole-call time of the matrix-core matcher over forced train splits, for one n x n pair per size:
    python tools/tune_mfma_splits.py [desc_bytes]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib, kernels
from slam_amd.utils.synth import matcher_workload
D = int(sys.argv[1]) if len(sys.argv) > 1 else 61
for n in (1024, 1500, 2048, 4096, 8192, 16384):
    qn, tn = matcher_workload(n, n, D, seed=3)
    q = kernels.to_device(qn, torch.uint8); t = kernels.to_device(tn, torch.uint8)
    out = (torch.empty((n, 2), dtype=torch.int32, device=q.device), torch.empty((n, 2), dtype=torch.int32, device=q.device),
           torch.empty(n, dtype=torch.uint8, device=q.device))
    row = []
    for v in (9, 10):
        for s in (0, 1, 2, 4, 8, 16, 32, 64):
            _lib.lib.slamhip_tune(0, v); _lib.lib.slamhip_tune(1, s)
            for _ in range(30):
                kernels.hamming_knn2_ratio(q, t, 0.8, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                kernels.hamming_knn2_ratio(q, t, 0.8, out=out)
            torch.cuda.synchronize()
            row.append(f"G{v - 8}/s{s}: {(time.perf_counter() - t0) / 200 * 1e6:5.1f}")
    print(f"n={n:5d} D={D}: " + "  ".join(row), flush=True)
