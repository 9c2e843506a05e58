#!/usr/bin/env python3
"""Top-2 kernel duration (HIP events on the launch stream, inside the library) and whole-call
time of the matcher for a few sizes: python tools/matcher_kernel_times.py [desc_bytes]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib, kernels          # noqa: E402
from slam_amd.utils.synth import matcher_workload   # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 61
for n in (1024, 2048, 4096, 8192, 16384):
    qn, tn = matcher_workload(n, n, D, seed=3)
    q = kernels.to_device(qn, torch.uint8)
    t = kernels.to_device(tn, torch.uint8)
    out = (torch.empty((n, 2), dtype=torch.int32, device=q.device),
           torch.empty((n, 2), dtype=torch.int32, device=q.device),
           torch.empty(n, dtype=torch.uint8, device=q.device))
    for _ in range(30):
        kernels.hamming_knn2_ratio(q, t, 0.8, out=out)
    torch.cuda.synchronize()
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        kernels.hamming_knn2_ratio(q, t, 0.8, out=out)
    torch.cuda.synchronize()
    call_us = (time.perf_counter() - t0) / reps * 1e6
    _lib.check(_lib.lib.slamhip_profile_begin(reps), "profile_begin")
    for _ in range(reps):
        kernels.hamming_knn2_ratio(q, t, 0.8, out=out)
    torch.cuda.synchronize()
    ms = C.c_double(0.0)
    k = _lib.lib.slamhip_profile_end(C.byref(ms))
    kern_us = ms.value / max(k, 1) * 1e3
    print(f"n={n:6d} D={D}: call {call_us:7.1f} us  top-2 kernel {kern_us:7.1f} us  "
          f"{n * n / (kern_us * 1e-6):.3e} compares/s in the kernel", flush=True)
