#!/usr/bin/env python3
"""Prints the shader-clock stamps that a diagnostics build of the matrix-core matcher
(SLAMHIP_ABLATE=8, library named by SLAMHIP_LIB_PATH) leaves behind the last row of dist2:
    python tools/stamps.py [splits] [variant]"""
import ctypes as C, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib, kernels
from slam_amd.utils.synth import matcher_workload
splits = int(sys.argv[1]) if len(sys.argv) > 1 else 16
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = 8192
qn, tn = matcher_workload(n, n, 61, seed=3)
q = kernels.to_device(qn, torch.uint8); t = kernels.to_device(tn, torch.uint8)
_lib.lib.slamhip_tune(0, variant); _lib.lib.slamhip_tune(1, splits)
extra = 64 * 40 + 8
i2 = torch.zeros(2 * n + extra, dtype=torch.int32, device="cuda")
d2 = torch.zeros(2 * n + extra, dtype=torch.int32, device="cuda")
need = _lib.lib.slamhip_hamming_knn2_workspace_bytes(n, n, 1, n, n, 61)
ws = torch.empty(need, dtype=torch.uint8, device="cuda")
for _ in range(5):
    rc = _lib.lib.slamhip_hamming_knn2(q.data_ptr(), n, t.data_ptr(), n, 61, 61, i2.data_ptr(), d2.data_ptr(),
                                       ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
tail = d2.cpu().numpy()[2 * n:]
st = tail[:64 * 40].reshape(64, 40)
np.set_printoptions(linewidth=250)
for b in (0, 1, 5, 17, 31, 40, 63):
    v = st[b][:38]; v = v[v > 0]
    print("block", b, "stamps:", v, " deltas:", np.diff(v))
life = st[:, 38].astype(np.float64); real = st[:, 39].astype(np.float64)
print("wave lifetime: shader cycles median", np.median(life), " 100 MHz ticks median", np.median(real),
      " -> in-kernel clock %.0f MHz" % (np.median(life / np.maximum(real, 1)) * 100))
