import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from slam_amd import _lib, kernels
from oracle import slam_oracle as so
for variant in (9, 10, 12):
  for splits in (1, 0, 3):
    _lib.lib.slamhip_tune(0, variant); _lib.lib.slamhip_tune(1, splits)
    rng = np.random.default_rng(8)
    for nq, nt, d in [(1, 1, 61), (63, 65, 61), (257, 129, 32), (600, 700, 61), (130, 513, 128), (64,256,64)]:
        q = rng.integers(0, 256, (nq, d), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, d), dtype=np.uint8)
        i, dd = kernels.hamming_knn2(q, t)
        i, dd = i.cpu().numpy(), dd.cpu().numpy()
        ei, ed = so.hamming_knn2(q, t)
        bad = np.argwhere((i != ei) | (dd != ed))
        print(variant, splits, (nq, nt, d), "bad:", len(bad))
        for (row, col) in bad[:6]:
            print("   q", row, "col", col, "got", i[row], dd[row], "want", ei[row], ed[row])
