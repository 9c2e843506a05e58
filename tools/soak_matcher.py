"""One-off soak of the matcher against the NumPy restatement: many seeded random
cases (sizes up to 3000 rows, now and then up to 9000 train rows, every packed width, both
kernel families, random split / block-size / split-fold knobs, batched pairs, the single-pair
call with the ratio test).  python tools/soak_matcher.py [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_amd import _lib, kernels
from oracle import slam_oracle as so

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
bad = 0
for case in range(cases):
    d = int(rng.integers(1, 129))
    _lib.lib.slamhip_tune(1, int(rng.choice([0, 0, 1, 2, 3, 5, 16, 64])))
    _lib.lib.slamhip_tune(3, int(rng.choice([0, 128, 256])))
    _lib.lib.slamhip_tune(0, int(rng.choice([-1, -1, 1, 9, 10])))
    _lib.lib.slamhip_tune(8, int(rng.choice([0, 1, 2])))
    _lib.lib.slamhip_tune(9, int(rng.choice([1, 1, 0])))
    _lib.lib.slamhip_tune(0, int(rng.choice([-1, -1, 1, 9, 10, 18])))
    n_pairs = int(rng.choice([1, 2, 5]))
    big = rng.random() < 0.2
    nqs = [int(rng.integers(0, 3000 if big else 400)) for _ in range(n_pairs)]
    nts = [int(rng.integers(0, 3000 if big else 400)) for _ in range(n_pairs)]
    if n_pairs == 1 and rng.random() < 0.15:
        nts = [int(rng.integers(3000, 9000))]
    hi = int(rng.choice([2, 3, 256]))
    qs = [rng.integers(0, hi, (n, d), dtype=np.uint8) for n in nqs]
    ts = [rng.integers(0, hi, (n, d), dtype=np.uint8) for n in nts]
    q_off = np.concatenate([[0], np.cumsum(nqs)]).astype(np.int32)
    t_off = np.concatenate([[0], np.cumsum(nts)]).astype(np.int32)
    if sum(nqs) == 0:
        continue
    i, dd = kernels.hamming_knn2_batched(np.vstack(qs), np.vstack(ts), q_off, t_off)
    i, dd = i.cpu().numpy(), dd.cpu().numpy()
    for p in range(n_pairs):
        ei, ed = so.hamming_knn2(qs[p], ts[p])
        if not (np.array_equal(i[q_off[p]:q_off[p + 1]], ei) and np.array_equal(dd[q_off[p]:q_off[p + 1]], ed)):
            bad += 1
            print(f"MISMATCH case {case} pair {p}: nq={nqs[p]} nt={nts[p]} D={d}", flush=True)
    if n_pairs == 1 and nts[0] > 0:
        ratio = float(rng.choice([0.7, 0.8, 0.99]))
        i, dd, keep = (x.cpu().numpy() for x in kernels.hamming_knn2_ratio(qs[0], ts[0], ratio))
        ei, ed = so.hamming_knn2(qs[0], ts[0])
        if not (np.array_equal(i, ei) and np.array_equal(dd, ed) and np.array_equal(keep.astype(bool), so.ratio_test(ed, ratio))):
            bad += 1
            print(f"MISMATCH (ratio call) case {case}: nq={nqs[0]} nt={nts[0]} D={d}", flush=True)
    if case % 50 == 0:
        print(f"case {case} ok so far, mismatches {bad}", flush=True)
print(f"done: {cases} cases, {bad} mismatching pairs")
sys.exit(1 if bad else 0)
