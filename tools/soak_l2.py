"""One-off soak of the L2 matcher against the float32 restatement: many seeded random
cases (0..1500 rows, 1..300 dimensions, aligned and unaligned rows, few-valued
descriptors for exact ties).  python tools/soak_l2.py [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_amd import _lib
from oracle import slam_oracle as so

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(4242)
bad = 0
for case in range(cases):
    d = int(rng.integers(1, 301))
    if rng.random() < 0.5:
        d = (d + 3) // 4 * 4
    big = rng.random() < 0.25
    nq = int(rng.integers(1, 1500 if big else 300))
    nt = int(rng.integers(0, 1500 if big else 300))
    stride = d + int(rng.choice([0, 0, 1, 4, 7]))
    shift = int(rng.choice([0, 0, 1, 2, 4]))                    # floats the first row starts after
    levels = rng.choice([0, 1, 2])
    Q = rng.normal(size=(nq, stride)).astype(np.float32)
    T = rng.normal(size=(nt, stride)).astype(np.float32)
    if levels:
        Q, T = np.round(Q * levels) / levels, np.round(T * levels) / levels
        Q, T = Q.astype(np.float32), T.astype(np.float32)
    if nt > 2 and rng.random() < 0.5:
        T[nt - 1, :d] = T[0, :d]
        Q[0, :d] = T[0, :d]
    ei, ed = so.l2_knn2(np.ascontiguousarray(Q[:, :d]), np.ascontiguousarray(T[:, :d]))
    pad = np.zeros(shift, np.float32)
    dq = torch.from_numpy(np.concatenate([pad, Q.ravel()])).cuda()
    dt = torch.from_numpy(np.concatenate([pad, T.ravel()])).cuda()
    idx2 = torch.empty((nq, 2), dtype=torch.int32, device="cuda")
    dist2 = torch.empty((nq, 2), dtype=torch.float32, device="cuda")
    need = _lib.lib.slamhip_l2_knn2_workspace_bytes(nq, nt, d)
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device="cuda")
    rc = _lib.lib.slamhip_l2_knn2(dq.data_ptr() + 4 * shift, nq, dt.data_ptr() + 4 * shift, nt, d, stride,
                                  idx2.data_ptr(), dist2.data_ptr(), ws.data_ptr(), need, 0)
    torch.cuda.synchronize()
    if rc != 0 or not (np.array_equal(idx2.cpu().numpy(), ei) and np.array_equal(dist2.cpu().numpy(), ed)):
        bad += 1
        print(f"MISMATCH case {case}: rc={rc} nq={nq} nt={nt} d={d} stride={stride} shift={shift} levels={levels}", flush=True)
    if case % 25 == 0:
        print(f"case {case} ok so far, mismatches {bad}", flush=True)
print(f"done: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
