#!/usr/bin/env python3
"""Block start/end spread of the matrix-core matcher from a SLAMHIP_ABLATE=64 build (library named
by SLAMHIP_LIB_PATH): first block start to last block end, against the kernel's event-timed duration.
    python tools/span.py [splits] [variant] [n]"""
import ctypes as C, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib, kernels
from slam_amd.utils.synth import matcher_workload
splits = int(sys.argv[1]) if len(sys.argv) > 1 else 16
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
qn, tn = matcher_workload(n, n, 61, seed=3)
q = kernels.to_device(qn, torch.uint8); t = kernels.to_device(tn, torch.uint8)
_lib.lib.slamhip_tune(0, variant); _lib.lib.slamhip_tune(1, splits)
extra = 8192
i2 = torch.zeros(2 * n + extra, dtype=torch.int32, device="cuda")
d2 = torch.zeros(2 * n + extra, dtype=torch.int32, device="cuda")
need = _lib.lib.slamhip_hamming_knn2_workspace_bytes(n, n, 1, n, n, 61)
ws = torch.empty(need, dtype=torch.uint8, device="cuda")
def call():
    rc = _lib.lib.slamhip_hamming_knn2(q.data_ptr(), n, t.data_ptr(), n, 61, 61, i2.data_ptr(), d2.data_ptr(),
                                       ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
for _ in range(300):
    call()
torch.cuda.synchronize()
_lib.check(_lib.lib.slamhip_profile_begin(100), "b")
for _ in range(100):
    call()
torch.cuda.synchronize()
ms = C.c_double(0.0)
k = _lib.lib.slamhip_profile_end(C.byref(ms))
tail = d2.cpu().numpy()[2 * n:].astype(np.int64) & 0xFFFFFFFF
qpb = 512 if variant == 18 else (256 if variant == 10 else 128)     # queries per block
nb = (n // qpb) * splits
rec = tail[:8 * nb].reshape(nb, 8)
se = rec[:, :2]
st, en = se[:, 0] - se[:, 0].min(), se[:, 1] - se[:, 0].min()
print(f"event-timed kernel {ms.value / k * 1e3:.2f} us; blocks {nb}: start spread {st.max() / 100:.2f} us, "
      f"first start -> last end {en.max() / 100:.2f} us, block lifetime median {np.median(en - st) / 100:.2f} "
      f"min {(en - st).min() / 100:.2f} max {(en - st).max() / 100:.2f} us, first end {en.min() / 100:.2f} us")
pro, loop = (rec[:, 4] - rec[:, 0]) / 100, (rec[:, 5] - rec[:, 4]) / 100
print(f"wave 0 of a block: prologue (to the first barrier) median {np.median(pro):.2f} us, stage loop {np.median(loop):.2f} us, "
      f"epilogue {np.median((rec[:, 1] - rec[:, 5]) / 100):.2f} us")
if rec[:, 6].any():
    print(f"  of the prologue: queries fetched and expanded after {np.median((rec[:, 6] - rec[:, 0]) / 100):.2f} us, first stage "
          f"in LDS after {np.median((rec[:, 7] - rec[:, 0]) / 100):.2f} us, barrier passed after {np.median(pro):.2f} us")
if nb <= 256 and variant == 18:
    w = tail[8 * 256:8 * 256 + 16 * nb].reshape(nb, 8, 2)
    if w.any():
        w0 = rec[:, 0][:, None]
        print("  per wave of a block, medians over blocks (us after the block's wave 0 started): start",
              np.round(np.median((w[:, :, 0] - w0) / 100, axis=0), 2), " its stage pieces in LDS",
              np.round(np.median((w[:, :, 1] - w0) / 100, axis=0), 2))
    w2 = tail[8 * 256 + 16 * 256:8 * 256 + 16 * 256 + 8 * nb].reshape(nb, 2, 4)
    if w2.any():
        for i, name in ((0, "wave 0"), (1, "wave 4")):
            d = (w2[:, i, :] - w2[:, i, :1]) / 100
            print(f"  {name}: set-up done (before the first row request) {np.median(d[:, 1]):.2f}, its stage pieces stored "
                  f"{np.median(d[:, 2]):.2f}, first query fragments expanded {np.median(d[:, 3]):.2f} us after its start")
life = (en - st) / 100
xcc = rec[:, 2] & 0xF
hw = rec[:, 3]
cu_key = xcc * 4096 + ((hw >> 13) & 7) * 256 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xF)
keys, counts = np.unique(cu_key, return_counts=True)
print("blocks per CU histogram:", dict(zip(*np.unique(counts, return_counts=True))), " distinct CUs:", len(keys))
per_cu = dict(zip(keys, counts))
nb_on_cu = np.array([per_cu[k] for k in cu_key])
for c in np.unique(nb_on_cu):
    m = nb_on_cu == c
    print(f"  blocks on CUs holding {c}: n={m.sum()} lifetime median {np.median(life[m]):.2f} max {life[m].max():.2f} us")
for x in range(8):
    m = xcc == x
    print(f"  xcc {x}: n={m.sum()} lifetime median {np.median(life[m]):.2f} max {life[m].max():.2f}, end max {en[m].max() / 100:.2f} us")
qb = n // qpb
bid = np.arange(nb)
print("  by split (median lifetime):", np.round([np.median(life[bid // qb == sp]) for sp in range(splits)], 2))
