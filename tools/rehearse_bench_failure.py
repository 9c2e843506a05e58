"""Rehearsal of bench.py's behaviour when a secondary section fails on one rank
(run under torch.distributed.run like bench.py, same arguments):
    SLAMHIP_REHEARSE_FAIL_RANK=1  -> bench_frontend raises on that rank
    SLAMHIP_REHEARSE_HANG_RANK=1  -> bench_frontend never returns on that rank
Rank 0 must still print the primary JSON line and every rank must end, with exit
code 3 (tests/test_gpu_rccl.py asserts it for the one-rank case)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

_orig = bench.bench_frontend


def _patched(args, rank, world, dist, *more, **kw):
    if str(rank) == os.environ.get("SLAMHIP_REHEARSE_FAIL_RANK"):
        raise RuntimeError("rehearsed failure")
    if str(rank) == os.environ.get("SLAMHIP_REHEARSE_HANG_RANK"):
        while True:
            time.sleep(1.0)
    return _orig(args, rank, world, dist, *more, **kw)


bench.bench_frontend = _patched
bench.main()
