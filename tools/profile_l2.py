"""L2 top-2 matcher on float descriptors: python3 tools/profile_l2.py [rows] [dim] [reps]
Prints the mean time of a call (HIP events on the launch stream); run under
rocprofv3 --kernel-trace --stats (or --pmc) for the per-kernel view."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import kernels  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    g = torch.Generator().manual_seed(0)
    t = torch.randn((n, dim), generator=g)
    q = t[torch.randperm(n, generator=g)] + 0.05 * torch.randn((n, dim), generator=g)
    q, t = q.cuda(), t.cuda()
    for _ in range(3):
        kernels.l2_knn2(q, t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        kernels.l2_knn2(q, t)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ops = 3.0 * n * n * dim
    print(f"l2_knn2 {n}x{n}x{dim}: {ms * 1e3:.1f} us per call, {ops / ms / 1e9:.1f} Tlane-op/s "
          f"(3 float32 operations per pair and dimension)")


if __name__ == "__main__":
    main()
