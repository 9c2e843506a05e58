"""The RCCL (torch.distributed backend "nccl") branch of the multi-GPU path, run
with the one rank a one-GPU box allows: process-group set-up bound to the device,
barrier, all_reduce of a device scalar, the variable-length all_gather / gather
of device-resident records, and run_sharded sent through the collective.  A
world of one moves no bytes between GPUs, but every call the N-GPU run makes is
made here with device tensors, so that run is not the first time they execute.
(Two ranks cannot share a GPU under RCCL, which rejects duplicate devices; the
two-rank tests use gloo.)"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rccl_worker(port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from slam_amd import CameraModel
        from slam_amd.backend.tracking import sharding
        from slam_amd.backend.tracking.frontend import TrackingFrontend
        from slam_amd.utils.synth import StereoSequence
        assert dist.get_backend() == "nccl"
        dist.barrier()
        tt = torch.tensor([1.25], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        assert float(tt.item()) == 1.25

        cam = CameraModel.kitti_like()
        seq = StereoSequence(cam, n_frames=7, n_features=400, seed=21)
        frames = []
        for k in range(7):
            f = seq.frame(k)
            frames.append((f.kp_left, f.kp_right, f.des_left, f.des_right))
        fe = TrackingFrontend(cam, n_iter_max=256, seed=5)
        direct = fe.run(frames)
        blob = sharding.pack_result(direct, (0, 7), (0, 6), 0, 0)
        assert blob.is_cuda

        got = sharding.gather_records(blob, dst=0)              # all_gather of sizes + gather
        assert len(got) == 1 and got[0].is_cuda and torch.equal(got[0], blob)
        every = sharding.all_gather_records(blob)               # padded all_gather
        assert len(every) == 1 and np.array_equal(every[0], blob.cpu().numpy())

        merged = sharding.run_sharded(fe, frames, 0, 1, force_collective=True)
        ok = True
        for f in range(7):
            a, c = merged.frame(f), direct.frame(f)
            for k in c:
                ok = ok and np.array_equal(a[k], c[k])
        for p in range(6):
            a, c = merged.pair(p), direct.pair(p)
            for k in ("query_idx", "train_idx", "dist", "mask", "Rt", "Rt_ransac"):
                ok = ok and np.array_equal(a[k], c[k])
            ok = ok and a["best_index"] == c["best_index"] and a["iterations"] == c["iterations"]
        ok = ok and np.array_equal(merged.relative_Rts, direct.relative_Rts())
        q.put(bool(ok))
    except Exception:  # pragma: no cover
        import traceback
        q.put(traceback.format_exc())
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_and_sharded_run_with_one_rank():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    try:
        result = q.get(timeout=240)
    finally:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    assert result is True, result


def test_shards_draw_the_samples_of_the_uncut_run():
    """Device sampler: a sequence cut into shards gets, pair for pair, the samples (and so the
    masks and poses) of the single run, because samples are keyed by the pair's index in the
    sequence, not in the batch."""
    from slam_amd import CameraModel
    from slam_amd.backend.tracking import sharding
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    from slam_amd.utils.synth import StereoSequence
    cam = CameraModel.kitti_like()
    seq = StereoSequence(cam, n_frames=9, n_features=400, seed=4)
    frames = []
    for k in range(9):
        f = seq.frame(k)
        frames.append((f.kp_left, f.kp_right, f.des_left, f.des_right))
    fe = TrackingFrontend(cam, n_iter_max=512, seed=11)
    whole = fe.run(frames)
    for world in (2, 3):
        for rank in range(world):
            lo, hi, (plo, phi), _ = sharding.frames_for_rank(len(frames), world, rank)
            if hi <= lo:
                continue
            part = fe.run(list(frames[lo:hi]), first_pair=plo)
            for p in range(plo, phi):
                a, b = part.pair(p - lo), whole.pair(p)
                assert a["best_index"] == b["best_index"] and a["iterations"] == b["iterations"]
                np.testing.assert_array_equal(a["mask"], b["mask"])
                np.testing.assert_array_equal(a["Rt"], b["Rt"])


def test_bench_multi_rank_path_runs_on_rccl_with_one_rank():
    """bench.py with its process group, barriers, all_reduce and record gather forced on at
    world size 1 (backend nccl): prints its line and exits 0."""
    env = dict(os.environ, SLAMHIP_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5",
                        "--warmup", "2", "--preheat", "0.05", "--frames", "12", "--frontend-steps", "1",
                        "--envelope-frames", "0", "--lc-keyframes", "6", "--sequence-frames", "31", "--no-sweep",
                        "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    import json
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["frontend"]["value"] > 0 and line["loop_verify"]["value"] > 0
    assert line["dist_backend"] == "nccl" and line["ranks_seen"] == 1
    assert line["sequence_sharded"]["poses"] == 30


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (here both
    on the one GPU of the box, over gloo: RCCL rejects two ranks on one device), relays rank 0's
    line and returns 0.  The line says how many ranks the collective backend joined and times
    the record hand-over on its own."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SLAMHIP_BENCH_BACKEND="gloo", SLAMHIP_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5",
                        "--warmup", "2", "--preheat", "0.05", "--frames", "12", "--frontend-steps", "1",
                        "--envelope-frames", "0", "--lc-keyframes", "6", "--sequence-frames", "31", "--no-sweep",
                        "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    import json
    rows = [x for x in r.stdout.strip().splitlines() if x.startswith("{")]
    assert len(rows) == 1, r.stdout[-2000:]
    line = json.loads(rows[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["dist_backend"] == "gloo"
    assert len(line["rank_devices"]) == 2
    ho = line["frontend"]["hand_over"]
    assert ho["gather_ms"] > 0 and ho["pack_ms"] > 0 and ho["bytes_on_rank0"] >= 2 * ho["record_bytes_this_rank"] * 0.5
    assert line["loop_verify"]["value"] > 0
    sq = line["sequence_sharded"]            # one 31-frame sequence cut in two: 30 poses come back merged
    assert sq["frames"] == 31 and sq["poses"] == 30 and sq["pairs_on_rank0"] == 15 and sq["scaling"] == "strong"
    assert sq["pose_error_vs_truth_max_m"] < 0.1


def test_bench_launcher_returns_the_failing_ranks_code():
    """A rank that fails makes the self-started run return non-zero (rank 1 raises in its
    front-end section; rank 0 prints the line and both end with code 3)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SLAMHIP_BENCH_BACKEND="gloo", SLAMHIP_BENCH_ONE_DEVICE="1", SLAMHIP_REHEARSE_FAIL_RANK="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rehearse_bench_failure.py"), "--gpus", "2",
                        "--steps", "5", "--warmup", "2", "--preheat", "0.05", "--frames", "4",
                        "--frontend-steps", "1", "--envelope-frames", "0", "--lc-keyframes", "0", "--sequence-frames", "0",
                        "--no-sweep",
                        "--no-cpu-baseline", "--section-timeout", "20"],
                       capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    assert r.returncode != 0, (r.stdout[-1500:], r.stderr[-3000:])
    import json
    rows = [x for x in r.stdout.strip().splitlines() if x.startswith("{")]
    assert len(rows) == 1 and json.loads(rows[0])["value"] > 0


def test_bench_reports_a_failed_section_with_a_nonzero_exit_code():
    env = dict(os.environ, SLAMHIP_REHEARSE_FAIL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rehearse_bench_failure.py"), "--gpus", "1",
                        "--steps", "5", "--warmup", "2", "--preheat", "0.05", "--frames", "4",
                        "--frontend-steps", "1", "--envelope-frames", "0", "--lc-keyframes", "0", "--sequence-frames", "0",
                        "--no-sweep",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    import json
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert "frontend" in line["secondary_sections_failed"] and line["value"] > 0
