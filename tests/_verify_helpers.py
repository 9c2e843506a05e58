"""Helpers of the pair-verification parity tests: a given idx4 sequence, the CPU loop run on it
with the device's minimal solutions, and the comparison of the two outcomes."""
import numpy as np

from oracle import pipeline_oracle as po
from oracle import slam_oracle as so


def _idx4_table(rng, counts, n_iter):
    """(pairs, n_iter, 4): what rng.choice(N, 4, replace=False) returns call after call
    (pnp.py:126), per pair with N matches."""
    table = np.zeros((len(counts), n_iter, 4), np.int32)
    for p, N in enumerate(counts):
        if N >= 4:
            table[p] = np.stack([rng.choice(int(N), 4, replace=False) for _ in range(n_iter)])
    return table


def _cpu_loop_on_device_poses(cam, kf_cand, kf_curr, samples, **kw):
    """The CPU restatement of consensus_pnp_match on a given idx4 sequence with the DEVICE's
    minimal solutions of those samples as its hypotheses (slamhip_p3p_solve on the same
    rows): what is left to compare is everything else -- projections, supporters, counts, the
    adaptive stop, the winner, its mask."""
    from slam_amd import kernels
    first = po.consensus_pnp_match(cam, kf_cand, kf_curr, samples=samples, **kw)
    if "P3D" not in first:
        return first, first
    Rts, valid = kernels.p3p_solve_indexed(first["P3D"], first["ptsL"], samples, cam.K)
    Rts, valid = Rts.cpu().numpy(), valid.cpu().numpy()
    calls = iter(range(len(samples)))

    def solver(obj4, img4):
        i = next(calls)
        return Rts[i] if valid[i] else None
    return first, po.consensus_pnp_match(cam, kf_cand, kf_curr, samples=samples, solver=solver, **kw)


def _assert_same_consensus(cam, res, p, sl, ref, thr=1.5):
    """Device result of pair p against a CPU run on the same samples and hypotheses: winner,
    iteration count, consensus count and mask (points within 1e-6 px of the threshold aside)."""
    assert res["best_index"][p] == ref["best_index"], (p, res["best_index"][p], ref["best_index"])
    assert res["iterations"][p] == ref["iterations"]
    assert res["n_iter_required"][p] == min(ref["n_iter_required"], 2 ** 31 - 1)      # the device keeps an int32
    diff = np.flatnonzero(res["mask"][sl] != ref["mask"])
    if len(diff):
        eL, eR = so.hypothesis_errors(cam.K, cam.M_right, ref["Rt_ransac"], ref["P3D"], ref["ptsL"], ref["ptsR"])
        assert all(min(abs(eL[k] - thr), abs(eR[k] - thr)) < 1e-6 for k in diff)
    else:
        assert res["best_count"][p] == ref["best_count"]
    if ref["best_index"] >= 0:
        np.testing.assert_allclose(res["Rt_ransac"][p], ref["Rt_ransac"], rtol=0, atol=1e-9)
