"""GPU parity tests of the L2 matcher (float32 descriptors; the norm the reference's
DescriptorMatcher selects for non-uint8 descriptors, descriptor_matcher.py:23-28):
HIP path against the float32 restatement, bit-exact on indices and distances."""
import numpy as np
import pytest
import torch

from oracle import slam_oracle as so

pytestmark = pytest.mark.gpu


def _desc(rng, n, d, levels=None):
    a = rng.normal(size=(n, d)).astype(np.float32)
    if levels:                     # few distinct values per dimension -> many exactly equal distances
        a = np.round(a * levels) / levels
    return a.astype(np.float32)


@pytest.mark.parametrize("levels", [None, 1])
def test_l2_knn2_equals_the_float32_restatement(levels):
    from slam_amd import kernels
    rng = np.random.default_rng(3 if levels else 4)
    for nq, nt, d in [(1, 1, 1), (5, 2, 3), (63, 65, 7), (64, 64, 32), (130, 257, 33), (300, 700, 128),
                      (257, 129, 256), (70, 1, 256), (33, 4000, 16)]:
        q, t = _desc(rng, nq, d, levels), _desc(rng, nt, d, levels)
        if nt > 3:
            t[3] = t[0]                                    # an exact duplicate: tie on the lowest index
            q[0] = t[0]
        i, dist = kernels.l2_knn2(q, t)
        torch.cuda.synchronize()
        ei, ed = so.l2_knn2(q, t)
        np.testing.assert_array_equal(i.cpu().numpy(), ei, err_msg=f"{nq}x{nt}x{d}")
        np.testing.assert_array_equal(dist.cpu().numpy(), ed, err_msg=f"{nq}x{nt}x{d}")
        if nt > 3:
            assert ei[0, 0] == 0 and ed[0, 0] == 0.0 and ei[0, 1] == 3


def test_l2_empty_inputs_and_limits():
    from slam_amd import _lib, kernels
    q = np.zeros((5, 8), np.float32)
    i, d = kernels.l2_knn2(q, np.zeros((0, 8), np.float32))
    assert (i.cpu().numpy() == -1).all() and (d.cpu().numpy() == -1).all()
    i, d = kernels.l2_knn2(np.zeros((0, 8), np.float32), q)
    assert i.shape == (0, 2)
    assert _lib.lib.slamhip_l2_knn2_workspace_bytes(10, 10, 5000) == 0       # above SLAMHIP_MAX_L2_DIM


@pytest.mark.parametrize("cross_check", [False, True])
def test_descriptor_matcher_with_float_descriptors(cross_check):
    """match / knn_match / ratio_test on float32 descriptors (what SuperPoint hands over
    with matcher_type='bf'): same matches in the same order as the restatement of
    cv2.BFMatcher(NORM_L2, crossCheck).match followed by the reference's stable sort."""
    from slam_amd import DescriptorMatcher
    rng = np.random.default_rng(11)
    base = _desc(rng, 400, 64)
    q = base[rng.permutation(400)[:300]] + rng.normal(0, 0.05, (300, 64)).astype(np.float32)
    t = np.vstack([base[:350] + rng.normal(0, 0.05, (350, 64)).astype(np.float32), _desc(rng, 60, 64)])
    q, t = q.astype(np.float32), t.astype(np.float32)
    m = DescriptorMatcher(cross_check=cross_check).match(q, t)
    eq, et, ed = so.bf_match_l2(q, t, cross_check=cross_check)
    assert [x.queryIdx for x in m] == eq.tolist()
    assert [x.trainIdx for x in m] == et.tolist()
    np.testing.assert_array_equal(np.array([x.distance for x in m], np.float32), ed)
    if cross_check:
        assert len(m) < 300
        return
    assert len(m) == 300                                   # one match per query row (database.py:332)
    raw = DescriptorMatcher(cross_check=False).knn_match(q, t, k=2)
    ei, edd = so.l2_knn2(q, t)
    assert [p[0].trainIdx for p in raw] == ei[:, 0].tolist() and [p[1].trainIdx for p in raw] == ei[:, 1].tolist()
    good, rejected = DescriptorMatcher.ratio_test(raw, ratio=0.7)
    keep = edd[:, 0].astype(np.float64) < 0.7 * edd[:, 1].astype(np.float64)
    assert [g.queryIdx for g in good] == np.flatnonzero(keep).tolist()
    assert len(good) + len(rejected) == 300 and 0 < len(good) < 300


def test_l2_equal_rounded_distances_from_different_sums():
    """Sums of squares 1 + 2 ulp and 1 + 3 ulp have the same rounded square root
    (1 + 1 ulp): the order among them is by train index, not by sum, and a second
    neighbour with the larger sum must not be displaced by a later row with the smaller."""
    from slam_amd import kernels
    z = np.float32(2.0 ** -11.5)
    hi = [1.0, 2.0 ** -11, z, 0.0]          # 1 + 2^-22 + ~2^-23 -> 1 + 3 ulp
    lo = [1.0, 2.0 ** -11, 0.0, 0.0]        # 1 + 2 ulp
    far = [3.0, 0.0, 0.0, 0.0]
    for rows, expect in [([hi, lo, lo, far], (0, 1)), ([far, lo, hi, lo], (1, 2)), ([hi, far, hi, lo], (0, 2)),
                         ([lo] + [far] * 300 + [hi, lo], (0, 301))]:
        t = np.array(rows, np.float32)
        q = np.zeros((3, 4), np.float32)
        acc = (t.astype(np.float32) ** 2)
        s = ((acc[:, 0] + acc[:, 1]) + acc[:, 2]) + acc[:, 3]
        assert len({float(np.sqrt(v)) for v in s[s < 2]}) == 1 and len({float(v) for v in s[s < 2]}) == 2
        i, d = kernels.l2_knn2(q, t)
        torch.cuda.synchronize()
        ei, ed = so.l2_knn2(q, t)
        np.testing.assert_array_equal(i.cpu().numpy(), ei)
        np.testing.assert_array_equal(d.cpu().numpy(), ed)
        assert tuple(ei[0]) == expect


def test_l2_strided_and_unaligned_rows():
    """Rows inside wider rows (row_stride > dim) and a base address that is not a
    multiple of 16 bytes: the kernel falls back from 16-byte loads; same result."""
    from slam_amd import _lib, kernels
    rng = np.random.default_rng(8)
    nq, nt, d, stride = 150, 333, 64, 71
    Q = rng.normal(size=(nq, stride)).astype(np.float32)
    T = rng.normal(size=(nt, stride)).astype(np.float32)
    ei, ed = so.l2_knn2(np.ascontiguousarray(Q[:, 3:3 + d]), np.ascontiguousarray(T[:, 3:3 + d]))
    dq, dt = torch.from_numpy(Q).cuda(), torch.from_numpy(T).cuda()
    idx2 = torch.empty((nq, 2), dtype=torch.int32, device="cuda")
    dist2 = torch.empty((nq, 2), dtype=torch.float32, device="cuda")
    need = _lib.lib.slamhip_l2_knn2_workspace_bytes(nq, nt, d)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    rc = _lib.lib.slamhip_l2_knn2(dq.data_ptr() + 12, nq, dt.data_ptr() + 12, nt, d, stride, idx2.data_ptr(),
                                  dist2.data_ptr(), ws.data_ptr(), need, 0)
    assert rc == 0
    torch.cuda.synchronize()
    np.testing.assert_array_equal(idx2.cpu().numpy(), ei)
    np.testing.assert_array_equal(dist2.cpu().numpy(), ed)
    # aligned rows of a multiple of four floats (the 16-byte path), many tiles per block
    q, t = _desc(rng, 300, 256), _desc(rng, 3000, 256)
    i, dd = kernels.l2_knn2(q, t)
    torch.cuda.synchronize()
    ei, ed = so.l2_knn2(q, t)
    np.testing.assert_array_equal(i.cpu().numpy(), ei)
    np.testing.assert_array_equal(dd.cpu().numpy(), ed)
