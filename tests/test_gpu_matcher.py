"""GPU parity tests of the Hamming matcher: HIP path (through the C ABI) vs the
CPU oracle, bit-exact on indices and distances."""
import numpy as np
import pytest
import torch

from oracle import slam_oracle as so

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["mfma", "popcount"])
def _reset_tuning(request):
    """Every test runs once per kernel family (tune knob 0: 8 = matrix-core kernel, 1 =
    popcount kernel); tests that pick a variant themselves override it."""
    from slam_amd import _lib
    _lib.lib.slamhip_tune(0, 8 if request.param == "mfma" else 1)
    yield
    _lib.lib.slamhip_tune(0, -1)
    _lib.lib.slamhip_tune(1, 0)
    _lib.lib.slamhip_tune(8, 2)
    _lib.lib.slamhip_tune(5, 0)
    _lib.lib.slamhip_tune(9, 1)


def _run(q, t):
    from slam_amd import kernels
    i, d = kernels.hamming_knn2(q, t)
    torch.cuda.synchronize()
    return i.cpu().numpy(), d.cpu().numpy()


def _knn2_raw(dq, q_off, nq, dt, t_off, nt, desc_bytes, stride):
    """slamhip_hamming_knn2 through ctypes on raw device byte buffers: rows start
    `*_off` bytes into the buffers and are `stride` bytes apart."""
    import ctypes as C
    from slam_amd import _lib
    idx2 = torch.full((nq, 2), -7, dtype=torch.int32, device="cuda")
    dist2 = torch.full((nq, 2), -7, dtype=torch.int32, device="cuda")
    need = _lib.lib.slamhip_hamming_knn2_workspace_bytes(nq, nt, 1, nq, nt, desc_bytes)
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device="cuda")
    rc = _lib.lib.slamhip_hamming_knn2(C.c_void_p(dq.data_ptr() + q_off), nq,
                                       C.c_void_p(dt.data_ptr() + t_off), nt, desc_bytes, stride,
                                       C.c_void_p(idx2.data_ptr()), C.c_void_p(dist2.data_ptr()),
                                       C.c_void_p(ws.data_ptr()), C.c_size_t(need),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    torch.cuda.synchronize()
    return idx2.cpu().numpy(), dist2.cpu().numpy()


def test_library_loaded_is_in_tree():
    from slam_amd import _lib
    assert _lib.LIB_PATH.endswith("slam_amd/libslamhip.so")


def test_golden_cases(golden):
    g = golden("oracle_hamming.npz")
    for name in ("akaze", "orb", "nt1", "nt2"):
        i, d = _run(g[f"{name}_q"], g[f"{name}_t"])
        np.testing.assert_array_equal(i, g[f"{name}_idx2"], err_msg=name)
        np.testing.assert_array_equal(d, g[f"{name}_dist2"], err_msg=name)


@pytest.mark.parametrize("variant", [0, 1, 4, 5, 8, 9, 10, 18])
@pytest.mark.parametrize("splits", [0, 1, 3, 64])
def test_every_kernel_variant_with_ties(variant, splits):
    from slam_amd import _lib
    _lib.lib.slamhip_tune(0, variant)
    _lib.lib.slamhip_tune(1, splits)
    rng = np.random.default_rng(variant * 10 + splits)
    for nq, nt, d in [(1, 1, 61), (63, 65, 61), (257, 129, 32), (600, 700, 61), (300, 40, 33),
                      (130, 513, 128), (70, 90, 1), (64, 256, 64), (5, 3, 17), (40, 33, 49), (2, 1, 15)]:
        q = rng.integers(0, 256, (nq, d), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, d), dtype=np.uint8)
        if nq % 2 == 0:  # few distinct byte values -> a lot of equal distances
            q &= 1
            t &= 1
        ei, ed = so.hamming_knn2(q, t)
        # knob 9: the matrix-core kernel reads the rows as they are (1) or packed by a launch in front (0)
        for raw in ((1, 0) if variant >= 8 else (1,)):
            _lib.lib.slamhip_tune(9, raw)
            i, dd = _run(q, t)
            np.testing.assert_array_equal(i, ei, err_msg=f"{nq}x{nt}x{d} raw={raw}")
            np.testing.assert_array_equal(dd, ed, err_msg=f"{nq}x{nt}x{d} raw={raw}")


@pytest.mark.parametrize("threads", [128, 256])
@pytest.mark.parametrize("splits", [0, 1, 5])
def test_both_block_sizes_with_ties_and_ragged_batches(threads, splits):
    """Matcher blocks of 256 and of 128 queries (tune knob 3; the plan picks 128 where
    it fills a pair's last block much better) give the same bit-exact results, on
    single pairs and on a ragged batch."""
    from slam_amd import _lib, kernels
    _lib.lib.slamhip_tune(3, threads)
    _lib.lib.slamhip_tune(1, splits)
    try:
        rng = np.random.default_rng(threads + splits)
        for nq, nt, d in [(1, 3, 61), (129, 300, 61), (610, 590, 61), (385, 77, 32), (128, 128, 128)]:
            q = rng.integers(0, 4, (nq, d), dtype=np.uint8)      # few values: many equal distances
            t = rng.integers(0, 4, (nt, d), dtype=np.uint8)
            i, dd = _run(q, t)
            ei, ed = so.hamming_knn2(q, t)
            np.testing.assert_array_equal(i, ei, err_msg=f"{nq}x{nt}x{d}")
            np.testing.assert_array_equal(dd, ed)
        sizes = [(610, 600), (0, 5), (131, 0), (257, 640), (5, 1)]
        qs = [rng.integers(0, 256, (a, 61), dtype=np.uint8) for a, _ in sizes]
        ts = [rng.integers(0, 256, (b, 61), dtype=np.uint8) for _, b in sizes]
        q_off = np.concatenate([[0], np.cumsum([len(x) for x in qs])]).astype(np.int32)
        t_off = np.concatenate([[0], np.cumsum([len(x) for x in ts])]).astype(np.int32)
        bi, bd = kernels.hamming_knn2_batched(np.concatenate(qs), np.concatenate(ts), q_off, t_off)
        bi, bd = bi.cpu().numpy(), bd.cpu().numpy()
        for p in range(len(sizes)):
            ei, ed = so.hamming_knn2(qs[p], ts[p])
            np.testing.assert_array_equal(bi[q_off[p]:q_off[p + 1]], ei, err_msg=f"pair {p}")
            np.testing.assert_array_equal(bd[q_off[p]:q_off[p + 1]], ed)
    finally:
        _lib.lib.slamhip_tune(3, 0)
        _lib.lib.slamhip_tune(1, 0)


@pytest.mark.parametrize("desc_bytes", [61, 32, 17])
def test_every_train_row_count_through_the_tile_scan(desc_bytes):
    """The tile scan is unrolled by two (16-word rows) or four (8-word rows) train rows
    with a tail, over LDS tiles of 128 rows: every row count from 1 to 140 and a few
    around further tile boundaries, one block scanning all of them (splits = 1), with
    many equal distances, in one ragged batched call."""
    from slam_amd import _lib, kernels
    _lib.lib.slamhip_tune(1, 1)
    rng = np.random.default_rng(desc_bytes)
    nts = list(range(1, 141)) + [255, 256, 257, 258, 259, 383, 384, 385, 386, 387]
    qs = [rng.integers(0, 3, (70, desc_bytes), dtype=np.uint8) for _ in nts]
    ts = [rng.integers(0, 3, (n, desc_bytes), dtype=np.uint8) for n in nts]
    q_off = (np.arange(len(nts) + 1) * 70).astype(np.int32)
    t_off = np.concatenate([[0], np.cumsum(nts)]).astype(np.int32)
    i, d = kernels.hamming_knn2_batched(np.vstack(qs), np.vstack(ts), q_off, t_off)
    i, d = i.cpu().numpy(), d.cpu().numpy()
    for p, n in enumerate(nts):
        ei, ed = so.hamming_knn2(qs[p], ts[p])
        np.testing.assert_array_equal(i[q_off[p]:q_off[p + 1]], ei, err_msg=f"{n} train rows")
        np.testing.assert_array_equal(d[q_off[p]:q_off[p + 1]], ed, err_msg=f"{n} train rows")


def test_seeded_random_shapes_strides_and_knobs():
    """Sixty seeded random cases through the C ABI: row counts 0..700, descriptor sizes
    1..128 bytes, rows at random byte offsets and strides, a random number of train
    splits and block size, single and batched calls; indices and distances equal the
    NumPy restatement exactly."""
    from slam_amd import _lib, kernels
    rng = np.random.default_rng(20240611)
    try:
        for case in range(60):
            d = int(rng.choice([1, 7, 16, 31, 32, 33, 61, 64, 65, 100, 128]))
            _lib.lib.slamhip_tune(1, int(rng.choice([0, 1, 2, 7, 64])))
            _lib.lib.slamhip_tune(3, int(rng.choice([0, 128, 256])))
            n_pairs = int(rng.choice([1, 1, 3, 6]))
            nqs = [int(rng.integers(0, 700)) for _ in range(n_pairs)]
            nts = [int(rng.integers(0, 700)) for _ in range(n_pairs)]
            hi = int(rng.choice([2, 4, 256]))          # few byte values -> many equal distances
            qs = [rng.integers(0, hi, (n, d), dtype=np.uint8) for n in nqs]
            ts = [rng.integers(0, hi, (n, d), dtype=np.uint8) for n in nts]
            if n_pairs == 1:
                stride = d + int(rng.integers(0, 9))
                off = int(rng.integers(0, 5))

                def strided(a):
                    # the buffer ends exactly at the last row's last byte
                    n = len(a)
                    buf = rng.integers(0, 256, off + (max(n, 1) - 1) * stride + d, dtype=np.uint8)
                    for r in range(n):
                        buf[off + r * stride: off + r * stride + d] = a[r]
                    return torch.from_numpy(buf).cuda()
                i, dd = _knn2_raw(strided(qs[0]), off, nqs[0], strided(ts[0]), off, nts[0], d, stride)
                ei, ed = so.hamming_knn2(qs[0], ts[0])
                np.testing.assert_array_equal(i, ei, err_msg=f"case {case}: {nqs[0]}x{nts[0]}x{d} stride {stride}")
                np.testing.assert_array_equal(dd, ed, err_msg=f"case {case}")
            else:
                q_off = np.concatenate([[0], np.cumsum(nqs)]).astype(np.int32)
                t_off = np.concatenate([[0], np.cumsum(nts)]).astype(np.int32)
                i, dd = kernels.hamming_knn2_batched(np.vstack(qs), np.vstack(ts), q_off, t_off)
                i, dd = i.cpu().numpy(), dd.cpu().numpy()
                for p in range(n_pairs):
                    ei, ed = so.hamming_knn2(qs[p], ts[p])
                    np.testing.assert_array_equal(i[q_off[p]:q_off[p + 1]], ei, err_msg=f"case {case} pair {p} D={d}")
                    np.testing.assert_array_equal(dd[q_off[p]:q_off[p + 1]], ed, err_msg=f"case {case} pair {p}")
    finally:
        _lib.lib.slamhip_tune(1, 0)
        _lib.lib.slamhip_tune(3, 0)


def test_maximum_train_rows(oracle_c):
    """The largest train set the packed keys allow (2^21 rows, the index field of a key):
    the best match planted in the very last row, an exact duplicate of another query's
    match earlier on (tie: the lower index wins), the rest random; against the C
    restatement.  One row more is refused."""
    import ctypes as C
    from slam_amd import _lib, kernels
    rng = np.random.default_rng(2021)
    nt, nq, d = 1 << 21, 70, 61
    t = rng.integers(0, 256, (nt, d), dtype=np.uint8)
    q = rng.integers(0, 256, (nq, d), dtype=np.uint8)
    q[0] = t[nt - 1]                    # distance 0 at the highest index a key can hold
    t[nt - 7] = t[5]
    q[1] = t[5]                         # two rows at distance 0: index 5 first, then 2^21 - 7
    i, dd = _run(q, t)
    ei, ed = oracle_c(q, t, threads=8)
    np.testing.assert_array_equal(i, ei)
    np.testing.assert_array_equal(dd, ed)
    assert i[0, 0] == nt - 1 and dd[0, 0] == 0
    assert i[1].tolist() == [5, nt - 7] and dd[1].tolist() == [0, 0]
    assert _lib.lib.slamhip_hamming_knn2_workspace_bytes(nq, nt, 1, nq, nt, d) > 0
    tq = torch.zeros((nq, d), dtype=torch.uint8, device="cuda")
    out = torch.empty((nq, 2), dtype=torch.int32, device="cuda")
    rc = _lib.lib.slamhip_hamming_knn2(C.c_void_p(tq.data_ptr()), nq, C.c_void_p(tq.data_ptr()), nt + 1, d, d,
                                       C.c_void_p(out.data_ptr()), C.c_void_p(out.data_ptr()),
                                       C.c_void_p(out.data_ptr()), C.c_size_t(1 << 30), None)
    assert rc == _lib.EUNSUPPORTED


def test_empty_inputs():
    q = np.zeros((5, 61), np.uint8)
    i, d = _run(q, np.zeros((0, 61), np.uint8))
    assert (i == -1).all() and (d == -1).all()
    i, d = _run(np.zeros((0, 61), np.uint8), q)
    assert i.shape == (0, 2)


def test_full_size_8k_vs_c_oracle(oracle_c):
    from slam_amd.utils.synth import matcher_workload
    q, t = matcher_workload(8192, 8192, 61, seed=0)
    i, d = _run(q, t)
    ei, ed = oracle_c(q, t, threads=8)
    np.testing.assert_array_equal(i, ei)
    np.testing.assert_array_equal(d, ed)
    # size-independent properties: a set matched against itself finds itself at
    # distance 0 (lowest index among exact duplicates); distances are ordered
    i2, d2 = _run(t, t)
    assert (d2[:, 0] == 0).all() and (i2[:, 0] <= np.arange(8192)).all()
    assert (d2[:, 0] <= d2[:, 1]).all()


def test_batched_ragged_pairs():
    from slam_amd import kernels
    rng = np.random.default_rng(9)
    nqs = [0, 5, 300, 64, 1, 777]
    nts = [10, 0, 257, 2, 1, 1500]
    qs = [rng.integers(0, 256, (n, 61), dtype=np.uint8) for n in nqs]
    ts = [rng.integers(0, 256, (n, 61), dtype=np.uint8) for n in nts]
    q_off = np.concatenate([[0], np.cumsum(nqs)]).astype(np.int32)
    t_off = np.concatenate([[0], np.cumsum(nts)]).astype(np.int32)
    i, d = kernels.hamming_knn2_batched(np.vstack(qs), np.vstack(ts), q_off, t_off)
    i, d = i.cpu().numpy(), d.cpu().numpy()
    for p in range(len(nqs)):
        ei, ed = so.hamming_knn2(qs[p], ts[p])
        np.testing.assert_array_equal(i[q_off[p]:q_off[p + 1]], ei, err_msg=f"pair {p}")
        np.testing.assert_array_equal(d[q_off[p]:q_off[p + 1]], ed, err_msg=f"pair {p}")


@pytest.mark.parametrize("n_pairs", [64, 67, 150])
@pytest.mark.parametrize("nbytes", [61, 32])
def test_many_ragged_pairs_dealt_to_the_xcds(n_pairs, nbytes):
    """Launches of 64 pairs or more deal the matrix-core matcher's blocks to the XCDs pair by pair
    (one-dimensional grid, (pair, split, query block) derived from the block id; ids of pairs that
    do not exist leave at once): pair counts that are and are not a multiple of eight, pairs of
    zero to several query blocks and stages, splits forced and not -- equal to the oracle and to
    the plain grid (tune knob 11 = 0)."""
    from slam_amd import kernels, _lib
    rng = np.random.default_rng(n_pairs + nbytes)
    nqs = [int(v) for v in rng.choice([0, 1, 31, 130, 257, 600], n_pairs)]
    nts = [int(v) for v in rng.choice([0, 1, 33, 129, 300, 700], n_pairs)]
    qs = [rng.integers(0, 256, (n, nbytes), dtype=np.uint8) for n in nqs]
    ts = [rng.integers(0, 256, (n, nbytes), dtype=np.uint8) for n in nts]
    for p in range(0, n_pairs, 5):            # ties: duplicated train rows
        if nts[p] > 40:
            ts[p][20:40] = ts[p][0:20]
    q_off = np.concatenate([[0], np.cumsum(nqs)]).astype(np.int32)
    t_off = np.concatenate([[0], np.cumsum(nts)]).astype(np.int32)
    q, t = np.vstack(qs), np.vstack(ts)
    try:
        for splits in (0, 3):
            _lib.lib.slamhip_tune(1, splits)
            got = {}
            for dealt in (1, 0):
                _lib.lib.slamhip_tune(11, dealt)
                i, d = kernels.hamming_knn2_batched(q, t, q_off, t_off)
                got[dealt] = (i.cpu().numpy(), d.cpu().numpy())
            np.testing.assert_array_equal(got[1][0], got[0][0])
            np.testing.assert_array_equal(got[1][1], got[0][1])
    finally:
        _lib.lib.slamhip_tune(1, 0)
        _lib.lib.slamhip_tune(11, 1)
    i, d = got[1]
    for p in range(n_pairs):
        ei, ed = so.hamming_knn2(qs[p], ts[p])
        np.testing.assert_array_equal(i[q_off[p]:q_off[p + 1]], ei, err_msg=f"pair {p}")
        np.testing.assert_array_equal(d[q_off[p]:q_off[p + 1]], ed, err_msg=f"pair {p}")


def test_ratio_test_and_sort_and_cross_check():
    from slam_amd import kernels
    from slam_amd.utils.synth import matcher_workload
    q, t = matcher_workload(1000, 900, 61, seed=2)
    i, d = kernels.hamming_knn2(q, t)
    for ratio in (0.5, 0.8, 0.99, 1.0):
        keep = kernels.ratio_test(d, ratio).cpu().numpy().astype(bool)
        np.testing.assert_array_equal(keep, so.ratio_test(d.cpu().numpy(), ratio))
    for ratio in (0.7, 0.99):   # fused call = knn2 followed by ratio_test
        fi, fd, fk = kernels.hamming_knn2_ratio(q, t, ratio)
        np.testing.assert_array_equal(fi.cpu().numpy(), i.cpu().numpy())
        np.testing.assert_array_equal(fd.cpu().numpy(), d.cpu().numpy())
        np.testing.assert_array_equal(fk.cpu().numpy().astype(bool), so.ratio_test(d.cpu().numpy(), ratio))
    fi, fd, fk = kernels.hamming_knn2_ratio(q[:100], t[:20], 0.8)   # single split: separate ratio pass
    np.testing.assert_array_equal(fk.cpu().numpy().astype(bool), so.ratio_test(fd.cpu().numpy(), 0.8))
    order = kernels.sort_by_distance(d[:, 0].contiguous()).cpu().numpy()
    np.testing.assert_array_equal(order, so.sort_by_distance(d[:, 0].cpu().numpy()))
    order2 = kernels.sort_by_distance(d, stride=2).cpu().numpy()
    np.testing.assert_array_equal(order2, order)
    mi, md = kernels.cross_check_match(q, t)
    t2q_i, t2q_d = so.hamming_knn2(t, q)
    ei, ed = so.cross_check_fold(t2q_i[:, 0], t2q_d[:, 0], q.shape[0])
    np.testing.assert_array_equal(mi.cpu().numpy(), ei)
    np.testing.assert_array_equal(md.cpu().numpy(), ed)


# tune knob 5: 0 = auto (radix passes in LDS where the segment fits), 1 = rank counting, 2 = bitonic network
@pytest.mark.parametrize("force_sort", [0, 1, 2])
def test_sort_large_and_segmented(force_sort):
    from slam_amd import _lib, kernels
    _lib.lib.slamhip_tune(5, force_sort)
    rng = np.random.default_rng(4)
    d = rng.integers(0, 489, 9000).astype(np.int32)
    d[rng.random(9000) < 0.01] = -1
    order = kernels.sort_by_distance(d).cpu().numpy()
    np.testing.assert_array_equal(order, so.sort_by_distance(d))
    sizes = [0, 3, 1025, 256, 1, 2047]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    dd = rng.integers(0, 30, int(off[-1])).astype(np.int32)
    o = kernels.sort_by_distance(dd, off=off).cpu().numpy()
    for s in range(len(sizes)):
        np.testing.assert_array_equal(o[off[s]:off[s + 1]], so.sort_by_distance(dd[off[s]:off[s + 1]]))
    # single segments (LDS kernels up to 8192 rows, rank counting beyond): powers of two, chunk
    # boundaries and their neighbours, absent distances (-1)
    for n in (1, 2, 3, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4097, 8191, 8192, 8193):
        d = rng.integers(0, 12, n).astype(np.int32)
        d[rng.random(n) < 0.05] = -1
        np.testing.assert_array_equal(kernels.sort_by_distance(d).cpu().numpy(), so.sort_by_distance(d),
                                      err_msg=f"n={n}")
    sizes = [8192, 0, 5000, 8192, 17] + [int(v) for v in rng.integers(0, 700, 140)]   # >= 128 segments: network
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    dd = rng.integers(0, 1025, int(off[-1])).astype(np.int32)         # 128-byte rows: distances to 1024
    o = kernels.sort_by_distance(dd, off=off).cpu().numpy()
    for s in range(len(sizes)):
        np.testing.assert_array_equal(o[off[s]:off[s + 1]], so.sort_by_distance(dd[off[s]:off[s + 1]]))


@pytest.mark.parametrize("cross_check", [False, True])
def test_descriptor_matcher_facade(cross_check):
    from slam_amd import DescriptorMatcher
    from slam_amd.utils.synth import matcher_workload
    q, t = matcher_workload(500, 450, 61, seed=7)
    m = DescriptorMatcher(cross_check=cross_check).match(q, t)
    qi, ti, di = so.bf_match(q, t, cross_check=cross_check)
    assert len(m) == len(qi)
    np.testing.assert_array_equal(m.query_idx, qi)
    np.testing.assert_array_equal(m.train_idx, ti)
    np.testing.assert_array_equal(m.distance, di.astype(np.float32))
    first = m[0]
    assert (first.queryIdx, first.trainIdx, first.imgIdx) == (int(qi[0]), int(ti[0]), 0)
    if not cross_check:
        assert len(m) == q.shape[0]  # one match per query row (asserted by the reference's DB)
        raw = DescriptorMatcher(cross_check=False).knn_match(q, t, k=2)
        good, rejected = DescriptorMatcher.ratio_test(raw, 0.8)
        idx2, dist2 = so.hamming_knn2(q, t)
        keep = so.ratio_test(dist2, 0.8)
        assert [g.queryIdx for g in good] == np.flatnonzero(keep).tolist()
        assert len(good) + len(rejected) == q.shape[0]


def test_facade_errors_like_reference():
    from slam_amd import DescriptorMatcher
    with pytest.raises(ValueError):
        DescriptorMatcher(cross_check=False, matcher_type="flann")


@pytest.mark.parametrize("desc_bytes,stride,offset", [(61, 61, 0), (61, 61, 1), (61, 67, 3), (32, 32, 2),
                                                      (64, 64, 1), (64, 70, 0), (1, 1, 0), (5, 9, 3),
                                                      (128, 128, 3), (100, 101, 2), (33, 40, 1), (17, 17, 0), (49, 49, 5), (16, 16, 1), (15, 15, 0),
                                                      # row pitches too wide for the query rows' copy through LDS
                                                      (61, 200, 2), (32, 150, 1), (128, 300, 0), (16, 1000, 3)])
def test_rows_at_any_alignment_and_stride(desc_bytes, stride, offset):
    """Through the C ABI with rows at an arbitrary byte offset and stride (the row
    packer fetches aligned dwords and funnel-shifts): results equal the oracle.
    The buffers end exactly at the last row's last byte."""
    import ctypes as C
    from slam_amd import _lib, kernels
    rng = np.random.default_rng(desc_bytes * 1000 + stride * 10 + offset)
    nq, nt = 333, 517
    q = rng.integers(0, 256, (nq, desc_bytes), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, desc_bytes), dtype=np.uint8)
    t[100] = t[7]; t[300] = t[7]; q[5] = t[7]                      # ties
    ref_i, ref_d = so.hamming_knn2(q, t)

    def strided(a):
        n = len(a)
        buf = rng.integers(0, 256, offset + (n - 1) * stride + desc_bytes, dtype=np.uint8)
        for r in range(n):
            buf[offset + r * stride: offset + r * stride + desc_bytes] = a[r]
        return torch.from_numpy(buf).cuda()

    dq, dt = strided(q), strided(t)
    idx2 = torch.empty((nq, 2), dtype=torch.int32, device="cuda")
    dist2 = torch.empty((nq, 2), dtype=torch.int32, device="cuda")
    need = _lib.lib.slamhip_hamming_knn2_workspace_bytes(nq, nt, 1, nq, nt, desc_bytes)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    idx2.fill_(-7); dist2.fill_(-7)
    rc = _lib.lib.slamhip_hamming_knn2(C.c_void_p(dq.data_ptr() + offset), nq,
                                       C.c_void_p(dt.data_ptr() + offset), nt, desc_bytes, stride,
                                       C.c_void_p(idx2.data_ptr()), C.c_void_p(dist2.data_ptr()),
                                       C.c_void_p(ws.data_ptr()), C.c_size_t(need),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    np.testing.assert_array_equal(idx2.cpu().numpy(), ref_i)
    np.testing.assert_array_equal(dist2.cpu().numpy(), ref_d)


@pytest.mark.parametrize("splits", [0, 2, 7, 64])
def test_splits_folded_inside_the_launch_equal_the_merge_launch(splits):
    """The matrix-core kernel folds its train splits itself (the last block to arrive of each
    query block reads the others' key pairs back; tune knob 8) or leaves them to a merge
    launch: same results bit for bit, with many equal distances, on single pairs, on a ragged
    batch with empty pairs, with the ratio test, and call after call in one workspace (the
    arrival counters must be back at zero every time), sizes changing in between."""
    from slam_amd import _lib, kernels
    rng = np.random.default_rng(splits)
    _lib.lib.slamhip_tune(0, 8)
    _lib.lib.slamhip_tune(1, splits)
    shapes = [(1, 1, 61), (700, 900, 61), (257, 4100, 32), (2049, 513, 61), (300, 8200, 61), (130, 513, 128),
              (4096, 4096, 61)]
    for rep in range(3):
        for nq, nt, d in shapes:
            q = rng.integers(0, 4, (nq, d), dtype=np.uint8)
            t = rng.integers(0, 4, (nt, d), dtype=np.uint8)
            out = {}
            for fold in (1, 0):
                _lib.lib.slamhip_tune(8, fold)
                i, dd, k = kernels.hamming_knn2_ratio(q, t, 0.8)
                torch.cuda.synchronize()
                out[fold] = (i.cpu().numpy(), dd.cpu().numpy(), k.cpu().numpy())
            for a, b in zip(out[1], out[0]):
                np.testing.assert_array_equal(a, b, err_msg=f"{nq}x{nt}x{d} rep {rep}")
            if nq * nt <= 700 * 900:
                ei, ed = so.hamming_knn2(q, t)
                np.testing.assert_array_equal(out[1][0], ei)
                np.testing.assert_array_equal(out[1][1], ed)
    sizes = [(610, 600), (0, 5), (131, 0), (257, 2640), (5, 1), (1300, 1290)]
    qs = [rng.integers(0, 256, (a, 61), dtype=np.uint8) for a, _ in sizes]
    ts = [rng.integers(0, 256, (b, 61), dtype=np.uint8) for _, b in sizes]
    q_off = np.concatenate([[0], np.cumsum([len(x) for x in qs])]).astype(np.int32)
    t_off = np.concatenate([[0], np.cumsum([len(x) for x in ts])]).astype(np.int32)
    res = {}
    for fold in (1, 0):
        _lib.lib.slamhip_tune(8, fold)
        bi, bd = kernels.hamming_knn2_batched(np.concatenate(qs), np.concatenate(ts), q_off, t_off)
        res[fold] = (bi.cpu().numpy(), bd.cpu().numpy())
    np.testing.assert_array_equal(res[1][0], res[0][0])
    np.testing.assert_array_equal(res[1][1], res[0][1])
    for p in range(len(sizes)):
        ei, ed = so.hamming_knn2(qs[p], ts[p])
        np.testing.assert_array_equal(res[1][0][q_off[p]:q_off[p + 1]], ei, err_msg=f"pair {p}")
        np.testing.assert_array_equal(res[1][1][q_off[p]:q_off[p + 1]], ed)


def test_bound_call_equals_the_array_native_call():
    """BoundRatioMatch (arguments checked and converted once) enqueues the same call: same
    results as hamming_knn2_ratio, call after call, into the tensors it was given."""
    from slam_amd import kernels
    rng = np.random.default_rng(8)
    q = rng.integers(0, 8, (900, 61), dtype=np.uint8)
    t = rng.integers(0, 8, (1100, 61), dtype=np.uint8)
    ei, ed, ek = (x.cpu().numpy() for x in kernels.hamming_knn2_ratio(q, t, 0.9))
    bound = kernels.BoundRatioMatch(q, t, 0.9)
    for _ in range(3):
        i, d, k = bound()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(i.cpu().numpy(), ei)
        np.testing.assert_array_equal(d.cpu().numpy(), ed)
        np.testing.assert_array_equal(k.cpu().numpy(), ek)
        i.zero_()
    with pytest.raises(ValueError):
        kernels.BoundRatioMatch(q, t, 0.9, out=(i, d, torch.empty(5, dtype=torch.uint8, device="cuda")))


def test_calls_in_flight_on_three_streams_keep_their_results_apart():
    """Three bound calls on three streams (different pairs, each with its own outputs and
    workspace), issued in turn without waiting: every call's result is its own pair's, as computed
    one at a time, while the launches of neighbouring calls overlap on the device."""
    from slam_amd import kernels
    rng = np.random.default_rng(21)
    pairs = [(rng.integers(0, 4, (nq, 61), dtype=np.uint8), rng.integers(0, 4, (nt, 61), dtype=np.uint8))
             for nq, nt in ((5000, 9000), (8192, 8192), (3000, 4100))]
    want = [tuple(x.cpu().numpy() for x in kernels.hamming_knn2_ratio(q, t, 0.85)) for q, t in pairs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in pairs]
    calls = []
    for st, (q, t) in zip(streams, pairs):
        with torch.cuda.stream(st):
            calls.append(kernels.BoundRatioMatch(q, t, 0.85))
    torch.cuda.synchronize()
    for rnd in range(40):
        for c in calls:
            c()
        if rnd == 20:
            for c in calls:                      # results wiped mid-way: the later calls must write them again
                for o in c.out:
                    o.zero_()
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    for c, w in zip(calls, want):
        for got, exp in zip(c.out, w):
            np.testing.assert_array_equal(got.cpu().numpy(), exp)
