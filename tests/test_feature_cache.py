"""CPU tests of the on-disk feature cache (one .npz per frame)."""
import numpy as np
import pytest

from slam_amd.backend.tracking import feature_cache as fc


def _frame(rng, nl, nr, d=61):
    return (rng.uniform(0, 1200, (nl, 2)).astype(np.float32), rng.uniform(0, 1200, (nr, 2)).astype(np.float32),
            rng.integers(0, 256, (nl, d), dtype=np.uint8), rng.integers(0, 256, (nr, d), dtype=np.uint8))


def test_roundtrip_slicing_and_laziness(tmp_path):
    rng = np.random.default_rng(0)
    frames = [_frame(rng, 10 + k, 7 + k) for k in range(5)]
    frames.append(_frame(rng, 0, 0))                       # an image with no features at all
    fc.write_sequence(str(tmp_path), frames)
    cache = fc.FeatureCache(str(tmp_path))
    assert len(cache) == 6
    for k in (0, 3, -1):
        for a, b in zip(cache[k], frames[k]):
            np.testing.assert_array_equal(a, b)
            assert a.dtype == b.dtype and a.flags.c_contiguous
    part = cache[2:4]
    assert len(part) == 2
    np.testing.assert_array_equal(part[1][2], frames[3][2])
    assert len(fc.FeatureCache(str(tmp_path), max_frames=3)) == 3
    with pytest.raises(IndexError):
        cache[6]


def test_keypoint_objects_and_validation(tmp_path):
    class KP:
        def __init__(self, x, y):
            self.pt = (x, y)
    rng = np.random.default_rng(1)
    _, _, dl, dr = _frame(rng, 3, 2)
    fc.save_frame(str(tmp_path), 0, [KP(1, 2), KP(3, 4), KP(5, 6)], [KP(7, 8), KP(9, 10)], dl, dr)
    kl, kr, _, _ = fc.FeatureCache(str(tmp_path))[0]
    assert kl.tolist() == [[1, 2], [3, 4], [5, 6]] and kr.tolist() == [[7, 8], [9, 10]]
    with pytest.raises(ValueError):
        fc.save_frame(str(tmp_path), 1, np.zeros((3, 2)), np.zeros((2, 2)), dl[:2], dr)   # rows != keypoints
    with pytest.raises(ValueError):
        fc.save_frame(str(tmp_path), 1, np.zeros((3, 2)), np.zeros((2, 2)), dl, dr[:, :32])  # D differs


def test_float_descriptors_are_refused_not_cast(tmp_path):
    """The reference matches uint8 descriptors under the Hamming norm and anything else under L2
    (descriptor_matcher.py:23-28); the batched paths are Hamming only and say so instead of
    casting float rows to bytes."""
    rng = np.random.default_rng(2)
    kl, kr, dl, dr = _frame(rng, 4, 3)
    fl, fr = rng.random((4, 256), dtype=np.float32), rng.random((3, 256), dtype=np.float32)
    with pytest.raises(TypeError, match="uint8"):
        fc.save_frame(str(tmp_path), 0, kl, kr, fl, fr)
    with pytest.raises(TypeError, match="uint8"):
        fc.PackedFrames.from_frames([(kl, kr, fl, fr)])
    with pytest.raises(TypeError, match="uint8"):
        fc.binary_descriptors(dl.astype(np.int32))
    assert fc.binary_descriptors(dl) is not None and fc.binary_descriptors(np.zeros((0, 61))).dtype == np.uint8
    fc.PackedFrames.from_frames([(kl, kr, dl, dr)])


def test_gaps_and_missing_directory(tmp_path):
    rng = np.random.default_rng(2)
    fc.save_frame(str(tmp_path), 0, *_frame(rng, 2, 2))
    fc.save_frame(str(tmp_path), 2, *_frame(rng, 2, 2))
    with pytest.raises(ValueError):
        fc.FeatureCache(str(tmp_path))
    with pytest.raises(FileNotFoundError):
        fc.FeatureCache(str(tmp_path / "nope"))
    (tmp_path / "empty").mkdir()
    with pytest.raises(FileNotFoundError):
        fc.FeatureCache(str(tmp_path / "empty"))


def test_extractor_script_says_what_it_needs_without_opencv(tmp_path):
    """tools/extract_features.py is the one place that imports cv2; where OpenCV is not
    installed it stops with a message instead of a traceback."""
    import importlib.util
    import os
    try:
        import cv2  # noqa: F401
        pytest.skip("OpenCV is installed here")
    except ImportError:
        pass
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("extract_features", os.path.join(root, "tools", "extract_features.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with pytest.raises(SystemExit, match="OpenCV"):
        mod.main(["--data", str(tmp_path), "--out", str(tmp_path / "cache")])


def test_packed_frames_roundtrip_slices_and_file(tmp_path):
    """One array set + offsets holds what the per-frame tuples hold: indexing gives the frames
    back, a slice is a PackedFrames of that frame range with re-based offsets, empty images are
    kept, and save / load keep everything."""
    from slam_amd.backend.tracking.feature_cache import PackedFrames
    rng = np.random.default_rng(4)
    frames = [_frame(rng, nl, nr) for nl, nr in [(5, 6), (0, 3), (7, 0), (4, 4)]]
    p = PackedFrames.from_frames(frames)
    assert len(p) == 4 and p.off_left.tolist() == [0, 5, 5, 12, 16] and p.off_right.tolist() == [0, 6, 9, 9, 13]
    for i, f in enumerate(frames):
        for a, b in zip(p[i], f):
            np.testing.assert_array_equal(a, b)
    q = p[1:4]
    assert len(q) == 3 and q.off_left.tolist() == [0, 0, 7, 11]
    for a, b in zip(q[2], frames[3]):
        np.testing.assert_array_equal(a, b)
    assert len(p[2:2]) == 0
    r = PackedFrames.load(p.save(str(tmp_path / "seq.npz")))
    for name in ("kp_left", "kp_right", "des_left", "des_right", "off_left", "off_right"):
        np.testing.assert_array_equal(getattr(r, name), getattr(p, name))
    with pytest.raises(ValueError):
        PackedFrames(p.kp_left[:3], p.kp_right, p.des_left, p.des_right, p.off_left, p.off_right)
    with pytest.raises(ValueError):
        p[0:4:2]
