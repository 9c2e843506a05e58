"""The reference's frame loop written against the drop-in classes.

The integration is a change of three imports at the reference's call sites
(backend/tracking/fill_database.py:7-10): DescriptorMatcher, Triangulation and PnP come
from slam_amd instead.  This test is that frame loop, statement for statement
(fill_database.py:102-240, features given, default matchers), on the frames of
tests/golden/ref_build_flow.npz, and compares what it would hand to
TrackingDB.add_frame with what the reference's own build() handed over when it ran on
the same frames (oracle/make_golden_flow.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _create_links(features, kp_left, kp_right, matches):
    """TrackingDB.create_links(..., inliers=None, keep_match_order=True) (database.py:256-285)."""
    links = [(kp_left[m.queryIdx].pt[0], kp_right[m.trainIdx].pt[0],
              (kp_left[m.queryIdx].pt[1] + kp_right[m.trainIdx].pt[1]) / 2) for m in matches]
    return features[[m.queryIdx for m in matches]], np.array(links, np.float64).reshape(-1, 3)


def test_reference_frame_loop_on_the_drop_in_classes(golden):
    from slam_amd import CameraModel, DescriptorMatcher, KeyPoint, PnP, Triangulation, SOLVEPNP_ITERATIVE
    g = golden("ref_build_flow.npz")
    F, seed0 = int(g["n_frames"]), int(g["seed0"])
    cam = CameraModel.from_projections(g["P0"], g["P1"])
    stereo_matcher = DescriptorMatcher(cross_check=False, matcher_type="bf")        # :67-78
    frame_matcher = DescriptorMatcher(cross_check=False, matcher_type="bf")
    triang = Triangulation(cam, 1.2, 4.0)                                           # :81
    pnp = PnP(cam, 4.0)                                                             # :82

    def detect(k):
        kl = [KeyPoint(x, y, 1) for x, y in g[f"kp_left_{k}"]]
        kr = [KeyPoint(x, y, 1) for x, y in g[f"kp_right_{k}"]]
        return kl, g[f"des_left_{k}"], kr, g[f"des_right_{k}"]

    kp_L_prev, des_L_prev, kp_R_prev, des_R_prev = detect(0)
    matches_LR_prev = stereo_matcher.match(des_L_prev, des_R_prev)                  # :114
    inliers_LR_prev = triang.create_inlier_outlier_matches(matches_LR_prev, kp_L_prev, kp_R_prev)[0]   # :116-118
    X_prev = triang.triangulate_opencv(kp_L_prev, kp_R_prev, inliers_LR_prev)       # :121
    inlier_des_L_prev, links_prev = _create_links(des_L_prev, kp_L_prev, kp_R_prev, inliers_LR_prev)  # :124-131
    np.testing.assert_array_equal(links_prev, g["links_0"])
    np.testing.assert_array_equal(inlier_des_L_prev, g["inlier_des_0"])
    assert X_prev.dtype == np.float32 and X_prev.shape == (len(links_prev), 3)

    for idx in range(1, F):                                                         # :140
        kp_L, des_L, kp_R, des_R = detect(idx)
        matches_LR = stereo_matcher.match(des_L, des_R)                             # :155
        inliers_LR = triang.create_inlier_outlier_matches(matches_LR, kp_L, kp_R)[0]   # :158-160
        X_curr = triang.triangulate_opencv(kp_L, kp_R, inliers_LR)                  # :163
        inlier_des_L, links_curr = _create_links(des_L, kp_L, kp_R, inliers_LR)     # :166-169
        matches01 = frame_matcher.match(inlier_des_L_prev, inlier_des_L)            # :185
        X_corr = X_prev[[m.queryIdx for m in matches01]]                            # :189-190
        inlier_idx_L = [m.queryIdx for m in inliers_LR]                             # :193
        pix_L, pix_R = pnp.create_L_R_pixels(matches01, inliers_LR, inlier_idx_L, kp_L, kp_R, pair_0_or_1=1)  # :195-197
        mask = pnp.ransac_pnp_pair1(X_corr, pix_L, pix_R,                           # :200-202
                                    rng=np.random.default_rng(seed0 + idx - 1))
        Rt = pnp.solve_pnp(X_corr[mask], pix_L[mask], SOLVEPNP_ITERATIVE)            # :204-207

        np.testing.assert_array_equal(links_curr, g[f"links_{idx}"], err_msg=f"frame {idx}")
        np.testing.assert_array_equal(inlier_des_L, g[f"inlier_des_{idx}"])
        assert len(matches01) == len(inlier_des_L_prev)                             # database.py:332
        np.testing.assert_array_equal([m.queryIdx for m in matches01], g[f"tm_query_{idx}"])
        np.testing.assert_array_equal([m.trainIdx for m in matches01], g[f"tm_train_{idx}"])
        np.testing.assert_array_equal(np.array([m.distance for m in matches01], np.float32), g[f"tm_dist_{idx}"])
        ref_mask = g[f"tm_mask_{idx}"]
        assert mask.dtype == bool and (mask != ref_mask).sum() <= 1
        if (mask == ref_mask).all():
            np.testing.assert_array_equal(X_corr[mask], g[f"refine_obj_{idx - 1}"].astype(np.float32))
            np.testing.assert_array_equal(pix_L[mask], g[f"refine_img_{idx - 1}"])
            assert np.abs(Rt - g["relative_Rts"][idx - 1]).max() < 1e-6
        X_prev, inlier_des_L_prev = X_curr, inlier_des_L                            # :234-236


@pytest.mark.parametrize("batched", [False, True])
def test_tracking_pipeline_with_the_references_constructor_and_build(golden, batched, tmp_path):
    """slam_amd's TrackingPipeline takes the reference's constructor arguments and build()
    (fill_database.py:27-38,85-255).  Image reading and feature extraction stay outside (OpenCV in
    the reference): here an image is a frame number and the extractor returns that frame's
    keypoints and descriptors of tests/golden/ref_build_flow.npz.  What build() hands to
    add_frame equals what the reference's own build() handed over on the same frames; the poses
    equal the batched front-end's bit for bit whether the frames are pushed one at a time or
    run in one call; the two files are written as the reference writes them."""
    import pickle
    from oracle.tracking_db_oracle import TrackDBOracle
    from slam_amd import CameraModel, KeyPoint
    from slam_amd.backend.tracking import db_adapter
    from slam_amd.backend.tracking.fill_database import TrackingPipeline
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    g = golden("ref_build_flow.npz")
    F = int(g["n_frames"])
    cam = CameraModel.from_projections(g["P0"], g["P1"])

    class Images:                                      # ImageSequence: (left image, right image) per index
        def __len__(self):
            return F

        def __getitem__(self, k):
            return ("left", k), ("right", k)

    class Extractor:                                   # FeatureExtractor.detect_and_compute
        def detect_and_compute(self, img):
            side, k = img
            return [KeyPoint(x, y, 1) for x, y in g[f"kp_{side}_{k}"]], g[f"des_{side}_{k}"]

    class DB(TrackDBOracle):
        def serialize(self, path):
            with open(path, "wb") as fh:
                pickle.dump(self.summary(), fh)

    with pytest.raises(ValueError):
        TrackingPipeline(cam, Images(), 2e-3, 1.2, 4.0, False, False, "akaze", "flann", False, feature_extractor=Extractor())
    pipe = TrackingPipeline(cam, Images(), 2e-3, 1.2, 4.0, False, False, "akaze", "bf", True,
                            feature_extractor=Extractor(), db_factory=DB, link_factory=db_adapter.Link,
                            batched=batched, seed=3)
    db = pipe.build(max_frame=2600, out_path=str(tmp_path / "tracks.pkl"), rel_path=str(tmp_path / "relative_Rts"))
    frames = [(g[f"kp_left_{k}"], g[f"kp_right_{k}"], g[f"des_left_{k}"], g[f"des_right_{k}"]) for k in range(F)]
    want = TrackingFrontend(cam, disparity_min=1.2, pixel_thresh=4.0, seed=3).run(frames)
    np.testing.assert_array_equal(pipe.get_relative_Rts(), want.relative_Rts())
    with open(tmp_path / "relative_Rts", "rb") as fh:
        np.testing.assert_array_equal(pickle.load(fh), want.relative_Rts())
    assert (tmp_path / "tracks.pkl").exists()
    assert np.abs(pipe.get_relative_Rts() - g["relative_Rts"]).max() < 0.05      # the reference's own samples differ
    for k in range(F):
        links = db.all_frame_links(k)
        np.testing.assert_array_equal(np.asarray(links, np.float64).reshape(-1, 3), g[f"links_{k}"], err_msg=f"frame {k}")
    ref_db = db_adapter.replay_into_db(TrackDBOracle(), want)
    a, b = db.summary(), ref_db.summary()
    assert a.keys() == b.keys()
    for key in a:
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)
    assert len(pipe.matches_per_frame) == F - 1 and all(i <= m for i, m in zip(pipe.inliers_per_frame, pipe.matches_per_frame))


def test_tracking_pipeline_with_float_descriptors_runs_the_three_classes(golden, tmp_path):
    """Float descriptors (the reference's SuperPoint features with the "bf" matcher) are paired
    under the L2 norm (descriptor_matcher.py:23-28), which the fused Hamming front-end does not
    do: build() then walks the reference's loop over DescriptorMatcher / Triangulation / PnP.
    Here the float rows are the golden binary rows unpacked to one 0.0 / 1.0 per bit, so an L2
    distance is the square root of the Hamming distance and every match, stereo inlier, 3-D
    point and link must equal the Hamming run's; only the sampler differs (NumPy's, as the
    reference's), so masks may differ in borderline rows and poses agree to the tolerance of
    the comparison with the reference's own poses."""
    from oracle.tracking_db_oracle import TrackDBOracle
    from slam_amd import CameraModel
    from slam_amd.backend.tracking import db_adapter
    from slam_amd.backend.tracking.fill_database import TrackingPipeline
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    g = golden("ref_build_flow.npz")
    F = int(g["n_frames"])
    cam = CameraModel.from_projections(g["P0"], g["P1"])

    class Images:
        def __len__(self):
            return F

        def __getitem__(self, k):
            return ("left", k), ("right", k)

    class FloatExtractor:
        def detect_and_compute(self, img):
            side, k = img
            return g[f"kp_{side}_{k}"], np.unpackbits(g[f"des_{side}_{k}"], axis=1).astype(np.float32)

    for cross in (False, True):
        pipe = TrackingPipeline(cam, Images(), 2e-3, 1.2, 4.0, False, cross, "superpoint", "bf", True,
                                feature_extractor=FloatExtractor(), db_factory=TrackDBOracle,
                                link_factory=db_adapter.Link, seed=3)
        db = pipe.build(max_frame=F, out_path=None, rel_path=str(tmp_path / "relative_Rts"))
        frames = [(g[f"kp_left_{k}"], g[f"kp_right_{k}"], g[f"des_left_{k}"], g[f"des_right_{k}"]) for k in range(F)]
        want = TrackingFrontend(cam, disparity_min=1.2, pixel_thresh=4.0, seed=3, frame_cross_check=cross).run(frames)
        for k in range(F):
            links = np.asarray(db.all_frame_links(k), np.float64).reshape(-1, 3)
            np.testing.assert_array_equal(links, g[f"links_{k}"], err_msg=f"frame {k}")
        got_Rts = pipe.get_relative_Rts()
        assert got_Rts.shape == (F - 1, 3, 4)
        assert np.abs(got_Rts - want.relative_Rts()).max() < 0.05
        assert np.abs(got_Rts - g["relative_Rts"]).max() < 0.05
        assert len(pipe.matches_per_frame) == F - 1
        for p in range(F - 1):
            assert pipe.matches_per_frame[p] == len(want.pair(p)["mask"])
            assert abs(pipe.inliers_per_frame[p] - int(want.pair(p)["mask"].sum())) <= 0.05 * len(want.pair(p)["mask"])

    class NoRightFeaturesInFrame1(FloatExtractor):       # frame 1 keeps no stereo inlier: its two pairs have no match
        def detect_and_compute(self, img):
            kp, des = super().detect_and_compute(img)
            return (kp[:0], des[:0]) if img == ("right", 1) else (kp, des)

    pipe = TrackingPipeline(cam, Images(), 2e-3, 1.2, 4.0, False, False, "superpoint", "bf", True,
                            feature_extractor=NoRightFeaturesInFrame1(), db_factory=TrackDBOracle,
                            link_factory=db_adapter.Link, seed=3)
    db = pipe.build(max_frame=F, out_path=None, rel_path=None)
    assert len(db.all_frame_links(1)) == 0 and pipe.matches_per_frame[:2] == [0, 0]
    identity = np.hstack([np.eye(3), np.zeros((3, 1))])
    np.testing.assert_array_equal(pipe.get_relative_Rts()[:2], [identity, identity])
    if F > 3:
        assert np.abs(pipe.get_relative_Rts()[2] - g["relative_Rts"][2]).max() < 0.05


@pytest.mark.parametrize("batched", [False, True])
def test_tracking_pipeline_survives_a_frame_without_stereo_inliers(golden, batched):
    """A frame whose right image gave no features keeps no stereo inlier, so the frames before
    and after it have nothing to match against.  The reference's loop fails there
    (fill_database.py:200-207 indexes with a missing mask, database.py:332 wants one match per
    previous row); here the two pairs are recorded without matches and with the identity as
    their pose, add_frame gets the densified hand-over, and the sequence goes on."""
    from oracle.tracking_db_oracle import TrackDBOracle
    from slam_amd import CameraModel
    from slam_amd.backend.tracking import db_adapter
    from slam_amd.backend.tracking.fill_database import TrackingPipeline
    g = golden("ref_build_flow.npz")
    F = int(g["n_frames"])
    cam = CameraModel.from_projections(g["P0"], g["P1"])

    class Images:
        def __len__(self):
            return F

        def __getitem__(self, k):
            return ("left", k), ("right", k)

    class Extractor:
        def detect_and_compute(self, img):
            side, k = img
            if img == ("right", 1):
                return [], None                       # what OpenCV returns for an image without features
            return g[f"kp_{side}_{k}"], g[f"des_{side}_{k}"]

    pipe = TrackingPipeline(cam, Images(), 2e-3, 1.2, 4.0, False, False, "akaze", "bf", True,
                            feature_extractor=Extractor(), db_factory=TrackDBOracle,
                            link_factory=db_adapter.Link, batched=batched, seed=3)
    db = pipe.build(max_frame=F, out_path=None, rel_path=None)
    assert len(db.all_frame_links(1)) == 0 and pipe.matches_per_frame[:2] == [0, 0]
    np.testing.assert_array_equal(np.asarray(db.all_frame_links(0), np.float64).reshape(-1, 3), g["links_0"])
    identity = np.hstack([np.eye(3), np.zeros((3, 1))])
    np.testing.assert_array_equal(pipe.get_relative_Rts()[:2], [identity, identity])
    if F > 3:
        assert np.abs(pipe.get_relative_Rts()[2] - g["relative_Rts"][2]).max() < 0.05
