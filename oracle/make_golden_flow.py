"""
Generates tests/golden/ref_build_flow.npz and ref_consensus_flow.npz.  Run in the
build container only:

    python oracle/make_golden_flow.py

REFERENCE-PINNED CONTROL FLOW.  The two callers of the hot path are run AS THEY ARE,
imported from /root/reference:

  * TrackingPipeline.build()            (backend/tracking/fill_database.py:85-255)
  * LoopClosure.consensus_pnp_match()   (backend/optimizers/loop_closure.py:167-235)

Everything between the OpenCV calls is the reference's own code and runs unmodified:
the frame loop, create_inlier_outlier_matches, TrackingDB.create_links / add_frame,
create_L_R_pixels, the whole ransac_pnp_pair1 loop (projection, supporters, adaptive
stop), the index plumbing and float32 casts of consensus_pnp_match, its two
`min_inliers_loop` gates, frame_data.

OpenCV, GTSAM, the image files and the AKAZE extractor are not available here, so the
leaves that are nothing but an OpenCV call are replaced, one reference method each, by
the CPU restatement this repository tests against (oracle/slam_oracle.py):

    ImageSequence.__getitem__            -> blank images tagged with the frame number
    FeatureExtractor.detect_and_compute  -> prepared synthetic keypoints + descriptors
    DescriptorMatcher.match              -> so.bf_match          (cv2.BFMatcher.match + sort)
    Triangulation.triangulate_opencv     -> so.triangulate_cv    (cv2.triangulatePoints)
    LoopClosure.compute_triangulation    -> so.triangulate_cv, fp64 points
    PnP.solve_pnp                        -> so.p3p_solve (AP3P flag) / a least-squares
                                            pose from the given points (ITERATIVE flag)
    numpy.random.default_rng()           -> seeded per call (pnp.py:122 is unseeded)
    cv2.KeyPoint, gtsam.Pose3/Rot3/Point3 -> plain holders

So these fixtures pin the restated DATA FLOW (oracle/pipeline_oracle.py) against the
reference's; the numerical behaviour of the OpenCV leaves stays unpinned.  The
reference itself never travels; only the arrays written here do.
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import make_golden as mg  # noqa: E402
from oracle import slam_oracle as so  # noqa: E402

N_FRAMES = 5
SEED0 = 4100          # numpy.random.default_rng() call k inside the reference gets SEED0 + k


class KP:
    """What the reference reads of a cv2.KeyPoint: `.pt`.  OpenCV keeps the point as a
    Point2f, so the coordinates come back rounded to float32 (and widened to Python
    floats); that matters for LoopClosure.frame_data, which builds key points from
    Link.y = (yL + yR) / 2, in general not a float32 value."""
    def __init__(self, x, y, size=1):
        self.pt = (float(np.float32(x)), float(np.float32(y)))


class DM:
    """What the reference reads of a cv2.DMatch."""
    def __init__(self, q, t, i=0, d=0.0):
        self.queryIdx, self.trainIdx, self.imgIdx, self.distance = int(q), int(t), int(i), float(d)


class Pose:
    def __init__(self, R, t):
        self.T = np.eye(4)
        self.T[:3, :3], self.T[:3, 3] = np.asarray(R), np.asarray(t).reshape(3)

    def inverse(self):
        p = Pose(np.eye(3), np.zeros(3))
        p.T = np.linalg.inv(self.T)
        return p


def synthetic_frames(cam, n_frames, rng):
    """Small KITTI-shaped frames: a static scene seen from a camera that moves forward,
    noisy pixels, clutter in both images, descriptors that survive with a few flipped bits."""
    n_pts = 260
    X = np.column_stack([rng.uniform(-18, 18, n_pts), rng.uniform(-2.5, 1.5, n_pts), rng.uniform(6, 60, n_pts)])
    base = rng.integers(0, 256, (n_pts, 61), dtype=np.uint8)
    base[:, 60] &= 0x3F
    frames, poses = [], []
    for k in range(n_frames):
        ang = 0.004 * k
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        t = np.array([0.02 * k, 0.0, -0.8 * k])
        poses.append(np.hstack([R, t[:, None]]))
        Xc = X @ R.T + t
        Xh = np.column_stack([Xc, np.ones(n_pts)])
        pl, pr = Xh @ cam.P_left.T, Xh @ cam.P_right.T
        uvl, uvr = pl[:, :2] / pl[:, 2:3], pr[:, :2] / pr[:, 2:3]
        vis = (Xc[:, 2] > 2) & (uvl[:, 0] > 0) & (uvl[:, 0] < 1226) & (uvl[:, 1] > 0) & (uvl[:, 1] < 370) \
            & (uvr[:, 0] > 0)
        ids = np.flatnonzero(vis)
        ids = ids[rng.random(len(ids)) < 0.9]
        ids_r = ids[rng.random(len(ids)) < 0.92]

        def view(idsel, uv):
            des = base[idsel].copy()
            flips = rng.random(des.shape + (8,)) < 0.02
            des ^= np.packbits(flips, axis=-1)[..., 0]
            des[:, 60] &= 0x3F
            kp = uv[idsel] + rng.normal(0, 0.4, (len(idsel), 2))
            nclut = 40
            kp = np.vstack([kp, np.column_stack([rng.uniform(0, 1226, nclut), rng.uniform(0, 370, nclut)])])
            cl = rng.integers(0, 256, (nclut, 61), dtype=np.uint8)
            cl[:, 60] &= 0x3F
            des = np.vstack([des, cl])
            perm = rng.permutation(len(kp))
            return kp[perm].astype(np.float32), des[perm]
        kl, dl = view(ids, uvl)
        kr, dr = view(ids_r, uvr)
        frames.append((kl, kr, dl, dr))
    return frames, poses


def main():
    PnP, Triangulation, CameraModel, _ = mg.import_reference()
    import cv2      # placeholders put there by import_reference  # noqa: E402
    import gtsam    # noqa: E402
    cv2.KeyPoint, cv2.DMatch = KP, DM
    gtsam.Rot3, gtsam.Point3, gtsam.Pose3 = (lambda R: R), (lambda t: t), Pose
    from slam.backend.tracking import fill_database as fd          # type: ignore
    from slam.backend.tracking.database import TrackingDB          # type: ignore
    from slam.backend.optimizers import loop_closure as lc         # type: ignore
    from slam.frontend.vision.descriptor_matcher import DescriptorMatcher   # type: ignore

    calib = tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False)
    calib.write("P0: " + " ".join(repr(v) for v in mg.KITTI_P0) + "\n")
    calib.write("P1: " + " ".join(repr(v) for v in mg.KITTI_P1) + "\n")
    calib.close()
    cam = CameraModel.from_kitti(calib.name)
    rng = np.random.default_rng(77)
    frames, _ = synthetic_frames(cam, N_FRAMES, rng)

    # ---------------------------------------------------------------- the leaves
    class Images:
        def __len__(self):
            return len(frames)

        def __getitem__(self, k):
            L = np.zeros((370, 1226), np.uint8)
            R = np.zeros((370, 1226), np.uint8)
            L[0, 0], R[0, 0], R[0, 1] = k, k, 1
            return L, R

    class Extractor:
        def detect_and_compute(self, image, mask=None):
            k, right = int(image[0, 0]), int(image[0, 1])
            kp, des = (frames[k][1], frames[k][3]) if right else (frames[k][0], frames[k][2])
            return [KP(x, y) for x, y in kp], des.copy()

    def match(self, desL, desR, kpsL=None, kpsR=None, shapeL=None, shapeR=None):
        qi, ti, di = so.bf_match(np.asarray(desL), np.asarray(desR), cross_check=bool(self.cross_check))
        return [DM(q, t, 0, float(np.float32(d))) for q, t, d in zip(qi, ti, di)]

    def triangulate_opencv(self, kps_left, kps_right, matches):
        pts_L, pts_R = DescriptorMatcher.extract_matched_pixels(kps_left, kps_right, matches)   # the reference's
        return so.triangulate_cv(self.P_left, self.P_right, pts_L, pts_R)

    def compute_triangulation(self, P_L, P_R, kpL, kpR):
        pts_L = np.array([k.pt for k in kpL])          # float64, as loop_closure.py:241-242
        pts_R = np.array([k.pt for k in kpR])
        return so.triangulate_cv(P_L, P_R, pts_L, pts_R, out_fp64=True)

    log = {"refine_obj": [], "refine_img": [], "refine_Rt": []}

    def solve_pnp(self, obj_pts, img_pts, flag):
        if flag.__name__ == "SOLVEPNP_AP3P":
            return so.p3p_solve(obj_pts, img_pts, self.K)
        assert flag.__name__ == "SOLVEPNP_ITERATIVE", flag
        obj, img = np.asarray(obj_pts), np.asarray(img_pts)
        Rt0 = None
        for s in range(0, len(obj) - 3):                       # any four of the consensus set
            Rt0 = so.p3p_solve(obj[s:s + 4], img[s:s + 4], self.K)
            if Rt0 is not None:
                break
        Rt = None if Rt0 is None else so.refine_pose(obj, img, self.K, Rt0)[0]
        log["refine_obj"].append(obj.copy())
        log["refine_img"].append(img.copy())
        log["refine_Rt"].append(Rt)
        return Rt

    DescriptorMatcher.match = match
    Triangulation.triangulate_opencv = triangulate_opencv
    PnP.solve_pnp = solve_pnp
    lc.LoopClosure.compute_triangulation = compute_triangulation

    calls = [0]
    real_rng = np.random.default_rng

    def seeded(*a, **k):
        if a or k:
            return real_rng(*a, **k)
        calls[0] += 1
        return real_rng(SEED0 + calls[0] - 1)

    # ------------------------------------------------- TrackingPipeline.build()
    cwd = os.getcwd()
    work = tempfile.mkdtemp(prefix="slamflow_")

    def run_build(cross_check: bool, name: str, prefix: str, out: dict):
        pipe = fd.TrackingPipeline.__new__(fd.TrackingPipeline)       # __init__ only builds cv2 objects
        pipe.cam, pipe.img_seq, pipe.plot, pipe.relative_Rts, pipe.matcher_type = cam, Images(), False, None, "bf"
        pipe.feature_extractor = Extractor()
        pipe.stereo_matcher = DescriptorMatcher.__new__(DescriptorMatcher)
        pipe.frame_matcher = DescriptorMatcher.__new__(DescriptorMatcher)
        for m in (pipe.stereo_matcher, pipe.frame_matcher):
            m.matcher_type, m.cross_check = "bf", cross_check         # run_slam.py:34-35 (default False)
        pipe.triang = Triangulation(cam, 1.2, 4.0)                    # run_slam.py:31-32
        pipe.pnp = PnP(cam, 4.0)

        added = []
        real_add = TrackingDB.add_frame

        def add_frame(self, links, left_features, matches_to_previous_frame=None, inliers=None):
            ms = matches_to_previous_frame or []
            added.append({
                "links": np.array([[l.x_left, l.x_right, l.y] for l in links], np.float64).reshape(-1, 3),
                "des": np.asarray(left_features).copy(),
                "mq": np.array([m.queryIdx for m in ms], np.int32),
                "mt": np.array([m.trainIdx for m in ms], np.int32),
                "md": np.array([m.distance for m in ms], np.float32),
                "mask": np.zeros(0, bool) if inliers is None else np.asarray(inliers, bool).copy()})
            return real_add(self, links, left_features, matches_to_previous_frame, inliers)

        TrackingDB.add_frame = add_frame
        n_ref0, calls0 = len(log["refine_obj"]), calls[0]
        os.chdir(work)
        np.random.default_rng = seeded
        try:
            pipe.build(max_frame=N_FRAMES, out_path=os.path.join(work, name))
        finally:
            np.random.default_rng = real_rng
            os.chdir(cwd)
            TrackingDB.add_frame = real_add
        assert calls[0] - calls0 == N_FRAMES - 1 and len(added) == N_FRAMES

        out[prefix + "seed0"] = np.int64(SEED0 + calls0)
        out[prefix + "relative_Rts"] = pipe.relative_Rts
        for k in range(N_FRAMES):
            a = added[k]
            out[f"{prefix}links_{k}"], out[f"{prefix}inlier_des_{k}"] = a["links"], a["des"]
            out[f"{prefix}tm_query_{k}"], out[f"{prefix}tm_train_{k}"] = a["mq"], a["mt"]
            out[f"{prefix}tm_dist_{k}"], out[f"{prefix}tm_mask_{k}"] = a["md"], a["mask"]
        for p in range(N_FRAMES - 1):
            out[f"{prefix}refine_obj_{p}"] = log["refine_obj"][n_ref0 + p]
            out[f"{prefix}refine_img_{p}"] = log["refine_img"][n_ref0 + p]
        # state of the TrackingDB that build() filled and serialised
        db0 = TrackingDB()
        db0.load(os.path.join(work, name))
        db0._check_consistency()
        fs = sorted(db0.frameId_to_trackIds_list)
        out[prefix + "db_track_ids"] = np.concatenate([np.asarray(db0.frameId_to_trackIds_list[f], np.int64) for f in fs])
        out[prefix + "db_track_ids_off"] = np.concatenate(
            [[0], np.cumsum([len(db0.frameId_to_trackIds_list[f]) for f in fs])]).astype(np.int64)
        out[prefix + "db_last_track"] = np.int64(db0.last_trackId)
        out[prefix + "db_last_frame"] = np.int64(db0.last_frameId)
        keys = sorted(db0.linkId_to_link)
        out[prefix + "db_link_keys"] = np.asarray(keys, np.int64).reshape(-1, 2)
        out[prefix + "db_link_xyz"] = np.asarray([[db0.linkId_to_link[k].x_left, db0.linkId_to_link[k].x_right,
                                                   db0.linkId_to_link[k].y] for k in keys], np.float64).reshape(-1, 3)
        tk = sorted(db0.trackId_to_frames)
        out[prefix + "db_track_keys"] = np.asarray(tk, np.int64)
        out[prefix + "db_track_len"] = np.asarray([len(db0.trackId_to_frames[t]) for t in tk], np.int64)
        out[prefix + "db_leftover_count"] = np.asarray([len(db0.leftover_links.get(f, [])) for f in fs], np.int64)
        print(f"{name} (cross-check {cross_check}):", [len(a["links"]) for a in added], "stereo inliers;",
              [int(a["mask"].sum()) for a in added[1:]], "PnP inliers of", [len(a["mq"]) for a in added[1:]],
              "entries;", len(tk), "tracks")

    out = {"n_frames": N_FRAMES, "P0": np.array(mg.KITTI_P0), "P1": np.array(mg.KITTI_P1)}
    for k, f in enumerate(frames):
        out[f"kp_left_{k}"], out[f"kp_right_{k}"], out[f"des_left_{k}"], out[f"des_right_{k}"] = f
    run_build(False, "tracks", "", out)
    n_rng_build = calls[0]
    # both matchers cross-checked: fewer matches, and the densified hand-over of fill_database.py:209-223
    run_build(True, "tracks_cc", "cc_", out)
    np.savez_compressed(os.path.join(OUT, "ref_build_flow.npz"), **out)

    # ------------------------------------- LoopClosure.consensus_pnp_match()
    from slam.backend.tracking.database import TrackingDB as DB    # type: ignore
    db = DB()
    db.load(os.path.join(work, "tracks"))
    loop = lc.LoopClosure(cam, None, None, None, None, frame_cross_check=False)
    # (curr, cand, min_inliers_loop, frame_cross_check)
    cases = [(3, 0, 60, 0), (4, 1, 40, 0), (2, 0, 100000, 0), (3, 0, 150, 0), (3, 1, 60, 1), (4, 0, 199, 1)]
    out = {"P0": np.array(mg.KITTI_P0), "P1": np.array(mg.KITTI_P1), "seed0": SEED0 + calls[0],
           "cases": np.array(cases, np.int64)}
    for frame in sorted({c[0] for c in cases} | {c[1] for c in cases}):
        links = db.all_frame_links(frame)
        out[f"links_{frame}"] = np.array([[l.x_left, l.x_right, l.y] for l in links], np.float64)
        out[f"des_{frame}"] = np.asarray(db.features(frame))
    np.random.default_rng = seeded
    try:
        for i, (curr, cand, gate, cc) in enumerate(cases):
            loop.min_inliers_loop, loop.frame_cross_check = gate, bool(cc)
            n_ref = len(log["refine_obj"])
            before = calls[0]
            ok, delta_inv, mask, ids, px = loop.consensus_pnp_match(curr, cand, db)
            out[f"ok_{i}"] = np.array(bool(ok))
            out[f"rng_calls_{i}"] = np.array(calls[0] - before)
            out[f"mask_{i}"] = np.zeros(0, bool) if mask is None else np.asarray(mask, bool)
            out[f"delta_inv_{i}"] = np.zeros((4, 4)) if delta_inv is None else delta_inv.T
            refined = len(log["refine_obj"]) > n_ref
            out[f"refine_obj_{i}"] = log["refine_obj"][-1] if refined else np.zeros((0, 3), np.float32)
            out[f"refine_img_{i}"] = log["refine_img"][-1] if refined else np.zeros((0, 2), np.float32)
            out[f"inl_curr_{i}"] = px.get("inl_curr", np.zeros((0, 2), np.float32))
            out[f"inl_cand_{i}"] = px.get("inl_cand", np.zeros((0, 2), np.float32))
            print(f"consensus case {i}: curr {curr} cand {cand} gate {gate} cross-check {cc}: ok={ok}",
                  "" if mask is None else f"{int(mask.sum())}/{len(mask)} inliers")
    finally:
        np.random.default_rng = real_rng
    np.savez_compressed(os.path.join(OUT, "ref_consensus_flow.npz"), **out)
    os.unlink(calib.name)


if __name__ == "__main__":
    main()
