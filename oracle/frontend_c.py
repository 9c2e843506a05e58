"""ctypes binding of oracle/frontend_oracle.c (the C/OpenMP restatement of the front-end).
TEST INFRASTRUCTURE ONLY: used by tests/ and by bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise OSError(f"{path} not found: build it with `make -C oracle`")
        _lib = C.CDLL(path)
        _lib.oracle_p3p_solve.restype = C.c_int
        _lib.oracle_ransac_loop.restype = C.c_int
        _lib.oracle_refine_pose.restype = C.c_int
        _lib.oracle_frontend_run.restype = C.c_int
        _lib.oracle_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def p3p_solve(obj4, img4, K):
    obj4, img4, K = _f64(obj4), _f64(img4), _f64(K)
    Rt = np.zeros((3, 4))
    return Rt if lib().oracle_p3p_solve(_p(obj4), _p(img4), _p(K), _p(Rt)) else None


def ransac_loop(K, M_right, X, mL, mR, pixel_thresh, p_success=0.99999, n_iter_max=2500, early_inliers=None,
                samples=None, seed=0, hyps=None, hyp_valid=None):
    """oracle_ransac_loop.  Raises ValueError where the reference's math.log(0) does."""
    K, M_right, X, mL, mR = _f64(K), _f64(M_right), _f64(X), _f64(mL), _f64(mR)
    n = len(X)
    mask = np.zeros(max(n, 1), np.uint8)
    Rt = np.zeros((3, 4))
    info = np.zeros(4, np.int32)
    if samples is not None:
        samples = np.ascontiguousarray(samples, dtype=np.int32)
        n_iter_max = min(n_iter_max, len(samples))
    n_hyps = 0
    if hyps is not None:
        hyps = _f64(hyps)
        n_hyps = len(hyps)
        hyp_valid = None if hyp_valid is None else np.ascontiguousarray(hyp_valid, dtype=np.uint8)
    rc = lib().oracle_ransac_loop(_p(K), _p(M_right), _p(X), _p(mL), _p(mR), C.c_int(n), C.c_double(pixel_thresh),
                                  C.c_double(p_success), C.c_int(n_iter_max),
                                  C.c_int(-1 if early_inliers is None else early_inliers), _p(samples),
                                  C.c_uint64(seed), _p(hyps), _p(hyp_valid), C.c_int(n_hyps), _p(mask), _p(Rt),
                                  _p(info))
    if rc == -2:
        raise ValueError("math domain error")
    return {"mask": mask[:n].astype(bool), "Rt_ransac": Rt if info[0] >= 0 else None, "best_index": int(info[0]),
            "best_count": int(info[1]), "iterations": int(info[2]), "n_iter_required": int(info[3])}


def refine_pose(K, X, mL, Rt0, mask=None):
    K, X, mL = _f64(K), _f64(X), _f64(mL)
    Rt = _f64(Rt0).copy()
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    lib().oracle_refine_pose(_p(K), _p(X), _p(mL), _p(m), C.c_int(len(X)), _p(Rt))
    return Rt


def frontend_run(cam, frames, stereo_thresh=4.0, disparity_min=1.2, pnp_thresh=4.0, p_success=0.99999,
                 n_iter_max=2500, samples=None, seed=0, refine=True, threads=0):
    """The whole front-end over `frames` = [(kp_left, kp_right, des_left, des_right), ...]
    (fill_database.py:140-240).  samples: per pair an (n_iter, 4) array, or None."""
    F = len(frames)
    offL = np.zeros(F + 1, np.int32)
    offR = np.zeros(F + 1, np.int32)
    offL[1:] = np.cumsum([len(f[0]) for f in frames])
    offR[1:] = np.cumsum([len(f[1]) for f in frames])
    kpL = _f64(np.concatenate([np.asarray(f[0]).reshape(-1, 2) for f in frames]))
    kpR = _f64(np.concatenate([np.asarray(f[1]).reshape(-1, 2) for f in frames]))
    desL = np.ascontiguousarray(np.concatenate([f[2] for f in frames]), dtype=np.uint8)
    desR = np.ascontiguousarray(np.concatenate([f[3] for f in frames]), dtype=np.uint8)
    nbytes = desL.shape[1]
    nL = int(offL[-1])
    smp = None
    if samples is not None:
        smp = np.zeros((F - 1, n_iter_max, 4), np.int32)
        for p, s in enumerate(samples):
            s = np.asarray(s, np.int32)[:n_iter_max]
            smp[p, :len(s)] = s
    i32 = lambda n: np.zeros(max(n, 1), np.int32)
    out = {"st_count": i32(F), "st_left": i32(nL), "st_right": i32(nL), "st_dist": i32(nL),
           "st_xyz": np.zeros((max(nL, 1), 3), np.float32), "tm_count": i32(F - 1), "tm_q": i32(nL), "tm_t": i32(nL),
           "tm_d": i32(nL), "mask": np.zeros(max(nL, 1), np.uint8), "info": np.zeros((max(F - 1, 1), 4), np.int32),
           "Rt_ransac": np.zeros((max(F - 1, 1), 3, 4)), "Rt": np.zeros((max(F - 1, 1), 3, 4)),
           "has_pose": np.zeros(max(F - 1, 1), np.uint8), "offL": offL}
    K, MR, PL, PR = _f64(cam.K), _f64(cam.M_right), _f64(cam.P_left), _f64(cam.P_right)
    rc = lib().oracle_frontend_run(
        _p(K), _p(MR), _p(PL), _p(PR), _p(kpL), _p(kpR), _p(desL), _p(desR), _p(offL), _p(offR), C.c_int(F),
        C.c_int(nbytes), C.c_double(stereo_thresh), C.c_double(disparity_min), C.c_double(pnp_thresh),
        C.c_double(p_success), C.c_int(n_iter_max), _p(smp), C.c_uint64(seed), C.c_int(int(refine)), C.c_int(threads),
        _p(out["st_count"]), _p(out["st_left"]), _p(out["st_right"]), _p(out["st_dist"]), _p(out["st_xyz"]),
        _p(out["tm_count"]), _p(out["tm_q"]), _p(out["tm_t"]), _p(out["tm_d"]), _p(out["mask"]), _p(out["info"]),
        _p(out["Rt_ransac"]), _p(out["Rt"]), _p(out["has_pose"]))
    if rc == -2:
        raise ValueError("math domain error")
    return out
