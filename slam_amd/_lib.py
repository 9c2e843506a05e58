"""ctypes binding of libslamhip.so (the C ABI declared in include/slamhip.h).

The library is the only compute path: if it is missing, import of this module
raises.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

# torch must be loaded first: it brings its own HIP runtime, and the library has
# to bind to that same runtime so that torch's device pointers and streams are
# valid in our launches.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# SLAMHIP_LIB_PATH points at another build of the same library (kernel experiments).
LIB_PATH = os.environ.get("SLAMHIP_LIB_PATH") or os.path.join(_HERE, "libslamhip.so")

OK, EINVAL, EUNSUPPORTED, EWORKSPACE, ELAUNCH = 0, -1, -2, -3, -4


class SlamHipError(RuntimeError):
    def __init__(self, code: int, where: str):
        self.code = code
        super().__init__(f"{where}: {strerror(code)} (code {code})")


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C slam_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "slam_amd has no CPU fallback."
        )
    return C.CDLL(LIB_PATH)


lib = _load()

_vp, _i, _d, _sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
_dp = C.POINTER(C.c_double)

# name -> (restype, argtypes); kept in the order of include/slamhip.h
SIGNATURES = {
    "slamhip_version": (_i, []),
    "slamhip_strerror": (C.c_char_p, [_i]),
    "slamhip_tune": (_i, [_i, _i]),
    "slamhip_matcher_kind": (_i, []),
    "slamhip_profile_begin": (_i, [_i]),
    "slamhip_profile_end": (_i, [_dp]),
    "slamhip_hamming_knn2_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "slamhip_hamming_knn2": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "slamhip_hamming_knn2_ratio": (
        _i, [_vp, _i, _vp, _i, _i, _i, _d, _vp, _vp, _vp, _vp, _sz, _vp]),
    "slamhip_hamming_knn2_ratio_static": (
        _i, [_vp, _i, _vp, _i, _i, _i, _d, _vp, _vp, _vp, _vp]),
    "slamhip_hamming_knn2_batched": (
        _i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "slamhip_ratio_test": (_i, [_vp, _i, _d, _vp, _vp]),
    "slamhip_l2_knn2_workspace_bytes": (_sz, [_i, _i, _i]),
    "slamhip_l2_knn2": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "slamhip_l2_cross_check": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "slamhip_ratio_test_f32": (_i, [_vp, _i, _d, _vp, _vp]),
    "slamhip_sort_by_distance_f32": (_i, [_vp, _i, _i, _vp, _vp]),
    "slamhip_cross_check": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "slamhip_sort_by_distance": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp]),
    "slamhip_stereo_filter_triangulate": (
        _i, [_vp, _i, _vp, _i, _i, _vp, _vp, _i, _dp, _dp, _d, _d, _vp, _vp, _i, _vp]),
    "slamhip_ransac_score": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _dp, _dp, _d, _vp, _vp]),
    "slamhip_ransac_mask": (_i, [_vp, _i, _vp, _vp, _i, _vp, _dp, _dp, _d, _vp, _vp]),
    "slamhip_p3p_solve": (_i, [_vp, _i, _vp, _i, _vp, _i, _dp, _vp, _vp, _vp]),
    "slamhip_refine_pose": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _dp, _vp, _i, _vp, _vp, _vp]),
    "slamhip_project": (_i, [_vp, _i, _i, _dp, _vp, _vp]),
    "slamhip_supporters_pair": (_i, [_vp, _vp, _vp, _vp, _i, _d, _vp, _vp]),
    "slamhip_compute_err": (_i, [_vp, _vp, _i, _vp, _vp]),
    "slamhip_ransac_replay": (_i, [_vp, _i, _i, _d, _i, _i, _vp, _vp]),
}

_i32, _u64 = C.c_int32, C.c_uint64


class FrontendArgs(C.Structure):
    """Mirror of `slamhip_frontend_args` (include/slamhip.h), field for field."""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n_frames", _i32), ("n_left_total", _i32), ("n_right_total", _i32),
        ("max_left", _i32), ("max_right", _i32), ("desc_bytes", _i32), ("desc_stride", _i32),
        ("d_kp_left", _vp), ("d_kp_right", _vp), ("d_des_left", _vp), ("d_des_right", _vp),
        ("d_off_left", _vp), ("d_off_right", _vp),
        ("K", C.c_double * 9), ("M_right", C.c_double * 12),
        ("P_left", C.c_double * 12), ("P_right", C.c_double * 12),
        ("pixel_thresh", _d), ("disparity_min", _d), ("ransac_thresh", _d), ("p_success", _d),
        ("n_iter_max", _i32), ("refine_iters", _i32),
        ("stereo_cross_check", _i32), ("temporal_cross_check", _i32),
        ("seed", _u64), ("d_samples", _vp),
        ("d_st_count", _vp), ("d_st_left", _vp), ("d_st_right", _vp), ("d_st_dist", _vp),
        ("d_st_xyz", _vp), ("d_st_pix_left", _vp), ("d_st_pix_right", _vp), ("d_st_desc", _vp),
        ("d_tm_query", _vp), ("d_tm_train", _vp), ("d_tm_dist", _vp), ("d_tm_mask", _vp),
        ("d_pose_ransac", _vp), ("d_pose_refined", _vp), ("d_ransac_info", _vp),
        ("d_workspace", _vp), ("workspace_bytes", _sz),
        ("first_pair", C.c_int64),
        ("first_stereo_frame", _i32), ("reserved0", _i32),
    ]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(self)


class VerifyArgs(C.Structure):
    """Mirror of `slamhip_verify_args` (include/slamhip.h), field for field."""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n_rows_total", _i32), ("desc_bytes", _i32), ("desc_stride", _i32),
        ("n_pairs", _i32), ("n_out_total", _i32), ("max_q", _i32), ("max_t", _i32),
        ("d_kp_left", _vp), ("d_kp_right", _vp), ("d_des", _vp),
        ("d_qbase", _vp), ("d_qcnt", _vp), ("d_tbase", _vp), ("d_tcnt", _vp), ("d_obase", _vp),
        ("K", C.c_double * 9), ("M_right", C.c_double * 12),
        ("P_left", C.c_double * 12), ("P_right", C.c_double * 12),
        ("ransac_thresh", _d), ("p_success", _d),
        ("n_iter_max", _i32), ("min_points", _i32), ("refine_iters", _i32), ("cross_check", _i32),
        ("seed", _u64),
        ("d_xyz", _vp), ("d_tm_query", _vp), ("d_tm_train", _vp), ("d_tm_dist", _vp),
        ("d_tm_mask", _vp), ("d_pose_ransac", _vp), ("d_pose_refined", _vp), ("d_info", _vp),
        ("d_workspace", _vp), ("workspace_bytes", _sz),
        ("d_tobase", _vp), ("n_tout_total", _i32), ("reserved", _i32),
        ("d_pair_ids", _vp), ("d_samples", _vp),
    ]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(self)


SIGNATURES["slamhip_verify_workspace_bytes"] = (_sz, [C.POINTER(VerifyArgs)])
SIGNATURES["slamhip_verify_pairs"] = (_i, [C.POINTER(VerifyArgs), _vp])
SIGNATURES["slamhip_frontend_workspace_bytes"] = (_sz, [C.POINTER(FrontendArgs)])
SIGNATURES["slamhip_frontend_batch"] = (_i, [C.POINTER(FrontendArgs), _vp])
SIGNATURES["slamhip_frontend_record_bytes"] = (_sz, [_i, _i, C.c_longlong, C.c_longlong, _i])
SIGNATURES["slamhip_frontend_pack_record"] = (_i, [C.POINTER(FrontendArgs), _i, _i, _i, _i, C.c_int64, C.c_int64,
                                                   _vp, _vp, C.c_int64, C.c_int64, _vp, _sz, _vp])
SIGNATURES["slamhip_frontend_unpack_frame"] = (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp])

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = header and library disagree
    _fn.restype = _res
    _fn.argtypes = _args


def strerror(code: int) -> str:
    return lib.slamhip_strerror(int(code)).decode()


def check(code: int, where: str) -> None:
    if code != OK:
        raise SlamHipError(code, where)


def host_doubles(arr, n: int):
    """Row-major float64 host block as a ctypes double* (keeps a reference)."""
    import numpy as np

    a = np.ascontiguousarray(np.asarray(arr, dtype=np.float64).reshape(-1))
    if a.size != n:
        raise ValueError(f"expected {n} values, got {a.size}")
    return a.ctypes.data_as(_dp), a
