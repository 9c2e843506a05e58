import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session")
def oracle_c():
    """ctypes handle of the C Hamming oracle (oracle/liboracle.so)."""
    import ctypes as C
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = C.CDLL(path)
    lib.oracle_hamming_knn2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_int]
    lib.oracle_hamming_knn2.restype = None

    def knn2(q, t, threads=0):
        import numpy as np
        q = np.ascontiguousarray(q, dtype=np.uint8)
        t = np.ascontiguousarray(t, dtype=np.uint8)
        nq, nt = q.shape[0], t.shape[0]
        d = q.shape[1]
        idx2 = np.empty((nq, 2), dtype=np.int32)
        dist2 = np.empty((nq, 2), dtype=np.int32)
        lib.oracle_hamming_knn2(q.ctypes.data, nq, t.ctypes.data, nt, d, d, idx2.ctypes.data,
                                dist2.ctypes.data, threads)
        return idx2, dist2
    return knn2
