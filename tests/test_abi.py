"""CPU tests of the boundary: the shared library loads, exports every symbol
the header declares, and the Python binding table covers the same set."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "slamhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(slamhip_\w+)\s*\(", text)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "slamhip_hamming_knn2" in syms and "slamhip_ransac_score" in syms


def test_library_exports_every_declared_symbol():
    from slam_amd import _lib
    for name in declared_symbols():
        assert hasattr(_lib.lib, name), f"{name} declared in slamhip.h but not exported"


def test_binding_table_matches_header():
    from slam_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_version_and_strerror_without_gpu():
    from slam_amd import _lib
    assert _lib.lib.slamhip_version() >= 102
    assert _lib.strerror(0) == "ok"
    assert "workspace" in _lib.strerror(_lib.EWORKSPACE)
    # argument validation happens before any device work
    assert _lib.lib.slamhip_hamming_knn2(None, -1, None, 0, 61, 61, None, None, None, 0, None) == _lib.EINVAL
    assert _lib.lib.slamhip_hamming_knn2_workspace_bytes(8192, 8192, 1, 8192, 8192, 61) > 0
    assert _lib.lib.slamhip_hamming_knn2_workspace_bytes(10, 10, 1, 10, 10, 500) == 0


def test_product_path_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    from slam_amd import kernels
    with pytest.raises(RuntimeError):
        kernels.hamming_knn2(np.zeros((4, 61), np.uint8), np.zeros((4, 61), np.uint8))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "slam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower().replace("# no oracle", ""), f


def test_header_is_plain_c_and_cxx(tmp_path):
    """include/slamhip.h is the boundary other hosts bind: it must compile as C99
    and as C++ on its own (no HIP or torch types)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "use.c"
    src.write_text('#include "slamhip.h"\nint main(void) { return slamhip_version(); }\n')
    for cc, flags in (("gcc", ["-std=c99", "-pedantic"]), ("g++", ["-std=c++17", "-x", "c++"])):
        if shutil.which(cc) is None:
            continue
        r = subprocess.run([cc, *flags, "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                            "-I", os.path.join(root, "include"), str(src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_argument_structs_match_the_header_and_are_size_checked(tmp_path):
    """The ctypes mirrors of the two argument structs have the C compiler's size and field
    offsets, they fill in struct_size themselves, and the entry points refuse a struct built
    against another header (a different struct_size) before touching any pointer in it."""
    import ctypes as C
    import shutil
    import subprocess
    from slam_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    probes = {"slamhip_frontend_args": ("struct_size", "n_frames", "d_samples", "d_workspace", "first_pair"),
              "slamhip_verify_args": ("struct_size", "n_rows_total", "seed", "d_tobase", "d_pair_ids", "d_samples")}
    body = "".join(f'printf("{s} %zu\\n", sizeof({s}));\n' + "".join(
        f'printf("{s}.{f} %zu\\n", offsetof({s}, {f}));\n' for f in fields) for s, fields in probes.items())
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "slamhip.h"\nint main(void) {\n' + body + "return 0; }\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, cls in (("slamhip_frontend_args", _lib.FrontendArgs), ("slamhip_verify_args", _lib.VerifyArgs)):
        assert C.sizeof(cls) == int(out[cname])
        for f in probes[cname]:
            assert getattr(cls, f).offset == int(out[f"{cname}.{f}"]), (cname, f)
        assert cls().struct_size == C.sizeof(cls)
    a = _lib.FrontendArgs()
    a.n_frames, a.desc_bytes, a.desc_stride, a.n_iter_max = 2, 61, 61, 10
    assert _lib.lib.slamhip_frontend_workspace_bytes(C.byref(a)) > 0
    a.struct_size -= 8                              # what a caller built against the previous header passes
    assert _lib.lib.slamhip_frontend_workspace_bytes(C.byref(a)) == 0
    assert _lib.lib.slamhip_frontend_batch(C.byref(a), None) == _lib.EINVAL
    assert _lib.lib.slamhip_frontend_pack_record(C.byref(a), 0, 1, 0, 0, 0, 0, None, None, 0, 0, None, 0, None) == _lib.EINVAL
    assert _lib.lib.slamhip_frontend_unpack_frame(None, 1, 1, 61, 61, None, None, None, None, None, None) == _lib.EINVAL
    v = _lib.VerifyArgs()
    v.desc_bytes = v.desc_stride = 61
    v.n_iter_max = 10
    assert _lib.lib.slamhip_verify_workspace_bytes(C.byref(v)) > 0
    v.struct_size = 0                               # a zero-initialised struct of a caller that never heard of the field
    assert _lib.lib.slamhip_verify_workspace_bytes(C.byref(v)) == 0
    assert _lib.lib.slamhip_verify_pairs(C.byref(v), None) == _lib.EINVAL
