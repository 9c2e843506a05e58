"""The allocation-free matcher call (slamhip_hamming_knn2_ratio_static: scratch in a
file-scope device array inside the library) against the workspace-taking call it is
a variant of, through the C ABI, bit for bit: two calls in a row on the same GPU, the
second with a smaller input, so that whatever the first left in the shared scratch
is seen not to leak into the second."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RATIO = 0.8


def _call_both(q_np, t_np, ratio=RATIO):
    from slam_amd import _lib
    lib = _lib.lib
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    q = torch.from_numpy(q_np).to(dev)
    t = torch.from_numpy(t_np).to(dev)
    nq, nt, d = q.shape[0], t.shape[0], q_np.shape[1]
    stream = torch.cuda.current_stream().cuda_stream

    def outputs():
        return (torch.full((max(nq, 1), 2), -7, dtype=torch.int32, device=dev),
                torch.full((max(nq, 1), 2), -7, dtype=torch.int32, device=dev),
                torch.full((max(nq, 1),), 9, dtype=torch.uint8, device=dev))
    i_ref, d_ref, k_ref = outputs()
    need = lib.slamhip_hamming_knn2_workspace_bytes(nq, nt, 1, nq, nt, d)
    ws = torch.empty(max(int(need), 1), dtype=torch.uint8, device=dev)
    rc = lib.slamhip_hamming_knn2_ratio(q.data_ptr(), nq, t.data_ptr(), nt, d, d, C.c_double(ratio),
                                        i_ref.data_ptr(), d_ref.data_ptr(), k_ref.data_ptr(),
                                        ws.data_ptr(), ws.numel(), stream)
    assert rc == 0, _lib.strerror(rc)
    i_new, d_new, k_new = outputs()

    def static_call():
        return lib.slamhip_hamming_knn2_ratio_static(q.data_ptr(), nq, t.data_ptr(), nt, d, d, C.c_double(ratio),
                                                     i_new.data_ptr(), d_new.data_ptr(), k_new.data_ptr(), stream)
    global _WARM
    if not _WARM:
        # the first call on a device looks the scratch array's address up (hipGetSymbolAddress,
        # once per device); the property below is about every call after it
        assert static_call() == 0
        _WARM = True
    torch.cuda.synchronize()
    # free device memory as the driver reports it (hipMemGetInfo): a hipMalloc / hipFree inside the
    # library would move it; torch's own counter only sees torch's caching allocator
    free_before, _ = torch.cuda.mem_get_info(dev)
    torch_before = torch.cuda.memory_allocated(dev)
    rc = static_call()
    assert rc == 0, _lib.strerror(rc)
    free_during, _ = torch.cuda.mem_get_info(dev)       # kernels still in flight: nothing allocated and held
    torch.cuda.synchronize()
    free_after, _ = torch.cuda.mem_get_info(dev)
    assert free_during == free_before and free_after == free_before
    assert torch.cuda.memory_allocated(dev) == torch_before
    return (i_ref, d_ref, k_ref), (i_new, d_new, k_new)


_WARM = False


def _workload(nq, nt, nbytes, seed):
    from slam_amd.utils.synth import matcher_workload
    return matcher_workload(nq, nt, nbytes, seed=seed)


def test_static_scratch_call_equals_the_workspace_call_twice_in_a_row():
    for (nq, nt, nbytes, seed) in ((8192, 8192, 61, 1), (700, 333, 61, 2)):   # large first, then small
        q, t = _workload(nq, nt, nbytes, seed)
        ref, new = _call_both(q, t)
        for a, b in zip(ref, new):
            assert torch.equal(a, b)
        assert int(new[2].sum().item()) > 0                      # some queries pass the ratio test
        assert int((new[0][:, 0] >= 0).sum().item()) == nq


def test_static_scratch_call_other_widths_and_small_inputs():
    for (nq, nt, nbytes, seed) in ((1500, 4096, 32, 3), (257, 8192, 128, 4), (5, 1, 61, 5), (64, 2, 16, 6)):
        q, t = _workload(nq, nt, nbytes, seed)
        ref, new = _call_both(q, t)
        for a, b in zip(ref, new):
            assert torch.equal(a, b)


def test_static_scratch_call_records_into_a_graph():
    """Stream capture rejects allocation: the call records into a HIP graph (torch captures with
    the error mode that makes any allocating runtime call on any thread fail the capture), and
    the replayed graph gives the workspace call's results."""
    from slam_amd import _lib
    lib = _lib.lib
    dev = torch.device("cuda", 0)
    q_np, t_np = _workload(3000, 2500, 61, 9)
    (i_ref, d_ref, k_ref), _ = _call_both(q_np, t_np)            # also the warm-up of the address lookup
    q, t = torch.from_numpy(q_np).to(dev), torch.from_numpy(t_np).to(dev)
    i2 = torch.full((3000, 2), -7, dtype=torch.int32, device=dev)
    d2 = torch.full((3000, 2), -7, dtype=torch.int32, device=dev)
    k = torch.full((3000,), 9, dtype=torch.uint8, device=dev)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        rc = lib.slamhip_hamming_knn2_ratio_static(q.data_ptr(), 3000, t.data_ptr(), 2500, 61, 61,
                                                   C.c_double(RATIO), i2.data_ptr(), d2.data_ptr(), k.data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _lib.strerror(rc)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(i2, i_ref) and torch.equal(d2, d_ref) and torch.equal(k, k_ref)


def test_static_scratch_call_rejects_what_it_cannot_hold():
    from slam_amd import _lib
    lib = _lib.lib
    dev = torch.device("cuda", 0)
    q = torch.zeros((8193, 61), dtype=torch.uint8, device=dev)
    t = torch.zeros((16, 61), dtype=torch.uint8, device=dev)
    i2 = torch.empty((8193, 2), dtype=torch.int32, device=dev)
    d2 = torch.empty((8193, 2), dtype=torch.int32, device=dev)
    k = torch.empty(8193, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    args = (i2.data_ptr(), d2.data_ptr(), k.data_ptr(), st)
    assert lib.slamhip_hamming_knn2_ratio_static(q.data_ptr(), 8193, t.data_ptr(), 16, 61, 61,
                                                 C.c_double(RATIO), *args) == _lib.EUNSUPPORTED
    assert lib.slamhip_hamming_knn2_ratio_static(t.data_ptr(), 16, q.data_ptr(), 8193, 61, 61,
                                                 C.c_double(RATIO), *args) == _lib.EUNSUPPORTED
    assert "unsupported" in _lib.strerror(_lib.EUNSUPPORTED).lower() or _lib.strerror(_lib.EUNSUPPORTED)
    assert lib.slamhip_hamming_knn2_ratio_static(q.data_ptr(), 16, t.data_ptr(), 16, 61, 61,
                                                 C.c_double(RATIO), i2.data_ptr(), d2.data_ptr(), None,
                                                 st) == _lib.EINVAL
    # rows longer than the matcher supports: the code the workspace call gives for them
    wide = torch.zeros((4, 200), dtype=torch.uint8, device=dev)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    assert lib.slamhip_hamming_knn2_ratio(wide.data_ptr(), 4, wide.data_ptr(), 4, 200, 200, C.c_double(RATIO),
                                          i2.data_ptr(), d2.data_ptr(), k.data_ptr(), ws.data_ptr(), ws.numel(),
                                          st) == _lib.EUNSUPPORTED
    assert lib.slamhip_hamming_knn2_ratio_static(wide.data_ptr(), 4, wide.data_ptr(), 4, 200, 200,
                                                 C.c_double(RATIO), *args) == _lib.EUNSUPPORTED
    assert lib.slamhip_hamming_knn2_ratio_static(wide.data_ptr(), 4, wide.data_ptr(), 4, 0, 0,
                                                 C.c_double(RATIO), *args) == _lib.EINVAL
    # empty query set: nothing to do, no error
    assert lib.slamhip_hamming_knn2_ratio_static(q.data_ptr(), 0, t.data_ptr(), 16, 61, 61,
                                                 C.c_double(RATIO), *args) == 0
