"""CPU: the placement arithmetic of the product-side RCCL gather (msf_gather_plan, csrc/msf_gather.cpp) against the
Python gather of bench.py (mono_slam_framework_amd/gather.py) on the same fake lists, and its argument checking.  The
transfers themselves need GPUs: tests/test_gather_rccl_gpu.py runs them on one MI355X with a one-rank communicator."""
import numpy as np

from mono_slam_framework_amd import _lib
from mono_slam_framework_amd.gather import gather_plan


def _offsets(counts):
    o = np.zeros((counts.shape[0], counts.shape[1] + 1), np.int32)
    o[:, 1:] = np.cumsum(counts, 1)
    return o


def test_plan_places_the_lists_in_rank_order_densely():
    rng = np.random.default_rng(3)
    counts = rng.integers(0, 9, size=(5, 17))
    counts[2] = 0                                           # a rank with nothing to send
    o = _offsets(counts)
    rc, totals, first = gather_plan(o, 1000)
    assert rc == _lib.MSF_OK
    np.testing.assert_array_equal(totals, counts.sum(1))
    np.testing.assert_array_equal(first, np.concatenate([[0], np.cumsum(counts.sum(1))[:-1]]))
    # the buffer rank 0 ends up with = the ranks' packed lists one after another; pair p of rank r at first[r] + o[r][p]
    lists = [np.arange(int(t) * 4, dtype=np.int32).reshape(-1, 4) + 1000 * r for r, t in enumerate(totals)]
    recv = np.concatenate(lists)
    for r in range(5):
        for p in (0, 5, 16):
            np.testing.assert_array_equal(recv[first[r] + o[r, p]:first[r] + o[r, p + 1]], lists[r][o[r, p]:o[r, p + 1]])


def test_plan_reports_capacity_and_malformed_offsets():
    o = _offsets(np.array([[3, 4], [50, 60]]))
    assert gather_plan(o, 100)[0] == _lib.MSF_ERR_CAPACITY      # rank 1 holds 110 > 100 records
    assert gather_plan(o, 110)[0] == _lib.MSF_OK
    bad = o.copy()
    bad[0, 0] = 1
    assert gather_plan(bad, 1000)[0] == _lib.MSF_ERR_INVALID_ARG
    bad = o.copy()
    bad[1, 1] = 200                                           # decreasing afterwards
    assert gather_plan(bad, 1000)[0] == _lib.MSF_ERR_INVALID_ARG


def test_create_without_a_gpu_is_loud():
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        return
    L = _lib.load()
    g = C.c_void_p()
    ident = (C.c_uint8 * 128)()
    assert L.msf_gather_create(0, 0, 1, ident, 4, 64, C.byref(g)) == _lib.MSF_ERR_HIP
    assert b"no HIP device" in L.msf_gather_last_error(None)
