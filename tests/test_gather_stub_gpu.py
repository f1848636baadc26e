"""msf_gather_matches_device with TWO ranks (csrc/msf_gather.cpp, N > 1 branch) on the one-GPU box: two processes share
cuda:0 and bind tests/stub_rccl/libstub_rccl.so -- a test-only stand-in for the ten RCCL entry points the product binds,
moving bytes through host shared memory + hipMemcpy -- through the existing MSF_RCCL_LIBRARY override.  Rank 0 must
receive exactly rank 1's packed records at the offsets msf_gather_plan predicts: ordinary lists, an empty rank, a list
of more than one transport chunk, a rank whose total equals the capacity, one beyond it (every rank returns
MSF_ERR_CAPACITY, the object stays usable), and an injected ncclSend failure (hard error on both ranks, the object is
dead, destroy is safe).  This proves offsets, stream ordering and error semantics of the leg that has never run on
more than one GPU -- not RCCL itself.  north_star: "RCCL gather of match lists"."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_DIR = os.path.join(ROOT, "tests", "stub_rccl")
STUB = os.path.join(STUB_DIR, "libstub_rccl.so")
SYMBOLS = ["ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclCommAbort", "ncclAllGather", "ncclSend",
           "ncclRecv", "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString"]


def build_stub():
    src = os.path.join(STUB_DIR, "stub_rccl.cpp")
    if not os.path.exists(STUB) or os.path.getmtime(src) > os.path.getmtime(STUB):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        subprocess.check_call([hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-o", STUB, src, "-lrt"])
    return STUB


def test_stub_exports_what_the_product_binds():
    """not gpu: the stand-in builds and exports every symbol csrc/msf_gather.cpp looks up with dlsym"""
    import ctypes as C
    L = C.CDLL(build_stub())
    for s in SYMBOLS:
        assert hasattr(L, s), s
    src = open(os.path.join(ROOT, "mono_slam_framework_amd", "csrc", "msf_gather.cpp")).read()
    for s in SYMBOLS:
        assert '"%s"' % s in src, "the product no longer binds " + s


@pytest.mark.gpu
def test_two_ranks_through_the_stub(tmp_path):
    build_stub()
    idfile = str(tmp_path / "id")
    procs = []
    for rank in range(2):
        env = dict(os.environ, MSF_RCCL_LIBRARY=STUB, PYTHONPATH=ROOT, MSF_STUB_RCCL_TIMEOUT_S="60")
        if rank == 1:
            env["MSF_STUB_RCCL_FAIL_SEND"] = "5"      # rank 1's sends: normal, two_chunks, exact_capacity, normal_again, then the injected one
        procs.append(subprocess.Popen([sys.executable, os.path.join(STUB_DIR, "worker.py"), str(rank), "2", idfile],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT))
    res = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, (so[-2000:], se[-4000:])
        line = [ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1]
        res.append(json.loads(line[len("RESULT "):]))
    r0, r1 = sorted(res, key=lambda r: r["rank"])
    exp_totals = {"normal": [37, 29], "empty_rank": [11, 0], "two_chunks": [5, 300000], "exact_capacity": [3, 400000],
                  "normal_again": [8, 13]}
    for name, tot in exp_totals.items():
        for r in (r0, r1):
            s = r["scenarios"][name]
            assert s["rc"] == 0 and s["totals"] == tot and s["offsets_ok"], (name, r["rank"], s)
        assert r0["scenarios"][name]["records_ok"], name
    for r in (r0, r1):                      # one record too many: the same soft error on EVERY rank, nothing transferred
        s = r["scenarios"]["over_capacity"]
        assert s["rc"] == -4 and "cap_records" in s["err"], s
    # the injected failure: rank 1's ncclSend fails -> hard error there, communicator aborted; rank 0's receive comes
    # back with an error instead of waiting for ever; both objects are dead, the next call says so, destroy is safe
    assert r1["scenarios"]["injected_send_failure"]["rc"] == -2 and "ncclSend" in r1["scenarios"]["injected_send_failure"]["err"]
    assert r0["scenarios"]["injected_send_failure"]["rc"] == -2
    for r in (r0, r1):
        s = r["scenarios"]["after_failure"]
        assert s["rc"] == -2 and "destroy this gather object" in s["err"], s
        assert r["destroyed"]
