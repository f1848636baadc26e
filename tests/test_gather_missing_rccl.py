"""A process without an RCCL to load (MSF_RCCL_LIBRARY names a library that does not exist) gets MSF_ERR_HIP and a
message from msf_gather_unique_id / msf_gather_create -- not a crash (round 3's build formed the message from two
dlerror() calls: the second returns NULL).  Runs in a child process: the binding is made once per process."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, sys
from mono_slam_framework_amd import _lib
L = _lib.load()
L.msf_gather_last_error.restype = C.c_char_p
L.msf_gather_last_error.argtypes = [C.c_void_p]
buf = (C.c_uint8 * 128)()
rc = L.msf_gather_unique_id(buf)
msg = L.msf_gather_last_error(None).decode()
print("RC", rc)
print("MSG", msg)
out = C.c_void_p()
rc2 = L.msf_gather_create(0, 0, 1, buf, 4, 64, C.byref(out))
print("RC2", rc2, bool(out.value))
print("MSG2", L.msf_gather_last_error(None).decode())
"""


def test_missing_rccl_is_an_error_code_with_a_message():
    env = dict(os.environ, MSF_RCCL_LIBRARY="libmsf_no_such_rccl.so", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)      # 139 = the null dereference of the old message code
    lines = dict(ln.split(" ", 1) for ln in r.stdout.strip().splitlines())
    assert lines["RC"] == "-2"                                        # MSF_ERR_HIP
    assert "libmsf_no_such_rccl.so" in lines["MSG"] and "not found" in lines["MSG"]
    assert lines["RC2"].split()[0] in ("-2", "-1") and lines["RC2"].split()[1] == "False"
    assert lines["MSG2"].strip()
