"""Host-side LoFTR weight loading (no GPU): the product reads the model file the caller names, like
DNNFeatureMatcher's constructor (/root/reference/src/dnnfeaturematcher.cpp:11-21) -- the reference's ONNX file directly,
or the MSFLTR01 blob, which is only a cache of the same tensors."""
import ctypes as C
import os

import pytest

from mono_slam_framework_amd import _lib
from tests import onnx_writer

REF_ONNX = "/root/reference/model/LoFTR_teacher.onnx"
BLOB = _lib.default_weights_path()


def _info(path):
    L = _lib.load()
    d, n, f = C.c_uint64(), C.c_int32(), C.c_int64()
    rc = L.msf_weights_info(path.encode(), C.byref(d), C.byref(n), C.byref(f))
    return rc, d.value, n.value, f.value, L.msf_last_error(None).decode()


def test_blob_loads_and_roundtrips(tmp_path):
    rc, dig, n, f, _ = _info(BLOB)
    assert rc == 0 and n == 106 and f == 203840      # 165,488 graph initializers + the 38,400-float PE constant
    out = str(tmp_path / "copy.bin")
    assert _lib.load().msf_convert_weights(BLOB.encode(), out.encode()) == 0
    assert _info(out)[:4] == (0, dig, n, f)


@pytest.mark.skipif(not os.path.exists(REF_ONNX), reason="the reference's model file exists only in the build container")
def test_reference_onnx_and_blob_hold_identical_tensors(tmp_path):
    a, b = _info(REF_ONNX), _info(BLOB)
    assert a[0] == 0, a[4]
    assert a[:4] == b[:4]
    out = str(tmp_path / "from_onnx.bin")
    assert _lib.load().msf_convert_weights(REF_ONNX.encode(), out.encode()) == 0
    assert _info(out)[:4] == b[:4]


def test_synthetic_onnx_with_the_same_topology(tmp_path):
    """An ONNX file written from the blob's tensors (tests/onnx_writer.py) loads to the same weights: the loader
    goes by graph structure (Conv / MatMul order, the PE constant, LayerNorm parameter names), not by value ids."""
    p = str(tmp_path / "synthetic.onnx")
    onnx_writer.write_onnx(p, onnx_writer.read_blob(BLOB))
    a, b = _info(p), _info(BLOB)
    assert a[0] == 0, a[4]
    assert a[:4] == b[:4]


def test_malformed_files_are_io_errors(tmp_path):
    raw = open(BLOB, "rb").read()
    cases = {"trunc.bin": raw[:len(raw) // 2], "hdr.bin": raw[:8] + b"\xff\xff\xff\x7f" + raw[12:],
             "junk.onnx": b"\x08\x07\x12\x04test\x3a\x05\x0a\x03\x0a\x01x", "empty": b"", "short": b"MSFLTR01"}
    t = onnx_writer.read_blob(BLOB)
    del t["blk3.wk"]
    for name, data in cases.items():
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        rc, _, _, _, msg = _info(p)
        assert rc == _lib.MSF_ERR_IO and msg.startswith("io:"), (name, rc, msg)
    assert _info(str(tmp_path / "missing.onnx"))[0] == _lib.MSF_ERR_IO
    # a record whose shape product disagrees with its size, and one with ndim > 4
    bad = bytearray(raw)
    bad[12 + 32:12 + 36] = (7).to_bytes(4, "little")
    p = str(tmp_path / "ndim.bin")
    open(p, "wb").write(bytes(bad))
    assert _info(p)[0] == _lib.MSF_ERR_IO
    # ONNX with a missing linear layer
    try:
        onnx_writer.write_onnx(str(tmp_path / "x.onnx"), t)
    except KeyError:
        pass


def test_no_exception_crosses_the_abi():
    """msf_abi.h promises plain status codes: a host exception inside an entry point (std::bad_alloc, forced here by a
    test hook) comes back as MSF_ERR_HIP with a message, not as a C++ exception through the C caller."""
    os.environ["MSF_TEST_HOOKS"] = "1"       # arms the hook (the library reads it at the call)
    try:
        rc, _, _, _, msg = _info("::throw::")
    finally:
        del os.environ["MSF_TEST_HOOKS"]
    assert rc == _lib.MSF_ERR_HIP and "host exception" in msg
    assert _info("::throw::")[0] == _lib.MSF_ERR_IO       # unarmed: an ordinary missing file
