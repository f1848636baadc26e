"""CPU: the C-ABI library builds, loads, and exports every symbol include/msf_abi.h declares; the product path
fails loudly without a GPU (no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "msf_abi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(msf_[a-z_0-9]+)\s*\(", hdr)))


def test_header_symbols_are_exported():
    from mono_slam_framework_amd import _lib
    L = _lib.load()
    names = _declared()
    assert len(names) >= 13
    for n in names:
        assert hasattr(L, n), n
    assert sorted(_lib.ABI_SYMBOLS) == names
    assert L.msf_abi_version() == 4


def test_header_is_plain_c(tmp_path):
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "msf_abi.h"\nint main(void){ msf_config c; msf_default_config(&c, MSF_KIND_ORB); return (int)sizeof(msf_match) - 16; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])


def test_struct_layouts_match_ctypes():
    import ctypes as C
    from mono_slam_framework_amd import _lib
    assert C.sizeof(_lib.Config) == 40 and C.sizeof(_lib.Image) == 24
    assert _lib.MATCH_DTYPE.itemsize == 16 and _lib.KP_DTYPE.itemsize == 32
    L = _lib.load()
    cfg = _lib.Config()
    L.msf_default_config(C.byref(cfg), _lib.MSF_KIND_LOFTR)
    assert cfg.struct_size == C.sizeof(_lib.Config) and abs(cfg.threshold - 0.15) < 1e-7
    assert (cfg.image_width, cfg.image_height) == (640, 480)
    L.msf_default_config(C.byref(cfg), _lib.MSF_KIND_ORB)
    assert abs(cfg.threshold - 0.8) < 1e-7


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mono_slam_framework_amd.matcher import DNNFeatureMatcher, FeatureMatcher, MsfError
    with pytest.raises(MsfError) as e:
        FeatureMatcher()
    assert e.value.code == -2
    with pytest.raises(MsfError):
        DNNFeatureMatcher()


def test_product_does_not_reference_oracle():
    pkg = os.path.join(ROOT, "mono_slam_framework_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".c")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.lower() or f in ("synth.c",), os.path.join(dp, f)
