"""ctypes binding of oracle/libloftr_oracle.so -- TEST INFRASTRUCTURE ONLY.
Restates ::DNNFeatureMatcher (src/dnnfeaturematcher.cpp:44-102); pinned by tests/golden/loftr_kat.npz."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libloftr_oracle.so")
WEIGHTS = os.path.join(os.path.dirname(_HERE), "mono_slam_framework_amd", "weights", "loftr_teacher.bin")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = [os.path.join(_HERE, f) for f in ("loftr_oracle.c", "loftr_oracle.h")]
        if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
            subprocess.check_call(["make", "-C", _HERE, "libloftr_oracle.so"])
        L = C.CDLL(_SO)
        L.loftr_oracle_create.restype = C.c_void_p
        L.loftr_oracle_create.argtypes = [C.c_char_p]
        L.loftr_oracle_destroy.argtypes = [C.c_void_p]
        L.loftr_oracle_run.argtypes = [C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t] + [C.c_void_p] * 5
        L.loftr_oracle_decode.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_int]
        L.loftr_oracle_set_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def set_threads(n):
    """OpenMP threads used by the restatement; returns the count in effect."""
    return lib().loftr_oracle_set_threads(int(n))


class DNNFeatureMatcherOracle:
    """Mirror of ::DNNFeatureMatcher (dnnfeaturematcher.h:9-36) on the CPU restatement."""

    def __init__(self, threshold=0.15, weights=WEIGHTS):
        self.L = lib()
        self.o = self.L.loftr_oracle_create(weights.encode())
        if not self.o:
            raise IOError("cannot load " + weights)
        self.threshold = threshold

    def __del__(self):
        if getattr(self, "o", None):
            self.L.loftr_oracle_destroy(self.o)
            self.o = None

    def SetThreshold(self, v):
        self.threshold = v

    def run(self, im1, im2):
        im1 = np.ascontiguousarray(im1, np.uint8)
        im2 = np.ascontiguousarray(im2, np.uint8)
        assert im1.shape == (480, 640) and im2.shape == (480, 640)
        conf = np.empty((1200, 1200), np.float32)
        sim = np.empty((1200, 1200), np.float32)
        f0 = np.empty((1200, 32), np.float32)
        f1 = np.empty((1200, 32), np.float32)
        tok = np.empty((2, 1200, 32), np.float32)
        rc = self.L.loftr_oracle_run(self.o, im1.ctypes.data, im1.strides[0], im2.ctypes.data, im2.strides[0],
                                     conf.ctypes.data, sim.ctypes.data, f0.ctypes.data, f1.ctypes.data, tok.ctypes.data)
        if rc:
            raise RuntimeError("loftr_oracle_run failed")
        return {"conf": conf, "sim": sim, "feat0": f0, "feat1": f1, "tok": tok}

    def decode(self, conf, threshold=None, cap=1 << 16):
        out = np.zeros((cap, 4), np.int32)
        n = self.L.loftr_oracle_decode(np.ascontiguousarray(conf, np.float32).ctypes.data,
                                       self.threshold if threshold is None else threshold, out.ctypes.data, cap)
        return out[:min(n, cap)].copy()

    def MatchFrames(self, im1, im2):
        return self.decode(self.run(im1, im2)["conf"])
