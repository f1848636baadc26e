"""TEST INFRASTRUCTURE ONLY (CPU oracle) -- ctypes wrapper of oracle/initializer_oracle.c
(Initializer::CheckHomography / CheckFundamental and the keep-the-best loop, slam_pipeline/src/Initializer.cc:152-245,
322-487).  Never imported by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_L = None


def lib():
    global _L
    if _L is None:
        so = os.path.join(HERE, "libinitializer_oracle.so")
        src = os.path.join(HERE, "initializer_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", HERE, "libinitializer_oracle.so"], stdout=subprocess.DEVNULL)
        _L = C.CDLL(so)
        _L.initializer_find_best.restype = C.c_int
        _L.initializer_find_best.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float,
                                             C.c_void_p, C.c_void_p, C.c_void_p]
    return _L


def find_best(model, m21, m12, matches, sigma):
    """-> (best index or -1, scores f32 [n_hyp], best inliers bool [n])"""
    m21 = np.ascontiguousarray(m21, np.float32).reshape(-1, 9)
    n_hyp = len(m21)
    m12 = np.ascontiguousarray(m12 if m12 is not None else np.zeros_like(m21), np.float32).reshape(-1, 9)
    m = np.ascontiguousarray(matches, np.int32).reshape(-1, 4)
    scores = np.zeros(n_hyp, np.float32)
    inl = np.zeros(max(len(m), 1), np.uint8)
    scratch = np.zeros(max(len(m), 1), np.uint8)
    best = lib().initializer_find_best(model, n_hyp, m21.ctypes.data, m12.ctypes.data, len(m), m.ctypes.data,
                                       C.c_float(sigma), scores.ctypes.data, inl.ctypes.data, scratch.ctypes.data)
    return best, scores, inl[:len(m)].astype(bool)
