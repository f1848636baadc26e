"""ctypes binding of oracle/liborb_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
The restated reference path is /root/reference/src/featurematcher.cpp:10-45
(see orb_oracle.c for the per-function citations).  Parity unpinned (no OpenCV here).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liborb_oracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("response", "<f4"), ("angle", "<f4"),
                     ("octave", "<i4"), ("lx", "<i4"), ("ly", "<i4"), ("fast_score", "<i4")])


class Opts(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("nlevels", C.c_int32), ("fast_threshold", C.c_int32),
                ("edge_threshold", C.c_int32), ("blur_tie_even", C.c_int32),
                ("level_size_mul_inv", C.c_int32), ("blur_kernel_sum256", C.c_int32)]


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("orb_oracle.c", "orb_oracle.h", "orb_pattern.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "liborb_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orb_oracle_create.restype = C.c_void_p
        L.orb_oracle_create.argtypes = [C.c_int, C.c_int, C.POINTER(Opts)]
        L.orb_oracle_destroy.argtypes = [C.c_void_p]
        L.orb_oracle_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_ssize_t]
        L.orb_oracle_keypoints.restype = C.c_void_p
        L.orb_oracle_keypoints.argtypes = [C.c_void_p]
        L.orb_oracle_descriptors.restype = C.c_void_p
        L.orb_oracle_descriptors.argtypes = [C.c_void_p]
        L.orb_oracle_num_keypoints.argtypes = [C.c_void_p]
        L.orb_oracle_level_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orb_oracle_level_scale.restype = C.c_float
        L.orb_oracle_level_scale.argtypes = [C.c_void_p, C.c_int]
        L.orb_oracle_level_quota.argtypes = [C.c_void_p, C.c_int]
        L.orb_oracle_level_pixels.restype = C.c_void_p
        L.orb_oracle_level_pixels.argtypes = [C.c_void_p, C.c_int]
        L.orb_oracle_level_blurred.restype = C.c_void_p
        L.orb_oracle_level_blurred.argtypes = [C.c_void_p, C.c_int]
        L.orb_oracle_fast_candidates.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.orb_oracle_stage1_keypoints.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.orb_oracle_fast_score_map.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orb_oracle_knn_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                           C.c_float, C.c_void_p, C.c_int]
        L.orb_oracle_knn2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orb_oracle_knn2.restype = None
        L.orb_oracle_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orb_oracle_sincosf.restype = None
        L.orb_oracle_fast_atan2.restype = C.c_float
        L.orb_oracle_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib = L
    return _lib


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros((0,), dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class OrbOracle:
    """cv::ORB::create() defaults, as the reference constructs it (featurematcher.cpp:4)."""

    def __init__(self, width, height, nfeatures=500, nlevels=8, fast_threshold=20, edge_threshold=31,
                 blur_tie_even=1, level_size_mul_inv=0, blur_kernel_sum256=0):
        self.L = lib()
        o = Opts(nfeatures, nlevels, fast_threshold, edge_threshold, blur_tie_even, level_size_mul_inv, blur_kernel_sum256)
        self.nlevels = nlevels
        self.w, self.h = width, height
        self.ctx = self.L.orb_oracle_create(width, height, C.byref(o))
        if not self.ctx:
            raise ValueError("orb_oracle_create failed")

    def __del__(self):
        if getattr(self, "ctx", None):
            self.L.orb_oracle_destroy(self.ctx)
            self.ctx = None

    def extract(self, img):
        img = np.asarray(img, dtype=np.uint8)
        assert img.shape == (self.h, self.w) and img.strides[1] == 1
        n = self.L.orb_oracle_extract(self.ctx, img.ctypes.data, img.strides[0])
        if n < 0:
            raise RuntimeError("orb_oracle_extract failed")
        kps = _arr(self.L.orb_oracle_keypoints(self.ctx), n, KP_DTYPE)
        desc = _arr(self.L.orb_oracle_descriptors(self.ctx), n * 32, np.uint8).reshape(n, 32)
        return kps, desc

    def level_size(self, l):
        w, h = C.c_int(), C.c_int()
        self.L.orb_oracle_level_size(self.ctx, l, C.byref(w), C.byref(h))
        return w.value, h.value

    def level_scale(self, l):
        return self.L.orb_oracle_level_scale(self.ctx, l)

    def level_quota(self, l):
        return self.L.orb_oracle_level_quota(self.ctx, l)

    def level_pixels(self, l, blurred=False):
        w, h = self.level_size(l)
        f = self.L.orb_oracle_level_blurred if blurred else self.L.orb_oracle_level_pixels
        return _arr(f(self.ctx, l), w * h, np.uint8).reshape(h, w)

    def fast_candidates(self, l):
        p = C.c_void_p()
        n = self.L.orb_oracle_fast_candidates(self.ctx, l, C.byref(p))
        return _arr(p.value, n * 3, np.int32).reshape(n, 3)

    def fast_score_map(self, l):
        """uint8 [h, w]: FAST score of every pixel before NMS and border reject (0 = not a corner); after extract()"""
        w, h = self.level_size(l)
        out = np.zeros((h, w), np.uint8)
        if self.L.orb_oracle_fast_score_map(self.ctx, l, out.ctypes.data):
            raise ValueError("bad level")
        return out

    def stage1_keypoints(self):
        p = C.c_void_p()
        n = self.L.orb_oracle_stage1_keypoints(self.ctx, C.byref(p))
        return _arr(p.value, n, KP_DTYPE)


def knn2(d1, d2):
    d1 = np.ascontiguousarray(d1, np.uint8)
    d2 = np.ascontiguousarray(d2, np.uint8)
    out = np.zeros((len(d1), 4), np.int32)
    lib().orb_oracle_knn2(d1.ctypes.data, len(d1), d2.ctypes.data, len(d2), out.ctypes.data)
    return out


def knn_match(k1, d1, k2, d2, ratio, cap=8192):
    """featurematcher.cpp:23-42 on extracted features -> int32 [m,4] (x1,y1,x2,y2)."""
    k1 = np.ascontiguousarray(k1)
    k2 = np.ascontiguousarray(k2)
    d1 = np.ascontiguousarray(d1, np.uint8)
    d2 = np.ascontiguousarray(d2, np.uint8)
    out = np.zeros((cap, 4), np.int32)
    m = lib().orb_oracle_knn_match(d1.ctypes.data, len(d1), k1.ctypes.data, d2.ctypes.data, len(d2),
                                   k2.ctypes.data, ratio, out.ctypes.data, cap)
    assert m <= cap
    return out[:m].copy()


class FeatureMatcherOracle:
    """Mirror of the reference's ::FeatureMatcher (featurematcher.h:7-22)."""

    def __init__(self, threshold=0.8, **orb_kw):
        self.threshold = threshold
        self.orb_kw = orb_kw
        self._ctx = {}

    def SetThreshold(self, v):
        self.threshold = v

    def _orb(self, shape):
        if shape not in self._ctx:
            self._ctx[shape] = (OrbOracle(shape[1], shape[0], **self.orb_kw),
                                OrbOracle(shape[1], shape[0], **self.orb_kw))
        return self._ctx[shape]

    def extract_both(self, im1, im2):
        oa, ob = self._orb(im1.shape)
        return oa.extract(im1), ob.extract(im2)

    def MatchFrames(self, im1, im2):
        (k1, d1), (k2, d2) = self.extract_both(im1, im2)
        return knn_match(k1, d1, k2, d2, self.threshold)


def sincosf(t):
    s, c = C.c_float(), C.c_float()
    lib().orb_oracle_sincosf(C.c_float(t), C.byref(s), C.byref(c))
    return s.value, c.value
