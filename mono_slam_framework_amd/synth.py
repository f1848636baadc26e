"""Deterministic synthetic frame pairs -- ctypes over csrc/synth.c."""
import ctypes as C

import numpy as np

from . import build as _build

_lib = None


def _L():
    global _lib
    if _lib is None:
        L = C.CDLL(_build.ensure_synth())
        L.msf_synth_pair.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.msf_synth_kat_pattern.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64]
        L.msf_synth_kat_pattern.restype = None
        _lib = L
    return _lib


SEED_BASE = 0x5EED0000


def pair_shift(pair_index):
    """Deterministic (dx, dy) in [-24, 24] for ORB pairs."""
    h = (pair_index * 2654435761 + 12345) & 0xFFFFFFFF
    return (h >> 8) % 49 - 24, (h >> 20) % 49 - 24


def synth_pair(pair_index, width, height, mode=0, noise=8, shift=None, out=None):
    """Returns (a, b) uint8 [H, W] for pair `pair_index`."""
    dx, dy = pair_shift(pair_index) if shift is None else shift
    if out is None:
        a = np.empty((height, width), np.uint8)
        b = np.empty((height, width), np.uint8)
    else:
        a, b = out
    rc = _L().msf_synth_pair(SEED_BASE + pair_index, width, height, dx, dy, mode, noise,
                             a.ctypes.data, a.strides[0], b.ctypes.data, b.strides[0])
    if rc:
        raise ValueError("msf_synth_pair rc=%d" % rc)
    return a, b


def synth_batch(first_pair, n_pairs, width, height, mode=0, noise=8, threads=8):
    """(A, B) uint8 [n, H, W]; generated with a small thread pool (ctypes drops the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    A = np.empty((n_pairs, height, width), np.uint8)
    B = np.empty((n_pairs, height, width), np.uint8)

    def one(i):
        synth_pair(first_pair + i, width, height, mode, noise, out=(A[i], B[i]))

    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(one, range(n_pairs)))
    return A, B


def kat_pattern(width=640, height=480, sx=0, sy=0):
    out = np.empty((height, width), np.uint8)
    _L().msf_synth_kat_pattern(width, height, sx, sy, out.ctypes.data, out.strides[0])
    return out
