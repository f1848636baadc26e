"""The transparent per-frame cache behind the drop-in MatchFrames: the unmodified callers loop
MatchFrames(X, KF_i) with X fixed (/root/reference/slam_pipeline/src/Tracking.cc:595-632, LocalMapping.cc:176,329,
KeyFrameDatabase.cc:32,64).  With the cache on, off, tiny (evictions) or with a crippled hash (collisions) the match
lists are the same, and they equal the CPU oracle's."""
import os
from contextlib import contextmanager

import numpy as np
import pytest

from mono_slam_framework_amd import _lib, synth
from oracle import orb as oracle_orb

pytestmark = pytest.mark.gpu
W, H = 640, 480


@contextmanager
def _env(**kv):
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update({k: str(v) for k, v in kv.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _sequence(n_cur=5, n_kf=6, seed=1200):
    """the call pattern of SearchLocalPoints: every current frame against the same local key frames"""
    kfs = [synth.synth_pair(seed + i, W, H, shift=(8 * i - 20, 5 * i - 12))[1] for i in range(n_kf)]
    curs = [synth.synth_pair(seed + 3, W, H, shift=(3 * j, -2 * j))[1] for j in range(n_cur)]
    return [(c, k) for c in curs for k in kfs], n_cur + n_kf


def _run(fm, calls):
    return [fm.MatchFrames(a, b) for a, b in calls]


def test_orb_cache_on_off_small_and_colliding_are_identical():
    from mono_slam_framework_amd.matcher import FeatureMatcher
    calls, n_frames = _sequence()
    off = FeatureMatcher(0.6, W, H, flags=_lib.MSF_FLAG_NO_FRAME_CACHE)
    assert off.frame_cache_stats()[2] == 0
    ref = _run(off, calls)
    on = FeatureMatcher(0.6, W, H)
    got = _run(on, calls)
    hits, misses, cap = on.frame_cache_stats()
    assert cap == 64 and misses == n_frames and hits == 2 * len(calls) - n_frames   # every frame extracted exactly once
    for g, r in zip(got, ref):
        np.testing.assert_array_equal(g, r)
    orc = oracle_orb.FeatureMatcherOracle(0.6)
    for i in (0, 7, len(calls) - 1):
        np.testing.assert_array_equal(got[i], orc.MatchFrames(*calls[i]))
    with _env(MSF_FRAME_CACHE_SLOTS=3):               # evictions all the time, including the frame just used
        small = FeatureMatcher(0.6, W, H)
    got = _run(small, calls)
    assert small.frame_cache_stats()[2] == 3 and small.frame_cache_stats()[1] > n_frames
    for g, r in zip(got, ref):
        np.testing.assert_array_equal(g, r)
    with _env(MSF_FRAME_CACHE_HASH_BITS=1):            # two hash values for eleven frames: the byte compare decides
        coll = FeatureMatcher(0.6, W, H)
    got = _run(coll, calls)
    assert coll.frame_cache_stats()[1] == n_frames
    for g, r in zip(got, ref):
        np.testing.assert_array_equal(g, r)


def test_orb_cache_special_cases():
    from mono_slam_framework_amd.matcher import FeatureMatcher
    fm = FeatureMatcher(0.8, W, H)
    off = FeatureMatcher(0.8, W, H, flags=_lib.MSF_FLAG_NO_FRAME_CACHE)
    a, b = synth.synth_pair(77, W, H)
    np.testing.assert_array_equal(fm.MatchFrames(a, a), off.MatchFrames(a, a))       # one frame on both sides
    assert fm.frame_cache_stats()[:2] == (1, 1)
    big = np.zeros((H, W + 29), np.uint8)                                             # another row stride, same content
    big[:, :W] = a
    np.testing.assert_array_equal(fm.MatchFrames(big[:, :W], b), off.MatchFrames(a, b))
    assert fm.frame_cache_stats()[:2] == (2, 2)
    a2 = a.copy()
    a2[H // 2, W // 2] ^= 1                                                           # one bit differs: a different frame
    np.testing.assert_array_equal(fm.MatchFrames(a2, b), off.MatchFrames(a2, b))
    assert fm.frame_cache_stats()[1] == 3
    fm.SetThreshold(0.6)                                                              # the ratio is applied at match time
    off.SetThreshold(0.6)
    np.testing.assert_array_equal(fm.MatchFrames(a, b), off.MatchFrames(a, b))
    assert fm.frame_cache_stats()[1] == 3
    # batches and the slot API do not touch the cache
    A, B = synth.synth_batch(910, 1, W, H)
    fm.match_batch(list(A) * 2, list(B) * 2)
    np.testing.assert_array_equal(fm.MatchFrames(a, b), off.MatchFrames(a, b))
    assert fm.frame_cache_stats()[1] == 3


def test_loftr_token_cache_behind_matchframes():
    from mono_slam_framework_amd.matcher import DNNFeatureMatcher
    kfs = [synth.synth_pair(1500 + i, W, H, mode=1, shift=(16 * i - 16, 16))[1] for i in range(3)]
    curs = [synth.synth_pair(1501, W, H, mode=1, shift=(16 * j, 0))[1] for j in range(2)]
    calls = [(c, k) for c in curs for k in kfs]
    off = DNNFeatureMatcher(threshold=0.15, flags=_lib.MSF_FLAG_NO_FRAME_CACHE)
    with _env(MSF_FRAME_CACHE_SLOTS=4):
        on = DNNFeatureMatcher(threshold=0.15)
    for a, b in calls:
        np.testing.assert_array_equal(on.MatchFrames(a, b, cap=8192), off.MatchFrames(a, b, cap=8192))
    hits, misses, cap = on.frame_cache_stats()
    assert cap == 4 and hits + misses == 2 * len(calls) and misses >= 5
