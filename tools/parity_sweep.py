#!/usr/bin/env python3
"""One-off wide parity sweep (not part of the test suite): GPU match lists vs the CPU oracle over many synthetic pairs,
sizes and texture modes.  Prints the number of mismatching pairs (expected 0)."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mono_slam_framework_amd import synth                      # noqa: E402
from mono_slam_framework_amd.matcher import FeatureMatcher     # noqa: E402
from oracle import orb as oracle_orb                           # noqa: E402


def sweep(w, h, n, mode, first, ratio=0.6, threads=16):
    fm = FeatureMatcher(ratio, w, h, max_batch_pairs=128)   # 256 frames per call: the one-launch walker with 240-row strips
    bad = 0
    tot = 0
    for p0 in range(0, n, 128):
        m = min(128, n - p0)
        A, B = synth.synth_batch(first + p0, m, w, h, mode=mode)
        got = fm.match_batch(list(A), list(B), cap=4096)

        def one(i):
            return oracle_orb.FeatureMatcherOracle(ratio).MatchFrames(A[i], B[i])
        with ThreadPoolExecutor(threads) as ex:
            exp = list(ex.map(one, range(m)))
        for g, e in zip(got, exp):
            tot += len(e)
            if g.shape != e.shape or not np.array_equal(g, e):
                bad += 1
    print("%dx%d mode %d: %d pairs, %d matches, %d mismatching pairs" % (w, h, mode, n, tot, bad), flush=True)
    return bad


if __name__ == "__main__":
    bad = 0
    for (w, h, n) in ((640, 480, 512), (1280, 720, 128), (752, 480, 128), (333, 251, 128), (1920, 1080, 32)):
        for mode in (0, 1, 2):
            bad += sweep(w, h, n, mode, 50000 + 1000 * mode)
    print("TOTAL mismatching pairs:", bad)
    sys.exit(1 if bad else 0)
