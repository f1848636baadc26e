#!/usr/bin/env python3
"""Mean FAST threshold (final / sampler's) and candidate count per level for one batch -- diagnostic of the walker's
in-launch refinement under the environment it is started with."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mono_slam_framework_amd import synth, _lib
from mono_slam_framework_amd.matcher import FeatureMatcher
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
w, h = 1280, 720
A, B = synth.synth_batch(5000, n, w, h, mode=0)
fm = FeatureMatcher(0.7, w, h, max_batch_pairs=n, flags=_lib.MSF_FLAG_NO_FRAME_CACHE)
got = fm.match_batch(list(A), list(B), cap=1024)
sl = list(range(0, 2 * n, max(1, 2 * n // 32)))
tau = np.stack([fm.fast_tau(s) for s in sl])
cnt = np.array([[len(fm.fast_candidates(s, l)) for l in range(8)] for s in sl[:8]])
print("final tau  ", np.round(tau[:, :, 0].mean(0), 1))
print("2nd column ", np.round(tau[:, :, 1].mean(0), 1))
print("cands >=tau", np.round(cnt.mean(0), 0), "matches", sum(len(m) for m in got))
