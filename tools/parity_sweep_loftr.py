#!/usr/bin/env python3
"""One-off LoFTR parity sweep (not part of the test suite): full 1200x1200 confidence matrix of the HIP path vs the
f32 C restatement of the ONNX graph, and the match lists wherever no confidence sits within the tolerance of the
threshold, over synthetic pairs of the three texture modes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mono_slam_framework_amd import synth                         # noqa: E402
from mono_slam_framework_amd.matcher import DNNFeatureMatcher     # noqa: E402
from oracle import loftr as oracle_loftr                          # noqa: E402

TOL = 1e-3
thr = 0.15
dm = DNNFeatureMatcher(None, thr, 640, 480, flags=4 | 16)   # MSF_FLAG_KEEP_DEBUG | MSF_FLAG_NO_FRAME_CACHE
orc = oracle_loftr.DNNFeatureMatcherOracle(thr)
worst = 0.0
bad_lists = 0
n = 0
for mode in (0, 1, 2):
    for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
        a, b = synth.synth_pair(70000 + 100 * mode + k, 640, 480, mode=mode)
        got = dm.MatchFrames(a, b, cap=8192)
        conf = dm.conf_matrix()
        ref = orc.run(a, b)["conf"]
        d = float(np.abs(conf.reshape(1200, 1200) - ref).max())
        worst = max(worst, d)
        sure = orc.decode(ref, thr + TOL)
        maybe = orc.decode(ref, thr - TOL)
        gs = set(map(tuple, got.tolist()))
        ok = set(map(tuple, sure.tolist())) <= gs <= set(map(tuple, maybe.tolist()))
        bad_lists += 0 if ok else 1
        n += 1
    print("mode %d done: worst |conf diff| so far %.3g, list violations %d of %d pairs" % (mode, worst, bad_lists, n), flush=True)
print("TOTAL pairs %d, worst |conf_gpu - conf_oracle| = %.3g (bar %.0e), list violations %d" % (n, worst, TOL, bad_lists))
sys.exit(1 if (worst > TOL or bad_lists) else 0)
