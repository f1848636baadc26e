import time, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mono_slam_framework_amd import synth, _lib
from mono_slam_framework_amd.matcher import FeatureMatcher
fm = FeatureMatcher(0.6, 640, 480, flags=_lib.MSF_FLAG_PROFILE)
a, b = synth.synth_pair(5, 640, 480)
for _ in range(20): fm.MatchFrames(a, b)
t0 = time.perf_counter()
for _ in range(200): fm.MatchFrames(a, b)
dt = (time.perf_counter() - t0) / 200
st = fm.stage_times()
print("orb latency ms", round(dt * 1e3, 3), "stages", {k: round(v, 3) for k, v in st.items()}, "sum", round(sum(st.values()), 3))
