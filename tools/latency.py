#!/usr/bin/env python3
"""Single-pair drop-in latency (msf_match_pair on host images, the call the reference app makes per MatchFrames) and
the per-stage kernel times of that call.  Prints one line per matcher."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mono_slam_framework_amd import _lib, synth                                     # noqa: E402
from mono_slam_framework_amd.matcher import DNNFeatureMatcher, FeatureMatcher       # noqa: E402

NOCACHE = _lib.MSF_FLAG_NO_FRAME_CACHE if "--no-cache" in sys.argv else 0   # --no-cache: both frames extracted on every call
for name, make in (("orb", lambda: FeatureMatcher(0.6, 640, 480, flags=_lib.MSF_FLAG_PROFILE | NOCACHE)),
                   ("loftr", lambda: DNNFeatureMatcher(None, 0.15, 640, 480, flags=_lib.MSF_FLAG_PROFILE | NOCACHE))):
    fm = make()
    a, b = synth.synth_pair(5, 640, 480, mode=0 if name == "orb" else 1)
    for _ in range(20):
        fm.MatchFrames(a, b)
    fm.stage_times()                 # msf_stage_times sums over the calls since the last query: start the sum here
    t0 = time.perf_counter()
    for _ in range(200):
        fm.MatchFrames(a, b)
    dt = (time.perf_counter() - t0) / 200
    st = {k: v / 200 for k, v in fm.stage_times().items()}      # mean per call
    print(name, "(stateless)" if NOCACHE else "(frames cached)", "640x480 MatchFrames latency %.3f ms; kernel stages (ms):" % (dt * 1e3),
          {k: round(v, 3) for k, v in st.items()}, "sum %.3f" % sum(st.values()))
