#!/usr/bin/env python3
"""Measurement for SURVEY.md 8f row 3 (KeyFrameMatchDatabase scoring): one query frame against N resident keyframes.

GPU: msf_store_frame(query) + msf_match_one_to_many (extract the query once, N matches, N map-point counts, 2 int32 per
keyframe back to the host).  CPU baseline: the reference's loop -- N x MatchFrames(query, KF_i), each re-extracting both
frames -- on the CPU oracle, over a bounded sample of keyframes.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--matcher", default="orb", choices=["orb", "loftr"])
    ap.add_argument("--keyframes", type=int, default=None)
    ap.add_argument("--queries", type=int, default=20)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--cpu-sample", type=int, default=8)
    a = ap.parse_args()
    from mono_slam_framework_amd import synth
    from mono_slam_framework_amd.matcher import DNNFeatureMatcher, FeatureMatcher
    orb = a.matcher == "orb"
    W = a.width or (1280 if orb else 640)
    H = a.height or (720 if orb else 480)
    N = a.keyframes or (1024 if orb else 128)
    fm = FeatureMatcher(0.6, W, H, max_batch_pairs=N) if orb else DNNFeatureMatcher(None, 0.15, W, H, max_batch_pairs=N)
    rng = np.random.RandomState(1)
    # keyframes: views of N/4 scenes; the queries revisit some of them
    frames = []
    for i in range(N):
        frames.append(synth.synth_pair(2000 + i // 4, W, H, mode=0 if orb else 1,
                                       shift=(int(rng.randint(-30, 31)), int(rng.randint(-30, 31))))[1])
    keys = [rng.randint(0, W * H, 400) for _ in range(N)]
    t0 = time.perf_counter()
    for i, f in enumerate(frames):
        fm.store_frame(1 + i, f)          # upload + per-frame extraction (ORB features / LoFTR tokens)
        fm.set_mappoints(1 + i, keys[i])
    t_add = (time.perf_counter() - t0) / N
    slots = np.arange(1, N + 1, dtype=np.int32)
    queries = [synth.synth_pair(2000 + int(rng.randint(0, N // 4)), W, H, mode=0 if orb else 1, shift=(5, -7))[1]
               for _ in range(a.queries)]
    fm.set_mappoints(0, rng.choice(W * H, 400, replace=False))
    fm.store_frame(0, queries[0])
    fm.match_one_to_many(0, slots, True)
    t0 = time.perf_counter()
    tot = 0
    for q in queries:
        fm.store_frame(0, q)
        num, nmp, _ = fm.match_one_to_many(0, slots, True)
        tot += int(np.clip(num, 0, None).sum())
    dt = (time.perf_counter() - t0) / len(queries)

    # CPU: the reference loop on the oracle, bounded sample
    if orb:
        from oracle import orb as oracle_orb
        orc = oracle_orb.FeatureMatcherOracle(0.6)
    else:
        from oracle import loftr as oracle_loftr
        orc = oracle_loftr.DNNFeatureMatcherOracle(0.15)
    t0 = time.perf_counter()
    for i in range(a.cpu_sample):
        orc.MatchFrames(queries[0], frames[i])
    cpu_pair = (time.perf_counter() - t0) / a.cpu_sample
    print(json.dumps({
        "workload": "%s one query vs %d resident keyframes, %dx%d" % (a.matcher.upper(), N, W, H),
        "query_ms": round(dt * 1e3, 3), "keyframe_pairs_per_s": round(N / dt, 1),
        "add_keyframe_ms": round(t_add * 1e3, 3), "mean_matches_per_pair": round(tot / len(queries) / N, 2),
        "cpu_reference_loop": {"pairs_per_s": round(1 / cpu_pair, 2), "cores": 1 if orb else "OpenMP default",
                               "kind": "port", "sample": "%d MatchFrames calls of the same query" % a.cpu_sample},
    }))


if __name__ == "__main__":
    main()
