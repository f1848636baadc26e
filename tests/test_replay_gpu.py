"""GPU: config-5 call-pattern replay (tools/replay_sequence.py, BASELINE.json configs[4]) at its full 1000 frames.  The
match lists the tracking / mapping loops would receive reproduce the known camera motion; the sequence issued as plain
MatchFrames calls on host images (what the untouched pipeline does: Tracking.cc:383,444,595-632, KeyFrameDatabase.cc:31-50,
LocalMapping.cc:325-358), served by the transparent frame cache, gives the very lists of the extract-once / match-many
path; a camera that also rotates and changes scale exercises steered rBRIEF and cross-octave matching end to end; and a
short LoFTR sequence with cell-aligned motion lands every match on the true cell."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_replay_1000_frames_slot_path_and_plain_matchframes_give_identical_lists():
    import replay_sequence
    n = 1000
    slots, la = replay_sequence.replay(n_frames=n, check_every=50, db_size=40, keep_lists=True)
    plain, lb = replay_sequence.replay(n_frames=n, db_size=40, mode="plain", keep_lists=True)
    assert slots["calls"] == plain["calls"] > 6000 and len(la) == len(lb) == slots["calls"]
    for k, (a, b) in enumerate(zip(la, lb)):
        assert a.shape == b.shape and np.array_equal(a, b), "call %d: slot path and plain MatchFrames differ" % k
    assert slots["inlier_ratio"] > 0.97 and slots["matches"] > 1_000_000, slots
    fc = plain["frame_cache"]
    # every MatchFrames call looks up two frames; a frame is extracted once, when it is first seen (identical consecutive
    # frames at the camera's turning points are hits: the cache is keyed by content)
    assert fc["hits"] + fc["misses"] == 2 * plain["calls"]
    assert 0.98 * n <= fc["misses"] <= n, fc
    # short lists ("lost": fewer than minLocalMatchCount = 15 matches against a local key frame) are not a defect of the
    # matcher: cv::ORB keeps the 500 strongest corners of the whole frame, and where the part two views share is the
    # weakly textured part of both, few of either frame's key points lie in it
    assert slots["lost"] < 0.06 * slots["calls"], slots
    assert slots["lost_max_common_keypoints"] < 100, slots       # of 500 per frame (tools/replay_sequence.py: common_keypoints)


def test_replay_with_rotation_and_scale_change():
    """in-plane rotation up to +-30 degrees and a height change of 1.3x either way (scale ratios up to 1.69 between a
    frame and a key frame): steered rBRIEF and cross-octave matches against the known similarity"""
    import replay_sequence
    st = replay_sequence.replay(n_frames=400, rotate_deg=30.0, zoom=1.3, check_every=20)
    assert st["calls"] > 3000 and st["cache_checks"] >= 30
    assert st["inlier_ratio"] > 0.95 and st["matches_per_call"] > 60, st
    rot = replay_sequence.replay(n_frames=200, rotate_deg=45.0, check_every=50)
    assert rot["inlier_ratio"] > 0.95, rot


def test_replay_loftr_short_sequence():
    """LoFTR through the same call pattern (token cache per frame slot): with cell-aligned camera motion every match
    joins a cell with its true image, and the slot path equals plain MatchFrames"""
    import replay_sequence
    st = replay_sequence.replay(n_frames=60, matcher="loftr", db_size=16, check_every=10)
    assert st["calls"] > 150 and st["cache_checks"] >= 8 and st["matches"] > 15000
    assert st["inlier_ratio"] >= 0.9999, st
