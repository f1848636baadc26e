"""GPU: config-5 call-pattern replay (tools/replay_sequence.py) on a short synthetic monocular sequence: the match lists
the tracking / mapping loops would receive reproduce the known camera motion, and the extract-once / match-many path is
indistinguishable from stateless MatchFrames calls."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_replay_120_frames():
    import replay_sequence
    st = replay_sequence.replay(n_frames=120, check_every=10)
    assert st["calls"] > 300 and st["cache_checks"] >= 10
    assert st["inlier_ratio"] > 0.95, st
    assert st["lost"] == 0, st
