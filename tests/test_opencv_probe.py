"""bench.py's OpenCV probe on a stand-in: no OpenCV exists in the build container or on the GPU
boxes, so the probe's plumbing -- the literal reference call sequence, the sweep over the restatement's open choices, the
verdict on which combination reproduces the library -- is exercised here against a fake `cv2` module that answers with
the oracle under a NON-default combination.  The probe must name exactly that combination.  (A stand-in for the test of
the probe; it pins nothing about the real library.)"""
import argparse
import sys
import types

import numpy as np

from mono_slam_framework_amd import synth
from oracle import orb as oracle_orb


class _KP:
    def __init__(self, q):
        self.pt = (float(q["x"]), float(q["y"]))
        self.octave = int(q["octave"])
        self.angle = float(q["angle"])
        self.response = float(q["response"])


class _DM:
    def __init__(self, q, t, d):
        self.queryIdx, self.trainIdx, self.distance = q, t, float(d)


def _fake_cv2(**switches):
    m = types.ModuleType("cv2")
    m.__version__ = "0.0-standin"
    m.NORM_HAMMING = 6

    class _Orb:
        def detectAndCompute(self, img, mask):
            assert mask.shape == img.shape and (mask == 255).all()          # the reference's all-255 mask (featurematcher.cpp:12)
            k, d = oracle_orb.OrbOracle(img.shape[1], img.shape[0], **switches).extract(img)
            return [_KP(q) for q in k], (d if len(k) else None)

    class _BF:
        def knnMatch(self, d1, d2, k=2):
            assert k == 2
            nn = oracle_orb.knn2(d1, d2)
            return [[_DM(q, r[0], r[1]), _DM(q, r[2], r[3])] for q, r in enumerate(nn)]

    m.ORB_create = lambda: _Orb()
    m.BFMatcher = lambda norm: _BF()
    return m


def test_probe_names_the_combination_the_library_uses(monkeypatch):
    import bench
    truth = {"blur_tie_even": 0, "level_size_mul_inv": 1, "blur_kernel_sum256": 1}
    monkeypatch.setitem(sys.modules, "cv2", _fake_cv2(**truth))
    w, h = 237, 153                       # a size for which the two level-size formulas differ (tests/test_orb_gpu.py lists them)
    A, B = synth.synth_batch(700, 3, w, h, mode=0)
    default = oracle_orb.FeatureMatcherOracle(0.8)
    gpu_lists = [default.MatchFrames(A[i], B[i]) for i in range(3)]       # what a GPU run with default flags returns
    r = bench.opencv_probe(argparse.Namespace(ratio=0.8), A, B, gpu_lists)
    assert r["opencv"] == "0.0-standin" and r["pairs"] == 3 and len(r["sweep"]) == 8
    assert r["best_switches"] == truth and r["best_is_exact"] and not r["default_is_best"]
    d = r["restatement_default_vs_opencv"]
    assert d["frames_with_identical_keypoint_set"] < d["of_frames"] or d["descriptors_identical"] < d["of_common_keypoints"]
    # with the default combination as the truth the default is best, exact, and the GPU lists agree
    monkeypatch.setitem(sys.modules, "cv2", _fake_cv2())
    r = bench.opencv_probe(argparse.Namespace(ratio=0.8), A, B, gpu_lists)
    assert r["default_is_best"] and r["best_is_exact"]
    assert r["gpu_default_vs_opencv"]["identical_ordered_lists"] == 3


def test_probe_reports_absence():
    import bench
    sys.modules.pop("cv2", None)
    assert bench.opencv_probe(argparse.Namespace(ratio=0.8), [], [], []) == {"opencv": "absent"}
