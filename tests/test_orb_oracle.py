"""CPU: the ORB oracle (oracle/orb_oracle.c).  The reference holds no golden vectors for this path and OpenCV
is not installed (parity unpinned); these tests pin the restatement to the published constants
(level sizes, per-level quotas, umax, the rBRIEF table) and to properties the algorithm must have."""
import os

import numpy as np
import pytest

from mono_slam_framework_amd import synth
from oracle import orb


def test_level_sizes_and_quotas():
    o = orb.OrbOracle(640, 480)
    assert [o.level_size(l) for l in range(8)] == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231),
                                                   (257, 193), (214, 161), (179, 134)]
    assert [o.level_quota(l) for l in range(8)] == [109, 90, 75, 63, 52, 44, 36, 31]
    o = orb.OrbOracle(1280, 720)
    assert [o.level_size(l) for l in range(8)] == [(1280, 720), (1067, 600), (889, 500), (741, 417), (617, 347),
                                                   (514, 289), (429, 241), (357, 201)]
    assert sum(o.level_quota(l) for l in range(8)) == 500


def test_pattern_table_fixture():
    pat = np.loadtxt(os.path.join(os.path.dirname(__file__), "golden", "orb_bit_pattern_31.txt"), dtype=int)
    assert pat.shape == (256, 4)
    assert pat[0].tolist() == [8, -3, 9, 5] and pat[-1].tolist() == [-1, -6, 0, -11]
    for hdr in ("oracle/orb_pattern.h", "mono_slam_framework_amd/csrc/orb_pattern.h"):
        txt = open(os.path.join(os.path.dirname(os.path.dirname(__file__)), hdr)).read()
        body = txt[txt.index("{") + 1:txt.rindex("}")]
        vals = [int(v) for v in body.replace("\n", " ").split(",") if v.strip()]
        assert vals == pat.ravel().tolist(), hdr


def test_fast_atan2_and_sincos():
    L = orb.lib()
    rng = np.random.default_rng(0)
    for _ in range(2000):
        y, x = rng.integers(-200000, 200000, 2)
        a = L.orb_oracle_fast_atan2(float(y), float(x))
        ref = np.degrees(np.arctan2(y, x)) % 360.0
        d = abs(a - ref)
        assert min(d, 360 - d) < 0.3          # cv::fastAtan2 is documented accurate to ~0.3 deg
    worst = 0.0
    for t in np.linspace(0, 2 * np.pi, 5001, dtype=np.float32):
        s, c = orb.sincosf(float(t))
        worst = max(worst, abs(s - np.sin(np.float64(t))), abs(c - np.cos(np.float64(t))))
        assert np.float32(s) == np.float32(np.sin(np.float64(t))) and np.float32(c) == np.float32(np.cos(np.float64(t)))
    assert worst < 1e-7


def test_extract_properties():
    a, b = synth.synth_pair(1, 640, 480)
    o = orb.OrbOracle(640, 480)
    k, d = o.extract(a)
    assert 400 <= len(k) <= 600 and d.shape == (len(k), 32)
    # canonical order: (level, y, x)
    key = k["octave"].astype(np.int64) << 40 | k["ly"].astype(np.int64) << 20 | k["lx"]
    assert np.all(np.diff(key) > 0)
    for l in range(8):
        w, h = o.level_size(l)
        kl = k[k["octave"] == l]
        assert len(kl) >= min(o.level_quota(l), len(o.fast_candidates(l)))
        assert np.all((kl["lx"] >= 31) & (kl["lx"] < w - 31) & (kl["ly"] >= 31) & (kl["ly"] < h - 31))
        np.testing.assert_array_equal(kl["x"], (kl["lx"].astype(np.float32) * np.float32(o.level_scale(l))))
    assert np.all((k["angle"] >= 0) & (k["angle"] <= 360))
    # progressive pyramid keeps the mean
    for l in range(1, 8):
        assert abs(float(o.level_pixels(l).mean()) - float(a.mean())) < 2.0
    # blur is a smoothing of the level: same mean, lower variance
    assert o.level_pixels(0, blurred=True).var() < o.level_pixels(0).var()


def test_determinism_and_shift_consistency():
    a, b = synth.synth_pair(2, 640, 480, shift=(10, -6))
    fm = orb.FeatureMatcherOracle(0.6)
    m1 = fm.MatchFrames(a, b)
    m2 = fm.MatchFrames(a, b)
    np.testing.assert_array_equal(m1, m2)
    assert len(m1) > 50
    d = m1[:, 2:] - m1[:, :2]
    # frame B is the canvas shifted by (+10, -6): features move by (-10, +6), up to the level scale
    good = (np.abs(d[:, 0] + 10) <= 4) & (np.abs(d[:, 1] - 6) <= 4)
    assert good.mean() > 0.9


def test_knn_tie_break_and_ratio():
    d2 = np.zeros((3, 32), np.uint8)
    d2[0, 0] = 0b1           # distance 1 from zero
    d2[1, 0] = 0b10          # distance 1 too (tie -> lower index first)
    d2[2, :2] = 0xFF         # distance 16
    d1 = np.zeros((1, 32), np.uint8)
    nn = orb.knn2(d1, d2)
    assert nn.tolist() == [[0, 1, 1, 1]]
    kp = np.zeros((3,), orb.KP_DTYPE)
    kp["x"] = [1.9, 5.5, 7.2]
    kp["y"] = [2.9, 6.5, 8.2]
    assert len(orb.knn_match(kp[:1], d1, kp, d2, 0.8)) == 0         # 1 < 0.8 * 1 is false
    d2[1, 0] = 0b111         # distances 1, 3
    m = orb.knn_match(kp[:1], d1, kp, d2, 0.8)
    assert m.tolist() == [[1, 2, 1, 2]]                              # truncation of the f32 coordinates
    assert len(orb.knn_match(kp[:1], d1, kp[:1], d2[:1], 0.8)) == 0  # single train descriptor: defined as no match


def test_empty_and_degenerate_inputs():
    o = orb.OrbOracle(640, 480)
    k, d = o.extract(np.full((480, 640), 128, np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)
    fm = orb.FeatureMatcherOracle(0.8)
    assert fm.MatchFrames(np.zeros((480, 640), np.uint8), np.zeros((480, 640), np.uint8)).shape == (0, 4)
    small = orb.OrbOracle(96, 80)
    k, d = small.extract(synth.synth_pair(0, 96, 80)[0])
    assert all(kk["octave"] < 3 for kk in k)   # upper levels are smaller than 2*31 px: cleared by runByImageBorder


def test_blur_tie_modes_differ_only_on_exact_ties():
    a, _ = synth.synth_pair(4, 640, 480)
    e = orb.OrbOracle(640, 480, blur_tie_even=1)
    u = orb.OrbOracle(640, 480, blur_tie_even=0)
    e.extract(a)
    u.extract(a)
    diff = e.level_pixels(0, blurred=True).astype(int) - u.level_pixels(0, blurred=True).astype(int)
    assert set(np.unique(diff)) <= {-1, 0}
    assert (diff != 0).mean() < 1e-3
