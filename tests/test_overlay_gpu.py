"""GPU: Tracking::CreateCurrentMatchImage on the device (msf_render_match_image) vs the CPU
restatement of the reference function + OpenCV's filled-circle loop (oracle/overlay.py): identical images."""
import numpy as np
import pytest

from mono_slam_framework_amd import synth
from oracle import overlay as oracle_overlay

pytestmark = pytest.mark.gpu

W, H = 640, 480


def test_match_image_equals_reference_drawing():
    from mono_slam_framework_amd.matcher import DNNFeatureMatcher, FeatureMatcher, MsfError
    fm = FeatureMatcher(0.7, W, H)
    a, b = synth.synth_pair(444, W, H, shift=(11, -6))
    m = fm.MatchFrames(a, b)
    assert len(m) > 100
    rng = np.random.RandomState(1)
    mp1, mp2 = rng.rand(len(m)) < 0.4, rng.rand(len(m)) < 0.4
    got = fm.render_match_image(a, b, m, mp1, mp2)
    exp = oracle_overlay.create_current_match_image(a, b, m, mp1, mp2)
    np.testing.assert_array_equal(got, exp)
    assert (got[:, :, 1] == 255).sum() > 100 and ((got[:, :, 0] == 255) & (got[:, :, 1] == 0)).sum() > 100
    # no flags: every circle is green; no matches: the plain side-by-side image
    np.testing.assert_array_equal(fm.render_match_image(a, b, m),
                                  oracle_overlay.create_current_match_image(a, b, m, np.zeros(len(m)), np.zeros(len(m))))
    plain = fm.render_match_image(a, b, np.zeros((0, 4), np.int32))
    np.testing.assert_array_equal(plain[:, :W, 2], a)
    np.testing.assert_array_equal(plain[:, W:, 0], b)
    # circles clipped at the border and overlapping each other (LoFTR cells sit on multiples of 16, also x = 0 / y = 0);
    # strided input frames (cv::Mat step > width)
    dm = DNNFeatureMatcher(None, 0.15, W, H)
    edge = np.array([[0, 0, 639, 479], [16, 0, 0, 464], [624, 464, 1, 1], [2, 2, 3, 3], [5, 3, 636, 477],
                     [-2, 5, 700, 10], [100, -1, 10, 481]], np.int32)
    f1 = np.array([1, 0, 0, 1, 0, 0, 1], bool)
    f2 = np.array([0, 0, 1, 0, 0, 1, 0], bool)
    big = np.zeros((H, 704), np.uint8)
    big[:, :W] = a
    got = dm.render_match_image(big[:, :W], b, edge, f1, f2)
    np.testing.assert_array_equal(got, oracle_overlay.create_current_match_image(a, b, edge, f1, f2))
    with pytest.raises(MsfError):
        fm.render_match_image(a[:100], b, m)
