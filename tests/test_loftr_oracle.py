"""CPU: the LoFTR C restatement (oracle/loftr_oracle.c) against the golden vectors produced from the reference's
model/LoFTR_teacher.onnx by oracle/onnx_oracle.py (tools/make_loftr_fixtures.py), and against the known-answer
numbers recorded in SURVEY.md section 8c.  This is what pins the LoFTR oracle."""
import os

import numpy as np
import pytest

from oracle import loftr

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "loftr_kat.npz"))
TOL = 1e-4  # restatement vs ONNX interpretation, f32 summation-order noise only (observed <= 3e-6)


@pytest.fixture(scope="module")
def orc():
    return loftr.DNNFeatureMatcherOracle(0.15)


@pytest.mark.parametrize("name", ["i", "ii", "iii", "synth"])
def test_restatement_matches_onnx_golden(orc, name):
    r = orc.run(G["img0_" + name], G["img1_" + name])
    conf = r["conf"]
    bi, bv = G["big_ij_" + name].astype(int), G["big_v_" + name]
    si, sv = G["samp_ij_" + name].astype(int), G["samp_v_" + name]
    if len(bv):
        assert np.abs(conf[bi[:, 0], bi[:, 1]] - bv).max() < TOL
        # nothing above 1e-3 + tol that the golden does not list
        big = np.argwhere(conf > 1e-3 + TOL)
        assert set(map(tuple, big)) <= set(map(tuple, bi))
    assert np.abs(conf[si[:, 0], si[:, 1]] - sv).max() < TOL
    assert np.abs(conf.sum(1) - G["rowsum_" + name]).max() < 1e-3
    assert np.abs(conf.sum(0) - G["colsum_" + name]).max() < 1e-3
    assert np.abs(r["feat0"] - G["feat0_" + name]).max() < 1e-3
    assert np.abs(r["feat1"] - G["feat1_" + name]).max() < 1e-3
    for thr, tag in ((0.15, "015"), (0.1, "010")):
        np.testing.assert_array_equal(orc.decode(conf, thr), G["matches_%s_%s" % (name, tag)])


def test_survey_known_answers():
    # (i) zeros; (ii) P vs P shifted (32,16); (iii) P vs P
    s = G["stats_i"]
    assert s[6] == 0 and s[7] == 0 and abs(s[0] - 0.004814) < 1e-6 and (s[2], s[3]) == (23, 23)
    assert abs(s[1] - 5.9520) < 1e-3 and abs(s[4] - 3.713395e-3) < 1e-8 and abs(s[5] - 2.474234) < 1e-5
    s = G["stats_ii"]
    assert (s[6], s[7]) == (41, 82) and abs(s[0] - 0.470346) < 1e-6 and (s[2], s[3]) == (860, 818)
    assert abs(s[1] - 55.8890) < 1e-3
    m = G["matches_ii_015"]
    assert len(m) == 41 and set(map(tuple, (m[:, 2:] - m[:, :2]) // 16)) == {(-2, -1)}
    s = G["stats_iii"]
    assert (s[6], s[7]) == (56, 134) and abs(s[0] - 0.489238) < 1e-6 and (s[2], s[3]) == (852, 852)


def test_decode_convention():
    # dnnfeaturematcher.cpp:88-99: row i -> (x0,y0), col j -> (x1,y1), top-left cell corners, strict '>'
    conf = np.zeros((1200, 1200), np.float32)
    conf[41, 85] = 0.2
    conf[41, 3] = 0.15   # not strictly greater
    conf[0, 1199] = 0.9
    o = loftr.DNNFeatureMatcherOracle(0.15)
    m = o.decode(conf)
    np.testing.assert_array_equal(m, [[0, 0, 39 * 16, 29 * 16], [16, 16, 5 * 16, 2 * 16]])
