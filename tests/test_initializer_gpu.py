"""GPU: Initializer::CheckHomography / CheckFundamental for all RANSAC hypotheses at once (SURVEY.md 8f row 4,
msf_check_hypotheses) vs the scalar restatement of slam_pipeline/src/Initializer.cc:152-245, 322-487.

Bar: f32 scores identical bit for bit (the reference accumulates in match order), the same kept hypothesis, the same
vbMatchesInliers.  Hypotheses are made the way FindHomography / FindFundamental make them (8 random matches,
Normalize, DLT through an SVD) -- here with numpy's SVD, they are inputs to both sides."""
import numpy as np
import pytest

from mono_slam_framework_amd import synth
from oracle import initializer as oracle_init

pytestmark = pytest.mark.gpu

W, H = 640, 480


def _normalize(p):
    """Initializer::Normalize (Initializer.cc:757-798): mean / mean absolute deviation"""
    p = p.astype(np.float32)
    mean = p.mean(0, dtype=np.float32)
    d = p - mean
    s = (1.0 / np.abs(d).mean(0, dtype=np.float32)).astype(np.float32)
    T = np.array([[s[0], 0, -mean[0] * s[0]], [0, s[1], -mean[1] * s[1]], [0, 0, 1]], np.float32)
    return d * s, T


def _hypotheses(matches, n_hyp, seed):
    rng = np.random.RandomState(seed)
    p1, p2 = matches[:, :2], matches[:, 2:]
    n1, T1 = _normalize(p1)
    n2, T2 = _normalize(p2)
    Hs, His, Fs = [], [], []
    for _ in range(n_hyp):
        idx = rng.choice(len(matches), 8, replace=False)
        a, b = n1[idx], n2[idx]
        A = np.zeros((16, 9), np.float32)
        for k in range(8):
            u1, v1, u2, v2 = a[k, 0], a[k, 1], b[k, 0], b[k, 1]
            A[2 * k] = [0, 0, 0, -u1, -v1, -1, v2 * u1, v2 * v1, v2]
            A[2 * k + 1] = [u1, v1, 1, 0, 0, 0, -u2 * u1, -u2 * v1, -u2]
        Hn = np.linalg.svd(A)[2][8].reshape(3, 3).astype(np.float32)
        H21 = (np.linalg.inv(T2) @ Hn @ T1).astype(np.float32)
        with np.errstate(all="ignore"):
            try:
                H12 = np.linalg.inv(H21).astype(np.float32)
            except np.linalg.LinAlgError:
                H12 = np.full((3, 3), np.inf, np.float32)
        B = np.stack([b[:, 0] * a[:, 0], b[:, 0] * a[:, 1], b[:, 0], b[:, 1] * a[:, 0], b[:, 1] * a[:, 1], b[:, 1],
                      a[:, 0], a[:, 1], np.ones(8, np.float32)], 1).astype(np.float32)
        Fp = np.linalg.svd(B)[2][8].reshape(3, 3)
        u, w, vt = np.linalg.svd(Fp)
        w[2] = 0
        Fn = (u @ np.diag(w) @ vt).astype(np.float32)
        Fs.append((T2.T @ Fn @ T1).astype(np.float32))
        Hs.append(H21)
        His.append(H12)
    return np.stack(Hs), np.stack(His), np.stack(Fs)


def _same(got, exp):
    gb, gs, gi = got
    eb, es, ei = exp
    # bit for bit; a NaN score must be NaN on both sides (its sign bit is the FPU's default-NaN convention: x86 SSE
    # produces 0xFFC00000, gfx950 0x7FC00000 -- no comparison in the reference can tell them apart)
    np.testing.assert_array_equal(np.isnan(gs), np.isnan(es))
    ok = ~np.isnan(es)
    np.testing.assert_array_equal(gs.view(np.uint32)[ok], es.view(np.uint32)[ok])
    assert gb == eb
    np.testing.assert_array_equal(gi, ei)


def test_check_hypotheses_bit_exact():
    from mono_slam_framework_amd.matcher import FeatureMatcher
    fm = FeatureMatcher(0.7, W, H)
    a, b = synth.synth_pair(321, W, H, shift=(17, -9))
    m = fm.MatchFrames(a, b)
    assert len(m) > 100
    rng = np.random.RandomState(2)
    # a third of the matches become outliers, as after a real ratio test
    bad = rng.rand(len(m)) < 0.33
    m = m.copy()
    m[bad, 2:] = np.stack([rng.randint(0, W, bad.sum()), rng.randint(0, H, bad.sum())], 1)
    H21, H12, F21 = _hypotheses(m, 200, 5)                               # mMaxIterations = 200 (Initializer.cc:20)
    for sigma in (1.0, 2.5):
        got = fm.check_hypotheses(0, H21, H12, m, sigma)
        _same(got, oracle_init.find_best(0, H21, H12, m, sigma))
        assert got[0] >= 0 and got[2].sum() > 0.5 * (~bad).sum()          # a translation is found
        got = fm.check_hypotheses(1, F21, None, m, sigma)
        _same(got, oracle_init.find_best(1, F21, None, m, sigma))
        assert got[0] >= 0

    # degenerate hypotheses: zero / huge / NaN / infinite entries -> divisions by zero, NaN scores (never kept)
    weird = np.stack([np.zeros((3, 3)), np.full((3, 3), 1e30), np.full((3, 3), np.nan), np.eye(3) * np.inf,
                      np.eye(3), -np.eye(3), H21[0], np.eye(3) * 1e-30]).astype(np.float32)
    _same(fm.check_hypotheses(0, weird, weird[::-1].copy(), m, 1.0), oracle_init.find_best(0, weird, weird[::-1].copy(), m, 1.0))
    _same(fm.check_hypotheses(1, weird, None, m, 1.0), oracle_init.find_best(1, weird, None, m, 1.0))
    # nothing scores above zero -> no hypothesis kept, all-false inliers
    none = np.full((5, 3, 3), np.nan, np.float32)
    got = fm.check_hypotheses(0, none, none, m, 1.0)
    assert got[0] == -1 and not got[2].any()
    _same(got, oracle_init.find_best(0, none, none, m, 1.0))


def test_check_hypotheses_sizes():
    from mono_slam_framework_amd.matcher import FeatureMatcher, MsfError
    fm = FeatureMatcher(0.7, W, H)
    rng = np.random.RandomState(9)
    for n in (0, 1, 7, 255, 256, 257, 2049, 8192):                        # workspace regrowth, LDS maximum
        p1 = np.stack([rng.randint(0, W, n), rng.randint(0, H, n)], 1)
        m = np.concatenate([p1, p1 + rng.randint(-2, 3, (n, 2)) + [5, 3]], 1).astype(np.int32)
        Hs = np.tile(np.array([[1, 0, 5], [0, 1, 3], [0, 0, 1]], np.float32), (3, 1, 1))
        Hs[1, 0, 2] = 4.0
        Hs[2] = 0
        His = np.stack([np.linalg.inv(h).astype(np.float32) if np.linalg.det(h) else h for h in Hs])
        _same(fm.check_hypotheses(0, Hs, His, m, 1.0), oracle_init.find_best(0, Hs, His, m, 1.0))
        Fs = rng.randn(300, 3, 3).astype(np.float32) * 1e-3              # more hypotheses than the first allocation
        _same(fm.check_hypotheses(1, Fs, None, m, 1.0), oracle_init.find_best(1, Fs, None, m, 1.0))
    got = fm.check_hypotheses(0, np.zeros((0, 3, 3), np.float32), np.zeros((0, 3, 3), np.float32), m, 1.0)
    assert got[0] == -1 and len(got[1]) == 0
    with pytest.raises(MsfError):
        fm.check_hypotheses(0, Hs, His, np.zeros((8193, 4), np.int32), 1.0)
    with pytest.raises(MsfError):
        fm.check_hypotheses(2, Hs, His, m, 1.0)
    with pytest.raises(MsfError):
        fm.check_hypotheses(0, Hs, None, m, 1.0)                          # a homography needs H12
