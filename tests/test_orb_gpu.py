"""GPU parity tests for the ORB path: every stage of the HIP pipeline against the CPU oracle (bit-exact),
called through the C ABI (include/msf_abi.h).  Reference path: src/featurematcher.cpp:10-45."""
import os

import numpy as np
import pytest

from mono_slam_framework_amd import synth
from oracle import orb as oracle_orb

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _matcher(w, h, thr=0.8, pairs=1, flags=0):
    # stateless MatchFrames: the stage-level getters below read the scratch slots of the last call
    # (the transparent frame cache has its own tests: tests/test_frame_cache_gpu.py)
    from mono_slam_framework_amd import _lib
    from mono_slam_framework_amd.matcher import FeatureMatcher
    # MSF_FLAG_FAST_STREAM: the streaming FAST pass also for these one-pair calls (it is the batch path by default)
    return FeatureMatcher(thr, w, h, max_batch_pairs=pairs,
                          flags=flags | _lib.MSF_FLAG_NO_FRAME_CACHE | _lib.MSF_FLAG_FAST_STREAM)


def _noise_image(w, h, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(h, w), dtype=np.uint8)


def _gradient_image(w, h):
    y, x = np.mgrid[0:h, 0:w]
    return ((x * 3 + y * 5) % 256).astype(np.uint8)


def _cases():
    c = []
    for (w, h) in [(640, 480), (1280, 720), (333, 257)]:
        for mode in (0, 1, 2):
            a, b = synth.synth_pair(7 + mode, w, h, mode=mode)
            c.append(("synth%d_%dx%d" % (mode, w, h), a, b))
    c.append(("noise_640", _noise_image(640, 480, 1), _noise_image(640, 480, 2)))
    c.append(("gradient_640", _gradient_image(640, 480), _gradient_image(640, 480)))
    a, b = synth.synth_pair(3, 640, 480, mode=0, noise=0)      # exact ties in FAST score and Harris
    c.append(("blocky_noiseless", a, b))
    c.append(("constant", np.full((480, 640), 77, np.uint8), np.full((480, 640), 77, np.uint8)))
    return c


CASES = _cases()


def _sorted_cands(c):
    c = np.asarray(c).reshape(-1, 3)
    return c[np.lexsort((c[:, 0], c[:, 1]))]


def _cands_at(oracle_cands, tau):
    c = np.asarray(oracle_cands).reshape(-1, 3)
    return c[c[:, 2] >= tau]


@pytest.mark.parametrize("dense", [False, True], ids=["tau", "dense"])
@pytest.mark.parametrize("name,a,b", CASES, ids=[c[0] for c in CASES])
def test_stage_parity(name, a, b, dense):
    """Every stage against the oracle, with the output-sensitive FAST pass (candidate list = the maxima with score >=
    the level's threshold; everything retainBest(2N) keeps is unchanged) and with MSF_FLAG_FAST_DENSE (all maxima)."""
    from mono_slam_framework_amd import _lib
    h, w = a.shape
    fm = _matcher(w, h, flags=_lib.MSF_FLAG_FAST_DENSE if dense else 0)
    got = fm.MatchFrames(a, b)
    orc = oracle_orb.FeatureMatcherOracle(0.8)
    exp = orc.MatchFrames(a, b)
    oa, ob = orc._orb(a.shape)
    for slot, o in ((0, oa), (1, ob)):
        sizes = fm.level_sizes()
        taus = fm.fast_tau(slot)
        if dense:
            assert (taus[:, 0] == 20).all()
        for l in range(8):
            assert (int(sizes[l][0]), int(sizes[l][1])) == o.level_size(l)
            assert int(sizes[l][3]) == o.level_quota(l)
            if l >= 1:
                np.testing.assert_array_equal(fm.level_pixels(slot, l), o.level_pixels(l), err_msg="pyramid L%d" % l)
            assert taus[l][0] >= 20 and taus[l][0] % 2 == 0 and (taus[l][0] == taus[l][1] or taus[l][0] == 20)
            np.testing.assert_array_equal(_sorted_cands(fm.fast_candidates(slot, l)),
                                          _sorted_cands(_cands_at(o.fast_candidates(l), taus[l][0])),
                                          err_msg="FAST candidates L%d (tau %d)" % (l, taus[l][0]))
        s1o = o.stage1_keypoints()
        for l in range(8):
            g = fm.stage1(slot, l)
            e = s1o[s1o["octave"] == l]
            g = g[np.lexsort((g["lx"], g["ly"]))]
            assert len(g) == len(e), "stage1 count L%d" % l
            np.testing.assert_array_equal(g["lx"], e["lx"])
            np.testing.assert_array_equal(g["ly"], e["ly"])
            np.testing.assert_array_equal(g["fast_score"], e["fast_score"])
            np.testing.assert_array_equal(g["response"].view(np.uint32), e["response"].view(np.uint32), err_msg="Harris L%d" % l)
        kg, dg = fm.keypoints(slot), fm.descriptors(slot)
        ko, do = o.extract(a if slot == 0 else b)
        assert len(kg) == len(ko)
        for f in ("lx", "ly", "octave", "fast_score"):
            np.testing.assert_array_equal(kg[f], ko[f], err_msg=f)
        for f in ("x", "y", "response", "angle"):
            np.testing.assert_array_equal(kg[f].view(np.uint32), ko[f].view(np.uint32), err_msg=f)
        np.testing.assert_array_equal(dg, do, err_msg="descriptors")
    np.testing.assert_array_equal(got, exp, err_msg="match list")


def test_fast_threshold_fallback_path():
    """MSF_ORB_FAST_TAU forces a first-pass threshold no level can satisfy (and one that only some can): every level
    must then go through k_fast_check + the dense second pass and still give the oracle's lists bit for bit."""
    import subprocess
    import sys
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from mono_slam_framework_amd import synth\n"
        "from mono_slam_framework_amd.matcher import FeatureMatcher\n"
        "from oracle import orb\n"
        "for (w, h, mode) in ((640, 480, 0), (333, 257, 1), (1280, 720, 2)):\n"
        "    A, B = synth.synth_batch(900, 3, w, h, mode=mode)\n"
        "    fm = FeatureMatcher(0.8, w, h, max_batch_pairs=3, flags=64)\n"
        "    got = fm.match_batch(list(A), list(B))\n"
        "    t = fm.fast_tau(0)\n"
        "    assert (t[:, 1] == int(sys.argv[1])).all(), t\n"
        "    assert (t[:, 0] == 20).any(), t\n"
        "    for i in range(3):\n"
        "        exp = orb.FeatureMatcherOracle(0.8).MatchFrames(A[i], B[i])\n"
        "        assert len(exp) > 20 and np.array_equal(got[i], exp), (w, h, mode, i)\n"
    ) % ROOT
    for tau in ("254", "150"):
        env = dict(os.environ, MSF_ORB_FAST_TAU=tau)
        r = subprocess.run([sys.executable, "-c", code, tau], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr


def test_level_size_formula_switch():
    """cvRound(W / scale_l) (OpenCV 3.x / 4.x; the default) and cvRound(W * (1.f / scale_l)) differ for
    139 widths in [64, 4096]; 69 is the smallest (level 1: 57 vs 58).  Both are exposed and each matches the oracle in
    the same mode on such a size; on the BASELINE sizes the two modes agree."""
    from mono_slam_framework_amd import _lib
    w, h = 69, 91
    a, b = synth.synth_pair(31, w, h, mode=0)
    sizes = {}
    for flag, mul in ((0, 0), (_lib.MSF_FLAG_LEVEL_SIZE_MUL_INV, 1)):
        fm = _matcher(w, h, thr=0.8, flags=flag)
        orc = oracle_orb.FeatureMatcherOracle(0.8, level_size_mul_inv=mul)
        np.testing.assert_array_equal(fm.MatchFrames(a, b), orc.MatchFrames(a, b))
        oa, _ = orc._orb(a.shape)
        sizes[mul] = [(int(r[0]), int(r[1])) for r in fm.level_sizes()]
        assert sizes[mul] == [oa.level_size(l) for l in range(8)]
        for slot, img in ((0, a), (1, b)):
            ko, do = oracle_orb.OrbOracle(w, h, level_size_mul_inv=mul).extract(img)
            assert len(fm.keypoints(slot)) == len(ko)
            np.testing.assert_array_equal(fm.descriptors(slot), do)
    assert sizes[0][1][0] == 57 and sizes[1][1][0] == 58
    for (w, h) in ((640, 480), (1280, 720)):
        s0 = _matcher(w, h).level_sizes()[:, :2]
        s1 = _matcher(w, h, flags=_lib.MSF_FLAG_LEVEL_SIZE_MUL_INV).level_sizes()[:, :2]
        np.testing.assert_array_equal(s0, s1)


def test_streaming_fast_in_the_dense_regime():
    """MSF_ORB_FAST_TAU=22 runs the streaming first pass just above fastThreshold on every level: nearly every pixel
    that is a corner is recorded, so its fixed-size lists overflow all the time (forced flushes, the dense NMS path,
    output-buffer flushes in mid-row, hit lists rebuilt from crowded rows) -- on textured, noisy and adversarial frames
    the lists must still be the oracle's."""
    import subprocess
    import sys
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from mono_slam_framework_amd import synth\n"
        "from mono_slam_framework_amd.matcher import FeatureMatcher\n"
        "from oracle import orb\n"
        "rng = np.random.default_rng(9)\n"
        "y, x = np.mgrid[0:480, 0:640]\n"
        "imgs = [synth.synth_pair(71, 640, 480, mode=0)[0], synth.synth_pair(72, 640, 480, mode=2, noise=0)[0],\n"
        "        rng.integers(0, 256, size=(480, 640), dtype=np.uint8),\n"
        "        (((x // 2) + (y // 2)) %% 2 * 200 + 20).astype(np.uint8),\n"
        "        ((rng.random((480, 640)) < 0.08).astype(np.uint8) * 220 + 10)]\n"
        "fm = FeatureMatcher(0.8, 640, 480, flags=16 | 64)\n"
        "orc = orb.FeatureMatcherOracle(0.8)\n"
        "for i, a in enumerate(imgs):\n"
        "    b = np.roll(a, (3, 5), (0, 1))\n"
        "    try:\n"
        "        got = fm.MatchFrames(a, b)\n"
        "    except Exception as e:\n"
        "        assert getattr(e, 'code', 0) == -4, e    # a candidate list overflowed: loud, never wrong\n"
        "        continue\n"
        "    assert (fm.fast_tau(0)[:, 1] == 22).all()\n"
        "    assert np.array_equal(got, orc.MatchFrames(a, b)), i\n"
        "    k, d = orb.OrbOracle(640, 480).extract(a)\n"
        "    assert len(fm.keypoints(0)) == len(k) and np.array_equal(fm.descriptors(0), d), i\n"
    ) % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MSF_ORB_FAST_TAU="22"), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_small_calls_take_the_dense_kernel_with_the_same_result():
    """Calls of fewer than 8 frames skip the streaming pass (latency); the lists are the same as with it."""
    from mono_slam_framework_amd import _lib
    from mono_slam_framework_amd.matcher import FeatureMatcher
    a, b = synth.synth_pair(17, 640, 480)
    f_dense = FeatureMatcher(0.7, 640, 480, flags=_lib.MSF_FLAG_NO_FRAME_CACHE)
    f_stream = FeatureMatcher(0.7, 640, 480, flags=_lib.MSF_FLAG_NO_FRAME_CACHE | _lib.MSF_FLAG_FAST_STREAM)
    m1, m2 = f_dense.MatchFrames(a, b), f_stream.MatchFrames(a, b)
    np.testing.assert_array_equal(m1, m2)
    assert (f_dense.fast_tau(0)[:, 0] == 20).all() and (f_stream.fast_tau(0)[:, 0] > 20).any()
    np.testing.assert_array_equal(f_dense.descriptors(0), f_stream.descriptors(0))
    np.testing.assert_array_equal(m1, oracle_orb.FeatureMatcherOracle(0.7).MatchFrames(a, b))


def test_threshold_and_blur_mode():
    a, b = synth.synth_pair(11, 640, 480)
    from mono_slam_framework_amd import _lib
    lists = []
    for flags, tie, s256 in ((0, 1, 0), (_lib.MSF_FLAG_BLUR_TIE_HALF_UP, 0, 0), (_lib.MSF_FLAG_BLUR_SUM256, 1, 1),
                             (_lib.MSF_FLAG_BLUR_SUM256 | _lib.MSF_FLAG_BLUR_TIE_HALF_UP, 0, 1)):
        fm = _matcher(640, 480, thr=0.6, flags=flags)
        orc = oracle_orb.FeatureMatcherOracle(0.6, blur_tie_even=tie, blur_kernel_sum256=s256)
        got = fm.MatchFrames(a, b)
        np.testing.assert_array_equal(got, orc.MatchFrames(a, b))
        np.testing.assert_array_equal(fm.descriptors(0), orc._orb(a.shape)[0].extract(a)[1])
        lists.append(fm.descriptors(0).copy())
        fm.SetThreshold(0.8)
        orc.SetThreshold(0.8)
        np.testing.assert_array_equal(fm.MatchFrames(a, b), orc.MatchFrames(a, b))
    # the sum-256 kernel (OpenCV's bit-exact fixed-point Gaussian) really is another blur: descriptor bits move; it rounds
    # half up whatever the tie flag says
    assert (lists[0] != lists[2]).any() and (lists[2] == lists[3]).all()


def test_strided_input_and_batch():
    w, h = 640, 480
    n = 6
    A, B = synth.synth_batch(100, n, w, h)
    fm = _matcher(w, h, thr=0.6, pairs=4)       # batch larger than max_batch_pairs: chunked
    orc = oracle_orb.FeatureMatcherOracle(0.6)
    exp = [orc.MatchFrames(A[i], B[i]) for i in range(n)]
    got = fm.match_batch(list(A), list(B))
    for g, e in zip(got, exp):
        np.testing.assert_array_equal(g, e)
    # arbitrary row stride (cv::Mat step), as FrameBase::imGray may have
    big = np.zeros((h, w + 37), np.uint8)
    big[:, :w] = A[0]
    big2 = np.zeros((h, w + 5), np.uint8)
    big2[:, :w] = B[0]
    np.testing.assert_array_equal(fm.MatchFrames(big[:, :w], big2[:, :w]), exp[0])


def test_device_resident_batch_matches_host_path():
    import torch
    w, h, n = 1280, 720, 3
    A, B = synth.synth_batch(200, n, w, h)
    fm = _matcher(w, h, thr=0.6, pairs=n)
    dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    out = torch.zeros((n, 1024, 4), dtype=torch.int32, device="cuda")
    cnt = torch.zeros((n,), dtype=torch.int32, device="cuda")
    fm.match_batch_device(dA, dB, out, cnt, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    orc = oracle_orb.FeatureMatcherOracle(0.6)
    for i in range(n):
        e = orc.MatchFrames(A[i], B[i])
        assert int(cnt[i]) == len(e)
        np.testing.assert_array_equal(out[i, :len(e)].cpu().numpy(), e)


def test_extract_once_match_many():
    import torch
    w, h = 640, 480
    A, B = synth.synth_batch(300, 3, w, h)
    frames = np.concatenate([A[:1], B], 0)     # frame 0 against frames 1..3
    fm = _matcher(w, h, thr=0.8, pairs=2)
    d = torch.from_numpy(frames).cuda()
    fm.extract_device(d, first_slot=0)
    sa = torch.tensor([0, 0, 0], dtype=torch.int32, device="cuda")
    sb = torch.tensor([1, 2, 3], dtype=torch.int32, device="cuda")
    out = torch.zeros((3, 1024, 4), dtype=torch.int32, device="cuda")
    cnt = torch.zeros((3,), dtype=torch.int32, device="cuda")
    fm.match_slots_device(sa, sb, out, cnt)
    orc = oracle_orb.FeatureMatcherOracle(0.8)
    for i in range(3):
        e = orc.MatchFrames(frames[0], frames[1 + i])
        assert int(cnt[i]) == len(e)
        np.testing.assert_array_equal(out[i, :len(e)].cpu().numpy(), e)
    # the caller's slots are [0, 2 * max_batch_pairs) = [0, 4): anything else -- the scratch slots of the stateless calls
    # and the frame cache included -- gives n_out = -1 for that pair
    sa = torch.tensor([0, 4, 0, -1], dtype=torch.int32, device="cuda")
    sb = torch.tensor([1, 1, 9, 2], dtype=torch.int32, device="cuda")
    out = torch.zeros((4, 1024, 4), dtype=torch.int32, device="cuda")
    cnt = torch.zeros((4,), dtype=torch.int32, device="cuda")
    fm.match_slots_device(sa, sb, out, cnt)
    assert cnt.tolist()[1:] == [-1, -1, -1] and int(cnt[0]) == len(orc.MatchFrames(frames[0], frames[1]))
    with pytest.raises(Exception):
        fm.extract_device(d[:1], first_slot=4)


def test_errors_are_loud():
    from mono_slam_framework_amd.matcher import MsfError
    fm = _matcher(640, 480)
    with pytest.raises(MsfError):
        fm.MatchFrames(np.zeros((100, 100), np.uint8), np.zeros((100, 100), np.uint8))
    with pytest.raises(MsfError):
        _matcher(16, 16)


def _adversarial_images(w, h):
    y, x = np.mgrid[0:h, 0:w]
    chk = (((x // 2) + (y // 2)) % 2 * 200 + 20).astype(np.uint8)             # 2x2 checkerboard: dense corners
    dots = np.full((h, w), 30, np.uint8)
    dots[::4, ::4] = 250                                                       # isolated bright dots every 4 px
    rng = np.random.default_rng(5)
    salt = (rng.random((h, w)) < 0.08).astype(np.uint8) * 220 + 10             # salt noise
    return [("checker2", chk), ("dots4", dots), ("salt", salt)]


@pytest.mark.parametrize("name,img", _adversarial_images(640, 480), ids=["checker2", "dots4", "salt"])
def test_adversarial_density_is_exact_or_loud(name, img):
    """Fixed-capacity device lists may overflow on pathological inputs; then the call must fail loudly
    (MSF_ERR_CAPACITY), otherwise the result must still be bit-exact."""
    from mono_slam_framework_amd.matcher import MsfError
    img2 = np.roll(img, (3, 5), (0, 1))
    fm = _matcher(640, 480)
    orc = oracle_orb.FeatureMatcherOracle(0.8)
    exp = orc.MatchFrames(img, img2)
    try:
        got = fm.MatchFrames(img, img2)
    except MsfError as e:
        assert e.code == -4
        return
    np.testing.assert_array_equal(got, exp)
    (k1, _), (k2, _) = orc.extract_both(img, img2)
    assert len(fm.keypoints(0)) == len(k1) and len(fm.keypoints(1)) == len(k2)


def test_chunked_match_path():
    """k_match stages train descriptors through LDS in chunks (1024 by default, reached only with heavy ties);
    a child process shrinks the chunk so ~500 keypoints cross several chunks, and must still be bit-exact."""
    import subprocess
    import sys
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from mono_slam_framework_amd import synth\n"
        "from mono_slam_framework_amd.matcher import FeatureMatcher\n"
        "from oracle import orb\n"
        "a, b = synth.synth_pair(21, 640, 480)\n"
        "got = FeatureMatcher(0.8, 640, 480).MatchFrames(a, b)\n"
        "exp = orb.FeatureMatcherOracle(0.8).MatchFrames(a, b)\n"
        "assert len(exp) > 100 and np.array_equal(got, exp)\n"
    ) % ROOT
    env = dict(os.environ, MSF_ORB_TRAIN_CHUNK="96")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_random_sizes_and_textures_parity():
    """Many small frames of arbitrary sizes (odd widths, levels that vanish below 2*31 px, different textures):
    match lists and extracted features stay bit-exact."""
    rng = np.random.default_rng(2026)
    sizes = [(64, 64), (65, 97), (127, 64), (200, 150), (257, 193), (333, 64), (401, 203), (96, 400), (511, 129)]
    for _ in range(6):
        sizes.append((int(rng.integers(64, 520)), int(rng.integers(64, 400))))
    for n, (w, h) in enumerate(sizes):
        kind = n % 4
        if kind == 0:
            a, b = synth.synth_pair(400 + n, w, h, mode=0)
        elif kind == 1:
            a, b = synth.synth_pair(400 + n, w, h, mode=1, noise=3)
        elif kind == 2:
            a = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
            b = np.roll(a, (2, -3), (0, 1))
        else:
            y, x = np.mgrid[0:h, 0:w]
            a = (((x // 7 + y // 5) % 2) * 120 + (x * 3 + y) % 64).astype(np.uint8)
            b = a[::-1].copy()
        fm = _matcher(w, h, thr=0.75)
        orc = oracle_orb.FeatureMatcherOracle(0.75)
        exp = orc.MatchFrames(a, b)
        got = fm.MatchFrames(a, b)
        np.testing.assert_array_equal(got, exp, err_msg="%dx%d kind %d" % (w, h, kind))
        (k1, d1), (k2, d2) = orc.extract_both(a, b)
        for slot, (ko, do) in ((0, (k1, d1)), (1, (k2, d2))):
            kg = fm.keypoints(slot)
            assert len(kg) == len(ko), "%dx%d kind %d slot %d" % (w, h, kind, slot)
            np.testing.assert_array_equal(kg["angle"].view(np.uint32), ko["angle"].view(np.uint32))
            np.testing.assert_array_equal(fm.descriptors(slot), do)
        fm.close()


def test_full_size_batch_properties():
    """At the bench's frame size, on an HBM-resident batch: results are deterministic, independent of a pair's
    position in the batch (pairs never interact), and reproduce the known shift of the synthetic pairs."""
    import torch
    w, h, n = 1280, 720, 48
    A, B = synth.synth_batch(700, n, w, h)
    fm = _matcher(w, h, thr=0.6, pairs=n)
    dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()

    def run(a, b):
        out = torch.zeros((n, 1024, 4), dtype=torch.int32, device="cuda")
        cnt = torch.zeros((n,), dtype=torch.int32, device="cuda")
        fm.match_batch_device(a, b, out, cnt, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return out.cpu().numpy(), cnt.cpu().numpy()

    o1, c1 = run(dA, dB)
    o2, c2 = run(dA, dB)
    np.testing.assert_array_equal(c1, c2)
    np.testing.assert_array_equal(o1, o2)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(3))
    o3, c3 = run(dA[perm].contiguous(), dB[perm].contiguous())
    np.testing.assert_array_equal(c3, c1[perm.numpy()])
    for k, p in enumerate(perm.numpy()):
        np.testing.assert_array_equal(o3[k, :c3[k]], o1[p, :c1[p]])
    assert (c1 > 50).all()
    for i in range(n):
        dx, dy = synth.pair_shift(700 + i)
        d = o1[i, :c1[i], 2:4] - o1[i, :c1[i], 0:2]
        good = (np.abs(d[:, 0] + dx) <= 4) & (np.abs(d[:, 1] + dy) <= 4)   # frame B = canvas shifted by (+dx, +dy)
        assert good.mean() > 0.9, (i, good.mean())
    # two spot checks against the oracle at full size
    orc = oracle_orb.FeatureMatcherOracle(0.6)
    for i in (0, n - 1):
        np.testing.assert_array_equal(o1[i, :c1[i]], orc.MatchFrames(A[i], B[i]))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(640, 480), (333, 251), (1280, 720)])
def test_resize_fallback_equals_shared_pair_kernel(monkeypatch, w, h):
    """k_resize<true> reads the taps of three output pixels from one dword pair (valid when the host's check of the
    column table passes, as for every 1.2x level); MSF_ORB_RESIZE_GENERIC=1 forces the per-pixel fallback.  Both must
    give the oracle's pyramid, on sizes whose level widths are and are not multiples of 4."""
    a, b = synth.synth_pair(77, w, h, mode=0)
    orc = oracle_orb.FeatureMatcherOracle(0.8)
    orc.MatchFrames(a, b)
    oa, _ = orc._orb(a.shape)
    for generic in ("0", "1"):
        monkeypatch.setenv("MSF_ORB_RESIZE_GENERIC", generic)
        fm = _matcher(w, h)
        fm.MatchFrames(a, b)
        for l in range(1, 8):
            np.testing.assert_array_equal(fm.level_pixels(0, l), oa.level_pixels(l), err_msg="generic=%s L%d" % (generic, l))
        fm.close()


@pytest.mark.gpu
def test_refined_fast_threshold_equals_unrefined_and_survives_an_overshooting_estimate(monkeypatch):
    """The streaming walker runs a sampled quarter of the strips at the first threshold; the other strips wait for that
    quarter and raise the threshold from its exact corners (k_walk).  Same match lists as without the refinement
    (MSF_ORB_FAST_ONE_PART=1); the refined threshold really is higher on the large levels; with the margin cut to 3 % of 2N
    it overshoots, levels fail k_fast_check, take the dense second pass -- and the lists are still the same.  Also the
    same: the walker as one launch per level instead of ONE launch over all levels (MSF_ORB_WALK_PER_LEVEL=1: every
    in-launch dependency wait is then met at once), and the unfused form (MSF_ORB_UNFUSED=1: k_resize x 7 + one FAST-only
    walker launch)."""
    from mono_slam_framework_amd import _lib
    from mono_slam_framework_amd.matcher import FeatureMatcher
    n, w, h = 12, 1280, 720
    A, B = synth.synth_batch(9300, n, w, h, mode=0)
    fl = _lib.MSF_FLAG_NO_FRAME_CACHE
    monkeypatch.setenv("MSF_ORB_FAST_ONE_PART", "1")
    one = FeatureMatcher(0.7, w, h, max_batch_pairs=n, flags=fl)
    ref = one.match_batch(list(A), list(B), cap=1024)
    tau_one = np.stack([one.fast_tau(s) for s in range(2 * n)])
    assert sum(len(m) for m in ref) > 50 * n
    monkeypatch.delenv("MSF_ORB_FAST_ONE_PART")
    # refinement alone (every level's first estimate from the sampler, as in the unrefined run): thresholds only rise
    monkeypatch.setenv("MSF_ORB_TAU_PREDICT", "0")
    samp = FeatureMatcher(0.7, w, h, max_batch_pairs=n, flags=fl)
    got = samp.match_batch(list(A), list(B), cap=1024)
    tau_s = np.stack([samp.fast_tau(s) for s in range(2 * n)])
    for r, g in zip(ref, got):
        np.testing.assert_array_equal(r, g)
    assert (tau_s[:, :, 0] >= tau_one[:, :, 0]).all() and (tau_s[:, :3, 0] > tau_one[:, :3, 0]).mean() > 0.5
    assert (tau_s[:, :, 0] == tau_s[:, :, 1]).all()                        # nothing redone at the default margin
    samp.close()
    monkeypatch.delenv("MSF_ORB_TAU_PREDICT")
    # the default: the first estimate of levels >= 1 predicted from the level above, then refined
    two = FeatureMatcher(0.7, w, h, max_batch_pairs=n, flags=fl)
    got = two.match_batch(list(A), list(B), cap=1024)
    tau_two = np.stack([two.fast_tau(s) for s in range(2 * n)])
    for r, g in zip(ref, got):
        np.testing.assert_array_equal(r, g)
    assert (tau_two[:, :3, 0] > tau_one[:, :3, 0]).mean() > 0.5
    assert (tau_two[:, :, 0] == tau_two[:, :, 1]).all()                    # nothing redone
    # an absurd prediction margin (5 % of the needed density: thresholds far too high) only costs dense passes
    # (these two settings send most levels of the batch through the dense pass: the pool of full-capacity candidate
    # lists, sized for an eighth of them, is made large enough for all -- r05, orb_pipeline.h)
    monkeypatch.setenv("MSF_ORB_POOL_ENTRIES", str(1 << 30))
    monkeypatch.setenv("MSF_ORB_TAU_PREDICT", "5")
    wildp = FeatureMatcher(0.7, w, h, max_batch_pairs=n, flags=fl)
    got_p = wildp.match_batch(list(A), list(B), cap=1024)
    assert (np.stack([wildp.fast_tau(s) for s in range(2 * n)])[:, 1:, 0] == 20).sum() > n
    for r, g in zip(ref, got_p):
        np.testing.assert_array_equal(r, g)
    wildp.close()
    monkeypatch.delenv("MSF_ORB_TAU_PREDICT")
    orc = oracle_orb.FeatureMatcherOracle(0.7)
    np.testing.assert_array_equal(got[0], orc.MatchFrames(A[0], B[0]))
    for l in range(1, 8):                                                  # the walker's pyramid is k_resize's
        np.testing.assert_array_equal(two.level_pixels(0, l), one.level_pixels(0, l), err_msg="pyramid L%d" % l)
    monkeypatch.setenv("MSF_ORB_TAU2_MARGIN_PCT", "3")
    wild = FeatureMatcher(0.7, w, h, max_batch_pairs=n, flags=fl)
    got2 = wild.match_batch(list(A), list(B), cap=1024)
    tau_w = np.stack([wild.fast_tau(s) for s in range(2 * n)])
    assert (tau_w[:, :, 0] == 20).sum() > 2 * n                            # many levels fell back to the dense pass
    for r, g in zip(ref, got2):
        np.testing.assert_array_equal(r, g)
    monkeypatch.delenv("MSF_ORB_TAU2_MARGIN_PCT")
    monkeypatch.delenv("MSF_ORB_POOL_ENTRIES")
    for env in ({"MSF_ORB_WALK_PER_LEVEL": "1"}, {"MSF_ORB_UNFUSED": "1"},
                {"MSF_ORB_WALK_PER_LEVEL": "1", "MSF_ORB_TAU_PREDICT": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        alt = FeatureMatcher(0.7, w, h, max_batch_pairs=n, flags=fl)
        got3 = alt.match_batch(list(A), list(B), cap=1024)
        for r, g in zip(ref, got3):
            np.testing.assert_array_equal(r, g, err_msg=str(env))
        for l in range(1, 8):
            np.testing.assert_array_equal(alt.level_pixels(1, l), one.level_pixels(1, l), err_msg="%s pyramid L%d" % (env, l))
        for s_ in (0, 5, 2 * n - 1):                                       # key points with their Harris response bits
            np.testing.assert_array_equal(alt.keypoints(s_), one.keypoints(s_), err_msg=str(env))
            for l in (0, 3, 7):
                a_, b_ = alt.stage1(s_, l), one.stage1(s_, l)
                np.testing.assert_array_equal(np.sort(a_, order=["ly", "lx"]), np.sort(b_, order=["ly", "lx"]), err_msg="%s stage 1 L%d" % (env, l))
        alt.close()
        for k in env:
            monkeypatch.delenv(k)


@pytest.mark.gpu
def test_one_launch_walker_equals_per_level_launches_and_the_unfused_form(monkeypatch):
    """A batch runs pyramid + FAST of all eight levels as ONE launch whose units wait for each other (threshold unit of
    level l for the strips of level l - 1, strips for their threshold, non-quarter strips for the quarter).  Same lists,
    key points and pyramids as one launch per level (MSF_ORB_WALK_PER_LEVEL=1) and as the unfused form, on a frame count
    that is no multiple of 8 (XCDs with different numbers of frames) and on odd level sizes; three frames are checked
    against the oracle; no frame is flagged as stalled."""
    from mono_slam_framework_amd import _lib
    from mono_slam_framework_amd.matcher import FeatureMatcher
    n, w, h = 147, 333, 251                                  # 294 frames
    A, B = synth.synth_batch(9500, n, w, h, mode=0)
    fl = _lib.MSF_FLAG_NO_FRAME_CACHE
    one = FeatureMatcher(0.7, w, h, max_batch_pairs=n, flags=fl)
    ref = one.match_batch(list(A), list(B), cap=1024)
    assert sum(len(m) for m in ref) > 5 * n
    orc = oracle_orb.FeatureMatcherOracle(0.7)
    for i in (0, n // 2 + 3, n - 1):
        np.testing.assert_array_equal(ref[i], orc.MatchFrames(A[i], B[i]))
    kp_ref = [one.keypoints(s) for s in (0, n - 1, n, 2 * n - 1)]
    for env in ("MSF_ORB_WALK_PER_LEVEL", "MSF_ORB_UNFUSED"):
        monkeypatch.setenv(env, "1")
        alt = FeatureMatcher(0.7, w, h, max_batch_pairs=n, flags=fl)
        got = alt.match_batch(list(A), list(B), cap=1024)
        for r, g_ in zip(ref, got):
            np.testing.assert_array_equal(r, g_, err_msg=env)
        for s_, k in zip((0, n - 1, n, 2 * n - 1), kp_ref):
            np.testing.assert_array_equal(alt.keypoints(s_), k, err_msg=env)
            for l in (1, 4, 7):
                np.testing.assert_array_equal(alt.level_pixels(s_, l), one.level_pixels(s_, l), err_msg="%s L%d" % (env, l))
        alt.close()
        monkeypatch.delenv(env)


@pytest.mark.gpu
def test_one_launch_walker_back_to_back_calls_and_small_batches():
    """The one-launch walker's per-(frame, level) state is reset by every call: the same handle run again, on other
    frames, and with batches of 8, 9 and 31 pairs (fewer units than wave slots: every dependency is really waited for)
    gives the oracle's lists each time."""
    from mono_slam_framework_amd import _lib
    from mono_slam_framework_amd.matcher import FeatureMatcher
    w, h = 640, 480
    fm = FeatureMatcher(0.7, w, h, max_batch_pairs=31, flags=_lib.MSF_FLAG_NO_FRAME_CACHE | _lib.MSF_FLAG_PROFILE)
    orc = oracle_orb.FeatureMatcherOracle(0.7)
    for rep, n in enumerate((31, 8, 9, 31)):
        A, B = synth.synth_batch(9700 + 50 * rep, n, w, h, mode=rep % 3)
        got = fm.match_batch(list(A), list(B), cap=1024)
        for i in (0, n // 2, n - 1):
            np.testing.assert_array_equal(got[i], orc.MatchFrames(A[i], B[i]), err_msg="call %d pair %d" % (rep, i))
    st = fm.stage_times()                            # sums over the four calls; the first stage is the one walker launch
    assert set(st) == {"pyramid_fast", "fast_nms", "select_harris", "orient_describe", "match"} and all(v > 0 for v in st.values())
    fm.close()


@pytest.mark.gpu
def test_a_unit_that_never_publishes_ends_in_flagged_frames_not_in_a_hang(monkeypatch):
    """The one-launch walker's waits are bounded: with the test hook that withholds the threshold of (frame 5, level 3),
    the strips of that (frame, level) give up after ~1 s, flag the frame, raise the abort word, every unit still waiting
    leaves, and the call returns: the pair of frame 5 reports n_out = -1 (MSF_ERR_CAPACITY from the batch call, never a
    silent wrong list), every other pair is the oracle's; the handle then falls back to one walker launch per level (the
    one-launch form rests on workgroups starting in index order, which is observed, not promised) and works normally."""
    import time
    from mono_slam_framework_amd import _lib
    from mono_slam_framework_amd.matcher import FeatureMatcher, MsfError
    n, w, h = 16, 640, 480
    A, B = synth.synth_batch(9900, n, w, h, mode=0)
    monkeypatch.setenv("MSF_TEST_HOOKS", "1")
    monkeypatch.setenv("MSF_ORB_TEST_STALL_FRAME", "5")
    fm = FeatureMatcher(0.7, w, h, max_batch_pairs=n, flags=_lib.MSF_FLAG_NO_FRAME_CACHE)
    t0 = time.time()
    num, lists = fm.match_batch_raw(list(A), list(B), cap=1024)
    assert time.time() - t0 < 30.0
    assert num[5] == -1 and (np.delete(num, 5) >= 0).all(), num
    orc = oracle_orb.FeatureMatcherOracle(0.7)
    for i in (0, 4, 6, n - 1):
        np.testing.assert_array_equal(lists[i], orc.MatchFrames(A[i], B[i]))
    # the hook stalls the first call only.  The handle has seen the abort word by now (match_batch waits for its call), so
    # the next call runs the walker one level at a time -- and says so once -- with every pair correct again
    got = fm.match_batch(list(A), list(B), cap=1024)
    assert "one level at a time" in fm.last_error()
    for i in (0, 5, n - 1):
        np.testing.assert_array_equal(got[i], orc.MatchFrames(A[i], B[i]))
    got = fm.match_batch(list(A), list(B), cap=1024)
    np.testing.assert_array_equal(got[5], orc.MatchFrames(A[5], B[5]))
    fm.close()


@pytest.mark.gpu
def test_two_walker_grids_at_once():
    """Two k_walk grids on one device at the same time: two handles with P >= 48 (240-row strips, the bench's form), each
    driven by its own host thread on its own stream for several calls, then one msf_multi with two shards on device 0.
    A unit only ever waits for units of lower index in ITS grid, so a second grid takes wave slots but cannot stall the
    first: every list is what the same handle gives alone (three per call also against the oracle), no frame is flagged
    (n_out >= 0 everywhere), no unit gave up a wait and no handle fell back to per-level launches (walk_mode,
    msf_last_error empty).  Reference call pattern: src/main.cpp:131-139 (the caller's thread changes every frame)."""
    import threading
    from mono_slam_framework_amd import _lib
    from mono_slam_framework_amd.matcher import FeatureMatcher, MultiDeviceMatcher
    n, w, h, calls = 64, 640, 480, 4
    fl = _lib.MSF_FLAG_NO_FRAME_CACHE
    data = [synth.synth_batch(12000 + 1000 * t, n, w, h, mode=t % 3) for t in range(2)]
    fms = [FeatureMatcher(0.7, w, h, max_batch_pairs=n, flags=fl) for _ in range(2)]
    alone = []
    for fm, (A, B) in zip(fms, data):                 # each handle on its own first: the reference lists
        num, lists = fm.match_batch_raw(list(A), list(B), cap=1024)
        assert (num >= 0).all()
        alone.append(lists)
    orc = oracle_orb.FeatureMatcherOracle(0.7)
    for (A, B), lists in zip(data, alone):
        for i in (0, 31, n - 1):
            np.testing.assert_array_equal(lists[i], orc.MatchFrames(A[i], B[i]))
    start = threading.Barrier(2)
    results, errors = [None, None], []

    def drive(t):
        try:
            A, B = data[t]
            out = []
            start.wait()
            for _ in range(calls):
                out.append(fms[t].match_batch_raw(list(A), list(B), cap=1024))
            results[t] = out
        except Exception as e:                           # noqa: BLE001 -- reported by the main thread
            errors.append((t, e))

    th = [threading.Thread(target=drive, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
    for t in range(2):
        for num, lists in results[t]:
            assert (num >= 0).all(), "handle %d: flagged frames %s" % (t, np.nonzero(num < 0)[0])
            for i in range(n):
                np.testing.assert_array_equal(lists[i], alone[t][i], err_msg="handle %d pair %d" % (t, i))
        assert fms[t].walk_mode() == (False, 0)
        assert fms[t].last_error() == ""
    for fm in fms:
        fm.close()
    # one msf_multi, two shards on device 0: 2 x 64 pairs, each shard's k_walk grid beside the other's
    multi = MultiDeviceMatcher("orb", 0.7, w, h, devices=(0, 0), max_batch_pairs=n, flags=fl)
    A = list(data[0][0]) + list(data[1][0])
    B = list(data[0][1]) + list(data[1][1])
    for _ in range(2):
        got = multi.match_batch(A, B, cap=1024)
        for i in range(n):
            np.testing.assert_array_equal(got[i], alone[0][i], err_msg="shard 0 pair %d" % i)
            np.testing.assert_array_equal(got[n + i], alone[1][i], err_msg="shard 1 pair %d" % i)
    multi.close()


@pytest.mark.gpu
def test_candidate_pool_exhaustion_is_loud_and_the_footprint_is_what_the_header_says(monkeypatch):
    """r05: a (frame, level) keeps a small primary candidate list (w h / 64 entries) and takes a full-capacity region of a
    shared pool only when it goes through the dense second pass (orb_pipeline.h).  (1) With the pool cut to nothing
    (MSF_ORB_POOL_ENTRIES=0) and a first threshold no level can satisfy (MSF_ORB_FAST_TAU=254: every level needs the dense
    pass) the pairs that find no region report n_out = -1 / MSF_ERR_CAPACITY -- never a short list; with the default pool the same batch is
    the oracle's (test_fast_threshold_fallback_path).  (2) A handle for 1024 pairs of 1280 x 720 takes less than the 8 GB
    include/msf_abi.h states (r04: 10.4 GB, r03: 20 GB)."""
    import torch
    from mono_slam_framework_amd import _lib
    from mono_slam_framework_amd.matcher import FeatureMatcher
    n, w, h = 8, 640, 480
    A, B = synth.synth_batch(12500, n, w, h, mode=0)
    monkeypatch.setenv("MSF_ORB_FAST_TAU", "254")
    monkeypatch.setenv("MSF_ORB_POOL_ENTRIES", "0")
    fm = FeatureMatcher(0.7, w, h, max_batch_pairs=n, flags=_lib.MSF_FLAG_NO_FRAME_CACHE)
    num, lists = fm.match_batch_raw(list(A), list(B), cap=1024)
    # (the pool keeps the regions dense CALLS need -- 7 frames' worth here -- and the overflowing levels race for them:
    # which frames lose is not determined, that most do is; a pair that got its regions must be the oracle's)
    assert (num == -1).sum() >= n // 2, num
    for i in np.nonzero(num >= 0)[0]:
        np.testing.assert_array_equal(lists[i], oracle_orb.FeatureMatcherOracle(0.7).MatchFrames(A[i], B[i]))
    fm.close()
    monkeypatch.delenv("MSF_ORB_POOL_ENTRIES")
    fm = FeatureMatcher(0.7, w, h, max_batch_pairs=n, flags=_lib.MSF_FLAG_NO_FRAME_CACHE)
    num, lists = fm.match_batch_raw(list(A), list(B), cap=1024)
    assert (num >= 0).all()
    np.testing.assert_array_equal(lists[3], oracle_orb.FeatureMatcherOracle(0.7).MatchFrames(A[3], B[3]))
    fm.close()
    monkeypatch.delenv("MSF_ORB_FAST_TAU")
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    big = FeatureMatcher(0.6, 1280, 720, max_batch_pairs=1024)
    free1, _ = torch.cuda.mem_get_info()
    used = (free0 - free1) / 1e9
    print("ORB handle, 1024 pairs of 1280x720: %.2f GB of device memory" % used)
    assert 5.0 < used < 8.0, used
    big.close()
