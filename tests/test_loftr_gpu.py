"""GPU parity tests for the LoFTR path through the C ABI: confidences within 1e-3 of the reference graph
(golden fixtures produced from model/LoFTR_teacher.onnx) and of the CPU restatement; match lists identical
wherever |conf - thr| > 1e-3.  Reference path: src/dnnfeaturematcher.cpp:44-102."""
import os

import numpy as np
import pytest

from mono_slam_framework_amd import synth
from oracle import loftr as oracle_loftr

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "loftr_kat.npz"))
CONF_TOL = 1e-3   # north_star: "LoFTR match confidences within 1e-3"


def _dm(thr=0.15, pairs=1):
    from mono_slam_framework_amd.matcher import DNNFeatureMatcher
    from mono_slam_framework_amd import _lib
    return DNNFeatureMatcher(threshold=thr, max_batch_pairs=pairs,
                             flags=_lib.MSF_FLAG_KEEP_DEBUG | _lib.MSF_FLAG_NO_FRAME_CACHE)


def _check_lists(got, conf_ref, thr):
    """identical match lists wherever |conf - thr| > tol: got must contain every sure hit and no sure miss"""
    sure = oracle_loftr.DNNFeatureMatcherOracle(thr).decode(conf_ref, thr + CONF_TOL)
    maybe = oracle_loftr.DNNFeatureMatcherOracle(thr).decode(conf_ref, thr - CONF_TOL)
    gs = set(map(tuple, got))
    assert set(map(tuple, sure)) <= gs <= set(map(tuple, maybe))
    # order: row-major over (i, j)  (cv::findNonZero)
    key = [(y1 // 16) * 40 + x1 // 16 for x1, y1, _, _ in got]
    key2 = [(y2 // 16) * 40 + x2 // 16 for _, _, x2, y2 in got]
    assert all((a, b) < (c, d) for (a, b), (c, d) in zip(zip(key, key2), list(zip(key, key2))[1:]))


@pytest.mark.parametrize("name", ["i", "ii", "iii", "synth"])
def test_conf_and_matches_vs_golden(name):
    a, b = G["img0_" + name], G["img1_" + name]
    dm = _dm(0.15)
    got = dm.MatchFrames(a, b, cap=8192)
    conf = dm.conf_matrix()
    feat = dm.coarse_features()
    assert np.abs(feat[0] - G["feat0_" + name]).max() < 1e-3
    assert np.abs(feat[1] - G["feat1_" + name]).max() < 1e-3
    bi, bv = G["big_ij_" + name].astype(int), G["big_v_" + name]
    si, sv = G["samp_ij_" + name].astype(int), G["samp_v_" + name]
    if len(bv):
        assert np.abs(conf[bi[:, 0], bi[:, 1]] - bv).max() <= CONF_TOL
    assert np.abs(conf[si[:, 0], si[:, 1]] - sv).max() <= CONF_TOL
    assert np.abs(conf.sum(1) - G["rowsum_" + name]).max() < 5e-3
    # golden match lists: exact unless an entry sits within tol of the threshold
    exp = G["matches_%s_015" % name]
    full_ref = oracle_loftr.DNNFeatureMatcherOracle(0.15).run(a, b)["conf"]
    assert np.abs(conf - full_ref).max() <= CONF_TOL
    _check_lists(got, full_ref, 0.15)
    near = np.abs(full_ref - 0.15) <= CONF_TOL
    if not near.any():
        np.testing.assert_array_equal(got, exp)
    dm.SetThreshold(0.1)
    got10 = dm.MatchFrames(a, b, cap=8192)
    _check_lists(got10, full_ref, 0.1)
    if not (np.abs(full_ref - 0.1) <= CONF_TOL).any():
        np.testing.assert_array_equal(got10, G["matches_%s_010" % name])


def test_batch_equals_single_and_chunks():
    n = 5
    A, B = synth.synth_batch(40, n, 640, 480, mode=1)
    single = _dm(0.15)
    ref = [single.MatchFrames(A[i], B[i], cap=8192) for i in range(n)]
    batched = _dm(0.15, pairs=2).match_batch(list(A), list(B), cap=8192)   # chunks of 2
    for r, g in zip(ref, batched):
        np.testing.assert_array_equal(r, g)


def test_streaming_batch_matches_the_oracle_per_pair():
    """A batch of 40 pairs = 80 images takes the streaming layer-1 / layer-2 kernels (backbone passes of >= 64 images;
    single pairs take the banded ones).  Spot-checked pairs must give the oracle's list wherever no confidence is
    within the tolerance of the threshold, and the batch must agree with single calls under the same rule."""
    n = 40
    A, B = synth.synth_batch(4100, n, 640, 480, mode=1)
    batched = _dm(0.15, pairs=n).match_batch(list(A), list(B), cap=8192)
    single = _dm(0.15)
    for k in (0, 17, 39):
        full_ref = oracle_loftr.DNNFeatureMatcherOracle(0.15).run(A[k], B[k])["conf"]
        assert len(batched[k]) > 10
        _check_lists(batched[k], full_ref, 0.15)
        _check_lists(single.MatchFrames(A[k], B[k], cap=8192), full_ref, 0.15)


def test_low_threshold_capacity_and_count():
    a, b = G["img0_ii"], G["img1_ii"]
    dm = _dm(1e-4)
    from mono_slam_framework_amd.matcher import MsfError
    full_ref = oracle_loftr.DNNFeatureMatcherOracle(1e-4).run(a, b)["conf"]
    n_ref = int((full_ref > 1e-4).sum())
    assert n_ref > 4096
    with pytest.raises(MsfError):         # the host staging list (4096) is shorter than the result: loud, not silent
        dm.MatchFrames(a, b, cap=100000)
    got = dm.MatchFrames(a, b, cap=64)    # caller-side truncation is part of the contract
    assert len(got) == 64


def test_wrong_size_is_unsupported():
    from mono_slam_framework_amd.matcher import DNNFeatureMatcher, MsfError
    with pytest.raises(MsfError) as e:
        DNNFeatureMatcher(threshold=0.15, image_width=1280, image_height=720)
    assert e.value.code == -3


def test_device_batch_across_backbone_chunks():
    """An HBM-resident batch larger than one backbone chunk (64 pairs): every pair's list equals the single-pair
    result regardless of its position, and KAT (ii) placed anywhere in the batch gives its golden list."""
    import torch
    n = 70
    A, B = synth.synth_batch(900, n, 640, 480, mode=1)
    A[66], B[66] = G["img0_ii"], G["img1_ii"]
    dm = _dm(0.15, pairs=n)
    dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    out = torch.zeros((n, 2048, 4), dtype=torch.int32, device="cuda")
    cnt = torch.zeros((n,), dtype=torch.int32, device="cuda")
    dm.match_batch_device(dA, dB, out, cnt, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out, cnt = out.cpu().numpy(), cnt.cpu().numpy()
    single = _dm(0.15)
    for i in (0, 31, 63, 64, 66, 69):
        ref = single.MatchFrames(A[i], B[i], cap=8192)
        assert cnt[i] == len(ref)
        np.testing.assert_array_equal(out[i, :cnt[i]], ref)
    np.testing.assert_array_equal(out[66, :cnt[66]], G["matches_ii_015"])


def test_token_cache_extract_once_match_many():
    """SURVEY.md 8f row 1 for LoFTR: backbone tokens cached per frame slot (msf_extract_device) + pairs of slots
    (msf_match_slots_device) give exactly the lists of the two-frame path, for any pairing (a frame against itself,
    a slot used on both sides, slots beyond one backbone chunk), and KAT (ii) through the cache is its golden list."""
    import torch
    n_frames = 140                                     # > one backbone chunk (128 images)
    A, B = synth.synth_batch(950, n_frames // 2, 640, 480, mode=1)
    F = np.concatenate([A, B])
    F[3], F[137] = G["img0_ii"], G["img1_ii"]
    dm = _dm(0.15, pairs=80)
    dF = torch.from_numpy(F).cuda()
    dm.extract_device(dF[:100], first_slot=0)
    dm.extract_device(dF[100:], first_slot=100)        # second call, other slots
    rng = np.random.RandomState(4)
    sa = rng.randint(0, n_frames, 40).astype(np.int32)
    sb = rng.randint(0, n_frames, 40).astype(np.int32)
    sa[:3], sb[:3] = [3, 5, 139], [137, 5, 0]          # KAT pair, a frame against itself, last slot
    out = torch.zeros((40, 2048, 4), dtype=torch.int32, device="cuda")
    cnt = torch.zeros((40,), dtype=torch.int32, device="cuda")
    dm.match_slots_device(torch.from_numpy(sa).cuda(), torch.from_numpy(sb).cuda(), out, cnt)
    out2 = torch.zeros_like(out)
    cnt2 = torch.zeros_like(cnt)
    dm.match_batch_device(dF[torch.from_numpy(sa).long().cuda()].contiguous(),
                          dF[torch.from_numpy(sb).long().cuda()].contiguous(), out2, cnt2)
    out, cnt, out2, cnt2 = out.cpu().numpy(), cnt.cpu().numpy(), out2.cpu().numpy(), cnt2.cpu().numpy()
    np.testing.assert_array_equal(cnt, cnt2)
    for i in range(40):
        np.testing.assert_array_equal(out[i, :cnt[i]], out2[i, :cnt2[i]])
    np.testing.assert_array_equal(out[0, :cnt[0]], G["matches_ii_015"])
    assert cnt[1] > 100                                # a frame matches itself on (nearly) every cell
    with pytest.raises(Exception):
        dm.extract_device(dF[:2], first_slot=159)      # slots are [0, 2*max_batch_pairs)
    with pytest.raises(Exception):
        dm.extract_device(dF[:1], first_slot=160)      # the first slot of the handle's private frame cache
    # a slot index outside [0, 2 * max_batch_pairs) gives n_out = -1 for that pair (the handle's own cache slots
    # included: they exist in the pipeline, but not for the caller); the other pairs are unaffected
    sa2 = np.array([3, 160, 3, -1], np.int32)
    sb2 = np.array([137, 5, 223, 5], np.int32)
    out3 = torch.zeros((4, 2048, 4), dtype=torch.int32, device="cuda")
    cnt3 = torch.zeros((4,), dtype=torch.int32, device="cuda")
    dm.match_slots_device(torch.from_numpy(sa2).cuda(), torch.from_numpy(sb2).cuda(), out3, cnt3)
    cnt3 = cnt3.cpu().numpy()
    assert cnt3[1] == -1 and cnt3[2] == -1 and cnt3[3] == -1, cnt3
    np.testing.assert_array_equal(out3[0, :cnt3[0]].cpu().numpy(), G["matches_ii_015"])


def test_sparse_head_equals_dense_head(monkeypatch):
    """The head lists, per row of S, the entries that can reach the threshold (conf <= softmax_j) and evaluates only
    those; below threshold 0.05 (and with MSF_LOFTR_DENSE_HEAD=1) every entry is evaluated.  Both must give the same
    match lists: textured pairs, the KATs, a frame against itself (a match on nearly every cell)."""
    A, B = synth.synth_batch(1200, 6, 640, 480, mode=1)
    pairs = [(A[i], B[i]) for i in range(6)] + [(G["img0_ii"], G["img1_ii"]), (A[0], A[0]), (B[3], B[3])]
    sparse = _dm(0.15, pairs=4)
    monkeypatch.setenv("MSF_LOFTR_DENSE_HEAD", "1")
    dense = _dm(0.15, pairs=4)
    monkeypatch.delenv("MSF_LOFTR_DENSE_HEAD")
    total = 0
    for thr in (0.05, 0.0501, 0.1, 0.15, 0.3, 0.9):
        sparse.SetThreshold(thr)
        dense.SetThreshold(thr)
        for i in range(0, len(pairs), 4):
            fa = [p[0] for p in pairs[i:i + 4]]
            fb = [p[1] for p in pairs[i:i + 4]]
            ls = sparse.match_batch(fa, fb, cap=8192)
            ld = dense.match_batch(fa, fb, cap=8192)
            for a, b in zip(ls, ld):
                np.testing.assert_array_equal(a, b)
                total += len(a)
    assert total > 5000
    sparse.SetThreshold(0.02)       # below the sparse path's validity: the handle switches to the dense head itself
    dense.SetThreshold(0.02)
    np.testing.assert_array_equal(sparse.MatchFrames(*pairs[6], cap=8192), dense.MatchFrames(*pairs[6], cap=8192))


def test_model_file_may_be_an_onnx_file(tmp_path):
    """DNNFeatureMatcher(model_file_path, ...) opens the model file its caller names
    (/root/reference/src/dnnfeaturematcher.cpp:11-21).  The reference's file cannot travel to the GPU box, so an ONNX
    file with the same topology is written from the blob's tensors (tests/onnx_writer.py): constructed from it, the
    matcher gives the confidences and the match list it gives from the blob, bit for bit."""
    from mono_slam_framework_amd import _lib
    from mono_slam_framework_amd.matcher import DNNFeatureMatcher, MsfError
    from tests import onnx_writer
    p = str(tmp_path / "LoFTR_teacher.onnx")
    onnx_writer.write_onnx(p, onnx_writer.read_blob(_lib.default_weights_path()))
    a, b = G["img0_ii"], G["img1_ii"]
    ref = _dm(0.15)
    exp = ref.MatchFrames(a, b, cap=8192)
    dm = DNNFeatureMatcher(p, threshold=0.15, flags=_lib.MSF_FLAG_KEEP_DEBUG | _lib.MSF_FLAG_NO_FRAME_CACHE)
    np.testing.assert_array_equal(dm.MatchFrames(a, b, cap=8192), exp)
    np.testing.assert_array_equal(dm.conf_matrix().view(np.uint32), ref.conf_matrix().view(np.uint32))
    np.testing.assert_array_equal(exp, G["matches_ii_015"])
    with pytest.raises(MsfError) as e:
        DNNFeatureMatcher(str(tmp_path / "absent.onnx"), threshold=0.15)
    assert e.value.code == _lib.MSF_ERR_IO


def _run_child(env_extra, seed=41):
    """one pair through a fresh handle in a child process (the kernel switches are read at msf_create)"""
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from mono_slam_framework_amd import synth\n"
        "from mono_slam_framework_amd.matcher import DNNFeatureMatcher\n"
        "a, b = synth.synth_pair(%d, 640, 480, mode=1, shift=(32, 16))\n"
        "dm = DNNFeatureMatcher(threshold=0.15, flags=4 | 16)\n"
        "m = dm.MatchFrames(a, b, cap=8192)\n"
        "np.savez(sys.argv[1], m=m, conf=dm.conf_matrix(), feat=dm.coarse_features(),"
        " **{'act%%d' %% l: dm.backbone_activation(l) for l in range(4)})\n"
    ) % (root, seed)
    f = tempfile.NamedTemporaryFile(suffix=".npz", delete=False).name
    r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **env_extra), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    out = dict(np.load(f))
    os.unlink(f)
    return out


def test_fused_basic_blocks_equal_the_layer_by_layer_path():
    """f32 kernels (MSF_LOFTR_F32=1): the fused BasicBlocks (k_block8 at 240x320, k_block16 at 120x160: the intermediate
    activation stays in LDS) run the same f32 MFMA chains in the same k order as one kernel per convolution
    (MSF_LOFTR_UNFUSED=1): confidences and features are bit-identical."""
    outs = [_run_child({"MSF_LOFTR_F32": "1", "MSF_LOFTR_UNFUSED": u}) for u in ("0", "1")]
    assert len(outs[0]["m"]) > 20
    np.testing.assert_array_equal(outs[0]["m"], outs[1]["m"])
    np.testing.assert_array_equal(outs[0]["feat"].view(np.uint32), outs[1]["feat"].view(np.uint32))
    np.testing.assert_array_equal(outs[0]["conf"].view(np.uint32), outs[1]["conf"].view(np.uint32))


_F32_RUNS = {}


def _f32_run(seed):
    if seed not in _F32_RUNS:
        _F32_RUNS[seed] = _run_child({"MSF_LOFTR_F32": "1"}, seed)
    return _F32_RUNS[seed]


@pytest.mark.parametrize("strip,seeds", [("3", (41, 7)), ("0", (41,))])
def test_split_bf16_blocks_stay_within_a_tenth_of_the_tolerance(strip, seeds):
    """Default path: the ResNet blocks run on bf16 MFMAs with every f32 operand split into hi + lo (three products,
    f32 accumulation); the ResNet runs as streaming strip kernels (default) or, MSF_LOFTR_STRIP=0, as the banded kernels calls
    of fewer than 64 images always take (MSF_LOFTR_STRIP_MIN=1 lifts that limit for the test).  Against the all-f32 path (MSF_LOFTR_F32=1)
    confidences must agree to 1e-4 -- a tenth of the north-star tolerance -- and the match lists wherever the f32
    confidence is not within 1e-4 of the threshold."""
    for seed in seeds:
        x, f = _run_child({"MSF_LOFTR_STRIP": strip, "MSF_LOFTR_STRIP_MIN": "1"}, seed), _f32_run(seed)
        dconf = np.abs(x["conf"] - f["conf"]).max()
        dfeat = np.abs(x["feat"] - f["feat"]).max()
        print("split-bf16 (strip mode %s) vs f32: max |dconf| %.3g, max |dfeat| %.3g" % (strip, dconf, dfeat))
        assert dconf <= 1e-4 and dfeat <= 1e-3
        # every ResNet stage's activation (MSF_DBG_LOFTR_ACT): a split product drops terms below 2^-16 of |x w|
        for l in range(4):
            d = np.abs(x["act%d" % l] - f["act%d" % l])
            assert np.abs(f["act%d" % l]).max() > 1.0
            assert d.max() <= 5e-4 and np.median(d) <= 2e-5, (l, d.max(), np.median(d))
        assert len(f["m"]) > 20
        sure = np.argwhere(f["conf"] > 0.15 + 1e-4)
        maybe = np.argwhere(f["conf"] > 0.15 - 1e-4)
        cell = lambda m: set(((y1 // 16) * 40 + x1 // 16, (y2 // 16) * 40 + x2 // 16) for x1, y1, x2, y2 in m)
        assert set(map(tuple, sure)) <= cell(x["m"]) <= set(map(tuple, maybe))


def test_single_pass_statistics_equal_the_running_maximum_passes(monkeypatch):
    """Batches of >= 8 pairs take both soft-max statistics from ONE evaluation of S with a per-pair exponent offset
    (k_sim_single / k_sim_finish / k_sim_cand3); MSF_LOFTR_SIM_SINGLE=0 keeps the two running-maximum passes.  The
    confidences of the debug pair must agree to 1e-5 (a hundredth of the tolerance), the lists wherever no confidence
    is that close to the threshold -- textured pairs, the KATs, a frame against itself -- on the sparse and the dense
    head; and a batch whose every pair is flagged for the fallback (MSF_LOFTR_SIM_FORCE_REDO=1: what a pair with
    logits outside the offset's range takes) must give the running-maximum lists exactly."""
    n = 12
    A, B = synth.synth_batch(7300, n, 640, 480, mode=1)
    A[3], B[3] = G["img0_ii"], G["img1_ii"]
    A[5], B[5] = G["img0_i"], G["img1_i"]
    B[7] = A[7]
    fa, fb = list(A), list(B)
    one = _dm(0.15, pairs=n)
    monkeypatch.setenv("MSF_LOFTR_SIM_SINGLE", "0")
    two = _dm(0.15, pairs=n)
    monkeypatch.delenv("MSF_LOFTR_SIM_SINGLE")
    monkeypatch.setenv("MSF_LOFTR_SIM_FORCE_REDO", "1")
    redo = _dm(0.15, pairs=n)
    monkeypatch.delenv("MSF_LOFTR_SIM_FORCE_REDO")
    total = 0
    for thr in (0.15, 0.05, 0.3, 0.02):
        for m in (one, two, redo):
            m.SetThreshold(thr)
        l1 = one.match_batch(fa, fb, cap=8192)
        c1 = one.conf_matrix()
        l2 = two.match_batch(fa, fb, cap=8192)
        c2 = two.conf_matrix()
        l3 = redo.match_batch(fa, fb, cap=8192)
        assert np.abs(c1 - c2).max() <= 1e-5
        for k in range(n):
            np.testing.assert_array_equal(l2[k], l3[k])
            total += len(l2[k])
        # pair 0: the lists may differ only at confidences within 1e-5 of the threshold
        s1, s2 = set(map(tuple, l1[0])), set(map(tuple, l2[0]))
        near = int((np.abs(c2 - thr) <= 1e-5).sum())
        assert len(s1 ^ s2) <= near
        differing = sum(1 for k in range(n) if len(l1[k]) != len(l2[k]) or not np.array_equal(l1[k], l2[k]))
        assert differing <= 1, differing
    assert total > 3000


def test_fused_output_convolution_and_tokens_equal_the_two_kernel_tail():
    """r05: the backbone's 1 x 1 output convolution, the positional encoding and the n c h w -> n (h w) c transpose run as
    one kernel (k_out_tokens) instead of k_conv<32, 32, 1, 1> + k_tokens: the same f32 MFMA chain in the same k order with
    the operands swapped, the same two adds behind it.  Confidences, features and the list are bit-identical to the
    two-kernel tail (MSF_LOFTR_OUT_FUSED=0) -- single pair (banded backbone) and the exact-f32 path alike."""
    for extra in ({}, {"MSF_LOFTR_F32": "1"}):
        outs = [_run_child(dict(extra, MSF_LOFTR_OUT_FUSED=u)) for u in ("1", "0")]
        assert len(outs[0]["m"]) > 20
        np.testing.assert_array_equal(outs[0]["m"], outs[1]["m"])
        np.testing.assert_array_equal(outs[0]["feat"].view(np.uint32), outs[1]["feat"].view(np.uint32))
        np.testing.assert_array_equal(outs[0]["conf"].view(np.uint32), outs[1]["conf"].view(np.uint32))


def test_candidate_pass_skips_tiles_without_changing_a_list(monkeypatch):
    """r05: the statistics pass (k_sim_single) leaves the largest dot product of every (row-tile triple, column tile) and the
    candidate pass (k_sim_cand3) evaluates only the tiles whose maximum can reach the smallest candidate limit of their rows.
    A skipped tile holds no candidate, so the lists are those of the pass over every tile (MSF_LOFTR_SIM_SKIP=0), pair by
    pair, at the default threshold and at the lowest one the sparse head takes; spot-checked against the oracle."""
    from mono_slam_framework_amd import _lib
    from mono_slam_framework_amd.matcher import DNNFeatureMatcher
    n = 24                                           # >= 8 pairs: the batch form of the head (three row tiles per wave)
    A, B = synth.synth_batch(8800, n, 640, 480, mode=1)
    A[5], B[5] = G["img0_ii"], G["img1_ii"]
    fl = _lib.MSF_FLAG_NO_FRAME_CACHE
    monkeypatch.setenv("MSF_LOFTR_SIM_SKIP", "0")
    every = DNNFeatureMatcher(threshold=0.15, max_batch_pairs=n, flags=fl)
    monkeypatch.delenv("MSF_LOFTR_SIM_SKIP")
    skip = DNNFeatureMatcher(threshold=0.15, max_batch_pairs=n, flags=fl)
    total = 0
    for thr in (0.15, 0.06):
        every.SetThreshold(thr)
        skip.SetThreshold(thr)
        l1 = every.match_batch(list(A), list(B), cap=8192)
        l2 = skip.match_batch(list(A), list(B), cap=8192)
        for i in range(n):
            np.testing.assert_array_equal(l1[i], l2[i], err_msg="thr %g pair %d" % (thr, i))
            total += len(l2[i])
    assert total > 2000
    skip.SetThreshold(0.15)
    got = skip.match_batch(list(A), list(B), cap=8192)
    full_ref = oracle_loftr.DNNFeatureMatcherOracle(0.15).run(A[5], B[5])["conf"]
    _check_lists(got[5], full_ref, 0.15)
    every.close()
    skip.close()


def test_paired_attention_launches_equal_one_launch_per_block():
    """r05: the two self-attention blocks of a layer pair (feat0 <- feat0, feat1 <- feat1) run as ONE launch of each
    attention kernel (12 launches per call instead of 16).  Same arithmetic, other grid: confidences, features and the
    match list are bit-identical to one launch per block (MSF_LOFTR_ATTN_PAIR=0)."""
    outs = [_run_child({"MSF_LOFTR_ATTN_PAIR": u}) for u in ("1", "0")]
    assert len(outs[0]["m"]) > 20
    np.testing.assert_array_equal(outs[0]["m"], outs[1]["m"])
    np.testing.assert_array_equal(outs[0]["feat"].view(np.uint32), outs[1]["feat"].view(np.uint32))
    np.testing.assert_array_equal(outs[0]["conf"].view(np.uint32), outs[1]["conf"].view(np.uint32))
