"""CPU: the host logic of the KeyFrameMatchDatabase mirror (mono_slam_framework_amd/keyframe_db.py) against the
line-by-line restatement of slam_pipeline/src/KeyFrameDatabase.cc:23-117 (oracle/keyframe_db.py), fuzzed over random
covisibility graphs and match counts through a stub matcher -- ties, zero counts, connected keyframes, mnLoopQuery
exclusions, keyframes erased and re-added.  (The device side of the same class is tests/test_keyframe_db_gpu.py.)
Also the known answers of the Initializer restatement (oracle/initializer_oracle.c)."""
import numpy as np

from oracle import initializer as oracle_init
from oracle import keyframe_db as oracle_db

W, H = 64, 48


class StubMatcher:
    """Stands in for libmsf.so: frames are tiny images whose pixel [0, 0..1] carries a frame number; the match list of
    (query, keyframe) is a deterministic function of the two numbers."""
    max_batch_pairs = 64
    width, height = W, H

    def __init__(self, seed):
        self.seed = seed
        self.frames = {}
        self.maps = {}

    @staticmethod
    def number(img):
        return int(img[0, 0]) * 256 + int(img[0, 1])

    def lists(self, qa, kb):
        rng = np.random.RandomState((self.seed * 1000003 + qa * 4099 + kb) % (2 ** 31))
        n = int(rng.choice([0, 0, 1, 3, 10, 10, 25, 25, 40]))           # ties and empties on purpose
        return np.stack([rng.randint(0, W, n), rng.randint(0, H, n), rng.randint(0, W, n), rng.randint(0, H, n)], 1)

    # the part of the matcher API the database uses
    def store_frame(self, slot, frame):
        self.frames[slot] = self.number(frame)

    def set_mappoints(self, slot, keys):
        self.maps[slot] = set(int(k) for k in keys)

    def match_one_to_many(self, query_slot, slots, with_map_points=False, cap=0):
        num, nmp = [], []
        for s in slots:
            m = self.lists(self.frames[query_slot], self.frames[s])
            num.append(len(m))
            nmp.append(sum(1 for x1, y1, x2, y2 in m
                           if y1 * W + x1 in self.maps.get(query_slot, ()) and y2 * W + x2 in self.maps.get(s, ())))
        return np.array(num, np.int32), (np.array(nmp, np.int32) if with_map_points else None), None


def _image(number):
    img = np.zeros((H, W), np.uint8)
    img[0, 0], img[0, 1] = number // 256, number % 256
    return img


def _graphs(n, rng_seed):
    from mono_slam_framework_amd.keyframe_db import KeyFrame
    out = []
    for _ in range(2):
        rng = np.random.RandomState(rng_seed)
        kfs = [KeyFrame(10 + i, _image(10 + i), rng.choice(W * H, rng.randint(0, W * H // 2), replace=False))
               for i in range(n)]
        for i, kf in enumerate(kfs):
            others = [k for j, k in enumerate(kfs) if j != i]
            order = rng.permutation(len(others))
            kf.ordered_covisibility = [others[j] for j in order[:rng.randint(0, 15)]]
            kf.connected = set(kf.ordered_covisibility[:rng.randint(0, 5)])
            kf.mnLoopQuery = int(rng.choice([0, 0, 0, 10 + rng.randint(0, n)]))
        out.append(kfs)
    return out


def test_selection_logic_fuzz():
    from mono_slam_framework_amd.keyframe_db import KeyFrame, KeyFrameMatchDatabase
    for trial in range(60):
        rng = np.random.RandomState(trial)
        n = int(rng.randint(1, 30))
        stub = StubMatcher(trial)
        gpu_kfs, cpu_kfs = _graphs(n, 100 + trial)
        db = KeyFrameMatchDatabase(stub)
        for kf in gpu_kfs:
            db.add(kf)
        live = list(cpu_kfs)
        for step in range(6):
            if step == 3 and n > 2:                                    # a keyframe leaves, another comes back later
                k = int(rng.randint(0, len(live)))
                db.erase(gpu_kfs[cpu_kfs.index(live[k])])
                gone = live.pop(k)
            if step == 5 and n > 2:
                db.add(gpu_kfs[cpu_kfs.index(gone)])
                live.append(gone)
            mf = lambda a, b: stub.lists(StubMatcher.number(a), StubMatcher.number(b))   # noqa: E731
            if rng.rand() < 0.5:
                qi = int(rng.randint(0, n))
                min_mp = int(rng.choice([0, 0, 1, 3, 8]))
                got = db.DetectLoopCandidate(gpu_kfs[qi], min_mp)
                exp, nums, nmps = oracle_db.detect_loop_candidate(live, mf, cpu_kfs[qi], min_mp)
                np.testing.assert_array_equal(db.last_num_matches, nums)
                np.testing.assert_array_equal(db.last_num_mp, nmps)
                assert (got.id() if got else None) == (exp.id() if exp else None)
            else:
                qn = 1000 + trial * 10 + step
                got = db.DetectRelocalizationCandidates(KeyFrame(qn, _image(qn)))
                exp, nums = oracle_db.detect_relocalization_candidates(live, mf, KeyFrame(qn, _image(qn)))
                np.testing.assert_array_equal(db.last_num_matches, nums)
                assert [k.id() for k in got] == [k.id() for k in exp]
                for a, b in zip(gpu_kfs, cpu_kfs):
                    assert a.mnRelocQuery == b.mnRelocQuery and float(a.mRelocScore) == float(b.mRelocScore)


def test_initializer_oracle_known_answers():
    """exact correspondences under the hypothesis: every term is th (resp. thScore), added in f32 in match order"""
    n = 300
    rng = np.random.RandomState(0)
    p1 = np.stack([rng.randint(0, 600, n), rng.randint(0, 440, n)], 1)
    m = np.concatenate([p1, p1 + [7, -3]], 1).astype(np.int32)
    H21 = np.array([[1, 0, 7], [0, 1, -3], [0, 0, 1]], np.float32)
    H12 = np.array([[1, 0, -7], [0, 1, 3], [0, 0, 1]], np.float32)
    bad = np.array([[1, 0, 90], [0, 1, 0], [0, 0, 1]], np.float32)
    bad_inv = np.array([[1, 0, -90], [0, 1, 0], [0, 0, 1]], np.float32)
    best, scores, inl = oracle_init.find_best(0, np.stack([bad, H21, H21]), np.stack([bad_inv, H12, H12]), m, 1.0)
    exp = np.float32(0)
    for _ in range(2 * n):
        exp = np.float32(exp + np.float32(5.991))
    assert best == 1 and scores[0] == 0 and scores[1] == exp and scores[2] == exp and inl.all()   # first maximum kept
    # pure translation: F = [t]x, every correspondence lies on its epipolar line
    F = np.array([[0, 0, -3], [0, 0, -7], [3, 7, 0]], np.float32)
    best, scores, inl = oracle_init.find_best(1, F[None], None, m, 1.0)
    assert best == 0 and scores[0] == exp and inl.all()
    # an outlier: thresholds 5.991 (H) and 3.841 (F) are strict '>' on chi-square
    m2 = m.copy()
    m2[5, 2] += 2                                                       # 2 px off: chi2 = 4 in each direction
    best, scores, inl = oracle_init.find_best(0, H21[None], H12[None], m2, 1.0)
    assert inl.sum() == n and best == 0                                 # 4 < 5.991: still an inlier for H
    m2[5, 2] += 1                                                       # 3 px: chi2 = 9 > 5.991
    best, scores, inl = oracle_init.find_best(0, H21[None], H12[None], m2, 1.0)
    assert inl.sum() == n - 1 and not inl[5]
    best, scores, inl = oracle_init.find_best(0, H21[None], H12[None], m2, 2.0)                     # sigma 2: 9/4 < th
    assert inl.all()
