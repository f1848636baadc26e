"""Host-side mirror of SLAM_PIPELINE::KeyFrameMatchDatabase (slam_pipeline/include/KeyFrameDatabase.h,
slam_pipeline/src/KeyFrameDatabase.cc:9-117) on top of the C ABI -- SURVEY.md 8f row 3.

The reference scores a query against every stored keyframe with one MatchFrames call each (re-extracting both frames
every time) and counts, on the host, the matches whose two endpoints carry a map point.  Here

  * a keyframe is uploaded once, on add() (msf_store_frame): its image stays in HBM, (ORB) its features are extracted
    once into a feature slot, and its KeyPointMap occupancy lives in a device bitmap (msf_set_mappoints);
  * a query is ONE launch sequence (msf_match_one_to_many): extract the query, match it against all keyframe slots,
    count map-point matches, read back two int32 per keyframe;
  * the candidate selection that follows (KeyFrameDatabase.cc:45-48, 73-115) is the reference's host logic on those
    counts, restated below with the same float/size_t conversions.

Same names and argument meaning as the reference class; keyframes are any objects with the attributes of `KeyFrame`
below (the slam_pipeline KeyFrame fields the two Detect* functions touch)."""
import numpy as np

from .matcher import FeatureMatcher


class KeyFrame:
    """The fields of SLAM_PIPELINE::KeyFrame / FrameBase that KeyFrameMatchDatabase reads or writes."""

    def __init__(self, frame_id, image, mappoint_keys=()):
        self.mnId = int(frame_id)
        self.imGray = image                         # FrameBase::imGray (FrameBase.h:45)
        self.mappoint_keys = set(int(k) for k in mappoint_keys)   # KeyPointMap: keys y*cols + x holding a map point
        self.connected = set()                      # KeyFrame::GetConnectedKeyFrames()
        self.ordered_covisibility = []              # KeyFrame::GetBestCovisibilityKeyFrames(N) = first N of these
        self.mnLoopQuery = 0
        self.mnRelocQuery = 0
        self.mRelocScore = 0.0

    def id(self):
        return self.mnId

    def GetConnectedKeyFrames(self):
        return self.connected

    def GetBestCovisibilityKeyFrames(self, n):
        return self.ordered_covisibility[:n]


class KeyFrameMatchDatabase:
    """KeyFrameMatchDatabase(featureMatcher): add / erase / clear / DetectLoopCandidate /
    DetectRelocalizationCandidates (KeyFrameDatabase.cc:9-21, 23-53, 55-117).  Holds up to
    `featureMatcher.max_batch_pairs` keyframes (feature/map slot 0 is the query's)."""

    def __init__(self, featureMatcher, cap=None):
        self.mFeatureMatcher = featureMatcher
        self.mFrames = []
        # the adapters' result capacities (hip_feature_matcher.h): MatchFramesResult never holds more
        self._cap = cap or (2048 if isinstance(featureMatcher, FeatureMatcher) else 4096)
        n_slots = min(2 * featureMatcher.max_batch_pairs, featureMatcher.max_batch_pairs + 1)
        self._free = list(range(n_slots - 1, 0, -1))      # slot 0 is the query's
        self._slot = {}          # id(kf object) -> frame / feature / map slot
        self._map_state = {}     # slot -> frozenset of keys last uploaded
        self.last_num_matches = None
        self.last_num_mp = None

    # --- KeyFrameDatabase.cc:12-21 -------------------------------------------------------------------------------
    def add(self, pKF):
        if not self._free:
            raise RuntimeError("KeyFrameMatchDatabase: more keyframes than max_batch_pairs")
        s = self._free.pop()
        self.mFeatureMatcher.store_frame(s, np.ascontiguousarray(pKF.imGray))
        self._slot[id(pKF)] = s
        self._map_state.pop(s, None)
        self.mFrames.append(pKF)

    def erase(self, pKF):
        for i, f in enumerate(self.mFrames):
            if f is pKF:
                del self.mFrames[i]
                self._free.append(self._slot.pop(id(pKF)))
                return

    def clear(self):
        for f in self.mFrames:
            self._free.append(self._slot.pop(id(f)))
        self.mFrames = []

    # --- the N x MatchFrames of both Detect* functions as one launch sequence -----------------------------------------
    def _sync_map(self, slot, keys):
        keys = frozenset(keys)
        if self._map_state.get(slot) != keys:
            self.mFeatureMatcher.set_mappoints(slot, sorted(keys))
            self._map_state[slot] = keys

    def _match_all(self, query, with_map_points):
        """-> (numMatches[n], numMP[n] or None) for the query frame against mFrames, in mFrames order."""
        if not self.mFrames:
            return np.zeros(0, np.int64), np.zeros(0, np.int64)
        m = self.mFeatureMatcher
        slots = [self._slot[id(f)] for f in self.mFrames]
        m.store_frame(0, np.ascontiguousarray(query.imGray))
        if with_map_points:
            self._sync_map(0, getattr(query, "mappoint_keys", ()))
            for f, s in zip(self.mFrames, slots):
                self._sync_map(s, f.mappoint_keys)
        cnt, nmp, _ = m.match_one_to_many(0, slots, with_map_points)
        # MatchFramesResult holds at most `cap` matches (the adapter's buffer); a failed pair gives an empty result
        num = np.clip(cnt.astype(np.int64), 0, self._cap)
        num_mp = nmp.astype(np.int64) if with_map_points else None
        self.last_num_matches, self.last_num_mp = num, num_mp
        return num, num_mp

    # --- the two queries (replace KeyFrameMatchDatabase::DetectLoopCandidate / DetectRelocalizationCandidates,
    # --- slam_pipeline/src/KeyFrameDatabase.cc:23-53, 55-117); rules as in csrc/hip_keyframe_database.h ---------------
    def DetectLoopCandidate(self, query, min_shared):
        """The stored key frame sharing the most map points with `query` (more than min_shared), among those that
        matched at all, were not already handed out for this query and are not its covisibility neighbours."""
        counts, shared = self._match_all(query, True)
        neighbours = query.GetConnectedKeyFrames()
        best, best_shared = None, min_shared
        for kf, n, mp in zip(self.mFrames, counts, shared):
            usable = n != 0 and kf.mnLoopQuery != query.id() and kf not in neighbours
            if usable and int(mp) > best_shared:
                best, best_shared = kf, int(mp)
        return best

    def DetectRelocalizationCandidates(self, query):
        """Representatives of the covisibility groups whose summed match count exceeds 75 % of the heaviest group's."""
        counts, _ = self._match_all(query, False)
        f32 = np.float32
        qid = query.id()
        for kf, n in zip(self.mFrames, counts):          # pass 1: stamp every key frame of this query with its count
            kf.mnRelocQuery = qid
            kf.mRelocScore = f32(int(n))
        top = int(counts.max()) if len(counts) else 0
        cutoff = int(f32(top) * f32(0.8))                 # the reference truncates the f32 product to an integer
        groups = []                                       # pass 2: (representative, f32 weight) per strong key frame
        for kf, n in zip(self.mFrames, counts):
            if int(n) < cutoff:
                continue
            weight = rep_score = f32(int(n))
            rep = kf
            for nb in kf.GetBestCovisibilityKeyFrames(10):
                if nb.mnRelocQuery != qid:                # did not take part in this query
                    continue
                weight = f32(weight + f32(nb.mRelocScore))
                if nb.mRelocScore > rep_score:
                    rep, rep_score = nb, f32(nb.mRelocScore)
            groups.append((rep, weight))
        keep_above = f32(0.75) * max([w for _, w in groups], default=f32(0))
        result = []                                       # pass 3: each representative once, in group order
        for rep, weight in groups:
            if weight > keep_above and not any(rep is r for r in result):
                result.append(rep)
        return result
