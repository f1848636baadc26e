"""TEST INFRASTRUCTURE ONLY (CPU oracle) -- never imported by the product package.

Literal restatement of SLAM_PIPELINE::KeyFrameMatchDatabase::DetectLoopCandidate and
::DetectRelocalizationCandidates (slam_pipeline/src/KeyFrameDatabase.cc:23-53 and :55-117) and of the KeyPointMap
lookup they use (slam_pipeline/src/KeyPointMap.cc:56-87), as per-keyframe loops over a CPU MatchFrames callable
(oracle.orb.FeatureMatcherOracle or oracle.loftr's matcher).  Parity: pinned only as far as the MatchFrames oracle
underneath is (ORB: unpinned, see oracle/orb_oracle.c; LoFTR: pinned by tests/golden/loftr_kat.npz); the selection
logic itself has no golden vectors in the reference (it has no tests), so it is restated line by line.

Keyframes are duck-typed: imGray, mappoint_keys (set of y*cols+x), id(), GetConnectedKeyFrames(),
GetBestCovisibilityKeyFrames(n), mnLoopQuery, mnRelocQuery, mRelocScore."""
import numpy as np


def get_map_point(frame, x, y, cols, rows):
    """KeyPointMap::GetMapPoint(keyPoint, diameter=5), KeyPointMap.cc:56-87: the sparse mask is read at the key point
    itself; the neighbourhood loop (:72-80) re-reads the same cell, so the result is the exact-cell lookup.
    Returns True when a map point is stored at (x, y)."""
    index = -1
    if 0 <= x < cols and 0 <= y < rows:
        key = y * cols + x                       # SetMapPoint: index = y*mCols + x, mask value index+1 (:40-43)
        index = key if key in frame.mappoint_keys else -1
    if index >= 0:
        radius = 5 // 2
        yy, xx = y - radius, x - radius
        while yy < y + radius:                   # both loops only ever re-evaluate the centre cell
            while xx < x + radius:
                if 0 <= xx < cols and 0 <= yy < rows:
                    key = y * cols + x
                    index = key if key in frame.mappoint_keys else -1
                xx += 1
            yy += 1
    return index >= 0


def detect_loop_candidate(frames, match_frames, pKF, minNumMPMatches):
    """KeyFrameDatabase.cc:23-53.  match_frames(img1, img2) -> int array [n, 4] (x1, y1, x2, y2).
    Returns (candidate or None, numMatches per keyframe, numMP per keyframe (-1 where the reference does not count))."""
    rows, cols = pKF.imGray.shape
    spConnectedKeyFrames = pKF.GetConnectedKeyFrames()
    loopCandidate, maxNumMP = None, 0
    nums, nmps = [], []
    for pKFi in frames:
        m = np.asarray(match_frames(pKF.imGray, pKFi.imGray)).reshape(-1, 4)
        numMatches = len(m)
        nums.append(numMatches)
        numMP = 0
        for x1, y1, x2, y2 in m:                 # counted for every keyframe here so tests can compare all counts
            if get_map_point(pKF, int(x1), int(y1), cols, rows) and get_map_point(pKFi, int(x2), int(y2), cols, rows):
                numMP += 1
        nmps.append(numMP)
        if numMatches != 0 and pKFi.mnLoopQuery != pKF.id():
            if pKFi not in spConnectedKeyFrames:
                if numMP > minNumMPMatches and numMP > maxNumMP:
                    loopCandidate, maxNumMP = pKFi, numMP
    return loopCandidate, nums, nmps


def detect_relocalization_candidates(frames, match_frames, pF):
    """KeyFrameDatabase.cc:55-117.  Returns (candidates, numMatches per keyframe)."""
    f32 = np.float32
    frameMatchCounts = []
    maxNumMatches = 0
    for pKFi in frames:
        numMatches = len(np.asarray(match_frames(pF.imGray, pKFi.imGray)).reshape(-1, 4))
        pKFi.mnRelocQuery = pF.id()
        pKFi.mRelocScore = f32(numMatches)
        frameMatchCounts.append((pKFi, numMatches))
        if numMatches > maxNumMatches:
            maxNumMatches = numMatches
    minNumMatches = int(f32(maxNumMatches) * f32(0.8))
    bestAccNumMatches = f32(0)
    accNumMatchFrames = []
    for pKFi, cnt in frameMatchCounts:
        if cnt >= minNumMatches:
            vpNeighs = pKFi.GetBestCovisibilityKeyFrames(10)
            bestNumMatches = f32(cnt)
            accNumMatches = bestNumMatches
            pBestKF = pKFi
            for pKF2 in vpNeighs:
                if pKF2.mnRelocQuery != pF.id():
                    continue
                accNumMatches = f32(accNumMatches + f32(pKF2.mRelocScore))
                if pKF2.mRelocScore > bestNumMatches:
                    pBestKF = pKF2
                    bestNumMatches = f32(pKF2.mRelocScore)
            accNumMatchFrames.append((pBestKF, accNumMatches))
            if accNumMatches > bestAccNumMatches:
                bestAccNumMatches = accNumMatches
    minNumMatchesToRetain = f32(0.75) * bestAccNumMatches
    spAlreadyAddedKF = []
    vpRelocCandidates = []
    for pKFi, acc in accNumMatchFrames:
        if f32(acc) > minNumMatchesToRetain:
            if not any(pKFi is k for k in spAlreadyAddedKF):
                vpRelocCandidates.append(pKFi)
                spAlreadyAddedKF.append(pKFi)
    return vpRelocCandidates, [c for _, c in frameMatchCounts]
