#!/usr/bin/env python3
"""Config-5 harness (BASELINE.json configs[4]), call-pattern replay form.

slam_pipeline itself needs OpenCV + g2o + PCL and cannot be built offline here, so this replays the SEQUENCE OF
MatchFrames CALLS its tracking / mapping loop issues on a synthetic monocular sequence with known motion, through the
plugged-in GPU matcher, and scores every match list against the ground-truth flow:

  per frame        TrackWithMotionModel      MatchFrames(current, last)              Tracking.cc:444
                   TrackReferenceKeyFrame    MatchFrames(current, refKF)             Tracking.cc:383
                   SearchLocalPoints         MatchFrames(current, KF_i) for the local key frames   Tracking.cc:595-632
  per new keyframe DetectLoopCandidate       MatchFrames(newKF, every KF in the DB)  KeyFrameDatabase.cc:31-50
                   SearchInNeighbors         MatchFrames(newKF, covisible KFs)       LocalMapping.cc:325-358

Two ways of issuing the same sequence, whose match lists must be identical call by call:
  * "slots": the one-vs-many loops through the extract-once / match-many entry points (msf_extract_device +
    msf_match_slots_device), what an adapted caller would use;
  * "plain": every call as the untouched pipeline makes it -- MatchFrames(frame, frame) on host images -- served by the
    handle's transparent per-frame cache (msf_match_pair); msf_frame_cache_stats must show one miss per distinct frame.

The camera looks straight down on a large synthetic plane and moves with a SIMILARITY: it pans, rotates in plane and
changes height (scale), so steered rBRIEF and cross-octave matching are exercised end to end.  Frames are rendered with an
integer-only warp (16.16 fixed point, nearest sampling), so the sequence is bit-reproducible and the true image of any
pixel of one frame in another is known: a match is an inlier when it lands within `tol` pixels of it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FX = 16                      # fraction bits of the warp


def make_canvas(w, h, seed=7):
    """A big textured plane: blocky + smooth layers from the deterministic synthetic generator, tiled."""
    from mono_slam_framework_amd import synth
    tw, th = 1280, 720
    nx, ny = (w + tw - 1) // tw, (h + th - 1) // th
    canvas = np.zeros((ny * th, nx * tw), np.uint8)
    k = 0
    for j in range(ny):
        for i in range(nx):
            a, _ = synth.synth_pair(seed * 100 + k, tw, th, mode=2 if (i + j) % 2 else 1, noise=3)
            canvas[j * th:(j + 1) * th, i * tw:(i + 1) * tw] = a
            k += 1
    return canvas[:h, :w]


class Camera:
    """frame pixel (u, v) sees canvas point c + s R(theta) (u - w/2, v - h/2), as 16.16 integers"""

    def __init__(self, n, w, h, cw, ch, rotate_deg=0.0, zoom=1.0, cell=0):
        t = np.arange(n)
        self.w, self.h = w, h
        theta = np.deg2rad(rotate_deg) * np.sin(2 * np.pi * t / 500.0)
        s = np.power(float(zoom), np.sin(2 * np.pi * t / 700.0 + 0.5))
        # keep the rotated / zoomed view inside the canvas
        reach = 0.5 * np.hypot(w, h) * max(float(zoom), 1.0) + 2
        ax, ay = (cw / 2 - reach) * 0.9, (ch / 2 - reach) * 0.9
        assert ax > 0 and ay > 0, "canvas too small for this camera"
        cx = cw / 2 + ax * np.sin(2 * np.pi * t / 400.0)
        cy = ch / 2 + ay * np.sin(2 * np.pi * t / 650.0 + 1.0)
        if cell:                                   # LoFTR: pure translation in whole cells, so matches sit on the 16-px grid
            cx = np.round((cx - w / 2) / cell) * cell + w / 2
            cy = np.round((cy - h / 2) / cell) * cell + h / 2
        one = 1 << FX
        self.a = np.round(s * np.cos(theta) * one).astype(np.int64)
        self.b = np.round(s * np.sin(theta) * one).astype(np.int64)
        self.cx = np.round(cx * one).astype(np.int64)
        self.cy = np.round(cy * one).astype(np.int64)
        u = np.arange(w, dtype=np.int64) * 2 - w       # 2 (u - w/2): half-pixel exact in integers
        v = np.arange(h, dtype=np.int64) * 2 - h
        self.u2, self.v2 = np.meshgrid(u, v)

    def render(self, canvas, i):
        a, b = self.a[i], self.b[i]
        px = (2 * self.cx[i] + a * self.u2 - b * self.v2 + (1 << FX)) >> (FX + 1)
        py = (2 * self.cy[i] + b * self.u2 + a * self.v2 + (1 << FX)) >> (FX + 1)
        return np.ascontiguousarray(canvas[py, px])

    def to_canvas(self, i, pts):
        """float canvas coordinates of frame-i pixels pts [n, 2]"""
        one = float(1 << FX)
        a, b = self.a[i] / one, self.b[i] / one
        du, dv = pts[:, 0] - self.w / 2.0, pts[:, 1] - self.h / 2.0
        return np.stack([self.cx[i] / one + a * du - b * dv, self.cy[i] / one + b * du + a * dv], 1)

    def from_canvas(self, i, p):
        one = float(1 << FX)
        a, b = self.a[i] / one, self.b[i] / one
        det = a * a + b * b
        dx, dy = p[:, 0] - self.cx[i] / one, p[:, 1] - self.cy[i] / one
        return np.stack([(a * dx + b * dy) / det + self.w / 2.0, (-b * dx + a * dy) / det + self.h / 2.0], 1)

    def overlap(self, i, f):
        """fraction of frame i's pixels (a coarse lattice) whose canvas point is visible in frame f"""
        g = np.stack(np.meshgrid(np.linspace(0, self.w - 1, 24), np.linspace(0, self.h - 1, 18)), -1).reshape(-1, 2)
        q = self.from_canvas(f, self.to_canvas(i, g))
        return float(((q[:, 0] >= 0) & (q[:, 0] < self.w) & (q[:, 1] >= 0) & (q[:, 1] < self.h)).mean())

    def scale_ratio(self, i, f):
        return float(np.hypot(self.a[i], self.b[i]) / np.hypot(self.a[f], self.b[f]))


def inliers(cam, i, f, m, tol):
    """matches m [n, 4] = (x1, y1) in frame i, (x2, y2) in frame f"""
    if len(m) == 0:
        return 0
    q = cam.from_canvas(f, cam.to_canvas(i, m[:, 0:2].astype(np.float64)))
    d = q - m[:, 2:4]
    return int(((np.abs(d[:, 0]) <= tol) & (np.abs(d[:, 1]) <= tol)).sum())


def call_sequence(cam, n_frames, max_local, kf_every, min_overlap, db_size):
    """per frame: (list of frames the current one is matched against, new-key-frame flag, frames of the DB loop)"""
    kfs, seq = [], []
    for i in range(n_frames):
        if i == 0:
            kfs.append(0)
            seq.append(([], True, []))
            continue
        local = [f for f in kfs[-max_local:] if cam.overlap(i, f) >= min_overlap]
        others = [i - 1] + [f for f in local if f != i - 1]
        newkf = i % kf_every == 0
        db = list(kfs[-db_size:]) if newkf else []
        seq.append((others, newkf, db))
        if newkf:
            kfs.append(i)
    return seq


def replay(n_frames=1000, width=640, height=480, ratio=0.6, max_local=8, kf_every=12, min_overlap=0.5, tol=4,
           check_every=25, verbose=False, rotate_deg=0.0, zoom=1.0, db_size=63, mode="slots", matcher="orb",
           threshold=0.15, keep_lists=False, min_matches=15):
    """mode "slots": extract-once / match-many; "plain": every call a MatchFrames on host images (transparent cache).
    Returns the statistics; with keep_lists also every list in call order (to compare two modes)."""
    import torch
    from mono_slam_framework_amd.matcher import DNNFeatureMatcher, FeatureMatcher
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    loftr = matcher == "loftr"
    canvas = make_canvas(width * 3 + 200, height * 3 + 200)
    cam = Camera(n_frames, width, height, canvas.shape[1], canvas.shape[0], rotate_deg, zoom, cell=16 if loftr else 0)
    seq = call_sequence(cam, n_frames, max_local, kf_every, min_overlap, db_size)
    n_slots = 2 + 64                       # slot 0 = current, 1 = last, 2.. = key frames (ring)
    cap = 4096 if loftr else 1024
    fm = (DNNFeatureMatcher(threshold=threshold, max_batch_pairs=n_slots // 2) if loftr
          else FeatureMatcher(ratio, width, height, max_batch_pairs=n_slots // 2))
    d_out = torch.zeros((64, cap, 4), dtype=torch.int32, device=dev)
    d_cnt = torch.zeros((64,), dtype=torch.int32, device=dev)
    stats = {"calls": 0, "matches": 0, "inliers": 0, "cache_checks": 0, "lost": 0, "keyframes": 0}
    lost, worst, all_lists = [], 1.0, []
    frames, kf_slot = {}, {}

    def upload(img, slot):
        fm.extract_device(torch.from_numpy(img[None]).to(dev), first_slot=slot)

    def many(cur, cur_slot, others):
        """MatchFrames(cur, other_k) for all k"""
        if mode == "plain":
            return [fm.MatchFrames(frames[cur], frames[f], cap=cap) for f in others]
        sa = torch.full((len(others),), cur_slot, dtype=torch.int32, device=dev)
        sb = torch.tensor([1 if f == cur - 1 and cur_slot == 0 else kf_slot[f] for f in others], dtype=torch.int32, device=dev)
        fm.match_slots_device(sa, sb, d_out[:len(others)], d_cnt[:len(others)])
        cnt = d_cnt[:len(others)].cpu().numpy()
        out = d_out[:len(others)].cpu().numpy()
        return [out[k, :cnt[k]].copy() for k in range(len(others))]

    def common_keypoints(i, f, slot_i, slot_f):
        """key points of frame i that frame f could have too -- their image lies inside f's key-point domain, 31 px inside
        the frame (edgeThreshold) -- and the other way round (slots mode, ORB).  cv::ORB keeps the 500 best corners of the
        WHOLE frame (no grid bucketing), so two views whose common part is the weakly textured part of one of them share
        few: a short list then says nothing about the matcher (r02's 89 "lost" lists; e.g. frame 68 against key frame 36:
        63 % overlap, but every key point of the key frame inside it falls in the other frame's 31-px border)"""
        if mode != "slots" or loftr:
            return None
        ki, kf = fm.keypoints(slot_i, cache=True), fm.keypoints(slot_f, cache=True)
        out = []
        for (a, b, k) in ((i, f, ki), (f, i, kf)):
            q = cam.from_canvas(b, cam.to_canvas(a, np.stack([k["x"], k["y"]], 1).astype(np.float64)))
            out.append(int(((q[:, 0] >= 31) & (q[:, 0] < width - 31) & (q[:, 1] >= 31) & (q[:, 1] < height - 31)).sum()))
        return out

    def score(i, f, m, tracked, slot_f=None):
        nonlocal worst
        k = inliers(cam, i, f, m, tol)
        stats["calls"] += 1
        stats["matches"] += len(m)
        stats["inliers"] += k
        if len(m) >= min_matches:           # minLocalMatchCount (SlamParameters.h)
            worst = min(worst, k / len(m))
        elif tracked:
            stats["lost"] += 1
            lost.append({"frame": i, "against": f, "matches": int(len(m)), "overlap": round(cam.overlap(i, f), 3),
                         "scale_ratio": round(cam.scale_ratio(i, f), 3),
                         "common_keypoints": common_keypoints(i, f, 0, slot_f) if slot_f is not None else None})

    t0 = time.perf_counter()
    n_kf = 0
    for i in range(n_frames):
        cur = cam.render(canvas, i)
        frames[i] = cur
        others, newkf, db = seq[i]
        if mode == "slots":
            upload(cur, 0)                  # features of the current frame, once
        if others:
            lists = many(i, 0, others)
            for f, m in zip(others, lists):
                score(i, f, m, True, (1 if f == i - 1 else kf_slot[f]) if mode == "slots" else None)
            if keep_lists:
                all_lists.extend(lists)
            if mode == "slots" and i % check_every == 0:   # the cache must be invisible: same lists as stateless calls
                for f, m in list(zip(others, lists))[:2]:
                    ref = fm.MatchFrames(cur, frames[f], cap=cap)
                    assert ref.shape == m.shape and np.array_equal(ref, m), "slot path differs from MatchFrames at frame %d" % i
                    stats["cache_checks"] += 1
        if newkf:
            # new key frame: DetectLoopCandidate + SearchInNeighbors = new KF against every KF in the DB
            slot = 2 + (n_kf % 64)
            if mode == "slots":
                upload(cur, slot)
            if db:
                lists = many(i, slot, db)
                for f, m in zip(db, lists):
                    score(i, f, m, False)
                if keep_lists:
                    all_lists.extend(lists)
            kf_slot[i] = slot
            n_kf += 1
            stats["keyframes"] += 1
        if mode == "slots":
            upload(cur, 1)                  # current becomes last
        frames.pop(i - max(kf_every * 70, 2), None)
        if verbose and i % 100 == 0:
            print("frame %d: %d calls, inlier ratio %.4f" % (i, stats["calls"], stats["inliers"] / max(stats["matches"], 1)))
    dt = time.perf_counter() - t0
    stats.update({"frames": n_frames, "mode": mode, "matcher": matcher, "rotate_deg": rotate_deg, "zoom": zoom,
                  "seconds": round(dt, 2), "frames_per_sec": round(n_frames / dt, 1),
                  "match_calls_per_sec": round(stats["calls"] / dt, 1),
                  "matches_per_call": round(stats["matches"] / max(stats["calls"], 1), 1),
                  "inlier_ratio": round(stats["inliers"] / max(stats["matches"], 1), 4), "worst_list_inlier_ratio": round(worst, 4),
                  "lost_max_overlap": max([e["overlap"] for e in lost], default=0.0),
                  # why a list is short: the two views share few of their 500 key points (see common_keypoints)
                  "lost_max_common_keypoints": max([min(e["common_keypoints"]) for e in lost if e["common_keypoints"]], default=0),
                  "lost_lists": lost[:12]})
    if mode == "plain":
        h, m_, c = fm.frame_cache_stats()
        stats["frame_cache"] = {"hits": h, "misses": m_, "capacity": c}
    fm.close()
    if keep_lists:
        return stats, all_lists
    return stats


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--mode", default="slots", choices=["slots", "plain"])
    ap.add_argument("--matcher", default="orb", choices=["orb", "loftr"])
    ap.add_argument("--rotate-deg", type=float, default=0.0)
    ap.add_argument("--zoom", type=float, default=1.0)
    ap.add_argument("--db-size", type=int, default=63)
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    print(json.dumps(replay(a.frames, verbose=a.verbose, mode=a.mode, matcher=a.matcher, rotate_deg=a.rotate_deg,
                            zoom=a.zoom, db_size=a.db_size)))
