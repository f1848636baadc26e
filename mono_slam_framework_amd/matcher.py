"""Host-side mirror of the reference's matcher classes over the C ABI (include/msf_abi.h).

  FeatureMatcher     <- ::FeatureMatcher     (src/featurematcher.h:7-22,   src/featurematcher.cpp:3-47)
  DNNFeatureMatcher  <- ::DNNFeatureMatcher  (src/dnnfeaturematcher.h:9-36, src/dnnfeaturematcher.cpp:11-103)

Same names and argument meaning (threshold, SetThreshold, MatchFrames(frame1, frame2)); a frame is the
8-bit single-channel image the reference reads from FrameBase::imGray (slam_pipeline/include/FrameBase.h:45).
MatchFrames returns an int32 [m, 4] array: columns 0:2 are MatchFramesResult::keyPoints1, 2:4 keyPoints2
(slam_pipeline/include/FeatureMatcher.h:15-19).  The C++ twin for linking into slam_pipeline is
csrc/hip_feature_matcher.h.  All compute happens in libmsf.so on the GPU; there is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _lib


class MsfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("msf error %d: %s" % (code, msg))
        self.code = code


class _Matcher:
    _kind = None

    def __init__(self, threshold, image_width, image_height, device=0, max_batch_pairs=1, flags=0,
                 weights_path=None):
        self._L = _lib.load()
        self._h = C.c_void_p()
        cfg = _lib.Config()
        self._L.msf_default_config(C.byref(cfg), self._kind)
        cfg.threshold = threshold
        cfg.device = device
        cfg.image_width = image_width
        cfg.image_height = image_height
        cfg.max_batch_pairs = max_batch_pairs
        cfg.flags = flags
        self._wpath = weights_path.encode() if weights_path else None
        cfg.weights_path = self._wpath
        rc = self._L.msf_create(C.byref(cfg), C.byref(self._h))
        if rc != _lib.MSF_OK:
            msg = self._L.msf_last_error(None).decode()
            self._h = C.c_void_p()
            raise MsfError(rc, msg)
        self.width, self.height = image_width, image_height
        self.max_batch_pairs = max_batch_pairs
        self.device = device
        self.threshold = threshold

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.msf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != _lib.MSF_OK:
            raise MsfError(rc, self._L.msf_last_error(self._h).decode())

    # --- reference API -------------------------------------------------------------------------
    def SetThreshold(self, value):
        self._check(self._L.msf_set_threshold(self._h, float(value)))
        self.threshold = float(value)

    @staticmethod
    def _image(arr):
        if arr.dtype != np.uint8 or arr.ndim != 2 or arr.strides[1] != 1:
            raise ValueError("frames must be 2-D uint8 with unit column stride (CV_8UC1)")
        return _lib.Image(arr.ctypes.data, arr.shape[1], arr.shape[0], arr.strides[0])

    def MatchFrames(self, frame1, frame2, cap=4096):
        a, b = self._image(frame1), self._image(frame2)
        out = np.zeros((cap,), _lib.MATCH_DTYPE)
        n = C.c_int32(0)
        self._check(self._L.msf_match_pair(self._h, C.byref(a), C.byref(b), out.ctypes.data, cap, C.byref(n)))
        m = min(n.value, cap)
        return out[:m].view(np.int32).reshape(m, 4).copy()

    def frame_cache_stats(self):
        """(hits, misses, capacity) of the transparent per-frame cache behind MatchFrames (a miss = one extraction)."""
        hits, miss, cap = C.c_uint64(0), C.c_uint64(0), C.c_int32(0)
        self._check(self._L.msf_frame_cache_stats(self._h, C.byref(hits), C.byref(miss), C.byref(cap)))
        return hits.value, miss.value, cap.value

    # --- batched forms -------------------------------------------------------------------------
    def match_batch(self, frames1, frames2, cap=4096):
        n = len(frames1)
        A = (_lib.Image * n)(*[self._image(f) for f in frames1])
        B = (_lib.Image * n)(*[self._image(f) for f in frames2])
        out = np.zeros((n, cap), _lib.MATCH_DTYPE)
        cnt = np.zeros((n,), np.int32)
        self._check(self._L.msf_match_batch(self._h, n, A, B, out.ctypes.data, cap, cnt.ctypes.data))
        return [out[i, :min(cnt[i], cap)].view(np.int32).reshape(-1, 4).copy() for i in range(n)]

    def match_batch_raw(self, frames1, frames2, cap=4096):
        """match_batch that tolerates MSF_ERR_CAPACITY: returns (n_out int32 [n] with -1 for a pair without a valid result,
        lists).  Any other error raises."""
        n = len(frames1)
        A = (_lib.Image * n)(*[self._image(f) for f in frames1])
        B = (_lib.Image * n)(*[self._image(f) for f in frames2])
        out = np.zeros((n, cap), _lib.MATCH_DTYPE)
        cnt = np.zeros((n,), np.int32)
        rc = self._L.msf_match_batch(self._h, n, A, B, out.ctypes.data, cap, cnt.ctypes.data)
        if rc not in (_lib.MSF_OK, _lib.MSF_ERR_CAPACITY):
            self._check(rc)
        return cnt, [out[i, :min(max(cnt[i], 0), cap)].view(np.int32).reshape(-1, 4).copy() for i in range(n)]

    def match_batch_device(self, d_a, d_b, d_out, d_n_out, stream=None):
        """d_a/d_b: torch uint8 CUDA tensors [n, H, W(pitch)] resident in HBM; d_out int32 [n, cap, 4];
        d_n_out int32 [n].  Asynchronous on `stream` (an int hipStream_t; None = handle stream + sync)."""
        n = d_a.shape[0]
        assert d_a.is_cuda and d_b.is_cuda and d_out.is_cuda and d_n_out.is_cuda
        assert d_a.stride(2) == 1 and d_b.stride() == d_a.stride()
        cap = d_out.shape[1]
        self._check(self._L.msf_match_batch_device(
            self._h, n, d_a.data_ptr(), d_b.data_ptr(), d_a.stride(0), d_a.stride(1),
            d_out.data_ptr(), cap, d_n_out.data_ptr(), stream))

    def extract_device(self, d_frames, first_slot=0, stream=None):
        """Per-frame part once (ORB features / LoFTR backbone tokens) into slots first_slot.. of the handle."""
        self._check(self._L.msf_extract_device(self._h, d_frames.shape[0], d_frames.data_ptr(), d_frames.stride(0),
                                               d_frames.stride(1), first_slot, stream))

    def match_slots_device(self, d_slot_a, d_slot_b, d_out, d_n_out, stream=None):
        """Pairs of slots (int32 CUDA tensors) -> match lists, as match_batch_device."""
        self._check(self._L.msf_match_slots_device(self._h, d_slot_a.shape[0], d_slot_a.data_ptr(),
                                                   d_slot_b.data_ptr(), d_out.data_ptr(), d_out.shape[1],
                                                   d_n_out.data_ptr(), stream))

    def pack_matches_device(self, d_out, d_n_out, d_packed, d_offsets, stream=None):
        """[n, cap, 4] + counts -> contiguous [total, 4] prefix of d_packed, offsets int32 [n + 1]."""
        self._check(self._L.msf_pack_matches_device(self._h, d_out.shape[0], d_out.data_ptr(), d_out.shape[1],
                                                    d_n_out.data_ptr(), d_packed.data_ptr(), d_offsets.data_ptr(),
                                                    stream))

    def set_mappoints(self, map_slot, keys):
        """KeyPointMap occupancy of one frame: `keys` = pixel keys y*cols + x that hold a map point."""
        keys = np.ascontiguousarray(np.asarray(keys, np.int32).reshape(-1))
        self._check(self._L.msf_set_mappoints(self._h, map_slot, keys.ctypes.data, keys.size))

    def count_mappoint_matches_device(self, d_out, d_n_out, d_map_a, d_map_b, d_num_mp, stream=None):
        """d_num_mp[i] = matches of pair i with a map point at both endpoints (KeyFrameDatabase.cc:37-44)."""
        self._check(self._L.msf_count_mappoint_matches_device(
            self._h, d_out.shape[0], d_out.data_ptr(), d_out.shape[1], d_n_out.data_ptr(), d_map_a.data_ptr(),
            d_map_b.data_ptr(), d_num_mp.data_ptr(), stream))

    def check_hypotheses(self, model, m21, m12, matches, sigma=1.0):
        """Initializer::CheckHomography (model 0: m21 = H21, m12 = H12) / CheckFundamental (model 1: m21 = F21) for
        all RANSAC hypotheses [n_hyp, 3, 3] on the match list [n, 4] (Initializer.cc:152-245, 322-487).
        -> (best index or -1, scores f32 [n_hyp], vbMatchesInliers of the best [n])"""
        m21 = np.ascontiguousarray(m21, np.float32).reshape(-1, 9)
        m12 = None if m12 is None else np.ascontiguousarray(m12, np.float32).reshape(-1, 9)
        m = np.ascontiguousarray(matches, np.int32).reshape(-1, 4)
        scores = np.zeros(len(m21), np.float32)
        inl = np.zeros(max(len(m), 1), np.uint8)
        best = C.c_int32(-1)
        self._check(self._L.msf_check_hypotheses(self._h, model, len(m21), m21.ctypes.data,
                                                 None if m12 is None else m12.ctypes.data, len(m), m.ctypes.data,
                                                 float(sigma), scores.ctypes.data, C.byref(best), inl.ctypes.data))
        return best.value, scores, inl[:len(m)].astype(bool)

    def render_match_image(self, frame1, frame2, matches, has_mp1=None, has_mp2=None):
        """Tracking::CreateCurrentMatchImage (Tracking.cc:899-940) -> uint8 [H, 2W, 3]."""
        a, b = self._image(frame1), self._image(frame2)
        m = np.ascontiguousarray(matches, np.int32).reshape(-1, 4)
        f1 = None if has_mp1 is None else np.ascontiguousarray(has_mp1, np.uint8)
        f2 = None if has_mp2 is None else np.ascontiguousarray(has_mp2, np.uint8)
        out = np.zeros((self.height, 2 * self.width, 3), np.uint8)
        self._check(self._L.msf_render_match_image(self._h, C.byref(a), C.byref(b), m.ctypes.data, len(m),
                                                   None if f1 is None else f1.ctypes.data,
                                                   None if f2 is None else f2.ctypes.data, out.ctypes.data,
                                                   out.strides[0]))
        return out

    def store_frame(self, slot, frame):
        """Uploads a host frame into resident frame slot `slot` (ORB: and extracts its features once)."""
        img = self._image(frame)
        self._check(self._L.msf_store_frame(self._h, slot, C.byref(img)))

    def match_one_to_many(self, query_slot, slots, with_map_points=False, cap=0):
        """MatchFrames(frame[query_slot], frame[s]) for s in slots, one launch sequence.
        -> (num_matches[n], num_mp[n] or None, lists or None)"""
        slots = np.ascontiguousarray(np.asarray(slots, np.int32).reshape(-1))
        n = slots.size
        num = np.zeros((n,), np.int32)
        nmp = np.zeros((n,), np.int32) if with_map_points else None
        out = np.zeros((n, cap), _lib.MATCH_DTYPE) if cap else None
        self._check(self._L.msf_match_one_to_many(self._h, query_slot, n, slots.ctypes.data, num.ctypes.data,
                                                  nmp.ctypes.data if with_map_points else None,
                                                  out.ctypes.data if cap else None, cap))
        lists = None
        if cap:
            lists = [out[i, :min(max(num[i], 0), cap)].view(np.int32).reshape(-1, 4).copy() for i in range(n)]
        return num, nmp, lists

    def last_error(self):
        """msf_last_error(h): the text of the handle's last failure -- or note (a call that succeeded may leave one)"""
        return self._L.msf_last_error(self._h).decode()

    def stage_times(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        n = self._L.msf_stage_times(self._h, names, ms, 16)
        return {names[i].decode(): float(ms[i]) for i in range(n)}

    # --- introspection (parity tests) ----------------------------------------------------------
    def _debug(self, what, slot, level, dtype, cap_bytes):
        buf = np.zeros((cap_bytes,), np.uint8)
        nb = C.c_size_t(0)
        self._check(self._L.msf_debug_get(self._h, what, slot, level, buf.ctypes.data, cap_bytes, C.byref(nb)))
        if nb.value > cap_bytes:
            return self._debug(what, slot, level, dtype, nb.value)
        return buf[:nb.value].view(dtype).copy()


class FeatureMatcher(_Matcher):
    """ORB (cv::ORB::create() defaults) + BruteForce-Hamming 2-NN + ratio test, on the GPU."""
    _kind = _lib.MSF_KIND_ORB

    def __init__(self, threshold=0.8, image_width=640, image_height=480, **kw):
        super().__init__(threshold, image_width, image_height, **kw)

    def level_sizes(self):
        return self._debug(_lib.DBG_LEVEL_SIZES, 0, 0, np.int32, 8 * 4 * 4).reshape(8, 4)

    def walk_mode(self):
        """(per_level, stalls): whether this handle launches the walker level by level -- MSF_ORB_WALK_PER_LEVEL=1, or
        after a unit of a one-launch walker gave up a bounded wait (include/msf_abi.h, "Walker stall") -- and how many
        units have given up since the handle was made.  Asked of the handle, not of the environment."""
        v = self._debug(_lib.DBG_WALK_MODE, 0, 0, np.int32, 8)
        return bool(v[0]), int(v[1])

    def walker_launches(self, stage):
        """Launches of the dominant kernel (the streaming walker k_walk) in one extraction of a batch: ONE -- all levels,
        their thresholds and the pyramid in one launch (r03: 2 chains x 8 levels of sampler + walker launches); eight when
        the handle runs level by level (walk_mode)."""
        return 8 if (stage == "pyramid_fast" and self.walk_mode()[0]) else 1

    # Introspection of the feature slots.  `slot` counts from the scratch slots of the last MatchFrames / match_batch
    # call (frame A of pair i = i, frame B = n_pairs + i); cache=True addresses the per-frame cache slots of
    # extract_device / store_frame instead (the two ranges are disjoint: [2P, 4P) and [0, 2P), P = max_batch_pairs).
    def _slot(self, slot, cache):
        return slot if cache else 2 * self.max_batch_pairs + slot

    def level_pixels(self, slot, level, cache=False):
        w, h, pitch, _ = self.level_sizes()[level]
        return self._debug(_lib.DBG_LEVEL_PIXELS, self._slot(slot, cache), level, np.uint8,
                           int(pitch) * int(h)).reshape(h, pitch)[:, :w]

    def fast_candidates(self, slot, level, cache=False):
        return self._debug(_lib.DBG_FAST_CANDS, self._slot(slot, cache), level, np.int32, 1 << 20).reshape(-1, 3)

    def fast_tau(self, slot, cache=False):
        """int32 [8, 2]: per level the FAST score threshold the candidate list was built with, and its first estimate."""
        return self._debug(_lib.DBG_FAST_TAU, self._slot(slot, cache), 0, np.int32, 64).reshape(-1, 2)

    def stage1(self, slot, level, cache=False):
        return self._debug(_lib.DBG_STAGE1, self._slot(slot, cache), level, _lib.KP_DTYPE, 1 << 18)

    def keypoints(self, slot, cache=False):
        return self._debug(_lib.DBG_KEYPOINTS, self._slot(slot, cache), 0, _lib.KP_DTYPE, 2048 * 32)

    def descriptors(self, slot, cache=False):
        return self._debug(_lib.DBG_DESCRIPTORS, self._slot(slot, cache), 0, np.uint8, 2048 * 32).reshape(-1, 32)


class DNNFeatureMatcher(_Matcher):
    """LoFTR_teacher (model/LoFTR_teacher.onnx restated as HIP kernels) + threshold + decode, on the GPU.
    model_file_path: the reference's .onnx file (read directly) or an MSFLTR01 blob; None = the blob shipped with the
    package."""
    _kind = _lib.MSF_KIND_LOFTR

    def __init__(self, model_file_path=None, threshold=0.15, image_width=640, image_height=480,
                 model_resolution=16, **kw):
        if model_resolution != 16:
            raise MsfError(_lib.MSF_ERR_UNSUPPORTED, "LoFTR_teacher works at 1/16 resolution only")
        super().__init__(threshold, image_width, image_height, weights_path=model_file_path, **kw)

    def conf_matrix(self, pair=0):
        return self._debug(_lib.DBG_LOFTR_CONF, pair, 0, np.float32, 1200 * 1200 * 4).reshape(1200, 1200)

    def coarse_features(self, pair=0):
        return self._debug(_lib.DBG_LOFTR_FEAT, pair, 0, np.float32, 2 * 1200 * 32 * 4).reshape(2, 1200, 32)

    def backbone_activation(self, stage):
        """NCHW activation of the first frame of the last call after ResNet stage `stage` + 1 (MSF_FLAG_KEEP_DEBUG)"""
        shape = [(8, 240, 320), (16, 120, 160), (32, 60, 80), (32, 30, 40)][stage]
        return self._debug(_lib.DBG_LOFTR_ACT, 0, stage, np.float32, int(np.prod(shape)) * 4).reshape(shape)


class MultiDeviceMatcher:
    """One matcher over several GPUs of one process (msf_multi_*, include/msf_abi.h): a batch is cut into contiguous
    blocks of ceil(n / G) pairs, one per device, each run by that device's own handle on its own host thread; the lists
    come back in pair order, identical to one handle's (bit for bit for ORB and for LoFTR with MSF_FLAG_LOFTR_F32; LoFTR's
    default split-bf16 kernels depend on the call size and agree to ~1e-5 in confidence: include/msf_abi.h).  `devices`
    may name a device twice (two shards share the card).
    kind: "orb" (::FeatureMatcher) or "loftr" (::DNNFeatureMatcher).  The reference has no multi-GPU form; a
    multi-process job (bench.py --gpus N) uses ordinary matchers, one per rank."""

    def __init__(self, kind, threshold, image_width, image_height, devices=(0,), max_batch_pairs=1, flags=0,
                 weights_path=None):
        self._L = _lib.load()
        self._m = C.c_void_p()
        cfg = _lib.Config()
        self._L.msf_default_config(C.byref(cfg), {"orb": _lib.MSF_KIND_ORB, "loftr": _lib.MSF_KIND_LOFTR}[kind])
        cfg.threshold = threshold
        cfg.image_width = image_width
        cfg.image_height = image_height
        cfg.max_batch_pairs = max_batch_pairs
        cfg.flags = flags
        self._wpath = weights_path.encode() if weights_path else None
        cfg.weights_path = self._wpath
        ids = (C.c_int32 * len(devices))(*devices)
        rc = self._L.msf_multi_create(C.byref(cfg), len(devices), ids, C.byref(self._m))
        if rc != _lib.MSF_OK:
            msg = self._L.msf_multi_last_error(None).decode()
            self._m = C.c_void_p()
            raise MsfError(rc, msg)
        self.devices = tuple(devices)

    def close(self):
        if getattr(self, "_m", None) and self._m.value:
            self._L.msf_multi_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != _lib.MSF_OK:
            raise MsfError(rc, self._L.msf_multi_last_error(self._m).decode())

    def SetThreshold(self, value):
        self._check(self._L.msf_multi_set_threshold(self._m, float(value)))

    def shard_range(self, n_pairs, shard):
        first, count = C.c_int32(0), C.c_int32(0)
        self._L.msf_multi_shard_range(n_pairs, len(self.devices), shard, C.byref(first), C.byref(count))
        return first.value, count.value

    def match_batch(self, frames1, frames2, cap=4096):
        n = len(frames1)
        A = (_lib.Image * max(n, 1))(*[_Matcher._image(f) for f in frames1])
        B = (_lib.Image * max(n, 1))(*[_Matcher._image(f) for f in frames2])
        out = np.zeros((max(n, 1), cap), _lib.MATCH_DTYPE)
        cnt = np.zeros((max(n, 1),), np.int32)
        self._check(self._L.msf_multi_match_batch(self._m, n, A, B, out.ctypes.data, cap, cnt.ctypes.data))
        return [out[i, :min(cnt[i], cap)].view(np.int32).reshape(-1, 4).copy() for i in range(n)]

    def match_batch_device(self, d_a, d_b, d_out, d_n_out):
        """Per shard r: d_a[r] / d_b[r] uint8 CUDA tensors [n_r, H, pitch] on that shard's device, d_out[r] int32
        [n_r, cap, 4], d_n_out[r] int32 [n_r] (lists of len(devices) tensors; n_r may be 0).  Returns when every shard
        has finished."""
        G = len(self.devices)
        assert len(d_a) == len(d_b) == len(d_out) == len(d_n_out) == G
        vp = C.c_void_p
        ref = next(t for t in d_a if t.shape[0] > 0)
        for r in range(G):
            assert d_a[r].is_cuda and d_a[r].stride()[1:] == ref.stride()[1:] and d_b[r].stride() == d_a[r].stride()
        n = (C.c_int32 * G)(*[int(t.shape[0]) for t in d_a])
        pa = (vp * G)(*[t.data_ptr() for t in d_a])
        pb = (vp * G)(*[t.data_ptr() for t in d_b])
        po = (vp * G)(*[t.data_ptr() for t in d_out])
        pn = (vp * G)(*[t.data_ptr() for t in d_n_out])
        self._check(self._L.msf_multi_match_batch_device(self._m, n, pa, pb, ref.stride(0), ref.stride(1), po,
                                                         d_out[0].shape[1], pn))
