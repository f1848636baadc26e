/*
 * msf_abi.h -- C ABI of the MI355X-native feature extraction + matching stage.
 *
 * This is the drop-in boundary for the matcher hot path of
 * Kolkir/mono_slam_framework.  Every entry point names the reference interface
 * it replaces (paths relative to the reference tree):
 *
 *   slam_pipeline/include/FeatureMatcher.h:41-47   abstract FeatureMatcher::MatchFrames
 *   slam_pipeline/include/FeatureMatcher.h:15-19   MatchFramesResult {keyPoints1, keyPoints2}
 *   src/featurematcher.h:7-22,   src/featurematcher.cpp:3-47      ORB matcher
 *   src/dnnfeaturematcher.h:9-36, src/dnnfeaturematcher.cpp:11-102 LoFTR (ONNX) matcher
 *
 * Plain C: no C++ types, no exceptions, no torch types.  All functions return
 * MSF_OK (0) or a negative msf_status; msf_last_error() gives the text.  The
 * reference signals no errors (it returns an empty MatchFramesResult when
 * either descriptor set is empty, featurematcher.cpp:23); the C++ adapter
 * (mono_slam_framework_amd/csrc/hip_feature_matcher.h) maps any non-zero status
 * to an empty result.
 *
 * Threading: the reference calls MatchFrames from one thread at a time, but a
 * different std::async thread per frame (src/main.cpp:131-139).  Every entry
 * point therefore selects the handle's device itself and serialises on a
 * per-handle mutex.
 *
 * Streams: the *_device entry points enqueue on the hipStream_t they are given
 * and return without waiting.  A handle's device workspaces are shared by all
 * its calls, so ONE stream may be in flight per handle: enqueue a handle's
 * calls on one stream (or synchronise between streams).  stream = NULL means
 * the handle's own stream, which is created with hipStreamDefault flags -- it
 * is ordered against the legacy null stream in both directions, so buffers a
 * caller prepared on the null stream (torch's default stream) are seen -- and
 * which is synchronised before the call returns.  Host-pointer entry points
 * always use the handle's stream and return when the results are in host memory.
 */
#ifndef MSF_ABI_H
#define MSF_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSF_ABI_VERSION 4 /* 4: MSF_DBG_WALK_MODE, the ORB candidate pool (MSF_ERR_CAPACITY when a call exhausts it); 3: MSF_FLAG_BLUR_SUM256, msf_gather_*; 2: msf_weights_info, msf_convert_weights, msf_frame_cache_stats, flags FAST_DENSE / NO_FRAME_CACHE / LEVEL_SIZE_MUL_INV */

typedef struct msf_handle msf_handle;

typedef enum msf_status {
  MSF_OK = 0,
  MSF_ERR_INVALID_ARG = -1,
  MSF_ERR_HIP = -2,          /* a HIP runtime call failed (no GPU, OOM, launch failure) */
  MSF_ERR_UNSUPPORTED = -3,  /* e.g. LoFTR at a size other than 640x480 (the ONNX graph is fixed-shape) */
  MSF_ERR_CAPACITY = -4,     /* a fixed-capacity device list overflowed; per-pair n_out is -1 for that pair */
  MSF_ERR_IO = -5            /* weights file missing / malformed */
} msf_status;

typedef enum msf_kind {
  MSF_KIND_ORB = 0,   /* replaces ::FeatureMatcher      (src/featurematcher.cpp) */
  MSF_KIND_LOFTR = 1  /* replaces ::DNNFeatureMatcher   (src/dnnfeaturematcher.cpp) */
} msf_kind;

/* msf_config.flags */
#define MSF_FLAG_BLUR_TIE_HALF_UP 1u /* ORB 7x7 blur: (sum+32768)>>16 instead of round-half-even */
#define MSF_FLAG_PROFILE 2u          /* record per-stage HIP events on the launch stream (msf_stage_times) */
#define MSF_FLAG_KEEP_DEBUG 4u       /* LoFTR: keep pair 0's confidence matrix and coarse features for msf_debug_get */
#define MSF_FLAG_LEVEL_SIZE_MUL_INV 32u /* ORB pyramid level size cvRound(W * (1.f / scale)) instead of cvRound(W / scale) */
#define MSF_FLAG_NO_FRAME_CACHE 16u  /* msf_match_pair: extract both frames on every call (no transparent per-frame cache) */
#define MSF_FLAG_FAST_STREAM 64u     /* ORB: the output-sensitive FAST pass also for calls of fewer than 8 frames (those use the
                                        dense kernel by default: lower latency, same results) */
#define MSF_FLAG_LOFTR_F32 128u      /* LoFTR: every product on the f32 MFMA (v_mfma_f32_16x16x4_f32), bit-identical to a
                                        k-ordered fmaf chain; default: the ResNet, the attention blocks and the similarity
                                        run as bf16 MFMA products of hi/lo-split f32 operands with f32 accumulation
                                        (|conf error| ~3e-5 against a 1e-3 bar, 1.9x the throughput) */
#define MSF_FLAG_BLUR_SUM256 256u     /* ORB 7x7 blur with OpenCV's bit-exact fixed-point kernel 18 34 48 56 48 34 18 (sum 256,
                                        rounding half up) instead of the sepFilter2D integer kernel 18 34 49 55 49 34 18
                                        (sum 257) that cv::ORB reaches in OpenCV 4.x */
#define MSF_FLAG_FAST_DENSE 8u       /* ORB: score every pixel at fastThreshold (no output-sensitive first pass); same results */

typedef struct msf_config {
  uint32_t struct_size;      /* sizeof(msf_config) */
  int32_t kind;              /* msf_kind */
  int32_t device;            /* HIP device ordinal */
  float threshold;           /* ORB: Lowe ratio (featurematcher.h:9 default 0.8f, app uses 0.6f src/main.cpp:66)
                                LoFTR: confidence threshold (dnnfeaturematcher.h:11 default 0.15f) */
  int32_t image_width;       /* all frames of one handle share one size (dnnfeaturematcher.h:12-13) */
  int32_t image_height;
  int32_t max_batch_pairs;   /* P: device workspace is sized for this many pairs per call.  ORB keeps the per-CALL arrays
                                (pyramid, FAST candidate lists, stage-1 lists, walker state) once per frame of a call --
                                2 P rows of about 3.5 MB at 1280x720 -- and key points + descriptors (128 KB) per feature
                                slot, of which there are 4 P + 64 (2 P caller-visible, 2 P scratch of the stateless calls,
                                64 of the frame cache): 7.8 GB at P = 1024 (measured, tests/test_orb_gpu.py; round 4: 10.4,
                                round 3: 20).  A level's candidate list holds w h / 64 entries (what the output-sensitive
                                FAST pass lists); a level that needs more -- a dense second pass over a frame of noise --
                                takes a full-size list from a pool shared by the call, and a call that exhausts the pool
                                returns MSF_ERR_CAPACITY for the frames concerned (n_out = -1), never a short list.
                                LoFTR: about 20 MB of activations per pair of a backbone chunk (<= 256) */
  uint32_t flags;
  const char* weights_path;  /* LoFTR: the model file, as DNNFeatureMatcher's model_file_path (dnnfeaturematcher.cpp:11-21):
                                the reference's model/LoFTR_teacher.onnx is read directly (its initializers and constants);
                                an MSFLTR01 blob made by msf_convert_weights loads the same tensors faster.
                                NULL = <library dir>/weights/loftr_teacher.bin */
} msf_config;

/* FrameBase::imGray as the matchers read it (slam_pipeline/include/FrameBase.h:45):
 * 8-bit single channel, any row stride. */
typedef struct msf_image {
  const uint8_t* data;
  int32_t width, height;
  int64_t stride; /* bytes between rows */
} msf_image;

/* one element of MatchFramesResult::keyPoints1/keyPoints2 (cv::Point2i pair) */
typedef struct msf_match {
  int32_t x1, y1, x2, y2;
} msf_match;

/* one extracted ORB feature (cv::KeyPoint subset + level coordinates), 32 bytes */
typedef struct msf_keypoint {
  float x, y;       /* cv::KeyPoint::pt, level-0 coordinates */
  float response;   /* Harris response */
  float angle;      /* degrees */
  int32_t octave;
  int32_t lx, ly;   /* integer coordinates inside the pyramid level */
  int32_t fast_score;
} msf_keypoint;

int msf_abi_version(void);
void msf_default_config(msf_config* cfg, int kind);

/* The LoFTR model file on the host, no GPU involved (what Ort::Session's constructor does with the path,
 * dnnfeaturematcher.cpp:18-21): msf_weights_info reads `path` -- ONNX model or MSFLTR01 blob -- and reports the tensor
 * count, the f32 count and a digest of names, shapes and values (equal digests <=> identical weights);
 * msf_convert_weights writes them as a blob.  Errors: MSF_ERR_IO with msf_last_error(NULL). */
int msf_weights_info(const char* path, uint64_t* digest, int32_t* n_tensors, int64_t* n_floats);
int msf_convert_weights(const char* src_path, const char* dst_blob_path);

/* FeatureMatcher::FeatureMatcher(threshold) / DNNFeatureMatcher::DNNFeatureMatcher(path, threshold, w, h, res)
 * (featurematcher.cpp:3-6, dnnfeaturematcher.cpp:11-40) */
int msf_create(const msf_config* cfg, msf_handle** out);
/* ~FeatureMatcher (featurematcher.cpp:8) */
void msf_destroy(msf_handle* h);
/* FeatureMatcher::SetThreshold / DNNFeatureMatcher::SetThreshold (featurematcher.cpp:47, dnnfeaturematcher.cpp:103) */
int msf_set_threshold(msf_handle* h, float value);
const char* msf_last_error(const msf_handle* h); /* h may be NULL: error of the last failed msf_create */

/* FeatureMatcher::MatchFrames(pF1, pF2) (featurematcher.cpp:10-45, dnnfeaturematcher.cpp:44-102):
 * host images in, host match list out.  Writes min(n, cap) matches, *n_out = n.
 * The reference extracts both frames on every call, and its callers loop MatchFrames(X, KF_i) with X fixed
 * (Tracking.cc:595-632, LocalMapping.cc:176,329, KeyFrameDatabase.cc:32,64).  msf_match_pair keeps the per-frame part
 * (ORB key points + descriptors / LoFTR backbone tokens) of the last frames it saw -- 64 by default, env
 * MSF_FRAME_CACHE_SLOTS, least recently used replaced -- keyed by a 64-bit content hash and confirmed by comparing the
 * frame's bytes with a host copy, so a hash collision is a miss, never a wrong result.  Results are identical with the
 * cache on or off (MSF_FLAG_NO_FRAME_CACHE); msf_frame_cache_stats counts hits and misses (a miss = one extraction). */
int msf_match_pair(msf_handle* h, const msf_image* a, const msf_image* b,
                   msf_match* out, int32_t cap, int32_t* n_out);
int msf_frame_cache_stats(msf_handle* h, uint64_t* hits, uint64_t* misses, int32_t* capacity);
/* n_pairs independent MatchFrames calls in one launch sequence; out is [n_pairs][cap_per_pair] */
int msf_match_batch(msf_handle* h, int32_t n_pairs, const msf_image* a, const msf_image* b,
                    msf_match* out, int32_t cap_per_pair, int32_t* n_out);

/* HBM-resident batch (throughput path): d_a/d_b are DEVICE pointers to n_pairs frames each,
 * frame i at d_x + i*frame_stride, rows row_stride bytes apart (both multiples of 16, base 16-aligned).
 * d_out [n_pairs][cap_per_pair] and d_n_out [n_pairs] are device buffers.  Asynchronous on `stream`
 * (a hipStream_t; NULL = the handle's own stream, which is then synchronised before returning). */
int msf_match_batch_device(msf_handle* h, int32_t n_pairs, const uint8_t* d_a, const uint8_t* d_b,
                           int64_t frame_stride, int64_t row_stride,
                           msf_match* d_out, int32_t cap_per_pair, int32_t* d_n_out, void* stream);

/* "next" row 1 of SURVEY.md 8f: extract once per frame, match many.  The per-frame part -- ORB key points and
 * descriptors, or the LoFTR backbone tokens ([1200][32] f32; the backbone sees one image at a time, only the attention
 * blocks and the correlation see the pair) -- stays in the handle's device slots [0, 2*max_batch_pairs).  MatchFrames
 * calls on the same handle (msf_match_pair/_batch/_batch_device) work in separate slots and leave these untouched.
 * msf_match_slots_device pairs slots (DEVICE index arrays; an index outside [0, 2*max_batch_pairs) gives n_out = -1
 * for that pair); LoFTR: n_pairs <= max_batch_pairs. */
int msf_extract_device(msf_handle* h, int32_t n_frames, const uint8_t* d_frames, int64_t frame_stride,
                       int64_t row_stride, int32_t first_slot, void* stream);
int msf_match_slots_device(msf_handle* h, int32_t n_pairs, const int32_t* d_slot_a, const int32_t* d_slot_b,
                           msf_match* d_out, int32_t cap_per_pair, int32_t* d_n_out, void* stream);

/* "next" row 3 of SURVEY.md 8f: KeyFrameMatchDatabase scoring (slam_pipeline/src/KeyFrameDatabase.cc:23-53).
 * msf_set_mappoints replaces the content of map slot [0, 2*max_batch_pairs) with a frame's KeyPointMap occupancy:
 * `keys` (HOST pointer) are the pixel keys y*cols + x that hold a map point (KeyPointMap.cc:36-52); keys outside the
 * image are ignored, as KeyPointMap::SetMapPoint ignores such points (KeyPointMap.cc:38-39).
 * msf_count_mappoint_matches_device: d_num_mp[i] = number of the first min(d_n_matches[i], cap_per_pair) matches of
 * pair i whose endpoint 1 is set in map slot d_map_a[i] and endpoint 2 in map slot d_map_b[i]
 * (MatchFramesResult::GetMapPoint1/2 both non-null, FeatureMatcher.h:23-29, KeyFrameDatabase.cc:37-44). */
int msf_set_mappoints(msf_handle* h, int32_t map_slot, const int32_t* keys, int32_t n_keys);
int msf_count_mappoint_matches_device(msf_handle* h, int32_t n_pairs, const msf_match* d_matches,
                                      int32_t cap_per_pair, const int32_t* d_n_matches, const int32_t* d_map_a,
                                      const int32_t* d_map_b, int32_t* d_num_mp, void* stream);

/* One-vs-many host entry points for KeyFrameMatchDatabase (KeyFrameDatabase.cc:31-32, 63-64: a loop of
 * MatchFrames(X, KF_i) with X fixed).  msf_store_frame uploads a host frame into resident frame slot
 * [0, 2*max_batch_pairs) and, for ORB, extracts its features once into the feature slot of the same index.
 * msf_match_one_to_many matches the frame in `query_slot` against the n <= max_batch_pairs frames in `slots`
 * (HOST array) in one launch sequence: num_matches[i] (HOST) as msf_match_batch's n_out; num_mp[i] (HOST, optional)
 * = matches with a map point at both endpoints, map slots being the frame slots; out (HOST, optional)
 * [n][cap_per_pair] receives the lists. */
int msf_store_frame(msf_handle* h, int32_t slot, const msf_image* img);
int msf_match_one_to_many(msf_handle* h, int32_t query_slot, int32_t n, const int32_t* slots, int32_t* num_matches,
                          int32_t* num_mp, msf_match* out, int32_t cap_per_pair);

/* "next" row 4 of SURVEY.md 8f: Initializer::CheckHomography / CheckFundamental (slam_pipeline/src/Initializer.cc:
 * 322-405, 407-487) for all n_hyp RANSAC hypotheses of FindHomography / FindFundamental (:152-199, :201-245) at once.
 * m21: [n_hyp][9] row-major H21 (or F21); m12: [n_hyp][9] H12 = H21^-1 (homography only, else NULL); matches: the
 * MatchFramesResult the Initializer was built from (mvKeys1/2, Initializer.cc:79-86); all HOST pointers.
 * scores[i] is the reference's f32 score of hypothesis i bit for bit; *best = the hypothesis the reference loop keeps
 * (first strict maximum above 0, -1 if none) and best_inliers[n_matches] its vbMatchesInliers (all 0 if none). */
#define MSF_MODEL_HOMOGRAPHY 0
#define MSF_MODEL_FUNDAMENTAL 1
int msf_check_hypotheses(msf_handle* h, int32_t model, int32_t n_hyp, const float* m21, const float* m12,
                         int32_t n_matches, const msf_match* matches, float sigma, float* scores, int32_t* best,
                         uint8_t* best_inliers);

/* Tracking::CreateCurrentMatchImage (slam_pipeline/src/Tracking.cc:899-940): out_rgb [H][2*W][3] (rows out_stride
 * bytes apart, HOST) = the two gray frames side by side as RGB with a filled radius-3 circle on every match end point:
 * (0,255,0) where neither side has a map point, then (255,0,0) over those where either side has one.
 * has_mp1 / has_mp2: [n_matches] bytes (GetMapPoint1/2 != null), NULL = none. */
int msf_render_match_image(msf_handle* h, const msf_image* f1, const msf_image* f2, const msf_match* matches,
                           int32_t n_matches, const uint8_t* has_mp1, const uint8_t* has_mp2, uint8_t* out_rgb,
                           int64_t out_stride);

/* Packs [n_pairs][cap_per_pair] match lists + counts into one contiguous device list:
 * d_offsets[i] = start of pair i, d_offsets[n_pairs] = total; pairs with n_out < 0 contribute nothing.
 * This is the payload of the multi-GPU gather of match lists (and of MatchFramesResult's vectors). */
int msf_pack_matches_device(msf_handle* h, int32_t n_pairs, const msf_match* d_in, int32_t cap_per_pair,
                            const int32_t* d_n_out, msf_match* d_packed, int32_t* d_offsets, void* stream);

/* -------- multi-device sharding of the batched call --------
 * The reference owns one matcher and calls it from one thread (src/main.cpp:65,78-82; System.cc:63-75): it has no
 * multi-GPU form.  Pairs are independent units, so a batch shards with no exchange step: msf_multi holds one msf_handle
 * per entry of device_ids (NULL = devices 0 .. n_devices-1; an id may repeat: two shards then share a card) and
 * msf_multi_match_batch gives shard r the contiguous block of ceil(n_pairs / G) pairs msf_multi_shard_range reports --
 * the partition of bench.py's ranks -- on one host thread per shard; every shard writes its block of out / n_out, which
 * therefore come back in pair order, exactly as from msf_match_batch on one handle with the same cfg -- bit for bit for
 * ORB, and for LoFTR under MSF_FLAG_LOFTR_F32.  On LoFTR's default split-bf16 path the kernel family follows the size of
 * a call (streaming ResNet kernels from 64 images, the three-tile similarity pass from 8 pairs), and the families agree
 * to ~1e-5 in confidence (bar 1e-3): a shard of another size than the whole batch can gain or lose a match whose
 * confidence lies that close to the threshold (tests/test_multi_device.py::test_loftr_lists_do_not_depend_on_...).
 * Returns MSF_OK, MSF_ERR_CAPACITY if that is the only failure of any shard, else the first hard error
 * (msf_multi_last_error names the shard).  msf_multi_handle: the shard's own handle, for callers that keep frames
 * resident per device and drive the *_device entry points themselves (one host thread per device); owned by the
 * msf_multi.  A multi-PROCESS job (one rank per GPU, as bench.py --gpus N) needs none of this: each rank creates an
 * ordinary handle and the match lists are gathered with msf_pack_matches_device + send/recv. */
typedef struct msf_multi msf_multi;
int msf_multi_create(const msf_config* cfg, int32_t n_devices, const int32_t* device_ids, msf_multi** out);
void msf_multi_destroy(msf_multi* m);
int32_t msf_multi_device_count(const msf_multi* m);
msf_handle* msf_multi_handle(msf_multi* m, int32_t shard);
int msf_multi_set_threshold(msf_multi* m, float value);
const char* msf_multi_last_error(const msf_multi* m); /* m may be NULL: error of the last failed msf_multi_create */
void msf_multi_shard_range(int32_t n_pairs, int32_t n_shards, int32_t shard, int32_t* first, int32_t* count);
int msf_multi_match_batch(msf_multi* m, int32_t n_pairs, const msf_image* a, const msf_image* b, msf_match* out,
                          int32_t cap_per_pair, int32_t* n_out);
/* HBM-resident form (the throughput path): shard r's n_pairs[r] pairs live on ITS device at d_a[r] / d_b[r] (layout
 * and alignment as msf_match_batch_device) and its results go to its device buffers d_out[r] [n_pairs[r]][cap_per_pair]
 * and d_n_out[r]; all arrays have msf_multi_device_count(m) entries.  Every shard runs on its handle's own stream and
 * the call returns when all have finished. */
int msf_multi_match_batch_device(msf_multi* m, const int32_t* n_pairs, const uint8_t* const* d_a,
                                 const uint8_t* const* d_b, int64_t frame_stride, int64_t row_stride,
                                 msf_match* const* d_out, int32_t cap_per_pair, int32_t* const* d_n_out);

/* -------- multi-process gather of the match lists --------
 * One process per GPU; rank r owns the contiguous block of pairs msf_multi_shard_range reports and computes it with no
 * data-path collective.  The one exchange step is the gather of the packed lists (msf_pack_matches_device) to rank 0:
 * an ncclAllGather of the per-pair offsets (pairs_per_rank + 1 int32 per rank), then exact-size ncclSend / ncclRecv of
 * the int32[4] records in one group call -- RCCL over xGMI, never an all-reduce.  The reference has no counterpart (one
 * matcher, one thread: src/main.cpp:65,78-82).  RCCL is bound at first use (dlopen), so libmsf.so does not depend on it.
 *   msf_gather_unique_id   rank 0: 128 bytes to hand to every rank through the job's own channel (ncclGetUniqueId)
 *   msf_gather_create      every rank, collectively (ncclCommInitRank on `device`)
 *   msf_gather_matches_device  every rank, collectively, asynchronous on `stream` except for ONE wait (the totals are
 *       needed on the host to size the transfers): d_packed / d_offsets as msf_pack_matches_device wrote them;
 *       d_all_offsets [n_ranks][pairs_per_rank + 1] (device, every rank) receives every rank's offsets; totals
 *       [n_ranks] (HOST) the record counts; on rank 0 d_recv [n_ranks * cap_records] receives the lists densely packed
 *       in rank order (pair p of rank r starts at sum(totals[0..r)) + d_all_offsets[r][p]); other ranks pass NULL.
 *       MSF_ERR_CAPACITY if a rank holds more than cap_records records (nothing is transferred then).
 *   msf_gather_plan        the placement arithmetic on its own (host only; unit-tested without a GPU).
 * Failure semantics of msf_gather_matches_device (a collective: every rank calls it the same number of times).
 *   MSF_ERR_CAPACITY / MSF_ERR_INVALID_ARG from the placement step are computed by every rank from the same gathered
 *   offsets: ALL ranks return the same code, none enters the send / recv leg, the object stays usable.  Any other error
 *   (a failing RCCL or HIP call on this rank) is HARD: the rank aborts its communicator (ncclCommAbort, where the library
 *   exports it) so that peers blocked in their send / recv come back with an error instead of waiting for ever, every
 *   later call on the object returns MSF_ERR_HIP, and the only valid operation left is msf_gather_destroy -- on every
 *   rank, followed by a new msf_gather_unique_id / msf_gather_create round if the job goes on.  EXPERIMENTAL beyond one
 *   rank: see the next line.
 * Missing RCCL (no librccl.so to dlopen; MSF_RCCL_LIBRARY overrides the name) gives MSF_ERR_HIP from
 * msf_gather_unique_id / msf_gather_create with the loader's message in msf_gather_last_error(NULL); nothing else in
 * libmsf.so needs RCCL.
 * Executed so far: on one MI355X with a one-rank communicator; the send / recv leg has not run on more than one GPU. */
typedef struct msf_gather msf_gather;
int msf_gather_unique_id(uint8_t* id128);
int msf_gather_create(int32_t device, int32_t rank, int32_t n_ranks, const uint8_t* id128, int32_t pairs_per_rank,
                      int64_t cap_records, msf_gather** out);
void msf_gather_destroy(msf_gather* g);
const char* msf_gather_last_error(const msf_gather* g); /* g may be NULL: error of the last failed create / unique_id */
int msf_gather_plan(int32_t n_ranks, int32_t pairs_per_rank, const int32_t* all_offsets, int64_t cap_records,
                    int32_t* totals, int64_t* recv_first);
int msf_gather_matches_device(msf_gather* g, const msf_match* d_packed, const int32_t* d_offsets, int32_t* d_all_offsets,
                              msf_match* d_recv, int32_t* totals, void* stream);

/* -------- Walker stall: a caller-visible failure mode of ORB batch extraction --------
 * Calls of >= 8 frames run pyramid + FAST of all eight levels as ONE launch whose one-wave units wait for units of lower
 * workgroup index (orb_kernels.hip, k_walk).  That is live as long as the hardware starts workgroups of a grid in index
 * order -- observed on gfx950, not promised by HIP -- so every wait is bounded (2^19 polls, about one second).  A unit
 * that gives up flags its frame: the pairs of that frame come back with n_out = -1 and the call returns MSF_ERR_CAPACITY
 * like any other per-pair failure (never a silent wrong list); every other waiter of the launch leaves at once, the grid
 * drains.  The stall is counted in a device word that is never cleared; the next extraction that finds it non-zero
 * switches the handle to one launch per level for the rest of its life (no in-launch waits; about 9 % slower at
 * 1280x720), and msf_last_error carries a note once.  MSF_DBG_WALK_MODE reports both.  Two handles extracting at the same
 * time on one device (msf_multi shards, per-size handles, a gather stream) do not stall each other: every grid's own
 * units still start in index order (tests/test_orb_gpu.py::test_two_walker_grids_at_once). */

/* -------- introspection used by the parity tests and bench.py (not by the drop-in path) -------- */
typedef enum msf_debug_what {
  MSF_DBG_LEVEL_SIZES = 0,   /* int32 [nlevels][4] = w, h, row pitch, quota */
  MSF_DBG_LEVEL_PIXELS = 1,  /* uint8 [h][pitch] of (slot, level); level 0 is the input frame itself (not kept) */
  MSF_DBG_FAST_CANDS = 2,    /* int32 [n][3] (x, y, score) of (slot, level), unordered: the strict 3x3 maxima with
                                score >= the level's MSF_DBG_FAST_TAU (all FAST corners after NMS when that is 20) */
  MSF_DBG_KEYPOINTS = 3,     /* msf_keypoint [n] of slot */
  MSF_DBG_DESCRIPTORS = 4,   /* uint8 [n][32] of slot */
  MSF_DBG_STAGE1 = 5,        /* msf_keypoint [n] (lx, ly, octave, fast_score, response) of (slot, level), unordered */
  MSF_DBG_LOFTR_CONF = 6,    /* float [1200][1200] confidence matrix of pair `slot` (debug launch only) */
  MSF_DBG_LOFTR_FEAT = 7,    /* float [2][1200][32] coarse features after the transformer of pair `slot` */
  MSF_DBG_FAST_TAU = 8,      /* int32 [nlevels][2] of slot: FAST score threshold the candidate list was built with
                                (20 = dense, also after a failed check), and the first estimate */
  MSF_DBG_WALK_MODE = 10,    /* ORB, int32 [2]: {1 if the handle launches the pyramid + FAST walker level by level (after a
                                stalled one-launch walker, or MSF_ORB_WALK_PER_LEVEL=1), else 0; units of one-launch
                                walkers that gave up a bounded wait since msf_create} -- see "Walker stall" below */
  MSF_DBG_LOFTR_ACT = 9      /* float NCHW activation of the first frame of the last backbone pass after ResNet stage
                                `level` + 1: [8][240][320], [16][120][160], [32][60][80], [32][30][40] (level 0..3) */
} msf_debug_what;
/* copies to host; *n_bytes = bytes available (may exceed cap_bytes, then only cap_bytes are written) */
int msf_debug_get(msf_handle* h, int32_t what, int32_t slot, int32_t level,
                  void* host_out, size_t cap_bytes, size_t* n_bytes);

/* per-stage device time, measured with HIP events on the launch stream (needs MSF_FLAG_PROFILE): the SUM over the batch
 * calls since the previous query -- queried after every call it is that call's times; a caller that enqueues many calls
 * ahead of the device (nothing in the recording waits) queries once afterwards and divides by its call count.  The events
 * live in a ring of 32 sets; a set needed again before it was queried is folded into the sum first.  Waits for the calls
 * it reports.  names[i] are static strings.  Returns the number of stages. */
int msf_stage_times(msf_handle* h, const char** names, float* ms, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* MSF_ABI_H */
