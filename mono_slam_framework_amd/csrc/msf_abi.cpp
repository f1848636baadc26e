// extern "C" boundary of libmsf.so (declarations + the reference interfaces they replace: include/msf_abi.h).
#include "msf_abi.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "loftr_pipeline.h"
#include "orb_pipeline.h"
#include "weights_io.h"

namespace msf {
hipError_t pack_matches(int n, const msf_match* d_in, int cap, const int32_t* d_cnt, msf_match* d_packed,
                        int32_t* d_offsets, hipStream_t st);
hipError_t count_mappoint_matches(int n, const msf_match* d_matches, int cap, const int32_t* d_cnt,
                                  const int32_t* d_map_a, const int32_t* d_map_b, const uint32_t* d_maps, int n_maps,
                                  int map_words, int width, int height, int32_t* d_num_mp, hipStream_t st);
hipError_t render_match_image(const uint8_t* d_f1, const uint8_t* d_f2, int w, int h, long long pitch,
                              const msf_match* d_m, const uint8_t* d_mp1, const uint8_t* d_mp2, int n, uint8_t* d_out,
                              long long out_stride, hipStream_t st);
hipError_t check_hypotheses(int model, int n_hyp, const float* d_m21, const float* d_m12, int n,
                            const msf_match* d_matches, float sigma, float* d_scores, uint8_t* d_inliers,
                            hipStream_t st);
}

namespace {
thread_local std::string g_create_error;
}

struct msf_handle {
  msf_config cfg{};
  std::mutex mu;
  std::string err;
  hipStream_t stream = nullptr;
  msf::OrbPipeline orb;
  msf::LoftrPipeline loftr;
  // staging for the host-image entry points
  uint8_t* d_stage = nullptr;   // [2 * max_pairs][H][pitch]
  msf_match* d_out_base = nullptr;
  msf_match* d_out = nullptr;   // [max_pairs][stage_cap] = d_out_base + 1
  msf_match* h_pin = nullptr;   // pinned: [1 + kPinMatches]
  int32_t* d_n = nullptr;       // [max_pairs]
  int stage_pitch = 0;
  long long stage_frame = 0;
  int stage_cap = 0;
  // KeyPointMap occupancy bitmaps (allocated by the first msf_set_mappoints): [n_maps][map_words]
  uint32_t* d_maps = nullptr;
  int n_maps = 0, map_words = 0;
  std::vector<uint32_t> map_stage;
  // resident frame store of the one-vs-many entry points (allocated by the first msf_store_frame): [2*max_pairs] frames
  uint8_t* d_store = nullptr;
  int32_t* d_idx = nullptr;     // [3][max_pairs]: query slot per pair, train slot per pair, map-point counts
  std::vector<int32_t> idx_stage;
  // msf_check_hypotheses workspace, grown on demand
  float* d_hyp = nullptr;        // [2][hyp_cap][9] + [hyp_cap] scores
  uint8_t* d_hyp_inl = nullptr;  // [hyp_cap * hyp_match_cap]
  msf_match* d_hyp_m = nullptr;  // [hyp_match_cap]
  int hyp_cap = 0, hyp_match_cap = 0;
  // msf_render_match_image workspace: the RGB image (allocated once) and the match list + flags (grown on demand)
  uint8_t* d_render = nullptr;      // [H][2 * W][3]
  msf_match* d_render_m = nullptr;  // [render_cap] matches, then 2 * render_cap flag bytes
  int render_cap = 0;
  // Transparent per-frame cache of the drop-in MatchFrames call: the callers loop
  // MatchFrames(X, KF_i) with X fixed (Tracking.cc:595-632, LocalMapping.cc:176,329, KeyFrameDatabase.cc:32,64) and the
  // reference re-extracts both frames every time.  Key = 64-bit content hash of the frame, confirmed by comparing the
  // bytes with the host copy kept per entry (a collision is a miss, never a wrong answer); value = a feature slot (ORB)
  // or token slot (LoFTR) beyond the caller-visible ones; least recently used entry is replaced.  FrameBase::id() is no
  // key: Frame and KeyFrame count separately (Frame.cc:29, KeyFrame.cc:30).
  struct CacheEntry {
    uint64_t hash = 0, used = 0;
    bool valid = false;
    std::vector<uint8_t> bytes;   // [H][W] contiguous
  };
  std::vector<CacheEntry> fc;
  int fc_slot0 = 0;                // first slot of the cache range
  int32_t* d_fc_slots = nullptr;   // [fc.size()] = fc_slot0 + i: one-element slot arrays for the match call
  uint64_t fc_tick = 0, fc_hits = 0, fc_misses = 0, fc_hash_mask = ~0ull;
};

namespace {

int fail(msf_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg; else g_create_error = msg;
  return code;
}

int hip_fail(msf_handle* h, const char* what, hipError_t e) {
  return fail(h, MSF_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

// No exception crosses the C ABI (msf_abi.h): every entry point that can allocate runs inside try / catch (...) and
// reports a host exception (std::bad_alloc from a std::vector / std::string, ...) as a status.  The message is
// assigned without allocating anything large; if even that throws, the status alone is returned.
int host_exception(msf_handle* h, const char* where) noexcept {
  try {
    fail(h, MSF_ERR_HIP, std::string(where) + ": host exception (out of memory?)");
  } catch (...) {
  }
  return MSF_ERR_HIP;
}

constexpr int kFrameCacheSlots = 64;   // default capacity of the transparent frame cache (MSF_FRAME_CACHE_SLOTS overrides)
constexpr int kPinMatches = 1024;  // matches fetched together with the count by the single-pair call
constexpr int kStageCap = 4096;  // matches per pair kept by the host-image path (ORB <= n1 <= 2048; LoFTR: see below)

int run_device(msf_handle* h, int n_pairs, const uint8_t* d_a, const uint8_t* d_b, long long frame_stride,
               long long row_stride, msf_match* d_out, int cap, int32_t* d_n_out, hipStream_t st) {
  if (n_pairs <= 0) return MSF_OK;
  if (n_pairs > h->cfg.max_batch_pairs) return fail(h, MSF_ERR_INVALID_ARG, "n_pairs exceeds max_batch_pairs");
  if (((uintptr_t)d_a | (uintptr_t)d_b | (uintptr_t)frame_stride | (uintptr_t)row_stride) & 15)
    return fail(h, MSF_ERR_INVALID_ARG, "device frames must be 16-byte aligned with strides multiple of 16");
  if (row_stride < h->cfg.image_width) return fail(h, MSF_ERR_INVALID_ARG, "row_stride < image_width");
  if (frame_stride < row_stride * (long long)h->cfg.image_height)
    return fail(h, MSF_ERR_INVALID_ARG, "frame_stride < row_stride * image_height (frames would overlap)");
  if (h->cfg.kind == MSF_KIND_ORB) {
    // the stateless MatchFrames path works in its own feature slots [2P, 4P): slots [0, 2P) are the per-frame cache of
    // msf_extract_device / msf_store_frame, which a MatchFrames call on the same handle must not disturb
    // (KeyFrameMatchDatabase and Tracking share one matcher, src/main.cpp:77-81)
    const int scratch0 = 2 * h->cfg.max_batch_pairs;
    msf::FrameSrc src{d_a, d_b, n_pairs, scratch0, frame_stride, (int)row_stride};
    hipError_t e = h->orb.extract(src, 2 * n_pairs, st);
    if (e != hipSuccess) return hip_fail(h, "orb extract", e);
    if (h->orb.take_degraded_note())    // not an error of THIS call: the text is there for whoever reads msf_last_error
      h->err = "note: a unit of an earlier ORB walker launch gave up a bounded wait (its pairs reported n_out = -1); "
               "this handle now launches the walker one level at a time";
    e = h->orb.match(n_pairs, nullptr, nullptr, h->cfg.threshold, d_out, cap, d_n_out, st, scratch0);
    if (e != hipSuccess) return hip_fail(h, "orb match", e);
    return MSF_OK;
  }
  hipError_t e = h->loftr.match(n_pairs, d_a, d_b, frame_stride, (int)row_stride, h->cfg.threshold, d_out, cap,
                                d_n_out, st);
  if (e != hipSuccess) return hip_fail(h, "loftr match", e);
  return MSF_OK;
}

int ensure_stage(msf_handle* h) {
  if (h->d_stage) return MSF_OK;
  const int W = h->cfg.image_width, H = h->cfg.image_height, maxp = h->cfg.max_batch_pairs;
  hipError_t e;
  h->stage_pitch = (W + 15) & ~15;
  h->stage_frame = (long long)h->stage_pitch * H;
  h->stage_cap = kStageCap;
  if ((e = hipMalloc(&h->d_stage, (size_t)2 * maxp * h->stage_frame)) != hipSuccess) return hip_fail(h, "hipMalloc stage", e);
  // one leading record in front of the lists: the single-pair call puts its count there, so count + list come back
  // in ONE device-to-host copy into pinned memory (one synchronisation per MatchFrames call instead of two)
  if ((e = hipMalloc(&h->d_out_base, ((size_t)maxp * h->stage_cap + 1) * sizeof(msf_match))) != hipSuccess) return hip_fail(h, "hipMalloc out", e);
  h->d_out = h->d_out_base + 1;
  if ((e = hipHostMalloc(&h->h_pin, (size_t)(kPinMatches + 1) * sizeof(msf_match), hipHostMallocDefault)) != hipSuccess) return hip_fail(h, "hipHostMalloc", e);
  if ((e = hipMalloc(&h->d_n, (size_t)maxp * sizeof(int32_t))) != hipSuccess) return hip_fail(h, "hipMalloc n", e);
  return MSF_OK;
}

int ensure_maps(msf_handle* h) {
  if (h->d_maps) return MSF_OK;
  const int n_maps = 2 * h->cfg.max_batch_pairs;
  const long long n_px = (long long)h->cfg.image_width * h->cfg.image_height;
  h->map_words = (int)((n_px + 31) / 32);
  const size_t bytes = (size_t)n_maps * h->map_words * sizeof(uint32_t);
  hipError_t e;
  if ((e = hipMalloc(&h->d_maps, bytes)) != hipSuccess) return hip_fail(h, "hipMalloc(map bitmaps)", e);
  if ((e = hipMemsetAsync(h->d_maps, 0, bytes, h->stream)) != hipSuccess) return hip_fail(h, "hipMemsetAsync", e);
  h->n_maps = n_maps;
  return MSF_OK;
}

// 64-bit content hash of a W x H frame with row stride: four independent multiply-xorshift lanes over 8-byte words
// (one dependent chain would run at a quarter of the speed), folded at the end.
uint64_t hash_image(const msf_image* im) {
  const uint64_t K0 = 0x9E3779B97F4A7C15ull, K1 = 0xC2B2AE3D27D4EB4Full, K2 = 0x165667B19E3779F9ull, K3 = 0xD6E8FEB86659FD93ull;
  uint64_t h0 = K0 ^ (uint64_t)im->width, h1 = K1 ^ (uint64_t)im->height, h2 = K2, h3 = K3;
  const int W = im->width;
  for (int y = 0; y < im->height; y++) {
    const uint8_t* p = im->data + (size_t)y * (size_t)im->stride;
    int x = 0;
    for (; x + 32 <= W; x += 32) {
      uint64_t a, b, c, d;
      std::memcpy(&a, p + x, 8); std::memcpy(&b, p + x + 8, 8); std::memcpy(&c, p + x + 16, 8); std::memcpy(&d, p + x + 24, 8);
      h0 = (h0 ^ a) * K1; h0 ^= h0 >> 29;
      h1 = (h1 ^ b) * K2; h1 ^= h1 >> 31;
      h2 = (h2 ^ c) * K3; h2 ^= h2 >> 30;
      h3 = (h3 ^ d) * K0; h3 ^= h3 >> 28;
    }
    uint64_t tail = 0;
    for (; x < W; x++) tail = (tail << 8) ^ p[x] ^ (tail >> 56);
    h0 = (h0 ^ tail ^ (uint64_t)y) * K2; h0 ^= h0 >> 32;
  }
  uint64_t h = h0 ^ (h1 * K3) ^ (h2 * K0) ^ (h3 * K1);
  h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull; h ^= h >> 33;
  return h;
}

bool same_bytes(const msf_image* im, const std::vector<uint8_t>& bytes) {
  const size_t W = (size_t)im->width;
  for (int y = 0; y < im->height; y++)
    if (std::memcmp(im->data + (size_t)y * (size_t)im->stride, bytes.data() + (size_t)y * W, W) != 0) return false;
  return true;
}

int fetch_single(msf_handle* h, msf_match* out, int32_t cap_per_pair, int32_t* n_out, hipStream_t st);

// Slot of `im` in the frame cache, extracting it first if it is not there.  `keep` = an entry that must not be evicted
// (the other frame of the same call), or -1.  `d_stage_frame` = where to upload the frame for an extraction.
int cached_slot(msf_handle* h, const msf_image* im, int keep, uint8_t* d_stage_frame, hipStream_t st, int* entry_out) {
  const int W = h->cfg.image_width, H = h->cfg.image_height;
  const uint64_t hv = hash_image(im) & h->fc_hash_mask;
  int victim = -1;   // an unused entry, else the least recently used one; never `keep`
  for (int i = 0; i < (int)h->fc.size(); i++) {
    msf_handle::CacheEntry& e = h->fc[i];
    if (e.valid && e.hash == hv && same_bytes(im, e.bytes)) {
      e.used = ++h->fc_tick;
      h->fc_hits++;
      *entry_out = i;
      return MSF_OK;
    }
    if (i == keep) continue;
    const uint64_t age = e.valid ? e.used : 0;            // invalid entries first
    if (victim < 0 || age < (h->fc[victim].valid ? h->fc[victim].used : 0)) victim = i;
  }
  if (victim < 0) return fail(h, MSF_ERR_INVALID_ARG, "frame cache has no replaceable entry");
  msf_handle::CacheEntry& e = h->fc[victim];
  e.valid = false;
  e.bytes.resize((size_t)W * H);
  for (int y = 0; y < H; y++) std::memcpy(e.bytes.data() + (size_t)y * W, im->data + (size_t)y * (size_t)im->stride, (size_t)W);
  hipError_t err = hipMemcpy2DAsync(d_stage_frame, h->stage_pitch, e.bytes.data(), W, W, H, hipMemcpyHostToDevice, st);
  if (err != hipSuccess) return hip_fail(h, "hipMemcpy2DAsync", err);
  const int slot = h->fc_slot0 + victim;
  if (h->cfg.kind == MSF_KIND_ORB) {
    msf::FrameSrc src{d_stage_frame, d_stage_frame, 1, slot, h->stage_frame, h->stage_pitch};
    if ((err = h->orb.extract(src, 1, st)) != hipSuccess) return hip_fail(h, "orb extract", err);
  } else if ((err = h->loftr.extract(1, d_stage_frame, h->stage_frame, h->stage_pitch, slot, st)) != hipSuccess) {
    return hip_fail(h, "loftr extract", err);
  }
  e.hash = hv;
  e.used = ++h->fc_tick;
  e.valid = true;
  h->fc_misses++;
  *entry_out = victim;
  return MSF_OK;
}

// MatchFrames(a, b) through the frame cache: per frame a hash + byte compare on the host; only frames not seen lately
// are uploaded and extracted; then one slot-pair match.  Same lists as the stateless path (tests/test_frame_cache_gpu.py).
int match_pair_cached(msf_handle* h, const msf_image* a, const msf_image* b, msf_match* out, int32_t cap, int32_t* n_out) {
  hipStream_t st = h->stream;
  uint8_t* dA = h->d_stage;
  uint8_t* dB = h->d_stage + (size_t)h->cfg.max_batch_pairs * h->stage_frame;
  int ea = -1, eb = -1;
  if (int rc = cached_slot(h, a, -1, dA, st, &ea)) return rc;
  if (int rc = cached_slot(h, b, ea, dB, st, &eb)) {
    return rc;
  }
  hipError_t e =
      h->cfg.kind == MSF_KIND_ORB
          ? h->orb.match(1, h->d_fc_slots + ea, h->d_fc_slots + eb, h->cfg.threshold, h->d_out, h->stage_cap,
                         reinterpret_cast<int32_t*>(h->d_out_base), st)
          : h->loftr.match_slots(1, h->d_fc_slots + ea, h->d_fc_slots + eb, h->cfg.threshold, h->d_out, h->stage_cap,
                                 reinterpret_cast<int32_t*>(h->d_out_base), st);
  if (e != hipSuccess) {
    // the extractions of this call may not have completed: forget both entries
    h->fc[ea].valid = false;
    h->fc[eb].valid = false;
    return hip_fail(h, "match slots", e);
  }
  const int rc = fetch_single(h, out, cap, n_out, st);
  if (rc != MSF_OK) {   // a failed (or overflowed) frame must not be served from the cache again
    h->fc[ea].valid = false;
    h->fc[eb].valid = false;
  }
  return rc;
}

// count (in the record before the list) + list of the single-pair call: one copy, one synchronisation
int fetch_single(msf_handle* h, msf_match* out, int32_t cap_per_pair, int32_t* n_out, hipStream_t st) {
  hipError_t e;
  int wmax = cap_per_pair < h->stage_cap ? cap_per_pair : h->stage_cap;
  const int wfirst = wmax < kPinMatches ? wmax : kPinMatches;
  if ((e = hipMemcpyAsync(h->h_pin, h->d_out_base, (size_t)(1 + wfirst) * sizeof(msf_match), hipMemcpyDeviceToHost, st)) != hipSuccess)
    return hip_fail(h, "hipMemcpyAsync", e);
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(h, "hipStreamSynchronize", e);
  const int32_t c = *reinterpret_cast<const int32_t*>(h->h_pin);
  n_out[0] = c;
  if (c < 0) return fail(h, MSF_ERR_CAPACITY, "at least one pair has no valid result (n_out = -1): a fixed-capacity device list overflowed, or a unit of the ORB walker launch gave up a bounded wait");
  const int avail = c < h->stage_cap ? c : h->stage_cap;
  const int w = avail < cap_per_pair ? avail : cap_per_pair;
  const int w1 = w < wfirst ? w : wfirst;
  if (w1 > 0) std::memcpy(out, h->h_pin + 1, (size_t)w1 * sizeof(msf_match));
  if (w > w1 && (e = hipMemcpy(out + w1, h->d_out + w1, (size_t)(w - w1) * sizeof(msf_match), hipMemcpyDeviceToHost)) != hipSuccess)
    return hip_fail(h, "hipMemcpy", e);
  if (avail < c && cap_per_pair > avail) return fail(h, MSF_ERR_CAPACITY, "at least one pair has no valid result (n_out = -1): a fixed-capacity device list overflowed, or a unit of the ORB walker launch gave up a bounded wait");
  return MSF_OK;
}

}  // namespace

extern "C" {

int msf_abi_version(void) { return MSF_ABI_VERSION; }

void msf_default_config(msf_config* cfg, int kind) {
  if (!cfg) return;
  *cfg = msf_config{};
  cfg->struct_size = sizeof(msf_config);
  cfg->kind = kind;
  cfg->device = 0;
  cfg->threshold = kind == MSF_KIND_LOFTR ? 0.15f : 0.8f;  // dnnfeaturematcher.h:11 / featurematcher.h:9
  cfg->image_width = 640;                                  // dnnfeaturematcher.h:12-13
  cfg->image_height = 480;
  cfg->max_batch_pairs = 1;
  cfg->flags = 0;
  cfg->weights_path = nullptr;
}

int msf_create(const msf_config* cfg, msf_handle** out) {
  try {
    if (!cfg || !out) return fail(nullptr, MSF_ERR_INVALID_ARG, "msf_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(msf_config)) return fail(nullptr, MSF_ERR_INVALID_ARG, "msf_create: struct_size mismatch");
    if (cfg->kind != MSF_KIND_ORB && cfg->kind != MSF_KIND_LOFTR) return fail(nullptr, MSF_ERR_INVALID_ARG, "msf_create: bad kind");
    if (cfg->max_batch_pairs < 1) return fail(nullptr, MSF_ERR_INVALID_ARG, "msf_create: max_batch_pairs < 1");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(nullptr, MSF_ERR_HIP, "msf_create: no HIP device (this library has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, MSF_ERR_INVALID_ARG, "msf_create: device ordinal out of range");
    if ((e = hipSetDevice(cfg->device)) != hipSuccess) return hip_fail(nullptr, "hipSetDevice", e);
    msf_handle* h = new (std::nothrow) msf_handle();
    if (!h) return fail(nullptr, MSF_ERR_HIP, "out of host memory");
    struct Guard {   // whatever path leaves this function without success (an exception included) frees the handle
      msf_handle* h;
      ~Guard() { if (h) msf_destroy(h); }
    } guard{h};
    h->cfg = *cfg;
    h->cfg.weights_path = nullptr;
    std::string err;
    const bool profile = (cfg->flags & MSF_FLAG_PROFILE) != 0;
    int n_cache = kFrameCacheSlots;
    if (const char* ev = getenv("MSF_FRAME_CACHE_SLOTS")) n_cache = atoi(ev);
    if (cfg->flags & MSF_FLAG_NO_FRAME_CACHE) n_cache = 0;          // the flag wins over the environment
    n_cache = n_cache < 2 ? 0 : n_cache > 4096 ? 4096 : n_cache;   // a pair needs two entries
    if (const char* ev = getenv("MSF_FRAME_CACHE_HASH_BITS")) {    // tests: a few bits make hash collisions the rule
      const int bits = atoi(ev);
      if (bits >= 0 && bits < 64) h->fc_hash_mask = (1ull << bits) - 1ull;
    }
    if (cfg->kind == MSF_KIND_ORB) {
      h->fc_slot0 = 4 * cfg->max_batch_pairs;
      err = h->orb.init(cfg->image_width, cfg->image_height, 4 * cfg->max_batch_pairs + n_cache,
                        (cfg->flags & MSF_FLAG_BLUR_TIE_HALF_UP) != 0, profile, (cfg->flags & MSF_FLAG_FAST_DENSE) != 0,
                        (cfg->flags & MSF_FLAG_LEVEL_SIZE_MUL_INV) != 0, (cfg->flags & MSF_FLAG_FAST_STREAM) ? 1 : 8,
                        (cfg->flags & MSF_FLAG_BLUR_SUM256) != 0, 2 * cfg->max_batch_pairs);
    } else {
      if (cfg->image_width != 640 || cfg->image_height != 480) {
          return fail(nullptr, MSF_ERR_UNSUPPORTED, "LoFTR_teacher is a fixed-shape 1x1x480x640 graph (model/LoFTR_teacher.onnx)");
      }
      h->fc_slot0 = 2 * cfg->max_batch_pairs;
      err = h->loftr.init(cfg->weights_path, cfg->max_batch_pairs, profile, (cfg->flags & MSF_FLAG_KEEP_DEBUG) != 0, n_cache,
                          (cfg->flags & MSF_FLAG_LOFTR_F32) != 0);
    }
    if (!err.empty()) {
      const bool io = err.rfind("io:", 0) == 0, arg = err.rfind("arg:", 0) == 0;
      return fail(nullptr, io ? MSF_ERR_IO : arg ? MSF_ERR_INVALID_ARG : MSF_ERR_HIP, err);
    }
    // hipStreamDefault (a "blocking" stream): work on the handle's own stream is ordered against the legacy null stream,
    // which is where a caller that passes stream = NULL (e.g. torch's default stream) has its own work (msf_abi.h)
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamDefault)) != hipSuccess) {
      return hip_fail(nullptr, "hipStreamCreate", e);
    }
    if (n_cache > 0) {
      std::vector<int32_t> ids(n_cache);
      for (int i = 0; i < n_cache; i++) ids[i] = h->fc_slot0 + i;
      if ((e = hipMalloc(&h->d_fc_slots, (size_t)n_cache * sizeof(int32_t))) != hipSuccess ||
          (e = hipMemcpy(h->d_fc_slots, ids.data(), (size_t)n_cache * sizeof(int32_t), hipMemcpyHostToDevice)) != hipSuccess) {
        return hip_fail(nullptr, "hipMalloc(frame cache slots)", e);
      }
      h->fc.resize(n_cache);
    }
    guard.h = nullptr;
    *out = h;
    return MSF_OK;
  } catch (...) {
    return host_exception(nullptr, "msf_create");
  }
}

void msf_destroy(msf_handle* h) {
  if (!h) return;
  hipSetDevice(h->cfg.device);
  hipDeviceSynchronize();
  h->orb.destroy();
  h->loftr.destroy();
  hipFree(h->d_stage);
  hipFree(h->d_out_base);
  if (h->h_pin) hipHostFree(h->h_pin);
  hipFree(h->d_n);
  hipFree(h->d_maps);
  hipFree(h->d_store);
  hipFree(h->d_idx);
  hipFree(h->d_hyp);
  hipFree(h->d_hyp_inl);
  hipFree(h->d_hyp_m);
  hipFree(h->d_render);
  hipFree(h->d_render_m);
  hipFree(h->d_fc_slots);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}

int msf_set_threshold(msf_handle* h, float value) {
  try {
    if (!h) return MSF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    h->cfg.threshold = value;
    return MSF_OK;
  } catch (...) {
    return host_exception(h, "msf_set_threshold");
  }
}

const char* msf_last_error(const msf_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int msf_match_batch_device(msf_handle* h, int32_t n_pairs, const uint8_t* d_a, const uint8_t* d_b,
                           int64_t frame_stride, int64_t row_stride, msf_match* d_out, int32_t cap_per_pair,
                           int32_t* d_n_out, void* stream) {
  try {
    if (!h) return MSF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (n_pairs < 0 || !d_a || !d_b || !d_out || !d_n_out || cap_per_pair < 1)
      return fail(h, MSF_ERR_INVALID_ARG, "msf_match_batch_device: bad argument");
    hipError_t e = hipSetDevice(h->cfg.device);
    if (e != hipSuccess) return hip_fail(h, "hipSetDevice", e);
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    int rc = run_device(h, n_pairs, d_a, d_b, frame_stride, row_stride, d_out, cap_per_pair, d_n_out, st);
    if (rc != MSF_OK) return rc;
    if (!stream && (e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(h, "hipStreamSynchronize", e);
    return MSF_OK;
  } catch (...) {
    return host_exception(h, "msf_match_batch_device");
  }
}

int msf_match_batch(msf_handle* h, int32_t n_pairs, const msf_image* a, const msf_image* b, msf_match* out,
                    int32_t cap_per_pair, int32_t* n_out) {
  try {
    if (!h) return MSF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (n_pairs < 0 || !a || !b || !out || !n_out || cap_per_pair < 1)
      return fail(h, MSF_ERR_INVALID_ARG, "msf_match_batch: bad argument");
    hipError_t e = hipSetDevice(h->cfg.device);
    if (e != hipSuccess) return hip_fail(h, "hipSetDevice", e);
    const int W = h->cfg.image_width, H = h->cfg.image_height, maxp = h->cfg.max_batch_pairs;
    for (int i = 0; i < n_pairs; i++) {
      if (!a[i].data || !b[i].data || a[i].width != W || a[i].height != H || b[i].width != W || b[i].height != H ||
          a[i].stride < W || b[i].stride < W)
        return fail(h, MSF_ERR_INVALID_ARG, "msf_match_batch: image size differs from the handle's, or null data");
    }
    if (int rc = ensure_stage(h)) return rc;
    if (n_pairs == 1 && !h->fc.empty()) return match_pair_cached(h, &a[0], &b[0], out, cap_per_pair, n_out);
    hipStream_t st = h->stream;
    std::vector<msf_match> tmp;
    std::vector<int32_t> cnt(maxp);
    bool capacity = false;
    for (int p0 = 0; p0 < n_pairs; p0 += maxp) {
      const int n = n_pairs - p0 < maxp ? n_pairs - p0 : maxp;
      uint8_t* dA = h->d_stage;
      uint8_t* dB = h->d_stage + (size_t)maxp * h->stage_frame;
      for (int i = 0; i < n; i++) {
        if ((e = hipMemcpy2DAsync(dA + (size_t)i * h->stage_frame, h->stage_pitch, a[p0 + i].data, a[p0 + i].stride, W, H,
                                  hipMemcpyHostToDevice, st)) != hipSuccess) return hip_fail(h, "hipMemcpy2DAsync", e);
        if ((e = hipMemcpy2DAsync(dB + (size_t)i * h->stage_frame, h->stage_pitch, b[p0 + i].data, b[p0 + i].stride, W, H,
                                  hipMemcpyHostToDevice, st)) != hipSuccess) return hip_fail(h, "hipMemcpy2DAsync", e);
      }
      if (n_pairs == 1) {   // the drop-in call: count in the record before the list, one copy, one synchronisation
        int rc = run_device(h, 1, dA, dB, h->stage_frame, h->stage_pitch, h->d_out, h->stage_cap,
                            reinterpret_cast<int32_t*>(h->d_out_base), st);
        if (rc != MSF_OK) return rc;
        return fetch_single(h, out, cap_per_pair, n_out, st);
      }
      int rc = run_device(h, n, dA, dB, h->stage_frame, h->stage_pitch, h->d_out, h->stage_cap, h->d_n, st);
      if (rc != MSF_OK) return rc;
      if ((e = hipMemcpyAsync(cnt.data(), h->d_n, (size_t)n * 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return hip_fail(h, "hipMemcpyAsync", e);
      if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(h, "hipStreamSynchronize", e);
      for (int i = 0; i < n; i++) {
        int32_t c = cnt[i];
        n_out[p0 + i] = c;
        if (c < 0) { capacity = true; continue; }
        int avail = c < h->stage_cap ? c : h->stage_cap;
        if (avail < c && cap_per_pair > avail) capacity = true;  // staging list shorter than what the caller asked for
        int w = avail < cap_per_pair ? avail : cap_per_pair;
        if (w > 0 && (e = hipMemcpy(out + (size_t)(p0 + i) * cap_per_pair, h->d_out + (size_t)i * h->stage_cap,
                                    (size_t)w * sizeof(msf_match), hipMemcpyDeviceToHost)) != hipSuccess)
          return hip_fail(h, "hipMemcpy", e);
      }
    }
    if (capacity) return fail(h, MSF_ERR_CAPACITY, "at least one pair has no valid result (n_out = -1): a fixed-capacity device list overflowed, or a unit of the ORB walker launch gave up a bounded wait");
    return MSF_OK;
  } catch (...) {
    return host_exception(h, "msf_match_batch");
  }
}

int msf_match_pair(msf_handle* h, const msf_image* a, const msf_image* b, msf_match* out, int32_t cap,
                   int32_t* n_out) {
  return msf_match_batch(h, 1, a, b, out, cap, n_out);
}

int msf_extract_device(msf_handle* h, int32_t n_frames, const uint8_t* d_frames, int64_t frame_stride,
                       int64_t row_stride, int32_t first_slot, void* stream) {
  try {
    if (!h) return MSF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    const bool is_orb = h->cfg.kind == MSF_KIND_ORB;
    // both kinds: the caller's slots are [0, 2P); what lies beyond (scratch of the stateless calls, the transparent
    // frame cache) belongs to the handle
    if (n_frames < 0 || !d_frames || first_slot < 0 || (long long)first_slot + n_frames > 2ll * h->cfg.max_batch_pairs)
      return fail(h, MSF_ERR_INVALID_ARG, "msf_extract_device: slot range outside [0, 2*max_batch_pairs)");
    if (((uintptr_t)d_frames | (uintptr_t)frame_stride | (uintptr_t)row_stride) & 15)
      return fail(h, MSF_ERR_INVALID_ARG, "device frames must be 16-byte aligned with strides multiple of 16");
    hipError_t e = hipSetDevice(h->cfg.device);
    if (e != hipSuccess) return hip_fail(h, "hipSetDevice", e);
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    if (row_stride < h->cfg.image_width ||
        (n_frames > 1 && frame_stride < row_stride * (long long)h->cfg.image_height))   // one frame: its stride is unused
      return fail(h, MSF_ERR_INVALID_ARG, "msf_extract_device: strides smaller than the frame");
    if (is_orb) {
      msf::FrameSrc src{d_frames, d_frames, n_frames, first_slot, frame_stride, (int)row_stride};
      if ((e = h->orb.extract(src, n_frames, st)) != hipSuccess) return hip_fail(h, "orb extract", e);
    } else if ((e = h->loftr.extract(n_frames, d_frames, frame_stride, (int)row_stride, first_slot, st)) != hipSuccess) {
      return hip_fail(h, "loftr extract", e);
    }
    if (!stream && (e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(h, "hipStreamSynchronize", e);
    return MSF_OK;
  } catch (...) {
    return host_exception(h, "msf_extract_device");
  }
}

int msf_match_slots_device(msf_handle* h, int32_t n_pairs, const int32_t* d_slot_a, const int32_t* d_slot_b,
                           msf_match* d_out, int32_t cap_per_pair, int32_t* d_n_out, void* stream) {
  try {
    if (!h) return MSF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (n_pairs < 0 || !d_slot_a || !d_slot_b || !d_out || !d_n_out || cap_per_pair < 1)
      return fail(h, MSF_ERR_INVALID_ARG, "msf_match_slots_device: bad argument");
    hipError_t e = hipSetDevice(h->cfg.device);
    if (e != hipSuccess) return hip_fail(h, "hipSetDevice", e);
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    if (h->cfg.kind != MSF_KIND_ORB && n_pairs > h->cfg.max_batch_pairs)   // LoFTR works on per-pair token buffers
      return fail(h, MSF_ERR_INVALID_ARG, "n_pairs exceeds max_batch_pairs");
    const int public_slots = 2 * h->cfg.max_batch_pairs;
    e = h->cfg.kind == MSF_KIND_ORB
            ? h->orb.match(n_pairs, d_slot_a, d_slot_b, h->cfg.threshold, d_out, cap_per_pair, d_n_out, st, 0, public_slots)
            : h->loftr.match_slots(n_pairs, d_slot_a, d_slot_b, h->cfg.threshold, d_out, cap_per_pair, d_n_out, st, public_slots);
    if (e != hipSuccess) return hip_fail(h, "match slots", e);
    if (!stream && (e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(h, "hipStreamSynchronize", e);
    return MSF_OK;
  } catch (...) {
    return host_exception(h, "msf_match_slots_device");
  }
}

int msf_pack_matches_device(msf_handle* h, int32_t n_pairs, const msf_match* d_in, int32_t cap_per_pair,
                            const int32_t* d_n_out, msf_match* d_packed, int32_t* d_offsets, void* stream) {
  try {
    if (!h) return MSF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (n_pairs < 0 || !d_in || !d_n_out || !d_packed || !d_offsets || cap_per_pair < 1)
      return fail(h, MSF_ERR_INVALID_ARG, "msf_pack_matches_device: bad argument");
    hipError_t e = hipSetDevice(h->cfg.device);
    if (e != hipSuccess) return hip_fail(h, "hipSetDevice", e);
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    if ((e = msf::pack_matches(n_pairs, d_in, cap_per_pair, d_n_out, d_packed, d_offsets, st)) != hipSuccess)
      return hip_fail(h, "pack_matches", e);
    if (!stream && (e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(h, "hipStreamSynchronize", e);
    return MSF_OK;
  } catch (...) {
    return host_exception(h, "msf_pack_matches_device");
  }
}

int msf_set_mappoints(msf_handle* h, int32_t map_slot, const int32_t* keys, int32_t n_keys) {
  try {
    if (!h) return MSF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    const int n_maps = 2 * h->cfg.max_batch_pairs;
    const long long n_px = (long long)h->cfg.image_width * h->cfg.image_height;
    if (map_slot < 0 || map_slot >= n_maps || n_keys < 0 || (n_keys > 0 && !keys))
      return fail(h, MSF_ERR_INVALID_ARG, "msf_set_mappoints: bad argument");
    hipError_t e = hipSetDevice(h->cfg.device);
    if (e != hipSuccess) return hip_fail(h, "hipSetDevice", e);
    if (int rc = ensure_maps(h)) return rc;
    h->map_stage.assign(h->map_words, 0u);
    for (int i = 0; i < n_keys; i++) {
      // KeyPointMap::SetMapPoint ignores points outside the image (KeyPointMap.cc:38-39); a key is y*cols + x
      if (keys[i] < 0 || keys[i] >= n_px) continue;
      h->map_stage[keys[i] >> 5] |= 1u << (keys[i] & 31);
    }
    e = hipMemcpyAsync(h->d_maps + (size_t)map_slot * h->map_words, h->map_stage.data(),
                       (size_t)h->map_words * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return hip_fail(h, "hipMemcpyAsync(map bitmap)", e);
    if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return hip_fail(h, "hipStreamSynchronize", e);
    return MSF_OK;
  } catch (...) {
    return host_exception(h, "msf_set_mappoints");
  }
}

int msf_count_mappoint_matches_device(msf_handle* h, int32_t n_pairs, const msf_match* d_matches,
                                      int32_t cap_per_pair, const int32_t* d_n_matches, const int32_t* d_map_a,
                                      const int32_t* d_map_b, int32_t* d_num_mp, void* stream) {
  try {
    if (!h) return MSF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (n_pairs < 0 || !d_matches || !d_n_matches || !d_map_a || !d_map_b || !d_num_mp || cap_per_pair < 1)
      return fail(h, MSF_ERR_INVALID_ARG, "msf_count_mappoint_matches_device: bad argument");
    if (!h->d_maps) return fail(h, MSF_ERR_INVALID_ARG, "msf_count_mappoint_matches_device: no map slot was ever set");
    hipError_t e = hipSetDevice(h->cfg.device);
    if (e != hipSuccess) return hip_fail(h, "hipSetDevice", e);
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    e = msf::count_mappoint_matches(n_pairs, d_matches, cap_per_pair, d_n_matches, d_map_a, d_map_b, h->d_maps, h->n_maps,
                                    h->map_words, h->cfg.image_width, h->cfg.image_height, d_num_mp, st);
    if (e != hipSuccess) return hip_fail(h, "count_mappoint_matches", e);
    if (!stream && (e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(h, "hipStreamSynchronize", e);
    return MSF_OK;
  } catch (...) {
    return host_exception(h, "msf_count_mappoint_matches_device");
  }
}

int msf_store_frame(msf_handle* h, int32_t slot, const msf_image* img) {
  try {
    if (!h) return MSF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    const int W = h->cfg.image_width, H = h->cfg.image_height, maxp = h->cfg.max_batch_pairs;
    if (slot < 0 || slot >= 2 * maxp || !img || !img->data || img->width != W || img->height != H || img->stride < W)
      return fail(h, MSF_ERR_INVALID_ARG, "msf_store_frame: slot outside [0, 2*max_batch_pairs) or image size differs from the handle's");
    hipError_t e = hipSetDevice(h->cfg.device);
    if (e != hipSuccess) return hip_fail(h, "hipSetDevice", e);
    if (int rc = ensure_stage(h)) return rc;
    if (!h->d_store) {
      if ((e = hipMalloc(&h->d_store, (size_t)2 * maxp * h->stage_frame)) != hipSuccess) return hip_fail(h, "hipMalloc(frame store)", e);
      if ((e = hipMalloc(&h->d_idx, (size_t)3 * maxp * sizeof(int32_t))) != hipSuccess) return hip_fail(h, "hipMalloc(idx)", e);
    }
    hipStream_t st = h->stream;
    uint8_t* dst = h->d_store + (size_t)slot * h->stage_frame;
    if ((e = hipMemcpy2DAsync(dst, h->stage_pitch, img->data, img->stride, W, H, hipMemcpyHostToDevice, st)) != hipSuccess)
      return hip_fail(h, "hipMemcpy2DAsync", e);
    if (h->cfg.kind == MSF_KIND_ORB) {   // features are extracted once, here
      msf::FrameSrc src{dst, dst, 1, slot, h->stage_frame, h->stage_pitch};
      if ((e = h->orb.extract(src, 1, st)) != hipSuccess) return hip_fail(h, "orb extract", e);
    } else if ((e = h->loftr.extract(1, dst, h->stage_frame, h->stage_pitch, slot, st)) != hipSuccess) {
      return hip_fail(h, "loftr extract", e);   // backbone tokens of the frame, once
    }
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(h, "hipStreamSynchronize", e);
    return MSF_OK;
  } catch (...) {
    return host_exception(h, "msf_store_frame");
  }
}

int msf_match_one_to_many(msf_handle* h, int32_t query_slot, int32_t n, const int32_t* slots, int32_t* num_matches,
                          int32_t* num_mp, msf_match* out, int32_t cap_per_pair) {
  try {
    if (!h) return MSF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    const int maxp = h->cfg.max_batch_pairs;
    if (n < 0 || (n > 0 && (!slots || !num_matches)) || (out && cap_per_pair < 1))
      return fail(h, MSF_ERR_INVALID_ARG, "msf_match_one_to_many: bad argument");
    if (n == 0) return MSF_OK;
    if (n > maxp) return fail(h, MSF_ERR_INVALID_ARG, "msf_match_one_to_many: n exceeds max_batch_pairs");
    if (!h->d_store) return fail(h, MSF_ERR_INVALID_ARG, "msf_match_one_to_many: no frame was stored");
    if (query_slot < 0 || query_slot >= 2 * maxp) return fail(h, MSF_ERR_INVALID_ARG, "msf_match_one_to_many: bad query slot");
    for (int i = 0; i < n; i++)
      if (slots[i] < 0 || slots[i] >= 2 * maxp) return fail(h, MSF_ERR_INVALID_ARG, "msf_match_one_to_many: bad slot");
    if (num_mp && !h->d_maps) return fail(h, MSF_ERR_INVALID_ARG, "msf_match_one_to_many: no map slot was ever set");
    hipError_t e = hipSetDevice(h->cfg.device);
    if (e != hipSuccess) return hip_fail(h, "hipSetDevice", e);
    hipStream_t st = h->stream;
    h->idx_stage.resize((size_t)2 * maxp);
    for (int i = 0; i < n; i++) { h->idx_stage[i] = query_slot; h->idx_stage[maxp + i] = slots[i]; }
    if ((e = hipMemcpyAsync(h->d_idx, h->idx_stage.data(), (size_t)2 * maxp * sizeof(int32_t), hipMemcpyHostToDevice, st)) != hipSuccess)
      return hip_fail(h, "hipMemcpyAsync(idx)", e);
    e = h->cfg.kind == MSF_KIND_ORB
            ? h->orb.match(n, h->d_idx, h->d_idx + maxp, h->cfg.threshold, h->d_out, h->stage_cap, h->d_n, st)
            : h->loftr.match_slots(n, h->d_idx, h->d_idx + maxp, h->cfg.threshold, h->d_out, h->stage_cap, h->d_n, st);
    if (e != hipSuccess) return hip_fail(h, "match slots", e);
    if (num_mp) {
      e = msf::count_mappoint_matches(n, h->d_out, h->stage_cap, h->d_n, h->d_idx, h->d_idx + maxp, h->d_maps, h->n_maps,
                                      h->map_words, h->cfg.image_width, h->cfg.image_height, h->d_idx + 2 * maxp, st);
      if (e != hipSuccess) return hip_fail(h, "count_mappoint_matches", e);
      if ((e = hipMemcpyAsync(num_mp, h->d_idx + 2 * maxp, (size_t)n * 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return hip_fail(h, "hipMemcpyAsync", e);
    }
    if ((e = hipMemcpyAsync(num_matches, h->d_n, (size_t)n * 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return hip_fail(h, "hipMemcpyAsync", e);
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(h, "hipStreamSynchronize", e);
    bool capacity = false;
    for (int i = 0; i < n; i++) {
      const int32_t c = num_matches[i];
      if (c < 0) { capacity = true; continue; }
      const int avail = c < h->stage_cap ? c : h->stage_cap;
      if (out) {
        if (avail < c && cap_per_pair > avail) capacity = true;
        const int w = avail < cap_per_pair ? avail : cap_per_pair;
        if (w > 0 && (e = hipMemcpy(out + (size_t)i * cap_per_pair, h->d_out + (size_t)i * h->stage_cap,
                                    (size_t)w * sizeof(msf_match), hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(h, "hipMemcpy", e);
      }
    }
    if (capacity) return fail(h, MSF_ERR_CAPACITY, "at least one pair has no valid result (n_out = -1): a fixed-capacity device list overflowed, or a unit of the ORB walker launch gave up a bounded wait");
    return MSF_OK;
  } catch (...) {
    return host_exception(h, "msf_match_one_to_many");
  }
}

int msf_check_hypotheses(msf_handle* h, int32_t model, int32_t n_hyp, const float* m21, const float* m12,
                         int32_t n_matches, const msf_match* matches, float sigma, float* scores, int32_t* best,
                         uint8_t* best_inliers) {
  try {
    if (!h) return MSF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    const bool homography = model == MSF_MODEL_HOMOGRAPHY;
    if ((!homography && model != MSF_MODEL_FUNDAMENTAL) || n_hyp < 0 || n_matches < 0 || n_matches > 8192 ||
        (n_hyp > 0 && (!m21 || !scores || (homography && !m12))) || (n_matches > 0 && !matches) || !best ||
        (n_matches > 0 && !best_inliers))
      return fail(h, MSF_ERR_INVALID_ARG, "msf_check_hypotheses: bad argument (models 0/1, at most 8192 matches)");
    *best = -1;
    for (int i = 0; i < n_matches; i++) best_inliers[i] = 0;   // FindHomography: vbMatchesInliers(N, false) (Initializer.cc:167)
    if (n_hyp == 0) return MSF_OK;
    hipError_t e = hipSetDevice(h->cfg.device);
    if (e != hipSuccess) return hip_fail(h, "hipSetDevice", e);
    if (n_hyp > h->hyp_cap || n_matches > h->hyp_match_cap) {
      hipFree(h->d_hyp); hipFree(h->d_hyp_inl); hipFree(h->d_hyp_m);
      h->d_hyp = nullptr; h->d_hyp_inl = nullptr; h->d_hyp_m = nullptr;
      h->hyp_cap = h->hyp_match_cap = 0;
      const int hc = n_hyp > 256 ? n_hyp : 256, mc = n_matches > 2048 ? n_matches : 2048;
      if ((e = hipMalloc(&h->d_hyp, (size_t)hc * 19 * sizeof(float))) != hipSuccess) return hip_fail(h, "hipMalloc", e);
      if ((e = hipMalloc(&h->d_hyp_inl, (size_t)hc * mc)) != hipSuccess) return hip_fail(h, "hipMalloc", e);
      if ((e = hipMalloc(&h->d_hyp_m, (size_t)mc * sizeof(msf_match))) != hipSuccess) return hip_fail(h, "hipMalloc", e);
      h->hyp_cap = hc;
      h->hyp_match_cap = mc;
    }
    hipStream_t st = h->stream;
    float* d21 = h->d_hyp;
    float* d12 = h->d_hyp + (size_t)9 * h->hyp_cap;
    float* dsc = h->d_hyp + (size_t)18 * h->hyp_cap;
    if ((e = hipMemcpyAsync(d21, m21, (size_t)n_hyp * 9 * sizeof(float), hipMemcpyHostToDevice, st)) != hipSuccess) return hip_fail(h, "hipMemcpyAsync", e);
    if (homography && (e = hipMemcpyAsync(d12, m12, (size_t)n_hyp * 9 * sizeof(float), hipMemcpyHostToDevice, st)) != hipSuccess) return hip_fail(h, "hipMemcpyAsync", e);
    if (n_matches && (e = hipMemcpyAsync(h->d_hyp_m, matches, (size_t)n_matches * sizeof(msf_match), hipMemcpyHostToDevice, st)) != hipSuccess) return hip_fail(h, "hipMemcpyAsync", e);
    if ((e = msf::check_hypotheses(model, n_hyp, d21, d12, n_matches, h->d_hyp_m, sigma, dsc, h->d_hyp_inl, st)) != hipSuccess)
      return hip_fail(h, "check_hypotheses", e);
    if ((e = hipMemcpyAsync(scores, dsc, (size_t)n_hyp * sizeof(float), hipMemcpyDeviceToHost, st)) != hipSuccess) return hip_fail(h, "hipMemcpyAsync", e);
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(h, "hipStreamSynchronize", e);
    // FindHomography / FindFundamental keep the first hypothesis whose score beats every earlier one (:190-194, :236-240)
    float score = 0.0f;
    for (int i = 0; i < n_hyp; i++)
      if (scores[i] > score) { score = scores[i]; *best = i; }
    if (*best >= 0 && n_matches &&
        (e = hipMemcpy(best_inliers, h->d_hyp_inl + (size_t)*best * n_matches, (size_t)n_matches, hipMemcpyDeviceToHost)) != hipSuccess)
      return hip_fail(h, "hipMemcpy", e);
    return MSF_OK;
  } catch (...) {
    return host_exception(h, "msf_check_hypotheses");
  }
}

int msf_render_match_image(msf_handle* h, const msf_image* f1, const msf_image* f2, const msf_match* matches,
                           int32_t n_matches, const uint8_t* has_mp1, const uint8_t* has_mp2, uint8_t* out_rgb,
                           int64_t out_stride) {
  try {
    if (!h) return MSF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    const int W = h->cfg.image_width, H = h->cfg.image_height;
    if (!f1 || !f2 || !f1->data || !f2->data || f1->width != W || f1->height != H || f2->width != W || f2->height != H ||
        f1->stride < W || f2->stride < W || n_matches < 0 || (n_matches > 0 && !matches) || !out_rgb || out_stride < 6ll * W)
      return fail(h, MSF_ERR_INVALID_ARG, "msf_render_match_image: bad argument or image size differs from the handle's");
    hipError_t e = hipSetDevice(h->cfg.device);
    if (e != hipSuccess) return hip_fail(h, "hipSetDevice", e);
    if (int rc = ensure_stage(h)) return rc;
    hipStream_t st = h->stream;
    // workspace: the two frames use the staging buffers; the RGB image is allocated by the first call, the list + flags
    // buffer grows when a call brings more matches than any before it (no allocation in the steady state)
    uint8_t* dA = h->d_stage;
    uint8_t* dB = h->d_stage + (size_t)h->cfg.max_batch_pairs * h->stage_frame;
    const size_t out_bytes = (size_t)6 * W * H;
    if (!h->d_render && (e = hipMalloc(&h->d_render, out_bytes)) != hipSuccess) return hip_fail(h, "hipMalloc(render image)", e);
    if (n_matches > h->render_cap) {
      if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(h, "hipStreamSynchronize", e);
      hipFree(h->d_render_m);
      h->d_render_m = nullptr;
      h->render_cap = 0;
      const int cap = n_matches > 4096 ? n_matches : 4096;
      if ((e = hipMalloc(&h->d_render_m, (size_t)cap * (sizeof(msf_match) + 2))) != hipSuccess) return hip_fail(h, "hipMalloc(render list)", e);
      h->render_cap = cap;
    }
    msf_match* d_m = n_matches ? h->d_render_m : nullptr;
    uint8_t* d_flags = n_matches ? reinterpret_cast<uint8_t*>(h->d_render_m + h->render_cap) : nullptr;
    // From here on asynchronous copies from / to the CALLER's buffers are in flight: whichever way the call leaves --
    // an error branch included -- the stream is drained first, so the caller may free or reuse them on return.
    struct Drain { hipStream_t s; bool armed = true; ~Drain() { if (armed) hipStreamSynchronize(s); } } drain{st};
    if (n_matches) {
      if ((e = hipMemcpyAsync(d_m, matches, (size_t)n_matches * sizeof(msf_match), hipMemcpyHostToDevice, st)) != hipSuccess) return hip_fail(h, "hipMemcpyAsync", e);
      if ((e = hipMemsetAsync(d_flags, 0, (size_t)2 * n_matches, st)) != hipSuccess) return hip_fail(h, "hipMemsetAsync", e);
      if (has_mp1 && (e = hipMemcpyAsync(d_flags, has_mp1, n_matches, hipMemcpyHostToDevice, st)) != hipSuccess) return hip_fail(h, "hipMemcpyAsync", e);
      if (has_mp2 && (e = hipMemcpyAsync(d_flags + n_matches, has_mp2, n_matches, hipMemcpyHostToDevice, st)) != hipSuccess) return hip_fail(h, "hipMemcpyAsync", e);
    }
    if ((e = hipMemcpy2DAsync(dA, h->stage_pitch, f1->data, f1->stride, W, H, hipMemcpyHostToDevice, st)) != hipSuccess) return hip_fail(h, "hipMemcpy2DAsync", e);
    if ((e = hipMemcpy2DAsync(dB, h->stage_pitch, f2->data, f2->stride, W, H, hipMemcpyHostToDevice, st)) != hipSuccess) return hip_fail(h, "hipMemcpy2DAsync", e);
    if ((e = msf::render_match_image(dA, dB, W, H, h->stage_pitch, d_m, d_flags, d_flags ? d_flags + n_matches : nullptr,
                                     n_matches, h->d_render, 6ll * W, st)) != hipSuccess) return hip_fail(h, "render_match_image", e);
    if ((e = hipMemcpy2DAsync(out_rgb, out_stride, h->d_render, (size_t)6 * W, (size_t)6 * W, H, hipMemcpyDeviceToHost, st)) != hipSuccess) return hip_fail(h, "hipMemcpy2DAsync", e);
    drain.armed = false;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(h, "hipStreamSynchronize", e);   // the one wait of the call
    return MSF_OK;
  } catch (...) {
    return host_exception(h, "msf_render_match_image");
  }
}

int msf_debug_get(msf_handle* h, int32_t what, int32_t slot, int32_t level, void* host_out, size_t cap_bytes,
                  size_t* n_bytes) {
  try {
    if (!h || !n_bytes || (!host_out && cap_bytes)) return MSF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    hipError_t e = hipSetDevice(h->cfg.device);
    if (e != hipSuccess) return hip_fail(h, "hipSetDevice", e);
    std::string err;
    int rc = h->cfg.kind == MSF_KIND_ORB ? h->orb.debug_get(what, slot, level, host_out, cap_bytes, n_bytes, &err)
                                          : h->loftr.debug_get(what, slot, level, host_out, cap_bytes, n_bytes, &err);
    if (rc != 0) return fail(h, rc, err);
    return MSF_OK;
  } catch (...) {
    return host_exception(h, "msf_debug_get");
  }
}

int msf_stage_times(msf_handle* h, const char** names, float* ms, int32_t cap) {
  try {
    if (!h || !names || !ms || cap < 1) return 0;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->cfg.device);
    return h->cfg.kind == MSF_KIND_ORB ? h->orb.stage_times(names, ms, cap) : h->loftr.stage_times(names, ms, cap);
  } catch (...) {
    return 0;
  }
}

int msf_frame_cache_stats(msf_handle* h, uint64_t* hits, uint64_t* misses, int32_t* capacity) {
  if (!h) return MSF_ERR_INVALID_ARG;
  try {
    std::lock_guard<std::mutex> lk(h->mu);
    if (hits) *hits = h->fc_hits;
    if (misses) *misses = h->fc_misses;
    if (capacity) *capacity = (int32_t)h->fc.size();
    return MSF_OK;
  } catch (...) {
    return host_exception(h, "msf_frame_cache_stats");
  }
}

/* LoFTR weights, on the host (no GPU needed): reads `path` -- the reference's ONNX model or the MSFLTR01 blob -- and
 * reports the number of tensors / floats and a digest of names, shapes and values; equal digests <=> identical weights. */
int msf_weights_info(const char* path, uint64_t* digest, int32_t* n_tensors, int64_t* n_floats) {
  try {
    if (!path) return fail(nullptr, MSF_ERR_INVALID_ARG, "msf_weights_info: null path");
    // test hook for the no-exceptions guarantee, armed only by the test's environment
    if (std::strcmp(path, "::throw::") == 0 && getenv("MSF_TEST_HOOKS")) throw std::bad_alloc();
    msf::WeightMap w;
    const std::string err = msf::load_weights(path, &w);
    if (!err.empty()) return fail(nullptr, MSF_ERR_IO, err);
    int64_t nf = 0;
    const uint64_t d = msf::weights_digest(w, &nf);
    if (digest) *digest = d;
    if (n_tensors) *n_tensors = (int32_t)w.size();
    if (n_floats) *n_floats = nf;
    return MSF_OK;
  } catch (...) {
    return host_exception(nullptr, "msf_weights_info");
  }
}

/* Writes the weights of `src_path` (ONNX model or blob) as an MSFLTR01 blob: a load-time cache, nothing more. */
int msf_convert_weights(const char* src_path, const char* dst_blob_path) {
  try {
    if (!src_path || !dst_blob_path) return fail(nullptr, MSF_ERR_INVALID_ARG, "msf_convert_weights: null path");
    msf::WeightMap w;
    std::string err = msf::load_weights(src_path, &w);
    if (err.empty()) err = msf::save_blob(dst_blob_path, w);
    if (!err.empty()) return fail(nullptr, MSF_ERR_IO, err);
    return MSF_OK;
  } catch (...) {
    return host_exception(nullptr, "msf_convert_weights");
  }
}

}  // extern "C"
