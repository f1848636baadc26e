// ORB extract + match on gfx950: host-side launcher interface (implemented in orb_kernels.hip).
//
// Replaces the arithmetic behind the reference's ::FeatureMatcher::MatchFrames
// (src/featurematcher.cpp:10-45): cv::ORB::detectAndCompute x2 + BFMatcher::knnMatch(k=2)
// + ratio test.  Stage names follow SURVEY.md 2.4 (K1..K11).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "msf_abi.h"

namespace msf {

constexpr int kOrbLevels = 8;
constexpr int kKpCap = 2048;       // keypoints per frame (cv::ORB nfeatures = 500, + ties)
constexpr int kOrbStages = 6;
constexpr int kS1Cap = 8192;       // per level: keypoints kept by retainBest(2N) on the FAST score (+ ties)

constexpr int kWkMaxNx = 17;   // walker strips across a level (4096 px)
struct OrbLevelInfo {
  int w, h, pitch;     // level size, row pitch in bytes (multiple of 16)
  int quota;           // nfeaturesPerLevel
  float scale;         // getScale(level)
  int cand_cap;        // FULL capacity of the FAST candidate list of this level (the dense detector: w h / 8)
  int cand_off;        // offset (entries) of this level inside a frame's full-capacity region (dense calls)
  int s1_off;          // offset (entries) of this level inside a slot's stage-1 array
  int tiles_x, tiles_y, tile_base;  // FAST tiling (128 x 32 tiles, dense kernel)
  // column strips of the streaming walker (one wave each) over the WHOLE level: strip (sx, sy) holds the 256-px window
  // starting at x = sx * wk_px and owns rows [sy * wk_rows, (sy + 1) * wk_rows)
  int wk_nx, wk_ny, wk_px, wk_rows, wk_base;   // wk_base: flat index of the level's first strip
  int wk_fused;        // this level (as the DESTINATION of a resize) can be made by the walker of level l - 1
  int tab_yemit;       // offset (entries) of the per-source-row emit table of the resize INTO this level
  int tab_xstrip;      // offset (entries) of the first output group of each walker strip of level l - 1
  int wk_xg[kWkMaxNx + 1];   // the same numbers inside the kernel arguments (a scalar load instead of a dependent memory
                             // round trip in front of the strip's table loads); more strips than kWkMaxNx: not fused
  int tab_off;         // offset (entries) of this level's resize tables
  int samp_sx, samp_sy, samp_rows, samp_cols;   // tau_unit's sample lattice over [31, w-31) x [31, h-31); rows 0 = none
  long long pix_off;   // byte offset of this level inside a slot's pyramid blob (levels >= 1)
};

// per level: capacity of the PRIMARY candidate list (the output-sensitive pass: w h / 64, at least 4096; small levels keep
// their full capacity) and its offset (entries) inside a work row's primary region.  (A struct of its own, passed to the
// one kernel that needs it: inside OrbGeometry or OrbLevelInfo the extra members put the walker's copies into scratch.)
struct OrbPrimLists {
  int cap[kOrbLevels], off[kOrbLevels];
};

struct OrbGeometry {
  int nlevels;
  int w0, h0;
  int total_tiles;
  int total_strips;        // walker strips of all levels
  int max_level_tiles;     // largest tile count of one level
  int cand_total;          // candidate entries of a frame at full capacity (dense calls: one such region per frame in the pool)
  int prim_total;          // candidate entries per work row (primary lists)
  // r05: the candidate arrays are [work rows][prim_total] followed by a POOL of pool_entries entries.  The dense second
  // pass of a (frame, level) first writes into the level's primary list (a smooth frame's dense list is short); only a
  // level that overflows it takes a region of its full capacity from the pool (a bump allocation per call) and is redone
  // once more.  A call that is dense altogether (MSF_FLAG_FAST_DENSE, fewer than eight frames) lays its frames'
  // full-capacity regions over the pool.  Exhausted pool: MSF_ERR_CAPACITY for that frame, never a short list.
  long long pool_base;     // first pool entry
  unsigned pool_entries;
  int s1_total;            // stage-1 entries per slot
  long long pyr_bytes;     // pyramid blob bytes per slot (levels 1..)
  OrbLevelInfo lv[kOrbLevels];
  int umax[16];
};

// where level 0 of slot s lives: slot < n_a ? a + slot*frame_stride : b + (slot-n_a)*frame_stride
struct FrameSrc {
  const uint8_t* a;
  const uint8_t* b;
  int n_a;
  int slot0;               // first feature slot written by this call
  long long frame_stride;
  int row_stride;
};

class OrbPipeline {
 public:
  OrbPipeline() = default;
  ~OrbPipeline();
  // returns empty string on success
  std::string init(int width, int height, int max_slots, bool blur_half_up, bool profile, bool dense_fast = false,
                   bool level_size_mul_inv = false, int stream_min_frames = 8, bool blur_sum256 = false,
                   int work_frames = 0);
  void destroy();

  // extract features of n frames into slots [src.slot0, src.slot0 + n)
  hipError_t extract(const FrameSrc& src, int n_frames, hipStream_t st);
  // match slot pairs; d_slot_a/d_slot_b may be null => pair i = (slot_base + i, slot_base + n_pairs + i)
  // slot_limit > 0: slots at or beyond it give n_out = -1 (public entry points: 2 * max_batch_pairs); 0: max_slots()
  hipError_t match(int n_pairs, const int32_t* d_slot_a, const int32_t* d_slot_b, float ratio,
                   msf_match* d_out, int cap_per_pair, int32_t* d_n_out, hipStream_t st, int slot_base = 0,
                   int slot_limit = 0);

  const OrbGeometry& geom() const { return g_; }
  int max_slots() const { return max_slots_; }
  int debug_get(int what, int slot, int level, void* host_out, size_t cap, size_t* n_bytes, std::string* err);
  int stage_times(const char** names, float* ms, int cap);
  // true once after the handle switched to per-level walker launches because a unit of an earlier launch gave up waiting
  bool take_degraded_note() { const bool d = degraded_note_; degraded_note_ = false; return d; }
  bool walk_per_level() const { return walk_per_level_; }

 private:
  OrbGeometry g_{};
  OrbPrimLists prim_{};
  int max_slots_ = 0;
  int work_frames_ = 0;            // frames one extraction may hold = rows of the per-call arrays (0 at init: max_slots)
  int last_n_ = 0;                 // frames of the last extraction (debug_get maps a slot to its work row)
  bool half_up_ = false, profile_ = false, blur_sum256_ = false;
  int stream_min_frames_ = 8;      // calls with fewer frames take the dense FAST kernel (latency), others the streaming pass
  int force_tau_ = 0;              // > 0: every (frame, level) starts at this FAST score threshold (20 = dense)
  // device storage
  // per CALL (work_frames_ rows): pyramid, candidate lists, thresholds, walker state, stage-1 lists
  uint8_t* d_pyr_ = nullptr;
  uint32_t* d_tab_ = nullptr;      // resize tables per level: per group of 4 columns selectors / weights / pair offsets, per row source row | w1 << 16
  bool resize_shared_[kOrbLevels] = {};   // per level: k_resize may read three pixels' taps from one dword pair
  uint32_t* d_qstat_ = nullptr;           // [slots][levels][kQStat]: per (slot, level) the score histogram of the sampled quarter's
                                          // corners, the thresholds and the done counters of the walker launch (see k_walk)
  uint32_t* d_walk_abort_ = nullptr;      // [4] the walker launch's abort word (a unit gave up waiting); zeroed per call
  uint32_t* h_walk_abort_ = nullptr;      // pinned copy of it, written behind every walker launch, read by later calls
  bool degraded_note_ = false;            // the handle has just fallen back to per-level launches (reported once)
  int test_stall_calls_ = 0;
  bool fast_two_part_ = true;             // MSF_ORB_FAST_ONE_PART=1 clears it: no refinement of the first threshold
  int tau_predict_pct_ = 300;             // MSF_ORB_TAU_PREDICT (0 = sample every level)
  bool fused_ = true;                     // MSF_ORB_UNFUSED=1 clears it: k_resize x 7, then one FAST-only walker launch
  bool walk_per_level_ = false;           // MSF_ORB_WALK_PER_LEVEL=1: the fused walker as one launch per level (the
                                          // in-launch waits are then met at once); tests compare it with the one-launch default
  int test_stall_frame_ = -1;             // MSF_TEST_HOOKS + MSF_ORB_TEST_STALL_FRAME: see k_walk
  int tau2_margin_pct_ = 200;             // MSF_ORB_TAU2_MARGIN_PCT
  bool resize_generic_ = false;           // MSF_ORB_RESIZE_GENERIC: never
  uint32_t* d_cand_cnt_ = nullptr; // [slots][8]
  uint32_t* d_tau_ = nullptr;      // [2][slots][8] FAST score threshold used per (slot, level) | first estimate
  uint32_t* d_redo_ = nullptr;     // [1 + slots * 8] dense-pass queue: count, entries (frame * 8 + level)
  uint32_t* d_cand_ = nullptr;     // [work rows][prim_total] + pool: key = y << 16 | x
  uint8_t* d_cand_sc_ = nullptr;   // the same shape: FAST score
  uint2* d_cmap_ = nullptr;        // [work rows][8] where the list of (frame, level) lives: (first entry, capacity); reset per call
  uint32_t* d_pool_cnt_ = nullptr; // pool entries handed out in the current call
  uint32_t* d_s1_cnt_ = nullptr;   // [slots][8]
  uint4* d_s1_ = nullptr;          // [slots][s1_total] (key, response bits, score, 0)
  // per feature SLOT (max_slots_): what a later match reads
  msf_keypoint* d_kp_ = nullptr;   // [slots][kKpCap]
  uint8_t* d_desc_ = nullptr;      // [slots][kKpCap][32]
  uint32_t* d_kp_cnt_ = nullptr;   // [slots]
  uint32_t* d_status_ = nullptr;   // [slots]
  uint32_t* d_qres_ = nullptr;     // [kSplitMaxPairs][kKpCap] per-query results of the small-batch matcher
  uint32_t* d_done_ = nullptr;     // [kSplitMaxPairs] ticket counters (left at 0)
  // Stage events (MSF_FLAG_PROFILE).  A RING of event sets: a call records into the next free set and nobody waits, so a
  // caller can enqueue many batches ahead of the device; stage_times() harvests every finished set and returns the SUM
  // of the stage times since the last query (a query after every call sees that call's times, as before).  A set that
  // is needed again before it was queried is harvested first -- it is kEvRing calls old, long finished.
  static constexpr int kEvRing = 32;
  struct EvSet {
    hipEvent_t ev[kOrbStages + 1] = {};
    bool recorded = false, match_only = false;
  };
  EvSet evr_[kEvRing];
  int ev_cur_ = 0;
  hipEvent_t* ev_ = evr_[0].ev;                 // the current set
  float acc_ms_[5] = {};                        // harvested, not yet returned
  int acc_full_ = 0, acc_match_only_ = 0;       // calls behind acc_ms_: with an extraction / slot-pair matches only
  void ev_begin_call();
  void ev_harvest(int i);
  bool ev_ok_ = false, ev_extract_pending_ = false;
  FrameSrc last_src_{};
  bool last_fused_ = false;
  hipError_t extract_range(const FrameSrc& src, int n, hipStream_t st, hipEvent_t* evs);
};

}  // namespace msf
