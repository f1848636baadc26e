// LoFTR_teacher (coarse-only, d_model 32, linear attention) on gfx950: host-side launcher interface.
// Replaces Ort::Session::Run + the threshold/decode loop of ::DNNFeatureMatcher::MatchFrames
// (src/dnnfeaturematcher.cpp:44-102).  Implemented in loftr_kernels.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "msf_abi.h"

namespace msf {

class LoftrPipeline {
 public:
  LoftrPipeline() = default;
  ~LoftrPipeline();
  // returns empty string on success; "io: ..." for weight-file problems
  // extra_slots: token slots beyond the 2 * max_pairs caller-visible ones (the handle's transparent frame cache)
  // f32_convs: MSF_FLAG_LOFTR_F32 (no split-bf16 kernels)
  std::string init(const char* weights_path, int max_pairs, bool profile, bool keep_debug, int extra_slots = 0,
                   bool f32_convs = false);
  void destroy();
  hipError_t match(int n_pairs, const uint8_t* d_a, const uint8_t* d_b, long long frame_stride, int row_stride,
                   float threshold, msf_match* d_out, int cap_per_pair, int32_t* d_n_out, hipStream_t st);
  // per-frame token cache: frame -> slot [0, 2*max_pairs), then pairs of slots
  hipError_t extract(int n_frames, const uint8_t* d_frames, long long frame_stride, int row_stride, int first_slot,
                     hipStream_t st);
  // slot_limit > 0: slots at or beyond it give n_out = -1 for the pair (the public entry points pass 2 * max_pairs, so a
  // caller cannot reach the handle's private cache slots); 0: every slot of the pipeline
  hipError_t match_slots(int n_pairs, const int32_t* d_slot_a, const int32_t* d_slot_b, float threshold,
                         msf_match* d_out, int cap_per_pair, int32_t* d_n_out, hipStream_t st, int slot_limit = 0);
  int max_slots() const;
  int debug_get(int what, int slot, int level, void* host_out, size_t cap, size_t* n_bytes, std::string* err);
  int stage_times(const char** names, float* ms, int cap);

  struct Impl;

 private:
  hipError_t transformer_and_head(int n_pairs, float threshold, msf_match* d_out, int cap_per_pair, int32_t* d_n_out,
                                  hipStream_t st);
  Impl* p_ = nullptr;
};

}  // namespace msf
