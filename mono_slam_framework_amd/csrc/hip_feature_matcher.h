// hip_feature_matcher.h -- C++ host side above the C ABI (include/msf_abi.h).
//
// Two layers:
//  (1) msf::HipFeatureMatcher / msf::HipDNNFeatureMatcher: OpenCV-free mirrors of the reference's
//      ::FeatureMatcher (src/featurematcher.h:7-22) and ::DNNFeatureMatcher (src/dnnfeaturematcher.h:9-36):
//      same constructor arguments, SetThreshold, MatchFrames on a plain image view.  These compile anywhere
//      (tests/cpp/test_host_mirror.cpp builds them with g++ against libmsf.so).
//  (2) with -DMSF_WITH_SLAM_PIPELINE (needs the reference's slam_pipeline + OpenCV headers): drop-in subclasses
//      of SLAM_PIPELINE::FeatureMatcher (slam_pipeline/include/FeatureMatcher.h:41-47) that read
//      FrameBase::imGray (FrameBase.h:45) and fill MatchFramesResult (FeatureMatcher.h:15-19).
//      See INTEGRATION.md for the three-line change in src/main.cpp:65-66.
//
// Error behaviour mirrors the reference: the matchers never throw from MatchFrames; any non-zero ABI status
// gives an empty result (the reference returns an empty MatchFramesResult when a descriptor set is empty,
// featurematcher.cpp:23).  Copying is deleted like in the reference (featurematcher.h:10-11).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "msf_abi.h"

namespace msf {

struct ImageView {  // what the matchers read of FrameBase::imGray: CV_8UC1 data + step
  const uint8_t* data;
  int width, height;
  int64_t stride;
};

struct Point2i {
  int x, y;
};

struct MatchResult {  // MatchFramesResult without the frame pointers
  std::vector<Point2i> keyPoints1, keyPoints2;
  size_t GetNumMatches() const { return keyPoints1.size(); }
  void DeleteMatch(size_t idx) {  // slam_pipeline/include/FeatureMatcher.h:31-38
    if (idx < GetNumMatches()) {
      keyPoints1.erase(keyPoints1.begin() + idx);
      keyPoints2.erase(keyPoints2.begin() + idx);
    }
  }
};

class HipMatcherBase {
 public:
  HipMatcherBase(const HipMatcherBase&) = delete;
  HipMatcherBase& operator=(const HipMatcherBase&) = delete;
  virtual ~HipMatcherBase() { msf_destroy(h_); }

  void SetThreshold(float value) { msf_set_threshold(h_, value); }

  MatchResult MatchFrames(const ImageView& f1, const ImageView& f2) {
    MatchResult r;
    msf_image a{f1.data, f1.width, f1.height, f1.stride}, b{f2.data, f2.width, f2.height, f2.stride};
    int32_t n = 0;
    buf_.resize(cap_);
    if (msf_match_pair(h_, &a, &b, buf_.data(), cap_, &n) != MSF_OK || n <= 0) return r;  // empty on any error
    if (n > cap_) n = cap_;
    r.keyPoints1.reserve(n);
    r.keyPoints2.reserve(n);
    for (int i = 0; i < n; i++) {
      r.keyPoints1.push_back({buf_[i].x1, buf_[i].y1});
      r.keyPoints2.push_back({buf_[i].x2, buf_[i].y2});
    }
    return r;
  }

  const char* LastError() const { return msf_last_error(h_); }
  msf_handle* handle() { return h_; }
  int max_batch_pairs() const { return max_pairs_; }   // capacity of the one-vs-many callers (hip_keyframe_database.h)
  int result_cap() const { return cap_; }

 protected:
  explicit HipMatcherBase(const msf_config& cfg, int cap) : cap_(cap), max_pairs_(cfg.max_batch_pairs) {
    if (msf_create(&cfg, &h_) != MSF_OK)
      throw std::runtime_error(std::string("msf_create: ") + msf_last_error(nullptr));  // the reference's ctor throws too (Ort::Session)
  }
  msf_handle* h_ = nullptr;
  int cap_;
  int max_pairs_;
  std::vector<msf_match> buf_;
};

// ::FeatureMatcher(float threshold = 0.8f)  (featurematcher.h:9)
class HipFeatureMatcher : public HipMatcherBase {
 public:
  explicit HipFeatureMatcher(float threshold = 0.8f, int image_width = 640, int image_height = 480, int device = 0,
                             int max_batch_pairs = 1)
      : HipMatcherBase(make(threshold, image_width, image_height, device, max_batch_pairs), 2048) {}

 private:
  static msf_config make(float thr, int w, int h, int dev, int max_pairs) {
    msf_config c;
    msf_default_config(&c, MSF_KIND_ORB);
    c.threshold = thr; c.image_width = w; c.image_height = h; c.device = dev; c.max_batch_pairs = max_pairs;
    return c;
  }
};

// ::DNNFeatureMatcher(model_file_path, threshold = 0.15f, image_width = 640, image_height = 480,
//                     model_resolution = 16)  (dnnfeaturematcher.h:11-13); model_file_path is the model file the
// reference opens, model/LoFTR_teacher.onnx (read directly: csrc/weights_io.cpp), or an MSFLTR01 blob written by
// msf_convert_weights; empty = the blob shipped next to libmsf.so.
class HipDNNFeatureMatcher : public HipMatcherBase {
 public:
  explicit HipDNNFeatureMatcher(const std::string& model_file_path = "", float threshold = 0.15f,
                                int64_t image_width = 640, int64_t image_height = 480, int model_resolution = 16,
                                int device = 0, int max_batch_pairs = 1)
      : HipMatcherBase(make(model_file_path, threshold, image_width, image_height, model_resolution, device,
                            max_batch_pairs), 4096),
        path_(model_file_path) {}

 private:
  static msf_config make(const std::string& path, float thr, int64_t w, int64_t h, int res, int dev, int max_pairs) {
    if (res != 16) throw std::runtime_error("LoFTR_teacher works at 1/16 resolution only");
    msf_config c;
    msf_default_config(&c, MSF_KIND_LOFTR);
    c.threshold = thr; c.image_width = (int)w; c.image_height = (int)h; c.device = dev; c.max_batch_pairs = max_pairs;
    c.weights_path = path.empty() ? nullptr : path.c_str();
    return c;
  }
  std::string path_;
};

}  // namespace msf

#ifdef MSF_WITH_SLAM_PIPELINE
// Drop-in plugins for the unmodified slam_pipeline (needs OpenCV + the reference headers on the include path).
#include <opencv2/core/core.hpp>

#include "slam_pipeline/include/FeatureMatcher.h"

namespace msf {

template <class Impl>
class SlamPipelineMatcher : public SLAM_PIPELINE::FeatureMatcher {
 public:
  template <class... A>
  explicit SlamPipelineMatcher(A&&... a) : impl_(std::forward<A>(a)...) {}
  SlamPipelineMatcher(const SlamPipelineMatcher&) = delete;
  SlamPipelineMatcher& operator=(const SlamPipelineMatcher&) = delete;

  SLAM_PIPELINE::MatchFramesResult MatchFrames(SLAM_PIPELINE::FrameBase& pF1, SLAM_PIPELINE::FrameBase& pF2) override {
    SLAM_PIPELINE::MatchFramesResult out;
    out.pF1 = &pF1;   // both reference matchers set these (featurematcher.cpp:20-21); callers dereference them
    out.pF2 = &pF2;
    const cv::Mat& a = pF1.imGray;
    const cv::Mat& b = pF2.imGray;
    if (a.type() != CV_8UC1 || b.type() != CV_8UC1) return out;
    MatchResult r = impl_.MatchFrames(ImageView{a.data, a.cols, a.rows, (int64_t)a.step},
                                      ImageView{b.data, b.cols, b.rows, (int64_t)b.step});
    out.keyPoints1.reserve(r.keyPoints1.size());
    out.keyPoints2.reserve(r.keyPoints2.size());
    for (size_t i = 0; i < r.keyPoints1.size(); i++) {
      out.keyPoints1.emplace_back(r.keyPoints1[i].x, r.keyPoints1[i].y);
      out.keyPoints2.emplace_back(r.keyPoints2[i].x, r.keyPoints2[i].y);
    }
    return out;
  }
  void SetThreshold(float v) { impl_.SetThreshold(v); }
  Impl& impl() { return impl_; }   // the HipMatcherBase the one-vs-many callers share (hip_keyframe_database.h)

 private:
  Impl impl_;
};

using HipOrbMatcher = SlamPipelineMatcher<HipFeatureMatcher>;      // replaces ::FeatureMatcher
using HipLoftrMatcher = SlamPipelineMatcher<HipDNNFeatureMatcher>; // replaces ::DNNFeatureMatcher

}  // namespace msf
#endif  // MSF_WITH_SLAM_PIPELINE
