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

  virtual void SetThreshold(float value) { msf_set_threshold(h_, value); }

  virtual MatchResult MatchFrames(const ImageView& f1, const ImageView& f2) { return match_on(h_, f1, f2); }

  const char* LastError() const { return msf_last_error(h_); }
  msf_handle* handle() { return h_; }
  int max_batch_pairs() const { return max_pairs_; }   // capacity of the one-vs-many callers (hip_keyframe_database.h)
  int result_cap() const { return cap_; }

 protected:
  MatchResult match_on(msf_handle* h, const ImageView& f1, const ImageView& f2) {
    MatchResult r;
    msf_image a{f1.data, f1.width, f1.height, f1.stride}, b{f2.data, f2.width, f2.height, f2.stride};
    int32_t n = 0;
    buf_.resize(cap_);
    if (msf_match_pair(h, &a, &b, buf_.data(), cap_, &n) != MSF_OK || n <= 0) return r;  // empty on any error
    if (n > cap_) n = cap_;
    r.keyPoints1.reserve(n);
    r.keyPoints2.reserve(n);
    for (int i = 0; i < n; i++) {
      r.keyPoints1.push_back({buf_[i].x1, buf_[i].y1});
      r.keyPoints2.push_back({buf_[i].x2, buf_[i].y2});
    }
    return r;
  }

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
// The reference matcher takes whatever size the frames it is handed have (cv::ORB sizes its pyramid per call,
// featurematcher.cpp:10-17).  A device handle is built for one size (image_width x image_height: the size the
// one-vs-many callers and the batch entry points use), so MatchFrames on a pair of another size goes to a handle of that
// size, created on first use and kept in a small least-recently-used set; SetThreshold reaches all of them.  The two
// frames of one call must have the same size (every frame of one camera does); a mixed pair gives an empty result.
class HipFeatureMatcher : public HipMatcherBase {
 public:
  explicit HipFeatureMatcher(float threshold = 0.8f, int image_width = 640, int image_height = 480, int device = 0,
                             int max_batch_pairs = 1)
      : HipMatcherBase(make(threshold, image_width, image_height, device, max_batch_pairs), 2048),
        thr_(threshold), w_(image_width), h0_(image_height), dev_(device) {}
  ~HipFeatureMatcher() override {
    for (Extra& e : extra_) msf_destroy(e.h);
  }

  void SetThreshold(float value) override {
    thr_ = value;
    msf_set_threshold(h_, value);
    for (Extra& e : extra_) msf_set_threshold(e.h, value);
  }

  MatchResult MatchFrames(const ImageView& f1, const ImageView& f2) override {
    if (f1.width != f2.width || f1.height != f2.height) return MatchResult();
    if (f1.width == w_ && f1.height == h0_) return match_on(h_, f1, f2);
    msf_handle* h = handle_for(f1.width, f1.height);
    return h ? match_on(h, f1, f2) : MatchResult();
  }

  size_t extra_sizes() const { return extra_.size(); }   // handles held for other frame sizes (tests)
  static constexpr size_t kMaxExtraSizes = 4;

 private:
  struct Extra {
    int w, h_dim;
    msf_handle* h;
    uint64_t used;
  };
  msf_handle* handle_for(int w, int h) {
    for (Extra& e : extra_)
      if (e.w == w && e.h_dim == h) { e.used = ++tick_; return e.h; }
    msf_config c = make(thr_, w, h, dev_, 1);
    msf_handle* nh = nullptr;
    if (msf_create(&c, &nh) != MSF_OK) return nullptr;       // e.g. a size outside [64, 8192]: empty result
    if (extra_.size() >= kMaxExtraSizes) {                     // replace the least recently used size
      size_t v = 0;
      for (size_t i = 1; i < extra_.size(); i++)
        if (extra_[i].used < extra_[v].used) v = i;
      msf_destroy(extra_[v].h);
      extra_.erase(extra_.begin() + v);
    }
    try {
      extra_.push_back(Extra{w, h, nh, ++tick_});
    } catch (...) {
      msf_destroy(nh);
      return nullptr;
    }
    return nh;
  }
  static msf_config make(float thr, int w, int h, int dev, int max_pairs) {
    msf_config c;
    msf_default_config(&c, MSF_KIND_ORB);
    c.threshold = thr; c.image_width = w; c.image_height = h; c.device = dev; c.max_batch_pairs = max_pairs;
    return c;
  }
  float thr_;
  int w_, h0_, dev_;
  uint64_t tick_ = 0;
  std::vector<Extra> extra_;
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
  // the reference's own signature, dnnfeaturematcher.h:11-13 (a wide path because Ort::Session takes ORTCHAR_T = wchar_t on
  // the platform the app was written on); its call, src/main.cpp:62: DNNFeatureMatcher featureMatcher(L"model/LoFTR_teacher.onnx");
  explicit HipDNNFeatureMatcher(const std::wstring& model_file_path, float threshold = 0.15f, int64_t image_width = 640,
                                int64_t image_height = 480, int model_resolution = 16, int device = 0,
                                int max_batch_pairs = 1)
      : HipDNNFeatureMatcher(narrow(model_file_path), threshold, image_width, image_height, model_resolution, device,
                             max_batch_pairs) {}
  // wide and narrow string LITERALS would otherwise be ambiguous between the two string classes' converting constructors
  explicit HipDNNFeatureMatcher(const wchar_t* model_file_path, float threshold = 0.15f, int64_t image_width = 640,
                                int64_t image_height = 480, int model_resolution = 16, int device = 0,
                                int max_batch_pairs = 1)
      : HipDNNFeatureMatcher(std::wstring(model_file_path ? model_file_path : L""), threshold, image_width, image_height,
                             model_resolution, device, max_batch_pairs) {}
  explicit HipDNNFeatureMatcher(const char* model_file_path, float threshold = 0.15f, int64_t image_width = 640,
                                int64_t image_height = 480, int model_resolution = 16, int device = 0,
                                int max_batch_pairs = 1)
      : HipDNNFeatureMatcher(std::string(model_file_path ? model_file_path : ""), threshold, image_width, image_height,
                             model_resolution, device, max_batch_pairs) {}

  // wchar_t path -> UTF-8 (wchar_t is UTF-32 on Linux, UTF-16 with surrogate pairs where it is 16 bits wide)
  static std::string narrow(const std::wstring& w) {
    std::string out;
    out.reserve(w.size());
    for (size_t i = 0; i < w.size(); i++) {
      uint32_t c = (uint32_t)w[i];
      if (sizeof(wchar_t) == 2 && c >= 0xD800u && c < 0xDC00u && i + 1 < w.size()) {
        const uint32_t lo = (uint32_t)w[i + 1];
        if (lo >= 0xDC00u && lo < 0xE000u) { c = 0x10000u + ((c - 0xD800u) << 10) + (lo - 0xDC00u); i++; }
      }
      if (c < 0x80u) out.push_back((char)c);
      else if (c < 0x800u) { out.push_back((char)(0xC0u | (c >> 6))); out.push_back((char)(0x80u | (c & 0x3Fu))); }
      else if (c < 0x10000u) {
        out.push_back((char)(0xE0u | (c >> 12))); out.push_back((char)(0x80u | ((c >> 6) & 0x3Fu)));
        out.push_back((char)(0x80u | (c & 0x3Fu)));
      } else {
        out.push_back((char)(0xF0u | ((c >> 18) & 7u))); out.push_back((char)(0x80u | ((c >> 12) & 0x3Fu)));
        out.push_back((char)(0x80u | ((c >> 6) & 0x3Fu))); out.push_back((char)(0x80u | (c & 0x3Fu)));
      }
    }
    return out;
  }

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
