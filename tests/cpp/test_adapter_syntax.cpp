// -fsyntax-only translation unit (tests/test_adapter_syntax.py, build container only): the drop-in subclasses of
// hip_feature_matcher.h against the reference's real plugin interface, /root/reference/slam_pipeline/include/
// FeatureMatcher.h:41-47.  A drift of MatchFrames' signature, of MatchFramesResult's members or of FrameBase::imGray
// stops this from compiling.
#define MSF_WITH_SLAM_PIPELINE 1
#include "hip_feature_matcher.h"

#include <type_traits>

static_assert(std::is_base_of<SLAM_PIPELINE::FeatureMatcher, msf::HipOrbMatcher>::value, "ORB plugin is a FeatureMatcher");
static_assert(std::is_base_of<SLAM_PIPELINE::FeatureMatcher, msf::HipLoftrMatcher>::value, "LoFTR plugin is a FeatureMatcher");
static_assert(!std::is_abstract<msf::HipOrbMatcher>::value && !std::is_abstract<msf::HipLoftrMatcher>::value,
              "MatchFrames overrides the reference's pure virtual");
static_assert(!std::is_copy_constructible<msf::HipOrbMatcher>::value, "copying is deleted like featurematcher.h:10-11");

// what src/main.cpp:61-66,78-82 does with its matcher: construct on the stack, hand out a FeatureMatcher*
SLAM_PIPELINE::MatchFramesResult through_the_interface(SLAM_PIPELINE::FrameBase& cur, SLAM_PIPELINE::FrameBase& kf) {
  msf::HipOrbMatcher featureMatcher(0.6f);
  msf::HipLoftrMatcher dnnMatcher("../model/LoFTR_teacher.onnx", 0.1f);
  dnnMatcher.SetThreshold(0.15f);
  // the reference's own constructor call, src/main.cpp:62 (a wide literal: dnnfeaturematcher.h:11 takes std::wstring)
  msf::HipLoftrMatcher featureMatcherW(L"model/LoFTR_teacher.onnx");
  featureMatcherW.SetThreshold(0.1f);
  const std::wstring wide_path = L"model/LoFTR_teacher.onnx";
  msf::HipLoftrMatcher fromWString(wide_path, 0.15f, 640, 480, 16);
  const std::string narrow_path = "model/LoFTR_teacher.onnx";
  msf::HipLoftrMatcher fromString(narrow_path);
  SLAM_PIPELINE::FeatureMatcher* plugins[2] = {&featureMatcher, &dnnMatcher};
  SLAM_PIPELINE::MatchFramesResult r = plugins[0]->MatchFrames(cur, kf);
  SLAM_PIPELINE::MatchFramesResult r2 = plugins[1]->MatchFrames(cur, kf);
  r.DeleteMatch(0);
  return r.GetNumMatches() >= r2.GetNumMatches() ? r : r2;
}
