// C++ host-side mirror (csrc/hip_feature_matcher.h) against the CPU oracle, the way the reference's own code would
// call it: construct a matcher, call MatchFrames on two 8-bit frames, read keyPoints1/keyPoints2.
// Exit code 0 = parity, 2 = no GPU (constructor threw), 1 = mismatch.
#include <cstdio>
#include <cstring>
#include <vector>

#include "hip_feature_matcher.h"
#include "orb_oracle.h"

extern "C" int msf_synth_pair(uint64_t seed, int w, int h, int dx, int dy, int mode, int noise, uint8_t* a,
                              int64_t stride_a, uint8_t* b, int64_t stride_b);

int main() {
  // host only: the wide model path of the reference's constructor (dnnfeaturematcher.h:11, src/main.cpp:62) as UTF-8
  if (msf::HipDNNFeatureMatcher::narrow(L"model/LoFTR_teacher.onnx") != "model/LoFTR_teacher.onnx" ||
      msf::HipDNNFeatureMatcher::narrow(L"mod\u00e8le/\u20ac.onnx") != "mod\xc3\xa8le/\xe2\x82\xac.onnx" ||
      msf::HipDNNFeatureMatcher::narrow(L"\U0001F600") != "\xf0\x9f\x98\x80" || !msf::HipDNNFeatureMatcher::narrow(L"").empty()) {
    std::printf("wide path conversion\n");
    return 3;
  }
  const int W = 640, H = 480, STEP = 704;  // cv::Mat with a row step larger than the width
  std::vector<uint8_t> a((size_t)STEP * H), b((size_t)STEP * H);
  if (msf_synth_pair(0x5EED0000ull + 77, W, H, 9, -5, 0, 8, a.data(), STEP, b.data(), STEP)) return 1;
  try {
    msf::HipFeatureMatcher matcher(0.6f, W, H);   // src/main.cpp:66 constructs FeatureMatcher(0.6f)
    msf::MatchResult r = matcher.MatchFrames({a.data(), W, H, STEP}, {b.data(), W, H, STEP});
    orb_oracle_opts o;
    orb_oracle_default_opts(&o);
    orb_oracle_ctx* ca = orb_oracle_create(W, H, &o);
    orb_oracle_ctx* cb = orb_oracle_create(W, H, &o);
    std::vector<int32_t> exp(4 * 4096);
    int n = orb_oracle_match_frames(ca, cb, a.data(), STEP, b.data(), STEP, 0.6f, exp.data(), 4096);
    orb_oracle_destroy(ca);
    orb_oracle_destroy(cb);
    if (n != (int)r.GetNumMatches()) { std::printf("count %d vs %zu\n", n, r.GetNumMatches()); return 1; }
    for (int i = 0; i < n; i++)
      if (exp[4 * i] != r.keyPoints1[i].x || exp[4 * i + 1] != r.keyPoints1[i].y || exp[4 * i + 2] != r.keyPoints2[i].x ||
          exp[4 * i + 3] != r.keyPoints2[i].y) { std::printf("mismatch at %d\n", i); return 1; }
    matcher.SetThreshold(0.8f);
    msf::MatchResult r2 = matcher.MatchFrames({a.data(), W, H, STEP}, {b.data(), W, H, STEP});
    if (r2.GetNumMatches() < r.GetNumMatches()) return 1;
    // Frames of another size through the SAME object (the reference matcher takes whatever imGray it is given,
    // src/featurematcher.cpp:10-17): views of 333 x 251 cut out of the same buffers, checked against the oracle at
    // that size; then the first size again (its handle and frame cache are untouched).
    {
      const int W2 = 333, H2 = 251;
      matcher.SetThreshold(0.6f);
      const uint8_t* a2 = a.data() + 40 * STEP + 24;
      const uint8_t* b2 = b.data() + 40 * STEP + 24;
      msf::MatchResult q = matcher.MatchFrames({a2, W2, H2, STEP}, {b2, W2, H2, STEP});
      orb_oracle_ctx* c2a = orb_oracle_create(W2, H2, &o);
      orb_oracle_ctx* c2b = orb_oracle_create(W2, H2, &o);
      const int n2 = orb_oracle_match_frames(c2a, c2b, a2, STEP, b2, STEP, 0.6f, exp.data(), 4096);
      orb_oracle_destroy(c2a);
      orb_oracle_destroy(c2b);
      if (n2 < 10 || n2 != (int)q.GetNumMatches()) { std::printf("second size: count %d vs %zu\n", n2, q.GetNumMatches()); return 1; }
      for (int i = 0; i < n2; i++)
        if (exp[4 * i] != q.keyPoints1[i].x || exp[4 * i + 1] != q.keyPoints1[i].y || exp[4 * i + 2] != q.keyPoints2[i].x ||
            exp[4 * i + 3] != q.keyPoints2[i].y) { std::printf("second size: mismatch at %d\n", i); return 1; }
      if (matcher.extra_sizes() != 1) return 1;
      msf::MatchResult again = matcher.MatchFrames({a.data(), W, H, STEP}, {b.data(), W, H, STEP});
      if (again.GetNumMatches() != r.GetNumMatches()) return 1;
      // six more sizes: the set of extra handles stays bounded (least recently used size replaced)
      for (int k = 0; k < 6; k++) {
        msf::MatchResult t = matcher.MatchFrames({a2, 200 + 16 * k, 160, STEP}, {b2, 200 + 16 * k, 160, STEP});
        (void)t;
      }
      if (matcher.extra_sizes() > msf::HipFeatureMatcher::kMaxExtraSizes) return 1;
    }
    // frames of two different sizes in one call, or a size no handle can be built for: empty result, no throw
    msf::MatchResult r3 = matcher.MatchFrames({a.data(), 320, 240, STEP}, {b.data(), 640, 480, STEP});
    if (r3.GetNumMatches() != 0) return 1;
    msf::MatchResult r4 = matcher.MatchFrames({a.data(), 32, 32, STEP}, {b.data(), 32, 32, STEP});
    if (r4.GetNumMatches() != 0) return 1;
    std::printf("host mirror ok: %d matches\n", n);
    return 0;
  } catch (const std::exception& e) {
    std::printf("constructor threw: %s\n", e.what());
    return 2;
  }
}
