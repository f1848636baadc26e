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
    // wrong-size frame: empty result, no throw (reference error behaviour)
    msf::MatchResult r3 = matcher.MatchFrames({a.data(), 320, 240, STEP}, {b.data(), 320, 240, STEP});
    if (r3.GetNumMatches() != 0) return 1;
    std::printf("host mirror ok: %d matches\n", n);
    return 0;
  } catch (const std::exception& e) {
    std::printf("constructor threw: %s\n", e.what());
    return 2;
  }
}
