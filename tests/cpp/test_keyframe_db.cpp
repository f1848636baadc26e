// C++ host mirror of KeyFrameMatchDatabase (csrc/hip_keyframe_database.h) against the reference's own loops
// (slam_pipeline/src/KeyFrameDatabase.cc:23-117) run over the CPU ORB oracle: per-keyframe counts and the returned
// candidates must be identical.  Exit code 0 = parity, 2 = no GPU (constructor threw), 1 = mismatch.
#include <cstdio>
#include <memory>
#include <set>
#include <vector>

#include "hip_keyframe_database.h"
#include "orb_oracle.h"

extern "C" int msf_synth_pair(uint64_t seed, int w, int h, int dx, int dy, int mode, int noise, uint8_t* a,
                              int64_t stride_a, uint8_t* b, int64_t stride_b);

namespace {

constexpr int W = 640, H = 480;

struct TestKeyFrame {   // the KeyFrame fields the database touches
  unsigned long mnId = 0, mnLoopQuery = 0, mnRelocQuery = 0;
  float mRelocScore = 0;
  std::vector<uint8_t> imGray;
  std::set<int32_t> keys;                                   // KeyPointMap entries with a map point
  std::set<std::shared_ptr<TestKeyFrame>> connected;
  std::vector<std::shared_ptr<TestKeyFrame>> covisible;
};
using KFPtr = std::shared_ptr<TestKeyFrame>;

struct Traits {
  static msf::ImageView Image(const TestKeyFrame& f) { return {f.imGray.data(), W, H, W}; }
  static unsigned long Id(const TestKeyFrame& f) { return f.mnId; }
  static void MapPointKeys(const TestKeyFrame& f, std::vector<int32_t>* k) { k->assign(f.keys.begin(), f.keys.end()); }
  static bool IsConnected(const TestKeyFrame& q, const KFPtr& o) { return q.connected.count(o) != 0; }
  static std::vector<KFPtr> BestCovisibility(TestKeyFrame& f, int n) {
    return std::vector<KFPtr>(f.covisible.begin(), f.covisible.begin() + std::min<size_t>(n, f.covisible.size()));
  }
  static unsigned long& LoopQuery(TestKeyFrame& f) { return f.mnLoopQuery; }
  static unsigned long& RelocQuery(TestKeyFrame& f) { return f.mnRelocQuery; }
  static float& RelocScore(TestKeyFrame& f) { return f.mRelocScore; }
};

struct Oracle {   // MatchFrames on the CPU restatement
  orb_oracle_ctx *a, *b;
  std::vector<int32_t> m = std::vector<int32_t>(4 * 4096);
  Oracle() {
    orb_oracle_opts o;
    orb_oracle_default_opts(&o);
    a = orb_oracle_create(W, H, &o);
    b = orb_oracle_create(W, H, &o);
  }
  ~Oracle() { orb_oracle_destroy(a); orb_oracle_destroy(b); }
  int match(const TestKeyFrame& f1, const TestKeyFrame& f2) {
    return orb_oracle_match_frames(a, b, f1.imGray.data(), W, f2.imGray.data(), W, 0.6f, m.data(), 4096);
  }
};

bool has_mp(const TestKeyFrame& f, int x, int y) {   // KeyPointMap::GetMapPoint: exact cell, inside the image
  return x >= 0 && x < W && y >= 0 && y < H && f.keys.count(y * W + x) != 0;
}

// KeyFrameDatabase.cc:23-53 over the oracle
KFPtr ref_loop(Oracle& orc, std::vector<KFPtr>& frames, TestKeyFrame& q, size_t minNumMP, std::vector<int>* nums,
               std::vector<int>* nmps) {
  KFPtr cand;
  size_t maxNumMP = 0;
  for (auto& kf : frames) {
    const int n = orc.match(q, *kf);
    size_t numMP = 0;
    for (int i = 0; i < n; i++)
      if (has_mp(q, orc.m[4 * i], orc.m[4 * i + 1]) && has_mp(*kf, orc.m[4 * i + 2], orc.m[4 * i + 3])) numMP++;
    nums->push_back(n);
    nmps->push_back((int)numMP);
    if (n != 0 && kf->mnLoopQuery != q.mnId && !q.connected.count(kf) && numMP > minNumMP && numMP > maxNumMP) {
      cand = kf;
      maxNumMP = numMP;
    }
  }
  return cand;
}

// KeyFrameDatabase.cc:55-117 over the oracle
std::vector<KFPtr> ref_reloc(Oracle& orc, std::vector<KFPtr>& frames, TestKeyFrame& q, std::vector<int>* nums) {
  std::vector<std::pair<KFPtr, size_t>> counts;
  size_t maxNum = 0;
  for (auto& kf : frames) {
    const int n = orc.match(q, *kf);
    nums->push_back(n);
    kf->mnRelocQuery = q.mnId;
    kf->mRelocScore = static_cast<float>(n);
    counts.emplace_back(kf, n);
    if ((size_t)n > maxNum) maxNum = n;
  }
  const auto minNum = static_cast<size_t>(maxNum * 0.8f);
  float bestAcc = 0;
  std::vector<std::pair<KFPtr, float>> acc;
  for (auto& it : counts) {
    if (it.second < minNum) continue;
    float best = static_cast<float>(it.second), a = best;
    KFPtr bestKF = it.first;
    for (auto& k2 : Traits::BestCovisibility(*it.first, 10)) {
      if (k2->mnRelocQuery != q.mnId) continue;
      a += k2->mRelocScore;
      if (k2->mRelocScore > best) { bestKF = k2; best = k2->mRelocScore; }
    }
    acc.emplace_back(bestKF, a);
    if (a > bestAcc) bestAcc = a;
  }
  const float retain = 0.75f * bestAcc;
  std::set<KFPtr> added;
  std::vector<KFPtr> out;
  for (auto& it : acc)
    if (it.second > retain && !added.count(it.first)) { out.push_back(it.first); added.insert(it.first); }
  return out;
}

uint32_t lcg(uint32_t& s) { return s = s * 1664525u + 1013904223u; }

// two identical graphs: index g = 0 is handed to the database, g = 1 to the reference loops
void make_graph(std::vector<KFPtr> (&g)[2]) {
  const int shifts[4][2] = {{0, 0}, {24, 8}, {-30, 20}, {12, -28}};
  Oracle orc;
  std::vector<uint8_t> a((size_t)W * H);
  for (int sc = 0; sc < 3; sc++)
    for (int v = 0; v < 4; v++) {
      auto kf = std::make_shared<TestKeyFrame>();
      kf->mnId = 100 + g[0].size();
      kf->imGray.resize((size_t)W * H);
      msf_synth_pair(0x5EED0000ull + 900 + sc, W, H, shifts[v][0], shifts[v][1], 0, 8, a.data(), W, kf->imGray.data(), W);
      // map points on 60 % of the (truncated) key point coordinates
      orb_oracle_extract(orc.a, kf->imGray.data(), W);
      const int n = orb_oracle_num_keypoints(orc.a);
      const orb_oracle_kp* kp = orb_oracle_keypoints(orc.a);
      uint32_t s = 7 + (uint32_t)kf->mnId;
      for (int i = 0; i < n; i++)
        if (lcg(s) % 10 < 6) kf->keys.insert((int)kp[i].y * W + (int)kp[i].x);
      g[0].push_back(kf);
      g[1].push_back(std::make_shared<TestKeyFrame>(*kf));
    }
  for (int t = 0; t < 2; t++) {
    uint32_t s = 12345;
    for (size_t i = 0; i < g[t].size(); i++) {
      const int ncov = lcg(s) % 12, ncon = lcg(s) % 4;
      for (int k = 0; k < ncov; k++) {
        const size_t j = lcg(s) % g[t].size();
        if (j != i) g[t][i]->covisible.push_back(g[t][j]);
      }
      for (int k = 0; k < ncon && k < (int)g[t][i]->covisible.size(); k++) g[t][i]->connected.insert(g[t][i]->covisible[k]);
    }
  }
}

bool same(const std::vector<int32_t>& got, const std::vector<int>& exp, const char* what) {
  if (got.size() != exp.size()) { std::printf("%s: size %zu vs %zu\n", what, got.size(), exp.size()); return false; }
  for (size_t i = 0; i < got.size(); i++)
    if (got[i] != exp[i]) { std::printf("%s[%zu]: %d vs %d\n", what, i, got[i], exp[i]); return false; }
  return true;
}

}  // namespace

int main() {
  try {
    msf::HipFeatureMatcher matcher(0.6f, W, H, 0, /*max_batch_pairs=*/16);
    msf::HipKeyFrameMatchDatabase<TestKeyFrame, TestKeyFrame, Traits> db(&matcher);
    std::vector<KFPtr> g[2];
    make_graph(g);
    for (auto& kf : g[0]) db.add(kf);
    Oracle orc;

    // relocalisation: a new view of scene 1
    TestKeyFrame q;
    q.mnId = 500;
    q.imGray.resize((size_t)W * H);
    std::vector<uint8_t> tmp((size_t)W * H);
    msf_synth_pair(0x5EED0000ull + 901, W, H, 30, 14, 0, 8, tmp.data(), W, q.imGray.data(), W);
    TestKeyFrame q2 = q;
    std::vector<int> nums, nmps;
    auto got = db.DetectRelocalizationCandidates(q);
    auto exp = ref_reloc(orc, g[1], q2, &nums);
    if (!same(db.LastNumMatches(), nums, "reloc numMatches")) return 1;
    if (got.size() != exp.size() || got.empty()) { std::printf("reloc: %zu vs %zu candidates\n", got.size(), exp.size()); return 1; }
    for (size_t i = 0; i < got.size(); i++)
      if (got[i]->mnId != exp[i]->mnId) { std::printf("reloc candidate %zu differs\n", i); return 1; }
    for (size_t i = 0; i < g[0].size(); i++)
      if (g[0][i]->mRelocScore != g[1][i]->mRelocScore || g[0][i]->mnRelocQuery != 500) return 1;

    // loop detection from keyframe 5 (itself stored, like in LoopClosing), three thresholds
    int max_mp = 0;
    for (size_t min_mp : {(size_t)0, (size_t)5, (size_t)100000}) {
      nums.clear();
      nmps.clear();
      KFPtr c = db.DetectLoopCandidate(*g[0][5], min_mp);
      KFPtr e = ref_loop(orc, g[1], *g[1][5], min_mp, &nums, &nmps);
      if (!same(db.LastNumMatches(), nums, "loop numMatches")) return 1;
      if (!same(db.LastNumMapPointMatches(), nmps, "loop numMP")) return 1;
      if ((c ? c->mnId : 0) != (e ? e->mnId : 0)) { std::printf("loop candidate differs (min %zu)\n", min_mp); return 1; }
      for (int v : nmps) max_mp = std::max(max_mp, v);
    }
    if (max_mp < 20) { std::printf("test data too weak: max numMP %d\n", max_mp); return 1; }

    // the map changes, a keyframe leaves and comes back (at the end of mFrames)
    for (int t = 0; t < 2; t++) {
      g[t][4]->keys.clear();
      g[t][7]->mnLoopQuery = g[t][5]->mnId;
    }
    db.erase(g[0][2]);
    std::vector<KFPtr> live(g[1]);
    live.erase(live.begin() + 2);
    nums.clear();
    nmps.clear();
    KFPtr c = db.DetectLoopCandidate(*g[0][5], 3);
    KFPtr e = ref_loop(orc, live, *g[1][5], 3, &nums, &nmps);
    if (!same(db.LastNumMatches(), nums, "loop2 numMatches") || !same(db.LastNumMapPointMatches(), nmps, "loop2 numMP")) return 1;
    if ((c ? c->mnId : 0) != (e ? e->mnId : 0)) { std::printf("loop2 candidate differs\n"); return 1; }
    db.add(g[0][2]);
    live.push_back(g[1][2]);
    nums.clear();
    got = db.DetectRelocalizationCandidates(*g[0][2]);
    exp = ref_reloc(orc, live, *g[1][2], &nums);
    if (!same(db.LastNumMatches(), nums, "reloc2 numMatches") || got.size() != exp.size()) return 1;
    for (size_t i = 0; i < got.size(); i++)
      if (got[i]->mnId != exp[i]->mnId) return 1;
    db.clear();
    if (db.DetectLoopCandidate(*g[0][0], 0) || !db.DetectRelocalizationCandidates(*g[0][0]).empty()) return 1;
    std::printf("keyframe database ok: %zu keyframes, max numMP %d\n", g[0].size(), max_mp);
    return 0;
  } catch (const std::exception& e) {
    std::printf("constructor threw: %s\n", e.what());
    return 2;
  }
}
