// hip_keyframe_database.h -- C++ host mirror of SLAM_PIPELINE::KeyFrameMatchDatabase
// (slam_pipeline/include/KeyFrameDatabase.h:31-51, slam_pipeline/src/KeyFrameDatabase.cc:9-117) above the C ABI:
// SURVEY.md 8f row 3.
//
// The reference calls MatchFrames(query, KF_i) once per stored keyframe (re-extracting both frames every time) and
// counts on the host.  Here a keyframe is uploaded once in add() (msf_store_frame: image resident in HBM, ORB features
// extracted once), its KeyPointMap occupancy is a device bitmap (msf_set_mappoints), and a query is one
// msf_match_one_to_many call that returns two int32 per keyframe.  The selection rules applied to those counts are
// stated at the two Detect* functions; tests/cpp/test_keyframe_db.cpp and tests/test_keyframe_db_gpu.py hold them
// against the reference's behaviour (including its integer / f32 conversions).
//
// Two layers, like hip_feature_matcher.h:
//  (1) msf::HipKeyFrameMatchDatabase<KF, Frame, Traits>: OpenCV-free; Traits says how to read a keyframe
//      (tests/cpp/test_keyframe_db.cpp builds it with g++ against libmsf.so);
//  (2) with -DMSF_WITH_SLAM_PIPELINE: msf::HipSlamKeyFrameDatabase, a SLAM_PIPELINE::KeyFrameDatabase that replaces
//      `KeyFrameMatchDatabase keyFrameDatabase(&featureMatcher)` (src/main.cpp:78).  See INTEGRATION.md.
#pragma once

#include <algorithm>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <vector>

#include "hip_feature_matcher.h"

namespace msf {

// Traits a keyframe type must provide (all static):
//   ImageView Image(const Frame&)                      FrameBase::imGray
//   unsigned long Id(const Frame&)                     FrameBase::id()
//   void MapPointKeys(const KF&, std::vector<int32_t>*)   keys y*cols + x of KeyPointMap entries with a map point
//   bool IsConnected(const KF& query, const std::shared_ptr<KF>&)   query.GetConnectedKeyFrames().count(other)
//   std::vector<std::shared_ptr<KF>> BestCovisibility(KF&, int n)   GetBestCovisibilityKeyFrames(n)
//   unsigned long& LoopQuery(KF&), unsigned long& RelocQuery(KF&), float& RelocScore(KF&)
template <class KF, class Frame, class Traits>
class HipKeyFrameMatchDatabase {
 public:
  using KeyFramePtr = std::shared_ptr<KF>;

  // `matcher` is not owned (the reference passes a raw FeatureMatcher*, src/main.cpp:78); it must have been created
  // with max_batch_pairs >= the number of keyframes the database will hold.
  explicit HipKeyFrameMatchDatabase(HipMatcherBase* matcher) : m_(matcher) {
    const int n_slots = std::min(2 * m_->max_batch_pairs(), m_->max_batch_pairs() + 1);
    for (int s = n_slots - 1; s >= 1; s--) free_.push_back(s);   // slot 0 is the query's
  }

  void add(KeyFramePtr pKF) {   // KeyFrameDatabase.cc:12
    if (free_.empty()) throw std::runtime_error("HipKeyFrameMatchDatabase: more keyframes than max_batch_pairs");
    const int s = free_.back();
    const ImageView v = Traits::Image(*pKF);
    const msf_image img{v.data, v.width, v.height, v.stride};
    if (msf_store_frame(m_->handle(), s, &img) != MSF_OK) throw std::runtime_error(m_->LastError());
    free_.pop_back();
    slot_[pKF.get()] = s;
    uploaded_.erase(s);
    mFrames.push_back(pKF);
  }

  void erase(KeyFramePtr pKF) {   // KeyFrameDatabase.cc:14-19
    auto i = std::find(mFrames.begin(), mFrames.end(), pKF);
    if (i != mFrames.end()) {
      mFrames.erase(i);
      free_.push_back(slot_[pKF.get()]);
      slot_.erase(pKF.get());
    }
  }

  void clear() {   // KeyFrameDatabase.cc:21
    for (auto& f : mFrames) free_.push_back(slot_[f.get()]);
    slot_.clear();
    mFrames.clear();
  }

  // Place recognition for loop closing (replaces KeyFrameMatchDatabase::DetectLoopCandidate, KeyFrameDatabase.cc:23-53).
  // One batched query gives, per stored key frame, the match count and the number of matches that carry a map point at
  // both ends.  A key frame can be the loop candidate when it matched at all, was not already handed out for this query
  // (its LoopQuery stamp) and is not a covisibility neighbour of the query; among those the one sharing the most map
  // points wins, provided it shares more than `min_shared`.  Earlier key frames win ties.
  KeyFramePtr DetectLoopCandidate(KF& query, size_t min_shared) {
    if (!MatchAll(Traits::Image(query), &query)) return KeyFramePtr();
    const unsigned long qid = Traits::Id(query);
    int best = -1;
    size_t best_shared = min_shared;
    for (size_t i = 0; i < mFrames.size(); i++) {
      const bool usable = num_[i] != 0 && Traits::LoopQuery(*mFrames[i]) != qid && !Traits::IsConnected(query, mFrames[i]);
      const size_t shared = (size_t)num_mp_[i];
      if (usable && shared > best_shared) {
        best = (int)i;
        best_shared = shared;
      }
    }
    return best < 0 ? KeyFramePtr() : mFrames[best];
  }

  // Relocalisation (replaces KeyFrameMatchDatabase::DetectRelocalizationCandidates, KeyFrameDatabase.cc:55-117), in
  // three passes over the counts of one batched query:
  //  1. every stored key frame is stamped with the query's id and its match count (the stamp is how pass 2 knows
  //     which covisible key frames took part in THIS query);
  //  2. each key frame with at least 80 % of the best count (the product is truncated to an integer, as the reference
  //     does) forms a group with those of its ten best covisible key frames that carry the stamp: the group's weight is
  //     the f32 sum of the members' counts (own count first, then covisibility order), its representative the member
  //     with the largest count (the key frame itself on ties);
  //  3. the representatives of groups heavier than 75 % of the heaviest group are returned, each once, in group order.
  std::vector<KeyFramePtr> DetectRelocalizationCandidates(Frame& query) {
    std::vector<KeyFramePtr> result;
    if (!MatchAll(Traits::Image(query), nullptr)) return result;
    const unsigned long qid = Traits::Id(query);
    int top = 0;
    for (size_t i = 0; i < mFrames.size(); i++) {
      Traits::RelocQuery(*mFrames[i]) = qid;
      Traits::RelocScore(*mFrames[i]) = (float)num_[i];
      top = std::max(top, (int)num_[i]);
    }
    const size_t cutoff = (size_t)((float)(size_t)top * 0.8f);

    struct Group { KeyFramePtr rep; float weight; };
    std::vector<Group> groups;
    float heaviest = 0.f;
    for (size_t i = 0; i < mFrames.size(); i++) {
      if ((size_t)num_[i] < cutoff) continue;
      Group g{mFrames[i], (float)num_[i]};
      float rep_score = g.weight;
      for (const KeyFramePtr& nb : Traits::BestCovisibility(*mFrames[i], 10)) {
        if (Traits::RelocQuery(*nb) != qid) continue;      // did not take part in this query
        const float s = Traits::RelocScore(*nb);
        g.weight += s;
        if (s > rep_score) { g.rep = nb; rep_score = s; }
      }
      heaviest = std::max(heaviest, g.weight);
      groups.push_back(g);
    }
    const float keep_above = 0.75f * heaviest;
    for (const Group& g : groups)
      if (g.weight > keep_above && std::find(result.begin(), result.end(), g.rep) == result.end()) result.push_back(g.rep);
    return result;
  }

  // per-keyframe counts of the last query, in mFrames order (diagnostics and tests)
  const std::vector<int32_t>& LastNumMatches() const { return num_; }
  const std::vector<int32_t>& LastNumMapPointMatches() const { return num_mp_; }
  size_t size() const { return mFrames.size(); }

 private:
  bool SyncMap(int slot, const KF& kf) {
    keys_.clear();
    Traits::MapPointKeys(kf, &keys_);
    std::sort(keys_.begin(), keys_.end());
    auto it = uploaded_.find(slot);
    if (it != uploaded_.end() && it->second == keys_) return true;
    if (msf_set_mappoints(m_->handle(), slot, keys_.data(), (int32_t)keys_.size()) != MSF_OK) return false;
    uploaded_[slot] = keys_;
    return true;
  }

  // one launch sequence for the N MatchFrames calls of the reference loop; a failure behaves like the reference's
  // empty MatchFramesResult for every keyframe (no candidate)
  bool MatchAll(const ImageView& q, const KF* query_kf) {
    const size_t n = mFrames.size();
    num_.assign(n, 0);
    num_mp_.assign(n, 0);
    if (n == 0) return true;
    const msf_image img{q.data, q.width, q.height, q.stride};
    if (msf_store_frame(m_->handle(), 0, &img) != MSF_OK) return false;
    slots_.resize(n);
    for (size_t i = 0; i < n; i++) slots_[i] = slot_[mFrames[i].get()];
    if (query_kf) {
      if (!SyncMap(0, *query_kf)) return false;
      for (size_t i = 0; i < n; i++)
        if (!SyncMap(slots_[i], *mFrames[i])) return false;
    }
    const int rc = msf_match_one_to_many(m_->handle(), 0, (int32_t)n, slots_.data(), num_.data(),
                                         query_kf ? num_mp_.data() : nullptr, nullptr, 0);
    if (rc != MSF_OK && rc != MSF_ERR_CAPACITY) {
      num_.assign(n, 0);
      return false;
    }
    for (auto& c : num_) c = std::min(std::max(c, 0), m_->result_cap());   // what MatchFramesResult would hold
    return true;
  }

  HipMatcherBase* m_;
  std::vector<KeyFramePtr> mFrames;
  std::vector<int> free_;
  std::map<const KF*, int> slot_;
  std::map<int, std::vector<int32_t>> uploaded_;
  std::vector<int32_t> keys_, slots_, num_, num_mp_;
};

}  // namespace msf

#ifdef MSF_WITH_SLAM_PIPELINE
#include "slam_pipeline/include/KeyFrame.h"
#include "slam_pipeline/include/KeyFrameDatabase.h"

namespace msf {

struct SlamKeyFrameTraits {
  static ImageView Image(const SLAM_PIPELINE::FrameBase& f) {
    return ImageView{f.imGray.data, f.imGray.cols, f.imGray.rows, (int64_t)f.imGray.step};
  }
  static unsigned long Id(const SLAM_PIPELINE::FrameBase& f) { return f.id(); }
  static void MapPointKeys(const SLAM_PIPELINE::KeyFrame& kf, std::vector<int32_t>* keys) {
    auto& map = const_cast<SLAM_PIPELINE::KeyFrame&>(kf).mKeyPointMap;   // Begin()/End() are non-const (KeyPointMap.h:46-47)
    for (auto i = map.Begin(); i != map.End(); ++i)
      if (i->second.mapPoint) keys->push_back(i->first);                 // key = y*cols + x (KeyPointMap.cc:40)
  }
  static bool IsConnected(const SLAM_PIPELINE::KeyFrame& q, const SLAM_PIPELINE::KeyFramePtr& o) {
    return q.GetConnectedKeyFrames().count(o) != 0;
  }
  static std::vector<SLAM_PIPELINE::KeyFramePtr> BestCovisibility(SLAM_PIPELINE::KeyFrame& kf, int n) {
    return kf.GetBestCovisibilityKeyFrames(n);
  }
  static unsigned long& LoopQuery(SLAM_PIPELINE::KeyFrame& kf) { return kf.mnLoopQuery; }
  static unsigned long& RelocQuery(SLAM_PIPELINE::KeyFrame& kf) { return kf.mnRelocQuery; }
  static float& RelocScore(SLAM_PIPELINE::KeyFrame& kf) { return kf.mRelocScore; }
};

// Replaces SLAM_PIPELINE::KeyFrameMatchDatabase (src/main.cpp:78); `Matcher` is HipOrbMatcher or HipLoftrMatcher.
class HipSlamKeyFrameDatabase : public SLAM_PIPELINE::KeyFrameDatabase {
 public:
  explicit HipSlamKeyFrameDatabase(HipMatcherBase* matcher) : db_(matcher) {}
  void add(SLAM_PIPELINE::KeyFramePtr pKF) override { db_.add(pKF); }
  void erase(SLAM_PIPELINE::KeyFramePtr pKF) override { db_.erase(pKF); }
  void clear() override { db_.clear(); }
  SLAM_PIPELINE::KeyFramePtr DetectLoopCandidate(SLAM_PIPELINE::KeyFrame& pKF, size_t minNumMPMatches) override {
    return db_.DetectLoopCandidate(pKF, minNumMPMatches);
  }
  std::vector<SLAM_PIPELINE::KeyFramePtr> DetectRelocalizationCandidates(SLAM_PIPELINE::FrameBase& pF) override {
    return db_.DetectRelocalizationCandidates(pF);
  }

 private:
  HipKeyFrameMatchDatabase<SLAM_PIPELINE::KeyFrame, SLAM_PIPELINE::FrameBase, SlamKeyFrameTraits> db_;
};

}  // namespace msf
#endif  // MSF_WITH_SLAM_PIPELINE
