// Multi-device sharding of the batched MatchFrames call (include/msf_abi.h, section "multi-device").
// Built on the public single-device entry points only: one msf_handle per shard.  Pairs are independent units
// (SURVEY.md section 8e), so there is no exchange step: every shard writes its block of the caller's output arrays.
// Shard 0 runs on the calling thread; every other shard has ONE worker thread that lives as long as the msf_multi
// (started by msf_multi_create, parked on a condition variable between calls) -- a call costs two notifications per
// shard instead of a thread start and join.
#include "msf_abi.h"

#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// one parked thread; run(job) hands it a job, wait() returns when the job is done
class Worker {
 public:
  Worker() : th_([this] { loop(); }) {}
  ~Worker() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    if (th_.joinable()) th_.join();
  }
  Worker(const Worker&) = delete;
  Worker& operator=(const Worker&) = delete;
  void run(const std::function<void()>* job) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = job;
      busy_ = true;
    }
    cv_.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [this] { return !busy_; });
  }

 private:
  void loop() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      cv_.wait(lk, [this] { return quit_ || job_ != nullptr; });
      if (quit_) return;
      const std::function<void()>* j = job_;
      job_ = nullptr;
      lk.unlock();
      try {
        (*j)();          // the jobs only call C entry points, which do not throw; belt and braces
      } catch (...) {
      }
      lk.lock();
      busy_ = false;
      cv_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  const std::function<void()>* job_ = nullptr;
  bool busy_ = false, quit_ = false;
  std::thread th_;      // last member: the thread starts when everything above exists
};

}  // namespace

struct msf_multi {
  std::vector<msf_handle*> shard;
  std::vector<int32_t> device;
  std::vector<std::unique_ptr<Worker>> worker;   // worker[r - 1] serves shard r
  std::mutex mu;
  std::string err;
  ~msf_multi() {
    worker.clear();                              // join the threads before their handles go
    for (msf_handle* h : shard)
      if (h) msf_destroy(h);
  }
};

namespace {
thread_local std::string g_multi_create_error;

int mfail(msf_multi* m, int code, const std::string& msg) {
  if (m) m->err = msg; else g_multi_create_error = msg;
  return code;
}

int mexception(msf_multi* m, const char* where) noexcept {
  try {
    mfail(m, MSF_ERR_HIP, std::string(where) + ": host exception (out of memory, or no thread could be started)");
  } catch (...) {
  }
  return MSF_ERR_HIP;
}

// every shard's job on its own thread (shard 0 here), then the merged status: MSF_OK, MSF_ERR_CAPACITY if that is the
// only failure of any shard, else the first hard error; the message names the shard
int run_shards(msf_multi* m, const std::function<void(int)>& job, std::vector<int>& rc) {
  const int G = (int)m->shard.size();
  std::vector<std::function<void()>> jobs((size_t)G);
  for (int r = 1; r < G; r++) {
    jobs[r] = [&job, r] { job(r); };
    m->worker[r - 1]->run(&jobs[r]);
  }
  job(0);
  for (int r = 1; r < G; r++) m->worker[r - 1]->wait();
  int result = MSF_OK;
  for (int r = 0; r < G; r++) {
    if (rc[r] == MSF_OK) continue;
    // a hard error wins over MSF_ERR_CAPACITY (every list is still complete up to its capacity then)
    if (result == MSF_OK || (result == MSF_ERR_CAPACITY && rc[r] != MSF_ERR_CAPACITY)) {
      result = rc[r];
      const char* e = msf_last_error(m->shard[r]);
      m->err = "shard " + std::to_string(r) + " (device " + std::to_string(m->device[r]) + "): " + (e ? e : "error");
    }
  }
  return result;
}
}  // namespace

extern "C" {

int msf_multi_create(const msf_config* cfg, int32_t n_devices, const int32_t* device_ids, msf_multi** out) {
  if (out) *out = nullptr;
  try {
    if (!cfg || !out || n_devices < 1 || n_devices > 64)
      return mfail(nullptr, MSF_ERR_INVALID_ARG, "msf_multi_create: bad argument (1 <= n_devices <= 64)");
    std::unique_ptr<msf_multi> m(new msf_multi());     // whatever leaves this function early frees handles and threads
    m->shard.reserve((size_t)n_devices);
    m->device.reserve((size_t)n_devices);
    m->worker.reserve((size_t)n_devices);
    for (int i = 0; i < n_devices; i++) {
      msf_config c = *cfg;
      c.device = device_ids ? device_ids[i] : i;
      msf_handle* h = nullptr;
      const int rc = msf_create(&c, &h);
      if (rc != MSF_OK) {
        const char* e = msf_last_error(nullptr);
        return mfail(nullptr, rc, "msf_multi_create: device " + std::to_string(c.device) + ": " + (e ? e : "msf_create failed"));
      }
      m->shard.push_back(h);                            // cannot throw: capacity reserved above
      m->device.push_back(c.device);
      if (i > 0) m->worker.emplace_back(new Worker());
    }
    *out = m.release();
    return MSF_OK;
  } catch (...) {
    return mexception(nullptr, "msf_multi_create");
  }
}

void msf_multi_destroy(msf_multi* m) { delete m; }

int32_t msf_multi_device_count(const msf_multi* m) { return m ? (int32_t)m->shard.size() : 0; }

msf_handle* msf_multi_handle(msf_multi* m, int32_t shard) {
  return (m && shard >= 0 && shard < (int32_t)m->shard.size()) ? m->shard[shard] : nullptr;
}

const char* msf_multi_last_error(const msf_multi* m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }

int msf_multi_set_threshold(msf_multi* m, float value) {
  try {
    if (!m) return MSF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    for (size_t i = 0; i < m->shard.size(); i++) {
      const int rc = msf_set_threshold(m->shard[i], value);
      if (rc != MSF_OK) return mfail(m, rc, std::string("msf_multi_set_threshold: ") + msf_last_error(m->shard[i]));
    }
    return MSF_OK;
  } catch (...) {
    return mexception(m, "msf_multi_set_threshold");
  }
}

void msf_multi_shard_range(int32_t n_pairs, int32_t n_shards, int32_t shard, int32_t* first, int32_t* count) {
  // contiguous blocks of ceil(n / G) pairs: the partition of gather.shard_pairs and bench.py
  int32_t f = 0, c = 0;
  if (n_pairs > 0 && n_shards > 0 && shard >= 0 && shard < n_shards) {
    const int64_t per = ((int64_t)n_pairs + n_shards - 1) / n_shards;
    const int64_t lo = per * shard < n_pairs ? per * shard : n_pairs;
    const int64_t hi = per * (shard + 1) < n_pairs ? per * (shard + 1) : n_pairs;
    f = (int32_t)lo;
    c = (int32_t)(hi - lo);
  }
  if (first) *first = f;
  if (count) *count = c;
}

int msf_multi_match_batch(msf_multi* m, int32_t n_pairs, const msf_image* a, const msf_image* b, msf_match* out,
                          int32_t cap_per_pair, int32_t* n_out) {
  try {
    if (!m) return MSF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    if (n_pairs < 0 || !a || !b || !out || !n_out || cap_per_pair < 1)
      return mfail(m, MSF_ERR_INVALID_ARG, "msf_multi_match_batch: bad argument");
    const int G = (int)m->shard.size();
    std::vector<int> rc((size_t)G, MSF_OK);
    const std::function<void(int)> job = [&](int r) {
      int32_t first = 0, count = 0;
      msf_multi_shard_range(n_pairs, G, r, &first, &count);
      if (count > 0)
        rc[r] = msf_match_batch(m->shard[r], count, a + first, b + first, out + (size_t)first * cap_per_pair, cap_per_pair,
                                n_out + first);
    };
    return run_shards(m, job, rc);
  } catch (...) {
    return mexception(m, "msf_multi_match_batch");
  }
}

int msf_multi_match_batch_device(msf_multi* m, const int32_t* n_pairs, const uint8_t* const* d_a, const uint8_t* const* d_b,
                                 int64_t frame_stride, int64_t row_stride, msf_match* const* d_out, int32_t cap_per_pair,
                                 int32_t* const* d_n_out) {
  try {
    if (!m) return MSF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    if (!n_pairs || !d_a || !d_b || !d_out || !d_n_out || cap_per_pair < 1)
      return mfail(m, MSF_ERR_INVALID_ARG, "msf_multi_match_batch_device: bad argument");
    const int G = (int)m->shard.size();
    std::vector<int> rc((size_t)G, MSF_OK);
    const std::function<void(int)> job = [&](int r) {
      // stream NULL: the shard's own stream, synchronised before the call returns
      if (n_pairs[r] > 0)
        rc[r] = msf_match_batch_device(m->shard[r], n_pairs[r], d_a[r], d_b[r], frame_stride, row_stride, d_out[r], cap_per_pair,
                                       d_n_out[r], nullptr);
    };
    return run_shards(m, job, rc);
  } catch (...) {
    return mexception(m, "msf_multi_match_batch_device");
  }
}

}  // extern "C"
