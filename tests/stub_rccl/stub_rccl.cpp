// TEST INFRASTRUCTURE ONLY -- never linked or loaded by the product unless a test points MSF_RCCL_LIBRARY at it.
//
// A stand-in for the ten RCCL entry points csrc/msf_gather.cpp binds (ncclGetUniqueId, ncclCommInitRank,
// ncclCommDestroy, ncclCommAbort, ncclAllGather, ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd,
// ncclGetErrorString), moving the bytes through a POSIX shared-memory segment and hipMemcpy instead of xGMI, so that two
// PROCESSES that share one GPU can drive msf_gather_matches_device's N > 1 branch (real RCCL refuses two ranks on one
// device, and no multi-GPU box is available to the build).  What it proves: the offsets, the placement of the received
// records, the stream ordering around the call and the failure semantics of the product code.  What it does not: RCCL.
//
// Semantics kept from RCCL: collectives are ordered on the stream they are given (the stub waits for the stream before
// it reads a send buffer, and returns only when the receive buffer is written -- stronger than RCCL's asynchrony, never
// weaker); Send / Recv inside a group run at ncclGroupEnd; a peer of an aborted communicator comes back with an error
// instead of waiting for ever.  Every wait is bounded (MSF_STUB_RCCL_TIMEOUT_S, default 30 s).
//
// Fault injection: MSF_STUB_RCCL_FAIL_SEND=k makes this process's k-th ncclSend (1-based) return ncclInternalError
// before it moves anything.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <vector>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kSlot = 4u << 20;            // bytes a rank can have in flight towards one peer / in one all-gather
constexpr int kOk = 0, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4;

struct Header {
  std::atomic<uint32_t> arrived;              // ranks that have mapped the segment
  std::atomic<uint32_t> aborted;              // a rank called ncclCommAbort
  std::atomic<uint32_t> bar_count, bar_gen;   // sense-reversing barrier
  std::atomic<uint32_t> detached;
  // mailbox src -> dst: seq_full = number of chunks written, seq_empty = number consumed, len of the current chunk
  struct Box { std::atomic<uint64_t> full, empty; uint64_t len; char pad[40]; } box[kMaxRanks][kMaxRanks];
};

struct Comm {
  int rank = 0, n = 1;
  char name[64] = {};
  Header* h = nullptr;
  char* data = nullptr;                        // [n][n][kSlot] mailboxes, then [n][kSlot] all-gather slots
  size_t bytes = 0;
  bool in_group = false;
  struct Op { bool send; void* buf; size_t bytes; int peer; hipStream_t st; };
  std::vector<Op> ops;
  int sends = 0;
  char* mail(int src, int dst) { return data + ((size_t)src * n + dst) * kSlot; }
  char* ag(int r) { return data + ((size_t)n * n + r) * kSlot; }
};

double now_s() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}
double timeout_s() {
  const char* e = getenv("MSF_STUB_RCCL_TIMEOUT_S");
  return e && atof(e) > 0 ? atof(e) : 30.0;
}
// spin until pred() or abort or timeout; 0 = ok
template <class P>
int wait_for(Comm* c, P pred) {
  const double t0 = now_s(), lim = timeout_s();
  for (uint64_t it = 0;; it++) {
    if (pred()) return kOk;
    if (c->h->aborted.load(std::memory_order_acquire)) return kSystemError;
    if ((it & 1023) == 1023 && now_s() - t0 > lim) return kSystemError;
    sched_yield();
  }
}
int barrier(Comm* c) {
  Header* h = c->h;
  const uint32_t gen = h->bar_gen.load(std::memory_order_acquire);
  if (h->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->n) {
    h->bar_count.store(0, std::memory_order_relaxed);
    h->bar_gen.fetch_add(1, std::memory_order_release);
    return kOk;
  }
  return wait_for(c, [&] { return h->bar_gen.load(std::memory_order_acquire) != gen; });
}
size_t type_size(int dt) {
  switch (dt) {
    case 0: case 1: return 1;       // int8 / uint8
    case 2: case 3: case 7: return 4;   // int32 / uint32 / float32
    case 4: case 5: case 8: return 8;   // int64 / uint64 / float64
    case 6: case 9: return 2;       // float16 / bfloat16
    default: return 0;
  }
}
int do_send(Comm* c, const Comm::Op& op) {
  if (hipStreamSynchronize(op.st) != hipSuccess) return kSystemError;      // everything enqueued before the send is done
  Header::Box& b = c->h->box[c->rank][op.peer];
  const char* src = (const char*)op.buf;
  size_t left = op.bytes;
  while (left) {
    const size_t len = left < kSlot ? left : kSlot;
    const uint64_t seq = b.full.load(std::memory_order_relaxed);
    if (int rc = wait_for(c, [&] { return b.empty.load(std::memory_order_acquire) == seq; })) return rc;
    if (hipMemcpy(c->mail(c->rank, op.peer), src, len, hipMemcpyDeviceToHost) != hipSuccess) return kSystemError;
    b.len = len;
    b.full.store(seq + 1, std::memory_order_release);
    src += len;
    left -= len;
  }
  return kOk;
}
int do_recv(Comm* c, const Comm::Op& op) {
  if (hipStreamSynchronize(op.st) != hipSuccess) return kSystemError;
  Header::Box& b = c->h->box[op.peer][c->rank];
  char* dst = (char*)op.buf;
  size_t left = op.bytes;
  while (left) {
    const uint64_t seq = b.empty.load(std::memory_order_relaxed);
    if (int rc = wait_for(c, [&] { return b.full.load(std::memory_order_acquire) == seq + 1; })) return rc;
    const size_t len = b.len;
    if (len > left) return kInternalError;                                   // sender and receiver disagree on the size
    if (hipMemcpy(dst, c->mail(op.peer, c->rank), len, hipMemcpyHostToDevice) != hipSuccess) return kSystemError;
    b.empty.store(seq + 1, std::memory_order_release);
    dst += len;
    left -= len;
  }
  return kOk;
}
void detach(Comm* c, bool abort) {
  if (!c) return;
  if (c->h) {
    if (abort) c->h->aborted.store(1, std::memory_order_release);
    const uint32_t d = c->h->detached.fetch_add(1, std::memory_order_acq_rel) + 1;
    munmap(c->h, c->bytes);
    if (d == (uint32_t)c->n || abort) shm_unlink(c->name);                  // last one out (or the aborting rank) removes it
  }
  delete c;
}

struct UniqueId { char internal[128]; };
thread_local Comm* g_group_comm = nullptr;
thread_local bool g_in_group = false;

}  // namespace

extern "C" {

int ncclGetUniqueId(UniqueId* id) {
  if (!id) return kInvalidArgument;
  memset(id->internal, 0, 128);
  timespec t;
  clock_gettime(CLOCK_REALTIME, &t);
  snprintf(id->internal, 64, "/msf_stub_rccl_%d_%lld_%ld", (int)getpid(), (long long)t.tv_sec, t.tv_nsec);
  return kOk;
}

int ncclCommInitRank(void** comm, int n, UniqueId id, int rank) {
  if (!comm || n < 1 || n > kMaxRanks || rank < 0 || rank >= n || id.internal[0] != '/') return kInvalidArgument;
  Comm* c = new Comm();
  c->rank = rank;
  c->n = n;
  memcpy(c->name, id.internal, 63);
  c->bytes = sizeof(Header) + ((size_t)n * n + n) * kSlot;
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { if (fd >= 0) close(fd); delete c; return kSystemError; }
  void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { delete c; return kSystemError; }
  c->h = (Header*)p;                          // a fresh segment is zero-filled: every atomic starts at 0
  c->data = (char*)p + sizeof(Header);
  c->h->arrived.fetch_add(1, std::memory_order_acq_rel);
  if (wait_for(c, [&] { return c->h->arrived.load(std::memory_order_acquire) >= (uint32_t)n; })) { detach(c, true); return kSystemError; }
  *comm = c;
  return kOk;
}

int ncclCommDestroy(void* comm) { detach((Comm*)comm, false); return kOk; }
int ncclCommAbort(void* comm) { detach((Comm*)comm, true); return kOk; }

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t st) {
  Comm* c = (Comm*)comm;
  const size_t bytes = count * type_size(dtype);
  if (!c || !send || !recv || !bytes || bytes > kSlot) return kInvalidArgument;
  if (hipStreamSynchronize(st) != hipSuccess) return kSystemError;
  if (hipMemcpy(c->ag(c->rank), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return kSystemError;
  if (int rc = barrier(c)) return rc;
  for (int r = 0; r < c->n; r++)
    if (hipMemcpy((char*)recv + (size_t)r * bytes, c->ag(r), bytes, hipMemcpyHostToDevice) != hipSuccess) return kSystemError;
  return barrier(c);                          // nobody overwrites a slot before every rank has read it
}

int ncclGroupStart() { g_in_group = true; return kOk; }

int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t st) {
  Comm* c = (Comm*)comm;
  if (!c || !buf || peer < 0 || peer >= c->n || peer == c->rank || !type_size(dtype)) return kInvalidArgument;
  c->sends++;
  const char* f = getenv("MSF_STUB_RCCL_FAIL_SEND");
  if (f && atoi(f) == c->sends) return kInternalError;
  Comm::Op op{true, const_cast<void*>(buf), count * type_size(dtype), peer, st};
  if (g_in_group) { g_group_comm = c; c->ops.push_back(op); return kOk; }
  return do_send(c, op);
}

int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t st) {
  Comm* c = (Comm*)comm;
  if (!c || !buf || peer < 0 || peer >= c->n || peer == c->rank || !type_size(dtype)) return kInvalidArgument;
  Comm::Op op{false, buf, count * type_size(dtype), peer, st};
  if (g_in_group) { g_group_comm = c; c->ops.push_back(op); return kOk; }
  return do_recv(c, op);
}

int ncclGroupEnd() {
  g_in_group = false;
  Comm* c = g_group_comm;
  g_group_comm = nullptr;
  if (!c) return kOk;
  int rc = kOk;
  // the product's groups are one-directional per rank (rank 0 only receives, the others only send), so running the
  // queued operations in order cannot deadlock
  for (const Comm::Op& op : c->ops) {
    rc = op.send ? do_send(c, op) : do_recv(c, op);
    if (rc) break;
  }
  c->ops.clear();
  return rc;
}

const char* ncclGetErrorString(int rc) {
  switch (rc) {
    case kOk: return "stub rccl: no error";
    case kSystemError: return "stub rccl: unhandled system error (peer aborted, timed out, or a HIP call failed)";
    case kInternalError: return "stub rccl: internal error (injected, or size mismatch)";
    case kInvalidArgument: return "stub rccl: invalid argument";
    default: return "stub rccl: unknown error";
  }
}

}  // extern "C"
