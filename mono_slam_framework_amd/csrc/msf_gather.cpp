// Product-side gather of the match lists of a multi-process job (include/msf_abi.h
// "multi-process gather"): one rank per GPU, pairs sharded with no data-path collective; the one exchange step is this
// gather of the packed lists (msf_pack_matches_device) to rank 0 -- ncclAllGather of the per-pair offsets, then
// exact-size ncclSend / ncclRecv inside one group call, over xGMI under RCCL.  Never an all-reduce.
//
// RCCL is bound at first use (dlopen), not at link time: a single-GPU user of libmsf.so needs no librccl, and a process
// that already holds an RCCL (torch.distributed's) shares that one instead of loading a second.
//
// Status: the offset / placement logic (msf_gather_plan) is unit-tested on the CPU; communicator set-up, the
// all-gather and the self-copy have run on ONE MI355X with a communicator of one rank (tests/test_gather_rccl_gpu.py).
// The send / recv leg has never executed on more than one GPU (no multi-GPU box is available to the build).
#include "msf_abi.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

namespace {

// the handful of RCCL entry points used, with the types of rccl.h (ncclResult_t = int, ncclDataType_t ncclInt32 = 2,
// ncclUniqueId = 128 opaque bytes passed BY VALUE)
struct UniqueId { char internal[128]; };
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommAbort)(void*) = nullptr;   // optional
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string err;
};
constexpr int kNcclInt32 = 2;

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // MSF_RCCL_LIBRARY names the one library to bind instead (a site's own RCCL build; tests point it at a name that
    // does not exist to take the not-found path)
    const char* over = getenv("MSF_RCCL_LIBRARY");
    const char* names[] = {over && *over ? over : "librccl.so", over && *over ? over : "librccl.so.1"};
    for (const char* n : names)                       // an RCCL this process already holds (torch.distributed) first
      if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    for (const char* n : names)
      if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) {
      const char* de = dlerror();                     // ONE call: dlerror() clears the message it returns
      r.err = std::string(names[0]) + " not found: " + (de ? de : "(no dlerror text)");
      return;
    }
    auto sym = [&](const char* s) { void* p = dlsym(r.lib, s); if (!p && r.err.empty()) r.err = std::string("RCCL symbol missing: ") + s; return p; };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.lib, "ncclCommAbort"));
  });
  return &r;
}

thread_local std::string g_gather_create_error;

}  // namespace

struct msf_gather {
  int device = 0, rank = 0, n_ranks = 1, pairs = 0;
  long long cap = 0;
  void* comm = nullptr;
  int32_t* h_offs = nullptr;          // pinned [n_ranks * (pairs + 1)]
  std::vector<int64_t> first;
  std::mutex mu;
  std::string err;
  bool dead = false;                  // a collective failed on this rank: the communicator was aborted, only destroy is left
};

namespace {
int gfail(msf_gather* g, int code, const std::string& msg) {
  if (g) g->err = msg; else g_gather_create_error = msg;
  return code;
}
int nccl_fail(msf_gather* g, const char* what, int rc) {
  Rccl* r = rccl();
  return gfail(g, MSF_ERR_HIP, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(rc) : "RCCL error"));
}
// A failure on ONE rank between the all-gather and the end of the group call leaves its peers inside their send / recv:
// the communicator is aborted (ncclCommAbort, where the library has it) so that they return with an error instead of
// waiting for ever, and the object only accepts msf_gather_destroy from then on (include/msf_abi.h, "failure semantics").
int hard_fail(msf_gather* g, int code) {
  g->dead = true;
  Rccl* r = rccl();
  if (g->comm && r->CommAbort) { r->CommAbort(g->comm); g->comm = nullptr; }
  return code;
}
}  // namespace

extern "C" {

int msf_gather_plan(int32_t n_ranks, int32_t pairs_per_rank, const int32_t* all_offsets, int64_t cap_records,
                    int32_t* totals, int64_t* recv_first) {
  if (n_ranks < 1 || pairs_per_rank < 0 || !all_offsets || !totals || !recv_first || cap_records < 0) return MSF_ERR_INVALID_ARG;
  int64_t at = 0;
  int rc = MSF_OK;
  for (int r = 0; r < n_ranks; r++) {
    const int32_t* o = all_offsets + (size_t)r * (pairs_per_rank + 1);
    // offsets of one rank: 0 = o[0] <= o[1] <= ... <= o[P] = its total (msf_pack_matches_device)
    if (o[0] != 0) return MSF_ERR_INVALID_ARG;
    for (int p = 0; p < pairs_per_rank; p++)
      if (o[p + 1] < o[p]) return MSF_ERR_INVALID_ARG;
    totals[r] = o[pairs_per_rank];
    recv_first[r] = at;                         // rank order, densely packed: pair p of rank r starts at recv_first[r] + o[p]
    if ((int64_t)totals[r] > cap_records) rc = MSF_ERR_CAPACITY;
    at += totals[r];
  }
  return rc;
}

int msf_gather_unique_id(uint8_t* id128) {
  try {
    if (!id128) return gfail(nullptr, MSF_ERR_INVALID_ARG, "msf_gather_unique_id: null");
    Rccl* r = rccl();
    if (!r->err.empty()) return gfail(nullptr, MSF_ERR_HIP, r->err);
    UniqueId u;
    const int rc = r->GetUniqueId(&u);
    if (rc != 0) return nccl_fail(nullptr, "ncclGetUniqueId", rc);
    std::memcpy(id128, u.internal, 128);
    return MSF_OK;
  } catch (...) {
    return gfail(nullptr, MSF_ERR_HIP, "msf_gather_unique_id: host exception");
  }
}

int msf_gather_create(int32_t device, int32_t rank, int32_t n_ranks, const uint8_t* id128, int32_t pairs_per_rank,
                      int64_t cap_records, msf_gather** out) {
  if (out) *out = nullptr;
  try {
    if (!out || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks || pairs_per_rank < 1 || cap_records < 1)
      return gfail(nullptr, MSF_ERR_INVALID_ARG, "msf_gather_create: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      return gfail(nullptr, MSF_ERR_HIP, "msf_gather_create: no HIP device");
    if (device < 0 || device >= ndev) return gfail(nullptr, MSF_ERR_INVALID_ARG, "msf_gather_create: device ordinal out of range");
    Rccl* r = rccl();
    if (!r->err.empty()) return gfail(nullptr, MSF_ERR_HIP, r->err);
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return gfail(nullptr, MSF_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    msf_gather* g = new (std::nothrow) msf_gather();
    if (!g) return gfail(nullptr, MSF_ERR_HIP, "out of host memory");
    struct Guard { msf_gather* g; ~Guard() { if (g) msf_gather_destroy(g); } } guard{g};
    g->device = device; g->rank = rank; g->n_ranks = n_ranks; g->pairs = pairs_per_rank; g->cap = cap_records;
    g->first.resize((size_t)n_ranks);
    if ((e = hipHostMalloc(&g->h_offs, (size_t)n_ranks * (pairs_per_rank + 1) * sizeof(int32_t), hipHostMallocDefault)) != hipSuccess)
      return gfail(nullptr, MSF_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    UniqueId u;
    std::memcpy(u.internal, id128, 128);
    const int rc = r->CommInitRank(&g->comm, n_ranks, u, rank);
    if (rc != 0) return nccl_fail(nullptr, "ncclCommInitRank", rc);
    guard.g = nullptr;
    *out = g;
    return MSF_OK;
  } catch (...) {
    return gfail(nullptr, MSF_ERR_HIP, "msf_gather_create: host exception");
  }
}

void msf_gather_destroy(msf_gather* g) {
  if (!g) return;
  hipSetDevice(g->device);
  if (g->comm && rccl()->CommDestroy) rccl()->CommDestroy(g->comm);   // (an aborted communicator is already gone)
  if (g->h_offs) hipHostFree(g->h_offs);
  delete g;
}

const char* msf_gather_last_error(const msf_gather* g) { return g ? g->err.c_str() : g_gather_create_error.c_str(); }

int msf_gather_matches_device(msf_gather* g, const msf_match* d_packed, const int32_t* d_offsets, int32_t* d_all_offsets,
                              msf_match* d_recv, int32_t* totals, void* stream) {
  try {
    if (!g) return MSF_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(g->mu);
    if (!d_packed || !d_offsets || !d_all_offsets || !totals || (g->rank == 0 && !d_recv))
      return gfail(g, MSF_ERR_INVALID_ARG, "msf_gather_matches_device: null argument");
    if (g->dead) return gfail(g, MSF_ERR_HIP, "msf_gather_matches_device: an earlier call failed; destroy this gather object");
    Rccl* r = rccl();
    hipError_t e = hipSetDevice(g->device);
    if (e != hipSuccess) return gfail(g, MSF_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    hipStream_t st = (hipStream_t)stream;
    const size_t n_off = (size_t)g->pairs + 1;
    int rc = r->AllGather(d_offsets, d_all_offsets, n_off, kNcclInt32, g->comm, st);
    if (rc != 0) return hard_fail(g, nccl_fail(g, "ncclAllGather(offsets)", rc));
    // exact-size send / recv need the totals on the host: one small copy + wait per step (a padded all-gather of
    // capacity-sized lists would move ~10x the bytes over xGMI instead)
    if ((e = hipMemcpyAsync(g->h_offs, d_all_offsets, (size_t)g->n_ranks * n_off * sizeof(int32_t), hipMemcpyDeviceToHost, st)) != hipSuccess ||
        (e = hipStreamSynchronize(st)) != hipSuccess)
      return hard_fail(g, gfail(g, MSF_ERR_HIP, std::string("offsets to host: ") + hipGetErrorString(e)));
    const int plan = msf_gather_plan(g->n_ranks, g->pairs, g->h_offs, g->cap, totals, g->first.data());
    // every rank plans from the same gathered offsets: a capacity / format error is returned by ALL ranks, none enters
    // the send / recv leg, and the object stays usable
    if (plan != MSF_OK)
      return gfail(g, plan, plan == MSF_ERR_CAPACITY ? "msf_gather_matches_device: a rank holds more records than cap_records"
                                                     : "msf_gather_matches_device: malformed offsets");
    if ((rc = r->GroupStart()) != 0) return hard_fail(g, nccl_fail(g, "ncclGroupStart", rc));
    if (g->rank == 0) {
      for (int q = 1; q < g->n_ranks && rc == 0; q++)
        if (totals[q] > 0) rc = r->Recv(d_recv + g->first[q], (size_t)totals[q] * 4, kNcclInt32, q, g->comm, st);
    } else if (totals[g->rank] > 0) {
      rc = r->Send(d_packed, (size_t)totals[g->rank] * 4, kNcclInt32, 0, g->comm, st);
    }
    const int rc_end = r->GroupEnd();
    if (rc != 0) return hard_fail(g, nccl_fail(g, "ncclSend/ncclRecv", rc));
    if (rc_end != 0) return hard_fail(g, nccl_fail(g, "ncclGroupEnd", rc_end));
    if (g->rank == 0 && totals[0] > 0 &&
        (e = hipMemcpyAsync(d_recv + g->first[0], d_packed, (size_t)totals[0] * sizeof(msf_match), hipMemcpyDeviceToDevice, st)) != hipSuccess)
      return gfail(g, MSF_ERR_HIP, std::string("own records: ") + hipGetErrorString(e));   // local copy: peers are not waiting on it
    return MSF_OK;
  } catch (...) {
    return gfail(g, MSF_ERR_HIP, "msf_gather_matches_device: host exception");
  }
}

}  // extern "C"
