// ggs_exchange.hpp -- host side of the multi-GPU exchange (include/ggs_hip.h, "multi-GPU"): the three collectives of a
// doc-sharded sweep behind one small table of function pointers, with three providers:
//   RCCL        ncclReduceScatter / ncclAllGather on the handle's stream (librccl.so.1 dlopen'ed on first use, so that
//               libggs_hip.so itself has no link-time dependency on it and shares the copy a host process already holds)
//   callbacks   the caller's transport (ggs_attach_exchange)
//   null        local copies standing in for the peers: a timing aid (ggs_attach_null_exchange)
//
// What is exchanged replaces the reference's thread-shared merge: updateCounts/updateTopics (UPLDA:1107-1221), the
// topic batches of samplePhi (GGS:139-171, EvenSplitTopicBatchBuilder.java:28-39) and, structurally, ADLDA's
// sumTypeTopicCounts + copy-back (ADLDA.java:302-332).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/ggs_hip.h"

namespace ggs {

struct RcclApi {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclReduceScatter) ReduceScatter = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclSend) Send = nullptr;                    // optional: the sparse count exchange (all_to_all_v_i32)
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;           // optional: what the communicator itself says (ggs_get_exchange_provider)
  decltype(&ncclCommUserRank) CommUserRank = nullptr;

  // nullptr + err when librccl cannot be loaded
  static RcclApi *get(std::string &err) {
    static RcclApi api;
    static std::string load_err;
    static std::once_flag once;
    std::call_once(once, [] {
      const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
      for (const char *n : names)
        if ((api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
      if (!api.lib) { load_err = std::string("cannot load librccl.so.1: ") + dlerror(); return; }
      auto sym = [&](const char *name) {
        void *s = dlsym(api.lib, name);
        if (!s && load_err.empty()) load_err = std::string("librccl lacks ") + name;
        return s;
      };
      api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
      api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
      api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
      api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
      api.ReduceScatter = reinterpret_cast<decltype(api.ReduceScatter)>(sym("ncclReduceScatter"));
      api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
      api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
      api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
      api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
      api.Send = reinterpret_cast<decltype(api.Send)>(dlsym(api.lib, "ncclSend"));
      api.Recv = reinterpret_cast<decltype(api.Recv)>(dlsym(api.lib, "ncclRecv"));
      api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.lib, "ncclCommCount"));
      api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(dlsym(api.lib, "ncclCommUserRank"));
    });
    if (!load_err.empty()) { err = load_err; return nullptr; }
    return &api;
  }
};

struct Exchange {
  int32_t rank = 0, nranks = 1;
  ggs_exchange_ops ops{};          // what the sweep calls; ctx = this for the built-in providers
  ncclComm_t comm = nullptr;       // RCCL provider
  bool own_comm = false;
  RcclApi *api = nullptr;
  bool is_null = false;            // the timing aid (ggs_attach_null_exchange)
  std::string err;                 // text of the last failed collective
};

namespace xops {

inline int rccl_fail(Exchange *x, ncclResult_t r, const char *what) {
  x->err = std::string(what) + ": " + (x->api && x->api->GetErrorString ? x->api->GetErrorString(r) : "rccl error");
  return 1;
}
inline int rccl_reduce_scatter_i32(void *ctx, const void *send, void *recv, int64_t recv_count, void *stream) {
  auto *x = static_cast<Exchange *>(ctx);
  const ncclResult_t r = x->api->ReduceScatter(send, recv, (size_t)recv_count, ncclInt32, ncclSum, x->comm, static_cast<hipStream_t>(stream));
  return r == ncclSuccess ? 0 : rccl_fail(x, r, "ncclReduceScatter");
}
inline int rccl_all_gather_f64(void *ctx, const void *send, void *recv, int64_t send_count, void *stream) {
  auto *x = static_cast<Exchange *>(ctx);
  const ncclResult_t r = x->api->AllGather(send, recv, (size_t)send_count, ncclFloat64, x->comm, static_cast<hipStream_t>(stream));
  return r == ncclSuccess ? 0 : rccl_fail(x, r, "ncclAllGather(f64)");
}
inline int rccl_all_gather_i32(void *ctx, const void *send, void *recv, int64_t send_count, void *stream) {
  auto *x = static_cast<Exchange *>(ctx);
  const ncclResult_t r = x->api->AllGather(send, recv, (size_t)send_count, ncclInt32, x->comm, static_cast<hipStream_t>(stream));
  return r == ncclSuccess ? 0 : rccl_fail(x, r, "ncclAllGather(i32)");
}

// all-to-all of variable blocks: one ncclSend + ncclRecv per peer inside a group (the own block: a device copy)
inline int rccl_all_to_all_v_i32(void *ctx, const void *send, const int64_t *soff, const int64_t *scnt, void *recv, const int64_t *roff,
                                 const int64_t *rcnt, void *stream) {
  auto *x = static_cast<Exchange *>(ctx);
  auto st = static_cast<hipStream_t>(stream);
  const int32_t *sp = static_cast<const int32_t *>(send);
  int32_t *rp = static_cast<int32_t *>(recv);
  if (scnt[x->rank] != rcnt[x->rank]) { x->err = "all_to_all_v: own block counts differ"; return 1; }
  if (scnt[x->rank] > 0 &&
      hipMemcpyAsync(rp + roff[x->rank], sp + soff[x->rank], (size_t)scnt[x->rank] * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) { x->err = "all_to_all_v: own block copy"; return 1; }
  if (x->nranks == 1) return 0;
  ncclResult_t r = x->api->GroupStart();
  for (int p = 0; p < x->nranks && r == ncclSuccess; ++p) {
    if (p == x->rank) continue;
    if (scnt[p] > 0) r = x->api->Send(sp + soff[p], (size_t)scnt[p], ncclInt32, p, x->comm, st);
    if (r == ncclSuccess && rcnt[p] > 0) r = x->api->Recv(rp + roff[p], (size_t)rcnt[p], ncclInt32, p, x->comm, st);
  }
  const ncclResult_t e = x->api->GroupEnd();
  if (r == ncclSuccess) r = e;
  return r == ncclSuccess ? 0 : rccl_fail(x, r, "ncclSend/ncclRecv (all_to_all_v)");
}
// the timing aid: the block this rank addresses to itself stands in for every peer's (the null all-gather of the pair
// counts has told the caller to expect exactly that many elements from each)
inline int null_all_to_all_v_i32(void *ctx, const void *send, const int64_t *soff, const int64_t *scnt, void *recv, const int64_t *roff,
                                 const int64_t *rcnt, void *stream) {
  auto *x = static_cast<Exchange *>(ctx);
  const int me = x->rank;
  for (int s = 0; s < x->nranks; ++s) {
    const int64_t n = rcnt[s] < scnt[me] ? rcnt[s] : scnt[me];
    if (n > 0 && hipMemcpyAsync(static_cast<int32_t *>(recv) + roff[s], static_cast<const int32_t *>(send) + soff[me], (size_t)n * 4, hipMemcpyDeviceToDevice,
                                static_cast<hipStream_t>(stream)) != hipSuccess)
      return 1;
  }
  return 0;
}

// the timing aid: this rank's own contribution stands in for every peer's
inline int null_reduce_scatter_i32(void *ctx, const void *send, void *recv, int64_t recv_count, void *stream) {
  auto *x = static_cast<Exchange *>(ctx);
  return hipMemcpyAsync(recv, static_cast<const int32_t *>(send) + (size_t)x->rank * recv_count, (size_t)recv_count * 4, hipMemcpyDeviceToDevice,
                        static_cast<hipStream_t>(stream)) == hipSuccess ? 0 : 1;
}
template <typename T>
inline int null_all_gather(void *ctx, const void *send, void *recv, int64_t send_count, void *stream) {
  auto *x = static_cast<Exchange *>(ctx);
  for (int r = 0; r < x->nranks; ++r)
    if (hipMemcpyAsync(static_cast<T *>(recv) + (size_t)r * send_count, send, (size_t)send_count * sizeof(T), hipMemcpyDeviceToDevice,
                       static_cast<hipStream_t>(stream)) != hipSuccess)
      return 1;
  return 0;
}

}  // namespace xops

// the topic slices: sizes K/n + (K % n > r), the rule of EvenSplitTopicBatchBuilder.java:28-39 with one batch per rank
inline std::vector<int32_t> topic_slices(int32_t K, int32_t n) {
  std::vector<int32_t> b((size_t)n + 1, 0);
  const int32_t size = K / n, rem = K % n;
  for (int32_t r = 0; r < n; ++r) b[(size_t)r + 1] = b[(size_t)r] + size + (rem > r ? 1 : 0);
  return b;
}

}  // namespace ggs
