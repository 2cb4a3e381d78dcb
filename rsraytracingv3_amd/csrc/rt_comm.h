// rt_comm.h — RCCL bound at run time (dlopen), for the multi-GPU image gather of the C-ABI (rt_comm_init).
//
// librt355.so does not link RCCL: a single-GPU host needs none, and a process that already carries an RCCL (a
// PyTorch wheel ships its own) must keep using THAT copy — two copies of RCCL in one process do not mix.  Order:
// $RT355_RCCL_LIB, a librccl already mapped into the process (/proc/self/maps), librccl.so.1 on the loader path,
// /opt/rocm/lib/librccl.so.1.  Only point-to-point calls are used: every peer sends its strips straight to rank 0
// (one xGMI link per peer, all seven concurrently), never a ring.
#pragma once

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include <rccl/rccl.h>  // types and enums only; every function is looked up with dlsym

namespace rt355 {

struct RcclApi {
  void *handle = nullptr;
  std::string path;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // required: the only way out of an exchange that never completes
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool ok() const { return handle != nullptr; }
};

inline std::string find_mapped_rccl() {
  FILE *f = std::fopen("/proc/self/maps", "r");
  if (!f) return std::string();
  char line[1024];
  std::string found;
  while (std::fgets(line, sizeof line, f)) {
    const char *slash = std::strchr(line, '/');
    if (!slash || !std::strstr(slash, "librccl.so")) continue;
    found = slash;
    while (!found.empty() && (found.back() == '\n' || found.back() == ' ')) found.pop_back();
    break;
  }
  std::fclose(f);
  return found;
}

// Loads (once per process) and returns the API table; ok() is false, with `why` filled, when no RCCL can be bound.
inline const RcclApi &rccl_api(std::string *why) {
  static RcclApi api;
  static std::string error;
  static bool tried = false;
  if (!tried) {
    tried = true;
    std::string candidates[4];
    int n = 0;
    if (const char *env = std::getenv("RT355_RCCL_LIB")) candidates[n++] = env;
    const std::string mapped = find_mapped_rccl();
    if (!mapped.empty()) candidates[n++] = mapped;
    candidates[n++] = "librccl.so.1";
    candidates[n++] = "/opt/rocm/lib/librccl.so.1";
    for (int i = 0; i < n && !api.handle; ++i) {
      void *h = dlopen(candidates[i].c_str(), RTLD_NOW | RTLD_LOCAL);
      if (!h) {
        const char *e = dlerror();  // one call: dlerror() clears the message it returns
        error += candidates[i] + ": " + (e ? e : "dlopen failed") + "; ";
        continue;
      }
      RcclApi a;
      a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
      a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
      a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
      a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(dlsym(h, "ncclCommAbort"));
      a.Send = reinterpret_cast<decltype(a.Send)>(dlsym(h, "ncclSend"));
      a.Recv = reinterpret_cast<decltype(a.Recv)>(dlsym(h, "ncclRecv"));
      a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
      a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
      a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
      if (a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.CommAbort && a.Send && a.Recv && a.GroupStart && a.GroupEnd && a.GetErrorString) {
        a.handle = h;
        a.path = candidates[i];
        api = a;
      } else {
        error += candidates[i] + ": not an RCCL (symbols missing); ";
        dlclose(h);
      }
    }
  }
  if (why) *why = error;
  return api;
}

}  // namespace rt355
