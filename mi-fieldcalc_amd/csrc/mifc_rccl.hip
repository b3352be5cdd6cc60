// mifc_rccl.hip -- see mifc_rccl.h (host code only).
#include "mifc_rccl.h"

#include <dlfcn.h>

#include <mutex>
#include <string>

namespace mifc {

namespace {
std::once_flag g_once;
RcclApi g_api;
bool g_ok = false;
std::string g_why;

template <typename F>
bool bind(void* lib, const char* name, F& slot)
{
  slot = reinterpret_cast<F>(dlsym(lib, name));
  if (!slot) {
    g_why = std::string("RCCL: symbol ") + name + " not found";
    return false;
  }
  return true;
}

void load()
{
  void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib)
    lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib)
    lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) {
    const char* e = dlerror();
    g_why = std::string("RCCL: cannot load librccl.so.1 (") + (e ? e : "?") + ")";
    return;
  }
  g_ok = bind(lib, "ncclGetUniqueId", g_api.GetUniqueId) && bind(lib, "ncclCommInitRank", g_api.CommInitRank) &&
         bind(lib, "ncclCommDestroy", g_api.CommDestroy) && bind(lib, "ncclCommCount", g_api.CommCount) &&
         bind(lib, "ncclCommUserRank", g_api.CommUserRank) && bind(lib, "ncclGetErrorString", g_api.GetErrorString) &&
         bind(lib, "ncclGroupStart", g_api.GroupStart) && bind(lib, "ncclGroupEnd", g_api.GroupEnd) && bind(lib, "ncclSend", g_api.Send) &&
         bind(lib, "ncclRecv", g_api.Recv) && bind(lib, "ncclAllReduce", g_api.AllReduce);
}
} // namespace

const RcclApi* rccl_api(const char** why)
{
  std::call_once(g_once, load);
  if (!g_ok) {
    if (why)
      *why = g_why.c_str();
    return nullptr;
  }
  return &g_api;
}

} // namespace mifc
