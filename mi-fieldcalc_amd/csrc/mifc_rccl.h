// mifc_rccl.h -- RCCL (send/recv over xGMI, all-reduce) for the row-slab path, loaded on first use.
// The library does not link librccl: a single-GPU caller never loads it.  dlopen by soname ("librccl.so.1") finds the copy a
// process has already loaded (PyTorch ships its own) -- a communicator handed in by the caller must be driven by the very
// library that created it -- and otherwise the system's (/opt/rocm/lib).
#ifndef MIFC_RCCL_H
#define MIFC_RCCL_H

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

namespace mifc {

struct RcclApi
{
  decltype(&ncclGetUniqueId) GetUniqueId;
  decltype(&ncclCommInitRank) CommInitRank;
  decltype(&ncclCommDestroy) CommDestroy;
  decltype(&ncclCommCount) CommCount;
  decltype(&ncclCommUserRank) CommUserRank;
  decltype(&ncclGetErrorString) GetErrorString;
  decltype(&ncclGroupStart) GroupStart;
  decltype(&ncclGroupEnd) GroupEnd;
  decltype(&ncclSend) Send;
  decltype(&ncclRecv) Recv;
  decltype(&ncclAllReduce) AllReduce;
};

// nullptr when the library cannot be loaded (*why then says so)
const RcclApi* rccl_api(const char** why);

} // namespace mifc

#endif // MIFC_RCCL_H
