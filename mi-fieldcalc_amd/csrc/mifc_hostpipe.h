// mifc_hostpipe.h -- streaming of a level batch that lives in HOST memory
// through the device (the legacy host-pointer path).
//
// A caller of the unchanged reference signatures hands over pageable host
// pointers.  Copying a whole batch in, computing, and copying it out uses the
// host link in one direction at a time (measured on the MI355X box, 1 GiB:
// 56 GB/s either way alone; pageable copies issued from two threads do NOT
// overlap, 28 GB/s each).  Copies from/to PINNED memory do run full duplex
// (48.6 GB/s each way at once), and a few host threads move pageable <-> pinned
// faster than the link (29 GB/s with one thread, 77 with four).  So the batch is
// cut into chunks of levels and pipelined:
//
//   host threads: caller's u,v --> pinned_in[b]            pinned_out[b] --> caller's outputs
//   copy engine :              pinned_in[b] --H2D--> dev_in[b]     dev_out[b] --D2H--> pinned_out[b]
//   compute     :                              kernel(dev_in[b]) -> dev_out[b]
//
// with two buffers per stage (b = chunk parity), three HIP streams and events
// between them.  profiles/r01/hostpath_probe.txt holds the link measurements.
#ifndef MIFC_HOSTPIPE_H
#define MIFC_HOSTPIPE_H

#include <hip/hip_runtime.h>

#include <cstddef>
#include <functional>
#include <string>

namespace mifc {

struct HostPipe; // pinned staging, device chunk buffers, streams, events, copy threads

HostPipe* hostpipe_create(int device);
void hostpipe_destroy(HostPipe* hp);

// Enqueues the kernel(s) for levels [l0, l0 + nl) on `stream`; d_in / d_out
// hold that chunk only (level l0 first).
typedef std::function<hipError_t(int l0, int nl, const float* const* d_in, float* const* d_out, hipStream_t stream)> ChunkLaunch;

// in[k] / out[k]: host arrays [nlev][n] (out[k] may be null: that output is not produced).
// Returns false and fills *err on a HIP error; everything is complete (outputs
// in the caller's memory) when it returns true.
bool hostpipe_run(HostPipe* hp, size_t n, int nlev, int n_in, const float* const* in, int n_out, float* const* out, const ChunkLaunch& launch,
                  std::string* err);

// how many levels go into one chunk for fields of n cells (0: batch too small to pipeline)
int hostpipe_chunk_levels(size_t n, int nlev);

} // namespace mifc

#endif // MIFC_HOSTPIPE_H
