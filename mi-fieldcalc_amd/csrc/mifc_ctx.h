// mifc_ctx.h -- internals shared by the host-side translation units of the C
// ABI (mifc_capi.hip: hot path and stencil family; mifc_capi_catalogue.hip: the
// rest of the pointwise catalogue and the ensemble reductions).  Not installed.
#ifndef MIFC_CTX_H
#define MIFC_CTX_H

#include "../../include/mifc.h"
#include "mifc_hostpipe.h"
#include "mifc_kernels.h"

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

using mifc::u64;

struct mifc_ctx
{
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  // grow-only device scratch slots for staged host fields
  static const int NSLOT = 10;
  void* slot[NSLOT] = {nullptr};
  size_t slot_bytes[NSLOT] = {0};
  // per-level flags / counters
  unsigned char* d_flags = nullptr; // 2 * cap_lev bytes (wind | thermo, or just one set)
  u64* d_counts = nullptr;          // 5 * cap_lev
  float* d_ab = nullptr;            // 2 * cap_lev (alevel | blevel; per-level scalars of the batched f1 operators)
  int* d_levels = nullptr;          // cap_lev: level lists of the batched two-stage operators (ALL_DEFINED levels first)
  unsigned int* d_partials = nullptr; // per-workgroup undefined counts of big one-shot launches (grow-only), see mifc_device.h
  size_t partials_cap = 0;
  void* h_pinned = nullptr;         // pinned mirror: counts (3*cap u64) + flags (2*cap) + ab (2*cap float)
  size_t cap_lev = 0;
  // recorded after every async copy that READS the pinned mirror (enqueue
  // variants); waited on before the mirror is rewritten
  hipEvent_t pinned_read = nullptr;
  bool pinned_read_pending = false;
  // recorded after every asynchronous launch that READS the context-owned device scratch
  // (d_flags, d_ab): a switch to another stream makes that stream wait on it before the scratch
  // can be rewritten there (the *_enqueue entry points never synchronise)
  hipEvent_t scratch_read = nullptr;
  bool scratch_read_pending = false;
  // RCCL communicator of the row-slab path (mifc_comm_init / mifc_comm_adopt; mifc_slab.hip): ncclComm_t, its size and
  // this process' rank in it; owned = created by the library (destroyed with the context)
  void* comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  bool comm_owned = false;
  // mifc_counts_accumulate: the *_enqueue entries add to the caller's counters instead of zeroing them first
  bool counts_accumulate = false;
  // mifc_graph_begin .. mifc_graph_end: the *_enqueue calls in between are recorded on capture_stream instead of running
  bool capturing = false;
  hipStream_t capture_stream = nullptr, stream_before_capture = nullptr;
  // lanes of a capture: independent calls recorded side by side (mifc_graph_lane); lane 0 is capture_stream
  std::vector<hipStream_t> lane_streams;
  std::vector<hipEvent_t> lane_events; // [0] fork, [k] join of lane k
  int n_lanes = 1, lane = 0;
  // chunked, full-duplex streaming of host-resident level batches (created on first use)
  mifc::HostPipe* pipe = nullptr;
  // host fields the caller declared constant (mifc_hold_field): device copies that stage_in reuses
  struct HeldField
  {
    const float* host;
    size_t n;
    float* dev;
  };
  std::vector<HeldField> held;
  // measurement aid (mifc_timing_begin / mifc_timing_end_ms): HIP event pairs around every
  // kernel launch of the calls in between, on the stream the kernels are launched on
  static const int NTIMED = 16;
  bool timing = false;
  int n_timed = 0;
  hipEvent_t tev[2 * NTIMED] = {nullptr};
};

namespace mifc_host {

bool fail(mifc_ctx* c, const char* what, hipError_t e);

// Start of every entry point: forget the last error and make the context's device current
// for the calling thread (a process may hold contexts on several GPUs).
inline void enter(mifc_ctx* c)
{
  c->err.clear();
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != c->device)
    (void)hipSetDevice(c->device);
}

#define MIFC_HIP(c, call)                    \
  do {                                       \
    hipError_t e_ = (call);                  \
    if (e_ != hipSuccess) {                  \
      mifc_host::fail((c), #call, e_);       \
      return 0;                              \
    }                                        \
  } while (0)

// launches wrapped in event pairs while a timing section is open (mifc_timing_begin: the measurement build only)
#ifdef MIFC_MEASUREMENT_BUILD
#define MIFC_LAUNCH(c, call)                                                      \
  do {                                                                            \
    const bool timed_ = (c)->timing && (c)->n_timed < mifc_ctx::NTIMED;           \
    if (timed_)                                                                   \
      (void)hipEventRecord((c)->tev[2 * (c)->n_timed], (c)->stream);              \
    MIFC_HIP(c, call);                                                            \
    if (timed_) {                                                                 \
      (void)hipEventRecord((c)->tev[2 * (c)->n_timed + 1], (c)->stream);          \
      (c)->n_timed += 1;                                                          \
    }                                                                             \
  } while (0)
#else
#define MIFC_LAUNCH(c, call) MIFC_HIP(c, call)
#endif

bool ensure_slot(mifc_ctx* c, int s, size_t bytes);
bool ensure_levels(mifc_ctx* c, size_t nlev);
// scratch for the per-workgroup counts of a counted one-shot launch over n cells (nullptr below the size where it pays)
unsigned int* partials_for(mifc_ctx* c, size_t n_cells, int* cap);
bool pinned_acquire(mifc_ctx* c);
bool pinned_release(mifc_ctx* c);
bool scratch_release(mifc_ctx* c);
u64* pinned_counts(mifc_ctx* c);
unsigned char* pinned_flags(mifc_ctx* c);
float* pinned_ab(mifc_ctx* c);
// Brings a field to the device if the caller handed a host pointer (slot s of the context's scratch).
const float* stage_in(mifc_ctx* c, int s, const float* p, size_t n, int memkind, bool* ok);
float* stage_out(mifc_ctx* c, int s, float* p, size_t n, int memkind, bool* ok, bool preload = false);
bool fetch_out(mifc_ctx* c, int s, float* p, size_t n, int memkind);

// MetConstants.h:43-53 (host copies, evaluated like the reference does on the CPU)
const float K_CP = 1004.f, K_T0 = 273.15f;
const float K_P0INV = (float)(1. / 1000.0);
const float K_KAPPA = 287.f / 1004.f;

inline bool bad_hlevel(float a, float b) // FieldCalculations.cc:298-301
{
  return (a < 0.0) || (b < 0.0) || (a == 0.0 && b == 0.0) || (b > 1.0);
}

} // namespace mifc_host

#endif // MIFC_CTX_H
