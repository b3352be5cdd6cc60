// mifc_capi.hip -- the extern "C" boundary declared in include/mifc.h.
//
// Host-side logic only: argument validation exactly as the reference does it
// (what makes an operator `return false`), the unit/compute remaps, staging of
// legacy host pointers through device scratch, launching the HIP kernels and
// turning the per-field undefined counts into ValuesDefined flags.  There is
// no CPU compute path: every operator body runs on the GPU.

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "mifc_ctx.h"

namespace mifc_host {


bool fail(mifc_ctx* c, const char* what, hipError_t e)
{
  char buf[256];
  std::snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  if (c)
    c->err = buf;
  return false;
}


bool ensure_slot(mifc_ctx* c, int s, size_t bytes)
{
  if (c->slot_bytes[s] >= bytes)
    return true;
  if (c->slot[s]) {
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess)
      return fail(c, "hipStreamSynchronize", e);
    (void)hipFree(c->slot[s]);
    c->slot[s] = nullptr;
    c->slot_bytes[s] = 0;
  }
  const size_t want = (bytes + 255) & ~size_t(255);
  hipError_t e = hipMalloc(&c->slot[s], want);
  if (e != hipSuccess)
    return fail(c, "hipMalloc(scratch)", e);
  c->slot_bytes[s] = want;
  return true;
}

bool ensure_levels(mifc_ctx* c, size_t nlev)
{
  if (c->cap_lev >= nlev)
    return true;
  hipError_t e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess)
    return fail(c, "hipStreamSynchronize", e);
  if (c->d_flags)
    (void)hipFree(c->d_flags);
  if (c->d_counts)
    (void)hipFree(c->d_counts);
  if (c->d_ab)
    (void)hipFree(c->d_ab);
  if (c->d_levels)
    (void)hipFree(c->d_levels);
  if (c->h_pinned)
    (void)hipHostFree(c->h_pinned);
  c->d_levels = nullptr;
  c->d_flags = nullptr;
  c->d_counts = nullptr;
  c->d_ab = nullptr;
  c->h_pinned = nullptr;
  c->cap_lev = 0;
  size_t cap = 256;
  while (cap < nlev)
    cap *= 2;
  if ((e = hipMalloc((void**)&c->d_flags, 2 * cap)) != hipSuccess)
    return fail(c, "hipMalloc(flags)", e);
  if ((e = hipMalloc((void**)&c->d_counts, 5 * cap * sizeof(u64))) != hipSuccess)
    return fail(c, "hipMalloc(counts)", e);
  if ((e = hipMalloc((void**)&c->d_ab, 2 * cap * sizeof(float))) != hipSuccess)
    return fail(c, "hipMalloc(ab)", e);
  if ((e = hipMalloc((void**)&c->d_levels, cap * sizeof(int))) != hipSuccess)
    return fail(c, "hipMalloc(levels)", e);
  if ((e = hipHostMalloc(&c->h_pinned, 5 * cap * sizeof(u64) + 2 * cap + 2 * cap * sizeof(float), hipHostMallocDefault)) != hipSuccess)
    return fail(c, "hipHostMalloc", e);
  c->cap_lev = cap;
  return true;
}

// One tested level of a big field through the one-shot stencil kernels: room for their workgroups' counts (StencilParams::partials).
// Not while a capture is recorded (the buffer is the context's ONE, calls recorded side by side would share it; and growing
// it frees memory): those launches keep one atomic per workgroup.
void stencil_partials(mifc_ctx* c, mifc::StencilParams& P)
{
  P.partials = nullptr;
  P.partials_cap = 0;
  if (P.every_level_all_defined || c->capturing)
    return;
  // the forms' units per level: 4-row (one-input one-shot), 8-row (wind one-shot tiles) or 8- to 14-row (level-walking tiles)
  // blocks x 256-column segments; bounded from above.  Small levels keep their atomics (a few hundred per counter).
  const size_t per_level = (size_t)(P.ny_local / 4 + 2) * (size_t)(P.nx / 256 + 1);
  const size_t units = per_level * (size_t)P.nlev;
  if (per_level < 2048 || units > ((size_t)1 << 24))
    return;
  int cap = 0;
  P.partials = partials_for(c, units * 1024, &cap);
  P.partials_cap = P.partials ? cap : 0;
}

unsigned int* partials_for(mifc_ctx* c, size_t n_cells, int* cap)
{
  *cap = 0;
  const size_t blocks = (n_cells / 4 + 255) / 256; // one float4 per lane, 256 lanes
  if (blocks < 2048)
    return nullptr;
  if (blocks > c->partials_cap) {
    if (c->d_partials)
      (void)hipFree(c->d_partials);
    c->d_partials = nullptr;
    c->partials_cap = 0;
    if (hipMalloc((void**)&c->d_partials, blocks * sizeof(unsigned int)) != hipSuccess)
      return nullptr; // the launch then counts with one atomic per workgroup
    c->partials_cap = blocks;
  }
  *cap = (int)c->partials_cap;
  return c->d_partials;
}

bool pinned_acquire(mifc_ctx* c)
{
  if (c->pinned_read_pending) {
    hipError_t e = hipEventSynchronize(c->pinned_read);
    if (e != hipSuccess)
      return fail(c, "hipEventSynchronize", e);
    c->pinned_read_pending = false;
  }
  return true;
}

bool pinned_release(mifc_ctx* c)
{
  hipError_t e = hipEventRecord(c->pinned_read, c->stream);
  if (e != hipSuccess)
    return fail(c, "hipEventRecord", e);
  c->pinned_read_pending = true;
  return true;
}

bool scratch_release(mifc_ctx* c)
{
  if (c->capturing && c->n_lanes > 1) {
    // the context's per-level scratch (flags, hybrid coefficients) is ONE set: calls recorded side by side would overwrite it
    // under each other's kernels
    c->err = "a call that needs the context's per-level scratch was recorded into a capture with several lanes: record level batches of at most 8 "
             "levels (their flags and coefficients travel in the kernel arguments), or use one lane";
    return false;
  }
  hipError_t e = hipEventRecord(c->scratch_read, c->stream);
  if (e != hipSuccess)
    return fail(c, "hipEventRecord", e);
  c->scratch_read_pending = true;
  return true;
}

u64* pinned_counts(mifc_ctx* c)
{
  return reinterpret_cast<u64*>(c->h_pinned);
}
unsigned char* pinned_flags(mifc_ctx* c)
{
  return reinterpret_cast<unsigned char*>(c->h_pinned) + 5 * c->cap_lev * sizeof(u64);
}
float* pinned_ab(mifc_ctx* c)
{
  return reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(c->h_pinned) + 5 * c->cap_lev * sizeof(u64) + 2 * c->cap_lev);
}

// Brings a field to the device if the caller handed a host pointer.
const float* stage_in(mifc_ctx* c, int s, const float* p, size_t n, int memkind, bool* ok)
{
  if (!p || memkind == MIFC_MEM_DEVICE)
    return p;
  for (const mifc_ctx::HeldField& h : c->held)
    if (h.host == p && h.n >= n)
      return h.dev; // declared constant by the caller: already resident
  if (!ensure_slot(c, s, n * sizeof(float))) {
    *ok = false;
    return nullptr;
  }
  hipError_t e = hipMemcpyAsync(c->slot[s], p, n * sizeof(float), hipMemcpyHostToDevice, c->stream);
  if (e != hipSuccess) {
    fail(c, "hipMemcpyAsync(H2D)", e);
    *ok = false;
    return nullptr;
  }
  return static_cast<const float*>(c->slot[s]);
}

float* stage_out(mifc_ctx* c, int s, float* p, size_t n, int memkind, bool* ok, bool preload)
{
  if (!p || memkind == MIFC_MEM_DEVICE)
    return p;
  if (!ensure_slot(c, s, n * sizeof(float))) {
    *ok = false;
    return nullptr;
  }
  if (preload) { // operator may leave cells unwritten: start from the caller's content
    hipError_t e = hipMemcpyAsync(c->slot[s], p, n * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) {
      fail(c, "hipMemcpyAsync(H2D)", e);
      *ok = false;
      return nullptr;
    }
  }
  return static_cast<float*>(c->slot[s]);
}

bool fetch_out(mifc_ctx* c, int s, float* p, size_t n, int memkind)
{
  if (!p || memkind == MIFC_MEM_DEVICE)
    return true;
  hipError_t e = hipMemcpyAsync(p, c->slot[s], n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
  if (e != hipSuccess)
    return fail(c, "hipMemcpyAsync(D2H)", e);
  return true;
}

inline bool unit_is(const char* unit, const char* what)
{
  return unit && std::strcmp(unit, what) == 0;
}

} // namespace mifc_host

using namespace mifc_host;

namespace {

// ---- single-field elementwise driver --------------------------------------
int run_ewise(mifc_ctx* c, mifc::EwiseParams P, const float* in0, const float* in1, const float* in2, float* out, int* fdefined, int memkind,
              bool may_keep)
{
  const size_t n = (size_t)P.n;
  bool ok = true;
  P.in0 = stage_in(c, 0, in0, n, memkind, &ok);
  P.in1 = stage_in(c, 1, in1, n, memkind, &ok);
  P.in2 = stage_in(c, 2, in2, n, memkind, &ok);
  P.out = stage_out(c, 3, out, n, memkind, &ok, may_keep);
  if (!ok || !ensure_levels(c, 1))
    return 0;
  // With an ALL_DEFINED input nothing is tested, and the operators without a saturation table cannot
  // reject a cell on their own: the count is known to be zero, no counter round trip (5 us of a 19 us call)
  const bool table_free = P.op == mifc::EW_VECTORABS || P.op == mifc::EW_MOMENTUM_X || P.op == mifc::EW_MOMENTUM_Y ||
                          (P.op == mifc::EW_TEMP && P.compute >= 1 && P.compute <= 3);
  const bool counted = P.count && !(P.all_defined && table_free);
  const int want_flag = P.count;
  if (!counted)
    P.count = 0;
  P.n_undefined = c->d_counts;
  if (counted) {
    MIFC_HIP(c, hipMemsetAsync(c->d_counts, 0, sizeof(u64), c->stream));
    P.partials = partials_for(c, n, &P.partials_cap);
  }
  MIFC_LAUNCH(c, mifc::launch_ewise(P, c->stream));
  if (counted)
    MIFC_HIP(c, hipMemcpyAsync(pinned_counts(c), c->d_counts, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  if (!fetch_out(c, 3, out, n, memkind))
    return 0;
  MIFC_HIP(c, hipStreamSynchronize(c->stream));
  if (counted)
    *fdefined = mifc_classify(pinned_counts(c)[0], (u64)n);
  else if (want_flag)
    *fdefined = MIFC_ALL_DEFINED; // checkDefined(0, n)
  return 1;
}

mifc::EwiseParams ewise_base(int op, int nx, int ny, const int* fdefined, float undef)
{
  mifc::EwiseParams P;
  std::memset(&P, 0, sizeof P);
  P.op = op;
  P.n = nx * ny;
  P.all_defined = (*fdefined == MIFC_ALL_DEFINED);
  P.count = 1;
  P.undef = undef;
  P.unit_scale = 100.f;
  return P;
}

// ---- single-field / batched stencil driver ---------------------------------
struct StencilCall
{
  int op;
  int nx, ny, nlev;
  const float *f0, *f1, *xm, *ym, *fc;
  float *o0, *o1;
  const float* f2; // third input field (advection)
  float scale;     // advection, Q-vector
  float scale2;    // Q-vector
  const float* scale_lev;  // Q-vector over a level batch: per-level tables on the device (or null)
  const float* scale2_lev;
};

// count range of the raw loop -> what the flag is classified against
u64 stencil_denominator(int op, int nx, int ny)
{
  const u64 n = (u64)nx * (u64)ny;
  if (op == mifc::ST_IGWIND)
    return n; // FieldCalculations.cc:1543
  return n - 2 * (u64)nx; // :1868 and friends, also gradient compute 1 (:2068)
}

bool host_pipeline_enabled()
{
  return mifc::env().host_pipeline; // MIFC_HOST_PIPELINE=0: stage whole batches (for A/B measurements)
}

int run_stencil(mifc_ctx* c, const StencilCall& sc, int* fdefined /* [nlev] */, float undef, int memkind)
{
  if (sc.nx < 3 || sc.ny < 3 || sc.nlev < 1)
    return 0;
  const size_t n = (size_t)sc.nx * sc.ny;
  const size_t nb = n * (size_t)sc.nlev;
  bool ok = true;
  mifc::StencilParams P;
  std::memset(&P, 0, sizeof P);
  P.op = sc.op;
  P.nx = sc.nx;
  P.ny_global = sc.ny;
  P.j0 = 0;
  P.ny_local = sc.ny;
  P.nlev = sc.nlev;
  // A large level batch in host memory is streamed through the device in
  // chunks, copies in both directions overlapping the kernels (mifc_hostpipe.h);
  // everything else is staged whole.
  const bool piped = memkind == MIFC_MEM_HOST && !sc.f2 && sc.f0 && (sc.o0 || sc.o1) && mifc::hostpipe_chunk_levels(n, sc.nlev) > 0 && host_pipeline_enabled();
  if (piped) {
    if (!c->pipe && !(c->pipe = mifc::hostpipe_create(c->device))) {
      c->err = "host pipeline: cannot create streams";
      return 0;
    }
    P.f0 = sc.f0; // placeholders (non-null where the operator has the field); the chunk launcher substitutes device buffers
    P.f1 = sc.f1;
    P.out0 = sc.o0;
    P.out1 = sc.o1;
  } else {
    P.f0 = stage_in(c, 0, sc.f0, nb, memkind, &ok);
    P.f1 = stage_in(c, 1, sc.f1, nb, memkind, &ok);
    P.out0 = stage_out(c, 5, sc.o0, nb, memkind, &ok);
    P.out1 = stage_out(c, 6, sc.o1, nb, memkind, &ok);
    P.f2 = stage_in(c, 7, sc.f2, nb, memkind, &ok);
  }
  P.xmapr = stage_in(c, 2, sc.xm, n, memkind, &ok);
  P.ymapr = stage_in(c, 3, sc.ym, n, memkind, &ok);
  P.fcoriolis = stage_in(c, 4, sc.fc, n, memkind, &ok);
  P.scale = sc.scale;
  P.scale2 = sc.scale2;
  P.scale_lev = sc.scale_lev;
  P.scale2_lev = sc.scale2_lev;
  if (!ok || !ensure_levels(c, (size_t)sc.nlev))
    return 0;
  if (sc.op == mifc::ST_VORTDIV && !P.out0 && P.out1) {
    // only divergence requested
    P.op = mifc::ST_DIVERGENCE;
    P.out0 = P.out1;
    P.out1 = nullptr;
  } else if (sc.op == mifc::ST_VORTDIV && !P.out1) {
    P.op = mifc::ST_RELVORT;
  }
  P.in_level_stride = (long)n;
  P.out_level_stride = (long)n;
  P.undef = undef;
  P.n_undefined = c->d_counts;
  if (!pinned_acquire(c))
    return 0;
  bool every_all = true, any_all = false;
  for (int l = 0; l < sc.nlev; ++l) {
    const bool a = (fdefined[l] == MIFC_ALL_DEFINED);
    pinned_flags(c)[l] = a ? 1 : 0;
    every_all = every_all && a;
    any_all = any_all || a;
  }
  // the second pass of thermalFrontParameter rejects cells (|grad T| == 0) even
  // when its input flag is ALL_DEFINED: it always runs the counting variant
  // (and so does the last pass of plevelqvector, :570)
  if (sc.op == mifc::ST_TFP || sc.op == mifc::ST_QVEC_X || sc.op == mifc::ST_QVEC_Y)
    every_all = false;
  P.every_level_all_defined = every_all ? 1 : 0;
  // the kernels read a null flag array as "no level is ALL_DEFINED": the usual single-field call with
  // undefined values in it needs no flag upload
  P.all_defined = any_all ? c->d_flags : nullptr;
  if (!every_all) {
    // flags (bit-packed in the kernel arguments) and zeroed counters by one small kernel; very deep batches copy and fill
    if (sc.nlev <= mifc::kPrepMaxLevels) {
      MIFC_HIP(c, mifc::launch_prep_levels(any_all ? pinned_flags(c) : nullptr, sc.nlev, c->d_flags, c->d_counts, sc.nlev, c->stream));
    } else {
      if (any_all)
        MIFC_HIP(c, hipMemcpyAsync(c->d_flags, pinned_flags(c), (size_t)sc.nlev, hipMemcpyHostToDevice, c->stream));
      MIFC_HIP(c, hipMemsetAsync(c->d_counts, 0, sizeof(u64) * (size_t)sc.nlev, c->stream));
    }
  }
  if (piped) {
    MIFC_HIP(c, hipStreamSynchronize(c->stream)); // maps, flags and zeroed counters are in place
    // P.out0 / P.out1 may have been swapped above (only one of the two wanted)
    const float* h_in[2] = {sc.f0, sc.f1};
    float* h_out[2] = {P.out0, P.out1};
    const int n_in = sc.f1 ? 2 : 1;
    const mifc::StencilParams base = P;
    const mifc::ChunkLaunch launch = [&base](int l0, int nl, const float* const* d_in, float* const* d_out, hipStream_t stream) {
      mifc::StencilParams q = base;
      q.nlev = nl;
      q.f0 = d_in[0];
      q.f1 = base.f1 ? d_in[1] : nullptr;
      q.out0 = d_out[0];
      q.out1 = d_out[1];
      q.all_defined = base.all_defined ? base.all_defined + l0 : nullptr;
      q.n_undefined = base.n_undefined + l0;
      return mifc::launch_stencil(q, stream);
    };
    if (!mifc::hostpipe_run(c->pipe, n, sc.nlev, n_in, h_in, 2, h_out, launch, &c->err))
      return 0;
    if (!every_all)
      MIFC_HIP(c, hipMemcpyAsync(pinned_counts(c), c->d_counts, sizeof(u64) * (size_t)sc.nlev, hipMemcpyDeviceToHost, c->stream));
  } else {
    stencil_partials(c, P);
    MIFC_LAUNCH(c, mifc::launch_stencil(P, c->stream));
    if (!every_all)
      MIFC_HIP(c, hipMemcpyAsync(pinned_counts(c), c->d_counts, sizeof(u64) * (size_t)sc.nlev, hipMemcpyDeviceToHost, c->stream));
    if (!fetch_out(c, 5, sc.o0, nb, memkind) || !fetch_out(c, 6, sc.o1, nb, memkind))
      return 0;
  }
  MIFC_HIP(c, hipStreamSynchronize(c->stream));
  const u64 denom = stencil_denominator(sc.op, sc.nx, sc.ny);
  for (int l = 0; l < sc.nlev; ++l) {
    if (sc.op == mifc::ST_GWIND_X)
      fdefined[l] = mifc_classify(denom, denom); // FieldCalculations.cc:664: every cell is counted
    else
      fdefined[l] = every_all ? MIFC_ALL_DEFINED : mifc_classify(pinned_counts(c)[l], denom);
  }
  return 1;
}

} // namespace

extern "C" {

int mifc_abi_version(void)
{
  return MIFC_ABI_VERSION;
}

int mifc_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

mifc_ctx* mifc_create(int device)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n)
    return nullptr;
  if (hipSetDevice(device) != hipSuccess)
    return nullptr;
  mifc_ctx* c = new (std::nothrow) mifc_ctx();
  if (!c)
    return nullptr;
  c->device = device;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return nullptr;
  }
  c->stream = c->own_stream;
  if (hipEventCreateWithFlags(&c->pinned_read, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->scratch_read, hipEventDisableTiming) != hipSuccess) {
    if (c->pinned_read)
      (void)hipEventDestroy(c->pinned_read);
    (void)hipStreamDestroy(c->own_stream);
    delete c;
    return nullptr;
  }
  mifc::env_reload(); // the tuning / diagnostic environment is read here, never on a launch path
  return c;
}

int mifc_reload_env(mifc_ctx* c)
{
  if (!c)
    return 0;
  mifc::env_reload();
  return 1;
}

void mifc_destroy(mifc_ctx* c)
{
  if (!c)
    return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)mifc_comm_release(c); // a communicator the library created goes with the context
  for (int s = 0; s < mifc_ctx::NSLOT; ++s)
    if (c->slot[s])
      (void)hipFree(c->slot[s]);
  if (c->d_flags)
    (void)hipFree(c->d_flags);
  if (c->d_counts)
    (void)hipFree(c->d_counts);
  if (c->d_ab)
    (void)hipFree(c->d_ab);
  if (c->d_partials)
    (void)hipFree(c->d_partials);
  if (c->d_levels)
    (void)hipFree(c->d_levels);
  if (c->h_pinned)
    (void)hipHostFree(c->h_pinned);
  if (c->pinned_read)
    (void)hipEventDestroy(c->pinned_read);
  if (c->scratch_read)
    (void)hipEventDestroy(c->scratch_read);
  for (hipEvent_t e : c->tev)
    if (e)
      (void)hipEventDestroy(e);
  mifc::hostpipe_destroy(c->pipe);
  for (const mifc_ctx::HeldField& h : c->held)
    (void)hipFree(h.dev);
  if (c->capture_stream)
    (void)hipStreamDestroy(c->capture_stream);
  if (c->own_stream)
    (void)hipStreamDestroy(c->own_stream);
  delete c;
}

const char* mifc_last_error(const mifc_ctx* c)
{
  return c ? c->err.c_str() : "no context (no usable HIP device)";
}

// Work already queued on the old stream may still read the context's device scratch (per-level
// flags, hybrid coefficients of an *_enqueue call): the new stream waits for it before anything
// issued there can rewrite that scratch.
static int switch_stream(mifc_ctx* c, hipStream_t s)
{
  if (s == c->stream)
    return 1;
  if (c->scratch_read_pending)
    MIFC_HIP(c, hipStreamWaitEvent(s, c->scratch_read, 0));
  c->stream = s;
  return 1;
}

int mifc_set_stream(mifc_ctx* c, void* hip_stream)
{
  if (!c)
    return 0;
  enter(c);
  if (c->capturing) {
    c->err = "mifc_set_stream: a graph capture is open on this context (mifc_graph_begin)";
    return 0;
  }
  return switch_stream(c, static_cast<hipStream_t>(hip_stream)); // null = HIP's default stream
}

int mifc_use_own_stream(mifc_ctx* c)
{
  if (!c)
    return 0;
  enter(c);
  if (c->capturing) {
    c->err = "mifc_use_own_stream: a graph capture is open on this context (mifc_graph_begin)";
    return 0;
  }
  return switch_stream(c, c->own_stream);
}

int mifc_not_built(mifc_ctx* c, const char* what)
{
  if (c)
    c->err = std::string(what ? what : "?") + ": not built on the GPU (outside the hot-path scope; there is no CPU fallback)";
  return 0;
}

int mifc_synchronize(mifc_ctx* c)
{
  if (!c)
    return 0;
  enter(c);
  MIFC_HIP(c, hipStreamSynchronize(c->stream));
  return 1;
}

void* mifc_device_alloc(mifc_ctx* c, size_t bytes)
{
  if (!c)
    return nullptr;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    fail(c, "hipMalloc", e);
    return nullptr;
  }
  return p;
}

int mifc_device_free(mifc_ctx* c, void* dptr)
{
  if (!c)
    return 0;
  enter(c);
  MIFC_HIP(c, hipFree(dptr));
  return 1;
}

int mifc_copy_to_device(mifc_ctx* c, void* dst_dev, const void* src_host, size_t bytes)
{
  if (!c)
    return 0;
  enter(c);
  MIFC_HIP(c, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream));
  MIFC_HIP(c, hipStreamSynchronize(c->stream));
  return 1;
}

int mifc_copy_to_host(mifc_ctx* c, void* dst_host, const void* src_dev, size_t bytes)
{
  if (!c)
    return 0;
  enter(c);
  MIFC_HIP(c, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
  MIFC_HIP(c, hipStreamSynchronize(c->stream));
  return 1;
}

// Constant host fields (map ratios, Coriolis parameter): a caller that passes
// the same host array to many calls declares it once; host-pointer calls then
// find the device copy instead of uploading it again.  The caller promises not
// to change the content while it is held (there is no cheap way to notice: a
// checksum of the field costs more host time than the upload it would save).
int mifc_hold_field(mifc_ctx* c, const float* host_field, size_t n_floats)
{
  if (!c || !host_field || n_floats == 0)
    return 0;
  enter(c);
  for (mifc_ctx::HeldField& h : c->held) {
    if (h.host == host_field) { // refresh (content or size may have changed)
      if (h.n < n_floats) {
        MIFC_HIP(c, hipStreamSynchronize(c->stream));
        (void)hipFree(h.dev);
        h.dev = nullptr;
        h.n = 0;
        MIFC_HIP(c, hipMalloc((void**)&h.dev, n_floats * sizeof(float)));
        h.n = n_floats;
      }
      MIFC_HIP(c, hipMemcpyAsync(h.dev, host_field, n_floats * sizeof(float), hipMemcpyHostToDevice, c->stream));
      MIFC_HIP(c, hipStreamSynchronize(c->stream));
      return 1;
    }
  }
  mifc_ctx::HeldField h = {host_field, n_floats, nullptr};
  MIFC_HIP(c, hipMalloc((void**)&h.dev, n_floats * sizeof(float)));
  hipError_t e = hipMemcpyAsync(h.dev, host_field, n_floats * sizeof(float), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess)
    e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) {
    (void)hipFree(h.dev);
    fail(c, "mifc_hold_field", e);
    return 0;
  }
  c->held.push_back(h);
  return 1;
}

int mifc_release_field(mifc_ctx* c, const float* host_field)
{
  if (!c)
    return 0;
  enter(c);
  for (size_t k = 0; k < c->held.size(); ++k) {
    if (c->held[k].host == host_field) {
      MIFC_HIP(c, hipStreamSynchronize(c->stream));
      (void)hipFree(c->held[k].dev);
      c->held.erase(c->held.begin() + (long)k);
      return 1;
    }
  }
  return 0;
}

#ifdef MIFC_MEASUREMENT_BUILD // libmifc_measure.so only (include/mifc_measure.h)
// Measurement aid: between begin and end every kernel launch of this context is bracketed by a
// HIP event pair on its launch stream; end returns the summed kernel time in milliseconds
// (-1 on error, or when more than 16 launches happened in between).
int mifc_timing_begin(mifc_ctx* c)
{
  if (!c)
    return 0;
  enter(c);
  for (hipEvent_t& e : c->tev)
    if (!e)
      MIFC_HIP(c, hipEventCreate(&e));
  c->n_timed = 0;
  c->timing = true;
  return 1;
}

float mifc_timing_end_ms(mifc_ctx* c)
{
  if (!c || !c->timing)
    return -1.f;
  c->timing = false;
  float total = 0.f;
  for (int k = 0; k < c->n_timed; ++k) {
    float ms = 0.f;
    if (hipEventSynchronize(c->tev[2 * k + 1]) != hipSuccess || hipEventElapsedTime(&ms, c->tev[2 * k], c->tev[2 * k + 1]) != hipSuccess)
      return -1.f;
    total += ms;
  }
  return c->n_timed >= mifc_ctx::NTIMED ? -1.f : total;
}
#endif // MIFC_MEASUREMENT_BUILD

int mifc_counts_accumulate(mifc_ctx* c, int on)
{
  if (!c)
    return 0;
  c->counts_accumulate = on != 0;
  return 1;
}

int mifc_zero_counts_enqueue(mifc_ctx* c, unsigned long long* counts_dev, size_t n)
{
  if (!c || !counts_dev)
    return 0;
  enter(c);
  if (n)
    MIFC_HIP(c, hipMemsetAsync(counts_dev, 0, n * sizeof(u64), c->stream));
  return 1;
}

int mifc_classify(unsigned long long n_undefined, unsigned long long n)
{
  if (n_undefined == 0)
    return MIFC_ALL_DEFINED;
  if (n_undefined == n)
    return MIFC_NONE_DEFINED;
  return MIFC_SOME_DEFINED;
}

// ------------------------------------------------------------- elementwise

int mifc_vectorabs(mifc_ctx* c, int nx, int ny, const float* u, const float* v, float* ff, int* fdefined, float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  if (nx * ny <= 0) { // empty loop, checkDefined(0, 0)
    *fdefined = MIFC_ALL_DEFINED;
    return 1;
  }
  mifc::EwiseParams P = ewise_base(mifc::EW_VECTORABS, nx, ny, fdefined, undef);
  return run_ewise(c, P, u, v, nullptr, ff, fdefined, memkind, false);
}

int mifc_pleveltemp(mifc_ctx* c, int nx, int ny, const float* tinp, float p, const char* unit, int compute, float* tout, int* fdefined, float undef,
                    int memkind)
{
  if (!c)
    return 0;
  enter(c);
  if (p <= 0) // FieldCalculations.cc:330
    return 0;
  if (compute < 3) { // :340-345
    if (unit_is(unit, "celsius"))
      compute = 1;
    else if (unit_is(unit, "kelvin"))
      compute = 2;
  }
  if (compute < 1 || compute > 5) // :364
    return 0;
  mifc::EwiseParams P = ewise_base(mifc::EW_TEMP, nx, ny, fdefined, undef);
  P.psrc = mifc::PS_SCALAR;
  P.compute = compute;
  P.p = p;
  P.pidcp = powf(p * K_P0INV, K_KAPPA); // :347, on the host like the reference
  P.pi = P.pidcp * K_CP;
  P.count = (compute >= 4); // compute 1..3 leave fDefined untouched (:94-122)
  if (P.n <= 0) {
    if (P.count)
      *fdefined = MIFC_ALL_DEFINED;
    return 1;
  }
  return run_ewise(c, P, tinp, nullptr, nullptr, tout, fdefined, memkind, false);
}

int mifc_hleveltemp(mifc_ctx* c, int nx, int ny, const float* tinp, const float* ps, float alevel, float blevel, const char* unit, int compute,
                    float* tout, int* fdefined, float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  if (compute < 3) { // :1060-1065
    if (unit_is(unit, "celsius"))
      compute = 1;
    else if (unit_is(unit, "kelvin"))
      compute = 2;
  }
  if (bad_hlevel(alevel, blevel)) // :1070
    return 0;
  mifc::EwiseParams P = ewise_base(mifc::EW_TEMP, nx, ny, fdefined, undef);
  P.psrc = mifc::PS_HYBRID;
  P.compute = compute; // no range check in the reference: other values leave defined cells unwritten
  P.alevel = alevel;
  P.blevel = blevel;
  if (P.n <= 0) {
    *fdefined = MIFC_ALL_DEFINED; // checkDefined(0, 0)
    return 1;
  }
  return run_ewise(c, P, tinp, nullptr, ps, tout, fdefined, memkind, compute < 1 || compute > 5);
}

int mifc_aleveltemp(mifc_ctx* c, int nx, int ny, const float* tinp, const float* p, const char* unit, int compute, float* tout, int* fdefined,
                    float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  if (compute <= 0 || compute >= 6) // :1319
    return 0;
  if (compute < 3) {
    if (unit_is(unit, "celsius"))
      compute = 1;
    else if (unit_is(unit, "kelvin"))
      compute = 2;
  }
  mifc::EwiseParams P = ewise_base(mifc::EW_TEMP, nx, ny, fdefined, undef);
  P.psrc = mifc::PS_FIELD;
  P.compute = compute;
  if (P.n <= 0) {
    *fdefined = MIFC_ALL_DEFINED;
    return 1;
  }
  return run_ewise(c, P, tinp, nullptr, p, tout, fdefined, memkind, false);
}

static int hum_kind_ah(int compute) // numbering of alevelhum / hlevelhum (:1157-1164)
{
  if (compute <= 2)
    return mifc::HUM_Q_RH;
  if (compute <= 4)
    return mifc::HUM_RH_Q;
  if (compute == 5 || compute == 6 || compute == 9 || compute == 10)
    return mifc::HUM_Q_TD;
  return mifc::HUM_RH_TD;
}

int mifc_plevelhum(mifc_ctx* c, int nx, int ny, const float* t, const float* huminp, float p, const char* unit, int compute, float* humout,
                   int* fdefined, float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  if (p <= 0 || compute <= 0 || compute >= 13) // :419
    return 0;
  if (compute > 8 && unit_is(unit, "celsius")) // :422-425
    compute -= 4;
  else if (compute > 4 && compute <= 8 && unit_is(unit, "kelvin"))
    compute += 4;
  const int n = nx * ny;
  const bool rh_td = (compute == 5 || compute == 6 || compute == 9 || compute == 10);
  mifc::EwiseParams P = ewise_base(mifc::EW_HUM, nx, ny, fdefined, undef);
  P.psrc = mifc::PS_SCALAR;
  P.p = p;
  if (p == undef && !rh_td) { // :429-432 fillUndef (:76-82): result undef everywhere, NONE_DEFINED
    if (n > 0) {
      bool ok = true;
      float* out = stage_out(c, 3, humout, (size_t)n, memkind, &ok);
      if (!ok)
        return 0;
      unsigned int bits;
      std::memcpy(&bits, &undef, sizeof bits);
      MIFC_HIP(c, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(out), (int)bits, (size_t)n, c->stream));
      if (!fetch_out(c, 3, humout, (size_t)n, memkind))
        return 0;
      MIFC_HIP(c, hipStreamSynchronize(c->stream));
    }
    *fdefined = MIFC_NONE_DEFINED;
    return 1;
  }
  const float pi = K_CP * powf(p * K_P0INV, K_KAPPA); // :434 pi_from_p, on the host
  P.pi = pi;
  P.tconv = (compute % 2 == 0) ? (pi / K_CP) : 1; // :436
  P.tdconv = (compute >= 9) ? K_T0 : 0;           // :437
  if (compute <= 2) // numbering of plevelhum (:408-415)
    P.kind = mifc::HUM_Q_RH;
  else if (compute <= 4)
    P.kind = mifc::HUM_RH_Q;
  else if (rh_td)
    P.kind = mifc::HUM_RH_TD;
  else
    P.kind = mifc::HUM_Q_TD;
  P.ptest = mifc::PT_NONE;
  if (n <= 0) {
    *fdefined = MIFC_ALL_DEFINED;
    return 1;
  }
  return run_ewise(c, P, t, huminp, nullptr, humout, fdefined, memkind, false);
}

int mifc_hlevelhum(mifc_ctx* c, int nx, int ny, const float* t, const float* huminp, const float* ps, float alevel, float blevel, const char* unit,
                   int compute, float* humout, int* fdefined, float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  if (compute <= 0 || compute >= 13) // :1168
    return 0;
  if (bad_hlevel(alevel, blevel)) // :1170
    return 0;
  if (compute > 8 && unit_is(unit, "celsius")) // :1174-1177
    compute -= 4;
  else if (compute > 4 && compute <= 8 && unit_is(unit, "kelvin"))
    compute += 4;
  mifc::EwiseParams P = ewise_base(mifc::EW_HUM, nx, ny, fdefined, undef);
  P.psrc = mifc::PS_HYBRID;
  P.alevel = alevel;
  P.blevel = blevel;
  P.tdconv = (compute >= 9) ? K_T0 : 0; // :1181
  P.kind = hum_kind_ah(compute);
  P.from_theta = (compute % 2 == 0);
  const bool need_p = !(compute == 7 || compute == 11); // :1182
  P.ptest = need_p ? mifc::PT_NEQ : mifc::PT_NONE;
  if (P.n <= 0) {
    *fdefined = MIFC_ALL_DEFINED;
    return 1;
  }
  return run_ewise(c, P, t, huminp, need_p ? ps : nullptr, humout, fdefined, memkind, false);
}

int mifc_alevelhum(mifc_ctx* c, int nx, int ny, const float* t, const float* huminp, const float* p, const char* unit, int compute, float* humout,
                   int* fdefined, float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  if (compute <= 0 || compute >= 13) // :1414
    return 0;
  if (compute > 8 && unit_is(unit, "celsius")) // :1417-1420
    compute -= 4;
  else if (compute > 4 && compute <= 8 && unit_is(unit, "kelvin"))
    compute += 4;
  mifc::EwiseParams P = ewise_base(mifc::EW_HUM, nx, ny, fdefined, undef);
  P.psrc = mifc::PS_FIELD;
  P.tdconv = (compute >= 9) ? K_T0 : 0; // :1423
  P.kind = hum_kind_ah(compute);
  P.from_theta = (compute % 2 == 0);
  // :1429 -- p is tested (with != undef) only for compute 7/11, which do not use it
  const bool tests_p = (compute == 7 || compute == 11);
  P.ptest = tests_p ? mifc::PT_NEQ : mifc::PT_NONE;
  const bool reads_p = !tests_p || !P.all_defined;
  if (P.n <= 0) {
    *fdefined = MIFC_ALL_DEFINED;
    return 1;
  }
  return run_ewise(c, P, t, huminp, reads_p ? p : nullptr, humout, fdefined, memkind, false);
}

int mifc_cvhum(mifc_ctx* c, int nx, int ny, const float* t, const float* huminp, const char* unit, int compute, float* humout, int* fdefined,
               float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  float unit_scale = 100; // :1746-1750
  if (compute == 1 && unit_is(unit, "celsius"))
    compute = 2;
  if ((compute == 4 || compute == 5) && unit_is(unit, "1"))
    unit_scale = 1;
  if (compute < 1 || compute > 5) // :1813
    return 0;
  mifc::EwiseParams P = ewise_base(compute <= 3 ? mifc::EW_CVHUM_TD : mifc::EW_CVHUM_RH, nx, ny, fdefined, undef);
  P.tconv = (compute == 1 || compute == 2 || compute == 4) ? K_T0 : 0; // :1753
  P.tdconv = (compute == 1) ? K_T0 : 0;                                 // :1754
  P.unit_scale = unit_scale;
  if (P.n <= 0) {
    *fdefined = MIFC_ALL_DEFINED;
    return 1;
  }
  return run_ewise(c, P, t, huminp, nullptr, humout, fdefined, memkind, false);
}

// ---------------------------------------------------------------- stencils

int mifc_relvort(mifc_ctx* c, int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr, float* rvort, int* fdefined,
                 float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  const StencilCall sc = {mifc::ST_RELVORT, nx, ny, 1, u, v, xmapr, ymapr, nullptr, rvort, nullptr};
  return run_stencil(c, sc, fdefined, undef, memkind);
}

int mifc_absvort(mifc_ctx* c, int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr, const float* fcoriolis,
                 float* avort, int* fdefined, float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  const StencilCall sc = {mifc::ST_ABSVORT, nx, ny, 1, u, v, xmapr, ymapr, fcoriolis, avort, nullptr};
  return run_stencil(c, sc, fdefined, undef, memkind);
}

int mifc_divergence(mifc_ctx* c, int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr, float* diverg,
                    int* fdefined, float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  const StencilCall sc = {mifc::ST_DIVERGENCE, nx, ny, 1, u, v, xmapr, ymapr, nullptr, diverg, nullptr};
  return run_stencil(c, sc, fdefined, undef, memkind);
}

int mifc_gradient(mifc_ctx* c, int nx, int ny, const float* field, const float* xmapr, const float* ymapr, int compute, float* fgrad, int* fdefined,
                  float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  if (compute < 1 || compute > 4) // :2064 (size check comes first in the reference, both return false)
    return 0;
  const int op = mifc::ST_GRAD_X + (compute - 1);
  const StencilCall sc = {op, nx, ny, 1, field, nullptr, xmapr, ymapr, nullptr, fgrad, nullptr};
  return run_stencil(c, sc, fdefined, undef, memkind);
}

int mifc_plevelgwind_xcomp(mifc_ctx* c, int nx, int ny, const float* z, const float* xmapr, const float* ymapr, const float* fcoriolis, float* ug,
                           int* fdefined, float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  (void)xmapr; // unused by the reference as well (:638)
  const StencilCall sc = {mifc::ST_GWIND_X, nx, ny, 1, z, nullptr, nullptr, ymapr, fcoriolis, ug, nullptr};
  return run_stencil(c, sc, fdefined, undef, memkind);
}

int mifc_plevelgwind_ycomp(mifc_ctx* c, int nx, int ny, const float* z, const float* xmapr, const float* ymapr, const float* fcoriolis, float* vg,
                           int* fdefined, float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  (void)ymapr;
  // the reference lacks the nx<3||ny<3 guard here and would read out of bounds;
  // this implementation returns false instead
  const StencilCall sc = {mifc::ST_GWIND_Y, nx, ny, 1, z, nullptr, xmapr, nullptr, fcoriolis, vg, nullptr};
  return run_stencil(c, sc, fdefined, undef, memkind);
}

int mifc_plevelgvort(mifc_ctx* c, int nx, int ny, const float* z, const float* xmapr, const float* ymapr, const float* fcoriolis, float* gvort,
                     int* fdefined, float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  const StencilCall sc = {mifc::ST_GVORT, nx, ny, 1, z, nullptr, xmapr, ymapr, fcoriolis, gvort, nullptr};
  return run_stencil(c, sc, fdefined, undef, memkind);
}

int mifc_ilevelgwind(mifc_ctx* c, int nx, int ny, const float* mpot, const float* xmapr, const float* ymapr, const float* fcoriolis, float* ug,
                     float* vg, int* fdefined, float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  const StencilCall sc = {mifc::ST_IGWIND, nx, ny, 1, mpot, nullptr, xmapr, ymapr, fcoriolis, ug, vg};
  return run_stencil(c, sc, fdefined, undef, memkind);
}

// ------------------------------------------------ SURVEY.md 8f-1 operators

int mifc_advection(mifc_ctx* c, int nx, int ny, const float* f, const float* u, const float* v, const float* xmapr, const float* ymapr, float hours,
                   float* advec, int* fdefined, float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  StencilCall sc = {mifc::ST_ADVECTION, nx, ny, 1, f, u, xmapr, ymapr, nullptr, advec, nullptr};
  sc.f2 = v;
  sc.scale = (float)(-3600. * (double)hours); // FieldCalculations.cc:1963
  return run_stencil(c, sc, fdefined, undef, memkind);
}

int mifc_jacobian(mifc_ctx* c, int nx, int ny, const float* field1, const float* field2, const float* xmapr, const float* ymapr, float* fjacobian,
                  int* fdefined, float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  const StencilCall sc = {mifc::ST_JACOBIAN, nx, ny, 1, field1, field2, xmapr, ymapr, nullptr, fjacobian, nullptr};
  return run_stencil(c, sc, fdefined, undef, memkind);
}

static int momentum_coordinate(mifc_ctx* c, int op, int nx, int ny, const float* wind, const float* mapr, const float* fcoriolis, float fcoriolisMin,
                               float* out, int* fdefined, float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  if (nx < 3 || ny < 3) // :2363, :2397
    return 0;
  mifc::EwiseParams P = ewise_base(op, nx, ny, fdefined, undef);
  P.nx = nx;
  P.fcormin = fabsf(fcoriolisMin); // :2366
  return run_ewise(c, P, wind, mapr, fcoriolis, out, fdefined, memkind, false);
}

int mifc_momentumXcoordinate(mifc_ctx* c, int nx, int ny, const float* v, const float* xmapr, const float* fcoriolis, float fcoriolisMin, float* mxy,
                             int* fdefined, float undef, int memkind)
{
  return momentum_coordinate(c, mifc::EW_MOMENTUM_X, nx, ny, v, xmapr, fcoriolis, fcoriolisMin, mxy, fdefined, undef, memkind);
}

int mifc_momentumYcoordinate(mifc_ctx* c, int nx, int ny, const float* u, const float* ymapr, const float* fcoriolis, float fcoriolisMin, float* nxy,
                             int* fdefined, float undef, int memkind)
{
  return momentum_coordinate(c, mifc::EW_MOMENTUM_Y, nx, ny, u, ymapr, fcoriolis, fcoriolisMin, nxy, fdefined, undef, memkind);
}

// One launch of mifc_fused2.hip on device pointers.  Returns 0 on error, 1 when
// the result stands (flag written), 2 when the caller has to take the
// multi-pass path after all (see the thermalFrontParameter note below).
static int run_fused2(mifc_ctx* c, mifc::Fused2Params& P, int* fdefined)
{
  if (!ensure_levels(c, 4) || !pinned_acquire(c))
    return 0;
  P.counts = c->d_counts;
  P.check = (*fdefined != MIFC_ALL_DEFINED) ? 1 : 0;
  if (!mifc::fused2_supported(P))
    return 2;
  MIFC_HIP(c, hipMemsetAsync(c->d_counts, 0, 3 * sizeof(u64), c->stream));
  MIFC_LAUNCH(c, mifc::launch_fused2(P, c->stream));
  MIFC_HIP(c, hipMemcpyAsync(pinned_counts(c), c->d_counts, 3 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  MIFC_HIP(c, hipStreamSynchronize(c->stream));
  const u64* n = pinned_counts(c);
  // thermalFrontParameter's second pass tests its inputs only if the first pass
  // left something undefined (:2286).  The kernel ran it tested; if the first
  // pass turned out clean AND the test rejected a cell the untested loop would
  // have computed (a NaN gradient from defined inputs), the result differs.
  if (P.op == mifc::F2_TFP && P.check && n[0] == 0 && n[2] != 0)
    return 2;
  *fdefined = mifc_classify(n[1], (u64)P.nx * (u64)P.ny - 2 * (u64)P.nx); // :2303, :590
  return 1;
}

static bool fused2_enabled()
{
  return mifc::env().fused2; // MIFC_FUSED2=0: always the multi-pass path (A/B measurements, tests)
}

// thermalFrontParameter pass by pass on device pointers: |grad T| into the context's scratch, then the
// front parameter; the second pass takes its "all defined" from the flag the first one returned (:2286)
// Device pointers; nlev levels at once (fields nx * ny floats apart, the intermediate batch in the context's scratch): each
// pass is ONE launch over the levels, the second takes every level's "all defined" from what the first returned for it.
static int tfp_two_passes(mifc_ctx* c, int nx, int ny, const float* d_tx, const float* d_xm, const float* d_ym, float* d_out, int* fdefined,
                          float undef, int nlev = 1)
{
  const size_t n = (size_t)nx * ny;
  if (!ensure_slot(c, 8, n * (size_t)nlev * sizeof(float)))
    return 0;
  float* d_absdelt = static_cast<float*>(c->slot[8]);
  const StencilCall pass1 = {mifc::ST_GRAD_ABS, nx, ny, nlev, d_tx, nullptr, d_xm, d_ym, nullptr, d_absdelt, nullptr};
  if (!run_stencil(c, pass1, fdefined, undef, MIFC_MEM_DEVICE))
    return 0;
  const StencilCall pass2 = {mifc::ST_TFP, nx, ny, nlev, d_tx, d_absdelt, d_xm, d_ym, nullptr, d_out, nullptr};
  return run_stencil(c, pass2, fdefined, undef, MIFC_MEM_DEVICE);
}

// thermalFrontParameter, FieldCalculations.cc:2266-2309.  One fused launch where
// the grid allows it; otherwise two passes with an intermediate |grad T| field
// that lives in the context's scratch.  The second pass takes its "all defined"
// from the flag the first pass returned (:2286).
int mifc_thermalFrontParameter(mifc_ctx* c, int nx, int ny, const float* tx, const float* xmapr, const float* ymapr, float* tfp, int* fdefined,
                               float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  if (nx < 3 || ny < 3) // gradient() :2004
    return 0;
  const size_t n = (size_t)nx * ny;
  bool ok = true;
  // bring the inputs to the device once; both passes then run on device pointers
  const float* d_tx = stage_in(c, 0, tx, n, memkind, &ok);
  const float* d_xm = stage_in(c, 2, xmapr, n, memkind, &ok);
  const float* d_ym = stage_in(c, 3, ymapr, n, memkind, &ok);
  float* d_out = stage_out(c, 5, tfp, n, memkind, &ok);
  if (!ok)
    return 0;
  if (fused2_enabled()) {
    mifc::Fused2Params F;
    std::memset(&F, 0, sizeof F);
    F.op = mifc::F2_TFP;
    F.nx = nx;
    F.ny = ny;
    F.a = d_tx;
    F.xmapr = d_xm;
    F.ymapr = d_ym;
    F.out = d_out;
    F.undef = undef;
    const int r = run_fused2(c, F, fdefined);
    if (r == 0)
      return 0;
    if (r == 1) {
      if (!fetch_out(c, 5, tfp, n, memkind))
        return 0;
      MIFC_HIP(c, hipStreamSynchronize(c->stream));
      return 1;
    }
  }
  if (!tfp_two_passes(c, nx, ny, d_tx, d_xm, d_ym, d_out, fdefined, undef))
    return 0;
  if (!fetch_out(c, 5, tfp, n, memkind))
    return 0;
  MIFC_HIP(c, hipStreamSynchronize(c->stream));
  return 1;
}

// plevelqvector, FieldCalculations.cc:505-595: geostrophic wind x and y into
// the context's scratch, then the Q-vector component.  The flag threads through
// the three passes like the reference's fDefined: the x pass leaves NONE_DEFINED
// (:664), so the y pass always tests; the last pass tests whatever it is handed.
int mifc_plevelqvector(mifc_ctx* c, int nx, int ny, const float* z, const float* t, const float* xmapr, const float* ymapr, const float* fcoriolis,
                       float p, int compute, float* qcomp, int* fdefined, float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  if (p <= 0.0 || nx < 3 || ny < 3) // :526-530
    return 0;
  float tscale;
  if (compute == 1 || compute == 3) {
    tscale = 1.0f;
  } else if (compute == 2 || compute == 4) {
    const float pi = K_CP * powf(p / 1000.0f, 287.f / K_CP); // :539, host powf like the reference
    tscale = pi / K_CP;
  } else {
    return 0;
  }
  const size_t n = (size_t)nx * ny;
  bool ok = true;
  const float* d_z = stage_in(c, 0, z, n, memkind, &ok);
  const float* d_t = stage_in(c, 1, t, n, memkind, &ok);
  const float* d_xm = stage_in(c, 2, xmapr, n, memkind, &ok);
  const float* d_ym = stage_in(c, 3, ymapr, n, memkind, &ok);
  const float* d_fc = stage_in(c, 4, fcoriolis, n, memkind, &ok);
  float* d_out = stage_out(c, 5, qcomp, n, memkind, &ok);
  if (!ok)
    return 0;
  const float cscale = (float)((double)(-287.f) / ((double)p * 100.)); // :564
  if (fused2_enabled()) {
    mifc::Fused2Params F;
    std::memset(&F, 0, sizeof F);
    F.op = compute < 3 ? mifc::F2_QVEC_X : mifc::F2_QVEC_Y;
    F.nx = nx;
    F.ny = ny;
    F.a = d_z;
    F.t = d_t;
    F.xmapr = d_xm;
    F.ymapr = d_ym;
    F.fcoriolis = d_fc;
    F.out = d_out;
    F.undef = undef;
    F.scale = tscale;
    F.scale2 = cscale;
    const int r = run_fused2(c, F, fdefined);
    if (r == 0)
      return 0;
    if (r == 1) {
      if (!fetch_out(c, 5, qcomp, n, memkind))
        return 0;
      MIFC_HIP(c, hipStreamSynchronize(c->stream));
      return 1;
    }
  }
  if (!ensure_slot(c, 8, n * sizeof(float)) || !ensure_slot(c, 9, n * sizeof(float)))
    return 0;
  float* d_ug = static_cast<float*>(c->slot[8]);
  float* d_vg = static_cast<float*>(c->slot[9]);
  const StencilCall pass1 = {mifc::ST_GWIND_X, nx, ny, 1, d_z, nullptr, d_xm, d_ym, d_fc, d_ug, nullptr};
  if (!run_stencil(c, pass1, fdefined, undef, MIFC_MEM_DEVICE))
    return 0;
  const StencilCall pass2 = {mifc::ST_GWIND_Y, nx, ny, 1, d_z, nullptr, d_xm, d_ym, d_fc, d_vg, nullptr};
  if (!run_stencil(c, pass2, fdefined, undef, MIFC_MEM_DEVICE))
    return 0;
  StencilCall pass3 = {compute < 3 ? mifc::ST_QVEC_X : mifc::ST_QVEC_Y, nx, ny, 1, d_ug, d_vg, d_xm, d_ym, nullptr, d_out, nullptr};
  pass3.f2 = d_t;
  pass3.scale = tscale;
  pass3.scale2 = cscale;
  if (!run_stencil(c, pass3, fdefined, undef, MIFC_MEM_DEVICE))
    return 0;
  if (!fetch_out(c, 5, qcomp, n, memkind))
    return 0;
  MIFC_HIP(c, hipStreamSynchronize(c->stream));
  return 1;
}

// ----------------------------------------------------------------- batched

int mifc_vortdiv_levels(mifc_ctx* c, int nx, int ny, int nlev, const float* u, const float* v, const float* xmapr, const float* ymapr, float* rvort,
                        float* diverg, int* fdefined, float undef, int memkind)
{
  if (!c || (!rvort && !diverg))
    return 0;
  enter(c);
  const StencilCall sc = {mifc::ST_VORTDIV, nx, ny, nlev, u, v, xmapr, ymapr, nullptr, rvort, diverg};
  return run_stencil(c, sc, fdefined, undef, memkind);
}

int mifc_stencil_levels(mifc_ctx* c, int op, int nx, int ny, int nlev, const float* f0, const float* f1, const float* xmapr, const float* ymapr,
                        const float* fcoriolis, float* out0, float* out1, int* fdefined, float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  if (!((op >= mifc::ST_RELVORT && op <= mifc::ST_IGWIND) || op == mifc::ST_JACOBIAN) || !f0 || !out0)
    return 0;
  const bool wind = (op <= mifc::ST_VORTDIV) || op == mifc::ST_JACOBIAN; // two input fields per level
  if (wind && !f1)
    return 0;
  const bool two_out = (op == mifc::ST_VORTDIV || op == mifc::ST_IGWIND);
  const StencilCall sc = {op, nx, ny, nlev, f0, wind ? f1 : nullptr, xmapr, ymapr, fcoriolis, out0, two_out ? out1 : nullptr};
  return run_stencil(c, sc, fdefined, undef, memkind);
}

// ---- the f1 operators over a batch of levels (shared map factors) -----------------------------
// advection runs on the batched stencil driver; thermalFrontParameter, plevelqvector and shapiro2_filter on
// their one-launch kernels with grid.y = level, in two groups: the levels whose input flag is ALL_DEFINED
// (no tests) and the others.  Grids those kernels do not take (nx % 4 != 0, unaligned) go level by level
// through the single-field entry points.
static int f1_levels_fallback(mifc_ctx* c, int op, int nx, int ny, int nlev, const float* f0, const float* f1, const float* xm, const float* ym,
                              const float* fc, const float* level_p, int compute, float* out0, int* fdefined, float undef,
                              const float* tscale = nullptr, const float* cscale = nullptr)
{
  const size_t n = (size_t)nx * ny;
  if (op == MIFC_OP_TFP && nlev > 1) // widths the one-launch kernel does not take: the two passes, each over all levels
    return tfp_two_passes(c, nx, ny, f0, xm, ym, out0, fdefined, undef, nlev);
  if (op == MIFC_OP_QVECTOR && nlev > 1 && tscale && cscale) {
    if (!ensure_levels(c, (size_t)nlev))
      return 0;
    // the three passes (:555-590), each ONE launch over all levels; the last one takes its two scalars per level from
    // device tables (they depend on the level's pressure)
    const size_t nb = n * (size_t)nlev;
    if (!ensure_slot(c, 8, nb * sizeof(float)) || !ensure_slot(c, 9, nb * sizeof(float)))
      return 0;
    float* d_ug = static_cast<float*>(c->slot[8]);
    float* d_vg = static_cast<float*>(c->slot[9]);
    StencilCall pass1 = {mifc::ST_GWIND_X, nx, ny, nlev, f0, nullptr, xm, ym, fc, d_ug, nullptr};
    if (!run_stencil(c, pass1, fdefined, undef, MIFC_MEM_DEVICE))
      return 0;
    StencilCall pass2 = {mifc::ST_GWIND_Y, nx, ny, nlev, f0, nullptr, xm, ym, fc, d_vg, nullptr};
    if (!run_stencil(c, pass2, fdefined, undef, MIFC_MEM_DEVICE))
      return 0;
    // the tables go up on the stream the passes run on; run_stencil() synchronises before it returns, so the host
    // vectors outlive the copies
    MIFC_HIP(c, hipMemcpyAsync(c->d_ab, tscale, sizeof(float) * (size_t)nlev, hipMemcpyHostToDevice, c->stream));
    MIFC_HIP(c, hipMemcpyAsync(c->d_ab + c->cap_lev, cscale, sizeof(float) * (size_t)nlev, hipMemcpyHostToDevice, c->stream));
    StencilCall pass3 = {compute < 3 ? mifc::ST_QVEC_X : mifc::ST_QVEC_Y, nx, ny, nlev, d_ug, d_vg, xm, ym, nullptr, out0, nullptr};
    pass3.f2 = f1;
    pass3.scale_lev = c->d_ab;
    pass3.scale2_lev = c->d_ab + c->cap_lev;
    return run_stencil(c, pass3, fdefined, undef, MIFC_MEM_DEVICE);
  }
  for (int l = 0; l < nlev; ++l) {
    int rc;
    if (op == MIFC_OP_TFP)
      rc = mifc_thermalFrontParameter(c, nx, ny, f0 + l * n, xm, ym, out0 + l * n, fdefined + l, undef, MIFC_MEM_DEVICE);
    else if (op == MIFC_OP_QVECTOR)
      rc = mifc_plevelqvector(c, nx, ny, f0 + l * n, f1 + l * n, xm, ym, fc, level_p[l], compute, out0 + l * n, fdefined + l, undef, MIFC_MEM_DEVICE);
    else
      rc = mifc_shapiro2_filter(c, nx, ny, f0 + l * n, out0 + l * n, fdefined + l, undef, MIFC_MEM_DEVICE);
    if (!rc)
      return 0;
  }
  return 1;
}

int mifc_stencil_levels_ex(mifc_ctx* c, int op, int nx, int ny, int nlev, const float* f0, const float* f1, const float* f2, const float* xmapr,
                           const float* ymapr, const float* fcoriolis, const float* level_scalars, float scalar, int compute, float* out0,
                           float* out1, int* fdefined, float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  if (op != MIFC_OP_ADVECTION && op != MIFC_OP_TFP && op != MIFC_OP_QVECTOR && op != MIFC_OP_SHAPIRO2)
    return mifc_stencil_levels(c, op, nx, ny, nlev, f0, f1, xmapr, ymapr, fcoriolis, out0, out1, fdefined, undef, memkind);
  if (nx < 3 || ny < 3 || nlev < 1 || !f0 || !out0 || !fdefined)
    return 0;
  if (op == MIFC_OP_ADVECTION) { // FieldCalculations.cc:1942-1983: f0 = f, f1 = u, f2 = v, scalar = hours
    if (!f1 || !f2)
      return 0;
    StencilCall sc = {mifc::ST_ADVECTION, nx, ny, nlev, f0, f1, xmapr, ymapr, nullptr, out0, nullptr};
    sc.f2 = f2;
    sc.scale = (float)(-3600. * (double)scalar); // :1963
    return run_stencil(c, sc, fdefined, undef, memkind);
  }
  const size_t n = (size_t)nx * ny, nb = n * (size_t)nlev;
  std::vector<float> tscale, cscale;
  if (op == MIFC_OP_QVECTOR) { // :526-540, per level like the single-field call
    if (!f1 || !fcoriolis || !level_scalars)
      return 0;
    tscale.resize(nlev);
    cscale.resize(nlev);
    for (int l = 0; l < nlev; ++l) {
      const float p = level_scalars[l];
      if (p <= 0.0)
        return 0;
      if (compute == 1 || compute == 3)
        tscale[l] = 1.0f;
      else if (compute == 2 || compute == 4)
        tscale[l] = K_CP * powf(p / 1000.0f, 287.f / K_CP) / K_CP; // :539, host powf like the reference
      else
        return 0;
      cscale[l] = (float)((double)(-287.f) / ((double)p * 100.)); // :564
    }
  }
  bool ok = true;
  const float* d0 = stage_in(c, 0, f0, nb, memkind, &ok);
  const float* d1 = (op == MIFC_OP_QVECTOR) ? stage_in(c, 1, f1, nb, memkind, &ok) : nullptr;
  const float* dxm = (op != MIFC_OP_SHAPIRO2) ? stage_in(c, 2, xmapr, n, memkind, &ok) : nullptr;
  const float* dym = (op != MIFC_OP_SHAPIRO2) ? stage_in(c, 3, ymapr, n, memkind, &ok) : nullptr;
  const float* dfc = (op == MIFC_OP_QVECTOR) ? stage_in(c, 4, fcoriolis, n, memkind, &ok) : nullptr;
  float* dout = stage_out(c, 5, out0, nb, memkind, &ok);
  if (!ok || !ensure_levels(c, (size_t)nlev))
    return 0;
  if (op != MIFC_OP_SHAPIRO2 && (!dxm || !dym))
    return 0;
  // the levels in two groups: ALL_DEFINED input first
  std::vector<int> order;
  order.reserve(nlev);
  for (int l = 0; l < nlev; ++l)
    if (fdefined[l] == MIFC_ALL_DEFINED)
      order.push_back(l);
  const int n_all = (int)order.size();
  for (int l = 0; l < nlev; ++l)
    if (fdefined[l] != MIFC_ALL_DEFINED)
      order.push_back(l);
  bool fused = nlev <= 65535;
  if (op == MIFC_OP_SHAPIRO2) {
    float* dst = dout;
    if (dout == d0) { // in place (allowed by the reference, :2088): through a scratch batch
      if (!ensure_slot(c, 8, nb * sizeof(float)))
        return 0;
      dst = static_cast<float*>(c->slot[8]);
    }
    fused = fused && mifc::env().shapiro_fused && mifc::shapiro2_fused_supported(nx, ny, d0, dst) && (n % 4 == 0 || mifc::env().shapiro_regs);
    if (fused) {
      MIFC_HIP(c, hipMemcpyAsync(c->d_levels, order.data(), sizeof(int) * (size_t)nlev, hipMemcpyHostToDevice, c->stream));
      if (n_all > 0)
        MIFC_LAUNCH(c, mifc::launch_shapiro2_fused_levels(nx, ny, 1, undef, d0, dst, n_all, (long)n, c->d_levels, c->stream));
      if (nlev - n_all > 0)
        MIFC_LAUNCH(c, mifc::launch_shapiro2_fused_levels(nx, ny, 0, undef, d0, dst, nlev - n_all, (long)n, c->d_levels + n_all, c->stream));
      if (dst != dout)
        MIFC_HIP(c, hipMemcpyAsync(dout, dst, nb * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
      if (!fetch_out(c, 5, out0, nb, memkind))
        return 0;
      MIFC_HIP(c, hipStreamSynchronize(c->stream));
      for (int l = 0; l < nlev; ++l)
        fdefined[l] = MIFC_ALL_DEFINED; // :2171
      return 1;
    }
    // widths the one-launch kernel does not take: the sweep-by-sweep path over the levels of each flag group (five launches
    // per group whatever the number of levels), in place on the output batch like the reference (:2099-2104)
    if (nlev > 1 && nlev <= 65535 && n <= 0x7fffffffu) {
      if (!ensure_slot(c, 9, nb * sizeof(float)))
        return 0;
      const bool any_tested = n_all < nlev;
      if (any_tested && !ensure_slot(c, 7, 2 * nb))
        return 0;
      if (dst != d0)
        MIFC_HIP(c, hipMemcpyAsync(dst, d0, nb * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
      MIFC_HIP(c, hipMemcpyAsync(c->d_levels, order.data(), sizeof(int) * (size_t)nlev, hipMemcpyHostToDevice, c->stream));
      mifc::ShapiroParams SP;
      SP.nx = nx;
      SP.ny = ny;
      SP.undef = undef;
      SP.f1 = dst;
      SP.f2 = static_cast<float*>(c->slot[9]);
      SP.mask_x = any_tested ? static_cast<unsigned char*>(c->slot[7]) : nullptr;
      SP.mask_y = any_tested ? static_cast<unsigned char*>(c->slot[7]) + nb : nullptr;
      if (n_all > 0) {
        SP.all_defined = 1;
        MIFC_LAUNCH(c, mifc::launch_shapiro2_levels(SP, n_all, c->d_levels, c->stream));
      }
      if (any_tested) {
        SP.all_defined = 0;
        MIFC_LAUNCH(c, mifc::launch_shapiro2_levels(SP, nlev - n_all, c->d_levels + n_all, c->stream));
      }
      if (dst != dout)
        MIFC_HIP(c, hipMemcpyAsync(dout, dst, nb * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
      if (!fetch_out(c, 5, out0, nb, memkind))
        return 0;
      MIFC_HIP(c, hipStreamSynchronize(c->stream));
      for (int l = 0; l < nlev; ++l)
        fdefined[l] = MIFC_ALL_DEFINED; // :2171
      return 1;
    }
  } else {
    mifc::Fused2Params F;
    std::memset(&F, 0, sizeof F);
    F.op = (op == MIFC_OP_TFP) ? mifc::F2_TFP : (compute < 3 ? mifc::F2_QVEC_X : mifc::F2_QVEC_Y);
    F.nx = nx;
    F.ny = ny;
    F.a = d0;
    F.t = d1;
    F.xmapr = dxm;
    F.ymapr = dym;
    F.fcoriolis = dfc;
    F.out = dout;
    F.undef = undef;
    F.counts = c->d_counts;
    F.level_stride = (long)n;
    fused = fused && fused2_enabled() && mifc::fused2_supported(F) && 3 * (size_t)nlev <= 5 * c->cap_lev;
    if (fused) {
      if (!pinned_acquire(c))
        return 0;
      MIFC_HIP(c, hipMemcpyAsync(c->d_levels, order.data(), sizeof(int) * (size_t)nlev, hipMemcpyHostToDevice, c->stream));
      if (op == MIFC_OP_QVECTOR) {
        MIFC_HIP(c, hipMemcpyAsync(c->d_ab, tscale.data(), sizeof(float) * (size_t)nlev, hipMemcpyHostToDevice, c->stream));
        MIFC_HIP(c, hipMemcpyAsync(c->d_ab + c->cap_lev, cscale.data(), sizeof(float) * (size_t)nlev, hipMemcpyHostToDevice, c->stream));
        F.scale_lev = c->d_ab;
        F.scale2_lev = c->d_ab + c->cap_lev;
      }
      MIFC_HIP(c, hipMemsetAsync(c->d_counts, 0, 3 * sizeof(u64) * (size_t)nlev, c->stream));
      for (int group = 0; group < 2; ++group) {
        const int first = group == 0 ? 0 : n_all, count = group == 0 ? n_all : nlev - n_all;
        if (count == 0)
          continue;
        F.check = group;
        F.n_launch_levels = count;
        F.levels = c->d_levels + first;
        MIFC_LAUNCH(c, mifc::launch_fused2(F, c->stream));
      }
      MIFC_HIP(c, hipMemcpyAsync(pinned_counts(c), c->d_counts, 3 * sizeof(u64) * (size_t)nlev, hipMemcpyDeviceToHost, c->stream));
      MIFC_HIP(c, hipStreamSynchronize(c->stream));
      const u64* cnt = pinned_counts(c);
      std::vector<int> redo;
      for (int l = 0; l < nlev; ++l) {
        const u64* k = cnt + 3 * (size_t)l;
        // thermalFrontParameter: the second pass is tested only if the first left something undefined (:2286); see run_fused2()
        if (op == MIFC_OP_TFP && fdefined[l] != MIFC_ALL_DEFINED && k[0] == 0 && k[2] != 0)
          redo.push_back(l);
        else
          fdefined[l] = mifc_classify(k[1], (u64)n - 2 * (u64)nx); // :2303, :590
      }
      for (int l : redo) // rare: those levels again, pass by pass
        if (!tfp_two_passes(c, nx, ny, d0 + (size_t)l * n, dxm, dym, dout + (size_t)l * n, fdefined + l, undef))
          return 0;
      if (!fetch_out(c, 5, out0, nb, memkind))
        return 0;
      MIFC_HIP(c, hipStreamSynchronize(c->stream));
      return 1;
    }
  }
  // level by level on the staged (device) batch
  if (!f1_levels_fallback(c, op, nx, ny, nlev, d0, d1, dxm, dym, dfc, level_scalars, compute, dout, fdefined, undef,
                          tscale.empty() ? nullptr : tscale.data(), cscale.empty() ? nullptr : cscale.data()))
    return 0;
  if (!fetch_out(c, 5, out0, nb, memkind))
    return 0;
  MIFC_HIP(c, hipStreamSynchronize(c->stream));
  return 1;
}

int mifc_vortdiv_levels_enqueue(mifc_ctx* c, int nx, int ny, int nlev, const float* u, const float* v, const float* xmapr, const float* ymapr,
                                float* rvort, float* diverg, const int* fdefined_in, float undef, unsigned long long* n_undefined_dev)
{
  const size_t n = (size_t)(nx > 0 ? nx : 0) * (size_t)(ny > 0 ? ny : 0);
  return mifc_vortdiv_levels_strided_enqueue(c, nx, ny, nlev, u, v, xmapr, ymapr, rvort, diverg, n, n, fdefined_in, undef, n_undefined_dev);
}

int mifc_vortdiv_ff_levels_enqueue(mifc_ctx* c, int nx, int ny, int nlev, const float* u, const float* v, const float* xmapr, const float* ymapr,
                                   float* rvort, float* diverg, float* ff, const int* fdefined_in, float undef, unsigned long long* n_undefined_dev,
                                   unsigned long long* n_undefined_ff_dev)
{
  if (!c || !rvort || !diverg || !ff || !u || !v || !xmapr || !ymapr)
    return 0;
  enter(c);
  if (nx < 3 || ny < 3 || nlev < 1)
    return 0;
  if (!ensure_levels(c, (size_t)nlev))
    return 0;
  bool every_all = (fdefined_in != nullptr);
  std::vector<unsigned char> hf((size_t)nlev);
  for (int l = 0; l < nlev; ++l) {
    hf[l] = (fdefined_in && fdefined_in[l] == MIFC_ALL_DEFINED) ? 1 : 0;
    every_all = every_all && hf[l];
  }
  if (!every_all && (!n_undefined_dev || !n_undefined_ff_dev)) {
    c->err = "mifc_vortdiv_ff_levels_enqueue: both counter arrays are required unless every level is ALL_DEFINED";
    return 0;
  }
  const bool piecewise = nlev > mifc::kPrepMaxLevels;
  if (!every_all) {
    if (piecewise) {
      if (!pinned_acquire(c))
        return 0;
      std::memcpy(pinned_flags(c), hf.data(), (size_t)nlev);
      MIFC_HIP(c, hipMemcpyAsync(c->d_flags, pinned_flags(c), (size_t)nlev, hipMemcpyHostToDevice, c->stream));
      if (!pinned_release(c))
        return 0;
      MIFC_HIP(c, hipMemsetAsync(n_undefined_dev, 0, sizeof(u64) * (size_t)nlev, c->stream));
      MIFC_HIP(c, hipMemsetAsync(n_undefined_ff_dev, 0, sizeof(u64) * (size_t)nlev, c->stream));
    } else {
      MIFC_HIP(c, mifc::launch_prep_levels(hf.data(), nlev, c->d_flags, n_undefined_dev, nlev, c->stream));
      MIFC_HIP(c, mifc::launch_prep_levels(nullptr, nlev, nullptr, n_undefined_ff_dev, nlev, c->stream));
    }
  }
  mifc::StencilParams P;
  std::memset(&P, 0, sizeof P);
  P.op = mifc::ST_VORTDIV;
  P.nx = nx;
  P.ny_global = ny;
  P.ny_local = ny;
  P.nlev = nlev;
  P.f0 = u;
  P.f1 = v;
  P.xmapr = xmapr;
  P.ymapr = ymapr;
  P.out0 = rvort;
  P.out1 = diverg;
  P.out_ff = ff;
  P.n_undefined_ff = n_undefined_ff_dev;
  P.in_level_stride = (long)nx * ny;
  P.out_level_stride = (long)nx * ny;
  P.undef = undef;
  P.every_level_all_defined = every_all ? 1 : 0;
  P.all_defined = c->d_flags;
  P.n_undefined = n_undefined_dev;
  const bool timed = c->timing && c->n_timed < mifc_ctx::NTIMED;
  if (timed)
    (void)hipEventRecord(c->tev[2 * c->n_timed], c->stream);
  hipError_t e = mifc::launch_stencil(P, c->stream);
  if (e == hipErrorNotSupported) {
    // not a launch the three-output kernel takes (shallow or small batch, ragged width, NaN undef, a forced tuning): the
    // pair as usual and the wind speed as a launch of its own (the batched vectorabs of mifc_derived.hip)
    (void)hipGetLastError();
    P.out_ff = nullptr;
    P.n_undefined_ff = nullptr;
    e = mifc::launch_stencil(P, c->stream);
    if (e == hipSuccess && (nx * ny) % 4 != 0) {
      // a cell count the batched vectorabs does not take: level by level on the single-field kernel
      for (int l = 0; l < nlev && e == hipSuccess; ++l) {
        const int fl = hf[l] ? MIFC_ALL_DEFINED : MIFC_SOME_DEFINED;
        mifc::EwiseParams E = ewise_base(mifc::EW_VECTORABS, nx, ny, &fl, undef);
        E.in0 = u + (size_t)l * nx * ny;
        E.in1 = v + (size_t)l * nx * ny;
        E.out = ff + (size_t)l * nx * ny;
        E.count = hf[l] ? 0 : 1;
        E.n_undefined = hf[l] ? nullptr : n_undefined_ff_dev + l;
        e = mifc::launch_ewise(E, c->stream);
      }
    } else if (e == hipSuccess) {
      mifc::DerivedParams D;
      std::memset(&D, 0, sizeof D);
      D.n = nx * ny;
      D.nlev = nlev;
      D.u = u;
      D.v = v;
      D.ff = ff;
      D.wind_all_defined = c->d_flags;
      D.thermo_all_defined = c->d_flags;
      D.every_level_all_defined = every_all ? 1 : 0;
      D.undef = undef;
      D.cnt_ff = n_undefined_ff_dev;
      e = mifc::launch_derived_levels(D, c->stream);
    }
  }
  if (timed) {
    (void)hipEventRecord(c->tev[2 * c->n_timed + 1], c->stream);
    c->n_timed += 1;
  }
  if (e != hipSuccess) {
    fail(c, "mifc_vortdiv_ff_levels_enqueue: launch", e);
    return 0;
  }
  if (!every_all && !scratch_release(c)) // the kernels read c->d_flags
    return 0;
  return 1;
}

const char* mifc_last_stencil_form(void)
{
  return mifc::last_form();
}

unsigned long long mifc_stencil_count_domain(int op, int nx, int ny)
{
  return stencil_denominator(op, nx, ny);
}

size_t mifc_batch_level_stride(int nx, int ny)
{
  if (nx <= 0 || ny <= 0)
    return 0;
  return mifc::padded_level_stride((size_t)nx * (size_t)ny);
}

// what the asynchronous level-batch entries share: flags up, counters zeroed, one launch, nothing read back
static int stencil_enqueue(mifc_ctx* c, const char* who, mifc::StencilParams& P, const int* fdefined_in, unsigned long long* n_undefined_dev)
{
  if (!ensure_levels(c, (size_t)P.nlev))
    return 0;
  bool every_all = (fdefined_in != nullptr), any_all = false;
  for (int l = 0; l < P.nlev; ++l) {
    const bool a = fdefined_in && fdefined_in[l] == MIFC_ALL_DEFINED;
    every_all = every_all && a;
    any_all = any_all || a;
  }
  P.every_level_all_defined = every_all ? 1 : 0;
  P.all_defined = any_all ? c->d_flags : nullptr; // the kernels read a null flag array as "no level is ALL_DEFINED"
  P.n_undefined = n_undefined_dev;
  if (!every_all && !n_undefined_dev) {
    c->err = std::string(who) + ": n_undefined_dev is required unless every level is ALL_DEFINED";
    return 0;
  }
  const bool upload = !every_all && any_all;
  if (P.nlev <= mifc::kPrepMaxLevels) {
    // flags (bit-packed in the kernel arguments) and zeroed counters by ONE small kernel in front of the operator's
    std::vector<unsigned char> hf;
    if (upload) {
      hf.resize((size_t)P.nlev);
      for (int l = 0; l < P.nlev; ++l)
        hf[l] = fdefined_in[l] == MIFC_ALL_DEFINED ? 1 : 0;
    }
    MIFC_HIP(c, mifc::launch_prep_levels(upload ? hf.data() : nullptr, P.nlev, c->d_flags, c->counts_accumulate ? nullptr : n_undefined_dev, P.nlev,
                                         c->stream));
  } else {
    if (upload) {
      if (!pinned_acquire(c))
        return 0;
      for (int l = 0; l < P.nlev; ++l)
        pinned_flags(c)[l] = fdefined_in[l] == MIFC_ALL_DEFINED ? 1 : 0;
      MIFC_HIP(c, hipMemcpyAsync(c->d_flags, pinned_flags(c), (size_t)P.nlev, hipMemcpyHostToDevice, c->stream));
      if (!pinned_release(c))
        return 0;
    }
    if (n_undefined_dev && !c->counts_accumulate)
      MIFC_HIP(c, hipMemsetAsync(n_undefined_dev, 0, sizeof(u64) * (size_t)P.nlev, c->stream));
  }
  stencil_partials(c, P);
  MIFC_LAUNCH(c, mifc::launch_stencil(P, c->stream));
  if ((upload || P.partials) && !scratch_release(c)) // the kernels read c->d_flags / write and read c->d_partials
    return 0;
  return 1;
}

int mifc_vortdiv_levels_strided_enqueue(mifc_ctx* c, int nx, int ny, int nlev, const float* u, const float* v, const float* xmapr, const float* ymapr,
                                        float* rvort, float* diverg, size_t in_level_stride, size_t out_level_stride, const int* fdefined_in,
                                        float undef, unsigned long long* n_undefined_dev)
{
  if (!c || (!rvort && !diverg))
    return 0;
  enter(c);
  if (nx < 3 || ny < 3 || nlev < 1)
    return 0;
  if (in_level_stride < (size_t)nx * ny || out_level_stride < (size_t)nx * ny) {
    c->err = "mifc_vortdiv_levels_strided_enqueue: a level stride is smaller than one field";
    return 0;
  }
  mifc::StencilParams P;
  std::memset(&P, 0, sizeof P);
  P.op = mifc::ST_VORTDIV;
  P.out0 = rvort;
  P.out1 = diverg;
  if (!rvort) {
    P.op = mifc::ST_DIVERGENCE;
    P.out0 = diverg;
    P.out1 = nullptr;
  } else if (!diverg) {
    P.op = mifc::ST_RELVORT;
  }
  P.nx = nx;
  P.ny_global = ny;
  P.j0 = 0;
  P.ny_local = ny;
  P.nlev = nlev;
  P.f0 = u;
  P.f1 = v;
  P.xmapr = xmapr;
  P.ymapr = ymapr;
  P.in_level_stride = (long)in_level_stride;
  P.out_level_stride = (long)out_level_stride;
  P.undef = undef;
  return stencil_enqueue(c, "mifc_vortdiv_levels_enqueue", P, fdefined_in, n_undefined_dev);
}

int mifc_stencil_levels_enqueue(mifc_ctx* c, int op, int nx, int ny, int nlev, const float* f0, const float* f1, const float* xmapr, const float* ymapr,
                                const float* fcoriolis, float* out0, float* out1, const int* fdefined_in, float undef,
                                unsigned long long* n_undefined_dev)
{
  if (!c)
    return 0;
  enter(c);
  if (!((op >= mifc::ST_RELVORT && op <= mifc::ST_IGWIND) || op == mifc::ST_JACOBIAN) || !f0 || nx < 3 || ny < 3 || nlev < 1 || !xmapr || !ymapr)
    return 0;
  const bool wind = (op <= mifc::ST_VORTDIV) || op == mifc::ST_JACOBIAN; // two input fields per level
  const bool needs_fc = op == mifc::ST_ABSVORT || (op >= mifc::ST_GWIND_X && op <= mifc::ST_IGWIND);
  if ((wind && !f1) || (needs_fc && !fcoriolis) || (op == mifc::ST_IGWIND && !out1))
    return 0;
  if (op == mifc::ST_VORTDIV ? (!out0 && !out1) : !out0)
    return 0;
  mifc::StencilParams P;
  std::memset(&P, 0, sizeof P);
  P.op = op;
  P.out0 = out0;
  P.out1 = (op == mifc::ST_VORTDIV || op == mifc::ST_IGWIND) ? out1 : nullptr;
  if (op == mifc::ST_VORTDIV && !out0) {
    P.op = mifc::ST_DIVERGENCE;
    P.out0 = out1;
    P.out1 = nullptr;
  } else if (op == mifc::ST_VORTDIV && !out1) {
    P.op = mifc::ST_RELVORT;
  }
  P.nx = nx;
  P.ny_global = ny;
  P.j0 = 0;
  P.ny_local = ny;
  P.nlev = nlev;
  P.f0 = f0;
  P.f1 = wind ? f1 : nullptr;
  P.xmapr = xmapr;
  P.ymapr = ymapr;
  P.fcoriolis = needs_fc ? fcoriolis : nullptr;
  P.in_level_stride = (long)nx * ny;
  P.out_level_stride = (long)nx * ny;
  P.undef = undef;
  return stencil_enqueue(c, "mifc_stencil_levels_enqueue", P, fdefined_in, n_undefined_dev);
}

} // extern "C"

// ---- fused derived variables on hybrid levels --------------------------------
namespace {

struct DerivedRequest
{
  const float *u, *v, *t, *h, *ps;
  const float *alevel, *blevel;
  float *ff, *temp, *hum, *hum2;
  const char *temp_unit, *hum_unit, *hum2_unit;
  int temp_compute, hum_compute, hum2_compute;
  float* dd; // extension output: wind direction
};

// hlevelhum's remaps (:1168-1182) for one humidity output; false = the reference returns false
bool derived_hum_variant(const char* unit, int compute, int* code, float* tdconv)
{
  if (compute <= 0 || compute >= 13) // :1168
    return false;
  if (compute > 8 && unit_is(unit, "celsius")) // :1174-1177
    compute -= 4;
  else if (compute > 4 && compute <= 8 && unit_is(unit, "kelvin"))
    compute += 4;
  *tdconv = (compute >= 9) ? K_T0 : 0; // :1181
  *code = 1 + hum_kind_ah(compute) + 4 * ((compute % 2 == 0) ? 1 : 0);
  return true;
}

// Validates like the per-level reference calls would, uploads the per-level scalars and launches (or,
// with prepared_only, hands the parameters to the host pipeline).  counts_dev: u64[5 * nlev], ff | temp | hum | hum2 | dd.
int derived_common(mifc_ctx* c, int nx, int ny, int nlev, const DerivedRequest& rq, const int* fdef_wind, const int* fdef_thermo, float undef,
                   u64* counts_dev, mifc::DerivedParams* prepared_only = nullptr)
{
  if (nlev < 1 || nx * ny <= 0)
    return 0;
  if (!rq.ff && !rq.temp && !rq.hum && !rq.hum2 && !rq.dd)
    return 0;
  const bool thermo = rq.temp || rq.hum || rq.hum2;
  const bool wind = rq.ff || rq.dd;
  mifc::DerivedParams P;
  std::memset(&P, 0, sizeof P);
  if (rq.temp) {
    int compute = rq.temp_compute;
    if (compute < 3) { // :1060-1065
      if (unit_is(rq.temp_unit, "celsius"))
        compute = 1;
      else if (unit_is(rq.temp_unit, "kelvin"))
        compute = 2;
    }
    if (compute < 1 || compute > 5) { // the reference leaves such cells unwritten (:1080-1090): not offered in the batched form
      c->err = "mifc_hlevel_derived_batch: temp_compute must be 1..5";
      return 0;
    }
    P.temp_compute = compute;
  }
  if (rq.hum && !derived_hum_variant(rq.hum_unit, rq.hum_compute, &P.hum_code, &P.hum_tdconv))
    return 0;
  if (rq.hum2 && !derived_hum_variant(rq.hum2_unit, rq.hum2_compute, &P.td_code, &P.td_tdconv))
    return 0;
  if (thermo) {
    for (int l = 0; l < nlev; ++l)
      if (bad_hlevel(rq.alevel[l], rq.blevel[l])) // :1070, :1170
        return 0;
  }
  if (!ensure_levels(c, (size_t)nlev))
    return 0;
  P.n = nx * ny;
  P.nlev = nlev;
  P.u = rq.u;
  P.v = rq.v;
  P.t = rq.t;
  P.h = rq.h;
  P.ps = rq.ps;
  P.ff = rq.ff;
  P.temp = rq.temp;
  P.hum = rq.hum;
  P.td = rq.hum2;
  P.dd = rq.dd;
  P.undef = undef;
  P.cnt_ff = counts_dev;
  P.cnt_temp = counts_dev + nlev;
  P.cnt_hum = counts_dev + 2 * (size_t)nlev;
  P.cnt_td = counts_dev + 3 * (size_t)nlev;
  P.cnt_dd = counts_dev + 4 * (size_t)nlev;
  bool every_all = true;
  if (nlev <= 8 && !prepared_only) {
    // small batch: per-level scalars travel in the kernel arguments
    P.n_inline = 1;
    for (int l = 0; l < nlev; ++l) {
      const bool w = !wind || (fdef_wind && fdef_wind[l] == MIFC_ALL_DEFINED);
      const bool th = !thermo || (fdef_thermo && fdef_thermo[l] == MIFC_ALL_DEFINED);
      P.wind_inline[l] = w ? 1 : 0;
      P.thermo_inline[l] = th ? 1 : 0;
      P.a_inline[l] = thermo ? rq.alevel[l] : 0.f;
      P.b_inline[l] = thermo ? rq.blevel[l] : 0.f;
      every_all = every_all && w && th;
    }
  } else {
    if (!pinned_acquire(c))
      return 0;
    unsigned char* hf = pinned_flags(c);
    for (int l = 0; l < nlev; ++l) {
      const bool w = !wind || (fdef_wind && fdef_wind[l] == MIFC_ALL_DEFINED);
      const bool th = !thermo || (fdef_thermo && fdef_thermo[l] == MIFC_ALL_DEFINED);
      hf[l] = w ? 1 : 0;
      hf[c->cap_lev + l] = th ? 1 : 0;
      every_all = every_all && w && th;
    }
    float* hab = pinned_ab(c);
    for (int l = 0; l < nlev; ++l) {
      hab[l] = thermo ? rq.alevel[l] : 0.f;
      hab[c->cap_lev + l] = thermo ? rq.blevel[l] : 0.f;
    }
    MIFC_HIP(c, hipMemcpyAsync(c->d_ab, hab, 2 * c->cap_lev * sizeof(float), hipMemcpyHostToDevice, c->stream));
    MIFC_HIP(c, hipMemcpyAsync(c->d_flags, hf, 2 * c->cap_lev, hipMemcpyHostToDevice, c->stream));
    if (!pinned_release(c))
      return 0;
    P.alevel = c->d_ab;
    P.blevel = c->d_ab + c->cap_lev;
    P.wind_all_defined = c->d_flags;
    P.thermo_all_defined = c->d_flags + c->cap_lev;
  }
  P.every_level_all_defined = every_all ? 1 : 0;
  if (!(c->counts_accumulate && !prepared_only && counts_dev != c->d_counts)) // (accumulate mode: the caller zeroed its counters)
    MIFC_HIP(c, hipMemsetAsync(counts_dev, 0, 5 * sizeof(u64) * (size_t)nlev, c->stream));
  if (prepared_only) { // the caller launches chunk by chunk (host pipeline)
    *prepared_only = P;
  } else {
    MIFC_LAUNCH(c, mifc::launch_derived_levels(P, c->stream));
    if (!P.n_inline && !scratch_release(c)) // the kernel reads c->d_flags and c->d_ab
      return 0;
  }
  return 1;
}

void derived_flags(const u64* cnt, int nlev, size_t n, const DerivedRequest& rq, int* fdef_ff, int* fdef_temp, int* fdef_hum, int* fdef_hum2,
                   int* fdef_dd)
{
  for (int l = 0; l < nlev; ++l) {
    if (rq.ff && fdef_ff)
      fdef_ff[l] = mifc_classify(cnt[l], (u64)n);
    if (rq.temp && fdef_temp)
      fdef_temp[l] = mifc_classify(cnt[nlev + l], (u64)n);
    if (rq.hum && fdef_hum)
      fdef_hum[l] = mifc_classify(cnt[2 * (size_t)nlev + l], (u64)n);
    if (rq.hum2 && fdef_hum2)
      fdef_hum2[l] = mifc_classify(cnt[3 * (size_t)nlev + l], (u64)n);
    if (rq.dd && fdef_dd)
      fdef_dd[l] = mifc_classify(cnt[4 * (size_t)nlev + l], (u64)n);
  }
}

int derived_sync(mifc_ctx* c, int nx, int ny, int nlev, const DerivedRequest& rq0, const int* fdef_wind, const int* fdef_thermo, int* fdef_ff,
                 int* fdef_temp, int* fdef_hum, int* fdef_hum2, int* fdef_dd, float undef, int memkind)
{
  if (nlev < 1 || nx * ny <= 0)
    return 0;
  if ((nx * ny) % 4 != 0) {
    c->err = "mifc_hlevel_derived_batch: nx*ny must be a multiple of 4 (use the per-field operators otherwise)";
    return 0;
  }
  const size_t n = (size_t)nx * ny, nb = n * (size_t)nlev;
  const bool thermo = rq0.temp || rq0.hum || rq0.hum2;
  const bool humid = rq0.hum || rq0.hum2;
  const bool wind = rq0.ff || rq0.dd;
  DerivedRequest rq = rq0;
  bool ok = true;
  // (the chunked pipeline carries four outputs; a request with the wind direction on top is staged whole)
  if (memkind == MIFC_MEM_HOST && mifc::hostpipe_chunk_levels(n, nlev) > 0 && host_pipeline_enabled() && !(rq0.dd && rq0.ff && rq0.temp && rq0.hum && rq0.hum2)) {
    // a large batch in host memory: chunks of levels stream through the device, copies
    // in both directions overlapping the kernels (mifc_hostpipe.h)
    if (!c->pipe && !(c->pipe = mifc::hostpipe_create(c->device))) {
      c->err = "host pipeline: cannot create streams";
      return 0;
    }
    rq.ps = thermo ? stage_in(c, 4, rq0.ps, n, memkind, &ok) : nullptr;
    if (!ok || !ensure_levels(c, (size_t)nlev))
      return 0;
    mifc::DerivedParams base;
    // the host pointers are placeholders that mark which fields take part; the chunk launcher substitutes device buffers
    rq.u = wind ? rq0.u : nullptr;
    rq.v = wind ? rq0.v : nullptr;
    rq.t = thermo ? rq0.t : nullptr;
    rq.h = humid ? rq0.h : nullptr;
    if (!derived_common(c, nx, ny, nlev, rq, fdef_wind, fdef_thermo, undef, c->d_counts, &base))
      return 0;
    MIFC_HIP(c, hipStreamSynchronize(c->stream)); // ps, flags, level coefficients, zeroed counters are in place
    const float* h_in[4];
    int slot_u = -1, slot_v = -1, slot_t = -1, slot_h = -1, n_in = 0;
    if (wind) {
      slot_u = n_in;
      h_in[n_in++] = rq0.u;
      slot_v = n_in;
      h_in[n_in++] = rq0.v;
    }
    if (thermo) {
      slot_t = n_in;
      h_in[n_in++] = rq0.t;
    }
    if (humid) {
      slot_h = n_in;
      h_in[n_in++] = rq0.h;
    }
    // the (at most four) requested outputs share the pipeline's four output slots
    float* all_out[5] = {rq0.ff, rq0.temp, rq0.hum, rq0.hum2, rq0.dd};
    float* h_out[4] = {nullptr, nullptr, nullptr, nullptr};
    int out_slot[5] = {-1, -1, -1, -1, -1}, n_out = 0;
    for (int k = 0; k < 5; ++k)
      if (all_out[k]) {
        out_slot[k] = n_out;
        h_out[n_out++] = all_out[k];
      }
    const mifc::ChunkLaunch launch = [&](int l0, int nl, const float* const* d_in, float* const* d_out, hipStream_t stream) {
      mifc::DerivedParams p = base;
      p.nlev = nl;
      p.u = slot_u >= 0 ? d_in[slot_u] : nullptr;
      p.v = slot_v >= 0 ? d_in[slot_v] : nullptr;
      p.t = slot_t >= 0 ? d_in[slot_t] : nullptr;
      p.h = slot_h >= 0 ? d_in[slot_h] : nullptr;
      p.ff = out_slot[0] >= 0 ? d_out[out_slot[0]] : nullptr;
      p.temp = out_slot[1] >= 0 ? d_out[out_slot[1]] : nullptr;
      p.hum = out_slot[2] >= 0 ? d_out[out_slot[2]] : nullptr;
      p.td = out_slot[3] >= 0 ? d_out[out_slot[3]] : nullptr;
      p.dd = out_slot[4] >= 0 ? d_out[out_slot[4]] : nullptr;
      p.alevel = base.alevel + l0;
      p.blevel = base.blevel + l0;
      p.wind_all_defined = base.wind_all_defined + l0;
      p.thermo_all_defined = base.thermo_all_defined + l0;
      p.cnt_ff = base.cnt_ff + l0;
      p.cnt_temp = base.cnt_temp + l0;
      p.cnt_hum = base.cnt_hum + l0;
      p.cnt_td = base.cnt_td + l0;
      p.cnt_dd = base.cnt_dd + l0;
      return mifc::launch_derived_levels(p, stream);
    };
    if (!mifc::hostpipe_run(c->pipe, n, nlev, n_in, h_in, 4, h_out, launch, &c->err))
      return 0;
  } else {
    rq.u = wind ? stage_in(c, 0, rq0.u, nb, memkind, &ok) : nullptr;
    rq.v = wind ? stage_in(c, 1, rq0.v, nb, memkind, &ok) : nullptr;
    rq.t = thermo ? stage_in(c, 2, rq0.t, nb, memkind, &ok) : nullptr;
    rq.h = humid ? stage_in(c, 3, rq0.h, nb, memkind, &ok) : nullptr;
    rq.ps = thermo ? stage_in(c, 4, rq0.ps, n, memkind, &ok) : nullptr;
    rq.ff = stage_out(c, 5, rq0.ff, nb, memkind, &ok);
    rq.temp = stage_out(c, 6, rq0.temp, nb, memkind, &ok);
    rq.hum = stage_out(c, 7, rq0.hum, nb, memkind, &ok);
    rq.hum2 = stage_out(c, 8, rq0.hum2, nb, memkind, &ok);
    rq.dd = stage_out(c, 9, rq0.dd, nb, memkind, &ok);
    if (!ok || !ensure_levels(c, (size_t)nlev))
      return 0;
    if (!derived_common(c, nx, ny, nlev, rq, fdef_wind, fdef_thermo, undef, c->d_counts))
      return 0;
    if (!fetch_out(c, 5, rq0.ff, nb, memkind) || !fetch_out(c, 6, rq0.temp, nb, memkind) || !fetch_out(c, 7, rq0.hum, nb, memkind) ||
        !fetch_out(c, 8, rq0.hum2, nb, memkind) || !fetch_out(c, 9, rq0.dd, nb, memkind))
      return 0;
  }
  MIFC_HIP(c, hipMemcpyAsync(pinned_counts(c), c->d_counts, 5 * sizeof(u64) * (size_t)nlev, hipMemcpyDeviceToHost, c->stream));
  MIFC_HIP(c, hipStreamSynchronize(c->stream));
  derived_flags(pinned_counts(c), nlev, n, rq0, fdef_ff, fdef_temp, fdef_hum, fdef_hum2, fdef_dd);
  return 1;
}

} // namespace

extern "C" {

int mifc_hlevel_derived_batch(mifc_ctx* c, int nx, int ny, int nlev, const float* u, const float* v, const float* t, const float* h, const float* ps,
                              const float* alevel, const float* blevel, float* ff, float* temp, const char* temp_unit, int temp_compute, float* hum,
                              const char* hum_unit, int hum_compute, float* hum2, const char* hum2_unit, int hum2_compute, float* dd,
                              const int* fdef_wind, const int* fdef_thermo, int* fdef_ff, int* fdef_temp, int* fdef_hum, int* fdef_hum2, int* fdef_dd,
                              float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  const DerivedRequest rq = {u, v, t, h, ps, alevel, blevel, ff, temp, hum, hum2, temp_unit, hum_unit, hum2_unit, temp_compute, hum_compute, hum2_compute, dd};
  return derived_sync(c, nx, ny, nlev, rq, fdef_wind, fdef_thermo, fdef_ff, fdef_temp, fdef_hum, fdef_hum2, fdef_dd, undef, memkind);
}

int mifc_hlevel_derived_batch_enqueue(mifc_ctx* c, int nx, int ny, int nlev, const float* u, const float* v, const float* t, const float* h,
                                      const float* ps, const float* alevel, const float* blevel, float* ff, float* temp, const char* temp_unit,
                                      int temp_compute, float* hum, const char* hum_unit, int hum_compute, float* hum2, const char* hum2_unit,
                                      int hum2_compute, float* dd, const int* fdef_wind, const int* fdef_thermo, float undef,
                                      unsigned long long* n_undefined_dev)
{
  if (!c || !n_undefined_dev)
    return 0;
  enter(c);
  if ((nx * ny) % 4 != 0) {
    c->err = "mifc_hlevel_derived_batch: nx*ny must be a multiple of 4 (use the per-field operators otherwise)";
    return 0;
  }
  const DerivedRequest rq = {u, v, t, h, ps, alevel, blevel, ff, temp, hum, hum2, temp_unit, hum_unit, hum2_unit, temp_compute, hum_compute, hum2_compute, dd};
  return derived_common(c, nx, ny, nlev, rq, fdef_wind, fdef_thermo, undef, n_undefined_dev);
}

// The original trio: ff, RH (hlevelhum compute 1), theta (hleveltemp compute 3).  n_undefined_dev keeps its
// documented layout u64[3 * nlev] = ff | rh | theta: the counters are collected in the context's own
// 4-array scratch and copied out in that order on the stream.
int mifc_hlevel_derived_levels_enqueue(mifc_ctx* c, int nx, int ny, int nlev, const float* u, const float* v, const float* t, const float* q,
                                       const float* ps, const float* alevel, const float* blevel, float* ff, float* rh, float* theta,
                                       const int* fdef_wind, const int* fdef_thermo, float undef, unsigned long long* n_undefined_dev)
{
  if (!c || !n_undefined_dev)
    return 0;
  enter(c);
  if ((nx * ny) % 4 != 0) {
    c->err = "mifc_hlevel_derived_levels: nx*ny must be a multiple of 4 (use the per-field operators otherwise)";
    return 0;
  }
  if (nlev < 1 || !ensure_levels(c, (size_t)nlev))
    return 0;
  const DerivedRequest rq = {u, v, t, q, ps, alevel, blevel, ff, theta, rh, nullptr, "", "", "", 3, 1, 0, nullptr};
  if (!derived_common(c, nx, ny, nlev, rq, fdef_wind, fdef_thermo, undef, c->d_counts))
    return 0;
  const size_t row = sizeof(u64) * (size_t)nlev;
  MIFC_HIP(c, hipMemcpyAsync(n_undefined_dev, c->d_counts, row, hipMemcpyDeviceToDevice, c->stream));                      // ff
  MIFC_HIP(c, hipMemcpyAsync(n_undefined_dev + nlev, c->d_counts + 2 * (size_t)nlev, row, hipMemcpyDeviceToDevice, c->stream)); // rh  <- hum
  MIFC_HIP(c, hipMemcpyAsync(n_undefined_dev + 2 * (size_t)nlev, c->d_counts + nlev, row, hipMemcpyDeviceToDevice, c->stream)); // theta <- temp
  return scratch_release(c) ? 1 : 0; // the copies read c->d_counts
}

int mifc_hlevel_derived_levels(mifc_ctx* c, int nx, int ny, int nlev, const float* u, const float* v, const float* t, const float* q, const float* ps,
                               const float* alevel, const float* blevel, float* ff, float* rh, float* theta, const int* fdef_wind,
                               const int* fdef_thermo, int* fdef_ff, int* fdef_rh, int* fdef_theta, float undef, int memkind)
{
  if (!c)
    return 0;
  enter(c);
  const DerivedRequest rq = {u, v, t, q, ps, alevel, blevel, ff, theta, rh, nullptr, "", "", "", 3, 1, 0, nullptr};
  return derived_sync(c, nx, ny, nlev, rq, fdef_wind, fdef_thermo, fdef_ff, fdef_theta, fdef_rh, nullptr, nullptr, undef, memkind);
}

int mifc_vortdiv_slab_enqueue(mifc_ctx* c, int nx, int ny_global, int j0, int ny_local, const float* u_halo, const float* v_halo, const float* xmapr,
                              const float* ymapr, float* rvort, float* diverg, int fdefined_in, float undef, unsigned long long* n_undefined_dev)
{
  return mifc_vortdiv_slab_rows_enqueue(c, nx, ny_global, j0, ny_local, 0, ny_local, u_halo, v_halo, xmapr, ymapr, rvort, diverg, fdefined_in, undef,
                                        n_undefined_dev, 0);
}

int mifc_vortdiv_slab_rows_enqueue(mifc_ctx* c, int nx, int ny_global, int j0, int ny_local, int row_begin, int row_end, const float* u_halo,
                                   const float* v_halo, const float* xmapr, const float* ymapr, float* rvort, float* diverg, int fdefined_in,
                                   float undef, unsigned long long* n_undefined_dev, int accumulate_count)
{
  if (!c || (!rvort && !diverg))
    return 0;
  enter(c);
  if (nx < 3 || ny_global < 3 || ny_local < 1 || j0 < 0 || j0 + ny_local > ny_global)
    return 0;
  // a slab that owns a global edge row must also own the row it is filled from
  if ((j0 == 0 || j0 + ny_local == ny_global) && ny_local < 2)
    return 0;
  if (row_begin < 0 || row_end > ny_local || row_begin >= row_end)
    return 0;
  // ... and a row range must keep the two together (fillEdges copies row 1 to row 0, row ny-2 to row ny-1)
  if ((j0 == 0 && (row_begin == 1 || row_end == 1)) || (j0 + ny_local == ny_global && (row_begin == ny_local - 1 || row_end == ny_local - 1))) {
    c->err = "mifc_vortdiv_slab_rows_enqueue: a row range must not separate a global edge row from the row it is filled from";
    return 0;
  }
  mifc::StencilParams P;
  std::memset(&P, 0, sizeof P);
  P.op = mifc::ST_VORTDIV;
  P.out0 = rvort;
  P.out1 = diverg;
  if (!rvort) {
    P.op = mifc::ST_DIVERGENCE;
    P.out0 = diverg;
    P.out1 = nullptr;
  } else if (!diverg) {
    P.op = mifc::ST_RELVORT;
  }
  P.nx = nx;
  P.ny_global = ny_global;
  P.j0 = j0;
  P.ny_local = ny_local;
  P.nlev = 1;
  P.f0 = u_halo + nx; // owned row 0; halo rows sit directly before and after
  P.f1 = v_halo + nx;
  P.xmapr = xmapr;
  P.ymapr = ymapr;
  P.undef = undef;
  P.every_level_all_defined = (fdefined_in == MIFC_ALL_DEFINED) ? 1 : 0;
  P.all_defined = nullptr;
  P.n_undefined = n_undefined_dev;
  if (row_begin != 0 || row_end != ny_local) {
    P.row_begin = row_begin;
    P.row_end = row_end;
  }
  if (!P.every_level_all_defined && !n_undefined_dev) {
    c->err = "mifc_vortdiv_slab_enqueue: n_undefined_dev is required unless the input is ALL_DEFINED";
    return 0;
  }
  if (n_undefined_dev && !accumulate_count)
    MIFC_HIP(c, hipMemsetAsync(n_undefined_dev, 0, sizeof(u64), c->stream));
  MIFC_LAUNCH(c, mifc::launch_stencil(P, c->stream));
  return 1;
}

int mifc_halo_copy_enqueue(mifc_ctx* dst_ctx, float* dst_dev, mifc_ctx* src_ctx, const float* src_dev, size_t n_floats)
{
  if (!dst_ctx || !src_ctx || !dst_dev || !src_dev)
    return 0;
  enter(src_ctx);
  if (n_floats == 0)
    return 1;
  // the rows must have been produced: order the copy after what is queued on the source context's stream
  if (src_ctx != dst_ctx || src_ctx->stream != dst_ctx->stream) {
    hipEvent_t ready = nullptr;
    MIFC_HIP(src_ctx, hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ready, src_ctx->stream);
    enter(dst_ctx);
    if (e == hipSuccess)
      e = hipStreamWaitEvent(dst_ctx->stream, ready, 0);
    (void)hipEventDestroy(ready); // released once the wait has consumed it
    if (e != hipSuccess) {
      fail(dst_ctx, "halo copy: event", e);
      return 0;
    }
  }
  enter(dst_ctx);
  if (src_ctx->device == dst_ctx->device)
    MIFC_HIP(dst_ctx, hipMemcpyAsync(dst_dev, src_dev, n_floats * sizeof(float), hipMemcpyDeviceToDevice, dst_ctx->stream));
  else
    MIFC_HIP(dst_ctx, hipMemcpyPeerAsync(dst_dev, dst_ctx->device, src_dev, src_ctx->device, n_floats * sizeof(float), dst_ctx->stream));
  return 1;
}

#ifdef MIFC_MEASUREMENT_BUILD // libmifc_measure.so only (include/mifc_measure.h)
int mifc_bench_stream2(mifc_ctx* c, int variant, int blocks, float* dst0, float* dst1, const float* src0, const float* src1, size_t n_floats)
{
  if (!c)
    return 0;
  enter(c);
  if (n_floats % 4 != 0)
    return 0;
  MIFC_HIP(c, mifc::launch_stream2(variant, blocks, dst0, dst1, src0, src1, n_floats, c->stream));
  return 1;
}

int mifc_diag_division(mifc_ctx* c, const float* a, const float* b, const float* g, float* shared, float* plain, size_t n)
{
  if (!c)
    return 0;
  enter(c);
  MIFC_HIP(c, mifc::launch_division_check(a, b, g, shared, plain, n, c->stream));
  return 1;
}
#endif // MIFC_MEASUREMENT_BUILD

} // extern "C"
