// mifc_slab.hip -- one decomposed step of a horizontally split field (BASELINE.json config 4) as ONE call
// of the C ABI: halo exchange over RCCL, interior rows while it is in flight, boundary strips, global undefined counts.
//
// The reference has no distributed code; what a decomposed caller needs around relvort / divergence
// (FieldCalculations.cc:1843-1940) is: one row of u and of v from each neighbour BEFORE the rows next to a slab boundary
// can be computed, and the sum of the slabs' undefined counts to classify the whole field (checkDefined, :1868).
//
//   stream S (the context's)                        stream C (the plan's)
//   zero the counters (tested input only)
//   record fork ------------------------------------> wait fork
//   interior rows [2, ny_local - 2) of every level    ncclGroupStart; per level and field: send row 1 / recv row 0 with the
//      (they read no halo row)                         rank above, send row ny_local / recv row ny_local + 1 with the rank
//                                                      below; ncclGroupEnd          (nx * 4 bytes each, nearest neighbour)
//   wait join <-------------------------------------- record join
//   strip [0, 2) and strip [ny_local - 2, ny_local)   (only where a neighbour exists; a strip keeps rows 0/1 and ny-2/ny-1
//   ncclAllReduce(counts, nlev x u64, sum)             of the WHOLE field together: fillEdges copies one from the other)
//
// The sequence is captured ONCE into a HIP graph and replayed per step (one hipGraphLaunch instead of ~10 runtime calls and
// an RCCL group on the host: a 500 x 4000 slab is 13 us of kernel time); MIFC_SLAB_GRAPH=0, or a capture the runtime
// refuses, falls back to enqueuing the sequence directly.  A slab is a level BATCH ([nlev][ny_local + 2][nx]): one exchange
// of nlev rows per neighbour and field amortises the step over the levels.
#include "mifc_ctx.h"
#include "mifc_rccl.h"

#include <cstring>
#include <new>

using namespace mifc_host;

struct mifc_slab_plan
{
  mifc_ctx* c = nullptr;
  int nx = 0, nyg = 0, j0 = 0, nyl = 0, nlev = 0;
  float *u = nullptr, *v = nullptr; // [nlev][nyl + 2][nx]
  const float *xm = nullptr, *ym = nullptr;
  float *rv = nullptr, *dv = nullptr; // [nlev][nyl][nx]
  int fdef = MIFC_SOME_DEFINED;
  float undef = 0.f;
  u64* counts = nullptr; // device u64[nlev]
  bool north = false, south = false; // a neighbouring slab exists above / below
  hipStream_t comm_stream = nullptr, capture_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool want_graph = true, graph_failed = false;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  // one tested level of a big slab: the interior launch's workgroups leave their counts here (StencilParams::partials); the
  // plan's own buffer, so the launch can be part of the plan's graph
  unsigned int* d_partials = nullptr;
  int partials_cap = 0;
};

namespace {

constexpr int kStrip = 2; // rows of a boundary strip

bool rccl_ok(mifc_ctx* c, const mifc::RcclApi* api, ncclResult_t r, const char* what)
{
  if (r == ncclSuccess)
    return true;
  c->err = std::string("RCCL: ") + what + ": " + (api ? api->GetErrorString(r) : "?");
  return false;
}

int launch_rows(mifc_slab_plan* p, int row_begin, int row_end)
{
  mifc_ctx* c = p->c;
  mifc::StencilParams P;
  std::memset(&P, 0, sizeof P);
  P.op = mifc::ST_VORTDIV;
  P.out0 = p->rv;
  P.out1 = p->dv;
  if (!p->rv) {
    P.op = mifc::ST_DIVERGENCE;
    P.out0 = p->dv;
    P.out1 = nullptr;
  } else if (!p->dv) {
    P.op = mifc::ST_RELVORT;
  }
  P.nx = p->nx;
  P.ny_global = p->nyg;
  P.j0 = p->j0;
  P.ny_local = p->nyl;
  P.nlev = p->nlev;
  P.f0 = p->u + p->nx; // owned row 0 of level 0; the halo rows sit directly before and after the owned rows of a level
  P.f1 = p->v + p->nx;
  P.xmapr = p->xm;
  P.ymapr = p->ym;
  P.in_level_stride = (long)(p->nyl + 2) * p->nx;
  P.out_level_stride = (long)p->nyl * p->nx;
  P.undef = p->undef;
  P.every_level_all_defined = (p->fdef == MIFC_ALL_DEFINED) ? 1 : 0;
  P.all_defined = nullptr;
  P.n_undefined = p->counts;
  P.partials = p->d_partials; // the launches of a step follow each other on one stream
  P.partials_cap = p->partials_cap;
  if (row_begin != 0 || row_end != p->nyl) {
    P.row_begin = row_begin;
    P.row_end = row_end;
  }
  MIFC_LAUNCH(c, mifc::launch_stencil(P, c->stream));
  return 1;
}

// rows exchanged with the neighbours: per level and field, in the same order on both sides (RCCL matches the sends to a
// peer with that peer's receives in issue order)
int exchange(mifc_slab_plan* p, hipStream_t on)
{
  mifc_ctx* c = p->c;
  const char* why = nullptr;
  const mifc::RcclApi* api = mifc::rccl_api(&why);
  if (!api) {
    c->err = why ? why : "RCCL not available";
    return 0;
  }
  if (!c->comm) {
    c->err = "mifc_slab_plan_step: the slab has neighbours but the context has no communicator (mifc_comm_init / mifc_comm_adopt)";
    return 0;
  }
  ncclComm_t comm = static_cast<ncclComm_t>(c->comm);
  const size_t nx = (size_t)p->nx, ls = (size_t)(p->nyl + 2) * nx;
  if (!rccl_ok(c, api, api->GroupStart(), "ncclGroupStart"))
    return 0;
  bool ok = true;
  for (int l = 0; l < p->nlev && ok; ++l) {
    float* f[2] = {p->u + (size_t)l * ls, p->v + (size_t)l * ls};
    for (int k = 0; k < 2 && ok; ++k) {
      if (p->north) { // my first owned row goes up, the row above it comes down
        ok = ok && rccl_ok(c, api, api->Send(f[k] + nx, nx, ncclFloat32, c->comm_rank - 1, comm, on), "ncclSend");
        ok = ok && rccl_ok(c, api, api->Recv(f[k], nx, ncclFloat32, c->comm_rank - 1, comm, on), "ncclRecv");
      }
      if (p->south) {
        ok = ok && rccl_ok(c, api, api->Send(f[k] + (size_t)p->nyl * nx, nx, ncclFloat32, c->comm_rank + 1, comm, on), "ncclSend");
        ok = ok && rccl_ok(c, api, api->Recv(f[k] + (size_t)(p->nyl + 1) * nx, nx, ncclFloat32, c->comm_rank + 1, comm, on), "ncclRecv");
      }
    }
  }
  const ncclResult_t r = api->GroupEnd(); // always closes the group
  return (ok && rccl_ok(c, api, r, "ncclGroupEnd")) ? 1 : 0;
}

bool tested(const mifc_slab_plan* p)
{
  return p->fdef != MIFC_ALL_DEFINED;
}

int zero_counts(mifc_slab_plan* p)
{
  if (tested(p))
    MIFC_HIP(p->c, mifc::launch_prep_levels(nullptr, p->nlev, nullptr, p->counts, p->nlev, p->c->stream));
  return 1;
}

// the owned rows that read no halo row
void interior_range(const mifc_slab_plan* p, int* lo, int* hi)
{
  *lo = p->north ? kStrip : 0;
  *hi = p->south ? p->nyl - kStrip : p->nyl;
}

bool worth_splitting(const mifc_slab_plan* p)
{
  return (p->north || p->south) && p->nyl >= 3 * kStrip + 1;
}

int strips(mifc_slab_plan* p)
{
  if (!worth_splitting(p))
    return launch_rows(p, 0, p->nyl);
  if (p->north && !launch_rows(p, 0, kStrip))
    return 0;
  if (p->south && !launch_rows(p, p->nyl - kStrip, p->nyl))
    return 0;
  return 1;
}

int enqueue_step(mifc_slab_plan* p)
{
  mifc_ctx* c = p->c;
  if (!zero_counts(p))
    return 0;
  if (!p->north && !p->south) // the whole field in one slab: nothing to exchange, nothing to reduce
    return launch_rows(p, 0, p->nyl);
  MIFC_HIP(c, hipEventRecord(p->ev_fork, c->stream)); // the rows to send are whatever the caller queued on the stream before
  MIFC_HIP(c, hipStreamWaitEvent(p->comm_stream, p->ev_fork, 0));
  if (!exchange(p, p->comm_stream))
    return 0;
  MIFC_HIP(c, hipEventRecord(p->ev_join, p->comm_stream));
  if (worth_splitting(p)) {
    int lo, hi;
    interior_range(p, &lo, &hi);
    if (!launch_rows(p, lo, hi))
      return 0;
  }
  MIFC_HIP(c, hipStreamWaitEvent(c->stream, p->ev_join, 0));
  if (!strips(p))
    return 0;
  if (tested(p) && c->comm_world > 1) { // every rank ends up with the whole field's counts
    const mifc::RcclApi* api = mifc::rccl_api(nullptr);
    if (!rccl_ok(c, api, api->AllReduce(p->counts, p->counts, (size_t)p->nlev, ncclUint64, ncclSum, static_cast<ncclComm_t>(c->comm), c->stream),
                 "ncclAllReduce"))
      return 0;
  }
  return 1;
}

void drop_graph(mifc_slab_plan* p)
{
  if (p->exec)
    (void)hipGraphExecDestroy(p->exec);
  if (p->graph)
    (void)hipGraphDestroy(p->graph);
  p->exec = nullptr;
  p->graph = nullptr;
}

bool graph_wanted()
{
  return mifc::env().slab_graph;
}

} // namespace

extern "C" {

int mifc_comm_unique_id(char* id_out)
{
  if (!id_out)
    return 0;
  const mifc::RcclApi* api = mifc::rccl_api(nullptr);
  if (!api)
    return 0;
  ncclUniqueId id;
  if (api->GetUniqueId(&id) != ncclSuccess)
    return 0;
  static_assert(sizeof id.internal == MIFC_COMM_ID_BYTES, "include/mifc.h and rccl.h disagree about the id size");
  std::memcpy(id_out, id.internal, MIFC_COMM_ID_BYTES);
  return 1;
}

int mifc_comm_release(mifc_ctx* c)
{
  if (!c)
    return 0;
  enter(c);
  if (c->comm && c->comm_owned) {
    (void)hipStreamSynchronize(c->stream);
    const mifc::RcclApi* api = mifc::rccl_api(nullptr);
    if (api)
      (void)api->CommDestroy(static_cast<ncclComm_t>(c->comm));
  }
  c->comm = nullptr;
  c->comm_owned = false;
  c->comm_rank = 0;
  c->comm_world = 1;
  return 1;
}

int mifc_comm_init(mifc_ctx* c, const char* id, int rank, int world)
{
  if (!c || !id || world < 1 || rank < 0 || rank >= world)
    return 0;
  enter(c);
  const char* why = nullptr;
  const mifc::RcclApi* api = mifc::rccl_api(&why);
  if (!api) {
    c->err = why ? why : "RCCL not available";
    return 0;
  }
  mifc_comm_release(c);
  ncclUniqueId uid;
  std::memcpy(uid.internal, id, MIFC_COMM_ID_BYTES);
  ncclComm_t comm = nullptr;
  if (!rccl_ok(c, api, api->CommInitRank(&comm, world, uid, rank), "ncclCommInitRank"))
    return 0;
  c->comm = comm;
  c->comm_owned = true;
  c->comm_rank = rank;
  c->comm_world = world;
  return 1;
}

int mifc_comm_adopt(mifc_ctx* c, void* nccl_comm)
{
  if (!c || !nccl_comm)
    return 0;
  enter(c);
  const char* why = nullptr;
  const mifc::RcclApi* api = mifc::rccl_api(&why);
  if (!api) {
    c->err = why ? why : "RCCL not available";
    return 0;
  }
  int rank = 0, world = 0;
  ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
  if (!rccl_ok(c, api, api->CommCount(comm, &world), "ncclCommCount") || !rccl_ok(c, api, api->CommUserRank(comm, &rank), "ncclCommUserRank"))
    return 0;
  mifc_comm_release(c);
  c->comm = nccl_comm;
  c->comm_owned = false;
  c->comm_rank = rank;
  c->comm_world = world;
  return 1;
}

int mifc_comm_info(const mifc_ctx* c, int* rank, int* world)
{
  if (!c)
    return 0;
  if (rank)
    *rank = c->comm_rank;
  if (world)
    *world = c->comm_world;
  return c->comm ? 1 : 0;
}

mifc_slab_plan* mifc_slab_plan_create(mifc_ctx* c, int nx, int ny_global, int j0, int ny_local, int nlev, float* u_halo, float* v_halo,
                                      const float* xmapr, const float* ymapr, float* rvort, float* diverg, int fdefined_in, float undef,
                                      unsigned long long* n_undefined_dev)
{
  if (!c)
    return nullptr;
  enter(c);
  if (!u_halo || !v_halo || !xmapr || !ymapr || (!rvort && !diverg) || nx < 3 || ny_global < 3 || ny_local < 1 || nlev < 1 || j0 < 0 ||
      j0 + ny_local > ny_global)
    return nullptr;
  // a slab that owns a global edge row must also own the row it is filled from (fillEdges, FieldCalculations.cc:70-73)
  if ((j0 == 0 || j0 + ny_local == ny_global) && ny_local < 2) {
    c->err = "mifc_slab_plan_create: a slab that owns row 0 or row ny-1 of the field needs at least two rows";
    return nullptr;
  }
  if (fdefined_in != MIFC_ALL_DEFINED && !n_undefined_dev) {
    c->err = "mifc_slab_plan_create: n_undefined_dev is required unless the input is ALL_DEFINED";
    return nullptr;
  }
  mifc_slab_plan* p = new (std::nothrow) mifc_slab_plan();
  if (!p)
    return nullptr;
  p->c = c;
  p->nx = nx;
  p->nyg = ny_global;
  p->j0 = j0;
  p->nyl = ny_local;
  p->nlev = nlev;
  p->u = u_halo;
  p->v = v_halo;
  p->xm = xmapr;
  p->ym = ymapr;
  p->rv = rvort;
  p->dv = diverg;
  p->fdef = fdefined_in;
  p->undef = undef;
  p->counts = n_undefined_dev;
  p->north = j0 > 0;
  p->south = j0 + ny_local < ny_global;
  p->want_graph = graph_wanted();
  if (fdefined_in != MIFC_ALL_DEFINED) {
    const size_t per_level = (size_t)(ny_local / 4 + 2) * (size_t)(nx / 256 + 1), units = per_level * (size_t)nlev;
    if (per_level >= 2048 && units <= ((size_t)1 << 24) && hipMalloc((void**)&p->d_partials, units * sizeof(unsigned int)) == hipSuccess)
      p->partials_cap = (int)units; // (no buffer: one atomic per workgroup and level, as before)
  }
  if (hipStreamCreateWithFlags(&p->comm_stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&p->capture_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming) != hipSuccess) {
    c->err = "mifc_slab_plan_create: cannot create the exchange stream / events";
    mifc_slab_plan_destroy(p);
    return nullptr;
  }
  return p;
}

void mifc_slab_plan_destroy(mifc_slab_plan* p)
{
  if (!p)
    return;
  enter(p->c);
  (void)hipStreamSynchronize(p->c->stream);
  drop_graph(p);
  if (p->comm_stream) {
    (void)hipStreamSynchronize(p->comm_stream);
    (void)hipStreamDestroy(p->comm_stream);
  }
  if (p->capture_stream)
    (void)hipStreamDestroy(p->capture_stream);
  if (p->ev_fork)
    (void)hipEventDestroy(p->ev_fork);
  if (p->ev_join)
    (void)hipEventDestroy(p->ev_join);
  if (p->d_partials)
    (void)hipFree(p->d_partials);
  delete p;
}

int mifc_slab_plan_step(mifc_slab_plan* p)
{
  if (!p)
    return 0;
  mifc_ctx* c = p->c;
  enter(c);
  if ((p->north || p->south) && (!c->comm || c->comm_rank - (p->north ? 1 : 0) < 0 || c->comm_rank + (p->south ? 1 : 0) >= c->comm_world)) {
    c->err = "mifc_slab_plan_step: the slab has neighbours the context's communicator does not reach (rank - 1 / rank + 1)";
    return 0;
  }
  if (p->want_graph && !p->exec && !p->graph_failed && !c->timing) {
    // capture the sequence once; what the runtime (or RCCL) refuses to capture runs directly from then on
    // (captured on a stream of the plan's own: the caller's may be the legacy default stream, which cannot capture; the
    // graph is launched on the caller's stream all the same)
    bool captured = false;
    hipStream_t callers = c->stream;
    c->stream = p->capture_stream;
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      const int ok = enqueue_step(p);
      hipGraph_t g = nullptr;
      const hipError_t e = hipStreamEndCapture(c->stream, &g);
      c->stream = callers;
      if (ok && e == hipSuccess && g && hipGraphInstantiate(&p->exec, g, nullptr, nullptr, 0) == hipSuccess) {
        p->graph = g;
        captured = true;
      } else {
        if (g)
          (void)hipGraphDestroy(g);
        p->exec = nullptr;
        (void)hipGetLastError();
        c->err.clear();
      }
    } else {
      c->stream = callers;
      (void)hipGetLastError();
    }
    if (!captured)
      p->graph_failed = true;
  }
  if (p->exec && !c->timing) {
    MIFC_HIP(c, hipGraphLaunch(p->exec, c->stream));
    return 1;
  }
  return enqueue_step(p);
}

int mifc_slab_plan_uses_graph(const mifc_slab_plan* p)
{
  return (p && p->exec) ? 1 : 0;
}

int mifc_slab_plan_begin(mifc_slab_plan* p)
{
  if (!p)
    return 0;
  enter(p->c);
  if (!zero_counts(p))
    return 0;
  if (!worth_splitting(p))
    return 1; // everything waits for the halo rows
  int lo, hi;
  interior_range(p, &lo, &hi);
  return launch_rows(p, lo, hi);
}

int mifc_slab_plan_finish(mifc_slab_plan* p)
{
  if (!p)
    return 0;
  enter(p->c);
  return strips(p);
}

} // extern "C"
