// mifc_graph.hip -- launch-bound work as ONE launch: a caller's sequence of *_enqueue calls captured into a HIP graph
// and replayed (host code only).
//
// The reference is called once per 2-D field; a caller that keeps that loop -- one level per call --
// pays a launch per level for 4 us of traffic (one 1440x720 level of the fused derived kernel is 33 MB).  Between
// mifc_graph_begin and mifc_graph_end the context's asynchronous entry points (mifc_*_enqueue, mifc_slab_plan_begin / finish)
// do not run: their launches are recorded, with the arguments of the recording calls; mifc_graph_launch replays the whole
// sequence with one runtime call.  Flags and per-level scalars travel in kernel arguments (launch_prep_levels, the
// derived kernel's inline scalars), so a replay needs nothing from the host.
#include "mifc_ctx.h"

#include <new>

using namespace mifc_host;

struct mifc_graph
{
  mifc_ctx* c = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int launches = 0;
};

extern "C" {

int mifc_graph_begin_lanes(mifc_ctx* c, int max_levels_per_call, int n_lanes);

int mifc_graph_begin(mifc_ctx* c, int max_levels_per_call)
{
  return mifc_graph_begin_lanes(c, max_levels_per_call, 1);
}

int mifc_graph_begin_lanes(mifc_ctx* c, int max_levels_per_call, int n_lanes)
{
  if (!c || n_lanes < 1 || n_lanes > 16)
    return 0;
  enter(c);
  if (c->capturing) {
    c->err = "mifc_graph_begin: a capture is already open on this context";
    return 0;
  }
  // nothing may be allocated while a stream captures: the per-level scratch is grown now
  if (!ensure_levels(c, (size_t)(max_levels_per_call > 0 ? max_levels_per_call : 256)))
    return 0;
  if (!c->capture_stream && hipStreamCreateWithFlags(&c->capture_stream, hipStreamNonBlocking) != hipSuccess) {
    c->err = "mifc_graph_begin: cannot create the capture stream";
    return 0;
  }
  MIFC_HIP(c, hipStreamSynchronize(c->stream)); // the recorded work starts from a quiet context
  c->stream_before_capture = c->stream;
  c->stream = c->capture_stream; // the caller's stream may be the legacy default stream, which cannot capture
  if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    c->stream = c->stream_before_capture;
    (void)hipGetLastError();
    c->err = "mifc_graph_begin: hipStreamBeginCapture failed";
    return 0;
  }
  c->capturing = true;
  c->n_lanes = n_lanes;
  c->lane = 0;
  // the other lanes fork from lane 0 here and join it in mifc_graph_end
  while ((int)c->lane_streams.size() < n_lanes) {
    hipStream_t s = nullptr;
    if (c->lane_streams.empty())
      s = c->capture_stream;
    else if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
      s = nullptr;
    c->lane_streams.push_back(s);
  }
  while ((int)c->lane_events.size() < n_lanes) {
    hipEvent_t e = nullptr;
    (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    c->lane_events.push_back(e);
  }
  bool ok = true;
  for (int k = 0; k < n_lanes; ++k)
    ok = ok && c->lane_streams[k] && c->lane_events[k];
  if (ok && n_lanes > 1) {
    ok = hipEventRecord(c->lane_events[0], c->stream) == hipSuccess;
    for (int k = 1; k < n_lanes && ok; ++k)
      ok = hipStreamWaitEvent(c->lane_streams[k], c->lane_events[0], 0) == hipSuccess;
  }
  if (!ok) {
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(c->stream, &g);
    if (g)
      (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    c->stream = c->stream_before_capture;
    c->capturing = false;
    c->err = "mifc_graph_begin: cannot set up the capture lanes";
    return 0;
  }
  return 1;
}

int mifc_graph_lane(mifc_ctx* c, int lane)
{
  if (!c)
    return 0;
  enter(c);
  if (!c->capturing || lane < 0 || lane >= c->n_lanes) {
    c->err = "mifc_graph_lane: no capture open, or no such lane";
    return 0;
  }
  c->lane = lane;
  c->stream = c->lane_streams[lane];
  return 1;
}

mifc_graph* mifc_graph_end(mifc_ctx* c)
{
  if (!c || !c->capturing)
    return nullptr;
  enter(c);
  hipGraph_t g = nullptr;
  hipError_t e = hipSuccess;
  for (int k = 1; k < c->n_lanes && e == hipSuccess; ++k) { // the lanes join lane 0
    e = hipEventRecord(c->lane_events[k], c->lane_streams[k]);
    if (e == hipSuccess)
      e = hipStreamWaitEvent(c->capture_stream, c->lane_events[k], 0);
  }
  const hipError_t e2 = hipStreamEndCapture(c->capture_stream, &g);
  if (e == hipSuccess)
    e = e2;
  c->stream = c->stream_before_capture;
  c->capturing = false;
  c->n_lanes = 1;
  c->lane = 0;
  c->scratch_read_pending = false; // the events recorded while capturing belong to the graph, not to the stream
  c->pinned_read_pending = false;
  if (e != hipSuccess || !g) {
    (void)hipGetLastError();
    c->err = "mifc_graph_end: the capture was invalidated (a call in between synchronised, allocated or copied from pageable memory?)";
    return nullptr;
  }
  mifc_graph* mg = new (std::nothrow) mifc_graph();
  if (!mg || hipGraphInstantiate(&mg->exec, g, nullptr, nullptr, 0) != hipSuccess) {
    (void)hipGraphDestroy(g);
    delete mg;
    c->err = "mifc_graph_end: hipGraphInstantiate failed";
    return nullptr;
  }
  mg->c = c;
  mg->graph = g;
  return mg;
}

int mifc_graph_launch(mifc_graph* g)
{
  if (!g || !g->exec)
    return 0;
  mifc_ctx* c = g->c;
  enter(c);
  if (c->capturing) {
    c->err = "mifc_graph_launch: a capture is open on this context";
    return 0;
  }
  MIFC_HIP(c, hipGraphLaunch(g->exec, c->stream));
  g->launches += 1;
  return 1;
}

void mifc_graph_destroy(mifc_graph* g)
{
  if (!g)
    return;
  enter(g->c);
  (void)hipStreamSynchronize(g->c->stream);
  if (g->exec)
    (void)hipGraphExecDestroy(g->exec);
  if (g->graph)
    (void)hipGraphDestroy(g->graph);
  delete g;
}

} // extern "C"
