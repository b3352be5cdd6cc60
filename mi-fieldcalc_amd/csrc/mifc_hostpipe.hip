// mifc_hostpipe.hip -- see mifc_hostpipe.h.
#include "mifc_hostpipe.h"
#include "mifc_env.h"

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace mifc {

namespace {

// A handful of threads that do nothing but memcpy between the caller's
// pageable memory and the pinned staging buffers.  copy() may be called from
// two threads at once (the feeding and the draining side of the pipeline).
class CopyPool
{
public:
  explicit CopyPool(int nthreads)
  {
    for (int t = 0; t < nthreads; ++t)
      workers_.emplace_back([this] { work(); });
  }
  ~CopyPool()
  {
    {
      std::lock_guard<std::mutex> g(m_);
      stop_ = true;
    }
    cv_job_.notify_all();
    for (auto& t : workers_)
      t.join();
  }
  void copy(void* dst, const void* src, size_t bytes)
  {
    const size_t slice = size_t(2) << 20;
    int pending = 0;
    {
      std::lock_guard<std::mutex> g(m_);
      for (size_t off = 0; off < bytes; off += slice) {
        Slice s;
        s.dst = static_cast<char*>(dst) + off;
        s.src = static_cast<const char*>(src) + off;
        s.bytes = (off + slice > bytes) ? bytes - off : slice;
        s.pending = &pending;
        jobs_.push_back(s);
        ++pending;
      }
    }
    cv_job_.notify_all();
    std::unique_lock<std::mutex> g(m_);
    cv_done_.wait(g, [&] { return pending == 0; });
  }

private:
  struct Slice
  {
    char* dst;
    const char* src;
    size_t bytes;
    int* pending; // guarded by m_
  };
  void work()
  {
    for (;;) {
      Slice s;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_job_.wait(g, [&] { return stop_ || !jobs_.empty(); });
        if (jobs_.empty())
          return;
        s = jobs_.front();
        jobs_.pop_front();
      }
      std::memcpy(s.dst, s.src, s.bytes);
      {
        std::lock_guard<std::mutex> g(m_);
        if (--*s.pending == 0)
          cv_done_.notify_all();
      }
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_job_, cv_done_;
  std::deque<Slice> jobs_;
  bool stop_ = false;
};

int copy_threads()
{
  // two per direction keep ahead of the link (profiles/r01/hostpath_probe.txt: 29 GB/s per
  // thread); more did not help in the sweep (profiles/r01/hostpipe_sweep.txt)
  int n = 4;
  if (env().host_threads > 0)
    n = env().host_threads;
  const int hw = (int)std::thread::hardware_concurrency();
  if (hw > 0 && n > hw)
    n = hw;
  return n < 1 ? 1 : n;
}

const int MAXF = 4; // fields per direction

} // namespace

struct HostPipe
{
  int device = 0;
  hipStream_t s_h2d = nullptr, s_cmp = nullptr, s_d2h = nullptr;
  hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_kernel[2] = {nullptr, nullptr}, ev_d2h[2] = {nullptr, nullptr};
  char* pinned = nullptr;
  size_t pinned_bytes = 0;
  char* dev = nullptr;
  size_t dev_bytes = 0;
  CopyPool pool;
  HostPipe()
      : pool(copy_threads())
  {
  }
};

HostPipe* hostpipe_create(int device)
{
  HostPipe* hp = nullptr;
  try { // the copy threads are started here; nothing may be thrown across the C ABI above us
    hp = new (std::nothrow) HostPipe();
  } catch (...) {
    return nullptr;
  }
  if (!hp)
    return nullptr;
  hp->device = device;
  bool ok = hipStreamCreateWithFlags(&hp->s_h2d, hipStreamNonBlocking) == hipSuccess &&
            hipStreamCreateWithFlags(&hp->s_cmp, hipStreamNonBlocking) == hipSuccess &&
            hipStreamCreateWithFlags(&hp->s_d2h, hipStreamNonBlocking) == hipSuccess;
  for (int b = 0; b < 2 && ok; ++b)
    ok = hipEventCreateWithFlags(&hp->ev_h2d[b], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&hp->ev_kernel[b], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&hp->ev_d2h[b], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    hostpipe_destroy(hp);
    return nullptr;
  }
  return hp;
}

void hostpipe_destroy(HostPipe* hp)
{
  if (!hp)
    return;
  for (int b = 0; b < 2; ++b) {
    if (hp->ev_h2d[b])
      (void)hipEventDestroy(hp->ev_h2d[b]);
    if (hp->ev_kernel[b])
      (void)hipEventDestroy(hp->ev_kernel[b]);
    if (hp->ev_d2h[b])
      (void)hipEventDestroy(hp->ev_d2h[b]);
  }
  if (hp->s_h2d)
    (void)hipStreamDestroy(hp->s_h2d);
  if (hp->s_cmp)
    (void)hipStreamDestroy(hp->s_cmp);
  if (hp->s_d2h)
    (void)hipStreamDestroy(hp->s_d2h);
  if (hp->pinned)
    (void)hipHostFree(hp->pinned);
  if (hp->dev)
    (void)hipFree(hp->dev);
  delete hp;
}

int hostpipe_chunk_levels(size_t n, int nlev)
{
  const size_t field_bytes = n * sizeof(float);
  if (field_bytes == 0 || (size_t)nlev * field_bytes < (size_t(64) << 20))
    return 0; // under 64 MiB per field the ramp-up costs more than the overlap gains
  size_t chunk_mib = 32; // per field and chunk
  if (env().host_chunk_mib > 0)
    chunk_mib = (size_t)env().host_chunk_mib;
  size_t lev = (chunk_mib << 20) / field_bytes;
  if (lev < 1)
    lev = 1;
  if (lev > (size_t)nlev / 4)
    lev = (size_t)nlev / 4;
  return lev < 1 ? 1 : (int)lev;
}

bool hostpipe_run(HostPipe* hp, size_t n, int nlev, int n_in, const float* const* in, int n_out, float* const* out, const ChunkLaunch& launch,
                  std::string* err)
{
  auto fail = [&](const char* what, hipError_t e) {
    char buf[256];
    std::snprintf(buf, sizeof buf, "host pipeline: %s: %s", what, hipGetErrorString(e));
    if (err)
      *err = buf;
    return false;
  };
  if (n_in > MAXF || n_out > MAXF)
    return fail("too many fields", hipErrorInvalidValue);
  const int L = hostpipe_chunk_levels(n, nlev);
  if (L < 1)
    return fail("batch too small", hipErrorInvalidValue);
  const size_t chunk_bytes = (size_t)L * n * sizeof(float);
  const size_t stride = (chunk_bytes + 255) & ~size_t(255);
  const size_t need = 2 * (size_t)(n_in + n_out) * stride;
  hipError_t e;
  if (hp->pinned_bytes < need) {
    if (hp->pinned)
      (void)hipHostFree(hp->pinned);
    hp->pinned = nullptr;
    hp->pinned_bytes = 0;
    if ((e = hipHostMalloc((void**)&hp->pinned, need, hipHostMallocDefault)) != hipSuccess)
      return fail("hipHostMalloc(staging)", e);
    hp->pinned_bytes = need;
  }
  if (hp->dev_bytes < need) {
    if (hp->dev)
      (void)hipFree(hp->dev);
    hp->dev = nullptr;
    hp->dev_bytes = 0;
    if ((e = hipMalloc((void**)&hp->dev, need)) != hipSuccess)
      return fail("hipMalloc(chunks)", e);
    hp->dev_bytes = need;
  }
  const float* d_in[2][MAXF];
  float* d_out[2][MAXF];
  char *p_in[2][MAXF], *p_out[2][MAXF];
  for (int b = 0; b < 2; ++b) {
    for (int k = 0; k < n_in; ++k) {
      const size_t off = ((size_t)b * (n_in + n_out) + k) * stride;
      d_in[b][k] = reinterpret_cast<const float*>(hp->dev + off);
      p_in[b][k] = hp->pinned + off;
    }
    for (int k = 0; k < n_out; ++k) {
      const size_t off = ((size_t)b * (n_in + n_out) + n_in + k) * stride;
      d_out[b][k] = out[k] ? reinterpret_cast<float*>(hp->dev + off) : nullptr;
      p_out[b][k] = hp->pinned + off;
    }
  }
  const int nchunks = (nlev + L - 1) / L;

  // feeder (this thread) <-> drainer hand-over
  std::mutex m;
  std::condition_variable cv;
  int enqueued = 0; // chunks whose D2H copies are in the d2h stream
  int drained = 0;  // chunks whose outputs are back in the caller's memory
  bool abort = false;
  hipError_t drain_error = hipSuccess;

  std::thread drainer;
  try {
    drainer = std::thread([&] {
    (void)hipSetDevice(hp->device);
    for (int k = 0; k < nchunks; ++k) {
      {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [&] { return abort || enqueued > k; });
        if (abort)
          return;
      }
      const int b = k & 1;
      const hipError_t de = hipEventSynchronize(hp->ev_d2h[b]);
      if (de != hipSuccess) {
        std::lock_guard<std::mutex> g(m);
        drain_error = de;
        abort = true;
        cv.notify_all();
        return;
      }
      const int l0 = k * L;
      const int nl = (nlev - l0 < L) ? nlev - l0 : L;
      const size_t bytes = (size_t)nl * n * sizeof(float);
      for (int o = 0; o < n_out; ++o)
        if (out[o])
          hp->pool.copy(out[o] + (size_t)l0 * n, p_out[b][o], bytes);
      {
        std::lock_guard<std::mutex> g(m);
        drained = k + 1;
      }
      cv.notify_all();
    }
    });
  } catch (...) {
    return fail("cannot start the drain thread", hipErrorOutOfMemory);
  }

  bool ok = true;
  const char* what = "";
  e = hipSuccess;
#define PIPE_CK(call)          \
  if (ok) {                    \
    e = (call);                \
    if (e != hipSuccess) {     \
      ok = false;              \
      what = #call;            \
    }                          \
  }
  for (int k = 0; k < nchunks && ok; ++k) {
    const int b = k & 1;
    const int l0 = k * L;
    const int nl = (nlev - l0 < L) ? nlev - l0 : L;
    const size_t bytes = (size_t)nl * n * sizeof(float);
    if (k >= 2)
      PIPE_CK(hipEventSynchronize(hp->ev_h2d[b])); // the H2D of chunk k-2 has left pinned_in[b]
    if (!ok)
      break;
    for (int i = 0; i < n_in; ++i)
      hp->pool.copy(p_in[b][i], in[i] + (size_t)l0 * n, bytes);
    if (k >= 2)
      PIPE_CK(hipStreamWaitEvent(hp->s_h2d, hp->ev_kernel[b], 0)); // kernel k-2 has read dev_in[b]
    for (int i = 0; i < n_in; ++i)
      PIPE_CK(hipMemcpyAsync(const_cast<float*>(d_in[b][i]), p_in[b][i], bytes, hipMemcpyHostToDevice, hp->s_h2d));
    PIPE_CK(hipEventRecord(hp->ev_h2d[b], hp->s_h2d));
    PIPE_CK(hipStreamWaitEvent(hp->s_cmp, hp->ev_h2d[b], 0));
    if (k >= 2)
      PIPE_CK(hipStreamWaitEvent(hp->s_cmp, hp->ev_d2h[b], 0)); // D2H k-2 has read dev_out[b]
    PIPE_CK(launch(l0, nl, d_in[b], d_out[b], hp->s_cmp));
    PIPE_CK(hipEventRecord(hp->ev_kernel[b], hp->s_cmp));
    if (k >= 2) { // pinned_out[b] still holds chunk k-2 until the drainer has copied it out
      std::unique_lock<std::mutex> g(m);
      cv.wait(g, [&] { return abort || drained >= k - 1; });
      if (abort)
        break;
    }
    PIPE_CK(hipStreamWaitEvent(hp->s_d2h, hp->ev_kernel[b], 0));
    for (int o = 0; o < n_out; ++o)
      if (out[o])
        PIPE_CK(hipMemcpyAsync(p_out[b][o], d_out[b][o], bytes, hipMemcpyDeviceToHost, hp->s_d2h));
    PIPE_CK(hipEventRecord(hp->ev_d2h[b], hp->s_d2h));
    if (ok) {
      std::lock_guard<std::mutex> g(m);
      enqueued = k + 1;
    }
    cv.notify_all();
  }
#undef PIPE_CK
  if (!ok) {
    std::lock_guard<std::mutex> g(m);
    abort = true;
  }
  cv.notify_all();
  drainer.join();
  // leave nothing in flight, whatever happened
  const hipError_t e1 = hipStreamSynchronize(hp->s_h2d);
  const hipError_t e2 = hipStreamSynchronize(hp->s_cmp);
  const hipError_t e3 = hipStreamSynchronize(hp->s_d2h);
  if (!ok)
    return fail(what, e);
  if (drain_error != hipSuccess)
    return fail("hipEventSynchronize(d2h)", drain_error);
  if (abort)
    return fail("aborted", hipErrorUnknown);
  if (e1 != hipSuccess)
    return fail("hipStreamSynchronize(h2d)", e1);
  if (e2 != hipSuccess)
    return fail("hipStreamSynchronize(compute)", e2);
  if (e3 != hipSuccess)
    return fail("hipStreamSynchronize(d2h)", e3);
  return true;
}

} // namespace mifc
