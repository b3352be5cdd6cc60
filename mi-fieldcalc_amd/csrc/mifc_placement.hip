// mifc_placement.hip -- where in HBM a long-lived batch lives, for callers of the C ABI (host code only).
//
// Which physical memory the arrays of a batch lie on changes the time of a streaming kernel over them by up to 12 %,
// stably for as long as the arrays live.  Round 2 chose a fast set of arrays from a pool in Python
// (mi-fieldcalc_amd/placement.py); this is the same search behind the C ABI, with a stated memory budget, plus -- opt-in --
// a second strategy that came out of round 3's experiment with HIP's virtual-memory API
// (profiles/r03/experiments/vmm_placement_*.txt): ONE physical allocation made with hipMemCreate and mapped by the
// library, the arrays a chosen distance apart inside it, ran the headline kernel at 74.3-74.5 % of 8 TB/s on every box
// tried without any search, where arrays from hipMalloc (also one slab of them, also hipDeviceMallocContiguous) gave
// 65-72 %.  Two probe runs that mapped and unmapped such allocations a dozen times in a row ended in a GPU memory fault,
// cause not found, so that strategy maps ONCE per batch, never remaps, and has to be asked for.
#include "mifc_ctx.h"

#include <algorithm>
#include <cstring>
#include <random>
#include <vector>

using namespace mifc_host;

namespace {

struct VmmBatch
{
  void* va = nullptr;
  size_t bytes = 0;
  hipMemGenericAllocationHandle_t handle;
};
// batches the library mapped itself (MIFC_PLACE_VMM), by base address, for mifc_batch_free_placed
std::vector<VmmBatch>& vmm_batches()
{
  static std::vector<VmmBatch> v;
  return v;
}

struct Probe
{
  mifc_ctx* c;
  mifc_placement_probe_fn fn;
  void* user;
  int nx, ny, nlev;
  const float *xm, *ym;
  float run(void* const* a) const
  {
    if (fn)
      return fn(user, a);
    // the library's own probe: the fused vorticity + divergence launch over the first four arrays (u, v, rvort, diverg)
    std::vector<int> flags((size_t)nlev, MIFC_ALL_DEFINED);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
      return -1.f;
    float best = -1.f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0, c->stream);
      for (int k = 0; k < 4; ++k)
        if (!mifc_vortdiv_levels_enqueue(c, nx, ny, nlev, (const float*)a[0], (const float*)a[1], xm, ym, (float*)a[2], (float*)a[3], flags.data(), 1e35f,
                                         nullptr))
          return -1.f;
      (void)hipEventRecord(e1, c->stream);
      float ms = -1.f;
      if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
        return -1.f;
      if (rep > 0 && (best < 0.f || ms / 4 < best)) // the first burst warms up
        best = ms / 4;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return best;
  }
};

} // namespace

extern "C" {

int mifc_batch_alloc_placed(mifc_ctx* c, int n_arrays, size_t bytes_each, int strategy, size_t budget_bytes, int nx, int ny, int nlev,
                            mifc_placement_probe_fn probe, void* probe_user, void** arrays_out, mifc_placement_report* report)
{
  if (!c || !arrays_out || n_arrays < 1 || n_arrays > 16 || bytes_each == 0)
    return 0;
  enter(c);
  mifc_placement_report rep;
  std::memset(&rep, 0, sizeof rep);
  rep.strategy = strategy;
  const bool own_probe = !probe;
  if (own_probe && strategy == MIFC_PLACE_SEARCH && !(n_arrays >= 4 && nx >= 3 && ny >= 3 && nlev >= 1 && (size_t)nx * ny * nlev * sizeof(float) <= bytes_each)) {
    c->err = "mifc_batch_alloc_placed: without a probe callback the search times the fused vorticity+divergence launch: give nx, ny, nlev of a batch "
             "that fits the arrays, and at least four arrays";
    return 0;
  }
  if (strategy == MIFC_PLACE_VMM) {
    // ONE physical allocation, mapped once; the arrays `stride` apart: the array size rounded up to 16 MiB, then moved to the
    // next distance that is 96 .. 128 MiB past a multiple of 256 MiB -- the fast end of the measured curve
    // (profiles/r03/experiments/vmm_placement_strides.txt: 544 MiB apart 71.5 %, 608 74.5 %, 640 74.4 %, 704 73.4 %, 768 72.1 %)
    const size_t MIB = (size_t)1 << 20;
    size_t stride = ((bytes_each + 16 * MIB - 1) / (16 * MIB)) * (16 * MIB);
    while ((stride / MIB) % 256 < 96 || (stride / MIB) % 256 > 128)
      stride += 16 * MIB;
    const size_t total = stride * (size_t)n_arrays;
    if (budget_bytes && total > budget_bytes) {
      c->err = "mifc_batch_alloc_placed: the budget does not hold the batch at the chosen array distance";
      return 0;
    }
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = c->device;
    VmmBatch b;
    b.bytes = total;
    MIFC_HIP(c, hipMemAddressReserve(&b.va, total, 0, nullptr, 0));
    if (hipMemCreate(&b.handle, total, &prop, 0) != hipSuccess) {
      (void)hipMemAddressFree(b.va, total);
      c->err = "mifc_batch_alloc_placed: hipMemCreate failed";
      return 0;
    }
    hipMemAccessDesc acc;
    std::memset(&acc, 0, sizeof acc);
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = c->device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemMap(b.va, total, 0, b.handle, 0) != hipSuccess || hipMemSetAccess(b.va, total, &acc, 1) != hipSuccess) {
      (void)hipMemRelease(b.handle);
      (void)hipMemAddressFree(b.va, total);
      c->err = "mifc_batch_alloc_placed: mapping the allocation failed";
      return 0;
    }
    vmm_batches().push_back(b);
    for (int k = 0; k < n_arrays; ++k)
      arrays_out[k] = static_cast<char*>(b.va) + (size_t)k * stride;
    rep.pool_size = 1;
    rep.array_distance_bytes = stride;
    if (!own_probe || (n_arrays >= 4 && nx >= 3 && (size_t)nx * ny * nlev * sizeof(float) <= bytes_each)) {
      // timed for the report only
      float *xm = nullptr, *ym = nullptr;
      if (own_probe) {
        MIFC_HIP(c, hipMalloc((void**)&xm, (size_t)nx * ny * 4));
        MIFC_HIP(c, hipMalloc((void**)&ym, (size_t)nx * ny * 4));
        MIFC_HIP(c, hipMemsetAsync(xm, 0, (size_t)nx * ny * 4, c->stream));
        MIFC_HIP(c, hipMemsetAsync(ym, 0, (size_t)nx * ny * 4, c->stream));
      }
      const Probe p = {c, probe, probe_user, nx, ny, nlev, xm, ym};
      rep.chosen_ms = p.run(arrays_out);
      rep.probes = 1;
      if (xm)
        (void)hipFree(xm);
      if (ym)
        (void)hipFree(ym);
    }
    if (report)
      *report = rep;
    return 1;
  }
  if (strategy != MIFC_PLACE_SEARCH && strategy != MIFC_PLACE_AS_ALLOCATED)
    return 0;

  // ---- the search of mi-fieldcalc_amd/placement.py::choose_search: a pool of arrays allocated in one go; the first n
  // ("as allocated", for the record), the sets that lie pool / n allocations apart and random index sets are probed, then
  // coordinate descent from the best; everything but the chosen arrays is freed.
  int pool_size = n_arrays;
  if (strategy == MIFC_PLACE_SEARCH) {
    pool_size = 48;
    if (budget_bytes)
      pool_size = (int)std::min<size_t>(48, budget_bytes / bytes_each);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
      pool_size = (int)std::min<size_t>((size_t)pool_size, (free_b / 10 * 9) / bytes_each); // never more than 90 % of what is free
    if (pool_size < n_arrays) {
      c->err = "mifc_batch_alloc_placed: the budget (or the free device memory) does not hold the batch";
      return 0;
    }
  }
  std::vector<void*> pool((size_t)pool_size, nullptr);
  for (int i = 0; i < pool_size; ++i)
    if (hipMalloc(&pool[i], bytes_each) != hipSuccess) {
      (void)hipGetLastError();
      if (i < n_arrays) {
        for (void* q : pool)
          if (q)
            (void)hipFree(q);
        c->err = "mifc_batch_alloc_placed: out of device memory";
        return 0;
      }
      pool.resize((size_t)i); // a smaller pool will do
      pool_size = i;
      break;
    }
  rep.pool_size = pool_size;
  std::vector<int> best(n_arrays);
  for (int k = 0; k < n_arrays; ++k)
    best[k] = k;
  if (strategy == MIFC_PLACE_SEARCH && pool_size > n_arrays) {
    float *xm = nullptr, *ym = nullptr;
    if (own_probe) {
      const size_t nb = (size_t)nx * ny * 4;
      if (hipMalloc((void**)&xm, nb) != hipSuccess || hipMalloc((void**)&ym, nb) != hipSuccess || hipMemsetAsync(xm, 0, nb, c->stream) != hipSuccess ||
          hipMemsetAsync(ym, 0, nb, c->stream) != hipSuccess) {
        c->err = "mifc_batch_alloc_placed: cannot allocate the probe's map factors";
        for (void* q : pool)
          (void)hipFree(q);
        return 0;
      }
      // defined values in the probe's inputs: the time does not depend on them, but NaN patterns need not be streamed either
      for (int i = 0; i < pool_size; ++i)
        (void)hipMemsetAsync(pool[i], 0, bytes_each, c->stream);
    }
    const Probe p = {c, probe, probe_user, nx, ny, nlev, xm, ym};
    std::vector<std::pair<std::vector<int>, float>> seen;
    const int max_probes = 160;
    auto timed = [&](const std::vector<int>& idx) -> float {
      for (auto& s : seen)
        if (s.first == idx)
          return s.second;
      std::vector<void*> a((size_t)n_arrays);
      for (int k = 0; k < n_arrays; ++k)
        a[k] = pool[idx[k]];
      const float ms = p.run(a.data());
      seen.push_back({idx, ms});
      return ms;
    };
    float best_ms = timed(best);
    rep.as_allocated_ms = best_ms;
    const int step = pool_size / n_arrays;
    for (int b = 0; b < step && best_ms >= 0.f; ++b) {
      std::vector<int> idx(n_arrays);
      for (int k = 0; k < n_arrays; ++k)
        idx[k] = b + k * step;
      const float ms = timed(idx);
      if (ms >= 0.f && ms < best_ms) {
        best_ms = ms;
        best = idx;
      }
    }
    std::mt19937 rng(5);
    for (int r = 0; r < 24 && (int)seen.size() < max_probes && best_ms >= 0.f; ++r) {
      std::vector<int> all(pool_size);
      for (int i = 0; i < pool_size; ++i)
        all[i] = i;
      std::shuffle(all.begin(), all.end(), rng);
      std::vector<int> idx(all.begin(), all.begin() + n_arrays);
      const float ms = timed(idx);
      if (ms >= 0.f && ms < best_ms) {
        best_ms = ms;
        best = idx;
      }
    }
    bool improved = best_ms >= 0.f;
    while (improved && (int)seen.size() < max_probes) {
      improved = false;
      for (int pos = n_arrays - 1; pos >= 0; --pos) { // outputs first: the stores are the slower side
        const std::vector<int> cur = best;
        for (int i = 0; i < pool_size && (int)seen.size() < max_probes; ++i) {
          if (std::find(cur.begin(), cur.end(), i) != cur.end())
            continue;
          std::vector<int> cand = cur;
          cand[pos] = i;
          const float ms = timed(cand);
          if (ms >= 0.f && ms < best_ms) {
            best_ms = ms;
            best = cand;
            improved = true;
          }
        }
      }
    }
    if (xm)
      (void)hipFree(xm);
    if (ym)
      (void)hipFree(ym);
    if (best_ms < 0.f) {
      for (void* q : pool)
        (void)hipFree(q);
      c->err = "mifc_batch_alloc_placed: the probe failed" + (c->err.empty() ? std::string() : (": " + c->err));
      return 0;
    }
    rep.probes = (int)seen.size();
    rep.chosen_ms = best_ms;
    std::vector<float> t;
    for (auto& s : seen)
      if (s.second >= 0.f)
        t.push_back(s.second);
    std::sort(t.begin(), t.end());
    rep.probe_ms_min = t.front();
    rep.probe_ms_median = t[t.size() / 2];
    rep.probe_ms_max = t.back();
  }
  MIFC_HIP(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < n_arrays; ++k) {
    arrays_out[k] = pool[best[k]];
    pool[best[k]] = nullptr;
  }
  for (void* q : pool)
    if (q)
      (void)hipFree(q);
  if (report)
    *report = rep;
  return 1;
}

int mifc_batch_free_placed(mifc_ctx* c, void** arrays, int n_arrays)
{
  if (!c || !arrays || n_arrays < 1)
    return 0;
  enter(c);
  MIFC_HIP(c, hipStreamSynchronize(c->stream));
  std::vector<VmmBatch>& vb = vmm_batches();
  for (size_t i = 0; i < vb.size(); ++i)
    if (vb[i].va == arrays[0]) { // a batch the library mapped itself: one unmap for all its arrays
      const VmmBatch b = vb[i];
      vb.erase(vb.begin() + (long)i);
      MIFC_HIP(c, hipMemUnmap(b.va, b.bytes));
      MIFC_HIP(c, hipMemRelease(b.handle));
      MIFC_HIP(c, hipMemAddressFree(b.va, b.bytes));
      for (int k = 0; k < n_arrays; ++k)
        arrays[k] = nullptr;
      return 1;
    }
  for (int k = 0; k < n_arrays; ++k)
    if (arrays[k]) {
      MIFC_HIP(c, hipFree(arrays[k]));
      arrays[k] = nullptr;
    }
  return 1;
}

} // extern "C"
