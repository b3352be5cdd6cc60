// Does the CHOICE of physical memory decide the time of the headline kernel, and can a library choose it?
// (VERDICT r2 "next" 4b.)  Builds the four arrays of a 1440x720x137 batch from physical chunks the program creates itself
// with HIP's virtual-memory API (hipMemCreate / hipMemAddressReserve / hipMemMap) in several creation orders and chunk
// sizes, and times the fused vorticity+divergence launch (through the C ABI, libmifc.so) on each layout, next to plain
// hipMalloc arrays.  Every layout is built `repeats` times from scratch: is a layout's time a property of the layout?
//   hipcc -O2 -o tools/vmm_placement_probe tools/vmm_placement_probe.cc -Iinclude -Lmi-fieldcalc_amd -lmifc -Wl,-rpath,'$ORIGIN/../mi-fieldcalc_amd'
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "mifc.h"

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);       \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

static const int NX = 1440, NY = 720;
static int NLEV = 137; // VMM_NLEV overrides

struct Mapped
{
  void* va = nullptr;
  size_t bytes = 0;
  std::vector<hipMemGenericAllocationHandle_t> handles;
};

static hipMemAllocationProp prop_for(int dev)
{
  hipMemAllocationProp p;
  memset(&p, 0, sizeof p);
  p.type = hipMemAllocationTypePinned;
  p.location.type = hipMemLocationTypeDevice;
  p.location.id = dev;
  return p;
}

static void unmap(Mapped& m)
{
  if (m.va) {
    CK(hipMemUnmap(m.va, m.bytes));
    CK(hipMemAddressFree(m.va, m.bytes));
  }
  for (auto h : m.handles)
    CK(hipMemRelease(h));
  m = Mapped();
}

// creates chunks for the four arrays in the order given by `order` (a list of (array, chunk) pairs), maps chunk k of
// array a at offset k*chunk of array a's range
static void build(std::vector<Mapped>& arr, size_t bytes, size_t chunk, const std::vector<std::pair<int, int>>& order, int dev)
{
  const hipMemAllocationProp p = prop_for(dev);
  const size_t nchunks = (bytes + chunk - 1) / chunk;
  arr.assign(4, Mapped());
  std::vector<bool> used(4, false);
  for (auto& ac : order)
    used[ac.first] = true;
  for (int a = 0; a < 4; ++a) {
    if (!used[a])
      continue;
    Mapped& m = arr[a];
    m.bytes = nchunks * chunk;
    CK(hipMemAddressReserve(&m.va, m.bytes, 0, nullptr, 0));
  }
  for (auto& ac : order) {
    Mapped& m = arr[ac.first];
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, chunk, &p, 0));
    m.handles.push_back(h);
    CK(hipMemMap((char*)m.va + (size_t)ac.second * chunk, chunk, 0, h, 0));
  }
  hipMemAccessDesc acc;
  memset(&acc, 0, sizeof acc);
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = dev;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  for (auto& m : arr)
    if (m.va)
      CK(hipMemSetAccess(m.va, m.bytes, &acc, 1));
}

static float time_kernel(mifc_ctx* ctx, float* u, float* v, float* xm, float* ym, float* rv, float* dg, hipStream_t s)
{
  std::vector<int> flags(NLEV, MIFC_ALL_DEFINED);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int k = 0; k < 6; ++k)
    if (!mifc_vortdiv_levels_enqueue(ctx, NX, NY, NLEV, u, v, xm, ym, rv, dg, flags.data(), 1e35f, nullptr)) {
      printf("enqueue failed: %s\n", mifc_last_error(ctx));
      exit(1);
    }
  std::vector<float> ms;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0, s));
    for (int k = 0; k < 8; ++k)
      mifc_vortdiv_levels_enqueue(ctx, NX, NY, NLEV, u, v, xm, ym, rv, dg, flags.data(), 1e35f, nullptr);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    ms.push_back(t / 8);
  }
  std::sort(ms.begin(), ms.end());
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return ms[ms.size() / 2];
}

__global__ void fill(float* p, size_t n, float a)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = a + 1e-3f * (float)(i % 9973);
}

int main(int argc, char** argv)
{
  const int repeats = argc > 1 ? atoi(argv[1]) : 3;
  if (getenv("VMM_NLEV"))
    NLEV = atoi(getenv("VMM_NLEV"));
  const int dev = 0;
  CK(hipSetDevice(dev));
  mifc_ctx* ctx = mifc_create(dev);
  if (!ctx) {
    printf("no context\n");
    return 1;
  }
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  mifc_set_stream(ctx, s);
  const size_t n = (size_t)NX * NY, nb = n * NLEV, bytes = nb * sizeof(float);
  float *xm, *ym;
  CK(hipMalloc(&xm, n * 4));
  CK(hipMalloc(&ym, n * 4));
  hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, s, xm, n, 1e-5f);
  hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, s, ym, n, 2e-5f);
  const hipMemAllocationProp p = prop_for(dev);
  size_t gmin = 0, grec = 0;
  CK(hipMemGetAllocationGranularity(&gmin, &p, hipMemAllocationGranularityMinimum));
  CK(hipMemGetAllocationGranularity(&grec, &p, hipMemAllocationGranularityRecommended));
  printf("allocation granularity: minimum %zu B, recommended %zu B; one array = %zu B (%.1f MiB)\n", gmin, grec, bytes, bytes / 1048576.0);

  auto run_on = [&](float* a0, float* a1, float* a2, float* a3) {
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, s, a0, nb, 10.f);
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, s, a1, nb, -5.f);
    return time_kernel(ctx, a0, a1, xm, ym, a2, a3, s);
  };

  // (0) plain hipMalloc, four arrays one after the other
  for (int r = 0; r < repeats && !getenv("VMM_SKIP_MALLOC"); ++r) {
    float* a[4];
    for (auto& q : a)
      CK(hipMalloc(&q, bytes));
    printf("hipMalloc x4 in a row                                   repeat %d: %.4f ms\n", r, run_on(a[0], a[1], a[2], a[3]));
    for (auto& q : a)
      CK(hipFree(q));
  }
  // chunk sizes in MiB from the command line (default: a sweep); a size >= the array makes ONE chunk per array of that size,
  // i.e. the arrays' physical starts that many MiB apart when the chunks come out of fresh memory one after the other
  std::vector<size_t> chunks;
  for (int a = 2; a < argc; ++a)
    chunks.push_back((size_t)atol(argv[a]) << 20);
  if (chunks.empty())
    for (size_t c : {2, 4, 8, 16, 32, 64, 128, 256, 512, 542, 544, 576, 640, 768, 1024})
      chunks.push_back(c << 20);
  if (getenv("VMM_SKIP_CHUNKS"))
    chunks.clear();
  for (size_t chunk : chunks) {
    if (chunk % gmin != 0)
      continue;
    const int nch = (int)((bytes + chunk - 1) / chunk);
    std::vector<std::pair<int, int>> order;
    for (int a = 0; a < 4; ++a)
      for (int k = 0; k < nch; ++k)
        order.push_back({a, k});
    for (int r = 0; r < repeats; ++r) {
      std::vector<Mapped> arr;
      build(arr, bytes, chunk, order, dev);
      const float ms = run_on((float*)arr[0].va, (float*)arr[1].va, (float*)arr[2].va, (float*)arr[3].va);
      printf("chunk %4zu MiB x %3d per array (array = %4zu MiB of physical memory)  repeat %d: %.4f ms  = %.1f %% of 8 TB/s\n", chunk >> 20, nch,
             (nch * chunk) >> 20, r, ms, (16.0 * NX * NY * NLEV + 8.0 * NX * NY) / ms / 1e6 / 8000 * 100);
      fflush(stdout);
      CK(hipStreamSynchronize(s));
      for (auto& m : arr)
        unmap(m);
    }
  }
  // ONE physical allocation for the whole batch, the arrays `stride` MiB apart inside it (argv: "one" then strides)
  if (getenv("VMM_ONE_HANDLE")) {
    std::vector<size_t> strides;
    for (const char* q = getenv("VMM_ONE_HANDLE"); q && *q;) {
      strides.push_back((size_t)atol(q) << 20);
      q = strchr(q, ',');
      if (q)
        ++q;
    }
    for (size_t d : strides) {
      if (d < bytes || 4 * d > ((size_t)3072 << 20)) // (a 3.5 GiB handle faulted on first touch: stay at or below 3 GiB)
        continue;
      for (int r = 0; r < repeats; ++r) {
        std::vector<Mapped> one;
        build(one, 4 * d, 4 * d, {{0, 0}}, dev); // arrays 1..3 of `one` stay empty
        char* base = (char*)one[0].va;
        const float ms = run_on((float*)base, (float*)(base + d), (float*)(base + 2 * d), (float*)(base + 3 * d));
        printf("ONE handle of %5zu MiB, arrays %4zu MiB apart  repeat %d: %.4f ms  = %.1f %% of 8 TB/s\n", (4 * d) >> 20, d >> 20, r, ms,
               (16.0 * NX * NY * NLEV + 8.0 * NX * NY) / ms / 1e6 / 8000 * 100);
        fflush(stdout);
        CK(hipStreamSynchronize(s));
        unmap(one[0]);
      }
    }
  }
  mifc_destroy(ctx);
  return 0;
}
