// hipExtMallocWithFlags(hipDeviceMallocContiguous) against plain hipMalloc for the four arrays of the headline batch:
// is physical contiguity what separates a fast placement from a slow one?  (profiles/r03/experiments/vmm_placement_*.txt:
// memory the program maps itself in large chunks runs the kernel at the rate of the best placement a search finds.)
// Every variant is allocated `repeats` times from scratch, with ballast allocations in between.
//   hipcc -O2 -o tools/contiguous_alloc_probe tools/contiguous_alloc_probe.cc -Iinclude -Lmi-fieldcalc_amd -lmifc -Wl,-rpath,'$ORIGIN/../mi-fieldcalc_amd'
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mifc.h"

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

static const int NX = 1440, NY = 720, NLEV = 137;

__global__ void fill(float* p, size_t n, float a)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = a + 1e-3f * (float)(i % 9973);
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

int main(int argc, char** argv)
{
  const int repeats = argc > 1 ? atoi(argv[1]) : 4;
  CK(hipSetDevice(0));
  mifc_ctx* ctx = mifc_create(0);
  if (!ctx)
    return 1;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  mifc_set_stream(ctx, s);
  const size_t n = (size_t)NX * NY, nb = n * NLEV, bytes = nb * sizeof(float);
  float *xm, *ym;
  CK(hipMalloc(&xm, n * 4));
  CK(hipMalloc(&ym, n * 4));
  hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, s, xm, n, 1e-5f);
  hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, s, ym, n, 2e-5f);
  auto run_on = [&](float* a0, float* a1, float* a2, float* a3) {
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, s, a0, nb, 10.f);
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, s, a1, nb, -5.f);
    return time_kernel(ctx, a0, a1, xm, ym, a2, a3, s);
  };
  std::vector<void*> ballast;
  const size_t MIB = 1 << 20;
  for (int r = 0; r < repeats; ++r) {
    float* a[4];
    for (auto& q : a)
      CK(hipMalloc(&q, bytes));
    const float t0 = run_on(a[0], a[1], a[2], a[3]);
    for (auto& q : a)
      CK(hipFree(q));
    float* c[4];
    bool ok = true;
    for (auto& q : c) {
      q = nullptr;
      if (hipExtMallocWithFlags((void**)&q, bytes, hipDeviceMallocContiguous) != hipSuccess) {
        ok = false;
        (void)hipGetLastError();
      }
    }
    float t1 = -1.f;
    if (ok)
      t1 = run_on(c[0], c[1], c[2], c[3]);
    for (auto& q : c)
      if (q)
        CK(hipFree(q));
    // one contiguous slab, arrays 544 / 608 / 640 MiB apart
    float ts[3] = {-1.f, -1.f, -1.f};
    const size_t strides[3] = {544 * MIB, 608 * MIB, 640 * MIB};
    char* slab = nullptr;
    if (hipExtMallocWithFlags((void**)&slab, 4 * strides[2], hipDeviceMallocContiguous) == hipSuccess) {
      for (int k = 0; k < 3; ++k)
        ts[k] = run_on((float*)slab, (float*)(slab + strides[k]), (float*)(slab + 2 * strides[k]), (float*)(slab + 3 * strides[k]));
      CK(hipFree(slab));
    } else {
      (void)hipGetLastError();
    }
    auto pct = [](float ms) { return ms > 0 ? 2280960000.0 / ms / 1e6 / 8000 * 100 : 0.0; };
    printf("repeat %d: hipMalloc x4 %.4f ms (%.1f %%) | contiguous x4 %.4f ms (%.1f %%) | contiguous slab, arrays 544 / 608 / 640 MiB apart: %.4f (%.1f %%) %.4f (%.1f %%) %.4f (%.1f %%)\n",
           r, t0, pct(t0), t1, pct(t1), ts[0], pct(ts[0]), ts[1], pct(ts[1]), ts[2], pct(ts[2]));
    fflush(stdout);
    void* b = nullptr;
    CK(hipMalloc(&b, (300 + 211 * (size_t)r) * MIB)); // the next round lands somewhere else
    ballast.push_back(b);
  }
  for (void* b : ballast)
    CK(hipFree(b));
  mifc_destroy(ctx);
  return 0;
}
