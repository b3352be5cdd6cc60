// call_latency_probe.cc -- wall time of the synchronous C-ABI calls on ONE
// 1440x720 level that is already resident in HBM, from a C++ caller (no Python
// in the path): what a per-level caller of the reference API pays per call.
//
//   hipcc -O2 -I include -o tools/call_latency_probe tools/call_latency_probe.cc -L mi-fieldcalc_amd -lmifc -Wl,-rpath,'$ORIGIN/../mi-fieldcalc_amd'
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <string>
#include <vector>

#include "mifc.h"

#define CK(x)                                                      \
  do {                                                             \
    hipError_t e_ = (x);                                           \
    if (e_ != hipSuccess) {                                        \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
      return 1;                                                    \
    }                                                              \
  } while (0)

static double now_us()
{
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class F>
static double per_call_us(F f, int n = 2000)
{
  for (int i = 0; i < 50; ++i)
    f();
  const double t0 = now_us();
  for (int i = 0; i < n; ++i)
    f();
  return (now_us() - t0) / n;
}

int main()
{
  const int nx = 1440, ny = 720, nmem = 51;
  const size_t n = (size_t)nx * ny;
  mifc_ctx* c = mifc_create(0);
  if (!c) {
    std::fprintf(stderr, "no device\n");
    return 1;
  }
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i)
    h[i] = 280.0f + 0.001f * (float)(i % 977);
  float *u, *v, *xm, *ym, *out;
  CK(hipMalloc(&u, n * 4));
  CK(hipMalloc(&v, n * 4));
  CK(hipMalloc(&xm, n * 4));
  CK(hipMalloc(&ym, n * 4));
  CK(hipMalloc(&out, n * 4));
  for (float* p : {u, v, xm, ym})
    CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
  std::vector<const float*> members(nmem);
  std::vector<int> mflags(nmem, MIFC_ALL_DEFINED);
  for (int k = 0; k < nmem; ++k) {
    float* m;
    CK(hipMalloc(&m, n * 4));
    CK(hipMemcpy(m, h.data(), n * 4, hipMemcpyHostToDevice));
    members[k] = m;
  }
  const float undef = 1.0e35f;
  int ok = 1;
  std::printf("one %dx%d level, device-resident, synchronous C-ABI calls from C++ (us per call)\n", nx, ny);
  for (int flag_in : {MIFC_ALL_DEFINED, MIFC_SOME_DEFINED}) {
    const char* fn = flag_in == MIFC_ALL_DEFINED ? "ALL_DEFINED" : "SOME_DEFINED";
    std::printf("%-44s %8.1f\n", (std::string("relvort, ") + fn).c_str(), per_call_us([&] {
                  int f = flag_in;
                  ok &= mifc_relvort(c, nx, ny, u, v, xm, ym, out, &f, undef, MIFC_MEM_DEVICE);
                }));
    std::printf("%-44s %8.1f\n", (std::string("vectorabs, ") + fn).c_str(), per_call_us([&] {
                  int f = flag_in;
                  ok &= mifc_vectorabs(c, nx, ny, u, v, out, &f, undef, MIFC_MEM_DEVICE);
                }));
    std::printf("%-44s %8.1f\n", (std::string("thermalFrontParameter, ") + fn).c_str(), per_call_us([&] {
                  int f = flag_in;
                  ok &= mifc_thermalFrontParameter(c, nx, ny, u, xm, ym, out, &f, undef, MIFC_MEM_DEVICE);
                }));
  }
  std::printf("%-44s %8.1f\n", "meanValue, 51 members", per_call_us([&] {
                int f = MIFC_SOME_DEFINED;
                ok &= mifc_meanValue(c, nx, ny, members.data(), mflags.data(), nmem, out, &f, undef, MIFC_MEM_DEVICE);
              }, 500));
  // floor: one empty-ish launch + stream synchronize through the same runtime
  hipStream_t s;
  CK(hipStreamCreate(&s));
  std::printf("%-44s %8.1f\n", "hipMemsetAsync(8 B) + hipStreamSynchronize", per_call_us([&] {
                (void)hipMemsetAsync(out, 0, 8, s);
                (void)hipStreamSynchronize(s);
              }));
  std::printf("ok=%d\n", ok);
  mifc_destroy(c);
  return ok ? 0 : 1;
}
