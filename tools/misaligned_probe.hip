// misaligned_probe.hip -- measurement only (tools/): what do 16-byte-per-lane loads and stores cost when the address is
// only dword-aligned (a field whose width is not a multiple of 4: every row starts 4, 8 or 12 bytes off)?
// Plain 2-in / 2-out float4 stream over 568-MB arrays, all pointers shifted by 0..3 floats; and loads alone / stores alone.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
      std::exit(1);                                                               \
    }                                                                             \
  } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) U4
{
  v4f v;
};

template <int OP> // 0 copy 2->2, 1 loads only, 2 stores only
__global__ __launch_bounds__(256) void k(const float* a, const float* b, float* c, float* d, unsigned n4, float* sink)
{
  const unsigned q = blockIdx.x * 256u + threadIdx.x;
  if (q >= n4)
    return;
  v4f x = {1.f, 2.f, 3.f, 4.f}, y = x;
  if (OP != 2) {
    x = reinterpret_cast<const U4*>(a + 4 * (size_t)q)->v;
    y = reinterpret_cast<const U4*>(b + 4 * (size_t)q)->v;
  }
  if (OP == 1) {
    if (x.x + y.y == 123456.789f)
      sink[0] = x.x;
    return;
  }
  U4 s, t;
  s.v = x + y;
  t.v = x - y;
  *reinterpret_cast<U4*>(c + 4 * (size_t)q) = s;
  *reinterpret_cast<U4*>(d + 4 * (size_t)q) = t;
}

int main()
{
  const size_t N = (size_t)1440 * 720 * 137;
  float *a, *b, *c, *d, *sink;
  CHECK(hipMalloc(&a, (N + 64) * 4));
  CHECK(hipMalloc(&b, (N + 64) * 4));
  CHECK(hipMalloc(&c, (N + 64) * 4));
  CHECK(hipMalloc(&d, (N + 64) * 4));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(a, 0, (N + 64) * 4));
  CHECK(hipMemset(b, 0, (N + 64) * 4));
  const unsigned n4 = (unsigned)(N / 4);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const char* names[3] = {"2 in / 2 out", "loads only", "stores only"};
  for (int op = 0; op < 3; ++op) {
    for (int sh = 0; sh < 4; ++sh) {
      std::vector<float> ms;
      for (int rep = 0; rep < 7; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        for (int it = 0; it < 3; ++it) {
          if (op == 0)
            hipLaunchKernelGGL(k<0>, dim3((n4 + 255) / 256), dim3(256), 0, 0, a + sh, b + sh, c + sh, d + sh, n4, sink);
          else if (op == 1)
            hipLaunchKernelGGL(k<1>, dim3((n4 + 255) / 256), dim3(256), 0, 0, a + sh, b + sh, c + sh, d + sh, n4, sink);
          else
            hipLaunchKernelGGL(k<2>, dim3((n4 + 255) / 256), dim3(256), 0, 0, a + sh, b + sh, c + sh, d + sh, n4, sink);
        }
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float t;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        if (rep)
          ms.push_back(t / 3);
      }
      std::sort(ms.begin(), ms.end());
      const double bytes = (double)N * 4 * (op == 0 ? 4 : 2);
      std::printf("%-14s pointers + %d floats: %.4f ms  %.0f GB/s\n", names[op], sh, ms[ms.size() / 2], bytes / (ms[ms.size() / 2] * 1e-3) / 1e9);
    }
  }
  return 0;
}
