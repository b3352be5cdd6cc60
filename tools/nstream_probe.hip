// nstream_probe.hip -- measurement only (tools/): the rate of a plain float4 stream kernel as a function of HOW MANY
// arrays it reads and writes at once (same total bytes per array: the headline batch's 568 MB), one-shot
// (a workgroup = one 4-KiB page of every array) and as a grid-stride loop over 8 workgroups per CU -- the
// yardstick each operator's "fraction of 8 TB/s" has to be read against: xcd_run_probe measured 7.0 TB/s for one
// read stream, 6.6 for one write stream, 6.3 for a 1 -> 1 copy and 5.7 for 2 -> 2.
//
// Usage: nstream_probe [spacer MiB between the arrays, default 0]
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

struct Ptrs
{
  const v4f* in[5];
  v4f* out[5];
};

template <int NI, int NO, bool LOOP>
__global__ __launch_bounds__(256) void stream_kernel(const Ptrs P, unsigned n4)
{
  unsigned q = blockIdx.x * 256u + threadIdx.x;
  const unsigned stride = LOOP ? gridDim.x * 256u : 0xffffffffu;
  for (; q < n4; q += stride) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    v4f x[NI > 0 ? NI : 1];
#pragma unroll
    for (int i = 0; i < NI; ++i)
      x[i] = P.in[i][q];
#pragma unroll
    for (int i = 0; i < NI; ++i)
      acc += x[i];
    if (NI == 0)
      acc.x = (float)q;
#pragma unroll
    for (int o = 0; o < NO; ++o)
      __builtin_nontemporal_store(acc + (float)o, P.out[o] + q);
    if (NO == 0 && acc.x == 123456.789f)
      P.out[0][0] = acc;
    if (!LOOP)
      break;
  }
}

template <int NI, int NO>
static void measure(const Ptrs& P, unsigned n4, hipEvent_t e0, hipEvent_t e1)
{
  double rate[2];
  for (int loop = 0; loop < 2; ++loop) {
    const unsigned grid = loop ? 256 * 8 : (n4 + 255) / 256;
    std::vector<float> ms;
    for (int rep = 0; rep < 7; ++rep) {
      CHECK(hipEventRecord(e0, 0));
      for (int k = 0; k < 3; ++k) {
        if (loop)
          hipLaunchKernelGGL((stream_kernel<NI, NO, true>), dim3(grid), dim3(256), 0, 0, P, n4);
        else
          hipLaunchKernelGGL((stream_kernel<NI, NO, false>), dim3(grid), dim3(256), 0, 0, P, n4);
      }
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float t;
      CHECK(hipEventElapsedTime(&t, e0, e1));
      if (rep)
        ms.push_back(t / 3);
    }
    std::sort(ms.begin(), ms.end());
    rate[loop] = (double)n4 * 16.0 * (NI + NO) / (ms[ms.size() / 2] * 1e-3) / 1e9;
  }
  std::printf("  %d in, %d out: one-shot %5.0f GB/s (%.1f %% of 8 TB/s)   grid-stride loop %5.0f GB/s (%.1f %%)\n", NI, NO, rate[0], rate[0] / 80.0, rate[1],
              rate[1] / 80.0);
  std::fflush(stdout);
}

int main(int argc, char** argv)
{
  const size_t spacer = (argc > 1 ? (size_t)std::atoi(argv[1]) : 0) << 20;
  const size_t N = (size_t)1440 * 720 * 137 * 4;
  const unsigned n4 = (unsigned)(N / 16);
  Ptrs P;
  std::vector<void*> keep;
  for (int i = 0; i < 10; ++i) {
    if (spacer) {
      void* s;
      CHECK(hipMalloc(&s, spacer));
      keep.push_back(s);
    }
    v4f* a;
    CHECK(hipMalloc(&a, N));
    CHECK(hipMemset(a, 0, N));
    // inputs and outputs alternate in allocation order
    if (i & 1)
      P.out[i / 2] = a;
    else
      P.in[i / 2] = a;
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  std::printf("plain float4 streams over arrays of %zu bytes, spacer between allocations %zu MiB\n", N, spacer >> 20);
  for (int k = 0; k < 10; ++k)
    hipLaunchKernelGGL((stream_kernel<2, 2, false>), dim3((n4 + 255) / 256), dim3(256), 0, 0, P, n4);
  measure<1, 0>(P, n4, e0, e1);
  measure<0, 1>(P, n4, e0, e1);
  measure<1, 1>(P, n4, e0, e1);
  measure<2, 1>(P, n4, e0, e1);
  measure<2, 2>(P, n4, e0, e1);
  measure<3, 1>(P, n4, e0, e1);
  measure<4, 1>(P, n4, e0, e1);
  measure<2, 4>(P, n4, e0, e1);
  measure<4, 3>(P, n4, e0, e1);
  measure<4, 4>(P, n4, e0, e1);
  measure<5, 5>(P, n4, e0, e1);
  return 0;
}
