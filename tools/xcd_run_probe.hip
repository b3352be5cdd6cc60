// xcd_run_probe.hip -- measurement only (tools/): does a stream run faster when every XCD works on RUNS of
// consecutive 4-KiB pages instead of every eighth page?
//
// Workgroups are dispatched round-robin over the 8 XCDs (workgroup b -> XCD b % 8, checked by xcd_stack_probe), so
// a one-shot elementwise kernel whose workgroup b takes page b gives XCD x the pages = x (mod 8): every XCD's L2
// holds an eighth of every 32-KiB stretch.  xcd_half_probe's write rate rose from 4.6 to 5.6 TB/s when the pages
// of an XCD were grouped.  Here: ONE-SHOT kernels (a workgroup = one page of each array), page = perm_R(b):
//   s = b / 8 (sequence number within the XCD), x = b % 8;  page = (s / R) * 8 R + x * R + s % R
// R = 1 is the identity.  The dispatch order, the number of workgroups and the bytes are the same for every R.
//
// Usage: xcd_run_probe [arrays in the pool, default 6]
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

__device__ __forceinline__ unsigned perm(unsigned b, unsigned R, unsigned pages)
{
  const unsigned s = b >> 3, x = b & 7;
  const unsigned p = (s / R) * 8 * R + x * R + s % R;
  return p < pages ? p : b; // the ragged tail keeps the identity
}

// op 0: read a; 1: write c; 2: copy a -> c; 3: 2-in / 2-out (a, b -> c, d)
template <int OP>
__global__ __launch_bounds__(256) void run_kernel(const v4f* __restrict__ a, const v4f* __restrict__ b2, v4f* __restrict__ c, v4f* __restrict__ d, unsigned pages,
                                                  unsigned R, unsigned full, float* sink)
{
  const unsigned b = blockIdx.x;
  const unsigned p = b < full ? perm(b, R, full) : b;
  const size_t o = (size_t)p * 256 + threadIdx.x;
  if (OP == 0) {
    const v4f t = a[o];
    if (t.x + t.y + t.z + t.w == 123456.789f)
      sink[0] = t.x;
  } else if (OP == 1) {
    const float f = (float)b;
    __builtin_nontemporal_store(v4f{f, f, f, f}, c + o);
  } else if (OP == 2) {
    __builtin_nontemporal_store(a[o], c + o);
  } else {
    const v4f x = a[o], y = b2[o];
    __builtin_nontemporal_store(x + y, c + o);
    __builtin_nontemporal_store(x - y, d + o);
  }
}

int main(int argc, char** argv)
{
  const int m = argc > 1 ? std::atoi(argv[1]) : 6;
  const size_t N = (size_t)1440 * 720 * 137 * 4;
  const unsigned pages = (unsigned)(N / 4096);
  std::vector<v4f*> arr(m);
  for (int i = 0; i < m; ++i) {
    CHECK(hipMalloc(&arr[i], N));
    CHECK(hipMemset(arr[i], 0, N));
  }
  float* sink;
  CHECK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto run = [&](int op, int i0, unsigned R) {
    const unsigned full = pages / (8 * R) * (8 * R);
    std::vector<float> ms;
    for (int rep = 0; rep < 7; ++rep) {
      const v4f *a = arr[i0 % m], *b = arr[(i0 + 1) % m];
      v4f *c = arr[(i0 + 2) % m], *d = arr[(i0 + 3) % m];
      CHECK(hipEventRecord(e0, 0));
      for (int k = 0; k < 3; ++k) {
        if (op == 0)
          hipLaunchKernelGGL(run_kernel<0>, dim3(pages), dim3(256), 0, 0, a, b, c, d, pages, R, full, sink);
        else if (op == 1)
          hipLaunchKernelGGL(run_kernel<1>, dim3(pages), dim3(256), 0, 0, a, b, c, d, pages, R, full, sink);
        else if (op == 2)
          hipLaunchKernelGGL(run_kernel<2>, dim3(pages), dim3(256), 0, 0, a, b, c, d, pages, R, full, sink);
        else
          hipLaunchKernelGGL(run_kernel<3>, dim3(pages), dim3(256), 0, 0, a, b, c, d, pages, R, full, sink);
      }
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float t;
      CHECK(hipEventElapsedTime(&t, e0, e1));
      if (rep)
        ms.push_back(t / 3);
    }
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)pages * 4096.0 * (op == 3 ? 4 : op == 2 ? 2 : 1);
    return bytes / (ms[ms.size() / 2] * 1e-3) / 1e9;
  };
  for (int k = 0; k < 30; ++k)
    run(3, 0, 1);
  const unsigned Rs[] = {1, 2, 4, 8, 16, 32, 64, 128, 512, 2048, 16384};
  const char* opname[4] = {"read", "write", "copy 1->1", "stream 2->2"};
  std::printf("GB/s, one-shot kernels, a workgroup = one 4-KiB page per array; R = pages per XCD run\n%-22s", "op / arrays");
  for (unsigned R : Rs)
    std::printf(" %6u", R);
  std::printf("\n");
  for (int op = 0; op < 4; ++op) {
    for (int i0 = 0; i0 < (m >= 6 ? 3 : 1); ++i0) {
      char label[64];
      std::snprintf(label, sizeof label, "%s @%d", opname[op], i0);
      std::printf("%-22s", label);
      for (unsigned R : Rs)
        std::printf(" %6.0f", run(op, i0, R));
      std::printf("\n");
      std::fflush(stdout);
    }
  }
  return 0;
}
