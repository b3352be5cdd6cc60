// xcd_half_probe.hip -- measurement only (tools/): all 8 XCDs stream over one array (read, write) or two (copy);
// the 4-KiB pages an XCD takes are chosen by ONE address bit: XCDs 0-3 take the pages whose bit `s` (12..17) is
// `flip`, XCDs 4-7 the others.  If that bit decides which half of the package (which IO dies' HBM stacks) a page
// lives in, one of flip = 0 / 1 is all-local traffic and the other all-remote; "none" is the usual mapping
// (page index = workgroup sequence).  xcd_stack_probe showed a 1.5 % write-rate pattern with exactly this shape
// (period 16 KiB, phase opposite for XCDs 0-3 and 4-7) on an idle fabric; this probe loads the fabric.
//
// Usage: xcd_half_probe [arrays, default 3]
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

// op 0 read, 1 write, 2 copy src -> dst.  bit < 0: page = j * 8 + xcd (plain order).  Otherwise the workgroup's
// k-th page is the k-th page (in address order) whose bit `bit` of the byte offset equals want = (xcd >> 2) ^ flip,
// the four XCDs of a half sharing those pages round-robin.
template <int OP>
__global__ __launch_bounds__(256) void half_kernel(const v4f* __restrict__ src, v4f* __restrict__ dst, unsigned pages, int bit, unsigned flip,
                                                   unsigned wgs_per_xcd, float* sink)
{
  const unsigned x = blockIdx.x & 7, j = blockIdx.x >> 3;
  float acc = 0.f;
  const unsigned half_pages = pages >> 1;
  for (unsigned k = j;; k += wgs_per_xcd) {
    size_t p;
    if (bit < 0) {
      p = (size_t)k * 8 + x;
      if (p >= pages)
        break;
    } else {
      // index among the pages of this half: m = k * 4 + (x & 3); m-th page with the wanted bit
      const unsigned m = k * 4 + (x & 3);
      if (m >= half_pages)
        break;
      const unsigned sh = (unsigned)bit - 12; // bit position within the page index
      const unsigned want = (x >> 2) ^ flip;
      const unsigned lowmask = (1u << sh) - 1;
      p = ((size_t)(m >> sh) << (sh + 1)) | ((size_t)want << sh) | (m & lowmask);
      if (p >= pages)
        break;
    }
    const v4f* s = src + p * 256 + threadIdx.x;
    v4f* d = dst + p * 256 + threadIdx.x;
    if (OP == 0) {
      const v4f t = *s;
      acc += t.x + t.y + t.z + t.w;
    } else if (OP == 1) {
      const float f = (float)k;
      __builtin_nontemporal_store(v4f{f, f, f, f}, d);
    } else {
      __builtin_nontemporal_store(*s, d);
    }
  }
  if (OP == 0 && acc == 123456.789f)
    sink[0] = acc;
}

int main(int argc, char** argv)
{
  const int narr = argc > 1 ? std::atoi(argv[1]) : 3;
  const size_t N = (size_t)1440 * 720 * 137 * 4;
  const unsigned pages = (unsigned)(N / 4096) & ~63u;
  std::vector<v4f*> arr(narr + 1);
  for (int i = 0; i <= narr; ++i) {
    CHECK(hipMalloc(&arr[i], N));
    CHECK(hipMemset(arr[i], 0, N));
  }
  float* sink;
  CHECK(hipMalloc(&sink, 64));
  const unsigned wgs = 512; // per XCD: 16 per CU
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto run = [&](int op, const v4f* s, v4f* d, int bit, unsigned flip) {
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CHECK(hipEventRecord(e0, 0));
      if (op == 0)
        hipLaunchKernelGGL(half_kernel<0>, dim3(8 * wgs), dim3(256), 0, 0, s, d, pages, bit, flip, wgs, sink);
      else if (op == 1)
        hipLaunchKernelGGL(half_kernel<1>, dim3(8 * wgs), dim3(256), 0, 0, s, d, pages, bit, flip, wgs, sink);
      else
        hipLaunchKernelGGL(half_kernel<2>, dim3(8 * wgs), dim3(256), 0, 0, s, d, pages, bit, flip, wgs, sink);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float t;
      CHECK(hipEventElapsedTime(&t, e0, e1));
      best = std::min(best, t);
    }
    return (double)pages * 4096.0 * (op == 2 ? 2 : 1) / (best * 1e-3) / 1e9;
  };
  for (int k = 0; k < 20; ++k)
    run(2, arr[0], arr[1], -1, 0);
  const char* opname[3] = {"read", "write", "copy"};
  for (int op = 0; op < 3; ++op) {
    for (int i = 0; i < narr; ++i) {
      std::printf("%-5s array %d%s: plain %5.0f GB/s |", opname[op], i, op == 2 ? " -> next" : "", run(op, arr[i], op == 2 ? arr[i + 1] : arr[i], -1, 0));
      for (int bit = 12; bit <= 18; ++bit)
        std::printf("  bit %d: %5.0f / %5.0f", bit, run(op, arr[i], op == 2 ? arr[i + 1] : arr[i], bit, 0), run(op, arr[i], op == 2 ? arr[i + 1] : arr[i], bit, 1));
      std::printf("\n");
      std::fflush(stdout);
    }
  }
  return 0;
}
