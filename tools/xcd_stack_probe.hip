// xcd_stack_probe.hip -- measurement only (tools/): is the read / write rate of ONE XCD on ONE residue class of
// 4-KiB pages of an array the same for every (XCD, class) pair?  If physical memory is interleaved over the HBM
// stacks page by page and an XCD reaches the stacks of its own IO die faster than the others, the matrix shows it,
// and the offsets between the rows of different arrays are what a batch placement has to line up.
//
// Usage: xcd_stack_probe [arrays, default 4] [page classes, default 16] [page bytes, default 4096]
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

__device__ __forceinline__ unsigned xcc_id()
{
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

__global__ void which_xcd(unsigned* out)
{
  if (threadIdx.x == 0)
    out[blockIdx.x] = xcc_id();
}

// Workgroups on XCD `x` read (WRITE = false) or write pages p = r + classes * k of the array; the others leave.
// page_q = float4 per page (256 for 4 KiB); a workgroup of 256 lanes takes page_q / 256 float4 per lane and page.
template <bool WRITE>
__global__ __launch_bounds__(256) void probe(v4f* __restrict__ a, unsigned x, unsigned r, unsigned classes, unsigned page_q, unsigned pages_per_class,
                                             unsigned wgs_per_xcd, float* sink)
{
  if (xcc_id() != x)
    return;
  const unsigned j = blockIdx.x >> 3; // index among the workgroups of this XCD (round-robin dispatch; checked by which_xcd)
  float acc = 0.f;
  for (unsigned k = j; k < pages_per_class; k += wgs_per_xcd) {
    const size_t p = (size_t)r + (size_t)classes * k;
    v4f* base = a + p * page_q;
    for (unsigned o = threadIdx.x; o < page_q; o += 256) {
      if (WRITE) {
        const float f = (float)k;
        __builtin_nontemporal_store(v4f{f, f, f, f}, base + o);
      } else {
        const v4f t = base[o];
        acc += t.x + t.y + t.z + t.w;
      }
    }
  }
  if (!WRITE && acc == 123456.789f)
    sink[0] = acc;
}

int main(int argc, char** argv)
{
  const int narr = argc > 1 ? std::atoi(argv[1]) : 4;
  const unsigned classes = argc > 2 ? (unsigned)std::atoi(argv[2]) : 16;
  const unsigned page_bytes = argc > 3 ? (unsigned)std::atoi(argv[3]) : 4096;
  const size_t N = (size_t)1440 * 720 * 137 * 4; // bytes per array: the headline batch
  const unsigned page_q = page_bytes / 16;
  const unsigned pages = (unsigned)(N / page_bytes);
  const unsigned pages_per_class = pages / classes;
  std::vector<v4f*> arr(narr);
  for (int i = 0; i < narr; ++i) {
    CHECK(hipMalloc(&arr[i], N));
    CHECK(hipMemset(arr[i], 0, N));
  }
  float* sink;
  CHECK(hipMalloc(&sink, 64));
  const unsigned grid = 8 * 256; // 256 workgroups per XCD: 8 per CU
  unsigned* d_x;
  CHECK(hipMalloc(&d_x, grid * 4));
  hipLaunchKernelGGL(which_xcd, dim3(grid), dim3(64), 0, 0, d_x);
  std::vector<unsigned> hx(grid);
  CHECK(hipMemcpy(hx.data(), d_x, grid * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (unsigned b = 0; b < grid; ++b)
    bad += hx[b] != (b & 7);
  std::printf("workgroup b runs on XCD b %% 8: %s (%d of %u differ); first 16: ", bad ? "NO" : "yes", bad, grid);
  for (int b = 0; b < 16; ++b)
    std::printf("%u ", hx[b]);
  std::printf("\n%u pages of %u B per array, %u classes, %u pages (%.1f MB) per probe\n", pages, page_bytes, classes, pages_per_class,
              pages_per_class * (double)page_bytes / 1e6);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int wr = 0; wr < 2; ++wr) {
    for (int i = 0; i < narr; ++i) {
      std::printf("array %d (%p) %s, GB/s by XCD (rows) and page class (columns)\n", i, (void*)arr[i], wr ? "WRITE" : "READ");
      for (unsigned x = 0; x < 8; ++x) {
        std::printf("  xcd %u:", x);
        for (unsigned r = 0; r < classes; ++r) {
          float best = 1e30f;
          for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            if (wr)
              hipLaunchKernelGGL(probe<true>, dim3(grid), dim3(256), 0, 0, arr[i], x, r, classes, page_q, pages_per_class, 256u, sink);
            else
              hipLaunchKernelGGL(probe<false>, dim3(grid), dim3(256), 0, 0, arr[i], x, r, classes, page_q, pages_per_class, 256u, sink);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float t;
            CHECK(hipEventElapsedTime(&t, e0, e1));
            best = std::min(best, t);
          }
          std::printf(" %5.0f", pages_per_class * (double)page_bytes / (best * 1e-3) / 1e9);
        }
        std::printf("\n");
        std::fflush(stdout);
      }
    }
  }
  return 0;
}
