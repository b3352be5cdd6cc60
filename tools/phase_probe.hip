// phase_probe.hip -- measurement only (tools/): does the time of a 2-in / 2-out stream depend on the PHASE
// between its streams?
//
// Round 2 found the headline kernel (and a plain copy of the same bytes) 8 % slower on some placements of its
// four arrays than on others, bimodally (0.40 / 0.435 ms), stable for the life of the arrays, worst for arrays
// allocated one after the other.  Hypothesis: the element a wave reads from u and the one it reads from v at
// the same moment (same index i) fall into the same DRAM bank at different rows whenever the two arrays'
// physical base addresses differ by the wrong amount -- the classic STREAM array-alignment conflict -- and the
// same for the two stores.  Then shifting WHICH index of v (of diverg) a wave touches relative to u (rvort)
// must change the time on a slow placement, and must not on a fast one.
//
//   thread q:  c[q] = a[q] + b[q (+) sb];   d[q (+) sd] = a[q] - b[q (+) sb]      ((+) wraps at the array end)
//
// Usage: phase_probe [pool size, default 16]
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

__global__ __launch_bounds__(256) void stream22(const v4f* __restrict__ a, const v4f* __restrict__ b, v4f* __restrict__ c, v4f* __restrict__ d,
                                                 unsigned n4, unsigned sb, unsigned sd)
{
  const unsigned q = blockIdx.x * 256u + threadIdx.x;
  if (q >= n4)
    return;
  unsigned qb = q + sb;
  if (qb >= n4)
    qb -= n4;
  unsigned qd = q + sd;
  if (qd >= n4)
    qd -= n4;
  const v4f x = a[q], y = b[qb];
  __builtin_nontemporal_store(x + y, c + q);
  __builtin_nontemporal_store(x - y, d + qd);
}

static const size_t N = (size_t)1440 * 720 * 137; // floats per array: the headline batch

static float time_ms(const v4f* a, const v4f* b, v4f* c, v4f* d, unsigned sb, unsigned sd)
{
  const unsigned n4 = (unsigned)(N / 4);
  const dim3 grid((n4 + 255) / 256), block(256);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int rep = 0; rep < 6; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    for (int k = 0; k < 4; ++k)
      hipLaunchKernelGGL(stream22, grid, block, 0, 0, a, b, c, d, n4, sb, sd);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float t;
    CHECK(hipEventElapsedTime(&t, e0, e1));
    if (rep)
      ms.push_back(t / 4);
  }
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2];
}

int main(int argc, char** argv)
{
  const int m = argc > 1 ? std::atoi(argv[1]) : 16;
  std::vector<v4f*> pool(m);
  for (int i = 0; i < m; ++i) {
    CHECK(hipMalloc(&pool[i], N * 4));
    CHECK(hipMemset(pool[i], 0, N * 4));
  }
  std::printf("pool of %d arrays of %zu bytes; addresses (MiB, relative to the lowest):", m, N * 4);
  size_t lo = (size_t)-1;
  for (int i = 0; i < m; ++i)
    lo = std::min(lo, (size_t)pool[i]);
  for (int i = 0; i < m; ++i)
    std::printf(" %zu", ((size_t)pool[i] - lo) >> 20);
  std::printf("\n");
  // warm the clocks
  for (int k = 0; k < 50; ++k)
    time_ms(pool[0], pool[1], pool[2], pool[3], 0, 0);

  struct Shift
  {
    const char* name;
    unsigned q;
  };
  const Shift shifts[] = {{"0", 0},           {"256B", 16},        {"1KiB", 64},          {"row 5760B", 360},      {"4KiB", 256},
                          {"64KiB", 4096},    {"1MiB", 65536},     {"level 4147200B", 259200}, {"64MiB", 1u << 22}, {"half", (unsigned)(N / 8)}};
  const int ns = sizeof(shifts) / sizeof(shifts[0]);
  std::vector<std::vector<int>> combos;
  for (int b = 0; b + 3 < m && b < 12; b += 4)
    combos.push_back({b, b + 1, b + 2, b + 3}); // allocated one after the other
  if (m >= 16) {
    combos.push_back({0, 5, 10, 15});
    combos.push_back({1, 5, 9, 13});
    combos.push_back({0, 8, 1, 9}); // u,rv adjacent; v,dg adjacent
    combos.push_back({0, 1, 8, 9}); // u,v adjacent; rv,dg adjacent
  }
  std::printf("%-16s", "combo \\ shift");
  for (int s = 0; s < ns; ++s)
    std::printf(" %14s", shifts[s].name);
  std::printf("\n");
  for (auto& c : combos) {
    for (int which = 0; which < 3; ++which) { // 0: both shifted, 1: only the second input, 2: only the second output
      char label[64];
      std::snprintf(label, sizeof label, "%d,%d,%d,%d %s", c[0], c[1], c[2], c[3], which == 0 ? "in+out" : which == 1 ? "in" : "out");
      std::printf("%-16s", label);
      for (int s = 0; s < ns; ++s) {
        const unsigned q = shifts[s].q;
        const float t = time_ms(pool[c[0]], pool[c[1]], pool[c[2]], pool[c[3]], which == 2 ? 0 : q, which == 1 ? 0 : q);
        std::printf(" %14.4f", t);
      }
      std::printf("\n");
      std::fflush(stdout);
    }
  }
  return 0;
}
