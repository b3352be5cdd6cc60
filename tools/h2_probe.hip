// h2_probe.hip -- measurement only (tools/): how much contiguous memory must ONE workgroup touch
// at ONE time for the HBM system of MI355X to stream at its best rate?
//
// Round 1 found the headline kernel's write side at 5.1 TB/s against 6.4 TB/s for a linear fill,
// its read side at 6.4, and a one-shot 4-row x 1-KiB tile fill in between (5.6) -- with any row
// pitch (round 2: 1440, 1536, 1280, 2048 columns all alike).  Hypothesis: what matters is the extent
// a workgroup covers contiguously in one go.  This probe issues exactly the same number of 1-KiB
// wave-instructions (64 lanes x 16 B) in every mode and only permutes WHICH KiB a wave touches:
//
//   kib(b, w, j) for workgroup b, wave w (of W), instruction j (of I) -- q = w*I + j in [0, W*I):
//     piece p = q / c, offset o = q % c             (c = contiguous KiB per piece)
//     kib = (b / G) * G*W*I + (p * G + b % G) * c + o    (G workgroups interleave their pieces)
//   c = W*I or G = 1 is the linear streaming order.
//
// Usage: h2_probe [MiB per array, default 1083]   -> one line per mode: write-only, read-only, copy.
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

struct Mode
{
  int W, I, c, G;
};

__device__ __forceinline__ size_t kib_index(const Mode m, size_t b, int w, int j)
{
  const int q = w * m.I + j;
  const int p = q / m.c, o = q % m.c;
  const size_t per_wg = (size_t)m.W * m.I;
  return (b / m.G) * m.G * per_wg + ((size_t)p * m.G + b % m.G) * m.c + o;
}

// op 0: write only, 1: read only, 2: copy (read with mode mr, write with mode mw)
template <int I, int OP, bool NT>
__global__ __launch_bounds__(1024) void probe_kernel(const Mode mr, const Mode mw, const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n_kib,
                                                      float* sink)
{
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t b = blockIdx.x;
  v4f val[I];
  if (OP != 0) {
#pragma unroll
    for (int j = 0; j < I; ++j) {
      const size_t k = kib_index(mr, b, w, j);
      const v4f* p = src + (k < n_kib ? k : 0) * 64 + lane;
      val[j] = NT ? __builtin_nontemporal_load(p) : *p;
    }
  } else {
#pragma unroll
    for (int j = 0; j < I; ++j) {
      const float f = (float)(b + j);
      val[j] = v4f{f, f + 1.f, f + 2.f, f + 3.f};
    }
  }
  if (OP == 1) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < I; ++j)
      acc += val[j].x + val[j].y + val[j].z + val[j].w;
    if (acc == 123456.789f) // never: keeps the loads alive
      sink[0] = acc;
    return;
  }
#pragma unroll
  for (int j = 0; j < I; ++j) {
    const size_t k = kib_index(mw, b, w, j);
    if (k < n_kib) {
      v4f* p = dst + k * 64 + lane;
      if (NT)
        __builtin_nontemporal_store(val[j], p);
      else
        *p = val[j];
    }
  }
}

template <int OP>
float run(const Mode mr, const Mode mw, const v4f* src, v4f* dst, size_t n_kib, float* sink, bool nt, int reps)
{
  const Mode m = (OP == 1) ? mr : mw;
  const size_t per_wg = (size_t)m.W * m.I;
  const size_t groups = (n_kib + per_wg * m.G - 1) / (per_wg * m.G);
  const unsigned grid = (unsigned)(groups * m.G);
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  std::vector<float> ms;
  for (int r = 0; r < reps + 2; ++r) {
    CHECK(hipEventRecord(a, 0));
#define LAUNCH(II)                                                                                                                   \
  if (nt)                                                                                                                            \
    hipLaunchKernelGGL((probe_kernel<II, OP, true>), dim3(grid), dim3(64 * m.W), 0, 0, mr, mw, src, dst, n_kib, sink);               \
  else                                                                                                                               \
    hipLaunchKernelGGL((probe_kernel<II, OP, false>), dim3(grid), dim3(64 * m.W), 0, 0, mr, mw, src, dst, n_kib, sink)
    if (m.I == 1) {
      LAUNCH(1);
    } else if (m.I == 2) {
      LAUNCH(2);
    } else {
      LAUNCH(4);
    }
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float t = 0;
    CHECK(hipEventElapsedTime(&t, a, b));
    if (r >= 2)
      ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  CHECK(hipEventDestroy(a));
  CHECK(hipEventDestroy(b));
  return ms[ms.size() / 2];
}

int main(int argc, char** argv)
{
  const size_t mib = argc > 1 ? (size_t)std::atol(argv[1]) : 1083; // two 568 MB outputs of the headline batch
  const size_t n_kib = mib * 1024;
  v4f *src = nullptr, *dst = nullptr;
  float* sink = nullptr;
  CHECK(hipMalloc((void**)&src, n_kib * 1024));
  CHECK(hipMalloc((void**)&dst, n_kib * 1024));
  CHECK(hipMalloc((void**)&sink, 64));
  CHECK(hipMemset(src, 1, n_kib * 1024));
  CHECK(hipMemset(dst, 0, n_kib * 1024));
  const double gb = (double)n_kib * 1024 / 1e9;
  std::printf("%zu MiB per array (%.3f GB); times are medians of 7 in ms; TB/s counts the bytes moved (copy: read + written)\n", mib, gb);
  std::printf("%-44s %9s %7s %9s %7s %9s %7s\n", "mode  W waves x I instr, c KiB pieces, G-way", "write ms", "TB/s", "read ms", "TB/s", "copy ms", "TB/s");
  const Mode lin = {4, 1, 4, 1};
  struct Row
  {
    const char* name;
    Mode m;
  };
  const Row rows[] = {
      {"linear 4 KiB per WG (W4 I1)", {4, 1, 4, 1}},
      {"linear 8 KiB per WG (W8 I1)", {8, 1, 8, 1}},
      {"linear 16 KiB per WG (W16 I1)", {16, 1, 16, 1}},
      {"linear 8 KiB per WG (W4 I2)", {4, 2, 8, 1}},
      {"linear 16 KiB per WG (W4 I4)", {4, 4, 16, 1}},
      {"1-KiB pieces, 4-way  (W4 I1 c1 G4)", {4, 1, 1, 4}},
      {"1-KiB pieces, 16-way (W4 I1 c1 G16)", {4, 1, 1, 16}},
      {"1-KiB pieces, 64-way (W4 I1 c1 G64)", {4, 1, 1, 64}},
      {"2-KiB pieces, 16-way (W4 I1 c2 G16)", {4, 1, 2, 16}},
      {"2-KiB pieces, 16-way (W8 I1 c2 G16)", {8, 1, 2, 16}},
      {"4-KiB pieces, 16-way (W8 I1 c4 G16)", {8, 1, 4, 16}},
      {"4-KiB pieces, 16-way (W16 I1 c4 G16)", {16, 1, 4, 16}},
      {"8-KiB pieces, 16-way (W16 I1 c8 G16)", {16, 1, 8, 16}},
      {"2-KiB pieces, 16-way (W4 I4 c2 G16)", {4, 4, 2, 16}},
      {"4-KiB pieces, 16-way (W4 I4 c4 G16)", {4, 4, 4, 16}},
      {"8-KiB pieces, 16-way (W4 I4 c8 G16)", {4, 4, 8, 16}},
      {"2-KiB pieces, 4-way  (W8 I1 c2 G4)", {8, 1, 2, 4}},
      {"2-KiB pieces, 64-way (W8 I1 c2 G64)", {8, 1, 2, 64}},
  };
  for (int nt = 0; nt < 2; ++nt) {
    std::printf("-- %s loads/stores\n", nt ? "nontemporal" : "plain");
    for (const Row& r : rows) {
      const float tw = run<0>(r.m, r.m, src, dst, n_kib, sink, nt, 7);
      const float tr = run<1>(r.m, r.m, src, dst, n_kib, sink, nt, 7);
      const float tc = run<2>(r.m, r.m, src, dst, n_kib, sink, nt, 7);
      std::printf("%-44s %9.4f %7.2f %9.4f %7.2f %9.4f %7.2f\n", r.name, tw, gb / tw, tr, gb / tr, tc, 2 * gb / tc);
    }
    // mixed: scattered reads with linear writes and the other way round (same W, I)
    const Mode sc = {4, 1, 1, 16};
    const float t1 = run<2>(sc, lin, src, dst, n_kib, sink, nt, 7);
    const float t2 = run<2>(lin, sc, src, dst, n_kib, sink, nt, 7);
    std::printf("%-44s %9s %7s %9s %7s %9.4f %7.2f\n", "copy: 1-KiB 16-way reads, linear writes", "", "", "", "", t1, 2 * gb / t1);
    std::printf("%-44s %9s %7s %9s %7s %9.4f %7.2f\n", "copy: linear reads, 1-KiB 16-way writes", "", "", "", "", t2, 2 * gb / t2);
  }
  return 0;
}
