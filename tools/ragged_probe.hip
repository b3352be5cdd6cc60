// Two questions behind a row-tiled kernel for widths that are not a multiple of 4 (rows then start at dword-aligned, not
// 16-byte-aligned, addresses):
//  (1) does global_load_lds_dwordx4 accept a global address that is only dword-aligned, and deliver the right bytes?
//  (2) what do the stores cost: 16-byte stores at misaligned addresses, against four dword stores per lane in which the
//      lanes of a wave write 64 CONSECUTIVE floats (each instruction one contiguous 256-byte run at dword alignment),
//      against aligned 16-byte stores -- plain and nontemporal.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ragged_probe tools/ragged_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void lds_dma_probe(const float* src, float* dst, int shift)
{
  __shared__ v4f buf[64];
  const float* p = src + shift + 4 * threadIdx.x; // dword-aligned only when shift % 4 != 0
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)p, (void __attribute__((address_space(3)))*)&buf[0], 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const v4f q = buf[threadIdx.x];
  dst[4 * threadIdx.x + 0] = q.x;
  dst[4 * threadIdx.x + 1] = q.y;
  dst[4 * threadIdx.x + 2] = q.z;
  dst[4 * threadIdx.x + 3] = q.w;
}

// MODE 0: aligned 16-byte stores; 1: the same nontemporal; 2: 16-byte stores, every row `shift` floats off alignment;
// 3: four dword stores per lane, lane l of a wave writes float (k * 64 + l) of its 256-float run, rows `shift` floats off;
// 4: as 3, nontemporal
template <int MODE>
__global__ __launch_bounds__(256) void store_probe(float* dst, size_t n4, int shift)
{
  struct __attribute__((packed, aligned(4))) U4
  {
    v4f v;
  };
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < n4; q += stride) {
    const v4f val = {(float)q, 1.f, 2.f, 3.f};
    if (MODE == 0)
      *reinterpret_cast<v4f*>(dst + 4 * q) = val;
    else if (MODE == 1)
      __builtin_nontemporal_store(val, reinterpret_cast<v4f*>(dst + 4 * q));
    else if (MODE == 2) {
      U4 t;
      t.v = val;
      *reinterpret_cast<U4*>(dst + 4 * q + shift) = t;
    } else {
      const size_t wave_base = (q & ~(size_t)63) * 4; // first float of this wave's 256-float run
      const int lane = (int)(q & 63);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float* p = dst + wave_base + shift + k * 64 + lane;
        if (MODE == 3)
          *p = val[k];
        else
          __builtin_nontemporal_store(val[k], p);
      }
    }
  }
}

template <int MODE>
float time_mode(float* d, size_t n4, int shift)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int k = 0; k < 3; ++k)
    hipLaunchKernelGGL(store_probe<MODE>, dim3(16384), dim3(256), 0, 0, d, n4, shift);
  CK(hipEventRecord(e0, 0));
  for (int k = 0; k < 10; ++k)
    hipLaunchKernelGGL(store_probe<MODE>, dim3(16384), dim3(256), 0, 0, d, n4, shift);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 10;
}

int main()
{
  // (1)
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i)
    h[i] = (float)i;
  float *s, *d;
  CK(hipMalloc(&s, 4096));
  CK(hipMalloc(&d, 4096));
  CK(hipMemcpy(s, h.data(), 4096, hipMemcpyHostToDevice));
  for (int shift = 0; shift < 4; ++shift) {
    hipLaunchKernelGGL(lds_dma_probe, dim3(1), dim3(64), 0, 0, s, d, shift);
    CK(hipDeviceSynchronize());
    std::vector<float> r(256);
    CK(hipMemcpy(r.data(), d, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 256; ++i)
      bad += r[i] != (float)(i + shift);
    printf("global_load_lds_dwordx4 from an address %d floats off 16-byte alignment: %s (%d of 256 values wrong)\n", shift, bad ? "WRONG" : "correct", bad);
  }
  // (2)
  const size_t n4 = (size_t)1 << 27; // 2 GiB of floats
  float* big;
  CK(hipMalloc(&big, n4 * 16 + 4096));
  for (int shift = 0; shift < 4; shift += (shift ? 2 : 1)) {
    const float a = time_mode<0>(big, n4, 0), nt = time_mode<1>(big, n4, 0), mis = time_mode<2>(big, n4, shift), dw = time_mode<3>(big, n4, shift),
                dwnt = time_mode<4>(big, n4, shift);
    const double gb = n4 * 16 / 1e6;
    printf("shift %d floats: aligned 16 B %.3f ms (%.0f GB/s) | nontemporal %.3f (%.0f) | 16 B at the shifted address %.3f (%.0f) | 4 dword stores, 256-B runs %.3f (%.0f) | "
           "the same nontemporal %.3f (%.0f)\n",
           shift, a, gb / a, nt, gb / nt, mis, gb / mis, dw, gb / dw, dwnt, gb / dwnt);
  }
  return 0;
}
