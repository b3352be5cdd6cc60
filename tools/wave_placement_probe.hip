// Where do the waves of a workgroup land?  Launches workgroups of W waves holding L KiB of LDS (so that only as many
// fit a CU as in the real kernels), lets every wave record its hardware SIMD / CU / SE ids (s_getreg HW_ID) and XCC id,
// and prints, per workgroup size, how the waves of a workgroup spread over the four SIMDs of its CU.
//   hipcc --offload-arch=gfx950 -O2 -o tools/wave_placement_probe tools/wave_placement_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

__global__ void probe(unsigned int* out, unsigned long long* times, int spin)
{
  extern __shared__ unsigned int lds[];
  const unsigned long long t0 = __builtin_readcyclecounter();
  const int wave = threadIdx.x >> 6;
  unsigned int hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // stay resident for a while so that the chip fills up like under a real launch
  unsigned int x = threadIdx.x;
  for (int i = 0; i < spin; ++i)
    x = x * 1664525u + 1013904223u;
  lds[threadIdx.x] = x;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 32 + wave) * 2 + 0] = hw;
    out[(blockIdx.x * 32 + wave) * 2 + 1] = (xcc & 0xf) | (lds[(threadIdx.x + 1) % blockDim.x] & 0x80000000u ? 0 : 0);
  }
  if (threadIdx.x == 0) {
    const unsigned long long t1 = __builtin_readcyclecounter();
    times[blockIdx.x * 2] = t0;
    times[blockIdx.x * 2 + 1] = t1;
  }
}

int main(int argc, char** argv)
{
  const int nblocks = argc > 1 ? atoi(argv[1]) : 2048;
  const int configs[][2] = {{10, 31}, {12, 37}, {13, 43}, {14, 43}, {14, 85}, {15, 49}, {16, 49}};
  unsigned int* d;
  (void)hipMalloc(&d, sizeof(unsigned int) * nblocks * 64);
  std::vector<unsigned int> h(nblocks * 64);
  unsigned long long* dt;
  (void)hipMalloc(&dt, sizeof(unsigned long long) * nblocks * 2);
  std::vector<unsigned long long> ht(nblocks * 2);
  for (auto& cfg : configs) {
    const int W = cfg[0], L = cfg[1];
    (void)hipMemset(d, 0xff, sizeof(unsigned int) * nblocks * 64);
    (void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, L * 1024);
    hipLaunchKernelGGL(probe, dim3(nblocks), dim3(64 * W), L * 1024, 0, d, dt, 20000);
    if (hipDeviceSynchronize() != hipSuccess) {
      printf("launch failed W=%d\n", W);
      continue;
    }
    (void)hipMemcpy(h.data(), d, sizeof(unsigned int) * nblocks * 64, hipMemcpyDeviceToHost);
    (void)hipMemcpy(ht.data(), dt, sizeof(unsigned long long) * nblocks * 2, hipMemcpyDeviceToHost);
    // how many workgroups were resident on one CU at the same time (the clock is per XCC: compare within a CU only)
    std::map<unsigned int, std::vector<std::pair<unsigned long long, int>>> per_cu;
    for (int b = 0; b < nblocks; ++b) {
      const unsigned int hw = h[(b * 32) * 2], xcc = h[(b * 32) * 2 + 1] & 0xf;
      const unsigned int cu = (xcc << 16) | ((hw >> 8) & 0xff); // cu_id, sh_id, se_id
      per_cu[cu].push_back({ht[b * 2], +1});
      per_cu[cu].push_back({ht[b * 2 + 1], -1});
    }
    std::map<int, int> max_resident;
    for (auto& c : per_cu) {
      std::sort(c.second.begin(), c.second.end());
      int cur = 0, mx = 0;
      for (auto& e : c.second) {
        cur += e.second;
        mx = cur > mx ? cur : mx;
      }
      max_resident[mx]++;
    }
    std::map<std::string, int> patterns;    // waves per SIMD of a workgroup, e.g. "4 4 3 3"
    std::map<std::string, int> last_two;    // SIMDs of the last two waves
    std::map<std::string, int> seq;         // SIMD sequence of the waves of a workgroup
    for (int b = 0; b < nblocks; ++b) {
      int per[4] = {0, 0, 0, 0};
      char s[128];
      int n = 0;
      for (int w = 0; w < W; ++w) {
        const unsigned int hw = h[(b * 32 + w) * 2];
        const int simd = (hw >> 4) & 3;
        per[simd]++;
        n += snprintf(s + n, sizeof s - n, "%d", simd);
      }
      seq[s]++;
      snprintf(s, sizeof s, "%d %d %d %d", per[0], per[1], per[2], per[3]);
      patterns[s]++;
      snprintf(s, sizeof s, "%d %d", (h[(b * 32 + W - 2) * 2] >> 4) & 3, (h[(b * 32 + W - 1) * 2] >> 4) & 3);
      last_two[s]++;
    }
    printf("== %d waves per workgroup, %d KiB LDS, %d workgroups on %zu CUs\n  workgroups resident on a CU at once (max):", W, L, nblocks, per_cu.size());
    for (auto& p : max_resident)
      printf("  %d: %d CUs", p.first, p.second);
    printf("\n  waves per SIMD (0 1 2 3):");
    for (auto& p : patterns)
      printf("  [%s] x%d", p.first.c_str(), p.second);
    printf("\n  SIMDs of the last two waves:");
    for (auto& p : last_two)
      printf("  [%s] x%d", p.first.c_str(), p.second);
    printf("\n  most frequent SIMD sequences:");
    std::vector<std::pair<int, std::string>> v;
    for (auto& p : seq)
      v.push_back({p.second, p.first});
    std::sort(v.rbegin(), v.rend());
    for (size_t i = 0; i < v.size() && i < 4; ++i)
      printf("  %s x%d", v[i].second.c_str(), v[i].first);
    printf("\n");
  }
  (void)hipFree(d);
  return 0;
}
