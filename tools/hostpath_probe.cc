// hostpath_probe.cc -- what the legacy host-pointer path can
// get out of the host <-> HBM link on the box it runs on.  Prints one line per
// measurement; the numbers pick the staging strategy in mifc_capi.hip.
//
//   hipcc -O2 -o hostpath_probe hostpath_probe.cc -lpthread && ./hostpath_probe [MiB]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
      return 1;                                                                \
    }                                                                          \
  } while (0)

static double now()
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void par_memcpy(char* dst, const char* src, size_t n, int threads)
{
  if (threads <= 1) {
    std::memcpy(dst, src, n);
    return;
  }
  std::vector<std::thread> th;
  const size_t per = (n / threads + 4095) & ~size_t(4095);
  for (int t = 0; t < threads; ++t) {
    const size_t a = (size_t)t * per;
    if (a >= n)
      break;
    const size_t len = (a + per > n) ? n - a : per;
    th.emplace_back([=] { std::memcpy(dst + a, src + a, len); });
  }
  for (auto& t : th)
    t.join();
}

int main(int argc, char** argv)
{
  const size_t mib = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 1024;
  const size_t n = mib << 20;
  std::printf("probe size %zu MiB, host threads available %u\n", mib, std::thread::hardware_concurrency());

  char* pageable = static_cast<char*>(std::malloc(n));
  char* pageable2 = static_cast<char*>(std::malloc(n));
  std::memset(pageable, 1, n);
  std::memset(pageable2, 2, n);
  char *pinned = nullptr, *pinned2 = nullptr, *dev = nullptr, *dev2 = nullptr;
  CK(hipHostMalloc((void**)&pinned, n, hipHostMallocDefault));
  CK(hipHostMalloc((void**)&pinned2, n, hipHostMallocDefault));
  std::memset(pinned, 3, n);
  std::memset(pinned2, 4, n);
  CK(hipMalloc((void**)&dev, n));
  CK(hipMalloc((void**)&dev2, n));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const double gb = (double)n / 1e9;

  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now();
    CK(hipMemcpy(dev, pageable, n, hipMemcpyHostToDevice));
    double t1 = now();
    std::printf("H2D pageable  hipMemcpy        %7.2f GB/s\n", gb / (t1 - t0));
    t0 = now();
    CK(hipMemcpy(pageable2, dev, n, hipMemcpyDeviceToHost));
    t1 = now();
    std::printf("D2H pageable  hipMemcpy        %7.2f GB/s\n", gb / (t1 - t0));
    t0 = now();
    CK(hipMemcpyAsync(dev, pinned, n, hipMemcpyHostToDevice, s1));
    CK(hipStreamSynchronize(s1));
    t1 = now();
    std::printf("H2D pinned    hipMemcpyAsync   %7.2f GB/s\n", gb / (t1 - t0));
    t0 = now();
    CK(hipMemcpyAsync(pinned2, dev2, n, hipMemcpyDeviceToHost, s2));
    CK(hipStreamSynchronize(s2));
    t1 = now();
    std::printf("D2H pinned    hipMemcpyAsync   %7.2f GB/s\n", gb / (t1 - t0));
    t0 = now();
    CK(hipMemcpyAsync(dev, pinned, n, hipMemcpyHostToDevice, s1));
    CK(hipMemcpyAsync(pinned2, dev2, n, hipMemcpyDeviceToHost, s2));
    CK(hipStreamSynchronize(s1));
    CK(hipStreamSynchronize(s2));
    t1 = now();
    std::printf("duplex pinned H2D+D2H          %7.2f GB/s each way (%.2f total)\n", gb / (t1 - t0), 2 * gb / (t1 - t0));
  }

  // pageable copies issued from two host threads at once (the runtime stages them itself)
  {
    const double t0 = now();
    std::thread a([&] { (void)hipMemcpy(dev, pageable, n, hipMemcpyHostToDevice); });
    std::thread b([&] { (void)hipMemcpy(pageable2, dev2, n, hipMemcpyDeviceToHost); });
    a.join();
    b.join();
    const double t1 = now();
    std::printf("duplex pageable, two threads   %7.2f GB/s each way\n", gb / (t1 - t0));
  }

  // in-place pinning of a caller's buffer
  {
    double t0 = now();
    CK(hipHostRegister(pageable, n, hipHostRegisterDefault));
    double t1 = now();
    std::printf("hipHostRegister                %7.2f GB/s (%.1f ms)\n", gb / (t1 - t0), 1e3 * (t1 - t0));
    t0 = now();
    CK(hipMemcpyAsync(dev, pageable, n, hipMemcpyHostToDevice, s1));
    CK(hipStreamSynchronize(s1));
    t1 = now();
    std::printf("H2D registered                 %7.2f GB/s\n", gb / (t1 - t0));
    t0 = now();
    CK(hipHostUnregister(pageable));
    t1 = now();
    std::printf("hipHostUnregister              %7.2f GB/s (%.1f ms)\n", gb / (t1 - t0), 1e3 * (t1 - t0));
  }

  // host bounce copies pageable -> pinned
  for (int threads : {1, 2, 4, 8, 16}) {
    const double t0 = now();
    par_memcpy(pinned, pageable, n, threads);
    const double t1 = now();
    std::printf("memcpy pageable->pinned x%-2d    %7.2f GB/s\n", threads, gb / (t1 - t0));
  }
  for (int threads : {1, 4, 8}) {
    const double t0 = now();
    par_memcpy(pageable2, pinned2, n, threads);
    const double t1 = now();
    std::printf("memcpy pinned->pageable x%-2d    %7.2f GB/s\n", threads, gb / (t1 - t0));
  }

  // small transfers: what one 1440x720 field (4 MiB) costs
  {
    const size_t f = 1440 * 720 * 4;
    for (int rep = 0; rep < 3; ++rep) {
      double t0 = now();
      CK(hipMemcpyAsync(dev, pageable, f, hipMemcpyHostToDevice, s1));
      CK(hipStreamSynchronize(s1));
      double t1 = now();
      CK(hipMemcpyAsync(dev, pinned, f, hipMemcpyHostToDevice, s1));
      CK(hipStreamSynchronize(s1));
      double t2 = now();
      CK(hipMemcpyAsync(pageable2, dev, f, hipMemcpyDeviceToHost, s1));
      CK(hipStreamSynchronize(s1));
      double t3 = now();
      CK(hipMemcpyAsync(pinned2, dev, f, hipMemcpyDeviceToHost, s1));
      CK(hipStreamSynchronize(s1));
      double t4 = now();
      std::printf("one 4 MiB field: H2D pageable %.0f us, pinned %.0f us; D2H pageable %.0f us, pinned %.0f us\n", 1e6 * (t1 - t0), 1e6 * (t2 - t1),
                  1e6 * (t3 - t2), 1e6 * (t4 - t3));
    }
  }
  return 0;
}
