// mifc_shapiro.hip -- second-order Shapiro filter (FieldCalculations.cc:2076-2179).
//
// Four Jacobi-style sweeps (x, y with weight s; x, y again) between the output
// field f1 and a scratch field f2 -- each sweep reads one array and writes the
// other, so every sweep is one embarrassingly parallel launch.  8 B per cell and
// sweep; a 1440x720 field is launch-bound (4 MB).
//
// Reference behaviour kept as it is:
//   * ALL_DEFINED input: weights +0.25 for the first x/y pair, -0.25 for the
//     second; the update contains the literal `2.` and is evaluated in double.
//   * otherwise the per-cell weights (0.25 where the three cells of the stencil
//     are defined in the UNSMOOTHED field, else 0) are computed once (:2141-2145)
//     and used by BOTH pairs -- the second pair smooths again, it does not
//     restore (the `s = -0.25` of :2167 never reaches them); float arithmetic.
//   * the x sweep runs over the flat range, then columns 0 and nx-1 are restored;
//     the y sweep leaves rows 0 and ny-1 as they are.
#include "mifc_device.h"
#include "mifc_kernels.h"

namespace mifc {

namespace {

__global__ __launch_bounds__(256) void shapiro_masks_kernel(const float* __restrict__ f, int nx, int n, float undef, unsigned char* __restrict__ m1,
                                                            unsigned char* __restrict__ m2)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const bool c = is_def(f[i], undef);
    m1[i] = (i >= 1 && i < n - 1 && c && is_def(f[i - 1], undef) && is_def(f[i + 1], undef)) ? 1 : 0;     // :2142
    m2[i] = (i >= nx && i < n - nx && c && is_def(f[i - nx], undef) && is_def(f[i + nx], undef)) ? 1 : 0; // :2145
  }
}

// dst = sweep(src) along x (STEP = 1, edge = first/last column) or y (STEP = nx, edge = first/last row)
template <bool ALL, bool ALONG_X>
__global__ __launch_bounds__(256) void shapiro_sweep_kernel(const float* __restrict__ src, float* __restrict__ dst, const unsigned char* __restrict__ mask,
                                                            int nx, int ny, float s)
{
  const int n = nx * ny;
  const int step = ALONG_X ? 1 : nx;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int row = i / nx;
    const int col = i - row * nx;
    const bool edge = ALONG_X ? (col == 0 || col == nx - 1) : (row == 0 || row == ny - 1);
    const float c = src[i];
    float r = c;
    if (!edge) {
      const float a = src[i - step], b = src[i + step];
      if (ALL) // :2115, :2123
        r = (float)((double)c + (double)s * ((double)(a + b) - 2. * (double)c));
      else // :2152, :2160
        r = c + (mask[i] ? 0.25f : 0.f) * (a + b - 2 * c);
    }
    dst[i] = r;
  }
}

} // namespace

hipError_t launch_shapiro2(const ShapiroParams& P, hipStream_t stream)
{
  const int n = P.nx * P.ny;
  if (n <= 0)
    return hipSuccess;
  int grid = (n + 255) / 256;
  if (grid > 65536)
    grid = 65536;
  if (!P.all_defined)
    hipLaunchKernelGGL(shapiro_masks_kernel, dim3(grid), dim3(256), 0, stream, P.f1, P.nx, n, P.undef, P.mask_x, P.mask_y);
  float s = 0.25f;
  for (int pass = 0; pass < 2; ++pass) {
    if (P.all_defined) {
      hipLaunchKernelGGL((shapiro_sweep_kernel<true, true>), dim3(grid), dim3(256), 0, stream, P.f1, P.f2, nullptr, P.nx, P.ny, s);
      hipLaunchKernelGGL((shapiro_sweep_kernel<true, false>), dim3(grid), dim3(256), 0, stream, P.f2, P.f1, nullptr, P.nx, P.ny, s);
    } else {
      hipLaunchKernelGGL((shapiro_sweep_kernel<false, true>), dim3(grid), dim3(256), 0, stream, P.f1, P.f2, P.mask_x, P.nx, P.ny, s);
      hipLaunchKernelGGL((shapiro_sweep_kernel<false, false>), dim3(grid), dim3(256), 0, stream, P.f2, P.f1, P.mask_y, P.nx, P.ny, s);
    }
    s = -0.25f;
  }
  return hipGetLastError();
}

} // namespace mifc
