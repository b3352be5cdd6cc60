// mifc_advection.hip -- advection (FieldCalculations.cc:1942-1983) as a one-shot vector kernel.
//
// advec = (u * dfdx + v * dfdy) * (-3600 * hours): a 5-point stencil on ONE field with the wind at the
// centre, so there is no row reuse worth a row walk -- seven float4 loads per lane (f above / centre /
// below, u, v, xmapr, ymapr), the x-neighbours of the centre row from the adjacent lanes (DPP wave
// shifts) and one edge scalar per wave, one float4 store, no loop.  A workgroup is 4 waves = 4 rows x
// 256 columns of one level.  (The one-lane-per-cell kernel of mifc_stencil.hip, which this replaces for
// nx % 4 == 0, moved 4 bytes per lane and load instruction.)
//
// Reference semantics (mifc_stencil.hip header): the flat loop evaluates the edge columns with
// neighbours wrapped into the adjacent row -- they take part in the count -- then fillEdges overwrites
// columns 0 / nx-1 and rows 0 / ny-1.
#include "mifc_device.h"
#include "mifc_kernels.h"

namespace mifc {

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float adv_from_lower_lane(float keep_if_none, float x) // lane i <- lane i-1
{
  return __builtin_bit_cast(float,
                            __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep_if_none), __builtin_bit_cast(int, x), 0x138 /*wave_shr:1*/, 0xf, 0xf, false));
}
__device__ __forceinline__ float adv_from_upper_lane(float keep_if_none, float x) // lane i <- lane i+1
{
  return __builtin_bit_cast(float,
                            __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep_if_none), __builtin_bit_cast(int, x), 0x130 /*wave_shl:1*/, 0xf, 0xf, false));
}
__device__ __forceinline__ float adv_readlane(float x, int src_lane)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src_lane));
}
__device__ __forceinline__ v4f adv_load4(const float* p)
{
  return *reinterpret_cast<const v4f*>(p);
}

template <bool CHECK>
__global__ __launch_bounds__(256) void advection_oneshot_kernel(const StencilParams P, const int uB, const int uW)
{
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int seq = blockIdx.x; // address order: column segment fastest, then row block, then level
  const int per_level = uB * uW;
  const int lev = seq / per_level;
  const int rem = seq - lev * per_level;
  const int rblock = rem / uW;
  const int wc = rem - rblock * uW;
  const int nx = P.nx, ny = P.ny_global;
  // a wave past the last computed row works on that row and keeps the result to itself: it stays for the workgroup-wide
  // count at the end (one atomic per workgroup: mifc_device.h, undefined-cell counting)
  const int j_raw = 1 + rblock * 4 + wave; // rows 1 .. ny-2 are computed
  const bool live = j_raw <= ny - 2;
  const int j = live ? j_raw : ny - 2;
  const int col = wc * 256 + lane * 4;
  const bool act = live && col < nx;
  const int col_c = (col < nx) ? col : nx - 4;
  int east_col = wc * 256 + 256;
  if (east_col > nx)
    east_col = nx;
  const int edge_col = (lane == 63) ? east_col : (wc * 256 - 1); // flat neighbours: (nx-1, j-1) west of column 0, (0, j+1) east of column nx-1
  const float undef = P.undef;
  const bool all = CHECK ? (P.all_defined && P.all_defined[lev] != 0) : true;

  const float* __restrict__ f = P.f0 + (size_t)lev * P.in_level_stride;
  const float* __restrict__ u = P.f1 + (size_t)lev * P.in_level_stride;
  const float* __restrict__ v = P.f2 + (size_t)lev * P.in_level_stride;
  const long base = (long)j * nx;
  const long o = base + col_c;
  const v4f fc = adv_load4(f + o), fn = adv_load4(f + o + nx), fs = adv_load4(f + o - nx);
  const v4f u4 = adv_load4(u + o), v4 = adv_load4(v + o);
  const v4f xm4 = adv_load4(P.xmapr + o), ym4 = adv_load4(P.ymapr + o);
  const float ef = f[base + edge_col];

  const float east_f = adv_readlane(ef, 63);
  const float fW = adv_from_lower_lane(ef, fc.w); // lane 0 keeps the west scalar
  float fE = adv_from_upper_lane(ef, fc.x);       // lane 63 keeps the east scalar
  if (col + 4 >= east_col)
    fE = east_f; // my east neighbour is outside the segment (or the field)
  const float fx[6] = {fW, fc.x, fc.y, fc.z, fc.w, fE};
  float z[4];
  unsigned int bad = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float w = fx[k], e = fx[k + 2], s = fs[k], n = fn[k], uc = u4[k], vc = v4[k];
    bool ok = true;
    if (CHECK) // :1971
      ok = all | all_def(undef, uc, vc, s, w, e, n);
    const float r = (float)(((double)uc * 0.5 * (double)xm4[k] * (double)(e - w) + (double)vc * 0.5 * (double)ym4[k] * (double)(n - s)) * (double)P.scale); // :1972
    z[k] = ok ? r : undef;
    if (CHECK)
      bad += (!ok & act) ? 1u : 0u;
  }
  if (col == 0) // fillEdges, column part (:65-68)
    z[0] = z[1];
  if (col + 4 == nx)
    z[3] = z[2];
  if (act) {
    float* out = P.out0 + (size_t)lev * P.out_level_stride;
    const v4f q = {z[0], z[1], z[2], z[3]};
    const long oo = base + col;
    __builtin_nontemporal_store(q, reinterpret_cast<v4f*>(out + oo));
    if (j == 1) // fillEdges, row part (:70-73)
      __builtin_nontemporal_store(q, reinterpret_cast<v4f*>(out + oo - nx));
    if (j == ny - 2)
      __builtin_nontemporal_store(q, reinterpret_cast<v4f*>(out + oo + nx));
  }
  if (CHECK)
    block_count_add(P.n_undefined ? P.n_undefined + lev : nullptr, bad); // every wave of the workgroup is on this level
}

inline bool a16(const void* p)
{
  return (reinterpret_cast<size_t>(p) & 15u) == 0;
}

} // namespace

// Takes whole fields with nx % 4 == 0 and 16-byte aligned arrays; leaves *handled false otherwise.
hipError_t launch_advection_oneshot(const StencilParams& prm, hipStream_t stream, bool* handled)
{
  *handled = false;
  const int nx = prm.nx, ny = prm.ny_global;
  if (prm.op != ST_ADVECTION || nx % 4 != 0 || nx < 8 || ny < 3 || prm.j0 != 0 || prm.ny_local != ny)
    return hipSuccess;
  if (!prm.f0 || !prm.f1 || !prm.f2 || !prm.xmapr || !prm.ymapr || !prm.out0)
    return hipSuccess;
  if (!a16(prm.f0) || !a16(prm.f1) || !a16(prm.f2) || !a16(prm.xmapr) || !a16(prm.ymapr) || !a16(prm.out0))
    return hipSuccess;
  if (prm.in_level_stride % 4 != 0 || prm.out_level_stride % 4 != 0)
    return hipSuccess;
  const int uB = (ny - 2 + 3) / 4, uW = (nx + 255) / 256;
  const long units = (long)prm.nlev * uB * uW;
  if (units <= 0 || units > 0x7fffffffL)
    return hipSuccess;
  *handled = true;
  if (prm.every_level_all_defined)
    hipLaunchKernelGGL((advection_oneshot_kernel<false>), dim3((unsigned)units), dim3(256), 0, stream, prm, uB, uW);
  else
    hipLaunchKernelGGL((advection_oneshot_kernel<true>), dim3((unsigned)units), dim3(256), 0, stream, prm, uB, uW);
  return hipGetLastError();
}

} // namespace mifc
