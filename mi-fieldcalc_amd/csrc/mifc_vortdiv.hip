// mifc_vortdiv.hip -- fused relative vorticity + divergence, the headline
// kernel (BASELINE.json: 1440x720x137 float32, HBM-bound, 16 B/cell).
//
// Restates relvort (FieldCalculations.cc:1843-1873) and divergence
// (:1910-1940) for a whole batch of levels in one pass; results per level are
// bit-identical to the two reference calls (double-promoted combine, no fma,
// same undefined test on (v[i-1], v[i+1], u[i-nx], u[i+nx]) for both outputs,
// flat-loop count including the wrapped edge columns, fillEdges folded into
// the store).
//
// Mapping to the hardware
//   * work unit = one wavefront = 256 columns (64 lanes x float4) x R rows of
//     one level.  The wave walks DOWN the rows keeping rows j-1, j, j+1 of u
//     and v in registers (row-sliding window): every u/v value is fetched from
//     memory once per band, halo overhead (R+2)/R on the reads only.
//   * all global accesses are 16 B per lane, 1 KiB per wave-instruction,
//     row-major and coalesced; D further rows are kept in flight per wave
//     (software prefetch ring) so that ~2(D+1) KiB per wave are outstanding.
//   * x neighbours (i-1, i+4) come from the adjacent lanes with one DPP
//     wave-shift each (v_mov_b32_dpp wave_shr:1 / wave_shl:1) -- no LDS, no
//     barrier; only lane 0 and lane 63 fetch one extra scalar per row from the
//     neighbouring wave-column (an L1/L2 hit, the neighbour streams that line).
//   * the four waves of a workgroup take FOUR CONSECUTIVE LEVELS of the same
//     (band, wave-column) tile, and the blockIdx -> tile map is XCD-aware:
//     blocks are dealt round-robin over the 8 XCDs, so block b runs tile
//     sequence (b % 8) * per_xcd + b / 8; inside one XCD consecutive blocks
//     are consecutive level-chunks of one tile, then the next band below.
//     xmapr / ymapr of a tile (2 * R KiB) are therefore re-read from that
//     XCD's L2 by every level instead of from HBM -- that is what makes
//     "map factors once per batch"  true on the chip.
//   * undefined cells are counted per lane, reduced per wave (butterfly) and
//     added with ONE atomic per wave to n_undefined[level]; the all-defined
//     instantiation contains no test and no atomic at all.
//   * no MFMA: ~14 fp64 + 16 fp32 VALU ops per cell, <20 % of the vector ALU
//     at HBM speed.
#include "mifc_device.h"
#include "mifc_kernels.h"

#include <cstdlib>
#include <cstring>

namespace mifc {

namespace {

struct RowsParams
{
  int nx;
  int nyg;       // rows of the whole field
  int j0;        // global row of owned row 0
  int ny_local;  // owned rows
  int lo, hi;    // owned local rows that are computed: [lo, hi)
  int R;         // rows per band
  int nbands, nwc, nlev;
  int chunks_per_tile; // ceil(nlev / waves per block)
  int n_logical;       // nbands * nwc * chunks_per_tile
  int per_xcd;         // ceil(n_logical / 8)
  const float *u, *v, *xm, *ym;
  float *rv, *dv;
  long in_stride, out_stride;
  const unsigned char* all_defined;
  float undef;
  u64* n_undefined;
};

__device__ __forceinline__ float dpp_from_lower_lane(float keep_if_none, float x)
{
  // lane i <- lane i-1; lane 0 keeps `keep_if_none`
  return __builtin_bit_cast(float,
                            __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep_if_none), __builtin_bit_cast(int, x), 0x138 /*wave_shr:1*/, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_upper_lane(float keep_if_none, float x)
{
  // lane i <- lane i+1; lane 63 keeps `keep_if_none`
  return __builtin_bit_cast(float,
                            __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep_if_none), __builtin_bit_cast(int, x), 0x130 /*wave_shl:1*/, 0xf, 0xf, false));
}

template <bool NT>
__device__ __forceinline__ void store4(float* p, const float4& v)
{
  typedef float v4f __attribute__((ext_vector_type(4)));
  if (NT) {
    v4f t;
    t.x = v.x;
    t.y = v.y;
    t.z = v.z;
    t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
  } else
    *reinterpret_cast<float4*>(p) = v;
}

struct RowRegs
{
  float4 u, v;
  float eu, ev; // lane 0: value west of the wave-column; lane 63: value east of it
};

template <bool CHECK, bool WANT_V, bool WANT_D, int D, bool NT>
__global__ __launch_bounds__(256) void vortdiv_rows_kernel(const RowsParams P)
{
  // ---- which tile / level does this wave own -----------------------------
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int bid = blockIdx.x;
  const int logical = (bid & 7) * P.per_xcd + (bid >> 3);
  if (logical >= P.n_logical || (bid >> 3) >= P.per_xcd)
    return;
  const int tile = logical / P.chunks_per_tile;
  const int chunk = logical - tile * P.chunks_per_tile;
  const int lev = chunk * 4 + wave;
  if (lev >= P.nlev)
    return;
  const int wc = tile / P.nbands;
  const int band = tile - wc * P.nbands;

  const int nx = P.nx;
  const int c0 = wc * 256 + lane * 4;
  const bool active = c0 < nx;
  int east_col = wc * 256 + 256; // first column east of this wave-column (may be nx: wraps to the next row)
  if (east_col > nx)
    east_col = nx;
  const bool take_east_scalar = (c0 + 4 >= east_col); // my east neighbour is outside the wave-column
  const bool edge_lane = (lane == 0) || (lane == 63);
  const int edge_col = (lane == 0) ? (c0 - 1) : east_col;

  const int jb = P.lo + band * P.R;                        // first local row of the band
  const int nr = (P.hi - jb < P.R) ? (P.hi - jb) : P.R;    // rows in this band
  const bool all = CHECK ? (P.all_defined && P.all_defined[lev] != 0) : true;
  const float undef = P.undef;

  const float* __restrict__ u = P.u + (size_t)lev * P.in_stride;
  const float* __restrict__ v = P.v + (size_t)lev * P.in_stride;
  float* rv = WANT_V ? P.rv + (size_t)lev * P.out_stride : nullptr;
  float* dv = WANT_D ? P.dv + (size_t)lev * P.out_stride : nullptr;
  const float* __restrict__ xmp = P.xm;
  const float* __restrict__ ymp = P.ym;

  // Rows jb-1 and jb+nr are only ever the north/south operands, never a centre
  // row, so they need no x-neighbour scalars (and at the global first/last row
  // those addresses would lie outside the field).
  auto load_row = [&](int row_local) -> RowRegs {
    RowRegs r;
    r.u = make_float4(0.f, 0.f, 0.f, 0.f);
    r.v = make_float4(0.f, 0.f, 0.f, 0.f);
    r.eu = 0.f;
    r.ev = 0.f;
    const long base = (long)row_local * nx;
    if (active) {
      r.u = *reinterpret_cast<const float4*>(u + base + c0);
      r.v = *reinterpret_cast<const float4*>(v + base + c0);
    }
    if (edge_lane && row_local >= jb && row_local < jb + nr) {
      r.eu = u[base + edge_col];
      r.ev = v[base + edge_col];
    }
    return r;
  };

  // ---- prologue: rows jb-1, jb, jb+1 and D more in flight -----------------
  RowRegs rp = load_row(jb - 1);
  RowRegs rc = load_row(jb);
  RowRegs rn = load_row(jb + 1); // nr >= 1, so row jb+1 exists (south halo at worst)
  RowRegs rf[D > 0 ? D : 1];
#pragma unroll
  for (int d = 0; d < D; ++d)
    rf[d] = (d + 2 <= nr) ? load_row(jb + 2 + d) : rn;

  float4 xm_c = make_float4(0.f, 0.f, 0.f, 0.f), ym_c = xm_c;
  if (active) {
    xm_c = *reinterpret_cast<const float4*>(xmp + (long)jb * nx + c0);
    ym_c = *reinterpret_cast<const float4*>(ymp + (long)jb * nx + c0);
  }

  const bool owns_top_edge = (P.j0 == 0);
  const bool owns_bottom_edge = (P.j0 + P.ny_local == P.nyg);
  unsigned int bad = 0;

  for (int r = 0; r < nr; ++r) {
    const int jl = jb + r;
    // map factors one row ahead, issued BEFORE the deep prefetch so that the
    // in-order vmcnt wait for them does not drain the prefetch
    float4 xm_n = xm_c, ym_n = ym_c;
    if (active && r + 1 < nr) {
      xm_n = *reinterpret_cast<const float4*>(xmp + (long)(jl + 1) * nx + c0);
      ym_n = *reinterpret_cast<const float4*>(ymp + (long)(jl + 1) * nx + c0);
    }
    RowRegs rnew = rn;
    if (r + 2 + D <= nr)
      rnew = load_row(jl + 2 + D);

    // ---- x neighbours of the centre row from the adjacent lanes ----------
    const float east_u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rc.eu), 63));
    const float east_v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rc.ev), 63));
    float uW = dpp_from_lower_lane(rc.eu, rc.u.w);
    float vW = dpp_from_lower_lane(rc.ev, rc.v.w);
    float uE = dpp_from_upper_lane(rc.eu, rc.u.x);
    float vE = dpp_from_upper_lane(rc.ev, rc.v.x);
    if (take_east_scalar) {
      uE = east_u;
      vE = east_v;
    }

    const float uc[6] = {uW, rc.u.x, rc.u.y, rc.u.z, rc.u.w, uE};
    const float vc[6] = {vW, rc.v.x, rc.v.y, rc.v.z, rc.v.w, vE};
    const float us[4] = {rp.u.x, rp.u.y, rp.u.z, rp.u.w}; // row j-1
    const float un[4] = {rn.u.x, rn.u.y, rn.u.z, rn.u.w}; // row j+1
    const float vs[4] = {rp.v.x, rp.v.y, rp.v.z, rp.v.w};
    const float vn[4] = {rn.v.x, rn.v.y, rn.v.z, rn.v.w};
    const float xm[4] = {xm_c.x, xm_c.y, xm_c.z, xm_c.w};
    const float ym[4] = {ym_c.x, ym_c.y, ym_c.z, ym_c.w};
    float zv[4] = {0.f, 0.f, 0.f, 0.f}, zd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float vw = vc[k], ve = vc[k + 2], uw = uc[k], ue = uc[k + 2];
      bool ok = true;
      if (CHECK)
        ok = all || (is_def(vw, undef) && is_def(ve, undef) && is_def(us[k], undef) && is_def(un[k], undef)); // :1861, :1927
      if (WANT_V)
        zv[k] = ok ? f_relvort(xm[k], ym[k], ve - vw, un[k] - us[k]) : undef;
      if (WANT_D)
        zd[k] = ok ? f_diverg(xm[k], ym[k], ue - uw, vn[k] - vs[k]) : undef;
      if (CHECK && !ok && active)
        bad += 1;
    }
    // ---- fillEdges, column part (:65-68), folded into the store ----------
    if (c0 == 0) {
      zv[0] = zv[1];
      zd[0] = zd[1];
    }
    if (c0 + 4 == nx) {
      zv[3] = zv[2];
      zd[3] = zd[2];
    }
    if (active) {
      const long o = (long)jl * nx + c0;
      const int j = P.j0 + jl;
      if (WANT_V) {
        const float4 z4 = make_float4(zv[0], zv[1], zv[2], zv[3]);
        store4<NT>(rv + o, z4);
        if (j == 1 && owns_top_edge) // row part of fillEdges (:70-73)
          store4<NT>(rv + o - nx, z4);
        if (j == P.nyg - 2 && owns_bottom_edge)
          store4<NT>(rv + o + nx, z4);
      }
      if (WANT_D) {
        const float4 d4 = make_float4(zd[0], zd[1], zd[2], zd[3]);
        store4<NT>(dv + o, d4);
        if (j == 1 && owns_top_edge)
          store4<NT>(dv + o - nx, d4);
        if (j == P.nyg - 2 && owns_bottom_edge)
          store4<NT>(dv + o + nx, d4);
      }
    }
    // ---- slide the window --------------------------------------------------
    rp = rc;
    rc = rn;
    if constexpr (D == 0) {
      rn = rnew;
    } else {
      rn = rf[0];
#pragma unroll
      for (int d = 0; d + 1 < D; ++d)
        rf[d] = rf[d + 1];
      rf[D - 1] = rnew;
    }
    xm_c = xm_n;
    ym_c = ym_n;
  }

  if (CHECK && P.n_undefined)
    wave_count_add(P.n_undefined + lev, bad);
}

struct Tuning
{
  int R;  // rows per band
  int D;  // rows kept in flight beyond the 3-row window
  int NT; // nontemporal stores
};

Tuning current_tuning()
{
  Tuning t = {32, 2, 0};
  // MIFC_VORTDIV_TUNE="R=32,D=2,NT=0" -- used by the sweep tool and the tests
  if (const char* s = std::getenv("MIFC_VORTDIV_TUNE")) {
    const char* p;
    if ((p = std::strstr(s, "R=")))
      t.R = std::atoi(p + 2);
    if ((p = std::strstr(s, "D=")))
      t.D = std::atoi(p + 2);
    if ((p = std::strstr(s, "NT=")))
      t.NT = std::atoi(p + 3);
  }
  if (t.R < 1)
    t.R = 1;
  if (t.D < 0)
    t.D = 0;
  if (t.D > 3)
    t.D = 3;
  return t;
}

template <bool CHECK, bool WV, bool WD, int D>
void launch_nt(const RowsParams& rp, int nt, int grid, hipStream_t stream)
{
  if (nt)
    hipLaunchKernelGGL((vortdiv_rows_kernel<CHECK, WV, WD, D, true>), dim3(grid), dim3(256), 0, stream, rp);
  else
    hipLaunchKernelGGL((vortdiv_rows_kernel<CHECK, WV, WD, D, false>), dim3(grid), dim3(256), 0, stream, rp);
}

template <bool CHECK, bool WV, bool WD>
void launch_d(const RowsParams& rp, const Tuning& t, int grid, hipStream_t stream)
{
  switch (t.D) {
  case 0:
    launch_nt<CHECK, WV, WD, 0>(rp, t.NT, grid, stream);
    break;
  case 1:
    launch_nt<CHECK, WV, WD, 1>(rp, t.NT, grid, stream);
    break;
  case 2:
    launch_nt<CHECK, WV, WD, 2>(rp, t.NT, grid, stream);
    break;
  default:
    launch_nt<CHECK, WV, WD, 3>(rp, t.NT, grid, stream);
    break;
  }
}

template <bool CHECK>
void launch_outputs(const RowsParams& rp, const Tuning& t, int grid, hipStream_t stream)
{
  if (rp.rv && rp.dv)
    launch_d<CHECK, true, true>(rp, t, grid, stream);
  else if (rp.rv)
    launch_d<CHECK, true, false>(rp, t, grid, stream);
  else
    launch_d<CHECK, false, true>(rp, t, grid, stream);
}

inline bool aligned16(const void* p)
{
  return (reinterpret_cast<size_t>(p) & 15u) == 0;
}

} // namespace

// Takes the request when the fast path applies (nx % 4 == 0, 16-byte aligned
// bases and strides); otherwise leaves *handled false and the caller falls
// back to the one-lane-per-cell kernel.
hipError_t launch_vortdiv_rows(const StencilParams& prm, hipStream_t stream, bool* handled)
{
  *handled = false;
  float* rv = nullptr;
  float* dv = nullptr;
  if (prm.op == ST_VORTDIV) {
    rv = prm.out0;
    dv = prm.out1;
  } else if (prm.op == ST_RELVORT) {
    rv = prm.out0;
  } else if (prm.op == ST_DIVERGENCE) {
    dv = prm.out0;
  } else {
    return hipSuccess;
  }
  if (!rv && !dv)
    return hipSuccess;
  const int nx = prm.nx;
  if (nx % 4 != 0 || nx < 8 || prm.ny_global < 3)
    return hipSuccess;
  if (!aligned16(prm.f0) || !aligned16(prm.f1) || !aligned16(prm.xmapr) || !aligned16(prm.ymapr) || (rv && !aligned16(rv)) || (dv && !aligned16(dv)))
    return hipSuccess;
  if (prm.in_level_stride % 4 != 0 || prm.out_level_stride % 4 != 0)
    return hipSuccess;
  if (std::getenv("MIFC_FORCE_CELL_KERNEL"))
    return hipSuccess;

  const Tuning t = current_tuning();
  RowsParams rp;
  rp.nx = nx;
  rp.nyg = prm.ny_global;
  rp.j0 = prm.j0;
  rp.ny_local = prm.ny_local;
  rp.lo = (prm.j0 >= 1) ? 0 : (1 - prm.j0);
  const int last = prm.ny_global - 1 - prm.j0; // local index of the global last row
  rp.hi = (prm.ny_local < last) ? prm.ny_local : last;
  if (rp.hi <= rp.lo) {
    // slab without a single computed row (can only be a 1-row edge slab): not supported here
    return hipSuccess;
  }
  rp.R = t.R;
  rp.nbands = (rp.hi - rp.lo + t.R - 1) / t.R;
  rp.nwc = (nx + 255) / 256;
  rp.nlev = prm.nlev;
  rp.chunks_per_tile = (prm.nlev + 3) / 4;
  const long n_logical = (long)rp.nbands * rp.nwc * rp.chunks_per_tile;
  if (n_logical > 0x3fffffffL)
    return hipSuccess;
  rp.n_logical = (int)n_logical;
  rp.per_xcd = (rp.n_logical + 7) / 8;
  rp.u = prm.f0;
  rp.v = prm.f1;
  rp.xm = prm.xmapr;
  rp.ym = prm.ymapr;
  rp.rv = rv;
  rp.dv = dv;
  rp.in_stride = prm.in_level_stride;
  rp.out_stride = prm.out_level_stride;
  rp.all_defined = prm.all_defined;
  rp.undef = prm.undef;
  rp.n_undefined = prm.n_undefined;
  const int grid = rp.per_xcd * 8;

  *handled = true;
  if (prm.every_level_all_defined)
    launch_outputs<false>(rp, t, grid, stream);
  else
    launch_outputs<true>(rp, t, grid, stream);
  return hipGetLastError();
}

} // namespace mifc
