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
//     and v in registers (row-sliding window): every u/v value is fetched
//     once per band, halo overhead (R+2)/R on the reads only.
//   * all global accesses are 16 B per lane, 1 KiB per wave-instruction,
//     row-major and coalesced.  The window is a STATIC register ring of
//     W = D+3 rows and the row loop is unrolled over the ring, so no loaded
//     value is ever moved between registers and the compiler can keep D rows
//     in flight with counted s_waitcnt vmcnt(N) (a rotating-variable window
//     forces vmcnt(0) every iteration: moving an in-flight register needs its
//     data).  Loads are branch-free: out-of-range lanes/rows read a clamped,
//     valid address and only the STORES are predicated.
//   * x neighbours (i-1, i+4) come from the adjacent lanes with one DPP
//     wave-shift each (v_mov_b32_dpp wave_shr:1 / wave_shl:1) -- no LDS, no
//     barrier; lane 0 and lane 63 take one scalar per row from the neighbouring
//     wave-column (one dword load per field and row, lanes 0..62 share an
//     address).
//   * the four waves of a workgroup take FOUR CONSECUTIVE LEVELS of the same
//     (band, wave-column) tile; the blockIdx -> tile map is XCD-aware (blocks
//     are dealt round-robin over the 8 XCDs, block b runs sequence number
//     (b % 8) * per_xcd + b / 8), so the levels of one tile meet in one L2 and
//     xmapr / ymapr are fetched from HBM once per tile, not once per level.
//   * undefined cells are counted per lane, reduced per wave (butterfly) and
//     added with ONE atomic per wave to n_undefined[level]; the all-defined
//     instantiation contains no test and no atomic at all.
//   * no MFMA: 4 fp32 subtractions + 16 fp64-pipe ops per cell (6 cvt, 6 mul,
//     2 add, 2 cvt back), the reference's double-promoted combine.
#include "mifc_device.h"
#include "mifc_kernels.h"

#include <cstdlib>
#include <cstring>

namespace mifc {

namespace {

struct RowsParams
{
  int nx;
  int nyg;      // rows of the whole field
  int j0;       // global row of owned row 0
  int ny_local; // owned rows
  int lo, hi;   // owned local rows that are computed: [lo, hi)
  int R;        // rows per band
  int nbands, nwc, nlev;
  int wpb;             // waves per workgroup (4, 8 or 16): that many levels side by side
  int lpw;             // levels each wave walks through one after the other
  int uL, uB, uW;      // workgroup-unit counts along (level groups, bands, wave-columns)
  int n_logical;       // uL * uB * uW
  int per_xcd;         // ceil(n_logical / 8)
  int order;           // block sequence -> unit order, see decode_block()
  int xcd_remap;       // 1: sequence = (b % 8) * per_xcd + b / 8
  long idx_lo, idx_hi; // valid flat element range relative to owned row 0 (for clamped scalar loads)
  const float *u, *v, *xm, *ym;
  float *rv, *dv;
  long in_stride, out_stride;
  const unsigned char* all_defined;
  float undef;
  u64* n_undefined;
};

typedef float v4f __attribute__((ext_vector_type(4)));

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
__device__ __forceinline__ void store4(float* p, const v4f& v)
{
  if (NT)
    __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
  else
    *reinterpret_cast<v4f*>(p) = v;
}

template <bool NT>
__device__ __forceinline__ v4f load4(const float* p)
{
  if (NT)
    return __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  return *reinterpret_cast<const v4f*>(p);
}

struct RowRegs
{
  v4f u, v;
  float eu, ev; // lane 63: value east of the wave-column; other lanes: value west of it
};

// sequence number -> workgroup unit (l, b, w) along (levels, bands, wave-columns)
//   order 1: address order -- level slowest, then band, wave-column fastest
//   order 0: (wave-column, band) tiles, the levels of one tile consecutive
//   order 2: (band, wave-column) tiles, the levels of one tile consecutive
__device__ __forceinline__ void decode_block(const RowsParams& P, int seq, int& l, int& b, int& w)
{
  if (P.order == 1) {
    const int per_level = P.uB * P.uW;
    l = seq / per_level;
    const int rem = seq - l * per_level;
    b = rem / P.uW;
    w = rem - b * P.uW;
  } else {
    const int t = seq / P.uL;
    l = seq - t * P.uL;
    if (P.order == 2) {
      b = t / P.uW;
      w = t - b * P.uW;
    } else {
      w = t / P.uB;
      b = t - w * P.uB;
    }
  }
}

template <bool CHECK, bool WANT_V, bool WANT_D, int D, bool NT, bool NTL>
__global__ __launch_bounds__(1024) void vortdiv_rows_kernel(const RowsParams P)
{
  constexpr int W = D + 3;                    // ring slots: rows r-2 (being refilled), r-1, r, r+1, r+2 .. r+D

  // ---- which tile / level does this wave own -----------------------------
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int bid = blockIdx.x;
  const int seq = P.xcd_remap ? ((bid & 7) * P.per_xcd + (bid >> 3)) : bid;
  if (seq >= P.n_logical)
    return; // whole workgroup: nobody reaches the barrier below
  int lgroup, band, wc;
  decode_block(P, seq, lgroup, band, wc);
  // this workgroup: levels [lev0, lev0 + wpb*lpw); wave w takes lev0 + w, lev0 + w + wpb, ...
  const int lev0 = lgroup * (P.wpb * P.lpw);

  const int nx = P.nx;
  const int c0 = wc * 256 + lane * 4;
  const bool active = c0 < nx;
  const int c0c = active ? c0 : nx - 4; // clamped column for loads
  int east_col = wc * 256 + 256;        // first column east of this wave-column (may be nx: wraps to the next row)
  if (east_col > nx)
    east_col = nx;
  const bool take_east_scalar = (c0 + 4 >= east_col); // my east neighbour is outside the wave-column
  const int edge_col = (lane == 63) ? east_col : (wc * 256 - 1);

  const int jb = P.lo + band * P.R;                     // first local row of the band
  const int nr = (P.hi - jb < P.R) ? (P.hi - jb) : P.R; // rows in this band
  const float undef = P.undef;
  const bool owns_top_edge = (P.j0 == 0);
  const bool owns_bottom_edge = (P.j0 + P.ny_local == P.nyg);

  // ---- map factors of the tile: HBM/L2 -> LDS once per workgroup -------------
  // xmapr/ymapr do not depend on the level; every wave of the workgroup and
  // every level it walks through re-reads them from LDS (2 KiB per tile row).
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  v4f* lds_xm = reinterpret_cast<v4f*>(lds_raw);
  v4f* lds_ym = lds_xm + P.R * 64;
  for (int i = threadIdx.x; i < nr * 64; i += blockDim.x) {
    const int row = i >> 6;
    int col = wc * 256 + (i & 63) * 4;
    col = col < nx ? col : nx - 4;
    const long o = (long)(jb + row) * nx + col;
    lds_xm[i] = load4<false>(P.xm + o);
    lds_ym[i] = load4<false>(P.ym + o);
  }
  __syncthreads();

  for (int li = 0; li < P.lpw; ++li) {
  const int lev = lev0 + li * P.wpb + wave;
  if (lev >= P.nlev)
    break;
  const bool all = CHECK ? (P.all_defined && P.all_defined[lev] != 0) : true;

  const float* __restrict__ u = P.u + (size_t)lev * P.in_stride;
  const float* __restrict__ v = P.v + (size_t)lev * P.in_stride;
  float* rv = WANT_V ? P.rv + (size_t)lev * P.out_stride : nullptr;
  float* dv = WANT_D ? P.dv + (size_t)lev * P.out_stride : nullptr;

  // band-relative row rho in [-1, nr]; rows past the south halo are clamped to it
  auto load_row = [&](int rho) -> RowRegs {
    const int rc = rho > nr ? nr : rho;
    const long base = (long)(jb + rc) * nx;
    RowRegs r;
    r.u = load4<NTL>(u + base + c0c);
    r.v = load4<NTL>(v + base + c0c);
    // x-neighbour scalar of the wave-column edge.  Only centre rows use it; the
    // clamp keeps the address inside the buffer for the rows that do not.
    long e = base + edge_col;
    e = e < P.idx_lo ? P.idx_lo : (e > P.idx_hi ? P.idx_hi : e);
    r.eu = u[e];
    r.ev = v[e];
    return r;
  };
  // ---- prologue: rows -1 .. D into ring slots (rho + 2) % W ----------------
  RowRegs ring[W];
#pragma unroll
  for (int rho = -1; rho <= D; ++rho)
    ring[(rho + 2) % W] = load_row(rho);

  unsigned int bad = 0;

  for (int rb = 0; rb < nr; rb += W) {
#pragma unroll
    for (int s = 0; s < W; ++s) {
      const int r = rb + s;
      if (r >= nr)
        goto level_done;
      // row r+1+D replaces row r-2, which nobody needs any more
      ring[s % W] = load_row(r + 1 + D);

      const RowRegs& rp = ring[(s + 1) % W]; // row r-1
      const RowRegs& rc = ring[(s + 2) % W]; // row r
      const RowRegs& rn = ring[(s + 3) % W]; // row r+1
      const v4f xm4 = lds_xm[r * 64 + lane], ym4 = lds_ym[r * 64 + lane];

      // ---- x neighbours of the centre row from the adjacent lanes ----------
      const float east_u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rc.eu), 63));
      const float east_v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rc.ev), 63));
      float uW = dpp_from_lower_lane(rc.eu, rc.u.w); // lane 0 keeps its own west scalar
      float vW = dpp_from_lower_lane(rc.ev, rc.v.w);
      float uE = dpp_from_upper_lane(rc.eu, rc.u.x); // lane 63 keeps its own east scalar
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
      const float xm[4] = {xm4.x, xm4.y, xm4.z, xm4.w};
      const float ym[4] = {ym4.x, ym4.y, ym4.z, ym4.w};
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
        const int jl = jb + r;
        const long o = (long)jl * nx + c0;
        const int j = P.j0 + jl;
        if (WANT_V) {
          v4f z4;
          z4.x = zv[0];
          z4.y = zv[1];
          z4.z = zv[2];
          z4.w = zv[3];
          store4<NT>(rv + o, z4);
          if (j == 1 && owns_top_edge) // row part of fillEdges (:70-73)
            store4<NT>(rv + o - nx, z4);
          if (j == P.nyg - 2 && owns_bottom_edge)
            store4<NT>(rv + o + nx, z4);
        }
        if (WANT_D) {
          v4f d4;
          d4.x = zd[0];
          d4.y = zd[1];
          d4.z = zd[2];
          d4.w = zd[3];
          store4<NT>(dv + o, d4);
          if (j == 1 && owns_top_edge)
            store4<NT>(dv + o - nx, d4);
          if (j == P.nyg - 2 && owns_bottom_edge)
            store4<NT>(dv + o + nx, d4);
        }
      }
    }
  }
level_done:
  if (CHECK && P.n_undefined)
    wave_count_add(P.n_undefined + lev, bad);
  } // levels of this wave
}

struct Tuning
{
  int R;     // rows per band
  int D;     // rows kept in flight beyond the 3-row window (0..2)
  int NT;    // nontemporal stores
  int NTL;   // nontemporal loads of u, v
  int ORDER; // block order, see decode_block()
  int XCD;   // XCD-aware blockIdx remap
  int WPB;   // waves per workgroup: 4, 8 or 16 (levels side by side)
  int LPW;   // levels each wave walks through one after the other
};

int tune_value(const char* s, const char* key, int dflt)
{
  // finds "KEY=" at the start of the string or after a comma
  const size_t n = std::strlen(key);
  for (const char* p = s; p && *p;) {
    if (std::strncmp(p, key, n) == 0 && p[n] == '=')
      return std::atoi(p + n + 1);
    p = std::strchr(p, ',');
    if (p)
      ++p;
  }
  return dflt;
}

Tuning current_tuning()
{
  Tuning t = {8, 1, 1, 0, 1, 1, 4, 1};
  // MIFC_VORTDIV_TUNE="R=8,D=1,NT=1,NTL=0,ORDER=1,XCD=1,WPB=4,LPW=1" -- used by the sweep tool and the tests
  if (const char* s = std::getenv("MIFC_VORTDIV_TUNE")) {
    t.R = tune_value(s, "R", t.R);
    t.D = tune_value(s, "D", t.D);
    t.NT = tune_value(s, "NT", t.NT);
    t.NTL = tune_value(s, "NTL", t.NTL);
    t.ORDER = tune_value(s, "ORDER", t.ORDER);
    t.XCD = tune_value(s, "XCD", t.XCD);
    t.WPB = tune_value(s, "WPB", t.WPB);
    t.LPW = tune_value(s, "LPW", t.LPW);
  }
  if (t.WPB != 4 && t.WPB != 8 && t.WPB != 16)
    t.WPB = 4;
  if (t.LPW < 1)
    t.LPW = 1;
  if (t.R < 1)
    t.R = 1;
  if (t.R > 32)
    t.R = 32; // map-factor tile in LDS: 2 KiB per row, 64 KiB at most
  if (t.D < 0)
    t.D = 0;
  if (t.D > 2)
    t.D = 2;
  return t;
}

template <bool CHECK, bool WV, bool WD, int D>
void launch_nt(const RowsParams& rp, const Tuning& t, int grid, hipStream_t stream)
{
  const int sel = (t.NT ? 1 : 0) | (t.NTL ? 2 : 0);
  switch (sel) {
  case 0:
    hipLaunchKernelGGL((vortdiv_rows_kernel<CHECK, WV, WD, D, false, false>), dim3(grid), dim3(64 * t.WPB), (size_t)rp.R * 2048, stream, rp);
    break;
  case 1:
    hipLaunchKernelGGL((vortdiv_rows_kernel<CHECK, WV, WD, D, true, false>), dim3(grid), dim3(64 * t.WPB), (size_t)rp.R * 2048, stream, rp);
    break;
  case 2:
    hipLaunchKernelGGL((vortdiv_rows_kernel<CHECK, WV, WD, D, false, true>), dim3(grid), dim3(64 * t.WPB), (size_t)rp.R * 2048, stream, rp);
    break;
  default:
    hipLaunchKernelGGL((vortdiv_rows_kernel<CHECK, WV, WD, D, true, true>), dim3(grid), dim3(64 * t.WPB), (size_t)rp.R * 2048, stream, rp);
    break;
  }
}

template <bool CHECK, bool WV, bool WD>
void launch_d(const RowsParams& rp, const Tuning& t, int grid, hipStream_t stream)
{
  switch (t.D) {
  case 0:
    launch_nt<CHECK, WV, WD, 0>(rp, t, grid, stream);
    break;
  case 1:
    launch_nt<CHECK, WV, WD, 1>(rp, t, grid, stream);
    break;
  default:
    launch_nt<CHECK, WV, WD, 2>(rp, t, grid, stream);
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
  // A row slab carries one halo row before owned row 0 and one after the last
  // owned row; a whole field has neither.
  const bool has_north_halo = prm.j0 > 0;
  const bool has_south_halo = prm.j0 + prm.ny_local < prm.ny_global;
  rp.idx_lo = has_north_halo ? -(long)nx : 0;
  rp.idx_hi = (long)nx * (prm.ny_local + (has_south_halo ? 1 : 0)) - 1;
  rp.R = t.R;
  rp.nbands = (rp.hi - rp.lo + t.R - 1) / t.R;
  rp.nwc = (nx + 255) / 256;
  rp.nlev = prm.nlev;
  rp.wpb = t.WPB;
  rp.lpw = t.LPW;
  rp.uL = (prm.nlev + t.WPB * t.LPW - 1) / (t.WPB * t.LPW);
  rp.uB = rp.nbands;
  rp.uW = rp.nwc;
  const long n_logical = (long)rp.uL * rp.uB * rp.uW;
  if (n_logical > 0x3fffffffL)
    return hipSuccess;
  rp.n_logical = (int)n_logical;
  rp.per_xcd = (rp.n_logical + 7) / 8;
  rp.order = t.ORDER;
  rp.xcd_remap = t.XCD;
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
