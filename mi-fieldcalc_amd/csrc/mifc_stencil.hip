// mifc_stencil.hip -- horizontal 5-point-stencil kernels for gfx950.
//
// Reference semantics restated once:
//   * the operators run ONE flat loop over i in [nx, nx*ny-nx) (gradient
//     compute 1: [1, nx*ny-1)), so the left/right edge columns are computed
//     with neighbours that wrap into the adjacent row; those cells take part
//     in the undefined COUNT, then fillEdges (FieldCalculations.cc:59-74)
//     overwrites them: columns first, then rows 0 / ny-1 including corners.
//     Net effect on the values: final(j,i) = raw(clamp(j,1,ny-2), clamp(i,1,nx-2)).
//   * the count is taken over the raw loop range, before the edge fill.
//   * an undefined cell gets `undef`; with an ALL_DEFINED input flag no test
//     runs at all.
//
// Two kernel families live here:
//   stencil_cell_kernel   one lane per cell, any nx, every operator; used for
//                         ragged widths, row slabs and the less common operators.
//   (the row-walking kernels that take over whenever nx % 4 == 0 are in
//    mifc_vortdiv.hip -- wind operators -- and mifc_stencil_rows.hip)
#include <cstdlib>

#include "mifc_device.h"
#include "mifc_kernels.h"

namespace mifc {

namespace {

struct CellOut
{
  float o0, o1;
};

// Raw value(s) at flat local index p (relative to owned row 0; may reach into
// the halo rows of a slab).  Returns false if the cell is undefined.
template <int OP, bool CHECK>
__device__ __forceinline__ bool stencil_raw(const StencilParams& P, const float* __restrict__ f0, const float* __restrict__ f1, long p, bool all,
                                            CellOut& o)
{
  const int nx = P.nx;
  const float undef = P.undef;
  if (OP == ST_RELVORT || OP == ST_ABSVORT || OP == ST_DIVERGENCE || OP == ST_VORTDIV) {
    const float* u = f0;
    const float* v = f1;
    const float vw = v[p - 1], ve = v[p + 1], us = u[p - nx], un = u[p + nx];
    if (CHECK && !(all || (is_def(vw, undef) && is_def(ve, undef) && is_def(us, undef) && is_def(un, undef)))) // :1861 == :1895 == :1927
      return false;
    const float xm = P.xmapr[p], ym = P.ymapr[p];
    if (OP == ST_RELVORT) {
      o.o0 = f_relvort(xm, ym, ve - vw, un - us);
    } else if (OP == ST_ABSVORT) {
      o.o0 = f_absvort(xm, ym, ve - vw, un - us, P.fcoriolis[p]);
    } else {
      const float uw = u[p - 1], ue = u[p + 1], vs = v[p - nx], vn = v[p + nx];
      if (OP == ST_DIVERGENCE) {
        o.o0 = f_diverg(xm, ym, ue - uw, vn - vs);
      } else {
        o.o0 = f_relvort(xm, ym, ve - vw, un - us);
        o.o1 = f_diverg(xm, ym, ue - uw, vn - vs);
      }
    }
    return true;
  }
  if (OP == ST_ADVECTION) { // :1971-1972
    const float* f = f0;
    const float uc = f1[p], vc = P.f2[p + (size_t)blockIdx.y * P.in_level_stride]; // blockIdx.y = level of this launch
    const float s = f[p - nx], w = f[p - 1], e = f[p + 1], n = f[p + nx];
    if (CHECK && !(all || (is_def(uc, undef) && is_def(vc, undef) && is_def(s, undef) && is_def(w, undef) && is_def(e, undef) && is_def(n, undef))))
      return false;
    o.o0 = (float)(((double)uc * 0.5 * (double)P.xmapr[p] * (double)(e - w) + (double)vc * 0.5 * (double)P.ymapr[p] * (double)(n - s)) * (double)P.scale);
    return true;
  }
  if (OP == ST_JACOBIAN) { // :2443-2451: four float-rounded partials, float combination
    const float* a = f0;
    const float* b = f1;
    const float as = a[p - nx], aw = a[p - 1], ae = a[p + 1], an = a[p + nx];
    const float bs = b[p - nx], bw = b[p - 1], be = b[p + 1], bn = b[p + nx];
    if (CHECK && !(all || (is_def(as, undef) && is_def(aw, undef) && is_def(ae, undef) && is_def(an, undef) && is_def(bs, undef) && is_def(bw, undef) &&
                           is_def(be, undef) && is_def(bn, undef))))
      return false;
    const float xm = P.xmapr[p], ym = P.ymapr[p];
    const float df1dx = half_prod(xm, ae - aw);
    const float df1dy = half_prod(ym, an - as);
    const float df2dx = half_prod(xm, be - bw);
    const float df2dy = half_prod(ym, bn - bs);
    o.o0 = df1dx * df2dy - df1dy * df2dx;
    return true;
  }
  if (OP == ST_TFP) { // :2290-2298; the "!= 0" test runs even when the flag says ALL_DEFINED
    const float* t = f0;
    const float* g = f1; // |grad t| with its edges filled
    const float ts = t[p - nx], tw = t[p - 1], te = t[p + 1], tn = t[p + nx];
    const float gs = g[p - nx], gw = g[p - 1], gc = g[p], ge = g[p + 1], gn = g[p + nx];
    const bool def = all || (is_def(ts, undef) && is_def(tw, undef) && is_def(te, undef) && is_def(tn, undef) && is_def(gs, undef) && is_def(gw, undef) &&
                             is_def(gc, undef) && is_def(ge, undef) && is_def(gn, undef));
    if (!(def && gc != 0))
      return false;
    const double hx = 0.5 * (double)P.xmapr[p], hy = 0.5 * (double)P.ymapr[p];
    const float dabsdeltdx = half_prod(P.xmapr[p], ge - gw);
    const float dabsdeltdy = half_prod(P.ymapr[p], gn - gs);
    const float dtdxa = (float)(hx * (double)(te - tw) / (double)gc);
    const float dtdya = (float)(hy * (double)(tn - ts) / (double)gc);
    o.o0 = -(dabsdeltdx * dtdxa + dabsdeltdy * dtdya);
    return true;
  }
  if (OP == ST_QVEC_X || OP == ST_QVEC_Y) { // :570-584; "!= undef" only (no NaN test), whatever the input flag says
    const float* ug = f0;
    const float* vg = f1;
    const float* t = P.f2 + (size_t)blockIdx.y * P.in_level_stride;
    const float us = ug[p - nx], uw = ug[p - 1], ue = ug[p + 1], un = ug[p + nx];
    const float vs = vg[p - nx], vw = vg[p - 1], ve = vg[p + 1], vn = vg[p + nx];
    const float ts = t[p - nx], tw = t[p - 1], te = t[p + 1], tn = t[p + nx];
    if (!(us != undef && uw != undef && ue != undef && un != undef && vs != undef && vw != undef && ve != undef && vn != undef && ts != undef &&
          tw != undef && te != undef && tn != undef))
      return false;
    const double hx = 0.5 * (double)P.xmapr[p], hy = 0.5 * (double)P.ymapr[p];
    const float scale = P.scale_lev ? P.scale_lev[blockIdx.y] : P.scale; // level batch: the pressure differs per level
    const float scale2 = P.scale2_lev ? P.scale2_lev[blockIdx.y] : P.scale2;
    const float dtdx = (float)(hx * (double)scale * (double)(te - tw));
    const float dtdy = (float)(hy * (double)scale * (double)(tn - ts));
    if (OP == ST_QVEC_X) {
      const float dugdx = half_prod(P.xmapr[p], ue - uw);
      const float dvgdx = half_prod(P.xmapr[p], ve - vw);
      o.o0 = scale2 * (dugdx * dtdx + dvgdx * dtdy);
    } else {
      const float dugdy = half_prod(P.ymapr[p], un - us);
      const float dvgdy = half_prod(P.ymapr[p], vn - vs);
      o.o0 = scale2 * (dugdy * dtdx + dvgdy * dtdy);
    }
    return true;
  }
  const float* f = f0;
  if (OP == ST_GRAD_X) { // :2015-2016
    const float w = f[p - 1], e = f[p + 1];
    if (CHECK && !(all || (is_def(w, undef) && is_def(e, undef))))
      return false;
    o.o0 = half_prod(P.xmapr[p], e - w);
    return true;
  }
  if (OP == ST_GRAD_Y) { // :2027-2028
    const float s = f[p - nx], n = f[p + nx];
    if (CHECK && !(all || (is_def(s, undef) && is_def(n, undef))))
      return false;
    o.o0 = half_prod(P.ymapr[p], n - s);
    return true;
  }
  const float s = f[p - nx], w = f[p - 1], e = f[p + 1], n = f[p + nx];
  if (OP == ST_GRAD_LAP || OP == ST_GVORT) {
    const float c = f[p];
    if (CHECK && !(all || (is_def(s, undef) && is_def(w, undef) && is_def(c, undef) && is_def(e, undef) && is_def(n, undef)))) // :2053, :729
      return false;
    const double xm = P.xmapr[p], ym = P.ymapr[p];
    if (OP == ST_GRAD_LAP) { // :2054-2056, second differences rounded to float first
      const float d2x = (float)((double)w - 2.0 * (double)c + (double)e);
      const float d2y = (float)((double)s - 2.0 * (double)c + (double)n);
      o.o0 = (float)(4.0 * (0.25 * xm * xm * (double)d2x + 0.25 * ym * ym * (double)d2y));
    } else { // :730-731, second differences stay double
      const float g4 = (float)((double)MIFC_K_G * 4.);
      const double d2x = (double)w - 2. * (double)c + (double)e;
      const double d2y = (double)s - 2. * (double)c + (double)n;
      o.o0 = (float)((0.25 * xm * xm * d2x + 0.25 * ym * ym * d2y) * (double)g4 / (double)P.fcoriolis[p]);
    }
    return true;
  }
  if (CHECK && !(all || (is_def(s, undef) && is_def(w, undef) && is_def(e, undef) && is_def(n, undef)))) // :2039, :660, :693, :1534
    return false;
  if (OP == ST_GRAD_ABS) { // :2040-2042
    const float dfdx = half_prod(P.xmapr[p], e - w);
    const float dfdy = half_prod(P.ymapr[p], n - s);
    o.o0 = absval(dfdx, dfdy);
  } else if (OP == ST_GWIND_X) { // :661
    o.o0 = (float)(-0.5 * (double)P.ymapr[p] * (double)(n - s) * (double)MIFC_K_G / (double)P.fcoriolis[p]);
  } else if (OP == ST_GWIND_Y) { // :694
    o.o0 = (float)(0.5 * (double)P.xmapr[p] * (double)(e - w) * (double)MIFC_K_G / (double)P.fcoriolis[p]);
  } else { // ST_IGWIND :1535-1536
    const double fc = P.fcoriolis[p];
    o.o0 = (float)(-0.5 * (double)P.ymapr[p] * (double)(n - s) / fc);
    o.o1 = (float)(0.5 * (double)P.xmapr[p] * (double)(e - w) / fc);
  }
  return true;
}

// The one-input operators of stencil_raw() from VALUES (the flat four-cells-per-lane kernel has them in registers); the
// same expressions, the same tests.  ST_GRAD_X is not here: its count runs over another range (cell_one).
template <int OP, bool CHECK>
__device__ __forceinline__ bool scalar_from_values(bool all, float undef, float s, float w, float c, float e, float n, float xmf, float ymf, float fcf,
                                                   CellOut& o)
{
  // straight-line code (all_def()): the formula runs unconditionally, the caller's select discards what undefined inputs made of it
  if (OP == ST_GRAD_Y) { // :2027-2028
    o.o0 = half_prod(ymf, n - s);
    return !CHECK || (all | all_def(undef, s, n));
  }
  if (OP == ST_GRAD_LAP || OP == ST_GVORT) {
    const double xm = xmf, ym = ymf;
    if (OP == ST_GRAD_LAP) { // :2054-2056, second differences rounded to float first
      const float d2x = (float)((double)w - 2.0 * (double)c + (double)e);
      const float d2y = (float)((double)s - 2.0 * (double)c + (double)n);
      o.o0 = (float)(4.0 * (0.25 * xm * xm * (double)d2x + 0.25 * ym * ym * (double)d2y));
    } else { // :730-731, second differences stay double
      const float g4 = (float)((double)MIFC_K_G * 4.);
      const double d2x = (double)w - 2. * (double)c + (double)e;
      const double d2y = (double)s - 2. * (double)c + (double)n;
      const double fd = (double)fcf; // the division through the refined reciprocal, as in mifc_stencil_rows.hip
      o.o0 = (float)quotient((0.25 * xm * xm * d2x + 0.25 * ym * ym * d2y) * (double)g4, fd, shared_reciprocal(fd));
    }
    return !CHECK || (all | all_def(undef, s, w, c, e, n)); // :2053, :729
  }
  if (OP == ST_GRAD_ABS) { // :2040-2042
    const float dfdx = half_prod(xmf, e - w);
    const float dfdy = half_prod(ymf, n - s);
    o.o0 = absval(dfdx, dfdy);
  } else if (OP == ST_GWIND_X) { // :661
    const double fd = (double)fcf;
    o.o0 = (float)quotient(-0.5 * (double)ymf * (double)(n - s) * (double)MIFC_K_G, fd, shared_reciprocal(fd));
  } else if (OP == ST_GWIND_Y) { // :694
    const double fd = (double)fcf;
    o.o0 = (float)quotient(0.5 * (double)xmf * (double)(e - w) * (double)MIFC_K_G, fd, shared_reciprocal(fd));
  } else { // ST_IGWIND :1535-1536: two quotients, one reciprocal
    const double fd = (double)fcf, finv = shared_reciprocal(fd);
    o.o0 = (float)quotient(-0.5 * (double)ymf * (double)(n - s), fd, finv);
    o.o1 = (float)quotient(0.5 * (double)xmf * (double)(e - w), fd, finv);
  }
  return !CHECK || (all | all_def(undef, s, w, e, n)); // :2039, :660, :693, :1534
}

// Final value(s) of owned cell i (flat local index) after fillEdges, and its contribution to the undefined count.
template <int OP, bool CHECK>
__device__ __forceinline__ void cell_one(const StencilParams& P, const float* __restrict__ f0, const float* __restrict__ f1, long i, bool all, int jmax,
                                         CellOut& o, unsigned int& bad)
{
  const int nx = P.nx;
  const int jl = (int)(i / nx);
  const int c = (int)(i - (long)jl * nx);
  const int j = P.j0 + jl;
  // where the value of this cell comes from after fillEdges
  const int jj = j < 1 ? 1 : (j > jmax ? jmax : j);
  const int cc = c < 1 ? 1 : (c > nx - 2 ? nx - 2 : c);
  const long p = (long)(jj - P.j0) * nx + cc;
  o.o0 = P.undef;
  o.o1 = P.undef;
  const bool ok = stencil_raw<OP, CHECK>(P, f0, f1, p, all, o);
  if (!ok) {
    o.o0 = P.undef;
    o.o1 = P.undef;
  }
  // TFP rejects |grad T| == 0 cells, the Q-vector pass tests its inputs, whatever the input flag says
  if (CHECK && (!all || OP == ST_TFP || OP == ST_QVEC_X || OP == ST_QVEC_Y)) {
    // Count over the raw loop range, with the wrapped neighbours the flat
    // loop sees at the edge columns (Appendix A #6).
    const long gi = (long)j * nx + c;
    const long ng = (long)nx * P.ny_global;
    const bool in_range = (OP == ST_GRAD_X) ? (gi >= 1 && gi < ng - 1) : (j >= 1 && j <= jmax);
    if (in_range) {
      if (p == i) {
        bad += ok ? 0u : 1u;
      } else {
        CellOut dummy;
        bad += stencil_raw<OP, CHECK>(P, f0, f1, i, all, dummy) ? 0u : 1u;
      }
    }
  }
}

// One lane per owned cell; blockIdx.y = level.
template <int OP, bool CHECK>
__global__ __launch_bounds__(256) void stencil_cell_kernel(const StencilParams P)
{
  const int lev = blockIdx.y;
  const int nx = P.nx;
  const long n_local = (long)nx * P.ny_local;
  const float* __restrict__ f0 = P.f0 + (size_t)lev * P.in_level_stride;
  const float* __restrict__ f1 = P.f1 ? P.f1 + (size_t)lev * P.in_level_stride : nullptr;
  float* out0 = P.out0 + (size_t)lev * P.out_level_stride;
  float* out1 = P.out1 ? P.out1 + (size_t)lev * P.out_level_stride : nullptr;
  const bool all = CHECK ? (P.all_defined && P.all_defined[lev] != 0) : true;
  const int jmax = P.ny_global - 2; // last computed global row
  unsigned int bad = 0;

  const long cell_begin = (P.row_end > P.row_begin) ? (long)P.row_begin * nx : 0;
  const long cell_end = (P.row_end > P.row_begin) ? (long)P.row_end * nx : n_local;
  for (long i = cell_begin + (long)blockIdx.x * blockDim.x + threadIdx.x; i < cell_end; i += (long)gridDim.x * blockDim.x) {
    CellOut o;
    cell_one<OP, CHECK>(P, f0, f1, i, all, jmax, o, bad);
    out0[i] = o.o0;
    if ((OP == ST_VORTDIV || OP == ST_IGWIND) && out1)
      out1[i] = o.o1;
  }
  if (CHECK)
    block_count_add(P.n_undefined ? P.n_undefined + lev : nullptr, bad); // one atomic per workgroup
}

// ---------------------------------------------------------------------------
// Wind operators on a width that is NOT a multiple of 4 (949 x 1069, say): the row-walking / level-walking kernels
// want 16-byte aligned rows, the one-lane-per-cell kernel above moves one dword per lane and instruction and divides
// per cell (39 % of 8 TB/s for the fused pair where the aligned kernels reach 60 % on a grid of that size).  The
// reference's loop is FLAT (i - 1, i + 1, i - nx, i + nx: section header), so this kernel is too: a lane takes four
// consecutive flat cells, 16 bytes per load and store at dword alignment (the hardware takes them), the rows above and
// below as the same loads nx cells earlier / later, x-neighbours from the adjacent lanes (DPP) -- consecutive lanes
// hold consecutive cells whatever the width.  What is not a plain interior cell goes through cell_one(): the first
// and last column of every row and rows 0 / ny-1 (their values come from a clamped position), the lanes whose group
// reaches over the first or last loadable row, and the last, partial wave of a level.
typedef float v4f_s __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) U4
{
  v4f_s v;
};
__device__ __forceinline__ v4f_s ldu4(const float* p)
{
  return reinterpret_cast<const U4*>(p)->v;
}
__device__ __forceinline__ void stu4(float* p, const v4f_s v)
{
  U4 t;
  t.v = v;
  *reinterpret_cast<U4*>(p) = t;
}
__device__ __forceinline__ float flat_from_lower_lane(float keep_if_none, float x)
{
  return __builtin_bit_cast(float,
                            __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep_if_none), __builtin_bit_cast(int, x), 0x138 /*wave_shr:1*/, 0xf, 0xf, false));
}
__device__ __forceinline__ float flat_from_upper_lane(float keep_if_none, float x)
{
  return __builtin_bit_cast(float,
                            __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep_if_none), __builtin_bit_cast(int, x), 0x130 /*wave_shl:1*/, 0xf, 0xf, false));
}

// Units are (level, 1024 consecutive cells); the blocks of the 1-D launch are dealt round-robin to the 8 XCDs, so unit
// seq = (b % 8) * per_xcd + b / 8 puts CONSECUTIVE units on one XCD: the rows above and below a unit's cells belong to the
// units next to it and are then served by that XCD's L2 instead of crossing the fabric three times.
template <int OP, bool CHECK> // the wind operators and the one-input operators but ST_GRAD_X
__global__ __launch_bounds__(256) void wind_flat4_kernel(const StencilParams P, const int blocks_per_level, const int n_units, const int per_xcd)
{
  constexpr bool WIND = OP == ST_RELVORT || OP == ST_ABSVORT || OP == ST_DIVERGENCE || OP == ST_VORTDIV || OP == ST_JACOBIAN; // two input fields
  constexpr bool WANT_D = OP == ST_DIVERGENCE || OP == ST_VORTDIV || OP == ST_JACOBIAN;                                        // rows above / below of the second field too
  constexpr bool USE_FC = OP == ST_ABSVORT || OP == ST_GWIND_X || OP == ST_GWIND_Y || OP == ST_GVORT || OP == ST_IGWIND;
  constexpr bool TWO_OUT = OP == ST_VORTDIV || OP == ST_IGWIND;
  const int seq = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  if (seq >= n_units)
    return;
  const int lev = seq / blocks_per_level;
  const int blk = seq - lev * blocks_per_level;
  const int nx = P.nx;
  const int n_local = nx * P.ny_local; // fits 32 bits: the launcher checks
  const float* __restrict__ u = P.f0 + (size_t)lev * P.in_level_stride;                    // the one-input operators: the field
  const float* __restrict__ v = WIND ? P.f1 + (size_t)lev * P.in_level_stride : u;
  float* out0 = P.out0 + (size_t)lev * P.out_level_stride;
  float* out1 = P.out1 ? P.out1 + (size_t)lev * P.out_level_stride : nullptr;
  const bool all = CHECK ? (P.all_defined && P.all_defined[lev] != 0) : true;
  const int jmax = P.ny_global - 2;
  const float undef = P.undef;
  unsigned int bad = 0;
  const int cell_begin = (P.row_end > P.row_begin) ? P.row_begin * nx : 0;
  const int cell_end = (P.row_end > P.row_begin) ? P.row_end * nx : n_local;
  // loadable flat range of the level: a slab has a halo row above / below its owned rows
  const int lo_idx = (P.j0 > 0) ? -nx : 0;
  const int hi_idx = n_local + ((P.j0 + P.ny_local < P.ny_global) ? nx : 0) - 1;
  const int lane = threadIdx.x & 63;
  // all lanes of a wave stay together (the x-neighbours come from the adjacent lanes)
  const int w0 = cell_begin + 4 * (blk * 256 + (int)(threadIdx.x & ~63u));
  if (w0 < cell_end) {
    const int i0 = w0 + 4 * lane;
    const bool whole_wave = w0 + 4 * 64 <= cell_end; // wave-uniform
    // rows above / below as whole groups: only if the group does not reach over the loadable range
    const bool vec = whole_wave && i0 - nx >= lo_idx && i0 + nx + 3 <= hi_idx;
    CellOut o[4];
    bool done[4] = {false, false, false, false};
    unsigned int nocount = 0;
    if (whole_wave) {
      const int is = vec ? i0 - nx : i0, in = vec ? i0 + nx : i0;
      const v4f_s uc = ldu4(u + i0);
      const v4f_s vc = WIND ? ldu4(v + i0) : uc;
      const v4f_s us = ldu4(u + is), un = ldu4(u + in);
      v4f_s vs = vc, vn = vc;
      if (WANT_D) {
        vs = ldu4(v + is);
        vn = ldu4(v + in);
      }
      const v4f_s xm4 = ldu4(P.xmapr + i0), ym4 = ldu4(P.ymapr + i0);
      v4f_s fc4 = xm4;
      if (USE_FC)
        fc4 = ldu4(P.fcoriolis + i0);
      // the cell before the wave's first and behind its last: lanes 0 and 63 keep what they load themselves
      int e = (lane == 63) ? i0 + 4 : i0 - 1;
      e = e < lo_idx ? lo_idx : (e > hi_idx ? hi_idx : e);
      const float eu = u[e];
      const float ev = WIND ? v[e] : eu;
      const float uW = flat_from_lower_lane(eu, uc.w), vW = flat_from_lower_lane(ev, vc.w);
      const float uE = flat_from_upper_lane(eu, uc.x), vE = flat_from_upper_lane(ev, vc.x);
      const float ucx[6] = {uW, uc.x, uc.y, uc.z, uc.w, uE};
      const float vcx[6] = {vW, vc.x, vc.y, vc.z, vc.w, vE};
      const int jl0 = i0 / nx; // one division per group
      const int c0 = i0 - jl0 * nx;
      // raw values of the flat loop for every cell of a computed row -- the first and last column included: with their
      // wrapped neighbours they are what the reference COUNTS there (Appendix A #6)
      CellOut raw[4];
      bool row_ok[4];
      int col[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int c = c0 + k, jl = jl0;
        if (c >= nx) {
          c -= nx;
          jl += 1;
        }
        const int j = P.j0 + jl;
        col[k] = c;
        row_ok[k] = vec && j >= 1 && j <= jmax;
        const float vw = vcx[k], ve = vcx[k + 2], uw = ucx[k], ue = ucx[k + 2];
        bool ok = true;
        raw[k].o0 = undef;
        raw[k].o1 = undef;
        // straight-line code (all_def()): formulas unconditional, selects behind them
        if (OP == ST_JACOBIAN) { // :2443-2451: all eight neighbours tested, four float-rounded partials, float combination
          if (CHECK)
            ok = all | all_def(undef, us[k], uw, ue, un[k], vs[k], vw, ve, vn[k]);
          const float df1dx = half_prod(xm4[k], ue - uw);
          const float df1dy = half_prod(ym4[k], un[k] - us[k]);
          const float df2dx = half_prod(xm4[k], ve - vw);
          const float df2dy = half_prod(ym4[k], vn[k] - vs[k]);
          raw[k].o0 = pick(ok, df1dx * df2dy - df1dy * df2dx, undef);
        } else if (WIND) {
          if (CHECK) // :1861 == :1895 == :1927
            ok = all | all_def(undef, vw, ve, us[k], un[k]);
          if (OP == ST_RELVORT)
            raw[k].o0 = pick(ok, f_relvort(xm4[k], ym4[k], ve - vw, un[k] - us[k]), undef);
          else if (OP == ST_ABSVORT)
            raw[k].o0 = pick(ok, f_absvort(xm4[k], ym4[k], ve - vw, un[k] - us[k], fc4[k]), undef);
          else if (OP == ST_DIVERGENCE)
            raw[k].o0 = pick(ok, f_diverg(xm4[k], ym4[k], ue - uw, vn[k] - vs[k]), undef);
          else {
            raw[k].o0 = pick(ok, f_relvort(xm4[k], ym4[k], ve - vw, un[k] - us[k]), undef);
            raw[k].o1 = pick(ok, f_diverg(xm4[k], ym4[k], ue - uw, vn[k] - vs[k]), undef);
          }
        } else {
          ok = scalar_from_values<OP, CHECK>(all, undef, us[k], uw, ucx[k + 1], ue, un[k], xm4[k], ym4[k], fc4[k], raw[k]);
          raw[k].o0 = pick(ok, raw[k].o0, undef);
          raw[k].o1 = pick(ok, raw[k].o1, undef);
        }
        if (CHECK)
          bad += (!all & !ok & row_ok[k]) ? 1u : 0u;
      }
      // fillEdges, column part: column 0 takes the raw value of column 1 (the next flat cell), column nx-1 that of
      // column nx-2 (the previous one) -- from this lane's group or from the adjacent lane's
      const float nx0 = flat_from_upper_lane(0.f, raw[0].o0), nx1 = flat_from_upper_lane(0.f, raw[0].o1);
      const float pv0 = flat_from_lower_lane(0.f, raw[3].o0), pv1 = flat_from_lower_lane(0.f, raw[3].o1);
      const int vec_i = vec ? 1 : 0;
      const bool next_vec = __builtin_amdgcn_update_dpp(0, vec_i, 0x130 /*wave_shl:1*/, 0xf, 0xf, false) != 0 && lane != 63;
      const bool prev_vec = __builtin_amdgcn_update_dpp(0, vec_i, 0x138 /*wave_shr:1*/, 0xf, 0xf, false) != 0 && lane != 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!row_ok[k])
          continue; // rows 0 / ny-1 and the groups at the ends of the loadable range: the per-cell path below
        if (col[k] == 0) {
          if (k < 3) {
            o[k] = raw[k + 1];
            done[k] = true;
          } else if (next_vec) {
            o[k].o0 = nx0;
            o[k].o1 = nx1;
            done[k] = true;
          } else {
            nocount |= 1u << k; // value through the per-cell path, counted here already
          }
        } else if (col[k] == nx - 1) {
          if (k > 0) {
            o[k] = raw[k - 1];
            done[k] = true;
          } else if (prev_vec) {
            o[k].o0 = pv0;
            o[k].o1 = pv1;
            done[k] = true;
          } else {
            nocount |= 1u << k;
          }
        } else {
          o[k] = raw[k];
          done[k] = true;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!done[k] && i0 + k < cell_end) {
        unsigned int b = 0;
        cell_one<OP, CHECK>(P, u, v, (long)(i0 + k), all, jmax, o[k], b);
        if (!((nocount >> k) & 1u))
          bad += b;
      }
    }
    if (i0 + 3 < cell_end) {
      stu4(out0 + i0, v4f_s{o[0].o0, o[1].o0, o[2].o0, o[3].o0});
      if (TWO_OUT && out1)
        stu4(out1 + i0, v4f_s{o[0].o1, o[1].o1, o[2].o1, o[3].o1});
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (i0 + k < cell_end) {
          out0[i0 + k] = o[k].o0;
          if (TWO_OUT && out1)
            out1[i0 + k] = o[k].o1;
        }
      }
    }
  }
  if (CHECK)
    block_count_add(P.n_undefined ? P.n_undefined + lev : nullptr, bad); // one atomic per workgroup
}

template <int OP>
hipError_t launch_cell(const StencilParams& prm, hipStream_t stream)
{
  const long n_local = (long)prm.nx * prm.ny_local;
  const int block = 256;
  long gx = (n_local + block - 1) / block;
  if (gx > 8192)
    gx = 8192;
  for (int l0 = 0; l0 < prm.nlev; l0 += 65535) {
    StencilParams p = prm;
    const int nl = (prm.nlev - l0 > 65535) ? 65535 : (prm.nlev - l0);
    p.f0 = prm.f0 + (size_t)l0 * prm.in_level_stride;
    p.f1 = prm.f1 ? prm.f1 + (size_t)l0 * prm.in_level_stride : nullptr;
    p.f2 = prm.f2 ? prm.f2 + (size_t)l0 * prm.in_level_stride : nullptr;
    p.out0 = prm.out0 + (size_t)l0 * prm.out_level_stride;
    p.out1 = prm.out1 ? prm.out1 + (size_t)l0 * prm.out_level_stride : nullptr;
    p.all_defined = prm.all_defined ? prm.all_defined + l0 : nullptr;
    p.n_undefined = prm.n_undefined ? prm.n_undefined + l0 : nullptr;
    p.scale_lev = prm.scale_lev ? prm.scale_lev + l0 : nullptr;
    p.scale2_lev = prm.scale2_lev ? prm.scale2_lev + l0 : nullptr;
    if (prm.every_level_all_defined)
      hipLaunchKernelGGL((stencil_cell_kernel<OP, false>), dim3((unsigned)gx, nl), dim3(block), 0, stream, p);
    else
      hipLaunchKernelGGL((stencil_cell_kernel<OP, true>), dim3((unsigned)gx, nl), dim3(block), 0, stream, p);
  }
  return hipGetLastError();
}

template <int OP>
hipError_t launch_wind_flat4(const StencilParams& prm, hipStream_t stream)
{
  const long rows = (prm.row_end > prm.row_begin) ? (long)(prm.row_end - prm.row_begin) : (long)prm.ny_local;
  const long cells = rows * prm.nx;
  const int blocks_per_level = (int)((cells + 1023) / 1024);
  // levels in slices that keep the unit count in 31 bits
  const int max_levels = (int)(0x3fffffffL / blocks_per_level) > 0 ? (int)(0x3fffffffL / blocks_per_level) : 1;
  for (int l0 = 0; l0 < prm.nlev; l0 += max_levels) {
    StencilParams p = prm;
    const int nl = (prm.nlev - l0 > max_levels) ? max_levels : (prm.nlev - l0);
    p.f0 = prm.f0 + (size_t)l0 * prm.in_level_stride;
    p.f1 = prm.f1 ? prm.f1 + (size_t)l0 * prm.in_level_stride : nullptr;
    p.out0 = prm.out0 + (size_t)l0 * prm.out_level_stride;
    p.out1 = prm.out1 ? prm.out1 + (size_t)l0 * prm.out_level_stride : nullptr;
    p.all_defined = prm.all_defined ? prm.all_defined + l0 : nullptr;
    p.n_undefined = prm.n_undefined ? prm.n_undefined + l0 : nullptr;
    const int n_units = blocks_per_level * nl;
    const int per_xcd = (n_units + 7) / 8;
    if (prm.every_level_all_defined)
      hipLaunchKernelGGL((wind_flat4_kernel<OP, false>), dim3((unsigned)(per_xcd * 8)), dim3(256), 0, stream, p, blocks_per_level, n_units, per_xcd);
    else
      hipLaunchKernelGGL((wind_flat4_kernel<OP, true>), dim3((unsigned)(per_xcd * 8)), dim3(256), 0, stream, p, blocks_per_level, n_units, per_xcd);
  }
  return hipGetLastError();
}

// the flat kernel wants dword-aligned fields (any float array is), 32-bit cell indices and at least four columns
inline bool wind_flat4_applies(const StencilParams& prm)
{
  const bool wind = prm.op == ST_RELVORT || prm.op == ST_ABSVORT || prm.op == ST_DIVERGENCE || prm.op == ST_VORTDIV || prm.op == ST_JACOBIAN;
  const bool needs_fc = prm.op == ST_ABSVORT || prm.op == ST_GWIND_X || prm.op == ST_GWIND_Y || prm.op == ST_GVORT || prm.op == ST_IGWIND;
  return !env().force_cell_kernel && prm.nx >= 4 && prm.ny_global >= 3 && (long)prm.nx * (prm.ny_local + 2) < 0x7fffff00L && (!wind || prm.f1) &&
         (!needs_fc || prm.fcoriolis) && prm.xmapr && prm.ymapr;
}

} // namespace

hipError_t launch_vortdiv_rows(const StencilParams& prm, hipStream_t stream, bool* handled); // mifc_vortdiv.hip
hipError_t launch_scalar_rows(const StencilParams& prm, hipStream_t stream, bool* handled);  // mifc_stencil_rows.hip
hipError_t launch_advection_oneshot(const StencilParams& prm, hipStream_t stream, bool* handled); // mifc_advection.hip
hipError_t launch_advection_split(const StencilParams& prm, hipStream_t stream, bool* handled);   // mifc_stencil_split.hip

namespace {
struct PrepLevels
{
  unsigned char* flags;
  u64* counts;
  int nlev, n_counts;
  unsigned int bits[kPrepMaxLevels / 32];
};
__global__ __launch_bounds__(256) void prep_levels_kernel(const PrepLevels P)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (P.flags && i < P.nlev)
    P.flags[i] = (unsigned char)((P.bits[i >> 5] >> (i & 31)) & 1u);
  if (P.counts && i < P.n_counts)
    P.counts[i] = 0;
}
} // namespace

hipError_t launch_prep_levels(const unsigned char* host_flags, int nlev, unsigned char* d_flags, u64* d_counts, int n_counts, hipStream_t stream)
{
  if (!host_flags)
    d_flags = nullptr;
  if ((d_flags && nlev > kPrepMaxLevels) || nlev < 0 || n_counts < 0)
    return hipErrorInvalidValue;
  const int n = (d_flags ? nlev : 0) > (d_counts ? n_counts : 0) ? nlev : (d_counts ? n_counts : 0);
  if (n == 0)
    return hipSuccess;
  PrepLevels P;
  P.flags = d_flags;
  P.counts = d_counts;
  P.nlev = nlev;
  P.n_counts = n_counts;
  for (unsigned int& w : P.bits)
    w = 0;
  if (d_flags)
    for (int l = 0; l < nlev; ++l)
      if (host_flags[l])
        P.bits[l >> 5] |= 1u << (l & 31);
  hipLaunchKernelGGL(prep_levels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, P);
  return hipGetLastError();
}

namespace {
thread_local const char* t_last_form = "";
}
void note_form(const char* form)
{
  t_last_form = form;
}
const char* last_form()
{
  return t_last_form;
}

hipError_t launch_stencil(const StencilParams& prm, hipStream_t stream)
{
  if (prm.nlev <= 0)
    return hipSuccess;
  note_form("");
  if (prm.op == ST_VORTDIV || prm.op == ST_RELVORT || prm.op == ST_DIVERGENCE || prm.op == ST_ABSVORT || prm.op == ST_JACOBIAN) {
    bool handled = false;
    const hipError_t e = launch_vortdiv_rows(prm, stream, &handled);
    if (handled)
      return e;
    if (prm.out_ff)
      return hipErrorNotSupported; // the caller computes the wind speed as a launch of its own
  }
  if (prm.op >= ST_GRAD_X && prm.op <= ST_IGWIND) {
    bool handled = false;
    const hipError_t e = launch_scalar_rows(prm, stream, &handled);
    if (handled)
      return e;
  }
  if (prm.op == ST_ADVECTION && !env().force_cell_kernel) {
    bool handled = false;
    const hipError_t es = launch_advection_split(prm, stream, &handled); // deep batches (mifc_stencil_split.hip)
    if (handled)
      return es;
    note_form("advection_oneshot");
    const hipError_t e = launch_advection_oneshot(prm, stream, &handled);
    if (handled)
      return e;
  }
  {
    const bool cell_only = prm.op == ST_GRAD_X || prm.op == ST_ADVECTION || prm.op == ST_TFP || prm.op == ST_QVEC_X || prm.op == ST_QVEC_Y;
    note_form(!cell_only && wind_flat4_applies(prm) ? "flat4" : "cell");
  }
  switch (prm.op) {
  case ST_RELVORT:
    return wind_flat4_applies(prm) ? launch_wind_flat4<ST_RELVORT>(prm, stream) : launch_cell<ST_RELVORT>(prm, stream);
  case ST_ABSVORT:
    return wind_flat4_applies(prm) ? launch_wind_flat4<ST_ABSVORT>(prm, stream) : launch_cell<ST_ABSVORT>(prm, stream);
  case ST_DIVERGENCE:
    return wind_flat4_applies(prm) ? launch_wind_flat4<ST_DIVERGENCE>(prm, stream) : launch_cell<ST_DIVERGENCE>(prm, stream);
  case ST_VORTDIV:
    return wind_flat4_applies(prm) ? launch_wind_flat4<ST_VORTDIV>(prm, stream) : launch_cell<ST_VORTDIV>(prm, stream);
  case ST_GRAD_X:
    return launch_cell<ST_GRAD_X>(prm, stream);
  case ST_GRAD_Y:
    return wind_flat4_applies(prm) ? launch_wind_flat4<ST_GRAD_Y>(prm, stream) : launch_cell<ST_GRAD_Y>(prm, stream);
  case ST_GRAD_ABS:
    return wind_flat4_applies(prm) ? launch_wind_flat4<ST_GRAD_ABS>(prm, stream) : launch_cell<ST_GRAD_ABS>(prm, stream);
  case ST_GRAD_LAP:
    return wind_flat4_applies(prm) ? launch_wind_flat4<ST_GRAD_LAP>(prm, stream) : launch_cell<ST_GRAD_LAP>(prm, stream);
  case ST_GWIND_X:
    return wind_flat4_applies(prm) ? launch_wind_flat4<ST_GWIND_X>(prm, stream) : launch_cell<ST_GWIND_X>(prm, stream);
  case ST_GWIND_Y:
    return wind_flat4_applies(prm) ? launch_wind_flat4<ST_GWIND_Y>(prm, stream) : launch_cell<ST_GWIND_Y>(prm, stream);
  case ST_GVORT:
    return wind_flat4_applies(prm) ? launch_wind_flat4<ST_GVORT>(prm, stream) : launch_cell<ST_GVORT>(prm, stream);
  case ST_IGWIND:
    return wind_flat4_applies(prm) ? launch_wind_flat4<ST_IGWIND>(prm, stream) : launch_cell<ST_IGWIND>(prm, stream);
  case ST_ADVECTION:
    return launch_cell<ST_ADVECTION>(prm, stream);
  case ST_JACOBIAN:
    return wind_flat4_applies(prm) ? launch_wind_flat4<ST_JACOBIAN>(prm, stream) : launch_cell<ST_JACOBIAN>(prm, stream);
  case ST_TFP:
    return launch_cell<ST_TFP>(prm, stream);
  case ST_QVEC_X:
    return launch_cell<ST_QVEC_X>(prm, stream);
  case ST_QVEC_Y:
    return launch_cell<ST_QVEC_Y>(prm, stream);
  default:
    return hipErrorInvalidValue;
  }
}

} // namespace mifc
