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
    const float dtdx = (float)(hx * (double)P.scale * (double)(te - tw));
    const float dtdy = (float)(hy * (double)P.scale * (double)(tn - ts));
    if (OP == ST_QVEC_X) {
      const float dugdx = half_prod(P.xmapr[p], ue - uw);
      const float dvgdx = half_prod(P.xmapr[p], ve - vw);
      o.o0 = P.scale2 * (dugdx * dtdx + dvgdx * dtdy);
    } else {
      const float dugdy = half_prod(P.ymapr[p], un - us);
      const float dvgdy = half_prod(P.ymapr[p], vn - vs);
      o.o0 = P.scale2 * (dugdy * dtdx + dvgdy * dtdy);
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
    const int jl = (int)(i / nx);
    const int c = (int)(i - (long)jl * nx);
    const int j = P.j0 + jl;
    // where the value of this cell comes from after fillEdges
    const int jj = j < 1 ? 1 : (j > jmax ? jmax : j);
    const int cc = c < 1 ? 1 : (c > nx - 2 ? nx - 2 : c);
    const long p = (long)(jj - P.j0) * nx + cc;
    CellOut o = {P.undef, P.undef};
    const bool ok = stencil_raw<OP, CHECK>(P, f0, f1, p, all, o);
    if (!ok) {
      o.o0 = P.undef;
      o.o1 = P.undef;
    }
    out0[i] = o.o0;
    if ((OP == ST_VORTDIV || OP == ST_IGWIND) && out1)
      out1[i] = o.o1;

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
    if (prm.every_level_all_defined)
      hipLaunchKernelGGL((stencil_cell_kernel<OP, false>), dim3((unsigned)gx, nl), dim3(block), 0, stream, p);
    else
      hipLaunchKernelGGL((stencil_cell_kernel<OP, true>), dim3((unsigned)gx, nl), dim3(block), 0, stream, p);
  }
  return hipGetLastError();
}

} // namespace

hipError_t launch_vortdiv_rows(const StencilParams& prm, hipStream_t stream, bool* handled); // mifc_vortdiv.hip
hipError_t launch_scalar_rows(const StencilParams& prm, hipStream_t stream, bool* handled);  // mifc_stencil_rows.hip
hipError_t launch_advection_oneshot(const StencilParams& prm, hipStream_t stream, bool* handled); // mifc_advection.hip

hipError_t launch_stencil(const StencilParams& prm, hipStream_t stream)
{
  if (prm.nlev <= 0)
    return hipSuccess;
  if (prm.op == ST_VORTDIV || prm.op == ST_RELVORT || prm.op == ST_DIVERGENCE || prm.op == ST_ABSVORT || prm.op == ST_JACOBIAN) {
    bool handled = false;
    const hipError_t e = launch_vortdiv_rows(prm, stream, &handled);
    if (handled)
      return e;
  }
  if (prm.op >= ST_GRAD_X && prm.op <= ST_IGWIND) {
    bool handled = false;
    const hipError_t e = launch_scalar_rows(prm, stream, &handled);
    if (handled)
      return e;
  }
  if (prm.op == ST_ADVECTION && !env().force_cell_kernel) {
    bool handled = false;
    const hipError_t e = launch_advection_oneshot(prm, stream, &handled);
    if (handled)
      return e;
  }
  switch (prm.op) {
  case ST_RELVORT:
    return launch_cell<ST_RELVORT>(prm, stream);
  case ST_ABSVORT:
    return launch_cell<ST_ABSVORT>(prm, stream);
  case ST_DIVERGENCE:
    return launch_cell<ST_DIVERGENCE>(prm, stream);
  case ST_VORTDIV:
    return launch_cell<ST_VORTDIV>(prm, stream);
  case ST_GRAD_X:
    return launch_cell<ST_GRAD_X>(prm, stream);
  case ST_GRAD_Y:
    return launch_cell<ST_GRAD_Y>(prm, stream);
  case ST_GRAD_ABS:
    return launch_cell<ST_GRAD_ABS>(prm, stream);
  case ST_GRAD_LAP:
    return launch_cell<ST_GRAD_LAP>(prm, stream);
  case ST_GWIND_X:
    return launch_cell<ST_GWIND_X>(prm, stream);
  case ST_GWIND_Y:
    return launch_cell<ST_GWIND_Y>(prm, stream);
  case ST_GVORT:
    return launch_cell<ST_GVORT>(prm, stream);
  case ST_IGWIND:
    return launch_cell<ST_IGWIND>(prm, stream);
  case ST_ADVECTION:
    return launch_cell<ST_ADVECTION>(prm, stream);
  case ST_JACOBIAN:
    return launch_cell<ST_JACOBIAN>(prm, stream);
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
