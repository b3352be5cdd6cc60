// mifc_scalar_cell.h -- what the kernels of the single-input stencil operators share (mifc_stencil_rows.hip: row-walking,
// one-shot and level-walking forms; mifc_stencil_split.hip: split-role level-walking form): the launch parameters, the
// lane-neighbour helpers and the point formulas of gradient compute 1..4 (FieldCalculations.cc:1985-2074),
// plevelgwind_xcomp (:638), plevelgwind_ycomp (:674), plevelgvort (:708), ilevelgwind (:1511).
#ifndef MIFC_SCALAR_CELL_H
#define MIFC_SCALAR_CELL_H

#include "mifc_device.h"
#include "mifc_kernels.h"

namespace mifc {

struct SRowsParams
{
  int nx, ny;
  int R, nbands, nwc, nlev, wpb;
  int uL, uB, uW, n_logical, per_xcd;
  const float* f;
  const float *xm, *ym, *fc; // any may be null when the operator does not use it
  float *o0, *o1;
  long in_stride, out_stride;
  const unsigned char* all_defined;
  float undef;
  u64* n_undefined;
  const float *g1, *g2; // advection_split_kernel: the wind components u, v (own rows only)
  float scale;          // ... and -3600 * hours
  unsigned int* partials; // big tested levels: the workgroup's count of a level goes to partials[level][unit] (StencilParams::partials)
  long partials_cap;      // entries of it
  int ragged; // split-role form only: rows at any alignment (a width that is not a multiple of 4, unaligned fields or level strides)
};

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));


__device__ __forceinline__ float dpp_lower(float keep_if_none, float x)
{
  return __builtin_bit_cast(float,
                            __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep_if_none), __builtin_bit_cast(int, x), 0x138 /*wave_shr:1*/, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_upper(float keep_if_none, float x)
{
  return __builtin_bit_cast(float,
                            __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep_if_none), __builtin_bit_cast(int, x), 0x130 /*wave_shl:1*/, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_value(float x, int src_lane) // by value: see mifc_vortdiv.hip
{
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src_lane));
}
__device__ __forceinline__ v4f ld4(const float* p)
{
  return *reinterpret_cast<const v4f*>(p);
}
__device__ __forceinline__ void st4_stream(float* p, const float (&z)[4])
{
  v4f t;
  t.x = z[0];
  t.y = z[1];
  t.z = z[2];
  t.w = z[3];
  __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
}

// One cell of operator OP.  w, c, e: row j; s, n: rows j-1, j+1.  Returns false
// when the cell is undefined.  Formulas are those of stencil_raw() in
// mifc_stencil.hip, i.e. of the reference lines cited there.
template <int OP, bool CHECK>
__device__ __forceinline__ bool scalar_cell(bool all, float undef, float w, float c, float e, float s, float n, float xm, float ym, float fc, float& o0,
                                            float& o1)
{
  // straight-line code: the tests are combined without short-circuits, the formula runs unconditionally (on undefined
  // inputs it produces some number, infinity or NaN that the caller's select discards) -- see all_def()
  if (OP == ST_GRAD_X) { // :2015-2016
    o0 = half_prod(xm, e - w);
    return !CHECK || (all | all_def(undef, w, e));
  }
  if (OP == ST_GRAD_Y) { // :2027-2028
    o0 = half_prod(ym, n - s);
    return !CHECK || (all | all_def(undef, s, n));
  }
  if (OP == ST_GRAD_LAP || OP == ST_GVORT) {
    const double dxm = xm, dym = ym;
    if (OP == ST_GRAD_LAP) { // :2054-2056
      const float d2x = (float)((double)w - 2.0 * (double)c + (double)e);
      const float d2y = (float)((double)s - 2.0 * (double)c + (double)n);
      o0 = (float)(4.0 * (0.25 * dxm * dxm * (double)d2x + 0.25 * dym * dym * (double)d2y));
    } else { // :730-731
      const float g4 = (float)((double)MIFC_K_G * 4.);
      const double d2x = (double)w - 2. * (double)c + (double)e;
      const double d2y = (double)s - 2. * (double)c + (double)n;
      // the f64 division through the refined reciprocal: for a float-born divisor it IS the IEEE quotient, bit for bit
      // (shared_reciprocal() in mifc_device.h; mifc_diag_division checks it), a few instructions shorter than the expansion
      const double fd = (double)fc;
      o0 = (float)quotient((0.25 * dxm * dxm * d2x + 0.25 * dym * dym * d2y) * (double)g4, fd, shared_reciprocal(fd));
    }
    return !CHECK || (all | all_def(undef, s, w, c, e, n)); // :2053, :729
  }
  if (OP == ST_GRAD_ABS) { // :2040-2042
    const float dfdx = half_prod(xm, e - w);
    const float dfdy = half_prod(ym, n - s);
    o0 = absval(dfdx, dfdy);
  } else if (OP == ST_GWIND_X) { // :661
    const double fd = (double)fc;
    o0 = (float)quotient(-0.5 * (double)ym * (double)(n - s) * (double)MIFC_K_G, fd, shared_reciprocal(fd));
  } else if (OP == ST_GWIND_Y) { // :694
    const double fd = (double)fc;
    o0 = (float)quotient(0.5 * (double)xm * (double)(e - w) * (double)MIFC_K_G, fd, shared_reciprocal(fd));
  } else { // ST_IGWIND :1535-1536: two quotients by the same divisor share its reciprocal
    const double fd = (double)fc, finv = shared_reciprocal(fd);
    o0 = (float)quotient(-0.5 * (double)ym * (double)(n - s), fd, finv);
    o1 = (float)quotient(0.5 * (double)xm * (double)(e - w), fd, finv);
  }
  return !CHECK || (all | all_def(undef, s, w, e, n)); // :2039, :660, :693, :1534
}

// ---- the same point formulas for kernels that STAY on a tile while they walk through the levels (mifc_stencil_split.hip):
// whatever depends only on the map factors and the Coriolis parameter is evaluated once per chunk of levels, not once
// per level -- the double conversions, the products 0.25*xm*xm / 0.25*ym*ym of the Laplacian-type operators, and above all
// the refined reciprocal of the Coriolis parameter (v_rcp_f64 + four fmas) behind every geostrophic quotient.  Each hoisted
// value is exactly the intermediate the per-cell formula above computes (same operations, same order), so the results are
// the same bits; what changes is 6-12 fp64-pipe instructions per cell and level fewer for the operators that were bound
// by instruction issue (gradient compute 4, plevelgwind_x/ycomp, plevelgvort, ilevelgwind).
template <int OP>
struct ScalarHoist
{
  double a[4]; // GRAD_LAP / GVORT: 0.25*xm*xm;  GWIND_X / IGWIND: -0.5*ym;  GWIND_Y: 0.5*xm
  double b[4]; // GRAD_LAP / GVORT: 0.25*ym*ym;  IGWIND: 0.5*xm
  double fd[4], finv[4]; // Coriolis parameter and its refined reciprocal (shared_reciprocal)
  float xm[4], ym[4];    // the one-sided gradients keep the float factors (half_prod)
  float hx[4], hy[4];    // ... and their halves: 0.5f * m is exact for |m| >= 2^-125, and then (0.5f * m) * d IS half_prod(m, d)
  bool halves_exact;     // every map factor of this lane is that large (or zero): the lane may skip half_prod's selects
  __device__ __forceinline__ void init(const v4f& xm4, const v4f& ym4, const v4f& fc4)
  {
    halves_exact = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xm[k] = xm4[k];
      ym[k] = ym4[k];
      hx[k] = 0.5f * xm4[k];
      hy[k] = 0.5f * ym4[k];
      if (OP == ST_GRAD_X || OP == ST_GRAD_ABS)
        halves_exact = halves_exact & (__builtin_fabsf(xm4[k]) >= 0x1p-125f || xm4[k] == 0.f);
      if (OP == ST_GRAD_Y || OP == ST_GRAD_ABS)
        halves_exact = halves_exact & (__builtin_fabsf(ym4[k]) >= 0x1p-125f || ym4[k] == 0.f);
      if (OP == ST_GRAD_LAP || OP == ST_GVORT) {
        const double dxm = xm4[k], dym = ym4[k];
        a[k] = 0.25 * dxm * dxm;
        b[k] = 0.25 * dym * dym;
      } else if (OP == ST_GWIND_X) {
        a[k] = -0.5 * (double)ym4[k];
      } else if (OP == ST_GWIND_Y) {
        a[k] = 0.5 * (double)xm4[k];
      } else if (OP == ST_IGWIND) {
        a[k] = -0.5 * (double)ym4[k];
        b[k] = 0.5 * (double)xm4[k];
      }
      if (OP == ST_GWIND_X || OP == ST_GWIND_Y || OP == ST_GVORT || OP == ST_IGWIND) {
        fd[k] = (double)fc4[k];
        finv[k] = shared_reciprocal(fd[k]);
      }
    }
  }
};

// "ordered and different from undef": is_def() for an undef that is not NaN, in ONE compare (the launcher keeps a NaN undef
// away from the kernels that use it)
template <typename... T>
__device__ __forceinline__ bool all_lg(float undef, T... x)
{
  return ((bool)__builtin_islessgreater(x, undef) & ...);
}

// HALVES: the wave's map factors all have exact halves (ScalarHoist::halves_exact in every lane): (float)(0.5 * m * d) is then the
// one float multiplication (0.5f * m) * d -- the exact product 0.5*m*d rounded once, like the reference's double product
// rounded to float (see half_prod) -- instead of half_prod's compare, two selects and two multiplications.
template <int OP, bool CHECK, bool HALVES>
__device__ __forceinline__ bool scalar_cell_hoisted(bool all, float undef, float w, float c, float e, float s, float n, const ScalarHoist<OP>& H, int k,
                                                    float& o0, float& o1)
{
  if (OP == ST_GRAD_X) { // :2015-2016
    o0 = HALVES ? H.hx[k] * (e - w) : half_prod(H.xm[k], e - w);
    return !CHECK || (all | all_lg(undef, w, e));
  }
  if (OP == ST_GRAD_Y) { // :2027-2028
    o0 = HALVES ? H.hy[k] * (n - s) : half_prod(H.ym[k], n - s);
    return !CHECK || (all | all_lg(undef, s, n));
  }
  if (OP == ST_GRAD_LAP || OP == ST_GVORT) {
    // w - 2.0*c: the product 2.0*c is exact, so the difference is ONE rounding -- which is what fma(-2.0, c, w) delivers
    const double dc = (double)c;
    const double sx = __builtin_fma(-2.0, dc, (double)w) + (double)e;
    const double sy = __builtin_fma(-2.0, dc, (double)s) + (double)n;
    if (OP == ST_GRAD_LAP) { // :2054-2056: the second differences are rounded to float first
      const float d2x = (float)sx;
      const float d2y = (float)sy;
      o0 = (float)(4.0 * (H.a[k] * (double)d2x + H.b[k] * (double)d2y));
    } else { // :730-731
      const float g4 = (float)((double)MIFC_K_G * 4.);
      o0 = (float)quotient((H.a[k] * sx + H.b[k] * sy) * (double)g4, H.fd[k], H.finv[k]);
    }
    return !CHECK || (all | all_lg(undef, s, w, c, e, n)); // :2053, :729
  }
  if (OP == ST_GRAD_ABS) { // :2040-2042
    const float dfdx = HALVES ? H.hx[k] * (e - w) : half_prod(H.xm[k], e - w);
    const float dfdy = HALVES ? H.hy[k] * (n - s) : half_prod(H.ym[k], n - s);
    o0 = absval(dfdx, dfdy);
  } else if (OP == ST_GWIND_X) { // :661
    o0 = (float)quotient(H.a[k] * (double)(n - s) * (double)MIFC_K_G, H.fd[k], H.finv[k]);
  } else if (OP == ST_GWIND_Y) { // :694
    o0 = (float)quotient(H.a[k] * (double)(e - w) * (double)MIFC_K_G, H.fd[k], H.finv[k]);
  } else { // ST_IGWIND :1535-1536
    o0 = (float)quotient(H.a[k] * (double)(n - s), H.fd[k], H.finv[k]);
    o1 = (float)quotient(H.b[k] * (double)(e - w), H.fd[k], H.finv[k]);
  }
  return !CHECK || (all | all_lg(undef, s, w, e, n)); // :2039, :660, :693, :1534
}

} // namespace

// split-role level-walking form (mifc_stencil_split.hip); rp.uB / rp.uW / rp.wpb / rp.n_logical / rp.per_xcd are set by it
// advection (three input fields) on split-role level-walking tiles; *handled stays false where that form does not apply
hipError_t launch_advection_split(const StencilParams& prm, hipStream_t stream, bool* handled);
bool scalar_split_applies(int op, int nx, int ny, int nlev, bool check, float undef, bool ragged);
hipError_t launch_scalar_split(int op, SRowsParams& rp, bool check, hipStream_t stream);

} // namespace mifc

#endif // MIFC_SCALAR_CELL_H
