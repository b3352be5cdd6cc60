// mifc_device.h -- device-side building blocks shared by the gfx950 kernels.
//
// Arithmetic contract: every reference
// expression that contains a bare double literal (0.5, 2., 100., 0.01 ...) is
// evaluated in double and rounded to float once, on the store; expressions of
// float variables only stay float.  The helpers below spell the promotion out
// operation by operation; the library is compiled with -ffp-contract=off so no
// multiply-add is fused (the reference build has no FMA: -mavx2 without -mfma,
// src/mi_fieldcalc/CMakeLists.txt:55-56).
#ifndef MIFC_DEVICE_H
#define MIFC_DEVICE_H

#include <hip/hip_runtime.h>

namespace mifc {

typedef unsigned long long u64;

// MetConstants.h:43-53.  Same float values as the reference: each is the
// double literal (or double expression) rounded to float.
#define MIFC_K_R 287.0f
#define MIFC_K_CP 1004.0f
#define MIFC_K_P0INV ((float)(1. / 1000.0))
#define MIFC_K_T0 ((float)273.15)
#define MIFC_K_EPS ((float)0.622)
#define MIFC_K_XLH ((float)2.501e+6)
#define MIFC_K_KAPPA (287.0f / 1004.0f)
#define MIFC_K_G ((float)9.8)
#define MIFC_K_RHMIN ((float)0.02)
#define MIFC_K_RHMAX 1.0f
#define MIFC_N_EWT 41

// FieldCalculations.h:42-45
__device__ __forceinline__ bool is_def(float x, float undef)
{
  return !(x != x) && x != undef;
}

// ---- saturation vapour pressure table (MetConstants.h:56-84, MetConstants.cc:37-45)
// The 41-entry table lives in LDS (164 B per workgroup): lookups are per-lane
// dynamic indices, which LDS serves without a trip through the vector cache.
struct EwtTable
{
  const float* tab; // LDS
};

__device__ __forceinline__ void ewt_table_init(float* lds_tab)
{
  // values are the reference's double literals rounded to float (MetConstants.h:57-59)
  const float init[MIFC_N_EWT] = {
      (float).000034, (float).000089, (float).000220, (float).000517, (float).001155, (float).002472, (float).005080, (float).01005, (float).01921,
      (float).03553,  (float).06356,  (float).1111,   (float).1891,   (float).3139,   (float).5088,   (float).8070,   (float)1.2540, (float)1.9118,
      (float)2.8627,  (float)4.2148,  (float)6.1078,  (float)8.7192,  (float)12.272,  (float)17.044,  (float)23.373,  (float)31.671, (float)42.430,
      (float)56.236,  (float)73.777,  (float)95.855,  (float)123.40,  (float)157.46,  (float)199.26,  (float)250.16,  (float)311.69, (float)385.56,
      (float)473.67,  (float)578.09,  (float)701.13,  (float)845.28,  (float)1013.25};
  for (int k = threadIdx.x; k < MIFC_N_EWT; k += blockDim.x)
    lds_tab[k] = init[k];
  __syncthreads();
}

struct Ewt
{
  float x;
  int l;
  __device__ __forceinline__ explicit Ewt(float t_celsius)
  {
    x = (float)(((double)t_celsius + 100.) * 0.2); // MetConstants.h:65
    // int(x): the compiled reference (x86-64 cvttss2si) yields INT_MIN for NaN
    // and out-of-range values; v_cvt_i32_f32 would give 0 / saturate.
    l = (x >= -2147483648.0f && x < 2147483648.0f) ? (int)x : (int)0x80000000;
  }
  __device__ __forceinline__ bool ok() const { return l >= 0 && l < MIFC_N_EWT - 1; }
  __device__ __forceinline__ float value(const float* tab) const { return tab[l] + (tab[l + 1] - tab[l]) * (x - (float)l); }
  __device__ __forceinline__ float inverse(const float* tab, float et) const
  {
    int ll = l;
    while (ll > 0 && ll < MIFC_N_EWT - 1 && tab[ll] > et)
      ll--;
    const float r = (et - tab[ll]) / (tab[ll + 1] - tab[ll]);
    return (float)(-100. + (double)((float)ll + r) * 5.);
  }
};

// FieldCalculations.cc:186-194
__device__ __forceinline__ float clamp_rh(float rh)
{
  if (rh < MIFC_K_RHMIN)
    return MIFC_K_RHMIN;
  if (rh > MIFC_K_RHMAX)
    return MIFC_K_RHMAX;
  return rh;
}

// FieldCalculations.cc:308-311: powf(p * p0inv, kappa).
// The reference calls glibc's powf, which evaluates in double and returns a
// nearly correctly rounded float.  The device's float powf is several ulp off,
// and the saturation-pressure table amplifies a temperature error (up to
// ~0.3 per kelvin at its cold end), so the power is taken in double here:
// exp2(kappa * log2(x)) with the double-precision device functions rounds to
// the correctly rounded float except in rare near-halfway cases.  Agreement
// with glibc is therefore "identical in almost every cell, 1 ulp otherwise";
// the parity bound for the operators that use it is 1e-5 relative
// (BASELINE.json), not bit-exact.  Special values behave like powf with a
// positive non-integer exponent: x<0 -> NaN, 0 -> 0, inf -> inf, NaN -> NaN.
__device__ __forceinline__ float pidcp_of(float p)
{
  const float x = p * MIFC_K_P0INV;
  return (float)exp2((double)MIFC_K_KAPPA * log2((double)x));
}

// The point functions return false where the table does not cover tk
// (reference: cell := undef, n_undefined += 1).
// FieldCalculations.cc:196-205
__device__ __forceinline__ bool t_thesat(const float* tab, float tk, float p, float pi, float& out)
{
  Ewt e(tk - MIFC_K_T0);
  if (!e.ok())
    return false;
  const float qsat = MIFC_K_EPS * e.value(tab) / p;
  out = (MIFC_K_CP * tk + MIFC_K_XLH * qsat) / pi;
  return true;
}
// FieldCalculations.cc:207-216
__device__ __forceinline__ bool th_thesat(const float* tab, float th, float p, float pi, float& out)
{
  Ewt e(th * pi / MIFC_K_CP - MIFC_K_T0);
  if (!e.ok())
    return false;
  const float qsat = MIFC_K_EPS * e.value(tab) / p;
  out = th + MIFC_K_XLH * qsat / pi;
  return true;
}
// FieldCalculations.cc:218-227
__device__ __forceinline__ bool tk_q_rh(const float* tab, float tk, float q, float p, float& out)
{
  Ewt e(tk - MIFC_K_T0);
  if (!e.ok())
    return false;
  const float qsat = MIFC_K_EPS * e.value(tab) / p;
  out = (float)(100. * (double)q / (double)qsat);
  return true;
}
// FieldCalculations.cc:229-238
__device__ __forceinline__ bool tk_rh_q(const float* tab, float tk, float rh, float p, float& out)
{
  Ewt e(tk - MIFC_K_T0);
  if (!e.ok())
    return false;
  const float qsat = MIFC_K_EPS * e.value(tab) / p;
  out = (float)(0.01 * (double)rh * (double)qsat);
  return true;
}
// FieldCalculations.cc:240-253
__device__ __forceinline__ bool tk_q_td(const float* tab, float tk, float q, float p, float tdconv, float& out)
{
  Ewt e(tk - MIFC_K_T0);
  if (!e.ok())
    return false;
  const float et = e.value(tab);
  const float qsat = MIFC_K_EPS * et / p;
  const float rh = clamp_rh(q / qsat);
  out = e.inverse(tab, rh * et) + tdconv;
  return true;
}
// FieldCalculations.cc:255-267
__device__ __forceinline__ bool tk_rh_td(const float* tab, float tk, float rh100, float tdconv, float& out)
{
  Ewt e(tk - MIFC_K_T0);
  if (!e.ok())
    return false;
  const float et = e.value(tab);
  const float rh = clamp_rh((float)(0.01 * (double)rh100));
  out = e.inverse(tab, rh * et) + tdconv;
  return true;
}

// math_util.h:57-60: sqrt(x*x + y*y) in float, correctly rounded sqrt, no fma
__device__ __forceinline__ float absval(float x, float y)
{
  return sqrtf(x * x + y * y);
}

// ---- stencil point formulas (double-promoted; inputs are float differences)
#ifdef MIFC_EXPERIMENT_F32_COMBINE
// NOT parity-correct: float-only combine, exists solely so that tools/ can
// measure how much of the kernel time is the fp64 pipe (never built into the product)
__device__ __forceinline__ float f_relvort(float xm, float ym, float dvdx, float dudy) { return 0.5f * xm * dvdx - 0.5f * ym * dudy; }
__device__ __forceinline__ float f_diverg(float xm, float ym, float dudx, float dvdy) { return 0.5f * xm * dudx + 0.5f * ym * dvdy; }
#define MIFC_SKIP_F64_COMBINE 1
#endif
#ifndef MIFC_SKIP_F64_COMBINE
// FieldCalculations.cc:1862
__device__ __forceinline__ float f_relvort(float xm, float ym, float dvdx, float dudy)
{
  return (float)(0.5 * (double)xm * (double)dvdx - 0.5 * (double)ym * (double)dudy);
}
// FieldCalculations.cc:1896
__device__ __forceinline__ float f_absvort(float xm, float ym, float dvdx, float dudy, float fc)
{
  return (float)(0.5 * (double)xm * (double)dvdx - 0.5 * (double)ym * (double)dudy + (double)fc);
}
// FieldCalculations.cc:1928
__device__ __forceinline__ float f_diverg(float xm, float ym, float dudx, float dvdy)
{
  return (float)(0.5 * (double)xm * (double)dudx + 0.5 * (double)ym * (double)dvdy);
}
#endif

// ---- undefined-cell counting: one atomic per wave, none when nothing to add
__device__ __forceinline__ void wave_count_add(u64* counter, unsigned int my_count)
{
  // wave64 butterfly sum
  unsigned int s = my_count;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0 && s != 0)
    atomicAdd(counter, (u64)s);
}

} // namespace mifc

#endif // MIFC_DEVICE_H
