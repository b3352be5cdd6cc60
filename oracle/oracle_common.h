/*
 * oracle_common.h -- TEST INFRASTRUCTURE, not product code.
 *
 * Constants and point helpers shared by the translation units of the CPU
 * restatement (oracle/mifc_oracle.cc, oracle/mifc_oracle_catalogue.cc).
 * Reference lines are cited per item (paths relative to
 * /root/reference/src/mi_fieldcalc/).
 */
#ifndef MIFC_ORACLE_COMMON_H
#define MIFC_ORACLE_COMMON_H

#include <climits>
#include <cmath>
#include <cstddef>
#include <cstring>

namespace {

enum { ALL_DEFINED = 0, NONE_DEFINED = 1, SOME_DEFINED = 2 }; // FieldDefined.h:41

// MetConstants.h:43-53 (all float; literals are double and get rounded on initialisation)
const float K_R = 287., K_CP = 1004., K_P0 = 1000., K_T0 = 273.15;
const float K_EPS = 0.622;
const float K_XLH = 2.501e+6;
const float K_P0INV = 1. / K_P0;
const float K_KAPPA = K_R / K_CP;
const float K_G = 9.8;
const float K_RHMIN = 0.02, K_RHMAX = 1.00;

// MetConstants.h:56-59: saturation vapour pressure over water, -100..+100 C step 5
const int K_NEWT = 41;
const float K_EWT[K_NEWT] = {.000034, .000089, .000220, .000517, .001155, .002472, .005080, .01005, .01921, .03553, .06356,
                             .1111,   .1891,   .3139,   .5088,   .8070,   1.2540,  1.9118,  2.8627, 4.2148, 6.1078, 8.7192,
                             12.272,  17.044,  23.373,  31.671,  42.430,  56.236,  73.777,  95.855, 123.40, 157.46, 199.26,
                             250.16,  311.69,  385.56,  473.67,  578.09,  701.13,  845.28,  1013.25};

// FieldCalculations.h:42-45
inline bool defined1(float x, float undef)
{
  return !std::isnan(x) && x != undef;
}

// FieldDefined.cc:62-70
inline int classify(size_t n_undefined, size_t n)
{
  if (n_undefined == 0)
    return ALL_DEFINED;
  if (n_undefined == n)
    return NONE_DEFINED;
  return SOME_DEFINED;
}

inline int trunc_like_x86(float x)
{
  if (!(x >= -2147483648.0f && x < 2147483648.0f))
    return INT_MIN;
  return (int)x;
}

// MetConstants.h:61-84 + MetConstants.cc:37-45
struct Ewt
{
  float x;
  int l;
  explicit Ewt(float t_celsius)
      : x((float)(((double)t_celsius + 100.) * 0.2))
      , l(trunc_like_x86(x))
  {
  }
  bool ok() const { return l >= 0 && l < K_NEWT - 1; }
  float value() const { return K_EWT[l] + (K_EWT[l + 1] - K_EWT[l]) * (x - (float)l); }
  float inverse(float et) const
  {
    int ll = l;
    while (ll > 0 && ll < K_NEWT - 1 && K_EWT[ll] > et)
      ll--;
    const float r = (et - K_EWT[ll]) / (K_EWT[ll + 1] - K_EWT[ll]);
    return (float)(-100. + (double)((float)ll + r) * 5.);
  }
};

// FieldCalculations.cc:186-194
inline float clamp_rh(float rh)
{
  if (rh < K_RHMIN)
    return K_RHMIN;
  if (rh > K_RHMAX)
    return K_RHMAX;
  return rh;
}

// FieldCalculations.cc:308-316
inline float pidcp_of(float p)
{
  return powf(p * K_P0INV, K_KAPPA);
}
inline float pi_of(float p)
{
  return K_CP * pidcp_of(p);
}

// FieldCalculations.cc:298-301
inline bool bad_hlevel(float a, float b)
{
  return (a < 0.0) || (b < 0.0) || (a == 0.0 && b == 0.0) || (b > 1.0);
}

// The humidity / theta-e point functions return false when the ewt table does
// not cover the temperature (cell becomes undef and is counted).
// FieldCalculations.cc:196-205
inline bool t_thesat(float tk, float p, float pi, float& out)
{
  Ewt e(tk - K_T0);
  if (!e.ok())
    return false;
  const float qsat = K_EPS * e.value() / p;
  out = (K_CP * tk + K_XLH * qsat) / pi;
  return true;
}
// FieldCalculations.cc:207-216
inline bool th_thesat(float th, float p, float pi, float& out)
{
  Ewt e(th * pi / K_CP - K_T0);
  if (!e.ok())
    return false;
  const float qsat = K_EPS * e.value() / p;
  out = th + K_XLH * qsat / pi;
  return true;
}
// FieldCalculations.cc:218-227
inline bool tk_q_rh(float tk, float q, float p, float& out)
{
  Ewt e(tk - K_T0);
  if (!e.ok())
    return false;
  const float qsat = K_EPS * e.value() / p;
  out = (float)(100. * (double)q / (double)qsat);
  return true;
}
// FieldCalculations.cc:229-238
inline bool tk_rh_q(float tk, float rh, float p, float& out)
{
  Ewt e(tk - K_T0);
  if (!e.ok())
    return false;
  const float qsat = K_EPS * e.value() / p;
  out = (float)(0.01 * (double)rh * (double)qsat);
  return true;
}
// FieldCalculations.cc:240-253
inline bool tk_q_td(float tk, float q, float p, float tdconv, float& out)
{
  Ewt e(tk - K_T0);
  if (!e.ok())
    return false;
  const float et = e.value();
  const float qsat = K_EPS * et / p;
  const float rh = clamp_rh(q / qsat);
  out = e.inverse(rh * et) + tdconv;
  return true;
}
// FieldCalculations.cc:255-267
inline bool tk_rh_td(float tk, float rh100, float tdconv, float& out)
{
  Ewt e(tk - K_T0);
  if (!e.ok())
    return false;
  const float et = e.value();
  const float rh = clamp_rh((float)(0.01 * (double)rh100));
  out = e.inverse(rh * et) + tdconv;
  return true;
}

// FieldCalculations.cc:59-74: columns first (rows 1..ny-2), then rows 0 and ny-1
inline void fill_edges(int nx, int ny, float* f)
{
  for (int j = 1; j < ny - 1; ++j) {
    f[j * nx] = f[j * nx + 1];
    f[j * nx + nx - 1] = f[j * nx + nx - 2];
  }
  for (int i = 0; i < nx; ++i) {
    f[i] = f[i + nx];
    f[(ny - 1) * nx + i] = f[(ny - 2) * nx + i];
  }
}

inline bool unit_is(const char* unit, const char* what)
{
  return unit && std::strcmp(unit, what) == 0;
}

// Shared driver for the flat stencil loops.  `cell(i, out_ok)` computes cell i,
// returns false if the cell is undefined.  Loop range and the count the flag
// is classified against are per operator (they differ: Appendix A #4, #5).
template <class Cell>
inline size_t flat_loop(int first, int last, const Cell& cell)
{
  size_t n_undefined = 0;
  for (int i = first; i < last; ++i)
    if (!cell(i))
      n_undefined += 1;
  return n_undefined;
}

} // namespace

#endif // MIFC_ORACLE_COMMON_H
