// Meteorological constants and the saturation-vapour-pressure table used by
// the humidity / theta operators.  Names, types and values are those of the
// reference's src/mi_fieldcalc/MetConstants.h:39-84 so that callers keep
// compiling; the GPU kernels hold their own copy (csrc/mifc_device.h).
#ifndef MI_FIELDCALC_METCONSTANTS_H
#define MI_FIELDCALC_METCONSTANTS_H

namespace miutil {
namespace constants {

const float r = 287., cp = 1004., p0 = 1000., t0 = 273.15;
const float eps = 0.622;
const float xlh = 2.501e+6;
const float rcp = r / cp, cplr = xlh / rcp, exl = eps * xlh;
const float p0inv = 1. / p0;
const float kappa = r / cp;
const float g = 9.8;
const float ginv = 1. / g;
const float rhmin = 0.02, rhmax = 1.00;

// e_sat over water at -100, -95, ... +100 degrees Celsius
const int N_EWT = 41;
extern const float ewt[N_EWT];

// Linear interpolation in ewt[] and its inverse (host-side helper; the
// operators evaluate the same arithmetic on the GPU).
class ewt_calculator
{
public:
  ewt_calculator(float t_celsius);
  bool defined() const { return l >= 0 && l < N_EWT - 1; }
  bool defined(bool& allDefined, float undef, float& out) const
  {
    if (defined())
      return true;
    allDefined = false;
    out = undef;
    return false;
  }
  float value() const { return ewt[l] + (ewt[l + 1] - ewt[l]) * (x - l); }
  float inverse(float et) const;

private:
  float x;
  int l;
};

} // namespace constants
} // namespace miutil

#endif // MI_FIELDCALC_METCONSTANTS_H
