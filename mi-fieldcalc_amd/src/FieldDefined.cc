// Host-side helpers of the undefined-value vocabulary
// (behaviour of the reference's src/mi_fieldcalc/FieldDefined.cc:34-87).
#include "mi_fieldcalc/FieldDefined.h"
#include "mi_fieldcalc/MetConstants.h"

namespace miutil {

const float UNDEF = 1.0e35f;

ValuesDefined checkDefined(const float* data, size_t n)
{
  // "defined" here means strictly below UNDEF (FieldDefined.cc:36-39); NaN
  // therefore counts as undefined.  Stops as soon as both kinds were seen.
  bool seen_defined = false, seen_undefined = false;
  for (size_t i = 0; i < n && !(seen_defined && seen_undefined); ++i) {
    if (data[i] < UNDEF)
      seen_defined = true;
    else
      seen_undefined = true;
  }
  if (seen_defined && seen_undefined)
    return SOME_DEFINED;
  return seen_defined ? ALL_DEFINED : NONE_DEFINED;
}

ValuesDefined checkDefined(size_t n_undefined, size_t n)
{
  if (n_undefined == 0)
    return ALL_DEFINED;
  return (n_undefined == n) ? NONE_DEFINED : SOME_DEFINED;
}

ValuesDefined combineDefined(ValuesDefined a, ValuesDefined b)
{
  if (a == ALL_DEFINED)
    return b;
  if (a == NONE_DEFINED)
    return NONE_DEFINED;
  return (b != ALL_DEFINED) ? b : SOME_DEFINED; // a == SOME_DEFINED
}

namespace constants {

const float ewt[N_EWT] = {.000034, .000089, .000220, .000517, .001155, .002472, .005080, .01005, .01921, .03553, .06356,
                          .1111,   .1891,   .3139,   .5088,   .8070,   1.2540,  1.9118,  2.8627, 4.2148, 6.1078, 8.7192,
                          12.272,  17.044,  23.373,  31.671,  42.430,  56.236,  73.777,  95.855, 123.40, 157.46, 199.26,
                          250.16,  311.69,  385.56,  473.67,  578.09,  701.13,  845.28,  1013.25};

ewt_calculator::ewt_calculator(float t_celsius)
    : x((t_celsius + 100.) * 0.2)
    , l(int(x))
{
}

float ewt_calculator::inverse(float et) const
{
  int ll = l;
  while (ll > 0 && ll < N_EWT - 1 && ewt[ll] > et)
    ll--;
  const float frac = (et - ewt[ll]) / (ewt[ll + 1] - ewt[ll]);
  return -100. + (float(ll) + frac) * 5.;
}

} // namespace constants
} // namespace miutil

const float fieldUndef = miutil::UNDEF;
