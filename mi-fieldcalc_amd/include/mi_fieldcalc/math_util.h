// Small numeric helpers of the operator API (subset used on the hot path).
// Source-compatible with the reference's src/mi_fieldcalc/math_util.h.
#ifndef MI_FIELDCALC_MATH_UTIL_H
#define MI_FIELDCALC_MATH_UTIL_H 1

#include <algorithm>
#include <cmath>

namespace miutil {

template <typename T>
inline T square(T x) { return x * x; }

// x*x + y*y and its square root, evaluated in T (math_util.h:48-60 of the
// reference; this is what defines the arithmetic of vectorabs)
template <typename T>
inline T absval2(T x, T y) { return square(x) + square(y); }

template <typename T>
inline T absval(T x, T y) { return std::sqrt(absval2(x, y)); }

template <typename T1, typename T2>
inline void minimize(T1& a, const T2& b) { if (b < a) a = b; }

template <typename T1, typename T2>
inline void maximize(T1& a, const T2& b) { if (b > a) a = b; }

template <typename T1>
inline bool value_between(const T1& v, const T1& lim0, const T1& lim1)
{
  return (lim0 <= lim1) ? (lim0 <= v && v <= lim1) : (lim1 <= v && v <= lim0);
}

template <typename T1>
inline T1 constrain_value(const T1& v, const T1& lim0, const T1& lim1)
{
  const T1& lo = (lim0 <= lim1) ? lim0 : lim1;
  const T1& hi = (lim0 <= lim1) ? lim1 : lim0;
  return (v < lo) ? lo : ((hi < v) ? hi : v);
}

} // namespace miutil

#endif // MI_FIELDCALC_MATH_UTIL_H
