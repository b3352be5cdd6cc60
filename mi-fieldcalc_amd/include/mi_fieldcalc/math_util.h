// Small numeric helpers of the operator API.
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

template <typename T>
inline void sort2(T& a, T& b) { if (b < a) std::swap(a, b); }

template <typename T1, typename T2>
inline void minimize(T1& a, const T2& b) { if (b < a) a = b; }

template <typename T1, typename T2>
inline void maximize(T1& a, const T2& b) { if (b > a) a = b; }

template <typename T1, typename T2>
inline void minimaximize(T1& mi, T1& ma, const T2& b)
{
  maximize(ma, b);
  minimize(mi, b);
}

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

// 10^t, evaluated in double like the reference's std::pow(10, t) (math_util.h:121-125)
template <typename T>
inline T pow10(T t) { return std::pow(10, t); }

} // namespace miutil

#endif // MI_FIELDCALC_MATH_UTIL_H
