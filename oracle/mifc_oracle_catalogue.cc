/*
 * mifc_oracle_catalogue.cc -- TEST INFRASTRUCTURE, not product code.
 *
 * Second translation unit of the CPU restatement: the rest of the pointwise
 * catalogue (theta-e, ducting, cvtemp, abshum, the stability
 * indices, field algebra, and the other pointwise operators the reference's
 * Python module exposes) and the ensemble reductions (8f-4).  Same rules as
 * mifc_oracle.cc: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product never does.
 *
 * Parity status: PINNED -- bit-identical to the compiled reference
 * (oracle/_ref/libmifc_ref.so) on the seeded sweep of
 * tests/test_oracle_vs_ref.py and on the golden vectors under tests/golden/.
 *
 * Promotion rules as in mifc_oracle.cc: a bare double literal makes its
 * sub-expression double; unqualified sqrt()/exp() of a float are the C double
 * functions (the reference has no `using namespace std`), std::log10 & co.
 * passed as float(*)(float) are the float functions.
 */
#define MIFC_ORACLE_PREFIX mifcorc_
#include "oracle_abi.h"

#include "oracle_common.h"

#include <algorithm>

namespace {

// MetConstants.h:46 (rcp, cplr, exl) and :53 (ms2knots)
const float K_RCP = K_R / K_CP, K_CPLR = K_XLH / K_RCP, K_EXL = K_EPS * K_XLH;
const double K_MS2KNOTS = 3600.0 / 1852.0;

inline float ms2knots(float ff) // MetConstants.h:132-135
{
  return (float)((double)ff * K_MS2KNOTS);
}

inline float absval(float x, float y) // math_util.h:57-60
{
  return std::sqrt(x * x + y * y);
}

// FieldCalculations.cc:269-278
inline bool tk_rh_the(float tk, float rh, float thconv, float& out)
{
  Ewt e(tk - K_T0);
  if (!e.ok())
    return false;
  out = tk * thconv + e.value() * rh;
  return true;
}
// :280-283
inline float tk_q_duct(float tk, float q, float p)
{
  return (float)(77.6 * (double)(p / tk) + 373000. * (double)(q * p) / (double)(K_EPS * tk * tk));
}
// :285-296
inline bool tk_rh_duct(float tk, float q, float p, float& out)
{
  Ewt e(tk - K_T0);
  if (!e.ok())
    return false;
  const float et = e.value();
  const float rh = clamp_rh((float)((double)q * 0.01));
  out = (float)(77.6 * (double)(p / tk) + 373000. * (double)rh * (double)et / (double)(tk * tk));
  return true;
}

// :76-82
inline int fill_undef(int n, float* fres, int* fdefined, float undef)
{
  *fdefined = NONE_DEFINED;
  std::fill(fres, fres + n, undef);
  return 1;
}

// Drivers shaped like the reference's helper templates (:94-179).  The plain
// ones leave the flag alone; the "Undef" ones count and classify.
template <class F>
inline int unary_plain(const F& f, int n, const float* a, float* r, const int* fdefined, float undef)
{
  const bool all = *fdefined == ALL_DEFINED;
  for (int i = 0; i < n; ++i)
    r[i] = (all || defined1(a[i], undef)) ? f(a[i]) : undef;
  return 1;
}
template <class F>
inline int binary_plain(const F& f, int n, const float* a, const float* b, float* r, const int* fdefined, float undef)
{
  const bool all = *fdefined == ALL_DEFINED;
  for (int i = 0; i < n; ++i)
    r[i] = (all || (defined1(a[i], undef) && defined1(b[i], undef))) ? f(a[i], b[i]) : undef;
  return 1;
}
// f(a, out) / f(a, b, out) return false for an undefined result
template <class F>
inline int unary_counting(const F& f, int n, const float* a, float* r, int* fdefined, float undef)
{
  const bool all = *fdefined == ALL_DEFINED;
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    float o;
    if ((all || defined1(a[i], undef)) && f(a[i], o)) {
      r[i] = o;
    } else {
      r[i] = undef;
      bad += 1;
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}
template <class F>
inline int binary_counting(const F& f, int n, const float* a, const float* b, float* r, int* fdefined, float undef)
{
  const bool all = *fdefined == ALL_DEFINED;
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    float o;
    if ((all || (defined1(a[i], undef) && defined1(b[i], undef))) && f(a[i], b[i], o)) {
      r[i] = o;
    } else {
      r[i] = undef;
      bad += 1;
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}

inline bool divide_undef(float a, float b, float& o) // :84-92
{
  if (b != 0) {
    o = a / b;
    return true;
  }
  return false;
}

// MetConstants.h:88-90
const int K_NLEVELTABLE = 16;
const float K_PLEVELTABLE[K_NLEVELTABLE] = {1000, 925, 850, 800, 700, 500, 400, 300, 250, 200, 150, 100, 70, 50, 30, 10};
const float K_FLEVELTABLE[K_NLEVELTABLE] = {5, 25, 50, 65, 100, 185, 235, 300, 340, 385, 445, 530, 605, 675, 780, 1020};

} // namespace

extern "C" {

// ------------------------------------------------------------- theta-e
// FieldCalculations.cc:369-398
int mifcorc_plevelthe(int nx, int ny, const float* t, const float* rh, float p, int compute, float* the, int* fdefined, float undef)
{
  if (compute != 1 && compute != 2)
    return 0;
  if (p <= 0.0)
    return 0;
  const float pidcp = pidcp_of(p), pi = pidcp * K_CP;
  const float cvrh = (float)(0.01 * (double)(K_XLH / pi) * (double)K_EPS / (double)p);
  const float tconv = (compute == 2) ? pidcp : 1;
  const float thconv = 1 / pidcp;
  return binary_counting([=](float tt, float r, float& o) { return tk_rh_the(tt * tconv, r * cvrh, thconv, o); }, nx * ny, t, rh, the, fdefined, undef);
}

// :1100-1143 (a defined cell stays unwritten for compute outside 1, 2) and :1355-1392
static int xlevelthe(int n, const float* t, const float* q, const float* ps, bool hybrid, float alevel, float blevel, int compute, float* the, int* fdefined,
                     float undef)
{
  const bool all = *fdefined == ALL_DEFINED;
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    if (all || (defined1(t[i], undef) && defined1(q[i], undef) && defined1(ps[i], undef))) {
      const float p = hybrid ? alevel + blevel * ps[i] : ps[i];
      const float pi = pi_of(p);
      if (compute == 1)
        the[i] = (t[i] * K_CP + q[i] * K_XLH) / pi;
      else if (compute == 2)
        the[i] = t[i] + q[i] * K_XLH / pi;
    } else {
      the[i] = undef;
      bad += 1;
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}
int mifcorc_hlevelthe(int nx, int ny, const float* t, const float* q, const float* ps, float alevel, float blevel, int compute, float* the, int* fdefined,
                      float undef)
{
  if (bad_hlevel(alevel, blevel))
    return 0;
  return xlevelthe(nx * ny, t, q, ps, true, alevel, blevel, compute, the, fdefined, undef);
}
int mifcorc_alevelthe(int nx, int ny, const float* t, const float* q, const float* p, int compute, float* the, int* fdefined, float undef)
{
  if (compute != 1 && compute != 2)
    return 0;
  return xlevelthe(nx * ny, t, q, p, false, 0, 0, compute, the, fdefined, undef);
}

// ------------------------------------------------------------- ducting
// :597-636
int mifcorc_plevelducting(int nx, int ny, const float* t, const float* h, float p, int compute, float* duct, int* fdefined, float undef)
{
  if (p <= 0)
    return 0;
  const float tconv = (compute % 2 == 0) ? pidcp_of(p) : 1;
  if (compute == 1 || compute == 2)
    return binary_plain([=](float tt, float hh) { return tk_q_duct(tt * tconv, hh, p); }, nx * ny, t, h, duct, fdefined, undef);
  if (compute == 3 || compute == 4)
    return binary_counting([=](float tt, float hh, float& o) { return tk_rh_duct(tt * tconv, hh, p, o); }, nx * ny, t, h, duct, fdefined, undef);
  return 0;
}

// :1219-1274 (counts, updates the flag) and :1460-1505 (neither)
static void xlevelducting(int n, const float* t, const float* h, const float* ps, bool hybrid, float alevel, float blevel, int compute, float* duct,
                          float undef, bool all, size_t& bad)
{
  for (int i = 0; i < n; ++i) {
    if (all || (defined1(t[i], undef) && defined1(h[i], undef) && defined1(ps[i], undef))) {
      const float p = hybrid ? alevel + blevel * ps[i] : ps[i];
      float tk = t[i];
      if (compute % 2 == 0)
        tk *= pidcp_of(p);
      if (compute == 1 || compute == 2) {
        duct[i] = tk_q_duct(tk, h[i], p);
      } else if (compute == 3 || compute == 4) {
        float o;
        if (tk_rh_duct(tk, h[i], p, o)) {
          duct[i] = o;
        } else {
          duct[i] = undef;
          bad += 1;
        }
      }
    } else {
      duct[i] = undef;
      bad += 1;
    }
  }
}
int mifcorc_hlevelducting(int nx, int ny, const float* t, const float* h, const float* ps, float alevel, float blevel, int compute, float* duct,
                          int* fdefined, float undef)
{
  if (bad_hlevel(alevel, blevel))
    return 0;
  size_t bad = 0;
  xlevelducting(nx * ny, t, h, ps, true, alevel, blevel, compute, duct, undef, *fdefined == ALL_DEFINED, bad);
  *fdefined = classify(bad, nx * ny);
  return 1;
}
int mifcorc_alevelducting(int nx, int ny, const float* t, const float* h, const float* p, int compute, float* duct, int* fdefined, float undef)
{
  size_t bad = 0; // :1488 counted but never used: the flag stays as it came in
  xlevelducting(nx * ny, t, h, p, false, 0, 0, compute, duct, undef, *fdefined == ALL_DEFINED, bad);
  return 1;
}

// :1276-1304
int mifcorc_hlevelpressure(int nx, int ny, const float* ps, float alevel, float blevel, float* p, int* fdefined, float undef)
{
  if (bad_hlevel(alevel, blevel))
    return 0;
  return unary_counting(
      [=](float s, float& o) {
        o = alevel + blevel * s;
        return true;
      },
      nx * ny, ps, p, fdefined, undef);
}

// :466-503
int mifcorc_pleveldz2tmean(int nx, int ny, const float* z1, const float* z2, float p1, float p2, int compute, float* tmean, int* fdefined, float undef)
{
  if (p1 <= 0 || p2 <= 0 || p1 == p2)
    return 0;
  const float pi1 = pi_of(p1), pi2 = pi_of(p2);
  float convert, tconvert;
  switch (compute) {
  case 1:
    convert = (float)((double)K_G * 0.5 * (double)(pi1 + pi2) / (double)((pi2 - pi1) * K_CP));
    tconvert = -K_T0;
    break;
  case 2:
    convert = (float)((double)K_G * 0.5 * (double)(pi1 + pi2) / (double)((pi2 - pi1) * K_CP));
    tconvert = 0.;
    break;
  case 3:
    convert = K_G / (pi2 - pi1);
    tconvert = 0.;
    break;
  default:
    return 0;
  }
  return binary_plain([=](float a, float b) { return (a - b) * convert + tconvert; }, nx * ny, z1, z2, tmean, fdefined, undef);
}

// ------------------------------------------------------------- indices
// :745-814
int mifcorc_kIndex(int nx, int ny, const float* t500, const float* t700, const float* rh700, const float* t850, const float* rh850, float p500,
                   float p700, float p850, int compute, float* kfield, int* fdefined, float undef)
{
  if (p500 <= 0.0 || p500 >= p700 || p700 >= p850)
    return 0;
  float cvt500, cvt700, cvt850;
  switch (compute) {
  case 1:
    cvt500 = cvt700 = cvt850 = 1.;
    break;
  case 2:
    cvt500 = pidcp_of(p500);
    cvt700 = pidcp_of(p700);
    cvt850 = pidcp_of(p850);
    break;
  default:
    return 0;
  }
  const int n = nx * ny;
  const bool all = *fdefined == ALL_DEFINED;
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    bool ok = all || (defined1(t500[i], undef) && defined1(t700[i], undef) && defined1(rh700[i], undef) && defined1(t850[i], undef) &&
                      defined1(rh850[i], undef));
    if (ok) {
      const float rh8 = clamp_rh((float)(0.01 * (double)rh850[i]));
      const float tc850 = cvt850 * t850[i] - K_T0;
      const float tc700 = cvt700 * t700[i] - K_T0;
      const Ewt e850(tc850), e700(tc700);
      if (!(e850.ok() && e700.ok())) {
        ok = false;
      } else {
        const float tdc850 = e850.inverse(e850.value() * rh8);
        const float rh7 = clamp_rh((float)(0.01 * (double)rh700[i]));
        const float tdc700 = e700.inverse(e700.value() * rh7);
        const float tc500 = cvt500 * t500[i] - K_T0;
        kfield[i] = (tc850 + tdc850) - (tc700 - tdc700) - tc500;
      }
    }
    if (!ok) {
      kfield[i] = undef;
      bad += 1;
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}

// :816-870
int mifcorc_ductingIndex(int nx, int ny, const float* t850, const float* rh850, float p850, int compute, float* duct, int* fdefined, float undef)
{
  const float bduct = 3.8e+5;
  if (p850 <= 0.0)
    return 0;
  float tconvert;
  switch (compute) {
  case 1:
    tconvert = 1.;
    break;
  case 2:
    tconvert = pidcp_of(p850);
    break;
  default:
    return 0;
  }
  return binary_counting(
      [=](float t, float r, float& o) {
        const float rh = clamp_rh((float)(0.01 * (double)r));
        const float tk = t * tconvert;
        const Ewt e(tk - K_T0);
        if (!e.ok())
          return false;
        const float et = e.value();
        const float etd = et * rh;
        const float tdk = e.inverse(etd) + K_T0;
        o = bduct * (et / (tk * tk) - etd / (tdk * tdk));
        return true;
      },
      nx * ny, t850, rh850, duct, fdefined, undef);
}

// :872-971.  A cell with an undefined input is counted but NOT written (:965-967).
int mifcorc_showalterIndex(int nx, int ny, const float* t500, const float* t850, const float* rh850, float p500, float p850, int compute, float* sfield,
                           int* fdefined, float undef)
{
  if (p500 <= 0.0 || p500 >= p850)
    return 0;
  const float pi500 = pi_of(p500), pi850 = pi_of(p850);
  float cvt500, cvt850, dryadiabat;
  switch (compute) {
  case 1:
    cvt500 = 1.;
    cvt850 = 1.;
    dryadiabat = K_CP * (K_CP / pi850) * (pi500 / K_CP);
    break;
  case 2:
    cvt500 = pi500 / K_CP;
    cvt850 = pi850 / K_CP;
    dryadiabat = K_CP * (pi500 / K_CP);
    break;
  default:
    return 0;
  }
  const int niter = 7;
  const int n = nx * ny;
  const bool all = *fdefined == ALL_DEFINED;
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    if (all || (defined1(t500[i], undef) && defined1(t850[i], undef) && defined1(rh850[i], undef))) {
      const float tk500 = cvt500 * t500[i];
      const float tk850 = cvt850 * t850[i];
      const float rh = clamp_rh((float)(0.01 * (double)rh850[i]));
      const Ewt e(tk850 - K_T0);
      if (!e.ok()) {
        sfield[i] = undef;
        bad += 1;
      } else {
        const float etd = e.value() * rh;
        float tcl = dryadiabat * t850[i];
        float qcl = K_EPS * etd / p850;
        for (int it = 0; it < niter; ++it) {
          const Ewt e2(tcl / K_CP - K_T0);
          if (!e2.ok())
            break;
          const float esat = e2.value();
          const float qsat = K_EPS * esat / p500;
          float dq = qcl - qsat;
          const float a1 = K_CPLR * qcl / tcl;
          const float a2 = K_EXL / tcl;
          dq = (float)((double)dq / (1. + (double)(a1 * a2)));
          qcl = qcl - dq;
          tcl = tcl + dq * K_XLH;
        }
        const float tx500 = tcl / K_CP;
        sfield[i] = tk500 - tx500;
      }
    } else {
      bad += 1;
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}

// :973-1014
int mifcorc_boydenIndex(int nx, int ny, const float* t700, const float* z700, const float* z1000, float p700, float p1000, int compute, float* bfield,
                        int* fdefined, float undef)
{
  if (compute <= 0 || compute >= 3)
    return 0;
  if (p700 <= 0.0 || p700 >= p1000)
    return 0;
  const float pi700 = K_CP * powf(p700 / K_P0, K_R / K_CP);
  const float tconv = (compute == 2) ? pi700 / K_CP : 1;
  const int n = nx * ny;
  const bool all = *fdefined == ALL_DEFINED;
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    if (all || (defined1(t700[i], undef) && defined1(z700[i], undef) && defined1(z1000[i], undef))) {
      const float tc700 = t700[i] * tconv - K_T0;
      bfield[i] = (float)((double)(z700[i] - z1000[i]) / 10. - (double)tc700 - 200.);
    } else {
      bfield[i] = undef;
      bad += 1;
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}

// :1016-1040
int mifcorc_sweatIndex(int nx, int ny, const float* t850, const float* t500, const float* td850, const float* td500, const float* u850, const float* v850,
                       const float* u500, const float* v500, float* sindex, int* fdefined, float undef)
{
  const int n = nx * ny;
  const bool all = *fdefined == ALL_DEFINED;
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    if (all || (defined1(t850[i], undef) && defined1(t500[i], undef) && defined1(td850[i], undef) && defined1(td500[i], undef) &&
                defined1(u850[i], undef) && defined1(v850[i], undef) && defined1(u500[i], undef) && defined1(v500[i], undef))) {
      const float ff850 = absval(u850[i], v850[i]);
      const float ff500 = absval(u500[i], v500[i]);
      const float sind = (u500[i] * v850[i] - v500[i] * u850[i]) / (ff850 * ff500);
      const float acc = 32 * td850[i] + 20 * t850[i] - 40 * t500[i] - 20 * 49 + 2 * ms2knots(ff850) + ms2knots(ff500);
      sindex[i] = (float)((double)acc + 125 * ((double)sind + 0.2));
    } else {
      sindex[i] = undef;
      bad += 1;
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}

// ------------------------------------------------------------- misc pointwise
// :1555-1602
int mifcorc_seaSoundSpeed(int nx, int ny, const float* t, const float* s, float z_, int compute, float* soundspeed, int* fdefined, float undef)
{
  if (compute != 1 && compute != 2)
    return 0;
  const float tconv = (compute == 1) ? 0 : K_T0;
  const double Z = fabsf(z_);
  const double Cz = 0.01635 * Z + 0.000000175 * Z * Z;
  return binary_counting(
      [=](float tt, float S, float& o) {
        const float T = tt - tconv;
        const double Ct = 4.565 * T - 0.0517 * T * T + 0.000221 * T * T * T;
        const double Cs = (1.338 - 0.013 * T + 0.0001 * T * T) * (S - 35.0);
        o = float(1449.1 + Ct + Cs + Cz);
        return true;
      },
      nx * ny, t, s, soundspeed, fdefined, undef);
}

// :1608-1674
int mifcorc_cvtemp(int nx, int ny, const float* tinp, int compute, float* tout, int* fdefined, float undef)
{
  const int n = nx * ny;
  float tconvert;
  switch (compute) {
  case 1:
  case 3:
    tconvert = -K_T0;
    break;
  case 2:
  case 4:
    tconvert = +K_T0;
    break;
  default:
    return 0;
  }
  const bool all = *fdefined == ALL_DEFINED;
  if (compute == 3 || compute == 4) {
    float tavg = 0.;
    int navg = 0;
    for (int i = 0; i < n; ++i) {
      if (all || defined1(tinp[i], undef)) {
        tavg += tinp[i];
        navg += 1;
      }
    }
    if (navg > 0)
      tavg /= float(navg);
    if ((compute == 3 && tavg < K_T0 / 2.) || (compute == 4 && tavg > K_T0 / 2.)) {
      if (tout != tinp)
        for (int i = 0; i < n; ++i)
          tout[i] = tinp[i];
      return 1; // flag untouched (:1658)
    }
  }
  return unary_counting(
      [=](float t, float& o) {
        o = t + tconvert;
        return true;
      },
      n, tinp, tout, fdefined, undef);
}

// :1676-1736
int mifcorc_abshum(int nx, int ny, const float* t, const float* rhum, float* abshumout, int* fdefined, float undef)
{
  const float C = 2.16679;
  const float C1 = -7.85951783, C2 = 1.84408259, C3 = -11.7866497, C4 = 22.6807411, C5 = -15.9618719, C6 = 1.80122502;
  const float Tc = 647.096;
  const float Pc = 220640;
  return binary_counting(
      [=](float tt, float rh, float& o) {
        const float v = 1 - tt / Tc, tii = 1 / tt;
        const float v2 = v * v, v3 = v * v2, v4 = v2 * v2, v1_5 = (float)((double)v * sqrt((double)v)), v3_5 = v2 * v1_5, v7_5 = v4 * v3_5;
        const float Pws = (float)((double)Pc * exp((double)(Tc * tii * (C1 * v + C2 * v1_5 + C3 * v3 + C4 * v3_5 + C5 * v4 + C6 * v7_5))));
        const float Pw = Pws * rh;
        o = C * Pw * 100 * tii;
        return true;
      },
      nx * ny, t, rhum, abshumout, fdefined, undef);
}

// :2181-2229 (counts but never updates the flag)
int mifcorc_windCooling(int nx, int ny, const float* t, const float* u, const float* v, int compute, float* dtcool, int* fdefined, float undef)
{
  if (compute != 1 && compute != 2)
    return 0;
  const float tconv = (compute == 1) ? K_T0 : 0.f;
  const int n = nx * ny;
  const bool all = *fdefined == ALL_DEFINED;
  for (int i = 0; i < n; ++i) {
    if (all || (defined1(t[i], undef) && defined1(u[i], undef) && defined1(v[i], undef))) {
      const float tc = t[i] - tconv;
      const float ff = (float)((double)absval(u[i], v[i]) * 3.6);
      const float ffpow = powf(ff, 0.16);
      float d = (float)(13.12 + 0.6215 * (double)tc - 11.37 * (double)ffpow + 0.3965 * (double)tc * (double)ffpow);
      if (d > 0.)
        d = 0.;
      dtcool[i] = d;
    } else {
      dtcool[i] = undef;
    }
  }
  return 1;
}

// :2231-2264
int mifcorc_underCooledRain(int nx, int ny, const float* precip, const float* snow, const float* tk, float precipMin, float snowRateMax, float tcMax,
                            float* undercooled, int* fdefined, float undef)
{
  const int n = nx * ny;
  const float tkMax = tcMax + K_T0;
  const bool all = *fdefined == ALL_DEFINED;
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    if (all || (defined1(precip[i], undef) && defined1(snow[i], undef) && defined1(tk[i], undef))) {
      undercooled[i] = (precip[i] >= precipMin && tk[i] <= tkMax && snow[i] <= precip[i] * snowRateMax) ? 1. : 0.;
    } else {
      undercooled[i] = undef;
      bad += 1;
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}

// :2311-2349
int mifcorc_pressure2FlightLevel(int nx, int ny, const float* pressure, float* flightlevel, int* fdefined, float undef)
{
  const int nTab = K_NLEVELTABLE - 1;
  return unary_counting(
      [=](float p, float& o) {
        if (p > K_PLEVELTABLE[0])
          p = K_PLEVELTABLE[0];
        if (p < K_PLEVELTABLE[nTab])
          p = K_PLEVELTABLE[nTab];
        int k = 1;
        while (k < nTab && K_PLEVELTABLE[k] > p)
          k++;
        const float ratio = (p - K_PLEVELTABLE[k - 1]) / (K_PLEVELTABLE[k] - K_PLEVELTABLE[k - 1]);
        o = K_FLEVELTABLE[k - 1] + (K_FLEVELTABLE[k] - K_FLEVELTABLE[k - 1]) * ratio;
        return true;
      },
      nx * ny, pressure, flightlevel, fdefined, undef);
}

// :3063-3118
int mifcorc_snow_in_cm(int nx, int ny, const float* snow_water, const float* tk2m, const float* td2m, float* snow_cm, int* fdefined, float undef)
{
  const int n = nx * ny;
  const bool all = *fdefined == ALL_DEFINED;
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    if (all || (defined1(snow_water[i], undef) && defined1(tk2m[i], undef) && defined1(td2m[i], undef))) {
      if (snow_water[i] <= 0.) {
        snow_cm[i] = 0.;
        continue;
      }
      const float t = (float)((double)(tk2m[i] + td2m[i]) / 2.);
      const float logit_t = (float)((1 - exp(((double)t - 274.3) * 3.5)) / (1 + exp(((double)t - 274.3) * 3.5)));
      const float mm2cm_t = (float)(0.13 / (0.02 + 0.1 * (((double)t - 252.0) / 20.0) * (((double)t - 252.0) / 20.0)));
      const float fac = logit_t * mm2cm_t;
      snow_cm[i] = (fac <= 1.) ? snow_water[i] : snow_water[i] * fac;
    } else {
      snow_cm[i] = undef;
      bad += 1;
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}

// :2462-2499
int mifcorc_values2classes(int nx, int ny, const float* fvalue, float* fclass, const float* values, int nvalues_total, int* fdefined, float undef)
{
  if (nvalues_total < 2)
    return 0;
  const int nvalues = nvalues_total - 2;
  const float fmin = values[0], fmax = values[nvalues + 1];
  return unary_counting(
      [=](float f, float& o) {
        if (!(f >= fmin && f < fmax))
          return false;
        int j = 1;
        while (j < nvalues && values[j] < f)
          j++;
        o = float(j - 1);
        return true;
      },
      nx * ny, fvalue, fclass, fdefined, undef);
}

// ------------------------------------------------------------- vessel icing, the two closed-form models
// FieldCalculationsVesselIcing.cc:77-112 (Overland 1990) and :114-180 (Mertins 1968); temperatures in Celsius
static int vessel_icing_simple(bool mertins, int nx, int ny, const float* airtemp, const float* seatemp, const float* u, const float* v,
                               const float* sal, const float* aice, float* icing, int* fdefined, float undef)
{
  const int n = nx * ny;
  const double A = 2.73e-2, B = 2.91e-4, C = 1.84e-6;
  const bool all = *fdefined == ALL_DEFINED;
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    const bool def = all || (defined1(airtemp[i], undef) && defined1(seatemp[i], undef) && defined1(u[i], undef) && defined1(v[i], undef) &&
                             defined1(sal[i], undef) && defined1(aice[i], undef));
    bool ok = def && aice[i] < 0.4;
    if (ok) {
      const double Tf = (-0.002 - 0.0524 * (double)sal[i]) - 6.0E-5 * (double)(sal[i] * sal[i]); // freezing point of sea water
      if (seatemp[i] < Tf) {
        ok = false;
      } else {
        const double ff = absval(u[i], v[i]);
        if (!mertins) {
          const double ppr = ff * (Tf - (double)airtemp[i]) / (1 + 0.3 * ((double)seatemp[i] - Tf));
          icing[i] = (float)(A * ppr + B * (ppr * ppr) + C * ppr * ppr * ppr);
        } else {
          const double temperature = airtemp[i], sst = seatemp[i];
          float r = 0;
          if (ff >= 10.8) {
            double temp1, temp2, temp3;
            if (ff < 17.2) {
              temp1 = -1.15 * sst - 4.3;
              temp2 = -1.5 * sst - 10;
              temp3 = -10000;
            } else if (ff < 20.8) {
              temp1 = -0.6 * sst - 3.2;
              temp2 = -1.05 * sst - 5.6;
              temp3 = -1.75 * sst - 12.5;
            } else if (ff < 28.5) {
              temp1 = -0.3 * sst - 2.6;
              temp2 = -0.66 * sst - 3.32;
              temp3 = -1.325 * sst - 7.651;
            } else {
              temp1 = -0.14 * sst - 2.28;
              temp2 = -0.3 * sst - 2.6;
              temp3 = -1.16 * sst - 5.22;
            }
            if (temperature > -2)
              r = 0;
            else if (temperature > temp1)
              r = (float)0.8333;
            else if (temperature > temp2)
              r = (float)2.0833;
            else
              r = (temperature <= temp3 || ff < 17.2) ? (float)4.375 : (float)6.25;
          }
          icing[i] = r;
        }
      }
    }
    if (!ok) {
      icing[i] = undef;
      bad += 1;
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}
int mifcorc_vesselIcingOverland(int nx, int ny, const float* airtemp, const float* seatemp, const float* u, const float* v, const float* sal,
                                const float* aice, float* icing, int* fdefined, float undef)
{
  return vessel_icing_simple(false, nx, ny, airtemp, seatemp, u, v, sal, aice, icing, fdefined, undef);
}
int mifcorc_vesselIcingMertins(int nx, int ny, const float* airtemp, const float* seatemp, const float* u, const float* v, const float* sal,
                               const float* aice, float* icing, int* fdefined, float undef)
{
  return vessel_icing_simple(true, nx, ny, airtemp, seatemp, u, v, sal, aice, icing, fdefined, undef);
}

// ------------------------------------------------------------- second-order Shapiro filter
// FieldCalculations.cc:2076-2179.  Two sweeps (x then y) with weight +0.25, two with -0.25 --
// for ALL_DEFINED input.  Otherwise the per-cell weights s1 / s2 are taken ONCE from the
// unsmoothed field with s = +0.25 (:2141-2145) and reused by both sweeps, i.e. the second
// sweep smooths again instead of restoring (the assignment s = -0.25 at :2167 never reaches
// them).  ALL branch: `2. * f` makes the update double; the other branch is float throughout.
// The flag becomes ALL_DEFINED whatever the content (:2176).  field == fsmooth is allowed.
int mifcorc_shapiro2_filter(int nx, int ny, const float* field, float* fsmooth, int* fdefined, float undef)
{
  const int n = nx * ny;
  if (nx < 3 || ny < 3)
    return 0;
  float* f1 = fsmooth;
  if (field != fsmooth)
    for (int i = 0; i < n; ++i)
      fsmooth[i] = field[i];
  float* f2 = new float[n];
  const bool all = *fdefined == ALL_DEFINED;
  float* s1 = nullptr;
  float* s2 = nullptr;
  if (!all) {
    s1 = new float[n];
    s2 = new float[n];
    for (int i = 1; i < n - 1; ++i)
      s1[i] = (defined1(f1[i - 1], undef) && defined1(f1[i], undef) && defined1(f1[i + 1], undef)) ? 0.25f : 0.f;
    for (int i = nx; i < n - nx; ++i)
      s2[i] = (defined1(f1[i - nx], undef) && defined1(f1[i], undef) && defined1(f1[i + nx], undef)) ? 0.25f : 0.f;
  }
  float s = 0.25;
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 1; i < n - 1; ++i) {
      if (all)
        f2[i] = (float)((double)f1[i] + (double)s * ((double)(f1[i - 1] + f1[i + 1]) - 2. * (double)f1[i]));
      else
        f2[i] = f1[i] + s1[i] * (f1[i - 1] + f1[i + 1] - 2 * f1[i]);
    }
    for (int j = 0; j < ny; ++j) {
      f2[j * nx] = f1[j * nx];
      f2[j * nx + nx - 1] = f1[j * nx + nx - 1];
    }
    for (int i = nx; i < n - nx; ++i) {
      if (all)
        f1[i] = (float)((double)f2[i] + (double)s * ((double)(f2[i - nx] + f2[i + nx]) - 2. * (double)f2[i]));
      else
        f1[i] = f2[i] + s2[i] * (f2[i - nx] + f2[i + nx] - 2 * f2[i]);
    }
    for (int i = 0; i < nx; ++i) {
      f1[i] = f2[i];
      f1[n - nx + i] = f2[n - nx + i];
    }
    s = -0.25;
  }
  delete[] f2;
  delete[] s1;
  delete[] s2;
  *fdefined = ALL_DEFINED;
  return 1;
}

// ------------------------------------------------------------- field algebra (:2501-2669)
int mifcorc_minvalueFields(int nx, int ny, const float* field1, const float* field2, float* fres, int* fdefined, float undef)
{
  return binary_plain([](float a, float b) { return std::min(a, b); }, nx * ny, field1, field2, fres, fdefined, undef);
}
int mifcorc_maxvalueFields(int nx, int ny, const float* field1, const float* field2, float* fres, int* fdefined, float undef)
{
  return binary_plain([](float a, float b) { return std::max(a, b); }, nx * ny, field1, field2, fres, fdefined, undef);
}
int mifcorc_minvalueFieldConst(int nx, int ny, const float* field1, float value, float* fres, int* fdefined, float undef)
{
  if (value == undef)
    return fill_undef(nx * ny, fres, fdefined, undef);
  return unary_plain([=](float a) { return std::min(a, value); }, nx * ny, field1, fres, fdefined, undef);
}
int mifcorc_maxvalueFieldConst(int nx, int ny, const float* field1, float value, float* fres, int* fdefined, float undef)
{
  if (value == undef)
    return fill_undef(nx * ny, fres, fdefined, undef);
  return unary_plain([=](float a) { return std::max(a, value); }, nx * ny, field1, fres, fdefined, undef);
}
int mifcorc_absvalueField(int nx, int ny, const float* field, float* fres, int* fdefined, float undef)
{
  return unary_plain([](float a) { return fabsf(a); }, nx * ny, field, fres, fdefined, undef);
}
int mifcorc_log10Field(int nx, int ny, const float* field, float* fres, int* fdefined, float undef)
{
  return unary_plain([](float a) { return log10f(a); }, nx * ny, field, fres, fdefined, undef);
}
int mifcorc_pow10Field(int nx, int ny, const float* field, float* fres, int* fdefined, float undef)
{
  return unary_plain([](float a) { return (float)pow(10.0, (double)a); }, nx * ny, field, fres, fdefined, undef); // math_util.h:121-125
}
int mifcorc_logField(int nx, int ny, const float* field, float* fres, int* fdefined, float undef)
{
  return unary_plain([](float a) { return logf(a); }, nx * ny, field, fres, fdefined, undef);
}
int mifcorc_expField(int nx, int ny, const float* field, float* fres, int* fdefined, float undef)
{
  return unary_plain([](float a) { return expf(a); }, nx * ny, field, fres, fdefined, undef);
}
int mifcorc_powerField(int nx, int ny, const float* field, float value, float* fres, int* fdefined, float undef)
{
  if (value == undef)
    return fill_undef(nx * ny, fres, fdefined, undef);
  return unary_plain([=](float a) { return powf(a, value); }, nx * ny, field, fres, fdefined, undef);
}
// :2565-2585
int mifcorc_replaceUndefined(int nx, int ny, const float* field, float value, float* fres, int* fdefined, float undef)
{
  const int n = nx * ny;
  if (value == undef || *fdefined == ALL_DEFINED) {
    if (fres != field)
      std::memcpy(fres, field, sizeof(float) * n);
    return 1;
  }
  if (*fdefined == NONE_DEFINED) {
    std::fill(fres, fres + n, value);
  } else {
    for (int i = 0; i < n; ++i)
      fres[i] = (field[i] == undef) ? value : field[i];
  }
  *fdefined = ALL_DEFINED;
  return 1;
}
// :2587-2608
int mifcorc_replaceDefined(int nx, int ny, const float* field, float value, float* fres, int* fdefined, float undef)
{
  const int n = nx * ny;
  if (value == undef || *fdefined == NONE_DEFINED) {
    std::fill(fres, fres + n, undef);
    *fdefined = NONE_DEFINED;
    return 1;
  }
  if (*fdefined == ALL_DEFINED) {
    std::fill(fres, fres + n, value);
  } else {
    for (int i = 0; i < n; ++i)
      fres[i] = (field[i] != undef) ? value : field[i];
  }
  *fdefined = ALL_DEFINED;
  return 1;
}
// :2611-2625
int mifcorc_fieldOPERfield(int compute, int nx, int ny, const float* field1, const float* field2, float* fres, int* fdefined, float undef)
{
  const int n = nx * ny;
  switch (compute) {
  case 1:
    return binary_plain([](float a, float b) { return a + b; }, n, field1, field2, fres, fdefined, undef);
  case 2:
    return binary_plain([](float a, float b) { return a - b; }, n, field1, field2, fres, fdefined, undef);
  case 3:
    return binary_plain([](float a, float b) { return a * b; }, n, field1, field2, fres, fdefined, undef);
  case 4:
    return binary_counting(divide_undef, n, field1, field2, fres, fdefined, undef);
  default:
    return 0;
  }
}
// :2627-2645
int mifcorc_fieldOPERconstant(int compute, int nx, int ny, const float* field, float value, float* fres, int* fdefined, float undef)
{
  const int n = nx * ny;
  if ((value == undef) || (compute == 4 && value == 0))
    return fill_undef(n, fres, fdefined, undef);
  switch (compute) {
  case 1:
    return unary_plain([=](float f) { return f + value; }, n, field, fres, fdefined, undef);
  case 2:
    return unary_plain([=](float f) { return f - value; }, n, field, fres, fdefined, undef);
  case 3:
    return unary_plain([=](float f) { return f * value; }, n, field, fres, fdefined, undef);
  case 4:
    return unary_plain([=](float f) { return f / value; }, n, field, fres, fdefined, undef);
  default:
    return 0;
  }
}
// :2647-2669
int mifcorc_constantOPERfield(int compute, int nx, int ny, float value, const float* field, float* fres, int* fdefined, float undef)
{
  const int n = nx * ny;
  if (value == undef)
    return fill_undef(n, fres, fdefined, undef);
  switch (compute) {
  case 1:
    return unary_plain([=](float f) { return value + f; }, n, field, fres, fdefined, undef);
  case 2:
    return unary_plain([=](float f) { return value - f; }, n, field, fres, fdefined, undef);
  case 3:
    return unary_plain([=](float f) { return value * f; }, n, field, fres, fdefined, undef);
  case 4:
    return unary_counting([=](float f, float& o) { return divide_undef(value, f, o); }, n, field, fres, fdefined, undef);
  default:
    return 0;
  }
}

// ------------------------------------------------------------- ensemble reductions (:2671-2860)
int mifcorc_sumFields(int nx, int ny, const float* const* fields, int nfields, float* fres, int* fdefined, float undef)
{
  const int n = nx * ny;
  const bool all = *fdefined == ALL_DEFINED;
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    fres[i] = 0;
    for (int j = 0; j < nfields; ++j) {
      if (all || defined1(fields[j][i], undef)) {
        fres[i] += fields[j][i];
      } else {
        fres[i] = undef;
        bad += 1;
        break;
      }
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}

int mifcorc_meanValue(int nx, int ny, const float* const* fields, const int* fdefined_in, int nfields, float* fres, int* fdefined_out, float undef)
{
  const int n = nx * ny;
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    fres[i] = 0;
    int ndef = 0;
    for (int j = 0; j < nfields; ++j) {
      if (fdefined_in[j] == ALL_DEFINED || defined1(fields[j][i], undef)) {
        ndef++;
        fres[i] += fields[j][i];
      }
    }
    if (ndef > 0) {
      fres[i] /= ndef;
    } else {
      fres[i] = undef;
      bad += 1;
    }
  }
  *fdefined_out = classify(bad, n);
  return 1;
}

int mifcorc_stddevValue(int nx, int ny, const float* const* fields, const int* fdefined_in, int nfields, float* fres, int* fdefined_out, float undef)
{
  const int n = nx * ny;
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    int cnt = 0;
    float m = 0, m2 = 0;
    for (int j = 0; j < nfields; ++j) {
      if (fdefined_in[j] == ALL_DEFINED || defined1(fields[j][i], undef)) {
        const float x = fields[j][i], delta = x - m;
        cnt += 1;
        m += delta / cnt;
        m2 += delta * (x - m);
      }
    }
    if (cnt > 0) {
      fres[i] = (float)sqrt((double)(m2 / cnt));
    } else {
      fres[i] = undef;
      bad += 1;
    }
  }
  *fdefined_out = classify(bad, n);
  return 1;
}

int mifcorc_extremeValue(int compute, int nx, int ny, const float* const* fields, int nfields, float* fres, int* fdefined, float undef)
{
  if (nfields == 0)
    return 0;
  const int n = nx * ny;
  const bool all = *fdefined == ALL_DEFINED;
  size_t bad = 0;
  if (compute == 1 || compute == 2) {
    for (int i = 0; i < n; ++i) {
      fres[i] = undef;
      for (int j = 0; j < nfields; ++j) {
        const float f = fields[j][i];
        if (fres[i] == undef || ((all || defined1(f, undef)) && ((compute == 1 && fres[i] < f) || (compute == 2 && fres[i] > f))))
          fres[i] = f;
      }
      if (fres[i] == undef)
        bad += 1;
    }
  } else if (compute == 3 || compute == 4) {
    for (int i = 0; i < n; ++i) {
      fres[i] = undef;
      float tmp = undef;
      for (int j = 0; j < nfields; ++j) {
        const float f = fields[j][i];
        if (tmp == undef || ((all || defined1(f, undef)) && ((compute == 3 && tmp < f) || (compute == 4 && tmp > f)))) {
          tmp = f;
          fres[i] = j;
        }
      }
      if (fres[i] == undef)
        bad += 1;
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}

int mifcorc_probability(int compute, int nx, int ny, const float* const* fields, const int* fdefined_in, int nfields, const float* limits, int nlimits,
                        float* fres, int* fdefined_out, float undef)
{
  const size_t n = (size_t)nx * ny;
  const bool check_between = (nlimits >= 2) && (compute == 3 || compute == 6);
  const bool check_above = (nlimits >= 1) && (compute == 1 || compute == 4 || check_between);
  const bool check_below = (nlimits >= 1) && (compute == 2 || compute == 5 || check_between);
  if (!(check_above || check_below)) {
    // the reference reads limits[0] before this test (:2824); an empty limits vector is the caller's error
    for (size_t i = 0; i < n; ++i)
      fres[i] = undef;
    *fdefined_out = NONE_DEFINED;
    return 0;
  }
  const float value_above = limits[0];
  const float value_below = check_between ? limits[1] : limits[0];
  size_t bad = 0;
  for (size_t i = 0; i < n; ++i) {
    fres[i] = 0;
    int ndef = 0;
    for (int j = 0; j < nfields; ++j) {
      if (fdefined_in[j] != NONE_DEFINED) {
        ndef += 1;
        const float value = fields[j][i];
        if ((value != undef) && (!check_above || value > value_above) && (!check_below || value < value_below))
          fres[i] += 1;
      }
    }
    if (ndef == 0) {
      fres[i] = undef;
      bad += 1;
    } else if (compute < 4) {
      fres[i] = (float)((double)fres[i] / (ndef / 100.0));
    }
  }
  *fdefined_out = classify(bad, n);
  return 1;
}

} // extern "C"
