// mifc_pointwise.hip -- the rest of the pointwise catalogue on
// one templated elementwise kernel: up to 8 input fields, 16-byte coalesced
// loads (4 cells per lane), point function per cell, nontemporal 16-byte
// stores, per-wave undefined count -> one atomic per wave.  All HBM-bound
// (8..36 B per cell); each operator is its own instantiation, so the point
// function is straight-line code and register use is per operator.
//
// Reference lines are cited per point function (FieldCalculations.cc).  The
// promotion rules are those of mifc_device.h: a bare double literal makes its
// sub-expression double, rounded to float once on the store; no fused
// multiply-add (-ffp-contract=off).
#include "mifc_device.h"
#include "mifc_kernels.h"

namespace mifc {

namespace {

enum PwStatus {
  PW_OK = 0,
  PW_UNDEF = 1, // cell := undef, counted
  PW_SKIP = 2,  // cell left as it is, counted (showalterIndex :965-967)
  PW_KEEP = 3   // cell left as it is, not counted (compute outside the operator's range)
};

template <int OP>
struct PwTraits;
#define PW_TRAITS(OP, NIN, EWT, POW)     \
  template <>                            \
  struct PwTraits<OP>                    \
  {                                      \
    static const int nin = NIN;          \
    static const bool ewt = EWT;         \
    static const int pow = POW;          \
  }
PW_TRAITS(PW_PLEVELTHE, 2, true, 0);
PW_TRAITS(PW_XLEVELTHE, 3, false, 2);
PW_TRAITS(PW_PDUCT, 2, true, 0);
PW_TRAITS(PW_XDUCT, 3, true, 2);
PW_TRAITS(PW_HPRESSURE, 1, false, 0);
PW_TRAITS(PW_DZ2TMEAN, 2, false, 0);
PW_TRAITS(PW_KINDEX, 5, true, 0);
PW_TRAITS(PW_DUCTINDEX, 2, true, 0);
PW_TRAITS(PW_SHOWALTER, 3, true, 0);
PW_TRAITS(PW_BOYDEN, 3, false, 0);
PW_TRAITS(PW_SWEAT, 8, false, 0);
PW_TRAITS(PW_SOUNDSPEED, 2, false, 0);
PW_TRAITS(PW_ADDCONST, 1, false, 0);
PW_TRAITS(PW_ABSHUM, 2, false, 1);
PW_TRAITS(PW_WINDCOOLING, 3, false, 1);
PW_TRAITS(PW_UNDERCOOLED, 3, false, 0);
PW_TRAITS(PW_FLIGHTLEVEL, 1, false, 0);
PW_TRAITS(PW_SNOWCM, 3, false, 1);
PW_TRAITS(PW_CLASSES, 1, false, 0);
PW_TRAITS(PW_MINMAX_FIELDS, 2, false, 0);
PW_TRAITS(PW_MINMAX_CONST, 1, false, 0);
PW_TRAITS(PW_MATH, 1, false, 1);
PW_TRAITS(PW_REPLACE, 1, false, 0);
PW_TRAITS(PW_FILL, 0, false, 0);
PW_TRAITS(PW_FIELD_OP_FIELD, 2, false, 0);
PW_TRAITS(PW_FIELD_OP_CONST, 1, false, 0);
PW_TRAITS(PW_CONST_OP_FIELD, 1, false, 0);
PW_TRAITS(PW_VESSEL_ICING, 6, false, 0);
PW_TRAITS(PW_WINDDIR, 2, false, 0);
#undef PW_TRAITS

// MetConstants.h:46 (rcp, cplr, exl), :53 (ms2knots), :88-90 (flight-level tables)
#define PW_K_CPLR (MIFC_K_XLH / (MIFC_K_R / MIFC_K_CP))
#define PW_K_EXL (MIFC_K_EPS * MIFC_K_XLH)

__device__ __forceinline__ float ms2knots(float ff) // MetConstants.h:132-135
{
  return (float)((double)ff * (3600.0 / 1852.0));
}

// FieldCalculations.cc:280-283
__device__ __forceinline__ float tk_q_duct(float tk, float q, float p)
{
  return (float)(77.6 * (double)(p / tk) + 373000. * (double)(q * p) / (double)(MIFC_K_EPS * tk * tk));
}
// :285-296
__device__ __forceinline__ bool tk_rh_duct(const float* tab, float tk, float q, float p, float& out)
{
  const Ewt e(tk - MIFC_K_T0);
  if (!e.ok())
    return false;
  const float et = e.value(tab);
  const float rh = clamp_rh((float)((double)q * 0.01));
  out = (float)(77.6 * (double)(p / tk) + 373000. * (double)rh * (double)et / (double)(tk * tk));
  return true;
}

// x: the cell's input values in operator order
template <int OP>
__device__ __forceinline__ int pw_point(const PwParams& P, const float* tab, const PowTables& PT, const float* x, float& r)
{
  const float* s = P.s;
  if (OP == PW_PLEVELTHE) { // :389-396 with tk_rh_the :269-278; s0 = tconv, s1 = cvrh, s2 = thconv
    const float tk = x[0] * s[0];
    const Ewt e(tk - MIFC_K_T0);
    if (!e.ok())
      return PW_UNDEF;
    r = tk * s[2] + e.value(tab) * (x[1] * s[1]);
    return PW_OK;
  }
  if (OP == PW_XLEVELTHE) { // hlevelthe :1129-1135, alevelthe :1379-1384; x = t, q, ps|p
    const float p = P.hybrid ? (s[0] + s[1] * x[2]) : x[2];
    const float pi = MIFC_K_CP * pidcp_of(PT, p);
    if (P.compute == 1)
      r = (x[0] * MIFC_K_CP + x[1] * MIFC_K_XLH) / pi;
    else if (P.compute == 2)
      r = x[0] + x[1] * MIFC_K_XLH / pi;
    else
      return PW_KEEP;
    return PW_OK;
  }
  if (OP == PW_PDUCT) { // plevelducting :625-632; s0 = tconv, s1 = p
    const float tk = x[0] * s[0];
    if (P.compute == 1 || P.compute == 2) {
      r = tk_q_duct(tk, x[1], s[1]);
      return PW_OK;
    }
    return tk_rh_duct(tab, tk, x[1], s[1], r) ? PW_OK : PW_UNDEF;
  }
  if (OP == PW_XDUCT) { // hlevelducting :1257-1266, alevelducting :1491-1499; x = t, h, ps|p
    const float p = P.hybrid ? (s[0] + s[1] * x[2]) : x[2];
    float tk = x[0];
    if (P.compute % 2 == 0)
      tk *= pidcp_of(PT, p);
    if (P.compute == 1 || P.compute == 2) {
      r = tk_q_duct(tk, x[1], p);
      return PW_OK;
    }
    if (P.compute == 3 || P.compute == 4)
      return tk_rh_duct(tab, tk, x[1], p, r) ? PW_OK : PW_UNDEF;
    return PW_KEEP;
  }
  if (OP == PW_HPRESSURE) { // :1295-1296
    r = s[0] + s[1] * x[0];
    return PW_OK;
  }
  if (OP == PW_DZ2TMEAN) { // :501; s0 = convert, s1 = tconvert
    r = (x[0] - x[1]) * s[0] + s[1];
    return PW_OK;
  }
  if (OP == PW_KINDEX) { // :788-806; x = t500, t700, rh700, t850, rh850; s = cvt500, cvt700, cvt850
    const float rh8 = clamp_rh((float)(0.01 * (double)x[4]));
    const float tc850 = s[2] * x[3] - MIFC_K_T0;
    const float tc700 = s[1] * x[1] - MIFC_K_T0;
    const Ewt e850(tc850), e700(tc700);
    if (!(e850.ok() && e700.ok()))
      return PW_UNDEF;
    const float tdc850 = e850.inverse(tab, e850.value(tab) * rh8);
    const float rh7 = clamp_rh((float)(0.01 * (double)x[2]));
    const float tdc700 = e700.inverse(tab, e700.value(tab) * rh7);
    const float tc500 = s[0] * x[0] - MIFC_K_T0;
    r = (tc850 + tdc850) - (tc700 - tdc700) - tc500;
    return PW_OK;
  }
  if (OP == PW_DUCTINDEX) { // :849-862; s0 = tconvert
    const float bduct = 3.8e+5f;
    const float rh = clamp_rh((float)(0.01 * (double)x[1]));
    const float tk = x[0] * s[0];
    const Ewt e(tk - MIFC_K_T0);
    if (!e.ok())
      return PW_UNDEF;
    const float et = e.value(tab);
    const float etd = et * rh;
    const float tdk = e.inverse(tab, etd) + MIFC_K_T0;
    r = bduct * (et / (tk * tk) - etd / (tdk * tdk));
    return PW_OK;
  }
  if (OP == PW_SHOWALTER) { // :931-964; x = t500, t850, rh850; s = cvt500, cvt850, dryadiabat, p500, p850
    const float tk500 = s[0] * x[0];
    const float tk850 = s[1] * x[1];
    const float rh = clamp_rh((float)(0.01 * (double)x[2]));
    const Ewt e(tk850 - MIFC_K_T0);
    if (!e.ok())
      return PW_UNDEF;
    const float etd = e.value(tab) * rh;
    float tcl = s[2] * x[1];
    float qcl = MIFC_K_EPS * etd / s[4];
    // The loop has four float divisions and one double division per trip; none of the float ones is left as the
    // compiler's 11-instruction expansion, and every replacement is the IEEE quotient bit for bit:
    //   * tcl / cp, a constant divisor: q = tcl * y, r = fma(-cp, q, tcl), q + r * y with y = RN(1 / cp) -- three float
    //     instructions.  Checked against a / 1004 for EVERY finite float a with a normal or zero quotient
    //     (tools/verify_div_by_cp.c: 4 094 297 070 values, no mismatch); infinite, NaN and tiny tcl give a temperature
    //     the table does not cover either way, and the loop ends there.
    //   * / p500, a divisor fixed per call: the double product with the host-made reciprocal d[1] = 1 / p500, rounded to
    //     float -- the exact quotient of two floats is never closer than 2^-49 (relative) to a rounding boundary of float
    //     and the product is within 2^-52 of it (d[1] == 0 when p500 is so large or small that quotients could be
    //     subnormal: plain division then).
    //   * the two / tcl: the compiler's own sequence (LowerFDIV32: reciprocal, one Newton step, quotient, two residual
    //     corrections) WITHOUT its operand scaling, and with the refined reciprocal shared by both quotients.  v_div_scale
    //     leaves operands alone unless the divisor is subnormal or huge, the numerator nearly subnormal or the exponents
    //     96 apart; tcl is a temperature the table covers times cp (1.7e5 .. 3.8e5, known from e2.ok()), exl is a
    //     constant, and cplr * qcl is tested against [2^-64, 2^64) -- outside it (zero included: the sign of a zero
    //     quotient needs v_div_fixup) that quotient is a plain division.
    // The double division of dq keeps the refined double reciprocal (shared_reciprocal, mifc_device.h).
    const double inv_p500 = P.d[1];
    const float inv_cp = (float)(1.0 / (double)MIFC_K_CP);
    for (int it = 0; it < 7; ++it) { // moist adiabat, :948-960
      const float q0 = tcl * inv_cp;
      const float tq = __builtin_fmaf(__builtin_fmaf(-MIFC_K_CP, q0, tcl), inv_cp, q0); // tcl / cp
      const Ewt e2(tq - MIFC_K_T0);
      if (!e2.ok())
        break; // from here on tcl is a temperature the table covers times cp: positive, normal, far from the ends of the range
      const float esat = e2.value(tab);
      const float qsat = inv_p500 != 0. ? (float)((double)(MIFC_K_EPS * esat) * inv_p500) : MIFC_K_EPS * esat / s[3];
      float dq = qcl - qsat;
      float rc = __builtin_amdgcn_rcpf(tcl);
      rc = __builtin_fmaf(__builtin_fmaf(-tcl, rc, 1.0f), rc, rc);
      auto over_tcl = [&](float a) {
        float q = a * rc;
        q = __builtin_fmaf(__builtin_fmaf(-tcl, q, a), rc, q);
        return __builtin_fmaf(__builtin_fmaf(-tcl, q, a), rc, q);
      };
      const float num1 = PW_K_CPLR * qcl;
      float a1 = over_tcl(num1);
      if (__builtin_expect(!(__builtin_fabsf(num1) >= 0x1p-64f && __builtin_fabsf(num1) < 0x1p64f), 0))
        a1 = num1 / tcl;
      const float a2 = over_tcl(PW_K_EXL);
      const double den = 1. + (double)(a1 * a2);
      dq = (float)quotient((double)dq, den, shared_reciprocal(den));
      qcl = qcl - dq;
      tcl = tcl + dq * MIFC_K_XLH;
    }
    r = tk500 - tcl / MIFC_K_CP;
    return PW_OK;
  }
  if (OP == PW_BOYDEN) { // :1005-1006; s0 = tconv
    const float tc700 = x[0] * s[0] - MIFC_K_T0;
    r = (float)((double)(x[1] - x[2]) / 10. - (double)tc700 - 200.);
    return PW_OK;
  }
  if (OP == PW_SWEAT) { // :1029-1032; x = t850, t500, td850, td500, u850, v850, u500, v500
    const float ff850 = absval(x[4], x[5]);
    const float ff500 = absval(x[6], x[7]);
    const float sind = (x[6] * x[5] - x[7] * x[4]) / (ff850 * ff500);
    const float acc = 32 * x[2] + 20 * x[0] - 40 * x[1] - 20 * 49 + 2 * ms2knots(ff850) + ms2knots(ff500);
    r = (float)((double)acc + 125 * ((double)sind + 0.2));
    return PW_OK;
  }
  if (OP == PW_SOUNDSPEED) { // :1589-1594; s0 = tconv, d0 = Cz
    const double T = (double)(x[0] - s[0]);
    const double S = (double)x[1];
    const double Ct = 4.565 * T - 0.0517 * T * T + 0.000221 * T * T * T;
    const double Cs = (1.338 - 0.013 * T + 0.0001 * T * T) * (S - 35.0);
    r = (float)(1449.1 + Ct + Cs + P.d[0]);
    return PW_OK;
  }
  if (OP == PW_ADDCONST) { // cvtemp :1666
    r = x[0] + s[0];
    return PW_OK;
  }
  if (OP == PW_ABSHUM) { // :1715-1728 (sqrt and exp are the double functions there)
    const float C = 2.16679f;
    const float C1 = -7.85951783f, C2 = 1.84408259f, C3 = -11.7866497f, C4 = 22.6807411f, C5 = -15.9618719f, C6 = 1.80122502f;
    const float Tc = 647.096f, Pc = 220640.f;
    const float v = 1 - x[0] / Tc, tii = 1 / x[0];
    const float v2 = v * v, v3 = v * v2, v4 = v2 * v2, v1_5 = (float)((double)v * sqrt((double)v)), v3_5 = v2 * v1_5, v7_5 = v4 * v3_5;
    const double ex = (double)(Tc * tii * (C1 * v + C2 * v1_5 + C3 * v3 + C4 * v3_5 + C5 * v4 + C6 * v7_5));
    const double t2 = ex * 1.4426950408889634; // exp(ex) = 2^(ex * log2 e), see exp_float()
    const float Pws = (float)((double)Pc * ((t2 > -1000.0 && t2 < 1000.0) ? exp2_tab(PT, t2) : exp(ex)));
    const float Pw = Pws * x[1];
    r = C * Pw * 100 * tii;
    return PW_OK;
  }
  if (OP == PW_WINDCOOLING) { // :2211-2216; s0 = tconv
    const float tc = x[0] - s[0];
    const float ff = (float)((double)absval(x[1], x[2]) * 3.6);
    // powf(ff, 0.16f): taken in double and rounded once (glibc's powf is correctly rounded in all but rare cases)
    const float ffpow = pow_float(PT, ff, (double)0.16f);
    float d = (float)(13.12 + 0.6215 * (double)tc - 11.37 * (double)ffpow + 0.3965 * (double)tc * (double)ffpow);
    if (d > 0.f)
      d = 0.f;
    r = d;
    return PW_OK;
  }
  if (OP == PW_UNDERCOOLED) { // :2253-2256; s = precipMin, snowRateMax, tkMax; x = precip, snow, tk
    r = (x[0] >= s[0] && x[2] <= s[2] && x[1] <= x[0] * s[1]) ? 1.f : 0.f;
    return PW_OK;
  }
  if (OP == PW_FLIGHTLEVEL) { // :2331-2341, tables MetConstants.h:88-90
    const float pt[16] = {1000, 925, 850, 800, 700, 500, 400, 300, 250, 200, 150, 100, 70, 50, 30, 10};
    const float ft[16] = {5, 25, 50, 65, 100, 185, 235, 300, 340, 385, 445, 530, 605, 675, 780, 1020};
    float p = x[0];
    if (p > pt[0])
      p = pt[0];
    if (p < pt[15])
      p = pt[15];
    // first k in 1..15 with pt[k] <= p (15 if none), written as selects so that the tables stay in registers
    float p_lo = pt[14], p_hi = pt[15], f_lo = ft[14], f_hi = ft[15];
#pragma unroll
    for (int k = 14; k >= 1; --k) {
      if (pt[k] <= p) {
        p_lo = pt[k - 1];
        p_hi = pt[k];
        f_lo = ft[k - 1];
        f_hi = ft[k];
      }
    }
    const float ratio = (p - p_lo) / (p_hi - p_lo);
    r = f_lo + (f_hi - f_lo) * ratio;
    return PW_OK;
  }
  if (OP == PW_SNOWCM) { // :3093-3112; x = snow_water, tk2m, td2m
    if (x[0] <= 0.f) {
      r = 0.f;
      return PW_OK;
    }
    const float t = (float)((double)(x[1] + x[2]) / 2.);
    const double ea = ((double)t - 274.3) * 3.5, et2 = ea * 1.4426950408889634;
    const double ex = (et2 > -1000.0 && et2 < 1000.0) ? exp2_tab(PT, et2) : exp(ea);
    const float logit_t = (float)((1 - ex) / (1 + ex));
    const double q = ((double)t - 252.0) / 20.0;
    const float mm2cm_t = (float)(0.13 / (0.02 + 0.1 * q * q));
    const float fac = logit_t * mm2cm_t;
    r = (fac <= 1.f) ? x[0] : x[0] * fac;
    return PW_OK;
  }
  if (OP == PW_CLASSES) { // values2classes :2487-2491
    const int nvalues = P.nvalues - 2;
    const float f = x[0];
    if (!(f >= P.values[0] && f < P.values[nvalues + 1]))
      return PW_UNDEF;
    int j = 1;
    while (j < nvalues && P.values[j] < f)
      j++;
    r = (float)(j - 1);
    return PW_OK;
  }
  if (OP == PW_MINMAX_FIELDS) { // :2504 std::min(a,b) = b<a ? b : a; :2519 std::max(a,b) = a<b ? b : a
    r = (P.compute == 1) ? ((x[1] < x[0]) ? x[1] : x[0]) : ((x[0] < x[1]) ? x[1] : x[0]);
    return PW_OK;
  }
  if (OP == PW_MINMAX_CONST) { // :2512, :2527; s0 = value
    r = (P.compute == 1) ? ((s[0] < x[0]) ? s[0] : x[0]) : ((x[0] < s[0]) ? s[0] : x[0]);
    return PW_OK;
  }
  if (OP == PW_MATH) { // :2531-2563: the float libm functions, here taken in double and rounded once
    const double a = (double)x[0];
    switch (P.compute) {
    case 1:
      r = fabsf(x[0]);
      break;
    case 2:
      r = log_float(PT, x[0], true);
      break;
    case 3:
      r = exp_float(PT, a, true); // math_util.h:121-125 is double already
      break;
    case 4:
      r = log_float(PT, x[0], false);
      break;
    case 5:
      r = exp_float(PT, a, false);
      break;
    default:
      r = pow_float(PT, x[0], (double)s[0]);
      break;
    }
    return PW_OK;
  }
  if (OP == PW_REPLACE) { // replaceUndefined :2581, replaceDefined :2604 ("== undef" only); 3 = copy
    if (P.compute == 1)
      r = (x[0] == P.undef) ? s[0] : x[0];
    else if (P.compute == 2)
      r = (x[0] != P.undef) ? s[0] : x[0];
    else
      r = x[0];
    return PW_OK;
  }
  if (OP == PW_FILL) {
    r = s[0];
    return PW_OK;
  }
  if (OP == PW_FIELD_OP_FIELD) { // :2613-2621
    switch (P.compute) {
    case 1:
      r = x[0] + x[1];
      return PW_OK;
    case 2:
      r = x[0] - x[1];
      return PW_OK;
    case 3:
      r = x[0] * x[1];
      return PW_OK;
    default: // divideUndef :84-92
      if (x[1] != 0) {
        r = x[0] / x[1];
        return PW_OK;
      }
      return PW_UNDEF;
    }
  }
  if (OP == PW_FIELD_OP_CONST) { // :2633-2641; s0 = value
    switch (P.compute) {
    case 1:
      r = x[0] + s[0];
      break;
    case 2:
      r = x[0] - s[0];
      break;
    case 3:
      r = x[0] * s[0];
      break;
    default:
      r = x[0] / s[0];
      break;
    }
    return PW_OK;
  }
  if (OP == PW_WINDDIR) { // extension: see wind_direction() in mifc_device.h
    r = wind_direction(x[0], x[1]);
    return PW_OK;
  }
  if (OP == PW_VESSEL_ICING) { // FieldCalculationsVesselIcing.cc:93-104 (Overland), :125-171 (Mertins); x = airtemp, seatemp, u, v, sal, aice
    if (!((double)x[5] < 0.4))
      return PW_UNDEF;
    const double Tf = (-0.002 - 0.0524 * (double)x[4]) - 6.0E-5 * (double)(x[4] * x[4]); // freezing point of sea water, Stallabrass (1980)
    if ((double)x[1] < Tf)
      return PW_UNDEF;
    const double ff = (double)absval(x[2], x[3]);
    if (P.compute == 1) {
      const double A = 2.73e-2, B = 2.91e-4, C = 1.84e-6;
      const double ppr = ff * (Tf - (double)x[0]) / (1 + 0.3 * ((double)x[1] - Tf));
      r = (float)(A * ppr + B * (ppr * ppr) + C * ppr * ppr * ppr);
      return PW_OK;
    }
    const double temperature = (double)x[0], sst = (double)x[1];
    r = 0.f;
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
        r = 0.f;
      else if (temperature > temp1)
        r = (float)0.8333;
      else if (temperature > temp2)
        r = (float)2.0833;
      else
        r = (temperature <= temp3 || ff < 17.2) ? (float)4.375 : (float)6.25;
    }
    return PW_OK;
  }
  // PW_CONST_OP_FIELD :2655-2664
  switch (P.compute) {
  case 1:
    r = s[0] + x[0];
    return PW_OK;
  case 2:
    r = s[0] - x[0];
    return PW_OK;
  case 3:
    r = s[0] * x[0];
    return PW_OK;
  default:
    if (x[0] != 0) {
      r = s[0] / x[0];
      return PW_OK;
    }
    return PW_UNDEF;
  }
}

__device__ __forceinline__ void pw_store4(float* p, const float* o)
{
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f t;
  t.x = o[0];
  t.y = o[1];
  t.z = o[2];
  t.w = o[3];
  __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
}

template <int OP, bool VEC4>
__global__ __launch_bounds__(256) void pointwise_kernel(const PwParams P)
{
  constexpr int NIN = PwTraits<OP>::nin;
  constexpr int NV = NIN > 0 ? NIN : 1;
  __shared__ __attribute__((aligned(8))) float s_ewt[PwTraits<OP>::ewt ? MIFC_EWT_LDS : 2];
  // x^kappa tables (theta-e, ducting from theta) or the generic log2 / exp2 tables (the libm-class functions)
  __shared__ double s_pow[PwTraits<OP>::pow == 2 ? MIFC_KAPPA_LDS : (PwTraits<OP>::pow == 1 ? (2 * MIFC_POW_LOG_N + MIFC_POW_EXP_N) : 1)];
  PowTables PT = {s_pow, s_pow, s_pow, s_pow};

  const bool all = P.all_defined != 0;
  const float undef = P.undef;
  unsigned int bad = 0;

  if (VEC4) {
    const int n4 = P.n >> 2;
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    float4 v[NV];
    float4 old = make_float4(0.f, 0.f, 0.f, 0.f);
    // the loads of the (first) cells go out before the lookup tables are staged
    auto fetch = [&](int qq) {
#pragma unroll
      for (int k = 0; k < NIN; ++k)
        v[k] = reinterpret_cast<const float4*>(P.in[k])[qq];
      if (P.may_keep)
        old = reinterpret_cast<const float4*>(P.out)[qq];
    };
    if (q < n4)
      fetch(q);
    if (PwTraits<OP>::ewt)
      ewt_table_init(s_ewt, PwTraits<OP>::pow == 0); // one barrier for both tables
    if (PwTraits<OP>::pow == 1)
      PT = pow_tables_init(s_pow);
    if (PwTraits<OP>::pow == 2)
      PT = kappa_tables_init(s_pow);
    while (q < n4) {
      float o[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float x[NV];
        bool def = true;
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
          x[k] = (c == 0) ? v[k].x : (c == 1) ? v[k].y : (c == 2) ? v[k].z : v[k].w;
          def = def && is_def(x[k], undef);
        }
        int st;
        float r = 0.f;
        if (all || P.no_input_test || def)
          st = pw_point<OP>(P, s_ewt, PT, x, r);
        else
          st = P.skip_undefined_input ? PW_SKIP : PW_UNDEF;
        if (st == PW_OK)
          o[c] = r;
        else if (st == PW_UNDEF)
          o[c] = undef;
        bad += (st == PW_UNDEF || st == PW_SKIP) ? 1u : 0u;
      }
      pw_store4(P.out + (size_t)q * 4, o);
      q += gridDim.x * blockDim.x;
      if (q < n4)
        fetch(q);
    }
  } else {
    if (PwTraits<OP>::ewt)
      ewt_table_init(s_ewt, PwTraits<OP>::pow == 0); // one barrier for both tables
    if (PwTraits<OP>::pow == 1)
      PT = pow_tables_init(s_pow);
    if (PwTraits<OP>::pow == 2)
      PT = kappa_tables_init(s_pow);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += gridDim.x * blockDim.x) {
      float x[NV];
      bool def = true;
#pragma unroll
      for (int k = 0; k < NIN; ++k) {
        x[k] = P.in[k][i];
        def = def && is_def(x[k], undef);
      }
      int st;
      float r = 0.f;
      if (all || P.no_input_test || def)
        st = pw_point<OP>(P, s_ewt, PT, x, r);
      else
        st = P.skip_undefined_input ? PW_SKIP : PW_UNDEF;
      if (st == PW_OK)
        P.out[i] = r;
      else if (st == PW_UNDEF)
        P.out[i] = undef;
      bad += (st == PW_UNDEF || st == PW_SKIP) ? 1u : 0u;
    }
  }
  if (P.count && P.partials)
    block_count_store(P.partials + blockIdx.x, bad); // big launch: added up by count_partials_kernel behind it
  else
    block_count_add(P.count ? P.n_undefined : nullptr, bad); // one atomic per workgroup
}

inline bool aligned16(const void* p)
{
  return (reinterpret_cast<size_t>(p) & 15u) == 0;
}

template <int OP>
hipError_t launch_pw(const PwParams& prm, hipStream_t stream)
{
  constexpr int NIN = PwTraits<OP>::nin;
  bool vec_ok = aligned16(prm.out) && prm.n >= 4;
  for (int k = 0; k < NIN; ++k)
    vec_ok = vec_ok && aligned16(prm.in[k]);
  const int block = 256;
  const int cap = 0x7fffffff; // one float4 per lane: table staging (where an operator needs it) runs under the load latency
  if (vec_ok) {
    const int n4 = prm.n >> 2;
    int g = (n4 + block - 1) / block;
    g = g < 1 ? 1 : (g > cap ? cap : g);
    PwParams main = prm;
    const bool by_partials = prm.count && prm.partials && g >= 2048 && g <= prm.partials_cap; // see mifc_ewise.hip
    if (!by_partials)
      main.partials = nullptr;
    hipLaunchKernelGGL((pointwise_kernel<OP, true>), dim3(g), dim3(block), 0, stream, main);
    if (by_partials)
      (void)launch_count_partials(prm.partials, g, prm.n_undefined, stream);
    const int tail = prm.n - n4 * 4;
    if (tail > 0) {
      PwParams t = prm;
      t.partials = nullptr;
      t.n = tail;
      for (int k = 0; k < NIN; ++k)
        t.in[k] = prm.in[k] + n4 * 4;
      t.out = prm.out + n4 * 4;
      hipLaunchKernelGGL((pointwise_kernel<OP, false>), dim3(1), dim3(64), 0, stream, t);
    }
  } else {
    int g = (prm.n + block - 1) / block;
    g = g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g);
    PwParams q = prm;
    q.partials = nullptr;
    hipLaunchKernelGGL((pointwise_kernel<OP, false>), dim3(g), dim3(block), 0, stream, q);
  }
  return hipGetLastError();
}

// cvtemp compute 3, 4 (:1640-1650): sum and count of the defined cells.  The
// reference accumulates in float, serially; here the sum is taken in double
// (the decision it feeds compares the mean with 136.575 K -- fields are either
// around 280 or around 10, so the rounding of the mean never decides).
__global__ __launch_bounds__(256) void mean_defined_kernel(const float* __restrict__ f, int n, int all_defined, float undef, double* sum,
                                                           unsigned long long* count)
{
  double s = 0.0;
  unsigned int c = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float x = f[i];
    if (all_defined || is_def(x, undef)) {
      s += (double)x;
      c += 1;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_xor(s, off, 64);
    c += __shfl_xor(c, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(sum, s);
    atomicAdd(count, (unsigned long long)c);
  }
}

} // namespace

hipError_t launch_pointwise(const PwParams& prm, hipStream_t stream)
{
  if (prm.n <= 0)
    return hipSuccess;
  switch (prm.op) {
#define PW_CASE(OP) \
  case OP:          \
    return launch_pw<OP>(prm, stream)
    PW_CASE(PW_PLEVELTHE);
    PW_CASE(PW_XLEVELTHE);
    PW_CASE(PW_PDUCT);
    PW_CASE(PW_XDUCT);
    PW_CASE(PW_HPRESSURE);
    PW_CASE(PW_DZ2TMEAN);
    PW_CASE(PW_KINDEX);
    PW_CASE(PW_DUCTINDEX);
    PW_CASE(PW_SHOWALTER);
    PW_CASE(PW_BOYDEN);
    PW_CASE(PW_SWEAT);
    PW_CASE(PW_SOUNDSPEED);
    PW_CASE(PW_ADDCONST);
    PW_CASE(PW_ABSHUM);
    PW_CASE(PW_WINDCOOLING);
    PW_CASE(PW_UNDERCOOLED);
    PW_CASE(PW_FLIGHTLEVEL);
    PW_CASE(PW_SNOWCM);
    PW_CASE(PW_CLASSES);
    PW_CASE(PW_MINMAX_FIELDS);
    PW_CASE(PW_MINMAX_CONST);
    PW_CASE(PW_MATH);
    PW_CASE(PW_REPLACE);
    PW_CASE(PW_FILL);
    PW_CASE(PW_FIELD_OP_FIELD);
    PW_CASE(PW_FIELD_OP_CONST);
    PW_CASE(PW_CONST_OP_FIELD);
    PW_CASE(PW_VESSEL_ICING);
    PW_CASE(PW_WINDDIR);
#undef PW_CASE
  default:
    return hipErrorInvalidValue;
  }
}

hipError_t launch_mean_defined(const float* f, int n, int all_defined, float undef, double* sum, unsigned long long* count, hipStream_t stream)
{
  int g = (n + 255) / 256;
  g = g < 1 ? 1 : (g > 2048 ? 2048 : g);
  hipLaunchKernelGGL(mean_defined_kernel, dim3(g), dim3(256), 0, stream, f, n, all_defined, undef, sum, count);
  return hipGetLastError();
}

} // namespace mifc
