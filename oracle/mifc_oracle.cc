/*
 * mifc_oracle.cc -- TEST INFRASTRUCTURE, not product code.
 *
 * From-scratch, scalar CPU restatement of the mi-fieldcalc hot path (the
 * elementwise derived-variable operators and the 5-point-stencil operators
 * of SURVEY.md section 8a).  It is the CHECKER for the HIP kernels: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product library (mi-fieldcalc_amd/csrc) never links, loads or calls it.
 *
 * Parity status: PINNED.  Bit-identical to the real reference (compiled from
 * /root/reference into oracle/_ref/libmifc_ref.so, see oracle/Makefile) on the
 * seeded sweep of tests/test_oracle_vs_ref.py, on the golden vectors committed
 * under tests/golden/ (generated from that reference build by
 * tests/golden/make_golden.py), and on the reference's own known-answer table
 * test/FieldCalculationsTest.cc:72-83.
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference/src/mi_fieldcalc/).  Arithmetic notes that matter:
 *   - a bare literal such as 0.5, 2., 100., 0.01 makes the enclosing
 *     sub-expression double; the result is rounded to float once, on the
 *     store.  Written out explicitly below.
 *   - no fused multiply-add anywhere (the reference is built -mavx2 without
 *     -mfma); this file is compiled with -ffp-contract=off.
 *   - int(x) of a NaN / out-of-range float is undefined in C++; x86-64
 *     cvttss2si yields INT_MIN, which is what the compiled reference does and
 *     what trunc_like_x86() spells out.
 */
#define MIFC_ORACLE_PREFIX mifcorc_
#include "oracle_abi.h"

#include "oracle_common.h"

extern "C" {

const char* mifcorc_kind(void)
{
  return "restatement";
}

// FieldCalculations.cc:1819-1841, math_util.h:57-60
int mifcorc_vectorabs(int nx, int ny, const float* u, const float* v, float* ff, int* fdefined, float undef)
{
  const int n = nx * ny;
  const bool all = (*fdefined == ALL_DEFINED);
  size_t n_undefined = 0;
  for (int i = 0; i < n; ++i) {
    if (all || (defined1(u[i], undef) && defined1(v[i], undef))) {
      ff[i] = std::sqrt(u[i] * u[i] + v[i] * v[i]);
    } else {
      ff[i] = undef;
      n_undefined += 1;
    }
  }
  *fdefined = classify(n_undefined, n);
  return 1;
}

// EXTENSION -- NO REFERENCE FUNCTION: BASELINE.json's north_star names "wind direction from u/v", the
// reference only mentions one in comments (FieldCalculations.cc:1951-1952, :2190-2191).  The definition the product
// implements (mifc_winddir, csrc/mifc_device.h wind_direction): the meteorological direction the wind blows FROM, degrees
// clockwise from north, dd = 270 - atan2(v, u) * 180 / pi brought into [0, 360), calm (u == v == 0) -> 0; undefined
// handling and flag as vectorabs (:1831-1839).  Restated here in float64 and rounded to float ONCE: the yardstick the
// device's float evaluation is held to (1e-5 relative).  Parity of this operator is pinned by this definition only; the
// compiled reference has no counterpart (oracle/ref_shim.cc exports none).
int mifcorc_winddir(int nx, int ny, const float* u, const float* v, float* dd, int* fdefined, float undef)
{
  const int n = nx * ny;
  const bool all = (*fdefined == ALL_DEFINED);
  size_t n_undefined = 0;
  for (int i = 0; i < n; ++i) {
    if (all || (defined1(u[i], undef) && defined1(v[i], undef))) {
      double d = 0.0;
      if (!(u[i] == 0.f && v[i] == 0.f)) {
        d = 270.0 - std::atan2((double)v[i], (double)u[i]) * (180.0 / 3.14159265358979323846);
        if (d >= 360.0)
          d -= 360.0;
        if (d < 0.0)
          d += 360.0;
      }
      float f = (float)d;
      if (f >= 360.f) // 359.99999... rounds up to 360: the same direction as 0
        f = 0.f;
      dd[i] = f;
    } else {
      dd[i] = undef;
      n_undefined += 1;
    }
  }
  *fdefined = classify(n_undefined, n);
  return 1;
}

// FieldCalculations.cc:1843-1873
int mifcorc_relvort(int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr, float* out, int* fdefined, float undef)
{
  if (nx < 3 || ny < 3)
    return 0;
  const int n = nx * ny;
  const bool all = (*fdefined == ALL_DEFINED);
  const size_t bad = flat_loop(nx, n - nx, [&](int i) {
    if (!(all || (defined1(v[i - 1], undef) && defined1(v[i + 1], undef) && defined1(u[i - nx], undef) && defined1(u[i + nx], undef)))) {
      out[i] = undef;
      return false;
    }
    const float dvdx = v[i + 1] - v[i - 1];
    const float dudy = u[i + nx] - u[i - nx];
    out[i] = (float)(0.5 * (double)xmapr[i] * (double)dvdx - 0.5 * (double)ymapr[i] * (double)dudy);
    return true;
  });
  *fdefined = classify(bad, n - 2 * nx);
  fill_edges(nx, ny, out);
  return 1;
}

// FieldCalculations.cc:1875-1908
int mifcorc_absvort(int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr, const float* fcoriolis, float* out,
                    int* fdefined, float undef)
{
  if (nx < 3 || ny < 3)
    return 0;
  const int n = nx * ny;
  const bool all = (*fdefined == ALL_DEFINED);
  const size_t bad = flat_loop(nx, n - nx, [&](int i) {
    if (!(all || (defined1(v[i - 1], undef) && defined1(v[i + 1], undef) && defined1(u[i - nx], undef) && defined1(u[i + nx], undef)))) {
      out[i] = undef;
      return false;
    }
    const float dvdx = v[i + 1] - v[i - 1];
    const float dudy = u[i + nx] - u[i - nx];
    out[i] = (float)(0.5 * (double)xmapr[i] * (double)dvdx - 0.5 * (double)ymapr[i] * (double)dudy + (double)fcoriolis[i]);
    return true;
  });
  *fdefined = classify(bad, n - 2 * nx);
  fill_edges(nx, ny, out);
  return 1;
}

// FieldCalculations.cc:1910-1940.  The undefined test looks at the SAME four
// neighbours as relvort (:1927), not at the ones differenced (:1928).
int mifcorc_divergence(int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr, float* out, int* fdefined, float undef)
{
  if (nx < 3 || ny < 3)
    return 0;
  const int n = nx * ny;
  const bool all = (*fdefined == ALL_DEFINED);
  const size_t bad = flat_loop(nx, n - nx, [&](int i) {
    if (!(all || (defined1(v[i - 1], undef) && defined1(v[i + 1], undef) && defined1(u[i - nx], undef) && defined1(u[i + nx], undef)))) {
      out[i] = undef;
      return false;
    }
    const float dudx = u[i + 1] - u[i - 1];
    const float dvdy = v[i + nx] - v[i - nx];
    out[i] = (float)(0.5 * (double)xmapr[i] * (double)dudx + 0.5 * (double)ymapr[i] * (double)dvdy);
    return true;
  });
  *fdefined = classify(bad, n - 2 * nx);
  fill_edges(nx, ny, out);
  return 1;
}

// FieldCalculations.cc:1985-2074
int mifcorc_gradient(int nx, int ny, const float* f, const float* xmapr, const float* ymapr, int compute, float* out, int* fdefined, float undef)
{
  if (nx < 3 || ny < 3)
    return 0;
  const int n = nx * ny;
  const bool all = (*fdefined == ALL_DEFINED);
  size_t bad = 0;
  switch (compute) {
  case 1: // d/dx over the flat range [1, n-1)  (:2013-2021)
    bad = flat_loop(1, n - 1, [&](int i) {
      if (!(all || (defined1(f[i - 1], undef) && defined1(f[i + 1], undef)))) {
        out[i] = undef;
        return false;
      }
      const float d = f[i + 1] - f[i - 1];
      out[i] = (float)(0.5 * (double)xmapr[i] * (double)d);
      return true;
    });
    break;
  case 2: // d/dy (:2025-2033)
    bad = flat_loop(nx, n - nx, [&](int i) {
      if (!(all || (defined1(f[i - nx], undef) && defined1(f[i + nx], undef)))) {
        out[i] = undef;
        return false;
      }
      const float d = f[i + nx] - f[i - nx];
      out[i] = (float)(0.5 * (double)ymapr[i] * (double)d);
      return true;
    });
    break;
  case 3: // |grad f| from the two float-rounded partials (:2037-2047)
    bad = flat_loop(nx, n - nx, [&](int i) {
      if (!(all || (defined1(f[i - nx], undef) && defined1(f[i - 1], undef) && defined1(f[i + 1], undef) && defined1(f[i + nx], undef)))) {
        out[i] = undef;
        return false;
      }
      const float dx = f[i + 1] - f[i - 1];
      const float dy = f[i + nx] - f[i - nx];
      const float dfdx = (float)(0.5 * (double)xmapr[i] * (double)dx);
      const float dfdy = (float)(0.5 * (double)ymapr[i] * (double)dy);
      out[i] = std::sqrt(dfdx * dfdx + dfdy * dfdy);
      return true;
    });
    break;
  case 4: // laplacian (:2051-2061); second differences are rounded to float first
    bad = flat_loop(nx, n - nx, [&](int i) {
      if (!(all || (defined1(f[i - nx], undef) && defined1(f[i - 1], undef) && defined1(f[i], undef) && defined1(f[i + 1], undef) &&
                    defined1(f[i + nx], undef)))) {
        out[i] = undef;
        return false;
      }
      const float d2x = (float)((double)f[i - 1] - 2.0 * (double)f[i] + (double)f[i + 1]);
      const float d2y = (float)((double)f[i - nx] - 2.0 * (double)f[i] + (double)f[i + nx]);
      const double xm = xmapr[i], ym = ymapr[i];
      out[i] = (float)(4.0 * (0.25 * xm * xm * (double)d2x + 0.25 * ym * ym * (double)d2y));
      return true;
    });
    break;
  default:
    return 0;
  }
  *fdefined = classify(bad, n - 2 * nx); // also for compute==1 (:2068)
  fill_edges(nx, ny, out);
  return 1;
}

// FieldCalculations.cc:638-672.  n_undefined is bumped on EVERY cell (:664),
// so the flag always comes out NONE_DEFINED.
int mifcorc_plevelgwind_xcomp(int nx, int ny, const float* z, const float* /*xmapr*/, const float* ymapr, const float* fcoriolis, float* ug,
                              int* fdefined, float undef)
{
  if (nx < 3 || ny < 3)
    return 0;
  const int n = nx * ny;
  const bool all = (*fdefined == ALL_DEFINED);
  size_t bad = 0;
  for (int i = nx; i < n - nx; ++i) {
    if (all || (defined1(z[i - nx], undef) && defined1(z[i - 1], undef) && defined1(z[i + 1], undef) && defined1(z[i + nx], undef))) {
      const float dz = z[i + nx] - z[i - nx];
      ug[i] = (float)(-0.5 * (double)ymapr[i] * (double)dz * (double)K_G / (double)fcoriolis[i]);
    } else {
      ug[i] = undef;
    }
    bad += 1;
  }
  *fdefined = classify(bad, n - 2 * nx);
  fill_edges(nx, ny, ug);
  return 1;
}

// FieldCalculations.cc:674-706.  The reference has no size guard here and runs
// into out-of-bounds accesses for ny < 3; the restatement (and the product)
// return false instead.
int mifcorc_plevelgwind_ycomp(int nx, int ny, const float* z, const float* xmapr, const float* /*ymapr*/, const float* fcoriolis, float* vg,
                              int* fdefined, float undef)
{
  if (nx < 3 || ny < 3)
    return 0;
  const int n = nx * ny;
  const bool all = (*fdefined == ALL_DEFINED);
  const size_t bad = flat_loop(nx, n - nx, [&](int i) {
    if (!(all || (defined1(z[i - nx], undef) && defined1(z[i - 1], undef) && defined1(z[i + 1], undef) && defined1(z[i + nx], undef)))) {
      vg[i] = undef;
      return false;
    }
    const float dz = z[i + 1] - z[i - 1];
    vg[i] = (float)(0.5 * (double)xmapr[i] * (double)dz * (double)K_G / (double)fcoriolis[i]);
    return true;
  });
  *fdefined = classify(bad, n - 2 * nx);
  fill_edges(nx, ny, vg);
  return 1;
}

// FieldCalculations.cc:708-743
int mifcorc_plevelgvort(int nx, int ny, const float* z, const float* xmapr, const float* ymapr, const float* fcoriolis, float* gvort, int* fdefined,
                        float undef)
{
  if (nx < 3 || ny < 3)
    return 0;
  const float g4 = (float)((double)K_G * 4.);
  const int n = nx * ny;
  const bool all = (*fdefined == ALL_DEFINED);
  const size_t bad = flat_loop(nx, n - nx, [&](int i) {
    if (!(all || (defined1(z[i - nx], undef) && defined1(z[i - 1], undef) && defined1(z[i], undef) && defined1(z[i + 1], undef) &&
                  defined1(z[i + nx], undef)))) {
      gvort[i] = undef;
      return false;
    }
    const double xm = xmapr[i], ym = ymapr[i], zc = z[i];
    const double d2x = (double)z[i - 1] - 2. * zc + (double)z[i + 1];
    const double d2y = (double)z[i - nx] - 2. * zc + (double)z[i + nx];
    gvort[i] = (float)((0.25 * xm * xm * d2x + 0.25 * ym * ym * d2y) * (double)g4 / (double)fcoriolis[i]);
    return true;
  });
  *fdefined = classify(bad, n - 2 * nx);
  fill_edges(nx, ny, gvort);
  return 1;
}

// FieldCalculations.cc:1511-1549.  Flag is classified against nx*ny (:1543).
int mifcorc_ilevelgwind(int nx, int ny, const float* mpot, const float* xmapr, const float* ymapr, const float* fcoriolis, float* ug, float* vg,
                        int* fdefined, float undef)
{
  if (nx < 3 || ny < 3)
    return 0;
  const int n = nx * ny;
  const bool all = (*fdefined == ALL_DEFINED);
  const size_t bad = flat_loop(nx, n - nx, [&](int i) {
    if (!(all ||
          (defined1(mpot[i - nx], undef) && defined1(mpot[i - 1], undef) && defined1(mpot[i + 1], undef) && defined1(mpot[i + nx], undef)))) {
      ug[i] = undef;
      vg[i] = undef;
      return false;
    }
    const float dy = mpot[i + nx] - mpot[i - nx];
    const float dx = mpot[i + 1] - mpot[i - 1];
    ug[i] = (float)(-0.5 * (double)ymapr[i] * (double)dy / (double)fcoriolis[i]);
    vg[i] = (float)(0.5 * (double)xmapr[i] * (double)dx / (double)fcoriolis[i]);
    return true;
  });
  *fdefined = classify(bad, n);
  fill_edges(nx, ny, ug);
  fill_edges(nx, ny, vg);
  return 1;
}

// FieldCalculations.cc:328-367.  compute 1..3 go through unaryFunctionField
// (:94-122) which leaves fDefined untouched and does not count; 4,5 go through
// unaryFunctionFieldUndef (:142-159).
int mifcorc_pleveltemp(int nx, int ny, const float* tinp, float p, const char* unit, int compute, float* tout, int* fdefined, float undef)
{
  if (p <= 0)
    return 0;
  if (compute < 3) {
    if (unit_is(unit, "celsius"))
      compute = 1;
    else if (unit_is(unit, "kelvin"))
      compute = 2;
  }
  const float pidcp = pidcp_of(p), pi = pidcp * K_CP;
  const size_t n = (size_t)(nx * ny);
  const bool all = (*fdefined == ALL_DEFINED);
  switch (compute) {
  case 1:
  case 2:
  case 3:
    for (size_t i = 0; i < n; ++i) {
      const float t = tinp[i];
      if (!(all || defined1(t, undef)))
        tout[i] = undef;
      else if (compute == 1)
        tout[i] = t * pidcp - K_T0;
      else if (compute == 2)
        tout[i] = t * pidcp;
      else
        tout[i] = t / pidcp;
    }
    return 1;
  case 4:
  case 5: {
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) {
      const float t = tinp[i];
      float r;
      if ((all || defined1(t, undef)) && (compute == 4 ? t_thesat(t, p, pi, r) : th_thesat(t, p, pi, r))) {
        tout[i] = r;
      } else {
        tout[i] = undef;
        bad += 1;
      }
    }
    *fdefined = classify(bad, n);
    return 1;
  }
  default:
    return 0;
  }
}

// FieldCalculations.cc:1046-1098.  No range check on compute: for a value
// outside 1..5 defined cells are simply left unwritten.
int mifcorc_hleveltemp(int nx, int ny, const float* tinp, const float* ps, float alevel, float blevel, const char* unit, int compute, float* tout,
                       int* fdefined, float undef)
{
  if (compute < 3) {
    if (unit_is(unit, "celsius"))
      compute = 1;
    else if (unit_is(unit, "kelvin"))
      compute = 2;
  }
  const int n = nx * ny;
  const bool all = (*fdefined == ALL_DEFINED);
  if (bad_hlevel(alevel, blevel))
    return 0;
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    if (all || (defined1(tinp[i], undef) && defined1(ps[i], undef))) {
      const float p = alevel + blevel * ps[i];
      const float pidcp = pidcp_of(p);
      float r;
      if (compute == 1) {
        tout[i] = tinp[i] * pidcp - K_T0;
      } else if (compute == 2) {
        tout[i] = tinp[i] * pidcp;
      } else if (compute == 3) {
        tout[i] = tinp[i] / pidcp;
      } else if (compute == 4 || compute == 5) {
        const bool ok = (compute == 4) ? t_thesat(tinp[i], p, pidcp * K_CP, r) : th_thesat(tinp[i], p, pidcp * K_CP, r);
        if (ok) {
          tout[i] = r;
        } else {
          tout[i] = undef;
          bad += 1;
        }
      }
    } else {
      tout[i] = undef;
      bad += 1;
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}

// FieldCalculations.cc:1310-1353
int mifcorc_aleveltemp(int nx, int ny, const float* tinp, const float* p, const char* unit, int compute, float* tout, int* fdefined, float undef)
{
  if (compute <= 0 || compute >= 6)
    return 0;
  if (compute < 3) {
    if (unit_is(unit, "celsius"))
      compute = 1;
    else if (unit_is(unit, "kelvin"))
      compute = 2;
  }
  const int n = nx * ny;
  const bool all = (*fdefined == ALL_DEFINED);
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    if (all || (defined1(tinp[i], undef) && defined1(p[i], undef))) {
      float r;
      if (compute == 1) {
        tout[i] = tinp[i] * pidcp_of(p[i]) - K_T0;
      } else if (compute == 2) {
        tout[i] = tinp[i] * pidcp_of(p[i]);
      } else if (compute == 3) {
        tout[i] = tinp[i] / pidcp_of(p[i]);
      } else {
        const bool ok = (compute == 4) ? t_thesat(tinp[i], p[i], pi_of(p[i]), r) : th_thesat(tinp[i], p[i], pi_of(p[i]), r);
        if (ok) {
          tout[i] = r;
        } else {
          tout[i] = undef;
          bad += 1;
        }
      }
    } else {
      tout[i] = undef;
      bad += 1;
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}

namespace {
// Shared tail of the three *levelhum operators once tk and p are known.
// kind: 0 q->RH, 1 RH->q, 2 q->Td, 3 RH->Td
inline bool hum_point(int kind, float tk, float hum, float p, float tdconv, float& r)
{
  switch (kind) {
  case 0:
    return tk_q_rh(tk, hum, p, r);
  case 1:
    return tk_rh_q(tk, hum, p, r);
  case 2:
    return tk_q_td(tk, hum, p, tdconv, r);
  default:
    return tk_rh_td(tk, hum, tdconv, r);
  }
}
} // namespace

// FieldCalculations.cc:400-464.  Numbering: 5,6,9,10 = RH->Td; 7,8,11,12 = q->Td.
int mifcorc_plevelhum(int nx, int ny, const float* t, const float* huminp, float p, const char* unit, int compute, float* humout, int* fdefined,
                      float undef)
{
  if (p <= 0 || compute <= 0 || compute >= 13)
    return 0;
  if (compute > 8 && unit_is(unit, "celsius"))
    compute -= 4;
  else if (compute > 4 && compute <= 8 && unit_is(unit, "kelvin"))
    compute += 4;
  const int n = nx * ny;
  if (p == undef && (compute != 5 && compute != 6 && compute != 9 && compute != 10)) {
    *fdefined = NONE_DEFINED; // fillUndef, :76-82
    for (int i = 0; i < n; ++i)
      humout[i] = undef;
    return 1;
  }
  const float pi = pi_of(p);
  const float tconv = (compute % 2 == 0) ? (pi / K_CP) : 1;
  const float tdconv = (compute >= 9) ? K_T0 : 0;
  int kind;
  if (compute == 1 || compute == 2)
    kind = 0;
  else if (compute == 3 || compute == 4)
    kind = 1;
  else if (compute == 5 || compute == 6 || compute == 9 || compute == 10)
    kind = 3;
  else
    kind = 2;
  const bool all = (*fdefined == ALL_DEFINED);
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    float r;
    if ((all || (defined1(t[i], undef) && defined1(huminp[i], undef))) && hum_point(kind, t[i] * tconv, huminp[i], p, tdconv, r)) {
      humout[i] = r;
    } else {
      humout[i] = undef;
      bad += 1;
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}

// FieldCalculations.cc:1145-1217.  Numbering: 5,6,9,10 = q->Td; 7,8,11,12 = RH->Td.
// ps is tested with "!= undef" only, and only when p is needed (:1187).
int mifcorc_hlevelhum(int nx, int ny, const float* t, const float* huminp, const float* ps, float alevel, float blevel, const char* unit, int compute,
                      float* humout, int* fdefined, float undef)
{
  if (compute <= 0 || compute >= 13)
    return 0;
  if (bad_hlevel(alevel, blevel))
    return 0;
  if (compute > 8 && unit_is(unit, "celsius"))
    compute -= 4;
  else if (compute > 4 && compute <= 8 && unit_is(unit, "kelvin"))
    compute += 4;
  const int n = nx * ny;
  const bool all = (*fdefined == ALL_DEFINED);
  const float tdconv = (compute >= 9) ? K_T0 : 0;
  const bool need_p = !(compute == 7 || compute == 11);
  const bool from_theta = (compute % 2 == 0);
  int kind;
  if (compute <= 2)
    kind = 0;
  else if (compute <= 4)
    kind = 1;
  else if (compute == 5 || compute == 6 || compute == 9 || compute == 10)
    kind = 2;
  else
    kind = 3;
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    float r;
    bool ok = (all || (defined1(t[i], undef) && defined1(huminp[i], undef))) && (!need_p || all || ps[i] != undef);
    if (ok) {
      const float p = need_p ? (alevel + blevel * ps[i]) : 0;
      const float tk = from_theta ? t[i] * pidcp_of(p) : t[i];
      ok = hum_point(kind, tk, huminp[i], p, tdconv, r);
    }
    if (ok) {
      humout[i] = r;
    } else {
      humout[i] = undef;
      bad += 1;
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}

// FieldCalculations.cc:1394-1458.  The p clause is inverted with respect to
// hlevelhum (:1429): p is tested only for compute 7/11, which do not use it.
int mifcorc_alevelhum(int nx, int ny, const float* t, const float* huminp, const float* p, const char* unit, int compute, float* humout, int* fdefined,
                      float undef)
{
  if (compute <= 0 || compute >= 13)
    return 0;
  if (compute > 8 && unit_is(unit, "celsius"))
    compute -= 4;
  else if (compute > 4 && compute <= 8 && unit_is(unit, "kelvin"))
    compute += 4;
  const int n = nx * ny;
  const float tdconv = (compute >= 9) ? K_T0 : 0;
  const bool all = (*fdefined == ALL_DEFINED);
  const bool from_theta = (compute % 2 == 0);
  int kind;
  if (compute <= 2)
    kind = 0;
  else if (compute <= 4)
    kind = 1;
  else if (compute == 5 || compute == 6 || compute == 9 || compute == 10)
    kind = 2;
  else
    kind = 3;
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    float r;
    bool ok = (all || (defined1(t[i], undef) && defined1(huminp[i], undef))) && ((compute != 7 && compute != 11) || all || p[i] != undef);
    if (ok) {
      const float tk = from_theta ? t[i] * pidcp_of(p[i]) : t[i];
      ok = hum_point(kind, tk, huminp[i], p[i], tdconv, r);
    }
    if (ok) {
      humout[i] = r;
    } else {
      humout[i] = undef;
      bad += 1;
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}

// FieldCalculations.cc:1738-1817
int mifcorc_cvhum(int nx, int ny, const float* t, const float* huminp, const char* unit, int compute, float* humout, int* fdefined, float undef)
{
  float unit_scale = 100;
  if (compute == 1 && unit_is(unit, "celsius"))
    compute = 2;
  if ((compute == 4 || compute == 5) && unit_is(unit, "1"))
    unit_scale = 1;
  if (compute < 1 || compute > 5)
    return 0;
  const int n = nx * ny;
  const float tconv = (compute == 1 || compute == 2 || compute == 4) ? K_T0 : 0;
  const float tdconv = (compute == 1) ? K_T0 : 0;
  const bool all = (*fdefined == ALL_DEFINED);
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    bool ok = all || (defined1(t[i], undef) && defined1(huminp[i], undef));
    float r = undef;
    if (ok) {
      if (compute <= 3) { // T, RH(%) -> Td; table lookup built directly from (t - tconv), :1767
        const Ewt e(t[i] - tconv);
        ok = e.ok();
        if (ok) {
          const float et = e.value();
          const float rh = clamp_rh((float)(0.01 * (double)huminp[i]));
          r = e.inverse(rh * et) + tdconv;
        }
      } else { // T, Td -> RH
        const Ewt e(t[i] - tconv), e2(huminp[i] - tconv);
        ok = e.ok() && e2.ok();
        if (ok) {
          const float rh = e2.value() / e.value();
          r = rh * unit_scale;
        }
      }
    }
    if (ok) {
      humout[i] = r;
    } else {
      humout[i] = undef;
      bad += 1;
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}

// ---------------------------------------------------------------- SURVEY.md 8f-1

// FieldCalculations.cc:1942-1983
int mifcorc_advection(int nx, int ny, const float* f, const float* u, const float* v, const float* xmapr, const float* ymapr, float hours, float* advec,
                      int* fdefined, float undef)
{
  if (nx < 3 || ny < 3)
    return 0;
  const float scale = (float)(-3600. * (double)hours);
  const int n = nx * ny;
  const bool all = (*fdefined == ALL_DEFINED);
  const size_t bad = flat_loop(nx, n - nx, [&](int i) {
    if (!(all || (defined1(u[i], undef) && defined1(v[i], undef) && defined1(f[i - nx], undef) && defined1(f[i - 1], undef) &&
                  defined1(f[i + 1], undef) && defined1(f[i + nx], undef)))) {
      advec[i] = undef;
      return false;
    }
    const float dx = f[i + 1] - f[i - 1];
    const float dy = f[i + nx] - f[i - nx];
    advec[i] = (float)(((double)u[i] * 0.5 * (double)xmapr[i] * (double)dx + (double)v[i] * 0.5 * (double)ymapr[i] * (double)dy) * (double)scale);
    return true;
  });
  *fdefined = classify(bad, n - 2 * nx);
  fill_edges(nx, ny, advec);
  return 1;
}

// FieldCalculations.cc:2424-2460: the four partials are rounded to float, the
// combination is float arithmetic
int mifcorc_jacobian(int nx, int ny, const float* f1, const float* f2, const float* xmapr, const float* ymapr, float* out, int* fdefined, float undef)
{
  if (nx < 3 || ny < 3)
    return 0;
  const int n = nx * ny;
  const bool all = (*fdefined == ALL_DEFINED);
  const size_t bad = flat_loop(nx, n - nx, [&](int i) {
    if (!(all || (defined1(f1[i - nx], undef) && defined1(f1[i - 1], undef) && defined1(f1[i + 1], undef) && defined1(f1[i + nx], undef) &&
                  defined1(f2[i - nx], undef) && defined1(f2[i - 1], undef) && defined1(f2[i + 1], undef) && defined1(f2[i + nx], undef)))) {
      out[i] = undef;
      return false;
    }
    const float a = f1[i + 1] - f1[i - 1], b = f1[i + nx] - f1[i - nx], c = f2[i + 1] - f2[i - 1], d = f2[i + nx] - f2[i - nx];
    const float df1dx = (float)(0.5 * (double)xmapr[i] * (double)a);
    const float df1dy = (float)(0.5 * (double)ymapr[i] * (double)b);
    const float df2dx = (float)(0.5 * (double)xmapr[i] * (double)c);
    const float df2dy = (float)(0.5 * (double)ymapr[i] * (double)d);
    out[i] = df1dx * df2dy - df1dy * df2dx;
    return true;
  });
  *fdefined = classify(bad, n - 2 * nx);
  fill_edges(nx, ny, out);
  return 1;
}

namespace {
// FieldCalculations.cc:2374-2378 / :2410-2414: keep |f| away from zero
inline float clamp_coriolis(float fcor, float fcormin, float fcormax)
{
  if (fcor >= 0. && fcor < fcormin)
    return fcormin;
  if (fcor <= 0. && fcor > fcormax)
    return fcormax;
  return fcor;
}
} // namespace

// FieldCalculations.cc:2351-2385 (pointwise despite the name)
int mifcorc_momentumXcoordinate(int nx, int ny, const float* v, const float* xmapr, const float* fcoriolis, float fcoriolisMin, float* mxy, int* fdefined,
                                float undef)
{
  if (nx < 3 || ny < 3)
    return 0;
  const int n = nx * ny;
  const float fcormin = fabsf(fcoriolisMin), fcormax = -fcormin;
  const bool all = (*fdefined == ALL_DEFINED);
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    if (all || defined1(v[i], undef)) {
      const float fcor = clamp_coriolis(fcoriolis[i], fcormin, fcormax);
      mxy[i] = float(i % nx) + v[i] * xmapr[i] / fcor;
    } else {
      mxy[i] = undef;
      bad += 1;
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}

// FieldCalculations.cc:2387-2422
int mifcorc_momentumYcoordinate(int nx, int ny, const float* u, const float* ymapr, const float* fcoriolis, float fcoriolisMin, float* nxy, int* fdefined,
                                float undef)
{
  if (nx < 3 || ny < 3)
    return 0;
  const int n = nx * ny;
  const float fcormin = fabsf(fcoriolisMin), fcormax = -fcormin;
  const bool all = (*fdefined == ALL_DEFINED);
  size_t bad = 0;
  for (int i = 0; i < n; ++i) {
    if (all || defined1(u[i], undef)) {
      const float fcor = clamp_coriolis(fcoriolis[i], fcormin, fcormax);
      nxy[i] = float(i / nx) - u[i] * ymapr[i] / fcor;
    } else {
      nxy[i] = undef;
      bad += 1;
    }
  }
  *fdefined = classify(bad, n);
  return 1;
}

// FieldCalculations.cc:2266-2309.  Two passes: |grad T| (gradient compute 3, which
// also updates fDefined and fills its edges), then a stencil over T and |grad T|.
// The second pass decides "all defined" from the flag the FIRST pass returned.
int mifcorc_thermalFrontParameter(int nx, int ny, const float* tx, const float* xmapr, const float* ymapr, float* tfp, int* fdefined, float undef)
{
  const int n = nx * ny;
  float* absdelt = new float[n > 0 ? n : 1];
  if (!mifcorc_gradient(nx, ny, tx, xmapr, ymapr, 3, absdelt, fdefined, undef)) {
    delete[] absdelt;
    return 0;
  }
  const bool all = (*fdefined == ALL_DEFINED);
  const size_t bad = flat_loop(nx, n - nx, [&](int i) {
    const bool ok = (all || (defined1(tx[i - nx], undef) && defined1(tx[i - 1], undef) && defined1(tx[i + 1], undef) && defined1(tx[i + nx], undef) &&
                             defined1(absdelt[i - nx], undef) && defined1(absdelt[i - 1], undef) && defined1(absdelt[i], undef) &&
                             defined1(absdelt[i + 1], undef) && defined1(absdelt[i + nx], undef))) &&
                    absdelt[i] != 0;
    if (!ok) {
      tfp[i] = undef;
      return false;
    }
    const float gax = absdelt[i + 1] - absdelt[i - 1], gay = absdelt[i + nx] - absdelt[i - nx];
    const float tdx = tx[i + 1] - tx[i - 1], tdy = tx[i + nx] - tx[i - nx];
    const float dabsdeltdx = (float)(0.5 * (double)xmapr[i] * (double)gax);
    const float dabsdeltdy = (float)(0.5 * (double)ymapr[i] * (double)gay);
    const float dtdxa = (float)(0.5 * (double)xmapr[i] * (double)tdx / (double)absdelt[i]);
    const float dtdya = (float)(0.5 * (double)ymapr[i] * (double)tdy / (double)absdelt[i]);
    tfp[i] = -(dabsdeltdx * dtdxa + dabsdeltdy * dtdya);
    return true;
  });
  *fdefined = classify(bad, n - 2 * nx);
  fill_edges(nx, ny, tfp);
  delete[] absdelt;
  return 1;
}

// FieldCalculations.cc:505-595.  plevelgwind_xcomp leaves fDefined = NONE_DEFINED
// (:664), so the y component always runs with its tests switched on; the last
// pass tests with "!= undef" only (no NaN test) and ignores the input flag.
int mifcorc_plevelqvector(int nx, int ny, const float* z, const float* t, const float* xmapr, const float* ymapr, const float* fcoriolis, float p,
                          int compute, float* qcomp, int* fdefined, float undef)
{
  if (p <= 0.0)
    return 0;
  if (nx < 3 || ny < 3)
    return 0;
  float tscale;
  if (compute == 1 || compute == 3) {
    tscale = 1.0;
  } else if (compute == 2 || compute == 4) {
    const float pi = K_CP * powf(p / K_P0, K_R / K_CP); // :538
    tscale = pi / K_CP;
  } else {
    return 0;
  }
  const int n = nx * ny;
  float* ug = new float[n];
  float* vg = new float[n];
  int ok = mifcorc_plevelgwind_xcomp(nx, ny, z, xmapr, ymapr, fcoriolis, ug, fdefined, undef) &&
           mifcorc_plevelgwind_ycomp(nx, ny, z, xmapr, ymapr, fcoriolis, vg, fdefined, undef);
  if (ok) {
    const float c = (float)((double)(-K_R) / ((double)p * 100.));
    const size_t bad = flat_loop(nx, n - nx, [&](int i) {
      if (!(ug[i - nx] != undef && ug[i - 1] != undef && ug[i + 1] != undef && ug[i + nx] != undef && vg[i - nx] != undef && vg[i - 1] != undef &&
            vg[i + 1] != undef && vg[i + nx] != undef && t[i - nx] != undef && t[i - 1] != undef && t[i + 1] != undef && t[i + nx] != undef)) {
        qcomp[i] = undef;
        return false;
      }
      const float tdx = t[i + 1] - t[i - 1], tdy = t[i + nx] - t[i - nx];
      const float dtdx = (float)(0.5 * (double)xmapr[i] * (double)tscale * (double)tdx);
      const float dtdy = (float)(0.5 * (double)ymapr[i] * (double)tscale * (double)tdy);
      if (compute < 3) {
        const float a = ug[i + 1] - ug[i - 1], b = vg[i + 1] - vg[i - 1];
        const float dugdx = (float)(0.5 * (double)xmapr[i] * (double)a);
        const float dvgdx = (float)(0.5 * (double)xmapr[i] * (double)b);
        qcomp[i] = c * (dugdx * dtdx + dvgdx * dtdy);
      } else {
        const float a = ug[i + nx] - ug[i - nx], b = vg[i + nx] - vg[i - nx];
        const float dugdy = (float)(0.5 * (double)ymapr[i] * (double)a);
        const float dvgdy = (float)(0.5 * (double)ymapr[i] * (double)b);
        qcomp[i] = c * (dugdy * dtdx + dvgdy * dtdy);
      }
      return true;
    });
    *fdefined = classify(bad, n - 2 * nx);
    fill_edges(nx, ny, qcomp);
  }
  delete[] ug;
  delete[] vg;
  return ok;
}

} // extern "C"
