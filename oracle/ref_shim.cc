/*
 * ref_shim.cc -- TEST INFRASTRUCTURE, not product code.
 *
 * Flat C wrappers around the REAL reference operators. This file is compiled
 * together with the reference's own translation units, taken from
 * /root/reference/src/mi_fieldcalc/ where they lie (see oracle/Makefile,
 * target _ref/libmifc_ref.so); nothing of the reference is copied here.
 * It exists so that Python (ctypes) can call miutil::fieldcalc::* without
 * having to fabricate std::string / enum references.
 *
 * Only tests/, tests/golden/make_golden.py, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load the resulting library.
 */
#define MIFC_ORACLE_PREFIX mifcref_
#include "oracle_abi.h"

#include <mi_fieldcalc/FieldCalculations.h>
#include <mi_fieldcalc/mi_fieldcalc_version.h>

#include <string>

namespace fc = miutil::fieldcalc;

namespace {
struct Flag
{
  int* p;
  miutil::ValuesDefined v;
  explicit Flag(int* fdefined)
      : p(fdefined)
      , v(static_cast<miutil::ValuesDefined>(*fdefined))
  {
  }
  ~Flag() { *p = static_cast<int>(v); }
};
} // namespace

extern "C" {

int mifcref_vectorabs(int nx, int ny, const float* u, const float* v, float* ff, int* fdefined, float undef)
{
  Flag f(fdefined);
  return fc::vectorabs(nx, ny, u, v, ff, f.v, undef);
}

int mifcref_relvort(int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr, float* out, int* fdefined, float undef)
{
  Flag f(fdefined);
  return fc::relvort(nx, ny, u, v, xmapr, ymapr, out, f.v, undef);
}

int mifcref_absvort(int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr, const float* fcoriolis, float* out,
                    int* fdefined, float undef)
{
  Flag f(fdefined);
  return fc::absvort(nx, ny, u, v, xmapr, ymapr, fcoriolis, out, f.v, undef);
}

int mifcref_divergence(int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr, float* out, int* fdefined, float undef)
{
  Flag f(fdefined);
  return fc::divergence(nx, ny, u, v, xmapr, ymapr, out, f.v, undef);
}

int mifcref_gradient(int nx, int ny, const float* field, const float* xmapr, const float* ymapr, int compute, float* out, int* fdefined, float undef)
{
  Flag f(fdefined);
  return fc::gradient(nx, ny, field, xmapr, ymapr, compute, out, f.v, undef);
}

int mifcref_plevelgwind_xcomp(int nx, int ny, const float* z, const float* xmapr, const float* ymapr, const float* fcoriolis, float* ug, int* fdefined,
                              float undef)
{
  Flag f(fdefined);
  return fc::plevelgwind_xcomp(nx, ny, z, xmapr, ymapr, fcoriolis, ug, f.v, undef);
}

int mifcref_plevelgwind_ycomp(int nx, int ny, const float* z, const float* xmapr, const float* ymapr, const float* fcoriolis, float* vg, int* fdefined,
                              float undef)
{
  Flag f(fdefined);
  return fc::plevelgwind_ycomp(nx, ny, z, xmapr, ymapr, fcoriolis, vg, f.v, undef);
}

int mifcref_plevelgvort(int nx, int ny, const float* z, const float* xmapr, const float* ymapr, const float* fcoriolis, float* gvort, int* fdefined,
                        float undef)
{
  Flag f(fdefined);
  return fc::plevelgvort(nx, ny, z, xmapr, ymapr, fcoriolis, gvort, f.v, undef);
}

int mifcref_ilevelgwind(int nx, int ny, const float* mpot, const float* xmapr, const float* ymapr, const float* fcoriolis, float* ug, float* vg,
                        int* fdefined, float undef)
{
  Flag f(fdefined);
  return fc::ilevelgwind(nx, ny, mpot, xmapr, ymapr, fcoriolis, ug, vg, f.v, undef);
}

int mifcref_pleveltemp(int nx, int ny, const float* tinp, float p, const char* unit, int compute, float* tout, int* fdefined, float undef)
{
  Flag f(fdefined);
  return fc::pleveltemp(nx, ny, tinp, p, std::string(unit), compute, tout, f.v, undef);
}

int mifcref_hleveltemp(int nx, int ny, const float* tinp, const float* ps, float alevel, float blevel, const char* unit, int compute, float* tout,
                       int* fdefined, float undef)
{
  Flag f(fdefined);
  return fc::hleveltemp(nx, ny, tinp, ps, alevel, blevel, std::string(unit), compute, tout, f.v, undef);
}

int mifcref_aleveltemp(int nx, int ny, const float* tinp, const float* p, const char* unit, int compute, float* tout, int* fdefined, float undef)
{
  Flag f(fdefined);
  return fc::aleveltemp(nx, ny, tinp, p, std::string(unit), compute, tout, f.v, undef);
}

int mifcref_plevelhum(int nx, int ny, const float* t, const float* huminp, float p, const char* unit, int compute, float* humout, int* fdefined,
                      float undef)
{
  Flag f(fdefined);
  return fc::plevelhum(nx, ny, t, huminp, p, std::string(unit), compute, humout, f.v, undef);
}

int mifcref_hlevelhum(int nx, int ny, const float* t, const float* huminp, const float* ps, float alevel, float blevel, const char* unit, int compute,
                      float* humout, int* fdefined, float undef)
{
  Flag f(fdefined);
  return fc::hlevelhum(nx, ny, t, huminp, ps, alevel, blevel, std::string(unit), compute, humout, f.v, undef);
}

int mifcref_alevelhum(int nx, int ny, const float* t, const float* huminp, const float* p, const char* unit, int compute, float* humout, int* fdefined,
                      float undef)
{
  Flag f(fdefined);
  return fc::alevelhum(nx, ny, t, huminp, p, std::string(unit), compute, humout, f.v, undef);
}

int mifcref_cvhum(int nx, int ny, const float* t, const float* huminp, const char* unit, int compute, float* humout, int* fdefined, float undef)
{
  Flag f(fdefined);
  return fc::cvhum(nx, ny, t, huminp, std::string(unit), compute, humout, f.v, undef);
}

int mifcref_advection(int nx, int ny, const float* f_, const float* u, const float* v, const float* xmapr, const float* ymapr, float hours, float* advec,
                      int* fdefined, float undef)
{
  Flag f(fdefined);
  return fc::advection(nx, ny, f_, u, v, xmapr, ymapr, hours, advec, f.v, undef);
}

int mifcref_jacobian(int nx, int ny, const float* field1, const float* field2, const float* xmapr, const float* ymapr, float* fjacobian, int* fdefined,
                     float undef)
{
  Flag f(fdefined);
  return fc::jacobian(nx, ny, field1, field2, xmapr, ymapr, fjacobian, f.v, undef);
}

int mifcref_momentumXcoordinate(int nx, int ny, const float* v, const float* xmapr, const float* fcoriolis, float fcoriolisMin, float* mxy, int* fdefined,
                                float undef)
{
  Flag f(fdefined);
  return fc::momentumXcoordinate(nx, ny, v, xmapr, fcoriolis, fcoriolisMin, mxy, f.v, undef);
}

int mifcref_momentumYcoordinate(int nx, int ny, const float* u, const float* ymapr, const float* fcoriolis, float fcoriolisMin, float* nxy, int* fdefined,
                                float undef)
{
  Flag f(fdefined);
  return fc::momentumYcoordinate(nx, ny, u, ymapr, fcoriolis, fcoriolisMin, nxy, f.v, undef);
}

int mifcref_thermalFrontParameter(int nx, int ny, const float* tx, const float* xmapr, const float* ymapr, float* tfp, int* fdefined, float undef)
{
  Flag f(fdefined);
  return fc::thermalFrontParameter(nx, ny, tx, xmapr, ymapr, tfp, f.v, undef);
}

int mifcref_plevelqvector(int nx, int ny, const float* z, const float* t, const float* xmapr, const float* ymapr, const float* fcoriolis, float p,
                          int compute, float* qcomp, int* fdefined, float undef)
{
  Flag f(fdefined);
  return fc::plevelqvector(nx, ny, z, t, xmapr, ymapr, fcoriolis, p, compute, qcomp, f.v, undef);
}

#define MIFC_STR0(x) #x
#define MIFC_STR(x) MIFC_STR0(x)
const char* mifcref_kind(void)
{
  return "reference " MIFC_STR(MI_FIELDCALC_VERSION_MAJOR) "." MIFC_STR(MI_FIELDCALC_VERSION_MINOR) "." MIFC_STR(MI_FIELDCALC_VERSION_PATCH);
}

} // extern "C"
