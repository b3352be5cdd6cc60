/*
 * ref_shim_catalogue.cc -- TEST INFRASTRUCTURE, not product code.
 *
 * Flat C wrappers around the REAL reference operators of SURVEY.md 8f-3 / 8f-4
 * (the rest of the pointwise catalogue, the ensemble reductions).  Compiled
 * together with the reference's own translation units where they lie under
 * /root/reference (oracle/Makefile, target _ref/libmifc_ref.so); nothing of
 * the reference is copied here.  Same loading rules as ref_shim.cc.
 */
#define MIFC_ORACLE_PREFIX mifcref_
#include "oracle_abi.h"

#include <mi_fieldcalc/FieldCalculations.h>

#include <vector>

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

std::vector<float*> field_vector(const float* const* fields, int nfields)
{
  std::vector<float*> v(nfields);
  for (int j = 0; j < nfields; ++j)
    v[j] = const_cast<float*>(fields[j]); // the reference's signature is non-const; it only reads
  return v;
}

std::vector<miutil::ValuesDefined> flag_vector(const int* flags, int n)
{
  std::vector<miutil::ValuesDefined> v(n);
  for (int j = 0; j < n; ++j)
    v[j] = static_cast<miutil::ValuesDefined>(flags[j]);
  return v;
}
} // namespace

#define SHIM(call)    \
  Flag f(fdefined);   \
  return call

extern "C" {

int mifcref_plevelthe(int nx, int ny, const float* t, const float* rh, float p, int compute, float* the, int* fdefined, float undef)
{
  SHIM(fc::plevelthe(nx, ny, t, rh, p, compute, the, f.v, undef));
}
int mifcref_hlevelthe(int nx, int ny, const float* t, const float* q, const float* ps, float alevel, float blevel, int compute, float* the, int* fdefined,
                      float undef)
{
  SHIM(fc::hlevelthe(nx, ny, t, q, ps, alevel, blevel, compute, the, f.v, undef));
}
int mifcref_alevelthe(int nx, int ny, const float* t, const float* q, const float* p, int compute, float* the, int* fdefined, float undef)
{
  SHIM(fc::alevelthe(nx, ny, t, q, p, compute, the, f.v, undef));
}
int mifcref_plevelducting(int nx, int ny, const float* t, const float* h, float p, int compute, float* duct, int* fdefined, float undef)
{
  SHIM(fc::plevelducting(nx, ny, t, h, p, compute, duct, f.v, undef));
}
int mifcref_hlevelducting(int nx, int ny, const float* t, const float* h, const float* ps, float alevel, float blevel, int compute, float* duct,
                          int* fdefined, float undef)
{
  SHIM(fc::hlevelducting(nx, ny, t, h, ps, alevel, blevel, compute, duct, f.v, undef));
}
int mifcref_alevelducting(int nx, int ny, const float* t, const float* h, const float* p, int compute, float* duct, int* fdefined, float undef)
{
  SHIM(fc::alevelducting(nx, ny, t, h, p, compute, duct, f.v, undef));
}
int mifcref_hlevelpressure(int nx, int ny, const float* ps, float alevel, float blevel, float* p, int* fdefined, float undef)
{
  SHIM(fc::hlevelpressure(nx, ny, ps, alevel, blevel, p, f.v, undef));
}
int mifcref_pleveldz2tmean(int nx, int ny, const float* z1, const float* z2, float p1, float p2, int compute, float* tmean, int* fdefined, float undef)
{
  SHIM(fc::pleveldz2tmean(nx, ny, z1, z2, p1, p2, compute, tmean, f.v, undef));
}
int mifcref_kIndex(int nx, int ny, const float* t500, const float* t700, const float* rh700, const float* t850, const float* rh850, float p500,
                   float p700, float p850, int compute, float* kfield, int* fdefined, float undef)
{
  SHIM(fc::kIndex(nx, ny, t500, t700, rh700, t850, rh850, p500, p700, p850, compute, kfield, f.v, undef));
}
int mifcref_ductingIndex(int nx, int ny, const float* t850, const float* rh850, float p850, int compute, float* duct, int* fdefined, float undef)
{
  SHIM(fc::ductingIndex(nx, ny, t850, rh850, p850, compute, duct, f.v, undef));
}
int mifcref_showalterIndex(int nx, int ny, const float* t500, const float* t850, const float* rh850, float p500, float p850, int compute, float* sfield,
                           int* fdefined, float undef)
{
  SHIM(fc::showalterIndex(nx, ny, t500, t850, rh850, p500, p850, compute, sfield, f.v, undef));
}
int mifcref_boydenIndex(int nx, int ny, const float* t700, const float* z700, const float* z1000, float p700, float p1000, int compute, float* bfield,
                        int* fdefined, float undef)
{
  SHIM(fc::boydenIndex(nx, ny, t700, z700, z1000, p700, p1000, compute, bfield, f.v, undef));
}
int mifcref_sweatIndex(int nx, int ny, const float* t850, const float* t500, const float* td850, const float* td500, const float* u850, const float* v850,
                       const float* u500, const float* v500, float* sindex, int* fdefined, float undef)
{
  SHIM(fc::sweatIndex(nx, ny, t850, t500, td850, td500, u850, v850, u500, v500, sindex, f.v, undef));
}
int mifcref_seaSoundSpeed(int nx, int ny, const float* t, const float* s, float z, int compute, float* soundspeed, int* fdefined, float undef)
{
  SHIM(fc::seaSoundSpeed(nx, ny, t, s, z, compute, soundspeed, f.v, undef));
}
int mifcref_cvtemp(int nx, int ny, const float* tinp, int compute, float* tout, int* fdefined, float undef)
{
  SHIM(fc::cvtemp(nx, ny, tinp, compute, tout, f.v, undef));
}
int mifcref_abshum(int nx, int ny, const float* t, const float* rhum, float* abshumout, int* fdefined, float undef)
{
  SHIM(fc::abshum(nx, ny, t, rhum, abshumout, f.v, undef));
}
int mifcref_windCooling(int nx, int ny, const float* t, const float* u, const float* v, int compute, float* dtcool, int* fdefined, float undef)
{
  SHIM(fc::windCooling(nx, ny, t, u, v, compute, dtcool, f.v, undef));
}
int mifcref_underCooledRain(int nx, int ny, const float* precip, const float* snow, const float* tk, float precipMin, float snowRateMax, float tcMax,
                            float* undercooled, int* fdefined, float undef)
{
  SHIM(fc::underCooledRain(nx, ny, precip, snow, tk, precipMin, snowRateMax, tcMax, undercooled, f.v, undef));
}
int mifcref_pressure2FlightLevel(int nx, int ny, const float* pressure, float* flightlevel, int* fdefined, float undef)
{
  SHIM(fc::pressure2FlightLevel(nx, ny, pressure, flightlevel, f.v, undef));
}
int mifcref_snow_in_cm(int nx, int ny, const float* snow_water, const float* tk2m, const float* td2m, float* snow_cm, int* fdefined, float undef)
{
  SHIM(fc::snow_in_cm(nx, ny, snow_water, tk2m, td2m, snow_cm, f.v, undef));
}
int mifcref_values2classes(int nx, int ny, const float* fvalue, float* fclass, const float* values, int nvalues, int* fdefined, float undef)
{
  const std::vector<float> v(values, values + nvalues);
  SHIM(fc::values2classes(nx, ny, fvalue, fclass, v, f.v, undef));
}

int mifcref_vesselIcingOverland(int nx, int ny, const float* airtemp, const float* seatemp, const float* u, const float* v, const float* sal,
                                const float* aice, float* icing, int* fdefined, float undef)
{
  SHIM(fc::vesselIcingOverland(nx, ny, airtemp, seatemp, u, v, sal, aice, icing, f.v, undef));
}
int mifcref_vesselIcingMertins(int nx, int ny, const float* airtemp, const float* seatemp, const float* u, const float* v, const float* sal,
                               const float* aice, float* icing, int* fdefined, float undef)
{
  SHIM(fc::vesselIcingMertins(nx, ny, airtemp, seatemp, u, v, sal, aice, icing, f.v, undef));
}

int mifcref_shapiro2_filter(int nx, int ny, const float* field, float* fsmooth, int* fdefined, float undef)
{
  SHIM(fc::shapiro2_filter(nx, ny, const_cast<float*>(field), fsmooth, f.v, undef)); // only written when it IS fsmooth
}

int mifcref_minvalueFields(int nx, int ny, const float* field1, const float* field2, float* fres, int* fdefined, float undef)
{
  Flag f(fdefined);
  fc::minvalueFields(nx, ny, field1, field2, fres, f.v, undef);
  return 1;
}
int mifcref_maxvalueFields(int nx, int ny, const float* field1, const float* field2, float* fres, int* fdefined, float undef)
{
  Flag f(fdefined);
  fc::maxvalueFields(nx, ny, field1, field2, fres, f.v, undef);
  return 1;
}
int mifcref_minvalueFieldConst(int nx, int ny, const float* field1, float value, float* fres, int* fdefined, float undef)
{
  Flag f(fdefined);
  fc::minvalueFieldConst(nx, ny, field1, value, fres, f.v, undef);
  return 1;
}
int mifcref_maxvalueFieldConst(int nx, int ny, const float* field1, float value, float* fres, int* fdefined, float undef)
{
  Flag f(fdefined);
  fc::maxvalueFieldConst(nx, ny, field1, value, fres, f.v, undef);
  return 1;
}
#define SHIM_UNARY_VOID(name)                                                                              \
  int mifcref_##name(int nx, int ny, const float* field, float* fres, int* fdefined, float undef)          \
  {                                                                                                        \
    Flag f(fdefined);                                                                                      \
    fc::name(nx, ny, field, fres, f.v, undef);                                                             \
    return 1;                                                                                              \
  }
SHIM_UNARY_VOID(absvalueField)
SHIM_UNARY_VOID(log10Field)
SHIM_UNARY_VOID(pow10Field)
SHIM_UNARY_VOID(logField)
SHIM_UNARY_VOID(expField)
#define SHIM_CONST_VOID(name)                                                                                       \
  int mifcref_##name(int nx, int ny, const float* field, float value, float* fres, int* fdefined, float undef)      \
  {                                                                                                                 \
    Flag f(fdefined);                                                                                               \
    fc::name(nx, ny, field, value, fres, f.v, undef);                                                               \
    return 1;                                                                                                       \
  }
SHIM_CONST_VOID(powerField)
SHIM_CONST_VOID(replaceUndefined)
SHIM_CONST_VOID(replaceDefined)

int mifcref_fieldOPERfield(int compute, int nx, int ny, const float* field1, const float* field2, float* fres, int* fdefined, float undef)
{
  SHIM(fc::fieldOPERfield(compute, nx, ny, field1, field2, fres, f.v, undef));
}
int mifcref_fieldOPERconstant(int compute, int nx, int ny, const float* field, float value, float* fres, int* fdefined, float undef)
{
  SHIM(fc::fieldOPERconstant(compute, nx, ny, field, value, fres, f.v, undef));
}
int mifcref_constantOPERfield(int compute, int nx, int ny, float value, const float* field, float* fres, int* fdefined, float undef)
{
  SHIM(fc::constantOPERfield(compute, nx, ny, value, field, fres, f.v, undef));
}

int mifcref_sumFields(int nx, int ny, const float* const* fields, int nfields, float* fres, int* fdefined, float undef)
{
  SHIM(fc::sumFields(nx, ny, field_vector(fields, nfields), fres, f.v, undef));
}
int mifcref_meanValue(int nx, int ny, const float* const* fields, const int* fdefined_in, int nfields, float* fres, int* fdefined_out, float undef)
{
  Flag f(fdefined_out);
  return fc::meanValue(nx, ny, field_vector(fields, nfields), flag_vector(fdefined_in, nfields), fres, f.v, undef);
}
int mifcref_stddevValue(int nx, int ny, const float* const* fields, const int* fdefined_in, int nfields, float* fres, int* fdefined_out, float undef)
{
  Flag f(fdefined_out);
  return fc::stddevValue(nx, ny, field_vector(fields, nfields), flag_vector(fdefined_in, nfields), fres, f.v, undef);
}
int mifcref_extremeValue(int compute, int nx, int ny, const float* const* fields, int nfields, float* fres, int* fdefined, float undef)
{
  SHIM(fc::extremeValue(compute, nx, ny, field_vector(fields, nfields), fres, f.v, undef));
}
int mifcref_probability(int compute, int nx, int ny, const float* const* fields, const int* fdefined_in, int nfields, const float* limits, int nlimits,
                        float* fres, int* fdefined_out, float undef)
{
  Flag f(fdefined_out);
  const std::vector<float> lim(limits, limits + nlimits);
  return fc::probability(compute, nx, ny, field_vector(fields, nfields), flag_vector(fdefined_in, nfields), lim, fres, f.v, undef);
}

} // extern "C"
