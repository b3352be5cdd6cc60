// Operator API of mi-fieldcalc, MI355X (gfx950) implementation.
//
// Source-compatible with the reference header
// src/mi_fieldcalc/FieldCalculations.h for the operators on the accelerated
// hot path: same namespace, names, parameter order and failure behaviour
//   (nx, ny, const float* inputs..., scalars with "compute" last,
//    float* output(s), ValuesDefined& fDefined /*in+out*/, float undef)
// Every function below is a thin forwarder to the C ABI in include/mifc.h;
// the operator bodies are HIP kernels.  Pointers are HOST pointers exactly as
// with the reference library (fields are staged through the GPU per call); a
// caller that keeps its fields resident in HBM uses the mifc_* entry points
// (or the *_levels batched forms) with MIFC_MEM_DEVICE directly.
//
// Every function the reference header declares is declared here and exported
// by libmi-fieldcalc.so with the same mangled name, so existing callers compile
// and link unchanged.  Four of them are not built on the GPU and return false
// (they never compute on the CPU): vesselIcingModStall, vesselIcingMincog,
// neighbourProbFunctions, neighbourFunctions.
#ifndef MI_FIELDCALC_FIELDCALCULATIONS_H
#define MI_FIELDCALC_FIELDCALCULATIONS_H

#include "FieldDefined.h"

#include <cmath>
#include <cstddef>
#include <string>
#include <vector>

namespace miutil {
namespace fieldcalc {

// ---- per-cell definedness test ------------------------------------------------
inline bool is_defined(float in, float undef)
{
  return !std::isnan(in) && in != undef;
}

// is_defined(allDefined, in1, ..., inN, undef) for N = 1..10: true when the
// caller vouches for the inputs (allDefined) or when every in_k is neither
// NaN nor undef.  The last argument is always the undefined value.
namespace detail {
inline bool every_defined(const float* v, int n_values, float undef)
{
  for (int k = 0; k < n_values; ++k)
    if (!is_defined(v[k], undef))
      return false;
  return true;
}
} // namespace detail

template <typename... Floats>
inline bool is_defined(bool allDefined, float in1, Floats... rest_then_undef)
{
  static_assert(sizeof...(Floats) >= 1 && sizeof...(Floats) <= 10, "is_defined(allDefined, in1..in10, undef)");
  if (allDefined)
    return true;
  const float a[] = {in1, static_cast<float>(rest_then_undef)...};
  const int n = static_cast<int>(sizeof...(Floats)); // values are a[0..n-1], undef is a[n]
  return detail::every_defined(a, n, a[n]);
}

void copy_field(float* fout, const float* fin, size_t fsize);

// ---- pressure level -----------------------------------------------------------
bool pleveltemp(int nx, int ny, const float* tinp, float p, const std::string& unit, int compute,
                float* tout, ValuesDefined& fDefined, float undef);
bool plevelhum(int nx, int ny, const float* t, const float* huminp, float p, const std::string& unit, int compute,
               float* humout, ValuesDefined& fDefined, float undef);
bool plevelgwind_xcomp(int nx, int ny, const float* z, const float* xmapr, const float* ymapr, const float* fcoriolis,
                       float* ug, ValuesDefined& fDefined, float undef);
bool plevelgwind_ycomp(int nx, int ny, const float* z, const float* xmapr, const float* ymapr, const float* fcoriolis,
                       float* vg, ValuesDefined& fDefined, float undef);
bool plevelgvort(int nx, int ny, const float* z, const float* xmapr, const float* ymapr, const float* fcoriolis,
                 float* gvort, ValuesDefined& fDefined, float undef);

// ---- hybrid model level: p = alevel + blevel * ps[] ---------------------------
bool hleveltemp(int nx, int ny, const float* tinp, const float* ps, float alevel, float blevel, const std::string& unit, int compute,
                float* tout, ValuesDefined& fDefined, float undef);
bool hlevelhum(int nx, int ny, const float* t, const float* huminp, const float* ps, float alevel, float blevel, const std::string& unit,
               int compute, float* humout, ValuesDefined& fDefined, float undef);

// ---- model level with a pressure field ----------------------------------------
bool aleveltemp(int nx, int ny, const float* tinp, const float* p, const std::string& unit, int compute,
                float* tout, ValuesDefined& fDefined, float undef);
bool alevelhum(int nx, int ny, const float* t, const float* huminp, const float* p, const std::string& unit, int compute,
               float* humout, ValuesDefined& fDefined, float undef);

// ---- isentropic level ------------------------------------------------------------
bool ilevelgwind(int nx, int ny, const float* mpot, const float* xmapr, const float* ymapr, const float* fcoriolis,
                 float* ug, float* vg, ValuesDefined& fDefined, float undef);

// ---- level independent ---------------------------------------------------------
bool cvhum(int nx, int ny, const float* t, const float* huminp, const std::string& unit, int compute,
           float* humout, ValuesDefined& fDefined, float undef);
bool vectorabs(int nx, int ny, const float* u, const float* v, float* ff, ValuesDefined& fDefined, float undef);
bool relvort(int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr,
             float* rvort, ValuesDefined& fDefined, float undef);
bool absvort(int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr, const float* fcoriolis,
             float* avort, ValuesDefined& fDefined, float undef);
bool divergence(int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr,
                float* diverg, ValuesDefined& fDefined, float undef);
bool gradient(int nx, int ny, const float* field, const float* xmapr, const float* ymapr, int compute,
              float* fgrad, ValuesDefined& fDefined, float undef);

// ---- same family, next in line (advection .. thermal front parameter) ----------
bool advection(int nx, int ny, const float* f, const float* u, const float* v, const float* xmapr, const float* ymapr, float hours,
               float* advec, ValuesDefined& fDefined, float undef);
bool jacobian(int nx, int ny, const float* field1, const float* field2, const float* xmapr, const float* ymapr,
              float* fjacobian, ValuesDefined& fDefined, float undef);
bool momentumXcoordinate(int nx, int ny, const float* v, const float* xmapr, const float* fcoriolis, float fcoriolisMin,
                         float* mxy, ValuesDefined& fDefined, float undef);
bool momentumYcoordinate(int nx, int ny, const float* u, const float* ymapr, const float* fcoriolis, float fcoriolisMin,
                         float* nxy, ValuesDefined& fDefined, float undef);
bool thermalFrontParameter(int nx, int ny, const float* t, const float* xmapr, const float* ymapr,
                           float* tfp, ValuesDefined& fDefined, float undef);
bool plevelqvector(int nx, int ny, const float* z, const float* t, const float* xmapr, const float* ymapr, const float* fcoriolis,
                   float p, int compute, float* qcomp, ValuesDefined& fDefined, float undef);

// ---- the rest of the pointwise catalogue -----------------------------------------
bool plevelthe(int nx, int ny, const float* t, const float* rh, float p, int compute, float* the, ValuesDefined& fDefined, float undef);
bool hlevelthe(int nx, int ny, const float* t, const float* q, const float* ps, float alevel, float blevel, int compute,
               float* the, ValuesDefined& fDefined, float undef);
bool alevelthe(int nx, int ny, const float* t, const float* q, const float* p, int compute, float* the, ValuesDefined& fDefined, float undef);
bool plevelducting(int nx, int ny, const float* t, const float* h, float p, int compute, float* duct, ValuesDefined& fDefined, float undef);
bool hlevelducting(int nx, int ny, const float* t, const float* h, const float* ps, float alevel, float blevel, int compute,
                   float* duct, ValuesDefined& fDefined, float undef);
bool alevelducting(int nx, int ny, const float* t, const float* h, const float* p, int compute, float* duct, ValuesDefined& fDefined, float undef);
bool hlevelpressure(int nx, int ny, const float* ps, float alevel, float blevel, float* p, ValuesDefined& fDefined, float undef);
bool pleveldz2tmean(int nx, int ny, const float* z1, const float* z2, float p1, float p2, int compute, float* tmean, ValuesDefined& fDefined, float undef);
bool kIndex(int nx, int ny, const float* t500, const float* t700, const float* rh700, const float* t850, const float* rh850,
            float p500, float p700, float p850, int compute, float* kfield, ValuesDefined& fDefined, float undef);
bool ductingIndex(int nx, int ny, const float* t850, const float* rh850, float p850, int compute, float* duct, ValuesDefined& fDefined, float undef);
bool showalterIndex(int nx, int ny, const float* t500, const float* t850, const float* rh850, float p500, float p850, int compute,
                    float* sfield, ValuesDefined& fDefined, float undef);
bool boydenIndex(int nx, int ny, const float* t700, const float* z700, const float* z1000, float p700, float p1000, int compute,
                 float* bfield, ValuesDefined& fDefined, float undef);
bool sweatIndex(int nx, int ny, const float* t850, const float* t500, const float* td850, const float* td500, const float* u850, const float* v850,
                const float* u500, const float* v500, float* sindex, ValuesDefined& fDefined, float undef);
bool seaSoundSpeed(int nx, int ny, const float* t, const float* s, float z, int compute, float* soundspeed, ValuesDefined& fDefined, float undef);
bool cvtemp(int nx, int ny, const float* tinp, int compute, float* tout, ValuesDefined& fDefined, float undef);
bool abshum(int nx, int ny, const float* t, const float* rhum, float* abshumout, ValuesDefined& fDefined, float undef);
bool windCooling(int nx, int ny, const float* t, const float* u, const float* v, int compute, float* dtcool, ValuesDefined& fDefined, float undef);
bool underCooledRain(int nx, int ny, const float* precip, const float* snow, const float* tk, float precipMin, float snowRateMax,
                     float tcMax, float* undercooled, ValuesDefined& fDefined, float undef);
bool pressure2FlightLevel(int nx, int ny, const float* pressure, float* flightlevel, ValuesDefined& fDefined, float undef);
bool snow_in_cm(int nx, int ny, const float* snow_water, const float* tk2m, const float* td2m, float* snow_cm, ValuesDefined& fDefined, float undef);
bool values2classes(int nx, int ny, const float* fvalue, float* fclass, const std::vector<float>& values, ValuesDefined& fDefined, float undef);

bool shapiro2_filter(int nx, int ny, float* field, float* fsmooth, ValuesDefined& fDefined, float undef);
bool vesselIcingOverland(int nx, int ny, const float* airtemp, const float* seatemp, const float* u, const float* v, const float* sal,
                         const float* aice, float* icing, ValuesDefined& fDefined, float undef);
bool vesselIcingMertins(int nx, int ny, const float* airtemp, const float* seatemp, const float* u, const float* v, const float* sal,
                        const float* aice, float* icing, ValuesDefined& fDefined, float undef);
// The two iterative vessel-icing models are declared so that callers (the reference's
// pybind11 module among them) compile and link unchanged; they are not built on the GPU
// yet and return false without touching their outputs -- they do not compute on the CPU.
bool vesselIcingModStall(int nx, int ny, const float* sal, const float* wave, const float* x_wind, const float* y_wind, const float* airtemp,
                         const float* rh, const float* sst, const float* p, const float* Pw, const float* aice, const float* depth,
                         const float vs, const float alpha, const float zmin, const float zmax, float* icing, ValuesDefined& fDefined, float undef);
bool vesselIcingMincog(int nx, int ny, const float* sal, const float* wave, const float* x_wind, const float* y_wind, const float* airtemp,
                       const float* rh, const float* sst, const float* p, const float* Pw, const float* aice, const float* depth,
                       const float vs, const float alpha, const float zmin, const float zmax, const int alt, float* icing, ValuesDefined& fDefined,
                       float undef);

// ---- field algebra ----------------------------------------------------------------
void minvalueFields(int nx, int ny, const float* field1, const float* field2, float* fres, ValuesDefined& fDefined, float undef);
void minvalueFieldConst(int nx, int ny, const float* field1, const float value, float* fres, ValuesDefined& fDefined, float undef);
void maxvalueFields(int nx, int ny, const float* field1, const float* field2, float* fres, ValuesDefined& fDefined, float undef);
void maxvalueFieldConst(int nx, int ny, const float* field1, const float value, float* fres, ValuesDefined& fDefined, float undef);
void absvalueField(int nx, int ny, const float* field, float* fres, ValuesDefined& fDefined, float undef);
void log10Field(int nx, int ny, const float* field, float* fres, ValuesDefined& fDefined, float undef);
void pow10Field(int nx, int ny, const float* field, float* fres, ValuesDefined& fDefined, float undef);
void logField(int nx, int ny, const float* field, float* fres, ValuesDefined& fDefined, float undef);
void expField(int nx, int ny, const float* field, float* fres, ValuesDefined& fDefined, float undef);
void powerField(int nx, int ny, const float* field, float value, float* fres, ValuesDefined& fDefined, float undef);
void replaceUndefined(int nx, int ny, const float* field, float value, float* fres, ValuesDefined& fDefined, float undef);
void replaceDefined(int nx, int ny, const float* field, float value, float* fres, ValuesDefined& fDefined, float undef);
bool fieldOPERfield(int compute, int nx, int ny, const float* field1, const float* field2, float* fres, ValuesDefined& fDefined, float undef);
bool fieldOPERconstant(int compute, int nx, int ny, const float* field, float value, float* fres, ValuesDefined& fDefined, float undef);
bool constantOPERfield(int compute, int nx, int ny, float value, const float* field, float* fres, ValuesDefined& fDefined, float undef);

// ---- reductions over ensemble members ------------------------------------------------
bool sumFields(int nx, int ny, const std::vector<float*>& fields, float* fres, ValuesDefined& fDefined, float undef);
bool meanValue(int nx, int ny, const std::vector<float*>& fields, const std::vector<ValuesDefined>& fDefinedIn,
               float* fres, ValuesDefined& fDefinedOut, float undef);
bool stddevValue(int nx, int ny, const std::vector<float*>& fields, const std::vector<ValuesDefined>& fDefinedIn,
                 float* fres, ValuesDefined& fDefinedOut, float undef);
bool extremeValue(int compute, int nx, int ny, const std::vector<float*>& fields, float* fres, ValuesDefined& fDefined, float undef);
bool probability(int compute, int nx, int ny, const std::vector<float*>& fields, const std::vector<ValuesDefined>& fDefinedIn,
                 const std::vector<float>& limits, float* fres, ValuesDefined& fDefinedOut, float undef);

// Neighbourhood statistics: declared for link compatibility, not built on the GPU
// (outside the accelerated path); they return false and do not compute on the CPU.
bool neighbourProbFunctions(int nx, int ny, const float* field, const std::vector<float>& constants, int compute,
                            float* fres, ValuesDefined& fDefined, float undef);
bool neighbourFunctions(int nx, int ny, const float* field, const std::vector<float>& constants, int compute,
                        float* fres, ValuesDefined& fDefined, float undef);

// ---- extensions of this implementation (not in the reference) -----------------
// Fused relvort + divergence for nlev levels stored [nlev][ny][nx]; xmapr/ymapr
// are shared.  fDefined[l] in/out per level.  Either output may be null.
bool vortdiv_levels(int nx, int ny, int nlev, const float* u, const float* v, const float* xmapr, const float* ymapr,
                    float* rvort, float* diverg, std::vector<ValuesDefined>& fDefined, float undef);
// Declares a host array that is handed to many calls unchanged (xmapr, ymapr,
// fcoriolis of a grid): uploaded to the calling thread's GPU context once
// instead of once per call.  Its content must not change until released.
bool hold_constant_field(const float* field, size_t fsize);
bool release_constant_field(const float* field);
// Last error text of the calling thread's GPU context ("" if none).
const char* last_error();

} // namespace fieldcalc
} // namespace miutil

#endif // MI_FIELDCALC_FIELDCALCULATIONS_H
