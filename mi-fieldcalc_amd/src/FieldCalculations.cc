// C++ operator API -> C ABI forwarders (host-side mirror of the reference's
// operator interface for the accelerated path).  Each function has the
// signature of its namesake in the reference's FieldCalculations.h and calls
// the mifc_* entry point that replaces it, with MIFC_MEM_HOST: legacy callers
// hand over host pointers, the library stages them through HBM.
//
// One GPU context per calling thread (the reference is re-entrant and its
// Python binding releases the GIL, python/py_mi_fieldcalc.cc:75, so concurrent
// callers are real).  Device ordinal: $MIFC_DEVICE, default 0.
#include "mi_fieldcalc/FieldCalculations.h"

#include "mifc.h"

#include <cstdlib>
#include <cstring>

namespace miutil {
namespace fieldcalc {

namespace {

struct ThreadContext
{
  mifc_ctx* ctx;
  ThreadContext()
  {
    const char* dev = std::getenv("MIFC_DEVICE");
    ctx = mifc_create(dev ? std::atoi(dev) : 0);
  }
  ~ThreadContext() { mifc_destroy(ctx); }
};

mifc_ctx* context()
{
  static thread_local ThreadContext tc;
  return tc.ctx; // null when no gfx950 device is usable: every operator then returns false
}

// ValuesDefined& <-> int* across the C boundary
struct FlagIO
{
  ValuesDefined& ref;
  int value;
  explicit FlagIO(ValuesDefined& f)
      : ref(f)
      , value(static_cast<int>(f))
  {
  }
  ~FlagIO() { ref = static_cast<ValuesDefined>(value); }
  int* ptr() { return &value; }
};

} // namespace

const char* last_error()
{
  return mifc_last_error(context());
}

bool hold_constant_field(const float* field, size_t fsize)
{
  return mifc_hold_field(context(), field, fsize) != 0;
}

bool release_constant_field(const float* field)
{
  return mifc_release_field(context(), field) != 0;
}

void copy_field(float* fout, const float* fin, size_t fsize)
{
  if (fout != fin)
    std::memcpy(fout, fin, sizeof(float) * fsize);
}

bool pleveltemp(int nx, int ny, const float* tinp, float p, const std::string& unit, int compute, float* tout, ValuesDefined& fDefined, float undef)
{
  FlagIO f(fDefined);
  return mifc_pleveltemp(context(), nx, ny, tinp, p, unit.c_str(), compute, tout, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool plevelhum(int nx, int ny, const float* t, const float* huminp, float p, const std::string& unit, int compute, float* humout,
               ValuesDefined& fDefined, float undef)
{
  FlagIO f(fDefined);
  return mifc_plevelhum(context(), nx, ny, t, huminp, p, unit.c_str(), compute, humout, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool plevelgwind_xcomp(int nx, int ny, const float* z, const float* xmapr, const float* ymapr, const float* fcoriolis, float* ug,
                       ValuesDefined& fDefined, float undef)
{
  FlagIO f(fDefined);
  return mifc_plevelgwind_xcomp(context(), nx, ny, z, xmapr, ymapr, fcoriolis, ug, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool plevelgwind_ycomp(int nx, int ny, const float* z, const float* xmapr, const float* ymapr, const float* fcoriolis, float* vg,
                       ValuesDefined& fDefined, float undef)
{
  FlagIO f(fDefined);
  return mifc_plevelgwind_ycomp(context(), nx, ny, z, xmapr, ymapr, fcoriolis, vg, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool plevelgvort(int nx, int ny, const float* z, const float* xmapr, const float* ymapr, const float* fcoriolis, float* gvort,
                 ValuesDefined& fDefined, float undef)
{
  FlagIO f(fDefined);
  return mifc_plevelgvort(context(), nx, ny, z, xmapr, ymapr, fcoriolis, gvort, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool hleveltemp(int nx, int ny, const float* tinp, const float* ps, float alevel, float blevel, const std::string& unit, int compute, float* tout,
                ValuesDefined& fDefined, float undef)
{
  FlagIO f(fDefined);
  return mifc_hleveltemp(context(), nx, ny, tinp, ps, alevel, blevel, unit.c_str(), compute, tout, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool hlevelhum(int nx, int ny, const float* t, const float* huminp, const float* ps, float alevel, float blevel, const std::string& unit, int compute,
               float* humout, ValuesDefined& fDefined, float undef)
{
  FlagIO f(fDefined);
  return mifc_hlevelhum(context(), nx, ny, t, huminp, ps, alevel, blevel, unit.c_str(), compute, humout, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool aleveltemp(int nx, int ny, const float* tinp, const float* p, const std::string& unit, int compute, float* tout, ValuesDefined& fDefined,
                float undef)
{
  FlagIO f(fDefined);
  return mifc_aleveltemp(context(), nx, ny, tinp, p, unit.c_str(), compute, tout, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool alevelhum(int nx, int ny, const float* t, const float* huminp, const float* p, const std::string& unit, int compute, float* humout,
               ValuesDefined& fDefined, float undef)
{
  FlagIO f(fDefined);
  return mifc_alevelhum(context(), nx, ny, t, huminp, p, unit.c_str(), compute, humout, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool ilevelgwind(int nx, int ny, const float* mpot, const float* xmapr, const float* ymapr, const float* fcoriolis, float* ug, float* vg,
                 ValuesDefined& fDefined, float undef)
{
  FlagIO f(fDefined);
  return mifc_ilevelgwind(context(), nx, ny, mpot, xmapr, ymapr, fcoriolis, ug, vg, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool cvhum(int nx, int ny, const float* t, const float* huminp, const std::string& unit, int compute, float* humout, ValuesDefined& fDefined,
           float undef)
{
  FlagIO f(fDefined);
  return mifc_cvhum(context(), nx, ny, t, huminp, unit.c_str(), compute, humout, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool vectorabs(int nx, int ny, const float* u, const float* v, float* ff, ValuesDefined& fDefined, float undef)
{
  FlagIO f(fDefined);
  return mifc_vectorabs(context(), nx, ny, u, v, ff, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool relvort(int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr, float* rvort, ValuesDefined& fDefined,
             float undef)
{
  FlagIO f(fDefined);
  return mifc_relvort(context(), nx, ny, u, v, xmapr, ymapr, rvort, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool absvort(int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr, const float* fcoriolis, float* avort,
             ValuesDefined& fDefined, float undef)
{
  FlagIO f(fDefined);
  return mifc_absvort(context(), nx, ny, u, v, xmapr, ymapr, fcoriolis, avort, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool divergence(int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr, float* diverg, ValuesDefined& fDefined,
                float undef)
{
  FlagIO f(fDefined);
  return mifc_divergence(context(), nx, ny, u, v, xmapr, ymapr, diverg, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool gradient(int nx, int ny, const float* field, const float* xmapr, const float* ymapr, int compute, float* fgrad, ValuesDefined& fDefined,
              float undef)
{
  FlagIO f(fDefined);
  return mifc_gradient(context(), nx, ny, field, xmapr, ymapr, compute, fgrad, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool advection(int nx, int ny, const float* f_, const float* u, const float* v, const float* xmapr, const float* ymapr, float hours, float* advec,
               ValuesDefined& fDefined, float undef)
{
  FlagIO f(fDefined);
  return mifc_advection(context(), nx, ny, f_, u, v, xmapr, ymapr, hours, advec, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool jacobian(int nx, int ny, const float* field1, const float* field2, const float* xmapr, const float* ymapr, float* fjacobian,
              ValuesDefined& fDefined, float undef)
{
  FlagIO f(fDefined);
  return mifc_jacobian(context(), nx, ny, field1, field2, xmapr, ymapr, fjacobian, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool momentumXcoordinate(int nx, int ny, const float* v, const float* xmapr, const float* fcoriolis, float fcoriolisMin, float* mxy,
                         ValuesDefined& fDefined, float undef)
{
  FlagIO f(fDefined);
  return mifc_momentumXcoordinate(context(), nx, ny, v, xmapr, fcoriolis, fcoriolisMin, mxy, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool momentumYcoordinate(int nx, int ny, const float* u, const float* ymapr, const float* fcoriolis, float fcoriolisMin, float* nxy,
                         ValuesDefined& fDefined, float undef)
{
  FlagIO f(fDefined);
  return mifc_momentumYcoordinate(context(), nx, ny, u, ymapr, fcoriolis, fcoriolisMin, nxy, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool thermalFrontParameter(int nx, int ny, const float* t, const float* xmapr, const float* ymapr, float* tfp, ValuesDefined& fDefined, float undef)
{
  FlagIO f(fDefined);
  return mifc_thermalFrontParameter(context(), nx, ny, t, xmapr, ymapr, tfp, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool plevelqvector(int nx, int ny, const float* z, const float* t, const float* xmapr, const float* ymapr, const float* fcoriolis, float p,
                   int compute, float* qcomp, ValuesDefined& fDefined, float undef)
{
  FlagIO f(fDefined);
  return mifc_plevelqvector(context(), nx, ny, z, t, xmapr, ymapr, fcoriolis, p, compute, qcomp, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

// ---- the rest of the pointwise catalogue: same pattern, one line each --------------
#define MIFC_FORWARD(call)                    \
  FlagIO f(fDefined);                         \
  return (call) != 0

bool plevelthe(int nx, int ny, const float* t, const float* rh, float p, int compute, float* the, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_plevelthe(context(), nx, ny, t, rh, p, compute, the, f.ptr(), undef, MIFC_MEM_HOST));
}
bool hlevelthe(int nx, int ny, const float* t, const float* q, const float* ps, float alevel, float blevel, int compute, float* the,
               ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_hlevelthe(context(), nx, ny, t, q, ps, alevel, blevel, compute, the, f.ptr(), undef, MIFC_MEM_HOST));
}
bool alevelthe(int nx, int ny, const float* t, const float* q, const float* p, int compute, float* the, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_alevelthe(context(), nx, ny, t, q, p, compute, the, f.ptr(), undef, MIFC_MEM_HOST));
}
bool plevelducting(int nx, int ny, const float* t, const float* h, float p, int compute, float* duct, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_plevelducting(context(), nx, ny, t, h, p, compute, duct, f.ptr(), undef, MIFC_MEM_HOST));
}
bool hlevelducting(int nx, int ny, const float* t, const float* h, const float* ps, float alevel, float blevel, int compute, float* duct,
                   ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_hlevelducting(context(), nx, ny, t, h, ps, alevel, blevel, compute, duct, f.ptr(), undef, MIFC_MEM_HOST));
}
bool alevelducting(int nx, int ny, const float* t, const float* h, const float* p, int compute, float* duct, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_alevelducting(context(), nx, ny, t, h, p, compute, duct, f.ptr(), undef, MIFC_MEM_HOST));
}
bool hlevelpressure(int nx, int ny, const float* ps, float alevel, float blevel, float* p, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_hlevelpressure(context(), nx, ny, ps, alevel, blevel, p, f.ptr(), undef, MIFC_MEM_HOST));
}
bool pleveldz2tmean(int nx, int ny, const float* z1, const float* z2, float p1, float p2, int compute, float* tmean, ValuesDefined& fDefined,
                    float undef)
{
  MIFC_FORWARD(mifc_pleveldz2tmean(context(), nx, ny, z1, z2, p1, p2, compute, tmean, f.ptr(), undef, MIFC_MEM_HOST));
}
bool kIndex(int nx, int ny, const float* t500, const float* t700, const float* rh700, const float* t850, const float* rh850, float p500, float p700,
            float p850, int compute, float* kfield, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_kIndex(context(), nx, ny, t500, t700, rh700, t850, rh850, p500, p700, p850, compute, kfield, f.ptr(), undef, MIFC_MEM_HOST));
}
bool ductingIndex(int nx, int ny, const float* t850, const float* rh850, float p850, int compute, float* duct, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_ductingIndex(context(), nx, ny, t850, rh850, p850, compute, duct, f.ptr(), undef, MIFC_MEM_HOST));
}
bool showalterIndex(int nx, int ny, const float* t500, const float* t850, const float* rh850, float p500, float p850, int compute, float* sfield,
                    ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_showalterIndex(context(), nx, ny, t500, t850, rh850, p500, p850, compute, sfield, f.ptr(), undef, MIFC_MEM_HOST));
}
bool boydenIndex(int nx, int ny, const float* t700, const float* z700, const float* z1000, float p700, float p1000, int compute, float* bfield,
                 ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_boydenIndex(context(), nx, ny, t700, z700, z1000, p700, p1000, compute, bfield, f.ptr(), undef, MIFC_MEM_HOST));
}
bool sweatIndex(int nx, int ny, const float* t850, const float* t500, const float* td850, const float* td500, const float* u850, const float* v850,
                const float* u500, const float* v500, float* sindex, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_sweatIndex(context(), nx, ny, t850, t500, td850, td500, u850, v850, u500, v500, sindex, f.ptr(), undef, MIFC_MEM_HOST));
}
bool seaSoundSpeed(int nx, int ny, const float* t, const float* s, float z, int compute, float* soundspeed, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_seaSoundSpeed(context(), nx, ny, t, s, z, compute, soundspeed, f.ptr(), undef, MIFC_MEM_HOST));
}
bool cvtemp(int nx, int ny, const float* tinp, int compute, float* tout, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_cvtemp(context(), nx, ny, tinp, compute, tout, f.ptr(), undef, MIFC_MEM_HOST));
}
bool abshum(int nx, int ny, const float* t, const float* rhum, float* abshumout, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_abshum(context(), nx, ny, t, rhum, abshumout, f.ptr(), undef, MIFC_MEM_HOST));
}
bool windCooling(int nx, int ny, const float* t, const float* u, const float* v, int compute, float* dtcool, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_windCooling(context(), nx, ny, t, u, v, compute, dtcool, f.ptr(), undef, MIFC_MEM_HOST));
}
bool underCooledRain(int nx, int ny, const float* precip, const float* snow, const float* tk, float precipMin, float snowRateMax, float tcMax,
                     float* undercooled, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_underCooledRain(context(), nx, ny, precip, snow, tk, precipMin, snowRateMax, tcMax, undercooled, f.ptr(), undef, MIFC_MEM_HOST));
}
bool pressure2FlightLevel(int nx, int ny, const float* pressure, float* flightlevel, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_pressure2FlightLevel(context(), nx, ny, pressure, flightlevel, f.ptr(), undef, MIFC_MEM_HOST));
}
bool snow_in_cm(int nx, int ny, const float* snow_water, const float* tk2m, const float* td2m, float* snow_cm, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_snow_in_cm(context(), nx, ny, snow_water, tk2m, td2m, snow_cm, f.ptr(), undef, MIFC_MEM_HOST));
}
bool values2classes(int nx, int ny, const float* fvalue, float* fclass, const std::vector<float>& values, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_values2classes(context(), nx, ny, fvalue, fclass, values.data(), static_cast<int>(values.size()), f.ptr(), undef, MIFC_MEM_HOST));
}

bool shapiro2_filter(int nx, int ny, float* field, float* fsmooth, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_shapiro2_filter(context(), nx, ny, field, fsmooth, f.ptr(), undef, MIFC_MEM_HOST));
}
bool vesselIcingOverland(int nx, int ny, const float* airtemp, const float* seatemp, const float* u, const float* v, const float* sal,
                         const float* aice, float* icing, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_vesselIcingOverland(context(), nx, ny, airtemp, seatemp, u, v, sal, aice, icing, f.ptr(), undef, MIFC_MEM_HOST));
}
bool vesselIcingMertins(int nx, int ny, const float* airtemp, const float* seatemp, const float* u, const float* v, const float* sal,
                        const float* aice, float* icing, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_vesselIcingMertins(context(), nx, ny, airtemp, seatemp, u, v, sal, aice, icing, f.ptr(), undef, MIFC_MEM_HOST));
}

// Not built on the GPU (FieldCalculationsVesselIcing.cc:182, :677; out of the hot-path scope): false, and
// last_error() says why -- an argument-validation failure leaves last_error() empty.
bool vesselIcingModStall(int, int, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                         const float*, const float*, const float*, const float, const float, const float, const float, float*, ValuesDefined&, float)
{
  return mifc_not_built(context(), "vesselIcingModStall") != 0;
}
bool vesselIcingMincog(int, int, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                       const float*, const float*, const float*, const float, const float, const float, const float, const int, float*, ValuesDefined&,
                       float)
{
  return mifc_not_built(context(), "vesselIcingMincog") != 0;
}

bool neighbourProbFunctions(int, int, const float*, const std::vector<float>&, int, float*, ValuesDefined&, float)
{
  return mifc_not_built(context(), "neighbourProbFunctions") != 0; // FieldCalculations.cc:2862
}
bool neighbourFunctions(int, int, const float*, const std::vector<float>&, int, float*, ValuesDefined&, float)
{
  return mifc_not_built(context(), "neighbourFunctions") != 0; // FieldCalculations.cc:2955
}

#define MIFC_FORWARD_VOID(call) \
  FlagIO f(fDefined);           \
  (void)(call)

void minvalueFields(int nx, int ny, const float* field1, const float* field2, float* fres, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD_VOID(mifc_minvalueFields(context(), nx, ny, field1, field2, fres, f.ptr(), undef, MIFC_MEM_HOST));
}
void minvalueFieldConst(int nx, int ny, const float* field1, const float value, float* fres, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD_VOID(mifc_minvalueFieldConst(context(), nx, ny, field1, value, fres, f.ptr(), undef, MIFC_MEM_HOST));
}
void maxvalueFields(int nx, int ny, const float* field1, const float* field2, float* fres, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD_VOID(mifc_maxvalueFields(context(), nx, ny, field1, field2, fres, f.ptr(), undef, MIFC_MEM_HOST));
}
void maxvalueFieldConst(int nx, int ny, const float* field1, const float value, float* fres, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD_VOID(mifc_maxvalueFieldConst(context(), nx, ny, field1, value, fres, f.ptr(), undef, MIFC_MEM_HOST));
}
void absvalueField(int nx, int ny, const float* field, float* fres, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD_VOID(mifc_absvalueField(context(), nx, ny, field, fres, f.ptr(), undef, MIFC_MEM_HOST));
}
void log10Field(int nx, int ny, const float* field, float* fres, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD_VOID(mifc_log10Field(context(), nx, ny, field, fres, f.ptr(), undef, MIFC_MEM_HOST));
}
void pow10Field(int nx, int ny, const float* field, float* fres, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD_VOID(mifc_pow10Field(context(), nx, ny, field, fres, f.ptr(), undef, MIFC_MEM_HOST));
}
void logField(int nx, int ny, const float* field, float* fres, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD_VOID(mifc_logField(context(), nx, ny, field, fres, f.ptr(), undef, MIFC_MEM_HOST));
}
void expField(int nx, int ny, const float* field, float* fres, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD_VOID(mifc_expField(context(), nx, ny, field, fres, f.ptr(), undef, MIFC_MEM_HOST));
}
void powerField(int nx, int ny, const float* field, float value, float* fres, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD_VOID(mifc_powerField(context(), nx, ny, field, value, fres, f.ptr(), undef, MIFC_MEM_HOST));
}
void replaceUndefined(int nx, int ny, const float* field, float value, float* fres, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD_VOID(mifc_replaceUndefined(context(), nx, ny, field, value, fres, f.ptr(), undef, MIFC_MEM_HOST));
}
void replaceDefined(int nx, int ny, const float* field, float value, float* fres, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD_VOID(mifc_replaceDefined(context(), nx, ny, field, value, fres, f.ptr(), undef, MIFC_MEM_HOST));
}
bool fieldOPERfield(int compute, int nx, int ny, const float* field1, const float* field2, float* fres, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_fieldOPERfield(context(), compute, nx, ny, field1, field2, fres, f.ptr(), undef, MIFC_MEM_HOST));
}
bool fieldOPERconstant(int compute, int nx, int ny, const float* field, float value, float* fres, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_fieldOPERconstant(context(), compute, nx, ny, field, value, fres, f.ptr(), undef, MIFC_MEM_HOST));
}
bool constantOPERfield(int compute, int nx, int ny, float value, const float* field, float* fres, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_constantOPERfield(context(), compute, nx, ny, value, field, fres, f.ptr(), undef, MIFC_MEM_HOST));
}

// ---- reductions over ensemble members ----------------------------------------------
namespace {
std::vector<int> flags_of(const std::vector<ValuesDefined>& v)
{
  std::vector<int> f(v.size());
  for (size_t j = 0; j < v.size(); ++j)
    f[j] = static_cast<int>(v[j]);
  return f;
}
} // namespace

bool sumFields(int nx, int ny, const std::vector<float*>& fields, float* fres, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_sumFields(context(), nx, ny, fields.data(), static_cast<int>(fields.size()), fres, f.ptr(), undef, MIFC_MEM_HOST));
}
bool meanValue(int nx, int ny, const std::vector<float*>& fields, const std::vector<ValuesDefined>& fDefinedIn, float* fres,
               ValuesDefined& fDefinedOut, float undef)
{
  if (fDefinedIn.size() < fields.size())
    return false;
  const std::vector<int> in = flags_of(fDefinedIn);
  FlagIO f(fDefinedOut);
  return mifc_meanValue(context(), nx, ny, fields.data(), in.data(), static_cast<int>(fields.size()), fres, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}
bool stddevValue(int nx, int ny, const std::vector<float*>& fields, const std::vector<ValuesDefined>& fDefinedIn, float* fres,
                 ValuesDefined& fDefinedOut, float undef)
{
  if (fDefinedIn.size() < fields.size())
    return false;
  const std::vector<int> in = flags_of(fDefinedIn);
  FlagIO f(fDefinedOut);
  return mifc_stddevValue(context(), nx, ny, fields.data(), in.data(), static_cast<int>(fields.size()), fres, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}
bool extremeValue(int compute, int nx, int ny, const std::vector<float*>& fields, float* fres, ValuesDefined& fDefined, float undef)
{
  MIFC_FORWARD(mifc_extremeValue(context(), compute, nx, ny, fields.data(), static_cast<int>(fields.size()), fres, f.ptr(), undef, MIFC_MEM_HOST));
}
bool probability(int compute, int nx, int ny, const std::vector<float*>& fields, const std::vector<ValuesDefined>& fDefinedIn,
                 const std::vector<float>& limits, float* fres, ValuesDefined& fDefinedOut, float undef)
{
  if (fDefinedIn.size() < fields.size())
    return false;
  const std::vector<int> in = flags_of(fDefinedIn);
  FlagIO f(fDefinedOut);
  return mifc_probability(context(), compute, nx, ny, fields.data(), in.data(), static_cast<int>(fields.size()), limits.data(),
                          static_cast<int>(limits.size()), fres, f.ptr(), undef, MIFC_MEM_HOST) != 0;
}

bool vortdiv_levels(int nx, int ny, int nlev, const float* u, const float* v, const float* xmapr, const float* ymapr, float* rvort, float* diverg,
                    std::vector<ValuesDefined>& fDefined, float undef)
{
  if (nlev < 1 || fDefined.size() < static_cast<size_t>(nlev))
    return false;
  std::vector<int> flags(nlev);
  for (int l = 0; l < nlev; ++l)
    flags[l] = static_cast<int>(fDefined[l]);
  const bool ok = mifc_vortdiv_levels(context(), nx, ny, nlev, u, v, xmapr, ymapr, rvort, diverg, flags.data(), undef, MIFC_MEM_HOST) != 0;
  if (ok)
    for (int l = 0; l < nlev; ++l)
      fDefined[l] = static_cast<ValuesDefined>(flags[l]);
  return ok;
}

} // namespace fieldcalc
} // namespace miutil
