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
