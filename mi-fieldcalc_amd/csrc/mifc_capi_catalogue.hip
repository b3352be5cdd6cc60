// mifc_capi_catalogue.hip -- C ABI of the rest of the pointwise catalogue
// (SURVEY.md 8f-3).  Host-side logic only: the reference's argument validation
// and its host-evaluated scalars (powf of a level pressure, unit factors), then
// one launch of the templated pointwise kernel (mifc_pointwise.hip).
#include "mifc_ctx.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

using namespace mifc_host;

namespace {

const float K_R = 287.f, K_P0 = 1000.f, K_EPS = (float)0.622, K_XLH = (float)2.501e+6, K_G = (float)9.8;

inline float pidcp_from_p(float p) // FieldCalculations.cc:308-311, host powf like the reference
{
  return powf(p * K_P0INV, K_KAPPA);
}
inline float pi_from_p(float p) // :313-316
{
  return K_CP * pidcp_from_p(p);
}

struct PwCall
{
  mifc::PwParams P; // op, compute, scalars and behaviour flags; pointers are filled in by run_pointwise
  int nin;
  const float* in[8];
  bool updates_flag; // the operator ends with fDefined = checkDefined(n_undefined, fsize)
};

PwCall pw_call(int op, int compute, int nin, const int* fdefined, float undef)
{
  PwCall pc;
  std::memset(&pc, 0, sizeof pc);
  pc.P.op = op;
  pc.P.compute = compute;
  pc.P.all_defined = (*fdefined == MIFC_ALL_DEFINED) ? 1 : 0;
  pc.P.undef = undef;
  pc.nin = nin;
  pc.updates_flag = true;
  return pc;
}

int run_pointwise(mifc_ctx* c, int nx, int ny, PwCall& pc, float* out, int* fdefined, int memkind)
{
  const long n64 = (long)nx * (long)ny;
  if (nx < 0 || ny < 0 || n64 > 0x7fffffffL || !out)
    return 0;
  const size_t n = (size_t)n64;
  bool ok = true;
  mifc::PwParams& P = pc.P;
  P.n = (int)n;
  P.count = pc.updates_flag ? 1 : 0;
  for (int k = 0; k < pc.nin; ++k) {
    if (!pc.in[k])
      return 0;
    P.in[k] = stage_in(c, k, pc.in[k], n, memkind, &ok);
  }
  P.out = stage_out(c, 9, out, n, memkind, &ok, P.may_keep != 0);
  if (!ok || !ensure_levels(c, 1))
    return 0;
  P.n_undefined = c->d_counts;
  if (P.count) {
    MIFC_HIP(c, hipMemsetAsync(c->d_counts, 0, sizeof(u64), c->stream));
    P.partials = partials_for(c, n, &P.partials_cap);
  }
  MIFC_LAUNCH(c, mifc::launch_pointwise(P, c->stream));
  if (P.count) {
    if (!pinned_acquire(c))
      return 0;
    MIFC_HIP(c, hipMemcpyAsync(pinned_counts(c), c->d_counts, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  }
  if (!fetch_out(c, 9, out, n, memkind))
    return 0;
  MIFC_HIP(c, hipStreamSynchronize(c->stream));
  if (P.count)
    *fdefined = mifc_classify(pinned_counts(c)[0], (u64)n);
  return 1;
}

// fillUndef, FieldCalculations.cc:76-82
int fill_undef(mifc_ctx* c, int nx, int ny, float* out, int* fdefined, float undef, int memkind)
{
  PwCall pc = pw_call(mifc::PW_FILL, 0, 0, fdefined, undef);
  pc.P.s[0] = undef;
  pc.updates_flag = false;
  if (!run_pointwise(c, nx, ny, pc, out, fdefined, memkind))
    return 0;
  *fdefined = MIFC_NONE_DEFINED;
  return 1;
}

// Per-cell reduction over `nfields` member fields.  Host
// members are staged next to each other in scratch slot 0; the table of member
// pointers and the per-member flags live in slot 8.
int run_ensemble(mifc_ctx* c, mifc::EnsembleParams P, int nx, int ny, const float* const* fields, const int* member_flags, int nfields, float* out,
                 int* fdefined_out, int memkind, bool may_keep)
{
  const long n64 = (long)nx * (long)ny;
  if (nx < 0 || ny < 0 || n64 > 0x7fffffffL || nfields < 0 || !out || (nfields > 0 && !fields))
    return 0;
  const size_t n = (size_t)n64;
  for (int j = 0; j < nfields; ++j)
    if (!fields[j])
      return 0;
  std::vector<const float*> table;
  std::vector<unsigned char> flags;
  try { // nothing may be thrown across the C ABI
    table.resize((size_t)nfields);
    flags.resize((size_t)nfields);
  } catch (...) {
    c->err = "out of host memory";
    return 0;
  }
  if (memkind == MIFC_MEM_HOST) {
    const size_t stride = (n + 3) & ~size_t(3); // keeps every staged member 16-byte aligned
    if (nfields > 0 && !ensure_slot(c, 0, (size_t)nfields * stride * sizeof(float)))
      return 0;
    for (int j = 0; j < nfields; ++j) {
      float* d = static_cast<float*>(c->slot[0]) + (size_t)j * stride;
      MIFC_HIP(c, hipMemcpyAsync(d, fields[j], n * sizeof(float), hipMemcpyHostToDevice, c->stream));
      table[(size_t)j] = d;
    }
  } else {
    for (int j = 0; j < nfields; ++j)
      table[(size_t)j] = fields[j];
  }
  bool ok = true;
  P.out = stage_out(c, 9, out, n, memkind, &ok, may_keep);
  const size_t table_bytes = ((size_t)nfields * sizeof(float*) + 15) & ~size_t(15);
  const bool inline_table = nfields <= 64; // pointers and flags ride in the kernel arguments
  if (!ok || !ensure_levels(c, 1) || (!inline_table && !ensure_slot(c, 8, table_bytes + (size_t)nfields + 16)))
    return 0;
  P.n_inline = inline_table ? 1 : 0;
  P.has_member_flags = member_flags ? 1 : 0;
  if (inline_table) {
    for (int j = 0; j < nfields; ++j) {
      P.fields_inline[j] = table[(size_t)j];
      P.flags_inline[j] = member_flags ? (unsigned char)member_flags[j] : 0;
    }
  } else if (nfields > 0) {
    MIFC_HIP(c, hipMemcpyAsync(c->slot[8], table.data(), (size_t)nfields * sizeof(float*), hipMemcpyHostToDevice, c->stream));
    if (member_flags) {
      for (int j = 0; j < nfields; ++j)
        flags[(size_t)j] = (unsigned char)member_flags[j];
      MIFC_HIP(c, hipMemcpyAsync(static_cast<char*>(c->slot[8]) + table_bytes, flags.data(), (size_t)nfields, hipMemcpyHostToDevice, c->stream));
    }
  }
  P.n = (int)n;
  P.first = 0;
  P.nfields = nfields;
  P.fields = inline_table ? nullptr : static_cast<const float* const*>(c->slot[8]);
  P.member_flags = (member_flags && !inline_table) ? reinterpret_cast<const unsigned char*>(static_cast<char*>(c->slot[8]) + table_bytes) : nullptr;
  P.vector_ok = (reinterpret_cast<size_t>(P.out) & 15u) == 0;
  for (int j = 0; j < nfields; ++j)
    P.vector_ok = P.vector_ok && (reinterpret_cast<size_t>(table[(size_t)j]) & 15u) == 0;
  P.n_undefined = c->d_counts;
  if (!pinned_acquire(c))
    return 0;
  MIFC_HIP(c, hipMemsetAsync(c->d_counts, 0, sizeof(u64), c->stream));
  MIFC_LAUNCH(c, mifc::launch_ensemble(P, c->stream));
  MIFC_HIP(c, hipMemcpyAsync(pinned_counts(c), c->d_counts, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  if (!fetch_out(c, 9, out, n, memkind))
    return 0;
  MIFC_HIP(c, hipStreamSynchronize(c->stream)); // also: `table` and `flags` were read by their copies
  *fdefined_out = mifc_classify(pinned_counts(c)[0], (u64)n);
  return 1;
}

#define CTX_OR_FAIL(c) \
  if (!(c))            \
    return 0;          \
  mifc_host::enter(c)

} // namespace

extern "C" {

// ------------------------------------------------------------- theta-e
int mifc_plevelthe(mifc_ctx* c, int nx, int ny, const float* t, const float* rh, float p, int compute, float* the, int* fdefined, float undef,
                   int memkind)
{
  CTX_OR_FAIL(c);
  if (compute != 1 && compute != 2) // :383
    return 0;
  if (p <= 0.0)
    return 0;
  const float pidcp = pidcp_from_p(p), pi = pidcp * K_CP; // :389-392
  PwCall pc = pw_call(mifc::PW_PLEVELTHE, compute, 2, fdefined, undef);
  pc.P.s[0] = (compute == 2) ? pidcp : 1;
  pc.P.s[1] = (float)(0.01 * (double)(K_XLH / pi) * (double)K_EPS / (double)p);
  pc.P.s[2] = 1 / pidcp;
  pc.in[0] = t;
  pc.in[1] = rh;
  return run_pointwise(c, nx, ny, pc, the, fdefined, memkind);
}

int mifc_hlevelthe(mifc_ctx* c, int nx, int ny, const float* t, const float* q, const float* ps, float alevel, float blevel, int compute, float* the,
                   int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  if (bad_hlevel(alevel, blevel)) // :1121
    return 0;
  PwCall pc = pw_call(mifc::PW_XLEVELTHE, compute, 3, fdefined, undef);
  pc.P.hybrid = 1;
  pc.P.s[0] = alevel;
  pc.P.s[1] = blevel;
  pc.P.may_keep = (compute != 1 && compute != 2); // :1132-1135: no branch writes the cell
  pc.in[0] = t;
  pc.in[1] = q;
  pc.in[2] = ps;
  return run_pointwise(c, nx, ny, pc, the, fdefined, memkind);
}

int mifc_alevelthe(mifc_ctx* c, int nx, int ny, const float* t, const float* q, const float* p, int compute, float* the, int* fdefined, float undef,
                   int memkind)
{
  CTX_OR_FAIL(c);
  if (compute != 1 && compute != 2) // :1370
    return 0;
  PwCall pc = pw_call(mifc::PW_XLEVELTHE, compute, 3, fdefined, undef);
  pc.in[0] = t;
  pc.in[1] = q;
  pc.in[2] = p;
  return run_pointwise(c, nx, ny, pc, the, fdefined, memkind);
}

// ------------------------------------------------------------- ducting
int mifc_plevelducting(mifc_ctx* c, int nx, int ny, const float* t, const float* h, float p, int compute, float* duct, int* fdefined, float undef,
                       int memkind)
{
  CTX_OR_FAIL(c);
  if (p <= 0) // :622
    return 0;
  if (compute < 1 || compute > 4) // :633-635
    return 0;
  PwCall pc = pw_call(mifc::PW_PDUCT, compute, 2, fdefined, undef);
  pc.P.s[0] = (compute % 2 == 0) ? pidcp_from_p(p) : 1; // :625
  pc.P.s[1] = p;
  pc.updates_flag = (compute >= 3); // :629 binaryFunctionFieldField leaves the flag alone, :632 the "Undef" variant classifies
  pc.in[0] = t;
  pc.in[1] = h;
  return run_pointwise(c, nx, ny, pc, duct, fdefined, memkind);
}

int mifc_hlevelducting(mifc_ctx* c, int nx, int ny, const float* t, const float* h, const float* ps, float alevel, float blevel, int compute,
                       float* duct, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  if (bad_hlevel(alevel, blevel)) // :1248
    return 0;
  PwCall pc = pw_call(mifc::PW_XDUCT, compute, 3, fdefined, undef);
  pc.P.hybrid = 1;
  pc.P.s[0] = alevel;
  pc.P.s[1] = blevel;
  pc.P.may_keep = (compute < 1 || compute > 4);
  pc.in[0] = t;
  pc.in[1] = h;
  pc.in[2] = ps;
  return run_pointwise(c, nx, ny, pc, duct, fdefined, memkind);
}

int mifc_alevelducting(mifc_ctx* c, int nx, int ny, const float* t, const float* h, const float* p, int compute, float* duct, int* fdefined,
                       float undef, int memkind)
{
  CTX_OR_FAIL(c);
  PwCall pc = pw_call(mifc::PW_XDUCT, compute, 3, fdefined, undef);
  pc.P.may_keep = (compute < 1 || compute > 4);
  pc.updates_flag = false; // :1488-1504: counted, never classified
  pc.in[0] = t;
  pc.in[1] = h;
  pc.in[2] = p;
  return run_pointwise(c, nx, ny, pc, duct, fdefined, memkind);
}

int mifc_hlevelpressure(mifc_ctx* c, int nx, int ny, const float* ps, float alevel, float blevel, float* p, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  if (bad_hlevel(alevel, blevel)) // :1286
    return 0;
  PwCall pc = pw_call(mifc::PW_HPRESSURE, 0, 1, fdefined, undef);
  pc.P.s[0] = alevel;
  pc.P.s[1] = blevel;
  pc.in[0] = ps;
  return run_pointwise(c, nx, ny, pc, p, fdefined, memkind);
}

int mifc_pleveldz2tmean(mifc_ctx* c, int nx, int ny, const float* z1, const float* z2, float p1, float p2, int compute, float* tmean, int* fdefined,
                        float undef, int memkind)
{
  CTX_OR_FAIL(c);
  if (p1 <= 0 || p2 <= 0 || p1 == p2) // :477
    return 0;
  const float pi1 = pi_from_p(p1), pi2 = pi_from_p(p2);
  float convert, tconvert;
  switch (compute) { // :484-499
  case 1:
    convert = (float)((double)K_G * 0.5 * (double)(pi1 + pi2) / (double)((pi2 - pi1) * K_CP));
    tconvert = -K_T0;
    break;
  case 2:
    convert = (float)((double)K_G * 0.5 * (double)(pi1 + pi2) / (double)((pi2 - pi1) * K_CP));
    tconvert = 0.f;
    break;
  case 3:
    convert = K_G / (pi2 - pi1);
    tconvert = 0.f;
    break;
  default:
    return 0;
  }
  PwCall pc = pw_call(mifc::PW_DZ2TMEAN, compute, 2, fdefined, undef);
  pc.P.s[0] = convert;
  pc.P.s[1] = tconvert;
  pc.updates_flag = false; // :502 binaryFunctionFieldField
  pc.in[0] = z1;
  pc.in[1] = z2;
  return run_pointwise(c, nx, ny, pc, tmean, fdefined, memkind);
}

// ------------------------------------------------------------- indices
int mifc_kIndex(mifc_ctx* c, int nx, int ny, const float* t500, const float* t700, const float* rh700, const float* t850, const float* rh850,
                float p500, float p700, float p850, int compute, float* kfield, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  if (p500 <= 0.0 || p500 >= p700 || p700 >= p850) // :764
    return 0;
  PwCall pc = pw_call(mifc::PW_KINDEX, compute, 5, fdefined, undef);
  switch (compute) { // :768-781
  case 1:
    pc.P.s[0] = pc.P.s[1] = pc.P.s[2] = 1.f;
    break;
  case 2:
    pc.P.s[0] = pidcp_from_p(p500);
    pc.P.s[1] = pidcp_from_p(p700);
    pc.P.s[2] = pidcp_from_p(p850);
    break;
  default:
    return 0;
  }
  pc.in[0] = t500;
  pc.in[1] = t700;
  pc.in[2] = rh700;
  pc.in[3] = t850;
  pc.in[4] = rh850;
  return run_pointwise(c, nx, ny, pc, kfield, fdefined, memkind);
}

int mifc_ductingIndex(mifc_ctx* c, int nx, int ny, const float* t850, const float* rh850, float p850, int compute, float* duct, int* fdefined,
                      float undef, int memkind)
{
  CTX_OR_FAIL(c);
  if (p850 <= 0.0) // :828
    return 0;
  PwCall pc = pw_call(mifc::PW_DUCTINDEX, compute, 2, fdefined, undef);
  switch (compute) { // :832-841
  case 1:
    pc.P.s[0] = 1.f;
    break;
  case 2:
    pc.P.s[0] = pidcp_from_p(p850);
    break;
  default:
    return 0;
  }
  pc.in[0] = t850;
  pc.in[1] = rh850;
  return run_pointwise(c, nx, ny, pc, duct, fdefined, memkind);
}

int mifc_showalterIndex(mifc_ctx* c, int nx, int ny, const float* t500, const float* t850, const float* rh850, float p500, float p850, int compute,
                        float* sfield, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  if (p500 <= 0.0 || p500 >= p850) // :902
    return 0;
  const float pi500 = pi_from_p(p500), pi850 = pi_from_p(p850);
  PwCall pc = pw_call(mifc::PW_SHOWALTER, compute, 3, fdefined, undef);
  switch (compute) { // :909-922
  case 1:
    pc.P.s[0] = 1.f;
    pc.P.s[1] = 1.f;
    pc.P.s[2] = K_CP * (K_CP / pi850) * (pi500 / K_CP);
    break;
  case 2:
    pc.P.s[0] = pi500 / K_CP;
    pc.P.s[1] = pi850 / K_CP;
    pc.P.s[2] = K_CP * (pi500 / K_CP);
    break;
  default:
    return 0;
  }
  pc.P.s[3] = p500;
  pc.P.s[4] = p850;
  pc.P.d[0] = 1.0 / (double)K_CP; // the kernel's float divisions by cp and p500 as double multiplications (see PW_SHOWALTER)
  pc.P.d[1] = (p500 > 1e-6f && p500 < 1e9f) ? 1.0 / (double)p500 : 0.0;
  pc.P.skip_undefined_input = 1; // :965-967: counted, cell not written
  pc.P.may_keep = 1;
  pc.in[0] = t500;
  pc.in[1] = t850;
  pc.in[2] = rh850;
  return run_pointwise(c, nx, ny, pc, sfield, fdefined, memkind);
}

int mifc_boydenIndex(mifc_ctx* c, int nx, int ny, const float* t700, const float* z700, const float* z1000, float p700, float p1000, int compute,
                     float* bfield, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  if (compute <= 0 || compute >= 3) // :990
    return 0;
  if (p700 <= 0.0 || p700 >= p1000) // :993
    return 0;
  const float pi700 = K_CP * powf(p700 / K_P0, K_R / K_CP); // :996
  PwCall pc = pw_call(mifc::PW_BOYDEN, compute, 3, fdefined, undef);
  pc.P.s[0] = (compute == 2) ? pi700 / K_CP : 1;
  pc.in[0] = t700;
  pc.in[1] = z700;
  pc.in[2] = z1000;
  return run_pointwise(c, nx, ny, pc, bfield, fdefined, memkind);
}

int mifc_sweatIndex(mifc_ctx* c, int nx, int ny, const float* t850, const float* t500, const float* td850, const float* td500, const float* u850,
                    const float* v850, const float* u500, const float* v500, float* sindex, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  PwCall pc = pw_call(mifc::PW_SWEAT, 0, 8, fdefined, undef);
  const float* in[8] = {t850, t500, td850, td500, u850, v850, u500, v500};
  for (int k = 0; k < 8; ++k)
    pc.in[k] = in[k];
  return run_pointwise(c, nx, ny, pc, sindex, fdefined, memkind);
}

// ------------------------------------------------------------- misc pointwise
int mifc_seaSoundSpeed(mifc_ctx* c, int nx, int ny, const float* t, const float* s, float z, int compute, float* soundspeed, int* fdefined,
                       float undef, int memkind)
{
  CTX_OR_FAIL(c);
  if (compute != 1 && compute != 2) // :1575
    return 0;
  PwCall pc = pw_call(mifc::PW_SOUNDSPEED, compute, 2, fdefined, undef);
  pc.P.s[0] = (compute == 1) ? 0 : K_T0;
  const double Z = fabsf(z); // :1581-1582
  pc.P.d[0] = 0.01635 * Z + 0.000000175 * Z * Z;
  pc.in[0] = t;
  pc.in[1] = s;
  return run_pointwise(c, nx, ny, pc, soundspeed, fdefined, memkind);
}

int mifc_cvtemp(mifc_ctx* c, int nx, int ny, const float* tinp, int compute, float* tout, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  float tconvert;
  switch (compute) { // :1620-1635
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
  if (!tinp || !tout || nx < 0 || ny < 0)
    return 0;
  const size_t n = (size_t)nx * (size_t)ny;
  if (compute == 3 || compute == 4) {
    // "convert only if the input seems to be in the other unit" (:1639-1660): mean of the defined cells
    bool ok = true;
    const float* d_in = stage_in(c, 0, tinp, n, memkind, &ok);
    if (!ok || !ensure_levels(c, 1) || !pinned_acquire(c))
      return 0;
    MIFC_HIP(c, hipMemsetAsync(c->d_counts + 1, 0, 2 * sizeof(u64), c->stream));
    MIFC_LAUNCH(c, mifc::launch_mean_defined(d_in, (int)n, *fdefined == MIFC_ALL_DEFINED, undef, reinterpret_cast<double*>(c->d_counts + 1),
                                          reinterpret_cast<unsigned long long*>(c->d_counts + 2), c->stream));
    MIFC_HIP(c, hipMemcpyAsync(pinned_counts(c) + 1, c->d_counts + 1, 2 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    MIFC_HIP(c, hipStreamSynchronize(c->stream));
    double sum;
    std::memcpy(&sum, pinned_counts(c) + 1, sizeof sum);
    const u64 navg = pinned_counts(c)[2];
    const float tavg = navg > 0 ? (float)(sum / (double)navg) : 0.f;
    if ((compute == 3 && tavg < K_T0 / 2.) || (compute == 4 && tavg > K_T0 / 2.)) {
      if (tout != tinp) { // :1653-1657: plain copy, flag untouched
        if (memkind == MIFC_MEM_DEVICE) {
          MIFC_HIP(c, hipMemcpyAsync(tout, tinp, n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
          MIFC_HIP(c, hipStreamSynchronize(c->stream));
        } else {
          std::memcpy(tout, tinp, n * sizeof(float));
        }
      }
      return 1;
    }
  }
  PwCall pc = pw_call(mifc::PW_ADDCONST, compute, 1, fdefined, undef);
  pc.P.s[0] = tconvert;
  pc.in[0] = tinp;
  return run_pointwise(c, nx, ny, pc, tout, fdefined, memkind);
}

int mifc_abshum(mifc_ctx* c, int nx, int ny, const float* t, const float* rhum, float* abshumout, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  PwCall pc = pw_call(mifc::PW_ABSHUM, 0, 2, fdefined, undef);
  pc.in[0] = t;
  pc.in[1] = rhum;
  return run_pointwise(c, nx, ny, pc, abshumout, fdefined, memkind);
}

int mifc_windCooling(mifc_ctx* c, int nx, int ny, const float* t, const float* u, const float* v, int compute, float* dtcool, int* fdefined,
                     float undef, int memkind)
{
  CTX_OR_FAIL(c);
  if (compute != 1 && compute != 2) // :2225
    return 0;
  PwCall pc = pw_call(mifc::PW_WINDCOOLING, compute, 3, fdefined, undef);
  pc.P.s[0] = (compute == 1) ? K_T0 : 0.f;
  pc.updates_flag = false; // :2207-2228: counted, never classified
  pc.in[0] = t;
  pc.in[1] = u;
  pc.in[2] = v;
  return run_pointwise(c, nx, ny, pc, dtcool, fdefined, memkind);
}

int mifc_underCooledRain(mifc_ctx* c, int nx, int ny, const float* precip, const float* snow, const float* tk, float precipMin, float snowRateMax,
                         float tcMax, float* undercooled, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  PwCall pc = pw_call(mifc::PW_UNDERCOOLED, 0, 3, fdefined, undef);
  pc.P.s[0] = precipMin;
  pc.P.s[1] = snowRateMax;
  pc.P.s[2] = tcMax + K_T0; // :2246
  pc.in[0] = precip;
  pc.in[1] = snow;
  pc.in[2] = tk;
  return run_pointwise(c, nx, ny, pc, undercooled, fdefined, memkind);
}

int mifc_pressure2FlightLevel(mifc_ctx* c, int nx, int ny, const float* pressure, float* flightlevel, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  PwCall pc = pw_call(mifc::PW_FLIGHTLEVEL, 0, 1, fdefined, undef);
  pc.in[0] = pressure;
  return run_pointwise(c, nx, ny, pc, flightlevel, fdefined, memkind);
}

int mifc_snow_in_cm(mifc_ctx* c, int nx, int ny, const float* snow_water, const float* tk2m, const float* td2m, float* snow_cm, int* fdefined,
                    float undef, int memkind)
{
  CTX_OR_FAIL(c);
  PwCall pc = pw_call(mifc::PW_SNOWCM, 0, 3, fdefined, undef);
  pc.in[0] = snow_water;
  pc.in[1] = tk2m;
  pc.in[2] = td2m;
  return run_pointwise(c, nx, ny, pc, snow_cm, fdefined, memkind);
}

int mifc_values2classes(mifc_ctx* c, int nx, int ny, const float* fvalue, float* fclass, const float* values, int nvalues, int* fdefined, float undef,
                        int memkind)
{
  CTX_OR_FAIL(c);
  if (nvalues < 2 || !values) // :2476
    return 0;
  // the class limits are a host vector in the reference; they always come from the host here too
  if (!ensure_slot(c, 8, (size_t)nvalues * sizeof(float)))
    return 0;
  MIFC_HIP(c, hipMemcpyAsync(c->slot[8], values, (size_t)nvalues * sizeof(float), hipMemcpyHostToDevice, c->stream));
  MIFC_HIP(c, hipStreamSynchronize(c->stream)); // `values` may be a temporary of the caller
  PwCall pc = pw_call(mifc::PW_CLASSES, 0, 1, fdefined, undef);
  pc.P.values = static_cast<const float*>(c->slot[8]);
  pc.P.nvalues = nvalues;
  pc.in[0] = fvalue;
  return run_pointwise(c, nx, ny, pc, fclass, fdefined, memkind);
}

// ------------------------------------------------------------- second-order Shapiro filter
// FieldCalculations.cc:2076-2179: four sweeps between the output and a scratch field; `field`
// and `fsmooth` may be the same array; the flag always becomes ALL_DEFINED (:2176).
int mifc_shapiro2_filter(mifc_ctx* c, int nx, int ny, const float* field, float* fsmooth, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  if (nx < 3 || ny < 3 || !field || !fsmooth) // :2093
    return 0;
  const size_t n = (size_t)nx * (size_t)ny;
  if (n > 0x7fffffffu)
    return 0;
  bool ok = true;
  const float* d_in = stage_in(c, 0, field, n, memkind, &ok);
  float* d_out = stage_out(c, 5, fsmooth, n, memkind, &ok);
  const bool all = (*fdefined == MIFC_ALL_DEFINED);
  if (!ok || !ensure_slot(c, 8, n * sizeof(float)))
    return 0;
  {
    // the four sweeps in one launch; it needs source and destination to be different arrays, so an
    // in-place call goes through the scratch field and is copied back
    // MIFC_SHAPIRO_FUSED=0: the four-launch path (A/B measurements, tests)
    float* d_dst = (d_out != d_in) ? d_out : static_cast<float*>(c->slot[8]);
    if (mifc::env().shapiro_fused && mifc::shapiro2_fused_supported(nx, ny, d_in, d_dst)) {
      MIFC_LAUNCH(c, mifc::launch_shapiro2_fused(nx, ny, all ? 1 : 0, undef, d_in, d_dst, c->stream));
      if (d_dst != d_out)
        MIFC_HIP(c, hipMemcpyAsync(d_out, d_dst, n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
      if (!fetch_out(c, 5, fsmooth, n, memkind))
        return 0;
      MIFC_HIP(c, hipStreamSynchronize(c->stream));
      *fdefined = MIFC_ALL_DEFINED;
      return 1;
    }
  }
  if (!all && !ensure_slot(c, 9, 2 * n))
    return 0;
  if (d_out != d_in)
    MIFC_HIP(c, hipMemcpyAsync(d_out, d_in, n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  mifc::ShapiroParams P;
  P.nx = nx;
  P.ny = ny;
  P.all_defined = all ? 1 : 0;
  P.undef = undef;
  P.f1 = d_out;
  P.f2 = static_cast<float*>(c->slot[8]);
  P.mask_x = all ? nullptr : static_cast<unsigned char*>(c->slot[9]);
  P.mask_y = all ? nullptr : static_cast<unsigned char*>(c->slot[9]) + n;
  MIFC_LAUNCH(c, mifc::launch_shapiro2(P, c->stream));
  if (!fetch_out(c, 5, fsmooth, n, memkind))
    return 0;
  MIFC_HIP(c, hipStreamSynchronize(c->stream));
  *fdefined = MIFC_ALL_DEFINED;
  return 1;
}

// ------------------------------------------------------------- vessel icing (closed-form models)
static int vessel_icing(mifc_ctx* c, int model, int nx, int ny, const float* airtemp, const float* seatemp, const float* u, const float* v,
                        const float* sal, const float* aice, float* icing, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  PwCall pc = pw_call(mifc::PW_VESSEL_ICING, model, 6, fdefined, undef);
  const float* in[6] = {airtemp, seatemp, u, v, sal, aice};
  for (int k = 0; k < 6; ++k)
    pc.in[k] = in[k];
  return run_pointwise(c, nx, ny, pc, icing, fdefined, memkind);
}
// EXTENSION: not a miutil::fieldcalc function (see include/mifc.h)
int mifc_winddir(mifc_ctx* c, int nx, int ny, const float* u, const float* v, float* dd, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  PwCall pc = pw_call(mifc::PW_WINDDIR, 0, 2, fdefined, undef);
  pc.in[0] = u;
  pc.in[1] = v;
  return run_pointwise(c, nx, ny, pc, dd, fdefined, memkind);
}

int mifc_vesselIcingOverland(mifc_ctx* c, int nx, int ny, const float* airtemp, const float* seatemp, const float* u, const float* v,
                             const float* sal, const float* aice, float* icing, int* fdefined, float undef, int memkind)
{
  return vessel_icing(c, 1, nx, ny, airtemp, seatemp, u, v, sal, aice, icing, fdefined, undef, memkind);
}
int mifc_vesselIcingMertins(mifc_ctx* c, int nx, int ny, const float* airtemp, const float* seatemp, const float* u, const float* v,
                            const float* sal, const float* aice, float* icing, int* fdefined, float undef, int memkind)
{
  return vessel_icing(c, 2, nx, ny, airtemp, seatemp, u, v, sal, aice, icing, fdefined, undef, memkind);
}

// ------------------------------------------------------------- field algebra
static int minmax_fields(mifc_ctx* c, int which, int nx, int ny, const float* f1, const float* f2, float* fres, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  PwCall pc = pw_call(mifc::PW_MINMAX_FIELDS, which, 2, fdefined, undef);
  pc.updates_flag = false;
  pc.in[0] = f1;
  pc.in[1] = f2;
  return run_pointwise(c, nx, ny, pc, fres, fdefined, memkind);
}
int mifc_minvalueFields(mifc_ctx* c, int nx, int ny, const float* field1, const float* field2, float* fres, int* fdefined, float undef, int memkind)
{
  return minmax_fields(c, 1, nx, ny, field1, field2, fres, fdefined, undef, memkind);
}
int mifc_maxvalueFields(mifc_ctx* c, int nx, int ny, const float* field1, const float* field2, float* fres, int* fdefined, float undef, int memkind)
{
  return minmax_fields(c, 2, nx, ny, field1, field2, fres, fdefined, undef, memkind);
}

static int unary_with_constant(mifc_ctx* c, int op, int compute, int nx, int ny, const float* field, float value, float* fres, int* fdefined,
                               float undef, int memkind, bool updates_flag = false)
{
  PwCall pc = pw_call(op, compute, 1, fdefined, undef);
  pc.P.s[0] = value;
  pc.updates_flag = updates_flag;
  pc.in[0] = field;
  return run_pointwise(c, nx, ny, pc, fres, fdefined, memkind);
}
int mifc_minvalueFieldConst(mifc_ctx* c, int nx, int ny, const float* field1, float value, float* fres, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  if (value == undef) // :2509
    return fill_undef(c, nx, ny, fres, fdefined, undef, memkind);
  return unary_with_constant(c, mifc::PW_MINMAX_CONST, 1, nx, ny, field1, value, fres, fdefined, undef, memkind);
}
int mifc_maxvalueFieldConst(mifc_ctx* c, int nx, int ny, const float* field1, float value, float* fres, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  if (value == undef) // :2524
    return fill_undef(c, nx, ny, fres, fdefined, undef, memkind);
  return unary_with_constant(c, mifc::PW_MINMAX_CONST, 2, nx, ny, field1, value, fres, fdefined, undef, memkind);
}
int mifc_absvalueField(mifc_ctx* c, int nx, int ny, const float* field, float* fres, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  return unary_with_constant(c, mifc::PW_MATH, 1, nx, ny, field, 0.f, fres, fdefined, undef, memkind);
}
int mifc_log10Field(mifc_ctx* c, int nx, int ny, const float* field, float* fres, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  return unary_with_constant(c, mifc::PW_MATH, 2, nx, ny, field, 0.f, fres, fdefined, undef, memkind);
}
int mifc_pow10Field(mifc_ctx* c, int nx, int ny, const float* field, float* fres, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  return unary_with_constant(c, mifc::PW_MATH, 3, nx, ny, field, 0.f, fres, fdefined, undef, memkind);
}
int mifc_logField(mifc_ctx* c, int nx, int ny, const float* field, float* fres, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  return unary_with_constant(c, mifc::PW_MATH, 4, nx, ny, field, 0.f, fres, fdefined, undef, memkind);
}
int mifc_expField(mifc_ctx* c, int nx, int ny, const float* field, float* fres, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  return unary_with_constant(c, mifc::PW_MATH, 5, nx, ny, field, 0.f, fres, fdefined, undef, memkind);
}
int mifc_powerField(mifc_ctx* c, int nx, int ny, const float* field, float value, float* fres, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  if (value == undef) // :2558
    return fill_undef(c, nx, ny, fres, fdefined, undef, memkind);
  return unary_with_constant(c, mifc::PW_MATH, 6, nx, ny, field, value, fres, fdefined, undef, memkind);
}

static int replace_cells(mifc_ctx* c, int compute, int nx, int ny, const float* field, float value, float* fres, int* fdefined, float undef,
                         int memkind)
{
  PwCall pc = pw_call(mifc::PW_REPLACE, compute, 1, fdefined, undef);
  pc.P.s[0] = value;
  pc.P.no_input_test = 1; // :2581, :2604 compare with undef only
  pc.updates_flag = false;
  pc.in[0] = field;
  return run_pointwise(c, nx, ny, pc, fres, fdefined, memkind);
}
static int fill_value(mifc_ctx* c, int nx, int ny, float value, float* fres, int* fdefined, float undef, int memkind)
{
  PwCall pc = pw_call(mifc::PW_FILL, 0, 0, fdefined, undef);
  pc.P.s[0] = value;
  pc.updates_flag = false;
  return run_pointwise(c, nx, ny, pc, fres, fdefined, memkind);
}
int mifc_replaceUndefined(mifc_ctx* c, int nx, int ny, const float* field, float value, float* fres, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  if (value == undef || *fdefined == MIFC_ALL_DEFINED) // :2567-2571: copy, flag untouched
    return (fres == field) ? 1 : replace_cells(c, 3, nx, ny, field, value, fres, fdefined, undef, memkind);
  int ok;
  if (*fdefined == MIFC_NONE_DEFINED) // :2573-2575
    ok = fill_value(c, nx, ny, value, fres, fdefined, undef, memkind);
  else
    ok = replace_cells(c, 1, nx, ny, field, value, fres, fdefined, undef, memkind);
  if (ok)
    *fdefined = MIFC_ALL_DEFINED; // :2584
  return ok;
}
int mifc_replaceDefined(mifc_ctx* c, int nx, int ny, const float* field, float value, float* fres, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  if (value == undef || *fdefined == MIFC_NONE_DEFINED) // :2589-2594
    return fill_undef(c, nx, ny, fres, fdefined, undef, memkind);
  int ok;
  if (*fdefined == MIFC_ALL_DEFINED) // :2596-2598
    ok = fill_value(c, nx, ny, value, fres, fdefined, undef, memkind);
  else
    ok = replace_cells(c, 2, nx, ny, field, value, fres, fdefined, undef, memkind);
  if (ok)
    *fdefined = MIFC_ALL_DEFINED; // :2607
  return ok;
}

int mifc_fieldOPERfield(mifc_ctx* c, int compute, int nx, int ny, const float* field1, const float* field2, float* fres, int* fdefined, float undef,
                        int memkind)
{
  CTX_OR_FAIL(c);
  if (compute < 1 || compute > 4) // :2622
    return 0;
  PwCall pc = pw_call(mifc::PW_FIELD_OP_FIELD, compute, 2, fdefined, undef);
  pc.updates_flag = (compute == 4); // :2621 only the division classifies
  pc.in[0] = field1;
  pc.in[1] = field2;
  return run_pointwise(c, nx, ny, pc, fres, fdefined, memkind);
}

int mifc_fieldOPERconstant(mifc_ctx* c, int compute, int nx, int ny, const float* field, float value, float* fres, int* fdefined, float undef,
                           int memkind)
{
  CTX_OR_FAIL(c);
  if ((value == undef) || (compute == 4 && value == 0)) // :2629, before `compute` is validated
    return fill_undef(c, nx, ny, fres, fdefined, undef, memkind);
  if (compute < 1 || compute > 4)
    return 0;
  return unary_with_constant(c, mifc::PW_FIELD_OP_CONST, compute, nx, ny, field, value, fres, fdefined, undef, memkind);
}

int mifc_constantOPERfield(mifc_ctx* c, int compute, int nx, int ny, float value, const float* field, float* fres, int* fdefined, float undef,
                           int memkind)
{
  CTX_OR_FAIL(c);
  if (value == undef) // :2651
    return fill_undef(c, nx, ny, fres, fdefined, undef, memkind);
  if (compute < 1 || compute > 4)
    return 0;
  return unary_with_constant(c, mifc::PW_CONST_OP_FIELD, compute, nx, ny, field, value, fres, fdefined, undef, memkind, compute == 4);
}

// ------------------------------------------------------------- ensemble reductions
static mifc::EnsembleParams ens_params(int op, int compute, float undef)
{
  mifc::EnsembleParams P;
  std::memset(&P, 0, sizeof P);
  P.op = op;
  P.compute = compute;
  P.undef = undef;
  return P;
}

int mifc_sumFields(mifc_ctx* c, int nx, int ny, const float* const* fields, int nfields, float* fres, int* fdefined, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  mifc::EnsembleParams P = ens_params(mifc::ENS_SUM, 0, undef);
  P.all_defined = (*fdefined == MIFC_ALL_DEFINED) ? 1 : 0;
  return run_ensemble(c, P, nx, ny, fields, nullptr, nfields, fres, fdefined, memkind, false);
}

int mifc_meanValue(mifc_ctx* c, int nx, int ny, const float* const* fields, const int* fdefined_in, int nfields, float* fres, int* fdefined_out,
                   float undef, int memkind)
{
  CTX_OR_FAIL(c);
  if (nfields > 0 && !fdefined_in)
    return 0;
  return run_ensemble(c, ens_params(mifc::ENS_MEAN, 0, undef), nx, ny, fields, fdefined_in, nfields, fres, fdefined_out, memkind, false);
}

int mifc_stddevValue(mifc_ctx* c, int nx, int ny, const float* const* fields, const int* fdefined_in, int nfields, float* fres, int* fdefined_out,
                     float undef, int memkind)
{
  CTX_OR_FAIL(c);
  if (nfields > 0 && !fdefined_in)
    return 0;
  return run_ensemble(c, ens_params(mifc::ENS_STDDEV, 0, undef), nx, ny, fields, fdefined_in, nfields, fres, fdefined_out, memkind, false);
}

int mifc_extremeValue(mifc_ctx* c, int compute, int nx, int ny, const float* const* fields, int nfields, float* fres, int* fdefined, float undef,
                      int memkind)
{
  CTX_OR_FAIL(c);
  if (nfields == 0) // :2769
    return 0;
  mifc::EnsembleParams P = ens_params(mifc::ENS_EXTREME, compute, undef);
  P.all_defined = (*fdefined == MIFC_ALL_DEFINED) ? 1 : 0;
  // compute outside 1..4: neither loop runs, nothing is written, the flag becomes ALL_DEFINED (:2803)
  return run_ensemble(c, P, nx, ny, fields, nullptr, nfields, fres, fdefined, memkind, !(compute >= 1 && compute <= 4));
}

int mifc_probability(mifc_ctx* c, int compute, int nx, int ny, const float* const* fields, const int* fdefined_in, int nfields, const float* limits,
                     int nlimits, float* fres, int* fdefined_out, float undef, int memkind)
{
  CTX_OR_FAIL(c);
  if (nfields > 0 && !fdefined_in)
    return 0;
  const bool check_between = (nlimits >= 2) && (compute == 3 || compute == 6); // :2821-2825
  const bool check_above = (nlimits >= 1) && (compute == 1 || compute == 4 || check_between);
  const bool check_below = (nlimits >= 1) && (compute == 2 || compute == 5 || check_between);
  if (!(check_above || check_below)) { // :2827-2833: everything undefined, and false
    int ignored = MIFC_SOME_DEFINED;
    (void)fill_undef(c, nx, ny, fres, &ignored, undef, memkind);
    *fdefined_out = MIFC_NONE_DEFINED;
    return 0;
  }
  mifc::EnsembleParams P = ens_params(mifc::ENS_PROBABILITY, compute, undef);
  P.check_above = check_above ? 1 : 0;
  P.check_below = check_below ? 1 : 0;
  P.value_above = limits[0];
  P.value_below = check_between ? limits[1] : limits[0];
  return run_ensemble(c, P, nx, ny, fields, fdefined_in, nfields, fres, fdefined_out, memkind, false);
}

} // extern "C"
