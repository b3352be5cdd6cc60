/*
 * mifc.h -- C ABI of the MI355X (gfx950) implementation of the mi-fieldcalc
 * hot path: the elementwise derived-variable operators and the horizontal
 * 5-point-stencil operators of miutil::fieldcalc.
 *
 * This is the drop-in boundary.  The reference has no FFI layer of its own:
 * its operator API is the set of C++ free functions declared in
 * src/mi_fieldcalc/FieldCalculations.h.  Each entry point below replaces one
 * of them (file:line cited per function) with the same argument order
 *   nx, ny, input fields, scalars ("compute" last), output field(s),
 *   fDefined (in/out), undef                      (FieldCalculations.h:102-107)
 * plus a leading context and a trailing `memkind`.  The source-compatible C++
 * header mi-fieldcalc_amd/include/mi_fieldcalc/FieldCalculations.h forwards
 * the original signatures to these; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - fields are row-major float32, index = j*nx + i, x fastest
 *     (FieldCalculations.cc:65-73); batched calls take [nlev][ny][nx]
 *     contiguous, level-major.
 *   - `int* fdefined` carries miutil::ValuesDefined (FieldDefined.h:41) and is
 *     IN/OUT exactly as in the reference: in = state of the inputs
 *     (MIFC_ALL_DEFINED enables the no-test fast path, FieldCalculations.h:47-50),
 *     out = state of the result (FieldDefined.cc:62-70).  Always host memory.
 *   - return value: 1 = success, 0 = the reference's `return false`
 *     (bad sizes / compute / pressure) or a HIP failure; in the latter case
 *     mifc_last_error() is non-empty.  Nothing throws across this boundary.
 *   - memkind says where the FIELD pointers live: MIFC_MEM_HOST (legacy
 *     callers: staged through the context's device scratch, synchronous) or
 *     MIFC_MEM_DEVICE (resident in HBM, no copies; the call still returns
 *     only after the result flag is known).  The *_enqueue variants never
 *     synchronise.
 *   - every operator body is a hand-written HIP kernel; there is no CPU
 *     fallback.  Without a usable gfx950 device mifc_create() returns NULL.
 *   - in-place use (output aliasing an input) is supported for the
 *     elementwise operators only, as in the reference.
 */
#ifndef MIFC_H
#define MIFC_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIFC_ABI_VERSION 1

/* miutil::ValuesDefined, FieldDefined.h:41 */
enum { MIFC_ALL_DEFINED = 0, MIFC_NONE_DEFINED = 1, MIFC_SOME_DEFINED = 2 };
enum { MIFC_MEM_HOST = 0, MIFC_MEM_DEVICE = 1 };

typedef struct mifc_ctx mifc_ctx;

/* ---- context ------------------------------------------------------------ */
int mifc_abi_version(void);
/* number of visible HIP devices (0 when there is no GPU / no driver) */
int mifc_device_count(void);
/* Binds to HIP device `device`; owns a stream, scratch and staging buffers.
 * NULL on failure.  A context may be used from one thread at a time; create
 * one per thread for concurrent callers (the reference is re-entrant). */
mifc_ctx* mifc_create(int device);
void mifc_destroy(mifc_ctx* ctx);
const char* mifc_last_error(const mifc_ctx* ctx);
/* Run subsequent work on a caller-owned hipStream_t, e.g. the PyTorch current
 * stream.  NULL means HIP's null (default) stream -- which is what PyTorch's
 * default stream is.  mifc_use_own_stream() switches back to the context's own
 * non-blocking stream (the initial state). */
/* Switching streams is ordered against work already enqueued on the old stream
 * that still reads context-owned scratch: the new stream waits for it. */
int mifc_set_stream(mifc_ctx* ctx, void* hip_stream);
int mifc_use_own_stream(mifc_ctx* ctx);
int mifc_synchronize(mifc_ctx* ctx);
/* Always returns 0 and leaves "<what>: not built on the GPU (...)" in mifc_last_error(): what the
 * source-compatible C++ API calls for the reference functions outside the hot-path scope
 * (neighbourFunctions, neighbourProbFunctions, vesselIcingModStall, vesselIcingMincog), so that
 * their `false` can be told from an argument-validation failure. */
int mifc_not_built(mifc_ctx* ctx, const char* what);
/* The library's tuning / diagnostic environment variables (MIFC_*: which of several
 * equivalent kernel forms runs; none changes a result) are read when a context is
 * created, never on a call path.  Re-read them (tests, A/B tools). */
int mifc_reload_env(mifc_ctx* ctx);
/* Device memory for callers that do not link HIP themselves. */
void* mifc_device_alloc(mifc_ctx* ctx, size_t bytes);
int mifc_device_free(mifc_ctx* ctx, void* dptr);
int mifc_copy_to_device(mifc_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int mifc_copy_to_host(mifc_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* Device memory for a batch that is allocated once and computed on many times, PLACED: which physical memory the
 * arrays a streaming kernel touches lie on decides its time by up to 12 %, stably for as long as they live (DESIGN.md
 * 4.1).  n_arrays arrays of bytes_each bytes; their addresses come back in arrays_out[n_arrays].
 *   MIFC_PLACE_AS_ALLOCATED  n_arrays plain allocations, as they come
 *   MIFC_PLACE_SEARCH        a pool of up to 48 arrays (never more than budget_bytes, 0 = no budget of the caller's, and
 *                            never more than 90 % of the free device memory) is allocated, index sets of it are timed with
 *                            the probe -- structured and random sets, then coordinate descent, at most 160 probes -- the
 *                            fastest set is kept and the rest freed (what mi-fieldcalc_amd/placement.py does for bench.py)
 *   MIFC_PLACE_VMM           opt-in: ONE physical allocation made and mapped by the library (hipMemCreate / hipMemMap),
 *                            the arrays a measured-good distance apart inside it; no search
 *                            (profiles/r03/experiments/vmm_placement_*.txt, and the caveat in mifc_placement.hip)
 * probe(user, arrays) -> milliseconds of a representative kernel over the candidate arrays, in arrays_out order (a negative
 * value aborts the call); NULL = the library times its own fused vorticity+divergence launch over the first four arrays
 * as (u, v, rvort, diverg) of a nx x ny x nlev batch, which must fit bytes_each.  report may be NULL.
 * Free with mifc_batch_free_placed (all arrays of one call together). */
enum { MIFC_PLACE_AS_ALLOCATED = 0, MIFC_PLACE_SEARCH = 1, MIFC_PLACE_VMM = 2 };
typedef float (*mifc_placement_probe_fn)(void* user, void* const* arrays);
typedef struct mifc_placement_report {
  int strategy, pool_size, probes;
  float as_allocated_ms; /* the first n_arrays arrays of the pool: what allocating the batch in one go gives */
  float chosen_ms, probe_ms_min, probe_ms_median, probe_ms_max;
  size_t array_distance_bytes; /* MIFC_PLACE_VMM: distance between the arrays inside the one allocation */
} mifc_placement_report;
int mifc_batch_alloc_placed(mifc_ctx* ctx, int n_arrays, size_t bytes_each, int strategy, size_t budget_bytes, int nx, int ny, int nlev,
                            mifc_placement_probe_fn probe, void* probe_user, void** arrays_out, mifc_placement_report* report);
int mifc_batch_free_placed(mifc_ctx* ctx, void** arrays, int n_arrays);
/* Host-pointer (MIFC_MEM_HOST) callers: declare a host array that is passed to
 * many calls unchanged -- xmapr, ymapr, fcoriolis of a grid (FieldCalculations.h:
 * every stencil operator takes them) -- so that it is uploaded once instead of
 * per call.  The content must not change while held; hold again to refresh,
 * release before freeing the array.  No counterpart in the reference, whose
 * operators read the caller's memory directly. */
int mifc_hold_field(mifc_ctx* ctx, const float* host_field, size_t n_floats);
int mifc_release_field(mifc_ctx* ctx, const float* host_field);
/* The asynchronous entry points zero the caller's undefined counters before they count (one small fill per call).
 * A caller that issues many small calls back to back -- or records them into a graph, where every fill is a node of
 * ~5 us -- can zero ALL its counters with one mifc_zero_counts_enqueue and switch the per-call fills off:
 * mifc_counts_accumulate(ctx, 1) makes mifc_stencil_levels_enqueue, mifc_vortdiv_levels*_enqueue and
 * mifc_hlevel_derived_*_enqueue ADD to the counters they are given (0 switches back; the synchronous entries and the
 * slab plans are not affected). */
int mifc_counts_accumulate(mifc_ctx* ctx, int on);
int mifc_zero_counts_enqueue(mifc_ctx* ctx, unsigned long long* counts_dev, size_t n);
/* miutil::checkDefined(size_t,size_t), FieldDefined.cc:62-70 */
int mifc_classify(unsigned long long n_undefined, unsigned long long n);

/* ---- elementwise, one field per call ------------------------------------ */
/* miutil::fieldcalc::vectorabs, FieldCalculations.h:204 / FieldCalculations.cc:1819 */
int mifc_vectorabs(mifc_ctx* ctx, int nx, int ny, const float* u, const float* v, float* ff, int* fdefined, float undef, int memkind);
/* pleveltemp FieldCalculations.h:113 / .cc:328; hleveltemp .h:154 / .cc:1046; aleveltemp .h:172 / .cc:1310 */
int mifc_pleveltemp(mifc_ctx* ctx, int nx, int ny, const float* tinp, float p, const char* unit, int compute, float* tout, int* fdefined,
                    float undef, int memkind);
int mifc_hleveltemp(mifc_ctx* ctx, int nx, int ny, const float* tinp, const float* ps, float alevel, float blevel, const char* unit, int compute,
                    float* tout, int* fdefined, float undef, int memkind);
int mifc_aleveltemp(mifc_ctx* ctx, int nx, int ny, const float* tinp, const float* p, const char* unit, int compute, float* tout, int* fdefined,
                    float undef, int memkind);
/* plevelhum FieldCalculations.h:117 / .cc:400; hlevelhum .h:160 / .cc:1145; alevelhum .h:176 / .cc:1394; cvhum .h:200 / .cc:1738 */
int mifc_plevelhum(mifc_ctx* ctx, int nx, int ny, const float* t, const float* huminp, float p, const char* unit, int compute, float* humout,
                   int* fdefined, float undef, int memkind);
int mifc_hlevelhum(mifc_ctx* ctx, int nx, int ny, const float* t, const float* huminp, const float* ps, float alevel, float blevel,
                   const char* unit, int compute, float* humout, int* fdefined, float undef, int memkind);
int mifc_alevelhum(mifc_ctx* ctx, int nx, int ny, const float* t, const float* huminp, const float* p, const char* unit, int compute,
                   float* humout, int* fdefined, float undef, int memkind);
int mifc_cvhum(mifc_ctx* ctx, int nx, int ny, const float* t, const float* huminp, const char* unit, int compute, float* humout, int* fdefined,
               float undef, int memkind);

/* ---- 5-point stencils, one field per call ------------------------------- */
/* relvort FieldCalculations.h:206 / .cc:1843; absvort .h:208 / .cc:1875; divergence .h:211 / .cc:1910 */
int mifc_relvort(mifc_ctx* ctx, int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr, float* rvort,
                 int* fdefined, float undef, int memkind);
int mifc_absvort(mifc_ctx* ctx, int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr, const float* fcoriolis,
                 float* avort, int* fdefined, float undef, int memkind);
int mifc_divergence(mifc_ctx* ctx, int nx, int ny, const float* u, const float* v, const float* xmapr, const float* ymapr, float* diverg,
                    int* fdefined, float undef, int memkind);
/* gradient FieldCalculations.h:216 / .cc:1985 (compute 1..4) */
int mifc_gradient(mifc_ctx* ctx, int nx, int ny, const float* field, const float* xmapr, const float* ymapr, int compute, float* fgrad,
                  int* fdefined, float undef, int memkind);
/* plevelgwind_xcomp .h:127 / .cc:638; plevelgwind_ycomp .h:130 / .cc:674; plevelgvort .h:133 / .cc:708 */
int mifc_plevelgwind_xcomp(mifc_ctx* ctx, int nx, int ny, const float* z, const float* xmapr, const float* ymapr, const float* fcoriolis, float* ug,
                           int* fdefined, float undef, int memkind);
int mifc_plevelgwind_ycomp(mifc_ctx* ctx, int nx, int ny, const float* z, const float* xmapr, const float* ymapr, const float* fcoriolis, float* vg,
                           int* fdefined, float undef, int memkind);
int mifc_plevelgvort(mifc_ctx* ctx, int nx, int ny, const float* z, const float* xmapr, const float* ymapr, const float* fcoriolis, float* gvort,
                     int* fdefined, float undef, int memkind);
/* ilevelgwind FieldCalculations.h:185 / .cc:1511 */
int mifc_ilevelgwind(mifc_ctx* ctx, int nx, int ny, const float* mpot, const float* xmapr, const float* ymapr, const float* fcoriolis, float* ug,
                     float* vg, int* fdefined, float undef, int memkind);

/* ---- next operators of the same family, one field per call -- */
/* advection FieldCalculations.h:213 / .cc:1942; jacobian .h:235 / .cc:2424 */
int mifc_advection(mifc_ctx* ctx, int nx, int ny, const float* f, const float* u, const float* v, const float* xmapr, const float* ymapr, float hours,
                   float* advec, int* fdefined, float undef, int memkind);
int mifc_jacobian(mifc_ctx* ctx, int nx, int ny, const float* field1, const float* field2, const float* xmapr, const float* ymapr, float* fjacobian,
                  int* fdefined, float undef, int memkind);
/* momentumXcoordinate .h:229 / .cc:2351; momentumYcoordinate .h:232 / .cc:2387 (pointwise) */
int mifc_momentumXcoordinate(mifc_ctx* ctx, int nx, int ny, const float* v, const float* xmapr, const float* fcoriolis, float fcoriolisMin, float* mxy,
                             int* fdefined, float undef, int memkind);
int mifc_momentumYcoordinate(mifc_ctx* ctx, int nx, int ny, const float* u, const float* ymapr, const float* fcoriolis, float fcoriolisMin, float* nxy,
                             int* fdefined, float undef, int memkind);
/* thermalFrontParameter .h:225 / .cc:2266 (two passes: |grad T|, then the front parameter) */
int mifc_thermalFrontParameter(mifc_ctx* ctx, int nx, int ny, const float* tx, const float* xmapr, const float* ymapr, float* tfp, int* fdefined,
                               float undef, int memkind);
/* plevelqvector .h:122 / .cc:505 (three passes: geostrophic wind x, y, then the Q-vector component;
 * compute 1/2 = x component from T / theta, 3/4 = y component) */
int mifc_plevelqvector(mifc_ctx* ctx, int nx, int ny, const float* z, const float* t, const float* xmapr, const float* ymapr,
                       const float* fcoriolis, float p, int compute, float* qcomp, int* fdefined, float undef, int memkind);

/* ---- the rest of the pointwise catalogue, one field per call ----
 * Each entry replaces the miutil::fieldcalc function of the same name; the
 * comment gives FieldCalculations.h:line / FieldCalculations.cc:line.  Argument
 * order is the reference's (fieldOPER* take `compute` first, like there).  The
 * reference's void functions return 1 here. */
/* plevelthe .h:115 / .cc:369; hlevelthe .h:157 / .cc:1100; alevelthe .h:174 / .cc:1355 */
int mifc_plevelthe(mifc_ctx* ctx, int nx, int ny, const float* t, const float* rh, float p, int compute, float* the, int* fdefined, float undef,
                   int memkind);
int mifc_hlevelthe(mifc_ctx* ctx, int nx, int ny, const float* t, const float* q, const float* ps, float alevel, float blevel, int compute,
                   float* the, int* fdefined, float undef, int memkind);
int mifc_alevelthe(mifc_ctx* ctx, int nx, int ny, const float* t, const float* q, const float* p, int compute, float* the, int* fdefined, float undef,
                   int memkind);
/* plevelducting .h:125 / .cc:597; hlevelducting .h:163 / .cc:1219; alevelducting .h:179 / .cc:1460 */
int mifc_plevelducting(mifc_ctx* ctx, int nx, int ny, const float* t, const float* h, float p, int compute, float* duct, int* fdefined, float undef,
                       int memkind);
int mifc_hlevelducting(mifc_ctx* ctx, int nx, int ny, const float* t, const float* h, const float* ps, float alevel, float blevel, int compute,
                       float* duct, int* fdefined, float undef, int memkind);
int mifc_alevelducting(mifc_ctx* ctx, int nx, int ny, const float* t, const float* h, const float* p, int compute, float* duct, int* fdefined,
                       float undef, int memkind);
/* hlevelpressure .h:166 / .cc:1276; pleveldz2tmean .h:120 / .cc:466 */
int mifc_hlevelpressure(mifc_ctx* ctx, int nx, int ny, const float* ps, float alevel, float blevel, float* p, int* fdefined, float undef,
                        int memkind);
int mifc_pleveldz2tmean(mifc_ctx* ctx, int nx, int ny, const float* z1, const float* z2, float p1, float p2, int compute, float* tmean,
                        int* fdefined, float undef, int memkind);
/* kIndex .h:136 / .cc:745; ductingIndex .h:139 / .cc:816; showalterIndex .h:141 / .cc:872; boydenIndex .h:144 / .cc:973; sweatIndex .h:147 / .cc:1016 */
int mifc_kIndex(mifc_ctx* ctx, int nx, int ny, const float* t500, const float* t700, const float* rh700, const float* t850, const float* rh850,
                float p500, float p700, float p850, int compute, float* kfield, int* fdefined, float undef, int memkind);
int mifc_ductingIndex(mifc_ctx* ctx, int nx, int ny, const float* t850, const float* rh850, float p850, int compute, float* duct, int* fdefined,
                      float undef, int memkind);
int mifc_showalterIndex(mifc_ctx* ctx, int nx, int ny, const float* t500, const float* t850, const float* rh850, float p500, float p850, int compute,
                        float* sfield, int* fdefined, float undef, int memkind);
int mifc_boydenIndex(mifc_ctx* ctx, int nx, int ny, const float* t700, const float* z700, const float* z1000, float p700, float p1000, int compute,
                     float* bfield, int* fdefined, float undef, int memkind);
int mifc_sweatIndex(mifc_ctx* ctx, int nx, int ny, const float* t850, const float* t500, const float* td850, const float* td500, const float* u850,
                    const float* v850, const float* u500, const float* v500, float* sindex, int* fdefined, float undef, int memkind);
/* seaSoundSpeed .h:192 / .cc:1555; cvtemp .h:198 / .cc:1608; abshum .h:202 / .cc:1676; windCooling .h:220 / .cc:2181;
 * underCooledRain .h:222 / .cc:2231; pressure2FlightLevel .h:227 / .cc:2311; snow_in_cm .h:303 / .cc:3063 */
int mifc_seaSoundSpeed(mifc_ctx* ctx, int nx, int ny, const float* t, const float* s, float z, int compute, float* soundspeed, int* fdefined,
                       float undef, int memkind);
int mifc_cvtemp(mifc_ctx* ctx, int nx, int ny, const float* tinp, int compute, float* tout, int* fdefined, float undef, int memkind);
int mifc_abshum(mifc_ctx* ctx, int nx, int ny, const float* t, const float* rhum, float* abshumout, int* fdefined, float undef, int memkind);
int mifc_windCooling(mifc_ctx* ctx, int nx, int ny, const float* t, const float* u, const float* v, int compute, float* dtcool, int* fdefined,
                     float undef, int memkind);
int mifc_underCooledRain(mifc_ctx* ctx, int nx, int ny, const float* precip, const float* snow, const float* tk, float precipMin, float snowRateMax,
                         float tcMax, float* undercooled, int* fdefined, float undef, int memkind);
int mifc_pressure2FlightLevel(mifc_ctx* ctx, int nx, int ny, const float* pressure, float* flightlevel, int* fdefined, float undef, int memkind);
int mifc_snow_in_cm(mifc_ctx* ctx, int nx, int ny, const float* snow_water, const float* tk2m, const float* td2m, float* snow_cm, int* fdefined,
                    float undef, int memkind);
/* values2classes .h:252 / .cc:2462: `values` (class limits, std::vector<float> there) is always a HOST array */
int mifc_values2classes(mifc_ctx* ctx, int nx, int ny, const float* fvalue, float* fclass, const float* values, int nvalues, int* fdefined,
                        float undef, int memkind);
/* shapiro2_filter .h:218 / .cc:2076 (second-order Shapiro filter, four sweeps; field == fsmooth allowed,
 * the flag always becomes ALL_DEFINED) */
int mifc_shapiro2_filter(mifc_ctx* ctx, int nx, int ny, const float* field, float* fsmooth, int* fdefined, float undef, int memkind);
/* vesselIcingOverland .h:238 / FieldCalculationsVesselIcing.cc:77; vesselIcingMertins .h:241 / :114
 * (vesselIcingModStall .h:244 and vesselIcingMincog .h:248, the two iterative models, are not built yet) */
int mifc_vesselIcingOverland(mifc_ctx* ctx, int nx, int ny, const float* airtemp, const float* seatemp, const float* u, const float* v,
                             const float* sal, const float* aice, float* icing, int* fdefined, float undef, int memkind);
int mifc_vesselIcingMertins(mifc_ctx* ctx, int nx, int ny, const float* airtemp, const float* seatemp, const float* u, const float* v,
                            const float* sal, const float* aice, float* icing, int* fdefined, float undef, int memkind);
/* field algebra .h:254-282 / .cc:2501-2669 */
int mifc_minvalueFields(mifc_ctx* ctx, int nx, int ny, const float* field1, const float* field2, float* fres, int* fdefined, float undef, int memkind);
int mifc_maxvalueFields(mifc_ctx* ctx, int nx, int ny, const float* field1, const float* field2, float* fres, int* fdefined, float undef, int memkind);
int mifc_minvalueFieldConst(mifc_ctx* ctx, int nx, int ny, const float* field1, float value, float* fres, int* fdefined, float undef, int memkind);
int mifc_maxvalueFieldConst(mifc_ctx* ctx, int nx, int ny, const float* field1, float value, float* fres, int* fdefined, float undef, int memkind);
int mifc_absvalueField(mifc_ctx* ctx, int nx, int ny, const float* field, float* fres, int* fdefined, float undef, int memkind);
int mifc_log10Field(mifc_ctx* ctx, int nx, int ny, const float* field, float* fres, int* fdefined, float undef, int memkind);
int mifc_pow10Field(mifc_ctx* ctx, int nx, int ny, const float* field, float* fres, int* fdefined, float undef, int memkind);
int mifc_logField(mifc_ctx* ctx, int nx, int ny, const float* field, float* fres, int* fdefined, float undef, int memkind);
int mifc_expField(mifc_ctx* ctx, int nx, int ny, const float* field, float* fres, int* fdefined, float undef, int memkind);
int mifc_powerField(mifc_ctx* ctx, int nx, int ny, const float* field, float value, float* fres, int* fdefined, float undef, int memkind);
int mifc_replaceUndefined(mifc_ctx* ctx, int nx, int ny, const float* field, float value, float* fres, int* fdefined, float undef, int memkind);
int mifc_replaceDefined(mifc_ctx* ctx, int nx, int ny, const float* field, float value, float* fres, int* fdefined, float undef, int memkind);
int mifc_fieldOPERfield(mifc_ctx* ctx, int compute, int nx, int ny, const float* field1, const float* field2, float* fres, int* fdefined, float undef,
                        int memkind);
int mifc_fieldOPERconstant(mifc_ctx* ctx, int compute, int nx, int ny, const float* field, float value, float* fres, int* fdefined, float undef,
                           int memkind);
int mifc_constantOPERfield(mifc_ctx* ctx, int compute, int nx, int ny, float value, const float* field, float* fres, int* fdefined, float undef,
                           int memkind);

/* ---- EXTENSION, NOT part of miutil::fieldcalc ------------------------------------------------
 * BASELINE.json's north_star names "wind speed/direction from u/v"; the reference has the speed
 * (vectorabs) but no direction function (FieldCalculations.cc:1951-1952 mention it in a comment
 * only), so there is no reference result to be identical to.  Defined here as the meteorological
 * direction the wind blows FROM, degrees clockwise from north, in [0, 360):
 *   dd = 270 - atan2(v, u) * 180 / pi   (calm: 0),   undefined where u or v is undefined,
 * with the flag handling of vectorabs (.cc:1819-1841).  Parity is against this definition only. */
int mifc_winddir(mifc_ctx* ctx, int nx, int ny, const float* u, const float* v, float* dd, int* fdefined, float undef, int memkind);

/* ---- reductions over ensemble members ---------------------
 * sumFields .h:284 / .cc:2671; meanValue .h:286 / .cc:2696; stddevValue .h:289 / .cc:2726;
 * extremeValue .h:292 / .cc:2759; probability .h:294 / .cc:2807.
 * The reference takes std::vector<float*> fields, std::vector<ValuesDefined>
 * fDefinedIn and std::vector<float> limits; here `fields` is a HOST array of
 * nfields pointers (each a host or a device field according to memkind),
 * `fdefined_in` a HOST int array of nfields flags, `limits` a HOST array.
 * Members are reduced in index order, like the reference's inner loop. */
int mifc_sumFields(mifc_ctx* ctx, int nx, int ny, const float* const* fields, int nfields, float* fres, int* fdefined, float undef, int memkind);
int mifc_meanValue(mifc_ctx* ctx, int nx, int ny, const float* const* fields, const int* fdefined_in, int nfields, float* fres, int* fdefined_out,
                   float undef, int memkind);
int mifc_stddevValue(mifc_ctx* ctx, int nx, int ny, const float* const* fields, const int* fdefined_in, int nfields, float* fres,
                     int* fdefined_out, float undef, int memkind);
int mifc_extremeValue(mifc_ctx* ctx, int compute, int nx, int ny, const float* const* fields, int nfields, float* fres, int* fdefined, float undef,
                      int memkind);
int mifc_probability(mifc_ctx* ctx, int compute, int nx, int ny, const float* const* fields, const int* fdefined_in, int nfields,
                     const float* limits, int nlimits, float* fres, int* fdefined_out, float undef, int memkind);

/* ---- batched over vertical levels / ensemble members (new surface) ------ */
/* The reference is called once per 2-D field; a caller that wants vorticity
 * AND divergence on nlev levels makes 2*nlev calls.  These
 * entry points do the same work in one fused pass per level-batch.
 *
 * u, v, rvort, diverg : [nlev][ny][nx];  xmapr, ymapr (, fcoriolis) : [ny][nx],
 * shared by all levels.  rvort or diverg may be NULL to skip that output.
 * fdefined : host int[nlev], in = input state per level, out = state of the
 * result(s) of that level -- identical for rvort and diverg because
 * FieldCalculations.cc:1861 and :1927 test the same four neighbours.
 * Result per level is bit-identical to relvort()+divergence() on that level. */
int mifc_vortdiv_levels(mifc_ctx* ctx, int nx, int ny, int nlev, const float* u, const float* v, const float* xmapr, const float* ymapr,
                        float* rvort, float* diverg, int* fdefined, float undef, int memkind);
/* Asynchronous form, device pointers only: enqueues on the context's stream
 * and returns.  `fdefined_in` (host int[nlev], may be NULL = all
 * MIFC_SOME_DEFINED) is read before returning.  `n_undefined_dev` is a device
 * array of nlev 64-bit counters that the call zeroes and the kernel fills;
 * classify each against (nx*ny - 2*nx) with mifc_classify() once the stream
 * has drained.  It may be NULL when every level is MIFC_ALL_DEFINED. */
int mifc_vortdiv_levels_enqueue(mifc_ctx* ctx, int nx, int ny, int nlev, const float* u, const float* v, const float* xmapr, const float* ymapr,
                                float* rvort, float* diverg, const int* fdefined_in, float undef, unsigned long long* n_undefined_dev);

/* The same with explicit level strides (in floats, multiples of 4, >= nx*ny):
 * level l of u / v starts at u + l*in_level_stride, of rvort / diverg at
 * rvort + l*out_level_stride.  A batch whose levels are padded to
 * mifc_batch_level_stride(nx, ny) floats streams through HBM evenly whatever
 * nx*ny is (a level size close to a multiple of 4 MiB makes the
 * eight levels a workgroup walks side by side meet in the same memory channels). */
size_t mifc_batch_level_stride(int nx, int ny);
int mifc_vortdiv_levels_strided_enqueue(mifc_ctx* ctx, int nx, int ny, int nlev, const float* u, const float* v, const float* xmapr,
                                        const float* ymapr, float* rvort, float* diverg, size_t in_level_stride, size_t out_level_stride,
                                        const int* fdefined_in, float undef, unsigned long long* n_undefined_dev);

/* Vorticity, divergence AND the wind speed ff = vectorabs(u, v) (FieldCalculations.cc:1819) of a level batch in one
 * pass: what BASELINE.json config 5 computes per ensemble member from u and v.  As two calls ff reads u and v a second
 * time (12 B per cell); here it is a third output of the fused kernel, computed from the rows that are in LDS anyway
 * (20 B per cell for the three results instead of 16 + 12).  ff : [nlev][ny][nx], every cell (rows 0 and ny-1 included);
 * n_undefined_ff_dev : device u64[nlev], classify against nx*ny (:1839); n_undefined_dev as in
 * mifc_vortdiv_levels_enqueue.  The input flag of a level governs all three results, as it would in the three reference
 * calls.  Per level each result is bit-identical to its reference function.  Launches the three-output kernel does not
 * take (shallow or small batches, nx % 4 != 0, ...) run the pair and a batched vectorabs: same results, two launches. */
int mifc_vortdiv_ff_levels_enqueue(mifc_ctx* ctx, int nx, int ny, int nlev, const float* u, const float* v, const float* xmapr,
                                   const float* ymapr, float* rvort, float* diverg, float* ff, const int* fdefined_in, float undef,
                                   unsigned long long* n_undefined_dev, unsigned long long* n_undefined_ff_dev);

/* Any stencil operator over a batch of levels (generic form of the call above).
 * op selects the reference function:
 *   MIFC_OP_RELVORT .cc:1843, _ABSVORT :1875, _DIVERGENCE :1910, _VORTDIV (both),
 *   MIFC_OP_GRADIENT_X/_Y/_ABS/_LAPLACE = gradient compute 1..4 :1985,
 *   MIFC_OP_GWIND_X :638, _GWIND_Y :674, _GVORT :708, MIFC_OP_IGWIND :1511 (two outputs),
 *   MIFC_OP_JACOBIAN :2424 (f0 = field1, f1 = field2).
 * f0 = u | z | field | mpot, f1 = v (wind operators and the Jacobian); fcoriolis where the
 * operator takes it, else NULL; out1 only for _VORTDIV / _IGWIND.  f0, f1, out0,
 * out1 : [nlev][ny][nx]; xmapr, ymapr, fcoriolis : [ny][nx] shared.
 * fdefined : host int[nlev] in/out, per level exactly as the single-field call. */
enum {
  MIFC_OP_RELVORT = 0, MIFC_OP_ABSVORT = 1, MIFC_OP_DIVERGENCE = 2, MIFC_OP_VORTDIV = 3,
  MIFC_OP_GRADIENT_X = 4, MIFC_OP_GRADIENT_Y = 5, MIFC_OP_GRADIENT_ABS = 6, MIFC_OP_GRADIENT_LAPLACE = 7,
  MIFC_OP_GWIND_X = 8, MIFC_OP_GWIND_Y = 9, MIFC_OP_GVORT = 10, MIFC_OP_IGWIND = 11,
  MIFC_OP_JACOBIAN = 13,
  /* mifc_stencil_levels_ex only: */
  MIFC_OP_ADVECTION = 12, MIFC_OP_TFP = 14, MIFC_OP_QVECTOR = 15, MIFC_OP_SHAPIRO2 = 17
};
int mifc_stencil_levels(mifc_ctx* ctx, int op, int nx, int ny, int nlev, const float* f0, const float* f1, const float* xmapr, const float* ymapr,
                        const float* fcoriolis, float* out0, float* out1, int* fdefined, float undef, int memkind);

/* Asynchronous form of mifc_stencil_levels, device pointers only (the generic form of
 * mifc_vortdiv_levels_enqueue): enqueues on the context's stream and returns; nothing is read back.
 * fdefined_in : host int[nlev] or NULL (= every level MIFC_SOME_DEFINED), read before returning.
 * n_undefined_dev : device array of nlev 64-bit counters, zeroed by the call and filled by the kernel
 * (NULL allowed when every level is MIFC_ALL_DEFINED).  Once the stream has drained the flag of level l
 * is mifc_classify(n_undefined[l], mifc_stencil_count_domain(op, nx, ny)) -- except MIFC_OP_GWIND_X,
 * which is MIFC_NONE_DEFINED whatever the count (FieldCalculations.cc:664).  This is the entry a caller
 * captures into a HIP graph or issues back to back without a host round trip per operator. */
int mifc_stencil_levels_enqueue(mifc_ctx* ctx, int op, int nx, int ny, int nlev, const float* f0, const float* f1, const float* xmapr,
                                const float* ymapr, const float* fcoriolis, float* out0, float* out1, const int* fdefined_in, float undef,
                                unsigned long long* n_undefined_dev);
/* what a level's undefined count of `op` is classified against: nx*ny - 2*nx (FieldCalculations.cc:1868 and
 * friends, also gradient compute 1, :2068), nx*ny for MIFC_OP_IGWIND (:1543) */
unsigned long long mifc_stencil_count_domain(int op, int nx, int ny);

/* Diagnostic: the kernel form the calling thread's last stencil launch took -- "wind_split", "wind_split_ragged",
 * "wind_split_ff", "wind_levelwalk", "wind_rows", "wind_oneshot", "wind_oneshot_tiles", "scalar_split",
 * "scalar_split_ragged", "scalar_levelwalk", "scalar_rows", "scalar_oneshot", "advection_split", "advection_split_ragged", "advection_oneshot", "flat4", "cell"; "" before the first launch
 * and for the operators with launchers of their own.  The tests use it to check that a case reaches the kernel it is
 * meant for; nothing on the data path reads it.  The string is static. */
const char* mifc_last_stencil_form(void);

/* The same plus the rest of the stencil family over a batch of levels, the map
 * and Coriolis fields shared by all levels:
 *   MIFC_OP_ADVECTION .cc:1942  f0 = f, f1 = u, f2 = v, scalar = hours
 *   MIFC_OP_TFP       .cc:2266  f0 = tx                         (thermalFrontParameter)
 *   MIFC_OP_QVECTOR   .cc:505   f0 = z, f1 = t, level_scalars = host float[nlev] pressures p, compute 1..4
 *   MIFC_OP_SHAPIRO2  .cc:2076  f0 = field (out0 == f0 allowed; no map fields)
 * Every other op is forwarded to mifc_stencil_levels.  Per level the result and the flag are those
 * of the single-field reference call; the two-stage operators run as ONE launch per flag group
 * (levels whose input is ALL_DEFINED / the others) with their intermediate fields in LDS. */
int mifc_stencil_levels_ex(mifc_ctx* ctx, int op, int nx, int ny, int nlev, const float* f0, const float* f1, const float* f2, const float* xmapr,
                           const float* ymapr, const float* fcoriolis, const float* level_scalars, float scalar, int compute, float* out0,
                           float* out1, int* fdefined, float undef, int memkind);

/* Fused derived variables on hybrid model levels: one pass producing any of
 *   ff    = vectorabs(u, v)                                   (.cc:1819)
 *   rh    = hlevelhum(t, q, ps, a, b, "", compute=1)  RH in %   (.cc:1145)
 *   theta = hleveltemp(t, ps, a, b, "", compute=3)  T -> theta  (.cc:1046)
 * u, v, t, q, ff, rh, theta : [nlev][ny][nx]; ps : [ny][nx] shared;
 * alevel, blevel : host float[nlev].  Any of ff / rh / theta may be NULL
 * (its inputs are then not read).  fdef_wind / fdef_thermo : host int[nlev]
 * input states of (u,v) and of (t,q,ps); outputs fdef_ff, fdef_rh, fdef_theta
 * : host int[nlev].  Returns 0 if any level has a bad (a,b) pair (.cc:298). */
int mifc_hlevel_derived_levels(mifc_ctx* ctx, int nx, int ny, int nlev, const float* u, const float* v, const float* t, const float* q,
                               const float* ps, const float* alevel, const float* blevel, float* ff, float* rh, float* theta,
                               const int* fdef_wind, const int* fdef_thermo, int* fdef_ff, int* fdef_rh, int* fdef_theta, float undef,
                               int memkind);
/* Asynchronous, device pointers only.  n_undefined_dev: device u64[3*nlev]
 * laid out [ff | rh | theta], zeroed by the call; classify against nx*ny. */
int mifc_hlevel_derived_levels_enqueue(mifc_ctx* ctx, int nx, int ny, int nlev, const float* u, const float* v, const float* t, const float* q,
                                       const float* ps, const float* alevel, const float* blevel, float* ff, float* rh, float* theta,
                                       const int* fdef_wind, const int* fdef_thermo, float undef, unsigned long long* n_undefined_dev);

/* The general form: per level l, any subset of
 *   ff   = vectorabs(u, v)                                               (.cc:1819)
 *   temp = hleveltemp(t, ps, a_l, b_l, temp_unit, temp_compute)          (.cc:1046; compute 1..5)
 *   hum  = hlevelhum(t, h, ps, a_l, b_l, hum_unit, hum_compute)          (.cc:1145; compute 1..12)
 *   hum2 = hlevelhum(t, h, ps, a_l, b_l, hum2_unit, hum2_compute)        a second variant of the SAME inputs,
 *          e.g. hum = RH (compute 1) and hum2 = dew point (compute 9) from T and q
 *   dd   = mifc_winddir(u, v)   EXTENSION, not a reference function (see above): wind direction
 * in ONE pass over u, v, t, h ([nlev][ny][nx]; h = specific humidity or RH as the variants demand; ps [ny][nx]
 * shared).  NULL outputs are skipped (their inputs are not read); unit / compute arguments of a skipped
 * output are ignored.  Results and flags per level are those of the per-level reference calls.  Returns 0
 * where one of those calls would return false (bad (a, b) pair .cc:298, compute out of range) and for
 * temp_compute outside 1..5 (the reference leaves such cells unwritten, .cc:1080-1090).
 * fdef_wind / fdef_thermo: host int[nlev] input states of (u, v) and of (t, h, ps); fdef_ff, fdef_temp,
 * fdef_hum, fdef_hum2, fdef_dd: host int[nlev] results (may be NULL). */
int mifc_hlevel_derived_batch(mifc_ctx* ctx, int nx, int ny, int nlev, const float* u, const float* v, const float* t, const float* h,
                              const float* ps, const float* alevel, const float* blevel, float* ff, float* temp, const char* temp_unit,
                              int temp_compute, float* hum, const char* hum_unit, int hum_compute, float* hum2, const char* hum2_unit,
                              int hum2_compute, float* dd, const int* fdef_wind, const int* fdef_thermo, int* fdef_ff, int* fdef_temp,
                              int* fdef_hum, int* fdef_hum2, int* fdef_dd, float undef, int memkind);
/* Asynchronous, device pointers only.  n_undefined_dev: device u64[5*nlev] laid out
 * [ff | temp | hum | hum2 | dd], zeroed by the call; classify against nx*ny. */
int mifc_hlevel_derived_batch_enqueue(mifc_ctx* ctx, int nx, int ny, int nlev, const float* u, const float* v, const float* t, const float* h,
                                      const float* ps, const float* alevel, const float* blevel, float* ff, float* temp, const char* temp_unit,
                                      int temp_compute, float* hum, const char* hum_unit, int hum_compute, float* hum2, const char* hum2_unit,
                                      int hum2_compute, float* dd, const int* fdef_wind, const int* fdef_thermo, float undef,
                                      unsigned long long* n_undefined_dev);

/* ---- horizontally decomposed single field (row slabs) -------------------- */
/* Vorticity + divergence on one row slab of a larger field (config 4 of
 * BASELINE.json: a 4000x4000 field split over 8 GPUs along y).
 * The slab holds `ny_local` owned rows starting at global row `j0` of a field
 * with `ny_global` rows; u and v point at [ny_local + 2][nx] buffers whose
 * first and last rows are the halo rows (global rows j0-1 and j0+ny_local)
 * filled by the caller's exchange (RCCL send/recv over xGMI; halo content is
 * ignored where it would fall outside the global field).  xmapr, ymapr,
 * rvort, diverg : [ny_local][nx], owned rows only.  The global edge rules of
 * fillEdges (FieldCalculations.cc:59-74) are applied by whichever slab owns
 * the edge rows.  *n_undefined_dev (device u64[1], zeroed by the call)
 * receives this slab's share of the global undefined count; the sum over
 * slabs classifies against nx*ny_global - 2*nx.  Asynchronous, device only. */
int mifc_vortdiv_slab_enqueue(mifc_ctx* ctx, int nx, int ny_global, int j0, int ny_local, const float* u_halo, const float* v_halo,
                              const float* xmapr, const float* ymapr, float* rvort, float* diverg, int fdefined_in, float undef,
                              unsigned long long* n_undefined_dev);

/* The same restricted to the owned rows [row_begin, row_end) of the slab, so that a caller
 * can overlap the halo exchange with compute: enqueue the exchange, launch
 * the interior rows [2, ny_local-2) -- they read no halo row --, wait for the exchange on
 * the stream, launch [0, 2) and [ny_local-2, ny_local).  accumulate_count != 0 leaves
 * *n_undefined_dev as it is and adds to it (zero it with the first launch of a slab only).
 * A range must not separate row 0 from row 1 or row ny_global-1 from row ny_global-2 of
 * the whole field (fillEdges copies one from the other): such a call returns 0. */
int mifc_vortdiv_slab_rows_enqueue(mifc_ctx* ctx, int nx, int ny_global, int j0, int ny_local, int row_begin, int row_end, const float* u_halo,
                                   const float* v_halo, const float* xmapr, const float* ymapr, float* rvort, float* diverg, int fdefined_in,
                                   float undef, unsigned long long* n_undefined_dev, int accumulate_count);
/* ---- launch-bound work as one launch: HIP-graph capture of a sequence of calls ----------------- */
/* The reference is called once per 2-D field; a caller that keeps its loop over levels pays a launch per level for
 * microseconds of traffic (one 1440x720 level of the fused derived kernel: 33 MB = 4 us at peak).  Between
 * mifc_graph_begin and mifc_graph_end the context's ASYNCHRONOUS entry points (mifc_*_enqueue, mifc_slab_plan_begin /
 * _finish) do not run: their launches are recorded with the arguments given; mifc_graph_launch replays the recorded
 * sequence on the context's stream with one runtime call, any number of times (the buffers are read when the graph
 * runs, not when it was recorded).  The synchronous entry points (everything that returns a flag) cannot be recorded:
 * the capture is invalidated and mifc_graph_end returns NULL.  max_levels_per_call: the deepest level batch of the
 * recorded calls (the per-level scratch is sized before the capture starts; 0 = 256).  One capture per context at a
 * time; mifc_set_stream must not be called in between.
 * Lanes: calls that do not depend on each other -- the levels of a caller's loop -- may be recorded side by side
 * (mifc_graph_begin_lanes + mifc_graph_lane(k) before a call): every lane is a chain of its own in the graph, all lanes
 * start together and the graph ends when all have ended, so the launch gaps of one lane are filled by the others.  The
 * caller vouches that calls in different lanes touch different outputs.  Calls that need the context's per-level
 * scratch (stencil batches with mixed flags, derived batches of more than 8 levels) are refused in a multi-lane capture. */
typedef struct mifc_graph mifc_graph;
int mifc_graph_begin(mifc_ctx* ctx, int max_levels_per_call);
int mifc_graph_begin_lanes(mifc_ctx* ctx, int max_levels_per_call, int n_lanes /* 1 .. 16 */);
int mifc_graph_lane(mifc_ctx* ctx, int lane);
mifc_graph* mifc_graph_end(mifc_ctx* ctx);
int mifc_graph_launch(mifc_graph* graph);
void mifc_graph_destroy(mifc_graph* graph);

/* ---- the decomposed step as ONE call (BASELINE.json config 4) --------------- */
/* Communicator: RCCL over xGMI, one process per GPU.  Either the library creates it -- rank 0 calls
 * mifc_comm_unique_id(), the caller distributes the MIFC_COMM_ID_BYTES bytes to every rank by any means it
 * has (MPI, a file, torch.distributed), every rank calls mifc_comm_init() -- or the caller hands over an
 * ncclComm_t it already owns (mifc_comm_adopt: rank and size are read from it; e.g. PyTorch's
 * ProcessGroupNCCL._comm_ptr()).  RCCL is loaded on first use; a process that never calls these never loads it.
 * Slab k of a field lives on rank k: the neighbours of a slab are rank - 1 (above) and rank + 1 (below). */
#define MIFC_COMM_ID_BYTES 128
int mifc_comm_unique_id(char* id_out /* [MIFC_COMM_ID_BYTES] */);
int mifc_comm_init(mifc_ctx* ctx, const char* id /* [MIFC_COMM_ID_BYTES] */, int rank, int world);
int mifc_comm_adopt(mifc_ctx* ctx, void* nccl_comm);
int mifc_comm_release(mifc_ctx* ctx);
/* returns 1 when the context has a communicator; rank / world may be NULL */
int mifc_comm_info(const mifc_ctx* ctx, int* rank, int* world);

/* A plan binds the buffers of one rank's row slab of a level batch:
 *   u_halo, v_halo : [nlev][ny_local + 2][nx], rows 1 .. ny_local of a level owned, rows 0 and ny_local + 1 its
 *                    halo rows (filled by the step from the neighbours; never read at the edges of the whole field)
 *   xmapr, ymapr   : [ny_local][nx] (owned rows, shared by the levels);  rvort, diverg : [nlev][ny_local][nx]
 *   fdefined_in    : input state of every level (MIFC_ALL_DEFINED: no tests, no counters)
 *   n_undefined_dev: device u64[nlev] (NULL allowed for MIFC_ALL_DEFINED)
 * mifc_slab_plan_step() enqueues ONE decomposed step on the context's stream and returns: one row of u and of v per
 * level to / from each neighbour (grouped ncclSend / ncclRecv on a stream of the plan's own), the owned rows that read
 * no halo row meanwhile, then the two boundary strips, then -- tested input, more than one rank -- an ncclAllReduce
 * that leaves the WHOLE field's undefined counts of every level in n_undefined_dev on every rank (classify against
 * nx*ny_global - 2*nx; with one rank or MIFC_ALL_DEFINED nothing is reduced).  Results per level are bit-identical
 * to relvort() + divergence() on the whole field (FieldCalculations.cc:1843-1940), fillEdges included.  The
 * sequence is captured into a HIP graph on the first step and replayed afterwards (MIFC_SLAB_GRAPH=0, or a capture
 * the runtime refuses: enqueued call by call); the buffers, the communicator and the tuning environment are those
 * of the first step.  What the caller wrote to the owned rows of u_halo / v_halo on the context's stream before
 * the call is what is sent. */
typedef struct mifc_slab_plan mifc_slab_plan;
mifc_slab_plan* mifc_slab_plan_create(mifc_ctx* ctx, int nx, int ny_global, int j0, int ny_local, int nlev, float* u_halo, float* v_halo,
                                      const float* xmapr, const float* ymapr, float* rvort, float* diverg, int fdefined_in, float undef,
                                      unsigned long long* n_undefined_dev);
void mifc_slab_plan_destroy(mifc_slab_plan* plan);
int mifc_slab_plan_step(mifc_slab_plan* plan);
/* 1 when the steps replay a captured graph */
int mifc_slab_plan_uses_graph(const mifc_slab_plan* plan);
/* The same step for a caller with a transport of its own (MPI, a host relay, hipMemcpyPeer between the
 * contexts of one process): begin = counters zeroed + the rows that read no halo row; the caller then fills the
 * halo rows, ordered on the context's stream; finish = the boundary strips (the whole slab when it is too thin to
 * split).  Counts stay local: the caller sums them over the slabs. */
int mifc_slab_plan_begin(mifc_slab_plan* plan);
int mifc_slab_plan_finish(mifc_slab_plan* plan);

/* Halo transport for a process that drives several GPUs itself (one context per GPU):
 * copies n_floats from src_dev (memory of src_ctx's device) to dst_dev (dst_ctx's device)
 * over xGMI (hipMemcpyPeerAsync), enqueued on dst_ctx's stream and ordered after the work
 * already queued on src_ctx's stream.  One row of a slab is nx floats.  Processes that own
 * one GPU each exchange the rows with RCCL send/recv instead (mi-fieldcalc_amd/sharding.py,
 * tools/bench_multigpu.py); the slab entry points do not care how the halo rows got there. */
int mifc_halo_copy_enqueue(mifc_ctx* dst_ctx, float* dst_dev, mifc_ctx* src_ctx, const float* src_dev, size_t n_floats);

#ifdef __cplusplus
}
#endif

#endif /* MIFC_H */
