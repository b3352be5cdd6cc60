// mifc_kernels.h -- host-callable launchers of the gfx950 kernels.
// Everything here takes DEVICE pointers and a stream and only enqueues work.
#ifndef MIFC_KERNELS_H
#define MIFC_KERNELS_H

#include <hip/hip_runtime.h>

#include "mifc_env.h"

namespace mifc {

typedef unsigned long long u64;

// ---------------------------------------------------------------- elementwise
enum EwiseOp {
  EW_VECTORABS = 0, // FieldCalculations.cc:1819
  EW_TEMP = 1,      // pleveltemp :328, hleveltemp :1046, aleveltemp :1310
  EW_HUM = 2,       // plevelhum :400, hlevelhum :1145, alevelhum :1394
  EW_CVHUM_TD = 3,  // cvhum compute 1..3 :1759-1785
  EW_CVHUM_RH = 4,  // cvhum compute 4,5 :1787-1811
  EW_MOMENTUM_X = 5, // momentumXcoordinate :2351: in0 = v, in1 = xmapr, in2 = fcoriolis
  EW_MOMENTUM_Y = 6, // momentumYcoordinate :2387: in0 = u, in1 = ymapr, in2 = fcoriolis
  // kernel-internal: EW_TEMP with a scalar pressure and compute 1..3 (pleveltemp's plain conversions) as a
  // small kernel of its own; launch_ewise() selects it, callers keep passing EW_TEMP
  EW_TEMP_SCALAR = 7,
  // likewise: EW_TEMP with a hybrid / field pressure and compute 1..3 (no saturation table in the kernel),
  // EW_HUM for the two kinds that need no table inverse (q <-> RH)
  EW_TEMP_PLAIN = 8,
  EW_HUM_DIRECT = 9
};
enum PressureSource { PS_SCALAR = 0, PS_HYBRID = 1, PS_FIELD = 2 };
enum HumKind { HUM_Q_RH = 0, HUM_RH_Q = 1, HUM_Q_TD = 2, HUM_RH_TD = 3 };
// which extra undefined test the pressure field gets (hlevelhum :1187, alevelhum :1429)
enum PTest { PT_NONE = 0, PT_FULL = 1 /* is_defined(p) */, PT_NEQ = 2 /* p != undef only */ };

struct EwiseParams
{
  int op;
  int n;          // cells per field
  int all_defined; // input flag == ALL_DEFINED: no per-cell tests (FieldCalculations.h:47-50)
  int count;      // 0: operator does not count / touch the flag (pleveltemp 1..3)
  int psrc;       // PressureSource
  int ptest;      // PTest
  int compute;    // remapped compute (temp: 1..5, others see kernel)
  int kind;       // HumKind
  int from_theta; // humidity: t is potential temperature
  float p;        // PS_SCALAR: pressure
  float pidcp;    // PS_SCALAR: powf(p*p0inv,kappa) evaluated on the host like the reference (:347)
  float pi;       // PS_SCALAR: host-side pi (pidcp*cp for temp, cp*pidcp for hum)
  float tconv;    // plevelhum :436 / cvhum :1753
  float tdconv;   // :437, :1181, :1423, :1754
  float alevel, blevel;
  float unit_scale; // cvhum :1746-1750
  int nx;           // momentum coordinates: row length (cell index -> column / row)
  int cell0;        // index of in0[0] inside the field (non-zero only for the scalar tail launch)
  float fcormin;    // momentum coordinates: |fcoriolisMin| (:2366)
  float undef;
  const float* in0; // u | t
  const float* in1; // v | hum
  const float* in2; // ps | p field
  float* out;
  u64* n_undefined; // device counter (may be null when count == 0)
  // big launches: the workgroups leave their counts in partials[blockIdx.x] (plain stores) and a one-workgroup kernel
  // behind the launch adds them up -- no queue of same-address atomics (mifc_device.h: undefined-cell counting)
  unsigned int* partials;
  int partials_cap;
};

hipError_t launch_ewise(const EwiseParams& prm, hipStream_t stream);

// recommended distance (in floats) between the levels of a device-resident batch of fields of n cells
size_t padded_level_stride(size_t n);

// Fused derived variables over hybrid levels (mifc_derived.hip), BASELINE.json config 2 x nlev.
struct DerivedParams
{
  int n;    // cells per level
  int nlev;
  const float *u, *v, *t, *h, *ps; // h: humidity input (q or RH); ps shared by all levels
  const float *alevel, *blevel;    // device float[nlev]
  float *ff, *temp, *hum, *td;     // outputs, any may be null: vectorabs | hleveltemp | hlevelhum | a second hlevelhum variant
  float* dd;                       // EXTENSION output (may be null): wind direction from u, v, see wind_direction() in mifc_device.h
  int temp_compute;                // hleveltemp compute after the unit remap, 1..5
  int hum_code, td_code;           // 1 + HumKind + 4 * from_theta (0 = none)
  float hum_tdconv, td_tdconv;     // :1181
  const unsigned char* wind_all_defined;   // device u8[nlev]
  const unsigned char* thermo_all_defined; // device u8[nlev]
  int every_level_all_defined;             // host hint: skip all tests
  float undef;
  u64 *cnt_ff, *cnt_temp, *cnt_hum, *cnt_td, *cnt_dd; // device u64[nlev] each (level 0 of this launch first)
  // small batches (nlev <= 8, e.g. the single level of BASELINE.json config 2)
  // carry the per-level scalars in the kernel arguments: no upload before the launch
  int n_inline; // != 0: use the arrays below instead of the device arrays above
  float a_inline[8], b_inline[8];
  unsigned char wind_inline[8], thermo_inline[8];
};
hipError_t launch_derived_levels(const DerivedParams& prm, hipStream_t stream);

// ------------------------------------------- the rest of the pointwise catalogue
// (FieldCalculations.cc line of each operator in the comment)
enum PwOp {
  PW_PLEVELTHE = 0,   // plevelthe :369: in = t, rh
  PW_XLEVELTHE,       // hlevelthe :1100 (hybrid) / alevelthe :1355: in = t, q, ps|p
  PW_PDUCT,           // plevelducting :597: in = t, h
  PW_XDUCT,           // hlevelducting :1219 (hybrid) / alevelducting :1460: in = t, h, ps|p
  PW_HPRESSURE,       // hlevelpressure :1276
  PW_DZ2TMEAN,        // pleveldz2tmean :466
  PW_KINDEX,          // :745
  PW_DUCTINDEX,       // :816
  PW_SHOWALTER,       // :872
  PW_BOYDEN,          // :973
  PW_SWEAT,           // :1016
  PW_SOUNDSPEED,      // seaSoundSpeed :1555
  PW_ADDCONST,        // cvtemp :1608 (the conversion pass)
  PW_ABSHUM,          // :1676
  PW_WINDCOOLING,     // :2181
  PW_UNDERCOOLED,     // underCooledRain :2231
  PW_FLIGHTLEVEL,     // pressure2FlightLevel :2311
  PW_SNOWCM,          // snow_in_cm :3063
  PW_CLASSES,         // values2classes :2462
  PW_MINMAX_FIELDS,   // minvalueFields :2501 (compute 1), maxvalueFields :2516 (2)
  PW_MINMAX_CONST,    // minvalueFieldConst :2507 (1), maxvalueFieldConst :2522 (2)
  PW_MATH,            // abs 1, log10 2, pow10 3, log 4, exp 5, power 6 (:2531-2563)
  PW_REPLACE,         // replaceUndefined :2565 (1), replaceDefined :2587 (2), copy (3)
  PW_FILL,            // fillUndef :76 and the std::fill branches of replace*
  PW_FIELD_OP_FIELD,  // fieldOPERfield :2611
  PW_FIELD_OP_CONST,  // fieldOPERconstant :2627
  PW_CONST_OP_FIELD,  // constantOPERfield :2647
  // FieldCalculationsVesselIcing.cc: vesselIcingOverland :77 (compute 1), vesselIcingMertins :114 (2);
  // in = airtemp, seatemp, u, v, sal, aice
  PW_VESSEL_ICING,
  // EXTENSION, no reference function (BASELINE.json names "wind direction from u/v"):
  // meteorological direction the wind blows FROM, degrees clockwise from north; in = u, v
  PW_WINDDIR
};

struct PwParams
{
  int op;
  int n;
  int all_defined;          // input flag == ALL_DEFINED
  int count;                // operator classifies its output (the "Undef" helper templates :142-179 and the hand-written loops)
  int compute;
  int hybrid;               // p = s[0] + s[1] * ps (h-level) instead of the p field (a-level)
  int may_keep;             // some cells may stay unwritten: the kernel starts from the output's content
  int no_input_test;        // replace*: no is_defined test on the input
  int skip_undefined_input; // showalterIndex: an undefined input is counted but the cell is not written
  float undef;
  float s[8];  // operator scalars, evaluated on the host like the reference does
  double d[2];
  const float* in[8];
  float* out;
  u64* n_undefined;
  const float* values; // values2classes: device copy of the class limits
  int nvalues;
  unsigned int* partials; // see EwiseParams
  int partials_cap;
};
hipError_t launch_pointwise(const PwParams& prm, hipStream_t stream);
// sum and number of the defined cells (cvtemp compute 3, 4)
hipError_t launch_count_partials(const unsigned int* partials, int n, u64* counter, hipStream_t stream);
// partials[level][unit]: one workgroup per level adds its `per_level` entries to counters[level]
hipError_t launch_count_partials_levels(const unsigned int* partials, int per_level, int nlev, u64* counters, hipStream_t stream);
hipError_t launch_mean_defined(const float* f, int n, int all_defined, float undef, double* sum, unsigned long long* count, hipStream_t stream);

// ------------------------------------ reductions over ensemble members
enum EnsembleOp {
  ENS_SUM = 0,        // sumFields :2671
  ENS_MEAN = 1,       // meanValue :2696
  ENS_STDDEV = 2,     // stddevValue :2726
  ENS_EXTREME = 3,    // extremeValue :2759 (compute 1 max, 2 min, 3 index of max, 4 index of min)
  ENS_PROBABILITY = 4 // probability :2807 (compute 1..3 percent, 4..6 count)
};

struct EnsembleParams
{
  int op;
  int n;       // cells per member field
  int first;   // scalar tail launch: first cell
  int nfields; // members
  int compute;
  int all_defined;                    // sumFields / extremeValue: the one input flag
  const unsigned char* member_flags;  // device u8[nfields] (ValuesDefined of each member) or null
  int check_above, check_below;       // probability :2821-2823
  float value_above, value_below;     // :2824-2825
  int vector_ok;                      // every member field and the output are 16-byte aligned
  float undef;
  const float* const* fields; // device table of nfields device pointers
  float* out;
  u64* n_undefined;
  // up to 64 members (an ensemble: 51) travel in the kernel arguments: no table upload before the launch
  int n_inline;         // != 0: use the arrays below instead of `fields` / `member_flags`
  int has_member_flags; // with n_inline: flags_inline is meaningful
  const float* fields_inline[64];
  unsigned char flags_inline[64];
};
hipError_t launch_ensemble(const EnsembleParams& prm, hipStream_t stream);

// -------------------------------------------------------------------- stencils
enum StencilOp {
  ST_RELVORT = 0,    // :1843
  ST_ABSVORT = 1,    // :1875
  ST_DIVERGENCE = 2, // :1910
  ST_VORTDIV = 3,    // relvort + divergence fused, two outputs
  ST_GRAD_X = 4,     // gradient compute 1 :2013
  ST_GRAD_Y = 5,     // 2 :2025
  ST_GRAD_ABS = 6,   // 3 :2037
  ST_GRAD_LAP = 7,   // 4 :2051
  ST_GWIND_X = 8,    // plevelgwind_xcomp :638
  ST_GWIND_Y = 9,    // plevelgwind_ycomp :674
  ST_GVORT = 10,     // plevelgvort :708
  ST_IGWIND = 11,    // ilevelgwind :1511, two outputs
  // SURVEY.md 8f-1 (one-lane-per-cell kernel only, so far)
  ST_ADVECTION = 12, // advection :1942: f0 = f, f1 = u, f2 = v, scale = -3600*hours
  ST_JACOBIAN = 13,  // jacobian :2424: f0 = field1, f1 = field2
  ST_TFP = 14,       // second pass of thermalFrontParameter :2289-2302: f0 = tx, f1 = |grad tx|
  // last pass of plevelqvector :564-593: f0 = ug, f1 = vg, f2 = t, scale = tscale, scale2 = c
  ST_QVEC_X = 15, // compute 1, 2
  ST_QVEC_Y = 16  // compute 3, 4
};

struct StencilParams
{
  int op;
  int nx;
  int ny_global; // rows of the whole field
  int j0;        // global row of the first owned row (0 unless row-slab decomposed)
  int ny_local;  // owned rows in this buffer
  int nlev;      // fields in the batch (maps are shared)
  // f0/f1 point at OWNED row 0; for a slab the halo rows sit directly before and after
  const float* f0; // u | z | field | mpot
  const float* f1; // v (uv family only)
  const float* f2; // third input (advection: v)
  float scale;     // advection: -3600 * hours, rounded to float like the reference (:1963); Q-vector: tscale
  float scale2;    // Q-vector: c = -r / (p * 100) (:564)
  const float* scale_lev;  // Q-vector over a level batch: tscale / c per level of the launch (device), or null: scale / scale2
  const float* scale2_lev;
  const float* xmapr;
  const float* ymapr;
  const float* fcoriolis;
  float* out0;
  float* out1;   // second output (ST_VORTDIV diverg, ST_IGWIND vg); may be null
  // ST_VORTDIV over whole fields only, optional: the wind speed sqrt(u*u + v*v) of every cell as a third output and its
  // per-level undefined counts (vectorabs :1819, count domain nx*ny).  launch_vortdiv_rows leaves *handled false when the
  // launch is not one the split-role kernel takes; launch_stencil then returns hipErrorNotSupported
  float* out_ff;
  u64* n_undefined_ff;
  long in_level_stride;  // elements between consecutive levels of f0/f1
  long out_level_stride; // same for out0/out1
  const unsigned char* all_defined; // device u8[nlev] or null
  int every_level_all_defined;
  float undef;
  u64* n_undefined; // device u64[nlev]
  // row slabs of the wind operators only: restrict the launch to the owned output rows
  // [row_begin, row_end) (0, 0 = all).  Lets a caller compute the rows that need no halo while
  // the halo exchange is in flight.  A range must not separate a global edge row from the row
  // it is filled from (rows 0 / 1 and ny-2 / ny-1 of the whole field stay together).
  int row_begin, row_end;
  // Big levels (>= 2 048 workgroups per level), tested: the workgroups leave their counts in partials[level][unit] (plain
  // stores) and launch_count_partials_levels adds them up behind the kernel -- thousands of atomics on ONE counter address
  // (or on neighbouring counters: one cache line) take ~5 ns each, one after the other (4000 x 4000: 7 800 workgroups, 44 us
  // on top of a 65-us kernel; 8 levels on the level-walking kernel: 196 us on top of 390).  nullptr / 0: atomics.
  unsigned int* partials = nullptr;
  int partials_cap = 0;
};

hipError_t launch_stencil(const StencilParams& prm, hipStream_t stream);

// What a level-batch launch needs in place beforehand, as ONE small kernel instead of a host-to-device copy plus a fill
// (two stream operations on two engines: ~10 us of a 0.2 ms launch): the per-level input flags travel bit-packed in the
// kernel arguments and are expanded into d_flags (u8[nlev], null: none needed), d_counts[0 .. n_counts) is zeroed
// (null: nothing to zero).  Returns hipErrorInvalidValue beyond kPrepMaxLevels levels (the caller then copies and fills).
constexpr int kPrepMaxLevels = 2048;
// Which kernel form the calling thread's last stencil launch took ("wind_split", "scalar_oneshot", "flat4" ...): a
// diagnostic for tests and bug reports (mifc_last_stencil_form), nothing reads it on the data path.
void note_form(const char* form);
const char* last_form();
hipError_t launch_prep_levels(const unsigned char* host_flags, int nlev, unsigned char* d_flags, u64* d_counts, int n_counts, hipStream_t stream);

// second-order Shapiro filter, FieldCalculations.cc:2076 (mifc_shapiro.hip)
struct ShapiroParams
{
  int nx, ny;
  int all_defined;       // input flag == ALL_DEFINED
  float undef;
  float* f1;             // in: the field, out: the smoothed field
  float* f2;             // scratch, nx*ny floats
  unsigned char* mask_x; // scratch, nx*ny bytes each (unused when all_defined)
  unsigned char* mask_y;
};
hipError_t launch_shapiro2(const ShapiroParams& prm, hipStream_t stream);
hipError_t launch_shapiro2_levels(const ShapiroParams& prm, int n_launch_levels, const int* levels, hipStream_t stream); // batches, see mifc_shapiro.hip
// The four sweeps in one launch, src -> dst (two different arrays); nx % 4 == 0, 16-byte aligned.
bool shapiro2_fused_supported(int nx, int ny, const float* src, const float* dst);
hipError_t launch_shapiro2_fused(int nx, int ny, int all_defined, float undef, const float* src, float* dst, hipStream_t stream);
// the same over n_launch_levels fields level_stride floats apart; entry k of the launch is level levels[k] (k if null)
hipError_t launch_shapiro2_fused_levels(int nx, int ny, int all_defined, float undef, const float* src, float* dst, int n_launch_levels,
                                        long level_stride, const int* levels, hipStream_t stream);

// Stencil-of-a-stencil operators in one launch (mifc_fused2.hip): the
// intermediate field(s) of thermalFrontParameter (:2266) and plevelqvector
// (:505) live in LDS row rings instead of going through HBM.
enum Fused2Op
{
  F2_TFP = 0,    // a = tx
  F2_QVEC_X = 1, // a = z, t = temperature; compute 1, 2
  F2_QVEC_Y = 2  // compute 3, 4
};
struct Fused2Params
{
  int op;
  int nx, ny;
  int check; // input flag != ALL_DEFINED
  const float* a;
  const float* t;
  const float* xmapr;
  const float* ymapr;
  const float* fcoriolis; // Q-vector only
  float* out;
  float undef;
  float scale;  // Q-vector: tscale
  float scale2; // Q-vector: c (:564)
  // [0] cells the first pass left undefined (TFP, tested input only)
  // [1] cells the last pass left undefined (what the returned flag is made from)
  // [2] TFP, tested input only: cells of the last pass that only the defined-test rejected
  u64* counts;
  // level batches (0 / null: one field): grid.y walks n_launch_levels entries; entry k is level
  // levels[k] (or k when levels is null) -- the levels of a batch are launched in two groups, the ones
  // whose input flag is ALL_DEFINED (no tests) and the others.  a, t, out are level_stride floats
  // apart, the map / Coriolis fields are shared, counts holds three counters per level, and the
  // Q-vector's scalars (they depend on the level's pressure) come from scale_lev / scale2_lev.
  int n_launch_levels;
  long level_stride;
  const int* levels;
  const float* scale_lev;
  const float* scale2_lev;
};
bool fused2_supported(const Fused2Params& prm);
hipError_t launch_fused2(const Fused2Params& prm, hipStream_t stream);

#ifdef MIFC_MEASUREMENT_BUILD // libmifc_measure.so only
// diagnostic: (0.5*a*b*g0)/g through the shared-reciprocal quotient of mifc_device.h and through a plain f64 division
hipError_t launch_division_check(const float* a, const float* b, const float* g, float* shared, float* plain, size_t n, hipStream_t stream);

// diagnostic: two-in / two-out streaming copy (bandwidth yardstick)
hipError_t launch_stream2(int variant, int blocks, float* d0, float* d1, const float* s0, const float* s1, size_t n_floats, hipStream_t stream);
#endif

} // namespace mifc

#endif // MIFC_KERNELS_H
