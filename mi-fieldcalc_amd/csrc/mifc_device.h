// mifc_device.h -- device-side building blocks shared by the gfx950 kernels.
//
// Arithmetic contract: every reference
// expression that contains a bare double literal (0.5, 2., 100., 0.01 ...) is
// evaluated in double and rounded to float once, on the store; expressions of
// float variables only stay float.  The helpers below spell the promotion out
// operation by operation; the library is compiled with -ffp-contract=off so no
// multiply-add is fused (the reference build has no FMA: -mavx2 without -mfma,
// src/mi_fieldcalc/CMakeLists.txt:55-56).
#ifndef MIFC_DEVICE_H
#define MIFC_DEVICE_H

#include <hip/hip_runtime.h>

namespace mifc {

typedef unsigned long long u64;

// MetConstants.h:43-53.  Same float values as the reference: each is the
// double literal (or double expression) rounded to float.
#define MIFC_K_R 287.0f
#define MIFC_K_CP 1004.0f
#define MIFC_K_P0INV ((float)(1. / 1000.0))
#define MIFC_K_T0 ((float)273.15)
#define MIFC_K_EPS ((float)0.622)
#define MIFC_K_XLH ((float)2.501e+6)
#define MIFC_K_KAPPA (287.0f / 1004.0f)
#define MIFC_K_G ((float)9.8)
#define MIFC_K_RHMIN ((float)0.02)
#define MIFC_K_RHMAX 1.0f
#define MIFC_N_EWT 41
// LDS copy of the table: the 41 values, four times +infinity (the inverse lookup's compares read up to four entries past
// its start index without a range check), then 26 start indices for the inverse lookup (one per binary exponent of the
// argument, see Ewt::inverse)
#define MIFC_EWT_PAD 4
#define MIFC_EWT_FIRST_AT (MIFC_N_EWT + MIFC_EWT_PAD)
#define MIFC_EWT_FIRST_N 26
// floats of LDS: ewt[41], +inf[4], first[26], one pad, then 41 doubles: 1 / (ewt[k+1] - ewt[k]) (the table must be 8-byte aligned)
#define MIFC_EWT_RCP_AT (MIFC_EWT_FIRST_AT + MIFC_EWT_FIRST_N + 1)
#define MIFC_EWT_LDS (MIFC_EWT_RCP_AT + 2 * MIFC_N_EWT)

// FieldCalculations.h:42-45
__device__ __forceinline__ bool is_def(float x, float undef)
{
  return !(x != x) && x != undef;
}
// "every one of these is defined" as straight-line code: all the compares, combined without short-circuits.  A chain of
// is_def() && is_def() && ... in front of a conditionally evaluated formula becomes a nest of exec-mask regions and
// branches per cell (the tested variant of the fused wind kernel ran 11 % behind the untested one that way, 3.5 % now);
// kernels use this and a select behind the unconditionally computed formula: pick(ok, value, undef).
template <typename... T>
__device__ __forceinline__ bool all_def(float undef, T... x)
{
  return (is_def(x, undef) & ...);
}
__device__ __forceinline__ float pick(bool ok, float value, float undef)
{
  return ok ? value : undef;
}

// ---- saturation vapour pressure table (MetConstants.h:56-84, MetConstants.cc:37-45)
// The 41-entry table lives in LDS (164 B per workgroup): lookups are per-lane
// dynamic indices, which LDS serves without a trip through the vector cache.
struct EwtTable
{
  const float* tab; // LDS
};

#include "mifc_ewt_image.h"
// Stages the table image (tools/gen_ewt_image.py: the 41 values, the start indices of the inverse lookup and the
// reciprocals of the bin widths, 600 bytes) in LDS: a plain copy, one dword per lane.  `sync = false` leaves the
// barrier to a second staging call that follows (one barrier for both tables).
__device__ __forceinline__ void ewt_table_init(float* lds_tab, bool sync = true)
{
  static_assert(MIFC_EWT_IMAGE_WORDS == MIFC_EWT_LDS, "table image and LDS layout must agree");
  unsigned int* dst = reinterpret_cast<unsigned int*>(lds_tab);
  for (int k = threadIdx.x; k < MIFC_EWT_IMAGE_WORDS; k += blockDim.x)
    dst[k] = mifc_ewt_image[k];
  if (sync)
    __syncthreads();
}

struct Ewt
{
  float x;
  int lc; // int(x) brought into the table: what the lookups index with, so that they need no branch around them --
          // every point function below computes unconditionally and reports ok(); a caller discards the value of
          // a cell that is not ok (the reference: cell := undef, n_undefined += 1).  lc == int(x) whenever ok().
  __device__ __forceinline__ explicit Ewt(float t_celsius)
  {
    x = (float)(((double)t_celsius + 100.) * 0.2); // MetConstants.h:65
    // l = int(x), MetConstants.h:66; the compiled reference (x86-64 cvttss2si) yields INT_MIN for NaN and out-of-range
    // values, i.e. "not in the table" like any other l outside 0 .. 39.  Truncation toward zero: l >= 0 <=> x > -1,
    // l < 40 <=> x < 40, and neither holds for NaN -- ok() needs no integer; the index is taken from x clamped to the
    // table (NaN: v_med3_f32 returns the smallest operand, 0).
    lc = (int)__builtin_amdgcn_fmed3f(x, 0.f, (float)(MIFC_N_EWT - 2));
  }
  __device__ __forceinline__ bool ok() const { return x > -1.f && x < (float)(MIFC_N_EWT - 1); }
  __device__ __forceinline__ float value(const float* tab) const { return tab[lc] + (tab[lc + 1] - tab[lc]) * (x - (float)lc); }
  // MetConstants.cc:37-45: `ll = l; while (ll > 0 && ll < 40 && ewt[ll] > et) ll--;` -- a walk
  // down the (strictly increasing) table whose length differs from lane to lane.  The same ll
  // without a loop: the walk stops at the largest k <= l for which !(ewt[k] > et) holds, or at
  // 0.  That k is found from the binary exponent of et (first[]: where the table crosses each
  // power of two; the table crosses at most 4 entries per octave) plus four independent
  // compares.  NaN never satisfies `>`: the walk does not move (ll = l).
  __device__ __forceinline__ float inverse(const float* tab, float et) const
  {
#ifdef MIFC_EWT_INVERSE_WALK
    int ll = lc;
    while (ll > 0 && ll < MIFC_N_EWT - 1 && tab[ll] > et)
      ll--;
#else
    const int bits = __float_as_int(et);
    const int e = ((bits >> 23) & 0xff) - 127;
    int b = e < -15 ? 0 : (e > 10 ? 25 : e + 15);
    if (bits < 0)
      b = 0; // negative: no entry satisfies the predicate, the compares below all fail
    const int m0 = __float_as_int(tab[MIFC_EWT_FIRST_AT + b]);
    // ewt[m0 + 1 .. m0 + 4]: four reads at constant offsets from one address; past the table they find +infinity, which
    // is `> et` for every finite et.  (et = +inf or NaN may count up to m0 + 4 = 44: any m above lc means "stay at lc".)
    const float* at = tab + m0;
    int m = m0;
#pragma unroll
    for (int j = 1; j <= MIFC_EWT_PAD; ++j)
      m += !(at[j] > et) ? 1 : 0;
    if (et != et)
      m = MIFC_N_EWT - 1; // a NaN with the sign bit set started from 0
    const int ll = m < lc ? m : lc;
#endif
    // (et - ewt[ll]) / (ewt[ll+1] - ewt[ll]): a quotient of two floats is never closer than 2^-49 (relative) to a
    // rounding boundary of float, so the double product with the correctly rounded reciprocal of the bin width
    // (off by < 2^-52), rounded to float, IS the correctly rounded float quotient -- 4 instructions instead of 12
    const double* rcp = reinterpret_cast<const double*>(tab + MIFC_EWT_RCP_AT);
    const float r = (float)((double)(et - tab[ll]) * rcp[ll]);
    return (float)(-100. + (double)((float)ll + r) * 5.);
  }
};

// FieldCalculations.cc:186-194
__device__ __forceinline__ float clamp_rh(float rh)
{
  if (rh < MIFC_K_RHMIN)
    return MIFC_K_RHMIN;
  if (rh > MIFC_K_RHMAX)
    return MIFC_K_RHMAX;
  return rh;
}

// FieldCalculations.cc:308-311: powf(p * p0inv, kappa).
// The reference calls glibc's powf, which evaluates in double and returns a
// nearly correctly rounded float (1 ulp off in 0.07 % of the arguments).  The
// device's float powf is several ulp off, and the saturation-pressure table
// amplifies a temperature error (up to ~0.3 per kelvin at its cold end), so
// the power is taken in double here.  Round 2: DIRECT tables for this one
// exponent (mifc_kappa_tables.h, generated and verified by tools/gen_kappa_tables.py):
//   x = 2^e * m, m in [1, 2);  i = top 8 mantissa bits -> 256 intervals, centre c_i
//   r = m * (1/c_i) - 1, |r| <= 2^-9
//   x^kappa = 2^(kappa e) * c_i^kappa * (1 + k1 r + k2 r^2 + k3 r^3)     (truncation <= 6e-13 relative)
// 8 fp64-pipe operations + one 16-byte and one 8-byte LDS read, instead of the 26
// of the log2 / exp2 route below (kept for the generic powers of the catalogue);
// the float result equals the correctly rounded power in all but 2 of 4.5e6 sampled
// arguments (1 ulp).  Agreement with glibc is "identical in 99.9 % of the cells,
// 1 ulp otherwise"; the parity bound for the operators that use it is 1e-5 relative
// (BASELINE.json), not bit-exact.  The fused multiply-adds are explicit here
// (accuracy, not parity, matters).  Arguments outside [2^-32, 2^32) -- no pressure
// is -- and special values behave like powf with a positive non-integer exponent:
// x<0 -> NaN, 0 -> 0, inf -> inf, NaN -> NaN (pow_kappa_rare, inline).
#include "mifc_kappa_tables.h"
#define MIFC_POW_LOG_N 16
#define MIFC_POW_EXP_N 32
__device__ const double mifc_pow_log_tab[MIFC_POW_LOG_N][2] = { // {1/c_i, log2 c_i}
  {0x1.62362d911af4bp+0, -0x1.dfb604b80ff0bp-2},
  {0x1.5387df9e8cf1dp+0, -0x1.a12d0f7c7f665p-2},
  {0x1.4604b5723d1cdp+0, -0x1.652e54c9d2f3cp-2},
  {0x1.398a5602df4f6p+0, -0x1.2b8710cfa9b50p-2},
  {0x1.2dfb78d4d5815p+0, -0x1.e814714fed5aap-3},
  {0x1.233eff7fab36cp+0, -0x1.7d1f4c3ccae3fp-3},
  {0x1.193f3ead95327p+0, -0x1.15e6d11e0ebd3p-3},
  {0x1.0fe96b8d03f02p+0, -0x1.6454cd5fac0edp-4},
  {0x1.072d25821e305p+0, -0x1.46bf208b64f9bp-5},
  {0x1.fbf473240c30dp-1, 0x1.76faa2375b368p-7},
  {0x1.de4c26de910dfp-1, 0x1.925f8d1536355p-4},
  {0x1.c3e983554318cp-1, 0x1.70d90505e826bp-3},
  {0x1.ac492704f0bfbp-1, 0x1.07c076b5b9bfep-2},
  {0x1.9701cc88e61a3p-1, 0x1.530948ef4443fp-2},
  {0x1.83be1cbf011acp-1, 0x1.9aab6676ab89bp-2},
  {0x1.72382dda556d1p-1, 0x1.defd4ba085026p-2},
};
__device__ const double mifc_pow_exp_tab[MIFC_POW_EXP_N] = { // 2^(j/32)
  0x1.0000000000000p+0, 0x1.059b0d3158574p+0, 0x1.0b5586cf9890fp+0, 0x1.11301d0125b51p+0,
  0x1.172b83c7d517bp+0, 0x1.1d4873168b9aap+0, 0x1.2387a6e756238p+0, 0x1.29e9df51fdee1p+0,
  0x1.306fe0a31b715p+0, 0x1.371a7373aa9cbp+0, 0x1.3dea64c123422p+0, 0x1.44e086061892dp+0,
  0x1.4bfdad5362a27p+0, 0x1.5342b569d4f82p+0, 0x1.5ab07dd485429p+0, 0x1.6247eb03a5585p+0,
  0x1.6a09e667f3bcdp+0, 0x1.71f75e8ec5f74p+0, 0x1.7a11473eb0187p+0, 0x1.82589994cce13p+0,
  0x1.8ace5422aa0dbp+0, 0x1.93737b0cdc5e5p+0, 0x1.9c49182a3f090p+0, 0x1.a5503b23e255dp+0,
  0x1.ae89f995ad3adp+0, 0x1.b7f76f2fb5e47p+0, 0x1.c199bdd85529cp+0, 0x1.cb720dcef9069p+0,
  0x1.d5818dcfba487p+0, 0x1.dfc97337b9b5fp+0, 0x1.ea4afa2a490dap+0, 0x1.f50765b6e4540p+0,
};

struct PowTables
{
  const double* logt; // LDS, [16][2]: generic log2 (log_float / exp_float / pow_float)
  const double* expt; // LDS, [32]
  const double* kit;  // LDS, [256][2] = {1/c_i, c_i^kappa}: pow_kappa
  const double* ket;  // LDS, [64] = 2^(kappa e)
};
#define MIFC_KAPPA_LDS (2 * MIFC_KAPPA_N + MIFC_KAPPA_NE) // doubles

// stages both tables in LDS (512 B); call once per workgroup before the first use
__device__ __forceinline__ PowTables pow_tables_init(double* lds_tab /* 64 doubles */)
{
  for (int k = threadIdx.x; k < 2 * MIFC_POW_LOG_N; k += blockDim.x)
    lds_tab[k] = (&mifc_pow_log_tab[0][0])[k];
  for (int k = threadIdx.x; k < MIFC_POW_EXP_N; k += blockDim.x)
    lds_tab[2 * MIFC_POW_LOG_N + k] = mifc_pow_exp_tab[k];
  __syncthreads();
  PowTables t;
  t.logt = lds_tab;
  t.expt = lds_tab + 2 * MIFC_POW_LOG_N;
  t.kit = nullptr;
  t.ket = nullptr;
  return t;
}

// stages the x^kappa tables in LDS (4.6 KiB); call once per workgroup before the first pow_kappa
__device__ __forceinline__ PowTables kappa_tables_init(double* lds_tab /* MIFC_KAPPA_LDS doubles */)
{
  for (int k = threadIdx.x; k < 2 * MIFC_KAPPA_N; k += blockDim.x)
    lds_tab[k] = (&mifc_kappa_it[0][0])[k];
  for (int k = threadIdx.x; k < MIFC_KAPPA_NE; k += blockDim.x)
    lds_tab[2 * MIFC_KAPPA_N + k] = mifc_kappa_et[k];
  __syncthreads();
  PowTables t;
  t.logt = nullptr;
  t.expt = nullptr;
  t.kit = lds_tab;
  t.ket = lds_tab + 2 * MIFC_KAPPA_N;
  return t;
}

// log2 of a positive NORMAL float, in double.  Absolute error ~5e-12 (table of 16 centres, |r| <= 1/32,
// series to r^6), i.e. fine as the inner function of a power; functions that return the logarithm itself
// use log2_near_one() around 1, where the RELATIVE error matters.
__device__ __forceinline__ double log2_tab(const PowTables& T, float x)
{
  const int ix = __float_as_int(x);
  const int e = (ix - 0x3f3504f3) >> 23; // x / 2^e in [sqrt(1/2), sqrt(2))
  const int im = ix - (e << 23);
  const int i = (im - 0x3f3504f3) >> 19;
  const double m = (double)__int_as_float(im);
  const double r = fma(m, T.logt[2 * i], -1.0);
  double p = -1.0 / 6.0;
  p = fma(p, r, 1.0 / 5.0);
  p = fma(p, r, -1.0 / 4.0);
  p = fma(p, r, 1.0 / 3.0);
  p = fma(p, r, -0.5);
  p = fma(p, r, 1.0);
  return fma(r * 1.4426950408889634 /* 1/ln 2 */, p, T.logt[2 * i + 1] + (double)e);
}
// log2(x) for |x - 1| < 1/32: r = x - 1 is exact, the series then has a relative error below 2e-10
__device__ __forceinline__ double log2_near_one(float x)
{
  const double r = (double)x - 1.0;
  double p = 1.0 / 7.0;
  p = fma(p, r, -1.0 / 6.0);
  p = fma(p, r, 1.0 / 5.0);
  p = fma(p, r, -1.0 / 4.0);
  p = fma(p, r, 1.0 / 3.0);
  p = fma(p, r, -0.5);
  p = fma(p, r, 1.0);
  return r * 1.4426950408889634 * p;
}
// 2^t in double for |t| < 1024 (relative error ~2e-12): k = rint(32 t), 2^(k>>5) * 2^((k&31)/32) * e^g
__device__ __forceinline__ double exp2_tab(const PowTables& T, double t)
{
  const double k = rint(t * 32.0);
  const double g = fma(k, -1.0 / 32.0, t) * 0.6931471805599453 /* ln 2 */;
  const int ki = (int)k;
  double q = 1.0 / 24.0;
  q = fma(q, g, 1.0 / 6.0);
  q = fma(q, g, 0.5);
  q = fma(q, g, 1.0);
  q = fma(q, g, 1.0);
  return ldexp(T.expt[ki & 31] * q, ki >> 5);
}

// x^kappa in double for the bits of a float in [2^-32, 2^32)
__device__ __forceinline__ double pow_kappa_core(const PowTables& T, int ix)
{
  // exponent index e - EMIN = biased exponent - (127 + EMIN); masked into the table so that bits outside the domain (whose
  // result the callers replace) read some entry of it instead of memory beside it
  static_assert(MIFC_KAPPA_NE == 64, "the exponent index is masked with 63");
  const int ei = (((ix >> 23) & 0xff) - (127 + MIFC_KAPPA_EMIN)) & (MIFC_KAPPA_NE - 1);
  const int i = (ix >> 15) & 0xff;
  const double m = (double)__int_as_float((ix & 0x007fffff) | 0x3f800000);
  const double r = fma(m, T.kit[2 * i], -1.0);
  double p = MIFC_KAPPA_K3;
  p = fma(p, r, MIFC_KAPPA_K2);
  p = fma(p, r, MIFC_KAPPA_K1);
  p = fma(p, r, 1.0);
  return T.ket[ei] * T.kit[2 * i + 1] * p;
}
// The arguments outside [2^-32, 2^32) -- no pressure is; what reaches here are undefined cells a tested kernel
// computes and discards -- inline and without a function call: a call inside a kernel's main loop makes the
// compiler wait for EVERY outstanding load at the loop head (s_waitcnt vmcnt(0): the counters are unknown after a
// call), which silently undid the two-trip software pipeline of the fused derived kernel.  A finite positive x is
// brought into the table's range by exact scalings with 2^+-64, 2^(+-64 kappa) multiplied back in double.
__device__ __forceinline__ float pow_kappa_rare(const PowTables& T, float x)
{
  if (x != x || x < 0.f)
    return __int_as_float(0x7fc00000);
  if (x == 0.f)
    return 0.f;
  if (x == __int_as_float(0x7f800000))
    return x;
  float xs;
  double c;
  if (x < 1.f) {
    xs = x * 0x1p64f;
    c = 0x1.a15e6ebf53f28p-19; // 2^(-64 kappa)
    if (xs < 2.3283064365386963e-10f) {
      xs *= 0x1p64f;
      c = 0x1.543a63d069f35p-37; // 2^(-128 kappa)
    }
  } else {
    xs = x * 0x1p-64f;
    c = 0x1.3a0b257a9f5abp+18; // 2^(64 kappa)
    if (xs >= 4294967296.0f) {
      xs *= 0x1p-64f;
      c = 0x1.813f586d0cf0ep+36; // 2^(128 kappa)
    }
  }
  return (float)(pow_kappa_core(T, __float_as_int(xs)) * c);
}
__device__ __forceinline__ float pow_kappa(const PowTables& T, float x)
{
  // 2^-32 <= x < 2^32 on the bits: one subtraction and one unsigned compare (negative, NaN and infinite x fail it too)
  const int ix = __float_as_int(x);
  const bool fast = (unsigned)(ix - 0x2f800000) < 0x20000000u;
  float out = (float)pow_kappa_core(T, ix); // unconditional: the core keeps its table reads in range for any bits
  if (__builtin_expect(!fast, 0))
    out = pow_kappa_rare(T, x);
  return out;
}

// ---- the float libm functions of the catalogue (logf, log10f, expf, powf and the double exp / pow the
// reference rounds to float), evaluated in double with the tables above and rounded once: within 1 ulp of
// the correctly rounded float (glibc's own float functions are), far inside the 1e-5 bar.  Arguments
// outside the fast domain (zero, negative, subnormal, infinite, NaN, huge exponents) take the device
// library's double function, out of line.
__device__ __attribute__((noinline)) float libm_log_slow(float x, int base10)
{
  return (float)(base10 ? log10((double)x) : log((double)x));
}
__device__ __attribute__((noinline)) float libm_exp_slow(double t, int base10)
{
  return (float)(base10 ? pow(10.0, t) : exp(t));
}
__device__ __attribute__((noinline)) float libm_pow_slow(double a, double b)
{
  return (float)pow(a, b);
}
__device__ __forceinline__ bool positive_normal(float x)
{
  return x >= 1.17549435e-38f && x <= 3.40282347e38f;
}
__device__ __forceinline__ double log2_any(const PowTables& T, float x) // x positive normal
{
  return (__builtin_fabsf(x - 1.0f) < 0.03125f) ? log2_near_one(x) : log2_tab(T, x);
}
// logf / log10f
__device__ __forceinline__ float log_float(const PowTables& T, float x, bool base10)
{
  if (!positive_normal(x))
    return libm_log_slow(x, base10 ? 1 : 0);
  return (float)(log2_any(T, x) * (base10 ? 0.30102999566398120 /* log10 2 */ : 0.6931471805599453 /* ln 2 */));
}
// (float)exp(a) / (float)pow(10, a) for a double argument
__device__ __forceinline__ float exp_float(const PowTables& T, double a, bool base10)
{
  const double t = a * (base10 ? 3.3219280948873623 /* log2 10 */ : 1.4426950408889634 /* log2 e */);
  if (!(t > -1000.0 && t < 1000.0))
    return libm_exp_slow(a, base10 ? 1 : 0);
  return (float)exp2_tab(T, t);
}
// (float)pow((double)a, b)
__device__ __forceinline__ float pow_float(const PowTables& T, float a, double b)
{
  if (!positive_normal(a))
    return libm_pow_slow((double)a, b);
  const double t = b * log2_any(T, a);
  if (!(t > -1000.0 && t < 1000.0))
    return libm_pow_slow((double)a, b);
  return (float)exp2_tab(T, t);
}

__device__ __forceinline__ float pidcp_of(const PowTables& T, float p)
{
  return pow_kappa(T, p * MIFC_K_P0INV);
}

// ---- two double quotients with one denominator
// The compiler expands a / b (f64) into v_div_scale x2, v_rcp_f64, four Newton fmas, then
// q = a*y, r = fma(-b, q, a), v_div_fmas(r, y, q), v_div_fixup (LLVM AMDGPU LowerFDIV64).
// v_div_scale only rescales operands near the ends of the double range (denormal b, 1/b or
// a/b, |a| < 2^-969, exponents >= 768 apart); values converted from float, and products of
// two or three of them, never get there, so for those the sequence below IS that expansion,
// bit for bit -- and quotients that share b share the v_rcp_f64 and the four fmas.
// Zero, infinite and NaN operands are settled by v_div_fixup as in the expansion.
__device__ __forceinline__ double shared_reciprocal(double b)
{
  double y = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-b, y, 1.0);
  return __builtin_fma(y, e, y);
}
__device__ __forceinline__ double quotient(double a, double b, double y /* shared_reciprocal(b) */)
{
  const double q = a * y;
  const double r = __builtin_fma(-b, q, a);
  return __builtin_amdgcn_div_fixup(__builtin_fma(r, y, q), b, a);
}

// The point functions return false where the table does not cover tk
// (reference: cell := undef, n_undefined += 1).
// FieldCalculations.cc:196-205
__device__ __forceinline__ bool t_thesat(const float* tab, float tk, float p, float pi, float& out)
{
  Ewt e(tk - MIFC_K_T0);
  const float qsat = MIFC_K_EPS * e.value(tab) / p;
  out = (MIFC_K_CP * tk + MIFC_K_XLH * qsat) / pi;
  return e.ok();
}
// FieldCalculations.cc:207-216
__device__ __forceinline__ bool th_thesat(const float* tab, float th, float p, float pi, float& out)
{
  Ewt e(th * pi / MIFC_K_CP - MIFC_K_T0);
  const float qsat = MIFC_K_EPS * e.value(tab) / p;
  out = th + MIFC_K_XLH * qsat / pi;
  return e.ok();
}
// FieldCalculations.cc:218-227
__device__ __forceinline__ bool tk_q_rh(const float* tab, float tk, float q, float p, float& out)
{
  Ewt e(tk - MIFC_K_T0);
  const float qsat = MIFC_K_EPS * e.value(tab) / p;
  // 100. * q / qsat in double: the division through the refined reciprocal -- for float-born operands it IS the
  // f64 division, bit for bit (see shared_reciprocal above; mifc_diag_division checks it), three instructions shorter
  const double d = (double)qsat;
  out = (float)quotient(100. * (double)q, d, shared_reciprocal(d));
  return e.ok();
}
// FieldCalculations.cc:229-238
__device__ __forceinline__ bool tk_rh_q(const float* tab, float tk, float rh, float p, float& out)
{
  Ewt e(tk - MIFC_K_T0);
  const float qsat = MIFC_K_EPS * e.value(tab) / p;
  out = (float)(0.01 * (double)rh * (double)qsat);
  return e.ok();
}
// FieldCalculations.cc:240-253
__device__ __forceinline__ bool tk_q_td(const float* tab, float tk, float q, float p, float tdconv, float& out)
{
  Ewt e(tk - MIFC_K_T0);
  const float et = e.value(tab);
  const float qsat = MIFC_K_EPS * et / p;
  const float rh = clamp_rh(q / qsat);
  out = e.inverse(tab, rh * et) + tdconv;
  return e.ok();
}
// FieldCalculations.cc:255-267
__device__ __forceinline__ bool tk_rh_td(const float* tab, float tk, float rh100, float tdconv, float& out)
{
  Ewt e(tk - MIFC_K_T0);
  const float et = e.value(tab);
  const float rh = clamp_rh((float)(0.01 * (double)rh100));
  out = e.inverse(tab, rh * et) + tdconv;
  return e.ok();
}

// math_util.h:57-60: sqrt(x*x + y*y) in float, correctly rounded sqrt, no fma
__device__ __forceinline__ float absval(float x, float y)
{
  return sqrtf(x * x + y * y);
}

// (float)(0.5 * (double)m * (double)d) for floats m and d -- the "float-rounded partial" of the reference's
// gradient-type operators (e.g. FieldCalculations.cc:2015, :2040, :2445).  Both factors are float-born, so
// the double product 0.5*m*d is EXACT (25 + 24 significant bits) and the expression is the correctly rounded
// float of the exact product.  A float multiply of two floats whose exact product is the same number gives
// that very rounding; what has to be exact is the halving of one factor, and halving the factor of larger
// magnitude always is (it is either >= 2^-125, where halving only lowers the exponent, or both factors are
// so small that the product rounds to the same signed zero either way).  Five fp64-pipe instructions
// (three conversions, two multiplies) become two selects and two float multiplies; the result is the
// reference's bit for bit, infinities, zeros and NaNs included (a NaN's payload may differ).
__device__ __forceinline__ float half_prod(float m, float d)
{
  const bool m_larger = __builtin_fabsf(m) >= __builtin_fabsf(d);
  const float a = m_larger ? 0.5f * m : m;
  const float b = m_larger ? d : 0.5f * d;
  return a * b;
}

// ---- stencil point formulas (double-promoted; inputs are float differences)
#ifdef MIFC_EXPERIMENT_F32_COMBINE
// NOT parity-correct: float-only combine, exists solely so that tools/ can
// measure how much of the kernel time is the fp64 pipe (never built into the product)
__device__ __forceinline__ float f_relvort(float xm, float ym, float dvdx, float dudy) { return 0.5f * xm * dvdx - 0.5f * ym * dudy; }
__device__ __forceinline__ float f_diverg(float xm, float ym, float dudx, float dvdy) { return 0.5f * xm * dudx + 0.5f * ym * dvdy; }
#define MIFC_SKIP_F64_COMBINE 1
#endif
#ifndef MIFC_SKIP_F64_COMBINE
// The reference evaluates a*b -+ c*d in double with a, b, c, d converted from float (0.5 * xm is
// exact).  A product of two float-born doubles has at most 48 significant bits: it is EXACT in
// double, so the only rounding of the expression is the one of the sum -- which is precisely what
// one fused multiply-add on an exact second product delivers: fma(a, b, -+(c*d)) == a*b -+ c*d bit
// for bit (zeros, infinities and NaNs included; only a NaN's payload may differ).  One fp64
// instruction fewer per result; the library is otherwise built with -ffp-contract=off.
// FieldCalculations.cc:1862
__device__ __forceinline__ float f_relvort(float xm, float ym, float dvdx, float dudy)
{
  return (float)__builtin_fma(0.5 * (double)xm, (double)dvdx, -(0.5 * (double)ym * (double)dudy));
}
// FieldCalculations.cc:1896
__device__ __forceinline__ float f_absvort(float xm, float ym, float dvdx, float dudy, float fc)
{
  return (float)(__builtin_fma(0.5 * (double)xm, (double)dvdx, -(0.5 * (double)ym * (double)dudy)) + (double)fc);
}
// FieldCalculations.cc:1928
__device__ __forceinline__ float f_diverg(float xm, float ym, float dudx, float dvdy)
{
  return (float)__builtin_fma(0.5 * (double)xm, (double)dudx, 0.5 * (double)ym * (double)dvdy);
}
#endif

// ---- EXTENSION (no reference function; BASELINE.json's north_star names "wind direction from u/v",
// FieldCalculations.cc:1951-1952 only mentions it in a comment): the meteorological wind direction,
// i.e. where the wind blows FROM, in degrees clockwise from north:
//   dd = 270 - atan2(v, u) * 180 / pi, brought into [0, 360); calm (u == v == 0) gives 0.
// Evaluated in float as dd = atan2(-u, -v) * 180 / pi (+360 when negative): the same angle as 270 - atan2(v, u) * 180 / pi
// without the subtraction, whose cancellation near north (dd -> 0) cost all relative accuracy (round 2: 2e-3 degrees
// absolute).  Now within 1e-5 RELATIVE of the float64 definition rounded once (the test suite's CPU restatement)
// everywhere; the definition, not a reference result, is what the tests pin.
__device__ __forceinline__ float wind_direction(float u, float v)
{
  if (u == 0.f && v == 0.f)
    return 0.f;
  float dd = atan2f(-u, -v) * 57.29577951308232f;
  if (dd < 0.f)
    dd += 360.f;
  if (dd >= 360.f || dd == 0.f) // -1e-6 + 360 rounds to 360: the same direction as 0; and no negative zero (wind from due north)
    dd = 0.f;
  return dd;
}

// ---- the input flag of a level inside a level-walking kernel
// flags[lev] != 0 <=> the level's input flag is ALL_DEFINED (no tests).  gfx9 has no scalar byte load, so the plain
// expression flags[lev] compiles into global_load_ubyte + s_waitcnt vmcnt(0): one more entry in the wave's in-order
// vector-memory queue, BEHIND every store (and prefetch) the wave still has in flight -- the wait for one byte is a wait
// for all of them, once per level, which is exactly the load/store coupling the split-role kernels exist to avoid
// (round 3: their tested variants ran 4-50 % behind the untested ones because of it).  This reads the aligned dword
// around the byte through the scalar cache instead (lgkmcnt, which the level's barrier waits for anyway) and returns
// after the workgroup barrier of the level: `s_waitcnt lgkmcnt(0); s_barrier` with the flag for free.
// The flag arrays are the context's (capacity a multiple of 4, 256-byte aligned base), so the dword never leaves them.
__device__ __forceinline__ bool level_flag_then_barrier(const unsigned char* flags /* wave-uniform, may be null */, int lev /* wave-uniform */)
{
  if (flags == nullptr) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    return false;
  }
  const unsigned long long a = (unsigned long long)(flags + lev);
  // (wave-uniform by contract; said again here because the "s" operand below cannot take a value the compiler keeps in VGPRs)
  const unsigned long long base = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(a >> 32)) << 32) |
                                  (unsigned long long)((unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)a) & ~3u);
  unsigned int w;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" : "=&s"(w) : "s"(base) : "memory");
  return ((w >> ((unsigned int)(a & 3ull) * 8u)) & 0xffu) != 0u;
}

// ---- undefined-cell counting
// Atomics to ONE address are served one after the other (~12 ns each on MI355X: profiles/r02/experiments/undef_density.txt);
// a masked field, where every wave has something to count, turned the 0.3 ms of a one-shot elementwise kernel into
// 6.7 ms of queueing behind its 555 000 per-wave atomics.  So: per-lane counts -> wave butterfly -> LDS atomic ->
// ONE global atomic per WORKGROUP (block_count_add, for kernels in which every wave of the workgroup reaches the call),
// or one atomic per wave where the waves of a workgroup do not meet again (wave_count_add); nothing at all where
// there is nothing to count.
__device__ __forceinline__ unsigned int wave_sum(unsigned int my_count) // total in lane 0 (all lanes, in fact)
{
  unsigned int s = my_count;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    s += __shfl_xor(s, off, 64);
  return s;
}
__device__ __forceinline__ void wave_count_add(u64* counter, unsigned int my_count)
{
  if (__builtin_amdgcn_ballot_w64(my_count != 0) == 0) // nothing to count in this wave (the usual case): no shuffles, no atomic
    return;
  const unsigned int s = wave_sum(my_count);
  if ((threadIdx.x & 63) == 0 && s != 0)
    atomicAdd(counter, (u64)s);
}
// N counters at once; contains barriers: every wave of the workgroup must call it, with the same N counters
template <int N>
__device__ __forceinline__ void block_count_add(u64* const (&counter)[N], const unsigned int (&my_count)[N])
{
  __shared__ unsigned int s_total[N];
  if (threadIdx.x < N)
    s_total[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (counter[k] && __builtin_amdgcn_ballot_w64(my_count[k] != 0) != 0) {
      const unsigned int s = wave_sum(my_count[k]);
      if ((threadIdx.x & 63) == 0)
        atomicAdd(&s_total[k], s);
    }
  }
  __syncthreads();
  if (threadIdx.x < N && counter[threadIdx.x] && s_total[threadIdx.x] != 0)
    atomicAdd(counter[threadIdx.x], (u64)s_total[threadIdx.x]);
}
// the workgroup's count into its own slot (always written: the slots are not zeroed beforehand); barriers inside
__device__ __forceinline__ void block_count_store(unsigned int* slot, unsigned int my_count)
{
  __shared__ unsigned int s_total1;
  if (threadIdx.x == 0)
    s_total1 = 0;
  __syncthreads();
  if (__builtin_amdgcn_ballot_w64(my_count != 0) != 0) {
    const unsigned int s = wave_sum(my_count);
    if ((threadIdx.x & 63) == 0)
      atomicAdd(&s_total1, s);
  }
  __syncthreads();
  if (threadIdx.x == 0)
    *slot = s_total1;
}
__device__ __forceinline__ void block_count_add(u64* counter, unsigned int my_count)
{
  u64* const c[1] = {counter};
  const unsigned int n[1] = {my_count};
  block_count_add<1>(c, n);
}

} // namespace mifc

#endif // MIFC_DEVICE_H
