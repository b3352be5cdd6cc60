// mifc_derived.hip -- fused derived variables on hybrid model levels, a whole
// batch of levels in one launch (BASELINE.json config 2 x nlev; north_star:
// "wind speed, humidity/dewpoint, potential temperature fused per vertical level").
//
// Per level l and cell, any subset of
//   ff   = vectorabs(u, v)                                      FieldCalculations.cc:1819-1841
//   temp = hleveltemp(t, ps, a_l, b_l, unit, compute 1..5)      :1046-1098
//   hum  = hlevelhum(t, h, ps, a_l, b_l, unit, compute 1..12)   :1145-1217
//   td   = a second hlevelhum variant of the same inputs (e.g. compute 9: T, q -> Td)
// from ONE read of u, v, t, h (16 B/cell) and ps (shared by all levels: L2),
// p = a + b * ps (:303) and the Exner power (:308) evaluated once per cell.
// Every output equals the per-level reference call on that level.
//
// Mapping to the hardware
//   * grid = (gx, nlev) with gx chosen so that the launch has ~8 workgroups per CU:
//     a workgroup stays on its level and walks it with a grid-stride loop, so the
//     saturation-pressure and x^kappa tables (1.3 KiB of LDS) are staged ONCE per
//     workgroup -- not in front of every 1024 cells -- and the per-level scalars
//     are wave-uniform (SGPRs).
//   * all global accesses are 16 B per lane, coalesced; the loads of trip k+1 are
//     issued before trip k is computed (two register sets, swapped by unrolling),
//     so the ~140 VALU instructions per cell (fp64 x^kappa, correctly rounded
//     divisions, table walks: the reference's arithmetic) overlap the memory
//     round trip of the next trip instead of following it.
//   * which outputs exist and which variant each one is are COMPILE-TIME for the
//     common combinations (ff + RH + theta, with or without Td); a generic
//     instantiation takes any other combination from the kernel arguments.
//   * outputs leave with nontemporal stores; undefined cells are counted per
//     lane, reduced per wave, one atomic per wave and output.
#include "mifc_device.h"
#include "mifc_kernels.h"

namespace mifc {

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st4(float* p, const float (&r)[4])
{
  v4f t;
  t.x = r[0];
  t.y = r[1];
  t.z = r[2];
  t.w = r[3];
  __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
}

struct Trip
{
  v4f u, v, t, h, s;
};

// humidity variant code: 0 none, else 1 + HumKind + 4 * from_theta
__host__ __device__ inline int hum_code(int kind, int from_theta)
{
  return 1 + kind + 4 * (from_theta ? 1 : 0);
}

// One humidity output of one cell; mirrors hlevelhum's loop body (:1186-1213).
//   code: see hum_code(); pidcp: powf(p * p0inv, kappa), only read when the variant starts from theta
__device__ __forceinline__ bool hum_point(int code, const float* tab, float tt, float hh, float p, float pidcp, float tdconv, float& out)
{
  const int kind = (code - 1) & 3;
  const bool from_theta = (code - 1) >= 4;
  const float tk = from_theta ? tt * pidcp : tt; // :1188-1189
  switch (kind) {
  case HUM_Q_RH:
    return tk_q_rh(tab, tk, hh, p, out);
  case HUM_RH_Q:
    return tk_rh_q(tab, tk, hh, p, out);
  case HUM_Q_TD:
    return tk_q_td(tab, tk, hh, p, tdconv, out);
  default:
    return tk_rh_td(tab, tk, hh, tdconv, out);
  }
}

// FF: 0 / 1, or -1 = from the arguments.  TC: temperature compute 0 (none) .. 5, -1 = arguments.
// HC / DC: humidity / second-humidity variant code (0 none), -1 = arguments.
template <bool CHECK, int FF, int TC, int HC, int DC, bool PIPE = true>
__global__ __launch_bounds__(256) void derived_levels_kernel(const DerivedParams P)
{
  __shared__ __attribute__((aligned(8))) float s_ewt[MIFC_EWT_LDS];
  __shared__ double s_pow[MIFC_KAPPA_LDS];

  const bool want_ff = FF >= 0 ? (FF != 0) : (P.ff != nullptr);
  const bool want_dd = P.dd != nullptr; // extension output, wave-uniform at run time in every instantiation
  const int tc = TC >= 0 ? TC : P.temp_compute;
  const int hc = HC >= 0 ? HC : P.hum_code;
  const int dc = DC >= 0 ? DC : P.td_code;
  const bool want_t = tc != 0, want_h = hc != 0, want_d = dc != 0;
  const bool thermo = want_t || want_h || want_d;
  // tables: the saturation pressure for every humidity variant and theta-e (compute 4, 5);
  // x^kappa for every temperature variant and the humidity variants that start from theta
  const bool need_ewt = want_h || want_d || tc >= 4;
  const bool need_pow = want_t || hc >= 5 || dc >= 5;
  PowTables PT = {nullptr, nullptr, s_pow, s_pow + 2 * MIFC_KAPPA_N};

  const int lev = blockIdx.y;
  const size_t base = (size_t)lev * (size_t)P.n;
  const bool wind_all = CHECK ? ((P.n_inline ? P.wind_inline[lev] : P.wind_all_defined[lev]) != 0) : true;
  const bool thermo_all = CHECK ? ((P.n_inline ? P.thermo_inline[lev] : P.thermo_all_defined[lev]) != 0) : true;
  const float a = thermo ? (P.n_inline ? P.a_inline[lev] : P.alevel[lev]) : 0.f;
  const float b = thermo ? (P.n_inline ? P.b_inline[lev] : P.blevel[lev]) : 0.f;
  const float undef = P.undef;
  // hlevelhum reads ps only when the variant needs p (:1182: not for RH -> Td from T), and tests it with != undef only (:1187)
  const bool h_need_p = want_h && hc != hum_code(HUM_RH_TD, 0);
  const bool d_need_p = want_d && dc != hum_code(HUM_RH_TD, 0);
  const bool read_ps = want_t || h_need_p || d_need_p;
  const bool read_h = want_h || want_d;
  unsigned int bad_ff = 0, bad_t = 0, bad_h = 0, bad_d = 0, bad_dd = 0;

  const float* __restrict__ pu = P.u + base;
  const float* __restrict__ pv = P.v + base;
  const float* __restrict__ pt = P.t + base;
  const float* __restrict__ ph = P.h + base;
  const float* __restrict__ ps = P.ps;

  // unsigned 32-bit cell-group indices: base pointer in SGPRs + zero-extended lane offset (saddr addressing)
  const unsigned n4 = (unsigned)P.n >> 2;
  const unsigned stride = gridDim.x * blockDim.x;
  const v4f zero = {0.f, 0.f, 0.f, 0.f};

  auto load = [&](unsigned q) -> Trip {
    Trip r;
    const unsigned o = q * 4u;
    r.u = (want_ff || want_dd) ? *reinterpret_cast<const v4f*>(pu + o) : zero;
    r.v = (want_ff || want_dd) ? *reinterpret_cast<const v4f*>(pv + o) : zero;
    r.t = thermo ? *reinterpret_cast<const v4f*>(pt + o) : zero;
    r.h = read_h ? *reinterpret_cast<const v4f*>(ph + o) : zero;
    r.s = read_ps ? *reinterpret_cast<const v4f*>(ps + o) : zero;
    return r;
  };

  float* __restrict__ off = P.ff ? P.ff + base : nullptr;
  float* __restrict__ otemp = P.temp ? P.temp + base : nullptr;
  float* __restrict__ ohum = P.hum ? P.hum + base : nullptr;
  float* __restrict__ otd = P.td ? P.td + base : nullptr;
  float* __restrict__ odd = P.dd ? P.dd + base : nullptr;
  auto compute = [&](unsigned q, const Trip& in) {
    const unsigned o = q * 4u;
    if (want_ff) { // vectorabs :1831-1837
      float r[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { // straight-line: arithmetic unconditional, the test selects
        const bool ok = wind_all | all_def(undef, in.u[k], in.v[k]);
        r[k] = pick(ok, absval(in.u[k], in.v[k]), undef);
        bad_ff += ok ? 0u : 1u;
      }
      st4(off + o, r);
    }
    if (want_dd) { // extension: wind direction, flag handling of vectorabs
      float r[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool ok = wind_all | all_def(undef, in.u[k], in.v[k]);
        r[k] = pick(ok, wind_direction(in.u[k], in.v[k]), undef);
        bad_dd += ok ? 0u : 1u;
      }
      st4(odd + o, r);
    }
    if (thermo) {
      float rt[4], rh[4], rd[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float tt = in.t[k], hh = in.h[k], ss = in.s[k];
        const float p = a + b * ss; // p_hlevel :303
        const float pidcp = need_pow ? pidcp_of(PT, p) : 1.f;
        if (want_t) { // hleveltemp :1077-1090
          float r = 0.f;
          bool ok = thermo_all | all_def(undef, tt, ss);
          { // unconditional: the point functions are branch-free (clamped table reads), the tests only select
            const float pi = pidcp * MIFC_K_CP;
            bool okp = true;
            switch (tc) {
            case 1:
              r = tt * pidcp - MIFC_K_T0;
              break;
            case 2:
              r = tt * pidcp;
              break;
            case 3:
              r = tt / pidcp;
              break;
            case 4:
              okp = t_thesat(s_ewt, tt, p, pi, r);
              break;
            default:
              okp = th_thesat(s_ewt, tt, p, pi, r);
              break;
            }
            ok = ok & okp;
          }
          rt[k] = ok ? r : undef;
          bad_t += ok ? 0u : 1u;
        }
        if (want_h) { // hlevelhum :1186-1213
          float r = 0.f;
          // the point function runs unconditionally (branch-free: its table reads are clamped), the tests only select
          const bool okp = hum_point(hc, s_ewt, tt, hh, h_need_p ? p : 0.f, pidcp, P.hum_tdconv, r);
          const bool ok = (thermo_all | (all_def(undef, tt, hh) & (!h_need_p | (ss != undef)))) & okp;
          rh[k] = ok ? r : undef;
          bad_h += ok ? 0u : 1u;
        }
        if (want_d) {
          float r = 0.f;
          const bool okp = hum_point(dc, s_ewt, tt, hh, d_need_p ? p : 0.f, pidcp, P.td_tdconv, r);
          const bool ok = (thermo_all | (all_def(undef, tt, hh) & (!d_need_p | (ss != undef)))) & okp;
          rd[k] = ok ? r : undef;
          bad_d += ok ? 0u : 1u;
        }
      }
      if (want_t)
        st4(otemp + o, rt);
      if (want_h)
        st4(ohum + o, rh);
      if (want_d)
        st4(otd + o, rd);
    }
  };

  // two trips per iteration: the loads of the next trip are in flight while this one is computed
  unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
  // the loads of the first trip go out before the lookup tables (4.9 KiB) are staged in LDS
  Trip cur = load(q < n4 ? q : 0u);
  if (need_ewt)
    ewt_table_init(s_ewt, !need_pow); // one barrier for both tables
  if (need_pow)
    PT = kappa_tables_init(s_pow);
  if (!PIPE) {
    if (q < n4)
      compute(q, cur);
    for (q += stride; q < n4; q += stride)
      compute(q, load(q));
  } else if (q < n4) {
    for (;;) {
      const unsigned q1 = q + stride;
      const bool more1 = q1 < n4;
      Trip nxt = load(more1 ? q1 : q);
      compute(q, cur);
      if (!more1)
        break;
      const unsigned q2 = q1 + stride;
      const bool more2 = q2 < n4;
      cur = load(more2 ? q2 : q1);
      compute(q1, nxt);
      if (!more2)
        break;
      q = q2;
    }
  }
  // The saturation table can reject a cell even when the inputs are ALL_DEFINED (:220-224), so the
  // humidity outputs and theta-e are always counted; ff / plain temperature only when tests ran.
  // one atomic per workgroup and output (every workgroup stays on one level)
  u64* const ctr[5] = {(want_ff && CHECK && P.cnt_ff) ? P.cnt_ff + lev : nullptr, (want_t && (CHECK || tc >= 4) && P.cnt_temp) ? P.cnt_temp + lev : nullptr,
                       (want_h && P.cnt_hum) ? P.cnt_hum + lev : nullptr, (want_d && P.cnt_td) ? P.cnt_td + lev : nullptr,
                       (want_dd && CHECK && P.cnt_dd) ? P.cnt_dd + lev : nullptr};
  const unsigned int cnt[5] = {bad_ff, bad_t, bad_h, bad_d, bad_dd};
  block_count_add<5>(ctr, cnt);
}

template <bool CHECK>
void launch_one(const DerivedParams& p, dim3 grid, hipStream_t stream)
{
  const int hc = p.hum ? p.hum_code : 0, dc = p.td ? p.td_code : 0, tc = p.temp ? p.temp_compute : 0;
  const int rh = hum_code(HUM_Q_RH, 0), td = hum_code(HUM_Q_TD, 0);
  const bool ff = p.ff != nullptr;
  // the compile-time combinations: wind speed + RH + theta (BASELINE.json config 2), the same with the
  // dew point, and the two without the wind
  if (tc == 3 && hc == rh && dc == 0 && ff && env().derived_pipe == 0)
    hipLaunchKernelGGL((derived_levels_kernel<CHECK, 1, 3, 1 + HUM_Q_RH, 0, false>), grid, dim3(256), 0, stream, p);
  else if (tc == 3 && hc == rh && dc == 0 && ff)
    hipLaunchKernelGGL((derived_levels_kernel<CHECK, 1, 3, 1 + HUM_Q_RH, 0>), grid, dim3(256), 0, stream, p);
  else if (tc == 3 && hc == rh && dc == td && ff)
    hipLaunchKernelGGL((derived_levels_kernel<CHECK, 1, 3, 1 + HUM_Q_RH, 1 + HUM_Q_TD>), grid, dim3(256), 0, stream, p);
  else if (tc == 3 && hc == rh && dc == 0 && !ff)
    hipLaunchKernelGGL((derived_levels_kernel<CHECK, 0, 3, 1 + HUM_Q_RH, 0>), grid, dim3(256), 0, stream, p);
  else if (tc == 3 && hc == rh && dc == td && !ff)
    hipLaunchKernelGGL((derived_levels_kernel<CHECK, 0, 3, 1 + HUM_Q_RH, 1 + HUM_Q_TD>), grid, dim3(256), 0, stream, p);
  else if (tc == 0 && hc == 0 && dc == 0 && ff)
    hipLaunchKernelGGL((derived_levels_kernel<CHECK, 1, 0, 0, 0>), grid, dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((derived_levels_kernel<CHECK, -1, -1, -1, -1>), grid, dim3(256), 0, stream, p);
}

} // namespace

hipError_t launch_derived_levels(const DerivedParams& prm, hipStream_t stream)
{
  if (prm.n <= 0 || prm.nlev <= 0)
    return hipSuccess;
  if (prm.n % 4 != 0)
    return hipErrorInvalidValue; // callers route ragged sizes through the per-field operators
  const int block = 256;
  const int n4 = prm.n >> 2;
  // ~8 workgroups per CU over the whole launch; every workgroup stays on one level.  A single level
  // (BASELINE.json config 2 itself) is latency-bound: one trip per workgroup there.
  int gx = (n4 + block - 1) / block;
  // Measured (profiles/r02/bench_derived*.txt): the kernel is bound by instruction issue (the reference's
  // arithmetic), not by the table staging; ~32 K workgroups (a handful of trips each, ~25 dispatch rounds: no
  // tail) ran fastest, a chip-filling 2 K grid was 15 % slower because its 1.6 dispatch rounds end in a half-empty one.
  int want = env().derived_blocks > 0 ? env().derived_blocks : 32768;
  int per_level = (want + prm.nlev - 1) / prm.nlev;
  if (per_level < 1)
    per_level = 1;
  if (gx > per_level)
    gx = per_level;
  for (int l0 = 0; l0 < prm.nlev; l0 += 65535) {
    DerivedParams p = prm;
    const int nl = (prm.nlev - l0 > 65535) ? 65535 : (prm.nlev - l0);
    p.nlev = nl;
    const size_t off = (size_t)l0 * (size_t)prm.n;
    p.u = prm.u ? prm.u + off : nullptr;
    p.v = prm.v ? prm.v + off : nullptr;
    p.t = prm.t ? prm.t + off : nullptr;
    p.h = prm.h ? prm.h + off : nullptr;
    p.ff = prm.ff ? prm.ff + off : nullptr;
    p.temp = prm.temp ? prm.temp + off : nullptr;
    p.hum = prm.hum ? prm.hum + off : nullptr;
    p.td = prm.td ? prm.td + off : nullptr;
    p.dd = prm.dd ? prm.dd + off : nullptr;
    p.alevel = prm.alevel ? prm.alevel + l0 : nullptr;
    p.blevel = prm.blevel ? prm.blevel + l0 : nullptr;
    p.wind_all_defined = prm.wind_all_defined ? prm.wind_all_defined + l0 : nullptr;
    p.thermo_all_defined = prm.thermo_all_defined ? prm.thermo_all_defined + l0 : nullptr;
    p.cnt_ff = prm.cnt_ff ? prm.cnt_ff + l0 : nullptr;
    p.cnt_temp = prm.cnt_temp ? prm.cnt_temp + l0 : nullptr;
    p.cnt_hum = prm.cnt_hum ? prm.cnt_hum + l0 : nullptr;
    p.cnt_td = prm.cnt_td ? prm.cnt_td + l0 : nullptr;
    p.cnt_dd = prm.cnt_dd ? prm.cnt_dd + l0 : nullptr;
    if (prm.every_level_all_defined)
      launch_one<false>(p, dim3(gx, nl), stream);
    else
      launch_one<true>(p, dim3(gx, nl), stream);
  }
  return hipGetLastError();
}

} // namespace mifc
