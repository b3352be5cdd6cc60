// mifc_ewise.hip -- elementwise derived-variable kernels for gfx950.
//
// One pass over the field(s): 16-byte coalesced loads (4 cells per lane),
// point function per cell, 16-byte stores, per-wave undefined count -> one
// atomic per wave.  All of these operators are HBM-bound (8..32 B per cell);
// the only non-trivial ALU work is powf (theta) and the saturation-pressure
// table walk (dew point), both far under the vector-ALU roof at HBM speed.
#include "mifc_device.h"
#include "mifc_kernels.h"

#include <cstdlib>

namespace mifc {

// outputs are written once and never re-read by this library: nontemporal store
__device__ __forceinline__ void store4_stream(float* p, float a, float b, float c, float d)
{
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f t;
  t.x = a;
  t.y = b;
  t.z = c;
  t.w = d;
  __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
}

namespace {

// Point function of the single-field operators.  Returns true when the cell is
// defined (r holds the value); false => cell := undef and is counted.
// `keep` is set for the one case where the reference leaves a defined cell
// unwritten (hleveltemp with compute outside 1..5, FieldCalculations.cc:1080-1090).
template <int OP>
__device__ __forceinline__ bool ewise_point(const EwiseParams& P, const float* tab, const PowTables& PT, int cell, float a, float b, float c, float& r,
                                            bool& keep)
{
  const bool all = P.all_defined != 0;
  const float undef = P.undef;
  keep = false;
  switch (OP) { // compile-time: every operator is its own kernel instantiation
  case EW_VECTORABS: { // :1831-1837
    if (!(all || (is_def(a, undef) && is_def(b, undef))))
      return false;
    r = absval(a, b);
    return true;
  }
  case EW_TEMP_SCALAR: { // pleveltemp compute 1..3, :347-356
    if (!(all || is_def(a, undef)))
      return false;
    r = (P.compute == 1) ? (a * P.pidcp - MIFC_K_T0) : (P.compute == 2) ? (a * P.pidcp) : (a / P.pidcp);
    return true;
  }
  case EW_TEMP:
  case EW_TEMP_PLAIN: {
    float p, pidcp, pi;
    if (P.psrc == PS_SCALAR) { // pleveltemp :347-363 (p hoisted, evaluated on the host)
      if (!(all || is_def(a, undef)))
        return false;
      p = P.p;
      pidcp = P.pidcp;
      pi = P.pi;
    } else {
      if (!(all || (is_def(a, undef) && is_def(c, undef)))) // :1077, :1334
        return false;
      p = (P.psrc == PS_HYBRID) ? (P.alevel + P.blevel * c) : c; // :303
      pidcp = pidcp_of(PT, p);
      pi = pidcp * MIFC_K_CP;
    }
    if (OP == EW_TEMP_PLAIN) { // compute 1..3 only: no saturation table in this instantiation
      r = (P.compute == 1) ? (a * pidcp - MIFC_K_T0) : (P.compute == 2) ? (a * pidcp) : (a / pidcp);
      return true;
    }
    switch (P.compute) {
    case 1:
      r = a * pidcp - MIFC_K_T0;
      return true;
    case 2:
      r = a * pidcp;
      return true;
    case 3:
      r = a / pidcp;
      return true;
    case 4:
      return t_thesat(tab, a, p, pi, r);
    case 5:
      return th_thesat(tab, a, p, pi, r);
    default:
      keep = true;
      return true;
    }
  }
  case EW_HUM:
  case EW_HUM_DIRECT: {
    bool ok = all || (is_def(a, undef) && is_def(b, undef)); // :443, :1187, :1429
    if (P.ptest == PT_NEQ)
      ok = ok && (all || c != undef);
    if (!ok)
      return false;
    float p, tk;
    if (P.psrc == PS_SCALAR) { // plevelhum :434-456
      p = P.p;
      tk = a * P.tconv;
    } else {
      if (P.psrc == PS_HYBRID)
        p = (P.kind == HUM_RH_TD && !P.from_theta) ? 0.0f : (P.alevel + P.blevel * c); // need_p :1182,:1188
      else
        p = c;
      tk = P.from_theta ? a * pidcp_of(PT, p) : a;
    }
    if (OP == EW_HUM_DIRECT) // q <-> RH: no table inverse in this instantiation
      return (P.kind == HUM_Q_RH) ? tk_q_rh(tab, tk, b, p, r) : tk_rh_q(tab, tk, b, p, r);
    switch (P.kind) {
    case HUM_Q_RH:
      return tk_q_rh(tab, tk, b, p, r);
    case HUM_RH_Q:
      return tk_rh_q(tab, tk, b, p, r);
    case HUM_Q_TD:
      return tk_q_td(tab, tk, b, p, P.tdconv, r);
    default:
      return tk_rh_td(tab, tk, b, P.tdconv, r);
    }
  }
  case EW_CVHUM_TD: { // :1765-1782
    if (!(all || (is_def(a, undef) && is_def(b, undef))))
      return false;
    const Ewt e(a - P.tconv);
    if (!e.ok())
      return false;
    const float et = e.value(tab);
    const float rh = clamp_rh((float)(0.01 * (double)b));
    r = e.inverse(tab, rh * et) + P.tdconv;
    return true;
  }
  case EW_CVHUM_RH: { // :1792-1808
    if (!(all || (is_def(a, undef) && is_def(b, undef))))
      return false;
    const Ewt e(a - P.tconv), e2(b - P.tconv);
    if (!(e.ok() && e2.ok()))
      return false;
    const float rh = e2.value(tab) / e.value(tab);
    r = rh * P.unit_scale;
    return true;
  }
  default: { // EW_MOMENTUM_X / _Y :2371-2379, :2407-2415 (a = wind component, b = map ratio, c = coriolis)
    if (!(all || is_def(a, undef)))
      return false;
    float fcor = c;
    if (fcor >= 0.f && fcor < P.fcormin)
      fcor = P.fcormin;
    else if (fcor <= 0.f && fcor > -P.fcormin)
      fcor = -P.fcormin;
    if (OP == EW_MOMENTUM_X)
      r = (float)(cell % P.nx) + a * b / fcor;
    else
      r = (float)(cell / P.nx) - a * b / fcor;
    return true;
  }
  }
}

__host__ __device__ inline bool ewise_needs_ewt(const EwiseParams& P)
{
  return !(P.op == EW_VECTORABS || P.op == EW_MOMENTUM_X || P.op == EW_MOMENTUM_Y || (P.op == EW_TEMP && P.compute >= 1 && P.compute <= 3));
}
__host__ __device__ inline bool ewise_needs_pow(const EwiseParams& P)
{
  return P.psrc != PS_SCALAR && (P.op == EW_TEMP || (P.op == EW_HUM && P.from_theta));
}

template <int OP, bool VEC4>
__global__ __launch_bounds__(256) void ewise_kernel(const EwiseParams P)
{
  __shared__ __attribute__((aligned(8))) float s_ewt[MIFC_EWT_LDS];
  constexpr bool USES_POW = (OP == EW_TEMP || OP == EW_TEMP_PLAIN || OP == EW_HUM || OP == EW_HUM_DIRECT);
  __shared__ double s_pow[USES_POW ? MIFC_KAPPA_LDS : 1];
  // the lookup tables cost a few hundred cycles per workgroup: staged only
  // for the operator variants that read them (wave-uniform conditions)
  constexpr bool TABLE_FREE = (OP == EW_TEMP_SCALAR || OP == EW_VECTORABS || OP == EW_MOMENTUM_X || OP == EW_MOMENTUM_Y);
  const bool with_pow = USES_POW && ewise_needs_pow(P);
  if (!TABLE_FREE && ewise_needs_ewt(P))
    ewt_table_init(s_ewt, !with_pow); // one barrier for both tables
  PowTables PT = {nullptr, nullptr, s_pow, s_pow};
  if (with_pow)
    PT = kappa_tables_init(s_pow);

  const bool use1 = P.in1 != nullptr;
  const bool use2 = P.in2 != nullptr;
  const bool may_keep = (OP == EW_TEMP) && (P.compute < 1 || P.compute > 5); // the _SCALAR / _PLAIN instantiations only see compute 1..3
  unsigned int bad = 0;

  if (VEC4) {
    const int n4 = P.n >> 2;
    const float4* __restrict__ a4p = reinterpret_cast<const float4*>(P.in0);
    const float4* __restrict__ b4p = reinterpret_cast<const float4*>(P.in1);
    const float4* __restrict__ c4p = reinterpret_cast<const float4*>(P.in2);
    float4* o4p = reinterpret_cast<float4*>(P.out);
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += gridDim.x * blockDim.x) {
      const float4 a4 = a4p[q];
      const float4 b4 = use1 ? b4p[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 c4 = use2 ? c4p[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 o4 = may_keep ? o4p[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float av[4] = {a4.x, a4.y, a4.z, a4.w};
      const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
      const float cv[4] = {c4.x, c4.y, c4.z, c4.w};
      float ov[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float r = 0.f;
        bool keep;
        if (ewise_point<OP>(P, s_ewt, PT, q * 4 + k, av[k], bv[k], cv[k], r, keep)) {
          if (!keep)
            ov[k] = r;
        } else {
          ov[k] = P.undef;
          bad += 1;
        }
      }
      store4_stream(P.out + (size_t)q * 4, ov[0], ov[1], ov[2], ov[3]);
    }
    // tail cells (n not a multiple of 4) are handled by a second, scalar launch
  } else {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += gridDim.x * blockDim.x) {
      const float a = P.in0[i];
      const float b = use1 ? P.in1[i] : 0.f;
      const float c = use2 ? P.in2[i] : 0.f;
      float r = 0.f;
      bool keep;
      if (ewise_point<OP>(P, s_ewt, PT, i + P.cell0, a, b, c, r, keep)) {
        if (!keep)
          P.out[i] = r;
      } else {
        P.out[i] = P.undef;
        bad += 1;
      }
    }
  }
  if (P.count && P.partials)
    block_count_store(P.partials + blockIdx.x, bad); // big launch: added up by count_partials_kernel behind it
  else
    block_count_add(P.count ? P.n_undefined : nullptr, bad); // one atomic per workgroup
}

inline bool aligned16(const void* p)
{
  return (reinterpret_cast<size_t>(p) & 15u) == 0;
}

inline int grid_for(int work_items, int block, int max_blocks)
{
  int g = (work_items + block - 1) / block;
  if (g < 1)
    g = 1;
  return g > max_blocks ? max_blocks : g;
}

} // namespace

namespace {
// A few workgroups add up the per-workgroup counts of a big launch and hand their sums to the counter (at most 64 atomics).
__global__ __launch_bounds__(256) void count_partials_kernel(const unsigned int* __restrict__ partials, int n, u64* counter)
{
  unsigned int s = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    s += partials[i];
  block_count_add(counter, s);
}
} // namespace

namespace {
__global__ __launch_bounds__(256) void count_partials_levels_kernel(const unsigned int* __restrict__ partials, int per_level, u64* counters)
{
  const unsigned int* p = partials + (size_t)blockIdx.x * (size_t)per_level;
  unsigned int s = 0;
  for (int i = threadIdx.x; i < per_level; i += 256)
    s += p[i];
  block_count_add(counters + blockIdx.x, s);
}
} // namespace

hipError_t launch_count_partials_levels(const unsigned int* partials, int per_level, int nlev, u64* counters, hipStream_t stream)
{
  for (int l0 = 0; l0 < nlev; l0 += 65535) { // grid limit
    const int nl = nlev - l0 > 65535 ? 65535 : nlev - l0;
    hipLaunchKernelGGL(count_partials_levels_kernel, dim3(nl), dim3(256), 0, stream, partials + (size_t)l0 * per_level, per_level, counters + l0);
  }
  return hipGetLastError();
}

hipError_t launch_count_partials(const unsigned int* partials, int n, u64* counter, hipStream_t stream)
{
  int g = (n + 2047) / 2048;
  g = g < 1 ? 1 : (g > 64 ? 64 : g);
  hipLaunchKernelGGL(count_partials_kernel, dim3(g), dim3(256), 0, stream, partials, n, counter);
  return hipGetLastError();
}

namespace {
// from this many workgroups on a counted launch leaves its counts in partials (a second, tiny launch) instead of one
// atomic per workgroup on one address (~12 ns each, one after the other)
constexpr int kPartialsMinBlocks = 2048;

template <int OP>
hipError_t launch_ewise_op(const EwiseParams& prm, hipStream_t stream)
{
  const bool vec_ok = aligned16(prm.in0) && aligned16(prm.out) && (!prm.in1 || aligned16(prm.in1)) && (!prm.in2 || aligned16(prm.in2)) && prm.n >= 4;
  const int block = 256;
  if (vec_ok) {
    const int n4 = prm.n >> 2;
    // Table-free variants: one float4 per lane, workgroups in address order (the
    // streaming shape that measured fastest on MI355X).  Variants that stage
    // lookup tables per workgroup amortise that over a grid-stride loop
    // (MIFC_EWISE_MAX_BLOCKS overrides the cap, for A/B measurements).
    // Measured on 1440x720x137 (profiles/r02/experiments/ewise_grid_caps.txt), with the tables staged by a plain copy:
    // 65 536 workgroups (two or three trips each) run the table variants 10-17 % faster than 4 096 (hleveltemp
    // 0.343 -> 0.295 ms, hlevelhum 0.475 -> 0.397) and than one trip each; the variants that stage BOTH tables
    // (humidity from potential temperature) are the exception -- their time grows with the number of workgroups
    // (0.51 ms at 4 096, 2.3 ms at 65 536) -- and keep the small grid.
    // Round 3, with the leaner lookups (profiles/r03/experiments/ewise_grid_caps.txt, two A/B passes in one process pair): the
    // variants that stage both tables no longer pay for more workgroups (alevelhum 0.473 ms at 4 096, 0.434 at 32 768, 0.452
    // at 65 536); the dew-point variants (more arithmetic per trip) and the pressure-level ones (two inputs) run 8-9 % faster
    // with 32 768 workgroups than with 65 536 (hlevelhum T,q->Td 0.478 -> 0.438, plevelhum 0.382 -> 0.348); hleveltemp,
    // hlevelhum T,q->RH and cvhum are 4-10 % faster with 65 536 and keep them.
    const bool tables = ewise_needs_ewt(prm) || ewise_needs_pow(prm);
    const bool both = ewise_needs_ewt(prm) && ewise_needs_pow(prm);
    const bool dew_point = (prm.op == EW_HUM) && (prm.kind == HUM_Q_TD || prm.kind == HUM_RH_TD);
    int cap = !tables ? 0x7fffffff : ((both || dew_point || prm.psrc == PS_SCALAR) ? 256 * 128 : 256 * 256);
    if (env().ewise_max_blocks > 0)
      cap = env().ewise_max_blocks;
    const int grid = grid_for(n4, block, cap);
    EwiseParams main = prm;
    const bool by_partials = prm.count && prm.partials && grid >= kPartialsMinBlocks && grid <= prm.partials_cap;
    if (!by_partials)
      main.partials = nullptr;
    hipLaunchKernelGGL((ewise_kernel<OP, true>), dim3(grid), dim3(block), 0, stream, main);
    if (by_partials)
      (void)launch_count_partials(prm.partials, grid, prm.n_undefined, stream);
    const int tail = prm.n - n4 * 4;
    if (tail > 0) {
      EwiseParams t = prm;
      t.partials = nullptr;
      t.n = tail;
      t.in0 = prm.in0 + n4 * 4;
      t.in1 = prm.in1 ? prm.in1 + n4 * 4 : nullptr;
      t.in2 = prm.in2 ? prm.in2 + n4 * 4 : nullptr;
      t.out = prm.out + n4 * 4;
      t.cell0 = n4 * 4;
      hipLaunchKernelGGL((ewise_kernel<OP, false>), dim3(1), dim3(64), 0, stream, t);
    }
  } else {
    EwiseParams q = prm;
    q.partials = nullptr;
    hipLaunchKernelGGL((ewise_kernel<OP, false>), dim3(grid_for(prm.n, block, 256 * 16)), dim3(block), 0, stream, q);
  }
  return hipGetLastError();
}

} // namespace

hipError_t launch_ewise(const EwiseParams& prm, hipStream_t stream)
{
  if (prm.n <= 0)
    return hipSuccess;
  switch (prm.op) {
  case EW_VECTORABS:
    return launch_ewise_op<EW_VECTORABS>(prm, stream);
  case EW_TEMP:
    if (prm.compute >= 1 && prm.compute <= 3)
      return prm.psrc == PS_SCALAR ? launch_ewise_op<EW_TEMP_SCALAR>(prm, stream) : launch_ewise_op<EW_TEMP_PLAIN>(prm, stream);
    return launch_ewise_op<EW_TEMP>(prm, stream);
  case EW_HUM:
    if (prm.kind == HUM_Q_RH || prm.kind == HUM_RH_Q)
      return launch_ewise_op<EW_HUM_DIRECT>(prm, stream);
    return launch_ewise_op<EW_HUM>(prm, stream);
  case EW_CVHUM_TD:
    return launch_ewise_op<EW_CVHUM_TD>(prm, stream);
  case EW_CVHUM_RH:
    return launch_ewise_op<EW_CVHUM_RH>(prm, stream);
  case EW_MOMENTUM_X:
    return launch_ewise_op<EW_MOMENTUM_X>(prm, stream);
  case EW_MOMENTUM_Y:
    return launch_ewise_op<EW_MOMENTUM_Y>(prm, stream);
  default:
    return hipErrorInvalidValue;
  }
}

} // namespace mifc

#ifdef MIFC_MEASUREMENT_BUILD // libmifc_measure.so only (tools/): the bandwidth yardsticks are not part of the product
// ----------------------------------------------------------------------------
// Bandwidth yardstick (diagnostic): streams two input fields into two output
// fields with the same 16-byte-per-lane access shape as the operators and no
// arithmetic.  bench.py / tools report operator bandwidth next to this number,
// measured on the same device in the same process.
namespace mifc {
namespace {
template <int VARIANT>
__global__ __launch_bounds__(256) void stream2_kernel(float* __restrict__ d0, float* __restrict__ d1, const float* __restrict__ s0,
                                                      const float* __restrict__ s1, size_t n4)
{
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f* a = reinterpret_cast<const v4f*>(s0);
  const v4f* b = reinterpret_cast<const v4f*>(s1);
  v4f* x = reinterpret_cast<v4f*>(d0);
  v4f* y = reinterpret_cast<v4f*>(d1);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
    v4f va, vb;
    if (VARIANT == 2) {
      va = __builtin_nontemporal_load(a + q);
      vb = __builtin_nontemporal_load(b + q);
    } else {
      va = a[q];
      vb = b[q];
    }
    if (VARIANT >= 1) {
      __builtin_nontemporal_store(va, x + q);
      __builtin_nontemporal_store(vb, y + q);
    } else {
      x[q] = va;
      y[q] = vb;
    }
  }
}
// Variant 3: the same copy as a LOOP kernel with split roles -- waves 0..3 of a
// workgroup only load (global -> LDS), waves 4..7 only store (LDS -> global).
// A wave that loads never has a store in its vmcnt queue (the counter is shared
// and in-order on gfx9), a wave that stores never waits on memory at all.
// Yardstick for the question "do long-running waves lose bandwidth because
// their load waits are coupled to their own older stores?".
__global__ __launch_bounds__(512) void stream2_split_kernel(float* __restrict__ d0, float* __restrict__ d1, const float* __restrict__ s0,
                                                            const float* __restrict__ s1, size_t n4)
{
  typedef float v4f __attribute__((ext_vector_type(4)));
  __shared__ v4f buf[2][2][256]; // [stage][array][position in the 256-float4 tile]
  const v4f* a = reinterpret_cast<const v4f*>(s0);
  const v4f* b = reinterpret_cast<const v4f*>(s1);
  v4f* x = reinterpret_cast<v4f*>(d0);
  v4f* y = reinterpret_cast<v4f*>(d1);
  const bool loader = threadIdx.x < 256;
  const int t = threadIdx.x & 255;
  const size_t ntiles = (n4 + 255) / 256;
  const size_t step = gridDim.x;
  const v4f zero = {0.f, 0.f, 0.f, 0.f};
  v4f ra0 = zero, rb0 = zero, ra1 = zero, rb1 = zero;
  auto fetch = [&](size_t tile, v4f& ra, v4f& rb) {
    const size_t q = tile * 256 + t;
    if (tile < ntiles && q < n4) {
      ra = __builtin_nontemporal_load(a + q);
      rb = __builtin_nontemporal_load(b + q);
    }
  };
  auto put = [&](size_t tile, int stage) {
    const size_t q = tile * 256 + t;
    if (q < n4) {
      __builtin_nontemporal_store(buf[stage][0][t], x + q);
      __builtin_nontemporal_store(buf[stage][1][t], y + q);
    }
  };
  size_t tile = blockIdx.x;
  if (loader) { // two tiles in flight per loading wave
    fetch(tile, ra0, rb0);
    fetch(tile + step, ra1, rb1);
  }
  for (; tile < ntiles; tile += 2 * step) {
    if (loader) {
      buf[0][0][t] = ra0;
      buf[0][1][t] = rb0;
      fetch(tile + 2 * step, ra0, rb0);
    }
    __syncthreads();
    if (!loader)
      put(tile, 0);
    if (loader) {
      buf[1][0][t] = ra1;
      buf[1][1][t] = rb1;
      fetch(tile + 3 * step, ra1, rb1);
    }
    __syncthreads();
    if (!loader && tile + step < ntiles)
      put(tile + step, 1);
  }
}
// Variant 4: a LOOP copy in which every workgroup streams ONE contiguous chunk of
// each array front to back (384 lanes = one 1440-column row per step, next step
// prefetched): the memory access shape of a kernel whose workgroups span the
// whole row width and walk down a band of rows.
__global__ __launch_bounds__(384) void stream2_band_kernel(float* __restrict__ d0, float* __restrict__ d1, const float* __restrict__ s0,
                                                           const float* __restrict__ s1, size_t n4, size_t chunk)
{
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f* a = reinterpret_cast<const v4f*>(s0);
  const v4f* b = reinterpret_cast<const v4f*>(s1);
  v4f* x = reinterpret_cast<v4f*>(d0);
  v4f* y = reinterpret_cast<v4f*>(d1);
  const size_t beg = (size_t)blockIdx.x * chunk;
  if (beg >= n4)
    return;
  const size_t end = beg + chunk < n4 ? beg + chunk : n4;
  size_t q = beg + threadIdx.x;
  size_t qc = q < end ? q : end - 1; // loads are unconditional (clamped), see mifc_fused2.hip
  v4f pa = a[qc], pb = b[qc];
  for (; q < end; q += 384) {
    const v4f va = pa, vb = pb;
    const size_t qn = q + 384 < end ? q + 384 : end - 1;
    pa = a[qn];
    pb = b[qn];
    __builtin_nontemporal_store(va, x + q);
    __builtin_nontemporal_store(vb, y + q);
  }
}
// Variants 5-7: write-only yardsticks (the sources are not read).  5: one lane per 16 bytes of both
// outputs, linear; 6: the same bytes as 1440-column rows written in tiles of ROWS x 256 columns per
// 256-lane workgroup... (wave w of the workgroup writes row w of the tile), tiles in (row block,
// column segment) order; 7: like 6 but every wave LOOPS over 8 rows of its 256-column segment.
__global__ __launch_bounds__(256) void fill2_linear_kernel(float* __restrict__ d0, float* __restrict__ d1, size_t n4)
{
  typedef float v4f __attribute__((ext_vector_type(4)));
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n4)
    return;
  const v4f c = {1.f, 2.f, 3.f, 4.f};
  __builtin_nontemporal_store(c, reinterpret_cast<v4f*>(d0) + q);
  __builtin_nontemporal_store(c, reinterpret_cast<v4f*>(d1) + q);
}
template <int ROWS_PER_WAVE>
__global__ __launch_bounds__(256) void fill2_tiles_kernel(float* __restrict__ d0, float* __restrict__ d1, size_t nrows)
{
  typedef float v4f __attribute__((ext_vector_type(4)));
  constexpr int NX = 1440, SEG = 6; // 6 segments of 256 columns, the last one partly empty
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t rblock = blockIdx.x / SEG;
  const int seg = (int)(blockIdx.x % SEG);
  const int col = seg * 256 + lane * 4;
  if (col >= NX)
    return;
  const v4f c = {1.f, 2.f, 3.f, 4.f};
#pragma unroll 1
  for (int k = 0; k < ROWS_PER_WAVE; ++k) {
    const size_t row = (rblock * 4 + wave) * ROWS_PER_WAVE + k;
    if (row >= nrows)
      return;
    const size_t o = row * NX + col;
    __builtin_nontemporal_store(c, reinterpret_cast<v4f*>(d0 + o));
    __builtin_nontemporal_store(c, reinterpret_cast<v4f*>(d1 + o));
  }
}
// Variant 8: write-only, a workgroup of 6 waves writes FULL 1440-column rows (wave w = the w-th 256-column
// segment) and loops over 8 consecutive rows: what the operator's stores would look like if the waves of
// a workgroup sat side by side on one level instead of on 8 different levels.
__global__ __launch_bounds__(384) void fill2_fullrow_kernel(float* __restrict__ d0, float* __restrict__ d1, size_t nrows)
{
  typedef float v4f __attribute__((ext_vector_type(4)));
  constexpr int NX = 1440;
  const int col = (int)threadIdx.x * 4;
  if (col >= NX)
    return;
  const v4f c = {1.f, 2.f, 3.f, 4.f};
#pragma unroll 1
  for (int k = 0; k < 8; ++k) {
    const size_t row = (size_t)blockIdx.x * 8 + k;
    if (row >= nrows)
      return;
    const size_t o = row * NX + col;
    __builtin_nontemporal_store(c, reinterpret_cast<v4f*>(d0 + o));
    __builtin_nontemporal_store(c, reinterpret_cast<v4f*>(d1 + o));
  }
}
} // namespace

hipError_t launch_stream2(int variant, int blocks, float* d0, float* d1, const float* s0, const float* s1, size_t n_floats, hipStream_t stream)
{
  const size_t n4 = n_floats / 4;
  if (blocks <= 0) {
    const size_t want = (n4 + 255) / 256;
    blocks = (int)(want > 0x7fffffff ? 0x7fffffff : want);
  }
  switch (variant) {
  case 8: {
    const size_t nrows = n_floats / 1440;
    hipLaunchKernelGGL(fill2_fullrow_kernel, dim3((unsigned)((nrows + 7) / 8)), dim3(384), 0, stream, d0, d1, nrows);
    break;
  }
  case 5:
    hipLaunchKernelGGL(fill2_linear_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, d0, d1, n4);
    break;
  case 6:
  case 7: {
    const size_t nrows = n_floats / 1440;
    const int rpw = variant == 6 ? 1 : 8;
    const size_t rblocks = (nrows + 4 * rpw - 1) / (4 * rpw);
    if (variant == 6)
      hipLaunchKernelGGL(fill2_tiles_kernel<1>, dim3((unsigned)(rblocks * 6)), dim3(256), 0, stream, d0, d1, nrows);
    else
      hipLaunchKernelGGL(fill2_tiles_kernel<8>, dim3((unsigned)(rblocks * 6)), dim3(256), 0, stream, d0, d1, nrows);
    break;
  }
  case 4:
    hipLaunchKernelGGL(stream2_band_kernel, dim3(blocks), dim3(384), 0, stream, d0, d1, s0, s1, n4, (n4 + (size_t)blocks - 1) / (size_t)blocks);
    break;
  case 3:
    hipLaunchKernelGGL(stream2_split_kernel, dim3(blocks), dim3(512), 0, stream, d0, d1, s0, s1, n4);
    break;
  case 0:
    hipLaunchKernelGGL(stream2_kernel<0>, dim3(blocks), dim3(256), 0, stream, d0, d1, s0, s1, n4);
    break;
  case 1:
    hipLaunchKernelGGL(stream2_kernel<1>, dim3(blocks), dim3(256), 0, stream, d0, d1, s0, s1, n4);
    break;
  default:
    hipLaunchKernelGGL(stream2_kernel<2>, dim3(blocks), dim3(256), 0, stream, d0, d1, s0, s1, n4);
    break;
  }
  return hipGetLastError();
}
} // namespace mifc
#endif // MIFC_MEASUREMENT_BUILD
