// mifc_ensemble.hip -- per-cell reductions over ensemble members (SURVEY.md
// 8f-4; FieldCalculations.cc:2671-2860).  One lane owns four consecutive cells
// and walks the members in order, so every member field is read once with
// 16-byte coalesced loads (4 B per cell and member in, 4 B per cell out:
// HBM-bound) and the float accumulation order is the reference's (member 0
// first), which keeps sums, Welford updates and "first one wins" ties identical.
#include "mifc_device.h"
#include "mifc_kernels.h"

namespace mifc {

namespace {

struct Acc // per-cell state of one reduction
{
  float r;   // running result (sum, extreme value, count)
  float tmp; // extremeValue index variants: running extreme
  float m, m2;
  int n;
};

template <int OP>
__device__ __forceinline__ void ens_init(const EnsembleParams& P, Acc& a)
{
  a.r = (OP == ENS_EXTREME) ? P.undef : 0.f;
  a.tmp = P.undef;
  a.m = 0.f;
  a.m2 = 0.f;
  a.n = 0;
}

// returns false when the cell is finished early (sumFields meets an undefined member, :2685-2688)
template <int OP>
__device__ __forceinline__ bool ens_member(const EnsembleParams& P, Acc& a, int j, float f, bool member_all, bool member_none)
{
  const float undef = P.undef;
  if (OP == ENS_SUM) { // :2682-2690
    if (member_all || is_def(f, undef)) {
      a.r += f;
      return true;
    }
    a.r = undef;
    a.n = -1; // counted as undefined
    return false;
  }
  if (OP == ENS_MEAN) { // :2709-2714
    if (member_all || is_def(f, undef)) {
      a.n++;
      a.r += f;
    }
    return true;
  }
  if (OP == ENS_STDDEV) { // :2739-2747, Welford in float
    if (member_all || is_def(f, undef)) {
      const float delta = f - a.m;
      a.n += 1;
      a.m += delta / a.n;
      a.m2 += delta * (f - a.m);
    }
    return true;
  }
  if (OP == ENS_EXTREME) {
    const bool def = member_all || is_def(f, undef);
    if (P.compute == 1 || P.compute == 2) { // :2778-2783
      if (a.r == undef || (def && ((P.compute == 1 && a.r < f) || (P.compute == 2 && a.r > f))))
        a.r = f;
    } else if (P.compute == 3 || P.compute == 4) { // :2792-2798
      if (a.tmp == undef || (def && ((P.compute == 3 && a.tmp < f) || (P.compute == 4 && a.tmp > f)))) {
        a.tmp = f;
        a.r = (float)j;
      }
    }
    return true;
  }
  // ENS_PROBABILITY :2840-2848: members flagged NONE_DEFINED do not take part
  if (!member_none) {
    a.n += 1;
    if ((f != undef) && (!P.check_above || f > P.value_above) && (!P.check_below || f < P.value_below))
      a.r += 1;
  }
  return true;
}

// -> true when the cell's result is undefined (counted)
template <int OP>
__device__ __forceinline__ bool ens_finish(const EnsembleParams& P, Acc& a, float& out)
{
  const float undef = P.undef;
  if (OP == ENS_SUM) {
    out = a.r;
    return a.n < 0;
  }
  if (OP == ENS_MEAN) { // :2715-2720
    if (a.n > 0) {
      out = a.r / a.n;
      return false;
    }
    out = undef;
    return true;
  }
  if (OP == ENS_STDDEV) { // :2748-2753, sqrt is the double function there
    if (a.n > 0) {
      out = (float)sqrt((double)(a.m2 / a.n));
      return false;
    }
    out = undef;
    return true;
  }
  if (OP == ENS_EXTREME) { // :2784, :2799
    out = a.r;
    return (P.compute >= 1 && P.compute <= 4) && a.r == undef;
  }
  // ENS_PROBABILITY :2850-2856
  if (a.n == 0) {
    out = undef;
    return true;
  }
  out = (P.compute < 4) ? (float)((double)a.r / (a.n / 100.0)) : a.r;
  return false;
}

// member j's field and ValuesDefined flag: from the kernel arguments (small ensembles) or the device table
__device__ __forceinline__ const float* ens_field(const EnsembleParams& P, int j)
{
  return P.n_inline ? P.fields_inline[j] : P.fields[j];
}
__device__ __forceinline__ unsigned char ens_flag(const EnsembleParams& P, int j)
{
  if (P.n_inline)
    return P.has_member_flags ? P.flags_inline[j] : (unsigned char)(P.all_defined ? 0 : 2);
  return P.member_flags ? P.member_flags[j] : (unsigned char)(P.all_defined ? 0 : 2);
}

template <int OP, bool VEC4>
__global__ __launch_bounds__(256) void ensemble_kernel(const EnsembleParams P)
{
  unsigned int bad = 0;
  const bool keep_all = (OP == ENS_EXTREME) && !(P.compute >= 1 && P.compute <= 4); // nothing is written for other computes
  if (VEC4) {
    const int n4 = P.n >> 2;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += gridDim.x * blockDim.x) {
      Acc a[4];
      bool live[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        ens_init<OP>(P, a[c]);
        live[c] = true;
      }
      // members in groups of MB: all loads of a group are issued before the first
      // is consumed (the accumulation itself stays in member order)
      constexpr int MB = 8;
      for (int j0 = 0; j0 < P.nfields; j0 += MB) {
        float4 v[MB];
#pragma unroll
        for (int k = 0; k < MB; ++k)
          if (j0 + k < P.nfields)
            v[k] = reinterpret_cast<const float4*>(ens_field(P, j0 + k))[q];
#pragma unroll
        for (int k = 0; k < MB; ++k) {
          if (j0 + k < P.nfields) {
            const int j = j0 + k;
            const unsigned char fl = ens_flag(P, j);
            const bool m_all = fl == 0, m_none = fl == 1;
            const float f[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (live[c])
                live[c] = ens_member<OP>(P, a[c], j, f[c], m_all, m_none);
          }
        }
      }
      if (keep_all)
        continue;
      float o[4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
        bad += ens_finish<OP>(P, a[c], o[c]) ? 1u : 0u;
      typedef float v4f __attribute__((ext_vector_type(4)));
      v4f t;
      t.x = o[0];
      t.y = o[1];
      t.z = o[2];
      t.w = o[3];
      __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(P.out + (size_t)q * 4));
    }
  } else {
    for (int i = P.first + blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += gridDim.x * blockDim.x) {
      Acc a;
      ens_init<OP>(P, a);
      bool live = true;
      for (int j = 0; j < P.nfields && live; ++j) {
        const unsigned char fl = ens_flag(P, j);
        live = ens_member<OP>(P, a, j, ens_field(P, j)[i], fl == 0, fl == 1);
      }
      if (keep_all)
        continue;
      float o;
      bad += ens_finish<OP>(P, a, o) ? 1u : 0u;
      P.out[i] = o;
    }
  }
  block_count_add(P.n_undefined, bad); // one atomic per workgroup
}

template <int OP>
hipError_t launch_ens(const EnsembleParams& prm, hipStream_t stream)
{
  const int block = 256;
  if (prm.vector_ok && prm.n >= 4) {
    const int n4 = prm.n >> 2;
    int g = (n4 + block - 1) / block;
    hipLaunchKernelGGL((ensemble_kernel<OP, true>), dim3(g < 1 ? 1 : g), dim3(block), 0, stream, prm);
    if (prm.n - n4 * 4 > 0) {
      EnsembleParams t = prm;
      t.first = n4 * 4;
      hipLaunchKernelGGL((ensemble_kernel<OP, false>), dim3(1), dim3(64), 0, stream, t);
    }
  } else {
    int g = (prm.n + block - 1) / block;
    hipLaunchKernelGGL((ensemble_kernel<OP, false>), dim3(g < 1 ? 1 : g), dim3(block), 0, stream, prm);
  }
  return hipGetLastError();
}

} // namespace

hipError_t launch_ensemble(const EnsembleParams& prm, hipStream_t stream)
{
  if (prm.n <= 0)
    return hipSuccess;
  switch (prm.op) {
  case ENS_SUM:
    return launch_ens<ENS_SUM>(prm, stream);
  case ENS_MEAN:
    return launch_ens<ENS_MEAN>(prm, stream);
  case ENS_STDDEV:
    return launch_ens<ENS_STDDEV>(prm, stream);
  case ENS_EXTREME:
    return launch_ens<ENS_EXTREME>(prm, stream);
  case ENS_PROBABILITY:
    return launch_ens<ENS_PROBABILITY>(prm, stream);
  default:
    return hipErrorInvalidValue;
  }
}

} // namespace mifc
