// mifc_fused2.hip -- thermalFrontParameter and plevelqvector in ONE launch.
//
// Both reference operators are a stencil applied to the result of a stencil:
//   thermalFrontParameter (FieldCalculations.cc:2266-2309)
//       absdelt = gradient(tx, compute 3), edges filled;  tfp = f(tx, absdelt)
//   plevelqvector (:505-595)
//       ug = plevelgwind_xcomp(z), vg = plevelgwind_ycomp(z), edges filled;
//       qcomp = f(ug, vg, t)
// Run as separate launches the intermediate fields cost a write and five reads
// of HBM/L2 each.  Here a workgroup spans the whole row width and walks down a
// band of rows; the source rows and the intermediate rows it needs sit in LDS
// row rings, so every field is read from HBM once (plus the band's halo rows)
// and only the result is written:
//
//   iteration r:  top      issue the global loads of source row r+1 and of
//                          map-factor row r (registers)
//                 stage A  intermediate row r-1 from source rows r-2..r -> ring M
//                 stage B  result row r-2 from intermediate rows r-3..r-1
//                          (and source/temperature rows r-3..r-1)
//                 end      the loaded source row lands in ring A; store row r-2
//
// Reference semantics kept (mifc_stencil.hip header): every pass is a flat loop
// over rows 1..ny-2 whose edge-column cells see neighbours wrapped into the
// adjacent row and take part in the count; fillEdges then makes
// final(j,i) = raw(clamp(j,1,ny-2), clamp(i,1,nx-2)) -- for the intermediate
// fields as well, which is why ring M holds FILLED rows and rows 0 / ny-1 alias
// rows 1 / ny-2.
//
// Requirements (fused2_supported): nx % 4 == 0, nx <= 4096, 16-byte aligned
// fields, rings fit the 160 KiB of LDS.  Everything else takes the multi-pass path.
#include <cstdlib>

#include "mifc_device.h"
#include "mifc_kernels.h"

namespace mifc {

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int F2_PAD = 4; // floats in front of / behind every LDS row: keeps rows 16-byte aligned, and x0-1 / x0+4 in bounds

// LDS-only barrier: the register prefetch of the next source row stays in flight
// across it (a __syncthreads() would wait for vmcnt(0) as well)
__device__ __forceinline__ void lds_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ float4 ld4(const float* p)
{
  return *reinterpret_cast<const float4*>(p);
}
__device__ __forceinline__ void st4(float* p, const float (&v)[4])
{
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void unpack(const float4 q, float (&v)[4])
{
  v[0] = q.x;
  v[1] = q.y;
  v[2] = q.z;
  v[3] = q.w;
}
// {west, own four, east} of a row in LDS; the edge threads take the neighbour the
// reference's flat loop sees there: (nx-1, j-1) left of column 0, (0, j+1) right of column nx-1
__device__ __forceinline__ void row6(const float* south, const float* centre, const float* north, int x0, int nx, bool first, bool last, float (&v)[6])
{
  const float4 q = ld4(centre + x0);
  v[0] = first ? south[nx - 1] : centre[x0 - 1];
  v[1] = q.x;
  v[2] = q.y;
  v[3] = q.z;
  v[4] = q.w;
  v[5] = last ? north[0] : centre[x0 + 4];
}

template <int OP, bool CHECK>
__global__ __launch_bounds__(1024) void fused2_kernel(const Fused2Params P, const int band)
{
  constexpr bool TFP = OP == F2_TFP;
  constexpr int RA = TFP ? 5 : 3; // TFP reads its source rows in stage B too (see hazards below)
  extern __shared__ float4 lds4[];
  const int nx = P.nx, ny = P.ny;
  const int S = nx + 2 * F2_PAD;
  float* ringA = reinterpret_cast<float*>(lds4) + F2_PAD; // source rows: tx | z
  float* ringT = ringA + RA * S;                          // Q-vector: temperature rows (4)
  float* mid0 = ringT + (TFP ? 0 : 4 * S);                // |grad tx| | ug, filled (3)
  float* mid1 = mid0 + 3 * S;                             // Q-vector: vg, filled (3)

  const int c = threadIdx.x;
  const int nq = nx >> 2;
  const bool active = c < nq;
  const bool first = c == 0, last = c == nq - 1;
  const int x0 = c * 4;
  const float undef = P.undef;

  // rows 1..ny-2 are split into bands; rows 0 and ny-1 are written with rows 1 and ny-2
  const int jb0 = 1 + (int)blockIdx.x * band;
  const int jb1 = (jb0 + band < ny - 1) ? jb0 + band : ny - 1;
  const int rs = jb0 - 2, re = jb1 + 1;

  const size_t col = (size_t)x0;
  const size_t ccol = active ? col : 0; // lanes beyond the row width load column 0 and use nothing
  if (active && rs >= 0)
    *reinterpret_cast<float4*>(ringA + (rs % RA) * S + x0) = ld4(P.a + (size_t)rs * nx + col);
  unsigned int n1 = 0, n2 = 0, n2c = 0;

  // Map-factor rows live in registers from the iteration that loads them (r) through stage A
  // (r+1) to stage B (r+2): three sets, rotated by unrolling the row loop three times.
  struct RowMaps
  {
    float4 xm, ym, fc;
  };
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  RowMaps m0 = {zero4, zero4, zero4}, m1 = m0, m2 = m0;

  // One iteration.  Global loads are issued at the top and land in LDS at the end, just before
  // the result row is stored: the wait there covers loads only (the previous store is a whole
  // iteration old), and the store gets the next iteration to drain.
  // LDS hazards, with the two barriers per iteration:
  //   ring A, TFP (5): row r+1 written at the end replaces row r-4, last read by stage B of iteration r-1
  //   ring A, Q-vector (3): replaces row r-2, last read by stage A of this iteration (before its second barrier)
  //   ring T (4): row r written at the end replaces row r-4, last read by stage B of iteration r-1
  //   ring M (3): row r-1 written in stage A replaces row r-4, last read by stage B of iteration r-1 (before this iteration's first barrier)
  auto iteration = [&](const int r, RowMaps& m_new /* row r */, const RowMaps& m_a /* row r-1 */, const RowMaps& m_b /* row r-2 */)
                       __attribute__((always_inline)) {
    // Unconditional loads (row and column clamped into the field): a load under a condition
    // would make the compiler merge old and new register contents right here, i.e. wait for it.
    const bool load_a = r < re && r + 1 >= 0 && r + 1 < ny;
    const bool load_row = r < re && r >= 0 && r < ny;
    const size_t row_a = (size_t)(r + 1 < 0 ? 0 : (r + 1 > ny - 1 ? ny - 1 : r + 1)) * nx + ccol;
    const size_t row_m = (size_t)(r < 0 ? 0 : (r > ny - 1 ? ny - 1 : r)) * nx + ccol;
    const float4 pa = ld4(P.a + row_a); // in flight: A(r+1), t(r), maps(r)
    float4 pt = zero4;
    m_new.xm = ld4(P.xmapr + row_m);
    m_new.ym = ld4(P.ymapr + row_m);
    if (!TFP) {
      m_new.fc = ld4(P.fcoriolis + row_m);
      pt = ld4(P.t + row_m);
    }
    lds_barrier();

    // ---- stage A: intermediate row y = r-1
    const int y = r - 1;
    if (active && y >= 1 && y <= ny - 2 && y >= jb0 - 1 && y <= jb1) {
      const float* Sr = ringA + ((y - 1) % RA) * S;
      const float* Cr = ringA + (y % RA) * S;
      const float* Nr = ringA + ((y + 1) % RA) * S;
      float sv[4], nv[4], cv[6], xm[4], ym[4];
      unpack(ld4(Sr + x0), sv);
      unpack(ld4(Nr + x0), nv);
      row6(Sr, Cr, Nr, x0, nx, first, last, cv);
      unpack(m_a.xm, xm);
      unpack(m_a.ym, ym);
      if (TFP) {
        // gradient compute 3, :2037-2046
        const bool counted = CHECK && y >= jb0 && y < jb1; // every row is counted by the band that owns it
        float g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float s = sv[k], n = nv[k], w = cv[k], e = cv[k + 2];
          const bool ok = !CHECK || (is_def(s, undef) && is_def(w, undef) && is_def(e, undef) && is_def(n, undef));
          const float dfdx = (float)(0.5 * (double)xm[k] * (double)(e - w));
          const float dfdy = (float)(0.5 * (double)ym[k] * (double)(n - s));
          g[k] = ok ? absval(dfdx, dfdy) : undef;
          if (counted && !ok)
            ++n1;
        }
        if (first)
          g[0] = g[1];
        if (last)
          g[3] = g[2];
        st4(mid0 + (y % 3) * S + x0, g);
      } else {
        // plevelgwind_xcomp :660-663 (tests only if the caller's flag is not ALL_DEFINED),
        // plevelgwind_ycomp :693-698 (always tests: the x pass hands it NONE_DEFINED, :664)
        float fc[4], ug[4], vg[4];
        unpack(m_a.fc, fc);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float s = sv[k], n = nv[k], w = cv[k], e = cv[k + 2];
          const bool ok = is_def(s, undef) && is_def(w, undef) && is_def(e, undef) && is_def(n, undef);
          const double fd = (double)fc[k], finv = shared_reciprocal(fd);
          const float u = (float)quotient(-0.5 * (double)ym[k] * (double)(n - s) * (double)MIFC_K_G, fd, finv);
          const float v = (float)quotient(0.5 * (double)xm[k] * (double)(e - w) * (double)MIFC_K_G, fd, finv);
          ug[k] = (!CHECK || ok) ? u : undef;
          vg[k] = ok ? v : undef;
        }
        if (first) {
          ug[0] = ug[1];
          vg[0] = vg[1];
        }
        if (last) {
          ug[3] = ug[2];
          vg[3] = vg[2];
        }
        st4(mid0 + (y % 3) * S + x0, ug);
        st4(mid1 + (y % 3) * S + x0, vg);
      }
    }
    lds_barrier();

    // ---- stage B: result row j = r-2
    const int j = r - 2;
    const bool have_row = active && j >= jb0 && j < jb1;
    float o[4] = {undef, undef, undef, undef};
    if (have_row) {
      const int js = (j - 1 < 1) ? 1 : j - 1, jn = (j + 1 > ny - 2) ? ny - 2 : j + 1; // filled intermediate rows 0 / ny-1 are rows 1 / ny-2
      float xm[4], ym[4];
      unpack(m_b.xm, xm);
      unpack(m_b.ym, ym);
      if (TFP) {
        const float* Gs = mid0 + (js % 3) * S;
        const float* Gc = mid0 + (j % 3) * S;
        const float* Gn = mid0 + (jn % 3) * S;
        const float* Ts = ringA + ((j - 1) % RA) * S;
        const float* Tc = ringA + (j % RA) * S;
        const float* Tn = ringA + ((j + 1) % RA) * S;
        float gs[4], gn[4], gc[6], ts[4], tn[4], tc[6];
        unpack(ld4(Gs + x0), gs);
        unpack(ld4(Gn + x0), gn);
        row6(Gs, Gc, Gn, x0, nx, first, last, gc);
        unpack(ld4(Ts + x0), ts);
        unpack(ld4(Tn + x0), tn);
        row6(Ts, Tc, Tn, x0, nx, first, last, tc);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // :2290-2298
          const float g = gc[k + 1];
          const bool def = !CHECK || (is_def(ts[k], undef) && is_def(tc[k], undef) && is_def(tc[k + 2], undef) && is_def(tn[k], undef) &&
                                      is_def(gs[k], undef) && is_def(gc[k], undef) && is_def(g, undef) && is_def(gc[k + 2], undef) && is_def(gn[k], undef));
          const bool ok = def && g != 0;
          const double hx = 0.5 * (double)xm[k], hy = 0.5 * (double)ym[k];
          const float dabsdeltdx = (float)(hx * (double)(gc[k + 2] - gc[k]));
          const float dabsdeltdy = (float)(hy * (double)(gn[k] - gs[k]));
          const double gd = (double)g, ginv = shared_reciprocal(gd);
          const float dtdxa = (float)quotient(hx * (double)(tc[k + 2] - tc[k]), gd, ginv);
          const float dtdya = (float)quotient(hy * (double)(tn[k] - ts[k]), gd, ginv);
          o[k] = ok ? -(dabsdeltdx * dtdxa + dabsdeltdy * dtdya) : undef;
          n2 += ok ? 0u : 1u;
          if (CHECK && !def && g != 0)
            ++n2c;
        }
      } else {
        const float* Us = mid0 + (js % 3) * S;
        const float* Uc = mid0 + (j % 3) * S;
        const float* Un = mid0 + (jn % 3) * S;
        const float* Vs = mid1 + (js % 3) * S;
        const float* Vc = mid1 + (j % 3) * S;
        const float* Vn = mid1 + (jn % 3) * S;
        const float* Ts = ringT + ((j - 1) % 4) * S;
        const float* Tc = ringT + (j % 4) * S;
        const float* Tn = ringT + ((j + 1) % 4) * S;
        float us[4], un[4], uc[6], vs[4], vn[4], vc[6], ts[4], tn[4], tc[6];
        unpack(ld4(Us + x0), us);
        unpack(ld4(Un + x0), un);
        row6(Us, Uc, Un, x0, nx, first, last, uc);
        unpack(ld4(Vs + x0), vs);
        unpack(ld4(Vn + x0), vn);
        row6(Vs, Vc, Vn, x0, nx, first, last, vc);
        unpack(ld4(Ts + x0), ts);
        unpack(ld4(Tn + x0), tn);
        row6(Ts, Tc, Tn, x0, nx, first, last, tc);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // :570-584, "!= undef" only
          const bool ok = us[k] != undef && uc[k] != undef && uc[k + 2] != undef && un[k] != undef && vs[k] != undef && vc[k] != undef &&
                          vc[k + 2] != undef && vn[k] != undef && ts[k] != undef && tc[k] != undef && tc[k + 2] != undef && tn[k] != undef;
          const double hx = 0.5 * (double)xm[k], hy = 0.5 * (double)ym[k];
          const float dtdx = (float)(hx * (double)P.scale * (double)(tc[k + 2] - tc[k]));
          const float dtdy = (float)(hy * (double)P.scale * (double)(tn[k] - ts[k]));
          float q;
          if (OP == F2_QVEC_X) {
            const float dugdx = (float)(hx * (double)(uc[k + 2] - uc[k]));
            const float dvgdx = (float)(hx * (double)(vc[k + 2] - vc[k]));
            q = P.scale2 * (dugdx * dtdx + dvgdx * dtdy);
          } else {
            const float dugdy = (float)(hy * (double)(un[k] - us[k]));
            const float dvgdy = (float)(hy * (double)(vn[k] - vs[k]));
            q = P.scale2 * (dugdy * dtdx + dvgdy * dtdy);
          }
          o[k] = ok ? q : undef;
          n2 += ok ? 0u : 1u;
        }
      }
      // fillEdges on the result: columns, then rows 0 / ny-1
      if (first)
        o[0] = o[1];
      if (last)
        o[3] = o[2];
    }
    // rows that were in flight since the top of the iteration; the explicit vmcnt(0) (all paths,
    // loads only by now) keeps the compiler from waiting again -- behind the store -- at the loop edge
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (active) {
      if (load_a)
        *reinterpret_cast<float4*>(ringA + ((r + 1) % RA) * S + x0) = pa;
      if (!TFP && load_row)
        *reinterpret_cast<float4*>(ringT + (r % 4) * S + x0) = pt;
    }
    if (have_row) {
      // the result is written once and never re-read here: nontemporal
      const v4f q = {o[0], o[1], o[2], o[3]};
      __builtin_nontemporal_store(q, reinterpret_cast<v4f*>(P.out + (size_t)j * nx + col));
      if (j == 1)
        __builtin_nontemporal_store(q, reinterpret_cast<v4f*>(P.out + col));
      if (j == ny - 2)
        __builtin_nontemporal_store(q, reinterpret_cast<v4f*>(P.out + (size_t)(ny - 1) * nx + col));
    }
  };
  for (int r = rs; r <= re; r += 3) {
    iteration(r, m0, m2, m1);
    if (r + 1 > re)
      break;
    iteration(r + 1, m1, m0, m2);
    if (r + 2 > re)
      break;
    iteration(r + 2, m2, m1, m0);
  }
  if (TFP && CHECK) {
    wave_count_add(P.counts + 0, n1);
    wave_count_add(P.counts + 2, n2c);
  }
  wave_count_add(P.counts + 1, n2);
}

// diagnostic: the same quotient through shared_reciprocal()/quotient() and through the compiler's a / b
__global__ void division_check_kernel(const float* a, const float* b, const float* g, float* shared, float* plain, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double num = 0.5 * (double)a[i] * (double)b[i] * (double)MIFC_K_G, den = (double)g[i];
    shared[i] = (float)quotient(num, den, shared_reciprocal(den));
    plain[i] = (float)(num / den);
  }
}

size_t lds_bytes(const Fused2Params& p)
{
  const size_t rows = p.op == F2_TFP ? 5 + 3 : 3 + 4 + 3 + 3;
  return rows * (size_t)(p.nx + 2 * F2_PAD) * sizeof(float);
}

constexpr size_t LDS_PER_CU = 160 * 1024;

template <int OP, bool CHECK>
hipError_t launch(const Fused2Params& p, hipStream_t stream)
{
  const int threads = ((p.nx / 4 + 63) / 64) * 64;
  const size_t lds = lds_bytes(p);
  const int interior = p.ny - 2;
  // enough bands to fill the chip a few times over, tall enough that the
  // 4 (source) + 2 (maps) halo rows a band re-reads stay a small fraction
  size_t per_cu = LDS_PER_CU / lds;
  if (per_cu * threads > 2048)
    per_cu = 2048 / threads;
  if (per_cu < 1)
    per_cu = 1;
  const long want_blocks = 256L * (long)per_cu * 4;
  int band = (int)((interior + want_blocks - 1) / want_blocks);
  if (band < 4) // one 1440x720 level: 4-row bands 16 / 20 us (TFP / Q-vector), 8-row bands 23 / 30 us, multi-pass 30 / 55 us
    band = 4;
  if (band > 64)
    band = 64;
  if (const char* e = std::getenv("MIFC_FUSED2_BAND")) // A/B measurements
    if (std::atoi(e) > 0)
      band = std::atoi(e);
  const int blocks = (interior + band - 1) / band;
  if (lds > 64 * 1024) {
    const hipError_t e =
        hipFuncSetAttribute(reinterpret_cast<const void*>(&fused2_kernel<OP, CHECK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess)
      return e;
  }
  hipLaunchKernelGGL((fused2_kernel<OP, CHECK>), dim3((unsigned)blocks), dim3((unsigned)threads), lds, stream, p, band);
  return hipGetLastError();
}

template <int OP>
hipError_t launch_op(const Fused2Params& p, hipStream_t stream)
{
  return p.check ? launch<OP, true>(p, stream) : launch<OP, false>(p, stream);
}

bool aligned16(const void* p)
{
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

} // namespace

hipError_t launch_division_check(const float* a, const float* b, const float* g, float* shared, float* plain, size_t n, hipStream_t stream)
{
  hipLaunchKernelGGL(division_check_kernel, dim3(1024), dim3(256), 0, stream, a, b, g, shared, plain, n);
  return hipGetLastError();
}

// mifc_fused2_tile.hip: the same operators with one wave per workgroup and column tiles
bool fused2_tile_supported(const Fused2Params& p);
hipError_t launch_fused2_tile(const Fused2Params& p, hipStream_t stream);

static bool tiles_enabled()
{
  const char* e = std::getenv("MIFC_FUSED2_TILE"); // "0": the row-wide kernel of this file (A/B measurements, tests)
  return !(e && e[0] == '0');
}

bool fused2_supported(const Fused2Params& p)
{
  if (p.nx < 4 || (p.nx & 3) || p.ny < 3)
    return false;
  if (tiles_enabled()) {
    if (!fused2_tile_supported(p))
      return false;
  } else if (p.nx > 4096 || lds_bytes(p) > LDS_PER_CU) {
    return false;
  }
  if (!p.a || !p.xmapr || !p.ymapr || !p.out || !p.counts)
    return false;
  if (!aligned16(p.a) || !aligned16(p.xmapr) || !aligned16(p.ymapr) || !aligned16(p.out))
    return false;
  if (p.op != F2_TFP && (!p.t || !p.fcoriolis || !aligned16(p.t) || !aligned16(p.fcoriolis)))
    return false;
  return true;
}

hipError_t launch_fused2(const Fused2Params& p, hipStream_t stream)
{
  if (!fused2_supported(p))
    return hipErrorInvalidValue;
  if (tiles_enabled())
    return launch_fused2_tile(p, stream);
  switch (p.op) {
  case F2_TFP:
    return launch_op<F2_TFP>(p, stream);
  case F2_QVEC_X:
    return launch_op<F2_QVEC_X>(p, stream);
  case F2_QVEC_Y:
    return launch_op<F2_QVEC_Y>(p, stream);
  default:
    return hipErrorInvalidValue;
  }
}

} // namespace mifc
