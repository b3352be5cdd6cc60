// mifc_shapiro.hip -- second-order Shapiro filter (FieldCalculations.cc:2076-2179).
//
// Four Jacobi-style sweeps (x, y with weight s; x, y again) between the output
// field f1 and a scratch field f2 -- each sweep reads one array and writes the
// other, so every sweep is one embarrassingly parallel launch.  8 B per cell and
// sweep; a 1440x720 field is launch-bound (4 MB).
//
// Reference behaviour kept as it is:
//   * ALL_DEFINED input: weights +0.25 for the first x/y pair, -0.25 for the
//     second; the update contains the literal `2.` and is evaluated in double.
//   * otherwise the per-cell weights (0.25 where the three cells of the stencil
//     are defined in the UNSMOOTHED field, else 0) are computed once (:2141-2145)
//     and used by BOTH pairs -- the second pair smooths again, it does not
//     restore (the `s = -0.25` of :2167 never reaches them); float arithmetic.
//   * the x sweep runs over the flat range, then columns 0 and nx-1 are restored;
//     the y sweep leaves rows 0 and ny-1 as they are.
#include "mifc_device.h"
#include "mifc_kernels.h"

namespace mifc {

namespace {

// blockIdx.y = position in the launch's level list (levels == nullptr: the level itself); a level is n cells
__global__ __launch_bounds__(256) void shapiro_masks_kernel(const float* __restrict__ f, int nx, int n, float undef, unsigned char* __restrict__ m1,
                                                            unsigned char* __restrict__ m2, const int* __restrict__ levels)
{
  const size_t off = (size_t)(levels ? levels[blockIdx.y] : (int)blockIdx.y) * (size_t)n;
  f += off;
  m1 += off;
  m2 += off;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const bool c = is_def(f[i], undef);
    m1[i] = (i >= 1 && i < n - 1 && c && is_def(f[i - 1], undef) && is_def(f[i + 1], undef)) ? 1 : 0;     // :2142
    m2[i] = (i >= nx && i < n - nx && c && is_def(f[i - nx], undef) && is_def(f[i + nx], undef)) ? 1 : 0; // :2145
  }
}

// dst = sweep(src) along x (STEP = 1, edge = first/last column) or y (STEP = nx, edge = first/last row)
template <bool ALL, bool ALONG_X>
__global__ __launch_bounds__(256) void shapiro_sweep_kernel(const float* __restrict__ src, float* __restrict__ dst, const unsigned char* __restrict__ mask,
                                                            int nx, int ny, float s, const int* __restrict__ levels)
{
  const int n = nx * ny;
  const size_t off = (size_t)(levels ? levels[blockIdx.y] : (int)blockIdx.y) * (size_t)n;
  src += off;
  dst += off;
  if (!ALL)
    mask += off;
  const int step = ALONG_X ? 1 : nx;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int row = i / nx;
    const int col = i - row * nx;
    const bool edge = ALONG_X ? (col == 0 || col == nx - 1) : (row == 0 || row == ny - 1);
    const float c = src[i];
    float r = c;
    if (!edge) {
      const float a = src[i - step], b = src[i + step];
      if (ALL) // :2115, :2123
        r = (float)((double)c + (double)s * ((double)(a + b) - 2. * (double)c));
      else // :2152, :2160
        r = c + (mask[i] ? 0.25f : 0.f) * (a + b - 2 * c);
    }
    dst[i] = r;
  }
}

} // namespace

// ---------------------------------------------------------------------------
// The four sweeps in ONE launch, built like mifc_fused2.hip: one wave per workgroup owns a tile of 240
// columns (60 float4 column groups + one halo group on either side) and walks down a band of rows; the
// rows it still needs live in LDS rings of 1-KiB rows, and since nothing is shared between waves there
// is no barrier.  Per iteration r (F = the unsmoothed field, A/B/C/D = after sweep 1/2/3/4):
//   A(r)   from F(r) and its x-neighbours              -> ring A
//   B(r-1) from A(r-2), A(r-1), A(r)                   -> one LDS row (its x-neighbours are needed next)
//   C(r-1) from B(r-1) and its x-neighbours            -> ring C
//   D(r-2) from C(r-3), C(r-2), C(r-1)                 -> stored
// The per-cell weights of the tested variant come from the UNSMOOTHED field (:2141-2145), so F rows
// r-3..r stay in their ring.  The x sweeps keep columns 0 / nx-1, the y sweeps rows 0 / ny-1.
// Reads src, writes dst: the two must be different arrays (halo rows and columns are read by
// neighbouring waves after their owner may have been written).
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int S2_TW = 240, S2_TS = S2_TW + 16, S2_TQ = S2_TW / 4 + 2;

// A row written to LDS is read back by OTHER lanes of the wave (x-neighbours).  The hardware keeps a
// wave's LDS operations in order; this keeps the compiler from moving a neighbour's read above the write
// (for one lane the two addresses never overlap, so it would be free to).
__device__ __forceinline__ void lds_rows_visible()
{
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ float4 s2_ld4(const float* p)
{
  return *reinterpret_cast<const float4*>(p);
}
__device__ __forceinline__ void s2_unpack(const float4 q, float (&v)[4])
{
  v[0] = q.x;
  v[1] = q.y;
  v[2] = q.z;
  v[3] = q.w;
}
__device__ __forceinline__ void s2_row6(const float* row, int p, float (&v)[6])
{
  const float4 q = s2_ld4(row + p);
  v[0] = row[p - 1];
  v[1] = q.x;
  v[2] = q.y;
  v[3] = q.z;
  v[4] = q.w;
  v[5] = row[p + 4];
}
// one cell of a sweep: centre c, the two neighbours a and b along the sweep direction
template <bool ALL>
__device__ __forceinline__ float s2_update(float c, float a, float b, float s, bool weighted)
{
  if (ALL) { // :2115, :2123: (float)(c + s * ((a + b) - 2. * c)) in double.  2. * c is exact, so the inner difference is ONE rounding:
    // fma(-2, c, a + b); s is +-0.25 (a power of two: s * t is exact), so the outer sum is ONE rounding: fma(s, t, c).
    // Two f64 instructions instead of four (the library is built with -ffp-contract=off).
    const double cd = (double)c;
    const double t = __builtin_fma(-2.0, cd, (double)(a + b));
    return (float)__builtin_fma((double)s, t, cd);
  }
  return c + (weighted ? 0.25f : 0.f) * (a + b - 2 * c); // :2152, :2160
}

template <bool ALL>
__global__ __launch_bounds__(64) void shapiro2_tile_kernel(const float* __restrict__ src0, float* __restrict__ dst0, const int nx, const int ny,
                                                           const float undef, const int band, const int ntiles, const long level_stride,
                                                           const int* __restrict__ levels)
{
  // level batches: grid.y walks the levels of this launch (entry k = level levels[k], or k)
  const size_t level_off = (size_t)(levels ? levels[blockIdx.y] : (int)blockIdx.y) * (size_t)level_stride;
  const float* __restrict__ src = src0 + level_off;
  float* __restrict__ dst = dst0 + level_off;
  constexpr int RF = 5; // F rows r-3..r are live in an iteration, row r+1 lands at its end
  __shared__ float4 lds4[(RF + 3 + 1 + 3) * S2_TS / 4];
  float* ringF = reinterpret_cast<float*>(lds4);
  float* ringA = ringF + RF * S2_TS;
  float* rowB = ringA + 3 * S2_TS;
  float* ringC = rowB + S2_TS;

  const int lane = threadIdx.x;
  const int tile = (int)blockIdx.x % ntiles;
  const int bidx = (int)blockIdx.x / ntiles;
  const int xq = tile * S2_TW - 4 + 4 * lane;
  const bool loadable = lane < S2_TQ && xq >= 0 && xq < nx;
  const bool owned = loadable && lane >= 1 && lane <= S2_TW / 4;
  const int p = 4 + 4 * lane;
  const bool first_col = xq == 0, last_col = xq + 4 == nx; // the group holds column 0 / nx-1: the x sweeps keep them

  const int jb0 = bidx * band; // output rows [jb0, jb1)
  const int jb1 = (jb0 + band < ny) ? jb0 + band : ny;
  const int rs = jb0 - 2, re = jb1 + 1;
  const size_t ccol = (size_t)(loadable ? xq : tile * S2_TW);
  auto clamp_row = [&](int r) { return (size_t)(r < 0 ? 0 : (r > ny - 1 ? ny - 1 : r)) * nx + ccol; };
  auto slot = [](int r, int n) { return ((r % n) + n) % n; }; // r may be negative near the top of the field
  if (loadable && rs >= 0)
    *reinterpret_cast<float4*>(ringF + slot(rs, RF) * S2_TS + p) = s2_ld4(src + clamp_row(rs));

  // is_def of the three cells of a stencil in the unsmoothed field
  auto def3 = [&](float a, float c, float b) { return all_def(undef, a, c, b); };

#pragma unroll 1
  for (int r = rs; r <= re; ++r) {
    const bool land = r < re && r + 1 >= 0 && r + 1 < ny;
    const float4 pf = s2_ld4(src + clamp_row(r + 1)); // lands in ring F at the end of the iteration

    // ---- sweep 1 (x, weight s = 0.25): A(r)
    if (loadable && r >= 0 && r < ny) {
      float f[6], a[4];
      s2_row6(ringF + slot(r, RF) * S2_TS, p, f);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        a[k] = s2_update<ALL>(f[k + 1], f[k], f[k + 2], 0.25f, ALL || def3(f[k], f[k + 1], f[k + 2]));
      if (first_col)
        a[0] = f[1];
      if (last_col)
        a[3] = f[4];
      *reinterpret_cast<float4*>(ringA + slot(r, 3) * S2_TS + p) = make_float4(a[0], a[1], a[2], a[3]);
    }
    // ---- sweep 2 (y): B(r-1), then sweep 3 (x, weight -0.25 / the same per-cell weights again): C(r-1)
    const int y = r - 1;
    if (loadable && y >= 0 && y < ny && y >= jb0 - 1 && y <= jb1) {
      float ac[4], b[4];
      s2_unpack(s2_ld4(ringA + slot(y, 3) * S2_TS + p), ac);
      if (y >= 1 && y <= ny - 2) {
        float as[4], an[4], fs[4], fc[4], fn[4];
        s2_unpack(s2_ld4(ringA + slot(y - 1, 3) * S2_TS + p), as);
        s2_unpack(s2_ld4(ringA + slot(y + 1, 3) * S2_TS + p), an);
        if (!ALL) {
          s2_unpack(s2_ld4(ringF + slot(y - 1, RF) * S2_TS + p), fs);
          s2_unpack(s2_ld4(ringF + slot(y, RF) * S2_TS + p), fc);
          s2_unpack(s2_ld4(ringF + slot(y + 1, RF) * S2_TS + p), fn);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          b[k] = s2_update<ALL>(ac[k], as[k], an[k], 0.25f, ALL || def3(fs[k], fc[k], fn[k]));
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          b[k] = ac[k]; // rows 0 and ny-1 pass through (:2125-2128)
      }
      lds_rows_visible(); // the neighbours' reads of the previous B row are issued
      *reinterpret_cast<float4*>(rowB + p) = make_float4(b[0], b[1], b[2], b[3]);
      lds_rows_visible();
      float bb[6], f[6], c[4];
      s2_row6(rowB, p, bb);
      if (!ALL)
        s2_row6(ringF + slot(y, RF) * S2_TS, p, f);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        c[k] = s2_update<ALL>(bb[k + 1], bb[k], bb[k + 2], -0.25f, ALL || def3(f[k], f[k + 1], f[k + 2]));
      if (first_col)
        c[0] = bb[1];
      if (last_col)
        c[3] = bb[4];
      *reinterpret_cast<float4*>(ringC + slot(y, 3) * S2_TS + p) = make_float4(c[0], c[1], c[2], c[3]);
    }
    // ---- sweep 4 (y): D(r-2), owned groups of owned rows
    const int j = r - 2;
    const bool have_row = owned && j >= jb0 && j < jb1;
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    if (have_row) {
      float cc[4];
      s2_unpack(s2_ld4(ringC + slot(j, 3) * S2_TS + p), cc);
      if (j >= 1 && j <= ny - 2) {
        float cs[4], cn[4], fs[4], fc[4], fn[4];
        s2_unpack(s2_ld4(ringC + slot(j - 1, 3) * S2_TS + p), cs);
        s2_unpack(s2_ld4(ringC + slot(j + 1, 3) * S2_TS + p), cn);
        if (!ALL) {
          s2_unpack(s2_ld4(ringF + slot(j - 1, RF) * S2_TS + p), fs);
          s2_unpack(s2_ld4(ringF + slot(j, RF) * S2_TS + p), fc);
          s2_unpack(s2_ld4(ringF + slot(j + 1, RF) * S2_TS + p), fn);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          d[k] = s2_update<ALL>(cc[k], cs[k], cn[k], -0.25f, ALL || def3(fs[k], fc[k], fn[k]));
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          d[k] = cc[k];
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the loaded row; the previous store is an iteration old
    if (loadable && land)
      *reinterpret_cast<float4*>(ringF + slot(r + 1, RF) * S2_TS + p) = pf;
    lds_rows_visible();
    if (have_row) {
      const v4f q = {d[0], d[1], d[2], d[3]};
      __builtin_nontemporal_store(q, reinterpret_cast<v4f*>(dst + (size_t)j * nx + xq));
    }
  }
}

// ---------------------------------------------------------------------------
// The same four sweeps with NOTHING in LDS: a wave owns 240 columns (60 float4 column groups, lanes 0 and 61 hold the
// halo groups; tiles start 960 B apart: whole 64-B pieces) and walks down its band; the x-neighbours of a row come from the adjacent lanes (DPP wave shifts), the
// rows above and below from the wave's own registers of the two previous iterations.  Per iteration r:
//   A(r)   = x sweep of F(r)                                     (F(r) was requested two iterations ago)
//   B(r-1) = y sweep of A(r-2), A(r-1), A(r)    C(r-1) = x sweep of B(r-1)
//   D(r-2) = y sweep of C(r-3), C(r-2), C(r-1)  -> stored
// Why: no LDS instructions, no LDS address arithmetic, no bank conflicts (the unaligned x-neighbour dwords of the ring
// rows), 8 waves per SIMD instead of 13 per CU, two rows in flight per wave instead of one.  The three-row windows rotate by unrolling the row loop three times (a register move of a row
// that is still in flight would wait for it).  The tested variant carries, instead of the unsmoothed rows, their
// is_def bits (4 per lane and row): the x-triple mask of rows r and r-1, the row masks of rows r-3..r.
// RAGGED (round 3): any width, fields and level strides at dword alignment.  Nothing here needs aligned rows but the 16-byte
// accesses themselves: unaligned loads cost nothing, unaligned stores 25 % of their rate (profiles/r03/experiments/
// ragged_probe.txt); the group that holds column nx-1 may hold fewer than four cells -- it is loaded from nx-4 and shifted
// into place (never reading past the row, i.e. past the batch), keeps column nx-1 wherever it sits and stores its cells one
// by one.  Before: four sweep launches and a mask launch over scalar cells, six times the time of this kernel.
template <bool ALL, bool RAGGED = false>
__global__ __launch_bounds__(64, 8) void shapiro2_regs_kernel(const float* __restrict__ src0, float* __restrict__ dst0, const int nx, const int ny,
                                                              const float undef, const int band, const int ntiles, const long level_stride,
                                                              const int* __restrict__ levels)
{
  constexpr int TW = 240;
  const size_t level_off = (size_t)(levels ? levels[blockIdx.y] : (int)blockIdx.y) * (size_t)level_stride;
  const float* __restrict__ src = src0 + level_off;
  float* __restrict__ dst = dst0 + level_off;
  const int lane = threadIdx.x;
  const int tile = (int)blockIdx.x % ntiles;
  const int bidx = (int)blockIdx.x / ntiles;
  const int xq = tile * TW - 4 + 4 * lane;
  const bool infield = xq >= 0 && xq < nx;
  const bool owned = infield && lane >= 1 && lane <= TW / 4;
  const int k_last = nx - 1 - xq;                           // 0 .. 3 in the group that holds column nx-1
  const bool first_col = xq == 0, last_col = RAGGED ? (k_last >= 0 && k_last <= 3) : xq + 4 == nx; // the x sweeps keep columns 0 / nx-1
  const int nvalid = (RAGGED && infield && k_last < 3) ? k_last + 1 : 4; // cells of this group inside the row
  const int xc = (infield && nvalid == 4) ? xq : (xq < 0 ? 0 : nx - 4);   // lanes outside the field load a valid address and use nothing
  const int jb0 = bidx * band;                               // output rows [jb0, jb1)
  const int jb1 = (jb0 + band < ny) ? jb0 + band : ny;
  const int rs = jb0 - 2, re = jb1 + 1;
  struct __attribute__((packed, aligned(4))) V4Any
  {
    v4f v;
  };
  auto row_at = [&](int r) {
    const float* p = src + (size_t)(r < 0 ? 0 : (r > ny - 1 ? ny - 1 : r)) * nx + xc;
    if constexpr (!RAGGED) {
      return *reinterpret_cast<const v4f*>(p);
    } else {
      v4f q = reinterpret_cast<const V4Any*>(p)->v;
      if (infield && nvalid < 4) { // loaded from column nx-4: the group's cells are the last `nvalid` of it
        const int sh = 4 - nvalid;
        v4f t;
        t.x = sh == 1 ? q.y : (sh == 2 ? q.z : q.w);
        t.y = sh == 1 ? q.z : q.w;
        t.z = q.w;
        t.w = q.w;
        q = t;
      }
      return q;
    }
  };
  auto west_of = [](float keep, float x) { // lane i <- lane i-1 (lane 0 keeps `keep`); every lane is active where these run
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, false));
  };
  auto east_of = [](float keep, float x) { // lane i <- lane i+1 (lane 63 keeps `keep`)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, false));
  };
  // x sweep of a row: q = the lane's four cells, m = weights (tested variant: bit k = the x triple of cell k is defined)
  auto xsweep = [&](const v4f q, const float s, const unsigned m) {
    const float v[6] = {west_of(q.x, q.w), q.x, q.y, q.z, q.w, east_of(q.w, q.x)};
    v4f o;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      o[k] = s2_update<ALL>(v[k + 1], v[k], v[k + 2], s, ALL || ((m >> k) & 1u) != 0);
    if (first_col)
      o.x = q.x;
    if constexpr (RAGGED) {
      if (last_col) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k == k_last)
            o[k] = q[k];
      }
    } else {
      if (last_col)
        o.w = q.w;
    }
    return o;
  };
  auto ysweep = [&](const v4f c, const v4f sth, const v4f nth, const float s, const unsigned m) {
    v4f o;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      o[k] = s2_update<ALL>(c[k], sth[k], nth[k], s, ALL || ((m >> k) & 1u) != 0);
    return o;
  };
  const v4f zero = {0.f, 0.f, 0.f, 0.f};
  v4f F0 = row_at(rs), F1 = row_at(rs + 1), F2 = zero;
  v4f A0 = zero, A1 = zero, A2 = zero, C0 = zero, C1 = zero, C2 = zero;
  // tested variant: is_def bits of the unsmoothed rows r, r-1, r-2, r-3 (own cells) and the x-triple masks of rows r, r-1.
  // (Kept as packed bits in a VGPR: as 24 loop-carried flags -- lane masks combined on the scalar unit -- the compiler
  // spilled and the variant ran 0.42 instead of 0.285 ms.)
  unsigned d0 = 0, d1 = 0, d2 = 0, d3 = 0, mx0 = 0, mx1 = 0;

  // one iteration; f_use holds F(r), f_load receives F(r+2); a_new / c_new are the window slots that fall free
  auto step = [&](const int r, const v4f& f_use, v4f& f_load, v4f& a_new, const v4f& a_c, const v4f& a_s, v4f& c_new, const v4f& c_c, const v4f& c_s)
                  __attribute__((always_inline)) {
    f_load = row_at(r + 2);
    const v4f f = f_use;
    if (!ALL) {
      d3 = d2;
      d2 = d1;
      d1 = d0;
      mx1 = mx0;
      const float fw = west_of(f.x, f.w), fe = east_of(f.w, f.x);
      const bool df[6] = {is_def(fw, undef), is_def(f.x, undef), is_def(f.y, undef), is_def(f.z, undef), is_def(f.w, undef), is_def(fe, undef)};
      d0 = (df[1] ? 1u : 0u) | (df[2] ? 2u : 0u) | (df[3] ? 4u : 0u) | (df[4] ? 8u : 0u);
      mx0 = ((df[0] & df[1] & df[2]) ? 1u : 0u) | ((df[1] & df[2] & df[3]) ? 2u : 0u) | ((df[2] & df[3] & df[4]) ? 4u : 0u) |
            ((df[3] & df[4] & df[5]) ? 8u : 0u);
    }
    // sweep 1 (x, s = 0.25): A(r)
    a_new = xsweep(f, 0.25f, mx0);
    // sweep 2 (y): B(r-1); rows 0 and ny-1 pass through (:2125-2128); then sweep 3 (x; -0.25, or the same weights again): C(r-1)
    const int y = r - 1;
    const v4f b = (y >= 1 && y <= ny - 2) ? ysweep(a_c, a_s, a_new, 0.25f, d2 & d1 & d0) : a_c;
    c_new = xsweep(b, -0.25f, mx1);
    // sweep 4 (y): D(r-2)
    const int j = r - 2;
    const v4f d = (j >= 1 && j <= ny - 2) ? ysweep(c_c, c_s, c_new, -0.25f, d3 & d2 & d1) : c_c;
    if (owned && j >= jb0 && j < jb1) {
      float* q = dst + (size_t)j * nx + xq;
      if constexpr (!RAGGED) {
        __builtin_nontemporal_store(d, reinterpret_cast<v4f*>(q));
      } else if (nvalid == 4) {
        V4Any t;
        t.v = d;
        *reinterpret_cast<V4Any*>(q) = t;
      } else {
        q[0] = d.x;
        if (nvalid > 1)
          q[1] = d.y;
        if (nvalid > 2)
          q[2] = d.z;
      }
    }
  };
#pragma unroll 1
  for (int r = rs; r <= re; r += 3) {
    step(r, F0, F2, A0, A2, A1, C0, C2, C1);
    if (r + 1 <= re)
      step(r + 1, F1, F0, A1, A0, A2, C1, C0, C2);
    if (r + 2 <= re)
      step(r + 2, F2, F1, A2, A1, A0, C2, C1, C0);
  }
}

} // namespace

bool shapiro2_fused_supported(int nx, int ny, const float* src, const float* dst)
{
  if (nx < 4 || ny < 3 || src == dst)
    return false;
  if (env().shapiro_regs && nx >= 8)
    return true; // the register form takes any width and alignment (RAGGED)
  return (nx & 3) == 0 && (reinterpret_cast<size_t>(src) & 15u) == 0 && (reinterpret_cast<size_t>(dst) & 15u) == 0;
}

hipError_t launch_shapiro2_fused(int nx, int ny, int all_defined, float undef, const float* src, float* dst, hipStream_t stream)
{
  return launch_shapiro2_fused_levels(nx, ny, all_defined, undef, src, dst, 1, 0, nullptr, stream);
}

hipError_t launch_shapiro2_fused_levels(int nx, int ny, int all_defined, float undef, const float* src, float* dst, int n_launch_levels,
                                        long level_stride, const int* levels, hipStream_t stream)
{
  if (n_launch_levels <= 0)
    return hipSuccess;
  if (n_launch_levels > 65535)
    return hipErrorInvalidValue;
  if (env().shapiro_regs) { // the register-resident form: 8 waves per SIMD, bands for about two rounds of the chip
    const int ntiles = (nx + 239) / 240;
    // three rounds of the chip's 8192 wave slots: 24-row bands for 1440 x 720 x 137 (0.256 ms; 36 rows 0.264, 48 rows 0.265)
    const long waves_wanted = env().fused2_band > 0 ? 0 : 256L * 4 * 8 * 3 - 256;
    const long want_bands = (waves_wanted + (long)ntiles * n_launch_levels - 1) / ((long)ntiles * n_launch_levels);
    int band = env().fused2_band > 0 ? env().fused2_band : (int)((ny + (want_bands > 0 ? want_bands : 1) - 1) / (want_bands > 0 ? want_bands : 1));
    if (band < 4 && env().fused2_band <= 0) {
      // a launch too small for three rounds even with 4-row bands: bands for ONE round of the chip instead (a band of b rows
      // costs b + 4 iterations; 4000 x 4000, one level: 17 000 waves of 8 iterations -> 7 565 of 13)
      const long slots = 256L * 4 * 8 / ((long)ntiles * n_launch_levels);
      band = slots > 0 ? (int)((ny + slots - 1) / slots) : 128;
    }
    if (band < 4)
      band = 4;
    if (band > 128)
      band = 128;
    const int nbands = (ny + band - 1) / band;
    const dim3 grid((unsigned)(nbands * ntiles), (unsigned)n_launch_levels);
    const bool ragged = (nx & 3) != 0 || (reinterpret_cast<size_t>(src) & 15u) != 0 || (reinterpret_cast<size_t>(dst) & 15u) != 0 || (level_stride & 3) != 0;
    if (ragged) {
      if (all_defined)
        hipLaunchKernelGGL((shapiro2_regs_kernel<true, true>), grid, dim3(64), 0, stream, src, dst, nx, ny, undef, band, ntiles, level_stride, levels);
      else
        hipLaunchKernelGGL((shapiro2_regs_kernel<false, true>), grid, dim3(64), 0, stream, src, dst, nx, ny, undef, band, ntiles, level_stride, levels);
      return hipGetLastError();
    }
    if (all_defined)
      hipLaunchKernelGGL((shapiro2_regs_kernel<true>), grid, dim3(64), 0, stream, src, dst, nx, ny, undef, band, ntiles, level_stride, levels);
    else
      hipLaunchKernelGGL((shapiro2_regs_kernel<false>), grid, dim3(64), 0, stream, src, dst, nx, ny, undef, band, ntiles, level_stride, levels);
    return hipGetLastError();
  }
  const int ntiles = (nx + S2_TW - 1) / S2_TW;
  const long want_bands = (256L * 12 * 4 + (long)ntiles * n_launch_levels - 1) / ((long)ntiles * n_launch_levels);
  int band = (int)((ny + want_bands - 1) / want_bands);
  if (band < 4)
    band = 4;
  if (band > 64)
    band = 64;
  const int nbands = (ny + band - 1) / band;
  const dim3 grid((unsigned)(nbands * ntiles), (unsigned)n_launch_levels);
  if (all_defined)
    hipLaunchKernelGGL((shapiro2_tile_kernel<true>), grid, dim3(64), 0, stream, src, dst, nx, ny, undef, band, ntiles, level_stride, levels);
  else
    hipLaunchKernelGGL((shapiro2_tile_kernel<false>), grid, dim3(64), 0, stream, src, dst, nx, ny, undef, band, ntiles, level_stride, levels);
  return hipGetLastError();
}

hipError_t launch_shapiro2(const ShapiroParams& P, hipStream_t stream)
{
  return launch_shapiro2_levels(P, 1, nullptr, stream);
}

// The sweep-by-sweep path over n_launch_levels levels of a batch (P.f1, P.f2 and the masks are batches, a level nx * ny
// apart; `levels` lists the levels of this launch or is null): five launches whatever the number of levels.
hipError_t launch_shapiro2_levels(const ShapiroParams& P, int n_launch_levels, const int* levels, hipStream_t stream)
{
  const int n = P.nx * P.ny;
  if (n <= 0 || n_launch_levels <= 0)
    return hipSuccess;
  int gx = (n + 255) / 256;
  if (gx > 65536)
    gx = 65536;
  for (int l0 = 0; l0 < n_launch_levels; l0 += 65535) { // grid.y limit; an implicit numbering shifts the bases
    const int nl = n_launch_levels - l0 > 65535 ? 65535 : n_launch_levels - l0;
    const dim3 grid(gx, nl);
    const size_t off = levels ? 0 : (size_t)l0 * (size_t)n;
    const int* lv = levels ? levels + l0 : nullptr;
    float* f1 = P.f1 + off;
    float* f2 = P.f2 + off;
    unsigned char* mx = P.mask_x ? P.mask_x + off : nullptr;
    unsigned char* my = P.mask_y ? P.mask_y + off : nullptr;
    if (!P.all_defined)
      hipLaunchKernelGGL(shapiro_masks_kernel, grid, dim3(256), 0, stream, f1, P.nx, n, P.undef, mx, my, lv);
    float s = 0.25f;
    for (int pass = 0; pass < 2; ++pass) {
      if (P.all_defined) {
        hipLaunchKernelGGL((shapiro_sweep_kernel<true, true>), grid, dim3(256), 0, stream, f1, f2, nullptr, P.nx, P.ny, s, lv);
        hipLaunchKernelGGL((shapiro_sweep_kernel<true, false>), grid, dim3(256), 0, stream, f2, f1, nullptr, P.nx, P.ny, s, lv);
      } else {
        hipLaunchKernelGGL((shapiro_sweep_kernel<false, true>), grid, dim3(256), 0, stream, f1, f2, mx, P.nx, P.ny, s, lv);
        hipLaunchKernelGGL((shapiro_sweep_kernel<false, false>), grid, dim3(256), 0, stream, f2, f1, my, P.nx, P.ny, s, lv);
      }
      s = -0.25f;
    }
  }
  return hipGetLastError();
}

} // namespace mifc
