// mifc_fused2.hip -- thermalFrontParameter and plevelqvector in ONE launch.
//
// Both reference operators are a stencil applied to the result of a stencil:
//   thermalFrontParameter (FieldCalculations.cc:2266-2309)
//       absdelt = gradient(tx, compute 3), edges filled;  tfp = f(tx, absdelt)
//   plevelqvector (:505-595)
//       ug = plevelgwind_xcomp(z), vg = plevelgwind_ycomp(z), edges filled;
//       qcomp = f(ug, vg, t)
// Run as separate launches the intermediate fields cost a write and five reads
// of HBM/L2 each.  Here a WAVE (one per workgroup) owns a tile of 240 columns
// and walks down a band of rows; the source rows and the edge-filled
// intermediate rows it needs sit in LDS row rings (1 KiB per row), so every
// field is read from HBM once (plus halo rows / columns) and only the result is
// written:
//
//   iteration r:  top      issue the global loads of source row r+1, of
//                          map-factor row r and (Q-vector) temperature row r
//                 stage A  intermediate row r-1 from source rows r-2..r -> ring M
//                 stage B  result row r-2 from intermediate rows r-3..r-1
//                          (and source/temperature rows r-3..r-1)
//                 end      the loaded source row lands in ring A; store row r-2
//
// A wave owns 60 float4 column groups and loads one more on either side, so the
// x-neighbours of its intermediate rows are its own: nothing is shared between
// waves and there is no barrier at all -- the LDS queue of a wave is in order.
// (A first version had one workgroup span the whole row width; it spent half
// its time or more at its two barriers per row: profiles/r01/valu_by_kernel.txt.)
//
// Reference semantics kept (mifc_stencil.hip header): every pass is a flat loop
// over rows 1..ny-2 whose edge-column cells see neighbours wrapped into the
// adjacent row and take part in the count; fillEdges then makes
// final(j,i) = raw(clamp(j,1,ny-2), clamp(i,1,nx-2)) -- for the intermediate
// fields as well, which is why ring M holds FILLED rows and rows 0 / ny-1 alias
// rows 1 / ny-2.  What a tile cannot see is the far edge of the field: the
// VALUES of columns 0 and nx-1 are fill copies of columns 1 / nx-2, which the
// edge tiles own; their share of the undefined COUNT is taken by a few extra
// workgroups (edge_count_cells below), two cells per row straight from global
// memory.  (TFP with an ALL_DEFINED input needs no wrapped neighbour for the
// count -- only |grad| != 0 -- and has no such workgroups.)
//
// (A register-resident form like shapiro2_regs_kernel -- x-neighbours by DPP, three-row windows in registers, the first
// stage's differences carried to the second, no LDS -- was built for thermalFrontParameter at the end of round 2:
// bit-identical, 22 % fewer instructions per row, but 126 VGPRs = 4 waves per SIMD instead of 5, and no faster
// (0.472-0.482 against 0.465 ms, tested 0.60 against 0.58: profiles/r02/experiments/tfp_registers.txt).  Not kept.)
//
// Requirements (fused2_supported): nx >= 8.  Widths that are not a multiple of 4 and fields off the 16-byte grid take the
// RAGGED variant of the kernel (round 3); smaller fields the multi-pass path.
#include <cstdlib>

#include "mifc_device.h"
#include "mifc_kernels.h"

namespace mifc {

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int TW = 240;       // cells a tile owns per row
constexpr int TS = TW + 16;   // floats per LDS row: 4 pad | halo quad | 60 owned quads | halo quad | 4 pad
constexpr int TQ = TW / 4 + 2; // column groups a wave holds (62 of its 64 lanes)

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
// lane i <- lane i-1 / lane i+1 (wave shifts; the edge lane keeps `keep_if_none`).  Every lane of the
// wave must be active where these are called: a disabled source lane leaves the destination unchanged.
__device__ __forceinline__ float from_lower_lane(float keep_if_none, float x)
{
  return __builtin_bit_cast(float,
                            __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep_if_none), __builtin_bit_cast(int, x), 0x138 /*wave_shr:1*/, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_upper_lane(float keep_if_none, float x)
{
  return __builtin_bit_cast(float,
                            __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep_if_none), __builtin_bit_cast(int, x), 0x130 /*wave_shl:1*/, 0xf, 0xf, false));
}
// A row written to LDS is read back by OTHER lanes of the wave (x-neighbours).  The hardware keeps a
// wave's LDS operations in order; this keeps the compiler from moving a neighbour's read above the write
// (for one lane the two addresses never overlap, so it would be free to).
__device__ __forceinline__ void lds_rows_visible()
{
  asm volatile("" ::: "memory");
}
// {west, own four, east} of a ring row: the lane's own column group from LDS, the two x-neighbours from the ADJACENT LANES'
// groups (wave shifts).  Round 2 read them from LDS as scalars (row[p - 1], row[p + 4]): 64 lanes 16 bytes apart hit each
// bank four times -- 58 % of this kernel's LDS cycles were bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE,
// profiles/r02/experiments/sq_counters_two_stage_kernels_end.txt).  Every lane whose group a NEIGHBOUR needs must be active
// here (a disabled source lane leaves the destination as it is): the callers run this for all lanes that hold a group,
// the halo groups included, and select the lanes that compute afterwards.  The outermost groups keep their own edge value
// as the missing neighbour: it only reaches cells whose values are fill copies and whose counts the edge kernel takes.
__device__ __forceinline__ void row6(const float* row, int p, float (&v)[6])
{
  const float4 q = ld4(row + p);
  v[0] = from_lower_lane(q.x, q.w);
  v[1] = q.x;
  v[2] = q.y;
  v[3] = q.z;
  v[4] = q.w;
  v[5] = from_upper_lane(q.w, q.x);
}

// ---- undefined values as NaN inside the tiles of the tested TFP variant
// is_def(x) = "x is not NaN and x != undef" costs two compares per value and every value is tested by up to
// eight cells.  The tested TFP tiles therefore replace undefined values by NaN once, when a row enters LDS
// (the source rows as they land, |grad| rows as they are stored); after that "is one of these undefined"
// is "is one of these NaN", one v_cmp_u_f32 per PAIR of values.  A cell that passes its tests computes with
// exactly the values the reference computes with; a cell that fails becomes undef either way.
__device__ __forceinline__ float canon_nan(float x, float undef)
{
  return is_def(x, undef) ? x : __builtin_nanf("");
}
__device__ __forceinline__ float4 canon_nan4(float4 q, float undef)
{
  return make_float4(canon_nan(q.x, undef), canon_nan(q.y, undef), canon_nan(q.z, undef), canon_nan(q.w, undef));
}
__device__ __forceinline__ bool either_nan(float a, float b)
{
  return __builtin_isunordered(a, b);
}
// "any of these" without short-circuits (straight-line code, see all_def() in mifc_device.h)
template <typename... B>
__device__ __forceinline__ bool any_of(B... b)
{
  return ((b ? 1 : 0) | ...) != 0;
}

// ---- the point formulas, shared with the edge-count kernel
// gradient compute 3, FieldCalculations.cc:2037-2046
// CANON: the inputs carry undefined values as NaN (see above) and so does the result
template <bool CHECK, bool CANON = false>
__device__ __forceinline__ float tfp_absdelt(float s, float w, float e, float n, float xm, float ym, float undef, bool& ok)
{
  if (CANON)
    ok = !any_of(either_nan(s, w), either_nan(e, n));
  else
    ok = !CHECK || all_def(undef, s, w, e, n);
  const float dfdx = half_prod(xm, e - w);
  const float dfdy = half_prod(ym, n - s);
  const float g = absval(dfdx, dfdy);
  if (CANON) // a computed NaN, or a computed value that happens to equal undef, is undefined to the next pass (:2290)
    return (ok && is_def(g, undef)) ? g : __builtin_nanf("");
  return ok ? g : undef;
}
// plevelgwind_xcomp :660-663 (tests only if the caller's flag is not ALL_DEFINED),
// plevelgwind_ycomp :693-698 (always tests: the x pass hands it NONE_DEFINED, :664)
template <bool CHECK>
__device__ __forceinline__ void qvec_gwind(float s, float w, float e, float n, float xm, float ym, float fc, float undef, float& ug, float& vg)
{
  const bool ok = all_def(undef, s, w, e, n);
  const double fd = (double)fc, finv = shared_reciprocal(fd);
  const float u = (float)quotient(-0.5 * (double)ym * (double)(n - s) * (double)MIFC_K_G, fd, finv);
  const float v = (float)quotient(0.5 * (double)xm * (double)(e - w) * (double)MIFC_K_G, fd, finv);
  ug = (!CHECK || ok) ? u : undef;
  vg = ok ? v : undef;
}
// thermalFrontParameter :2290-2298
template <bool CHECK, bool CANON = false>
__device__ __forceinline__ float tfp_point(float ts, float tw, float te, float tn, float gs, float gw, float g, float ge, float gn, float xm, float ym,
                                           float undef, bool& ok, bool& rejected_by_test_only)
{
  bool def;
  if (CANON)
    def = !any_of(either_nan(ts, tw), either_nan(te, tn), either_nan(gs, gw), either_nan(ge, gn), g != g);
  else
    def = !CHECK || all_def(undef, ts, tw, te, tn, gs, gw, g, ge, gn);
  ok = def & (g != 0);
  rejected_by_test_only = CHECK && (!def & (g != 0));
  const double hx = 0.5 * (double)xm, hy = 0.5 * (double)ym;
  const float dabsdeltdx = half_prod(xm, ge - gw);
  const float dabsdeltdy = half_prod(ym, gn - gs);
  const double gd = (double)g, ginv = shared_reciprocal(gd);
  const float dtdxa = (float)quotient(hx * (double)(te - tw), gd, ginv);
  const float dtdya = (float)quotient(hy * (double)(tn - ts), gd, ginv);
  return pick(ok, -(dabsdeltdx * dtdxa + dabsdeltdy * dtdya), undef); // unconditional arithmetic + select, no branch per cell
}
// plevelqvector :570-584, "!= undef" only
template <int OP>
__device__ __forceinline__ float qvec_point(float us, float uw, float ue, float un, float vs, float vw, float ve, float vn, float ts, float tw, float te,
                                            float tn, float xm, float ym, float scale, float scale2, float undef, bool& ok)
{
  ok = (us != undef) & (uw != undef) & (ue != undef) & (un != undef) & (vs != undef) & (vw != undef) & (ve != undef) & (vn != undef) & (ts != undef) &
       (tw != undef) & (te != undef) & (tn != undef);
  const double hx = 0.5 * (double)xm, hy = 0.5 * (double)ym;
  // scale == 1 (compute 1, 3: temperature as it is): the product has two float-born factors again
  const float dtdx = scale == 1.0f ? half_prod(xm, te - tw) : (float)(hx * (double)scale * (double)(te - tw));
  const float dtdy = scale == 1.0f ? half_prod(ym, tn - ts) : (float)(hy * (double)scale * (double)(tn - ts));
  float q;
  if (OP == F2_QVEC_X) {
    const float dugdx = half_prod(xm, ue - uw);
    const float dvgdx = half_prod(xm, ve - vw);
    q = scale2 * (dugdx * dtdx + dvgdx * dtdy);
  } else {
    const float dugdy = half_prod(ym, un - us);
    const float dvgdy = half_prod(ym, vn - vs);
    q = scale2 * (dugdy * dtdx + dvgdy * dtdy);
  }
  return ok ? q : undef;
}

// |grad tx| after its fillEdges: the raw value at the clamped position (always an interior cell), from
// global memory.  Out of line on purpose: five of these inlined into one edge cell made the compiler
// spill a thousand scalar registers; the edge cells are 2 per row, a call costs nothing that matters.
template <bool CHECK>
__device__ __attribute__((noinline)) float absdelt_filled_at(const Fused2Params& P, int x, int y)
{
  const int nx = P.nx;
  x = x < 1 ? 1 : (x > nx - 2 ? nx - 2 : x);
  y = y < 1 ? 1 : (y > P.ny - 2 ? P.ny - 2 : y);
  const size_t c = (size_t)y * nx + x;
  bool ok;
  return tfp_absdelt<CHECK>(P.a[c - nx], P.a[c - 1], P.a[c + 1], P.a[c + nx], P.xmapr[c], P.ymapr[c], P.undef, ok);
}

// The cells of columns 0 and nx-1, rows 1..ny-2: only their contribution to the counts (their values
// are fill copies).  Run by a few extra workgroups at the end of the grid; lane = (row, side);
// everything comes from global memory, with the neighbours the flat loop of the reference sees:
// west of column 0 is (nx-1, j-1), east of column nx-1 is (0, j+1).
template <int OP, bool CHECK>
__device__ __forceinline__ void edge_count_cells(const Fused2Params& P, const int first, const int stride)
{
  constexpr bool TFP = OP == F2_TFP;
  const int nx = P.nx, ny = P.ny;
  const float undef = P.undef;
  auto at = [&](const float* f, int x, int y) { return f[(size_t)y * nx + x]; };
  auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
  // intermediate fields after their fillEdges: the raw value at the clamped position (always an interior cell)
  auto absdelt_filled = [&](int x, int y) { return absdelt_filled_at<CHECK>(P, x, y); };
  auto gwind_filled = [&](int x, int y, float& ug, float& vg) {
    x = clampi(x, 1, nx - 2);
    y = clampi(y, 1, ny - 2);
    qvec_gwind<CHECK>(at(P.a, x, y - 1), at(P.a, x - 1, y), at(P.a, x + 1, y), at(P.a, x, y + 1), at(P.xmapr, x, y), at(P.ymapr, x, y),
                      at(P.fcoriolis, x, y), undef, ug, vg);
  };
  unsigned int n1 = 0, n2 = 0, n2c = 0;
  const int cells = 2 * (ny - 2);
  for (int i = first; i < cells; i += stride) {
    const int j = 1 + (i >> 1);
    const bool left = (i & 1) == 0;
    const int x = left ? 0 : nx - 1;
    // flat-index neighbours
    const int wx = left ? nx - 1 : nx - 2, wy = left ? j - 1 : j;
    const int ex = left ? 1 : 0, ey = left ? j : j + 1;
    if (TFP) {
      const float ts = at(P.a, x, j - 1), tw = at(P.a, wx, wy), te = at(P.a, ex, ey), tn = at(P.a, x, j + 1);
      if (CHECK && !(is_def(ts, undef) && is_def(tw, undef) && is_def(te, undef) && is_def(tn, undef)))
        ++n1; // gradient compute 3 :2039
      bool ok, by_test;
      (void)tfp_point<CHECK>(ts, tw, te, tn, absdelt_filled(x, j - 1), absdelt_filled(wx, wy), absdelt_filled(x, j), absdelt_filled(ex, ey),
                             absdelt_filled(x, j + 1), at(P.xmapr, x, j), at(P.ymapr, x, j), undef, ok, by_test);
      n2 += ok ? 0u : 1u;
      if (by_test)
        ++n2c;
    } else {
      float us, uw, ue, un, vs, vw, ve, vn;
      gwind_filled(x, j - 1, us, vs);
      gwind_filled(wx, wy, uw, vw);
      gwind_filled(ex, ey, ue, ve);
      gwind_filled(x, j + 1, un, vn);
      bool ok;
      (void)qvec_point<OP>(us, uw, ue, un, vs, vw, ve, vn, at(P.t, x, j - 1), at(P.t, wx, wy), at(P.t, ex, ey), at(P.t, x, j + 1), at(P.xmapr, x, j),
                           at(P.ymapr, x, j), P.scale, P.scale2, undef, ok);
      n2 += ok ? 0u : 1u;
    }
  }
  if (TFP && CHECK) {
    wave_count_add(P.counts + 0, n1);
    wave_count_add(P.counts + 2, n2c);
  }
  wave_count_add(P.counts + 1, n2);
}

// Ring depths.  One wave reads and writes its rings in program order, so a row may be replaced as soon
// as its last read has been ISSUED: source rows r-2..r+1 are live after iteration r for TFP (its last
// stage reads rows r-3..r-1 of the source again), r-1..r+1 for the Q-vector; intermediate rows r-3..r-1.
// RAGGED (round 3): any width, fields and level strides at dword alignment.  The tile's groups still start at multiples of four
// COLUMNS; what changes is that a row's address is no longer a multiple of 16 bytes (unaligned 16-byte loads: free; unaligned
// stores: 75 % of the store rate, profiles/r03/experiments/ragged_probe.txt) and that the group holding column nx-1 may hold
// fewer than four cells: it is loaded from column nx-4 and shifted into place (never reading past the row), its fill copy of
// column nx-2 may come from the lane below, its cells beyond the row take part in nothing and its store is 1-3 dwords.
template <int OP, bool CHECK, bool RAGGED = false>
__global__ __launch_bounds__(64, (OP == F2_TFP && !CHECK) ? 5 : 4) void fused2_tile_kernel(const Fused2Params P0, const int band, const int ntiles, const int n_main)
{
  constexpr bool TFP = OP == F2_TFP;
  // level batches: this workgroup's level (wave-uniform) selects the fields, the counters and the scalars
  Fused2Params P = P0;
  if (P0.n_launch_levels > 0) {
    const int lev = P0.levels ? P0.levels[blockIdx.y] : (int)blockIdx.y;
    const size_t off = (size_t)lev * (size_t)P0.level_stride;
    P.a = P0.a + off;
    P.t = P0.t ? P0.t + off : nullptr;
    P.out = P0.out + off;
    P.counts = P0.counts + 3 * (size_t)lev;
    if (P0.scale_lev) {
      P.scale = P0.scale_lev[lev];
      P.scale2 = P0.scale2_lev[lev];
    }
  }
  if ((!TFP || CHECK) && (int)blockIdx.x >= n_main) { // the workgroups behind the tiles count the edge-column cells
    edge_count_cells<OP, CHECK>(P, ((int)blockIdx.x - n_main) * 64 + (int)threadIdx.x, ((int)gridDim.x - n_main) * 64);
    return;
  }
  constexpr bool CANON = TFP && CHECK; // undefined values travel as NaN inside the tiles
  constexpr int RA = TFP ? 4 : 3;
  constexpr int ROWS = TFP ? RA + 3 : RA + 3 + 3;
  __shared__ float4 lds4[ROWS * TS / 4];
  float* ringA = reinterpret_cast<float*>(lds4); // source rows: tx | z
  float* mid0 = ringA + RA * TS;                 // |grad tx| | ug, edge-filled (3)
  float* mid1 = mid0 + 3 * TS;                   // Q-vector: vg, edge-filled (3)
  // The Q-vector's temperature rows stay in registers (three rows of the lane's own column group; the
  // x-neighbours of the middle one come from the adjacent lanes): 9 instead of 12 KiB of LDS per wave
  // is the difference between 13 and 16 waves on a CU.

  const int nx = P.nx, ny = P.ny;
  const int lane = threadIdx.x;
  const int tile = (int)blockIdx.x % ntiles;
  const int bidx = (int)blockIdx.x / ntiles;
  const int xq = tile * TW - 4 + 4 * lane; // first column of this lane's group; lanes 0 and 61 hold the halo groups
  const bool loadable = lane < TQ && xq >= 0 && xq < nx;
  const bool owned = loadable && lane >= 1 && lane <= TW / 4;
  const int p = 4 + 4 * lane;                 // position of the group in a ring row
  const int k_last = nx - 1 - xq;             // 0 .. 3 in the group that holds column nx-1
  const bool fill_w = xq == 0, fill_e = RAGGED ? (k_last >= 0 && k_last <= 3) : xq + 4 == nx; // the group holds column 0 / column nx-1 of the field
  const int nvalid = (RAGGED && loadable && k_last < 3) ? k_last + 1 : 4; // cells of this group inside the row
  const int k_e = RAGGED ? k_last : 3;        // where column nx-1 sits in the group that holds it
  const float undef = P.undef;
  struct __attribute__((packed, aligned(4))) V4Any
  {
    v4f v;
  };
  // a group of a row from global memory: aligned 16 bytes, or (RAGGED) 16 bytes at any dword; the partial group at the end of a
  // row comes from column nx-4, its cells shifted to the front
  auto ldg = [&](const float* q) __attribute__((always_inline)) {
    if constexpr (!RAGGED) {
      return ld4(q);
    } else {
      const v4f u = reinterpret_cast<const V4Any*>(q)->v;
      float4 v = make_float4(u.x, u.y, u.z, u.w);
      if (nvalid < 4) {
        const int sh = 4 - nvalid;
        float4 t;
        t.x = sh == 1 ? v.y : (sh == 2 ? v.z : v.w);
        t.y = sh == 1 ? v.z : v.w;
        t.z = v.w;
        t.w = v.w;
        v = t;
      }
      return v;
    }
  };
  // fillEdges of a row held as four cells per lane: column 0 <- column 1, column nx-1 <- column nx-2 (RAGGED: wherever it sits;
  // with one cell in the group the value comes from the lane below, which every caller has active)
  auto fill_cols = [&](float (&z)[4]) __attribute__((always_inline)) {
    if constexpr (RAGGED) {
      const float below = from_lower_lane(z[3], z[3]);
      if (fill_w)
        z[0] = z[1];
      if (fill_e) {
        if (k_last == 0)
          z[0] = below;
        else if (k_last == 1)
          z[1] = z[0];
        else if (k_last == 2)
          z[2] = z[1];
        else
          z[3] = z[2];
      }
    } else {
      if (fill_w)
        z[0] = z[1];
      if (fill_e)
        z[3] = z[2];
    }
  };

  const int jb0 = 1 + bidx * band;
  const int jb1 = (jb0 + band < ny - 1) ? jb0 + band : ny - 1;
  const int rs = jb0 - 2, re = jb1 + 1;

  // other lanes load a valid address and use nothing (RAGGED: the last four columns of the row, like the partial group)
  const size_t ccol = RAGGED ? (size_t)((loadable && nvalid == 4) ? xq : nx - 4) : (size_t)(loadable ? xq : tile * TW);
  if (loadable && rs >= 0)
    *reinterpret_cast<float4*>(ringA + (rs % RA) * TS + p) = CANON ? canon_nan4(ldg(P.a + (size_t)rs * nx + ccol), undef) : ldg(P.a + (size_t)rs * nx + ccol);
  unsigned int n1 = 0, n2 = 0, n2c = 0;

  struct RowMaps // what a lane keeps of a row beyond the iteration that loads it: map factors, and the Q-vector's temperature
  {
    float4 xm, ym, fc, t;
  };
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  RowMaps m0 = {zero4, zero4, zero4, zero4}, m1 = m0, m2 = m0;
  float4 t_south = zero4; // temperature row r-3

  // One iteration (pipeline: file header; ring depths: above).  Global loads are issued at the top and
  // land in LDS at the end, just before the result row is stored: the wait there covers loads only (the
  // previous store is a whole iteration old), and the store gets the next iteration to drain.  The
  // loads are unconditional (row and column clamped into the field): a load under a condition would
  // make the compiler merge old and new register contents right there, i.e. wait for it.  Map rows are
  // live for three iterations (load, stage A, stage B): three register sets, rotated by unrolling the
  // row loop three times instead of by moves.
  auto iteration = [&](const int r, RowMaps& m_new /* row r */, const RowMaps& m_a /* row r-1 */, const RowMaps& m_b /* row r-2 */)
                       __attribute__((always_inline)) {
    const bool load_a = r < re && r + 1 >= 0 && r + 1 < ny;
    const size_t row_a = (size_t)(r + 1 < 0 ? 0 : (r + 1 > ny - 1 ? ny - 1 : r + 1)) * nx + ccol;
    const size_t row_m = (size_t)(r < 0 ? 0 : (r > ny - 1 ? ny - 1 : r)) * nx + ccol;
    const float4 pa = ldg(P.a + row_a); // in flight until the end of the iteration: A(r+1), t(r), maps(r)
    m_new.xm = ldg(P.xmapr + row_m);
    m_new.ym = ldg(P.ymapr + row_m);
    if (!TFP) {
      m_new.fc = ldg(P.fcoriolis + row_m);
      m_new.t = ldg(P.t + row_m);
    }

    // ---- stage A: intermediate row y = r-1, for every group the wave holds (halo groups included)
    const int y = r - 1;
    if (loadable && y >= 1 && y <= ny - 2 && y >= jb0 - 1 && y <= jb1) {
      const float* Sr = ringA + ((y - 1) % RA) * TS;
      const float* Cr = ringA + (y % RA) * TS;
      const float* Nr = ringA + ((y + 1) % RA) * TS;
      float sv[4], nv[4], cv[6], xm[4], ym[4];
      unpack(ld4(Sr + p), sv);
      unpack(ld4(Nr + p), nv);
      row6(Cr, p, cv);
      unpack(m_a.xm, xm);
      unpack(m_a.ym, ym);
      if (TFP) {
        const bool counted = CHECK && owned && y >= jb0 && y < jb1; // every cell is counted by the wave that owns it
        float g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          bool ok;
          g[k] = tfp_absdelt<CHECK, CANON>(sv[k], cv[k], cv[k + 2], nv[k], xm[k], ym[k], undef, ok);
          const bool edge_cell = (fill_w && k == 0) || (fill_e && k == k_e); // counted by edge_count_cells
          if (counted && !edge_cell && !ok && k < nvalid)
            ++n1;
        }
        fill_cols(g);
        st4(mid0 + (y % 3) * TS + p, g);
      } else {
        float fc[4], ug[4], vg[4];
        unpack(m_a.fc, fc);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          qvec_gwind<CHECK>(sv[k], cv[k], cv[k + 2], nv[k], xm[k], ym[k], fc[k], undef, ug[k], vg[k]);
        fill_cols(ug);
        fill_cols(vg);
        st4(mid0 + (y % 3) * TS + p, ug);
        st4(mid1 + (y % 3) * TS + p, vg);
      }
    }

    // (no compiler barrier needed here: stage B takes x-neighbours only from intermediate row r-2, written an
    // iteration ago -- the barrier at the end of every iteration lies in between -- and from row r-1 only
    // the lane's own column group)

    // ---- stage B: result row j = r-2, owned groups
    const int j = r - 2;
    const bool have_row = owned && j >= jb0 && j < jb1;
    // Q-vector: x-neighbours of the temperature row j from the adjacent lanes (all lanes active here)
    float t_west = 0.f, t_east = 0.f;
    if (!TFP) {
      const float tx_first = m_b.t.x, tx_last = m_b.t.w;
      t_west = from_lower_lane(0.f, tx_last);
      t_east = from_upper_lane(0.f, tx_first);
    }
    float o[4] = {undef, undef, undef, undef};
    // the rows whose x-neighbours stage B needs: read by every lane that holds a group (the owned groups' neighbours come
    // from the adjacent lanes, row6), before the owned lanes are selected
    float mid6a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, mid6b[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool row_b = j >= jb0 && j < jb1; // wave-uniform
    if (row_b && lane < TQ) {
      row6(mid0 + (j % 3) * TS, p, mid6a); // |grad tx| | ug
      if (TFP)
        row6(ringA + (j % RA) * TS, p, mid6b); // tx
      else
        row6(mid1 + (j % 3) * TS, p, mid6b); // vg
    }
    if (have_row) {
      const int js = (j - 1 < 1) ? 1 : j - 1, jn = (j + 1 > ny - 2) ? ny - 2 : j + 1; // filled intermediate rows 0 / ny-1 are rows 1 / ny-2
      float xm[4], ym[4];
      unpack(m_b.xm, xm);
      unpack(m_b.ym, ym);
      if (TFP) {
        float gs[4], gn[4], ts[4], tn[4];
        const float(&gc)[6] = mid6a;
        const float(&tc)[6] = mid6b;
        unpack(ld4(mid0 + (js % 3) * TS + p), gs);
        unpack(ld4(mid0 + (jn % 3) * TS + p), gn);
        unpack(ld4(ringA + ((j - 1) % RA) * TS + p), ts);
        unpack(ld4(ringA + ((j + 1) % RA) * TS + p), tn);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          bool ok, by_test;
          o[k] = tfp_point<CHECK, CANON>(ts[k], tc[k], tc[k + 2], tn[k], gs[k], gc[k], gc[k + 1], gc[k + 2], gn[k], xm[k], ym[k], undef, ok, by_test);
          // the cells of columns 0 / nx-1 need wrapped neighbours for their tests: the edge kernel counts
          // them -- unless nothing is tested, then |grad| != 0 of the filled value is all there is
          const bool edge_cell = (fill_w && k == 0) || (fill_e && k == k_e);
          if (!(CHECK && edge_cell) && k < nvalid) {
            n2 += ok ? 0u : 1u;
            if (by_test)
              ++n2c;
          }
        }
      } else {
        float us[4], un[4], vs[4], vn[4], ts[4], tn[4], tc[6];
        const float(&uc)[6] = mid6a;
        const float(&vc)[6] = mid6b;
        unpack(ld4(mid0 + (js % 3) * TS + p), us);
        unpack(ld4(mid0 + (jn % 3) * TS + p), un);
        unpack(ld4(mid1 + (js % 3) * TS + p), vs);
        unpack(ld4(mid1 + (jn % 3) * TS + p), vn);
        unpack(t_south, ts);
        unpack(m_a.t, tn);
        tc[0] = t_west;
        tc[1] = m_b.t.x;
        tc[2] = m_b.t.y;
        tc[3] = m_b.t.z;
        tc[4] = m_b.t.w;
        tc[5] = t_east;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          bool ok;
          o[k] = qvec_point<OP>(us[k], uc[k], uc[k + 2], un[k], vs[k], vc[k], vc[k + 2], vn[k], ts[k], tc[k], tc[k + 2], tn[k], xm[k], ym[k], P.scale,
                                P.scale2, undef, ok);
          const bool edge_cell = (fill_w && k == 0) || (fill_e && k == k_e);
          if (!edge_cell && k < nvalid)
            n2 += ok ? 0u : 1u;
        }
      }
    }
    // fillEdges on the result: columns (every lane that holds a group is here: the value may come from the lane below), then
    // rows 0 / ny-1 with the stores
    if (row_b && lane < TQ)
      fill_cols(o);
    // rows that were in flight since the top of the iteration; the explicit vmcnt(0) (all paths,
    // loads only by now) keeps the compiler from waiting again -- behind the store -- at the loop edge
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (loadable) {
      if (load_a)
        *reinterpret_cast<float4*>(ringA + ((r + 1) % RA) * TS + p) = CANON ? canon_nan4(pa, undef) : pa;
    }
    lds_rows_visible();
    if (!TFP)
      t_south = m_b.t; // row r-2 is row (r+1)-3
    if (have_row) {
      const v4f q = {o[0], o[1], o[2], o[3]};
      auto put = [&](float* at) __attribute__((always_inline)) {
        if constexpr (!RAGGED) {
          __builtin_nontemporal_store(q, reinterpret_cast<v4f*>(at));
        } else if (nvalid == 4) {
          V4Any t;
          t.v = q;
          *reinterpret_cast<V4Any*>(at) = t;
        } else {
          at[0] = o[0];
          if (nvalid > 1)
            at[1] = o[1];
          if (nvalid > 2)
            at[2] = o[2];
        }
      };
      put(P.out + (size_t)j * nx + xq);
      if (j == 1)
        put(P.out + xq);
      if (j == ny - 2)
        put(P.out + (size_t)(ny - 1) * nx + xq);
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

#ifdef MIFC_MEASUREMENT_BUILD // libmifc_measure.so only
// diagnostic: the same quotient through shared_reciprocal()/quotient() and through the compiler's a / b
__global__ void division_check_kernel(const float* a, const float* b, const float* g, float* shared, float* plain, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double num = 0.5 * (double)a[i] * (double)b[i] * (double)MIFC_K_G, den = (double)g[i];
    shared[i] = (float)quotient(num, den, shared_reciprocal(den));
    plain[i] = (float)(num / den);
  }
}
#endif

bool aligned16(const void* p)
{
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

template <int OP, bool CHECK>
hipError_t launch(const Fused2Params& p, hipStream_t stream)
{
  const int interior = p.ny - 2;
  const int ntiles = (p.nx + TW - 1) / TW;
  const int nl = p.n_launch_levels > 0 ? p.n_launch_levels : 1;
  // enough workgroups to fill the chip several times over (16-20 single-wave workgroups fit a CU),
  // bands tall enough that the 4 (source) + 2 (maps) halo rows a band re-reads stay a small fraction
  const long want_bands = (256L * 16 * 4 + (long)ntiles * nl - 1) / ((long)ntiles * nl);
  int band = (int)((interior + want_bands - 1) / want_bands);
  if (band < 4)
    band = 4;
  if (band > 64)
    band = 64;
  if (env().fused2_band > 0) // A/B measurements
    band = env().fused2_band;
  const int nbands = (interior + band - 1) / band;
  const int n_main = nbands * ntiles;
  int n_edge = 0; // TFP with an ALL_DEFINED input counts its edge cells in the tiles (nothing wrapped is tested)
  if (OP != F2_TFP || CHECK) {
    n_edge = (2 * interior + 63) / 64;
    if (n_edge > 256)
      n_edge = 256;
  }
  for (int l0 = 0; l0 < nl; l0 += 65535) { // grid.y limit
    Fused2Params q = p;
    const int n = nl - l0 > 65535 ? 65535 : nl - l0;
    if (p.n_launch_levels > 0) {
      q.n_launch_levels = n;
      if (p.levels) {
        q.levels = p.levels + l0;
      } else if (l0 > 0) { // implicit numbering: shift the bases instead
        const size_t off = (size_t)l0 * (size_t)p.level_stride;
        q.a = p.a + off;
        q.t = p.t ? p.t + off : nullptr;
        q.out = p.out + off;
        q.counts = p.counts + 3 * (size_t)l0;
        q.scale_lev = p.scale_lev ? p.scale_lev + l0 : nullptr;
        q.scale2_lev = p.scale2_lev ? p.scale2_lev + l0 : nullptr;
      }
    }
    const bool ragged = (p.nx & 3) != 0 || !aligned16(p.a) || !aligned16(p.xmapr) || !aligned16(p.ymapr) || !aligned16(p.out) ||
                        (p.op != F2_TFP && (!aligned16(p.t) || !aligned16(p.fcoriolis))) || (p.n_launch_levels > 0 && (p.level_stride & 3) != 0);
    if (ragged)
      hipLaunchKernelGGL((fused2_tile_kernel<OP, CHECK, true>), dim3((unsigned)(n_main + n_edge), (unsigned)n), dim3(64), 0, stream, q, band, ntiles, n_main);
    else
      hipLaunchKernelGGL((fused2_tile_kernel<OP, CHECK>), dim3((unsigned)(n_main + n_edge), (unsigned)n), dim3(64), 0, stream, q, band, ntiles, n_main);
  }
  return hipGetLastError();
}

template <int OP>
hipError_t launch_op(const Fused2Params& p, hipStream_t stream)
{
  return p.check ? launch<OP, true>(p, stream) : launch<OP, false>(p, stream);
}

} // namespace

#ifdef MIFC_MEASUREMENT_BUILD
hipError_t launch_division_check(const float* a, const float* b, const float* g, float* shared, float* plain, size_t n, hipStream_t stream)
{
  hipLaunchKernelGGL(division_check_kernel, dim3(1024), dim3(256), 0, stream, a, b, g, shared, plain, n);
  return hipGetLastError();
}
#endif

bool fused2_supported(const Fused2Params& p)
{
  if (p.nx < 8 || p.ny < 3) // (any width from 8 columns on, any dword alignment: the RAGGED variant)
    return false;
  if ((p.nx & 3) != 0 && p.nx % TW == 1) // column nx-1 alone in a tile: the column its fill copy comes from belongs to another wave
    return false;
  if ((long)((p.nx + TW - 1) / TW) * ((p.ny - 2 + 3) / 4) + 256 >= 0x7fffffffL) // workgroups of the launch
    return false;
  if (!p.a || !p.xmapr || !p.ymapr || !p.out || !p.counts)
    return false;
  if (p.op != F2_TFP && (!p.t || !p.fcoriolis))
    return false;
  return true;
}

hipError_t launch_fused2(const Fused2Params& p, hipStream_t stream)
{
  if (!fused2_supported(p))
    return hipErrorInvalidValue;
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
