// mifc_stencil_rows.hip -- row-walking kernel for the single-input stencil
// operators: gradient compute 1..4 (FieldCalculations.cc:1985-2074),
// plevelgwind_xcomp (:638), plevelgwind_ycomp (:674), plevelgvort (:708),
// ilevelgwind (:1511).  Same machinery as the fused vorticity+divergence
// kernel (mifc_vortdiv.hip), with one input field instead of two:
//   * one wavefront = 64 lanes x V float4 (V separate 1-KiB segments) x R rows
//     of one level; rows j-1, j, j+1 live in a static register ring, the next
//     row is in flight (counted vmcnt waits, branch-free clamped loads);
//   * x neighbours by DPP wave shifts, wave-column edges by one scalar per row;
//   * the waves of a workgroup take consecutive levels of one tile; xmapr,
//     ymapr, fcoriolis of the tile are staged once per workgroup in LDS;
//   * address-order, XCD-aware block schedule; odd bands walk upwards so that
//     shared halo rows meet in L2; nontemporal stores;
//   * flat-loop semantics of the reference: wrapped neighbours at the edge
//     columns take part in the undefined count, fillEdges is folded into the
//     stores.
// Algorithmic traffic: 4 B read + 4 B written per cell (8 B written for
// ilevelgwind), map factors once per batch.
#include "mifc_scalar_cell.h"

#include <cstdlib>

namespace mifc {

namespace {

template <int V>
struct SRow
{
  v4f f[V];
  float e; // lane 63: value east of the wave-column; other lanes: value west of it
};

template <int OP, bool CHECK, int V>
__global__ __launch_bounds__(1024) void scalar_rows_kernel(const SRowsParams P)
{
  constexpr int W = 4; // ring: rows r-2 (being refilled), r-1, r, r+1
  constexpr int WCOLS = 256 * V;
  constexpr bool USE_XM = (OP != ST_GRAD_Y && OP != ST_GWIND_X);
  constexpr bool USE_YM = (OP != ST_GRAD_X && OP != ST_GWIND_Y);
  constexpr bool USE_FC = (OP == ST_GWIND_X || OP == ST_GWIND_Y || OP == ST_GVORT || OP == ST_IGWIND);
  constexpr bool TWO_OUT = (OP == ST_IGWIND);

  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int bid = blockIdx.x;
  const int seq = (bid & 7) * P.per_xcd + (bid >> 3);
  if (seq >= P.n_logical)
    return;
  // address order: level group slowest, then band, wave-column fastest
  const int per_level = P.uB * P.uW;
  const int lgroup = seq / per_level;
  const int rem = seq - lgroup * per_level;
  const int band = rem / P.uW;
  const int wc = rem - band * P.uW;
  const int lev0 = lgroup * P.wpb;

  const int nx = P.nx;
  int colq[V], colq_c[V];
  bool actq[V];
#pragma unroll
  for (int q = 0; q < V; ++q) {
    colq[q] = wc * WCOLS + q * 256 + lane * 4;
    actq[q] = colq[q] < nx;
    colq_c[q] = actq[q] ? colq[q] : nx - 4;
  }
  int east_col = wc * WCOLS + WCOLS;
  if (east_col > nx)
    east_col = nx;
  const int edge_col = (lane == 63) ? east_col : (wc * WCOLS - 1);
  const long idx_hi = (long)nx * P.ny - 1;

  const int jb = 1 + band * P.R; // rows 1 .. ny-2 are computed
  const int nr = (P.ny - 1 - jb < P.R) ? (P.ny - 1 - jb) : P.R;
  const float undef = P.undef;
  const bool up = (band & 1) != 0;

  const bool valid = (lev0 + wave) < P.nlev;
  const int lev = valid ? (lev0 + wave) : (P.nlev - 1);
  const bool all = CHECK ? (P.all_defined && P.all_defined[lev] != 0) : true;
  const float* __restrict__ f = P.f + (size_t)lev * P.in_stride;
  float* o0p = P.o0 + (size_t)lev * P.out_stride;
  float* o1p = TWO_OUT ? P.o1 + (size_t)lev * P.out_stride : nullptr;

  auto load_row = [&](int t) -> SRow<V> {
    const int tc = t > nr ? nr : t;
    const int rowl = up ? (nr - 1 - tc) : tc;
    const long base = (long)(jb + rowl) * nx;
    SRow<V> r;
#pragma unroll
    for (int q = 0; q < V; ++q)
      r.f[q] = ld4(f + base + colq_c[q]);
    long e = base + edge_col;
    e = e < 0 ? 0 : (e > idx_hi ? idx_hi : e);
    r.e = f[e];
    return r;
  };
  SRow<V> ring[W];
#pragma unroll
  for (int t = -1; t <= 1; ++t)
    ring[(t + 2) % W] = load_row(t);

  // map factors of the tile -> LDS, layout [array][row][segment][lane]
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  v4f* lds = reinterpret_cast<v4f*>(lds_raw);
  const int tile4 = P.R * 64 * V;
  for (int i = threadIdx.x; i < nr * 64 * V; i += blockDim.x) {
    const int row = i / (64 * V);
    int col = wc * WCOLS + (i - row * 64 * V) * 4;
    col = col < nx ? col : nx - 4;
    const long o = (long)(jb + row) * nx + col;
    if (USE_XM)
      lds[i] = ld4(P.xm + o);
    if (USE_YM)
      lds[tile4 + i] = ld4(P.ym + o);
    if (USE_FC)
      lds[2 * tile4 + i] = ld4(P.fc + o);
  }
  __syncthreads();
  if (!valid)
    return;

  unsigned int bad = 0;
  for (int rb = 0; rb < nr; rb += W) {
#pragma unroll
    for (int s = 0; s < W; ++s) {
      const int r = rb + s;
      if (r >= nr)
        goto level_done;
      ring[s % W] = load_row(r + 2); // replaces row r-2
      const SRow<V>& rp = ring[(s + 1) % W];
      const SRow<V>& rc = ring[(s + 2) % W];
      const SRow<V>& rn = ring[(s + 3) % W];
      const float east = lane_value(rc.e, 63);
      const int rl = up ? (nr - 1 - r) : r;
      const int j = jb + rl;

#pragma unroll
      for (int q = 0; q < V; ++q) {
        float fW = dpp_lower(rc.e, rc.f[q].w); // lane 0 keeps the west scalar
        float fE = dpp_upper(rc.e, rc.f[q].x); // lane 63 keeps the east scalar
        if (q > 0) {
          const float t = lane_value(rc.f[q > 0 ? q - 1 : 0].w, 63);
          fW = (lane == 0) ? t : fW;
        }
        if (q < V - 1) {
          const float t = lane_value(rc.f[q < V - 1 ? q + 1 : q].x, 0);
          fE = (lane == 63) ? t : fE;
        }
        if (colq[q] + 4 >= east_col)
          fE = east;
        const int li = (rl * V + q) * 64 + lane;
        v4f xm4 = rc.f[q], ym4 = rc.f[q], fc4 = rc.f[q];
        if (USE_XM)
          xm4 = lds[li];
        if (USE_YM)
          ym4 = lds[tile4 + li];
        if (USE_FC)
          fc4 = lds[2 * tile4 + li];
        const float fc6[6] = {fW, rc.f[q].x, rc.f[q].y, rc.f[q].z, rc.f[q].w, fE};
        float z0[4], z1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a = rp.f[q][k], b = rn.f[q][k]; // previous / next step of the walk
          const float sv = up ? b : a;               // row j-1
          const float nv = up ? a : b;               // row j+1
          float r0 = undef, r1 = undef;
          const bool ok = scalar_cell<OP, CHECK>(all, undef, fc6[k], fc6[k + 1], fc6[k + 2], sv, nv, xm4[k], ym4[k], fc4[k], r0, r1);
          z0[k] = ok ? r0 : undef;
          z1[k] = ok ? r1 : undef;
          if (CHECK)
            bad += (!ok & actq[q]) ? 1u : 0u;
        }
        if (colq[q] == 0) { // fillEdges, column part
          z0[0] = z0[1];
          z1[0] = z1[1];
        }
        if (colq[q] + 4 == nx) {
          z0[3] = z0[2];
          z1[3] = z1[2];
        }
        if (actq[q]) {
          const long o = (long)j * nx + colq[q];
          st4_stream(o0p + o, z0);
          if (j == 1) // fillEdges, row part
            st4_stream(o0p + o - nx, z0);
          if (j == P.ny - 2)
            st4_stream(o0p + o + nx, z0);
          if (TWO_OUT) {
            st4_stream(o1p + o, z1);
            if (j == 1)
              st4_stream(o1p + o - nx, z1);
            if (j == P.ny - 2)
              st4_stream(o1p + o + nx, z1);
          }
        }
      }
    }
  }
level_done:
  if (CHECK && P.n_undefined)
    wave_count_add(P.n_undefined + lev, bad);
}

// One-shot form for small launches (the reference's single-field call): no row loop, no LDS -- a workgroup
// is 4 waves = 4 rows x 256 columns of one level; every lane loads the float4 of its own row and of the
// rows above and below plus the map factors it needs, takes its x-neighbours from the adjacent lanes and
// stores.  A wave lives for one memory round trip; with fewer than ~2000 waves in the launch that beats
// walking even 2-row bands.
template <int OP, bool CHECK>
__global__ __launch_bounds__(256) void scalar_oneshot_kernel(const SRowsParams P)
{
  constexpr bool USE_XM = (OP != ST_GRAD_Y && OP != ST_GWIND_X);
  constexpr bool USE_YM = (OP != ST_GRAD_X && OP != ST_GWIND_Y);
  constexpr bool USE_FC = (OP == ST_GWIND_X || OP == ST_GWIND_Y || OP == ST_GVORT || OP == ST_IGWIND);
  constexpr bool TWO_OUT = (OP == ST_IGWIND);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int seq = blockIdx.x; // address order: column segment fastest, then row block, then level
  const int per_level = P.uB * P.uW;
  const int lev = seq / per_level;
  const int rem = seq - lev * per_level;
  const int rblock = rem / P.uW;
  const int wc = rem - rblock * P.uW;
  const int nx = P.nx, ny = P.ny;
  // a wave past the last computed row works on that row and keeps the result to itself: it stays for the workgroup-wide
  // count at the end (one atomic per workgroup: mifc_device.h, undefined-cell counting)
  const int j_raw = 1 + rblock * 4 + wave;
  const bool live = j_raw <= ny - 2;
  const int j = live ? j_raw : ny - 2;
  const int col = wc * 256 + lane * 4;
  const bool act = live && col < nx;
  const int col_c = (col < nx) ? col : nx - 4;
  int east_col = wc * 256 + 256;
  if (east_col > nx)
    east_col = nx;
  const int edge_col = (lane == 63) ? east_col : (wc * 256 - 1);
  const float undef = P.undef;
  const bool all = CHECK ? (P.all_defined && P.all_defined[lev] != 0) : true;
  const float* __restrict__ f = P.f + (size_t)lev * P.in_stride;
  const long base = (long)j * nx;
  const long o = base + col_c;
  const v4f fc = ld4(f + o), fn = ld4(f + o + nx), fs = ld4(f + o - nx);
  v4f xm4 = fc, ym4 = fc, co4 = fc;
  if (USE_XM)
    xm4 = ld4(P.xm + o);
  if (USE_YM)
    ym4 = ld4(P.ym + o);
  if (USE_FC)
    co4 = ld4(P.fc + o);
  const float ef = f[base + edge_col]; // (nx-1, j-1) west of column 0, (0, j+1) east of column nx-1: the flat loop's neighbours

  const float east = lane_value(ef, 63);
  const float fW = dpp_lower(ef, fc.w);
  float fE = dpp_upper(ef, fc.x);
  if (col + 4 >= east_col)
    fE = east;
  const float fc6[6] = {fW, fc.x, fc.y, fc.z, fc.w, fE};
  float z0[4], z1[4];
  unsigned int bad = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float r0 = undef, r1 = undef;
    const bool ok = scalar_cell<OP, CHECK>(all, undef, fc6[k], fc6[k + 1], fc6[k + 2], fs[k], fn[k], xm4[k], ym4[k], co4[k], r0, r1);
    z0[k] = ok ? r0 : undef;
    z1[k] = ok ? r1 : undef;
    if (CHECK)
      bad += (!ok & act) ? 1u : 0u;
  }
  if (col == 0) { // fillEdges, column part
    z0[0] = z0[1];
    z1[0] = z1[1];
  }
  if (col + 4 == nx) {
    z0[3] = z0[2];
    z1[3] = z1[2];
  }
  if (act) {
    float* o0p = P.o0 + (size_t)lev * P.out_stride;
    const long oo = base + col;
    st4_stream(o0p + oo, z0);
    if (j == 1) // fillEdges, row part
      st4_stream(o0p + oo - nx, z0);
    if (j == ny - 2)
      st4_stream(o0p + oo + nx, z0);
    if (TWO_OUT) {
      float* o1p = P.o1 + (size_t)lev * P.out_stride;
      st4_stream(o1p + oo, z1);
      if (j == 1)
        st4_stream(o1p + oo - nx, z1);
      if (j == ny - 2)
        st4_stream(o1p + oo + nx, z1);
    }
  }
  if (CHECK) {
    if (P.partials)
      block_count_store(P.partials + seq, bad); // one big level: added up by launch_count_partials behind this launch
    else
      block_count_add(P.n_undefined ? P.n_undefined + lev : nullptr, bad); // every wave of the workgroup is on this level
  }
}

// Level-walking form for deep batches (same design as vortdiv_levelwalk_kernel, mifc_vortdiv.hip): a workgroup
// is 12 waves = a tile of 10 rows x 256 columns plus one wave each for the row above and below it; it stays on
// its tile and walks through a chunk of levels.  The tile's map factors live in registers for the whole walk;
// per level a wave loads one row of the field (the row of the next level is in flight meanwhile), parks it in
// one of two LDS buffers, one LDS-only barrier, rows above / below from LDS, x-neighbours by DPP, streaming stores.
constexpr int LW_WAVES = 12;
constexpr int LW_ROWS = LW_WAVES - 2;

template <int OP, bool CHECK>
__global__ __launch_bounds__(64 * LW_WAVES, 6) void scalar_levelwalk_kernel(const SRowsParams P)
{
  constexpr bool USE_XM = (OP != ST_GRAD_Y && OP != ST_GWIND_X);
  constexpr bool USE_YM = (OP != ST_GRAD_X && OP != ST_GWIND_Y);
  constexpr bool USE_FC = (OP == ST_GWIND_X || OP == ST_GWIND_Y || OP == ST_GVORT || OP == ST_IGWIND);
  constexpr bool TWO_OUT = (OP == ST_IGWIND);
  __shared__ v4f sf[2][LW_WAVES][64];
  // undefined counts of a level: the waves add theirs up in LDS, thread 0 (a halo wave) hands the total of level l to the
  // global counter after the barrier of level l + 1 -- ONE global atomic per workgroup and level (with
  // one per wave a masked field queued 4 320 same-address atomics per level)
  __shared__ unsigned int sbad[2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int bid = blockIdx.x;
  const int seq = (bid & 7) * P.per_xcd + (bid >> 3);
  if (seq >= P.n_logical)
    return;
  if (CHECK && threadIdx.x < 2)
    sbad[threadIdx.x] = 0; // ordered before the first add by the barrier of the first level
  // unit = (level chunk, row block, column segment), column segment fastest
  const int ntiles = P.uB * P.uW;
  const int lchunk = seq / ntiles;
  const int tile = seq - lchunk * ntiles;
  const int rblock = tile / P.uW;
  const int wc = tile - rblock * P.uW;
  const int lev0 = lchunk * P.wpb; // wpb: levels per chunk here
  const int lev1 = (lev0 + P.wpb < P.nlev) ? lev0 + P.wpb : P.nlev;

  const int nx = P.nx, ny = P.ny;
  const int first = 1 + rblock * LW_ROWS; // rows 1 .. ny-2 are computed
  const int j_raw = first + wave - 1;     // LDS slot = wave: slot 0 is the row above the tile
  const bool computes = wave >= 1 && wave <= LW_ROWS && j_raw <= ny - 2;
  const int j = j_raw < ny - 1 ? j_raw : ny - 1; // rows past the field: loaded from the last row, never used
  const int col = wc * 256 + lane * 4;
  const bool act = col < nx;
  const int col_c = act ? col : nx - 4;
  int east_col = wc * 256 + 256;
  if (east_col > nx)
    east_col = nx;
  const int edge_col = (lane == 63) ? east_col : (wc * 256 - 1);
  const float undef = P.undef;
  const int base = j * nx; // offsets inside a level fit 32 bits (the launcher checks)
  const int o = base + col_c;
  long e64 = (long)base + edge_col;
  const long idx_hi = (long)nx * ny - 1;
  e64 = e64 < 0 ? 0 : (e64 > idx_hi ? idx_hi : e64);
  const int e = (int)e64;
  const int oo = base + col;
  const bool last_in_seg = col + 4 >= east_col;

  v4f xm4 = {0.f, 0.f, 0.f, 0.f}, ym4 = xm4, fc4 = xm4;
  if (computes) {
    if (USE_XM)
      xm4 = ld4(P.xm + o);
    if (USE_YM)
      ym4 = ld4(P.ym + o);
    if (USE_FC)
      fc4 = ld4(P.fc + o);
  }
  struct Lev
  {
    v4f f;
    float e;
  };
  auto load_level = [&](int lev) -> Lev {
    const int l = lev < lev1 ? lev : lev1 - 1;
    const float* __restrict__ f = P.f + (size_t)l * P.in_stride;
    Lev r;
    r.f = ld4(f + o);
    r.e = f[e];
    return r;
  };
  Lev R[2];
  R[0] = load_level(lev0);
  for (int lb = lev0; lb < lev1; lb += 2) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int lev = lb + s;
      if (lev >= lev1)
        goto chunk_done;
      R[s ^ 1] = load_level(lev + 1); // in flight while this level is computed
      const Lev& C = R[s];
      const int buf = (lev - lev0) & 1;
      sf[buf][wave][lane] = C.f;
      // only the LDS counter is waited for: the prefetch above and earlier stores stay in flight.  Two buffers: a
      // wave can overwrite buffer b again only after the barrier of the level in between.
      // (the level's input flag through the scalar cache, with the same wait: level_flag_then_barrier, mifc_device.h)
      bool all = true;
      if (CHECK)
        all = level_flag_then_barrier(P.all_defined, lev);
      else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (CHECK && P.n_undefined && threadIdx.x == 0 && lev > lev0) { // the previous level's adds are complete
        const int q = (lev - 1 - lev0) & 1;
        const unsigned int n = sbad[q];
        if (n != 0) {
          atomicAdd(P.n_undefined + (lev - 1), (u64)n);
          sbad[q] = 0; // the next adds into this slot come after the next barrier
        }
      }
      if (computes) {
        const v4f fn = sf[buf][wave + 1][lane], fs = sf[buf][wave - 1][lane];
        const float east = lane_value(C.e, 63);
        const float fW = dpp_lower(C.e, C.f.w);
        float fE = dpp_upper(C.e, C.f.x);
        if (last_in_seg)
          fE = east;
        const float fc6[6] = {fW, C.f.x, C.f.y, C.f.z, C.f.w, fE};
        float z0[4], z1[4];
        unsigned int bad = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float r0 = undef, r1 = undef;
          const bool ok = scalar_cell<OP, CHECK>(all, undef, fc6[k], fc6[k + 1], fc6[k + 2], fs[k], fn[k], xm4[k], ym4[k], fc4[k], r0, r1);
          z0[k] = ok ? r0 : undef;
          z1[k] = ok ? r1 : undef;
          if (CHECK)
            bad += (!ok & act) ? 1u : 0u;
        }
        if (col == 0) { // fillEdges, column part
          z0[0] = z0[1];
          z1[0] = z1[1];
        }
        if (col + 4 == nx) {
          z0[3] = z0[2];
          z1[3] = z1[2];
        }
        if (act) {
          float* o0p = P.o0 + (size_t)lev * P.out_stride;
          st4_stream(o0p + oo, z0);
          if (j == 1) // fillEdges, row part
            st4_stream(o0p + oo - nx, z0);
          if (j == ny - 2)
            st4_stream(o0p + oo + nx, z0);
          if (TWO_OUT) {
            float* o1p = P.o1 + (size_t)lev * P.out_stride;
            st4_stream(o1p + oo, z1);
            if (j == 1)
              st4_stream(o1p + oo - nx, z1);
            if (j == ny - 2)
              st4_stream(o1p + oo + nx, z1);
          }
        }
        if (CHECK && P.n_undefined && !all && __builtin_amdgcn_ballot_w64(bad != 0) != 0) {
          const unsigned int n = wave_sum(bad);
          if (lane == 0)
            atomicAdd(&sbad[(lev - lev0) & 1], n);
        }
      }
    }
  }
chunk_done:;
  if (CHECK) { // the last level's count
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (P.n_undefined && threadIdx.x == 0 && lev1 > lev0) {
      const unsigned int n = sbad[(lev1 - 1 - lev0) & 1];
      if (n != 0)
        atomicAdd(P.n_undefined + (lev1 - 1), (u64)n);
    }
  }
}

template <int OP>
void launch_op(const SRowsParams& rp, bool check, int V, int grid, size_t lds, hipStream_t stream)
{
  if (V == 3) { // level-walking form: rp.uB / rp.uW are 10-row blocks / 256-column segments, rp.wpb the levels per chunk
    if (check)
      hipLaunchKernelGGL((scalar_levelwalk_kernel<OP, true>), dim3(grid), dim3(64 * LW_WAVES), 0, stream, rp);
    else
      hipLaunchKernelGGL((scalar_levelwalk_kernel<OP, false>), dim3(grid), dim3(64 * LW_WAVES), 0, stream, rp);
    return;
  }
  if (V == 0) { // one-shot form: rp.uB / rp.uW are row blocks of 4 / 256-column segments, grid = levels * uB * uW
    if (check)
      hipLaunchKernelGGL((scalar_oneshot_kernel<OP, true>), dim3(grid), dim3(256), 0, stream, rp);
    else
      hipLaunchKernelGGL((scalar_oneshot_kernel<OP, false>), dim3(grid), dim3(256), 0, stream, rp);
    return;
  }
  const dim3 block(64 * rp.wpb);
  if (check) {
    if (V == 2)
      hipLaunchKernelGGL((scalar_rows_kernel<OP, true, 2>), dim3(grid), block, lds, stream, rp);
    else
      hipLaunchKernelGGL((scalar_rows_kernel<OP, true, 1>), dim3(grid), block, lds, stream, rp);
  } else {
    if (V == 2)
      hipLaunchKernelGGL((scalar_rows_kernel<OP, false, 2>), dim3(grid), block, lds, stream, rp);
    else
      hipLaunchKernelGGL((scalar_rows_kernel<OP, false, 1>), dim3(grid), block, lds, stream, rp);
  }
}

inline bool a16(const void* p)
{
  return (reinterpret_cast<size_t>(p) & 15u) == 0;
}

} // namespace

// Fast path for whole fields with nx % 4 == 0 and 16-byte aligned pointers;
// otherwise *handled stays false and the one-lane-per-cell kernel runs.
// gradient compute 1 (FieldCalculations.cc:2013-2021) tests and counts over the flat cells 1 .. nx*ny-2: rows 0 and ny-1
// as well (cells 1 .. nx-1 of row 0 and 0 .. nx-2 of row ny-1, with the flat neighbours i-1 / i+1).  Their values are
// overwritten by fillEdges; their share of the undefined count is added here.  grid.y = level.
__global__ __launch_bounds__(256) void gradx_outer_rows_count_kernel(const SRowsParams P)
{
  const int lev = blockIdx.y;
  unsigned int bad = 0;
  if (!(P.all_defined && P.all_defined[lev] != 0)) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int per_row = P.nx - 1;
    if (t < 2 * per_row) {
      const long i = t < per_row ? 1 + t : (long)(P.ny - 1) * P.nx + (t - per_row);
      const float* f = P.f + (size_t)lev * P.in_stride;
      bad = all_def(P.undef, f[i - 1], f[i + 1]) ? 0u : 1u;
    }
  }
  block_count_add(P.n_undefined + lev, bad);
}

hipError_t launch_scalar_rows(const StencilParams& prm, hipStream_t stream, bool* handled)
{
  *handled = false;
  const int op = prm.op;
  if (op < ST_GRAD_X || op > ST_IGWIND)
    return hipSuccess;
  const int nx = prm.nx, ny = prm.ny_global;
  if (prm.j0 != 0 || prm.ny_local != ny || nx < 8 || ny < 3)
    return hipSuccess;
  // Rows at any alignment (a width that is not a multiple of 4, fields or level strides off the 16-byte grid): the
  // split-role level-walking form has a variant for them (RAGGED, mifc_stencil_split.hip); the other forms of this file
  // do not, and such a call outside that form's range goes to the flat kernels.
  {
    const bool uxm = (op != ST_GRAD_Y && op != ST_GWIND_X), uym = (op != ST_GRAD_X && op != ST_GWIND_Y);
    const bool ufc = (op == ST_GWIND_X || op == ST_GWIND_Y || op == ST_GVORT || op == ST_IGWIND);
    const bool present = prm.f0 && prm.out0 && (!uxm || prm.xmapr) && (!uym || prm.ymapr) && (!ufc || prm.fcoriolis) && (op != ST_IGWIND || prm.out1);
    const bool ragged = nx % 4 != 0 || !a16(prm.f0) || !a16(prm.out0) || (uxm && !a16(prm.xmapr)) || (uym && !a16(prm.ymapr)) || (ufc && !a16(prm.fcoriolis)) ||
                        (op == ST_IGWIND && !a16(prm.out1)) || prm.in_level_stride % 4 != 0 || prm.out_level_stride % 4 != 0;
    if (present && ragged) {
      const bool check = !prm.every_level_all_defined;
      // (nx % 256 == 1: the column whose value fillEdges copies into column nx-1 belongs to another workgroup)
      if (nx % 256 == 1 || env().force_cell_kernel || env().scalar_rows_r >= 0 || !scalar_split_applies(op, nx, ny, prm.nlev, check, prm.undef, true))
        return hipSuccess;
      SRowsParams rp{};
      rp.nx = nx;
      rp.ny = ny;
      rp.nlev = prm.nlev;
      rp.f = prm.f0;
      rp.xm = prm.xmapr;
      rp.ym = prm.ymapr;
      rp.fc = prm.fcoriolis;
      rp.o0 = prm.out0;
      rp.o1 = prm.out1;
      rp.in_stride = prm.in_level_stride;
      rp.out_stride = prm.out_level_stride;
      rp.all_defined = prm.all_defined;
      rp.undef = prm.undef;
      rp.n_undefined = prm.n_undefined;
      rp.ragged = 1;
      *handled = true;
      note_form("scalar_split_ragged");
      const hipError_t e = launch_scalar_split(op, rp, check, stream);
      if (e == hipSuccess && op == ST_GRAD_X && check && rp.n_undefined) {
        for (int l0 = 0; l0 < prm.nlev; l0 += 65535) { // grid.y limit; see count_outer_rows below
          SRowsParams cp = rp;
          const int nl = prm.nlev - l0 > 65535 ? 65535 : prm.nlev - l0;
          cp.f = rp.f + (size_t)l0 * rp.in_stride;
          cp.all_defined = rp.all_defined ? rp.all_defined + l0 : nullptr;
          cp.n_undefined = rp.n_undefined + l0;
          hipLaunchKernelGGL(gradx_outer_rows_count_kernel, dim3((unsigned)((2 * (nx - 1) + 255) / 256), (unsigned)nl), dim3(256), 0, stream, cp);
        }
      }
      return e != hipSuccess ? e : hipGetLastError();
    }
  }
  if (nx % 4 != 0)
    return hipSuccess;
  // gradient compute 1 counts over the flat range [1, nx*ny-1), i.e. also in rows 0 and ny-1, which the row kernels
  // do not walk: gradx_outer_rows_count_kernel adds those cells' share behind the main launch (values there are fill copies)
  const bool use_xm = (op != ST_GRAD_Y && op != ST_GWIND_X);
  const bool use_ym = (op != ST_GRAD_X && op != ST_GWIND_Y);
  const bool use_fc = (op == ST_GWIND_X || op == ST_GWIND_Y || op == ST_GVORT || op == ST_IGWIND);
  if (!a16(prm.f0) || !a16(prm.out0) || (use_xm && (!prm.xmapr || !a16(prm.xmapr))) || (use_ym && (!prm.ymapr || !a16(prm.ymapr))) ||
      (use_fc && (!prm.fcoriolis || !a16(prm.fcoriolis))))
    return hipSuccess;
  if (op == ST_IGWIND && (!prm.out1 || !a16(prm.out1)))
    return hipSuccess;
  if (prm.in_level_stride % 4 != 0 || prm.out_level_stride % 4 != 0)
    return hipSuccess;
  if (env().force_cell_kernel)
    return hipSuccess;

  SRowsParams rp{};
  const int V = (nx > 256) ? 2 : 1;
  rp.nx = nx;
  rp.ny = ny;
  rp.nwc = (nx + 256 * V - 1) / (256 * V);
  rp.nlev = prm.nlev;
  int wpb = 8;
  while (wpb > 1 && wpb / 2 >= prm.nlev)
    wpb /= 2;
  // 8-row bands; a small launch (the reference's single-field call: one level) is latency-bound and
  // gets shorter ones -- more waves on the chip, the halo re-reads stay in L2
  rp.R = 8;
  const int forced_r = env().scalar_rows_r; // -1: not set
  if (forced_r < 0) {
    while (rp.R > 2 && (long)prm.nlev * ((ny - 2 + rp.R - 1) / rp.R) * rp.nwc < 2048) // waves of the launch
      rp.R /= 2;
  } else if (forced_r > 0) {
    rp.R = forced_r; // A/B measurements
    if (rp.R > 8)
      rp.R = 8;
  }
  rp.nbands = (ny - 2 + rp.R - 1) / rp.R;
  rp.wpb = wpb;
  rp.uL = (prm.nlev + wpb - 1) / wpb;
  rp.uB = rp.nbands;
  rp.uW = rp.nwc;
  const long n_logical = (long)rp.uL * rp.uB * rp.uW;
  if (n_logical > 0x3fffffffL)
    return hipSuccess;
  rp.n_logical = (int)n_logical;
  rp.per_xcd = (rp.n_logical + 7) / 8;
  rp.f = prm.f0;
  rp.xm = prm.xmapr;
  rp.ym = prm.ymapr;
  rp.fc = prm.fcoriolis;
  rp.o0 = prm.out0;
  rp.o1 = prm.out1;
  rp.in_stride = prm.in_level_stride;
  rp.out_stride = prm.out_level_stride;
  rp.all_defined = prm.all_defined;
  rp.undef = prm.undef;
  rp.n_undefined = prm.n_undefined;
  int grid = rp.per_xcd * 8;
  const size_t lds = (size_t)rp.R * 1024 * V * 3;
  const bool check = !prm.every_level_all_defined;
  int form = V;
  // a small launch (fewer than 2048 waves even with 8-row bands) takes the one-shot form, and so do one or
  // two levels of any size: their row-walking workgroups would be single waves holding a 48-KiB tile of map
  // factors, three to a CU
  // (MIFC_LEVELWALK_MIN_UNITS, the tests' switch, sends launches of any size to the level-walking forms)
  if (forced_r < 0 && ((env().levelwalk_min_units <= 0 && (long)prm.nlev * ((ny - 2 + 7) / 8) * rp.nwc < 2048) || prm.nlev <= 2)) {
    rp.uB = (ny - 2 + 3) / 4;
    rp.uW = (nx + 255) / 256;
    const long units = (long)prm.nlev * rp.uB * rp.uW;
    if (units <= 0x7fffffffL) {
      grid = (int)units;
      form = 0;
      // one big level with tests: the workgroups' counts by plain stores + one small launch (StencilParams::partials)
      if (check && prm.partials && prm.n_undefined && (long)rp.uB * rp.uW >= 2048 && units <= prm.partials_cap)
        rp.partials = prm.partials;
    }
  }

  // Deep batches: tiles that stay put and walk the levels (see scalar_levelwalk_kernel).  Measured on 1440x720x137
  // (profiles/r02/experiments/ab_levelwalk_ops.txt): the two one-sided gradients gain 2-5 %, the operators with
  // more arithmetic per cell (|grad|: a square root; gvort, gwind: f64 divisions) lose 2-16 % -- all waves of a
  // workgroup compute at the same phase here, while the row-walking waves drift apart and overlap their
  // arithmetic with each other's memory time.  So only the light ones take this form by default; with
  // MIFC_LEVELWALK_MIN_UNITS set (tests) every operator does.
  const bool light = (op == ST_GRAD_X || op == ST_GRAD_Y);
  if (form != 0 && forced_r < 0 && env().levelwalk && (light || env().levelwalk_min_units > 0) && prm.nlev >= 3 && (long)nx * ny < 0x7fffffffL) {
    const int target = prm.nlev >= 48 ? 6 : 8; // levels per chunk, then balanced
    const int nchunks = (prm.nlev + target - 1) / target;
    const long tiles = (long)((ny - 2 + LW_ROWS - 1) / LW_ROWS) * ((nx + 255) / 256);
    if (tiles * nchunks >= (env().levelwalk_min_units > 0 ? env().levelwalk_min_units : 768) && tiles * nchunks <= 0x3fffffffL) {
      rp.uB = (ny - 2 + LW_ROWS - 1) / LW_ROWS;
      rp.uW = (nx + 255) / 256;
      rp.wpb = (prm.nlev + nchunks - 1) / nchunks;
      rp.n_logical = (int)(tiles * ((prm.nlev + rp.wpb - 1) / rp.wpb));
      rp.per_xcd = (rp.n_logical + 7) / 8;
      grid = rp.per_xcd * 8;
      form = 3;
    }
  }

  *handled = true;
  auto count_outer_rows = [&]() { // gradient compute 1 with tests: the share of rows 0 and ny-1 (see gradx_outer_rows_count_kernel)
    if (!(check && rp.n_undefined))
      return;
    for (int l0 = 0; l0 < prm.nlev; l0 += 65535) { // grid.y limit
      SRowsParams cp = rp;
      const int nl = prm.nlev - l0 > 65535 ? 65535 : prm.nlev - l0;
      cp.f = rp.f + (size_t)l0 * rp.in_stride;
      cp.all_defined = rp.all_defined ? rp.all_defined + l0 : nullptr;
      cp.n_undefined = rp.n_undefined + l0;
      hipLaunchKernelGGL(gradx_outer_rows_count_kernel, dim3((unsigned)((2 * (nx - 1) + 255) / 256), (unsigned)nl), dim3(256), 0, stream, cp);
    }
  };
  // Deep batches, round 3: the level-walking tiles with split roles (mifc_stencil_split.hip) -- loader waves fill LDS two
  // levels ahead, compute waves only read LDS and store, two workgroups per CU -- for every operator of the family.
  note_form(form == 0 ? "scalar_oneshot" : form == 3 ? "scalar_levelwalk" : "scalar_rows");
  if (form != 0 && forced_r < 0 && scalar_split_applies(op, nx, ny, prm.nlev, check, prm.undef, false)) {
    note_form("scalar_split");
    // big tested levels: room for the tiles' counts (the split-role launcher keeps it if the levels are big enough)
    if (check && prm.partials) {
      rp.partials = prm.partials;
      rp.partials_cap = prm.partials_cap;
    }
    const hipError_t e = launch_scalar_split(op, rp, check, stream);
    if (e == hipSuccess && rp.partials)
      (void)launch_count_partials_levels(rp.partials, rp.uB * rp.uW, prm.nlev, prm.n_undefined, stream);
    if (e == hipSuccess && op == ST_GRAD_X)
      count_outer_rows();
    return e != hipSuccess ? e : hipGetLastError();
  }
  switch (op) {
  case ST_GRAD_X:
    launch_op<ST_GRAD_X>(rp, check, form, grid, lds, stream);
    count_outer_rows();
    break;
  case ST_GRAD_Y:
    launch_op<ST_GRAD_Y>(rp, check, form, grid, lds, stream);
    break;
  case ST_GRAD_ABS:
    launch_op<ST_GRAD_ABS>(rp, check, form, grid, lds, stream);
    break;
  case ST_GRAD_LAP:
    launch_op<ST_GRAD_LAP>(rp, check, form, grid, lds, stream);
    break;
  case ST_GWIND_X:
    launch_op<ST_GWIND_X>(rp, check, form, grid, lds, stream);
    break;
  case ST_GWIND_Y:
    launch_op<ST_GWIND_Y>(rp, check, form, grid, lds, stream);
    break;
  case ST_GVORT:
    launch_op<ST_GVORT>(rp, check, form, grid, lds, stream);
    break;
  default:
    launch_op<ST_IGWIND>(rp, check, form, grid, lds, stream);
    break;
  }
  if (rp.partials)
    (void)launch_count_partials_levels(rp.partials, rp.uB * rp.uW, prm.nlev, prm.n_undefined, stream);
  return hipGetLastError();
}

} // namespace mifc
