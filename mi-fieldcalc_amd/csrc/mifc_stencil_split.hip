// mifc_stencil_split.hip -- split-role level-walking form of the single-input stencil operators: gradient compute 1..4
// (FieldCalculations.cc:1985-2074), plevelgwind_xcomp (:638), plevelgwind_ycomp (:674), plevelgvort (:708),
// ilevelgwind (:1511) over a deep batch of levels.
//
// The design of vortdiv_split_kernel (mifc_vortdiv.hip) with one input field:
//   * a workgroup owns a tile of TR rows x 256 columns and walks through a chunk of levels; the tile's map factors and
//     Coriolis parameter live in registers for the whole walk (map factors once per chunk of levels);
//   * NL loader waves bring the TR + 2 rows of a level (and the two edge scalars of every row) straight from global
//     memory into LDS (global_load_lds: no registers, no ds_write), PF levels ahead, into a ring of PF + 1 level buffers;
//   * TR compute waves (one row each) take their row, the rows above and below and the edge scalars from LDS, the
//     x-neighbours from the adjacent lanes (DPP), and store 1 KiB (2 for ilevelgwind) nontemporally: the only entries in
//     their vmcnt queue are stores, which nothing in the loop waits for.  On gfx9 a wave's loads and stores share one
//     in-order counter; a wave that does both sees the data of level l + 1 only after its stores of level l - 1 have been
//     acknowledged (the coupling the copy yardsticks price at 10-15 %);
//   * ONE barrier per level; a workgroup holds (PF + 1) x (TR + 2) KiB of LDS and at most 64 VGPRs per lane, so TWO
//     workgroups share a CU and their arithmetic phases (a correctly rounded square root, f64 quotients) overlap each
//     other's memory time -- what the first level-walking form (all waves of a CU in step) lacked for the heavy operators;
//   * undefined counts: compute waves add into one of two LDS slots, the last loader hands the total of level l to the
//     level's counter after the barrier of level l + 1 -- one global atomic per workgroup and level.
// Flat-loop semantics as in the other forms: the edge scalars are the flat neighbours i - 1 / i + 1 (wrapped into the
// adjacent row at columns 0 / nx - 1, clamped to the field), fillEdges is folded into the stores.
#include "mifc_scalar_cell.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace mifc {

namespace {

// RAGGED: any width, dword-aligned fields (see vortdiv_split_kernel, mifc_vortdiv.hip: rows start wherever they start; the loads
// do not care, the stores are 16-byte stores at dword alignment -- 75 % of the aligned store rate --, the last column group of
// a row may be partial, and the one group that would read past the end of the batch is loaded cell by cell).
struct __attribute__((packed, aligned(4))) V4Any
{
  v4f v;
};
__device__ __forceinline__ void st4_any_alignment(float* p, const float (&z)[4], int nvalid)
{
  if (nvalid >= 4) {
    V4Any t;
    t.v.x = z[0];
    t.v.y = z[1];
    t.v.z = z[2];
    t.v.w = z[3];
    *reinterpret_cast<V4Any*>(p) = t;
  } else {
    if (nvalid > 0)
      p[0] = z[0];
    if (nvalid > 1)
      p[1] = z[1];
    if (nvalid > 2)
      p[2] = z[2];
  }
}

template <int OP, bool CHECK, int TR, int NL, int PF, bool RAGGED = false>
__global__ __launch_bounds__(64 * (TR + NL), (TR + NL <= 16) ? (2 * (TR + NL) + 3) / 4 : 1) void scalar_split_kernel(const SRowsParams P)
{
  constexpr bool USE_XM = (OP != ST_GRAD_Y && OP != ST_GWIND_X);
  constexpr bool USE_YM = (OP != ST_GRAD_X && OP != ST_GWIND_Y);
  constexpr bool USE_FC = (OP == ST_GWIND_X || OP == ST_GWIND_Y || OP == ST_GVORT || OP == ST_IGWIND);
  constexpr bool TWO_OUT = (OP == ST_IGWIND);
  // the operator reads the x-neighbours, hence the edge scalars: in its formula, or -- plevelgwind_xcomp -- only in its test
  // ("check for y component input, too", FieldCalculations.cc:659-660)
  constexpr bool NEED_X = USE_XM || (CHECK && OP == ST_GWIND_X);
  constexpr int NB = PF + 1;               // level buffers
  constexpr int NS = TR + 2;               // row slots per level: slot s holds tile row s - 1
  constexpr int KMAX = (NS + NL - 1) / NL; // row slots per loader wave
  static_assert(2 * NS <= 64, "the edge scalars of a level are one dword per lane");
  __shared__ v4f srow[NB][NS][64];
  __shared__ float sedge[NB][64]; // [buffer][2 * slot + (west | east)]
  __shared__ unsigned int sbad[2];
  if (CHECK && threadIdx.x < 2)
    sbad[threadIdx.x] = 0; // ordered before the first use by the barrier of the first level
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int bid = blockIdx.x;
  const int seq = (bid & 7) * P.per_xcd + (bid >> 3);
  if (seq >= P.n_logical)
    return;
  // unit = (level chunk, row block, column segment), column segment fastest
  const int ntiles = P.uB * P.uW;
  const int lchunk = seq / ntiles;
  const int tile = seq - lchunk * ntiles;
  const int rblock = tile / P.uW;
  const int wc = tile - rblock * P.uW;
  const int lev0 = lchunk * P.wpb; // wpb: levels per chunk
  const int lev1 = (lev0 + P.wpb < P.nlev) ? lev0 + P.wpb : P.nlev;
  const int nx = P.nx, ny = P.ny;
  const int first = 1 + rblock * TR; // rows 1 .. ny-2 are computed
  const int col = wc * 256 + lane * 4;
  const bool act = col < nx;
  const int col_c = act ? col : (RAGGED ? 0 : nx - 4);
  const int nvalid = RAGGED ? ((nx - col) < 4 ? (nx - col) : 4) : 4; // cells of this lane's group inside the row (<= 0: none)
  int east_col = wc * 256 + 256;
  if (east_col > nx)
    east_col = nx;

  if (wave >= TR) {
    // ------------------------------------------------------------------ loader
    const int lw = wave - TR;
    int off[KMAX], slot_of[KMAX];
    int off_end[KMAX]; // RAGGED: the same with the group that would reach past the end of the level pulled back inside it
    bool fix[KMAX];
    const long last_idx = (long)nx * ny - 1;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      int s = lw + NL * k;
      if (s > NS - 1)
        s = NS - 1; // a wave with fewer rows repeats the last slot (same data to the same place)
      const int jr = first + s - 1;
      const int j = jr < ny - 1 ? jr : ny - 1; // rows past the field: the last row, never used
      slot_of[k] = s;
      off[k] = j * nx + col_c; // offsets inside a level fit 32 bits (the launcher checks)
      fix[k] = RAGGED && ((long)off[k] + 3 > last_idx);
      off_end[k] = fix[k] ? (int)last_idx - 3 : off[k];
    }
    // edge scalars: lane i gathers (slot i / 2, side i % 2); lanes past 2 NS repeat lane 0's
    const int ei = (lane < 2 * NS) ? lane : 0;
    const int ejr = first + (ei >> 1) - 1;
    const int ej = ejr < ny - 1 ? ejr : ny - 1;
    long e64 = (long)ej * nx + ((ei & 1) ? east_col : (wc * 256 - 1));
    const long idx_hi = (long)nx * ny - 1;
    e64 = e64 < 0 ? 0 : (e64 > idx_hi ? idx_hi : e64);
    const int eoff = (int)e64;

    // The wait count of the loop is an immediate, so a loader that also gathers the edge scalars (loader 0 of the
    // operators that read x-neighbours) runs its own copy of the loop.
    auto walk = [&](auto edge_tag) __attribute__((always_inline)) {
      constexpr bool EDGE = decltype(edge_tag)::value;
      constexpr int L = KMAX + (EDGE ? 1 : 0); // load instructions per level of this wave
      auto issue = [&](int lev, int b) {
        const int l = lev < lev1 ? lev : lev1 - 1; // past the chunk: a valid address into a buffer nobody reads
        const float* __restrict__ f = P.f + (size_t)l * P.in_stride;
        const bool at_end = RAGGED && l == P.nlev - 1; // no next level for a partial group to read into
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(f + (at_end ? off_end[k] : off[k])),
                                           (void __attribute__((address_space(3)))*)&srow[b][slot_of[k]][0], 16, 0, 0);
        if (EDGE)
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(f + eoff), (void __attribute__((address_space(3)))*)&sedge[b][0],
                                           4, 0, 0);
        if constexpr (RAGGED) {
          bool mine = false;
#pragma unroll
          for (int k = 0; k < KMAX; ++k)
            mine = mine | fix[k];
          if (at_end && __builtin_amdgcn_ballot_w64(mine) != 0) { // one workgroup of the launch, its last level
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the pulled-back copy has landed: it is overwritten now
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
              if (fix[k]) {
                v4f q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  const long i = (long)off[k] + c;
                  if (i <= last_idx)
                    q[c] = f[i];
                }
                srow[b][slot_of[k]][lane] = q;
              }
            }
          }
        }
      };
#pragma unroll
      for (int k = 0; k < PF; ++k)
        issue(lev0 + k, k);
      int b_next = PF % NB; // buffer of level lev + PF
      // the last loader hands a level's undefined count to the global counter one level late: it idles between its loads
      // and the next barrier anyway.  Its atomic is one more entry in this wave's vmcnt queue, younger than the loads the
      // next wait is about: that wait can only get stricter.
      auto hand_over = [&](int lev_done) {
        if (CHECK && lw == NL - 1 && P.n_undefined && lane == 0) {
          const int q = (lev_done - lev0) & 1;
          const unsigned int n = sbad[q];
          if (P.partials) { // big levels: a plain store per workgroup and level, added up behind the launch (StencilParams::partials)
            P.partials[(size_t)lev_done * (size_t)ntiles + tile] = n;
            if (n != 0)
              sbad[q] = 0;
          } else if (n != 0) {
            atomicAdd(P.n_undefined + lev_done, (u64)n);
            sbad[q] = 0; // the next adds into this slot come after the next barrier, which this wave reaches with lgkmcnt(0)
          }
        }
      };
      for (int lev = lev0; lev < lev1; ++lev) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((PF - 1) * L) : "memory"); // level lev has landed
        issue(lev + PF, b_next); // into the buffer of level lev - 1, whose readers all passed the barrier above
        b_next = (b_next + 1 == NB) ? 0 : b_next + 1;
        if (lev > lev0)
          hand_over(lev - 1); // complete since the barrier above
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // nothing may land in LDS after the workgroup has gone
      if (CHECK) {
        asm volatile("s_barrier" ::: "memory"); // the compute waves' last barrier
        if (lev1 > lev0)
          hand_over(lev1 - 1);
      }
    };
    if (NEED_X && lw == 0)
      walk(std::true_type());
    else
      walk(std::false_type());
    return;
  }

  // -------------------------------------------------------------------- compute
  const int slot = wave + 1;
  const int j_raw = first + wave;
  const bool computes = j_raw <= ny - 2;
  const int j = computes ? j_raw : ny - 2;
  const float undef = P.undef;
  const int base = j * nx;
  const int o = base + col_c;
  const int oo = base + col;
  const bool top = j == 1, bottom = j == ny - 2;
  const bool last_in_seg = col + 4 >= east_col;
  // the tile's map factors and Coriolis parameter: once, for every level of the chunk -- and with them everything of the
  // point formula that does not depend on the level (ScalarHoist, mifc_scalar_cell.h)
  ScalarHoist<OP> H;
  {
    v4f xm4 = {1.f, 1.f, 1.f, 1.f}, ym4 = xm4, fc4 = xm4;
    if (computes) {
      if constexpr (RAGGED) { // dword-aligned rows: unaligned 16-byte loads (a partial group reads into the next row, which exists)
        if (USE_XM)
          xm4 = reinterpret_cast<const V4Any*>(P.xm + o)->v;
        if (USE_YM)
          ym4 = reinterpret_cast<const V4Any*>(P.ym + o)->v;
        if (USE_FC)
          fc4 = reinterpret_cast<const V4Any*>(P.fc + o)->v;
      } else {
        if (USE_XM)
          xm4 = ld4(P.xm + o);
        if (USE_YM)
          ym4 = ld4(P.ym + o);
        if (USE_FC)
          fc4 = ld4(P.fc + o);
      }
    }
    H.init(xm4, ym4, fc4);
  }
  // the one-sided gradients: one multiplication per partial where every lane's halved map factors are exact (wave-uniform,
  // decided once per chunk; scalar_cell_hoisted), the general half_prod otherwise -- two copies of the walk
  constexpr bool USES_HALF = (OP == ST_GRAD_X || OP == ST_GRAD_Y || OP == ST_GRAD_ABS);
  const bool halves = USES_HALF && __builtin_amdgcn_ballot_w64(!H.halves_exact) == 0;
  auto walk_levels = [&](auto halves_tag) __attribute__((always_inline)) {
  constexpr bool HALVES = decltype(halves_tag)::value;
  int buf = 0;
  for (int lev = lev0; lev < lev1; ++lev) {
    // the LDS reads of the previous level are consumed (their values went into the stores); stores stay in flight.  The
    // level's input flag comes through the scalar cache with the barrier's own wait: a vector load of it would sit in this
    // wave's vmcnt queue behind all those stores (level_flag_then_barrier, mifc_device.h)
    bool all = true;
    if (CHECK)
      all = level_flag_then_barrier(P.all_defined, lev);
    else
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (computes) {
      const v4f fcur = srow[buf][slot][lane];
      const v4f fn = srow[buf][slot + 1][lane], fs = srow[buf][slot - 1][lane];
      float fW = fcur.x, fE = fcur.w;
      if (NEED_X) {
        const float We = sedge[buf][2 * slot], Ee = sedge[buf][2 * slot + 1];
        fW = dpp_lower(We, fcur.w); // lane 0 keeps the west scalar
        fE = dpp_upper(Ee, fcur.x); // lane 63 keeps the east scalar
        if (last_in_seg)
          fE = Ee;
      }
      const float fc6[6] = {fW, fcur.x, fcur.y, fcur.z, fcur.w, fE};
      float z0[4], z1[4];
      unsigned int bad = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float r0 = undef, r1 = undef;
        const bool ok = scalar_cell_hoisted<OP, CHECK, HALVES>(all, undef, fc6[k], fc6[k + 1], fc6[k + 2], fs[k], fn[k], H, k, r0, r1);
        z0[k] = ok ? r0 : undef;
        z1[k] = ok ? r1 : undef;
        if (CHECK)
          bad += (!ok & act & (k < nvalid)) ? 1u : 0u;
      }
      if (col == 0) { // fillEdges, column part (:65-68)
        z0[0] = z0[1];
        z1[0] = z1[1];
      }
      if constexpr (RAGGED) { // ... for any width: the value of column nx-2 may sit in the lane below
        const int k_last = nx - 1 - col; // 0 .. 3 in the lane that holds column nx-1
        const float below0 = dpp_lower(z0[3], z0[3]), below1 = dpp_lower(z1[3], z1[3]); // every lane of the wave is here
        if (k_last == 0) {
          z0[0] = below0;
          z1[0] = below1;
        } else if (k_last == 1) {
          z0[1] = z0[0];
          z1[1] = z1[0];
        } else if (k_last == 2) {
          z0[2] = z0[1];
          z1[2] = z1[1];
        } else if (k_last == 3) {
          z0[3] = z0[2];
          z1[3] = z1[2];
        }
      } else if (col + 4 == nx) {
        z0[3] = z0[2];
        z1[3] = z1[2];
      }
      if (act) {
        float* o0p = P.o0 + (size_t)lev * P.out_stride;
        if constexpr (RAGGED) {
          st4_any_alignment(o0p + oo, z0, nvalid);
          if (top) // fillEdges, row part (:70-73)
            st4_any_alignment(o0p + oo - nx, z0, nvalid);
          if (bottom)
            st4_any_alignment(o0p + oo + nx, z0, nvalid);
          if (TWO_OUT) {
            float* o1p = P.o1 + (size_t)lev * P.out_stride;
            st4_any_alignment(o1p + oo, z1, nvalid);
            if (top)
              st4_any_alignment(o1p + oo - nx, z1, nvalid);
            if (bottom)
              st4_any_alignment(o1p + oo + nx, z1, nvalid);
          }
        } else {
          st4_stream(o0p + oo, z0);
          if (top) // fillEdges, row part (:70-73)
            st4_stream(o0p + oo - nx, z0);
          if (bottom)
            st4_stream(o0p + oo + nx, z0);
          if (TWO_OUT) {
            float* o1p = P.o1 + (size_t)lev * P.out_stride;
            st4_stream(o1p + oo, z1);
            if (top)
              st4_stream(o1p + oo - nx, z1);
            if (bottom)
              st4_stream(o1p + oo + nx, z1);
          }
        }
      }
      if (CHECK && P.n_undefined && !all && __builtin_amdgcn_ballot_w64(bad != 0) != 0) {
        const unsigned int n = wave_sum(bad);
        if (lane == 0)
          atomicAdd(&sbad[(lev - lev0) & 1], n);
      }
    }
    buf = (buf + 1 == NB) ? 0 : buf + 1;
  }
  };
  if (USES_HALF && halves)
    walk_levels(std::true_type());
  else
    walk_levels(std::false_type());
  if (CHECK) // the last level's adds are complete: the last loader hands its total over
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- advection (FieldCalculations.cc:1942-1983) in the same form: a 5-point stencil on f with the wind at the centre.
// The loaders bring the TR + 2 rows of f AND the tile's TR rows of u and of v into LDS (38 one-KiB loads per level for
// TR = 12, 19 per loader wave); the compute waves read LDS and store -- 16 B per cell of HBM traffic, the map factors once
// per chunk of levels (the one-shot kernel of mifc_advection.hip reads them with every level: 24 B per cell, 64 % of
// peak on a 137-level batch).  advec = (u * 0.5 * xm * (e - w) + v * 0.5 * ym * (n - s)) * scale in double, left to right
// (:1972); 0.5 * xm per cell once per chunk: u * (0.5 * xm) and (u * 0.5) * xm are the same exact product.
// RAGGED: rows at any alignment, as in scalar_split_kernel.
template <bool CHECK, int TR, int NL, int PF, bool RAGGED = false>
__global__ __launch_bounds__(64 * (TR + NL)) void advection_split_kernel(const SRowsParams P)
{
  constexpr int NB = PF + 1;               // level buffers
  constexpr int NS = TR + 2;               // row slots of f per level: slot s holds tile row s - 1
  constexpr int NE = NS + 2 * TR;          // ... followed by (u, v) of tile row r at NS + 2 r, NS + 2 r + 1
  constexpr int KMAX = (NE + NL - 1) / NL; // entries per loader wave
  static_assert(2 * NS <= 64, "the edge scalars of a level are one dword per lane");
  __shared__ v4f srow[NB][NE][64];
  __shared__ float sedge[NB][64]; // [buffer][2 * slot + (west | east)]
  __shared__ unsigned int sbad[2];
  if (CHECK && threadIdx.x < 2)
    sbad[threadIdx.x] = 0; // ordered before the first use by the barrier of the first level
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int bid = blockIdx.x;
  const int seq = (bid & 7) * P.per_xcd + (bid >> 3);
  if (seq >= P.n_logical)
    return;
  const int ntiles = P.uB * P.uW;
  const int lchunk = seq / ntiles;
  const int tile = seq - lchunk * ntiles;
  const int rblock = tile / P.uW;
  const int wc = tile - rblock * P.uW;
  const int lev0 = lchunk * P.wpb; // wpb: levels per chunk
  const int lev1 = (lev0 + P.wpb < P.nlev) ? lev0 + P.wpb : P.nlev;
  const int nx = P.nx, ny = P.ny;
  const int first = 1 + rblock * TR; // rows 1 .. ny-2 are computed
  const int col = wc * 256 + lane * 4;
  const bool act = col < nx;
  const int col_c = act ? col : (RAGGED ? 0 : nx - 4);
  const int nvalid = RAGGED ? ((nx - col) < 4 ? (nx - col) : 4) : 4; // cells of this lane's group inside the row (<= 0: none)
  int east_col = wc * 256 + 256;
  if (east_col > nx)
    east_col = nx;

  if (wave >= TR) {
    // ------------------------------------------------------------------ loader
    const int lw = wave - TR;
    int off[KMAX], entry_of[KMAX];
    int off_end[KMAX]; // RAGGED: the same with the group that would reach past the end of the level pulled back inside it
    bool fix[KMAX];
    const long last_idx = (long)nx * ny - 1;
    const float* src[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      int e = lw + NL * k;
      if (e > NE - 1)
        e = NE - 1; // a wave with fewer entries repeats the last one (same data to the same place)
      const int w = e - NS; // >= 0: a wind row
      const int jr = (w < 0) ? first + e - 1 : first + (w >> 1);
      const int j = jr < ny - 1 ? jr : ny - 1; // rows past the field: the last row, never used
      entry_of[k] = e;
      src[k] = (w < 0) ? P.f : ((w & 1) ? P.g2 : P.g1);
      off[k] = j * nx + col_c; // offsets inside a level fit 32 bits (the launcher checks)
      fix[k] = RAGGED && ((long)off[k] + 3 > last_idx);
      off_end[k] = fix[k] ? (int)last_idx - 3 : off[k];
    }
    const int ei = (lane < 2 * NS) ? lane : 0;
    const int ejr = first + (ei >> 1) - 1;
    const int ej = ejr < ny - 1 ? ejr : ny - 1;
    long e64 = (long)ej * nx + ((ei & 1) ? east_col : (wc * 256 - 1));
    const long idx_hi = (long)nx * ny - 1;
    e64 = e64 < 0 ? 0 : (e64 > idx_hi ? idx_hi : e64);
    const int eoff = (int)e64;
    auto walk = [&](auto edge_tag) __attribute__((always_inline)) {
      constexpr bool EDGE = decltype(edge_tag)::value;
      constexpr int L = KMAX + (EDGE ? 1 : 0); // load instructions per level of this wave
      auto issue = [&](int lev, int b) {
        const int l = lev < lev1 ? lev : lev1 - 1; // past the chunk: a valid address into a buffer nobody reads
        const size_t lo = (size_t)l * P.in_stride;
        const bool at_end = RAGGED && l == P.nlev - 1; // no next level for a partial group to read into
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src[k] + lo + (at_end ? off_end[k] : off[k])),
                                           (void __attribute__((address_space(3)))*)&srow[b][entry_of[k]][0], 16, 0, 0);
        if (EDGE)
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(P.f + lo + eoff), (void __attribute__((address_space(3)))*)&sedge[b][0],
                                           4, 0, 0);
        if constexpr (RAGGED) {
          bool mine = false;
#pragma unroll
          for (int k = 0; k < KMAX; ++k)
            mine = mine | fix[k];
          if (at_end && __builtin_amdgcn_ballot_w64(mine) != 0) { // one workgroup of the launch, its last level
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the pulled-back copies have landed: they are overwritten now
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
              if (fix[k]) {
                v4f q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  const long i = (long)off[k] + c;
                  if (i <= last_idx)
                    q[c] = (src[k] + lo)[i];
                }
                srow[b][entry_of[k]][lane] = q;
              }
            }
          }
        }
      };
#pragma unroll
      for (int k = 0; k < PF; ++k)
        issue(lev0 + k, k);
      int b_next = PF % NB;
      auto hand_over = [&](int lev_done) { // see scalar_split_kernel
        if (CHECK && lw == NL - 1 && P.n_undefined && lane == 0) {
          const int q = (lev_done - lev0) & 1;
          const unsigned int n = sbad[q];
          if (P.partials) {
            P.partials[(size_t)lev_done * (size_t)ntiles + tile] = n;
            if (n != 0)
              sbad[q] = 0;
          } else if (n != 0) {
            atomicAdd(P.n_undefined + lev_done, (u64)n);
            sbad[q] = 0;
          }
        }
      };
      for (int lev = lev0; lev < lev1; ++lev) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((PF - 1) * L) : "memory"); // level lev has landed
        issue(lev + PF, b_next);
        b_next = (b_next + 1 == NB) ? 0 : b_next + 1;
        if (lev > lev0)
          hand_over(lev - 1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // nothing may land in LDS after the workgroup has gone
      if (CHECK) {
        asm volatile("s_barrier" ::: "memory"); // the compute waves' last barrier
        if (lev1 > lev0)
          hand_over(lev1 - 1);
      }
    };
    if (lw == 0)
      walk(std::true_type());
    else
      walk(std::false_type());
    return;
  }

  // -------------------------------------------------------------------- compute
  const int slot = wave + 1;
  const int j_raw = first + wave;
  const bool computes = j_raw <= ny - 2;
  const int j = computes ? j_raw : ny - 2;
  const float undef = P.undef;
  const int base = j * nx;
  const int o = base + col_c;
  const int oo = base + col;
  const bool top = j == 1, bottom = j == ny - 2;
  const bool last_in_seg = col + 4 >= east_col;
  double a[4] = {0., 0., 0., 0.}, b[4] = {0., 0., 0., 0.};
  if (computes) {
    v4f xm4, ym4;
    if constexpr (RAGGED) { // (a partial group reads into the next row, which exists: rows 1 .. ny-2 are computed)
      xm4 = reinterpret_cast<const V4Any*>(P.xm + o)->v;
      ym4 = reinterpret_cast<const V4Any*>(P.ym + o)->v;
    } else {
      xm4 = ld4(P.xm + o);
      ym4 = ld4(P.ym + o);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a[k] = 0.5 * (double)xm4[k];
      b[k] = 0.5 * (double)ym4[k];
    }
  }
  const double scale = (double)P.scale;
  int buf = 0;
  for (int lev = lev0; lev < lev1; ++lev) {
    bool all = true;
    if (CHECK)
      all = level_flag_then_barrier(P.all_defined, lev);
    else
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (computes) {
      const v4f fcur = srow[buf][slot][lane];
      const v4f fn = srow[buf][slot + 1][lane], fs = srow[buf][slot - 1][lane];
      const v4f u4 = srow[buf][NS + 2 * wave][lane], v4 = srow[buf][NS + 2 * wave + 1][lane];
      const float We = sedge[buf][2 * slot], Ee = sedge[buf][2 * slot + 1];
      const float fW = dpp_lower(We, fcur.w); // lane 0 keeps the west scalar
      float fE = dpp_upper(Ee, fcur.x);       // lane 63 keeps the east scalar
      if (last_in_seg)
        fE = Ee;
      const float fc6[6] = {fW, fcur.x, fcur.y, fcur.z, fcur.w, fE};
      float z0[4];
      unsigned int bad = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float w = fc6[k], e = fc6[k + 2], s = fs[k], n = fn[k], uc = u4[k], vc = v4[k];
        bool ok = true;
        if (CHECK) // :1971
          ok = all | all_lg(undef, uc, vc, s, w, e, n);
        const float r = (float)(((double)uc * a[k] * (double)(e - w) + (double)vc * b[k] * (double)(n - s)) * scale); // :1972
        z0[k] = ok ? r : undef;
        if (CHECK)
          bad += (!ok & act & (k < nvalid)) ? 1u : 0u;
      }
      if (col == 0) // fillEdges, column part (:65-68)
        z0[0] = z0[1];
      if constexpr (RAGGED) { // ... for any width: the value of column nx-2 may sit in the lane below
        const int k_last = nx - 1 - col; // 0 .. 3 in the lane that holds column nx-1
        const float below0 = dpp_lower(z0[3], z0[3]); // every lane of the wave is here
        if (k_last == 0)
          z0[0] = below0;
        else if (k_last == 1)
          z0[1] = z0[0];
        else if (k_last == 2)
          z0[2] = z0[1];
        else if (k_last == 3)
          z0[3] = z0[2];
      } else if (col + 4 == nx) {
        z0[3] = z0[2];
      }
      if (act) {
        float* o0p = P.o0 + (size_t)lev * P.out_stride;
        if constexpr (RAGGED) {
          st4_any_alignment(o0p + oo, z0, nvalid);
          if (top) // fillEdges, row part (:70-73)
            st4_any_alignment(o0p + oo - nx, z0, nvalid);
          if (bottom)
            st4_any_alignment(o0p + oo + nx, z0, nvalid);
        } else {
          st4_stream(o0p + oo, z0);
          if (top) // fillEdges, row part (:70-73)
            st4_stream(o0p + oo - nx, z0);
          if (bottom)
            st4_stream(o0p + oo + nx, z0);
        }
      }
      if (CHECK && P.n_undefined && !all && __builtin_amdgcn_ballot_w64(bad != 0) != 0) {
        const unsigned int n = wave_sum(bad);
        if (lane == 0)
          atomicAdd(&sbad[(lev - lev0) & 1], n);
      }
    }
    buf = (buf + 1 == NB) ? 0 : buf + 1;
  }
  if (CHECK) // the last level's adds are complete: the last loader hands its total over
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct SplitShape
{
  int tr, nl, pf, lg; // tile rows, loader waves, levels the loaders run ahead, levels per chunk (0: chosen from the batch)
};

int shape_value(const char* s, const char* key, int dflt)
{
  const size_t n = std::strlen(key);
  for (const char* p = s; p && *p;) {
    if (std::strncmp(p, key, n) == 0 && p[n] == '=')
      return std::atoi(p + n + 1);
    p = std::strchr(p, ',');
    if (p)
      ++p;
  }
  return dflt;
}

// Two shapes, chosen per operator and variant (profiles/r03/split_role_ops.txt, wave_placement.txt, sq_counters_split_shapes.txt):
//   * {12 rows, 2 loaders, 3 levels ahead}: 14-wave workgroups.  Two of them would fit a CU by every resource, but the
//     hardware keeps ONE resident (SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE is 46 % of the 16-wave shapes'; workgroups of 13, 14 and
//     15 waves all behave so, 12 and 16 do not) -- which suits the operators bound by memory: one workgroup per CU streaming
//     three levels ahead is the access pattern of the headline kernel, 2-5 % faster than two workgroups per CU.
//   * {14 rows, 2 loaders, 2 levels ahead}: 16-wave workgroups, two per CU, for the variants bound by instruction issue --
//     |grad f| (a correctly rounded square root), the Laplacian, plevelgvort (fp64 chains) and everything that tests its
//     inputs but d/dy, whose arithmetic phases then overlap (the 14-wave shape runs the tested ones 15-25 % slower;
//     profiles/r03/split_role_ops_placed.txt has both shapes for every operator on one set of placed arrays).
// MIFC_SCALAR_SPLIT_TUNE="TR=12,NL=2,PF=2,LG=6" overrides (A/B measurements and tests; only the shapes instantiated below exist).
SplitShape current_shape(int op, bool check, bool ragged = false)
{
  const bool issue_bound = op == ST_GRAD_ABS || op == ST_GRAD_LAP || op == ST_GVORT || (check && op != ST_GRAD_Y);
  SplitShape sh = issue_bound ? SplitShape{14, 2, 2, 0} : SplitShape{12, 2, 3, 0};
  const char* s = env().scalar_split_tune;
  if (s[0]) {
    sh.tr = shape_value(s, "TR", sh.tr);
    sh.nl = shape_value(s, "NL", sh.nl);
    sh.pf = shape_value(s, "PF", sh.pf);
    sh.lg = shape_value(s, "LG", sh.lg);
  }
  if (ragged) // the variant for rows at any alignment exists in the two default shapes
    sh = (sh.tr == 14) ? SplitShape{14, 2, 2, sh.lg} : SplitShape{12, 2, 3, sh.lg};
  return sh;
}

template <int OP, int TR, int NL, int PF>
void launch_shape(const SRowsParams& rp, bool check, int grid, hipStream_t stream)
{
  if (check)
    hipLaunchKernelGGL((scalar_split_kernel<OP, true, TR, NL, PF>), dim3(grid), dim3(64 * (TR + NL)), 0, stream, rp);
  else
    hipLaunchKernelGGL((scalar_split_kernel<OP, false, TR, NL, PF>), dim3(grid), dim3(64 * (TR + NL)), 0, stream, rp);
}

// rows at any alignment: the two default shapes only
template <int OP>
void launch_ragged(const SRowsParams& rp, const SplitShape& sh, bool check, int grid, hipStream_t stream)
{
  if (sh.tr == 14) {
    if (check)
      hipLaunchKernelGGL((scalar_split_kernel<OP, true, 14, 2, 2, true>), dim3(grid), dim3(64 * 16), 0, stream, rp);
    else
      hipLaunchKernelGGL((scalar_split_kernel<OP, false, 14, 2, 2, true>), dim3(grid), dim3(64 * 16), 0, stream, rp);
  } else {
    if (check)
      hipLaunchKernelGGL((scalar_split_kernel<OP, true, 12, 2, 3, true>), dim3(grid), dim3(64 * 14), 0, stream, rp);
    else
      hipLaunchKernelGGL((scalar_split_kernel<OP, false, 12, 2, 3, true>), dim3(grid), dim3(64 * 14), 0, stream, rp);
  }
}

// the instantiated shapes: {tile rows, loader waves, levels ahead}
#define MIFC_SPLIT_SHAPES(X) X(12, 2, 3) X(14, 2, 2) X(12, 2, 2) X(12, 4, 2) X(10, 2, 2) X(8, 2, 2) X(12, 2, 1) X(13, 3, 2)

bool shape_exists(const SplitShape& sh)
{
#define X(TR_, NL_, PF_) \
  if (sh.tr == TR_ && sh.nl == NL_ && sh.pf == PF_) \
    return true;
  MIFC_SPLIT_SHAPES(X)
#undef X
  return false;
}

template <int OP>
void launch_op(const SRowsParams& rp, const SplitShape& sh, bool check, int grid, hipStream_t stream)
{
  if (rp.ragged) {
    launch_ragged<OP>(rp, sh, check, grid, stream);
    return;
  }
#define X(TR_, NL_, PF_) \
  if (sh.tr == TR_ && sh.nl == NL_ && sh.pf == PF_) { \
    launch_shape<OP, TR_, NL_, PF_>(rp, check, grid, stream); \
    return; \
  }
  MIFC_SPLIT_SHAPES(X)
#undef X
}

// tiles x level chunks of the launch
bool plan(const SplitShape& sh, int nx, int ny, int nlev, int* levels_per_chunk, long* units)
{
  if (nlev < 3 || (long)nx * ny >= 0x7fffffffL)
    return false;
  const int target = sh.lg > 0 ? sh.lg : (nlev >= 48 ? 6 : 8); // levels per chunk; the chunks are then balanced
  const int nchunks = (nlev + target - 1) / target;
  const int lpc = (nlev + nchunks - 1) / nchunks;
  const long tiles = (long)((ny - 2 + sh.tr - 1) / sh.tr) * ((nx + 255) / 256);
  *levels_per_chunk = lpc;
  *units = tiles * ((nlev + lpc - 1) / lpc);
  return *units <= 0x3fffffffL;
}

} // namespace

bool scalar_split_applies(int op, int nx, int ny, int nlev, bool check, float undef, bool ragged)
{
  if (ragged && !env().ragged_split)
    return false;
  if (!env().split_roles || !env().levelwalk || op < ST_GRAD_X || op > ST_IGWIND)
    return false;
  if (check && undef != undef) // the kernel tests with ONE compare per value, which is is_def() only for an undef that is not NaN
    return false;
  const SplitShape sh = current_shape(op, check, ragged);
  int lpc;
  long units;
  if (!shape_exists(sh) || !plan(sh, nx, ny, nlev, &lpc, &units))
    return false;
  return units >= (env().levelwalk_min_units > 0 ? env().levelwalk_min_units : 768);
}

hipError_t launch_scalar_split(int op, SRowsParams& rp, bool check, hipStream_t stream)
{
  const SplitShape sh = current_shape(op, check, rp.ragged != 0);
  int lpc;
  long units;
  if (!shape_exists(sh) || !plan(sh, rp.nx, rp.ny, rp.nlev, &lpc, &units))
    return hipErrorInvalidValue; // scalar_split_applies() said otherwise
  rp.uB = (rp.ny - 2 + sh.tr - 1) / sh.tr;
  rp.uW = (rp.nx + 255) / 256;
  rp.wpb = lpc;
  rp.n_logical = (int)units;
  rp.per_xcd = (rp.n_logical + 7) / 8;
  const int grid = rp.per_xcd * 8;
  // big levels with tests: counts by plain stores (rp.partials[level][tile], set by the caller where its buffer is large
  // enough), added up by the caller behind this launch
  if (!(check && rp.partials && rp.n_undefined && (long)rp.uB * rp.uW >= 2048 && (long)rp.uB * rp.uW * rp.nlev <= rp.partials_cap))
    rp.partials = nullptr;
  switch (op) {
  case ST_GRAD_X:
    launch_op<ST_GRAD_X>(rp, sh, check, grid, stream);
    break;
  case ST_GRAD_Y:
    launch_op<ST_GRAD_Y>(rp, sh, check, grid, stream);
    break;
  case ST_GRAD_ABS:
    launch_op<ST_GRAD_ABS>(rp, sh, check, grid, stream);
    break;
  case ST_GRAD_LAP:
    launch_op<ST_GRAD_LAP>(rp, sh, check, grid, stream);
    break;
  case ST_GWIND_X:
    launch_op<ST_GWIND_X>(rp, sh, check, grid, stream);
    break;
  case ST_GWIND_Y:
    launch_op<ST_GWIND_Y>(rp, sh, check, grid, stream);
    break;
  case ST_GVORT:
    launch_op<ST_GVORT>(rp, sh, check, grid, stream);
    break;
  default:
    launch_op<ST_IGWIND>(rp, sh, check, grid, stream);
    break;
  }
  return hipGetLastError();
}

hipError_t launch_advection_split(const StencilParams& prm, hipStream_t stream, bool* handled)
{
  *handled = false;
  constexpr int TR = 12, NL = 2, PF = 2;
  const int nx = prm.nx, ny = prm.ny_global;
  if (prm.op != ST_ADVECTION || !env().split_roles || !env().levelwalk || env().force_cell_kernel)
    return hipSuccess;
  if (nx < 8 || ny < 3 || prm.j0 != 0 || prm.ny_local != ny || prm.nlev < 3 || (long)nx * ny >= 0x7fffffffL)
    return hipSuccess;
  if (!prm.f0 || !prm.f1 || !prm.f2 || !prm.xmapr || !prm.ymapr || !prm.out0)
    return hipSuccess;
  const auto a16 = [](const void* p) { return (reinterpret_cast<size_t>(p) & 15u) == 0; };
  const bool ragged = nx % 4 != 0 || !a16(prm.f0) || !a16(prm.f1) || !a16(prm.f2) || !a16(prm.xmapr) || !a16(prm.ymapr) || !a16(prm.out0) ||
                      prm.in_level_stride % 4 != 0 || prm.out_level_stride % 4 != 0;
  // (nx % 256 == 1: the column whose value fillEdges copies into column nx-1 belongs to another workgroup)
  if (ragged && (!env().ragged_split || nx % 256 == 1))
    return hipSuccess;
  const bool check = !prm.every_level_all_defined;
  if (check && prm.undef != prm.undef) // one-compare tests: not for NaN as undef
    return hipSuccess;
  const SplitShape sh = {TR, NL, PF, 0};
  int lpc;
  long units;
  if (!plan(sh, nx, ny, prm.nlev, &lpc, &units) || units < (env().levelwalk_min_units > 0 ? env().levelwalk_min_units : 768))
    return hipSuccess;
  SRowsParams rp{};
  rp.nx = nx;
  rp.ny = ny;
  rp.nlev = prm.nlev;
  rp.f = prm.f0;
  rp.g1 = prm.f1;
  rp.g2 = prm.f2;
  rp.scale = prm.scale;
  rp.xm = prm.xmapr;
  rp.ym = prm.ymapr;
  rp.o0 = prm.out0;
  rp.in_stride = prm.in_level_stride;
  rp.out_stride = prm.out_level_stride;
  rp.all_defined = prm.all_defined;
  rp.undef = prm.undef;
  rp.n_undefined = prm.n_undefined;
  rp.uB = (ny - 2 + TR - 1) / TR;
  rp.uW = (nx + 255) / 256;
  rp.wpb = lpc;
  rp.n_logical = (int)units;
  rp.per_xcd = (rp.n_logical + 7) / 8;
  const int grid = rp.per_xcd * 8;
  *handled = true;
  note_form("advection_split");
  const long tiles = (long)rp.uB * rp.uW;
  if (check && prm.partials && prm.n_undefined && tiles >= 2048 && tiles * prm.nlev <= prm.partials_cap)
    rp.partials = prm.partials;
  if (ragged) {
    note_form("advection_split_ragged");
    if (check)
      hipLaunchKernelGGL((advection_split_kernel<true, TR, NL, PF, true>), dim3(grid), dim3(64 * (TR + NL)), 0, stream, rp);
    else
      hipLaunchKernelGGL((advection_split_kernel<false, TR, NL, PF, true>), dim3(grid), dim3(64 * (TR + NL)), 0, stream, rp);
  } else if (check)
    hipLaunchKernelGGL((advection_split_kernel<true, TR, NL, PF>), dim3(grid), dim3(64 * (TR + NL)), 0, stream, rp);
  else
    hipLaunchKernelGGL((advection_split_kernel<false, TR, NL, PF>), dim3(grid), dim3(64 * (TR + NL)), 0, stream, rp);
  if (rp.partials)
    (void)launch_count_partials_levels(rp.partials, (int)tiles, prm.nlev, prm.n_undefined, stream);
  return hipGetLastError();
}

} // namespace mifc
