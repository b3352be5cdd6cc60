// mifc_vortdiv.hip -- fused relative vorticity + divergence, the headline
// kernel (BASELINE.json: 1440x720x137 float32, HBM-bound, 16 B/cell).
//
// Restates relvort (FieldCalculations.cc:1843-1873) and divergence
// (:1910-1940) for a whole batch of levels in one pass; results per level are
// bit-identical to the two reference calls (double-promoted combine, no fma,
// same undefined test on (v[i-1], v[i+1], u[i-nx], u[i+nx]) for both outputs,
// flat-loop count including the wrapped edge columns, fillEdges folded into
// the store).
//
// Mapping to the hardware
//   * work unit = one wavefront = 64 lanes x V float4 = 256*V columns x R rows
//     of one level.  The wave walks DOWN the rows keeping rows j-1, j, j+1 of u
//     and v in registers (row-sliding window): every u/v value is fetched
//     once per band, halo overhead (R+2)/R on the reads only.
//   * all global accesses are 16 B per lane, 1 KiB per wave-instruction,
//     row-major and coalesced.  The window is a STATIC register ring of
//     W = D+3 rows and the row loop is unrolled over the ring, so no loaded
//     value is ever moved between registers and the compiler keeps D rows in
//     flight with counted s_waitcnt vmcnt(N) (a rotating-variable window forces
//     vmcnt(0) every iteration: moving an in-flight register needs its data).
//     Loads are branch-free: out-of-range lanes/rows read a clamped, valid
//     address and only the STORES are predicated.
//   * x neighbours come from the adjacent lanes with one DPP wave-shift each
//     (v_mov_b32_dpp wave_shr:1 / wave_shl:1) -- no LDS traffic, no barrier in
//     the row loop; lane 0 and lane 63 take one scalar per row from the
//     neighbouring wave-column.  Those two scalars cost two extra 128-byte
//     lines per row, which is why V = 2 (512 columns per wave) is the default:
//     it halves that overhead.
//   * the waves of a workgroup take CONSECUTIVE LEVELS of the same (band,
//     wave-column) tile; xmapr / ymapr of the tile are staged ONCE per
//     workgroup in LDS (they do not depend on the level) and read from there
//     with ds_read_b128 in the row loop.
//   * blockIdx -> tile map is XCD-aware (blocks are dealt round-robin over the
//     8 XCDs; block b runs sequence number (b % 8) * per_xcd + b / 8) and walks
//     the batch in address order (level group, band, wave-column), so that the
//     halo rows and edge lines a tile shares with its neighbours are fetched
//     by the same XCD at about the same time.
//   * undefined cells are counted per lane, reduced per wave (butterfly) and
//     added with ONE atomic per wave to n_undefined[level]; the all-defined
//     instantiation contains no test and no atomic at all.
//   * no MFMA: 4 fp32 subtractions + 16 fp64-pipe ops per cell (6 cvt, 6 mul,
//     2 add, 2 cvt back), the reference's double-promoted combine; measured to
//     be fully hidden behind the memory traffic (a float-only build of the
//     same kernel runs in the same time).
#include "mifc_device.h"
#include "mifc_kernels.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace mifc {

namespace {

struct RowsParams
{
  int nx;
  int nyg;      // rows of the whole field
  int j0;       // global row of owned row 0
  int ny_local; // owned rows
  int lo, hi;   // owned local rows that are computed: [lo, hi)
  int R;        // rows per band
  int nbands, nwc, nlev;
  int wpb;             // waves per workgroup (4, 8 or 16): that many levels side by side
  int uL, uB, uW;      // workgroup-unit counts along (level groups, bands, wave-columns)
  int n_logical;       // uL * uB * uW
  int per_xcd;         // ceil(n_logical / 8)
  int order;           // block sequence -> unit order, see decode_block()
  int xcd_remap;       // 1: sequence = (b % 8) * per_xcd + b / 8
  int zigzag;          // 1: odd bands walk upwards
  int nt_interior;     // 1: rows no other band touches are loaded nontemporally
  int lgroup;          // one-shot forms: > 0 = consecutive units are the SAME tile on `lgroup` consecutive levels
                       // (the tile's map factors then come from L2); 0 = address order
#ifdef MIFC_MEASUREMENT_BUILD // libmifc_measure.so only (tools/): knobs that give wrong results by design
  int exp_nohalo;      // halo rows are not fetched, to price their traffic
  int exp_nostore;     // 1 = only a sliver of the stores is issued (read side alone)
  int exp_noload;      // the field rows are not loaded at all (write side alone)
  int exp_store_aux;   // != 0: results leave through buffer stores with this cache policy (2 nt, 16 sc1, 17 sc0 sc1, 18 nt sc1)
#endif
  long idx_lo, idx_hi; // valid flat element range relative to owned row 0 (for clamped scalar loads)
  const float *u, *v, *xm, *ym;
  const float* fc; // coriolis parameter, absvort only
  float *rv, *dv;
  float* ff;          // split-role kernel only: wind speed sqrt(u*u + v*v) of every cell as a third output (vectorabs, :1819), or null
  u64* n_undefined_ff; // its per-level undefined counts (count domain nx*ny, :1839)
  long in_stride, out_stride;
  const unsigned char* all_defined;
  float undef;
  u64* n_undefined;
  unsigned int* partials; // one-shot tiles of one or two big levels: the workgroup's count goes to partials[unit] instead (see StencilParams)
};

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float dpp_from_lower_lane(float keep_if_none, float x)
{
  // lane i <- lane i-1; lane 0 keeps `keep_if_none`
  return __builtin_bit_cast(float,
                            __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep_if_none), __builtin_bit_cast(int, x), 0x138 /*wave_shr:1*/, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_upper_lane(float keep_if_none, float x)
{
  // lane i <- lane i+1; lane 63 keeps `keep_if_none`
  return __builtin_bit_cast(float,
                            __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep_if_none), __builtin_bit_cast(int, x), 0x130 /*wave_shl:1*/, 0xf, 0xf, false));
}

// NB: take the float BY VALUE.  __builtin_bit_cast applied directly to a vector
// element expression (x.w) reads element 0 with this compiler.
__device__ __forceinline__ float readlane_f(float x, int src_lane)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src_lane));
}

template <bool NT>
__device__ __forceinline__ void store4(float* p, const v4f& v)
{
  if (NT)
    __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
  else
    *reinterpret_cast<v4f*>(p) = v;
}

// Rows of a field whose width is not a multiple of 4 start at dword-aligned addresses only.  16-byte stores there run at
// 75 % of the aligned store rate (profiles/r03/experiments/ragged_probe.txt: 3.8 against 5.0 TB/s; four dword stores per
// lane writing contiguous 256-byte runs are no better, nontemporal ones worse), loads -- global_load_lds_dwordx4 included --
// do not care.  A group that reaches over the end of its row stores its valid cells one by one.
struct __attribute__((packed, aligned(4))) V4Unaligned
{
  v4f v;
};
__device__ __forceinline__ void store4_any_alignment(float* p, const v4f& v, int nvalid)
{
  if (nvalid >= 4) {
    V4Unaligned t;
    t.v = v;
    *reinterpret_cast<V4Unaligned*>(p) = t;
  } else {
    if (nvalid > 0)
      p[0] = v.x;
    if (nvalid > 1)
      p[1] = v.y;
    if (nvalid > 2)
      p[2] = v.z;
  }
}

#ifdef MIFC_MEASUREMENT_BUILD
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
// base: wave-uniform start of the level's output field; off: element offset
__device__ __forceinline__ void store4_policy(int aux, float* base, long off, int n_bytes, const v4f& v)
{
  const unsigned long long a = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  void* b = (void*)(((unsigned long long)hi << 32) | lo);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(b, 0, n_bytes, 0x00020000);
  const v4u d = __builtin_bit_cast(v4u, v);
  const int o = (int)(off * 4);
  switch (aux) {
  case 2:
    __builtin_amdgcn_raw_buffer_store_b128(d, rs, o, 0, 2);
    break;
  case 16:
    __builtin_amdgcn_raw_buffer_store_b128(d, rs, o, 0, 16);
    break;
  case 17:
    __builtin_amdgcn_raw_buffer_store_b128(d, rs, o, 0, 17);
    break;
  case 18:
    __builtin_amdgcn_raw_buffer_store_b128(d, rs, o, 0, 18);
    break;
  default:
    __builtin_amdgcn_raw_buffer_store_b128(d, rs, o, 0, 0);
    break;
  }
}
#define MIFC_STORE4(NT_, base_, off_, v_)                                                        \
  do {                                                                                           \
    if (P.exp_store_aux)                                                                         \
      store4_policy(P.exp_store_aux, (base_), (off_), (int)((long)P.ny_local * P.nx * 4), (v_)); \
    else                                                                                         \
      store4<NT_>((base_) + (off_), (v_));                                                       \
  } while (0)
#else
#define MIFC_STORE4(NT_, base_, off_, v_) store4<NT_>((base_) + (off_), (v_))
#endif

__device__ __forceinline__ v4f load4(const float* p)
{
  return *reinterpret_cast<const v4f*>(p);
}

template <int V>
struct RowRegs
{
  v4f u[V], v[V];
  float eu, ev; // lane 63: value east of the wave-column; other lanes: value west of it
};

// sequence number -> workgroup unit (l, b, w) along (level groups, bands, wave-columns)
//   order 1: address order -- level group slowest, then band, wave-column fastest
//   order 0: (wave-column, band) tiles, the level groups of one tile consecutive
//   order 2: (band, wave-column) tiles, the level groups of one tile consecutive
__device__ __forceinline__ void decode_block(const RowsParams& P, int seq, int& l, int& b, int& w)
{
  if (P.order == 1) {
    const int per_level = P.uB * P.uW;
    l = seq / per_level;
    const int rem = seq - l * per_level;
    b = rem / P.uW;
    w = rem - b * P.uW;
  } else {
    const int t = seq / P.uL;
    l = seq - t * P.uL;
    if (P.order == 2) {
      b = t / P.uW;
      w = t - b * P.uW;
    } else {
      w = t / P.uB;
      b = t - w * P.uB;
    }
  }
}

// one-shot forms: sequence number -> (level, row block, column segment)
__device__ __forceinline__ void decode_oneshot(const RowsParams& P, int seq, int& lev, int& rblock, int& wc)
{
  const int per_level = P.uB * P.uW;
  if (P.lgroup > 0) {
    // groups of `lgroup` levels; inside a group the tile index is slow and the level fast
    const int per_group = P.lgroup * per_level;
    const int g = seq / per_group;
    const int r2 = seq - g * per_group;
    const int first = g * P.lgroup;
    const int ng = (P.nlev - first < P.lgroup) ? (P.nlev - first) : P.lgroup; // the last group may be short
    const int tile = r2 / ng;
    lev = first + (r2 - tile * ng);
    rblock = tile / P.uW;
    wc = tile - rblock * P.uW;
    return;
  }
  // address order: column segment fastest, then row block, then level
  lev = seq / per_level;
  const int rem = seq - lev * per_level;
  rblock = rem / P.uW;
  wc = rem - rblock * P.uW;
}

// JAC: the same data pattern (W/E/N/S neighbours of two fields) computes jacobian(field1, field2)
// (FieldCalculations.cc:2424-2460) into the first output instead.
template <bool CHECK, bool WANT_V, bool WANT_D, bool ABSV, int D, bool NT, int V, bool JAC = false>
__global__ __launch_bounds__(512) void vortdiv_rows_kernel(const RowsParams P)
{
  constexpr int W = D + 3;    // ring slots: rows r-2 (being refilled), r-1, r, r+1, r+2 .. r+D
  constexpr int WCOLS = 256 * V; // columns per wave

  // ---- which tile / level does this wave own -----------------------------
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int bid = blockIdx.x;
  const int seq = P.xcd_remap ? ((bid & 7) * P.per_xcd + (bid >> 3)) : bid;
  if (seq >= P.n_logical)
    return; // whole workgroup: nobody reaches the barrier below
  int lgroup, band, wc;
  decode_block(P, seq, lgroup, band, wc);
  // this workgroup: levels [lev0, lev0 + wpb), one per wave
  const int lev0 = lgroup * P.wpb;

  const int nx = P.nx;
  // Lane l owns V separate float4 per row: the q-th one at column
  // wc*WCOLS + q*256 + 4*l, so every load/store instruction of the wave covers
  // ONE contiguous KiB (64 lanes x 16 B).
  int colq[V], colq_c[V];
  bool actq[V];
#pragma unroll
  for (int q = 0; q < V; ++q) {
    colq[q] = wc * WCOLS + q * 256 + lane * 4;
    actq[q] = colq[q] < nx;
    colq_c[q] = actq[q] ? colq[q] : nx - 4; // clamped column for loads
  }
  int east_col = wc * WCOLS + WCOLS; // first column east of this wave-column (may be nx: wraps to the next row)
  if (east_col > nx)
    east_col = nx;
  const int edge_col = (lane == 63) ? east_col : (wc * WCOLS - 1);

  const int jb = P.lo + band * P.R;                     // first local row of the band
  const int nr = (P.hi - jb < P.R) ? (P.hi - jb) : P.R; // rows in this band
  const float undef = P.undef;
  const bool owns_top_edge = (P.j0 == 0);
  const bool owns_bottom_edge = (P.j0 + P.ny_local == P.nyg);

  // the last level group may have fewer levels than waves: such a wave still
  // helps staging the map factors, with its loads pointed at a valid level
  const bool valid = (lev0 + wave) < P.nlev;
  const int lev = valid ? (lev0 + wave) : (P.nlev - 1);
  const bool all = CHECK ? (P.all_defined && P.all_defined[lev] != 0) : true;

  const float* __restrict__ u = P.u + (size_t)lev * P.in_stride;
  const float* __restrict__ v = P.v + (size_t)lev * P.in_stride;
  float* rv = WANT_V ? P.rv + (size_t)lev * P.out_stride : nullptr;
  float* dv = WANT_D ? P.dv + (size_t)lev * P.out_stride : nullptr;

  // Bands alternate their walking direction: even bands go down, odd bands go
  // up.  The halo rows a band shares with its neighbours are then loaded by
  // both at the SAME phase of their walk (both at the start, or both at the
  // end), i.e. at about the same time on the same XCD, and the second fetch
  // hits L2 instead of going back to HBM.
  const bool up = (P.zigzag != 0) && ((band & 1) != 0);
  // step t in [-1, nr] of the walk -> local row; steps past the far halo are clamped to it
  auto load_row = [&](int t) -> RowRegs<V> {
    int tc = t > nr ? nr : t;
#ifdef MIFC_MEASUREMENT_BUILD
    if (P.exp_nohalo)
      tc = tc < 0 ? 0 : (tc > nr - 1 ? nr - 1 : tc);
#endif
    const int rowl = up ? (nr - 1 - tc) : tc;
    const long base = (long)(jb + rowl) * nx;
    const bool stream_row = P.nt_interior && tc >= 1 && tc <= nr - 2;
    RowRegs<V> r;
#ifdef MIFC_MEASUREMENT_BUILD
    if (P.exp_noload) { // write side alone: compute on something that costs no memory traffic
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const float f = (float)(tc + lane);
        const v4f z = {f, f + 1.f, f * 2.f, f + 3.f};
        r.u[q] = z;
        r.v[q] = z;
      }
      r.eu = 0.f;
      r.ev = 0.f;
      return r;
    }
#endif
#pragma unroll
    for (int q = 0; q < V; ++q) {
      if (stream_row) { // interior row of the band: read once by this wave only
        r.u[q] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(u + base + colq_c[q]));
        r.v[q] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(v + base + colq_c[q]));
      } else { // first/last row of the band or a halo row: the neighbouring band reads it too
        r.u[q] = load4(u + base + colq_c[q]);
        r.v[q] = load4(v + base + colq_c[q]);
      }
    }
    // x-neighbour scalar of the wave-column edge.  Only centre rows use it; the
    // clamp keeps the address inside the buffer for the rows that do not.
    long e = base + edge_col;
    e = e < P.idx_lo ? P.idx_lo : (e > P.idx_hi ? P.idx_hi : e);
    r.eu = u[e];
    r.ev = v[e];
    return r;
  };
  // ---- prologue: rows -1 .. D into ring slots (rho + 2) % W; issued before the
  // map-factor staging so that both latencies overlap -------------------------
  RowRegs<V> ring[W];
#pragma unroll
  for (int rho = -1; rho <= D; ++rho)
    ring[(rho + 2) % W] = load_row(rho);

  // ---- map factors of the tile: HBM/L2 -> LDS once per workgroup -------------
  // xmapr/ymapr do not depend on the level: the waves of the workgroup (one
  // level each) all read them from LDS (2*V KiB per tile row, memory order).
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  v4f* lds_xm = reinterpret_cast<v4f*>(lds_raw);
  v4f* lds_ym = lds_xm + P.R * 64 * V;
  v4f* lds_fc = lds_ym + P.R * 64 * V; // only allocated / filled for absvort
  for (int i = threadIdx.x; i < nr * 64 * V; i += blockDim.x) {
    const int row = i / (64 * V);
    int col = wc * WCOLS + (i - row * 64 * V) * 4;
    col = col < nx ? col : nx - 4;
    const long o = (long)(jb + row) * nx + col;
    lds_xm[i] = load4(P.xm + o);
    lds_ym[i] = load4(P.ym + o);
    if (ABSV)
      lds_fc[i] = load4(P.fc + o);
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
      // row r+1+D replaces row r-2, which nobody needs any more
      ring[s % W] = load_row(r + 1 + D);

      // previous / current / next step of the walk.  Walking down, "previous"
      // is row j-1 and "next" row j+1; walking up it is the other way round, so
      // the y-differences are formed in both orders and selected (a - b and
      // -(b - a) would differ in the sign of an exact zero)
      const RowRegs<V>& rp = ring[(s + 1) % W];
      const RowRegs<V>& rc = ring[(s + 2) % W];
      const RowRegs<V>& rn = ring[(s + 3) % W];

      const float east_u = readlane_f(rc.eu, 63);
      const float east_v = readlane_f(rc.ev, 63);
      const int rl = up ? (nr - 1 - r) : r; // local row inside the band
      const int jl = jb + rl;
      const int j = P.j0 + jl;

#pragma unroll
      for (int q = 0; q < V; ++q) {
        // ---- x neighbours of the centre row: adjacent lanes by DPP wave shift;
        // lane 0 / lane 63 continue into the neighbouring 1-KiB segment of this
        // wave (readlane) or take the wave-column edge scalar
        float uW = dpp_from_lower_lane(rc.eu, rc.u[q].w); // lane 0 keeps the west scalar
        float vW = dpp_from_lower_lane(rc.ev, rc.v[q].w);
        float uE = dpp_from_upper_lane(rc.eu, rc.u[q].x); // lane 63 keeps the east scalar
        float vE = dpp_from_upper_lane(rc.ev, rc.v[q].x);
        if (q > 0) { // lane 0 continues into the last lane of the previous segment
          const float tu = readlane_f(rc.u[q > 0 ? q - 1 : 0].w, 63);
          const float tv = readlane_f(rc.v[q > 0 ? q - 1 : 0].w, 63);
          uW = (lane == 0) ? tu : uW;
          vW = (lane == 0) ? tv : vW;
        }
        if (q < V - 1) { // lane 63 continues into the first lane of the next segment
          const float tu = readlane_f(rc.u[q < V - 1 ? q + 1 : q].x, 0);
          const float tv = readlane_f(rc.v[q < V - 1 ? q + 1 : q].x, 0);
          uE = (lane == 63) ? tu : uE;
          vE = (lane == 63) ? tv : vE;
        }
        if (colq[q] + 4 >= east_col) { // my east neighbour is outside the wave-column
          uE = east_u;
          vE = east_v;
        }
        const v4f xm4 = lds_xm[(rl * V + q) * 64 + lane], ym4 = lds_ym[(rl * V + q) * 64 + lane];
        v4f fc4 = xm4;
        if (ABSV)
          fc4 = lds_fc[(rl * V + q) * 64 + lane];
        const float uc[6] = {uW, rc.u[q].x, rc.u[q].y, rc.u[q].z, rc.u[q].w, uE};
        const float vc[6] = {vW, rc.v[q].x, rc.v[q].y, rc.v[q].z, rc.v[q].w, vE};
        float zv[4], zd[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float vw = vc[k], ve = vc[k + 2], uw = uc[k], ue = uc[k + 2];
          const float ua = rp.u[q][k], ub = rn.u[q][k], va = rp.v[q][k], vb = rn.v[q][k]; // rows j-1 / j+1 in walk order
          bool ok = true;
          if (CHECK && !JAC)
            ok = all | all_def(undef, vw, ve, ua, ub); // :1861, :1927
          if (CHECK && JAC) // :2443-2444: all eight neighbours
            ok = all | all_def(undef, uw, ue, ua, ub, vw, ve, va, vb);
          const float dudy = up ? (ua - ub) : (ub - ua); // u[i+nx] - u[i-nx]
          const float dvdy = up ? (va - vb) : (vb - va);
          zv[k] = 0.f;
          zd[k] = 0.f;
          if (JAC) { // :2445-2449: four float-rounded partials, float combination
            const float df1dx = half_prod(xm4[k], ue - uw);
            const float df1dy = half_prod(ym4[k], dudy);
            const float df2dx = half_prod(xm4[k], ve - vw);
            const float df2dy = half_prod(ym4[k], dvdy);
            zv[k] = pick(ok, df1dx * df2dy - df1dy * df2dx, undef);
          } else if (WANT_V)
            zv[k] = pick(ok, ABSV ? f_absvort(xm4[k], ym4[k], ve - vw, dudy, fc4[k]) : f_relvort(xm4[k], ym4[k], ve - vw, dudy), undef);
          if (WANT_D)
            zd[k] = pick(ok, f_diverg(xm4[k], ym4[k], ue - uw, dvdy), undef);
          if (CHECK)
            bad += (!ok & actq[q]) ? 1u : 0u;
        }
        // ---- fillEdges, column part (:65-68), folded into the store ----------
        if (colq[q] == 0) {
          zv[0] = zv[1];
          zd[0] = zd[1];
        }
        if (colq[q] + 4 == nx) {
          zv[3] = zv[2];
          zd[3] = zd[2];
        }
#ifdef MIFC_MEASUREMENT_BUILD
        if (actq[q] && !(P.exp_nostore && zv[0] != 12345.678f)) {
#else
        if (actq[q]) {
#endif
          const long o = (long)jl * nx + colq[q];
          if (WANT_V) {
            v4f z4;
            z4.x = zv[0];
            z4.y = zv[1];
            z4.z = zv[2];
            z4.w = zv[3];
            MIFC_STORE4(NT, rv, o, z4);
            if (j == 1 && owns_top_edge) // row part of fillEdges (:70-73)
              MIFC_STORE4(NT, rv, o - nx, z4);
            if (j == P.nyg - 2 && owns_bottom_edge)
              MIFC_STORE4(NT, rv, o + nx, z4);
          }
          if (WANT_D) {
            v4f d4;
            d4.x = zd[0];
            d4.y = zd[1];
            d4.z = zd[2];
            d4.w = zd[3];
            MIFC_STORE4(NT, dv, o, d4);
            if (j == 1 && owns_top_edge)
              MIFC_STORE4(NT, dv, o - nx, d4);
            if (j == P.nyg - 2 && owns_bottom_edge)
              MIFC_STORE4(NT, dv, o + nx, d4);
          }
        }
      }
    }
  }
level_done:
  if (CHECK && P.n_undefined)
    wave_count_add(P.n_undefined + lev, bad);
}

// ---------------------------------------------------------------------------
// One-shot form of the same operator (tuning K=1): no row loop.  A workgroup is
// 4 waves = 4 consecutive rows x 256 columns of one level; every lane loads the
// float4 of its own row and of the rows above and below (6 loads of 16 B, all
// issued at once), the map factors (2 loads), takes its x neighbours from the
// adjacent lanes (DPP) and stores.  Each value is requested three times, but
// two of the three requests hit in L1 / L2 (the rows are shared inside the
// workgroup and with the next row block on the same XCD); in exchange every
// wave has its whole traffic in flight at once and lives for one memory
// round trip, like a streaming copy.
template <bool CHECK, bool WANT_V, bool WANT_D, bool NT, bool JAC = false, bool ABSV = false>
__global__ __launch_bounds__(256) void vortdiv_oneshot_kernel(const RowsParams P)
{
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int bid = blockIdx.x;
  const int seq = P.xcd_remap ? ((bid & 7) * P.per_xcd + (bid >> 3)) : bid;
  if (seq >= P.n_logical)
    return;
  int lev, rblock, wc;
  decode_oneshot(P, seq, lev, rblock, wc);

  const int nx = P.nx;
  // a wave past the computed range works on the last computed row and keeps the result to itself: it stays for the
  // workgroup-wide count at the end (one atomic per workgroup: mifc_device.h, undefined-cell counting)
  const int jl_raw = P.lo + rblock * 4 + wave; // local row of this wave
  const bool live = jl_raw < P.hi;
  const int jl = live ? jl_raw : P.hi - 1;
  const int j = P.j0 + jl;
  const int col = wc * 256 + lane * 4;
  const bool act = live && col < nx;
  const int col_c = (col < nx) ? col : nx - 4;
  int east_col = wc * 256 + 256;
  if (east_col > nx)
    east_col = nx;
  const int edge_col = (lane == 63) ? east_col : (wc * 256 - 1);
  const float undef = P.undef;
  const bool all = CHECK ? (P.all_defined && P.all_defined[lev] != 0) : true;

  const float* __restrict__ u = P.u + (size_t)lev * P.in_stride;
  const float* __restrict__ v = P.v + (size_t)lev * P.in_stride;
  const long base = (long)jl * nx;
  const long o = base + col_c;
  const v4f uc = load4(u + o), vc = load4(v + o);
  const v4f un = load4(u + o + nx), us = load4(u + o - nx);
  const v4f vn = load4(v + o + nx), vs = load4(v + o - nx);
  const v4f xm4 = load4(P.xm + o), ym4 = load4(P.ym + o);
  v4f fc4 = xm4;
  if (ABSV)
    fc4 = load4(P.fc + o);
  long e = base + edge_col;
  e = e < P.idx_lo ? P.idx_lo : (e > P.idx_hi ? P.idx_hi : e);
  const float eu = u[e], ev = v[e];

  const float east_u = readlane_f(eu, 63), east_v = readlane_f(ev, 63);
  float uW = dpp_from_lower_lane(eu, uc.w), vW = dpp_from_lower_lane(ev, vc.w);
  float uE = dpp_from_upper_lane(eu, uc.x), vE = dpp_from_upper_lane(ev, vc.x);
  if (col + 4 >= east_col) {
    uE = east_u;
    vE = east_v;
  }
  const float ucx[6] = {uW, uc.x, uc.y, uc.z, uc.w, uE};
  const float vcx[6] = {vW, vc.x, vc.y, vc.z, vc.w, vE};
  float zv[4], zd[4];
  unsigned int bad = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float vw = vcx[k], ve = vcx[k + 2], uw = ucx[k], ue = ucx[k + 2];
    bool ok = true;
    if (CHECK && !JAC)
      ok = all | all_def(undef, vw, ve, us[k], un[k]); // :1861, :1927
    if (CHECK && JAC) // :2443-2444: all eight neighbours
      ok = all | all_def(undef, uw, ue, us[k], un[k], vw, ve, vs[k], vn[k]);
    zv[k] = 0.f;
    zd[k] = 0.f;
    if (JAC) { // :2445-2449: four float-rounded partials, float combination (u = field1, v = field2)
      const float df1dx = half_prod(xm4[k], ue - uw);
      const float df1dy = half_prod(ym4[k], un[k] - us[k]);
      const float df2dx = half_prod(xm4[k], ve - vw);
      const float df2dy = half_prod(ym4[k], vn[k] - vs[k]);
      zv[k] = pick(ok, df1dx * df2dy - df1dy * df2dx, undef);
    } else if (WANT_V)
      zv[k] = pick(ok, ABSV ? f_absvort(xm4[k], ym4[k], ve - vw, un[k] - us[k], fc4[k]) : f_relvort(xm4[k], ym4[k], ve - vw, un[k] - us[k]), undef);
    if (WANT_D)
      zd[k] = pick(ok, f_diverg(xm4[k], ym4[k], ue - uw, vn[k] - vs[k]), undef);
    if (CHECK)
      bad += (!ok & act) ? 1u : 0u;
  }
  if (col == 0) { // fillEdges, column part (:65-68)
    zv[0] = zv[1];
    zd[0] = zd[1];
  }
  if (col + 4 == nx) {
    zv[3] = zv[2];
    zd[3] = zd[2];
  }
  if (act) {
    const bool top = (j == 1) && (P.j0 == 0);
    const bool bottom = (j == P.nyg - 2) && (P.j0 + P.ny_local == P.nyg);
    const long oo = base + col;
    if (WANT_V) {
      float* rv = P.rv + (size_t)lev * P.out_stride;
      v4f z4;
      z4.x = zv[0];
      z4.y = zv[1];
      z4.z = zv[2];
      z4.w = zv[3];
      store4<NT>(rv + oo, z4);
      if (top) // fillEdges, row part (:70-73)
        store4<NT>(rv + oo - nx, z4);
      if (bottom)
        store4<NT>(rv + oo + nx, z4);
    }
    if (WANT_D) {
      float* dv = P.dv + (size_t)lev * P.out_stride;
      v4f d4;
      d4.x = zd[0];
      d4.y = zd[1];
      d4.z = zd[2];
      d4.w = zd[3];
      store4<NT>(dv + oo, d4);
      if (top)
        store4<NT>(dv + oo - nx, d4);
      if (bottom)
        store4<NT>(dv + oo + nx, d4);
    }
  }
  if (CHECK)
    block_count_add(P.n_undefined ? P.n_undefined + lev : nullptr, bad); // every wave of the workgroup is on this level
}

// ---------------------------------------------------------------------------
// One-shot form with the row reuse in LDS (tuning K=2): a workgroup is RB+2
// waves = RB+2 consecutive rows x 256 columns of one level.  Every wave loads
// ITS row once (u, v, map factors, one edge scalar), parks u and v in LDS,
// and after one barrier the RB inner waves take the rows above and below from
// there.  Loads per cell: (RB+2)/RB instead of the 3 of K=1, no row loop.
template <bool CHECK, bool WANT_V, bool WANT_D, bool NT, int RB, bool ABSV = false>
__global__ __launch_bounds__(64 * (RB + 2)) void vortdiv_tile_kernel(const RowsParams P)
{
  __shared__ v4f su[RB + 2][64];
  __shared__ v4f sv[RB + 2][64];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int bid = blockIdx.x;
  const int seq = P.xcd_remap ? ((bid & 7) * P.per_xcd + (bid >> 3)) : bid;
  if (seq >= P.n_logical)
    return;
  int lev, rblock, wc;
  decode_oneshot(P, seq, lev, rblock, wc);

  const int nx = P.nx;
  // wave 0 and wave RB+1 hold the halo rows; rows past the computed range are only loaded (clamped to the
  // row after the last computed one, which always exists) for the neighbour below them
  const int jl_raw = P.lo + rblock * RB + wave - 1;
  const int jl = jl_raw > P.hi ? P.hi : jl_raw;
  const bool computes = wave >= 1 && wave <= RB && jl_raw < P.hi;
  const int j = P.j0 + jl;
  const int col = wc * 256 + lane * 4;
  const bool act = col < nx;
  const int col_c = act ? col : nx - 4;
  int east_col = wc * 256 + 256;
  if (east_col > nx)
    east_col = nx;
  const int edge_col = (lane == 63) ? east_col : (wc * 256 - 1);
  const float undef = P.undef;
  const bool all = CHECK ? (P.all_defined && P.all_defined[lev] != 0) : true;

  const float* __restrict__ u = P.u + (size_t)lev * P.in_stride;
  const float* __restrict__ v = P.v + (size_t)lev * P.in_stride;
  const long base = (long)jl * nx;
  const long o = base + col_c;
  const v4f uc = load4(u + o), vc = load4(v + o);
  v4f xm4 = uc, ym4 = uc, fc4 = uc;
  float eu = 0.f, ev = 0.f;
  if (computes) {
    xm4 = load4(P.xm + o);
    ym4 = load4(P.ym + o);
    if (ABSV)
      fc4 = load4(P.fc + o);
    long e = base + edge_col;
    e = e < P.idx_lo ? P.idx_lo : (e > P.idx_hi ? P.idx_hi : e);
    eu = u[e];
    ev = v[e];
  }
  su[wave][lane] = uc;
  sv[wave][lane] = vc;
  // undefined count of the workgroup: the compute waves add theirs into LDS and tick an arrival counter (a wave's LDS
  // operations are in order), the last one to arrive hands the total to the level's counter -- one atomic per workgroup
  __shared__ unsigned int s_cnt[2]; // [0] total, [1] compute waves that have added theirs
  if (CHECK && threadIdx.x < 2)
    s_cnt[threadIdx.x] = 0;
  __syncthreads();
  if (!computes)
    return;
  const v4f un = su[wave + 1][lane], us = su[wave - 1][lane];
  const v4f vn = sv[wave + 1][lane], vs = sv[wave - 1][lane];

  const float east_u = readlane_f(eu, 63), east_v = readlane_f(ev, 63);
  float uW = dpp_from_lower_lane(eu, uc.w), vW = dpp_from_lower_lane(ev, vc.w);
  float uE = dpp_from_upper_lane(eu, uc.x), vE = dpp_from_upper_lane(ev, vc.x);
  if (col + 4 >= east_col) {
    uE = east_u;
    vE = east_v;
  }
  const float ucx[6] = {uW, uc.x, uc.y, uc.z, uc.w, uE};
  const float vcx[6] = {vW, vc.x, vc.y, vc.z, vc.w, vE};
  float zv[4], zd[4];
  unsigned int bad = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float vw = vcx[k], ve = vcx[k + 2], uw = ucx[k], ue = ucx[k + 2];
    bool ok = true;
    if (CHECK)
      ok = all | all_def(undef, vw, ve, us[k], un[k]); // :1861, :1927
    zv[k] = 0.f;
    zd[k] = 0.f;
    if (WANT_V)
      zv[k] = pick(ok, ABSV ? f_absvort(xm4[k], ym4[k], ve - vw, un[k] - us[k], fc4[k]) : f_relvort(xm4[k], ym4[k], ve - vw, un[k] - us[k]), undef);
    if (WANT_D)
      zd[k] = pick(ok, f_diverg(xm4[k], ym4[k], ue - uw, vn[k] - vs[k]), undef);
    if (CHECK)
      bad += (!ok & act) ? 1u : 0u;
  }
  if (col == 0) { // fillEdges, column part (:65-68)
    zv[0] = zv[1];
    zd[0] = zd[1];
  }
  if (col + 4 == nx) {
    zv[3] = zv[2];
    zd[3] = zd[2];
  }
  if (act) {
    const bool top = (j == 1) && (P.j0 == 0);
    const bool bottom = (j == P.nyg - 2) && (P.j0 + P.ny_local == P.nyg);
    const long oo = base + col;
    if (WANT_V) {
      float* rv = P.rv + (size_t)lev * P.out_stride;
      v4f z4;
      z4.x = zv[0];
      z4.y = zv[1];
      z4.z = zv[2];
      z4.w = zv[3];
      store4<NT>(rv + oo, z4);
      if (top) // fillEdges, row part (:70-73)
        store4<NT>(rv + oo - nx, z4);
      if (bottom)
        store4<NT>(rv + oo + nx, z4);
    }
    if (WANT_D) {
      float* dv = P.dv + (size_t)lev * P.out_stride;
      v4f d4;
      d4.x = zd[0];
      d4.y = zd[1];
      d4.z = zd[2];
      d4.w = zd[3];
      store4<NT>(dv + oo, d4);
      if (top)
        store4<NT>(dv + oo - nx, d4);
      if (bottom)
        store4<NT>(dv + oo + nx, d4);
    }
  }
  if (CHECK && P.n_undefined) {
    const unsigned int n = (__builtin_amdgcn_ballot_w64(bad != 0) != 0) ? wave_sum(bad) : 0u;
    if (lane == 0) {
      if (n != 0)
        atomicAdd(&s_cnt[0], n);
      const int rows_left = P.hi - (P.lo + rblock * RB); // compute waves of this workgroup
      const unsigned int ncompute = (unsigned int)(rows_left < RB ? rows_left : RB);
      if (atomicAdd(&s_cnt[1], 1u) + 1u == ncompute) {
        const unsigned int total = atomicAdd(&s_cnt[0], 0u);
        if (P.partials)
          P.partials[seq] = total; // always written: the slots are not zeroed beforehand; seq = level * units per level + unit
        else if (total != 0)
          atomicAdd(P.n_undefined + lev, (u64)total);
      }
    }
  }
}


// ---------------------------------------------------------------------------
// Level-walking form (tuning K=3): a workgroup is NW waves = NW consecutive rows x 256 columns of a
// tile -- and it STAYS on its tile and walks through the levels of the batch.
//   * xmapr / ymapr of the tile do not depend on the level: every wave loads its row of them ONCE,
//     into registers, for all its levels (map factors once per level chunk);
//   * the workgroups of the launch advance level by level together, so at any time the chip works
//     inside a window of a few levels of each array -- the access pattern of a streaming copy that
//     happens to be cut into tiles -- instead of the 16+ levels a row-walking launch has open at once;
//   * per level a wave loads its row of u and v (the loads of level l+PF are issued before level l is
//     computed; static register sets, the loop is unrolled over them so that no in-flight register
//     is ever moved), parks it in one of two LDS buffers, ONE barrier (LDS counter only: prefetches
//     and stores stay in flight across it), takes the rows above and below from LDS, x-neighbours
//     from the adjacent lanes (DPP), stores 2 x 1 KiB;
//   * the first and the last wave also load the halo row above / below the tile: (NW+2)/NW requests at
//     L1, but the workgroup above / below reads the same rows of the same level at about the same
//     time on the same XCD (consecutive units -> one XCD): L2 hits.
//   * HALO = true instead gives the two halo rows waves of their own (the first and the last wave of the
//     workgroup only load): NW - 2 computed rows per tile, but no extra row in anybody's registers.
// Registers: with halo waves and one level of prefetch the kernel fits 64 VGPRs, i.e. two 16-wave (or four
// 8-wave) workgroups per CU -- asked for through the second launch bound (waves per SIMD).
template <bool CHECK, bool WANT_V, bool WANT_D, bool NT, int NW, int PF, bool HALO>
__global__ __launch_bounds__(64 * NW, (HALO && PF == 1) ? (NW == 12 ? 6 : 8) : 1) void vortdiv_levelwalk_kernel(const RowsParams P)
{
  constexpr int TR = HALO ? NW - 2 : NW; // rows of the tile
  __shared__ v4f su[2][TR + 2][64];
  __shared__ v4f sv[2][TR + 2][64];
  // undefined counts of a level: added up in LDS, handed to the level's counter by thread 0 after the barrier of the next
  // level -- one global atomic per workgroup and level (mifc_device.h: undefined-cell counting)
  __shared__ unsigned int sbad[2];
  if (CHECK && threadIdx.x < 2)
    sbad[threadIdx.x] = 0; // ordered before the first use by the barrier of the first level
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int bid = blockIdx.x;
  const int seq = P.xcd_remap ? ((bid & 7) * P.per_xcd + (bid >> 3)) : bid;
  if (seq >= P.n_logical)
    return;
  // unit = (level chunk, row block, column segment): column segment fastest, level chunk slowest
  const int ntiles = P.uB * P.uW;
  const int lchunk = seq / ntiles;
  const int tile = seq - lchunk * ntiles;
  const int rblock = tile / P.uW;
  const int wc = tile - rblock * P.uW;
  const int lev0 = lchunk * P.lgroup;
  const int lev1 = (lev0 + P.lgroup < P.nlev) ? lev0 + P.lgroup : P.nlev;

  const int nx = P.nx;
  const int first = P.lo + rblock * TR; // first row of the tile (always computed)
  // LDS slot s holds tile row s - 1 (slot 0: the row above the tile, slot TR + 1: the row below)
  const int slot = HALO ? wave : wave + 1;
  const int jl_raw = first + slot - 1;
  const bool computes = (!HALO || (wave >= 1 && wave <= TR)) && jl_raw < P.hi;
  const int jl = jl_raw < P.hi ? jl_raw : P.hi; // rows past the computed range: only loaded (row `hi` always exists), for the wave above
  const int j = P.j0 + jl;
  // without halo waves the first wave also brings the row above the tile, the last wave the row below
  const bool has_extra = !HALO && ((wave == 0) || (wave == NW - 1));
  const int extra_row = (wave == 0) ? first - 1 : ((first + NW < P.hi) ? first + NW : P.hi);
  const int extra_slot = (wave == 0) ? 0 : NW + 1;
  const int col = wc * 256 + lane * 4;
  const bool act = col < nx;
  const int col_c = act ? col : nx - 4;
  int east_col = wc * 256 + 256;
  if (east_col > nx)
    east_col = nx;
  const int edge_col = (lane == 63) ? east_col : (wc * 256 - 1);
  const float undef = P.undef;
  // offsets inside one level fit 32 bits (the launcher checks): half the registers of 64-bit ones, and they live through the whole walk
  const int base = jl * nx;
  const int o = base + col_c;
  const int ox = extra_row * nx + col_c;
  long e64 = (long)base + edge_col;
  e64 = e64 < P.idx_lo ? P.idx_lo : (e64 > P.idx_hi ? P.idx_hi : e64);
  const int e = (int)e64;
  const bool top = (j == 1) && (P.j0 == 0);
  const bool bottom = (j == P.nyg - 2) && (P.j0 + P.ny_local == P.nyg);
  const bool last_in_seg = col + 4 >= east_col;
  const int oo = base + col;

  // rows 2 .. TR-1 of a tile are read by this workgroup only (rows 1 and TR are the neighbours' halo rows)
  const bool stream_rows = HALO && P.nt_interior && wave >= 2 && wave <= TR - 1;
  // the tile's map factors: once, for every level of the chunk
  v4f xm4 = {0.f, 0.f, 0.f, 0.f}, ym4 = xm4;
  if (computes) {
    xm4 = load4(P.xm + o);
    ym4 = load4(P.ym + o);
  }
  struct LevRegsX
  {
    v4f u, v, xu, xv;
    float eu, ev;
  };
  struct LevRegsH
  {
    v4f u, v;
    float eu, ev;
  };
  typedef typename std::conditional<HALO, LevRegsH, LevRegsX>::type LevRegs;
  auto load_level = [&](int lev) -> LevRegs {
    const int l = lev < lev1 ? lev : lev1 - 1; // past the chunk: a valid address, never used
    const float* __restrict__ u = P.u + (size_t)l * P.in_stride;
    const float* __restrict__ v = P.v + (size_t)l * P.in_stride;
    LevRegs r;
#ifdef MIFC_MEASUREMENT_BUILD
    if (P.exp_noload) { // write side alone
      const float f = (float)(lev + lane);
      const v4f z = {f, f + 1.f, f * 2.f, f + 3.f};
      r.u = z;
      r.v = z;
      if constexpr (!HALO) {
        r.xu = z;
        r.xv = z;
      }
      r.eu = 0.f;
      r.ev = 0.f;
      return r;
    }
#endif
    if (stream_rows) { // no other workgroup reads this row: do not keep it in L2, where the shared boundary rows should stay
      r.u = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(u + o));
      r.v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(v + o));
    } else {
      r.u = load4(u + o);
      r.v = load4(v + o);
    }
    if constexpr (!HALO) {
      r.xu = r.u;
      r.xv = r.v;
      if (has_extra) {
        r.xu = load4(u + ox);
        r.xv = load4(v + ox);
      }
    }
    r.eu = u[e];
    r.ev = v[e];
    return r;
  };
  LevRegs R[PF + 1];
#pragma unroll
  for (int k = 0; k < PF; ++k)
    R[k] = load_level(lev0 + k);

  for (int lb = lev0; lb < lev1; lb += PF + 1) {
#pragma unroll
    for (int s = 0; s <= PF; ++s) {
      const int lev = lb + s;
      if (lev >= lev1)
        goto chunk_done;
      R[(s + PF) % (PF + 1)] = load_level(lev + PF); // in flight while this level and the next PF-1 are computed
      const LevRegs& C = R[s];
      const int buf = (lev - lev0) & 1;
      su[buf][slot][lane] = C.u;
      sv[buf][slot][lane] = C.v;
      if constexpr (!HALO) {
        if (has_extra) {
          su[buf][extra_slot][lane] = C.xu;
          sv[buf][extra_slot][lane] = C.xv;
        }
      }
      // LDS rows of this level visible to the workgroup; only the LDS counter is waited for -- the prefetches
      // above and the stores of the previous levels stay in flight across the barrier.  Two buffers: a wave
      // can only overwrite buffer b again after the barrier of the level in between, which every wave reaches
      // with its reads of b consumed.
      // (the level's input flag through the scalar cache, with the same wait: a vector load of the byte would be the
      // youngest entry of this wave's vmcnt queue, and waiting for it would drain the prefetch and every store in flight)
      bool all = true;
      if (CHECK)
        all = level_flag_then_barrier(P.all_defined, lev);
      else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (CHECK && P.n_undefined && threadIdx.x == 0 && lev > lev0) { // the count of the previous level is complete
        const int q = (lev - 1 - lev0) & 1;
        const unsigned int n = sbad[q];
        if (n != 0) {
          atomicAdd(P.n_undefined + (lev - 1), (u64)n);
          sbad[q] = 0; // the next adds into this slot come after the next barrier
        }
      }
      if (computes) {
        // two passes, vorticity then divergence, each taking its rows from LDS when it needs them and storing at
        // once: fewer live registers than one fused pass (the variant with tests spilled otherwise)
        const v4f uc = C.u, vc = C.v;
        const float east_u = readlane_f(C.eu, 63), east_v = readlane_f(C.ev, 63);
        float uW = dpp_from_lower_lane(C.eu, uc.w), vW = dpp_from_lower_lane(C.ev, vc.w);
        float uE = dpp_from_upper_lane(C.eu, uc.x), vE = dpp_from_upper_lane(C.ev, vc.x);
        if (last_in_seg) {
          uE = east_u;
          vE = east_v;
        }
        bool ok[4] = {true, true, true, true};
        unsigned int bad = 0;
        {
          const v4f un = su[buf][slot + 1][lane], us = su[buf][slot - 1][lane];
          const float vcx[6] = {vW, vc.x, vc.y, vc.z, vc.w, vE};
          float zv[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float vw = vcx[k], ve = vcx[k + 2];
            if (CHECK) // both operators test these four values (:1861, :1927)
              ok[k] = all | all_def(undef, vw, ve, us[k], un[k]);
            if (WANT_V)
              zv[k] = pick(ok[k], f_relvort(xm4[k], ym4[k], ve - vw, un[k] - us[k]), undef);
            if (CHECK)
              bad += (!ok[k] & act) ? 1u : 0u;
          }
          if (WANT_V) {
            if (col == 0) // fillEdges, column part (:65-68)
              zv[0] = zv[1];
            if (col + 4 == nx)
              zv[3] = zv[2];
#ifdef MIFC_MEASUREMENT_BUILD
            if (act && !(P.exp_nostore && zv[0] != 12345.678f)) {
#else
            if (act) {
#endif
              float* rv = P.rv + (size_t)lev * P.out_stride;
              v4f z4;
              z4.x = zv[0];
              z4.y = zv[1];
              z4.z = zv[2];
              z4.w = zv[3];
              store4<NT>(rv + oo, z4);
              if (top) // fillEdges, row part (:70-73)
                store4<NT>(rv + oo - nx, z4);
              if (bottom)
                store4<NT>(rv + oo + nx, z4);
            }
          }
        }
        if (WANT_D) {
          if (WANT_V)
            __builtin_amdgcn_sched_barrier(0);
          const v4f vn = sv[buf][slot + 1][lane], vs = sv[buf][slot - 1][lane];
          const float ucx[6] = {uW, uc.x, uc.y, uc.z, uc.w, uE};
          float zd[4];
#pragma unroll
          for (int k = 0; k < 4; ++k)
            zd[k] = pick(ok[k], f_diverg(xm4[k], ym4[k], ucx[k + 2] - ucx[k], vn[k] - vs[k]), undef);
          if (col == 0)
            zd[0] = zd[1];
          if (col + 4 == nx)
            zd[3] = zd[2];
#ifdef MIFC_MEASUREMENT_BUILD
          if (act && !(P.exp_nostore && zd[0] != 12345.678f)) {
#else
          if (act) {
#endif
            float* dv = P.dv + (size_t)lev * P.out_stride;
            v4f d4;
            d4.x = zd[0];
            d4.y = zd[1];
            d4.z = zd[2];
            d4.w = zd[3];
            store4<NT>(dv + oo, d4);
            if (top)
              store4<NT>(dv + oo - nx, d4);
            if (bottom)
              store4<NT>(dv + oo + nx, d4);
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

// ---------------------------------------------------------------------------
// Split-role level-walking form (tuning K=4): the tile, the walk through a chunk of levels and the arithmetic of the
// level-walking kernel above, but the waves of a workgroup either LOAD or COMPUTE AND STORE, never both:
//   * NL loader waves bring the TR + 2 rows of u and of v of a level (and the 2 x 2 edge scalars of every row)
//     straight from global memory into LDS (global_load_lds_dwordx4: no registers, no ds_write), PF levels ahead of the
//     level being computed, into a ring of PF + 1 level buffers;
//   * TR compute waves (one row each) take everything from LDS -- own row, the rows above and below, the edge
//     scalars -- x-neighbours from the adjacent lanes, and store 2 x 1 KiB; the only thing in their vmcnt queue is
//     stores, which nothing in the loop ever waits for.
// Why: on gfx9 a wave's loads and stores share ONE in-order counter, so a wave that loads for level l + 1 after it
// stored level l - 1 gets its data "back" only when those older stores have been acknowledged.  The copy yardsticks
// show what that coupling costs (one process, 2 x 1.136 GB each way): loop copy 0.42-0.45 ms, the same loop with
// split roles 0.364-0.371, one-shot 0.367, one-shot with split roles 0.339.
//   iteration for level lev (t = lev - lev0), ONE barrier:
//       loaders:  s_waitcnt vmcnt((PF - 1) * L)      level lev has landed in buffer t % (PF + 1)
//       everyone: s_barrier
//       loaders:  issue level lev + PF into buffer (t + PF) % (PF + 1)   -- the buffer of level lev - 1, whose readers
//                                                                           all passed the barrier above
//       computes: level lev from buffer t % (PF + 1)
// A loader wave issues the same number L of load instructions for every level (a wave with fewer rows repeats its last
// one), so that the wait count is an immediate; loader 0, which also gathers the edge scalars, has its own copy of the loop.
// WANT_V / WANT_D: which of the two results the launch produces (relvort or divergence alone read the very same rows:
// 12 instead of 16 B per cell); ABSV: the vorticity gets the tile's Coriolis parameter added (absvort, :1896), which
// lives in registers next to the map factors for the whole walk.
// FF: a third output, the wind speed ff = sqrt(u*u + v*v) of every cell (vectorabs, FieldCalculations.cc:1819-1841: float
// arithmetic, its own test on exactly (u[i], v[i]), its own count over ALL nx*ny cells) from the rows of u and v that are in
// LDS anyway -- BASELINE.json config 5 computes ff next to vorticity and divergence per member, and as a separate launch ff
// reads u and v a second time (44 -> 36 B per cell for the member).  Rows 0 and ny-1 have no compute wave of their own: the
// waves of rows 1 and ny-2, which fill them for the stencil outputs, compute their ff from the halo slots.
// RAGGED (round 3): any width and dword-aligned fields.  The tiles stay 256 columns wide in COLUMN space, so rows start
// wherever they start: the loaders do not care (see store4_any_alignment), the stores are 16-byte stores at dword
// alignment, the last column group of a row may be partial (its trailing cells are the first cells of the next row -- exactly
// the flat neighbours the reference's loop sees at column nx-1), and the one group whose 16 bytes would reach past the END
// of the batch (last row of the last level) is loaded cell by cell by its loader.
// JAC (round 3): the same rows of two fields in LDS are all jacobian(field1, field2) needs (:2424-2455) -- its four partials
// are the fused pair's four differences, its test the pair's plus the other field's four values; one output, P.rv.
template <bool CHECK, bool NT, int TR, int NL, int PF, bool WANT_V = true, bool WANT_D = true, bool ABSV = false, bool FF = false, bool RAGGED = false,
          bool JAC = false>
__global__ __launch_bounds__(64 * (TR + NL)) void vortdiv_split_kernel(const RowsParams P)
{
  static_assert(WANT_V || WANT_D, "nothing to compute");
  static_assert(!JAC || (WANT_V && !WANT_D && !ABSV && !FF), "the Jacobian is a single-output operator");
  static_assert(!(RAGGED && FF), "the three-output form takes aligned fields only");
  static_assert(!FF || (WANT_V && WANT_D && !ABSV), "the wind speed rides on the fused pair");
  static_assert(!ABSV || (WANT_V && !WANT_D), "absvort is a single-output operator");
  constexpr int NB = PF + 1;                // level buffers
  constexpr int NS = TR + 2;                // row slots per level: slot s holds tile row s - 1
  constexpr int KMAX = (NS + NL - 1) / NL;  // row slots per loader wave
  static_assert(4 * NS <= 64, "the edge scalars of a level are one dword per lane");
  static_assert(NL >= 2, "loader 0 gathers the edge scalars, the last loader hands the undefined counts over");
  __shared__ v4f srow[NB][NS][2][64];       // [buffer][slot][u | v][lane]
  __shared__ float sedge[NB][64];           // [buffer][4 * slot + 2 * (u | v) + (west | east)]
  // undefined counts of a level: the compute waves add theirs into LDS, the last loader hands the total of level l to the
  // global counter after the barrier of level l + 1 -- ONE global atomic per workgroup and level (same-address atomics are
  // served one after the other: a masked field made the per-wave atomics cost twice the kernel)
  __shared__ unsigned int sbad[2];
  __shared__ unsigned int sbadf[2]; // the same for the wind speed's counts
  if (CHECK && threadIdx.x < 2) {
    sbad[threadIdx.x] = 0; // ordered before the first use by the barrier of the first level
    sbadf[threadIdx.x] = 0;
  }
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int bid = blockIdx.x;
  const int seq = P.xcd_remap ? ((bid & 7) * P.per_xcd + (bid >> 3)) : bid;
  if (seq >= P.n_logical)
    return;
  const int ntiles = P.uB * P.uW;
  const int lchunk = seq / ntiles;
  const int tile = seq - lchunk * ntiles;
  const int rblock = tile / P.uW;
  const int wc = tile - rblock * P.uW;
  const int lev0 = lchunk * P.lgroup;
  const int lev1 = (lev0 + P.lgroup < P.nlev) ? lev0 + P.lgroup : P.nlev;
  const int nx = P.nx;
  const int first = P.lo + rblock * TR; // first row of the tile (always computed)
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
    int off_end[KMAX]; // RAGGED: the same with the groups that would reach past the valid range pulled back inside it
    bool fix[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      int s = lw + NL * k;
      if (s > NS - 1)
        s = NS - 1; // a wave with fewer rows repeats the last slot (same data to the same place)
      const int jl_raw = first + s - 1;
      const int jl = jl_raw < P.hi ? jl_raw : P.hi; // rows past the computed range: row `hi` always exists
      slot_of[k] = s;
      off[k] = jl * nx + col_c;
      fix[k] = RAGGED && ((long)off[k] + 3 > P.idx_hi);
      off_end[k] = fix[k] ? (int)P.idx_hi - 3 : off[k];
    }
    // edge scalars: lane i gathers (slot i / 4, array (i / 2) % 2, side i % 2); lanes past 4 NS repeat lane 0's
    const int ei = (lane < 4 * NS) ? lane : 0;
    const int es = ei >> 2;
    const bool e_is_v = ((ei >> 1) & 1) != 0;
    const int ejl_raw = first + es - 1;
    const int ejl = ejl_raw < P.hi ? ejl_raw : P.hi;
    long e64 = (long)ejl * nx + ((ei & 1) ? east_col : (wc * 256 - 1));
    e64 = e64 < P.idx_lo ? P.idx_lo : (e64 > P.idx_hi ? P.idx_hi : e64);
    const int eoff = (int)e64;
    const float* __restrict__ ebase = e_is_v ? P.v : P.u;

    // loader 0 also gathers the edge scalars of the level (one dword per lane); the wait count is an immediate, so the
    // two kinds of loader run their own copy of the loop
    auto walk = [&](auto edge_tag) __attribute__((always_inline)) {
      constexpr bool EDGE = decltype(edge_tag)::value;
      constexpr int L = 2 * KMAX + (EDGE ? 1 : 0); // load instructions per level of this wave
      auto issue = [&](int lev, int b) {
        const int l = lev < lev1 ? lev : lev1 - 1; // past the chunk: a valid address into a buffer nobody reads
        const float* __restrict__ u = P.u + (size_t)l * P.in_stride;
        const float* __restrict__ v = P.v + (size_t)l * P.in_stride;
        // behind the last level there is no next level for a partial group to read into: there its 16 bytes are pulled back
        // inside the batch and the group is loaded again, cell by cell, below
        const bool at_end = RAGGED && l == P.nlev - 1;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
          const int o = at_end ? off_end[k] : off[k];
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(u + o),
                                           (void __attribute__((address_space(3)))*)&srow[b][slot_of[k]][0][0], 16, 0, 0);
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(v + o),
                                           (void __attribute__((address_space(3)))*)&srow[b][slot_of[k]][1][0], 16, 0, 0);
        }
        if (EDGE)
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(ebase + (size_t)l * P.in_stride + eoff),
                                           (void __attribute__((address_space(3)))*)&sedge[b][0], 4, 0, 0);
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
                v4f qu = {0.f, 0.f, 0.f, 0.f}, qv = qu;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  const long i = (long)off[k] + c;
                  if (i <= P.idx_hi) {
                    qu[c] = u[i];
                    qv[c] = v[i];
                  }
                }
                srow[b][slot_of[k]][0][lane] = qu;
                srow[b][slot_of[k]][1][lane] = qv;
              }
            }
          }
        }
      };
#pragma unroll
      for (int k = 0; k < PF; ++k)
        issue(lev0 + k, k);
      int b_next = PF % NB; // buffer of level lev + PF
      // the last loader also hands a level's undefined count (added up in LDS by the compute waves) to the global counter,
      // one level late: it has nothing else to do between its loads and the next barrier, a compute wave would make the
      // whole workgroup wait for the LDS round trip.  Its atomic is one more entry in this wave's vmcnt queue, younger
      // than the loads the next wait is about: that wait can only get stricter.
      auto hand_over = [&](int lev_done) {
        if (CHECK && !EDGE && lw == NL - 1 && P.n_undefined && lane == 0) {
          const int q = (lev_done - lev0) & 1;
          const unsigned int n = sbad[q];
          if (P.partials) { // big levels: a plain store per workgroup and level, added up behind the launch (StencilParams::partials)
            P.partials[(size_t)lev_done * (size_t)ntiles + tile] = n;
            if (n != 0)
              sbad[q] = 0;
          } else if (n != 0) {
            atomicAdd(P.n_undefined + lev_done, (u64)n);
            sbad[q] = 0; // the next adds into this slot come after the next barrier
          }
          if constexpr (FF) {
            const unsigned int nf = sbadf[q];
            if (nf != 0) {
              atomicAdd(P.n_undefined_ff + lev_done, (u64)nf);
              sbadf[q] = 0;
            }
          }
        }
      };
      for (int lev = lev0; lev < lev1; ++lev) {
        // lgkmcnt(0): the reset of a count slot below (a ds_write) has landed before the compute waves add into it again
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((PF - 1) * L) : "memory");
        issue(lev + PF, b_next);
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
    if (lw == 0)
      walk(std::true_type());
    else
      walk(std::false_type());
    return;
  }

  // -------------------------------------------------------------------- compute
  const int slot = wave + 1;
  const int jl_raw = first + wave;
  const bool computes = jl_raw < P.hi;
  const int jl = computes ? jl_raw : P.hi;
  const int j = P.j0 + jl;
  const float undef = P.undef;
  const int base = jl * nx;
  const int o = base + col_c;
  const bool top = (j == 1) && (P.j0 == 0);
  const bool bottom = (j == P.nyg - 2) && (P.j0 + P.ny_local == P.nyg);
  const bool last_in_seg = col + 4 >= east_col;
  const int oo = base + col;
  v4f xm4 = {0.f, 0.f, 0.f, 0.f}, ym4 = xm4, fc4 = xm4;
  if (computes) { // the tile's map factors: once, for every level of the chunk
    if constexpr (RAGGED) { // (dword-aligned rows: unaligned 16-byte loads; a partial group reads into the next row, which exists)
      xm4 = reinterpret_cast<const V4Unaligned*>(P.xm + o)->v;
      ym4 = reinterpret_cast<const V4Unaligned*>(P.ym + o)->v;
      if constexpr (ABSV)
        fc4 = reinterpret_cast<const V4Unaligned*>(P.fc + o)->v;
    } else {
      xm4 = load4(P.xm + o);
      ym4 = load4(P.ym + o);
      if constexpr (ABSV)
        fc4 = load4(P.fc + o);
    }
  }
  // fillEdges, column part (:65-68), for any width: the last column takes the value of the one before it, which may sit in
  // the lane below (the group that holds column nx-1 then holds nothing else)
  const int k_last = nx - 1 - col; // 0 .. 3 in the lane that holds column nx-1
  auto fill_columns = [&](float (&z)[4]) __attribute__((always_inline)) {
    if (col == 0)
      z[0] = z[1];
    if constexpr (RAGGED) {
      const float below = dpp_from_lower_lane(z[3], z[3]); // every lane of the wave is here
      if (k_last == 0)
        z[0] = below;
      else if (k_last == 1)
        z[1] = z[0];
      else if (k_last == 2)
        z[2] = z[1];
      else if (k_last == 3)
        z[3] = z[2];
    } else {
      if (col + 4 == nx)
        z[3] = z[2];
    }
  };
  auto store_rows = [&](float* field, const float (&z)[4]) __attribute__((always_inline)) {
    if (act) {
      v4f z4;
      z4.x = z[0];
      z4.y = z[1];
      z4.z = z[2];
      z4.w = z[3];
      if constexpr (RAGGED) {
        store4_any_alignment(field + oo, z4, nvalid);
        if (top) // fillEdges, row part (:70-73)
          store4_any_alignment(field + oo - nx, z4, nvalid);
        if (bottom)
          store4_any_alignment(field + oo + nx, z4, nvalid);
      } else {
        store4<NT>(field + oo, z4);
        if (top) // fillEdges, row part (:70-73)
          store4<NT>(field + oo - nx, z4);
        if (bottom)
          store4<NT>(field + oo + nx, z4);
      }
    }
  };
  // the Jacobian's partials are (float)(0.5 * m * d): one float multiplication with a pre-halved map factor where every
  // lane's halves are exact (wave-uniform, decided once per chunk; half_prod otherwise) -- see ScalarHoist, mifc_scalar_cell.h
  float hx4[4] = {0.f, 0.f, 0.f, 0.f}, hy4[4] = {0.f, 0.f, 0.f, 0.f};
  bool halves = false;
  if constexpr (JAC) {
    bool exact = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hx4[k] = 0.5f * xm4[k];
      hy4[k] = 0.5f * ym4[k];
      exact = exact & (__builtin_fabsf(xm4[k]) >= 0x1p-125f || xm4[k] == 0.f) & (__builtin_fabsf(ym4[k]) >= 0x1p-125f || ym4[k] == 0.f);
    }
    halves = __builtin_amdgcn_ballot_w64(!exact) == 0;
  }
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
      const v4f uc = srow[buf][slot][0][lane], vc = srow[buf][slot][1][lane];
      const float uWe = sedge[buf][4 * slot + 0], uEe = sedge[buf][4 * slot + 1];
      const float vWe = sedge[buf][4 * slot + 2], vEe = sedge[buf][4 * slot + 3];
      const float uW = dpp_from_lower_lane(uWe, uc.w), vW = dpp_from_lower_lane(vWe, vc.w); // lane 0 keeps the west scalar
      float uE = dpp_from_upper_lane(uEe, uc.x), vE = dpp_from_upper_lane(vEe, vc.x);       // lane 63 keeps the east scalar
      if (last_in_seg) {
        uE = uEe;
        vE = vEe;
      }
      bool ok[4] = {true, true, true, true};
      unsigned int bad = 0;
      if constexpr (JAC) { // jacobian(field1 = u, field2 = v), :2443-2449: all eight neighbours tested, four float-rounded partials
        const v4f un = srow[buf][slot + 1][0][lane], us = srow[buf][slot - 1][0][lane];
        const v4f vn = srow[buf][slot + 1][1][lane], vs = srow[buf][slot - 1][1][lane];
        const float ucx[6] = {uW, uc.x, uc.y, uc.z, uc.w, uE};
        const float vcx[6] = {vW, vc.x, vc.y, vc.z, vc.w, vE};
        bool dux[6] = {true, true, true, true, true, true}, dvx[6] = {true, true, true, true, true, true};
        if (CHECK) {
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            dux[k] = __builtin_islessgreater(ucx[k], undef);
            dvx[k] = __builtin_islessgreater(vcx[k], undef);
          }
        }
        float zj[4];
        auto cells = [&](auto halves_tag) __attribute__((always_inline)) {
          constexpr bool HALVES = decltype(halves_tag)::value;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (CHECK)
              ok[k] = all | (dux[k] & dux[k + 2] & dvx[k] & dvx[k + 2] & (bool)__builtin_islessgreater(us[k], undef) & (bool)__builtin_islessgreater(un[k], undef) &
                             (bool)__builtin_islessgreater(vs[k], undef) & (bool)__builtin_islessgreater(vn[k], undef));
            const float df1dx = HALVES ? hx4[k] * (ucx[k + 2] - ucx[k]) : half_prod(xm4[k], ucx[k + 2] - ucx[k]);
            const float df1dy = HALVES ? hy4[k] * (un[k] - us[k]) : half_prod(ym4[k], un[k] - us[k]);
            const float df2dx = HALVES ? hx4[k] * (vcx[k + 2] - vcx[k]) : half_prod(xm4[k], vcx[k + 2] - vcx[k]);
            const float df2dy = HALVES ? hy4[k] * (vn[k] - vs[k]) : half_prod(ym4[k], vn[k] - vs[k]);
            const float z = df1dx * df2dy - df1dy * df2dx;
            zj[k] = ok[k] ? z : undef;
            if (CHECK)
              bad += (!ok[k] & act & (k < nvalid)) ? 1u : 0u;
          }
        };
        if (halves)
          cells(std::true_type());
        else
          cells(std::false_type());
        fill_columns(zj);
        store_rows(P.rv + (size_t)lev * P.out_stride, zj);
      } else if constexpr (WANT_V || CHECK) { // divergence alone still tests the vorticity's four values (:1927)
        const v4f un = srow[buf][slot + 1][0][lane], us = srow[buf][slot - 1][0][lane];
        const float vcx[6] = {vW, vc.x, vc.y, vc.z, vc.w, vE};
        float zv[4];
        // The tests are straight-line code: every value is tested once (ONE compare: "ordered and different from undef",
        // which is is_def() for an undef that is not NaN -- the launcher sends a NaN undef to the other kernels), the
        // results are combined without short-circuits and the formula runs unconditionally with a select behind it.
        // Written with && and ?: the compiler built a nest of exec-mask regions and branches per cell.
        bool dvx[6] = {true, true, true, true, true, true};
        if (CHECK) {
#pragma unroll
          for (int k = 0; k < 6; ++k)
            dvx[k] = __builtin_islessgreater(vcx[k], undef);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float vw = vcx[k], ve = vcx[k + 2];
          if (CHECK) // both operators test these four values (:1861, :1927)
            ok[k] = all | (dvx[k] & dvx[k + 2] & (bool)__builtin_islessgreater(us[k], undef) & (bool)__builtin_islessgreater(un[k], undef));
          if constexpr (WANT_V) {
            const float z = ABSV ? f_absvort(xm4[k], ym4[k], ve - vw, un[k] - us[k], fc4[k]) : f_relvort(xm4[k], ym4[k], ve - vw, un[k] - us[k]);
            zv[k] = ok[k] ? z : undef;
          }
          if (CHECK)
            bad += (!ok[k] & act & (k < nvalid)) ? 1u : 0u;
        }
        if constexpr (WANT_V) {
          fill_columns(zv);
          store_rows(P.rv + (size_t)lev * P.out_stride, zv);
        }
      }
      if constexpr (WANT_D && !JAC) {
        const v4f vn = srow[buf][slot + 1][1][lane], vs = srow[buf][slot - 1][1][lane];
        const float ucx[6] = {uW, uc.x, uc.y, uc.z, uc.w, uE};
        float zd[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float d = f_diverg(xm4[k], ym4[k], ucx[k + 2] - ucx[k], vn[k] - vs[k]);
          zd[k] = ok[k] ? d : undef;
        }
        fill_columns(zd);
        store_rows(P.dv + (size_t)lev * P.out_stride, zd);
      }
      if (CHECK && P.n_undefined && !all && __builtin_amdgcn_ballot_w64(bad != 0) != 0) {
        const unsigned int n = wave_sum(bad);
        if (lane == 0)
          atomicAdd(&sbad[(lev - lev0) & 1], n);
      }
      if constexpr (FF) {
        // vectorabs (:1831-1837) of this wave's row -- and of row 0 / ny-1 in the waves next to them -- in float, tested on
        // the cell's own u and v
        unsigned int badf = 0;
        float* ffp = P.ff + (size_t)lev * P.out_stride;
        auto speed_row = [&](const v4f& ur, const v4f& vr, int at) {
          float f[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            bool okf = true;
            if (CHECK)
              okf = all | ((bool)__builtin_islessgreater(ur[k], undef) & (bool)__builtin_islessgreater(vr[k], undef));
            const float a = absval(ur[k], vr[k]);
            f[k] = okf ? a : undef;
            if (CHECK)
              badf += (!okf & act) ? 1u : 0u;
          }
          if (act) {
            v4f f4;
            f4.x = f[0];
            f4.y = f[1];
            f4.z = f[2];
            f4.w = f[3];
            store4<NT>(ffp + at, f4);
          }
        };
        speed_row(uc, vc, oo);
        if (top)
          speed_row(srow[buf][slot - 1][0][lane], srow[buf][slot - 1][1][lane], oo - nx);
        if (bottom)
          speed_row(srow[buf][slot + 1][0][lane], srow[buf][slot + 1][1][lane], oo + nx);
        if (CHECK && P.n_undefined_ff && !all && __builtin_amdgcn_ballot_w64(badf != 0) != 0) {
          const unsigned int n = wave_sum(badf);
          if (lane == 0)
            atomicAdd(&sbadf[(lev - lev0) & 1], n);
        }
      }
    }
    buf = (buf + 1 == NB) ? 0 : buf + 1;
  }
  if (CHECK) // the last level's adds are complete: the last loader hands its total over
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct Tuning
{
  int K;     // 0: row-walking kernel (default), 1: one-shot kernel, 2: one-shot tiles with the row reuse in LDS, 3: level-walking tiles
  int R;     // rows per band
  int D;     // rows kept in flight beyond the 3-row window (0 or 1)
  int NT;    // nontemporal stores
  int V;     // float4 per lane and row (1 or 2): the wave covers 256*V columns
  int ORDER; // block order, see decode_block()
  int XCD;   // XCD-aware blockIdx remap
  int WPB;   // waves per workgroup: 4, 8 or 16 (levels side by side)
  int ZZ;    // odd bands walk upwards (halo rows meet in L2)
  int NTI;   // nontemporal loads for the rows of a band that no other band reads
  int XH;    // measurement build only: skip the halo rows (results are wrong)
  int XS;    // measurement build only: skip (practically all) stores
  int PADROWS; // measurement build only: the last PADROWS rows of every level are padding (changes the level stride)
  int XL;      // measurement build only: skip the field loads (write side alone)
  int LDSX;    // extra KiB of LDS requested per workgroup: limits the workgroups resident on a CU (occupancy experiments)
  int STA;     // measurement build only: buffer stores with this cache policy (aux bits)
  int LG;      // one-shot forms: level-minor unit order in groups of LG levels (0: address order)
  int RB;      // one-shot tile form: rows per tile (8 or 14)
};

int tune_value(const char* s, const char* key, int dflt)
{
  // finds "KEY=" at the start of the string or after a comma
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

Tuning current_tuning(int nx)
{
  Tuning t = {0, 8, 1, 1, 2, 1, 1, 8, 1, 0, 0, 0, 0, 0, 0, 0, 0, 8}; // R = 6 / WPB = 4 run within 1 % of this but fetch more (halo rows, map factors): HBM traffic 1.13-1.14x vs 1.08x of the minimum
  // MIFC_VORTDIV_TUNE="R=8,D=1,NT=1,V=2,ORDER=1,XCD=1,WPB=8" -- used by the sweep tool and the tests
  if (env().has_vortdiv_tune) {
    const char* s = env().vortdiv_tune;
    t.K = tune_value(s, "K", t.K);
    t.R = tune_value(s, "R", t.R);
    t.D = tune_value(s, "D", t.D);
    t.NT = tune_value(s, "NT", t.NT);
    t.V = tune_value(s, "V", t.V);
    t.ORDER = tune_value(s, "ORDER", t.ORDER);
    t.XCD = tune_value(s, "XCD", t.XCD);
    t.WPB = tune_value(s, "WPB", t.WPB);
    t.ZZ = tune_value(s, "ZZ", t.ZZ);
    t.NTI = tune_value(s, "NTI", t.NTI);
    t.LDSX = tune_value(s, "LDSX", t.LDSX);
    t.LG = tune_value(s, "LG", t.LG);
    t.RB = tune_value(s, "RB", t.RB);
#ifdef MIFC_MEASUREMENT_BUILD
    // knobs that make the kernel compute something else (wrong results by design): they exist in
    // libmifc_measure.so only, which tools/ load explicitly; the product library has no such code
    t.XH = tune_value(s, "XH", 0);
    t.XS = tune_value(s, "XS", 0);
    t.PADROWS = tune_value(s, "PADROWS", 0);
    t.XL = tune_value(s, "XL", 0);
    t.STA = tune_value(s, "STA", 0);
#endif
  }
  if (t.WPB != 1 && t.WPB != 2 && t.WPB != 4 && t.WPB != 8)
    t.WPB = 4; // the kernel is compiled for workgroups of up to 8 waves
  if (t.V != 1 && t.V != 2 && t.V != 3)
    t.V = 2;
  if (nx <= 256)
    t.V = 1; // a second 256-column segment would be empty
  if (t.R < 1)
    t.R = 1;
  const int rmax = 32 / t.V; // map-factor tile in LDS: up to 3*V KiB per row (absvort), 96 KiB at most
  if (t.R > rmax)
    t.R = rmax;
  if (t.D < 0)
    t.D = 0;
  if (t.D > 1)
    t.D = 1; // deeper rings (2, 3) were measured -- same time, profiles/r01/experiments/sweep_d.txt -- and are not instantiated
  return t;
}

template <bool CHECK, bool WV, bool WD, bool ABSV, int D, bool NT>
void launch_v(const RowsParams& rp, const Tuning& t, int grid, hipStream_t stream)
{
  const size_t lds = (size_t)rp.R * 1024 * t.V * (ABSV ? 3 : 2) + (size_t)t.LDSX * 1024;
  if (t.V == 2)
    hipLaunchKernelGGL((vortdiv_rows_kernel<CHECK, WV, WD, ABSV, D, NT, 2>), dim3(grid), dim3(64 * t.WPB), lds, stream, rp);
  else if (t.V == 3)
    hipLaunchKernelGGL((vortdiv_rows_kernel<CHECK, WV, WD, ABSV, D, NT, 3>), dim3(grid), dim3(64 * t.WPB), lds, stream, rp);
  else
    hipLaunchKernelGGL((vortdiv_rows_kernel<CHECK, WV, WD, ABSV, D, NT, 1>), dim3(grid), dim3(64 * t.WPB), lds, stream, rp);
}

template <bool CHECK, bool WV, bool WD, bool ABSV>
void launch_d(const RowsParams& rp, const Tuning& t, int grid, hipStream_t stream)
{
  if (!t.NT) { // plain stores: only the default depth is instantiated
    launch_v<CHECK, WV, WD, ABSV, 1, false>(rp, t, grid, stream);
    return;
  }
  if (!(WV && WD)) { // single-output forms: default depth only (the depth sweep is about the fused kernel)
    launch_v<CHECK, WV, WD, ABSV, 1, true>(rp, t, grid, stream);
    return;
  }
  if (t.D == 0)
    launch_v<CHECK, WV, WD, ABSV, 0, true>(rp, t, grid, stream);
  else
    launch_v<CHECK, WV, WD, ABSV, 1, true>(rp, t, grid, stream);
}

template <bool CHECK>
void launch_jacobian(const RowsParams& rp, const Tuning& t, int grid, hipStream_t stream)
{
  const size_t lds = (size_t)rp.R * 1024 * t.V * 2;
  if (t.V == 2)
    hipLaunchKernelGGL((vortdiv_rows_kernel<CHECK, true, false, false, 1, true, 2, true>), dim3(grid), dim3(64 * t.WPB), lds, stream, rp);
  else
    hipLaunchKernelGGL((vortdiv_rows_kernel<CHECK, true, false, false, 1, true, 1, true>), dim3(grid), dim3(64 * t.WPB), lds, stream, rp);
}

template <bool CHECK>
void launch_outputs(const RowsParams& rp, const Tuning& t, int grid, hipStream_t stream)
{
  if (rp.fc)
    launch_d<CHECK, true, false, true>(rp, t, grid, stream);
  else if (rp.rv && rp.dv)
    launch_d<CHECK, true, true, false>(rp, t, grid, stream);
  else if (rp.rv)
    launch_d<CHECK, true, false, false>(rp, t, grid, stream);
  else
    launch_d<CHECK, false, true, false>(rp, t, grid, stream);
}

// when the level-walking form takes over from the row-walking one (measured: profiles/r02/experiments/levelwalk_threshold.txt)
constexpr int kLevelWalkMinLevels = 3;
constexpr long kLevelWalkMinUnits = 768;

inline bool aligned16(const void* p)
{
  return (reinterpret_cast<size_t>(p) & 15u) == 0;
}

} // namespace

// Distance between the levels of a device-resident batch.  The waves of a workgroup walk eight
// consecutive levels side by side; where those eight streams land in the memory channels depends on
// the level stride.  (Measured: profiles/r02/placement_sweep.txt.)
size_t padded_level_stride(size_t n)
{
  const size_t n4 = (n + 3) & ~size_t(3);
  return n4;
}

// Takes the request when the fast path applies (nx % 4 == 0, 16-byte aligned
// bases and strides); otherwise leaves *handled false and the caller falls
// back to the one-lane-per-cell kernel.
hipError_t launch_vortdiv_rows(const StencilParams& prm, hipStream_t stream, bool* handled)
{
  *handled = false;
  float* rv = nullptr;
  float* dv = nullptr;
  if (prm.op == ST_VORTDIV) {
    rv = prm.out0;
    dv = prm.out1;
  } else if (prm.op == ST_RELVORT) {
    rv = prm.out0;
  } else if (prm.op == ST_DIVERGENCE) {
    dv = prm.out0;
  } else if (prm.op == ST_ABSVORT) {
    rv = prm.out0;
    if (!prm.fcoriolis)
      return hipSuccess;
  } else if (prm.op == ST_JACOBIAN) {
    rv = prm.out0;
  } else {
    return hipSuccess;
  }
  if (!rv && !dv)
    return hipSuccess;
  const int nx = prm.nx;
  if (nx < 8 || prm.ny_global < 3)
    return hipSuccess;
  // Rows that do not start at 16-byte boundaries (a width that is not a multiple of 4, unaligned fields or level strides): only
  // the split-role kernel has a form for them (RAGGED, round 3), i.e. deep batches of the wind operators; everything else is
  // left to the flat four-cells-per-lane kernel
  const bool ragged = nx % 4 != 0 || !aligned16(prm.f0) || !aligned16(prm.f1) || !aligned16(prm.xmapr) || !aligned16(prm.ymapr) || (rv && !aligned16(rv)) ||
                      (dv && !aligned16(dv)) || prm.in_level_stride % 4 != 0 || prm.out_level_stride % 4 != 0 ||
                      (prm.op == ST_ABSVORT && prm.fcoriolis && !aligned16(prm.fcoriolis));
  // (nx % 256 == 1: the column whose value fillEdges copies into column nx-1 belongs to another workgroup)
  if (ragged && (nx % 256 == 1 || env().has_vortdiv_tune || !env().split_roles || !env().levelwalk || !env().ragged_split || prm.out_ff))
    return hipSuccess;
  if (env().force_cell_kernel)
    return hipSuccess;

  // the split-role kernel's tests are ONE compare per value ("ordered and != undef"), which is is_def() only for an
  // undef that is not NaN: a NaN undef takes the kernels with the generic two-compare test
  const bool nan_undef_tested = !prm.every_level_all_defined && prm.undef != prm.undef;
  // the wind speed as a third output exists in the split-role form only (whole fields, fused pair): anything else is left to
  // the caller, which runs vectorabs as a launch of its own
  if (prm.out_ff && !(rv && dv && prm.op == ST_VORTDIV && prm.j0 == 0 && prm.ny_local == prm.ny_global && prm.row_end <= prm.row_begin && !nan_undef_tested &&
                      env().split_roles && env().levelwalk && !env().has_vortdiv_tune && prm.nlev >= kLevelWalkMinLevels))
    return hipSuccess;
  Tuning t = current_tuning(nx);
  while (t.WPB > 1 && t.WPB / 2 >= prm.nlev)
    t.WPB /= 2; // fewer levels than waves: do not launch waves that only stage map factors
  if (!env().has_vortdiv_tune) {
    // A small launch (the reference's single-field call: one level) is latency-bound: shorter
    // bands put more waves on the chip, and their halo re-reads stay in L2.
    // One 1440x720 level: 8-row bands 270 waves, 2-row bands 1077; 8 levels (a chunk of the host pipeline) keep 8.
    const long rows = prm.ny_local, wcols = (nx + 256 * t.V - 1) / (256 * t.V), waves_per_band = (long)prm.nlev * wcols;
    // (MIFC_LEVELWALK_MIN_UNITS, the tests' switch, sends launches of any size to the level-walking forms)
    const bool small = env().levelwalk_min_units <= 0 && waves_per_band * ((rows + t.R - 1) / t.R) < 2048;
    if (small || prm.nlev <= 2) {
      // ... and the wind operators have forms without any row loop.  Small launches: one 1440x720 level takes
      // 6.7 us (7.5 us with tests and counts) instead of 7.3 (10.8) with 2-row bands, 12.9 (21.6) with 8-row
      // bands.  One or two levels of any size: the row-walking workgroup would be one or two waves holding a
      // 32-KiB map-factor tile (5 waves per CU); a 4000x4000 level straight from HBM runs at 48 % of peak
      // that way, 66 % one-shot, 69 % as one-shot tiles with the row reuse in LDS
      // (profiles/r01/other_configs.jsonl, cold numbers).
      t.K = (small || prm.op == ST_JACOBIAN) ? 1 : 2;
    }
    else if (((prm.op != ST_JACOBIAN && prm.op != ST_ABSVORT) || (env().split_roles && !nan_undef_tested)) && prm.nlev >= kLevelWalkMinLevels && env().levelwalk) {
      // Deep batches: tiles that stay put and walk the levels (map factors once per chunk of levels, a narrow
      // window of each array open at any time).  12-wave workgroups, 10 computed rows + 2 halo waves, chunks of
      // about 6 levels (8 in shallower batches), balanced; 3-6 % faster than the row-walking kernel on every device tried
      // (profiles/r02/experiments/sweep_k3_*.txt).
      const long tiles = ((rows + 9) / 10) * ((nx + 255) / 256);
      const int target = prm.nlev >= 48 ? 6 : 8; // levels per chunk; the chunks are then balanced
      const int nchunks = (prm.nlev + target - 1) / target;
      if (tiles * nchunks >= (env().levelwalk_min_units > 0 ? env().levelwalk_min_units : kLevelWalkMinUnits)) {
        t.K = 3;
        t.RB = 12;
        t.ZZ = 1;
        t.D = 0;
        t.LG = (prm.nlev + nchunks - 1) / nchunks;
        // The same tiles with split roles -- 2 loader waves bring 14 rows of u and v straight into LDS two levels
        // ahead, 12 compute waves only read LDS and store (vortdiv_split_kernel).  The fused pair: 1-5 % faster than the
        // form above on every box, placement and shape tried, 4 % on the tested variant, up to 18 % on shallow batches
        // (profiles/r02/experiments/sweep_k4_*.txt, ab_split_roles.txt).  Round 3: absvort too (+2 % on the row-walking
        // kernel it ran before, which has no level-walking form of the first kind); relvort / divergence ALONE measure
        // the same in both forms (12 B per cell: +-1 %, the sign depends on the box -- profiles/r03/split_role_ops.txt) and
        // keep the first, MIFC_VORTDIV_TUNE="K=4,..." selects the split-role one.
        // (single outputs on big tested levels too: that kernel leaves its counts in prm.partials, the first form adds them one by one)
        const bool big_tested = prm.partials && !prm.every_level_all_defined && tiles >= 2048;
        if (env().split_roles && !nan_undef_tested && ((rv && dv) || prm.op == ST_ABSVORT || prm.op == ST_JACOBIAN || ragged || big_tested)) {
          t.K = 4;
          t.D = 1;
          t.WPB = 2;
        }
      }
    }
    while (t.R > 2 && waves_per_band * ((rows + t.R - 1) / t.R) < 2048)
      t.R /= 2;
  }
  RowsParams rp;
  rp.nx = nx;
  rp.nyg = prm.ny_global - t.PADROWS;
  rp.j0 = prm.j0;
  rp.ny_local = prm.ny_local - t.PADROWS;
  rp.lo = (prm.j0 >= 1) ? 0 : (1 - prm.j0);
  const int last = rp.nyg - 1 - prm.j0; // local index of the global last row
  rp.hi = (rp.ny_local < last) ? rp.ny_local : last;
  if (prm.row_end > prm.row_begin) { // a caller-chosen range of owned rows (halo overlap)
    rp.lo = rp.lo > prm.row_begin ? rp.lo : prm.row_begin;
    rp.hi = rp.hi < prm.row_end ? rp.hi : prm.row_end;
  }
  if (rp.hi <= rp.lo) {
    // slab without a single computed row (can only be a 1-row edge slab): not supported here
    return hipSuccess;
  }
  // A row slab carries one halo row before owned row 0 and one after the last
  // owned row; a whole field has neither.
  const bool has_north_halo = prm.j0 > 0;
  const bool has_south_halo = prm.j0 + rp.ny_local < rp.nyg;
  rp.idx_lo = has_north_halo ? -(long)nx : 0;
  rp.idx_hi = (long)nx * (rp.ny_local + (has_south_halo ? 1 : 0)) - 1;
  rp.R = t.R;
  rp.nbands = (rp.hi - rp.lo + t.R - 1) / t.R;
  rp.nwc = (nx + 256 * t.V - 1) / (256 * t.V);
  rp.nlev = prm.nlev;
  rp.wpb = t.WPB;
  rp.uL = (prm.nlev + t.WPB - 1) / t.WPB;
  rp.uB = rp.nbands;
  rp.uW = rp.nwc;
  const long n_logical = (long)rp.uL * rp.uB * rp.uW;
  if (n_logical > 0x3fffffffL)
    return hipSuccess;
  rp.n_logical = (int)n_logical;
  rp.per_xcd = (rp.n_logical + 7) / 8;
  rp.order = t.ORDER;
  rp.xcd_remap = t.XCD;
  rp.zigzag = t.ZZ;
  rp.nt_interior = t.NTI;
  rp.lgroup = t.LG > 0 ? t.LG : 0;
#ifdef MIFC_MEASUREMENT_BUILD
  rp.exp_nohalo = t.XH;
  rp.exp_nostore = t.XS;
  rp.exp_noload = t.XL;
  rp.exp_store_aux = t.STA;
#endif
  rp.u = prm.f0;
  rp.v = prm.f1;
  rp.xm = prm.xmapr;
  rp.ym = prm.ymapr;
  rp.fc = (prm.op == ST_ABSVORT) ? prm.fcoriolis : nullptr;
  rp.rv = rv;
  rp.dv = dv;
  rp.ff = prm.out_ff;
  rp.n_undefined_ff = prm.n_undefined_ff;
  rp.in_stride = prm.in_level_stride;
  rp.out_stride = prm.out_level_stride;
  rp.all_defined = prm.all_defined;
  rp.undef = prm.undef;
  rp.n_undefined = prm.n_undefined;
  rp.partials = nullptr;
  // one big level with tests: the one-shot tiles leave their counts in prm.partials and one small launch adds them up (see StencilParams)
  auto counts_by_partials = [&](long units_per_level, bool level_walking = false) {
    // (the one-shot tiles index partials[unit of the launch]: level-major only in address order, lgroup == 0)
    const bool yes = prm.partials && prm.n_undefined && !prm.every_level_all_defined && !prm.out_ff && (level_walking || rp.lgroup == 0) &&
                     units_per_level >= 2048 && units_per_level * prm.nlev <= prm.partials_cap;
    rp.partials = yes ? prm.partials : nullptr;
    return yes;
  };
  int grid = rp.per_xcd * 8;

  if ((prm.out_ff || ragged) && t.K != 4)
    return hipSuccess; // (see above: only the split-role kernel has the third output / takes rows at any alignment)
  *handled = true;
  note_form(t.K == 1 ? "wind_oneshot" : t.K == 2 ? "wind_oneshot_tiles" : t.K == 3 ? "wind_levelwalk"
            : t.K == 4 ? (ragged ? "wind_split_ragged" : prm.out_ff ? "wind_split_ff" : "wind_split") : "wind_rows");
  if ((t.K == 1 || t.K == 2) && rp.fc) { // one-shot forms of absvort (relvort + the Coriolis parameter)
    const bool tiles = t.K == 2;
    rp.uB = (rp.hi - rp.lo + (tiles ? 7 : 3)) / (tiles ? 8 : 4);
    rp.uW = (nx + 255) / 256;
    const long units = (long)prm.nlev * rp.uB * rp.uW;
    if (units <= 0x3fffffffL) {
      rp.n_logical = (int)units;
      rp.per_xcd = (rp.n_logical + 7) / 8;
      grid = rp.per_xcd * 8;
      const bool chk = !prm.every_level_all_defined;
      if (tiles) {
        const bool partials = counts_by_partials((long)rp.uB * rp.uW);
        if (chk)
          hipLaunchKernelGGL((vortdiv_tile_kernel<true, true, false, true, 8, true>), dim3(grid), dim3(640), 0, stream, rp);
        else
          hipLaunchKernelGGL((vortdiv_tile_kernel<false, true, false, true, 8, true>), dim3(grid), dim3(640), 0, stream, rp);
        if (partials)
          (void)launch_count_partials_levels(prm.partials, rp.uB * rp.uW, prm.nlev, prm.n_undefined, stream);
      } else {
        if (chk)
          hipLaunchKernelGGL((vortdiv_oneshot_kernel<true, true, false, true, false, true>), dim3(grid), dim3(256), 0, stream, rp);
        else
          hipLaunchKernelGGL((vortdiv_oneshot_kernel<false, true, false, true, false, true>), dim3(grid), dim3(256), 0, stream, rp);
      }
      return hipGetLastError();
    }
    *handled = false; // a level or a launch beyond the 32-bit index range of these forms: rp was changed for them, so not the row kernel below either
    return hipSuccess;
  }
  if (t.K == 1 && prm.op == ST_JACOBIAN) { // one-shot form of the Jacobian
    rp.uB = (rp.hi - rp.lo + 3) / 4;
    rp.uW = (nx + 255) / 256;
    const long units = (long)prm.nlev * rp.uB * rp.uW;
    if (units <= 0x3fffffffL) {
      rp.n_logical = (int)units;
      rp.per_xcd = (rp.n_logical + 7) / 8;
      grid = rp.per_xcd * 8;
      if (prm.every_level_all_defined)
        hipLaunchKernelGGL((vortdiv_oneshot_kernel<false, true, false, true, true>), dim3(grid), dim3(256), 0, stream, rp);
      else
        hipLaunchKernelGGL((vortdiv_oneshot_kernel<true, true, false, true, true>), dim3(grid), dim3(256), 0, stream, rp);
      return hipGetLastError();
    }
    *handled = false; // a level or a launch beyond the 32-bit index range of these forms: rp was changed for them, so not the row kernel below either
    return hipSuccess;
  }
  if (t.K == 1 && !rp.fc && prm.op != ST_JACOBIAN) { // one-shot form: units are (level, block of 4 rows, 256-column segment)
    rp.uB = (rp.hi - rp.lo + 3) / 4;
    rp.uW = (nx + 255) / 256;
    const long units = (long)prm.nlev * rp.uB * rp.uW;
    if (units <= 0x3fffffffL) {
      rp.n_logical = (int)units;
      rp.per_xcd = (rp.n_logical + 7) / 8;
      grid = rp.per_xcd * 8;
      const bool chk = !prm.every_level_all_defined;
      const int sel = (chk ? 4 : 0) | (rv ? 2 : 0) | (dv ? 1 : 0);
      switch (sel) {
#define ONESHOT(C, WV, WD) \
  if (t.NT) \
    hipLaunchKernelGGL((vortdiv_oneshot_kernel<C, WV, WD, true>), dim3(grid), dim3(256), 0, stream, rp); \
  else \
    hipLaunchKernelGGL((vortdiv_oneshot_kernel<C, WV, WD, false>), dim3(grid), dim3(256), 0, stream, rp); \
  break
      case 1:
        ONESHOT(false, false, true);
      case 2:
        ONESHOT(false, true, false);
      case 3:
        ONESHOT(false, true, true);
      case 5:
        ONESHOT(true, false, true);
      case 6:
        ONESHOT(true, true, false);
      default:
        ONESHOT(true, true, true);
#undef ONESHOT
      }
      return hipGetLastError();
    }
    *handled = false; // a level or a launch beyond the 32-bit index range of these forms: rp was changed for them, so not the row kernel below either
    return hipSuccess;
  }
  if (t.K == 4 && nan_undef_tested) { // only a forced tuning gets here: the default selection above never picks K = 4 for these
    if (rp.fc) { // absvort has no other level-walking form: leave the request to the flat kernel
      *handled = false;
      return hipSuccess;
    }
    t.K = 3;
    t.D = 0;
    t.WPB = 8;
  }
  if (t.K == 3 && !rp.fc && prm.op != ST_JACOBIAN && !(rv && dv)) { // level-walking tiles, one output: the default shape only
    constexpr int NW = 12;
    rp.uB = (rp.hi - rp.lo + NW - 3) / (NW - 2);
    rp.uW = (nx + 255) / 256;
    rp.lgroup = (t.LG > 0 && t.LG < prm.nlev) ? t.LG : prm.nlev;
    const int nchunks = (prm.nlev + rp.lgroup - 1) / rp.lgroup;
    const long units = (long)nchunks * rp.uB * rp.uW;
    if (units <= 0x3fffffffL && (long)nx * (rp.ny_local + 2) < 0x7fffffffL) {
      rp.n_logical = (int)units;
      rp.per_xcd = (rp.n_logical + 7) / 8;
      grid = rp.per_xcd * 8;
      const bool chk = !prm.every_level_all_defined;
      if (chk && rv)
        hipLaunchKernelGGL((vortdiv_levelwalk_kernel<true, true, false, true, NW, 1, true>), dim3(grid), dim3(64 * NW), 0, stream, rp);
      else if (chk)
        hipLaunchKernelGGL((vortdiv_levelwalk_kernel<true, false, true, true, NW, 1, true>), dim3(grid), dim3(64 * NW), 0, stream, rp);
      else if (rv)
        hipLaunchKernelGGL((vortdiv_levelwalk_kernel<false, true, false, true, NW, 1, true>), dim3(grid), dim3(64 * NW), 0, stream, rp);
      else
        hipLaunchKernelGGL((vortdiv_levelwalk_kernel<false, false, true, true, NW, 1, true>), dim3(grid), dim3(64 * NW), 0, stream, rp);
      return hipGetLastError();
    }
    *handled = false; // a level or a launch beyond the 32-bit index range of these forms: rp was changed for them, so not the row kernel below either
    return hipSuccess;
  }
  if (t.K == 3 && !rp.fc && prm.op != ST_JACOBIAN && rv && dv) { // level-walking tiles: units are (level chunk, row block, 256-column segment)
    const int NWsel = (t.RB == 16 || t.RB == 12 || t.RB == 8) ? t.RB : 16; // RB doubles as the waves per workgroup here
    const bool halo_waves = t.ZZ != 0;                                     // ZZ doubles as "the halo rows have waves of their own"
    const int tile_rows = halo_waves ? NWsel - 2 : NWsel;
    rp.uB = (rp.hi - rp.lo + tile_rows - 1) / tile_rows;
    rp.uW = (nx + 255) / 256;
    rp.lgroup = (t.LG > 0 && t.LG < prm.nlev) ? t.LG : prm.nlev; // levels per workgroup
    const int nchunks = (prm.nlev + rp.lgroup - 1) / rp.lgroup;
    const long units = (long)nchunks * rp.uB * rp.uW;
    if (units <= 0x3fffffffL && (long)nx * (rp.ny_local + 2) < 0x7fffffffL) {
      rp.n_logical = (int)units;
      rp.per_xcd = (rp.n_logical + 7) / 8;
      grid = rp.per_xcd * 8;
      const bool chk = !prm.every_level_all_defined;
      const int pf = t.D >= 1 ? 2 : 1; // D doubles as the prefetch depth selector: D=0 -> one level ahead, D=1 (default) -> two
#define LEVELWALK(NW_, PF_)                                                                                                                          \
  if (chk && halo_waves)                                                                                                                             \
    hipLaunchKernelGGL((vortdiv_levelwalk_kernel<true, true, true, true, NW_, PF_, true>), dim3(grid), dim3(64 * NW_), 0, stream, rp);               \
  else if (chk)                                                                                                                                      \
    hipLaunchKernelGGL((vortdiv_levelwalk_kernel<true, true, true, true, NW_, PF_, false>), dim3(grid), dim3(64 * NW_), 0, stream, rp);              \
  else if (halo_waves)                                                                                                                               \
    hipLaunchKernelGGL((vortdiv_levelwalk_kernel<false, true, true, true, NW_, PF_, true>), dim3(grid), dim3(64 * NW_), 0, stream, rp);              \
  else                                                                                                                                               \
    hipLaunchKernelGGL((vortdiv_levelwalk_kernel<false, true, true, true, NW_, PF_, false>), dim3(grid), dim3(64 * NW_), 0, stream, rp)
      if (NWsel == 16 && pf == 2) {
        LEVELWALK(16, 2);
      } else if (NWsel == 16) {
        LEVELWALK(16, 1);
      } else if (NWsel == 12 && pf == 2) {
        LEVELWALK(12, 2);
      } else if (NWsel == 12) {
        LEVELWALK(12, 1);
      } else if (pf == 2) {
        LEVELWALK(8, 2);
      } else {
        LEVELWALK(8, 1);
      }
#undef LEVELWALK
      return hipGetLastError();
    }
    *handled = false; // a level or a launch beyond the 32-bit index range of these forms: rp was changed for them, so not the row kernel below either
    return hipSuccess;
  }
  if (t.K == 4 && prm.op == ST_JACOBIAN && env().has_vortdiv_tune) { // a forced tuning: the Jacobian has the default shape only
    *handled = false;
    return hipSuccess;
  }
  if (t.K == 4) { // split-role level-walking tiles (loader waves / compute waves)
    const bool single = !(rv && dv) || rp.fc; // one output: the default shape only (and its one-level-ahead sibling)
    const int tile_rows = (single || ragged || prm.out_ff) ? 12 : ((t.RB == 6 || t.RB == 8 || t.RB == 12 || t.RB == 14) ? t.RB : 10);
    const int nchunks_lg = (t.LG > 0 && t.LG < prm.nlev) ? t.LG : prm.nlev; // levels per workgroup
    const int nchunks = (prm.nlev + nchunks_lg - 1) / nchunks_lg;
    const long units = (long)nchunks * ((rp.hi - rp.lo + tile_rows - 1) / tile_rows) * ((nx + 255) / 256);
    if (units > 0x3fffffffL || (long)nx * (rp.ny_local + 2) >= 0x7fffffffL) { // 32-bit offsets inside a level: not this kernel's case
      *handled = false;
      return hipSuccess;
    }
    rp.uB = (rp.hi - rp.lo + tile_rows - 1) / tile_rows;
    rp.uW = (nx + 255) / 256;
    rp.lgroup = nchunks_lg;
    rp.n_logical = (int)units;
    rp.per_xcd = (rp.n_logical + 7) / 8;
    grid = rp.per_xcd * 8;
    const bool chk = !prm.every_level_all_defined;
    const bool partials = counts_by_partials((long)rp.uB * rp.uW, true); // big levels: counts by plain stores + one small launch
    auto finish = [&]() {
      if (partials)
        (void)launch_count_partials_levels(prm.partials, rp.uB * rp.uW, prm.nlev, prm.n_undefined, stream);
      return hipGetLastError();
    };
    const int pf = t.D >= 2 ? 3 : (t.D == 1 ? 2 : 1); // D selects how many levels the loaders run ahead
    const int nl = (t.WPB == 2 || t.WPB == 4) ? t.WPB : (tile_rows == 10 ? 4 : 2); // WPB doubles as the number of loader waves
#define SPLIT_AS(TR_, NL_, PF_, ...)                                                                                                       \
  if (chk)                                                                                                                                 \
    hipLaunchKernelGGL((vortdiv_split_kernel<true, true, TR_, NL_, PF_, ##__VA_ARGS__>), dim3(grid), dim3(64 * (TR_ + NL_)), 0, stream, rp); \
  else                                                                                                                                     \
    hipLaunchKernelGGL((vortdiv_split_kernel<false, true, TR_, NL_, PF_, ##__VA_ARGS__>), dim3(grid), dim3(64 * (TR_ + NL_)), 0, stream, rp)
    if (prm.out_ff) { // the fused pair plus the wind speed: the default shape
      SPLIT_AS(12, 2, 2, true, true, false, true);
      return finish();
    }
    if (prm.op == ST_JACOBIAN) { // the default shape, rows at any alignment or not
      if (ragged) {
        SPLIT_AS(12, 2, 2, true, false, false, false, true, true);
      } else {
        SPLIT_AS(12, 2, 2, true, false, false, false, false, true);
      }
      return finish();
    }
    if (ragged) { // rows at any alignment: the default shape
      if (rp.fc) {
        SPLIT_AS(12, 2, 2, true, false, true, false, true);
      } else if (rv && dv) {
        SPLIT_AS(12, 2, 2, true, true, false, false, true);
      } else if (rv) {
        SPLIT_AS(12, 2, 2, true, false, false, false, true);
      } else {
        SPLIT_AS(12, 2, 2, false, true, false, false, true);
      }
      return finish();
    }
    if (single) {
      const bool pf1 = pf == 1;
      if (rp.fc) {
        if (pf1) { SPLIT_AS(12, 2, 1, true, false, true); } else { SPLIT_AS(12, 2, 2, true, false, true); }
      } else if (rv) {
        if (pf1) { SPLIT_AS(12, 2, 1, true, false, false); } else { SPLIT_AS(12, 2, 2, true, false, false); }
      } else {
        if (pf1) { SPLIT_AS(12, 2, 1, false, true, false); } else { SPLIT_AS(12, 2, 2, false, true, false); }
      }
      return finish();
    }
#define SPLIT(TR_, NL_, PF_) SPLIT_AS(TR_, NL_, PF_)
    if (tile_rows == 6) {
      if (pf == 3) { SPLIT(6, 2, 3); } else if (pf == 2) { SPLIT(6, 2, 2); } else { SPLIT(6, 2, 1); }
    } else if (tile_rows == 8) {
      if (pf == 3) { SPLIT(8, 2, 3); } else if (pf == 2) { SPLIT(8, 2, 2); } else { SPLIT(8, 2, 1); }
    } else if (tile_rows == 12 && nl == 4) {
      if (pf >= 2) { SPLIT(12, 4, 2); } else { SPLIT(12, 4, 1); }
    } else if (tile_rows == 12) {
      if (pf == 3) { SPLIT(12, 2, 3); } else if (pf == 2) { SPLIT(12, 2, 2); } else { SPLIT(12, 2, 1); }
    } else if (tile_rows == 14) { // 16 waves: two loaders at most
      if (pf >= 2) { SPLIT(14, 2, 2); } else { SPLIT(14, 2, 1); }
    } else if (nl == 2) {
      if (pf >= 2) { SPLIT(10, 2, 2); } else { SPLIT(10, 2, 1); }
    } else {
      if (pf == 3) { SPLIT(10, 4, 3); } else if (pf == 2) { SPLIT(10, 4, 2); } else { SPLIT(10, 4, 1); }
    }
#undef SPLIT
#undef SPLIT_AS
    return finish();
  }
  if (t.K == 2 && !rp.fc && prm.op != ST_JACOBIAN && t.RB == 14 && rv && dv) { // one-shot tiles of 14 rows (16-wave workgroups)
    constexpr int RB = 14;
    rp.uB = (rp.hi - rp.lo + RB - 1) / RB;
    rp.uW = (nx + 255) / 256;
    const long units = (long)prm.nlev * rp.uB * rp.uW;
    if (units <= 0x3fffffffL) {
      rp.n_logical = (int)units;
      rp.per_xcd = (rp.n_logical + 7) / 8;
      grid = rp.per_xcd * 8;
      const bool partials = counts_by_partials((long)rp.uB * rp.uW);
      if (prm.every_level_all_defined)
        hipLaunchKernelGGL((vortdiv_tile_kernel<false, true, true, true, RB>), dim3(grid), dim3(64 * (RB + 2)), 0, stream, rp);
      else
        hipLaunchKernelGGL((vortdiv_tile_kernel<true, true, true, true, RB>), dim3(grid), dim3(64 * (RB + 2)), 0, stream, rp);
      if (partials)
        (void)launch_count_partials_levels(prm.partials, rp.uB * rp.uW, prm.nlev, prm.n_undefined, stream);
      return hipGetLastError();
    }
    *handled = false; // a level or a launch beyond the 32-bit index range of these forms: rp was changed for them, so not the row kernel below either
    return hipSuccess;
  }
  if (t.K == 2 && !rp.fc && prm.op != ST_JACOBIAN) { // one-shot tiles: units are (level, block of 8 rows, 256-column segment)
    constexpr int RB = 8;
    rp.uB = (rp.hi - rp.lo + RB - 1) / RB;
    rp.uW = (nx + 255) / 256;
    const long units = (long)prm.nlev * rp.uB * rp.uW;
    if (units <= 0x3fffffffL) {
      rp.n_logical = (int)units;
      rp.per_xcd = (rp.n_logical + 7) / 8;
      grid = rp.per_xcd * 8;
      const bool chk = !prm.every_level_all_defined;
      const int sel = (chk ? 4 : 0) | (rv ? 2 : 0) | (dv ? 1 : 0);
      const bool partials = counts_by_partials((long)rp.uB * rp.uW);
      switch (sel) {
#define TILE(C, WV, WD) \
  hipLaunchKernelGGL((vortdiv_tile_kernel<C, WV, WD, true, RB>), dim3(grid), dim3(64 * (RB + 2)), 0, stream, rp); \
  break
      case 1:
        TILE(false, false, true);
      case 2:
        TILE(false, true, false);
      case 3:
        TILE(false, true, true);
      case 5:
        TILE(true, false, true);
      case 6:
        TILE(true, true, false);
      default:
        TILE(true, true, true);
#undef TILE
      }
      if (partials)
        (void)launch_count_partials_levels(prm.partials, rp.uB * rp.uW, prm.nlev, prm.n_undefined, stream);
      return hipGetLastError();
    }
    *handled = false; // a level or a launch beyond the 32-bit index range of these forms: rp was changed for them, so not the row kernel below either
    return hipSuccess;
  }
  if (prm.op == ST_JACOBIAN) {
    if (t.V == 3)
      t.V = 2;
    if (prm.every_level_all_defined)
      launch_jacobian<false>(rp, t, grid, stream);
    else
      launch_jacobian<true>(rp, t, grid, stream);
    return hipGetLastError();
  }
  if (prm.every_level_all_defined)
    launch_outputs<false>(rp, t, grid, stream);
  else
    launch_outputs<true>(rp, t, grid, stream);
  return hipGetLastError();
}

} // namespace mifc
