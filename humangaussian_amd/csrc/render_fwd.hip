// render_fwd.hip - front-to-back alpha blending (stage F6), CELL-ROW mapping.
//
// Upstream runs one thread block per 16x16 tile, every thread walks the tile's whole depth-sorted list and
// evaluates every Gaussian at its pixel; on an avatar a Gaussian reaches alpha >= 1/255 on ~34 of those 256
// pixels.  Here the tile is cut into 16 cells of 4x4 pixels and the sort kernel has written, per tile, 16
// depth-ordered CELL LISTS holding only the records that can reach each cell (cellmask.h, exact), and has
// placed every non-empty cell into a table by LENGTH CLASS.  A wave64 is FOUR ROWS of 16 lanes; it takes four
// cells of the same class - any cells of any tiles, longest classes first - row j = one cell, lane = one pixel,
// and every row streams ITS OWN cell list (waves are independent: no workgroup barrier).  With the tile as
// the unit of work (first version) a CU got ~3 active tiles of very different weight and the kernel ended
// with the CU that drew the heaviest ones (48 us, half of it tail); rows of equal length also end together.
//   1. the row's 16 lanes gather the next 16 records of the list (index load two batches ahead, 48 B record
//      gather one batch ahead: both latencies hide behind the blend of the current batch) and stage them in
//      a wave-private LDS slice;
//   2. 16 blend iterations: lane reads "its row's record u" (a row-uniform LDS address: four different
//      records per ds_read_b128, one per row), upstream's sequential per-pixel loop with predication;
//      T is the only loop-carried dependency;
//   3. a row stops when its 16 pixels have all terminated (T < 1e-4) or its list ends; the wave stops with
//      its last row.
// When a backward will follow, the per-pixel state (T, C, D, W) is stored at every HGS_SEGLEN-th entry of the
// cell list (the backward's work items start there); n_contrib keeps the tile-list position of the pixel's
// last contributor (upstream's meaning).
//
// Roofline: VALU issue (about 22 instructions per row iteration); a record is evaluated only in the cells
// it can reach: 57 lane slots per tile entry instead of 256 (tools/cell_stats.py).  HBM traffic per tile
// entry: 4 B index + 48 B gather per cell reached (L2-served), 24 B/pixel out, 24 B/pixel per HGS_SEGLEN (128)
// cell-list entries of state.
#include "hgs_common.h"

namespace {

typedef float hgs_fwd_f32x2 __attribute__((ext_vector_type(2)));

struct PixState {
  hgs_fwd_f32x2 C01, C2D;       // (C0 C1), (C2 D): updated as pairs - v_pk_fma_f32 - written out explicitly (left to the
  float Wt;                     //  SLP vectoriser the packing came and went with unrelated edits of this function)
  float T;          // running transmittance while the pixel is alive (> 0); once it has terminated: MINUS the transmittance
                    // behind its last blended record (what the background sees); 0 outside the image
  uint32_t last;
};

// Blend one staged record into the lane's pixel, fully predicated.  r2.w carries the record's 1-based
// position in the TILE's list (upstream's `contributor` count); pad records have opacity 0 and never blend.
//
// Upstream's rule - skip (power > 0 or alpha < 1/255), stop at test_T < 1e-4 WITHOUT blending that Gaussian,
// nothing behind it counts - as arithmetic on ONE carried value: a terminated pixel's T turns NEGATIVE (it keeps its
// magnitude: the transmittance the background sees), so every later test_T is <= 0 < 1e-4 and blends nothing.  (Until
// round 6 it became 0 and a second carried value, updated per record, remembered the magnitude: one v_cndmask per
// record more for the same bits.)  The carried chain per record is v_mul -> v_cmp -> v_cndmask;
// with a separate `done` flag it ran through four scalar mask operations per record (VALU -> SALU -> VALU
// round trips), which is what a wave alone on its SIMD - the tail of this kernel - was waiting for.
// Same values as the flag formulation, bit for bit (a skipped record multiplies T by exactly 1).
__device__ __forceinline__ void blend_one(PixState& s, float pxf, float pyf, const float4 r0,
                                          const float4 r1, const float4 r2) {
  float G, alpha, m2, m3;
  const bool keep = hgs_eval_alpha(r0.x - pxf, r0.y - pyf, r0.z, r0.w, r1.x, r1.y, G, alpha, m2, m3);
  const float ak = keep ? alpha : 0.0f;                 // off the carried chain
  const float test_T = s.T * (1.0f - ak);
  const bool ok = test_T >= HGS_T_EPS;                  // false from the terminating record on
  const float wgt = ok ? ak * s.T : 0.0f;
  s.T = ok ? test_T : -__builtin_fabsf(s.T);            // (a skipped record: ak = 0, test_T = T: nothing changes)
  const bool upd = keep && ok;
  const hgs_fwd_f32x2 rg = {r1.z, r1.w}, bd = {r2.x, r2.y}, w2 = {wgt, wgt};
  s.C01 = __builtin_elementwise_fma(rg, w2, s.C01);
  s.C2D = __builtin_elementwise_fma(bd, w2, s.C2D);
  s.Wt += wgt;
  s.last = upd ? __float_as_uint(r2.w) : s.last;
}

}  // namespace

#define HGS_FWD_C4 4            // length classes 0 .. HGS_FWD_C4 - 1 (>= 13 batches of 16 records) are walked four records at a time
typedef unsigned hgs_u32x2 __attribute__((ext_vector_type(2)));

// ---- LONG cell lists (length classes < HGS_FWD_C4: more than 192 records): ONE cell per wave, FOUR RECORDS per
// iteration.  A wave issues one instruction per 4 cycles at best, so a 400-record list walked one record at a time
// is a 20 us chain even with the SIMD to itself - the tail of this kernel (max wave 46 us at 100k Gaussians).
// Here row r evaluates record 4 k + r at all 16 pixels; the transmittance in front of it is the carried T times the
// product of (1 - alpha) of the rows before it - two cross-row exchanges (v_permlane16_swap / 32_swap, no LDS).
// Upstream's stop rule needs no flag: products only shrink, so once a row fails test_T >= 1e-4 every later row
// of the iteration fails with it.  Each lane sums the contributions of ITS records; the four rows are added at the
// state boundaries (every 64 records = every staged block) and at the end, in a fixed order.
__device__ __forceinline__ float hgs_xor16(float x, int lane) {
  const hgs_u32x2 s2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float((lane & 16) ? s2.x : s2.y);
}
__device__ __forceinline__ float hgs_xor32(float x, int lane) {
  const hgs_u32x2 s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float((lane & 32) ? s2.x : s2.y);
}
// sum over the four rows (lanes l, l ^ 16, l ^ 32, l ^ 48), same association in every lane: (row 0 + row 1) + (row 2 + row 3)
// (the swap leaves "even row" / "odd row" of a pair in its two results in BOTH rows: no partner select)
__device__ __forceinline__ float hgs_rows_sum(float x, int lane) {
  (void)lane;
  const hgs_u32x2 s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float s01 = __uint_as_float(s16.x) + __uint_as_float(s16.y);       // row (even) + row (odd)
  const hgs_u32x2 s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s01), __float_as_uint(s01), false, false);
  return __uint_as_float(s32.x) + __uint_as_float(s32.y);                  // rows (0 + 1) + rows (2 + 3)
}

template <bool STORE>
__device__ __forceinline__ void render_fwd_cell4(const View& v, const Layout& L, uint32_t key,
                                                 const SortRec* __restrict__ recs_all,
                                                 float* __restrict__ cstate,
                                                 float* __restrict__ out_color,
                                                 float* __restrict__ out_depth,
                                                 float* __restrict__ out_alpha, float4* __restrict__ s_rec_w) {
  const int lane = (int)threadIdx.x & 63;
  const int r = lane >> 4, i = lane & 15;
  const int g = (int)(key >> 4), c = (int)(key & 15u);
  const int bview = g / v.T, t = g % v.T;
  const size_t HW = (size_t)v.H * v.W;
  const int px = (t % v.grid_x) * HGS_TILE + (c & 3) * HGS_CELL + (i & 3);
  const int py = (t / v.grid_x) * HGS_TILE + (c >> 2) * HGS_CELL + (i >> 2);
  const bool inside = (px < v.W) && (py < v.H);
  const float pxf = (float)px, pyf = (float)py;
  const uint32_t tstart1 = L.tile_start[g] - 1u;
  const CellInfo ci = L.cell_info[key];
  const uint32_t len = ci.len, sbase = ci.sbase;
  const uint2* __restrict__ list = L.cell_list + ci.base;
  const float4* __restrict__ recs = reinterpret_cast<const float4*>(recs_all);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

  float T = inside ? 1.0f : 0.0f;                 // carried transmittance of the pixel (the same in its four lanes)
  float tmin = 1.0f;                              // transmittance behind the lane's last blended record
  float C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, Wt = 0.f;
  uint32_t last = 0;

  // Unconditional loads (a lane beyond the list re-reads a valid slot; its record is neutralised - opacity 0 - when it
  // is staged, one block later, when the data has arrived anyway), and the index load goes BEFORE the gather of the
  // same iteration: vmcnt counts in order, so waiting for the youngest load waits for everything before it.
  auto load_idx = [&](uint32_t e) { return list[(e < len) ? e : 0u].x; };
  auto gather = [&](uint32_t idx_raw, bool valid, float4& r0, float4& r1, float4& r2) {
    const uint32_t idx = valid ? idx_raw : tstart1 + 1u;               // (the tile's first record)
    r0 = recs[3 * (size_t)idx]; r1 = recs[3 * (size_t)idx + 1];
    const float4 t2 = recs[3 * (size_t)idx + 2];
    r2 = make_float4(t2.x, t2.y, t2.z, __uint_as_float(valid ? idx - tstart1 : 0xffffffffu));
  };
  // blocks of 64 records, one per lane; indices two and three blocks ahead, records one block ahead
  float4 c0, c1, c2;
  gather(load_idx((uint32_t)lane), (uint32_t)lane < len, c0, c1, c2);
  uint32_t idx_a = load_idx(64u + (uint32_t)lane), idx_b = load_idx(128u + (uint32_t)lane);

  for (uint32_t it0 = 0; it0 < len; it0 += 64u) {
    if (__ballot(T != 0.0f) == 0ull) break;          // every pixel has terminated
    if (STORE && it0 > 0 && (it0 % HGS_SEGLEN) == 0) {      // pixel state in front of record it0 (blocks of 64; HGS_SEGLEN is a multiple)
      const float s0 = hgs_rows_sum(C0, lane), s1 = hgs_rows_sum(C1, lane), s2 = hgs_rows_sum(C2, lane);
      const float sd = hgs_rows_sum(D, lane), sw = hgs_rows_sum(Wt, lane);
      if (r == 0) {
        float* cs = cstate + (size_t)(sbase + it0 / HGS_SEGLEN - 1) * HGS_CSTATE_FLOATS + i;
        cs[0 * 16] = T; cs[1 * 16] = s0; cs[2 * 16] = s1; cs[3 * 16] = s2; cs[4 * 16] = sd; cs[5 * 16] = sw;
      }
    }
    __builtin_amdgcn_wave_barrier();                 // the previous block's LDS reads are done
    c1.y = (__float_as_uint(c2.w) == 0xffffffffu) ? 0.0f : c1.y;      // pad record: never blends
    s_rec_w[3 * lane + 0] = c0; s_rec_w[3 * lane + 1] = c1; s_rec_w[3 * lane + 2] = c2;
    const uint32_t idx_n = load_idx(it0 + 192u + (uint32_t)lane);
    gather(idx_a, it0 + 64u + (uint32_t)lane < len, c0, c1, c2);
    idx_a = idx_b; idx_b = idx_n;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float4* __restrict__ myrec = s_rec_w + 3 * r;          // record 4 k + r of the block
    float4 ra = myrec[0], rb = myrec[1], rc = myrec[2];
#pragma unroll 2
    for (int k = 0; k < 16; ++k) {
      float4 na = zero4, nb = zero4, nc = zero4;
      if (k + 1 < 16) { na = myrec[12 * (k + 1) + 0]; nb = myrec[12 * (k + 1) + 1]; nc = myrec[12 * (k + 1) + 2]; }
      __builtin_amdgcn_sched_barrier(0x7f);
      float G, alpha, m2, m3;
      const bool keep = hgs_eval_alpha(ra.x - pxf, ra.y - pyf, ra.z, ra.w, rb.x, rb.y, G, alpha, m2, m3);
      const float ak = keep ? alpha : 0.0f;
      const float a = 1.0f - ak;
      // exclusive prefix of a over the four rows of this pixel, and the product of all four.  v_permlane16_swap(a, a)
      // leaves the EVEN row's value of every row pair in its first result and the ODD row's in the second - in both rows
      // - so the pair's product needs no "which one is my partner" select (nor does the four-row product behind
      // v_permlane32_swap): two v_cndmask per iteration less than partner = select(...), a * partner; the same products
      // (multiplication commutes: the same bits).
      const hgs_u32x2 s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(a), false, false);
      const float a_even = __uint_as_float(s16.x), a_odd = __uint_as_float(s16.y);
      const float p01 = a_even * a_odd;                    // rows (0, 1) or (2, 3): the same value in both rows
      const hgs_u32x2 s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(p01), __float_as_uint(p01), false, false);
      const float p_lo = __uint_as_float(s32.x), p_hi = __uint_as_float(s32.y);      // rows (0, 1) / rows (2, 3), in every row
      const float excl = ((lane & 16) ? a_even : 1.0f) * ((lane & 32) ? p_lo : 1.0f);
      const float Tb = T * excl;                           // transmittance in front of this record
      const float test_T = Tb * a;
      const bool ok = test_T >= HGS_T_EPS;
      const float wgt = ok ? ak * Tb : 0.0f;
      const bool upd = keep && ok;
      C0 = __builtin_fmaf(rb.z, wgt, C0);
      C1 = __builtin_fmaf(rb.w, wgt, C1);
      C2 = __builtin_fmaf(rc.x, wgt, C2);
      D = __builtin_fmaf(rc.y, wgt, D);
      Wt += wgt;
      tmin = upd ? test_T : tmin;
      last = upd ? __float_as_uint(rc.w) : last;
      // the pixel goes on iff all four rows passed (a skipped record passes with a = 1)
      // (on the scalar unit: AND of the four rows' masks, replicated to the four rows, used as the select mask directly -
      //  `(all4 >> (lane & 15)) & 1` cost four v_and and a 64-bit compare per iteration)
      const unsigned long long okm = __ballot(ok);
      const uint32_t ok2 = (uint32_t)okm & (uint32_t)(okm >> 32);          // rows 0 & 2 | rows 1 & 3
      const uint32_t ok4 = (ok2 & (ok2 >> 16) & 0xffffu) * 0x10001u;       // all four rows, in both halves
      const bool go = __builtin_amdgcn_inverse_ballot_w64(((unsigned long long)ok4 << 32) | ok4);
      T = go ? T * (p_lo * p_hi) : 0.0f;
      ra = na; rb = nb; rc = nc;
    }
  }
  // the four rows of a pixel -> its outputs
  const float s0 = hgs_rows_sum(C0, lane), s1 = hgs_rows_sum(C1, lane), s2 = hgs_rows_sum(C2, lane);
  const float sd = hgs_rows_sum(D, lane), sw = hgs_rows_sum(Wt, lane);
  float tm = fminf(tmin, hgs_xor16(tmin, lane));
  tm = fminf(tm, hgs_xor32(tm, lane));
  uint32_t lm = max(last, __float_as_uint(hgs_xor16(__uint_as_float(last), lane)));
  lm = max(lm, __float_as_uint(hgs_xor32(__uint_as_float(lm), lane)));
  if (inside && r == 0) {
    const float* __restrict__ bg = v.cam[bview].bg;
    const size_t pix = (size_t)py * v.W + px, o1 = (size_t)bview * HW, o3 = 3 * o1;
    out_color[o3 + 0 * HW + pix] = s0 + tm * bg[0];
    out_color[o3 + 1 * HW + pix] = s1 + tm * bg[1];
    out_color[o3 + 2 * HW + pix] = s2 + tm * bg[2];
    out_depth[o1 + pix] = sd;
    out_alpha[o1 + pix] = sw;
    L.n_contrib[o1 + pix] = lm;
  }
}

// Cell q of length class c in the forward's order: the dies' tables of the class, one behind the other.  pre = the
// workgroup's LDS table [class][HGS_FWD_PRE_STRIDE]: [x] = cells of the class in the tables of dies < x, [8] = all.
#define HGS_FWD_PRE_STRIDE 12
__device__ __forceinline__ uint32_t fwd_cell_key(const Layout& L, size_t dcap, const uint32_t* pre, int c, uint32_t q) {
  const uint4 p0 = *reinterpret_cast<const uint4*>(pre + c * HGS_FWD_PRE_STRIDE);
  const uint4 p1 = *reinterpret_cast<const uint4*>(pre + c * HGS_FWD_PRE_STRIDE + 4);
  uint32_t x = 0, base = 0;
  if (q >= p0.y) { x = 1; base = p0.y; }
  if (q >= p0.z) { x = 2; base = p0.z; }
  if (q >= p0.w) { x = 3; base = p0.w; }
  if (q >= p1.x) { x = 4; base = p1.x; }
  if (q >= p1.y) { x = 5; base = p1.y; }
  if (q >= p1.z) { x = 6; base = p1.z; }
  if (q >= p1.w) { x = 7; base = p1.w; }
  return L.fwd_cells[((size_t)x * HGS_NFC + (size_t)c) * dcap + (q - base)];
}

// One wave = four cells of one length class.
template <bool STORE>
__device__ __forceinline__ void render_fwd_cells(const View& v, const Layout& L, const uint32_t* pre, uint32_t wave_id, int w,
                                                 const SortRec* __restrict__ recs_all,
                                                 float* __restrict__ cstate,
                                                 float* __restrict__ out_color,
                                                 float* __restrict__ out_depth,
                                                 float* __restrict__ out_alpha, float4* __restrict__ s_rec_w) {
  const int lane = (int)threadIdx.x & 63;
  const int j = lane >> 4, i = lane & 15;
  // item q of the class tables, longest class first
  uint32_t q = 4u * wave_id + (uint32_t)j;            // (wave_id counts from the first wave behind the long cells)
  uint32_t key = 0;
  bool have = false;
  int mycls = HGS_NFC;
  {
    uint32_t total = 0;
#pragma unroll
    for (int c = HGS_FWD_C4; c < HGS_NFC; ++c) {
      const uint32_t nc = pre[c * HGS_FWD_PRE_STRIDE + HGS_NXCD];
      if (!have && q < nc) { have = true; mycls = c; }
      q -= have ? 0u : nc;
      total += nc;
    }
    if (4u * wave_id >= total) return;                  // surplus wave
    // (one unconditional load: a row without a cell reads a valid slot and drops it)
    const uint32_t k_ = fwd_cell_key(L, hgs_die_cells(v.TT), pre, have ? mycls : HGS_FWD_C4, have ? q : 0u);
    key = have ? k_ : 0u;
  }
  const int g = (int)(key >> 4), c = (int)(key & 15u);
  const int bview = g / v.T, t = g % v.T;
  const size_t HW = (size_t)v.H * v.W;
  const int px = (t % v.grid_x) * HGS_TILE + (c & 3) * HGS_CELL + (i & 3);
  const int py = (t / v.grid_x) * HGS_TILE + (c >> 2) * HGS_CELL + (i >> 2);
  const bool inside = have && (px < v.W) && (py < v.H);
  const float pxf = (float)px, pyf = (float)py;

  const uint32_t tstart1 = have ? L.tile_start[g] - 1u : 0u;          // record index - tstart1 = 1-based list position
  uint32_t len = 0, base = 0, sbase = 0;
  if (have) {
    const CellInfo ci = L.cell_info[key];
    len = ci.len; base = ci.base; sbase = ci.sbase;
  }
  const uint2* __restrict__ list = L.cell_list + base;
  const float4* __restrict__ recs = reinterpret_cast<const float4*>(recs_all);
  float4* __restrict__ srow = s_rec_w + j * HGS_ROW_F4;               // this row's 16 staged records

  PixState s;
  s.T = inside ? 1.0f : 0.0f;
  s.C01 = s.C2D = (hgs_fwd_f32x2){0.f, 0.f};
  s.Wt = 0.f;
  s.last = 0;

  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // Software pipeline of the record stream: the index of batch b + 3 and the 48 B record gather of batch b + 2
  // are issued while batch b is blended.
  // Unconditional loads, index load before the gather of the same iteration (see render_fwd_cell4)
  const uint32_t safe_rec = have ? tstart1 + 1u : 0u;
  auto load_idx = [&](uint32_t e) { return list[(e < len) ? e : 0u].x; };
  auto gather = [&](uint32_t idx_raw, bool valid, float4& r0, float4& r1, float4& r2) {
    const uint32_t idx = valid ? idx_raw : safe_rec;
    r0 = recs[3 * (size_t)idx]; r1 = recs[3 * (size_t)idx + 1];
    const float4 t2 = recs[3 * (size_t)idx + 2];
    r2 = make_float4(t2.x, t2.y, t2.z, __uint_as_float(valid ? idx - tstart1 : 0xffffffffu));
  };
  float4 c0, c1, c2, d0, d1, d2;                    // records of batch b (c) and b + 1 (d)
  gather(load_idx((uint32_t)i), (uint32_t)i < len, c0, c1, c2);
  gather(load_idx(HGS_RB + (uint32_t)i), HGS_RB + (uint32_t)i < len, d0, d1, d2);
  uint32_t idx_a = load_idx(2 * HGS_RB + (uint32_t)i), idx_b = load_idx(3 * HGS_RB + (uint32_t)i);      // batches b + 2, b + 3

  for (uint32_t it0 = 0;; it0 += HGS_RB) {
    // rows still at work: list not exhausted and a pixel not finished
    const unsigned long long act = __ballot((it0 < len) && (s.T > 0.0f));
    if (act == 0ull) break;
    const bool row_on = ((act >> (lane & 48)) & 0xffffull) != 0ull;
    if (STORE && row_on && it0 > 0 && (it0 % HGS_SEGLEN) == 0) {
      float* cs = cstate + (size_t)(sbase + it0 / HGS_SEGLEN - 1) * HGS_CSTATE_FLOATS + i;
      cs[0 * 16] = s.T; cs[1 * 16] = s.C01.x; cs[2 * 16] = s.C01.y; cs[3 * 16] = s.C2D.x; cs[4 * 16] = s.C2D.y; cs[5 * 16] = s.Wt;
    }
    __builtin_amdgcn_wave_barrier();                 // the previous batch's LDS reads are done
    c1.y = (__float_as_uint(c2.w) == 0xffffffffu) ? 0.0f : c1.y;      // pad record: never blends
    // PAIRED stage: records 2p and 2p + 1 of the batch share six float4, field by field -
    //   (mx0 mx1 my0 my1 | qa0 qa1 qb0 qb1 | qc0 qc1 op0 op1 | r0 g0 r1 g1 | b0 d0 b1 d1 | pos0 pos1 - -)
    // - the geometry fields interleaved ACROSS the two records, the colour fields kept per record: the compiler packs the
    // alpha arithmetic of two records into v_pk_* instructions (dx0 dx1, ...) and the accumulation of one record's colour
    // into channel pairs ((C0 C1) += (r g) w, (C2 D) += (b d) w) either way, and with one record per three float4 it first
    // had to shuffle every geometry operand pair together (7 v_mov per record in a loop of ~28 instructions).
    {
      float* pb = reinterpret_cast<float*>(srow + 6 * (i >> 1)) + (i & 1);      // field f of record e: pb[2 f]
      float* pc = pb + (i & 1);                                                  // colour pair k of record e: pc[..]
      pb[0] = c0.x; pb[2] = c0.y; pb[4] = c0.z; pb[6] = c0.w;
      pb[8] = c1.x; pb[10] = c1.y; pc[12] = c1.z; pc[13] = c1.w;
      pc[16] = c2.x; pc[17] = c2.y; pb[20] = c2.w;
    }
    c0 = d0; c1 = d1; c2 = d2;
    const uint32_t idx_n = load_idx(it0 + 4 * HGS_RB + (uint32_t)i);
    gather(idx_a, row_on && (it0 + 2 * HGS_RB + (uint32_t)i < len), d0, d1, d2);
    idx_a = idx_b; idx_b = idx_n;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // pairs of records; the six LDS reads of pair k + 1 are in flight while pair k is blended
    float4 P[6], N[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) P[m] = srow[m];
#pragma unroll
    for (int u0 = 0; u0 < HGS_RB; u0 += 2) {
      if (u0 + 2 < HGS_RB) {
#pragma unroll
        for (int m = 0; m < 6; ++m) N[m] = srow[3 * (u0 + 2) + m];
      }
      // the machine scheduler would sink those reads to their first use (and expose one LDS round trip per pair):
      // LDS instructions may not cross this point, everything else may
      __builtin_amdgcn_sched_barrier(0x7f);
      blend_one(s, pxf, pyf, make_float4(P[0].x, P[0].z, P[1].x, P[1].z), make_float4(P[2].x, P[2].z, P[3].x, P[3].y),
                make_float4(P[4].x, P[4].y, 0.0f, P[5].x));
      blend_one(s, pxf, pyf, make_float4(P[0].y, P[0].w, P[1].y, P[1].w), make_float4(P[2].y, P[2].w, P[3].z, P[3].w),
                make_float4(P[4].z, P[4].w, 0.0f, P[5].y));
      __builtin_amdgcn_sched_barrier(0x7f);
      if (u0 + 2 < HGS_RB) {
#pragma unroll
        for (int m = 0; m < 6; ++m) P[m] = N[m];
      }
    }
  }

  if (inside) {
    const float* __restrict__ bg = v.cam[bview].bg;
    const size_t pix = (size_t)py * v.W + px, o1 = (size_t)bview * HW, o3 = 3 * o1;
    const float Tout = __builtin_fabsf(s.T);          // transmittance behind the last blended record
    out_color[o3 + 0 * HW + pix] = s.C01.x + Tout * bg[0];
    out_color[o3 + 1 * HW + pix] = s.C01.y + Tout * bg[1];
    out_color[o3 + 2 * HW + pix] = s.C2D.x + Tout * bg[2];
    out_depth[o1 + pix] = s.C2D.y;
    out_alpha[o1 + pix] = s.Wt;
    L.n_contrib[o1 + pix] = s.last;
  }
}

// The pixels of EMPTY cells (and of every cell when the lists are invalid): background, zero depth / alpha.
// One workgroup per tile, thread = pixel in row-major order (coalesced stores).
__device__ __forceinline__ void render_fwd_background(const View& v, const Layout& L, bool overflow, int g,
                                                      float* __restrict__ out_color, float* __restrict__ out_depth,
                                                      float* __restrict__ out_alpha) {
  const int bview = g / v.T, t = g % v.T;
  const int lx = (int)threadIdx.x & 15, ly = (int)threadIdx.x >> 4;
  const int px = (t % v.grid_x) * HGS_TILE + lx, py = (t / v.grid_x) * HGS_TILE + ly;
  if (px >= v.W || py >= v.H) return;
  bool empty = overflow || L.tile_n[g] == 0u;
  if (!empty) empty = L.cell_info[(size_t)g * 16 + (ly >> 2) * 4 + (lx >> 2)].len == 0u;
  if (!empty) return;
  const float* __restrict__ bg = v.cam[bview].bg;
  const size_t HW = (size_t)v.H * v.W, pix = (size_t)py * v.W + px, o1 = (size_t)bview * HW, o3 = 3 * o1;
  out_color[o3 + 0 * HW + pix] = bg[0];
  out_color[o3 + 1 * HW + pix] = bg[1];
  out_color[o3 + 2 * HW + pix] = bg[2];
  out_depth[o1 + pix] = 0.0f;
  out_alpha[o1 + pix] = 0.0f;
  L.n_contrib[o1 + pix] = 0u;
}

// Launch: blocks [0, cell_blocks) hold four PERSISTENT cell waves each: wave w takes work items w, w + W, ... (long cells
// first, one each; then groups of four cells by descending length class) - the item count is only known on the device,
// and a capacity-sized grid cost more in empty waves (56k of them, ~1 us each) than the blending itself; blocks
// [cell_blocks, cell_blocks + B*T) write the background of the empty cells.
//
// Blocks b, b + ncu, b + 2 ncu, ... share a CU (ncu = cell_blocks / 4), the four waves of a block sit on its four
// SIMDs (which wave on which rotates from block to block): the waves of a block take ADJACENT items of one round,
// the blocks of a CU different rounds, in SNAKE order (round rho: rho S + slot for even rho, rho S + S - 1 - slot for
// odd rho) - so every SIMD gets one item of every round, a heavy + light mix of about the same total, whatever the
// rotation.  A view has about one item per wave (3657 for 4096 at configs[1]; 6 us of SIMD time each): the kernel
// ends with its most loaded SIMD (per-SIMD totals 49 .. 87 us of wave time for a mean of 65), and every other way of
// handing the items out that was tried lost or changed nothing - LDS tickets per workgroup (also one 16-wave
// workgroup per CU: +3 us), 8 or 12 waves per CU (+7 us), issue priorities by item weight, groups of the longest
// short cells ahead of the long cells they cost as much as, a die's workgroups on the die's own tables (+3 us) -
// EXPERIMENTS.md, round 4.
#define HGS_RENDER_FWD_KERNEL(NAME, STORE)                                                                 \
  extern "C" __global__ void __launch_bounds__(HGS_FWD_THREADS) NAME(                                       \
      View v, Layout L, uint32_t cell_blocks, hgs_status* __restrict__ status,                               \
      hgs_status* __restrict__ status_host, const SortRec* __restrict__ recs, float* __restrict__ cstate, float* __restrict__ out_color,           \
      float* __restrict__ out_depth, float* __restrict__ out_alpha) {                                        \
    __shared__ float4 s_rec[HGS_FWD_THREADS / 64][4 * HGS_ROW_F4];      /* [wave][row][record][3] (+ pad) */  \
    __shared__ __attribute__((aligned(16))) uint32_t s_pre[HGS_NFC * HGS_FWD_PRE_STRIDE];                    \
    const bool overflow = status->overflow != 0;                                                             \
    if (blockIdx.x >= cell_blocks) {                                                                         \
      const uint32_t g = blockIdx.x - cell_blocks;                                                           \
      if (g < (uint32_t)v.TT) render_fwd_background(v, L, overflow, (int)g, out_color, out_depth, out_alpha); \
      return;                                                                                                \
    }                                                                                                        \
    if (overflow) return;                                                                                    \
    if (blockIdx.x == 0 && threadIdx.x == 0) {       /* the sort has finished: the pair total is final (hgs_status.num_pairs) */ \
      const uint32_t np = (uint32_t)L.ctr->alloc_ps;                                                         \
      status->num_pairs = np;                                                                                \
      if (status_host) *reinterpret_cast<volatile uint32_t*>(&status_host->num_pairs) = np;                  \
    }                                                                                                        \
    HGS_TL_BEGIN();                                                                                          \
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);                                     \
    /* the work tables are per die (Counters::sched); the forward takes the dies' tables of a class one behind the other */ \
    const size_t dcap = hgs_die_cells(v.TT);                                                                 \
    if (threadIdx.x < HGS_NFC) {                                                                             \
      uint32_t acc = 0;                                                                                      \
      for (int x = 0; x < HGS_NXCD; ++x) {                                                                   \
        s_pre[threadIdx.x * HGS_FWD_PRE_STRIDE + x] = acc;                                                   \
        acc += (uint32_t)L.ctr->sched[x][2 + threadIdx.x];                                                   \
      }                                                                                                      \
      s_pre[threadIdx.x * HGS_FWD_PRE_STRIDE + HGS_NXCD] = acc;                                              \
    }                                                                                                        \
    __syncthreads();                                                                                         \
    uint32_t cnt4[HGS_FWD_C4 + 1], n4 = 0, nrest = 0;     /* long cells per class; the other cells */        \
    _Pragma("unroll") for (int c = 0; c < HGS_NFC; ++c) {                                                    \
      const uint32_t nc = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_pre[c * HGS_FWD_PRE_STRIDE + HGS_NXCD]); \
      if (c < HGS_FWD_C4) { cnt4[c] = nc; n4 += nc; } else nrest += nc;                                      \
    }                                                                                                        \
    const uint32_t nitems = n4 + (nrest + 3u) / 4u;                                                          \
    const uint32_t ncu = max(1u, cell_blocks / 4u), S = ncu * 4u;                                            \
    const uint32_t slot = (blockIdx.x % ncu) * 4u + (uint32_t)w;                                             \
    HGS_TLI_DECL();                                                                                          \
    for (uint32_t rho = blockIdx.x / ncu; rho * S < nitems; rho += (cell_blocks + ncu - 1u) / ncu) {         \
      const uint32_t it = rho * S + ((rho & 1u) ? S - 1u - slot : slot);                                     \
      if (it >= nitems) continue;                                                                            \
      HGS_TLI_BEGIN();                                                                                       \
      if (it < n4) {                                                                                         \
        uint32_t q = it;                                                                                     \
        int c4 = 0;                                                                                          \
        _Pragma("unroll") for (int c = 0; c < HGS_FWD_C4 - 1; ++c) if (c4 == c && q >= cnt4[c]) { q -= cnt4[c]; c4 = c + 1; } \
        render_fwd_cell4<STORE>(v, L, fwd_cell_key(L, dcap, s_pre, c4, q), recs, cstate, out_color,         \
                                out_depth, out_alpha, s_rec[w]);                                             \
      } else {                                                                                               \
        render_fwd_cells<STORE>(v, L, s_pre, it - n4, w, recs, cstate, out_color, out_depth, out_alpha, s_rec[w]); \
      }                                                                                                      \
      HGS_TLI_END(it, (it < n4 ? 0 : -1));                                                                   \
    }                                                                                                        \
    HGS_TL_END(4, HGS_TLI_TAG());                                                                            \
  }
HGS_RENDER_FWD_KERNEL(hgs_k_render_fwd_store, true)
HGS_RENDER_FWD_KERNEL(hgs_k_render_fwd_nostore, false)
