// render_fwd.hip - front-to-back alpha blending (stage F6), CELL-ROW mapping.
//
// Upstream runs one thread block per 16x16 tile, every thread walks the tile's whole depth-sorted list and
// evaluates every Gaussian at its pixel; on an avatar a Gaussian reaches alpha >= 1/255 on ~34 of those 256
// pixels.  Here the tile is cut into 16 cells of 4x4 pixels and the sort kernel has written, per tile, 16
// depth-ordered CELL LISTS holding only the records that can reach each cell (cellmask.h, exact).  One
// workgroup = one tile = four independent wave64 (no workgroup barrier); wave w owns the 8x8 quadrant
// (w&1, w>>1) and is FOUR ROWS of 16 lanes: row j = one cell, lane = one pixel, and every row streams
// ITS OWN cell list:
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
// entry: 4 B index + 48 B gather per cell reached (L2-served), 24 B/pixel out, 24 B/pixel per 64 cell-list
// entries of state.
#include "hgs_common.h"

namespace {

struct PixState {
  float T, C0, C1, C2, D, Wt;
  uint32_t last;
  bool done;
};

// Blend one staged record into the lane's pixel, fully predicated.  r2.w carries the record's 1-based
// position in the TILE's list (upstream's `contributor` count); pad records have opacity 0 and never blend.
__device__ __forceinline__ void blend_one(PixState& s, float pxf, float pyf, const float4 r0,
                                          const float4 r1, const float4 r2) {
  float G, alpha, m2, m3;
  const bool keep = hgs_eval_alpha(r0.x - pxf, r0.y - pyf, r0.z, r0.w, r1.x, r1.y, G, alpha, m2, m3);
  const bool live = keep && !s.done;
  const float test_T = s.T * (1.0f - alpha);
  const bool stop = live && (test_T < HGS_T_EPS);
  const bool upd = live && !stop;
  s.done = s.done || stop;
  const float wgt = upd ? alpha * s.T : 0.0f;
  s.C0 = __builtin_fmaf(r1.z, wgt, s.C0);
  s.C1 = __builtin_fmaf(r1.w, wgt, s.C1);
  s.C2 = __builtin_fmaf(r2.x, wgt, s.C2);
  s.D = __builtin_fmaf(r2.y, wgt, s.D);
  s.Wt += wgt;
  s.T = upd ? test_T : s.T;
  s.last = upd ? __float_as_uint(r2.w) : s.last;
}

}  // namespace

#ifndef HGS_FWD_GROUP
#define HGS_FWD_GROUP 2          // records of a batch whose LDS reads are issued together (one group ahead of the blend)
#endif

template <bool STORE>
__device__ __forceinline__ void render_fwd_body(const View& v, const Layout& L,
                                                const hgs_status* __restrict__ status,
                                                const SortRec* __restrict__ recs_all,
                                                float* __restrict__ cstate,
                                                float* __restrict__ out_color,
                                                float* __restrict__ out_depth,
                                                float* __restrict__ out_alpha) {
  __shared__ float4 s_rec[HGS_FWD_THREADS / 64][4 * HGS_ROW_F4];      // [wave][row][record][3] (+ pad): 3 KB per wave
  const bool overflow = status->overflow != 0;
  const uint32_t p = blockIdx.x;
  if (p >= (uint32_t)v.TT) return;
  const int g = overflow ? (int)p : (int)L.tile_order[p];   // lists are invalid on overflow: background only
  const int bview = g / v.T, t = g % v.T;
  const size_t HW = (size_t)v.H * v.W;
  out_color += (size_t)bview * 3 * HW;
  out_depth += (size_t)bview * HW;
  out_alpha += (size_t)bview * HW;
  uint32_t* __restrict__ n_contrib = L.n_contrib + (size_t)bview * HW;
  const float* __restrict__ bg = v.cam[bview].bg;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int j = lane >> 4, i = lane & 15;
  const int c = hgs_cell_of(w, j);
  const int px = (t % v.grid_x) * HGS_TILE + (c & 3) * HGS_CELL + (i & 3);
  const int py = (t / v.grid_x) * HGS_TILE + (c >> 2) * HGS_CELL + (i >> 2);
  const bool inside = (px < v.W) && (py < v.H);
  const float pxf = (float)px, pyf = (float)py;

  const uint32_t n_tile = overflow ? 0u : L.tile_n[g];
  const uint32_t tstart1 = n_tile ? L.tile_start[g] - 1u : 0u;        // record index - tstart1 = 1-based list position
  uint32_t len = 0, base = 0, sbase = 0;
  if (n_tile) {                                       // (empty tiles have no cell table)
    const CellInfo ci = L.cell_info[(size_t)g * 16 + c];
    len = ci.len; base = ci.base; sbase = ci.sbase;
  }
  const uint32_t* __restrict__ list = L.cell_list + base;
  const float4* __restrict__ recs = reinterpret_cast<const float4*>(recs_all);
  float4* __restrict__ srow = s_rec[w] + j * HGS_ROW_F4;              // this row's 16 staged records

  PixState s;
  s.T = 1.0f; s.C0 = s.C1 = s.C2 = s.D = s.Wt = 0.f;
  s.last = 0;
  s.done = !inside;

  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // software pipeline: indices two batches ahead, records one batch ahead
  uint32_t idx_next = ((uint32_t)i < len) ? list[i] : 0xffffffffu;                 // batch 0
  float4 c0 = zero4, c1 = zero4, c2 = zero4;
  if (idx_next != 0xffffffffu) {
    c0 = recs[3 * (size_t)idx_next]; c1 = recs[3 * (size_t)idx_next + 1]; c2 = recs[3 * (size_t)idx_next + 2];
    c2.w = __uint_as_float(idx_next - tstart1);
  }
  idx_next = (HGS_RB + (uint32_t)i < len) ? list[HGS_RB + i] : 0xffffffffu;        // batch 1

  for (uint32_t it0 = 0;; it0 += HGS_RB) {
    // rows still at work: list not exhausted and a pixel not finished
    const unsigned long long act = __ballot((it0 < len) && !s.done);
    if (act == 0ull) break;
    const bool row_on = ((act >> (lane & 48)) & 0xffffull) != 0ull;
    if (STORE && row_on && it0 > 0 && (it0 % HGS_SEGLEN) == 0) {
      float* cs = cstate + (size_t)(sbase + it0 / HGS_SEGLEN - 1) * HGS_CSTATE_FLOATS + i;
      cs[0 * 16] = s.T; cs[1 * 16] = s.C0; cs[2 * 16] = s.C1; cs[3 * 16] = s.C2; cs[4 * 16] = s.D; cs[5 * 16] = s.Wt;
    }
    __builtin_amdgcn_wave_barrier();                 // the previous batch's LDS reads are done
    srow[3 * i + 0] = c0; srow[3 * i + 1] = c1; srow[3 * i + 2] = c2;
    // next batch's records (its indices arrived during the previous batch), then the indices after that
    c0 = zero4; c1 = zero4; c2 = zero4;
    if (row_on && idx_next != 0xffffffffu) {
      c0 = recs[3 * (size_t)idx_next]; c1 = recs[3 * (size_t)idx_next + 1]; c2 = recs[3 * (size_t)idx_next + 2];
      c2.w = __uint_as_float(idx_next - tstart1);
    }
    const uint32_t in2 = it0 + 2 * HGS_RB + (uint32_t)i;
    idx_next = (row_on && in2 < len) ? list[in2] : 0xffffffffu;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // groups of HGS_FWD_GROUP records; the LDS reads of group k + 1 are in flight while group k is blended
    float4 ra[HGS_FWD_GROUP], rb[HGS_FWD_GROUP], rc[HGS_FWD_GROUP];
#pragma unroll
    for (int u = 0; u < HGS_FWD_GROUP; ++u) { ra[u] = srow[3 * u + 0]; rb[u] = srow[3 * u + 1]; rc[u] = srow[3 * u + 2]; }
#pragma unroll
    for (int u0 = 0; u0 < HGS_RB; u0 += HGS_FWD_GROUP) {
      float4 na[HGS_FWD_GROUP], nb[HGS_FWD_GROUP], nc[HGS_FWD_GROUP];
      if (u0 + HGS_FWD_GROUP < HGS_RB) {
#pragma unroll
        for (int u = 0; u < HGS_FWD_GROUP; ++u) {
          na[u] = srow[3 * (u0 + HGS_FWD_GROUP + u) + 0]; nb[u] = srow[3 * (u0 + HGS_FWD_GROUP + u) + 1];
          nc[u] = srow[3 * (u0 + HGS_FWD_GROUP + u) + 2];
        }
      }
#pragma unroll
      for (int u = 0; u < HGS_FWD_GROUP; ++u) blend_one(s, pxf, pyf, ra[u], rb[u], rc[u]);
      if (u0 + HGS_FWD_GROUP < HGS_RB) {
#pragma unroll
        for (int u = 0; u < HGS_FWD_GROUP; ++u) { ra[u] = na[u]; rb[u] = nb[u]; rc[u] = nc[u]; }
      }
    }
  }

  if (inside) {
    const size_t pix = (size_t)py * v.W + px;
    out_color[0 * HW + pix] = s.C0 + s.T * bg[0];
    out_color[1 * HW + pix] = s.C1 + s.T * bg[1];
    out_color[2 * HW + pix] = s.C2 + s.T * bg[2];
    out_depth[pix] = s.D;
    out_alpha[pix] = s.Wt;
    n_contrib[pix] = s.last;
  }
}

#define HGS_RENDER_FWD_KERNEL(NAME, STORE)                                                                 \
  extern "C" __global__ void __launch_bounds__(HGS_FWD_THREADS) NAME(                                       \
      View v, Layout L, const hgs_status* __restrict__ status, const SortRec* __restrict__ recs,             \
      float* __restrict__ cstate, float* __restrict__ out_color, float* __restrict__ out_depth,              \
      float* __restrict__ out_alpha) {                                                                       \
    HGS_TL_BEGIN();                                                                                          \
    render_fwd_body<STORE>(v, L, status, recs, cstate, out_color, out_depth, out_alpha);                     \
    HGS_TL_END(4, blockIdx.x < (uint32_t)v.TT ? L.tile_n[L.tile_order[blockIdx.x]] : 0u);                    \
  }
HGS_RENDER_FWD_KERNEL(hgs_k_render_fwd_store, true)
HGS_RENDER_FWD_KERNEL(hgs_k_render_fwd_nostore, false)
