// hgs_common.h - shared device-side definitions for libhgs_rast (gfx950 / CDNA4 only).
//
// One render call = B views (B >= 1) of the SAME Gaussians at the same resolution, rendered by
// ONE launch set: every kernel runs over the B*P Gaussian instances / B*T tiles of the batch
// ("global tile" g = view * T + tile).  The single-view API is the B = 1 case of the same code.
//
// Data layout in HBM:
//
//   geom buffer   (saved for backward)          bin buffer (sized by entry capacity C, all views)
//   ---------------------------------           --------------------------------------------
//   GeomRec  geom[B*P]           64 B each      uint64  keys[C]     (depth_bits<<32 | idx)
//   uint32   tile_n[B*T]         list length    SortRec recs[C]     48 B, depth-sorted per tile
//   uint32   tile_start[B*T]     first entry    float   bstate[C/64][6][256]  per-bucket pixel state
//   uint32   tile_bstart[B*T]    first bucket state       float segT[2C/SEG][256], segP[2C/SEG][7][256]
//   uint32   tile_wgstart[B*T]   first backward item      uint2 seg_item[], wg_tile[]
//   uint32   tile_msegstart[B*T] first segment plane
//   uint32   tile_maxcontrib[B*T]               img buffer:  uint32 n_contrib[B][H*W]
//   uint32   tile_order[B*T]     heavy first    bwd scratch: float grad_rows[R][12]
//   uint32   hist[B*nwg][T]      per-binning-workgroup tile histograms (T <= 16384)
//   uint32   tile_gbase[RG][B*T] absolute base of a row group inside the tile's list
//   uint32   chunk_sums/base[B*P/256]  entry-id ranges of the 256-Gaussian chunks
//   Counters ctr                 bump allocators, class histogram, ticket
//   hgs_status
//
// A tile's list, its bucket states, its backward work items and its segment planes are RANGES
// handed out by bump allocation (one 64-bit atomic per 64 tiles), not by a prefix scan over the
// tiles: where a range lives does not influence any result, so nothing is lost, and the
// single-workgroup scan chain of the first design (21 us of latency for 32 KB of data) is gone.
// The same holds for a Gaussian's entry-id range (`GeomRec::offset`).
//
// wave = 64 lanes everywhere; a "bucket" is 64 consecutive entries of one tile's list.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hgs_rast.h"

#define HGS_TILE 16
#define HGS_TILE_PIX 256
#define HGS_BLOCK 256          // Gaussians per preprocess / fill chunk
#define HGS_BUCKET 64          // entries per backward bucket (= one wave)
#ifndef HGS_BWD_WAVES
#define HGS_BWD_WAVES 1        // waves per backward bucket (each sweeps 4 / HGS_BWD_WAVES quadrants);
                               // measured 1 / 2 / 4: 96 / 98 / 100 us (100k Gaussians), 165 / 207 / 231 us (500k)
#endif
#define HGS_ROW_GROUPS 16      // histogram row groups walked in parallel by hgs_k_tiles (one wave each)
#define HGS_TILES_PER_WG 64    // tiles per hgs_k_tiles workgroup
#ifndef HGS_SEG
#define HGS_SEG 256            // entries per forward segment (list-parallel blend), multiple of 64
#endif
#ifndef HGS_SEG_THRESH
#define HGS_SEG_THRESH 1024    // only tile lists longer than this are cut into segments: short
                               // lists blend faster in one piece (measured)
#endif
#define HGS_SEG_PLANES 7       // per-segment pixel planes: C0 C1 C2 D W Tend(signed) last(bits)
#define HGS_NEAR_Z 0.2f
#define HGS_ALPHA_MIN (1.0f / 255.0f)
#define HGS_ALPHA_MAX 0.99f
#define HGS_T_EPS 0.0001f
#define HGS_BSTATE_FLOATS (6 * HGS_TILE_PIX)   // T, C0, C1, C2, D, W per pixel
#define HGS_ROW_FLOATS 12                       // grad row per entry (10 used)
#define HGS_NCLS 33                             // tile classes by log2(list length); class 0 = empty

struct __attribute__((aligned(16))) GeomRec {   // 64 B, one per (view, Gaussian)
  float mx, my;         // pixel-space mean
  float ca, cb, cc;     // conic (inverse 2D covariance)
  float op;             // opacity
  float r, g, b;        // view-dependent colour (after +0.5, clamp)
  float depth;          // view-space z
  uint32_t rect_lo;     // minx | miny << 16   (tile units)
  uint32_t rect_hi;     // maxx | maxy << 16   (exclusive)
  uint32_t offset;      // first entry id of this Gaussian's contiguous range, RELATIVE to its 256-chunk's
                        // base (Layout::chunk_base, bump-allocated by hgs_k_tiles)
  int32_t radius;       // 0 => culled
  uint32_t clamped;     // bit c set: colour channel c was clamped at 0
  uint32_t flags;       // bit0: t.x/t.z frustum-clamped, bit1: t.y/t.z clamped
};

struct __attribute__((aligned(16))) SortRec {   // 48 B, one per (tile, Gaussian) entry
  float mx, my;         // pixel-space mean
  float qa, qb, qc;     // conic folded for exp2: qa=-0.5*ca*log2e, qb=-cb*log2e, qc=-0.5*cc*log2e
  float op, r, g, b, depth;
  uint32_t entry;       // entry id = geom.offset + position of the tile in the rect
  uint32_t idx_mask;    // Gaussian index within the view (low 28 bits) | quadrant cull mask << 28
};
#define HGS_LOG2E 1.4426950408889634f

// Device-side counters of one forward call (zeroed by the first workgroup of the preprocess kernel).
struct Counters {
  unsigned long long alloc_eb;   // low: entries handed out to tiles (= R when done), high: bucket states
  unsigned long long alloc_ws;   // low: backward work items,                         high: segment planes
  uint32_t entry_alloc;          // entry ids handed out to Gaussians (= R when done)
  uint32_t ticket;               // hgs_k_tiles workgroups that have finished
  uint32_t max_n;                // longest tile list
  uint32_t pad;
  uint32_t bwd_cur[4];           // backward work items placed so far, per cost class (0 = most expensive): classes 0 / 1
                                 // fill the first item table from its front / back, classes 2 / 3 the second one
  uint32_t cls_hist[HGS_NCLS];   // tiles per class
  uint32_t cls_cur[HGS_NCLS];    // tile_order cursor per class (starts at the class base, heavy first)
};

// Backward work items are dispatched in table order, and the kernel ends with its last item: the
// forward sorts them into four cost classes (kept (entry, quadrant) pairs of the bucket) so that
// expensive buckets start first and the cheapest ones fill the tail.  Classes 0 and 1 share one
// table (front / back), classes 2 and 3 a second one: no class needs to know another's size while
// the forward is still placing items.  Position r of class c among `total` items:
__host__ __device__ __forceinline__ size_t hgs_bwd_item_slot(uint32_t cls, uint32_t r, uint32_t total, uint32_t capacity) {
  const size_t table = (cls >> 1) ? (size_t)capacity + capacity / HGS_BUCKET + 2 : 0;
  return table + ((cls & 1u) ? (size_t)(total - 1u - r) : (size_t)r);
}

struct Layout {          // pointers carved out of the caller's buffers
  GeomRec* geom;
  uint32_t* tile_n;
  uint32_t* tile_start;
  uint32_t* tile_bstart;
  uint32_t* tile_wgstart;
  uint32_t* tile_msegstart;   // index of a tile's first segment plane (tiles with more than one segment)
  uint32_t* tile_maxcontrib;
  uint32_t* tile_order;
  uint32_t* hist;             // [B*nwg][T] per-workgroup tile histograms -> exclusive bases inside a row group
  uint32_t* tile_gbase;       // [HGS_ROW_GROUPS][B*T] absolute base of each row group in the tile's list
  uint32_t* tile_count;       // [B*T] global-atomic path only (T > 16384): counts, then fill cursor
  uint32_t* chunk_sums;       // [B*nblk] tiles_touched summed over a 256-Gaussian chunk
  uint32_t* chunk_base;       // [B*nblk] first entry id of the chunk (bump-allocated)
  Counters* ctr;
  uint2* seg_item;            // [<= 2C/HGS_SEG + 4] (tile, segment) of every segment of the long lists
  uint4* wg_tile;             // two tables of [C + C/64 + 2]: (tile, bucket, list start, list length) of every backward
                              // work item, by cost class (hgs_bwd_item_slot)
                              // (written by the forward: the backward wave finds its records with ONE load)
  unsigned long long* keys;
  SortRec* recs;
  float* bstate;
  float* segT;             // [2C/SEG][256]   product of (1-alpha) over a (non-last) segment
  float* segP;             // [2C/SEG][7][256] per-segment partial sums -> exclusive prefix (base)
  uint32_t* n_contrib;     // [B][H*W]
};

struct Cam {             // per-view constants (device pointers stay with the caller)
  const float* viewmatrix;
  const float* projmatrix;
  const float* campos;
  const float* bg;
  float tanfovx, tanfovy, focal_x, focal_y;
};

struct View {            // per-call constants, passed by value to every kernel
  Cam cam[HGS_MAX_VIEWS];
  float scale_modifier;
  int32_t W, H, grid_x, grid_y, T;   // T = tiles per view
  int32_t B, TT;                     // views, B*T
  int32_t P, M, D, nblk;             // Gaussians per view, SH coefficients, active degree, 256-chunks per view
  int32_t cpw, nwg, lds_bins;        // chunks per binning workgroup, binning workgroups PER VIEW, LDS path?
  uint32_t entry_capacity;
  int32_t max_tile_hint;             // >0: caller promises no tile list is longer (else overflow bit 2)
  int32_t seg_off;                   // 1: the caller's hint proves no list exceeds HGS_SEG_THRESH
  int32_t seg_recompute;             // 1: the hint proves lists have <= 12 segments: no segT pre-pass
  int32_t act;                       // HGS_ACT_* bits: inputs are RAW parameters, activations fused into preprocess
};

static inline size_t hgs_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// number of forward segments of a tile list of n entries (a pure function of n, so results do
// not depend on hints or call history)
__host__ __device__ __forceinline__ uint32_t hgs_nseg(uint32_t n) {
  return n > HGS_SEG_THRESH ? (n + HGS_SEG - 1) / HGS_SEG : 1u;
}

// Forward pixel ownership: four waves per tile, wave w owns the 8x8 quadrant (w&1, w>>1),
// lane l is (l&7, l>>3) inside it.  `pf` in [0,256) (= forward thread index) is the index
// under which the forward stores per-pixel bucket / segment state; the backward and the
// combine kernel map it back to a pixel with the same function.
__device__ __forceinline__ void hgs_fwd_thread_pixel(int pf, int& lx, int& ly) {
  const int w = pf >> 6, l = pf & 63;
  lx = ((w & 1) << 3) | (l & 7);
  ly = ((w >> 1) << 3) | (l >> 3);
}
#define HGS_FWD_THREADS 256

// The one place alpha is evaluated, shared by forward and backward so both take the
// identical instruction sequence (skip decisions must agree).  q* are the folded conic of
// SortRec; p2 = log2(e) * power.  Returns false when the pair is skipped (power > 0 or
// alpha < 1/255).  m2 = qa*dx + qb*dy and m3 = qc*dy are handed back for the backward.
__device__ __forceinline__ bool hgs_eval_alpha(float dx, float dy, float qa, float qb,
                                               float qc, float op, float& G, float& alpha,
                                               float& m2, float& m3) {
  m2 = __builtin_fmaf(qa, dx, qb * dy);
  m3 = qc * dy;
  const float p2 = __builtin_fmaf(dx, m2, m3 * dy);
  G = __builtin_amdgcn_exp2f(p2);
  alpha = fminf(HGS_ALPHA_MAX, op * G);
  return (p2 <= 0.0f) && (alpha >= HGS_ALPHA_MIN);
}

// ---- wave-level primitives on DPP (no LDS round trips) ---------------------------------------
// v_mov_b32_dpp with row_shr / row_bcast / wave_shr patterns; `update_dpp(old, src, ctrl, row_mask,
// bank_mask, bound_ctrl)`: lanes whose source is out of range keep `old`.
#define HGS_DPP_ROW_SHR(n) (0x110 + (n))
#define HGS_DPP_ROW_BCAST15 0x142
#define HGS_DPP_ROW_BCAST31 0x143
#define HGS_DPP_QUAD_PERM(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define HGS_DPP_ROW_MIRROR 0x140
#define HGS_DPP_ROW_HALF_MIRROR 0x141
#define HGS_DPP_ROW_ROR(n) (0x120 + (n))

// inclusive prefix sum over the 64 lanes: 4 row_shr steps inside each row of 16, then the two
// broadcast steps that carry row totals across rows (the classic GCN DPP scan)
__device__ __forceinline__ uint32_t hgs_wave_incl_scan(uint32_t v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, HGS_DPP_ROW_SHR(1), 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, HGS_DPP_ROW_SHR(2), 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, HGS_DPP_ROW_SHR(4), 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, HGS_DPP_ROW_SHR(8), 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, HGS_DPP_ROW_BCAST15, 0xa, 0xf, false);   // rows 1 and 3 += last lane of rows 0 / 2
  x += __builtin_amdgcn_update_dpp(0, x, HGS_DPP_ROW_BCAST31, 0xc, 0xf, false);   // rows 2, 3 += lane 31
  return (uint32_t)x;
}

// max over the 64 lanes, returned in every lane (values are unsigned)
__device__ __forceinline__ uint32_t hgs_wave_max_u32(uint32_t v) {
  int x = (int)v;
  x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, HGS_DPP_ROW_SHR(1), 0xf, 0xf, false));
  x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, HGS_DPP_ROW_SHR(2), 0xf, 0xf, false));
  x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, HGS_DPP_ROW_SHR(4), 0xf, 0xf, false));
  x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, HGS_DPP_ROW_SHR(8), 0xf, 0xf, false));
  x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, HGS_DPP_ROW_BCAST15, 0xa, 0xf, false));
  x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, HGS_DPP_ROW_BCAST31, 0xc, 0xf, false));
  return (uint32_t)__builtin_amdgcn_readlane(x, 63);       // lane 63 holds the max of all lanes
}

// Exclusive scan over the workgroup (NT threads, multiple of 64).  `wtot` needs NT/64
// entries of LDS.  Returns the exclusive prefix; `total` = workgroup sum.
template <int NT>
__device__ __forceinline__ uint32_t hgs_block_excl_scan(uint32_t v, uint32_t* wtot,
                                                        uint32_t& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t incl = hgs_wave_incl_scan(v);
  __syncthreads();                 // protect wtot from a previous use
  if (lane == 63) wtot[w] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < NT / 64; ++k) {
    const uint32_t t = wtot[k];
    if (k < w) base += t;
    tot += t;
  }
  total = tot;
  return base + incl - v;
}

// ---- device timeline (debug builds only: -DHGS_TIMELINE, tools/timeline.py) --------------------------
// Every wave of an instrumented kernel leaves (start, end, hardware id, tag) in a static device
// table (defined in api.hip: the kernels of that translation unit); wall_clock64 ticks at 100 MHz.
// Not compiled into the product library.
#ifdef HGS_TIMELINE
#define HGS_TL_KERNELS 6
#define HGS_TL_SLOTS (1 << 17)
#define HGS_TL_BEGIN() const unsigned long long tl_t0__ = wall_clock64()
#define HGS_TL_END(KID, TAG)                                                                        \
  do {                                                                                              \
    const unsigned long long tl_t1__ = wall_clock64();                                              \
    const uint32_t tl_slot__ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                 \
    if ((threadIdx.x & 63) == 0 && tl_slot__ < HGS_TL_SLOTS) {                                      \
      const uint32_t hw__ = __builtin_amdgcn_s_getreg((31 << 11) | 4);                              \
      const uint32_t xcc__ = __builtin_amdgcn_s_getreg((31 << 11) | 20);                            \
      hgs_tl[KID][tl_slot__][0] = tl_t0__;                                                          \
      hgs_tl[KID][tl_slot__][1] = tl_t1__;                                                          \
      hgs_tl[KID][tl_slot__][2] = ((unsigned long long)xcc__ << 32) | hw__;                         \
      hgs_tl[KID][tl_slot__][3] = (unsigned long long)(TAG);                                        \
    }                                                                                               \
  } while (0)
#else
#define HGS_TL_BEGIN() do {} while (0)
#define HGS_TL_END(KID, TAG) do {} while (0)
#endif
