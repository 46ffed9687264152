// hgs_common.h - shared device-side definitions for libhgs_rast (gfx950 / CDNA4 only).
//
// Data layout in HBM (one render call = one view):
//
//   geom buffer   (saved for backward)        bin buffer (sized by entry_capacity C)
//   ---------------------------------         --------------------------------------------
//   GeomRec  geom[P]            64 B each     uint64  keys[C]     (depth_bits<<32 | idx)
//   uint32   block_sums[NBLK]                 SortRec recs[C]     48 B, depth-sorted per tile
//   uint32   block_base[NBLK]                 float   bstate[C/64][6][256]  per-bucket pixel state
//   uint32   tile_count[T]    (fill cursor)
//   uint32   tile_start[T+1]                  img buffer:  uint32 n_contrib[H*W]
//   uint32   tile_order[T]    (heavy first)
//   uint32   tile_bstart[T+1] (bucket-state prefix)        bwd scratch: float grad_rows[R][12]
//   uint32   tile_wgstart[T+1](backward WG prefix)
//   uint32   tile_maxcontrib[T]                float segT[2C/SEG][256], segP[2C/SEG][7][256]
//   uint32   tile_msegstart[T+1]                      (forward segments of long lists)
//   uint32   hist[NWG][T]      (per-binning-workgroup tile histograms, T <= 16384)
//
// wave = 64 lanes everywhere; a "bucket" is 64 consecutive entries of one tile's list.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hgs_rast.h"

#define HGS_TILE 16
#define HGS_TILE_PIX 256
#define HGS_BLOCK 256          // Gaussians per preprocess / fill workgroup
#define HGS_BUCKET 64          // entries per backward bucket (= one wave)
#ifndef HGS_BWD_WAVES
#define HGS_BWD_WAVES 1        // waves per backward bucket (each sweeps 4 / HGS_BWD_WAVES quadrants);
                               // measured 1 / 2 / 4: 96 / 98 / 100 us (100k Gaussians), 165 / 207 / 231 us (500k)
#endif
#define HGS_ROW_GROUPS 4       // histogram row groups scanned in parallel by hgs_k_colscan
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

struct __attribute__((aligned(16))) GeomRec {   // 64 B, one per Gaussian
  float mx, my;         // pixel-space mean
  float ca, cb, cc;     // conic (inverse 2D covariance)
  float op;             // opacity
  float r, g, b;        // view-dependent colour (after +0.5, clamp)
  float depth;          // view-space z
  uint32_t rect_lo;     // minx | miny << 16   (tile units)
  uint32_t rect_hi;     // maxx | maxy << 16   (exclusive)
  uint32_t offset;      // exclusive prefix of tiles_touched = first entry id
  int32_t radius;       // 0 => culled
  uint32_t clamped;     // bit c set: colour channel c was clamped at 0
  uint32_t flags;       // bit0: t.x/t.z frustum-clamped, bit1: t.y/t.z clamped
};

struct __attribute__((aligned(16))) SortRec {   // 48 B, one per (tile, Gaussian) entry
  float mx, my;         // pixel-space mean
  float qa, qb, qc;     // conic folded for exp2: qa=-0.5*ca*log2e, qb=-cb*log2e, qc=-0.5*cc*log2e
  float op, r, g, b, depth;
  uint32_t entry;       // entry id = geom.offset + position of the tile in the rect
  uint32_t idx_mask;    // Gaussian index (low 28 bits) | quadrant cull mask << 28
};
#define HGS_LOG2E 1.4426950408889634f

struct Layout {          // pointers carved out of the caller's buffers
  GeomRec* geom;
  uint32_t* block_sums;
  uint32_t* block_base;
  uint32_t* tile_count;
  uint32_t* tile_start;
  uint32_t* tile_order;
  uint32_t* tile_bstart;
  uint32_t* tile_wgstart;
  uint32_t* tile_maxcontrib;
  uint32_t* tile_msegstart;   // [T+1] prefix of (nseg > 1 ? nseg : 0): index of a tile's segment planes
  uint2* seg_item;            // [<= 2C/HGS_SEG + 4] (tile, segment) of every segment of the long lists
  uint2* wg_tile;             // [<= C + C/64] (tile, bucket) of every backward work item (written by the forward)
  uint32_t* tile_pos;         // [T] position of a tile in tile_order
  uint32_t* pos_wgstart;      // [T] first backward work item of the tile at a tile_order position
  uint32_t* hist;          // [nwg][T] per-workgroup tile histograms -> exclusive bases
  uint32_t* tile_grp;      // [HGS_ROW_GROUPS][T] row-group totals (colscan)
  uint32_t* tile_gbase;    // [HGS_ROW_GROUPS][T] absolute base of each row group in the tile's list (scan)
  unsigned long long* keys;
  SortRec* recs;
  float* bstate;
  float* segT;             // [2C/SEG][256]   product of (1-alpha) over a (non-last) segment
  float* segP;             // [2C/SEG][7][256] per-segment partial sums -> exclusive prefix (base)
  uint32_t* n_contrib;
};

struct View {            // per-call constants, passed by value to every kernel
  const float* viewmatrix;
  const float* projmatrix;
  const float* campos;
  const float* bg;
  float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
  int32_t W, H, grid_x, grid_y, T;
  int32_t P, M, D, nblk;
  int32_t cpw, nwg, lds_bins;   // chunks per binning workgroup, #binning workgroups, LDS path?
  uint32_t entry_capacity;
  int32_t max_tile_hint;        // >0: caller promises no tile list is longer (else overflow bit 2)
  int32_t seg_off;              // 1: the caller's hint proves no list exceeds HGS_SEG_THRESH
  int32_t seg_recompute;        // 1: the hint proves lists have <= 12 segments: no segT pre-pass
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
// (A 2-pixels-per-lane / 2-waves-per-tile variant was measured: 168 vs 87 us - the cull is
// weaker and the extra ILP does not materialise; removed.)
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

__device__ __forceinline__ uint32_t hgs_wave_incl_scan(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
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
