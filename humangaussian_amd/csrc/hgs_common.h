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
//   GeomRec  geom[B*P]           64 B each      uint64  keys[C]        (depth_bits<<32 | idx)
//   uint32   tile_n[B*T]         list length    SortRec recs[C]        48 B, depth-sorted per tile
//   uint32   tile_start[B*T]     first entry    uint2   cell_list[16C] (record index, pair-row id), CELL-major per tile
//   uint32   tile_order[B*T]     heavy first (+ uint4 tile_rec[B*T]: tile, entries, first entry)
//                                                   pair-row ids: ENTRY-major (1-2 views) or chunk-cell-major (hgs_rec_tag)
//   CellInfo cell_info[B*T][16]  the 16 cell    float   cstate[16C/SEGLEN][6][16]  pixel state every HGS_SEGLEN (128) cell-list entries
//            lists of a tile                    uint4   items_full[16C/SEGLEN] backward work items (full segments)
//   uint4    items_part[2][16 B*T] backward work items (last, partial segment of every cell list)
//   uint32   fwd_cells[11][16 B*T] forward work items: non-empty cells by length class
//   uint32   hist[B*nwg][T]      per-binning-workgroup tile histograms (T <= 16384)
//   uint32   tile_gbase[RG][B*T] absolute base of a row group inside the tile's list
//   uint32   chunk_sums/base[B*P/256]  entry-id ranges of the 256-Gaussian chunks
//   Counters ctr                 bump allocators, class histogram, tickets
//   hgs_status                                  img buffer:  uint32 n_contrib[B][H*W]
//                                               bwd scratch: float grad_rows[R][12], pair_rows[num_pairs (<= 16R)][10]
//
// A 16x16 tile is cut into 16 CELLS of 4x4 pixels.  The sort kernel gives every entry a 16-bit cell
// mask (cellmask.h: the exact ellipse-vs-rectangle test of alpha >= 1/255) and writes, per tile, 16
// depth-ordered CELL LISTS (indices of the records that can touch the cell).  Both blend kernels walk
// cell lists: a wave64 is four ROWS of 16 lanes, row = one cell (4x4 pixels), and every row streams
// ITS OWN list - a Gaussian costs lane time only in the cells it reaches (57 lane slots per entry
// instead of 97 with 8x8 quadrants, 256 without culling, tools/cell_stats.py).
//
// A tile's list and its pair range are RANGES handed out by bump allocation, not by a prefix scan
// over the tiles: where a range lives does not influence any result.
//
// wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hgs_rast.h"
#include "cellmask.h"

#define HGS_TILE 16
#define HGS_TILE_PIX 256
#define HGS_BLOCK 256          // Gaussians per preprocess / fill chunk
#define HGS_ROW_GROUPS 16      // histogram row groups walked in parallel by hgs_k_tiles (one wave each)
#define HGS_TILES_PER_WG 64    // tiles per hgs_k_tiles workgroup
#define HGS_RB 16              // records a row stages per batch (= lanes of a row)
#define HGS_ROW_F4 (HGS_RB * 3 + 1)   // float4 per staged row: 16 records of 48 B + 16 B, so that the four rows of a wave
                               // (which read four DIFFERENT records per ds_read_b128) sit on different LDS banks
#define HGS_BWD_BLOCK_WAVES 12 // waves per workgroup of the blend backward: they share one LDS ticket for the workgroup's groups
#define HGS_SEGLEN 128         // cell-list entries per backward work item; the forward stores the pixel state
                               // of a cell at every multiple of this
#define HGS_PAIRS_PER_ENTRY 16 // capacity of the pair arrays per entry of capacity (worst case: every cell)
#define HGS_NEAR_Z 0.2f
#define HGS_ALPHA_MIN (1.0f / 255.0f)
#define HGS_ALPHA_MAX 0.99f
#define HGS_T_EPS 0.0001f
#define HGS_CSTATE_FLOATS (6 * 16)             // T, C0, C1, C2, D, W for the 16 pixels of a cell
#define HGS_ROW_FLOATS 12                       // gradient row per entry (10 used); 64 B rows (whole ECC granules per
                                                // scattered store) measured slower: the rows are bandwidth
#define HGS_GROW_F4 (HGS_ROW_FLOATS / 4)          // float4 per gradient row
#define HGS_PROW_FLOATS 10                      // (entry, cell) pair row: the ten sums, packed (40 B: a sixth less pair traffic than 48 B)
#define HGS_PROW_F2 (HGS_PROW_FLOATS / 2)         // float2 per pair row
#define HGS_SORT_LDS_MAX 4096                   // longest tile list of the 256-thread LDS sort class; longer: hgs_k_sort_large (1024 threads; 3072 / 2048 / 1024 swept: EXPERIMENTS.md)
#define HGS_NCLS 33                             // tile classes by log2(list length); class 0 = empty
#define HGS_NXCD 8                              // accelerator dies (XCDs), each with its own L2: workgroup b of a launch runs on die b % 8
#define HGS_NFC 11                              // length classes of the non-empty cells (forward work items)

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
  uint32_t entry;       // entry id = chunk base + geom.offset + position of the tile in the rect (< 2^27: HGS_MAX_ENTRY_CAPACITY)
  uint32_t pad;
};
#define HGS_LOG2E 1.4426950408889634f
// SortRec::pad of the record at list position k of a tile of n entries, written by the sort in calls that keep their pair rows
// CHUNK-cell-major (View::pairchunks: calls of >= HGS_CHUNK_ROWS_MIN_VIEWS views); both blend kernels overwrite the word with the
// list position when they gather a record, the pair reduction hgs_k_pair_reduce_ch reads it: the entry's 16-bit cell mask, its
// place in the 64-record CHUNK of the tile list it belongs to (chunks start at the tile's first record) and the chunk's record
// count, plus the layout of the chunk's pair rows - bit 28 set: chunk-cell-major (the rows of the chunk are one block starting
// at entpair.y, cell by cell, inside a cell in list order), clear: entry-major (entpair.y = the entry's first row; the long-list
// sort classes).  tests/test_pair_rows_cpu.py restates the id arithmetic of both kernels.
__host__ __device__ __forceinline__ uint32_t hgs_rec_tag(uint32_t mask, uint32_t k, uint32_t n, bool chunk_rows) {
  const uint32_t left = n - (k & ~63u);
  return (mask & 0xffffu) | ((k & 63u) << 16) | (((left < 64u ? left : 64u) - 1u) << 22) | (chunk_rows ? 1u << 28 : 0u);
}

struct __attribute__((aligned(16))) CellInfo {  // one of the 16 cell lists of a tile
  uint32_t base;        // first slot of the list in cell_list (absolute)
  uint32_t len;         // records in the list
  uint32_t sbase;       // first pixel-state slot (cstate): slot sbase + s - 1 holds the state before entry HGS_SEGLEN * s
  uint32_t pbase;       // the TILE's first pair id
};

// Device-side counters of one forward call (zeroed by the first workgroup of the preprocess kernel).
struct __attribute__((aligned(128))) CounterLine { uint32_t v; };
struct __attribute__((aligned(128))) Counters {   // One 128 B line per counter (group) that one kernel's atomics hit: atomics
  // on DIFFERENT words of a line queue behind each other like same-address ones (with the entry allocator of `fill` on
  // the line of the sort's allocators, `fill` ran 74 instead of 66 us with 8 views; with the tile allocator, the list
  // maximum and the class histogram of `tiles` on one line, `tiles` 25 us with 8 views, 14.5 with the three on a line
  // each (2048 class-histogram atomics queueing at ~7 ns).  Hence also: a line per tile class.
  unsigned long long alloc_eb;   // low: entries handed out to tiles (= R when done)
  __attribute__((aligned(128))) uint32_t entry_alloc;   // entry ids handed out to Gaussians (= R when done)
  __attribute__((aligned(128))) uint32_t max_n;         // longest tile list
  CounterLine cls_hist[HGS_NCLS];   // tiles per class (a line each: a view's tiles fall into three or four classes)
  CounterLine cls_cur[HGS_NCLS];    // tile_order cursor per class (starts at the class base, heavy first)
  __attribute__((aligned(128))) unsigned long long alloc_ps;   // bump allocator of the sort kernel: low: pairs (= cell-list slots) handed out to tiles, high: cell states
  // Work tables are kept PER DIE: the sort puts the work of the tile at tile_order position b into the tables of die
  // b % 8.  [x][0] low / high: backward work items of class 0 (full segments) / class 1; [x][1] low / high: class 2 /
  // class 3; [x][2 + c]: non-empty cells of length class c (forward items).  The sort issues ONE multi-lane atomic per
  // tile over alloc_ps and its die's row: same-address device-scope atomics serialise at ~10 ns each, eight rows cut
  // that queue by eight (global forward counters next to per-die backward ones: 39.3 -> 42 us for the sort of a view).
  //  - the backward's workgroup i draws from the tables of die i % 8 - the die the dispatcher puts it on - so the
  //    waves that gather one tile's records and lists share ONE L2 (a record crossed the fabric once per die that
  //    touched it: 205 -> 131 MB per view, 47 -> 44.5 us).  Placement only: any workgroup may process any table;
  //  - the forward takes the eight tables of a class one behind the other, as ONE list (with a die's workgroups on
  //    its own table its traffic fell as well, 77 -> 52 MB, but it ran 3 us longer per view, 10 us at 500k - it is
  //    bound by its most loaded SIMD, not by the gathers; EXPERIMENTS.md).
  __attribute__((aligned(128))) unsigned long long sched[HGS_NXCD][16];      // (rows padded to a line: 2 + HGS_NFC used)
};
static_assert(2 + HGS_NFC <= 16, "a die's row of work counters is one 128 B line");

// Backward work item = HGS_SEGLEN consecutive entries of one cell list (the last item of a list may be
// shorter).  Four items make a wave (one per row), so items are handed out longest first and in classes of
// similar length: class 0 = full segments; 1 / 2 / 3 = partial ones with >= 43 / >= 22 / fewer entries.
__host__ __device__ __forceinline__ uint32_t hgs_item_class(uint32_t cnt) {
  return cnt >= HGS_SEGLEN ? 0u : (cnt >= (2u * HGS_SEGLEN + 2u) / 3u ? 1u : (cnt >= (HGS_SEGLEN + 2u) / 3u ? 2u : 3u));
}

// Forward work item = one non-empty cell; a wave takes four cells of one class (rows of similar length end
// together), classes in descending length: batches of 16 records >= 33, 25, 17, 13, 9, 7, 5, then 4, 3, 2, 1.
__host__ __device__ __forceinline__ uint32_t hgs_cell_class(uint32_t len) {
  const uint32_t nb = (len + HGS_RB - 1) / HGS_RB;
  return nb >= 33u ? 0u : nb >= 25u ? 1u : nb >= 17u ? 2u : nb >= 13u ? 3u : nb >= 9u ? 4u : nb >= 7u ? 5u : nb >= 5u ? 6u : 11u - nb;
}

struct Layout {          // pointers carved out of the caller's buffers
  GeomRec* geom;
  uint32_t* tile_n;
  uint32_t* tile_start;
  uint32_t* tile_order;
  uint4* tile_rec;            // [B*T] by position in tile_order: (tile, entries, first entry, 0)
  uint32_t* hist;             // [B*nwg][T] per-workgroup tile histograms -> exclusive bases inside a row group
  uint32_t* tile_gbase;       // [HGS_ROW_GROUPS][B*T] absolute base of each row group in the tile's list
  uint32_t* tile_count;       // [B*T] global-atomic path only (T > 16384): counts, then fill cursor
  uint32_t* chunk_sums;       // [B*nblk] tiles_touched summed over a 256-Gaussian chunk
  uint32_t* chunk_base;       // [B*nblk] first entry id of the chunk (bump-allocated)
  CellInfo* cell_info;        // [B*T][16]
  uint32_t* fwd_cells;        // [HGS_NXCD][HGS_NFC][hgs_die_cells]: cell keys (g * 16 + c) of the non-empty cells, per die and length class
  uint4* items_part;          // [HGS_NXCD][2][hgs_die_cells]: per die, table 0 holds class 1 (from the front) and class 2 (from the back), table 1 class 3
  Counters* ctr;
  unsigned long long* keys;
  SortRec* recs;
  uint2* cell_list;           // [16 C]: (record index, id of the pair's gradient row), cell-major per tile
  uint2* entpair;             // [C] by record index: (entry id | pairs << 27, first pair row of the entry - or of its chunk, hgs_rec_tag) - what the pair reduction reads
  float* cstate;              // [C/4 + 1][6][16]
  uint32_t full_cap;          // slots of ONE die's items_full table (the full segments of all lists would fit in each)
  uint4* items_full;          // [HGS_NXCD][full_cap]: (cell key = g * 16 + c, entries, first cell-list slot, state slot or ~0): all a wave needs to start
  uint32_t* n_contrib;        // [B][H*W]  1-based TILE-list position of the pixel's last contributor (upstream's meaning)
};

// Cells of the tiles one die can get: tile_order positions b with b % HGS_NXCD == x, at most ceil(B*T / 8) of them.
__host__ __device__ __forceinline__ size_t hgs_die_cells(int TT) { return (size_t)16 * (size_t)((TT + HGS_NXCD - 1) / HGS_NXCD); }

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
  int32_t act;                       // HGS_ACT_* bits: inputs are RAW parameters, activations fused into preprocess
  int32_t pairchunks;                // != 0 (calls of >= 3 views): the backward's pair rows are cell-major inside every 64-record chunk of a
                                     // tile list (hgs_rec_tag): a batch of the blend backward writes runs of rows; 0: entry-major ids
};

static inline size_t hgs_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Pixel ownership in both blend kernels: wave w of a tile's workgroup owns the 8x8 quadrant (w&1, w>>1); its row j
// (lanes 16 j .. 16 j + 15) owns the 4x4 cell (2 (w&1) + (j&1), 2 (w>>1) + (j>>1)); lane i of the row is pixel
// (i & 3, i >> 2) of the cell.  Cell index c = cy * 4 + cx (the bit order of hgs_cell_mask).
__device__ __forceinline__ int hgs_cell_of(int w, int j) { return (((w >> 1) << 1) + (j >> 1)) * 4 + ((w & 1) << 1) + (j & 1); }
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
  const float og = op * G;
  alpha = fminf(HGS_ALPHA_MAX, og);
  // (the threshold test on the UN-clamped value: min(0.99, x) >= 1/255 <=> x >= 1/255 for every non-NaN x, and the
  //  backward - which needs op G, not alpha - saves the clamp in front of its test; render_bwd.hip restates this
  //  function on float2 with the same operations)
  return (p2 <= 0.0f) && (og >= HGS_ALPHA_MIN);
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
// per work item of a persistent wave: the longest one (10 ns ticks << 40 | items << 28 | item index << 4 | class + 1)
#define HGS_TLI_DECL() unsigned long long tli_max__ = 0, tli_t__ = 0; uint32_t tli_it__ = 0, tli_n__ = 0
#define HGS_TLI_BEGIN() tli_t__ = wall_clock64()
#define HGS_TLI_END(IT, CLS)                                                                        \
  do {                                                                                              \
    const unsigned long long d__ = wall_clock64() - tli_t__;                                        \
    ++tli_n__;                                                                                      \
    if (d__ > tli_max__) { tli_max__ = d__; tli_it__ = ((uint32_t)(IT) << 4) | (uint32_t)((CLS) + 1); } \
  } while (0)
#define HGS_TLI_TAG() ((tli_max__ << 40) | ((unsigned long long)tli_n__ << 28) | (tli_it__ & 0xfffffffu))
#else
#define HGS_TL_BEGIN() do {} while (0)
#define HGS_TL_END(KID, TAG) do {} while (0)
#define HGS_TLI_DECL() do {} while (0)
#define HGS_TLI_BEGIN() do {} while (0)
#define HGS_TLI_END(IT, CLS) do {} while (0)
#define HGS_TLI_TAG() 1u
#endif
