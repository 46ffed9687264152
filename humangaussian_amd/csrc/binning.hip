// binning.hip - tile binning and per-tile depth sort (stages F2-F5 of SURVEY.md 2.3(B)).
//
// Upstream: InclusiveSum over Gaussians -> duplicateWithKeys -> one GLOBAL 64-bit radix
// sort of all (tile|depth) keys -> identifyTileRanges.  Here (MI355X-first), for all B views of
// a call at once (global tile g = view * T + tile):
//   1. hgs_k_tiles     many workgroups, 64 tiles each: sums the per-workgroup histogram rows
//                      hgs_k_preprocess_fwd left (16 row groups walked by 16 waves in parallel,
//                      rows turned into exclusive bases in place), and hands every tile its list
//                      range by BUMP ALLOCATION (one 64-bit atomic per 64 tiles) - no prefix scan
//                      over the tiles, no single-workgroup latency chain, no ticket.
//   2. hgs_k_fill      per Gaussian: scatter (depth_bits<<32 | idx) keys into its tiles' list
//                      ranges (order inside a tile is arbitrary here); extra workgroups of the same
//                      launch place the tiles into tile_order (heavy classes first), hand out the
//                      entry-id bases of the Gaussian chunks and publish hgs_status.
//   3. hgs_k_sort_*    per tile: bitonic sort of the tile's keys IN LDS (unique keys =>
//                      deterministic result = upstream's stable order: depth, ties by index),
//                      then gathers the Gaussians into a depth-ordered, contiguous 48-byte
//                      record list ("duplicated Gaussian list"), gives every entry its 16-bit CELL mask
//                      (which 4x4-pixel cells of the tile it can reach, cellmask.h) and writes the tile's
//                      16 depth-ordered cell lists + the backward work items (gather_records).
//
// Roofline: HBM/latency-bound integer work: 8 B/entry written + read for keys, one 64 B
// gather + 48 B write per entry for the records.
#include "hgs_common.h"

// ---------------------------------------------------------------------------- 1. tiles
// Workgroup = 64 consecutive tiles of ONE view x HGS_ROW_GROUPS row groups (one wave each).
// LDS path: thread (tile, rg) walks rows [rg*rpg, (rg+1)*rpg) of the view's histogram column in
// place (hist[row][t] -> entries of tile t owned by earlier workgroups of the same group).
// Global-atomic path (T > 16384): the counts are already in tile_count.
extern "C" __global__ void __launch_bounds__(64 * HGS_ROW_GROUPS)
hgs_k_tiles(View v, Layout L) {
  __shared__ uint32_t gt[HGS_ROW_GROUPS][HGS_TILES_PER_WG];     // group totals per tile
  __shared__ uint32_t start_s[HGS_TILES_PER_WG];
  __shared__ uint32_t cls_s[HGS_NCLS];
  const int tid = threadIdx.x, tl = tid & 63;
  const int rg = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bpv = (v.T + HGS_TILES_PER_WG - 1) / HGS_TILES_PER_WG;      // workgroups per view
  const int b = (int)blockIdx.x / bpv;
  const int t = ((int)blockIdx.x % bpv) * HGS_TILES_PER_WG + tl;
  const bool valid = t < v.T;
  const size_t g = (size_t)b * v.T + t;
  if (tid < HGS_NCLS) cls_s[tid] = 0;
  uint32_t run = 0;
  if (v.lds_bins) {
    const int rpg = (v.nwg + HGS_ROW_GROUPS - 1) / HGS_ROW_GROUPS;
    const int r0 = rg * rpg, r1 = min(v.nwg, r0 + rpg);
    if (valid) {
      uint32_t* col = L.hist + (size_t)b * v.nwg * v.T + t;
      constexpr int BR = 16;
      for (int r = r0; r < r1; r += BR) {
        uint32_t c[BR];
#pragma unroll
        for (int k = 0; k < BR; ++k) c[k] = (r + k < r1) ? col[(size_t)(r + k) * v.T] : 0u;
#pragma unroll
        for (int k = 0; k < BR; ++k) {
          // (a row that has no entry in this tile keeps its 0: hgs_k_fill never uses the base of a tile its chunk does not
          //  touch, and seven of eight histogram words are such zeros - the stores were half of this kernel's traffic)
          if (r + k < r1 && c[k]) col[(size_t)(r + k) * v.T] = run;
          run += c[k];
        }
      }
    }
  } else if (rg == 0 && valid) {
    run = L.tile_count[g];
    L.tile_count[g] = 0;                 // becomes the fill cursor
  }
  gt[rg][tl] = run;
  __syncthreads();
  uint32_t gbase = 0;                    // entries of this tile in earlier row groups
#pragma unroll
  for (int k = 0; k < HGS_ROW_GROUPS; ++k) gbase += (k < rg) ? gt[k][tl] : 0u;
  if (rg == 0) {
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < HGS_ROW_GROUPS; ++k) n += gt[k][tl];
    if (!valid) n = 0;
    const uint32_t i0 = hgs_wave_incl_scan(n);
    // one bump allocation for the 64 tiles (lane 63 holds the total)
    unsigned long long b_eb = 0;
    if (tl == 63 && i0) b_eb = atomicAdd(&L.ctr->alloc_eb, (unsigned long long)i0);
    const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b_eb, 63);
    const uint32_t start = s0 + i0 - n;
    start_s[tl] = start;
    if (valid) {
      L.tile_n[g] = n;
      L.tile_start[g] = start;
    }
    // class histogram (class = bit length of n; 0 = empty); the empty class is counted per wave
    const unsigned long long eb = __ballot(valid && n == 0);
    if (tl == 0 && eb) atomicAdd(&cls_s[0], (uint32_t)__popcll(eb));
    if (valid && n) atomicAdd(&cls_s[32 - __clz(n)], 1u);
    const uint32_t mx = hgs_wave_max_u32(n);
    if (tl == 0 && mx) atomicMax(&L.ctr->max_n, mx);
  }
  __syncthreads();
  if (v.lds_bins && valid) L.tile_gbase[(size_t)rg * v.TT + g] = start_s[tl] + gbase;
  if (tid < HGS_NCLS && cls_s[tid]) atomicAdd(&L.ctr->cls_hist[tid].v, cls_s[tid]);

}

// ---------------------------------------------------------------------------- 2. fill
// LDS path: same workgroup -> chunk ownership as hgs_k_preprocess_fwd.  Slot of an entry =
// tile_gbase[rg][g] (list start + entries of earlier row groups) + hist[row][t] (entries of
// earlier workgroups of the group) + an LDS cursor.  Workgroups beyond the binning ones place
// tiles into tile_order: position = class base (heavy first) + a cursor; the order INSIDE a
// class is free (it only schedules work).
// overflow word of this call, recomputed from the counters hgs_k_tiles left (the status itself is
// published by the first tile-order workgroup of the SAME launch, so fill cannot read it yet)
__device__ __forceinline__ uint32_t overflow_from_counters(const View& v, const Layout& L) {
  const uint32_t R = (uint32_t)L.ctr->alloc_eb;
  uint32_t o = (R > v.entry_capacity) ? 1u : 0u;
  if (v.max_tile_hint > 0 && L.ctr->max_n > (uint32_t)v.max_tile_hint) o |= 2u;
  return o;
}

// The workgroups behind the binning ones ("order" workgroups, 256 tiles each):
//  * place their tiles into tile_order (class base, heavy classes first, + a cursor per class);
//  * bump-allocate the entry-id bases of their share of the 256-Gaussian chunks;
//  * the first of them publishes hgs_status (device copy + pinned host mirror).
// All of this used to sit at the end of the tile kernel behind a ticket; here it costs no extra
// latency chain (it runs beside the key scatter).
__device__ __forceinline__ void place_tiles(const View& v, const Layout& L, int blk, int nblks,
                                            hgs_status* __restrict__ status,
                                            hgs_status* __restrict__ status_host) {
  __shared__ uint32_t cls_base[HGS_NCLS];
  __shared__ uint32_t cls_cnt[HGS_NCLS];
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid < HGS_NCLS) cls_cnt[tid] = L.ctr->cls_hist[tid].v;
  __syncthreads();
  if (tid == 0) {
    uint32_t acc = 0;
    for (int c = HGS_NCLS - 1; c >= 0; --c) { cls_base[c] = acc; acc += cls_cnt[c]; }   // heavy classes first
  }
  if (blk == 0 && tid == 64) {
    const unsigned long long a_eb = L.ctr->alloc_eb;
    hgs_status st;
    st.num_rendered = (uint32_t)a_eb;
    st.active_tiles = (uint32_t)v.TT - cls_cnt[0];
    st.num_pairs = 0;                    // (known when the sort has run: the blend forward publishes it)
    st.bwd_groups = 0;                   // (ABI v10 field of the bucket design: unused since v11)
    st.overflow = overflow_from_counters(v, L);
    st.reserved[0] = v.entry_capacity;   // carve key for hgs_backward
    st.reserved[1] = L.ctr->max_n;       // longest tile list
    st.reserved[2] = 1;                  // "complete" marker (see the host mirror below)
    *status = st;
    if (status_host) {            // pinned, device-mapped host memory: no in-stream copy.  reserved[2] = 1 is written LAST,
      volatile uint32_t* hs = reinterpret_cast<volatile uint32_t*>(status_host);      // behind a system-scope fence: a host
      const uint32_t* sw = reinterpret_cast<const uint32_t*>(&st);                    // that cleared it can POLL the word
      for (int k = 0; k < 7; ++k) hs[k] = sw[k];                                      // instead of waiting for an event
      __threadfence_system();
      hs[7] = 1u;
      __threadfence_system();
    }
  }
  if (tid >= 128 && tid < 192) {
    // entry-id bases of this workgroup's share of the B*nblk chunks: one bump allocation per 64 chunks
    const int nchunk = v.B * v.nblk;
    const int per_wg = (nchunk + nblks - 1) / nblks;
    const int c_begin = blk * per_wg, c_end = min(nchunk, c_begin + per_wg);
    for (int c0 = c_begin; c0 < c_end; c0 += 64) {
      const int c = c0 + lane;
      const uint32_t sum = (c < c_end) ? L.chunk_sums[c] : 0u;
      const uint32_t inc = hgs_wave_incl_scan(sum);
      uint32_t base = 0;
      if (lane == 63 && inc) base = atomicAdd(&L.ctr->entry_alloc, inc);
      base = (uint32_t)__builtin_amdgcn_readlane((int)base, 63);
      if (c < c_end) L.chunk_base[c] = base + inc - sum;
    }
  }
  __syncthreads();
  const int g = blk * HGS_BLOCK + tid;
  const bool valid = g < v.TT;
  const uint32_t n = valid ? L.tile_n[g] : 0u;
  // position inside the class: ONE atomic per (wave, class) - the lanes of a class are counted with ballots (a view's
  // tiles fall into three or four classes; an atomic per tile queued 5870 of them at ~10 ns with 8 views)
  const int c = (valid && n) ? 32 - __clz(n) : 0;
  int leader_of = 0;                     // first lane of this lane's class
  uint32_t rank = 0, cnt = 0;            // this lane's rank inside its class, lanes of the class (in the wave)
  unsigned long long todo = __ballot(valid);
  while (todo) {
    const int leader = (int)__builtin_ctzll(todo);
    const int cl = __builtin_amdgcn_readlane(c, leader);
    const unsigned long long same = __ballot(valid && c == cl);
    if (valid && c == cl) {
      leader_of = leader;
      rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0u));
      cnt = (uint32_t)__popcll(same);
    }
    todo &= ~same;
  }
  uint32_t base = 0;                     // (the leaders' atomics leave together: one round trip per wave)
  if (valid && lane == leader_of) base = atomicAdd(&L.ctr->cls_cur[c].v, cnt);
  base = (uint32_t)__shfl((int)base, leader_of, 64);
  const uint32_t pos = cls_base[c] + base + rank;
  if (!valid) return;
  L.tile_order[pos] = (uint32_t)g;
  L.tile_rec[pos] = make_uint4((uint32_t)g, n, L.tile_start[g], 0u);      // what a sort workgroup needs to start, in one load
}

extern "C" __global__ void __launch_bounds__(HGS_BLOCK)
hgs_k_fill(View v, Layout L, hgs_status* __restrict__ status, hgs_status* __restrict__ status_host, int nbin) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_cur[];
  if ((int)blockIdx.x >= nbin) {
    place_tiles(v, L, (int)blockIdx.x - nbin, (int)gridDim.x - nbin, status, status_host);
    return;
  }
  if (overflow_from_counters(v, L)) return;
  const int b = (int)blockIdx.x / v.nwg, lw = (int)blockIdx.x % v.nwg;
  const uint32_t* __restrict__ base_row = L.hist + (size_t)blockIdx.x * v.T;
  const int rpg = (v.nwg + HGS_ROW_GROUPS - 1) / HGS_ROW_GROUPS;
  const uint32_t* __restrict__ grp_row = L.tile_gbase + (size_t)(lw / rpg) * v.TT + (size_t)b * v.T;
  for (int t = threadIdx.x; t < v.T; t += HGS_BLOCK) lds_cur[t] = grp_row[t] + base_row[t];
  __syncthreads();
  const GeomRec* __restrict__ geom = L.geom + (size_t)b * v.P;
  for (int c = 0; c < v.cpw; ++c) {
    const int chunk = lw * v.cpw + c;
    if (chunk >= v.nblk) break;
    const int i = chunk * HGS_BLOCK + threadIdx.x;
    if (i >= v.P) continue;
    const uint4 q2 = reinterpret_cast<const uint4*>(&geom[i])[2];   // b, depth, rect_lo, rect_hi
    const uint32_t depth_bits = q2.y, lo = q2.z, hi = q2.w;
    const int minx = lo & 0xffffu, miny = lo >> 16, maxx = hi & 0xffffu, maxy = hi >> 16;
    const unsigned long long key_hi = (unsigned long long)depth_bits << 32;
    for (int ty = miny; ty < maxy; ++ty)
      for (int tx = minx; tx < maxx; ++tx) {
        const uint32_t slot = atomicAdd(&lds_cur[ty * v.grid_x + tx], 1u);
        L.keys[slot] = key_hi | (uint32_t)i;
      }
  }
}

// Fallback (global atomics) for T*4 > 64 KB.
extern "C" __global__ void __launch_bounds__(HGS_BLOCK)
hgs_k_fill_ga(View v, Layout L, hgs_status* __restrict__ status, hgs_status* __restrict__ status_host, int nbin) {
  if ((int)blockIdx.x >= nbin) {
    place_tiles(v, L, (int)blockIdx.x - nbin, (int)gridDim.x - nbin, status, status_host);
    return;
  }
  if (overflow_from_counters(v, L)) return;
  const int b = (int)blockIdx.x / v.nblk, chunk = (int)blockIdx.x % v.nblk;
  const int i = chunk * HGS_BLOCK + threadIdx.x;
  if (i >= v.P) return;
  const uint4 q2 = reinterpret_cast<const uint4*>(&L.geom[(size_t)b * v.P + i])[2];   // b, depth, rect_lo, rect_hi
  const uint32_t depth_bits = q2.y, lo = q2.z, hi = q2.w;
  const int minx = lo & 0xffffu, miny = lo >> 16, maxx = hi & 0xffffu, maxy = hi >> 16;
  const unsigned long long key_hi = (unsigned long long)depth_bits << 32;
  for (int ty = miny; ty < maxy; ++ty)
    for (int tx = minx; tx < maxx; ++tx) {
      const size_t g = (size_t)b * v.T + ty * v.grid_x + tx;
      const uint32_t slot = L.tile_start[g] + atomicAdd(&L.tile_count[g], 1u);
      L.keys[slot] = key_hi | (uint32_t)i;
    }
}

// ---------------------------------------------------------------------------- 3. sort
namespace {

// One (entry, cell) pair of record `rec`: its cell-list element = (record index, id of the pair's gradient row).  Where the
// blend backward writes the row is a per-call choice (View::pairchunks):
//  * entry-major ids (1-2 views): the rows of an entry are neighbours, in cell order; the reduction streams the rows of 64
//    entries as one contiguous block (17 us per view) - the backward pays with isolated 40 B stores (+3 us);
//  * chunk-cell-major ids (>= 3 views): the rows of every 64-record chunk of the tile list are one block, cell by cell, inside
//    a cell in list order (hgs_rec_tag): a 16-record batch of the backward writes one or two contiguous runs of rows - with 8
//    views in flight the isolated stores made it bandwidth-bound (428 -> 273 us) - and the reduction still reads blocks.
__device__ __forceinline__ void hgs_put_pair(const Layout& L, uint32_t rec, uint32_t row_id, uint32_t slot) {
  L.cell_list[slot] = make_uint2(rec, row_id);
}

// Ranges of one tile, from the lengths of its 16 cell lists: pairs (= cell-list slots), cell states, work items.
// Wave 0 of the workgroup, lane c = cell c.  Two halves: `issue` sends the bump allocation (ONE atomic instruction,
// lanes 0..2 + one per forward class on 64-bit counters: a device-scope atomic is a ~2 us trip to the memory side of
// the fabric, five of them in a row were 13 us of every tile's chain), `finish` consumes its result and writes
// cell_info, the work items, cell_base[] and pair_base (LDS) - the caller may put independent work between the two.
struct CellAlloc {
  uint32_t len, nfull, rem, nst, i_len, i_st, i_full, pcls, fcls, frank, die;
  unsigned long long b1, b2, b3, got;
};

__device__ __forceinline__ void hgs_alloc_cell_ranges_issue(const Layout& L, uint32_t order_pos, uint32_t len_of_lane, CellAlloc& a) {
  const int lane = (int)threadIdx.x & 63;
  a.die = order_pos % HGS_NXCD;                         // the die whose work tables take this tile (Counters::sched)
  const bool cl = lane < 16;
  a.len = cl ? len_of_lane : 0u;
  a.nfull = a.len / HGS_SEGLEN; a.rem = a.len % HGS_SEGLEN;
  const uint32_t nseg = a.nfull + (a.rem ? 1u : 0u);
  a.nst = nseg ? nseg - 1u : 0u;
  a.i_len = hgs_wave_incl_scan(a.len); a.i_st = hgs_wave_incl_scan(a.nst); a.i_full = hgs_wave_incl_scan(a.nfull);
  a.pcls = a.rem ? hgs_item_class(a.rem) : 0u;            // 1..3 for a partial last segment
  a.b1 = __ballot(a.pcls == 1u); a.b2 = __ballot(a.pcls == 2u); a.b3 = __ballot(a.pcls == 3u);
  const uint32_t t_len = (uint32_t)__builtin_amdgcn_readlane((int)a.i_len, 63);
  const uint32_t t_st = (uint32_t)__builtin_amdgcn_readlane((int)a.i_st, 63);
  const uint32_t t_full = (uint32_t)__builtin_amdgcn_readlane((int)a.i_full, 63);
  // forward work items: every non-empty cell goes into its die's table of its length class; lane 3 + c allocates for class c
  a.fcls = hgs_cell_class(a.len);
  unsigned long long fb_ = 0;                          // cells of this tile in "my" class (lanes 3 .. 3 + HGS_NFC - 1)
  a.frank = 0;                                         // rank of this lane's cell inside its class, within the tile
#pragma unroll
  for (int c = 0; c < HGS_NFC; ++c) {
    const unsigned long long bc = __ballot(cl && a.len && a.fcls == (uint32_t)c);
    if (lane == 3 + c) fb_ = bc;
    if (a.fcls == (uint32_t)c) a.frank = (uint32_t)__popcll(bc & ((1ull << lane) - 1ull));
  }
  const unsigned long long add = lane == 0 ? ((unsigned long long)t_len | ((unsigned long long)t_st << 32))
                               : lane == 1 ? ((unsigned long long)t_full | ((unsigned long long)__popcll(a.b1) << 32))
                               : lane == 2 ? ((unsigned long long)__popcll(a.b2) | ((unsigned long long)__popcll(a.b3) << 32))
                                           : (unsigned long long)__popcll(fb_);
  a.got = 0;
  unsigned long long* ctr = lane == 0 ? &L.ctr->alloc_ps : &L.ctr->sched[a.die][min(lane, 2 + HGS_NFC) - 1];
  if (lane < 3 + HGS_NFC && add) a.got = atomicAdd(ctr, add);
}

__device__ __forceinline__ void hgs_alloc_cell_ranges_finish(const View& v, const Layout& L, int g, const CellAlloc& a,
                                                             uint32_t* cell_base, uint32_t& pair_base_out) {
  int lane = (int)threadIdx.x & 63;
  asm volatile("" : "+v"(lane));       // (keeps lane-dependent store addresses from being hoisted out of a persistent tile loop
  const bool cl = lane < 16;           //  and held - spilled - across it)
  const uint32_t fpos = (uint32_t)__shfl((int)(uint32_t)a.got, 3 + (int)a.fcls, 64) + a.frank;
  const uint32_t got_lo = (uint32_t)a.got, got_hi = (uint32_t)(a.got >> 32);
  const uint32_t pb = (uint32_t)__builtin_amdgcn_readlane((int)got_lo, 0);
  const uint32_t sb = (uint32_t)__builtin_amdgcn_readlane((int)got_hi, 0);
  const uint32_t fb = (uint32_t)__builtin_amdgcn_readlane((int)got_lo, 1);
  const uint32_t p1 = (uint32_t)__builtin_amdgcn_readlane((int)got_hi, 1);
  const uint32_t p2 = (uint32_t)__builtin_amdgcn_readlane((int)got_lo, 2);
  const uint32_t p3 = (uint32_t)__builtin_amdgcn_readlane((int)got_hi, 2);
  if (cl) {
    const uint32_t len = a.len, nfull = a.nfull, rem = a.rem;
    const uint32_t base = pb + a.i_len - len;
    cell_base[lane] = base;
    CellInfo ci;
    ci.base = base; ci.len = len; ci.sbase = sb + a.i_st - a.nst; ci.pbase = pb;
    L.cell_info[(size_t)g * 16 + lane] = ci;
    const uint32_t key = (uint32_t)g * 16u + (uint32_t)lane;
    const size_t dcap = hgs_die_cells(v.TT);
    if (len) L.fwd_cells[((size_t)a.die * HGS_NFC + a.fcls) * dcap + fpos] = key;
    // a work item carries all its wave needs to start: (cell, entries, first cell-list slot, state slot in front of it)
    uint4* full = L.items_full + ((size_t)a.die * L.full_cap + fb + a.i_full - nfull);
    for (uint32_t sgm = 0; sgm < nfull; ++sgm)
      full[sgm] = make_uint4(key, HGS_SEGLEN, base + sgm * HGS_SEGLEN, sgm ? ci.sbase + sgm - 1u : 0xffffffffu);
    if (rem) {
      const unsigned long long below = (1ull << lane) - 1ull;
      uint4* part = L.items_part + (size_t)a.die * 2 * dcap;
      const uint4 it = make_uint4(key, rem, base + nfull * HGS_SEGLEN, nfull ? ci.sbase + nfull - 1u : 0xffffffffu);
      if (a.pcls == 1u) part[p1 + (uint32_t)__popcll(a.b1 & below)] = it;
      else if (a.pcls == 2u) part[dcap - 1 - (p2 + (uint32_t)__popcll(a.b2 & below))] = it;
      else part[dcap + p3 + (uint32_t)__popcll(a.b3 & below)] = it;
    }
  }
  if (lane == 0) pair_base_out = pb;
}

__device__ __forceinline__ void hgs_alloc_cell_ranges(const View& v, const Layout& L, int g, uint32_t order_pos, const uint32_t* cell_tot,
                                                      uint32_t* cell_base, uint32_t& pair_base_out) {
  if (threadIdx.x < 64) {
    CellAlloc a;
    hgs_alloc_cell_ranges_issue(L, order_pos, (threadIdx.x & 63) < 16 ? cell_tot[threadIdx.x & 63] : 0u, a);
    hgs_alloc_cell_ranges_finish(v, L, g, a, cell_base, pair_base_out);
  }
}

// LDS tables of the record gather (one workgroup = one tile): MAXCH 64-record chunks per pass.
template <int MAXCH>
struct GatherLds {
  uint32_t tab[MAXCH][17];   // per chunk: records that touch cell c (columns 0..15), pairs of the chunk (16)
  uint32_t run[17];          // running totals over the passes of a long list (same columns)
  uint32_t cell_tot[16];     // length of the tile's 16 cell lists
  uint32_t cell_base[16];    // absolute first slot of each cell list
  uint32_t pair_base;        // first pair slot of the tile
};

// After the sort: (1) gather the Gaussians into the depth-ordered 48 B record list, computing every entry's
// 16-bit cell mask (cellmask.h); (2) allocate the tile's pair range, cell-state range and backward work
// items; (3) write the 16 depth-ordered cell lists: (record index, pair id) per element; pair ids are entry-major,
// so the backward's pair rows of one entry lie behind each other for the reduce kernel.  `sorted` (LDS or HBM) holds
// the sorted keys on entry; its slots are reused for the masks.
template <int MAXCH>
__device__ __forceinline__ void gather_records(const View& v, const Layout& L, int g, uint32_t order_pos,
                                               uint32_t start, uint32_t n,
                                               unsigned long long* sorted, int nt, GatherLds<MAXCH>& S) {
  const int t = g % v.T;
  const GeomRec* __restrict__ geom = L.geom + (size_t)(g / v.T) * v.P;
  const uint32_t* __restrict__ cbase = L.chunk_base + (size_t)(g / v.T) * v.nblk;
  const int tx = t % v.grid_x, ty = t / v.grid_x;
  const float x0 = (float)(tx * HGS_TILE), y0 = (float)(ty * HGS_TILE);
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wv = tid >> 6, nwaves = nt >> 6;
  if (tid < 16) S.cell_tot[tid] = 0;
  if (tid < 17) S.run[tid] = 0;
  __syncthreads();
  // ---- sweep 1: records + masks + cell-list lengths.
  // GU records per thread in flight: the 64 B geom gathers are dependent random reads (~1-2 us
  // each); issued one at a time they dominated the sort kernel of the heaviest tile
  constexpr int GU = 4;
  uint32_t mytot = 0;                                  // lane c < 16: records of this wave that touch cell c
  for (uint32_t kb0 = 0; kb0 < n; kb0 += (uint32_t)nt * GU) {      // (wave-uniform trip count: ballots inside)
    const uint32_t kb = kb0 + threadIdx.x;
    uint32_t idxv[GU];
    uint4 q0[GU], q1[GU], q2[GU];
    uint32_t q3[GU];
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const uint32_t k = kb + (uint32_t)u * nt;
      idxv[u] = (k < n) ? (uint32_t)sorted[k] : 0u;
    }
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const uint32_t k = kb + (uint32_t)u * nt;
      if (k < n) {
        const uint4* gp = reinterpret_cast<const uint4*>(&geom[idxv[u]]);
        q0[u] = gp[0]; q1[u] = gp[1]; q2[u] = gp[2]; q3[u] = gp[3].x;        // g3: only the entry-id offset
      }
    }
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const uint32_t k = kb + (uint32_t)u * nt;
      uint32_t mask = 0;
      if (k < n) {
        const uint32_t idx = idxv[u];
        const uint4 g0 = q0[u], g1 = q1[u], g2 = q2[u];
        // g0: mx my ca cb | g1: cc op r g | g2: b depth rect_lo rect_hi | q3: offset
        const int minx = g2.z & 0xffffu, miny = g2.z >> 16, maxx = g2.w & 0xffffu;
        const uint32_t entry = cbase[idx >> 8] + q3[u] + (uint32_t)((ty - miny) * (maxx - minx) + (tx - minx));
        const float mx = __uint_as_float(g0.x), my = __uint_as_float(g0.y);
        const float ca = __uint_as_float(g0.z), cb = __uint_as_float(g0.w), cc = __uint_as_float(g1.x);
        mask = hgs_cell_mask(mx, my, ca, cb, cc, __uint_as_float(g1.y), x0, y0);
        uint4* dst = reinterpret_cast<uint4*>(&L.recs[start + k]);
        const float qa = -0.5f * ca * HGS_LOG2E, qb = -cb * HGS_LOG2E, qc = -0.5f * cc * HGS_LOG2E;
        dst[0] = make_uint4(g0.x, g0.y, __float_as_uint(qa), __float_as_uint(qb));
        dst[1] = make_uint4(__float_as_uint(qc), g1.y, g1.z, g1.w);
        dst[2] = make_uint4(g2.x, g2.y, entry, v.pairchunks ? hgs_rec_tag(mask, k, n, false) : 0u);
        L.entpair[start + k].x = entry | ((uint32_t)__popc(mask) << 27);      // (.y, the first pair id, follows in sweep 2)
        sorted[k] = (unsigned long long)mask;               // the key is consumed: its slot keeps the mask
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const unsigned long long bal = __ballot((mask >> c) & 1u);
        if (lane == c) mytot += (uint32_t)__popcll(bal);
      }
    }
  }
  if (lane < 16 && mytot) atomicAdd(&S.cell_tot[lane], mytot);
  __syncthreads();
  hgs_alloc_cell_ranges(v, L, g, order_pos, S.cell_tot, S.cell_base, S.pair_base);
  __syncthreads();
  // ---- sweep 2: cell lists and pair slots, MAXCH chunks per pass
  const uint32_t pair_base = S.pair_base;
  for (uint32_t sc0 = 0; sc0 < n; sc0 += (uint32_t)MAXCH * 64u) {
    const uint32_t nch = min((uint32_t)MAXCH, (n - sc0 + 63u) / 64u);
    // (a) per chunk: how many of its records touch each cell
    for (uint32_t ch = wv; ch < nch; ch += nwaves) {
      const uint32_t k = sc0 + ch * 64u + lane;
      const uint32_t mask = (k < n) ? (uint32_t)sorted[k] : 0u;
      uint32_t mine = 0;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const unsigned long long bal = __ballot((mask >> c) & 1u);
        if (lane == c) mine = (uint32_t)__popcll(bal);
      }
      uint32_t tot = mine;                               // lanes >= 16 hold 0
      tot += __builtin_amdgcn_update_dpp(0, (int)tot, HGS_DPP_ROW_SHR(1), 0xf, 0xf, false);
      tot += __builtin_amdgcn_update_dpp(0, (int)tot, HGS_DPP_ROW_SHR(2), 0xf, 0xf, false);
      tot += __builtin_amdgcn_update_dpp(0, (int)tot, HGS_DPP_ROW_SHR(4), 0xf, 0xf, false);
      tot += __builtin_amdgcn_update_dpp(0, (int)tot, HGS_DPP_ROW_SHR(8), 0xf, 0xf, false);
      if (lane < 16) S.tab[ch][lane] = mine;
      if (lane == 15) S.tab[ch][16] = tot;               // pairs of the chunk
    }
    __syncthreads();
    // (b) exclusive prefix over the chunks, per column, on top of the running totals of earlier passes
    for (int col = wv; col < 17; col += nwaves) {
      uint32_t carry = S.run[col];
      for (uint32_t c0 = 0; c0 < nch; c0 += 64u) {
        const uint32_t ch = c0 + lane;
        const uint32_t x = ch < nch ? S.tab[ch][col] : 0u;
        const uint32_t inc = hgs_wave_incl_scan(x);
        if (ch < nch) S.tab[ch][col] = carry + inc - x;
        carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
      }
      if (lane == 0) S.run[col] = carry;
    }
    __syncthreads();
    // (c) emit
    for (uint32_t ch = wv; ch < nch; ch += nwaves) {
      const uint32_t k = sc0 + ch * 64u + lane;
      const bool in = k < n;
      const uint32_t mask = in ? (uint32_t)sorted[k] : 0u;
      const uint32_t cnt = (uint32_t)__popc(mask);
      const uint32_t rel = S.tab[ch][16] + hgs_wave_incl_scan(cnt) - cnt;      // entry-major, relative to the tile
      if (in) L.entpair[start + k].y = pair_base + rel;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const bool bit = (mask >> c) & 1u;
        const unsigned long long bal = __ballot(bit);
        if (bit) {
          const uint32_t rank = S.tab[ch][c] + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
          const uint32_t slot = S.cell_base[c] + rank;
          hgs_put_pair(L, start + k, pair_base + rel + (uint32_t)__popc(mask & ((1u << c) - 1u)), slot);
        }
      }
    }
    __syncthreads();
  }
}

// ---- the same for lists of at most MAXCH * 64 entries (every sort class but `huge`), without ballots:
// a record's 16 mask bits are SPREAD into four words of four byte counters (bit 4 w + b -> byte b of word w);
// a DPP wave scan of those four words then yields, for all 16 cells at once, how many records of the 64-record
// chunk touch each cell before this lane (<= 64: a byte holds it).  One table pass, no recount.
__device__ __forceinline__ uint32_t hgs_spread4(uint32_t nib) {          // 4 bits -> 4 bytes of 0 / 1
  return (nib * 0x00204081u) & 0x01010101u;
}
__device__ __forceinline__ uint32_t hgs_bytesum(uint32_t w) { return __builtin_amdgcn_sad_u8(w, 0u, 0u); }

template <int MAXCH>
__device__ __forceinline__ void gather_records_single(const View& v, const Layout& L, int g, uint32_t order_pos,
                                                      uint32_t start, uint32_t n,
                                                      unsigned long long* sorted, int nt, GatherLds<MAXCH>& S) {
  const int t = g % v.T;
  const GeomRec* __restrict__ geom = L.geom + (size_t)(g / v.T) * v.P;
  const uint32_t* __restrict__ cbase = L.chunk_base + (size_t)(g / v.T) * v.nblk;
  const int tx = t % v.grid_x, ty = t / v.grid_x;
  const float x0 = (float)(tx * HGS_TILE), y0 = (float)(ty * HGS_TILE);
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wv = tid >> 6, nwaves = nt >> 6;
  const uint32_t nch = (n + 63u) / 64u;                 // <= MAXCH
#ifdef HGS_TIMELINE
  const unsigned long long tg0 = wall_clock64();
#define HGS_TG(i) const unsigned long long tg##i = wall_clock64()
#else
#define HGS_TG(i)
#endif
  // ---- sweep 1: records + masks; per chunk the packed per-cell counts (tab[ch][0..3]) and its pairs (tab[ch][16])
  constexpr int GU = 2;            // gathers per thread in flight
  for (uint32_t kb0 = 0; kb0 < n; kb0 += (uint32_t)nt * GU) {      // (wave-uniform trip count: wave scans inside)
    const uint32_t kb = kb0 + threadIdx.x;
    uint32_t idxv[GU];
    uint4 q0[GU], q1[GU], q2[GU];
    uint32_t q3[GU];
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const uint32_t k = kb + (uint32_t)u * nt;
      idxv[u] = (k < n) ? (uint32_t)sorted[k] : 0u;
    }
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const uint32_t k = kb + (uint32_t)u * nt;
      if (k < n) {
        const uint4* gp = reinterpret_cast<const uint4*>(&geom[idxv[u]]);
        q0[u] = gp[0]; q1[u] = gp[1]; q2[u] = gp[2]; q3[u] = gp[3].x;        // g3: only the entry-id offset
      }
    }
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const uint32_t k = kb + (uint32_t)u * nt;
      if (kb0 + (uint32_t)u * nt + (uint32_t)(wv * 64) >= n) continue;       // (wave-uniform) chunk beyond the list
      uint32_t mask = 0;
      if (k < n) {
        const uint32_t idx = idxv[u];
        const uint4 g0 = q0[u], g1 = q1[u], g2 = q2[u];
        // g0: mx my ca cb | g1: cc op r g | g2: b depth rect_lo rect_hi | q3: offset
        const int minx = g2.z & 0xffffu, miny = g2.z >> 16, maxx = g2.w & 0xffffu;
        const uint32_t entry = cbase[idx >> 8] + q3[u] + (uint32_t)((ty - miny) * (maxx - minx) + (tx - minx));
        const float mx = __uint_as_float(g0.x), my = __uint_as_float(g0.y);
        const float ca = __uint_as_float(g0.z), cb = __uint_as_float(g0.w), cc = __uint_as_float(g1.x);
        mask = hgs_cell_mask(mx, my, ca, cb, cc, __uint_as_float(g1.y), x0, y0);
        uint4* dst = reinterpret_cast<uint4*>(&L.recs[start + k]);
        const float qa = -0.5f * ca * HGS_LOG2E, qb = -cb * HGS_LOG2E, qc = -0.5f * cc * HGS_LOG2E;
        dst[0] = make_uint4(g0.x, g0.y, __float_as_uint(qa), __float_as_uint(qb));
        dst[1] = make_uint4(__float_as_uint(qc), g1.y, g1.z, g1.w);
        dst[2] = make_uint4(g2.x, g2.y, entry, v.pairchunks ? hgs_rec_tag(mask, k, n, false) : 0u);
        L.entpair[start + k].x = entry | ((uint32_t)__popc(mask) << 27);      // (.y, the first pair id, follows in sweep 2)
        sorted[k] = (unsigned long long)mask;               // the key is consumed: its slot keeps the mask
      }
      const uint32_t ch = k >> 6;
      uint32_t tot[4];
#pragma unroll
      for (int wd = 0; wd < 4; ++wd) tot[wd] = hgs_wave_incl_scan(hgs_spread4((mask >> (4 * wd)) & 0xfu));
      if (lane == 63) {
        S.tab[ch][0] = tot[0]; S.tab[ch][1] = tot[1]; S.tab[ch][2] = tot[2]; S.tab[ch][3] = tot[3];
        S.tab[ch][16] = hgs_bytesum(tot[0]) + hgs_bytesum(tot[1]) + hgs_bytesum(tot[2]) + hgs_bytesum(tot[3]);
      }
    }
  }
  __syncthreads();
  HGS_TG(1);
  // ---- exclusive prefix over the chunks for the 16 cells (unpacked to 32 bits, columns 0..15 rewritten in
  // place AFTER every wave has read its packed words) and for the pairs (column 16); totals -> cell_tot
  {
    uint32_t pk[(MAXCH + 63) / 64][4];
#pragma unroll
    for (int r = 0; r < (MAXCH + 63) / 64; ++r) {
      const uint32_t ch = (uint32_t)r * 64u + lane;
#pragma unroll
      for (int wd = 0; wd < 4; ++wd) pk[r][wd] = (wv < 5 && ch < nch) ? S.tab[ch][wd] : 0u;
    }
    __syncthreads();
    // wave wd < 4 unpacks word wd (cells 4 wd .. 4 wd + 3); wave 4 (or wave 0 when there are fewer) scans the pairs
    for (int col = wv; col < 5; col += nwaves) {
      uint32_t carry[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int r = 0; r < (MAXCH + 63) / 64; ++r) {
        const uint32_t ch = (uint32_t)r * 64u + lane;
        if ((uint32_t)r * 64u >= nch) break;
        if (col < 4) {
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const uint32_t x = (pk[r][col] >> (8 * b)) & 0xffu;
            const uint32_t inc = hgs_wave_incl_scan(x);
            if (ch < nch) S.tab[ch][4 * col + b] = carry[b] + inc - x;
            carry[b] += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
          }
        } else {
          const uint32_t x = ch < nch ? S.tab[ch][16] : 0u;
          const uint32_t inc = hgs_wave_incl_scan(x);
          if (ch < nch) S.tab[ch][16] = carry[0] + inc - x;
          carry[0] += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
      }
      if (col < 4 && lane < 4) S.cell_tot[4 * col + lane] = lane == 0 ? carry[0] : lane == 1 ? carry[1] : lane == 2 ? carry[2] : carry[3];
    }
  }
  __syncthreads();
  HGS_TG(2);
  hgs_alloc_cell_ranges(v, L, g, order_pos, S.cell_tot, S.cell_base, S.pair_base);
  __syncthreads();
  HGS_TG(3);
  // ---- sweep 2: cell lists and pair slots
  const uint32_t pair_base = S.pair_base;
  for (uint32_t ch = wv; ch < nch; ch += nwaves) {
    const uint32_t k = ch * 64u + lane;
    const bool in = k < n;
    uint32_t mask = in ? (uint32_t)sorted[k] : 0u;
    uint32_t ex[4], before = 0;                          // records of this chunk before this lane, per cell (bytes)
#pragma unroll
    for (int wd = 0; wd < 4; ++wd) {
      const uint32_t mine = hgs_spread4((mask >> (4 * wd)) & 0xfu);
      ex[wd] = hgs_wave_incl_scan(mine) - mine;
      before += hgs_bytesum(ex[wd]);
    }
    const uint32_t rel = S.tab[ch][16] + before;          // entry-major, relative to the tile
    if (in) L.entpair[start + k].y = pair_base + rel;
    uint32_t r = 0;
    while (mask) {
      const int c = __builtin_ctz(mask);
      mask &= mask - 1u;
      const uint32_t exw = (c < 8) ? (c < 4 ? ex[0] : ex[1]) : (c < 12 ? ex[2] : ex[3]);
      const uint32_t slot = S.cell_base[c] + S.tab[ch][c] + ((exw >> (8 * (c & 3))) & 0xffu);
      hgs_put_pair(L, start + k, pair_base + rel + r, slot);
      ++r;
    }
  }
#ifdef HGS_TIMELINE
  if (threadIdx.x == 0 && blockIdx.x < HGS_TL_SLOTS) {      // phases of this tile's gather (wave 0), 10 ns ticks, 16 bits each
    const unsigned long long tg4 = wall_clock64();
    auto cl = [](unsigned long long d) { return d > 0xffffull ? 0xffffull : d; };
    hgs_tl[2][blockIdx.x][0] = cl(tg1 - tg0) | (cl(tg2 - tg1) << 16) | (cl(tg3 - tg2) << 32) | (cl(tg4 - tg3) << 48);
  }
#endif
}

// Bitonic network in its "all comparators ascending" form (first stage of every merge
// mirrors the upper half), which sorts any n <= npad correctly with VIRTUAL +inf padding:
// a comparator whose upper index is >= n is a no-op.  Keys are unique, so the result is
// the unique ascending order.  One __syncthreads() per stage.  Used for the HBM fallback.
template <int NT>
__device__ __forceinline__ void bitonic_sort(unsigned long long* keys, uint32_t n) {
  uint32_t npad = 2;
  while (npad < n) npad <<= 1;
  const uint32_t half = npad >> 1;
  for (uint32_t kk = 2; kk <= npad; kk <<= 1) {
    const uint32_t hk = kk >> 1;
    for (uint32_t i = threadIdx.x; i < half; i += NT) {
      const uint32_t blk = i / hk, off = i - blk * hk;
      const uint32_t lo = blk * kk + off, hi = blk * kk + (kk - 1 - off);
      if (hi < n) {
        const unsigned long long a = keys[lo], c = keys[hi];
        if (a > c) { keys[lo] = c; keys[hi] = a; }
      }
    }
    __syncthreads();
    for (uint32_t j = kk >> 2; j > 0; j >>= 1) {
      for (uint32_t i = threadIdx.x; i < half; i += NT) {
        const uint32_t lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        const uint32_t hi = lo | j;
        if (hi < n) {
          const unsigned long long a = keys[lo], c = keys[hi];
          if (a > c) { keys[lo] = c; keys[hi] = a; }
        }
      }
      __syncthreads();
    }
  }
}

typedef unsigned long long u64;

}  // namespace

// ---------------------------------------------------------------------------- 3b. rank sort (lists of 1..4096 entries)
// The bitonic network above costs log2(n) (log2(n) + 1) / 2 DEPENDENT stages (45 for a typical 436-entry tile, ~220 ns
// each) and the record gather behind it was a second chain of dependent round trips: a tile took 22 us, the heaviest
// 33 us, and the kernel was the latency of its heaviest tile.  Here a key never moves: the thread that LOADED a key
// computes the key's final RANK and writes the record straight to its slot.
//   1. keys -> registers (coalesced); the 64 B geometry gathers of the thread's first keys go out NOW and fly under
//      the whole sort (the thread knows its Gaussians from the start: nothing waits for the order);
//   2. depth range of the tile (wave reductions + two LDS atomics), NB ~ 2 n buckets (a power of two <= 2048),
//      bucket = floor((depth - dmin) * NB / range): a MONOTONE map (fp subtraction, multiplication by a positive
//      constant, truncation and the clamp are all non-decreasing), so keys of a smaller bucket sort first;
//   3. LDS histogram (one atomic per key; its return value = arrival position inside the bucket), exclusive scan;
//   4. keys scattered into bucket order (LDS); rank = bucket base + number of SMALLER 64-bit keys inside the bucket
//      (unique keys: depth bits, then Gaussian index - upstream's stable order); buckets hold ~1-3 keys, the probes of
//      all keys of a thread run interleaved;
//   5. record + cell mask at the rank (HBM), mask at the rank (LDS), per-cell totals on the way -> the bump
//      allocation of the tile's ranges is ISSUED, the cell tables are built while it travels, then the cell lists:
//      a branch-free emit (16 uniform iterations, list bases through v_readlane) instead of a per-lane loop.
// Degenerate depth distributions (a bucket of more than HGS_RANK_BUCKET_MAX keys: hundreds of exactly equal depths, one
// far outlier that stretches the range) take the plain bitonic network in LDS for step 4 - same results, the old speed.
// Workgroups are PERSISTENT (a capacity-sized grid spent its last third launching ~3000 workgroups that found no tile).
// Steps 1-4 exist in a register form for lists of up to 2 / 4 keys per thread and in a streaming form for longer ones;
// step 5 is ONE loop for all (it reads (Gaussian, rank) pairs back from LDS), which keeps the kernel's code within the
// instruction cache.
#define HGS_SORT_LIGHT 16             // heaviest tiles that get the lightest tiles as CU neighbours (sort_rank_body; 8 / 24 / 32 measured the same, 0 = off: +3.4 us)
#define HGS_RANK_BUCKET_MAX 192        // (< 256: bucket lengths travel in 8 bits)
#define HGS_RANK_NB_MAX 2048
// 256 threads: three workgroups per CU by LDS (45 KB each) are three waves per SIMD, which leaves a wave 168 VGPRs for
// up to 16 keys + their ranks + the gathers in flight.  A typical tile (~440 entries) is two keys per thread.
#define HGS_SORT_NT 256
#define HGS_SORT_WAVES_PER_EU 3
#define HGS_RANK_GU 2                  // geometry gathers per thread and round (16 registers each); two rounds are in flight

namespace {

#ifdef HGS_TIMELINE
__shared__ unsigned long long hgs_s_tq[8];          // finer stamps of the tile in flight (thread 0): ranking sub-phases, record rounds
#define HGS_TQ(i) { if (threadIdx.x == 0) hgs_s_tq[(i)] = wall_clock64(); }
#else
#define HGS_TQ(i) {}
#endif

// workgroup barrier that orders LDS traffic only (no global load / atomic is waited for: gathers and the bump
// allocation stay in flight across it).  Every cross-wave hand-off in this kernel goes through LDS.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct __attribute__((aligned(16))) RankLds {
  uint32_t hist[HGS_RANK_NB_MAX + 1];   // bucket counts -> exclusive bases (+ sentinel); from step 5 on: the 16-bit masks in list order
  uint32_t wtot[16];                     // block scan (up to 1024 threads: the large class)
  uint32_t tot16[8];                     // cell-list lengths of the tile, two 16-bit fields per word (word 2 q + (c & 1), field (c >> 1) & 1, q = c >> 2)
  uint32_t dmin, dmax, maxcnt, pad;
};
static_assert(sizeof(uint32_t) * (HGS_RANK_NB_MAX + 1) >= sizeof(uint16_t) * 4096, "the masks of the longest list fit into the histogram");

// Exclusive scan of the NB bucket counts in place (thread = PER = NB / NT consecutive buckets: 1, 2, 4 or 8), the largest
// bucket on the way.  The thread's counts are ONE or TWO vector LDS reads, in front of the barrier and again behind it (kept
// in registers across it they spilled in the eight-keys-per-thread instantiation); a loop of PER dependent reads in front
// and PER read-modify-writes behind was 1.2 of a typical tile's 15.7 us.
template <int NT>
__device__ __forceinline__ void bucket_scan(RankLds& R, uint32_t NB) {
  int tid = (int)threadIdx.x;
  asm volatile("" : "+v"(tid));        // (keeps the thread's LDS address local: hoisted out of the persistent tile loop it is spilled)
  const int lane = tid & 63, wv = tid >> 6;
  const uint32_t PER = NB / (uint32_t)NT;                 // (NB is a power of two >= 256 and >= NT for every caller)
  uint32_t* __restrict__ h = R.hist + (uint32_t)tid * PER;
  uint32_t c[8];
  auto load_counts = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = 0u;
    if (PER == 8u) {
      const uint4 a = *reinterpret_cast<const uint4*>(h), b = *reinterpret_cast<const uint4*>(h + 4);
      c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
    } else if (PER == 4u) {
      const uint4 a = *reinterpret_cast<const uint4*>(h);
      c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w;
    } else if (PER == 2u) {
      const uint2 a = *reinterpret_cast<const uint2*>(h);
      c[0] = a.x; c[1] = a.y;
    } else {
      c[0] = h[0];
    }
  };
  load_counts();
  uint32_t sum = 0, mx = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { sum += c[i]; mx = max(mx, c[i]); }
  const uint32_t inc = hgs_wave_incl_scan(sum);
  mx = hgs_wave_max_u32(mx);
  if (lane == 63) R.wtot[wv] = inc;
  if (lane == 0 && mx) atomicMax(&R.maxcnt, mx);
  uint32_t base = inc - sum;
  asm volatile("" : "+v"(base));                           // (the counts are re-read, not carried)
  lds_barrier();
  load_counts();
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) base += (w < wv) ? R.wtot[w] : 0u;
  uint32_t e[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { e[i] = base; base += c[i]; }
  if (PER == 8u) {
    *reinterpret_cast<uint4*>(h) = make_uint4(e[0], e[1], e[2], e[3]);
    *reinterpret_cast<uint4*>(h + 4) = make_uint4(e[4], e[5], e[6], e[7]);
  } else if (PER == 4u) {
    *reinterpret_cast<uint4*>(h) = make_uint4(e[0], e[1], e[2], e[3]);
  } else if (PER == 2u) {
    *reinterpret_cast<uint2*>(h) = make_uint2(e[0], e[1]);
  } else {
    h[0] = e[0];
  }
}

// Cell tables + cell lists of one tile from the masks of its records in list order (what gather_records_single does
// behind its sweep 1).  All NT threads of the workgroup take part; wave 0 has ISSUED the range allocation (`ca`).
template <int NT, bool CH>
__device__ __forceinline__ void cell_lists_from_masks(const View& v, const Layout& L, uint32_t order_pos, uint32_t start, uint32_t n,
                                                      const uint16_t* __restrict__ masks, GatherLds<64>& S, const CellAlloc& ca,
                                                      unsigned long long* tp) {
  const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int nwaves = NT / 64;
  const uint32_t nch = (n + 63u) / 64u;                 // <= 64
  // (1) per 64-record chunk: packed per-cell counts (tab[ch][0..3]) and the chunk's pairs (tab[ch][16])
  for (uint32_t ch = wv; ch < nch; ch += nwaves) {
    const uint32_t k = ch * 64u + lane;
    const uint32_t mask = (k < n) ? (uint32_t)masks[k] : 0u;
    uint32_t tot[4];
#pragma unroll
    for (int wd = 0; wd < 4; ++wd) tot[wd] = hgs_wave_incl_scan(hgs_spread4((mask >> (4 * wd)) & 0xfu));
    if (lane == 63) {
      S.tab[ch][0] = tot[0]; S.tab[ch][1] = tot[1]; S.tab[ch][2] = tot[2]; S.tab[ch][3] = tot[3];
      S.tab[ch][16] = hgs_bytesum(tot[0]) + hgs_bytesum(tot[1]) + hgs_bytesum(tot[2]) + hgs_bytesum(tot[3]);
    }
  }
  lds_barrier();
  // (2) exclusive prefix over the chunks for the 16 cells (unpacked to 32 bits in place, after every wave has read
  // its packed words) and for the pairs
  {
    uint32_t pk[4];
    const uint32_t ch = (uint32_t)lane;
#pragma unroll
    for (int wd = 0; wd < 4; ++wd) pk[wd] = (ch < nch) ? S.tab[ch][wd] : 0u;
    lds_barrier();
    for (int col = wv; col < 5; col += nwaves) {
      if (col < 4) {
        const uint32_t word = col == 0 ? pk[0] : col == 1 ? pk[1] : col == 2 ? pk[2] : pk[3];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const uint32_t x = (word >> (8 * b)) & 0xffu;
          const uint32_t inc = hgs_wave_incl_scan(x);
          if (ch < nch) S.tab[ch][4 * col + b] = inc - x;
        }
      } else {
        const uint32_t x = ch < nch ? S.tab[ch][16] : 0u;
        const uint32_t inc = hgs_wave_incl_scan(x);
        if (ch < nch) S.tab[ch][16] = inc - x;
      }
    }
  }
#ifdef HGS_TIMELINE
  tp[2] = wall_clock64();
#endif
  // (3) the ranges have arrived: cell_info, work items, list bases
  if (tid < 64) hgs_alloc_cell_ranges_finish(v, L, (int)L.tile_rec[order_pos].x, ca, S.cell_base, S.pair_base);   // (the tile id re-read: one register less across the kernel)
  lds_barrier();
#ifdef HGS_TIMELINE
  tp[3] = wall_clock64();
#endif
  // (4) cell lists and pair slots: 16 uniform iterations, the list bases of the chunk travel through v_readlane
  const uint32_t pair_base = S.pair_base;
  for (uint32_t ch = wv; ch < nch; ch += nwaves) {
    const uint32_t k = ch * 64u + lane;
    const bool in = k < n;
    const uint32_t mask = in ? (uint32_t)masks[k] : 0u;
    uint32_t ex[4], before = 0;                          // records of this chunk before this lane, per cell (bytes)
    uint32_t tot[4];                                     // (CH) records of the chunk, per cell (bytes)
#pragma unroll
    for (int wd = 0; wd < 4; ++wd) {
      const uint32_t mine = hgs_spread4((mask >> (4 * wd)) & 0xfu);
      const uint32_t inc = hgs_wave_incl_scan(mine);
      ex[wd] = inc - mine;
      if (CH) tot[wd] = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
      before += hgs_bytesum(ex[wd]);
    }
    const uint32_t rel_chunk = pair_base + S.tab[ch][16];          // first pair id of the chunk
    const uint32_t rel = rel_chunk + before;                       // first pair id of this entry (entry-major)
    if (in) L.entpair[start + k].y = CH ? rel_chunk : rel;
    const uint32_t cb = lane < 16 ? S.cell_base[lane] + S.tab[ch][lane] : 0u;
    uint32_t cp = 0;                                     // (CH) pairs of the chunk in the cells before c
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const uint32_t cbase = (uint32_t)__builtin_amdgcn_readlane((int)cb, c);
      if ((mask >> c) & 1u) {
        const uint32_t exc = (ex[c >> 2] >> (8 * (c & 3))) & 0xffu;
        const uint32_t slot = cbase + exc;
        hgs_put_pair(L, start + k, CH ? rel_chunk + cp + exc : rel + (uint32_t)__popc(mask & ((1u << c) - 1u)), slot);
      }
      if (CH) cp += (tot[c >> 2] >> (8 * (c & 3))) & 0xffu;
    }
  }
}

// Steps 1-4 for one tile of n <= E * NT keys: leaves (Gaussian index | rank << 32) of source position k in pairs[k]
// (k = e * NT + thread: every thread reads back only what it wrote).  keyp = the thread's first keys, already loaded.
// Returns true when the depth distribution was degenerate and the ranks came from the bitonic network (then the pair of
// source position k belongs to LIST position k and the caller's prefetched gathers are stale).
template <int E, int NT>
__device__ __forceinline__ bool rank_keys(const Layout& L, uint32_t start, uint32_t n, uint32_t NB, const u64 (&keyp)[HGS_RANK_GU],
                                          unsigned long long* pairs, RankLds& R) {
  int tid = (int)threadIdx.x;
  asm volatile("" : "+v"(tid));        // (the key offsets e * NT + tid stay local to the tile: hoisted out of the persistent loop they spill)
  const int lane = tid & 63, wv = tid >> 6;
  u64 key[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    if (e < HGS_RANK_GU) key[e] = keyp[e];
    else {                                               // (unconditional load from a clamped slot: a load inside a branch is
      const uint32_t k = (uint32_t)e * NT + (uint32_t)tid;      // waited for on the spot, one key at a time)
      const u64 kv = L.keys[start + min(k, n - 1u)];
      key[e] = (k < n) ? kv : ~0ull;
    }
  }
  // ---- 2. depth range
  {
    uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const uint32_t d = (uint32_t)(key[e] >> 32);
      lo = min(lo, d);                                   // (padding: 0xffffffff never lowers the minimum)
      hi = max(hi, key[e] != ~0ull ? d : 0u);
    }
    hi = hgs_wave_max_u32(hi);
    lo = ~hgs_wave_max_u32(~lo);
    lds_barrier();                                       // the zeroed histogram / range words are in place
    if (lane == 0) { atomicMin(&R.dmin, lo); atomicMax(&R.dmax, hi); }
  }
  lds_barrier();
  HGS_TQ(0)
  // ---- 3. histogram
  const float flo = __uint_as_float(R.dmin);
  const float range = __uint_as_float(R.dmax) - flo;
  const float scale = range > 1e-30f ? (float)NB / range : 0.0f;
  uint32_t bp[E];                                          // bucket | arrival position << 12, later the rank
#pragma unroll
  for (int e = 0; e < E; ++e) {
    bp[e] = 0u;
    if (key[e] != ~0ull) {
      const float x = (__uint_as_float((uint32_t)(key[e] >> 32)) - flo) * scale;
      const uint32_t bkt = (uint32_t)fminf(x, (float)(NB - 1u));
      bp[e] = bkt | (atomicAdd(&R.hist[bkt], 1u) << 12);
    }
  }
  lds_barrier();
  HGS_TQ(1)
  // exclusive scan of the NB counts in place (thread = PER consecutive buckets), largest bucket on the way
  bucket_scan<NT>(R, NB);
  if (tid == 0) R.hist[NB] = n;
  lds_barrier();
  HGS_TQ(2)
  const bool degenerate = R.maxcnt > (uint32_t)HGS_RANK_BUCKET_MAX;      // (workgroup-uniform)
  if (!degenerate) {
    // ---- 4. bucket order, then the rank inside the bucket
    constexpr int EG = E < 8 ? E : 8;                      // keys probed together (more: registers)
    uint32_t lmax[E / EG];
#pragma unroll
    for (int h = 0; h < E / EG; ++h) lmax[h] = 0u;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if (key[e] != ~0ull) {
        const uint32_t j0 = R.hist[bp[e] & 0xfffu], j1 = R.hist[(bp[e] & 0xfffu) + 1u];
        pairs[j0 + (bp[e] >> 12)] = key[e];
        bp[e] = j0 | ((j1 - j0) << 12);                    // bucket start | bucket length << 12 | (smaller keys << 20)
        lmax[e / EG] = max(lmax[e / EG], j1 - j0);
      }                                                    // (padding keeps 0: no probe)
    }
    lds_barrier();
    HGS_TQ(3)
    // the probes of the thread's keys interleaved: EG independent LDS reads per step (one key after the other, each
    // with its own data-dependent loop, was 5 us of a 1400-entry tile)
#pragma unroll
    for (int h = 0; h < E / EG; ++h) {
      for (uint32_t sidx = 0; sidx < lmax[h]; ++sidx) {
        u64 other[EG];
#pragma unroll
        for (int e = 0; e < EG; ++e) {
          const uint32_t w = bp[h * EG + e];
          const bool on = sidx < ((w >> 12) & 0xffu);
          other[e] = pairs[on ? (w & 0xfffu) + sidx : 0u];        // (unconditional read from a valid slot)
        }
#pragma unroll
        for (int e = 0; e < EG; ++e) {
          const uint32_t w = bp[h * EG + e];
          const bool on = sidx < ((w >> 12) & 0xffu);
          bp[h * EG + e] = w + ((on && other[e] < key[h * EG + e]) ? (1u << 20) : 0u);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) bp[e] = (bp[e] & 0xfffu) + (bp[e] >> 20);
  } else {
    // hundreds of keys in one bucket (exactly equal depths, or a range stretched by an outlier): the bitonic network
    // on the keys in LDS; afterwards the thread adopts the keys at ITS list positions
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const uint32_t k = (uint32_t)e * NT + (uint32_t)tid;
      if (k < n) pairs[k] = key[e];
    }
    __syncthreads();
    bitonic_sort<NT>(pairs, n);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const uint32_t k = (uint32_t)e * NT + (uint32_t)tid;
      key[e] = (k < n) ? pairs[k] : ~0ull;
      bp[e] = k;
    }
  }
  lds_barrier();                                           // every rank is known: `pairs` and the histogram are free
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const uint32_t k = (uint32_t)e * NT + (uint32_t)tid;
    if (k < n) pairs[k] = (key[e] & 0xffffffffull) | ((u64)bp[e] << 32);
  }
  return degenerate;
}

// The same steps for LONG lists (more than four keys per thread) without per-key register state: every phase re-reads
// the keys from the tile's key segment (coalesced, L2-resident) and only the ranks stay in registers (16 keys and
// their bucket words per thread did not fit beside everything else: scratch spills).  The scatter takes its slot from a
// second LDS atomic on the scanned histogram (hist[b] then ends bucket b), so no arrival position has to be kept.
// SORTED: leave the KEYS in list order in `pairs` (what the large class hands to gather_records_single) instead of
// (Gaussian | rank << 32) by source position.  Lists of up to 16 * NT keys; bucket starts travel in JB bits.
template <int NT, bool SORTED = false>
__device__ __forceinline__ bool rank_keys_stream(const Layout& L, uint32_t start, uint32_t n, uint32_t NB,
                                                 unsigned long long* pairs, RankLds& R) {
  const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int JB = NT >= 1024 ? 14 : 12;              // bits of a bucket start (< 16 * NT)
  constexpr uint32_t JM = (1u << JB) - 1u;
  const unsigned long long* __restrict__ keys = L.keys + start;
  const uint32_t last = n - 1u;
  // ---- 2. depth range
  {
    uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll 4
    for (uint32_t k = tid; k < n; k += NT) {
      const uint32_t d = (uint32_t)(keys[k] >> 32);
      lo = min(lo, d); hi = max(hi, d);
    }
    hi = hgs_wave_max_u32(hi);
    lo = ~hgs_wave_max_u32(~lo);
    lds_barrier();                                       // the zeroed histogram / range words are in place
    if (lane == 0) { atomicMin(&R.dmin, lo); atomicMax(&R.dmax, hi); }
  }
  lds_barrier();
  const float flo = __uint_as_float(R.dmin);
  const float range = __uint_as_float(R.dmax) - flo;
  const float scale = range > 1e-30f ? (float)NB / range : 0.0f;
  const float bmax = (float)(NB - 1u);
  auto bucket_of = [&](u64 key) { return (uint32_t)fminf((__uint_as_float((uint32_t)(key >> 32)) - flo) * scale, bmax); };
  // ---- 3. histogram, scan
#pragma unroll 4
  for (uint32_t k = tid; k < n; k += NT) atomicAdd(&R.hist[bucket_of(keys[k])], 1u);
  lds_barrier();
  bucket_scan<NT>(R, NB);
  lds_barrier();
  const bool degenerate = R.maxcnt > (uint32_t)HGS_RANK_BUCKET_MAX;      // (workgroup-uniform)
  uint32_t rk[16];
  if (!degenerate) {
    // ---- 4. bucket order (slot = the bucket's cursor), then the rank inside the bucket, four keys probed together
#pragma unroll 4
    for (uint32_t k = tid; k < n; k += NT) {
      const u64 key = keys[k];
      pairs[atomicAdd(&R.hist[bucket_of(key)], 1u)] = key;
    }
    lds_barrier();                                         // hist[b] = end of bucket b = start of bucket b + 1
#pragma unroll
    for (int e0 = 0; e0 < 16; e0 += 4) {
      if ((uint32_t)e0 * NT >= n) break;                   // (workgroup-uniform)
      u64 key[4];
      uint32_t w[4], lmax = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t k = (uint32_t)(e0 + e) * NT + (uint32_t)tid;
        key[e] = keys[min(k, last)];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t k = (uint32_t)(e0 + e) * NT + (uint32_t)tid;
        const uint32_t b = bucket_of(key[e]);
        const uint32_t j1 = R.hist[b], j0 = b ? R.hist[b - 1u] : 0u;
        const uint32_t len = k < n ? j1 - j0 : 0u;
        w[e] = j0 | (len << JB);                           // bucket start | bucket length << JB | (smaller keys << (JB + 8))
        lmax = max(lmax, len);
      }
      for (uint32_t sidx = 0; sidx < lmax; ++sidx) {
        u64 other[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool on = sidx < ((w[e] >> JB) & 0xffu);
          other[e] = pairs[on ? (w[e] & JM) + sidx : 0u];         // (unconditional read from a valid slot)
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool on = sidx < ((w[e] >> JB) & 0xffu);
          w[e] += (on && other[e] < key[e]) ? (1u << (JB + 8)) : 0u;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) rk[e0 + e] = (w[e] & JM) + (w[e] >> (JB + 8));
    }
    lds_barrier();                                         // every probe is done: `pairs` is free
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const uint32_t k = (uint32_t)e * NT + (uint32_t)tid;
      if (k < n) {
        if (SORTED) pairs[rk[e]] = keys[k];
        else pairs[k] = (keys[k] & 0xffffffffull) | ((u64)rk[e] << 32);
      }
    }
    if (SORTED) lds_barrier();
  } else {
    // degenerate depths: the bitonic network on the keys in LDS; list position k then holds (Gaussian, rank = k)
    for (uint32_t k = tid; k < n; k += NT) pairs[k] = keys[k];
    __syncthreads();
    bitonic_sort<NT>(pairs, n);
    if (!SORTED) {
      for (uint32_t k = tid; k < n; k += NT) pairs[k] = (pairs[k] & 0xffffffffull) | ((u64)k << 32);
      lds_barrier();
    }
  }
  return degenerate;
}

template <int NT, bool CH>
__device__ __forceinline__ void rank_sort_tile(const View& v, const Layout& L, uint32_t order_pos, int g, uint32_t start, uint32_t n,
                                               unsigned long long* pairs, RankLds& R, GatherLds<64>& S, bool& degenerate_out,
                                               unsigned long long* tp) {
  const int tid = (int)threadIdx.x, lane = tid & 63;
  constexpr int GU = HGS_RANK_GU;
  const int t = g % v.T;
  const GeomRec* __restrict__ geom = L.geom + (size_t)(g / v.T) * v.P;
  const uint32_t* __restrict__ cbase = L.chunk_base + (size_t)(g / v.T) * v.nblk;
  const int tx = t % v.grid_x, ty = t / v.grid_x;
  const float x0 = (float)(tx * HGS_TILE), y0 = (float)(ty * HGS_TILE);
  uint32_t NB = 256u;
  while (NB < 2u * n && NB < (uint32_t)HGS_RANK_NB_MAX) NB <<= 1;      // (wave-uniform)
  for (uint32_t i = tid; i <= NB; i += NT) R.hist[i] = 0u;
  if (tid < 8) R.tot16[tid] = 0u;
  if (tid == 0) { R.dmin = 0xffffffffu; R.dmax = 0u; R.maxcnt = 0u; }
  // ---- 1. the thread's first keys; their gathers leave at once (unconditional: padding re-reads record 0)
  u64 keyp[GU];
#pragma unroll
  for (int u = 0; u < GU; ++u) {
    const uint32_t k = (uint32_t)u * NT + (uint32_t)tid;
    const u64 kv = L.keys[start + min(k, n - 1u)];          // (unconditional, clamped)
    keyp[u] = (k < n) ? kv : ~0ull;
  }
  // two rounds of gathers in flight: A and B, plain register arrays (a struct handed to a lambda stayed in scratch memory)
  uint4 Aa[GU], Ab[GU], Ac[GU], Ba[GU], Bb[GU], Bc[GU];     // GeomRec words 0-11
  uint32_t Aoff[GU], Acb[GU], Boff[GU], Bcb[GU];            // entry-id offset inside the chunk, the chunk's base
  u64 Apr[GU], Bpr[GU];                                     // Gaussian | rank << 32, ~0 = none
#define HGS_RANK_ISSUE(X, u, PR)                                                              \
  {                                                                                           \
    const u64 pr__ = (PR);                                                                    \
    const uint32_t idx__ = pr__ == ~0ull ? 0u : (uint32_t)pr__;                               \
    const uint4* gp__ = reinterpret_cast<const uint4*>(&geom[idx__]);                         \
    X##a[u] = gp__[0]; X##b[u] = gp__[1]; X##c[u] = gp__[2]; X##off[u] = gp__[3].x;           \
    X##cb[u] = cbase[idx__ >> 8];                                                             \
    X##pr[u] = pr__;                                                                          \
  }
  // ---- 2-4. ranks.  Lists of up to 4 keys per thread keep keys and bucket words in registers and send their first
  // gathers out BEFORE the sort; longer ones stream the keys per phase and start the gathers behind the sort: their
  // many rounds amortise one exposed round trip.
  bool degenerate;
  const bool early = n <= 8u * NT;
  if (early) {
#pragma unroll
    for (int u = 0; u < GU; ++u) HGS_RANK_ISSUE(A, u, keyp[u]);
    if (n <= 2u * NT) degenerate = rank_keys<2, NT>(L, start, n, NB, keyp, pairs, R);
    else if (n <= 4u * NT) degenerate = rank_keys<4, NT>(L, start, n, NB, keyp, pairs, R);
    else degenerate = rank_keys<8, NT>(L, start, n, NB, keyp, pairs, R);
  } else {
    degenerate = rank_keys_stream<NT>(L, start, n, NB, pairs, R);
  }
  degenerate_out = degenerate;
#ifdef HGS_TIMELINE
  tp[0] = wall_clock64();
#endif
  // ---- 5. records + masks at the rank: rounds of GU keys per thread, two rounds in flight (A / B), the (Gaussian,
  // rank) pairs read back from LDS; per-cell totals of the thread's masks on the way (bytes: <= 16 keys per thread)
  uint16_t* __restrict__ masks = reinterpret_cast<uint16_t*>(R.hist);
  uint32_t acc[4] = {0u, 0u, 0u, 0u};
#define HGS_RANK_ROUND(X, r)       /* the pairs of round r and their gathers */                \
  _Pragma("unroll") for (int u = 0; u < GU; ++u) {                                            \
    const uint32_t k__ = ((r) * (uint32_t)GU + (uint32_t)u) * NT + (uint32_t)tid;             \
    HGS_RANK_ISSUE(X, u, (k__ < n) ? pairs[k__] : ~0ull);                                     \
  }
#define HGS_RANK_CONSUME(X)                                                                   \
  _Pragma("unroll") for (int u = 0; u < GU; ++u) {                                            \
    if (X##pr[u] != ~0ull) {                                                                  \
      const uint32_t k = (uint32_t)(X##pr[u] >> 32);                                          \
      const uint4 g0 = X##a[u], g1 = X##b[u], g2 = X##c[u];                                   \
      /* g0: mx my ca cb | g1: cc op r g | g2: b depth rect_lo rect_hi */                     \
      const int minx = g2.z & 0xffffu, miny = g2.z >> 16, maxx = g2.w & 0xffffu;              \
      const uint32_t entry = X##cb[u] + X##off[u] + (uint32_t)((ty - miny) * (maxx - minx) + (tx - minx)); \
      const float mx = __uint_as_float(g0.x), my = __uint_as_float(g0.y);                     \
      const float ca = __uint_as_float(g0.z), cb = __uint_as_float(g0.w), cc = __uint_as_float(g1.x); \
      const uint32_t mask = hgs_cell_mask(mx, my, ca, cb, cc, __uint_as_float(g1.y), x0, y0); \
      uint4* dst = reinterpret_cast<uint4*>(&L.recs[start + k]);                              \
      const float qa = -0.5f * ca * HGS_LOG2E, qb = -cb * HGS_LOG2E, qc = -0.5f * cc * HGS_LOG2E; \
      dst[0] = make_uint4(g0.x, g0.y, __float_as_uint(qa), __float_as_uint(qb));              \
      dst[1] = make_uint4(__float_as_uint(qc), g1.y, g1.z, g1.w);                             \
      dst[2] = make_uint4(g2.x, g2.y, entry, CH ? hgs_rec_tag(mask, k, n, true) : 0u);        \
      L.entpair[start + k].x = entry | ((uint32_t)__popc(mask) << 27);   /* (.y, the first pair id, follows with the lists) */ \
      masks[k] = (uint16_t)mask;                                                              \
      _Pragma("unroll") for (int wd = 0; wd < 4; ++wd) acc[wd] += hgs_spread4((mask >> (4 * wd)) & 0xfu); \
    }                                                                                         \
  }
  if (degenerate || !early) {                              // (degenerate: the thread now owns other Gaussians - gather again)
    HGS_RANK_ROUND(A, 0u)
  } else {
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const uint32_t k = (uint32_t)u * NT + (uint32_t)tid;
      Apr[u] = (k < n) ? pairs[k] : ~0ull;
    }
  }
  const uint32_t per_round = (uint32_t)GU * NT;
  for (uint32_t r = 0; r * per_round < n; r += 2u) {       // (workgroup-uniform trip count)
    const bool hb = (r + 1u) * per_round < n, ha = (r + 2u) * per_round < n;
    if (hb) { HGS_RANK_ROUND(B, r + 1u) }
    HGS_RANK_CONSUME(A)
    if (r < 4u) HGS_TQ(4 + r)        // (timeline builds: round ends)
    if (ha) { HGS_RANK_ROUND(A, r + 2u) }
    if (hb) { HGS_RANK_CONSUME(B) }
    if (r < 3u) HGS_TQ(5 + r)
  }
#undef HGS_RANK_ISSUE
#undef HGS_RANK_ROUND
#undef HGS_RANK_CONSUME
  // per-cell totals of the wave -> the tile's (two 16-bit fields per word: <= 1024 per wave, <= 4096 per tile)
  {
    uint32_t w16[8];
#pragma unroll
    for (int wd = 0; wd < 4; ++wd) { w16[2 * wd] = acc[wd] & 0x00ff00ffu; w16[2 * wd + 1] = (acc[wd] >> 8) & 0x00ff00ffu; }
#pragma unroll
    for (int i = 0; i < 8; ++i) w16[i] = hgs_wave_incl_scan(w16[i]);
    if (lane == 63) {
#pragma unroll
      for (int i = 0; i < 8; ++i) if (w16[i]) atomicAdd(&R.tot16[i], w16[i]);
    }
  }
  lds_barrier();
#ifdef HGS_TIMELINE
  tp[1] = wall_clock64();
#endif
  // the bump allocation of the tile's ranges leaves now; the cell tables are built while it travels
  CellAlloc ca;
  if (tid < 64) {
    const int c = lane & 15;
    const uint32_t word = R.tot16[2 * (c >> 2) + (c & 1)];
    hgs_alloc_cell_ranges_issue(L, order_pos, (word >> (16 * ((c >> 1) & 1))) & 0xffffu, ca);
  }
  cell_lists_from_masks<NT, CH>(v, L, order_pos, start, n, masks, S, ca, tp);
}

template <int NT, bool CH>
__device__ __forceinline__ void sort_rank_body(const View& v, const Layout& L, const hgs_status* __restrict__ status,
                                               unsigned long long* pairs, RankLds& R, GatherLds<64>& S) {
  if (status->overflow) return;
  const uint32_t active = status->active_tiles;
  // Light neighbours for the heaviest tiles.  Workgroups b, b + ncu, b + 2 ncu of a 3-per-CU grid share a CU (observed:
  // tools/timeline.py), and tile_order is heavy first: the heaviest tiles - whose chains are the kernel's length - sat beside
  // mid-sized ones that kept the CU's issue slots and memory queue busy through their record rounds.  In the first pass
  // the HGS_SORT_LIGHT workgroups beside each of the first HGS_SORT_LIGHT positions swap tiles with the workgroups of
  // the LAST positions (the lightest tiles: done in a third of the time, a tenth of the traffic).  Placement only: every
  // position is processed once, by whichever workgroup (the die its tables are filed under follows the POSITION).
  const uint32_t G = gridDim.x, ncu = G / 3u, na = min(active, G);
  const bool remap = HGS_SORT_LIGHT > 0 && G == 3u * ncu && status->reserved[1] >= 768u &&
                     na >= 2u * ncu + 3u * (uint32_t)HGS_SORT_LIGHT;
  for (uint32_t b0 = blockIdx.x; b0 < active; b0 += gridDim.x) {
    uint32_t b = b0;
    if (remap && b0 < na) {
      constexpr uint32_t H = (uint32_t)HGS_SORT_LIGHT;
      if (b0 >= ncu && b0 < ncu + H) b = na - 1u - (b0 - ncu);
      else if (b0 >= 2u * ncu && b0 < 2u * ncu + H) b = na - 1u - H - (b0 - 2u * ncu);
      else if (b0 >= na - 2u * H) { const uint32_t j = na - 1u - b0; b = j < H ? ncu + j : 2u * ncu + (j - H); }
    }
    const uint4 tr = L.tile_rec[b];                      // (tile, entries, first entry): one load
    const int g = (int)tr.x;
    const uint32_t n = tr.y, start = tr.z;
    if (n == 0 || n > (uint32_t)HGS_SORT_LDS_MAX) continue;   // (longer lists: hgs_k_sort_large / _huge)
#ifdef HGS_TIMELINE
    const unsigned long long tp0 = wall_clock64();
#endif
    bool degenerate;
    unsigned long long tp[4] = {0, 0, 0, 0};
    rank_sort_tile<NT, CH>(v, L, b, g, start, n, pairs, R, S, degenerate, tp);
#ifdef HGS_TIMELINE
    if (threadIdx.x == 0 && b < HGS_TL_SLOTS) {          // per tile (position in tile_order), wave 0: kernel ids 5 and 2
      const unsigned long long tpe = wall_clock64();
      auto cl = [](unsigned long long d) { return d > 0xffffull ? 0xffffull : d; };
      hgs_tl[5][b][0] = tp0;                             // absolute start / end (10 ns ticks)
      hgs_tl[5][b][1] = tpe;
      hgs_tl[5][b][2] = degenerate ? 1 : 0;
      hgs_tl[5][b][3] = ((unsigned long long)1 << 32) | n;
      // phases: ranks | records | tables | allocation (its exposed rest)   and   cell lists
      hgs_tl[2][b][0] = cl(tp[0] - tp0) | (cl(tp[1] - tp[0]) << 16) | (cl(tp[2] - tp[1]) << 32) | (cl(tp[3] - tp[2]) << 48);
      hgs_tl[2][b][1] = cl(tpe - tp[3]);
      // ranking: range | histogram | scan | scatter (then the probes up to tp[0]); record rounds 0..3 (ends, from tp[0])
      hgs_tl[2][b][2] = cl(hgs_s_tq[0] - tp0) | (cl(hgs_s_tq[1] - hgs_s_tq[0]) << 16) | (cl(hgs_s_tq[2] - hgs_s_tq[1]) << 32) | (cl(hgs_s_tq[3] - hgs_s_tq[2]) << 48);
      hgs_tl[2][b][3] = cl(hgs_s_tq[4] - tp[0]) | (cl(hgs_s_tq[5] - tp[0]) << 16) | (cl(hgs_s_tq[6] - tp[0]) << 32) | (cl(hgs_s_tq[7] - tp[0]) << 48);
    }
#endif
    lds_barrier();                                       // the LDS tables are reused by the next tile
  }
}
}  // namespace

static_assert(16 * HGS_SORT_NT >= 4096, "the rank sort takes every list of the LDS class");
// Two instantiations: pair-row ids entry-major (calls of 1-2 views) / chunk-cell-major (hgs_k_sort_lds_ch: >= 3 views,
// View::pairchunks) - a compile-time choice inside the kernel, so neither form pays for the other's id arithmetic.
#define HGS_SORT_LDS_KERNEL(NAME, CH)                                                                                  \
  extern "C" __global__ void __launch_bounds__(HGS_SORT_NT) __attribute__((amdgpu_waves_per_eu(HGS_SORT_WAVES_PER_EU, HGS_SORT_WAVES_PER_EU))) \
  NAME(View v, Layout L, const hgs_status* __restrict__ status) {                                                      \
    __shared__ unsigned long long pairs[4096];                                                                         \
    __shared__ GatherLds<64> S;                                                                                        \
    __shared__ RankLds R;                                                                                              \
    HGS_TL_BEGIN();                                                                                                    \
    sort_rank_body<HGS_SORT_NT, CH>(v, L, status, pairs, R, S);                                                        \
    HGS_TL_END(3, 0u);                                                                                                 \
  }
HGS_SORT_LDS_KERNEL(hgs_k_sort_lds, false)
HGS_SORT_LDS_KERNEL(hgs_k_sort_lds_ch, true)

extern "C" __global__ void __launch_bounds__(1024)
hgs_k_sort_large(View v, Layout L, const hgs_status* __restrict__ status) {
  __shared__ unsigned long long keys[16384];
  __shared__ GatherLds<256> S;
  if (status->overflow) return;
  const uint32_t b = blockIdx.x;
  if (b >= status->active_tiles) return;
  const int t = (int)L.tile_order[b];
  const uint32_t start = L.tile_start[t];
  const uint32_t n = L.tile_n[t];
  if (n <= (uint32_t)HGS_SORT_LDS_MAX || n > 16384u) return;
  // long lists (4097 .. 16384 entries): the same bucket ranking as hgs_k_sort_lds, streaming form, 16 keys per thread at
  // most; the keys land in list order in LDS and the gather below takes over.  (The bitonic network it replaces - 91
  // stages of two barriers for 8192 padded keys - was 45 of the 90 us ONE 4411-entry tile cost configs[3].)
  {
    __shared__ RankLds R;
    const uint32_t NB = (uint32_t)HGS_RANK_NB_MAX;
    for (uint32_t i = threadIdx.x; i <= NB; i += 1024) R.hist[i] = 0u;
    if (threadIdx.x == 0) { R.dmin = 0xffffffffu; R.dmax = 0u; R.maxcnt = 0u; }
    rank_keys_stream<1024, true>(L, start, n, NB, keys, R);
    __syncthreads();
  }
  gather_records_single<256>(v, L, t, b, start, n, keys, 1024, S);
}

// Tiles with n > 16384 (longer than LDS): the same network run in place on the tile's key
// segment in HBM by one 1024-thread workgroup (all its waves sit on one CU and share its
// L1; __syncthreads() is the workgroup-scope release/acquire).  Rare (e.g. 500k @ 512^2).
extern "C" __global__ void __launch_bounds__(1024)
hgs_k_sort_huge(View v, Layout L, const hgs_status* __restrict__ status) {
  if (status->overflow) return;
  const uint32_t b = blockIdx.x;
  if (b >= status->active_tiles) return;
  const int t = (int)L.tile_order[b];
  const uint32_t start = L.tile_start[t];
  const uint32_t n = L.tile_n[t];
  if (n <= 16384u) return;
  __shared__ GatherLds<256> S;
  bitonic_sort<1024>(L.keys + start, n);
  gather_records<256>(v, L, t, b, start, n, L.keys + start, 1024, S);
}
