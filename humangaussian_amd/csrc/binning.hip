// binning.hip - tile binning and per-tile depth sort (stages F2-F5 of SURVEY.md 2.3(B)).
//
// Upstream: InclusiveSum over Gaussians -> duplicateWithKeys -> one GLOBAL 64-bit radix
// sort of all (tile|depth) keys -> identifyTileRanges.  Here (MI355X-first):
//   1. hgs_k_scan      one workgroup: scans the per-workgroup tiles_touched sums and the
//                      per-tile counts (from the preprocess atomics), orders tiles heavy-first
//                      for scheduling, lays out bucket-state and backward-workgroup prefixes,
//                      publishes hgs_status.
//   2. hgs_k_fill      per Gaussian: entry-id prefix + scatter (depth_bits<<32 | idx) keys into
//                      its tiles' list segments (order inside a tile is arbitrary here).
//   3. hgs_k_sort_*    per tile: bitonic sort of the tile's keys IN LDS (unique keys =>
//                      deterministic result = upstream's stable order: depth, ties by index),
//                      then gathers the Gaussians into a depth-ordered, contiguous 48-byte
//                      record list ("duplicated Gaussian list") that both blend kernels stream.
//
// Roofline: HBM/latency-bound integer work: 8 B/entry written + read for keys, one 64 B
// gather + 48 B write per entry for the records.
#include "hgs_common.h"

namespace {
constexpr int SCAN_NT = 1024;
constexpr int SCAN_ITEMS = 4;      // tiles per thread per pass
constexpr int SCAN_LDS_TILES = 14336;   // 56 KB of dynamic LDS (stays under the 64 KB default limit)

// exclusive scan of three counters at once over the workgroup (one barrier pair)
constexpr int NSCAN = 4;
__device__ __forceinline__ void block_excl_scanN(const uint32_t (&v)[NSCAN], uint32_t (*wtot)[NSCAN],
                                                 uint32_t (&ex)[NSCAN], uint32_t (&tot)[NSCAN]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc[NSCAN];
#pragma unroll
  for (int q = 0; q < NSCAN; ++q) inc[q] = hgs_wave_incl_scan(v[q]);
  __syncthreads();
  if (lane == 63) {
#pragma unroll
    for (int q = 0; q < NSCAN; ++q) wtot[w][q] = inc[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NSCAN; ++q) {
    uint32_t b = 0, t = 0;
#pragma unroll
    for (int k = 0; k < SCAN_NT / 64; ++k) {
      const uint32_t x = wtot[k][q];
      if (k < w) b += x;
      t += x;
    }
    ex[q] = b + inc[q] - v[q];
    tot[q] = t;
  }
}
}  // namespace

// ---------------------------------------------------------------------------- 1. scan
// One workgroup.  LDS-bin path: the per-tile count is the sum of the HGS_ROW_GROUPS group
// totals hgs_k_colscan left in tile_grp[rg][t]; this kernel turns them into ABSOLUTE
// bases tile_start[t] + (entries of earlier row groups).
extern "C" __global__ void __launch_bounds__(SCAN_NT)
hgs_k_scan(View v, Layout L, hgs_status* __restrict__ status,
           hgs_status* __restrict__ status_host) {
  __shared__ uint32_t wtot[SCAN_NT / 64];
  __shared__ uint32_t wtotN[SCAN_NT / 64][NSCAN];
  __shared__ uint32_t carry_s;
  __shared__ uint32_t carry3[NSCAN];
  __shared__ uint32_t cls_hist[33];
  __shared__ uint32_t cls_base[33];
  __shared__ uint32_t max_n_s;
  // dynamic LDS (launched with 4*T bytes when T <= SCAN_LDS_TILES, else 0): the per-tile counts
  // stay on chip for phase (c) - each global round trip of this single workgroup costs 1.5-3 us
  extern __shared__ uint32_t s_dyn[];
  const bool lds_tiles = v.T <= SCAN_LDS_TILES;
  uint32_t* s_n = s_dyn;
  const int tid = threadIdx.x;

  // Three independent jobs.  On the LDS-bin path they run as three workgroups of one launch
  // (role = blockIdx.x) so that their latency chains overlap; on the global-atomic path (one
  // workgroup) they run one after the other.
  //   role 1: (a) block_base      role 0: (b) tile tables + status      role 2: (c) tile_order
  const int role = (gridDim.x > 1) ? (int)blockIdx.x : -1;
  if (tid == 0) { carry_s = 0; carry3[0] = carry3[1] = carry3[2] = carry3[3] = 0; max_n_s = 0; }
  if (tid < 33) cls_hist[tid] = 0;
  __syncthreads();
  // (a) exclusive scan of per-chunk tiles_touched sums -> block_base
  if (role == -1 || role == 1)
  for (int base = 0; base < v.nblk; base += SCAN_NT) {
    const int k = base + tid;
    const uint32_t val = (k < v.nblk) ? L.block_sums[k] : 0u;
    uint32_t total;
    const uint32_t ex = hgs_block_excl_scan<SCAN_NT>(val, wtot, total);
    const uint32_t carry = carry_s;
    if (k < v.nblk) L.block_base[k] = carry + ex;
    __syncthreads();
    if (tid == 0) carry_s = carry + total;
    __syncthreads();
  }
  if (role == 1) return;

  // (b) tile_start / bucket-state prefix / backward-workgroup prefix; class histogram
  // (role 2 runs the same loop for the per-tile counts and the class histogram only)
  for (int base = 0; base < v.T; base += SCAN_NT * SCAN_ITEMS) {
    const int t0 = base + tid * SCAN_ITEMS;
    uint32_t n[SCAN_ITEMS], grp[HGS_ROW_GROUPS][SCAN_ITEMS];
    // one CU does all of this: 16 B/lane vector accesses (4 consecutive tiles per thread)
    // keep its memory pipeline to a handful of fully coalesced instructions
    const bool vec = ((v.T & 3) == 0) && (t0 + SCAN_ITEMS <= v.T);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) n[k] = 0;
    if (v.lds_bins) {
#pragma unroll
      for (int rg = 0; rg < HGS_ROW_GROUPS; ++rg) {
        if (vec) {
          const uint4 q = *reinterpret_cast<const uint4*>(L.tile_grp + (size_t)rg * v.T + t0);
          grp[rg][0] = q.x; grp[rg][1] = q.y; grp[rg][2] = q.z; grp[rg][3] = q.w;
        } else {
#pragma unroll
          for (int k = 0; k < SCAN_ITEMS; ++k)
            grp[rg][k] = (t0 + k < v.T) ? L.tile_grp[(size_t)rg * v.T + t0 + k] : 0u;
        }
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) n[k] += grp[rg][k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < SCAN_ITEMS; ++k) n[k] = (t0 + k < v.T) ? L.tile_count[t0 + k] : 0u;
    }
    uint32_t mx = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) mx = max(mx, n[k]);
    if (role != 2) {
    uint32_t l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    uint32_t p0[SCAN_ITEMS], p1[SCAN_ITEMS], p2[SCAN_ITEMS], p3[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
      const uint32_t nb = (n[k] + HGS_BUCKET - 1) / HGS_BUCKET;
      const uint32_t nseg = hgs_nseg(n[k]);
      p0[k] = l0; p1[k] = l1; p2[k] = l2; p3[k] = l3;
      l0 += n[k];
      l1 += nb > 0 ? nb - 1 : 0;                               // stored bucket states
      l2 += nb;                                                // backward workgroups (1 per bucket)
      l3 += nseg > 1 ? nseg : 0;                               // segment planes of long lists
    }
    uint32_t ex[NSCAN], tot[NSCAN];
    const uint32_t lv[NSCAN] = {l0, l1, l2, l3};
    block_excl_scanN(lv, wtotN, ex, tot);
    const uint32_t c0 = carry3[0], c1 = carry3[1], c2 = carry3[2];
    const uint32_t c3 = carry3[3];
    uint32_t ts[SCAN_ITEMS], tb[SCAN_ITEMS], tw[SCAN_ITEMS], tm[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
      ts[k] = c0 + ex[0] + p0[k];
      tb[k] = c1 + ex[1] + p1[k];
      tw[k] = c2 + ex[2] + p2[k];
      tm[k] = c3 + ex[3] + p3[k];
    }
    if (vec) {
      *reinterpret_cast<uint4*>(L.tile_start + t0) = make_uint4(ts[0], ts[1], ts[2], ts[3]);
      *reinterpret_cast<uint4*>(L.tile_bstart + t0) = make_uint4(tb[0], tb[1], tb[2], tb[3]);
      *reinterpret_cast<uint4*>(L.tile_wgstart + t0) = make_uint4(tw[0], tw[1], tw[2], tw[3]);
      *reinterpret_cast<uint4*>(L.tile_maxcontrib + t0) = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(L.tile_msegstart + t0) = make_uint4(tm[0], tm[1], tm[2], tm[3]);
      if (v.lds_bins) {
        uint32_t acc[SCAN_ITEMS] = {ts[0], ts[1], ts[2], ts[3]};
#pragma unroll
        for (int rg = 0; rg < HGS_ROW_GROUPS; ++rg) {
          *reinterpret_cast<uint4*>(L.tile_gbase + (size_t)rg * v.T + t0) =
              make_uint4(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
          for (int k = 0; k < SCAN_ITEMS; ++k) acc[k] += grp[rg][k];
        }
      } else {
        *reinterpret_cast<uint4*>(L.tile_count + t0) = make_uint4(0u, 0u, 0u, 0u);
      }
    } else {
#pragma unroll
      for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int t = t0 + k;
        if (t < v.T) {
          L.tile_start[t] = ts[k];
          L.tile_bstart[t] = tb[k];
          L.tile_wgstart[t] = tw[k];
          L.tile_maxcontrib[t] = 0;
          L.tile_msegstart[t] = tm[k];
          if (v.lds_bins) {
            uint32_t acc = ts[k];
#pragma unroll
            for (int rg = 0; rg < HGS_ROW_GROUPS; ++rg) {
              L.tile_gbase[(size_t)rg * v.T + t] = acc;
              acc += grp[rg][k];
            }
          } else {
            L.tile_count[t] = 0;            // becomes the fill cursor
          }
        }
      }
    }
    __syncthreads();
    if (tid == 0) { carry3[0] = c0 + tot[0]; carry3[1] = c1 + tot[1]; carry3[2] = c2 + tot[2]; carry3[3] = c3 + tot[3]; }
    }  // role != 2
    if (lds_tiles) {
#pragma unroll
      for (int k = 0; k < SCAN_ITEMS; ++k)
        if (t0 + k < v.T) s_n[t0 + k] = n[k];
    }
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
      if (t0 + k < v.T && n[k]) atomicAdd(&cls_hist[32 - __clz(n[k])], 1u);
    {  // empty tiles are the bulk (80+ %): count them once per wave, not once per tile
      uint32_t empties = 0;
#pragma unroll
      for (int k = 0; k < SCAN_ITEMS; ++k) empties += (t0 + k < v.T && n[k] == 0) ? 1u : 0u;
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) empties += (uint32_t)__shfl_xor((int)empties, d, 64);
      if ((tid & 63) == 0 && empties) atomicAdd(&cls_hist[0], empties);
    }
    if (mx) atomicMax(&max_n_s, mx);
    __syncthreads();
  }
  if (tid == 0 && role != 2) {
    const uint32_t R = carry3[0];            // sum of the tile counts = sum of tiles_touched
    L.tile_start[v.T] = carry3[0];
    L.tile_bstart[v.T] = carry3[1];
    L.tile_wgstart[v.T] = carry3[2];
    L.tile_msegstart[v.T] = carry3[3];
    hgs_status st;
    st.num_rendered = R;
    st.active_tiles = (uint32_t)v.T - cls_hist[0];
    st.num_buckets = carry3[1];
    st.bwd_groups = carry3[2];
    st.overflow = (R > v.entry_capacity) ? 1u : 0u;
    if (v.max_tile_hint > 0 && max_n_s > (uint32_t)v.max_tile_hint) st.overflow |= 2u;
    st.reserved[0] = v.entry_capacity;   // carve key for hgs_backward
    st.reserved[1] = max_n_s;            // longest tile list
    st.reserved[2] = 0;
    *status = st;
    if (status_host) {            // pinned, device-mapped host memory: no in-stream copy
      *status_host = st;
      __threadfence_system();
    }
  }
  if (role == 0) return;
  if (tid == 0) {   // heavy classes first; class 0 (empty tiles) last
    uint32_t acc = 0;
    for (int c = 32; c >= 0; --c) { cls_base[c] = acc; acc += cls_hist[c]; }
  }
  __syncthreads();
  // (c) tile_order: a permutation of all tiles, heavy first (order inside a class is free).
  // Non-empty tiles take a slot with one LDS atomic each; the empty class is handed out
  // per wave (ballot + prefix popcount) - thousands of same-address atomics otherwise.
  for (int base = 0; base < v.T; base += SCAN_NT) {
    const int t = base + tid;
    uint32_t n = 0;
    if (t < v.T) {
      if (lds_tiles) {
        n = s_n[t];
      } else if (role == 2) {     // tile_start belongs to another workgroup of this launch
#pragma unroll
        for (int rg = 0; rg < HGS_ROW_GROUPS; ++rg) n += L.tile_grp[(size_t)rg * v.T + t];
      } else {
        n = L.tile_start[t + 1] - L.tile_start[t];
      }
    }
    const bool empty = (t < v.T) && (n == 0);
    const unsigned long long ball = __ballot(empty);
    uint32_t wbase = 0;
    if ((tid & 63) == 0 && ball) wbase = atomicAdd(&cls_base[0], (uint32_t)__popcll(ball));
    wbase = (uint32_t)__shfl((int)wbase, 0, 64);
    if (t < v.T) {
      uint32_t pos;
      if (empty)
        pos = wbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32),
                                                __builtin_amdgcn_mbcnt_lo((uint32_t)ball, 0u));
      else
        pos = atomicAdd(&cls_base[32 - __clz(n)], 1u);
      L.tile_order[pos] = (uint32_t)t;
      L.tile_pos[t] = pos;
      L.pos_wgstart[pos] = (n + HGS_BUCKET - 1) / HGS_BUCKET;      // buckets of the tile at this position
    }
  }
  // Backward work items in tile_order too (heavy tiles first, a tile's buckets consecutive):
  // pos_wgstart = exclusive prefix of the bucket counts over positions.  With ~5.6k items on
  // 4096 wave slots the items that start late must be the light ones.
  __syncthreads();
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < v.T; base += SCAN_NT * 4) {
    const int p0 = base + tid * 4;
    uint32_t nb[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (p0 + k < v.T) nb[k] = L.pos_wgstart[p0 + k];
    uint32_t total;
    const uint32_t ex = hgs_block_excl_scan<SCAN_NT>(nb[0] + nb[1] + nb[2] + nb[3], wtot, total);
    uint32_t run = carry_s + ex;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (p0 + k < v.T) { L.pos_wgstart[p0 + k] = run; run += nb[k]; }
    __syncthreads();
    if (tid == 0) carry_s += total;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------- 2. fill
// LDS path, part 1: grid (T/256, HGS_ROW_GROUPS).  Thread (t, rg) scans the rows of row group
// rg of histogram column t in place (hist[g][t] -> entries of tile t owned by earlier
// workgroups OF THE SAME GROUP) and leaves the group's total in tile_grp[rg][t].
extern "C" __global__ void __launch_bounds__(256)
hgs_k_colscan(View v, Layout L) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= v.T) return;
  const int rg = blockIdx.y;
  const int rpg = (v.nwg + HGS_ROW_GROUPS - 1) / HGS_ROW_GROUPS;
  const int g0 = rg * rpg, g1 = min(v.nwg, g0 + rpg);
  uint32_t run = 0;
  uint32_t* col = L.hist + t;
  constexpr int B = 16;
  for (int g = g0; g < g1; g += B) {
    uint32_t c[B];
#pragma unroll
    for (int k = 0; k < B; ++k) c[k] = (g + k < g1) ? col[(size_t)(g + k) * v.T] : 0u;
#pragma unroll
    for (int k = 0; k < B; ++k) {
      if (g + k < g1) col[(size_t)(g + k) * v.T] = run;
      run += c[k];
    }
  }
  L.tile_grp[(size_t)rg * v.T + t] = run;
}

// LDS path, part 2: same workgroup -> chunk ownership as hgs_k_preprocess_fwd.  Slot of an
// entry = tile_start[t] + hist[g][t] (entries of earlier workgroups) + an LDS cursor.
extern "C" __global__ void __launch_bounds__(HGS_BLOCK)
hgs_k_fill(View v, Layout L, const hgs_status* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_cur[];
  __shared__ uint32_t wtot[HGS_BLOCK / 64];
  if (status->overflow) return;
  const uint32_t* __restrict__ base_row = L.hist + (size_t)blockIdx.x * v.T;
  const int rpg = (v.nwg + HGS_ROW_GROUPS - 1) / HGS_ROW_GROUPS;
  const uint32_t* __restrict__ grp_row = L.tile_gbase + (size_t)(blockIdx.x / rpg) * v.T;
  for (int t = threadIdx.x; t < v.T; t += HGS_BLOCK) lds_cur[t] = grp_row[t] + base_row[t];
  __syncthreads();
  for (int c = 0; c < v.cpw; ++c) {
    const int chunk = blockIdx.x * v.cpw + c;
    if (chunk >= v.nblk) break;
    const int i = chunk * HGS_BLOCK + threadIdx.x;
    uint32_t lo = 0, hi = 0, depth_bits = 0;
    if (i < v.P) {
      const uint4 q2 = reinterpret_cast<const uint4*>(&L.geom[i])[2];   // b, depth, rect_lo, rect_hi
      depth_bits = q2.y; lo = q2.z; hi = q2.w;
    }
    const int minx = lo & 0xffffu, miny = lo >> 16, maxx = hi & 0xffffu, maxy = hi >> 16;
    const uint32_t tt = (uint32_t)((maxx - minx) * (maxy - miny));
    uint32_t total;
    const uint32_t ex = hgs_block_excl_scan<HGS_BLOCK>(tt, wtot, total);
    if (i < v.P) {
      L.geom[i].offset = L.block_base[chunk] + ex;
      const unsigned long long key_hi = (unsigned long long)depth_bits << 32;
      for (int ty = miny; ty < maxy; ++ty)
        for (int tx = minx; tx < maxx; ++tx) {
          const uint32_t slot = atomicAdd(&lds_cur[ty * v.grid_x + tx], 1u);
          L.keys[slot] = key_hi | (uint32_t)i;
        }
    }
  }
}

// Fallback (global atomics) for T*4 > 64 KB.
extern "C" __global__ void __launch_bounds__(HGS_BLOCK)
hgs_k_fill_ga(View v, Layout L, const hgs_status* __restrict__ status) {
  __shared__ uint32_t wtot[HGS_BLOCK / 64];
  if (status->overflow) return;
  const int i = blockIdx.x * HGS_BLOCK + threadIdx.x;
  uint32_t lo = 0, hi = 0, depth_bits = 0;
  if (i < v.P) {
    const uint4 q2 = reinterpret_cast<const uint4*>(&L.geom[i])[2];   // b, depth, rect_lo, rect_hi
    depth_bits = q2.y; lo = q2.z; hi = q2.w;
  }
  const int minx = lo & 0xffffu, miny = lo >> 16, maxx = hi & 0xffffu, maxy = hi >> 16;
  const uint32_t tt = (uint32_t)((maxx - minx) * (maxy - miny));
  uint32_t total;
  const uint32_t ex = hgs_block_excl_scan<HGS_BLOCK>(tt, wtot, total);
  if (i >= v.P) return;
  L.geom[i].offset = L.block_base[blockIdx.x] + ex;
  if (tt == 0) return;
  const unsigned long long key_hi = (unsigned long long)depth_bits << 32;
  for (int ty = miny; ty < maxy; ++ty)
    for (int tx = minx; tx < maxx; ++tx) {
      const int t = ty * v.grid_x + tx;
      const uint32_t slot = L.tile_start[t] + atomicAdd(&L.tile_count[t], 1u);
      L.keys[slot] = key_hi | (uint32_t)i;
    }
}

// ---------------------------------------------------------------------------- 3. sort
namespace {

__device__ __forceinline__ void gather_records(const View& v, const Layout& L, int t,
                                               uint32_t start, uint32_t n,
                                               const unsigned long long* sorted, int nt) {
  const int tx = t % v.grid_x, ty = t / v.grid_x;
  const float x0 = (float)(tx * HGS_TILE), y0 = (float)(ty * HGS_TILE);
  {  // (tile, segment) table of the long lists: the segment kernels find their work with one load
    const uint32_t nseg = hgs_nseg(n);
    if (nseg > 1) {
      const uint32_t ms0 = L.tile_msegstart[t], item_bound = 2u * (uint32_t)(v.entry_capacity / HGS_SEG) + 2u;
      for (uint32_t sg = threadIdx.x; sg < nseg && ms0 + sg < item_bound; sg += nt)
        L.seg_item[ms0 + sg] = make_uint2((uint32_t)t, sg);
    }
  }
  // GU records per thread in flight: the 64 B geom gathers are dependent random reads (~1-2 us
  // each); issued one at a time they dominated the sort kernel of the heaviest tile
  constexpr int GU = 4;
  for (uint32_t kb = threadIdx.x; kb < n; kb += (uint32_t)nt * GU) {
    uint32_t idxv[GU];
    uint4 q0[GU], q1[GU], q2[GU], q3[GU];
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const uint32_t k = kb + (uint32_t)u * nt;
      idxv[u] = (k < n) ? (uint32_t)sorted[k] : 0u;
    }
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const uint32_t k = kb + (uint32_t)u * nt;
      if (k < n) {
        const uint4* gp = reinterpret_cast<const uint4*>(&L.geom[idxv[u]]);
        q0[u] = gp[0]; q1[u] = gp[1]; q2[u] = gp[2]; q3[u] = gp[3];
      }
    }
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const uint32_t k = kb + (uint32_t)u * nt;
      if (k >= n) continue;
      const uint32_t idx = idxv[u];
      const uint4 g0 = q0[u], g1 = q1[u], g2 = q2[u], g3 = q3[u];
      // g0: mx my ca cb | g1: cc op r g | g2: b depth rect_lo rect_hi | g3: offset radius ..
      const int minx = g2.z & 0xffffu, miny = g2.z >> 16, maxx = g2.w & 0xffffu;
      const uint32_t entry = g3.x + (uint32_t)((ty - miny) * (maxx - minx) + (tx - minx));
      const float mx = __uint_as_float(g0.x), my = __uint_as_float(g0.y);
      const float ca = __uint_as_float(g0.z), cb = __uint_as_float(g0.w), cc = __uint_as_float(g1.x);
      const float op = __uint_as_float(g1.y);
      // Conservative quadrant cull: a pixel can only pass alpha >= 1/255 inside the ellipse
      // q(d) = ca dx^2 + 2 cb dx dy + cc dy^2 <= tau, tau = 2 ln(255 op).  A quadrant (its 8x8 pixel
      // centres span a rectangle) is kept iff the minimum of q over that rectangle is <= tau: the
      // minimum of a convex quadratic over a box is 0 if the centre is inside, else it lies on an
      // edge, where the free coordinate's optimum is the clamped 1-D minimiser.  (This exact test
      // keeps 9 % fewer (entry, quadrant) pairs than the ellipse's bounding box on the 100k-Gaussian
      // scene - tools/cull_stats.py - and every pair it drops has no live pixel.)  Margins absorb
      // rounding; a set bit never changes results, a cleared bit must be provably empty.
      uint32_t mask = 0;
      const float a255 = 255.0f * op;
      if (a255 >= 0.999f) {
        const float tau = 2.0f * __logf(fmaxf(a255, 1.0f)) * 1.001f + 0.02f;
        const float detc = ca * cc - cb * cb;
        if (detc > 0.0f && cc > 0.0f && ca > 0.0f) {
          const float bc = cb / cc, ba = cb / ca;
          auto qf = [&](float px, float py) {
            const float dx = px - mx, dy = py - my;
            return ca * dx * dx + 2.0f * cb * dx * dy + cc * dy * dy;
          };
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float qx0 = x0 + (float)((q & 1) * 8), qy0 = y0 + (float)((q >> 1) * 8);
            const float qx1 = qx0 + 7.0f, qy1 = qy0 + 7.0f;
            float best = qf(fminf(fmaxf(mx, qx0), qx1), fminf(fmaxf(my, qy0), qy1));   // 0 when inside
            best = fminf(best, qf(qx0, fminf(fmaxf(my - bc * (qx0 - mx), qy0), qy1)));
            best = fminf(best, qf(qx1, fminf(fmaxf(my - bc * (qx1 - mx), qy0), qy1)));
            best = fminf(best, qf(fminf(fmaxf(mx - ba * (qy0 - my), qx0), qx1), qy0));
            best = fminf(best, qf(fminf(fmaxf(mx - ba * (qy1 - my), qx0), qx1), qy1));
            const bool hit = best <= tau * 1.0005f + 1e-3f * best;
            mask |= hit ? (1u << q) : 0u;
          }
        } else {
          mask = 0xfu;   // degenerate conic: never cull
        }
      }
      uint4* dst = reinterpret_cast<uint4*>(&L.recs[start + k]);
      const float qa = -0.5f * ca * HGS_LOG2E, qb = -cb * HGS_LOG2E, qc = -0.5f * cc * HGS_LOG2E;
      dst[0] = make_uint4(g0.x, g0.y, __float_as_uint(qa), __float_as_uint(qb));
      dst[1] = make_uint4(__float_as_uint(qc), g1.y, g1.z, g1.w);
      dst[2] = make_uint4(g2.x, g2.y, entry, (idx & 0x0fffffffu) | (mask << 28));
    }
  }
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

// ---- register / wave-shuffle / LDS hybrid of the same network -------------------------
// Thread t owns E consecutive keys (indices t*E .. t*E+E-1) in REGISTERS.  A comparator
// of stride < E stays inside the thread; stride < 64*E pairs lanes of one wave and goes
// through ds_bpermute (no barrier); only strides >= 64*E (a handful of the ~50-80 stages)
// exchange through LDS with barriers.  Padding is explicit (+inf keys).
typedef unsigned long long u64;

__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int mask) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, mask, 64);
  const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), mask, 64);
  return ((u64)hi << 32) | lo;
}

template <int E>
__device__ __forceinline__ void reg_stage_xor(u64 (&k)[E], int j) {      // j < E, power of 2
#pragma unroll
  for (int e = 0; e < E; ++e) {
    if ((e & j) == 0) {
      const u64 a = k[e], c = k[e | j];
      k[e] = a < c ? a : c;
      k[e | j] = a < c ? c : a;
    }
  }
}

template <int E>
__device__ __forceinline__ void reg_stage_mirror(u64 (&k)[E], int kk) {   // kk <= E
#pragma unroll
  for (int e = 0; e < E; ++e) {
    if ((e & (kk >> 1)) == 0) {
      const int o = e ^ (kk - 1);
      const u64 a = k[e], c = k[o];
      k[e] = a < c ? a : c;
      k[o] = a < c ? c : a;
    }
  }
}

// one stage whose partner lives in another lane of the same wave
template <int E>
__device__ __forceinline__ void lane_stage(u64 (&k)[E], int lane_mask, bool mirror, bool keep_min) {
  u64 o[E];
#pragma unroll
  for (int e = 0; e < E; ++e) o[e] = shfl_xor_u64(mirror ? k[E - 1 - e] : k[e], lane_mask);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const u64 a = k[e], c = o[e];
    const u64 mn = a < c ? a : c, mx = a < c ? c : a;
    k[e] = keep_min ? mn : mx;
  }
}

// one stage whose partner lives in another wave: through LDS
template <int E>
__device__ __forceinline__ void lds_stage(u64 (&k)[E], u64* lds, uint32_t base, uint32_t xmask,
                                          bool keep_min) {
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; ++e) lds[base + e] = k[e];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const u64 a = k[e], c = lds[(base + e) ^ xmask];
    const u64 mn = a < c ? a : c, mx = a < c ? c : a;
    k[e] = keep_min ? mn : mx;
  }
}

// one LDS stage for a wave that holds only padding: it must keep the barrier count of the others
__device__ __forceinline__ void lds_stage_idle() {
  __syncthreads();
  __syncthreads();
}

template <int E, int NT>
__device__ __forceinline__ void hybrid_sort(u64 (&k)[E], u64* lds, uint32_t npad) {
  const uint32_t base = threadIdx.x * E;
  // Waves whose key slots all lie beyond npad hold +inf padding only; no comparator ever changes
  // them, so they skip the network and just keep the barrier count.  (Device timestamps showed
  // the kernel issue-bound with every tile running all of its 8 waves through every stage: 762
  // tiles, most of them far shorter than 512 * E keys.)  Wave-uniform: npad and 64 E are powers
  // of two.
  const bool active = base < npad;
  for (uint32_t kk = 2; kk <= npad; kk <<= 1) {
    // --- mirror stage of span kk: index i pairs with i ^ (kk-1)
    if (kk <= (uint32_t)E) {
      if (active) {
        if (kk == 2) reg_stage_mirror<E>(k, 2);
        else if (kk == 4) { if (E >= 4) reg_stage_mirror<E>(k, 4); }
        else if (kk == 8) { if (E >= 8) reg_stage_mirror<E>(k, 8); }
        else if (kk == 16) { if (E >= 16) reg_stage_mirror<E>(k, 16); }
      }
    } else {
      const bool keep_min = (base & (kk >> 1)) == 0;
      if (kk <= 64u * E) { if (active) lane_stage<E>(k, (int)(kk / E - 1), true, keep_min); }
      else if (active) lds_stage<E>(k, lds, base, kk - 1, keep_min);
      else lds_stage_idle();
    }
    // --- xor stages j = kk/4 .. 1
    for (uint32_t j = kk >> 2; j > 0; j >>= 1) {
      if (j < (uint32_t)E) {
        if (active) {
          if (j == 1) reg_stage_xor<E>(k, 1);
          else if (j == 2) { if (E > 2) reg_stage_xor<E>(k, 2); }
          else if (j == 4) { if (E > 4) reg_stage_xor<E>(k, 4); }
          else if (j == 8) { if (E > 8) reg_stage_xor<E>(k, 8); }
        }
      } else {
        const bool keep_min = (base & j) == 0;
        if (j < 64u * E) { if (active) lane_stage<E>(k, (int)(j / E), false, keep_min); }
        else if (active) lds_stage<E>(k, lds, base, j, keep_min);
        else lds_stage_idle();
      }
    }
  }
}

// One tile, n <= NT*E keys: load, sort in registers/LDS, gather the records.
template <int E, int NT>
__device__ __forceinline__ void sort_one_tile(const View& v, const Layout& L, int t,
                                              uint32_t start, uint32_t n, u64* keys) {
  uint32_t npad = E;
  while (npad < n) npad <<= 1;
  u64 k[E];
  const uint32_t base = threadIdx.x * E;
#ifdef HGS_SORT_TIMING
  unsigned long long tmk[4];
  tmk[0] = __builtin_readcyclecounter();
#endif
#pragma unroll
  for (int e = 0; e < E; ++e) k[e] = (base + e < n) ? L.keys[start + base + e] : ~0ull;
#ifdef HGS_SORT_TIMING
  asm volatile("s_waitcnt vmcnt(0)");
  tmk[1] = __builtin_readcyclecounter();
#endif
  hybrid_sort<E, NT>(k, keys, npad);
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; ++e) keys[base + e] = k[e];
  __syncthreads();
#ifdef HGS_SORT_TIMING
  tmk[2] = __builtin_readcyclecounter();
#endif
  gather_records(v, L, t, start, n, keys, NT);
#ifdef HGS_SORT_TIMING
  __syncthreads();
  tmk[3] = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(L.segT) + (size_t)blockIdx.x * 8;
    o[0] = tmk[0]; o[1] = tmk[1]; o[2] = tmk[2]; o[3] = tmk[3]; o[4] = n; o[5] = E;
  }
#endif
}

}  // namespace

// All tiles with 1..4096 entries in ONE launch (512 threads; 2, 4 or 8 keys per thread by
// list length): each sort is latency-bound on its own stage chain, so the few long lists
// overlap with the many short ones instead of running in a second kernel after them.
#ifndef HGS_SORT_NT
#define HGS_SORT_NT 512
#endif
extern "C" __global__ void __launch_bounds__(HGS_SORT_NT)
hgs_k_sort_lds(View v, Layout L, const hgs_status* __restrict__ status) {
  __shared__ unsigned long long keys[4096];
  if (status->overflow) return;
  const uint32_t b = blockIdx.x;
  if (b >= status->active_tiles) return;
  const int t = (int)L.tile_order[b];
  const uint32_t start = L.tile_start[t];
  const uint32_t n = L.tile_start[t + 1] - start;
  if (n == 0 || n > 4096u) return;
  constexpr int E0 = 1024 / HGS_SORT_NT;
  if (n <= 1024u) sort_one_tile<E0, HGS_SORT_NT>(v, L, t, start, n, keys);
  else if (n <= 2048u) sort_one_tile<2 * E0, HGS_SORT_NT>(v, L, t, start, n, keys);
  else sort_one_tile<4 * E0, HGS_SORT_NT>(v, L, t, start, n, keys);
}

extern "C" __global__ void __launch_bounds__(1024)
hgs_k_sort_large(View v, Layout L, const hgs_status* __restrict__ status) {
  __shared__ unsigned long long keys[16384];
  if (status->overflow) return;
  const uint32_t b = blockIdx.x;
  if (b >= status->active_tiles) return;
  const int t = (int)L.tile_order[b];
  const uint32_t start = L.tile_start[t];
  const uint32_t n = L.tile_start[t + 1] - start;
  if (n <= 4096u || n > 16384u) return;
  // long lists: the plain LDS network with all 1024 threads on 2 comparators per stage
  // beats 16 keys per thread in registers (measured at 500k Gaussians: 181 vs 220 us) and ties
  // with the register/shuffle hybrid at 8 keys x 1024 threads (153 vs 156 us)
  for (uint32_t k = threadIdx.x; k < n; k += 1024) keys[k] = L.keys[start + k];
  __syncthreads();
  bitonic_sort<1024>(keys, n);
  gather_records(v, L, t, start, n, keys, 1024);
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
  const uint32_t n = L.tile_start[t + 1] - start;
  if (n <= 16384u) return;
  bitonic_sort<1024>(L.keys + start, n);
  gather_records(v, L, t, start, n, L.keys + start, 1024);
}
