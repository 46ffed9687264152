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
}

// ---------------------------------------------------------------------------- 1. scan
extern "C" __global__ void __launch_bounds__(SCAN_NT)
hgs_k_scan(View v, Layout L, hgs_status* __restrict__ status) {
  __shared__ uint32_t wtot[SCAN_NT / 64];
  __shared__ uint32_t carry_s;
  __shared__ uint32_t cls_hist[33];
  __shared__ uint32_t cls_base[33];
  const int tid = threadIdx.x;

  // (a) exclusive scan of per-workgroup tiles_touched sums -> block_base
  if (tid == 0) carry_s = 0;
  __syncthreads();
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
  const uint32_t R = carry_s;
  __syncthreads();

  // (b) tile_start / bucket-state prefix / backward-workgroup prefix; class histogram
  if (tid < 33) cls_hist[tid] = 0;
  __shared__ uint32_t carry3[3];
  if (tid == 0) { carry3[0] = carry3[1] = carry3[2] = 0; }
  __syncthreads();
  for (int base = 0; base < v.T; base += SCAN_NT) {
    const int t = base + tid;
    const uint32_t n = (t < v.T) ? L.tile_count[t] : 0u;
    const uint32_t nb = (n + HGS_BUCKET - 1) / HGS_BUCKET;
    const uint32_t nbs = nb > 0 ? nb - 1 : 0;                          // stored bucket states
    const uint32_t nwg = (nb + HGS_BWD_WAVES - 1) / HGS_BWD_WAVES;     // backward workgroups
    uint32_t tot0, tot1, tot2;
    const uint32_t e0 = hgs_block_excl_scan<SCAN_NT>(n, wtot, tot0);
    const uint32_t e1 = hgs_block_excl_scan<SCAN_NT>(nbs, wtot, tot1);
    const uint32_t e2 = hgs_block_excl_scan<SCAN_NT>(nwg, wtot, tot2);
    const uint32_t c0 = carry3[0], c1 = carry3[1], c2 = carry3[2];
    if (t < v.T) {
      L.tile_start[t] = c0 + e0;
      L.tile_bstart[t] = c1 + e1;
      L.tile_wgstart[t] = c2 + e2;
      L.tile_count[t] = 0;            // becomes the fill cursor
      L.tile_maxcontrib[t] = 0;
      atomicAdd(&cls_hist[n ? 32 - __clz(n) : 0], 1u);
    }
    __syncthreads();
    if (tid == 0) { carry3[0] = c0 + tot0; carry3[1] = c1 + tot1; carry3[2] = c2 + tot2; }
    __syncthreads();
  }
  if (tid == 0) {
    L.tile_start[v.T] = carry3[0];
    L.tile_bstart[v.T] = carry3[1];
    L.tile_wgstart[v.T] = carry3[2];
    // heavy classes first; class 0 (empty tiles) last
    uint32_t acc = 0;
    for (int c = 32; c >= 0; --c) { cls_base[c] = acc; acc += cls_hist[c]; }
    hgs_status st;
    st.num_rendered = R;
    st.active_tiles = (uint32_t)v.T - cls_hist[0];
    st.num_buckets = carry3[1];
    st.bwd_groups = carry3[2];
    st.overflow = (R > v.entry_capacity) ? 1u : 0u;
    st.reserved[0] = v.entry_capacity;   // carve key for hgs_backward
    st.reserved[1] = st.reserved[2] = 0;
    *status = st;
  }
  __syncthreads();
  // (c) tile_order: a permutation of all tiles, heavy first (order inside a class is free)
  for (int base = 0; base < v.T; base += SCAN_NT) {
    const int t = base + tid;
    if (t < v.T) {
      const uint32_t n = L.tile_start[t + 1] - L.tile_start[t];
      const int c = n ? 32 - __clz(n) : 0;
      const uint32_t pos = atomicAdd(&cls_base[c], 1u);
      L.tile_order[pos] = (uint32_t)t;
    }
  }
}

// ---------------------------------------------------------------------------- 2. fill
// LDS path, part 1: per tile, exclusive scan of the per-workgroup histogram column
// (hist[g][t] -> number of entries of tile t owned by workgroups < g) and the tile total.
extern "C" __global__ void __launch_bounds__(256)
hgs_k_colscan(View v, Layout L) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= v.T) return;
  uint32_t run = 0;
  uint32_t* col = L.hist + t;
  int g = 0;
  for (; g + 4 <= v.nwg; g += 4) {
    const uint32_t c0 = col[(size_t)(g + 0) * v.T], c1 = col[(size_t)(g + 1) * v.T];
    const uint32_t c2 = col[(size_t)(g + 2) * v.T], c3 = col[(size_t)(g + 3) * v.T];
    col[(size_t)(g + 0) * v.T] = run; run += c0;
    col[(size_t)(g + 1) * v.T] = run; run += c1;
    col[(size_t)(g + 2) * v.T] = run; run += c2;
    col[(size_t)(g + 3) * v.T] = run; run += c3;
  }
  for (; g < v.nwg; ++g) {
    const uint32_t c = col[(size_t)g * v.T];
    col[(size_t)g * v.T] = run;
    run += c;
  }
  L.tile_count[t] = run;
}

// LDS path, part 2: same workgroup -> chunk ownership as hgs_k_preprocess_fwd.  Slot of an
// entry = tile_start[t] + hist[g][t] (entries of earlier workgroups) + an LDS cursor.
extern "C" __global__ void __launch_bounds__(HGS_BLOCK)
hgs_k_fill(View v, Layout L, const hgs_status* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_cur[];
  __shared__ uint32_t wtot[HGS_BLOCK / 64];
  if (status->overflow) return;
  const uint32_t* __restrict__ base_row = L.hist + (size_t)blockIdx.x * v.T;
  for (int t = threadIdx.x; t < v.T; t += HGS_BLOCK) lds_cur[t] = L.tile_start[t] + base_row[t];
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
  for (uint32_t k = threadIdx.x; k < n; k += nt) {
    const uint32_t idx = (uint32_t)sorted[k];
    const uint4* gp = reinterpret_cast<const uint4*>(&L.geom[idx]);
    const uint4 g0 = gp[0], g1 = gp[1], g2 = gp[2], g3 = gp[3];
    // g0: mx my ca cb | g1: cc op r g | g2: b depth rect_lo rect_hi | g3: offset radius ..
    const int minx = g2.z & 0xffffu, miny = g2.z >> 16, maxx = g2.w & 0xffffu;
    const uint32_t entry = g3.x + (uint32_t)((ty - miny) * (maxx - minx) + (tx - minx));
    const float mx = __uint_as_float(g0.x), my = __uint_as_float(g0.y);
    const float ca = __uint_as_float(g0.z), cb = __uint_as_float(g0.w), cc = __uint_as_float(g1.x);
    const float op = __uint_as_float(g1.y);
    // Conservative quadrant cull: a pixel can only pass alpha >= 1/255 inside the ellipse
    // d^T Sigma^-1 d <= tau, tau = 2 ln(255 op); its bounding box has half extents
    // sqrt(tau * Sigma_xx), sqrt(tau * Sigma_yy) with Sigma = conic^-1.  Margins absorb
    // rounding; a set bit never changes results, a cleared bit must be provably empty.
    uint32_t mask = 0;
    const float a255 = 255.0f * op;
    if (a255 >= 0.999f) {
      const float tau = 2.0f * __logf(fmaxf(a255, 1.0f)) * 1.001f + 0.01f;
      const float detc = ca * cc - cb * cb;
      if (detc > 0.0f && cc > 0.0f && ca > 0.0f) {
        const float ex = sqrtf(tau * cc / detc) * 1.001f + 0.01f;
        const float ey = sqrtf(tau * ca / detc) * 1.001f + 0.01f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float qx0 = x0 + (float)((q & 1) * 8), qy0 = y0 + (float)((q >> 1) * 8);
          const bool hit = (mx + ex >= qx0) && (mx - ex <= qx0 + 7.0f) && (my + ey >= qy0) &&
                           (my - ey <= qy0 + 7.0f);
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

// Bitonic network in its "all comparators ascending" form (first stage of every merge
// mirrors the upper half), which sorts any n <= npad correctly with VIRTUAL +inf padding:
// a comparator whose upper index is >= n is a no-op.  Keys are unique, so the result is
// the unique ascending order.  One __syncthreads() per stage.
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

// Tiles with LO < n <= CAP, keys sorted entirely in LDS.
template <int CAP, int NT, int LO>
__device__ __forceinline__ void sort_tiles_lds(const View& v, const Layout& L,
                                               const hgs_status* status,
                                               unsigned long long* keys) {
  if (status->overflow) return;
  const uint32_t b = blockIdx.x;
  if (b >= status->active_tiles) return;
  const int t = (int)L.tile_order[b];
  const uint32_t start = L.tile_start[t];
  const uint32_t n = L.tile_start[t + 1] - start;
  if (n <= (uint32_t)LO || n > (uint32_t)CAP) return;
  for (uint32_t k = threadIdx.x; k < n; k += NT) keys[k] = L.keys[start + k];
  __syncthreads();
  bitonic_sort<NT>(keys, n);
  gather_records(v, L, t, start, n, keys, NT);
}

}  // namespace

extern "C" __global__ void __launch_bounds__(256)
hgs_k_sort_small(View v, Layout L, const hgs_status* __restrict__ status) {
  __shared__ unsigned long long keys[1024];
  sort_tiles_lds<1024, 256, 0>(v, L, status, keys);
}
extern "C" __global__ void __launch_bounds__(512)
hgs_k_sort_medium(View v, Layout L, const hgs_status* __restrict__ status) {
  __shared__ unsigned long long keys[4096];
  sort_tiles_lds<4096, 512, 1024>(v, L, status, keys);
}
extern "C" __global__ void __launch_bounds__(1024)
hgs_k_sort_large(View v, Layout L, const hgs_status* __restrict__ status) {
  __shared__ unsigned long long keys[16384];
  sort_tiles_lds<16384, 1024, 4096>(v, L, status, keys);
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
