// knn.hip - mean squared distance to the 3 nearest neighbours of every point: the
// `simple_knn._C.distCUDA2` the reference calls when it creates a cloud from a point set
// (/root/reference/gaussiansplatting/scene/gaussian_model.py:134, gs_renderer.py:386-389;
// semantics in submodules/simple-knn/simple_knn.cu:147-183: neighbours exclude the point's own
// INDEX, duplicates at distance 0 count, result = (d0 + d1 + d2) / 3 with d0 <= d1 <= d2).
//
// Upstream sorts the points by Morton code (a global radix sort) and prunes 1024-point boxes
// (simple_knn.cu:63-221).  Here, MI355X-first, without a global sort and without a host round trip:
//   1. bounding box: wave min / max (DPP), one ordered-integer atomic per workgroup and axis;
//   2. a UNIFORM GRID sized on the device from the box (cell edge = cbrt(2 V / P): ~2 points per cell of the box
//      volume; a surface cloud fills ~5 % of the cells with ~20-80 points each); counting sort of the points by
//      cell: count (one atomic per point), exclusive scan over the cells (three passes), scatter - the order INSIDE
//      a cell is arbitrary and does not influence any result (the three smallest distances of a SET);
//   3. search: thread = point in cell order (neighbouring threads visit the same cells: L1 / L2 hits), rings of
//      cells around its own until the third-best distance cannot be beaten from outside the block searched
//      (exact: the same three distances as the brute force).  Near-linear: ~200-700 candidates per point instead
//      of P (100k points: 2.5 ms -> tens of microseconds; 5 M points after densification: ~6 s -> milliseconds).
// Degenerate clouds (a cell with more than HGS_KNN_CELL_MAX points: thousands of duplicates, two far outliers that
// stretch the box around one dense cluster) take the exact brute-force kernel of rounds 1-4 instead - the choice is
// made on the device (no host wait), both kernels are always launched, one of them returns at once.
#include "hgs_common.h"

#define HGS_KNN_TILE 1024
#define HGS_KNN_CELL_MAX 4096          // more points than this in one cell: brute force
#define HGS_KNN_MAX_CELLS (1u << 22)

// scratch layout (hgs_knn_scratch_bytes): header | cell_of[P] | count[NC + 1] -> start | cursor[NC] | block sums | sorted[P] float4
struct KnnGrid {
  uint32_t bmin[3], bmax[3];           // ordered-integer images of the box (knn_key)
  uint32_t gx, gy, gz, ncells;         // grid
  float ox, oy, oz, inv_h, h;          // origin, 1 / cell edge, cell edge
  uint32_t max_count;                  // fullest cell
  uint32_t brute;                      // 1: degenerate -> brute force
  uint32_t pad;
};

// float <-> unsigned key with the same order (atomicMin / atomicMax on floats of any sign)
__device__ __forceinline__ uint32_t knn_key(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float knn_unkey(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

extern "C" __global__ void __launch_bounds__(256)
hgs_k_knn_bbox(int P, const float* __restrict__ pts, KnnGrid* __restrict__ G) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  uint32_t lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
  if (i < P) {
#pragma unroll
    for (int a = 0; a < 3; ++a) lo[a] = hi[a] = knn_key(pts[3 * (size_t)i + a]);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    hi[a] = hgs_wave_max_u32(hi[a]);
    lo[a] = ~hgs_wave_max_u32(~lo[a]);
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { atomicMin(&G->bmin[a], lo[a]); atomicMax(&G->bmax[a], hi[a]); }
  }
}

// one thread: the grid from the box.  nc_max = cells the scratch was sized for.
extern "C" __global__ void hgs_k_knn_grid_setup(int P, uint32_t nc_max, KnnGrid* __restrict__ G) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float lo[3], ext[3];
  for (int a = 0; a < 3; ++a) {
    lo[a] = knn_unkey(G->bmin[a]);
    ext[a] = knn_unkey(G->bmax[a]) - lo[a];
    if (!(ext[a] >= 0.0f) || !(ext[a] < 3.0e38f)) ext[a] = 0.0f;      // NaN / inf coordinates: that axis collapses
  }
  const float emax = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
  // flat axes (a planar or collinear cloud) get the thickness of one cell instead of 0
  float h = 1.0f;
  uint32_t g[3] = {1u, 1u, 1u};
  if (emax > 0.0f && P > 8) {
    const float floor_ext = emax * 1e-3f;
    const float vol = fmaxf(ext[0], floor_ext) * fmaxf(ext[1], floor_ext) * fmaxf(ext[2], floor_ext);
    h = cbrtf(2.0f * vol / (float)P);
    for (int it = 0; it < 64; ++it) {                    // grow the cells until the grid fits the scratch
      unsigned long long n = 1;
      for (int a = 0; a < 3; ++a) {
        const float c = floorf(ext[a] / h) + 1.0f;
        g[a] = c < 1.0f ? 1u : (c > 4096.0f ? 4096u : (uint32_t)c);
        n *= g[a];
      }
      if (n <= nc_max && ext[0] / h < 4095.0f && ext[1] / h < 4095.0f && ext[2] / h < 4095.0f) break;
      h *= 1.26f;
    }
  }
  G->gx = g[0]; G->gy = g[1]; G->gz = g[2];
  G->ncells = g[0] * g[1] * g[2];
  if (G->ncells > nc_max) { G->gx = G->gy = G->gz = 1u; G->ncells = 1u; }      // (cannot happen; stays in bounds if it does)
  G->ox = lo[0]; G->oy = lo[1]; G->oz = lo[2];
  G->h = h; G->inv_h = 1.0f / h;
  G->max_count = 0u; G->brute = 0u;
}

__device__ __forceinline__ void knn_cell_of(const KnnGrid& G, float x, float y, float z, int& cx, int& cy, int& cz) {
  // (non-finite coordinates land in cell 0 of their axis: fmaxf / fminf drop NaN)
  cx = (int)fminf(fmaxf(floorf((x - G.ox) * G.inv_h), 0.0f), (float)(G.gx - 1u));
  cy = (int)fminf(fmaxf(floorf((y - G.oy) * G.inv_h), 0.0f), (float)(G.gy - 1u));
  cz = (int)fminf(fmaxf(floorf((z - G.oz) * G.inv_h), 0.0f), (float)(G.gz - 1u));
}

extern "C" __global__ void __launch_bounds__(256)
hgs_k_knn_count(int P, const float* __restrict__ pts, const KnnGrid* __restrict__ Gp, uint32_t* __restrict__ cell_of,
                uint32_t* __restrict__ count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const KnnGrid G = *Gp;
  int cx, cy, cz;
  knn_cell_of(G, pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], cx, cy, cz);
  const uint32_t c = ((uint32_t)cz * G.gy + (uint32_t)cy) * G.gx + (uint32_t)cx;
  cell_of[i] = c;
  atomicAdd(&count[c], 1u);
}

// exclusive scan of count[0 .. ncells) in place (count[ncells] = P), three passes of 1024-cell blocks
extern "C" __global__ void __launch_bounds__(1024)
hgs_k_knn_scan1(const KnnGrid* __restrict__ Gp, const uint32_t* __restrict__ count, uint32_t* __restrict__ bsum) {
  __shared__ uint32_t wtot[16];
  const uint32_t n = Gp->ncells, i = blockIdx.x * 1024u + threadIdx.x;
  if (blockIdx.x * 1024u >= n) return;
  uint32_t tot;
  hgs_block_excl_scan<1024>(i < n ? count[i] : 0u, wtot, tot);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
extern "C" __global__ void __launch_bounds__(1024)
hgs_k_knn_scan2(KnnGrid* __restrict__ Gp, uint32_t* __restrict__ bsum) {
  __shared__ uint32_t wtot[16];
  const uint32_t nb = (Gp->ncells + 1023u) / 1024u;
  uint32_t carry = 0;
  for (uint32_t b0 = 0; b0 < nb; b0 += 1024u) {
    const uint32_t b = b0 + threadIdx.x;
    const uint32_t v = b < nb ? bsum[b] : 0u;
    uint32_t tot;
    const uint32_t ex = hgs_block_excl_scan<1024>(v, wtot, tot);
    if (b < nb) bsum[b] = carry + ex;
    carry += tot;
    __syncthreads();
  }
}
extern "C" __global__ void __launch_bounds__(1024)
hgs_k_knn_scan3(int P, KnnGrid* __restrict__ Gp, uint32_t* __restrict__ count, uint32_t* __restrict__ cursor,
                const uint32_t* __restrict__ bsum) {
  __shared__ uint32_t wtot[16];
  const uint32_t n = Gp->ncells, i = blockIdx.x * 1024u + threadIdx.x;
  if (blockIdx.x * 1024u >= n) return;
  const uint32_t v = i < n ? count[i] : 0u;
  uint32_t tot;
  const uint32_t ex = hgs_block_excl_scan<1024>(v, wtot, tot) + bsum[blockIdx.x];
  if (i < n) { count[i] = ex; cursor[i] = ex; }
  if (i == n - 1u) count[n] = (uint32_t)P;
  const uint32_t mx = hgs_wave_max_u32(v);
  if ((threadIdx.x & 63) == 0 && mx) {
    atomicMax(&Gp->max_count, mx);
    if (mx > (uint32_t)HGS_KNN_CELL_MAX) Gp->brute = 1u;
  }
}

extern "C" __global__ void __launch_bounds__(256)
hgs_k_knn_scatter(int P, const float* __restrict__ pts, const uint32_t* __restrict__ cell_of, uint32_t* __restrict__ cursor,
                  float4* __restrict__ sorted) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const uint32_t slot = atomicAdd(&cursor[cell_of[i]], 1u);
  sorted[slot] = make_float4(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], __uint_as_float((uint32_t)i));
}

extern "C" __global__ void __launch_bounds__(256)
hgs_k_knn_search(int P, const KnnGrid* __restrict__ Gp, const uint32_t* __restrict__ start, const float4* __restrict__ sorted,
                 float* __restrict__ out) {
  const KnnGrid G = *Gp;
  if (G.brute) return;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= P) return;
  const float4 me = sorted[t];
  const uint32_t self = __float_as_uint(me.w);
  int cx, cy, cz;
  knn_cell_of(G, me.x, me.y, me.z, cx, cy, cz);
  const float big = 3.402823466e+38f;
  float b0 = big, b1 = big, b2 = big;
  const int gx = (int)G.gx, gy = (int)G.gy, gz = (int)G.gz;
  const int rmax = max(gx, max(gy, gz));
  auto visit = [&](int x, int y, int z) {
    const uint32_t c = ((uint32_t)z * G.gy + (uint32_t)y) * G.gx + (uint32_t)x;
    const uint32_t s = start[c], e = start[c + 1u];
    for (uint32_t k = s; k < e; ++k) {
      const float4 q = sorted[k];
      const float dx = q.x - me.x, dy = q.y - me.y, dz = q.z - me.z;
      float d = dx * dx + dy * dy + dz * dz;
      d = (__float_as_uint(q.w) == self) ? big : d;          // a point is not its own neighbour (its INDEX)
      const float t0 = fminf(b0, d), c0 = fmaxf(b0, d);
      const float t1 = fminf(b1, c0), c1 = fmaxf(b1, c0);
      b0 = t0; b1 = t1; b2 = fminf(b2, c1);
    }
  };
  for (int r = 0; r <= rmax; ++r) {
    // the SHELL of cells at Chebyshev distance exactly r around (cx, cy, cz), clipped to the grid: rows on a y / z face of
    // the block are walked in full, of the other rows only the two end cells are new
    const int z0 = max(cz - r, 0), z1 = min(cz + r, gz - 1), y0 = max(cy - r, 0), y1 = min(cy + r, gy - 1);
    const int x0 = max(cx - r, 0), x1 = min(cx + r, gx - 1);
    for (int z = z0; z <= z1; ++z)
      for (int y = y0; y <= y1; ++y) {
        if (abs(z - cz) == r || abs(y - cy) == r) {
          for (int x = x0; x <= x1; ++x) visit(x, y, z);
        } else {
          if (cx - r >= 0) visit(cx - r, y, z);
          if (cx + r <= gx - 1) visit(cx + r, y, z);
        }
      }
    // every point NOT yet seen lies outside the block [c - r, c + r]^3: at least `reach` away (conservative by 1e-3 h: the
    // cell of a point is floor((x - o) / h) in fp32, up to 2.4e-4 cells from the ideal boundary on a 4096-cell axis)
    if (x0 == 0 && y0 == 0 && z0 == 0 && x1 == gx - 1 && y1 == gy - 1 && z1 == gz - 1) break;      // the whole grid
    // (distances to the block's faces in GRID-RELATIVE coordinates: me - origin is the subtraction the cell assignment
    //  itself makes; `origin + k h` in absolute coordinates rounds at ulp(|origin|) / 2, which for a cloud translated
    //  far from the world origin exceeds the 1e-3 h margin - ADVICE r5)
    const float rx = me.x - G.ox, ry = me.y - G.oy, rz = me.z - G.oz;
    float reach = big;
    if (cx - r > 0) reach = fminf(reach, rx - (float)(cx - r) * G.h);
    if (cx + r < gx - 1) reach = fminf(reach, (float)(cx + r + 1) * G.h - rx);
    if (cy - r > 0) reach = fminf(reach, ry - (float)(cy - r) * G.h);
    if (cy + r < gy - 1) reach = fminf(reach, (float)(cy + r + 1) * G.h - ry);
    if (cz - r > 0) reach = fminf(reach, rz - (float)(cz - r) * G.h);
    if (cz + r < gz - 1) reach = fminf(reach, (float)(cz + r + 1) * G.h - rz);
    reach = fmaxf(reach - 1e-3f * G.h, 0.0f);
    if (b2 <= reach * reach) break;
  }
  out[self] = ((b0 + b1) + b2) / 3.0f;
}

// the exact brute force (rounds 1-4): every workgroup owns 256 query points and streams the whole set through LDS in
// 1024-point float4 tiles; a lane keeps its three best distances in registers with a branch-free insert.  P^2 pairs x
// ~14 VALU instructions: 2.5 ms at 100k points.  `gate` (may be NULL): run only if gate->brute is set.
extern "C" __global__ void __launch_bounds__(256)
hgs_k_knn3(int P, const float* __restrict__ pts, float* __restrict__ out, const KnnGrid* __restrict__ gate) {
  __shared__ float4 tile[HGS_KNN_TILE];
  if (gate && !gate->brute) return;
  const int tid = threadIdx.x;
  const int i = blockIdx.x * 256 + tid;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (i < P) { px = pts[3 * (size_t)i]; py = pts[3 * (size_t)i + 1]; pz = pts[3 * (size_t)i + 2]; }
  const float big = 3.402823466e+38f;
  float b0 = big, b1 = big, b2 = big;
  for (int base = 0; base < P; base += HGS_KNN_TILE) {
#pragma unroll
    for (int k = tid; k < HGS_KNN_TILE; k += 256) {
      const int j = base + k;
      tile[k] = (j < P) ? make_float4(pts[3 * (size_t)j], pts[3 * (size_t)j + 1], pts[3 * (size_t)j + 2], 0.f)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int cnt = min(HGS_KNN_TILE, P - base);
    const int self = i - base;                      // position of this lane's own point in the tile, if any
#pragma unroll 4
    for (int k = 0; k < cnt; ++k) {
      const float4 q = tile[k];                     // LDS broadcast
      const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
      float d = dx * dx + dy * dy + dz * dz;
      d = (k == self) ? big : d;                    // a point is not its own neighbour
      const float t0 = fminf(b0, d), c0 = fmaxf(b0, d);
      const float t1 = fminf(b1, c0), c1 = fmaxf(b1, c0);
      b0 = t0; b1 = t1; b2 = fminf(b2, c1);
    }
    __syncthreads();
  }
  if (i < P) out[i] = ((b0 + b1) + b2) / 3.0f;
}

// ---------------------------------------------------------------------------------------------
// View-parallel reduction: after the one all-gather every rank holds all ranks'
// packs [world][P][F] (gradient columns + radii as the last column).  out[p][f] = sum over ranks in
// RANK ORDER (deterministic, identical on every rank) for f < F-1, max for the radii column.
// One pass over world * P * F * 4 bytes at HBM speed; replaces 2 * (world - 1) strided torch
// kernels launched from a Python loop (host-bound: ~25 us per rank).
extern "C" __global__ void __launch_bounds__(256)
hgs_k_reduce_view_packs(int world, long long n, int F, const float* __restrict__ gathered,
                        const float* __restrict__ acc_in, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const bool is_max = (int)(i % F) == F - 1;
  // one left-to-right chain: (running total of the earlier collectives of the step) + rank 0 + rank 1 + ...
  float acc = acc_in ? acc_in[i] : gathered[i];
  for (int r = acc_in ? 0 : 1; r < world; ++r) {
    const float x = gathered[(long long)r * n + i];
    acc = is_max ? fmaxf(acc, x) : acc + x;
  }
  out[i] = acc;
}

// The same reduction with the UNPACK fused in: element (p, f) of the rank-ordered sum / max goes straight into the tensor
// its column belongs to (means3D 3 | means2D 3 | sh 3M | opacity 1 | scales 3 | rotations 4 | radii 1 as int32) - the last
// pass of a view-parallel step (hgs_reduce_view_packs_unpack): no (P, F) intermediate, no slicing / rounding kernels behind it.
extern "C" __global__ void __launch_bounds__(256)
hgs_k_reduce_view_packs_unpack(int world, long long n, int F, int M, const float* __restrict__ gathered,
                               const float* __restrict__ acc_in, float* __restrict__ g_means3D,
                               float* __restrict__ g_means2D, float* __restrict__ g_sh, float* __restrict__ g_opac,
                               float* __restrict__ g_scales, float* __restrict__ g_rot, int32_t* __restrict__ radii) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long p = i / F;
  const int f = (int)(i - p * F);
  const bool is_max = f == F - 1;
  float acc = acc_in ? acc_in[i] : gathered[i];
  for (int r = acc_in ? 0 : 1; r < world; ++r) {
    const float x = gathered[(long long)r * n + i];
    acc = is_max ? fmaxf(acc, x) : acc + x;
  }
  if (f < 3) g_means3D[3 * p + f] = acc;
  else if (f < 6) g_means2D[3 * p + (f - 3)] = acc;
  else if (f < 6 + 3 * M) g_sh[p * 3 * M + (f - 6)] = acc;
  else if (f == 6 + 3 * M) g_opac[p] = acc;
  else if (f < 10 + 3 * M) g_scales[3 * p + (f - 7 - 3 * M)] = acc;
  else if (f < 14 + 3 * M) g_rot[4 * p + (f - 10 - 3 * M)] = acc;
  else radii[p] = (int32_t)rintf(acc);
}

// Packs one rank's per-Gaussian contribution [means3D 3 | means2D 3 | sh 3M | opacity 1 | scale 3 |
// rot 4 | radii 1] into the (P, F) fp32 tensor that travels through the all-gather (radii as exact
// fp32 integers).  One pass instead of seven reshapes/casts and a torch.cat.
extern "C" __global__ void __launch_bounds__(256)
hgs_k_pack_view_contribution(int P, int M, const float* __restrict__ g_means3D,
                             const float* __restrict__ g_means2D, const float* __restrict__ g_sh,
                             const float* __restrict__ g_opac, const float* __restrict__ g_scales,
                             const float* __restrict__ g_rot, const int32_t* __restrict__ radii,
                             float* __restrict__ out) {
  const int F = 3 + 3 + 3 * M + 1 + 3 + 4 + 1;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)P * F) return;
  const int p = (int)(i / F), f = (int)(i % F);
  float v;
  if (f < 3) v = g_means3D[3 * (size_t)p + f];
  else if (f < 6) v = g_means2D[3 * (size_t)p + (f - 3)];
  else if (f < 6 + 3 * M) v = g_sh[(size_t)p * 3 * M + (f - 6)];
  else if (f < 7 + 3 * M) v = g_opac[p];
  else if (f < 10 + 3 * M) v = g_scales[3 * (size_t)p + (f - 7 - 3 * M)];
  else if (f < 14 + 3 * M) v = g_rot[4 * (size_t)p + (f - 10 - 3 * M)];
  else v = (float)radii[p];
  out[i] = v;
}
