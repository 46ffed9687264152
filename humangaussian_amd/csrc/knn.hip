// knn.hip - mean squared distance to the 3 nearest neighbours of every point: the
// `simple_knn._C.distCUDA2` the reference calls when it creates a cloud from a point set
// (/root/reference/gaussiansplatting/scene/gaussian_model.py:134, gs_renderer.py:386-389;
// semantics in submodules/simple-knn/simple_knn.cu:147-183: neighbours exclude the point's own
// INDEX, duplicates at distance 0 count, result = (d0 + d1 + d2) / 3 with d0 <= d1 <= d2).
//
// Upstream sorts by Morton code and prunes 1024-point boxes.  This runs once per cloud, not per
// step, so the MI355X version is the exact, sort-free form: every workgroup owns 256 query points
// (2 per lane would only help above ~1M points) and streams the whole set through LDS in 1024-point
// float4 tiles; a lane keeps its three best distances in registers with a branch-free insert.
// Cost: P^2 pairs x ~14 VALU instructions: 2.5 ms at 100k points, 60 ms at 500k.
// Roofline: VALU; HBM traffic P * (P / 256) * 16 B reads, all L2 hits after the first pass.
#include "hgs_common.h"

#define HGS_KNN_TILE 1024

extern "C" __global__ void __launch_bounds__(256)
hgs_k_knn3(int P, const float* __restrict__ pts, float* __restrict__ out) {
  __shared__ float4 tile[HGS_KNN_TILE];
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
