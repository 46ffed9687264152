// render_fwd.hip - per-tile front-to-back alpha blending (stage F6).
//
// One 256-thread workgroup (4 wave64) per 16x16 tile; wave w owns the 8x8 pixel block
// (w&1, w>>1).  The tile's depth-sorted 48-byte records are streamed from HBM with
// coalesced 16 B/lane loads into LDS in batches of 256 and broadcast-read by every lane.
// Workgroups are issued heavy-tile-first (tile_order) so the long lists start early.
// When the call needs a backward, the running per-pixel state (T, C, D, W) is stored at
// every 64-entry bucket boundary: the backward kernel is parallel over buckets.
//
// Semantics follow upstream's renderCUDA of the ashawkey fork exactly (skip rules, the
// terminating Gaussian is not blended, depth not normalised, out_alpha = sum of weights).
// Roofline: VALU/LDS-latency bound (about 25 flop per pixel-Gaussian pair); HBM traffic is
// 48 B/entry in + 24 B/pixel out (+ 24 B/pixel/bucket state when storing).
#include "hgs_common.h"

template <bool STORE>
__device__ __forceinline__ void render_fwd_body(const View& v, const Layout& L,
                                                const hgs_status* status,
                                                float* __restrict__ out_color,
                                                float* __restrict__ out_depth,
                                                float* __restrict__ out_alpha) {
  __shared__ float4 batch[3 * 256];        // 12 KB: records as 3 x float4
  __shared__ uint32_t max_contrib_s;

  const bool overflow = status->overflow != 0;
  const int t = overflow ? (int)blockIdx.x : (int)L.tile_order[blockIdx.x];
  const int tid = threadIdx.x;
  const int tile_x = t % v.grid_x, tile_y = t / v.grid_x;
  int lx, ly;
  hgs_fwd_thread_pixel(tid, lx, ly);
  const int px = tile_x * HGS_TILE + lx, py = tile_y * HGS_TILE + ly;
  const bool inside = (px < v.W) && (py < v.H);
  const float pxf = (float)px, pyf = (float)py;

  const uint32_t start = overflow ? 0u : L.tile_start[t];
  const uint32_t n = overflow ? 0u : (L.tile_start[t + 1] - start);
  const uint32_t bstart = overflow ? 0u : L.tile_bstart[t];

  if (tid == 0) max_contrib_s = 0;

  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, Wt = 0.f;
  uint32_t last = 0;
  bool done = !inside;

  for (uint32_t base = 0; base < n; base += 256) {
    if (__syncthreads_and(done)) break;
    const uint32_t m = min(256u, n - base);
    if ((uint32_t)tid < m) {
      const float4* src = reinterpret_cast<const float4*>(&L.recs[start + base + tid]);
      batch[3 * tid + 0] = src[0];
      batch[3 * tid + 1] = src[1];
      batch[3 * tid + 2] = src[2];
    }
    __syncthreads();
    for (uint32_t jb = 0; jb < m; jb += HGS_BUCKET) {
      if (STORE && (base + jb) > 0) {
        float* bs = L.bstate + (size_t)(bstart + (base + jb) / HGS_BUCKET - 1) * HGS_BSTATE_FLOATS;
        bs[0 * 256 + tid] = T;
        bs[1 * 256 + tid] = C0;
        bs[2 * 256 + tid] = C1;
        bs[3 * 256 + tid] = C2;
        bs[4 * 256 + tid] = D;
        bs[5 * 256 + tid] = Wt;
      }
      const uint32_t je = min(m, jb + HGS_BUCKET);
      for (uint32_t j = jb; j < je; ++j) {
        const float4 r0 = batch[3 * j + 0];   // mx my ca cb
        const float4 r1 = batch[3 * j + 1];   // cc op r g
        const float4 r2 = batch[3 * j + 2];   // b depth entry idx
        float G, alpha;
        const bool keep = hgs_eval_alpha(r0.x - pxf, r0.y - pyf, r0.z, r0.w, r1.x, r1.y, G, alpha);
        if (done || !keep) continue;
        const float test_T = T * (1.0f - alpha);
        if (test_T < HGS_T_EPS) { done = true; continue; }
        const float wgt = alpha * T;
        C0 += r1.z * wgt;
        C1 += r1.w * wgt;
        C2 += r2.x * wgt;
        D += r2.y * wgt;
        Wt += wgt;
        T = test_T;
        last = base + j + 1;
      }
    }
  }

  if (inside) {
    const size_t pix = (size_t)py * v.W + px;
    const size_t HW = (size_t)v.H * v.W;
    out_color[0 * HW + pix] = C0 + T * v.bg[0];
    out_color[1 * HW + pix] = C1 + T * v.bg[1];
    out_color[2 * HW + pix] = C2 + T * v.bg[2];
    out_depth[pix] = D;
    out_alpha[pix] = Wt;
    L.n_contrib[pix] = last;
  }
  if (STORE && !overflow) {
    // tile-wide max of n_contrib: buckets at or beyond it are skipped by the backward
    uint32_t mx = last;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    __syncthreads();
    if ((tid & 63) == 0) atomicMax(&max_contrib_s, mx);
    __syncthreads();
    if (tid == 0) L.tile_maxcontrib[t] = max_contrib_s;
  }
}

extern "C" __global__ void __launch_bounds__(256)
hgs_k_render_fwd_store(View v, Layout L, const hgs_status* __restrict__ status,
                       float* __restrict__ out_color, float* __restrict__ out_depth,
                       float* __restrict__ out_alpha) {
  render_fwd_body<true>(v, L, status, out_color, out_depth, out_alpha);
}

extern "C" __global__ void __launch_bounds__(256)
hgs_k_render_fwd_nostore(View v, Layout L, const hgs_status* __restrict__ status,
                         float* __restrict__ out_color, float* __restrict__ out_depth,
                         float* __restrict__ out_alpha) {
  render_fwd_body<false>(v, L, status, out_color, out_depth, out_alpha);
}
