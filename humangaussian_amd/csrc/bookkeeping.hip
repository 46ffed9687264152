// bookkeeping.hip - the per-step bookkeeping either side of the rasterize hot path (SURVEY.md 8(f)-3,
// 8(f)-4), as single-pass HIP kernels.  All of it is HBM-bound elementwise / gather work: one
// coalesced pass over the per-Gaussian arrays, no atomics except the three mask counters.
//
// Reference code replaced (torch elementwise kernels launched from Python, a dozen per call):
//   densification statistics   threestudio/systems/GaussianDreamer.py:253-256,289,385-391 +
//                              gaussiansplatting/scene/gaussian_model.py:434-438
//   clone / split / prune masks gaussian_model.py:359-438 (densify_and_clone, densify_and_split,
//                              densify_and_prune, prune_only)
//   row compaction             gaussian_model.py:283-337 (_prune_optimizer: boolean-mask indexing of
//                              every parameter tensor and of both Adam moments)
//   re-anchoring               animation.py:384-403 (numpy on the CPU + H2D per frame)
#include "hgs_common.h"

// ---- densification statistics of one training step (B views) -------------------------------------
// Per Gaussian: radii_max = max over views, visible = radii_max > 0 (and keep[i] when given: the
// reference masks near-hand points), g = sum over views (view order) of dL/dmeans2D, and for
// visible Gaussians  max_radii2D = max(max_radii2D, radii_max),  accum += |g.xy|,  denom += 1.
extern "C" __global__ void __launch_bounds__(256)
hgs_k_densify_stats(int B, int P, const float* __restrict__ g2d, const int32_t* __restrict__ radii,
                    const uint8_t* __restrict__ keep, float* __restrict__ accum, float* __restrict__ denom,
                    float* __restrict__ max_radii2D, int32_t* __restrict__ radii_max_out,
                    uint8_t* __restrict__ visible_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  int rmax = 0;
  float gx = 0.f, gy = 0.f;
  for (int b = 0; b < B; ++b) {
    rmax = max(rmax, radii[(size_t)b * P + i]);
    const float* g = g2d + ((size_t)b * P + i) * 3;
    gx += g[0];
    gy += g[1];
  }
  const bool vis = rmax > 0 && (!keep || keep[i]);
  if (vis) {
    max_radii2D[i] = fmaxf(max_radii2D[i], (float)rmax);
    accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.0f;
  }
  if (radii_max_out) radii_max_out[i] = rmax;
  if (visible_out) visible_out[i] = vis ? 1 : 0;
}

// ---- clone / split / prune masks ---------------------------------------------------------------
// grad = accum / denom (NaN -> 0).  big = max_k scale_k > percent_dense * extent.
//   clone = grad >= thr && !big         (densify_and_clone)
//   split = grad >= thr &&  big         (densify_and_split)
//   prune = opacity < min_opacity || (max_screen_size > 0 && (max_radii2D > max_screen_size ||
//           max scale > 0.1 * extent)) || (size_thresh > 0 && max scale > size_thresh)   (densify_and_prune / prune_only)
// scales / opacity arrive raw (log-scale, logit) when the flags say so: the activations are fused.
// counts[0..2] += number of set clone / split / prune bits (the host sizes its tensors with them).
extern "C" __global__ void __launch_bounds__(256)
hgs_k_densify_masks(int P, const float* __restrict__ accum, const float* __restrict__ denom,
                    const float* __restrict__ scales, int scales_are_log, const float* __restrict__ opacity,
                    int opacity_is_logit, const float* __restrict__ max_radii2D, float grad_threshold,
                    float percent_dense, float extent, float min_opacity, float max_screen_size,
                    float size_thresh, uint8_t* __restrict__ clone, uint8_t* __restrict__ split,
                    uint8_t* __restrict__ prune, uint32_t* __restrict__ counts) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool c = false, s = false, p = false;
  if (i < P) {
    float g = accum[i] / denom[i];
    if (g != g) g = 0.0f;
    float s0 = scales[3 * i + 0], s1 = scales[3 * i + 1], s2 = scales[3 * i + 2];
    if (scales_are_log) { s0 = expf(s0); s1 = expf(s1); s2 = expf(s2); }
    const float smax = fmaxf(s0, fmaxf(s1, s2));
    float op = opacity[i];
    if (opacity_is_logit) op = 1.0f / (1.0f + expf(-op));
    const bool big = smax > percent_dense * extent;
    c = (g >= grad_threshold) && !big;
    s = (g >= grad_threshold) && big;
    p = op < min_opacity;
    if (max_screen_size > 0.0f) p = p || (max_radii2D[i] > max_screen_size) || (smax > 0.1f * extent);
    if (size_thresh > 0.0f) p = p || (smax > size_thresh);
    if (clone) clone[i] = c;
    if (split) split[i] = s;
    if (prune) prune[i] = p;
  }
  if (counts) {
    const unsigned long long bc = __ballot(c), bs = __ballot(s), bp = __ballot(p);
    if ((threadIdx.x & 63) == 0) {
      if (bc) atomicAdd(&counts[0], (uint32_t)__popcll(bc));
      if (bs) atomicAdd(&counts[1], (uint32_t)__popcll(bs));
      if (bp) atomicAdd(&counts[2], (uint32_t)__popcll(bp));
    }
  }
}

// ---- stable row compaction (prune every parameter tensor and both Adam moments) ---------------------
// pass 1: per 1024-row block popcount of keep[]; pass 2 (one workgroup): exclusive scan of the block
// counts; pass 3: rows of the kept indices move to their compacted position, order preserved.
extern "C" __global__ void __launch_bounds__(1024)
hgs_k_keep_count(int P, const uint8_t* __restrict__ keep, uint32_t* __restrict__ block_count) {
  __shared__ uint32_t wsum[16];
  const int i = blockIdx.x * 1024 + threadIdx.x;
  const unsigned long long b = __ballot(i < P && keep[i]);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int k = 0; k < 16; ++k) t += wsum[k];
    block_count[blockIdx.x] = t;
  }
}

extern "C" __global__ void __launch_bounds__(1024)
hgs_k_keep_scan(int nblocks, uint32_t* __restrict__ block_count, uint32_t* __restrict__ total) {
  __shared__ uint32_t wtot[16];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    const int k = base + threadIdx.x;
    const uint32_t v = k < nblocks ? block_count[k] : 0u;
    uint32_t tot;
    const uint32_t ex = hgs_block_excl_scan<1024>(v, wtot, tot);
    const uint32_t c = carry;
    if (k < nblocks) block_count[k] = c + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

// dst row of kept source row i = block base + rank of i among the kept rows of its block
extern "C" __global__ void __launch_bounds__(1024)
hgs_k_keep_index(int P, const uint8_t* __restrict__ keep, const uint32_t* __restrict__ block_base,
                 int32_t* __restrict__ src_of_dst) {
  __shared__ uint32_t wsum[16];
  const int i = blockIdx.x * 1024 + threadIdx.x;
  const bool k = i < P && keep[i];
  const unsigned long long b = __ballot(k);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) wsum[w] = (uint32_t)__popcll(b);
  __syncthreads();
  uint32_t base = block_base[blockIdx.x];
  for (int q = 0; q < w; ++q) base += wsum[q];
  if (k) {
    const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
    src_of_dst[base + r] = i;
  }
}

// dst[j][0..row_floats) = src[src_of_dst[j]][..]; one thread per float, coalesced on the output
extern "C" __global__ void __launch_bounds__(256)
hgs_k_gather_rows(long long n_out_floats, int row_floats, const int32_t* __restrict__ src_of_dst,
                  const float* __restrict__ src, float* __restrict__ dst) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_out_floats) return;
  const long long j = e / row_floats;
  const int c = (int)(e - j * row_floats);
  dst[e] = src[(long long)src_of_dst[j] * row_floats + c];
}

// ---- re-anchoring of the Gaussians on a posed mesh (animation.py:384-403) -----------------------------
// xyz[i] = u*v0 + v*v1 + w*v2 + dist[i] * n,  n = cross(v1-v0, v2-v0) / (|.| + 1e-20), (v0,v1,v2) = the
// vertices of face mapping_face[i].
extern "C" __global__ void __launch_bounds__(256)
hgs_k_reanchor(int P, const float* __restrict__ vertices, const int32_t* __restrict__ faces,
               const int32_t* __restrict__ mapping_face, const float* __restrict__ mapping_uvw,
               const float* __restrict__ mapping_dist, float* __restrict__ xyz) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const int f = mapping_face[i];
  const int i0 = faces[3 * f + 0], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
  const float ax = vertices[3 * i0], ay = vertices[3 * i0 + 1], az = vertices[3 * i0 + 2];
  const float bx = vertices[3 * i1], by = vertices[3 * i1 + 1], bz = vertices[3 * i1 + 2];
  const float cx = vertices[3 * i2], cy = vertices[3 * i2 + 1], cz = vertices[3 * i2 + 2];
  const float e1x = bx - ax, e1y = by - ay, e1z = bz - az;
  const float e2x = cx - ax, e2y = cy - ay, e2z = cz - az;
  float nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
  const float inv = 1.0f / (sqrtf(nx * nx + ny * ny + nz * nz) + 1e-20f);
  nx *= inv; ny *= inv; nz *= inv;
  const float u = mapping_uvw[3 * i], vv = mapping_uvw[3 * i + 1], w = mapping_uvw[3 * i + 2];
  const float d = mapping_dist[i];
  xyz[3 * i + 0] = (ax * u + bx * vv + cx * w) + d * nx;
  xyz[3 * i + 1] = (ay * u + by * vv + cy * w) + d * ny;
  xyz[3 * i + 2] = (az * u + bz * vv + cz * w) + d * nz;
}
