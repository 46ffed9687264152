// render_bwd.hip - blend backward (stage B1 of SURVEY.md 2.3(B)), bucket-parallel.
//
// Upstream walks each pixel's list back-to-front and issues ~10 atomicAdd per
// (pixel, Gaussian) pair.  Here one workgroup owns one BUCKET (64 consecutive entries) of one
// tile's depth-sorted list; the forward stored the per-pixel running state (T, C, D, W) at
// every bucket boundary, so all buckets of all tiles run in parallel - no serial chain over
// long lists, no load imbalance.  Inside the workgroup the layout is the forward's: four
// independent wave64, wave w = 8x8 pixel quadrant, lane = pixel, and the bucket's records
// are ballot/prefix-popcount COMPACTED per quadrant with the conservative cull mask (only
// ~41 % of (quadrant, entry) pairs survive - work the previous lane=Gaussian systolic
// formulation could not skip; it needed 2.3x more instructions, see
// render_bwd_systolic.hip.txt and DESIGN.md section 4).
//
// With  S_j = c_j . g_C + d_j g_D + g_A   (g_* = incoming pixel gradients),
//       F   = running sum of w_j S_j       (front to back, like T),
//       F'  = out_color . g_C + out_depth g_D + out_alpha g_A   (= total + background term)
//   dL/dalpha_j = T_j S_j - (F' - F_j) / (1 - alpha_j)
// which is algebraically upstream's back-to-front recurrence including the background term.
// The ten per-pixel gradient terms of a record are summed over the wave's 64 pixels with a
// 28-instruction reduce-scatter (wave_reduce.h), dropped into LDS per (entry, quadrant), and
// the four quadrant partials are added in fixed order: no atomics, bitwise reproducible.
// One 48 B gradient row per entry goes to HBM; hgs_k_preprocess_bwd sums a Gaussian's rows.
//
// Roofline: instruction issue (~85 instructions per kept record per wave); HBM traffic per
// entry: 4 x 48 B record reads (L2-served), 24 B/pixel/bucket state in, 48 B row out.
//
// This file is its own translation unit (built with -fno-slp-vectorize: the kernel is
// throughput-bound, where v_pk_* packing only adds register moves).
#include "hgs_common.h"
#include "wave_reduce.h"

#define HGS_BWD_UNROLL 2

extern "C" __global__ void __launch_bounds__(256)
hgs_k_render_bwd(View v, Layout L, const hgs_status* __restrict__ status,
                 const SortRec* __restrict__ recs_all,
                 const float* __restrict__ bstate, const float* __restrict__ segP,
                 const float* __restrict__ out_color,
                 const float* __restrict__ out_depth, const float* __restrict__ out_alpha,
                 const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                 const float* __restrict__ dL_dalpha, float* __restrict__ grad_rows) {
  __shared__ float4 s_rec[4][3 * (HGS_BUCKET + HGS_BWD_UNROLL)];
  __shared__ float s_part[HGS_BUCKET + 1][4][12];    // [slot][quadrant][value]; slot 64 = sink of the pads

  // ---- which (tile, bucket) is this workgroup?  binary search the bucket prefix
  const uint32_t g = blockIdx.x;
  if (status->overflow || g >= L.tile_wgstart[v.T]) return;   // surplus workgroup
  int lo = 0, hi = v.T;                       // invariant: wgstart[lo] <= g < wgstart[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (L.tile_wgstart[mid] <= g) lo = mid; else hi = mid;
  }
  const int t = lo;
  const uint32_t b = g - L.tile_wgstart[t];
  const uint32_t start = L.tile_start[t];
  const uint32_t n = L.tile_start[t + 1] - start;
  const uint32_t maxc = L.tile_maxcontrib[t];
  const uint32_t q0 = b * HGS_BUCKET;
  const uint32_t m = min((uint32_t)HGS_BUCKET, n - q0);      // entries in this bucket
  const int tid = threadIdx.x;
  const SortRec* __restrict__ brecs = recs_all + start + q0;

  if (q0 >= maxc) {               // nothing in this bucket ever contributed: zero rows
    if ((uint32_t)(tid >> 2) < m) {
      float* row = grad_rows + (size_t)brecs[tid >> 2].entry * HGS_ROW_FLOATS + (tid & 3) * 3;
      row[0] = 0.f; row[1] = 0.f; row[2] = 0.f;
    }
    return;
  }

  // zero the per-(entry, quadrant) partials: quadrants that cull an entry leave zeros
  {
    float4* z = reinterpret_cast<float4*>(&s_part[0][0][0]);
#pragma unroll
    for (int k = 0; k < 3; ++k) z[k * 256 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // ---- this thread's pixel (same ownership as the forward: pf = tid)
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  int lx, ly;
  hgs_fwd_thread_pixel(tid, lx, ly);
  const int px = (t % v.grid_x) * HGS_TILE + lx, py = (t / v.grid_x) * HGS_TILE + ly;
  const float pxf = (float)px, pyf = (float)py;
  float g0 = 0.f, g1 = 0.f, g2 = 0.f, gd = 0.f, ga = 0.f, fp = 0.f;
  uint32_t nc = 0;
  if (px < v.W && py < v.H) {
    const size_t pix = (size_t)py * v.W + px, HW = (size_t)v.H * v.W;
    if (dL_dcolor) { g0 = dL_dcolor[pix]; g1 = dL_dcolor[HW + pix]; g2 = dL_dcolor[2 * HW + pix]; }
    if (dL_ddepth) gd = dL_ddepth[pix];
    if (dL_dalpha) ga = dL_dalpha[pix];
    fp = out_color[pix] * g0 + out_color[HW + pix] * g1 + out_color[2 * HW + pix] * g2 +
         out_depth[pix] * gd + out_alpha[pix] * ga;
    nc = L.n_contrib[pix];
  }
  // running state at the bucket start.  A pixel with n_contrib <= q0 finished before this
  // bucket (its forward wave may have exited without storing the state) and is never active.
  float T = 1.0f, F = 0.0f;
  if (b > 0 && nc > q0) {
    const float* bs = bstate + (size_t)(L.tile_bstart[t] + b - 1) * HGS_BSTATE_FLOATS;
    T = bs[0 * 256 + tid];
    float c0 = bs[1 * 256 + tid], c1 = bs[2 * 256 + tid], c2 = bs[3 * 256 + tid];
    float d = bs[4 * 256 + tid], wt = bs[5 * 256 + tid];
    // long lists are blended in segments of HGS_SEG entries: C, D, W are relative to the segment
    // start, the combine kernel left the segment's base (exclusive prefix) in segP
    const uint32_t kseg = (hgs_nseg(n) > 1) ? q0 / HGS_SEG : 0u;
    if (kseg > 0) {
      const float* base = segP + (size_t)(L.tile_msegstart[t] + kseg) * HGS_SEG_PLANES * HGS_TILE_PIX;
      c0 += base[0 * 256 + tid]; c1 += base[1 * 256 + tid]; c2 += base[2 * 256 + tid];
      d += base[3 * 256 + tid]; wt += base[4 * 256 + tid];
    }
    F = c0 * g0 + c1 * g1 + c2 * g2 + d * gd + wt * ga;
  }

  // ---- compaction of the bucket's records for this quadrant (ballot + prefix popcount)
  float4* __restrict__ srec = s_rec[w];
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 c0 = zero4, c1 = zero4, c2 = zero4;
  if ((uint32_t)lane < m) {
    const float4* src = reinterpret_cast<const float4*>(brecs + lane);
    c0 = src[0]; c1 = src[1]; c2 = src[2];
  }
  const bool hit = ((uint32_t)lane < m) && ((__float_as_uint(c2.w) >> (28 + w)) & 1u);
  const unsigned long long ball = __ballot(hit);
  const uint32_t cnt = (uint32_t)__popcll(ball);
  const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32),
                                                 __builtin_amdgcn_mbcnt_lo((uint32_t)ball, 0u));
  if (hit) {
    srec[3 * pos + 0] = c0;
    srec[3 * pos + 1] = c1;
    srec[3 * pos + 2] = make_float4(c2.x, c2.y, c2.z, __uint_as_float((uint32_t)lane));   // slot in bucket
  }
  if (lane < HGS_BWD_UNROLL) {                       // pad records: opacity 0 (never active),
    srec[3 * (cnt + lane) + 0] = zero4;              // their (zero) sums go to the sink slot
    srec[3 * (cnt + lane) + 1] = zero4;
    srec[3 * (cnt + lane) + 2] = make_float4(0.f, 0.f, 0.f, __uint_as_float((uint32_t)HGS_BUCKET));
  }
  __syncthreads();                                   // s_part zeroed, s_rec ready

  const int row16 = lane >> 4;                       // DPP row of this lane
  const bool writer = (lane & 15) == 0;
  const int v0 = hgsred::slot_of(0, row16), v1 = hgsred::slot_of(1, row16), v2 = hgsred::slot_of(2, row16);

  for (uint32_t k0 = 0; k0 < cnt; k0 += HGS_BWD_UNROLL) {
#pragma unroll
    for (int u = 0; u < HGS_BWD_UNROLL; ++u) {
      const float4 r0 = srec[3 * (k0 + u) + 0];      // mx my qa qb
      const float4 r1 = srec[3 * (k0 + u) + 1];      // qc op r g
      const float4 r2 = srec[3 * (k0 + u) + 2];      // b depth entry slot
      const uint32_t slot = __float_as_uint(r2.w);
      // same dx/dy expressions as the forward so skip decisions agree
      const float dx = r0.x - pxf, dy = r0.y - pyf;
      float G, alpha, m2, m3;
      const bool keep = hgs_eval_alpha(dx, dy, r0.z, r0.w, r1.x, r1.y, G, alpha, m2, m3);
      const bool act = keep && (q0 + slot < nc);
      const float am = act ? r1.y * G : 0.0f;        // un-clamped alpha (= op*G), 0 when inactive
      const float a = fminf(HGS_ALPHA_MAX, am);
      const float wgt = a * T;
      const float S = __builtin_fmaf(r1.z, g0, __builtin_fmaf(r1.w, g1, __builtin_fmaf(r2.x, g2,
                      __builtin_fmaf(r2.y, gd, ga))));
      F = __builtin_fmaf(wgt, S, F);
      const float om = 1.0f - a;
      // om >= 0.01, so dLda is always finite; inactive pixels are removed through am = 0
      const float dLda = __builtin_fmaf(T, S, -((fp - F) * __builtin_amdgcn_rcpf(om)));
      T *= om;
      const float k = am * dLda;                     // dL/dG * G  (= op * G * dL/dalpha)
      const float kdx = k * dx, kdy = k * dy;
      float x[10], o[3];
      // d(p2)/d(dx) = 2 qa dx + qb dy = m2 + qa dx ;  d(p2)/d(dy) = qb dx + 2 qc dy
      x[0] = k * __builtin_fmaf(r0.z, dx, m2);
      x[1] = k * __builtin_fmaf(r0.w, dx, m3 + m3);
      x[2] = kdx * dx;
      x[3] = kdx * dy;
      x[4] = kdy * dy;
      x[5] = k;
      x[6] = wgt * g0;
      x[7] = wgt * g1;
      x[8] = wgt * g2;
      x[9] = wgt * gd;
      hgsred::reduce10(x, o);
      if (writer) {
        float* dst = &s_part[slot][w][0];
        dst[v0] = o[0];
        dst[v1] = o[1];
        if (v2 >= 0) dst[v2] = o[2];
      }
    }
  }
  __syncthreads();

  // ---- one gradient row per entry: quadrant partials added in fixed order, exp2 folding
  // undone (d power = d p2 / log2e), conic factors applied, dL/dopacity = sum(k) / op
  {
    const uint32_t e = (uint32_t)(tid >> 2), part = (uint32_t)(tid & 3);
    if (e < m) {
      const SortRec& rec = brecs[e];
      float* row = grad_rows + (size_t)rec.entry * HGS_ROW_FLOATS + part * 3;
      const float il = 1.0f / HGS_LOG2E;
      const float opi = (rec.op != 0.0f) ? 1.0f / rec.op : 0.0f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int vi = (int)part * 3 + j;
        const float sum = ((s_part[e][0][vi] + s_part[e][1][vi]) + s_part[e][2][vi]) + s_part[e][3][vi];
        float sc = 1.0f;
        if (vi == 0 || vi == 1) sc = il;
        else if (vi == 2 || vi == 4) sc = -0.5f;
        else if (vi == 3) sc = -1.0f;
        else if (vi == 5) sc = opi;
        row[j] = (vi < 10) ? sum * sc : 0.0f;
      }
    }
  }
}
