// render_bwd.hip - blend backward (stage B1 of SURVEY.md 2.3(B)), re-designed for wave64.
//
// Upstream walks each pixel's list back-to-front and issues ~10 atomicAdd per
// (pixel, Gaussian) pair.  Here the roles are transposed: a wave owns one BUCKET of 64
// consecutive list entries of a tile (lane = Gaussian) and sweeps the tile's 256 pixels
// through the lanes as a systolic pipeline:
//
//     step s:  lane l handles pixel p = s - l;  the pixel's running state (T, F) enters
//              lane 0 from the bucket-boundary state the forward stored, and moves one
//              lane per step with a single DPP wave-shift.
//
// with  S_j   = c_j . g_C + d_j g_D + g_A                 (g_* = incoming pixel gradients)
//       F_i   = sum_{j<=i} w_j S_j,   w_j = alpha_j T_j   (prefix, flows with T)
//       F'    = out_color . g_C + out_depth g_D + out_alpha g_A   (= total + bg term)
//       dL/dalpha_i = T_i S_i - (F' - F_i) / (1 - alpha_i)
//
// which is algebraically upstream's back-to-front recurrence including the background
// term.  Each lane accumulates ITS Gaussian's 10 gradient sums over all pixels in
// registers - no cross-lane reduction, no atomics - and writes one 48 B row per entry;
// hgs_k_preprocess_bwd sums a Gaussian's rows in fixed order (deterministic).
// Buckets are independent => (#entries / 64) equal-sized work items: no load imbalance.
//
// Roofline: VALU-bound (~60 VALU per lane-step, 319 steps per bucket); HBM traffic per
// entry: 48 B record + 24 B/pixel/bucket state (= 96 B/entry) in, 48 B row out.
#include "hgs_common.h"

namespace {

// shift a value one lane up the wave (lane l receives lane l-1); lane 0 receives `first`.
__device__ __forceinline__ float wave_shift_in(float prev_out, float first) {
  // DPP wave_shr:1 (0x138): GFX9-family full-wave shift; lane 0 has no source and keeps
  // `old` (bound_ctrl = 0), which we preload with the value entering the pipeline.
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(first),
                                                    __float_as_int(prev_out), 0x138, 0xf, 0xf,
                                                    false));
}

}  // namespace

extern "C" __global__ void __launch_bounds__(256)
hgs_k_render_bwd(View v, Layout L, const float* __restrict__ out_color,
                 const float* __restrict__ out_depth, const float* __restrict__ out_alpha,
                 const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                 const float* __restrict__ dL_dalpha, float* __restrict__ grad_rows) {
  // pixel constants, row-major inside the tile (p = y*16 + x)
  __shared__ float s_gc0[256], s_gc1[256], s_gc2[256], s_gd[256], s_ga[256], s_fp[256];
  __shared__ uint32_t s_nc[256];
  __shared__ float s_T0[HGS_BWD_WAVES][256], s_F0[HGS_BWD_WAVES][256];

  // ---- which (tile, bucket group) is this workgroup?  binary search the WG prefix
  const uint32_t g = blockIdx.x;
  int lo = 0, hi = v.T;                       // invariant: wgstart[lo] <= g < wgstart[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (L.tile_wgstart[mid] <= g) lo = mid; else hi = mid;
  }
  const int t = lo;
  const uint32_t grp = g - L.tile_wgstart[t];
  const uint32_t start = L.tile_start[t];
  const uint32_t n = L.tile_start[t + 1] - start;
  const uint32_t maxc = L.tile_maxcontrib[t];
  const int tile_x = t % v.grid_x, tile_y = t / v.grid_x;
  const int tid = threadIdx.x;

  {  // pixel constants: thread tid <-> row-major pixel tid
    const int px = tile_x * HGS_TILE + (tid & 15), py = tile_y * HGS_TILE + (tid >> 4);
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, gd = 0.f, ga = 0.f, fp = 0.f;
    uint32_t nc = 0;
    if (px < v.W && py < v.H) {
      const size_t pix = (size_t)py * v.W + px, HW = (size_t)v.H * v.W;
      if (dL_dcolor) { c0 = dL_dcolor[pix]; c1 = dL_dcolor[HW + pix]; c2 = dL_dcolor[2 * HW + pix]; }
      if (dL_ddepth) gd = dL_ddepth[pix];
      if (dL_dalpha) ga = dL_dalpha[pix];
      fp = out_color[pix] * c0 + out_color[HW + pix] * c1 + out_color[2 * HW + pix] * c2 +
           out_depth[pix] * gd + out_alpha[pix] * ga;
      nc = L.n_contrib[pix];
    }
    s_gc0[tid] = c0; s_gc1[tid] = c1; s_gc2[tid] = c2; s_gd[tid] = gd; s_ga[tid] = ga;
    s_fp[tid] = fp; s_nc[tid] = nc;
  }
  __syncthreads();

  const int w = tid >> 6, lane = tid & 63;
  const uint32_t b = grp * HGS_BWD_WAVES + w;          // bucket inside the tile
  const uint32_t q0 = b * HGS_BUCKET;
  if (q0 >= n) return;                                  // no such bucket (wave-uniform)
  const uint32_t q = q0 + lane;
  const bool valid = q < n;

  float mx = 0.f, my = 0.f, ca = 0.f, cb = 0.f, cc = 0.f, op = 0.f;
  float cr = 0.f, cg = 0.f, cbl = 0.f, dep = 0.f;
  uint32_t entry = 0;
  if (valid) {
    const float4* src = reinterpret_cast<const float4*>(&L.recs[start + q]);
    const float4 r0 = src[0], r1 = src[1], r2 = src[2];
    mx = r0.x; my = r0.y; ca = r0.z; cb = r0.w; cc = r1.x; op = r1.y;
    cr = r1.z; cg = r1.w; cbl = r2.x; dep = r2.y;
    entry = __float_as_uint(r2.z);
  }
  float4* row = reinterpret_cast<float4*>(grad_rows) + 3 * (size_t)entry;

  if (q0 >= maxc) {               // nothing in this bucket ever contributed: zero rows
    if (valid) {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      row[0] = z; row[1] = z; row[2] = z;
    }
    return;
  }

  // ---- pipeline entry state for the 256 pixels (4 per lane), row-major order
  {
    const float* bs = (b > 0) ? L.bstate + (size_t)(L.tile_bstart[t] + b - 1) * HGS_BSTATE_FLOATS
                              : nullptr;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pf = k * 64 + lane;                 // forward thread index
      int lx, ly;
      hgs_fwd_thread_pixel(pf, lx, ly);
      const int p = ly * 16 + lx;                   // row-major pixel
      float T0 = 1.0f, F0 = 0.0f;
      if (bs) {
        T0 = bs[0 * 256 + pf];
        F0 = bs[1 * 256 + pf] * s_gc0[p] + bs[2 * 256 + pf] * s_gc1[p] +
             bs[3 * 256 + pf] * s_gc2[p] + bs[4 * 256 + pf] * s_gd[p] + bs[5 * 256 + pf] * s_ga[p];
      }
      s_T0[w][p] = T0;
      s_F0[w][p] = F0;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  const int tx0 = tile_x * HGS_TILE, ty0 = tile_y * HGS_TILE;
  const uint32_t m = min((uint32_t)HGS_BUCKET, n - q0);   // valid lanes in this bucket
  const int nsteps = 256 + (int)m - 1;

  float a_mx = 0.f, a_my = 0.f, a_ca = 0.f, a_cb = 0.f, a_cc = 0.f, a_op = 0.f;
  float a_r = 0.f, a_g = 0.f, a_b = 0.f, a_d = 0.f;
  float T_out = 1.0f, F_out = 0.0f;

  for (int s = 0; s < nsteps; ++s) {
    const int pe = min(s, 255);
    const float T_in = wave_shift_in(T_out, s_T0[w][pe]);
    const float F_in = wave_shift_in(F_out, s_F0[w][pe]);
    const int p = s - lane;
    const bool inrange = (p >= 0) && (p < 256);
    const int pc = min(max(p, 0), 255);
    const float g0 = s_gc0[pc], g1 = s_gc1[pc], g2 = s_gc2[pc], gd = s_gd[pc], ga = s_ga[pc];
    const float fp = s_fp[pc];
    const uint32_t nc = s_nc[pc];
    // same expression as the forward (absolute pixel centre) so skip decisions agree
    const float dx = mx - (float)(tx0 + (pc & 15)), dy = my - (float)(ty0 + (pc >> 4));
    float G, alpha;
    const bool keep = hgs_eval_alpha(dx, dy, ca, cb, cc, op, G, alpha);
    const bool act = keep && inrange && (q < nc);
    const float a = act ? alpha : 0.0f;
    const float wgt = a * T_in;
    const float S = cr * g0 + cg * g1 + cbl * g2 + dep * gd + ga;
    const float F_new = F_in + wgt * S;
    const float om = 1.0f - a;
    T_out = T_in * om;
    F_out = F_new;
    // om >= 0.01, so dLda is always finite; inactive pairs are removed by zeroing G
    const float dLda = T_in * S - (fp - F_new) * __frcp_rn(om);
    const float Gm = act ? G : 0.0f;
    a_r += wgt * g0; a_g += wgt * g1; a_b += wgt * g2; a_d += wgt * gd;
    a_op += Gm * dLda;
    const float dLdG = op * dLda;
    const float gdx = Gm * dx, gdy = Gm * dy;
    a_mx += dLdG * (-gdx * ca - gdy * cb);
    a_my += dLdG * (-gdy * cc - gdx * cb);
    a_ca += -0.5f * gdx * dx * dLdG;
    a_cb += -gdx * dy * dLdG;
    a_cc += -0.5f * gdy * dy * dLdG;
  }

  if (valid) {
    row[0] = make_float4(a_mx, a_my, a_ca, a_cb);
    row[1] = make_float4(a_cc, a_op, a_r, a_g);
    row[2] = make_float4(a_b, a_d, 0.f, 0.f);
  }
}
