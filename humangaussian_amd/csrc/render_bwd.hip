// render_bwd.hip - blend backward (stage B1 of SURVEY.md 2.3(B)), re-designed for wave64.
//
// Upstream walks each pixel's list back-to-front and issues ~10 atomicAdd per
// (pixel, Gaussian) pair.  Here the roles are transposed: a wave owns one BUCKET of 64
// consecutive list entries of a tile (lane = Gaussian) and sweeps the tile's 256 pixels
// through the lanes as a systolic pipeline:
//
//     step s:  lane l handles pixel p = s - l;  the pixel's running state (T, F) enters
//              lane 0 from the bucket-boundary state the forward stored, and moves one
//              lane per step (ds_bpermute through the LDS crossbar; a DPP wave_shr:1
//              variant is kept behind HGS_BWD_DPP_SHIFT).
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

// This file is its own translation unit, built with -fno-slp-vectorize: the kernel is
// VALU-throughput-bound with ~6 waves per SIMD, where v_pk_* packing (same flop rate as
// scalar fp32 ops on gfx950, measured with tools/valu_ubench.hip) only adds register moves
// (207 -> 180 us at config 2).

namespace {

// shift a value one lane up the wave (lane l receives lane l-1); lane 0 receives `first`.
__device__ __forceinline__ float wave_shift_in(float prev_out, float first, int lane) {
#ifndef HGS_BWD_DPP_SHIFT
  // through the LDS crossbar (ds_bpermute): measured 174 us vs 180 us for the DPP form at
  // config 2 - v_mov_dpp wave_shr costs ~8 cycles of VALU issue, the LDS pipe has slack
  const float up = __shfl_up(prev_out, 1, 64);
  return lane == 0 ? first : up;
#else
  // DPP wave_shr:1 (0x138): GFX9-family full-wave shift; lane 0 has no source and keeps
  // `old` (bound_ctrl = 0), which we preload with the value entering the pipeline.
  (void)lane;
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(first),
                                                    __float_as_int(prev_out), 0x138, 0xf, 0xf,
                                                    false));
#endif
}

}  // namespace

extern "C" __global__ void __launch_bounds__(256)
hgs_k_render_bwd(View v, Layout L, const hgs_status* __restrict__ status,
                 const SortRec* __restrict__ recs_all,
                 const float* __restrict__ bstate, const float* __restrict__ segP,
                 const float* __restrict__ out_color,
                 const float* __restrict__ out_depth, const float* __restrict__ out_alpha,
                 const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                 const float* __restrict__ dL_dalpha, float* __restrict__ grad_rows) {
  // per-pixel constants, row-major inside the tile (p = y*16 + x), two float4 per pixel:
  //   plane A [gC0 gC1 gC2 gD], plane B [gA F' n_contrib(bits) pixel_x]
  // 64 dummy pixels (n_contrib = 0) pad both ends, so the systolic loop indexes with
  // p + 64 and needs neither a clamp nor a range test for the lanes still outside 0..255
  // Two float4 planes (lane stride 16 B => conflict-free ds_read_b128) and one float2 plane
  // per wave for the pipeline entry state (one ds_read_b64 per step).
  __shared__ float4 s_pixA[256 + 128], s_pixB[256 + 128];
  __shared__ float2 s_TF0[HGS_BWD_WAVES][256 + 128];

  // ---- which (tile, bucket group) is this workgroup?  binary search the WG prefix
  const uint32_t g = blockIdx.x;
  if (status->overflow || g >= L.tile_wgstart[v.T]) return;   // surplus workgroup
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
  const int tx0 = tile_x * HGS_TILE, ty0 = tile_y * HGS_TILE;

  {  // pixel constants: thread tid <-> row-major pixel tid
    const int px = tx0 + (tid & 15), py = ty0 + (tid >> 4);
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
    s_pixA[tid + 64] = make_float4(c0, c1, c2, gd);
    s_pixB[tid + 64] = make_float4(ga, fp, __uint_as_float(nc), (float)px);
    if (tid < 128) {                                   // the two pads: never active
      const int d = tid < 64 ? tid : tid + 256;
      s_pixA[d] = make_float4(0.f, 0.f, 0.f, 0.f);
      s_pixB[d] = make_float4(0.f, 0.f, __uint_as_float(0u), 0.f);
    }
  }
  __syncthreads();

  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const uint32_t b = grp * HGS_BWD_WAVES + (uint32_t)w;          // bucket inside the tile
  const uint32_t q0 = b * HGS_BUCKET;
  if (q0 >= n) return;                                  // no such bucket (wave-uniform)
  const uint32_t q = q0 + lane;
  const bool valid = q < n;

  float mx = 0.f, my = 0.f, qa = 0.f, qb = 0.f, qc = 0.f, op = 0.f;
  float cr = 0.f, cg = 0.f, cbl = 0.f, dep = 0.f;
  uint32_t entry = 0;
  if (valid) {
    const float4* src = reinterpret_cast<const float4*>(recs_all + start + q);
    const float4 r0 = src[0], r1 = src[1], r2 = src[2];
    mx = r0.x; my = r0.y; qa = r0.z; qb = r0.w; qc = r1.x; op = r1.y;
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

  // ---- pipeline entry state for the 256 pixels (4 per lane), row-major order.  A pixel
  // whose n_contrib <= q0 finished before this bucket (its forward wave may have exited
  // without storing the state): it can never be active here, give it a finite dummy state.
  {
    const float* bs = (b > 0) ? bstate + (size_t)(L.tile_bstart[t] + b - 1) * HGS_BSTATE_FLOATS
                              : nullptr;
    // the forward blends long lists in segments of HGS_SEG entries: bucket states hold C, D, W
    // relative to the segment start, the combine kernel left the segment's base in segP
    const uint32_t kseg = q0 / HGS_SEG;
    const float* base = (kseg > 0) ? segP + (size_t)(L.tile_msegstart[t] + kseg) * HGS_SEG_PLANES * HGS_TILE_PIX
                                   : nullptr;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pf = k * 64 + lane;                 // forward thread index
      int lx, ly;
      hgs_fwd_thread_pixel(pf, lx, ly);
      const int p = ly * 16 + lx;                   // row-major pixel
      const float4 pa = s_pixA[p + 64], pb = s_pixB[p + 64];
      float T0 = 1.0f, F0 = 0.0f;
      if (bs && __float_as_uint(pb.z) > q0) {
        T0 = bs[0 * 256 + pf];
        float c0 = bs[1 * 256 + pf], c1 = bs[2 * 256 + pf], c2 = bs[3 * 256 + pf];
        float d = bs[4 * 256 + pf], wt = bs[5 * 256 + pf];
        if (base) {
          c0 += base[0 * 256 + pf]; c1 += base[1 * 256 + pf]; c2 += base[2 * 256 + pf];
          d += base[3 * 256 + pf]; wt += base[4 * 256 + pf];
        }
        F0 = c0 * pa.x + c1 * pa.y + c2 * pa.z + d * pa.w + wt * pb.x;
      }
      s_TF0[w][p + 64] = make_float2(T0, F0);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)min((uint32_t)HGS_BUCKET, n - q0));
  const int nsteps = 256 + (int)m - 1;
  const float ty0f = (float)ty0;
  const float2* __restrict__ TF0w = s_TF0[w];
  {   // pads of the entry-state array are read (by lanes > 0) but never used
    s_TF0[w][lane] = make_float2(1.0f, 0.0f);
    s_TF0[w][lane + 320] = make_float2(1.0f, 0.0f);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  float a_mx = 0.f, a_my = 0.f, a_ca = 0.f, a_cb = 0.f, a_cc = 0.f, a_op = 0.f;
  float a_r = 0.f, a_g = 0.f, a_b = 0.f, a_d = 0.f;
  float T_out = 1.0f, F_out = 0.0f;

  for (int s = 0; s < nsteps; ++s) {
    const int pi = s - lane + 64;             // padded pixel index, 1 .. 382
    // lane 0 takes the pipeline entry state of ITS pixel (= pixel s); every lane reads its
    // own slot (conflict-free) and only lane 0 keeps it
    const float2 tf0 = TF0w[pi];
    const float T_in = wave_shift_in(T_out, tf0.x, lane);
    const float F_in = wave_shift_in(F_out, tf0.y, lane);
    const float4 pa = s_pixA[pi];             // gC0 gC1 gC2 gD
    const float4 pb = s_pixB[pi];             // gA F' n_contrib pixel_x
    // same dx/dy expressions as the forward (absolute pixel centre) so skip decisions agree
    const float dx = mx - pb.w;
    const float dy = my - (ty0f + (float)((pi - 64) >> 4));
    float G, alpha, m2, m3;
    const bool keep = hgs_eval_alpha(dx, dy, qa, qb, qc, op, G, alpha, m2, m3);
    // pad pixels have n_contrib = 0, so `q < n_contrib` also rejects out-of-range steps
    const bool act = keep && (q < __float_as_uint(pb.z));
    const float am = act ? op * G : 0.0f;     // un-clamped alpha (= op*G), 0 when inactive
    const float a = fminf(HGS_ALPHA_MAX, am);
    const float wgt = a * T_in;
    const float S = __builtin_fmaf(cr, pa.x, __builtin_fmaf(cg, pa.y, __builtin_fmaf(cbl, pa.z,
                    __builtin_fmaf(dep, pa.w, pb.x))));
    const float F_new = __builtin_fmaf(wgt, S, F_in);
    const float om = 1.0f - a;
    T_out = T_in * om;
    F_out = F_new;
    // om >= 0.01, so dLda is always finite; inactive pairs are removed through am = 0
    const float dLda = __builtin_fmaf(T_in, S, -((pb.y - F_new) * __builtin_amdgcn_rcpf(om)));
    a_r = __builtin_fmaf(wgt, pa.x, a_r);
    a_g = __builtin_fmaf(wgt, pa.y, a_g);
    a_b = __builtin_fmaf(wgt, pa.z, a_b);
    a_d = __builtin_fmaf(wgt, pa.w, a_d);
    const float k = am * dLda;                            // dL/dG * G  (= op * G * dL/dalpha)
    a_op += k;                                            // = op * sum G dL/dalpha; /op below
    // d(p2)/d(dx) = 2 qa dx + qb dy = m2 + qa dx ;  d(p2)/d(dy) = qb dx + 2 qc dy
    a_mx = __builtin_fmaf(k, __builtin_fmaf(qa, dx, m2), a_mx);
    a_my = __builtin_fmaf(k, __builtin_fmaf(qb, dx, m3 + m3), a_my);
    const float kdx = k * dx, kdy = k * dy;
    a_ca = __builtin_fmaf(kdx, dx, a_ca);
    a_cb = __builtin_fmaf(kdx, dy, a_cb);
    a_cc = __builtin_fmaf(kdy, dy, a_cc);
  }
  // dL/dopacity = sum G dL/dalpha = a_op / op  (a_op is 0 whenever op is 0: never active)
  a_op = (op != 0.0f) ? a_op / op : 0.0f;

  if (valid) {
    // undo the exp2 folding (d power = d p2 / log2e) and apply the conic factors
    const float il = 1.0f / HGS_LOG2E;
    row[0] = make_float4(a_mx * il, a_my * il, -0.5f * a_ca, -a_cb);
    row[1] = make_float4(-0.5f * a_cc, a_op, a_r, a_g);
    row[2] = make_float4(a_b, a_d, 0.f, 0.f);
  }
}
