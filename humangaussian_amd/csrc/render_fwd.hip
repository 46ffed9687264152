// render_fwd.hip - per-tile front-to-back alpha blending (stage F6).
//
// One 256-thread workgroup per 16x16 tile; its four wave64 are INDEPENDENT (no barriers):
// wave w owns the 8x8 pixel quadrant (w&1, w>>1) and walks the tile's depth-sorted 48-byte
// record list in buckets of 64:
//   1. each lane loads one record of the bucket with coalesced 16 B/lane loads (the loads
//      for the NEXT bucket are issued before the current one is blended);
//   2. wavefront ballot + prefix popcount COMPACT the bucket to the records whose
//      conservative cull mask (computed at sort time) says they can reach alpha >= 1/255
//      somewhere in this wave's quadrant - typically about half - into a wave-private
//      3 KB LDS slice, each carrying its original list position;
//   3. a branch-free, 4x unrolled, software-pipelined loop broadcasts the compacted
//      records from LDS (ds_read_b128, same address in every lane) and blends them with
//      per-lane predication; the only loop-carried dependency is the transmittance T.
// Workgroups are issued heavy-tile-first (tile_order).  When the call needs a backward,
// the running per-pixel state (T, C, D, W) is stored at every 64-entry bucket boundary.
//
// Semantics follow upstream's renderCUDA of the ashawkey fork exactly (skip rules, the
// terminating Gaussian is not blended, depth not normalised, out_alpha = sum of weights,
// n_contrib = list position of the last blended Gaussian).
// Roofline: VALU-bound (about 24 VALU per kept pixel-Gaussian pair); HBM traffic is
// 48 B/entry/wave in (L2-served after the first wave) + 24 B/pixel out
// (+ 24 B/pixel/bucket state when storing).
#include "hgs_common.h"

namespace {

struct PixState {
  float T, C0, C1, C2, D, Wt;
  uint32_t last;
  bool done;
};

// Blend one compacted record into the lane's pixel, fully predicated.  r2.w carries the
// record's 1-based position in the tile list (pad records have opacity 0 and never blend).
__device__ __forceinline__ void blend_one(PixState& s, float pxf, float pyf, const float4 r0,
                                          const float4 r1, const float4 r2) {
  float G, alpha, m2, m3;
  const bool keep = hgs_eval_alpha(r0.x - pxf, r0.y - pyf, r0.z, r0.w, r1.x, r1.y, G, alpha, m2, m3);
  const bool live = keep && !s.done;
  const float test_T = s.T * (1.0f - alpha);
  const bool stop = live && (test_T < HGS_T_EPS);
  const bool upd = live && !stop;
  s.done = s.done || stop;
  const float wgt = upd ? alpha * s.T : 0.0f;
  s.C0 = __builtin_fmaf(r1.z, wgt, s.C0);
  s.C1 = __builtin_fmaf(r1.w, wgt, s.C1);
  s.C2 = __builtin_fmaf(r2.x, wgt, s.C2);
  s.D = __builtin_fmaf(r2.y, wgt, s.D);
  s.Wt += wgt;
  s.T = upd ? test_T : s.T;
  s.last = upd ? __float_as_uint(r2.w) : s.last;
}

}  // namespace

template <bool STORE>
__device__ __forceinline__ void render_fwd_body(const View& v, const Layout& L,
                                                const hgs_status* __restrict__ status,
                                                const SortRec* __restrict__ recs_all,
                                                float* __restrict__ bstate,
                                                float* __restrict__ out_color,
                                                float* __restrict__ out_depth,
                                                float* __restrict__ out_alpha) {
  constexpr int PPL = HGS_FWD_PPL;
  // wave-private compacted buckets (+4 zero-opacity pad records so the unrolled loop
  // needs neither index clamps nor a tail predicate)
  __shared__ float4 s_rec[HGS_FWD_WAVES][3 * (HGS_BUCKET + 4)];
  const bool overflow = status->overflow != 0;
  const int t = overflow ? (int)blockIdx.x : (int)L.tile_order[blockIdx.x];
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int tile_x = t % v.grid_x, tile_y = t / v.grid_x;

  int pf[PPL], px[PPL], py[PPL];
  float pxf[PPL], pyf[PPL];
  bool inside[PPL];
  PixState s[PPL];
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    pf[k] = w * (64 * PPL) + k * 64 + lane;
    int lx, ly;
    hgs_fwd_thread_pixel(pf[k], lx, ly);
    px[k] = tile_x * HGS_TILE + lx; py[k] = tile_y * HGS_TILE + ly;
    inside[k] = (px[k] < v.W) && (py[k] < v.H);
    pxf[k] = (float)px[k]; pyf[k] = (float)py[k];
    s[k].T = 1.0f; s[k].C0 = s[k].C1 = s[k].C2 = s[k].D = s[k].Wt = 0.f;
    s[k].last = 0;
    s[k].done = !inside[k];
  }

  const uint32_t start = overflow ? 0u : L.tile_start[t];
  const uint32_t n = overflow ? 0u : (L.tile_start[t + 1] - start);
  const uint32_t bstart = overflow ? 0u : L.tile_bstart[t];
  const float4* __restrict__ recs = reinterpret_cast<const float4*>(recs_all + start);
  const uint32_t wbits = hgs_fwd_wave_cullbits(w) << 28;
  float4* __restrict__ srec = s_rec[w];
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

  // records of the first bucket
  float4 c0 = zero4, c1 = zero4, c2 = zero4;
  if ((uint32_t)lane < n) { c0 = recs[3 * lane + 0]; c1 = recs[3 * lane + 1]; c2 = recs[3 * lane + 2]; }

  for (uint32_t j0 = 0; j0 < n; j0 += HGS_BUCKET) {
    bool alive = false;
#pragma unroll
    for (int k = 0; k < PPL; ++k) alive = alive || !s[k].done;
    if (__ballot(alive) == 0ull) break;              // every pixel of this wave is finished
    if (STORE && j0 > 0) {
      float* bs = bstate + (size_t)(bstart + j0 / HGS_BUCKET - 1) * HGS_BSTATE_FLOATS;
#pragma unroll
      for (int k = 0; k < PPL; ++k) {
        bs[0 * 256 + pf[k]] = s[k].T;
        bs[1 * 256 + pf[k]] = s[k].C0;
        bs[2 * 256 + pf[k]] = s[k].C1;
        bs[3 * 256 + pf[k]] = s[k].C2;
        bs[4 * 256 + pf[k]] = s[k].D;
        bs[5 * 256 + pf[k]] = s[k].Wt;
      }
    }
    // issue the next bucket's loads now; they land while this bucket is blended
    const uint32_t qn = j0 + HGS_BUCKET + lane;
    float4 n0 = zero4, n1 = zero4, n2 = zero4;
    if (qn < n) { n0 = recs[3 * qn + 0]; n1 = recs[3 * qn + 1]; n2 = recs[3 * qn + 2]; }

    // ballot + prefix popcount compaction of the records that can touch this wave's pixels
    const bool hit = (j0 + lane < n) && ((__float_as_uint(c2.w) & wbits) != 0u);
    const unsigned long long ball = __ballot(hit);
    const uint32_t cnt = (uint32_t)__popcll(ball);
    const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32),
                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)ball, 0u));
    __builtin_amdgcn_wave_barrier();                 // previous bucket's reads are done
    if (hit) {
      srec[3 * pos + 0] = c0;
      srec[3 * pos + 1] = c1;
      srec[3 * pos + 2] = make_float4(c2.x, c2.y, c2.z, __uint_as_float(j0 + lane + 1));
    }
    if (lane < 4) {                                  // 4 pad records behind the last real one
      srec[3 * (cnt + lane) + 0] = zero4;
      srec[3 * (cnt + lane) + 1] = zero4;            // opacity 0 => alpha 0 => skipped
      srec[3 * (cnt + lane) + 2] = zero4;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    constexpr int U = 4 / PPL;                       // records per unrolled group
    for (uint32_t k0 = 0; k0 < cnt; k0 += U) {
      float4 ra[U], rb[U], rc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ra[u] = srec[3 * (k0 + u) + 0]; rb[u] = srec[3 * (k0 + u) + 1]; rc[u] = srec[3 * (k0 + u) + 2];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int k = 0; k < PPL; ++k) blend_one(s[k], pxf[k], pyf[k], ra[u], rb[u], rc[u]);
      }
    }
    c0 = n0; c1 = n1; c2 = n2;
  }

  uint32_t mx = 0;
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    if (inside[k]) {
      const size_t pix = (size_t)py[k] * v.W + px[k];
      const size_t HW = (size_t)v.H * v.W;
      out_color[0 * HW + pix] = s[k].C0 + s[k].T * v.bg[0];
      out_color[1 * HW + pix] = s[k].C1 + s[k].T * v.bg[1];
      out_color[2 * HW + pix] = s[k].C2 + s[k].T * v.bg[2];
      out_depth[pix] = s[k].D;
      out_alpha[pix] = s[k].Wt;
      L.n_contrib[pix] = s[k].last;
    }
    mx = max(mx, s[k].last);
  }
  if (STORE && !overflow) {
    // tile-wide max of n_contrib: buckets at or beyond it are skipped by the backward
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    if ((tid & 63) == 0 && mx > 0) atomicMax(&L.tile_maxcontrib[t], mx);
  }
}

extern "C" __global__ void __launch_bounds__(HGS_FWD_THREADS)
hgs_k_render_fwd_store(View v, Layout L, const hgs_status* __restrict__ status,
                       const SortRec* __restrict__ recs, float* __restrict__ bstate,
                       float* __restrict__ out_color, float* __restrict__ out_depth,
                       float* __restrict__ out_alpha) {
  render_fwd_body<true>(v, L, status, recs, bstate, out_color, out_depth, out_alpha);
}

extern "C" __global__ void __launch_bounds__(HGS_FWD_THREADS)
hgs_k_render_fwd_nostore(View v, Layout L, const hgs_status* __restrict__ status,
                         const SortRec* __restrict__ recs, float* __restrict__ bstate,
                         float* __restrict__ out_color, float* __restrict__ out_depth,
                         float* __restrict__ out_alpha) {
  render_fwd_body<false>(v, L, status, recs, bstate, out_color, out_depth, out_alpha);
}
