// render_fwd.hip - front-to-back alpha blending (stage F6), list-parallel.
//
// Upstream runs one thread block per tile and walks the tile's whole depth-sorted list
// serially; on a human avatar a few tiles hold 1.5k+ entries and that serial chain, not the
// arithmetic, sets the kernel time.  Here a tile's list is cut into SEGMENTS of HGS_SEG
// (512) entries that are blended IN PARALLEL by different workgroups:
//
//   hgs_k_fwd_segT     for every segment that has a successor: P_k = product over the
//                      segment of (1 - alpha) per pixel (alpha evaluation only, no colours).
//   hgs_k_fwd_blend    one workgroup per segment (heavy tiles first).  Entry transmittance
//                      T_in = P_0 * ... * P_(k-1); because T only decreases, "the pixel was
//                      terminated by T < 1e-4 before this segment" is exactly T_in < 1e-4,
//                      so upstream's stop rule (the terminating Gaussian is not blended,
//                      nothing after it counts) is reproduced; inside the segment the blend
//                      is upstream's sequential loop.  Single-segment tiles write the image
//                      directly; others write per-segment partial sums.
//   hgs_k_fwd_combine  per multi-segment tile: sums the partials in segment order
//                      (deterministic), writes the image, and turns the partials into
//                      exclusive prefixes (segment bases) for the backward.
//
// Inside a segment the four wave64 of a workgroup are INDEPENDENT (no barriers): wave w owns
// the 8x8 pixel quadrant (w&1, w>>1) and walks the segment in buckets of 64 records:
//   1. coalesced 16 B/lane loads of the bucket's 48-byte records, issued one bucket ahead;
//   2. wavefront ballot + prefix popcount COMPACT the bucket to the records whose conservative
//      cull bit (computed at sort time) says they can reach alpha >= 1/255 in this quadrant
//      (about half) into a wave-private LDS slice, each carrying its list position;
//   3. a branch-free, 4x unrolled loop broadcasts the compacted records from LDS and blends
//      them with per-lane predication; T is the only loop-carried dependency.
// When a backward will follow, the per-pixel state (T absolute; C, D, W relative to the
// segment start) is stored at every 64-entry bucket boundary.
//
// Roofline: VALU issue (about 37 instructions per kept record per wave); HBM traffic
// 48 B/entry/wave in (L2-served after the first wave), 24 B/pixel out, 24 B/pixel/bucket
// state when storing, 32 B/pixel/segment for long lists.
#include "hgs_common.h"

// Compacted records per unrolled group of the blend loop (pad records >= this), a template parameter:
//   4: 88 VGPRs = 5 waves/SIMD - the shortest chain per tile: what a single view waits for (62 vs 66 us);
//   2: 64 VGPRs = 8 waves/SIMD - the best throughput: what counts with several views in flight
//      (8 views: 263 vs 306 us).  The host picks by the number of views of the call; the results are
//      bit-identical (the blend is sequential in list order either way).
// Cost classes of the backward work items (kept (entry, quadrant) pairs of the bucket, mean 97 at config 2):
// class 0 >= COST_0 > class 1 >= COST_1 > class 2 >= COST_2 > class 3.  render_bwd, single view: plain bump order
// 89.6 us; two classes split at 64 / 80 / 96 / 112 / 144: 77.6 / 76.6 / 78.7 / 80.9 / 85.3 us.
#ifndef HGS_BWD_COST_0
#define HGS_BWD_COST_0 128
#endif
#ifndef HGS_BWD_COST_1
#define HGS_BWD_COST_1 80
#endif
#ifndef HGS_BWD_COST_2
#define HGS_BWD_COST_2 40
#endif
#ifndef HGS_FWD_PAIRS
#define HGS_FWD_PAIRS 1          // calls of few views walk the lists with the PAIRED record stage (below): same bits
#endif                           // (tools/cmp_variant.py), 4.3 fewer VALU and 1.3 fewer LDS instructions per record,
                                 // render_fwd 70 -> 67 us; the many-view instantiation keeps the plain stage (its 64
                                 // registers / 8 waves per SIMD would become 68 / 7; not measured yet)
#ifndef HGS_FWD_UNROLL_FEW
#define HGS_FWD_UNROLL_FEW 4
#endif
#ifndef HGS_FWD_UNROLL_MANY
#define HGS_FWD_UNROLL_MANY 2
#endif
__host__ __device__ constexpr bool hgs_fwd_pairs(int unroll) {      // HGS_FWD_PAIRS = 2: every instantiation (to be measured)
  return HGS_FWD_PAIRS == 2 || (HGS_FWD_PAIRS == 1 && unroll == HGS_FWD_UNROLL_FEW);
}

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

// transmittance-only variant (no stop rule): P *= (1 - alpha) for kept pairs
__device__ __forceinline__ void tprod_one(float& P, float pxf, float pyf, const float4 r0,
                                          const float4 r1) {
  float G, alpha, m2, m3;
  const bool keep = hgs_eval_alpha(r0.x - pxf, r0.y - pyf, r0.z, r0.w, r1.x, r1.y, G, alpha, m2, m3);
  P *= keep ? (1.0f - alpha) : 1.0f;
}

// ---- PAIRED record stage (calls of few views, HGS_FWD_PAIRS).  The ISA of the loops below spends 4.3 of its
// 24 VALU instructions per record on v_mov_b32: the compiler packs the alpha arithmetic of two
// neighbouring records into v_pk_* instructions and has to shuffle (mx_u, mx_u+1) ... into register
// pairs first.  Here the compaction writes two records INTERLEAVED (24 dwords per pair)
//   mx0 mx1 my0 my1 | qa0 qa1 qb0 qb1 | qc0 qc1 op0 op1 | r0 g0 b0 d0 | r1 g1 b1 d1 | pos0 pos1 - -
// so that every ds_read_b128 already returns operand pairs: the per-record alpha evaluation runs packed
// across the two records, the colour / depth accumulation packed across channels, and the T chain stays
// scalar.  Same operations per element as blend_one / tprod_one (same bits).
typedef float hgs_f2 __attribute__((ext_vector_type(2)));
struct RecPair { float4 A, B, C, D0, D1; float2 E; };

__device__ __forceinline__ RecPair load_pair(const float* __restrict__ blk) {
  RecPair r;
  r.A = *reinterpret_cast<const float4*>(blk + 0);
  r.B = *reinterpret_cast<const float4*>(blk + 4);
  r.C = *reinterpret_cast<const float4*>(blk + 8);
  r.D0 = *reinterpret_cast<const float4*>(blk + 12);
  r.D1 = *reinterpret_cast<const float4*>(blk + 16);
  r.E = *reinterpret_cast<const float2*>(blk + 20);
  return r;
}

// alpha of the two records at the lane's pixel: hgs_eval_alpha on both halves of the pair
__device__ __forceinline__ void pair_alpha(const RecPair& r, float pxf, float pyf, float (&alpha)[2], bool (&keep)[2]) {
  const hgs_f2 dx = hgs_f2{r.A.x, r.A.y} - hgs_f2{pxf, pxf};
  const hgs_f2 dy = hgs_f2{r.A.z, r.A.w} - hgs_f2{pyf, pyf};
  const hgs_f2 m2 = __builtin_elementwise_fma(hgs_f2{r.B.x, r.B.y}, dx, hgs_f2{r.B.z, r.B.w} * dy);
  const hgs_f2 m3 = hgs_f2{r.C.x, r.C.y} * dy;
  const hgs_f2 p2 = __builtin_elementwise_fma(dx, m2, m3 * dy);
  const hgs_f2 G = {__builtin_amdgcn_exp2f(p2.x), __builtin_amdgcn_exp2f(p2.y)};
  const hgs_f2 og = hgs_f2{r.C.z, r.C.w} * G;
  alpha[0] = fminf(HGS_ALPHA_MAX, og.x);
  alpha[1] = fminf(HGS_ALPHA_MAX, og.y);
  keep[0] = (p2.x <= 0.0f) && (alpha[0] >= HGS_ALPHA_MIN);
  keep[1] = (p2.y <= 0.0f) && (alpha[1] >= HGS_ALPHA_MIN);
}

__device__ __forceinline__ void blend_pair(PixState& s, float pxf, float pyf, const RecPair& r) {
  float alpha[2];
  bool keep[2];
  pair_alpha(r, pxf, pyf, alpha, keep);
  hgs_f2 c01 = {s.C0, s.C1}, c2d = {s.C2, s.D};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float4 col = h ? r.D1 : r.D0;
    const bool live = keep[h] && !s.done;
    const float test_T = s.T * (1.0f - alpha[h]);
    const bool stop = live && (test_T < HGS_T_EPS);
    const bool upd = live && !stop;
    s.done = s.done || stop;
    const float wgt = upd ? alpha[h] * s.T : 0.0f;
    c01 = __builtin_elementwise_fma(hgs_f2{col.x, col.y}, hgs_f2{wgt, wgt}, c01);
    c2d = __builtin_elementwise_fma(hgs_f2{col.z, col.w}, hgs_f2{wgt, wgt}, c2d);
    s.Wt += wgt;
    s.T = upd ? test_T : s.T;
    s.last = upd ? __float_as_uint(h ? r.E.y : r.E.x) : s.last;
  }
  s.C0 = c01.x; s.C1 = c01.y; s.C2 = c2d.x; s.D = c2d.y;
}

__device__ __forceinline__ void tprod_pair(float& P, float pxf, float pyf, const RecPair& r) {
  float alpha[2];
  bool keep[2];
  pair_alpha(r, pxf, pyf, alpha, keep);
  P *= keep[0] ? (1.0f - alpha[0]) : 1.0f;
  P *= keep[1] ? (1.0f - alpha[1]) : 1.0f;
}

// Same walk as walk_segment below; BODY gets U / 2 record pairs.
template <int U, typename Pre, typename Alive, typename Body>
__device__ __forceinline__ void walk_segment_pairs(const float4* __restrict__ recs, uint32_t q_begin,
                                             uint32_t q_end, uint32_t wbit, float4* __restrict__ srec,
                                             int lane, Pre pre, Alive alive, Body body) {
  static_assert(U % 2 == 0, "pairs");
  float* __restrict__ sf = reinterpret_cast<float*>(srec);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 c0 = zero4, c1 = zero4, c2 = zero4;
  {
    const uint32_t q = q_begin + lane;
    if (q < q_end) { c0 = recs[3 * q + 0]; c1 = recs[3 * q + 1]; c2 = recs[3 * q + 2]; }
  }
  // one record into its half of its pair block
  auto put = [&](uint32_t p, const float4 a, const float4 b, const float4 c, float posf) {
    float* blk = sf + (p >> 1) * 24u;
    const uint32_t h = p & 1u;
    blk[0 + h] = a.x; blk[2 + h] = a.y; blk[4 + h] = a.z; blk[6 + h] = a.w;     // mx my qa qb
    blk[8 + h] = b.x; blk[10 + h] = b.y;                                        // qc op
    *reinterpret_cast<float4*>(blk + 12 + 4 * h) = make_float4(b.z, b.w, c.x, c.y);   // r g b depth
    blk[20 + h] = posf;
  };
  for (uint32_t j0 = q_begin; j0 < q_end; j0 += HGS_BUCKET) {
    if (!alive()) break;
    const uint32_t qn = j0 + HGS_BUCKET + lane;
    float4 n0 = zero4, n1 = zero4, n2 = zero4;
    if (qn < q_end) { n0 = recs[3 * qn + 0]; n1 = recs[3 * qn + 1]; n2 = recs[3 * qn + 2]; }
    const bool hit = (j0 + lane < q_end) && ((__float_as_uint(c2.w) & wbit) != 0u);
    const unsigned long long ball = __ballot(hit);
    const uint32_t cnt = (uint32_t)__popcll(ball);
    const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32),
                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)ball, 0u));
    pre(j0, cnt);
    __builtin_amdgcn_wave_barrier();                 // previous bucket's reads are done
    if (hit) put(pos, c0, c1, c2, __uint_as_float(j0 + lane + 1));
    if (lane < 2 * U) put(cnt + lane, zero4, zero4, zero4, 0.0f);      // pads: opacity 0 => alpha 0 => skipped
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t k0 = 0; k0 < cnt; k0 += U) {
      RecPair rp[U / 2];
#pragma unroll
      for (int u = 0; u < U / 2; ++u) rp[u] = load_pair(sf + ((k0 >> 1) + u) * 24u);
      body(rp);
    }
    c0 = n0; c1 = n1; c2 = n2;
  }
}
// Wave-level walk over list entries [q_begin, q_end) of one tile: loads, compaction, and a
// callback per group of 4 compacted records.  BODY(ra, rb, rc) gets float4[4] arrays;
// PRE(j0) runs at every bucket start (bucket-state stores); ALIVE() lets the wave stop early.
template <int U, typename Pre, typename Alive, typename Body>
__device__ __forceinline__ void walk_segment(const float4* __restrict__ recs, uint32_t q_begin,
                                             uint32_t q_end, uint32_t wbit, float4* __restrict__ srec,
                                             int lane, Pre pre, Alive alive, Body body) {
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 c0 = zero4, c1 = zero4, c2 = zero4;
  {
    const uint32_t q = q_begin + lane;
    if (q < q_end) { c0 = recs[3 * q + 0]; c1 = recs[3 * q + 1]; c2 = recs[3 * q + 2]; }
  }
  for (uint32_t j0 = q_begin; j0 < q_end; j0 += HGS_BUCKET) {
    if (!alive()) break;
    // issue the next bucket's loads now; they land while this bucket is processed
    const uint32_t qn = j0 + HGS_BUCKET + lane;
    float4 n0 = zero4, n1 = zero4, n2 = zero4;
    if (qn < q_end) { n0 = recs[3 * qn + 0]; n1 = recs[3 * qn + 1]; n2 = recs[3 * qn + 2]; }

    // ballot + prefix popcount compaction of the records that can touch this quadrant
    const bool hit = (j0 + lane < q_end) && ((__float_as_uint(c2.w) & wbit) != 0u);
    const unsigned long long ball = __ballot(hit);
    const uint32_t cnt = (uint32_t)__popcll(ball);
    const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32),
                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)ball, 0u));
    pre(j0, cnt);                                    // bucket start: state stores, cost bookkeeping
    __builtin_amdgcn_wave_barrier();                 // previous bucket's reads are done
    if (hit) {
      srec[3 * pos + 0] = c0;
      srec[3 * pos + 1] = c1;
      srec[3 * pos + 2] = make_float4(c2.x, c2.y, c2.z, __uint_as_float(j0 + lane + 1));
    }
    if (lane < 2 * U) {                 // pad records behind the last real one
      srec[3 * (cnt + lane) + 0] = zero4;
      srec[3 * (cnt + lane) + 1] = zero4;            // opacity 0 => alpha 0 => skipped
      srec[3 * (cnt + lane) + 2] = zero4;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    for (uint32_t k0 = 0; k0 < cnt; k0 += U) {
      float4 ra[U], rb[U], rc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ra[u] = srec[3 * (k0 + u) + 0]; rb[u] = srec[3 * (k0 + u) + 1]; rc[u] = srec[3 * (k0 + u) + 2];
      }
      body(ra, rb, rc);
    }
    c0 = n0; c1 = n1; c2 = n2;
  }
}


}  // namespace

// product of (1 - alpha) over segment i of a tile's list for this thread's pixel (no stop rule)
template <int U>
__device__ __forceinline__ float segment_tprod(const float4* __restrict__ recs, uint32_t i, int w,
                                               float4* __restrict__ srec, int lane, float pxf, float pyf) {
  float P = 1.0f;
  if constexpr (hgs_fwd_pairs(U)) {
    walk_segment_pairs<U>(recs, i * HGS_SEG, (i + 1) * HGS_SEG, 1u << (28 + w), srec, lane,
                          [](uint32_t, uint32_t) {}, [] { return true; },
                          [&](const RecPair (&rp)[U / 2]) {
#pragma unroll
                            for (int u = 0; u < U / 2; ++u) tprod_pair(P, pxf, pyf, rp[u]);
                          });
  } else {
    walk_segment<U>(recs, i * HGS_SEG, (i + 1) * HGS_SEG, 1u << (28 + w), srec, lane,
                    [](uint32_t, uint32_t) {}, [] { return true; },
                    [&](const float4 (&ra)[U], const float4 (&rb)[U], const float4 (&)[U]) {
#pragma unroll
                      for (int u = 0; u < U; ++u) tprod_one(P, pxf, pyf, ra[u], rb[u]);
                    });
  }
  return P;
}

// ------------------------------------------------------------------ segment transmittance
extern "C" __global__ void __launch_bounds__(HGS_FWD_THREADS)
hgs_k_fwd_segT(View v, Layout L, const hgs_status* __restrict__ status,
               const SortRec* __restrict__ recs_all, float* __restrict__ segT) {
  constexpr int U = HGS_FWD_UNROLL_FEW;
  __shared__ float4 s_rec[4][3 * (HGS_BUCKET + 2 * U)];
  const uint32_t ms = blockIdx.x;
  if (status->overflow || ms >= status->reserved[2]) return;
  const uint2 item = L.seg_item[ms];                // (global tile, segment): one load, no search
  const int g = (int)item.x;
  const uint32_t k = item.y;
  const uint32_t start = L.tile_start[g];
  const uint32_t n = L.tile_n[g];
  const uint32_t nseg = hgs_nseg(n);
  if (k + 1 >= nseg) return;                       // the last segment has no successor
  const int t = g % v.T;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  int lx, ly;
  hgs_fwd_thread_pixel(tid, lx, ly);
  const float pxf = (float)((t % v.grid_x) * HGS_TILE + lx), pyf = (float)((t / v.grid_x) * HGS_TILE + ly);
  const float4* __restrict__ recs = reinterpret_cast<const float4*>(recs_all + start);
  const float P = segment_tprod<U>(recs, k, w, s_rec[w], lane, pxf, pyf);
  segT[(size_t)ms * HGS_TILE_PIX + tid] = P;
}

// ---------------------------------------------------------------------------- blend
template <bool STORE, int U, bool FINE_CLASSES>
__device__ __forceinline__ void render_fwd_body(const View& v, const Layout& L, uint32_t seg_bound,
                                                const hgs_status* __restrict__ status,
                                                const SortRec* __restrict__ recs_all,
                                                float* __restrict__ bstate,
                                                const float* __restrict__ segT,
                                                float* __restrict__ segP,
                                                float* __restrict__ out_color,
                                                float* __restrict__ out_depth,
                                                float* __restrict__ out_alpha) {
  __shared__ float4 s_rec[4][3 * (HGS_BUCKET + 2 * U)];
  constexpr int MAXB = (HGS_SEG_THRESH > HGS_SEG ? HGS_SEG_THRESH : HGS_SEG) / HGS_BUCKET;   // buckets one workgroup blends
  __shared__ uint32_t s_cost[MAXB];                 // (entry, quadrant) pairs its waves kept, per bucket
  __shared__ uint32_t s_done;                       // waves of this workgroup that have finished blending
  const bool overflow = status->overflow != 0;
  const uint32_t total_items = status->bwd_groups;  // all backward work items of the call (not in the counters' cache line)
  int g;                                            // global tile = view * T + tile
  uint32_t k = 0;
  // Work items, in dispatch order: first the segments of the long lists (the heaviest tiles: their
  // chains start first), then one workgroup per remaining tile, heavy first.  The first
  // `seg_bound` blocks are a capacity bound on the number of segments; surplus ones exit.
  const bool LONG = blockIdx.x < seg_bound;
  if (LONG) {
    const uint32_t ms = blockIdx.x;
    if (overflow || ms >= status->reserved[2]) return;
    const uint2 item = L.seg_item[ms];              // (global tile, segment): one load, no search
    g = (int)item.x;
    k = item.y;
  } else {
    const uint32_t p = blockIdx.x - seg_bound;
    if (p >= (uint32_t)v.TT) return;
    g = overflow ? (int)p : (int)L.tile_order[p];   // lists are invalid on overflow: background only
  }
  const int bview = g / v.T, t = g % v.T;
  const size_t HW = (size_t)v.H * v.W;
  out_color += (size_t)bview * 3 * HW;
  out_depth += (size_t)bview * HW;
  out_alpha += (size_t)bview * HW;
  uint32_t* __restrict__ n_contrib = L.n_contrib + (size_t)bview * HW;
  const float* __restrict__ bg = v.cam[bview].bg;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int tile_x = t % v.grid_x, tile_y = t / v.grid_x;
  int lx, ly;
  hgs_fwd_thread_pixel(tid, lx, ly);
  const int px = tile_x * HGS_TILE + lx, py = tile_y * HGS_TILE + ly;
  const bool inside = (px < v.W) && (py < v.H);
  const float pxf = (float)px, pyf = (float)py;

  const uint32_t start = overflow ? 0u : L.tile_start[g];
  const uint32_t n = overflow ? 0u : L.tile_n[g];
  const uint32_t nseg = hgs_nseg(n);
  if (!LONG && nseg > 1) return;
  const uint32_t bstart = overflow ? 0u : L.tile_bstart[g];
  const uint32_t ms0 = (nseg > 1) ? L.tile_msegstart[g] : 0u;
  if (STORE) {
    if (tid < MAXB) s_cost[tid] = 0;
    if (tid == MAXB) s_done = 0;
    __syncthreads();
  }
  const float4* __restrict__ recs = reinterpret_cast<const float4*>(recs_all + start);

  PixState s;
  s.T = 1.0f; s.C0 = s.C1 = s.C2 = s.D = s.Wt = 0.f;
  s.last = 0;
  if (v.seg_recompute) {
    // lists of at most a few segments: the segment recomputes its predecessors' transmittance
    // products itself (same arithmetic as hgs_k_fwd_segT, so the same bits) and the pre-pass
    // kernel - 15 us alone on the GPU for ~60 long tiles - is not launched at all
    for (uint32_t i = 0; i < k; ++i) s.T *= segment_tprod<U>(recs, i, w, s_rec[w], lane, pxf, pyf);
  } else {
    for (uint32_t i = 0; i < k; ++i) s.T *= segT[(size_t)(ms0 + i) * HGS_TILE_PIX + tid];
  }
  // T only decreases: "terminated before this segment" <=> entry transmittance < 1e-4
  s.done = !inside || (s.T < HGS_T_EPS);

  const uint32_t seg_begin = nseg > 1 ? k * HGS_SEG : 0u, seg_end = nseg > 1 ? min(n, (k + 1) * HGS_SEG) : n;
  uint32_t wcost = 0;                               // this wave's kept records per bucket (lane = bucket)
  auto at_bucket = [&](uint32_t j0, uint32_t cnt) {
    // kept pairs per bucket: few views - lane i of a per-wave register keeps bucket i's count (one LDS atomic per
    // wave at the end); many views - one LDS add per bucket (the register would cost the eighth wave per SIMD)
    if (STORE && FINE_CLASSES) wcost = ((uint32_t)lane == (j0 - seg_begin) / HGS_BUCKET) ? cnt : wcost;
    if (STORE && !FINE_CLASSES && lane == 0 && cnt) atomicAdd(&s_cost[(j0 - seg_begin) / HGS_BUCKET], cnt);
    if (STORE && j0 > 0) {
      float* bs = bstate + (size_t)(bstart + j0 / HGS_BUCKET - 1) * HGS_BSTATE_FLOATS;
      bs[0 * 256 + tid] = s.T;
      bs[1 * 256 + tid] = s.C0;
      bs[2 * 256 + tid] = s.C1;
      bs[3 * 256 + tid] = s.C2;
      bs[4 * 256 + tid] = s.D;
      bs[5 * 256 + tid] = s.Wt;
    }
  };
  auto any_pixel_left = [&] { return __ballot(!s.done) != 0ull; };      // stop when every pixel is finished
  if constexpr (hgs_fwd_pairs(U)) {
    walk_segment_pairs<U>(recs, seg_begin, seg_end, 1u << (28 + w), s_rec[w], lane, at_bucket, any_pixel_left,
                          [&](const RecPair (&rp)[U / 2]) {
#pragma unroll
                            for (int u = 0; u < U / 2; ++u) blend_pair(s, pxf, pyf, rp[u]);
                          });
  } else {
    walk_segment<U>(recs, seg_begin, seg_end, 1u << (28 + w), s_rec[w], lane, at_bucket, any_pixel_left,
                    [&](const float4 (&ra)[U], const float4 (&rb)[U], const float4 (&rc)[U]) {
#pragma unroll
                      for (int u = 0; u < U; ++u) blend_one(s, pxf, pyf, ra[u], rb[u], rc[u]);
                    });
  }

  if (nseg == 1) {
    if (inside) {
      const size_t pix = (size_t)py * v.W + px;
      out_color[0 * HW + pix] = s.C0 + s.T * bg[0];
      out_color[1 * HW + pix] = s.C1 + s.T * bg[1];
      out_color[2 * HW + pix] = s.C2 + s.T * bg[2];
      out_depth[pix] = s.D;
      out_alpha[pix] = s.Wt;
      n_contrib[pix] = s.last;
    }
  } else {
    // partial sums of this segment; Tend < 0 marks "terminated (or already finished) here"
    float* sp = segP + (size_t)(ms0 + k) * HGS_SEG_PLANES * HGS_TILE_PIX;
    sp[0 * 256 + tid] = s.C0;
    sp[1 * 256 + tid] = s.C1;
    sp[2 * 256 + tid] = s.C2;
    sp[3 * 256 + tid] = s.D;
    sp[4 * 256 + tid] = s.Wt;
    sp[5 * 256 + tid] = s.done ? -s.T : s.T;
    sp[6 * 256 + tid] = __uint_as_float(s.last);
  }
  if (STORE && !overflow) {
    // tile-wide max of n_contrib: buckets at or beyond it are skipped by the backward
    const uint32_t mx = hgs_wave_max_u32(s.last);
    if ((tid & 63) == 0 && mx > 0) atomicMax(&L.tile_maxcontrib[g], mx);
    // Backward work items (one per 64-entry bucket this workgroup blended): (global tile, bucket, list
    // start, list length), so that a backward wave finds its records with ONE load.  An item's
    // duration follows the (entry, quadrant) pairs it must evaluate (device timestamps: 38k cycles
    // at the 10th percentile, 121k at the 90th, 161k max), and the kernel ends with its last
    // item: EXPENSIVE items are placed from the front of the table (dispatched first), cheap ones
    // from the back, with two bump cursors - a two-class longest-first schedule.  (Buckets no
    // wave reached - every pixel had terminated - cost 0 and still get their zero rows.)
    // The LAST of the four waves to get here places the items (no workgroup barrier: the waves are
    // independent and retire on their own; LDS atomics order the cost updates before the count).
    if (FINE_CLASSES && lane < MAXB && wcost) atomicAdd(&s_cost[lane], wcost);      // one LDS atomic per wave, not one per bucket
    uint32_t arrived = 0;
    if (lane == 0) arrived = atomicAdd(&s_done, 1u);
    arrived = (uint32_t)__builtin_amdgcn_readfirstlane((int)arrived);
    const uint32_t b0 = seg_begin / HGS_BUCKET, nbl = (seg_end - seg_begin + HGS_BUCKET - 1) / HGS_BUCKET;
    if (arrived == HGS_FWD_THREADS / 64 - 1) {       // nbl <= MAXB <= 64: one wave, at most four atomics per workgroup
      const bool mine = (uint32_t)lane < nbl;
      const uint32_t cost = s_cost[mine ? lane : 0];
      // four classes for calls of few views (the tail of the backward is what a single view waits for); two when
      // many views are in flight: every class is one more device-scope atomic on the same line per workgroup, and
      // 33k workgroups queueing on it cost the 8-view forward 65 us
      const uint32_t cls = FINE_CLASSES ? (cost >= HGS_BWD_COST_0 ? 0u : cost >= HGS_BWD_COST_1 ? 1u : cost >= HGS_BWD_COST_2 ? 2u : 3u)
                                        : (cost >= HGS_BWD_COST_1 ? 1u : 2u);
      unsigned long long bc[4];
#pragma unroll
      for (uint32_t c = 0; c < 4; ++c) bc[c] = __ballot(mine && cls == c);
      // the bump allocations travel together (lanes 0..3): one round trip at the end of the chain
      uint32_t base = 0;
#pragma unroll
      for (uint32_t c = 0; c < 4; ++c)
        if ((uint32_t)lane == c && bc[c]) base = atomicAdd(&L.ctr->bwd_cur[c], (uint32_t)__popcll(bc[c]));
      const unsigned long long below = (1ull << lane) - 1ull;
      uint32_t r = 0;
#pragma unroll
      for (uint32_t c = 0; c < 4; ++c) {
        const uint32_t cb = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)c);
        r = (cls == c) ? cb + (uint32_t)__popcll(bc[c] & below) : r;
      }
      if (mine)
        L.wg_tile[hgs_bwd_item_slot(cls, r, total_items, v.entry_capacity)] = make_uint4((uint32_t)g, b0 + (uint32_t)lane, start, n);
    }
  }
}

#define HGS_RENDER_FWD_KERNEL(NAME, STORE, U)                                                              \
  extern "C" __global__ void __launch_bounds__(HGS_FWD_THREADS) NAME(                                       \
      View v, Layout L, uint32_t seg_bound, const hgs_status* __restrict__ status,                           \
      const SortRec* __restrict__ recs, float* __restrict__ bstate, const float* __restrict__ segT,          \
      float* __restrict__ segP, float* __restrict__ out_color, float* __restrict__ out_depth,                \
      float* __restrict__ out_alpha) {                                                                       \
    HGS_TL_BEGIN();                                                                                          \
    render_fwd_body<STORE, U, (U == HGS_FWD_UNROLL_FEW)>(v, L, seg_bound, status, recs, bstate, segT, segP, out_color, out_depth, out_alpha); \
    HGS_TL_END(4, blockIdx.x < seg_bound                                                                     \
                      ? (blockIdx.x < status->reserved[2] ? (1ull << 32) | L.seg_item[blockIdx.x].y : 0ull) \
                      : (blockIdx.x - seg_bound < (uint32_t)v.TT ? L.tile_n[L.tile_order[blockIdx.x - seg_bound]] : 0u)); \
  }
HGS_RENDER_FWD_KERNEL(hgs_k_render_fwd_store, true, HGS_FWD_UNROLL_FEW)            // calls of < 3 views
HGS_RENDER_FWD_KERNEL(hgs_k_render_fwd_nostore, false, HGS_FWD_UNROLL_FEW)
HGS_RENDER_FWD_KERNEL(hgs_k_render_fwd_store_many, true, HGS_FWD_UNROLL_MANY)      // >= 3 views in flight
HGS_RENDER_FWD_KERNEL(hgs_k_render_fwd_nostore_many, false, HGS_FWD_UNROLL_MANY)

// -------------------------------------------------------------------------- combine
// One workgroup per tile_order position that can hold a tile with more than one segment (the
// launch covers the first capacity / HGS_SEG_THRESH positions); the others do nothing.  Thread = pf.
extern "C" __global__ void __launch_bounds__(HGS_FWD_THREADS)
hgs_k_fwd_combine(View v, Layout L, const hgs_status* __restrict__ status,
                  float* __restrict__ segP, float* __restrict__ out_color,
                  float* __restrict__ out_depth, float* __restrict__ out_alpha) {
  if (status->overflow || blockIdx.x >= status->active_tiles) return;
  const int g = (int)L.tile_order[blockIdx.x];      // heavy first: the multi-segment tiles lead the order
  const uint32_t n = L.tile_n[g];
  const uint32_t nseg = hgs_nseg(n);
  if (nseg <= 1) return;
  const int bview = g / v.T, t = g % v.T;
  const int tid = threadIdx.x;
  const uint32_t ms0 = L.tile_msegstart[g];
  float C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, Wt = 0.f, Tf = 1.0f;
  uint32_t last = 0;
  bool stopped = false;
  // The partials of FOUR segments are fetched together before any base is written back (same buffer: the
  // compiler may not move loads above the stores itself): one load latency per four segments instead of one
  // per segment - this kernel is nothing but that dependent chain.
  constexpr uint32_t CG = 4;
  for (uint32_t k0 = 0; k0 < nseg; k0 += CG) {
    float* sp0 = segP + (size_t)(ms0 + k0) * HGS_SEG_PLANES * HGS_TILE_PIX;
    float pv[CG][HGS_SEG_PLANES];
#pragma unroll
    for (uint32_t u = 0; u < CG; ++u) {
      const float* sp = sp0 + (size_t)min(u, nseg - 1u - k0) * HGS_SEG_PLANES * HGS_TILE_PIX;   // clamp: in range
#pragma unroll
      for (int pl = 0; pl < HGS_SEG_PLANES; ++pl) pv[u][pl] = sp[pl * 256 + tid];
    }
#pragma unroll
    for (uint32_t u = 0; u < CG; ++u) {
      if (k0 + u < nseg) {
        float* sp = sp0 + (size_t)u * HGS_SEG_PLANES * HGS_TILE_PIX;
        // exclusive prefix = what the backward adds to the segment-relative bucket states
        sp[0 * 256 + tid] = C0; sp[1 * 256 + tid] = C1; sp[2 * 256 + tid] = C2;
        sp[3 * 256 + tid] = D;  sp[4 * 256 + tid] = Wt;
        C0 += pv[u][0]; C1 += pv[u][1]; C2 += pv[u][2]; D += pv[u][3]; Wt += pv[u][4];   // later segments add exact zeros once stopped
        const float te = pv[u][5];
        if (!stopped) { Tf = fabsf(te); stopped = te < 0.0f; }
        last = max(last, __float_as_uint(pv[u][6]));
      }
    }
  }
  int lx, ly;
  hgs_fwd_thread_pixel(tid, lx, ly);
  const int px = (t % v.grid_x) * HGS_TILE + lx, py = (t / v.grid_x) * HGS_TILE + ly;
  if (px < v.W && py < v.H) {
    const size_t pix = (size_t)py * v.W + px;
    const size_t HW = (size_t)v.H * v.W;
    const float* __restrict__ bg = v.cam[bview].bg;
    float* oc = out_color + (size_t)bview * 3 * HW;
    oc[0 * HW + pix] = C0 + Tf * bg[0];
    oc[1 * HW + pix] = C1 + Tf * bg[1];
    oc[2 * HW + pix] = C2 + Tf * bg[2];
    out_depth[(size_t)bview * HW + pix] = D;
    out_alpha[(size_t)bview * HW + pix] = Wt;
    L.n_contrib[(size_t)bview * HW + pix] = last;
  }
}
