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
// tools/render_bwd_systolic.hip.txt and DESIGN.md section 4).
//
// With  S_j = c_j . g_C + d_j g_D + g_A   (g_* = incoming pixel gradients),
//       F   = running sum of w_j S_j       (front to back, like T),
//       F'  = out_color . g_C + out_depth g_D + out_alpha g_A   (= total + background term)
//   dL/dalpha_j = T_j S_j - (F' - F_j) / (1 - alpha_j)
// which is algebraically upstream's back-to-front recurrence including the background term.
// The ten sums over the wave's 64 pixels that a record needs ARE a dense contraction (two
// per-(record, pixel) quantities against per-pixel constants, see the operand-A comment below),
// so they run on the matrix cores: v_mfma_f32_16x16x4_f32 (exact fp32), 8 records per batch,
// operands transposed through a wave-private LDS stage.  This replaced a 28-instruction
// VALU/DPP reduce-scatter per record (82 -> ~40 VALU instructions per kept record).  Results
// drop into LDS per (entry, quadrant) and the four quadrant partials are added in fixed order:
// no atomics, bitwise reproducible.  One 48 B gradient row per entry goes to HBM;
// hgs_k_preprocess_bwd sums a Gaussian's rows.
//
// Roofline: VALU issue (~40 instructions per kept record per wave) with the MFMA pipe running
// beside it (2 x 32 cycles per record); HBM traffic per entry: 4 x 48 B record reads
// (L2-served), 24 B/pixel/bucket state in, 48 B row out.
//
// This file is its own translation unit (it compiles in parallel with api.hip; same flags.  SLP vectorisation
// off was worth 1-2 % while every record of a batch was its own basic block; with the branch-free
// full-batch evaluation the v_pk_* pairs across neighbouring records win: 8 views 441 -> 435 us).
#include "hgs_common.h"

#ifndef HGS_BWD_BATCH
#define HGS_BWD_BATCH 8                  // records per MFMA batch (8 records x {k, wgt} = 16 columns).
#endif                                   // 4 (half-empty MFMAs, 7.8 KB LDS, 5 waves/SIMD) was measured:
                                         // 96 -> 127 us - the fp32 MFMA time is not hidden behind VALU work
#ifndef HGS_BWD_PAIRS
#define HGS_BWD_PAIRS 0                  // 1 (NOT measured / verified yet - check with tools/cmp_variant.py first): paired record
#endif                                   // stage like the forward's: two compacted records interleaved in LDS,
                                         //   mx0 mx1 my0 my1 | qa0 qa1 qb0 qb1 | qc0 qc1 op0 op1 | r0 r1 g0 g1 | b0 b1 d0 d1 | slot0 slot1
                                         // so that the record-parallel part of the evaluation (alpha, S, 1 - a) runs in
                                         // v_pk_* across the pair without register shuffles: 21.5 instead of 29 VALU
                                         // instructions per record by the ISA; T and F stay a scalar chain
typedef float hgs_f2 __attribute__((ext_vector_type(2)));
#define HGS_STAGE_STRIDE 68              // floats per staged column: 64 pixels + 4 (bank spread)
#define HGS_PART_FLOATS 10               // sums per (entry, quadrant)

typedef float hgs_f32x4 __attribute__((ext_vector_type(4)));

#if HGS_BWD_PAIRS
#define HGS_BWD_OCC __attribute__((amdgpu_num_vgpr(112)))     // 112 + 16 (8 accumulators, 8 spill slots in AGPRs): keeps 4 waves per SIMD
#else
#define HGS_BWD_OCC
#endif
extern "C" __global__ void __launch_bounds__(64 * HGS_BWD_WAVES) HGS_BWD_OCC
hgs_k_render_bwd(View v, Layout L, const hgs_status* __restrict__ status,
                 const SortRec* __restrict__ recs_all,
                 const float* __restrict__ bstate, const float* __restrict__ segP,
                 const float* __restrict__ out_color,
                 const float* __restrict__ out_depth, const float* __restrict__ out_alpha,
                 const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                 const float* __restrict__ dL_dalpha, float* __restrict__ grad_rows) {
  // HGS_BWD_WAVES (1, 2 or 4) wave64 per (tile, bucket); each sweeps 4 / HGS_BWD_WAVES of the tile's
  // 8x8 quadrants one after the other (lane = pixel of the current quadrant) and accumulates the
  // per-entry sums of its quadrants in its own LDS block, in a fixed order; one barrier at the
  // end, then the blocks are added in wave order.  3072 + 4352 + 2560 B of LDS per wave: 16 waves
  // per CU.  Fewer waves per bucket amortise the per-bucket loads better, more waves make the
  // work items shorter (at 5.6k buckets on 4096 wave slots the tail of long items dominates).
  __shared__ float4 s_rec_all[HGS_BWD_WAVES][3 * HGS_BUCKET];
  __shared__ __attribute__((aligned(16))) float s_part_all[HGS_BWD_WAVES][HGS_BUCKET][HGS_PART_FLOATS];   // [wave][slot][value]
  __shared__ __attribute__((aligned(16))) float stage_all[HGS_BWD_WAVES][2 * HGS_BWD_BATCH * HGS_STAGE_STRIDE];   // 2 B columns x 64 pixels
  constexpr int QW = 4 / HGS_BWD_WAVES;                      // quadrants per wave
  const int h = (HGS_BWD_WAVES == 1) ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  float4* __restrict__ s_rec = s_rec_all[h];
  float (*__restrict__ s_part)[HGS_PART_FLOATS] = s_part_all[h];
  float* __restrict__ stage = stage_all[h];
  // scratch for the basis transposition: 11 rows; the stage if it is large enough, else the record
  // buffer (768 floats), which is filled only after the basis has been read back
  float* __restrict__ basis = (2 * HGS_BWD_BATCH >= 11) ? stage : reinterpret_cast<float*>(s_rec);

  // ---- which (tile, bucket)?  The forward left (tile, bucket, list start, length) of every work item in wg_tile, heavy
  // tiles first (a binary search over a prefix array here cost 12 dependent loads).
  const uint32_t g = blockIdx.x;
#ifdef HGS_BWD_TIMING
  unsigned long long tm[8];
  const unsigned long long wall0 = wall_clock64();      // 100 MHz, the same clock on every XCD (the cycle counter is per XCD)
  tm[0] = __builtin_readcyclecounter();
#define HGS_TM(i) tm[i] = __builtin_readcyclecounter()
  unsigned long long tacc[4] = {0, 0, 0, 0}, tlast = 0;
#define HGS_TACC(i) { const unsigned long long tn_ = __builtin_readcyclecounter(); tacc[i] += tn_ - tlast; tlast = tn_; }
#define HGS_TSTART() tlast = __builtin_readcyclecounter()
#else
#define HGS_TM(i)
#define HGS_TACC(i)
#define HGS_TSTART()
#endif
  const uint32_t total_items = status->bwd_groups;
  if (status->overflow || g >= total_items) return;          // surplus workgroup
  // dispatch position -> (cost class, rank): classes in order 0 (most expensive) .. 3, sizes from the forward's cursors
  uint32_t cls = 0, r = g;
  {
    const uint32_t n0 = L.ctr->bwd_cur[0], n1 = L.ctr->bwd_cur[1], n2 = L.ctr->bwd_cur[2];
    if (r >= n0) { r -= n0; cls = 1; if (r >= n1) { r -= n1; cls = 2; if (r >= n2) { r -= n2; cls = 3; } } }
  }
  const uint4 item = L.wg_tile[hgs_bwd_item_slot(cls, r, total_items, v.entry_capacity)];   // one load: everything needed to find the records
  const int gt = (int)item.x;                               // global tile = view * T + tile
  const uint32_t b = item.y;
  const uint32_t start = item.z;
  const uint32_t n = item.w;
  const uint32_t maxc = L.tile_maxcontrib[gt];
  const int bview = gt / v.T, t = gt % v.T;
  {   // this view's planes
    const size_t HW = (size_t)v.H * v.W;
    out_color += (size_t)bview * 3 * HW; out_depth += (size_t)bview * HW; out_alpha += (size_t)bview * HW;
    if (dL_dcolor) dL_dcolor += (size_t)bview * 3 * HW;
    if (dL_ddepth) dL_ddepth += (size_t)bview * HW;
    if (dL_dalpha) dL_dalpha += (size_t)bview * HW;
  }
  const uint32_t* __restrict__ n_contrib = L.n_contrib + (size_t)bview * v.H * v.W;
  const uint32_t q0 = b * HGS_BUCKET;
  const uint32_t m = min((uint32_t)HGS_BUCKET, n - q0);      // entries in this bucket
  const int lane = (int)threadIdx.x & 63;
  const SortRec* __restrict__ brecs = recs_all + start + q0;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

  // the bucket's records: one per lane, kept in registers for the four compactions
  float4 c0 = zero4, c1 = zero4, c2 = zero4;
  if ((uint32_t)lane < m) {
    const float4* src = reinterpret_cast<const float4*>(brecs + lane);
    c0 = src[0]; c1 = src[1]; c2 = src[2];
  }
  if (q0 >= maxc) {               // nothing in this bucket ever contributed: zero rows
    if (h == 0 && (uint32_t)lane < m) {
      float4* row = reinterpret_cast<float4*>(grad_rows + (size_t)__float_as_uint(c2.z) * HGS_ROW_FLOATS);
      row[0] = zero4; row[1] = zero4; row[2] = zero4;
    }
    return;
  }
  {  // zero the per-entry sums: 640 floats
    float* z = &s_part[0][0];
#pragma unroll
    for (int k = 0; k < HGS_PART_FLOATS; ++k) z[k * 64 + lane] = 0.0f;
  }
  const uint32_t bs_index = L.tile_bstart[gt] + b - 1;
  const uint32_t kseg = (hgs_nseg(n) > 1) ? q0 / HGS_SEG : 0u;
  const uint32_t ms_index = (kseg > 0) ? L.tile_msegstart[gt] + kseg : 0u;
  const int mrow = lane & 15, kk = lane >> 4;
  const int tile_x0 = (t % v.grid_x) * HGS_TILE, tile_y0 = (t / v.grid_x) * HGS_TILE;
  HGS_TM(1);

  // Raw per-pixel inputs of one quadrant.  They are fetched one quadrant AHEAD (software
  // prefetch): a single wave has nobody else to hide its HBM round trips behind.
  struct PixRaw { float g0, g1, g2, gd, ga, o0, o1, o2, od, oa, T, s0, s1, s2, d, wt, e0, e1, e2, e3, e4; uint32_t nc; };
  const bool have_state = b > 0;
  const float* __restrict__ bs = bstate + (size_t)bs_index * HGS_BSTATE_FLOATS;
  const float* __restrict__ sbase = segP + (size_t)ms_index * HGS_SEG_PLANES * HGS_TILE_PIX;
  auto fetch = [&](int w) {
    PixRaw r;
    r.g0 = r.g1 = r.g2 = r.gd = r.ga = r.o0 = r.o1 = r.o2 = r.od = r.oa = 0.f;
    r.T = 1.0f; r.s0 = r.s1 = r.s2 = r.d = r.wt = 0.f; r.nc = 0;
    r.e0 = r.e1 = r.e2 = r.e3 = r.e4 = 0.f;
    const int pf = w * 64 + lane;
    const int px = tile_x0 + ((w & 1) << 3) + (lane & 7), py = tile_y0 + ((w >> 1) << 3) + (lane >> 3);
    if (px < v.W && py < v.H) {
      const size_t pix = (size_t)py * v.W + px, HW = (size_t)v.H * v.W;
      if (dL_dcolor) { r.g0 = dL_dcolor[pix]; r.g1 = dL_dcolor[HW + pix]; r.g2 = dL_dcolor[2 * HW + pix]; }
      if (dL_ddepth) r.gd = dL_ddepth[pix];
      if (dL_dalpha) r.ga = dL_dalpha[pix];
      r.o0 = out_color[pix]; r.o1 = out_color[HW + pix]; r.o2 = out_color[2 * HW + pix];
      r.od = out_depth[pix]; r.oa = out_alpha[pix];
    }
    if (have_state) {            // unconditional on n_contrib: one load round, selected below
      r.T = bs[0 * 256 + pf];
      r.s0 = bs[1 * 256 + pf]; r.s1 = bs[2 * 256 + pf]; r.s2 = bs[3 * 256 + pf];
      r.d = bs[4 * 256 + pf]; r.wt = bs[5 * 256 + pf];
      // long lists are blended in segments of HGS_SEG entries: C, D, W are relative to the segment
      // start, the combine kernel left the segment's base (exclusive prefix) in segP
      // (added at consumption: an add here would make the prefetch wait for its own loads)
      if (kseg > 0) {
        r.e0 = sbase[0 * 256 + pf]; r.e1 = sbase[1 * 256 + pf]; r.e2 = sbase[2 * 256 + pf];
        r.e3 = sbase[3 * 256 + pf]; r.e4 = sbase[4 * 256 + pf];
      }
    }
    return r;
  };
  // This wave's quadrants.  An entry can be dropped from a quadrant's list when it lies beyond
  // the deepest pixel of that quadrant (slot >= max n_contrib - q0: no pixel ever reached it).
  const int w_begin = h * QW;
  uint32_t cntq[QW], ncq[QW];
  unsigned long long ballq[QW];
#pragma unroll
  for (int j = 0; j < QW; ++j) {
    const int w = w_begin + j;
    const int px = tile_x0 + ((w & 1) << 3) + (lane & 7), py = tile_y0 + ((w >> 1) << 3) + (lane >> 3);
    ncq[j] = (px < v.W && py < v.H) ? n_contrib[(size_t)py * v.W + px] : 0u;
  }
#pragma unroll
  for (int j = 0; j < QW; ++j) {
    const uint32_t mx = hgs_wave_max_u32(ncq[j]);
    const int w = w_begin + j;
    ballq[j] = __ballot(((uint32_t)lane < m) && ((__float_as_uint(c2.w) >> (28 + w)) & 1u) &&
                        (q0 + (uint32_t)lane < mx));
    cntq[j] = (uint32_t)__popcll(ballq[j]);
  }
  int jnext = 0;
  while (jnext < QW && cntq[jnext] == 0) ++jnext;

#pragma unroll 1
  for (int j = jnext; j < QW; j = jnext) {
    const int w = w_begin + j;
    const unsigned long long ball = ballq[j];
    const uint32_t cnt = cntq[j];
    const bool hit = (ball >> lane) & 1ull;
    const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32),
                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)ball, 0u));
    const int px = tile_x0 + ((w & 1) << 3) + (lane & 7), py = tile_y0 + ((w >> 1) << 3) + (lane >> 3);
    const float pxf = (float)px, pyf = (float)py;
    const PixRaw cur = fetch(w);       // (fetching one quadrant ahead: 22 more VGPRs, measured neutral)
    const uint32_t nc = ncq[j];
    jnext = j + 1;
    while (jnext < QW && cntq[jnext] == 0) ++jnext;
    const float g0 = cur.g0, g1 = cur.g1, g2 = cur.g2, gd = cur.gd, ga = cur.ga;
    const float fp = cur.o0 * g0 + cur.o1 * g1 + cur.o2 * g2 + cur.od * gd + cur.oa * ga;
    // running state at the bucket start.  A pixel with n_contrib <= q0 finished before this
    // bucket (its forward wave may have exited without storing the state) and is never active.
    float T = 1.0f, F = 0.0f;
    if (have_state && nc > q0) {
      T = cur.T;
      F = (cur.s0 + cur.e0) * g0 + (cur.s1 + cur.e1) * g1 + (cur.s2 + cur.e2) * g2 + (cur.d + cur.e3) * gd +
          (cur.wt + cur.e4) * ga;
    }

    // ---- MFMA operand A: the per-pixel basis of this quadrant.
    // The ten sums over the quadrant's 64 pixels that a record needs are contractions of two
    // per-(record, pixel) quantities with per-pixel constants:
    //   k   (= op G dL/dalpha)  against  1, u, v, u^2, uv, v^2     (u, v = pixel - quadrant centre)
    //   wgt (= alpha T)         against  g_C0, g_C1, g_C2, g_D
    // dx = a - u, dy = b - v with (a, b) = mean - quadrant centre, so sum k dx^2 etc. follow from
    // the six moments.  One v_mfma_f32_16x16x4_f32 chain per batch of 8 records computes
    //   D[m][n] = sum_p A[m][p] B[p][n],  columns n < 8: k of record n, n >= 8: wgt of record n-8,
    // rows m < 6: moment basis, rows 6..9: pixel gradients (the cross blocks are not used).
    // Lane l supplies A[m = l & 15][k = l >> 4] and B[k = l >> 4][n = l & 15]; instruction
    // i = 4c + r contracts pixels p = 16c + 4(l >> 4) + r, so a lane fetches its four B values of
    // a c-group with one 16 B LDS read.
    {
      // every lane writes ITS pixel's ten basis values as one column of the stage; every lane then
      // reads the row it supplies to the MFMA (row 10 = zeros for the six unused rows of A)
      const float ub = (float)(lane & 7) - 3.5f, vb = (float)(lane >> 3) - 3.5f;
      basis[0 * HGS_STAGE_STRIDE + lane] = 1.0f;
      basis[1 * HGS_STAGE_STRIDE + lane] = ub;
      basis[2 * HGS_STAGE_STRIDE + lane] = vb;
      basis[3 * HGS_STAGE_STRIDE + lane] = ub * ub;
      basis[4 * HGS_STAGE_STRIDE + lane] = ub * vb;
      basis[5 * HGS_STAGE_STRIDE + lane] = vb * vb;
      basis[6 * HGS_STAGE_STRIDE + lane] = g0;
      basis[7 * HGS_STAGE_STRIDE + lane] = g1;
      basis[8 * HGS_STAGE_STRIDE + lane] = g2;
      basis[9 * HGS_STAGE_STRIDE + lane] = gd;
      basis[10 * HGS_STAGE_STRIDE + lane] = 0.0f;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float Areg[16];
    {
      const float* arow = basis + min(mrow, 10) * HGS_STAGE_STRIDE + 4 * kk;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 aq = *reinterpret_cast<const float4*>(arow + 16 * c);
        Areg[4 * c + 0] = aq.x; Areg[4 * c + 1] = aq.y; Areg[4 * c + 2] = aq.z; Areg[4 * c + 3] = aq.w;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();                 // basis reads done: its scratch may be overwritten
#if HGS_BWD_PAIRS
    float* __restrict__ s_recf = reinterpret_cast<float*>(s_rec);
    // field f (0 mx, 1 my, 2 qa, 3 qb, 4 qc, 5 op, 6 r, 7 g, 8 b, 9 depth, 10 slot) of compacted record p
    auto rec_at = [&](uint32_t p2_, int f) -> float& { return s_recf[(p2_ >> 1) * 24u + 2u * (uint32_t)f + (p2_ & 1u)]; };
    if (hit) {
      rec_at(pos, 0) = c0.x; rec_at(pos, 1) = c0.y; rec_at(pos, 2) = c0.z; rec_at(pos, 3) = c0.w;
      rec_at(pos, 4) = c1.x; rec_at(pos, 5) = c1.y; rec_at(pos, 6) = c1.z; rec_at(pos, 7) = c1.w;
      rec_at(pos, 8) = c2.x; rec_at(pos, 9) = c2.y; rec_at(pos, 10) = __uint_as_float((uint32_t)lane);   // slot in bucket
    }
#else
    if (hit) {
      s_rec[3 * pos + 0] = c0;
      s_rec[3 * pos + 1] = c1;
      s_rec[3 * pos + 2] = make_float4(c2.x, c2.y, c2.z, __uint_as_float((uint32_t)lane));   // slot in bucket
    }
#endif
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float cxq = (float)(tile_x0 + ((w & 1) << 3)) + 3.5f;
    const float cyq = (float)(tile_y0 + ((w >> 1) << 3)) + 3.5f;

    // Finishes a batch: D layout is lane l -> rows 4 (l >> 4) + r (r = register) of column l & 15.
    // The sums are ADDED to the entry's totals: quadrants 0..3 in this fixed order.
    auto finish = [&](const hgs_f32x4& a0, const hgs_f32x4& a1, uint32_t k0, uint32_t nrec) {
      const float d0 = a0[0] + a1[0], d1 = a0[1] + a1[1], d2 = a0[2] + a1[2], d3 = a0[3] + a1[3];
      // rows 4, 5 (uv, v^2 moments) of column n live in lane 16 + n: bring them to lane n
      const float k11 = __shfl(d0, (lane + 16) & 63, 64), k02 = __shfl(d1, (lane + 16) & 63, 64);
      const uint32_t rsel = (uint32_t)lane & (HGS_BWD_BATCH - 1);   // record of the batch this lane finishes
      if (rsel >= nrec) return;
#if HGS_BWD_PAIRS
      const uint32_t pr_ = k0 + rsel;
      const float4 q0r = make_float4(rec_at(pr_, 0), rec_at(pr_, 1), rec_at(pr_, 2), rec_at(pr_, 3));
      const float4 q1r = make_float4(rec_at(pr_, 4), 0.f, 0.f, 0.f);
      const uint32_t slot_l = __float_as_uint(rec_at(pr_, 10));
#else
      const float4 q0r = s_rec[3 * (k0 + rsel) + 0];
      const float4 q1r = s_rec[3 * (k0 + rsel) + 1];
      const uint32_t slot_l = __float_as_uint(s_rec[3 * (k0 + rsel) + 2].w);
#endif
      float* dst = &s_part[slot_l][0];
      if (lane < HGS_BWD_BATCH) {
        const float a = q0r.x - cxq, bb = q0r.y - cyq;
        const float k00 = d0, k10 = d1, k01 = d2, k20 = d3;
        const float sdx = __builtin_fmaf(a, k00, -k10), sdy = __builtin_fmaf(bb, k00, -k01);
        const float sxx = __builtin_fmaf(a, __builtin_fmaf(a, k00, -(k10 + k10)), k20);
        const float sxy = __builtin_fmaf(a, sdy, __builtin_fmaf(-bb, k10, k11));
        const float syy = __builtin_fmaf(bb, __builtin_fmaf(bb, k00, -(k01 + k01)), k02);
        // d(p2)/d(dx) = 2 qa dx + qb dy ;  d(p2)/d(dy) = qb dx + 2 qc dy
        const float x0 = __builtin_fmaf(q0r.z + q0r.z, sdx, q0r.w * sdy);
        const float x1 = __builtin_fmaf(q0r.w, sdx, (q1r.x + q1r.x) * sdy);
        float2* d2p = reinterpret_cast<float2*>(dst);
        const float2 o0 = d2p[0], o1 = d2p[1], o2 = d2p[2];
        d2p[0] = make_float2(o0.x + x0, o0.y + x1);
        d2p[1] = make_float2(o1.x + sxx, o1.y + sxy);
        d2p[2] = make_float2(o2.x + syy, o2.y + k00);
      } else if (lane >= 16 + HGS_BWD_BATCH && lane < 16 + 2 * HGS_BWD_BATCH) {   // rows 6, 7 of columns B..2B-1: sum wgt g_C0, g_C1
        float2* d2p = reinterpret_cast<float2*>(dst + 6);
        const float2 o = *d2p;
        *d2p = make_float2(o.x + d2, o.y + d3);
      } else if (lane >= 32 + HGS_BWD_BATCH && lane < 32 + 2 * HGS_BWD_BATCH) {   // rows 8, 9 of columns B..2B-1: sum wgt g_C2, g_D
        float2* d2p = reinterpret_cast<float2*>(dst + 8);
        const float2 o = *d2p;
        *d2p = make_float2(o.x + d0, o.y + d1);
      }
    };

    // Software pipeline: the MFMA chain of batch i runs while the wave evaluates batch i + 1; its
    // results are picked up (finish) only after that.
    hgs_f32x4 pa0 = {0.f, 0.f, 0.f, 0.f}, pa1 = {0.f, 0.f, 0.f, 0.f};
    uint32_t pk0 = 0, pn = 0;
    HGS_TSTART();
    // one record of the batch: everything that depends on (pixel, record); T and F are the only
    // values carried from record to record
    // the record's slot in the bucket = the position of the next set bit of the quadrant's ballot: SALU
    // work instead of a fourth LDS read per record (73.2 -> 72.2 us, 8 views 433.7 -> 428.6: the LDS pipe is
    // the co-limit of this kernel).  Reading the whole record through the scalar cache (s_load from the
    // sorted list) or with v_readlane from the bucket's registers was measured too: 77 -> 89 us both.
    unsigned long long mrem = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ball >> 32)) << 32) |
                              (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ball);
    auto eval_record = [&](uint32_t idx, float& kq, float& wgt) {
      const uint32_t slot = (uint32_t)__builtin_ctzll(mrem);
      mrem &= mrem - 1ull;
#if HGS_BWD_PAIRS                              // (partial batches only: single records out of the paired layout)
      const float4 r0 = make_float4(rec_at(idx, 0), rec_at(idx, 1), rec_at(idx, 2), rec_at(idx, 3));
      const float4 r1 = make_float4(rec_at(idx, 4), rec_at(idx, 5), rec_at(idx, 6), rec_at(idx, 7));
      const float2 r2 = make_float2(rec_at(idx, 8), rec_at(idx, 9));
#else
      const float4 r0 = s_rec[3 * idx + 0];    // mx my qa qb
      const float4 r1 = s_rec[3 * idx + 1];    // qc op r g
      const float2 r2 = *reinterpret_cast<const float2*>(&s_rec[3 * idx + 2]);    // b depth
#endif
      // same dx/dy expressions as the forward so skip decisions agree
      const float dx = r0.x - pxf, dy = r0.y - pyf;
      float G, alpha, m2, m3;
      const bool keep = hgs_eval_alpha(dx, dy, r0.z, r0.w, r1.x, r1.y, G, alpha, m2, m3);
      // `&`, not `&&`: a short-circuit here makes the compiler sink the slot read into a divergent
      // branch, which cuts the eight records of a batch into eight basic blocks (no overlap of one
      // record's LDS / exp / rcp latencies with its neighbours' arithmetic)
      const bool act = keep & (q0 + slot < nc);
      const float am = act ? r1.y * G : 0.0f;    // un-clamped alpha (= op*G), 0 when inactive
      const float a = fminf(HGS_ALPHA_MAX, am);
      wgt = a * T;
      const float S = __builtin_fmaf(r1.z, g0, __builtin_fmaf(r1.w, g1, __builtin_fmaf(r2.x, g2,
                      __builtin_fmaf(r2.y, gd, ga))));
      F = __builtin_fmaf(wgt, S, F);
      const float om = 1.0f - a;
      // om >= 0.01, so dLda is always finite; inactive pixels are removed through am = 0
      const float dLda = __builtin_fmaf(T, S, -((fp - F) * __builtin_amdgcn_rcpf(om)));
      T *= om;
      kq = am * dLda;                            // k = dL/dG * G
    };
#if HGS_BWD_PAIRS
    // two records of a full batch: the same operations per element as eval_record (same bits); the record-parallel
    // part packed across the pair, then the T / F chain for record 0 and record 1 in list order
    auto eval_pair = [&](uint32_t pair, float& kq0, float& wgt0, float& kq1, float& wgt1) {
      const float* __restrict__ blk = s_recf + pair * 24u;
      const float4 A = *reinterpret_cast<const float4*>(blk + 0);     // mx0 mx1 my0 my1
      const float4 Bq = *reinterpret_cast<const float4*>(blk + 4);    // qa0 qa1 qb0 qb1
      const float4 Cq = *reinterpret_cast<const float4*>(blk + 8);    // qc0 qc1 op0 op1
      const float4 D = *reinterpret_cast<const float4*>(blk + 12);    // r0 r1 g0 g1
      const float4 E = *reinterpret_cast<const float4*>(blk + 16);    // b0 b1 d0 d1
      const uint32_t slot0 = (uint32_t)__builtin_ctzll(mrem);
      mrem &= mrem - 1ull;
      const uint32_t slot1 = (uint32_t)__builtin_ctzll(mrem);
      mrem &= mrem - 1ull;
      const hgs_f2 dx = hgs_f2{A.x, A.y} - hgs_f2{pxf, pxf};
      const hgs_f2 dy = hgs_f2{A.z, A.w} - hgs_f2{pyf, pyf};
      const hgs_f2 m2 = __builtin_elementwise_fma(hgs_f2{Bq.x, Bq.y}, dx, hgs_f2{Bq.z, Bq.w} * dy);
      const hgs_f2 m3 = hgs_f2{Cq.x, Cq.y} * dy;
      const hgs_f2 p2 = __builtin_elementwise_fma(dx, m2, m3 * dy);
      const hgs_f2 G = {__builtin_amdgcn_exp2f(p2.x), __builtin_amdgcn_exp2f(p2.y)};
      const hgs_f2 og = hgs_f2{Cq.z, Cq.w} * G;                        // op * G
      const bool keep0 = (p2.x <= 0.0f) && (fminf(HGS_ALPHA_MAX, og.x) >= HGS_ALPHA_MIN);
      const bool keep1 = (p2.y <= 0.0f) && (fminf(HGS_ALPHA_MAX, og.y) >= HGS_ALPHA_MIN);
      const bool act0 = keep0 & (q0 + slot0 < nc), act1 = keep1 & (q0 + slot1 < nc);
      const float am0 = act0 ? og.x : 0.0f, am1 = act1 ? og.y : 0.0f;     // un-clamped alpha, 0 when inactive
      const hgs_f2 a = {fminf(HGS_ALPHA_MAX, am0), fminf(HGS_ALPHA_MAX, am1)};
      const hgs_f2 S = __builtin_elementwise_fma(hgs_f2{D.x, D.y}, hgs_f2{g0, g0},
                       __builtin_elementwise_fma(hgs_f2{D.z, D.w}, hgs_f2{g1, g1},
                       __builtin_elementwise_fma(hgs_f2{E.x, E.y}, hgs_f2{g2, g2},
                       __builtin_elementwise_fma(hgs_f2{E.z, E.w}, hgs_f2{gd, gd}, hgs_f2{ga, ga}))));
      const hgs_f2 om = hgs_f2{1.0f, 1.0f} - a;
      const float ri0 = __builtin_amdgcn_rcpf(om.x), ri1 = __builtin_amdgcn_rcpf(om.y);
      wgt0 = a.x * T;
      F = __builtin_fmaf(wgt0, S.x, F);
      const float dLda0 = __builtin_fmaf(T, S.x, -((fp - F) * ri0));
      T *= om.x;
      kq0 = am0 * dLda0;
      wgt1 = a.y * T;
      F = __builtin_fmaf(wgt1, S.y, F);
      const float dLda1 = __builtin_fmaf(T, S.y, -((fp - F) * ri1));
      T *= om.y;
      kq1 = am1 * dLda1;
    };
#endif
    for (uint32_t k0 = 0; k0 < cnt; k0 += HGS_BWD_BATCH) {
      const uint32_t nrec = min((uint32_t)HGS_BWD_BATCH, cnt - k0);
      if (nrec == HGS_BWD_BATCH) {
        // FULL batch (three of four at config 2): no per-record branch, ONE basic block for the eight
        // records, so the scheduler overlaps the LDS reads and the exp / rcp latencies of one record
        // with the arithmetic of its neighbours.  (With a wave-uniform `u < nrec` test per record
        // every record was its own block: read, wait, compute - 14 cycles per instruction per wave.)
#if HGS_BWD_PAIRS
#pragma unroll
        for (int u = 0; u < HGS_BWD_BATCH; u += 2) {      // staged pair by pair: the results do not pile up in registers
          float kqa, wga, kqb, wgb;
          // (20 record reads in flight would need 122 + 8 registers = 3 waves per SIMD: the batch is scheduled in two halves)
          if (u == HGS_BWD_BATCH / 2) __builtin_amdgcn_sched_barrier(0);
          eval_pair((k0 + u) >> 1, kqa, wga, kqb, wgb);
          stage[u * HGS_STAGE_STRIDE + lane] = kqa;
          stage[(u + 1) * HGS_STAGE_STRIDE + lane] = kqb;
          stage[(HGS_BWD_BATCH + u) * HGS_STAGE_STRIDE + lane] = wga;
          stage[(HGS_BWD_BATCH + u + 1) * HGS_STAGE_STRIDE + lane] = wgb;
        }
#else
        float kqv[HGS_BWD_BATCH], wgv[HGS_BWD_BATCH];
#pragma unroll
        for (int u = 0; u < HGS_BWD_BATCH; ++u) eval_record(k0 + u, kqv[u], wgv[u]);
#pragma unroll
        for (int u = 0; u < HGS_BWD_BATCH; ++u) {
          stage[u * HGS_STAGE_STRIDE + lane] = kqv[u];
          stage[(HGS_BWD_BATCH + u) * HGS_STAGE_STRIDE + lane] = wgv[u];
        }
#endif
      } else {
#pragma unroll
        for (int u = 0; u < HGS_BWD_BATCH; ++u) {
          float kq = 0.0f, wgt = 0.0f;
          if ((uint32_t)u < nrec) eval_record(k0 + u, kq, wgt);      // wave-uniform
          stage[u * HGS_STAGE_STRIDE + lane] = kq;
          stage[(HGS_BWD_BATCH + u) * HGS_STAGE_STRIDE + lane] = wgt;
        }
      }
      HGS_TACC(0);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float4 bq[4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
        bq[c] = *reinterpret_cast<const float4*>(&stage[min(mrow, 2 * HGS_BWD_BATCH - 1) * HGS_STAGE_STRIDE + 16 * c + 4 * kk]);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();               // the next batch overwrites the stage
      HGS_TACC(1);
      // keep the previous batch's accumulators in AGPRs up to here: read any earlier and the wave
      // waits for its MFMA chain before evaluating this batch (no overlap)
      asm volatile("" : "+a"(pa0), "+a"(pa1));
      if (pn) finish(pa0, pa1, pk0, pn);
      HGS_TACC(2);
      // ---- contraction over the 64 pixels on the matrix cores (two accumulators: no dependent stall)
      hgs_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(Areg[4 * c + 0], bq[c].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Areg[4 * c + 1], bq[c].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(Areg[4 * c + 2], bq[c].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Areg[4 * c + 3], bq[c].w, acc1, 0, 0, 0);
      }
      pa0 = acc0; pa1 = acc1; pk0 = k0; pn = nrec;
      HGS_TACC(3);
    }
    if (pn) finish(pa0, pa1, pk0, pn);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();                 // s_rec / stage are rewritten by the next quadrant
  }
  HGS_TM(4);

  // ---- one gradient row per entry (wave 0, lane = entry): the waves' blocks added in wave order,
  // exp2 folding undone (d power = d p2 / log2e), conic factors applied, dL/dopacity = sum(k) / op
  if (HGS_BWD_WAVES > 1) __syncthreads();
  if (h == 0 && (uint32_t)lane < m) {
    float sp[HGS_PART_FLOATS];
#pragma unroll
    for (int k = 0; k < HGS_PART_FLOATS; ++k) {
      sp[k] = s_part_all[0][lane][k];
#pragma unroll
      for (int hh = 1; hh < HGS_BWD_WAVES; ++hh) sp[k] += s_part_all[hh][lane][k];
    }
    const float op = c1.y;
    const float il = 1.0f / HGS_LOG2E;
    const float opi = (op != 0.0f) ? 1.0f / op : 0.0f;
    float4* row = reinterpret_cast<float4*>(grad_rows + (size_t)__float_as_uint(c2.z) * HGS_ROW_FLOATS);
    row[0] = make_float4(sp[0] * il, sp[1] * il, sp[2] * -0.5f, sp[3] * -1.0f);
    row[1] = make_float4(sp[4] * -0.5f, sp[5] * opi, sp[6], sp[7]);
    row[2] = make_float4(sp[8], sp[9], 0.0f, 0.0f);
  }
#ifdef HGS_BWD_TIMING
  HGS_TM(6);
  if (threadIdx.x == 0) {
    unsigned long long* o = L.keys + (size_t)g * 10;
    o[0] = tm[0]; o[1] = tm[1]; o[2] = tacc[0]; o[3] = tacc[1]; o[4] = tm[4]; o[5] = tacc[2]; o[6] = tm[6]; o[7] = tacc[3];
    o[8] = wall0; o[9] = wall_clock64();
  }
#endif
}
