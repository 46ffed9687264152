// render_bwd.hip - blend backward (stage B1 of SURVEY.md 2.3(B)), CELL-ROW mapping, and the pair reduction.
//
// Upstream walks each pixel's list back-to-front and issues ~10 atomicAdd per (pixel, Gaussian) pair.
// Here the unit of work is HGS_SEGLEN (128) consecutive entries of ONE CELL LIST (a 4x4-pixel cell of a tile, the
// records that can reach it, depth-ordered; written by the sort kernel): the forward stored the cell's per-pixel
// running state (T, C, D, W) at every HGS_SEGLEN-th entry, so all items are independent.  A wave64 is FOUR ROWS of
// 16 lanes; each row takes one work item (any cell of any tile), lane = pixel of the row's cell.  Persistent
// waves take groups of four items round-robin from a longest-first table (full segments, then partial ones by
// length class).
//
// Per row iteration (one record at 16 pixels), with
//       S_j = c_j . g_C + d_j g_D + g_A   (g_* = incoming pixel gradients),
//       F   = running sum of w_j S_j       (front to back, like T),
//       F'  = out_color . g_C + out_depth g_D + out_alpha g_A   (= total + background term)
//   dL/dalpha_j = T_j S_j - (F' - F_j) / (1 - alpha_j)
// which is algebraically upstream's back-to-front recurrence including the background term.
//
// The ten sums over the cell's 16 pixels that a record needs are contractions of two per-(record, pixel)
// quantities with per-pixel constants:
//   k   (= op G dL/dalpha)  against  1, u, v, u^2, uv, v^2     (u, v = pixel - cell centre)
//   wgt (= alpha T)         against  g_C0, g_C1, g_C2, g_D
// and run on the matrix cores, once per batch of 16 iterations, as v_mfma_f32_4x4x1_16b_f32 (exact fp32): SIXTEEN
// independent 4x4 outer-product blocks per instruction, block = 4 consecutive lanes,
//   D_b[q][jn] += A_b[q] * B_b[jn],      instruction t = in-cell pixel t of ALL FOUR cells:
//   block b = (cell b >> 2, records 4 (b & 3) .. + 3 of the batch), q = one of four quantities;
//   B: lane l = 16 cell + n supplies k (or wgt) of (cell, record n) at pixel t - it reads it from the LDS stage;
//   A: lane l supplies basis_{l & 3}(pixel t)  (for the gradient chain: g_{l & 3} at pixel t of cell l >> 4).
// Three chains (A = {1, u, v, u^2} x k; {uv, v^2} x k; {g_C0, g_C1, g_C2, g_D} x wgt), 16 instructions of 8 cycles
// each.  The accumulator layout puts ALL TEN sums of (cell, record n) into lane 16 cell + n - the lane that
// gathered that record - so the moment -> gradient conversion and the 40 B pair row need no shuffles.
// (First version: v_mfma_f32_16x16x4_f32 with a block-diagonal A - K slot = cell - to get the same layout: three
// quarters of its 32 cycles multiplied zeros, and the matrix pipe was 40 % of the kernel's issue time.)
//
// An (entry, cell) pair row lands at the row id the list element carries (the pair's entry-major id, or its chunk-cell-major
// id in calls of >= 3 views: binning.hip::hgs_put_pair); hgs_k_pair_reduce_{em,ch} adds the pair rows of
// every entry in cell order (fixed order: no atomics, bitwise reproducible) into one 48 B gradient row per
// entry, which hgs_k_preprocess_bwd sums per Gaussian.  Pair rows are 40 B (the ten sums, packed).
//
// Roofline: VALU issue (~35 instructions per row iteration) beside the MFMA pipe (3 x 8 cycles per
// iteration); HBM traffic per (entry, cell) pair: 4 B index + 48 B record gather (L2-served), 40 B pair row
// out, 24 B/pixel state per 128 pairs in.
//
// This file is its own translation unit (it compiles in parallel with api.hip; same flags).
#include "hgs_common.h"

#define HGS_STAGE_STRIDE 68              // floats per staged column: 64 pixels + 4 (bank spread)

typedef float hgs_f32x4 __attribute__((ext_vector_type(4)));
typedef float hgs_f32x2 __attribute__((ext_vector_type(2)));

namespace {

// the four work items of group `grp`, one per row: longest first (class 0 = full segments, then 1, 2, 3)
__device__ __forceinline__ bool fetch_item(const uint4* __restrict__ full, const uint4* __restrict__ part, size_t dcap,
                                           uint32_t q, uint32_t n0, uint32_t n1, uint32_t n2, uint32_t n3, uint4& item) {
  // ONE unconditional load from a selected address (a load per branch made the compiler wait for it on the spot)
  const bool ok = q < n0 + n1 + n2 + n3;
  const uint4* src = full + q;
  if (q >= n0) {
    const uint32_t q1 = q - n0;
    src = part + q1;
    if (q1 >= n1) {
      const uint32_t q2 = q1 - n1;
      src = (q2 < n2) ? part + (dcap - 1 - q2) : part + (dcap + (q2 - n2));
    }
  }
  if (!ok) src = part;                                  // (any valid slot; the caller masks every use with the result)
  item = *src;
  return ok;
}

}  // namespace

extern "C" __global__ void __launch_bounds__(64 * HGS_BWD_BLOCK_WAVES)
hgs_k_render_bwd(View v, Layout L, const hgs_status* __restrict__ status,
                 const SortRec* __restrict__ recs_all, const float* __restrict__ cstate,
                 const float* __restrict__ out_color,
                 const float* __restrict__ out_depth, const float* __restrict__ out_alpha,
                 const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                 const float* __restrict__ dL_dalpha, float* __restrict__ pair_rows, uint32_t pair_cap) {
  // wave-private LDS slices
  __shared__ float4 s_rec_all[HGS_BWD_BLOCK_WAVES][4 * HGS_ROW_F4];                 // [row][record][3] (+ pad: rows on different banks)
  __shared__ __attribute__((aligned(16))) float stage_k_all[HGS_BWD_BLOCK_WAVES][HGS_RB * HGS_STAGE_STRIDE];   // [iteration][pixel lane]
  __shared__ __attribute__((aligned(16))) float stage_w_all[HGS_BWD_BLOCK_WAVES][HGS_RB * HGS_STAGE_STRIDE];
  __shared__ uint32_t s_ticket;
  __shared__ __attribute__((aligned(16))) float s_tabA[4][36];      // MFMA operand A of the two moment chains: [l & 3][A1 16 | A2 16 | pad:
                                                                    // rows 36 floats apart keep the four rows' b128 reads on disjoint banks]
  if (status->overflow) return;
  // the caller sized the scratch for pair_cap pair rows (hgs_backward*: status->num_pairs, or 16 per entry); a count the
  // device does not confirm writes nothing here and poisons the gradient rows in the reduction (NaN: loud, in bounds)
  if ((uint32_t)L.ctr->alloc_ps > pair_cap) return;
  const int lane = (int)threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  float4* __restrict__ s_rec = s_rec_all[wv];
  float* __restrict__ stage_k = stage_k_all[wv];
  float* __restrict__ stage_w = stage_w_all[wv];
  if (threadIdx.x == 0) s_ticket = HGS_BWD_BLOCK_WAVES;      // tickets 0 .. waves - 1: every wave's first group
  if (threadIdx.x < 128) {
    const int qs = (int)threadIdx.x >> 5, e = (int)threadIdx.x & 31, t = e & 15;
    const float u = (float)(t & 3) - 1.5f, w_ = (float)(t >> 2) - 1.5f;
    s_tabA[qs][e] = e < 16 ? (qs == 0 ? 1.0f : (qs == 1 ? u : (qs == 2 ? w_ : u * u)))
                           : (qs == 0 ? u * w_ : (qs == 1 ? w_ * w_ : 0.0f));
  }
  __syncthreads();
  const int j = lane >> 4, i = lane & 15;
  // the work tables are per die: this workgroup draws from those of the die the dispatcher puts it on (Counters::sched)
  const uint32_t die = blockIdx.x % HGS_NXCD, lb = blockIdx.x / HGS_NXCD, nlb = gridDim.x / HGS_NXCD;
  const size_t dcap = hgs_die_cells(v.TT);
  const uint4* __restrict__ it_full = L.items_full + (size_t)die * L.full_cap;
  const uint4* __restrict__ it_part = L.items_part + (size_t)die * 2 * dcap;
  const uint32_t n0 = (uint32_t)L.ctr->sched[die][0], n1 = (uint32_t)(L.ctr->sched[die][0] >> 32);      // items per class
  const uint32_t n2 = (uint32_t)L.ctr->sched[die][1], n3 = (uint32_t)(L.ctr->sched[die][1] >> 32);
  const uint32_t ngroups = (n0 + n1 + n2 + n3 + 3u) / 4u;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* __restrict__ recs = reinterpret_cast<const float4*>(recs_all);
  float4* __restrict__ srow = s_rec + j * HGS_ROW_F4;

  // ---- MFMA operand A of the two moment chains: constants of the lane - lane l supplies A[q = l & 3] of block l >> 2 (its
  // cell: l >> 4); instruction t: in-cell pixel t.  They live in a 512 B LDS table (s_tabA) and are read four at a time
  // beside the B operands: as 32 registers per lane they were parked in AGPRs and copied back before every use
  // (93 v_accvgpr_read per 16-record batch in a kernel that is VALU-issue bound).
  const int qsel = lane & 3;
  const float* __restrict__ tabA = s_tabA[qsel];

#ifdef HGS_TIMELINE
  // per group: wall start | end | batches  (into L.keys: free after the sort)
  unsigned long long tl_w0 = 0, tl_w1 = 0;
  uint32_t tl_nb = 0, tl_live = 0;
#endif
  // Persistent workgroups of HGS_BWD_BLOCK_WAVES waves (12: one workgroup per CU, three waves per SIMD).  A die's group
  // table is longest first; its workgroup b (of G)
  // owns groups b, 2 G - 1 - b, 2 G + b, ... (odd rounds run backwards, so every workgroup gets a
  // similar total), and its waves DRAW them in that order through an LDS ticket: the wave that finishes first takes
  // the next, the SIMDs of the CU end within one short group of each other.  (Static round-robin per wave: a view
  // has ~1.16 groups per wave, the kernel ran 60 us for 47 us of mean load, a quarter of it with SIMDs running dry.
  // A device-wide ticket - one device-scope atomic per group on ONE address - serialised at the memory side of the
  // fabric: ~10 ns each, 110 us per view.)
  auto group_of = [&](uint32_t tk) {
    return tk * nlb + ((tk & 1u) ? nlb - 1u - lb : lb);      // (odd rounds run through the group table backwards)
  };
  // the item of the NEXT group is fetched while this one is processed (a group's start is a chain of dependent
  // loads - item, cell list, records - at ~2 us each, and a view has more than one group per wave)
  uint32_t grp = group_of((uint32_t)wv);
  uint4 item;                                            // (cell key, entries, first cell-list slot, state slot or ~0)
  bool have = fetch_item(it_full, it_part, dcap, 4u * grp + (uint32_t)j, n0, n1, n2, n3, item);
  while (grp < ngroups) {                                // (every later ticket of this workgroup lies behind it as well)
    uint32_t grp_next;
    {
      uint32_t t_next = 0;
      if (lane == 0) t_next = atomicAdd(&s_ticket, 1u);  // (LDS)
      grp_next = group_of((uint32_t)__builtin_amdgcn_readfirstlane((int)t_next));
    }
#ifdef HGS_TIMELINE
    tl_w0 = wall_clock64(); tl_nb = 0; tl_live = 0;
#endif
    const uint32_t key = item.x, cnt = have ? item.y : 0u;
    const int g = (int)(key >> 4), c = (int)(key & 15u);
    const int bview = g / v.T, t_ = g % v.T;
    const int cx0 = (t_ % v.grid_x) * HGS_TILE + (c & 3) * HGS_CELL, cy0 = (t_ / v.grid_x) * HGS_TILE + (c >> 2) * HGS_CELL;
    const int px = cx0 + (i & 3), py = cy0 + (i >> 2);
    const float pxf = (float)px, pyf = (float)py;
    const float cxq = (float)cx0 + 1.5f, cyq = (float)cy0 + 1.5f;      // cell centre
    const uint32_t tstart1 = L.tile_start[have ? g : 0] - 1u;         // record index - tstart1 = 1-based position in the tile list
    const uint2* __restrict__ list = L.cell_list + (have ? item.z : 0u);      // (record index, pair id)

    // ---- the group's start: ONE round of independent loads (cell list, per-pixel inputs, stored state - every one
    // unconditional, from a clamped address, masked afterwards), then the first record gather.  (With the loads inside
    // `if (inside)` blocks each block waited for its own data before the next one was issued: item -> pixels -> list ->
    // records, four memory round trips of ~2 us in front of every group.)
    const uint32_t safe_rec = tstart1 + 1u;              // the tile's first record (tile 0's for a row without item): always a valid slot
    auto list_at = [&](uint32_t e) { return list[(e < cnt) ? e : 0u]; };
    auto gather = [&](const uint2 le, bool valid, float4& r0, float4& r1, float4& r2) {
      const uint32_t idx = valid ? le.x : safe_rec;
      r0 = recs[3 * (size_t)idx]; r1 = recs[3 * (size_t)idx + 1];
      const float4 t2 = recs[3 * (size_t)idx + 2];
      r2 = make_float4(t2.x, t2.y, t2.z, __uint_as_float(valid ? idx - tstart1 : 0xffffffffu));   // .w: 1-based position in the tile list
    };
    float4 c0, c1, c2;
    uint2 le_next = list_at((uint32_t)i);
    const uint2 le_second = list_at(HGS_RB + (uint32_t)i);
    const bool inside = have && px < v.W && py < v.H;
    const bool has_state = inside && item.w != 0xffffffffu;
    float g0, g1, g2, gd, ga, fp, T, F;
    uint32_t nc;
    {
      const size_t HW = (size_t)v.H * v.W, pix = inside ? (size_t)py * v.W + px : 0;
      const size_t o1 = inside ? (size_t)bview * HW : 0, o3 = 3 * o1;
      const float* pc = dL_dcolor ? dL_dcolor : out_color;          // (absent gradient: any readable address, value masked)
      const float* pd = dL_ddepth ? dL_ddepth : out_depth;
      const float* pa = dL_dalpha ? dL_dalpha : out_alpha;
      const float r_g0 = pc[o3 + pix], r_g1 = pc[o3 + HW + pix], r_g2 = pc[o3 + 2 * HW + pix];
      const float r_gd = pd[o1 + pix], r_ga = pa[o1 + pix];
      const float r_c0 = out_color[o3 + pix], r_c1 = out_color[o3 + HW + pix], r_c2 = out_color[o3 + 2 * HW + pix];
      const float r_d = out_depth[o1 + pix], r_a = out_alpha[o1 + pix];
      const uint32_t r_nc = L.n_contrib[o1 + pix];
      // running state at the item's first entry (used only by pixels that still contribute, see below)
      const float* cs = cstate + (size_t)(has_state ? item.w : 0u) * HGS_CSTATE_FLOATS + i;
      const float r_s0 = cs[0 * 16], r_s1 = cs[1 * 16], r_s2 = cs[2 * 16], r_s3 = cs[3 * 16], r_s4 = cs[4 * 16], r_s5 = cs[5 * 16];
      // the first records go out before anything above is consumed
      gather(le_next, (uint32_t)i < cnt, c0, c1, c2);
      g0 = (inside && dL_dcolor) ? r_g0 : 0.0f; g1 = (inside && dL_dcolor) ? r_g1 : 0.0f; g2 = (inside && dL_dcolor) ? r_g2 : 0.0f;
      gd = (inside && dL_ddepth) ? r_gd : 0.0f;
      ga = (inside && dL_dalpha) ? r_ga : 0.0f;
      fp = inside ? r_c0 * g0 + r_c1 * g1 + r_c2 * g2 + r_d * gd + r_a * ga : 0.0f;
      nc = inside ? r_nc : 0u;
      T = has_state ? r_s0 : 1.0f;
      F = has_state ? r_s1 * g0 + r_s2 * g1 + r_s3 * g2 + r_s4 * gd + r_s5 * ga : 0.0f;
      // From here on F is what REMAINS behind the current record: R = F' - (w . S of everything in front of it), carried DOWN
      // from ONE subtraction per work item instead of carried up and subtracted from F' at every record: one VALU
      // instruction per record pair less (render_bwd 40.8 -> 39.5 us by stage events, same box), the same algebra.  It does
      // NOT change the accuracy (measured on the 200 extreme cameras, EXPERIMENTS.md round 6): the rounding that matters is
      // that one subtraction - F' and the stored prefix state are both O(|F'|), the remainder deep in a list is 1e-2 .. 1e-4
      // of that - and it is inherent to walking the list front to back from stored prefix states (upstream sums from the back).
      F = fp - F;
    }
    // ---- operand A of the gradient chain: the row's pixel gradients, transposed through LDS
    // (the k stage is free here: every batch of the previous group has been consumed)
    __builtin_amdgcn_wave_barrier();
    stage_k[0 * 64 + lane] = g0; stage_k[1 * 64 + lane] = g1; stage_k[2 * 64 + lane] = g2; stage_k[3 * 64 + lane] = gd;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float A3[16];
    {
      const float* ar = stage_k + qsel * 64 + 16 * (lane >> 4);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 a = *reinterpret_cast<const float4*>(ar + 4 * q4);
        A3[4 * q4 + 0] = a.x; A3[4 * q4 + 1] = a.y; A3[4 * q4 + 2] = a.z; A3[4 * q4 + 3] = a.w;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // longest row of the group, in batches
    uint32_t maxcnt = (uint32_t)__builtin_amdgcn_readlane((int)cnt, 0);
    maxcnt = max(maxcnt, (uint32_t)__builtin_amdgcn_readlane((int)cnt, 16));
    maxcnt = max(maxcnt, (uint32_t)__builtin_amdgcn_readlane((int)cnt, 32));
    maxcnt = max(maxcnt, (uint32_t)__builtin_amdgcn_readlane((int)cnt, 48));

    // Software pipeline of the record stream: indices two batches ahead, records one batch ahead.  Every load is
    // UNCONDITIONAL (a lane beyond the item's end re-reads a valid slot and is marked through its list position,
    // 0xffffffff = never active) and the index load is issued BEFORE the record gather of the same iteration: with
    // predicated loads the compiler merged the loaded registers into the loop-carried ones right behind the load
    // (s_waitcnt vmcnt(1) two instructions after the gather: the whole memory latency exposed once per batch), and
    // with the index load last, the wait for it (vmcnt is in order) also waited for the gather and the row stores.
    uint32_t pid_cur = le_next.y;                       // pair id of the lane's record of the current batch
    le_next = le_second;
    // (issued behind the first gather: the wait for that gather covers it; at the top of the group it sat in front of
    // the setup's waits and its latency was exposed)
    uint4 item_next;
    const bool have_next = fetch_item(it_full, it_part, dcap, 4u * grp_next + (uint32_t)j, n0, n1, n2, n3, item_next);
    {
      // A pixel whose last contributor (n_contrib, a tile-list position) lies before the item's first record
      // finished before this item - its forward row may have stopped without storing the state - and is never active.
      const uint32_t first = (uint32_t)__shfl((int)__float_as_uint(c2.w), lane & 48, 64);
      if (!(have && nc >= first)) { T = 1.0f; F = 0.0f; nc = 0; }      // (never active: every k is 0 whatever F holds)
    }

    float pmx = 0.f, pmy = 0.f, pqa = 0.f, pqb = 0.f, pqc = 0.f, pop = 0.f;      // the lane's record of the batch being finished
    uint32_t ppid = 0;
    // moments -> gradient sums of (this row, record i of the batch), written to the pair's slot
    auto finish = [&](const hgs_f32x4& a1, const hgs_f32x4& a2, const hgs_f32x4& a3, uint32_t it0) {
      if (it0 + (uint32_t)i >= cnt) return;
      const float a = pmx - cxq, bb = pmy - cyq;
      const float k00 = a1[0], k10 = a1[1], k01 = a1[2], k20 = a1[3], k11 = a2[0], k02 = a2[1];
      const float sdx = __builtin_fmaf(a, k00, -k10), sdy = __builtin_fmaf(bb, k00, -k01);
      const float sxx = __builtin_fmaf(a, __builtin_fmaf(a, k00, -(k10 + k10)), k20);
      const float sxy = __builtin_fmaf(a, sdy, __builtin_fmaf(-bb, k10, k11));
      const float syy = __builtin_fmaf(bb, __builtin_fmaf(bb, k00, -(k01 + k01)), k02);
      // d(p2)/d(dx) = 2 qa dx + qb dy ;  d(p2)/d(dy) = qb dx + 2 qc dy ; exp2 folding undone (d power = d p2 / log2e)
      const float x0 = __builtin_fmaf(pqa + pqa, sdx, pqb * sdy);
      const float x1 = __builtin_fmaf(pqb, sdx, (pqc + pqc) * sdy);
      const float il = 1.0f / HGS_LOG2E;
      const float opi = (pop != 0.0f) ? 1.0f / pop : 0.0f;
      float2* row = reinterpret_cast<float2*>(pair_rows + (size_t)ppid * HGS_PROW_FLOATS);      // (40 B rows: 8 B aligned)
      row[0] = make_float2(x0 * il, x1 * il);
      row[1] = make_float2(sxx * -0.5f, sxy * -1.0f);
      row[2] = make_float2(syy * -0.5f, k00 * opi);
      row[3] = make_float2(a3[0], a3[1]);
      row[4] = make_float2(a3[2], a3[3]);
    };
    for (uint32_t it0 = 0; it0 < maxcnt; it0 += HGS_RB) {
      // does any pixel of any row still contribute at or behind this batch?  (positions: lane 16 j holds the batch's first record)
      const uint32_t bfirst = (uint32_t)__shfl((int)__float_as_uint(c2.w), lane & 48, 64);
      const unsigned long long act = __ballot((it0 < cnt) && (bfirst <= nc));
#ifdef HGS_TIMELINE
      if (tl_nb == 0) tl_w1 = wall_clock64() + (act & 1ull) * 0ull;      // (the first batch's records have arrived: end of the group's start chain)
      ++tl_nb;
      tl_live += ((act & 0xffffull) != 0) + (((act >> 16) & 0xffffull) != 0) + (((act >> 32) & 0xffffull) != 0) + ((act >> 48) != 0);      // rows with work in this batch
#endif
      const float mxr = c0.x, myr = c0.y, qar = c0.z, qbr = c0.w, qcr = c1.x, opr = c1.y;   // this lane's gathered record
      if (act != 0ull) {
        __builtin_amdgcn_wave_barrier();               // the previous batch's LDS reads are done
        // PAIRED stage (as in render_fwd.hip, here every field): records 2p and 2p + 1 of the batch share six float4 -
        //   (mx0 mx1 my0 my1 | qa0 qa1 qb0 qb1 | qc0 qc1 op0 op1 | r0 r1 g0 g1 | b0 b1 d0 d1 | pos0 pos1 - -)
        // - so that the v_pk_* instructions the compiler forms over two records find their operand pairs in neighbouring
        // registers (one record per three float4: 4 v_mov per record to shuffle them together)
        float* pb = reinterpret_cast<float*>(srow + 6 * (i >> 1)) + (i & 1);
        pb[0] = c0.x; pb[2] = c0.y; pb[4] = c0.z; pb[6] = c0.w;
        pb[8] = c1.x; pb[10] = c1.y; pb[12] = c1.z; pb[14] = c1.w;
        pb[16] = c2.x; pb[18] = c2.y; pb[20] = c2.w;
      }
      // the index of the batch after next, then the next batch's records (also when this batch is skipped)
      const uint2 le_use = le_next;
      const uint32_t in2 = it0 + 2 * HGS_RB + (uint32_t)i;
      le_next = list_at(in2);
      const uint32_t pid_this = pid_cur;
      pid_cur = le_use.y;
      gather(le_use, it0 + HGS_RB + (uint32_t)i < cnt, c0, c1, c2);
      if (act == 0ull) {
        // nothing contributes any more (every pixel terminated before): zero pair rows, no evaluation
        if (it0 + (uint32_t)i < cnt) {
          float2* row = reinterpret_cast<float2*>(pair_rows + (size_t)pid_this * HGS_PROW_FLOATS);
          const float2 zero2 = make_float2(0.f, 0.f);
          row[0] = zero2; row[1] = zero2; row[2] = zero2; row[3] = zero2; row[4] = zero2;
        }
        continue;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // ---- 16 iterations in 8 pairs: everything that depends on (pixel, record); T and F are the only carried values
      // (the six LDS reads of pair k + 1 are in flight while pair k is evaluated).  Everything that is not on the T / F
      // chains is written on float2 (records 2p, 2p + 1): v_pk_add / mul / fma_f32 do two records per issue slot, and
      // the kernel is issue-bound.  The SAME IEEE operations in the same order as the scalar form (and as the forward's
      // hgs_eval_alpha: skip decisions must agree): results are bit-identical.
      auto eval_pair = [&](int u, const float4& P0, const float4& P1, const float4& P2, const float4& P3, const float4& P4,
                           const float4& P5) {
        const hgs_f32x2 mx2 = {P0.x, P0.y}, my2 = {P0.z, P0.w}, qa2 = {P1.x, P1.y}, qb2 = {P1.z, P1.w};
        const hgs_f32x2 qc2 = {P2.x, P2.y}, op2 = {P2.z, P2.w}, cr2 = {P3.x, P3.y}, cg2 = {P3.z, P3.w};
        const hgs_f32x2 cb2 = {P4.x, P4.y}, cd2 = {P4.z, P4.w};
        // same dx/dy expressions as the forward
        const hgs_f32x2 dx2 = mx2 - pxf, dy2 = my2 - pyf;
        const hgs_f32x2 m2 = __builtin_elementwise_fma(qa2, dx2, qb2 * dy2);
        const hgs_f32x2 m3 = qc2 * dy2;
        const hgs_f32x2 p2 = __builtin_elementwise_fma(dx2, m2, m3 * dy2);
        const hgs_f32x2 G2 = {__builtin_amdgcn_exp2f(p2.x), __builtin_amdgcn_exp2f(p2.y)};
        const hgs_f32x2 og2 = op2 * G2;                      // un-clamped alpha
        const bool on0 = (p2.x <= 0.0f) & (og2.x >= HGS_ALPHA_MIN) & (__float_as_uint(P5.x) <= nc);
        const bool on1 = (p2.y <= 0.0f) & (og2.y >= HGS_ALPHA_MIN) & (__float_as_uint(P5.y) <= nc);
        const hgs_f32x2 am2 = {on0 ? og2.x : 0.0f, on1 ? og2.y : 0.0f};      // 0 when inactive
        const hgs_f32x2 a2 = {fminf(HGS_ALPHA_MAX, am2.x), fminf(HGS_ALPHA_MAX, am2.y)};
        const hgs_f32x2 om2 = 1.0f - a2;
        // om >= 0.01, so dLda is always finite; inactive pixels are removed through am = 0
        const hgs_f32x2 rc2 = {__builtin_amdgcn_rcpf(om2.x), __builtin_amdgcn_rcpf(om2.y)};
        const hgs_f32x2 S2 = __builtin_elementwise_fma(cr2, (hgs_f32x2)(g0), __builtin_elementwise_fma(cg2, (hgs_f32x2)(g1),
                             __builtin_elementwise_fma(cb2, (hgs_f32x2)(g2), __builtin_elementwise_fma(cd2, (hgs_f32x2)(gd), (hgs_f32x2)(ga)))));
        // the carried chains: T and F, record 2p then 2p + 1
        const float T1 = T * om2.x;
        const hgs_f32x2 T2 = {T, T1};
        const hgs_f32x2 wgt2 = a2 * T2;
        const float F0 = __builtin_fmaf(-wgt2.x, S2.x, F);      // what remains behind record 2p
        const float F1 = __builtin_fmaf(-wgt2.y, S2.y, F0);     //                      ... 2p + 1
        const hgs_f32x2 Fn2 = {F0, F1};
        const hgs_f32x2 dl2 = __builtin_elementwise_fma(T2, S2, -(Fn2 * rc2));
        const hgs_f32x2 k2 = am2 * dl2;                      // k = dL/dG * G
        T = T1 * om2.y;
        F = F1;
        stage_k[u * HGS_STAGE_STRIDE + lane] = k2.x;
        stage_k[(u + 1) * HGS_STAGE_STRIDE + lane] = k2.y;
        stage_w[u * HGS_STAGE_STRIDE + lane] = wgt2.x;
        stage_w[(u + 1) * HGS_STAGE_STRIDE + lane] = wgt2.y;
      };
      float4 P[6], N[6];
#pragma unroll
      for (int m = 0; m < 6; ++m) P[m] = srow[m];
#pragma unroll
      for (int u = 0; u < HGS_RB; u += 2) {
        if (u + 2 < HGS_RB) {
#pragma unroll
          for (int m = 0; m < 6; ++m) N[m] = srow[3 * (u + 2) + m];
        }
        __builtin_amdgcn_sched_barrier(0x7f);       // LDS reads stay ahead of the evaluation (the scheduler would sink them to their use)
        eval_pair(u, P[0], P[1], P[2], P[3], P[4], P[5]);
        if (u + 2 < HGS_RB) {
#pragma unroll
          for (int m = 0; m < 6; ++m) P[m] = N[m];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // operand B: lane 16 kk + n reads (iteration n, pixels of row kk): 16 consecutive floats per stage
      hgs_f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
      {
        const float* sk = stage_k + i * HGS_STAGE_STRIDE + 16 * j;
        const float* sw = stage_w + i * HGS_STAGE_STRIDE + 16 * j;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 bk = *reinterpret_cast<const float4*>(sk + 4 * q4);
          const float4 bw = *reinterpret_cast<const float4*>(sw + 4 * q4);
          const float4 t1 = *reinterpret_cast<const float4*>(tabA + 4 * q4);
          const float4 t2 = *reinterpret_cast<const float4*>(tabA + 16 + 4 * q4);
          const float kx[4] = {bk.x, bk.y, bk.z, bk.w};
          const float wx[4] = {bw.x, bw.y, bw.z, bw.w};
          const float a1x[4] = {t1.x, t1.y, t1.z, t1.w};
          const float a2x[4] = {t2.x, t2.y, t2.z, t2.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1x[r], kx[r], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a2x[r], kx[r], acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(A3[4 * q4 + r], wx[r], acc3, 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();                 // the next batch overwrites the stages
      // moments -> the batch's pair rows right away: three interleaved chains of 16 x 8 cycles are over a few cycles after
      // the last issue (the "pending batch" pipeline that picked the results up one batch later dates from the 16x16x4
      // form, 1536 cycles per chain; it cost 12 AGPRs, 8 VGPRs and their copies in a kernel that is issue-bound)
      pmx = mxr; pmy = myr; pqa = qar; pqb = qbr; pqc = qcr; pop = opr; ppid = pid_this;
      finish(acc1, acc2, acc3, it0);
    }
#ifdef HGS_TIMELINE
    if (lane == 0) {
      unsigned long long* o = L.keys + ((size_t)grp * HGS_NXCD + die) * 4;
      // (physical SIMD: XCC id and the SE / SH / CU / SIMD fields of HW_ID, for the per-SIMD balance in tools/timeline.py)
      const unsigned long long simd_key = ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu) << 16) |
                                          (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xff30u);
      o[0] = tl_w0; o[1] = wall_clock64(); o[2] = tl_nb | ((tl_w1 - tl_w0) << 32); o[3] = (unsigned long long)(cnt) | (simd_key << 8) | ((unsigned long long)tl_live << 32) | 1ull << 63;
    }
#endif
    grp = grp_next; item = item_next; have = have_next;
  }
}

// ------------------------------------------------------------------------------ pair reduction, ENTRY-major rows (calls of 1-2 views)
// One gradient row per tile entry = the sum of the entry's (entry, cell) pair rows, cells in ascending order
// (deterministic).  A wave64 takes 64 consecutive entries; pair ids are entry-major, so their pair rows are ONE
// contiguous range (inside a tile): the wave streams it through LDS with fully coalesced 8 B loads, 128 rows at a
// time, and every lane (= entry) then adds its own rows from LDS in order.  (Per-thread row loads - 48 B at a
// stride of ~170 B per lane - reached 3 TB/s; at 8 views the 456 MB of pair rows made this the second-largest
// kernel of the step.)  A wave whose entries straddle tiles (pair ranges apart) streams one run per tile.
#define HGS_RED_ROWS 128
#define HGS_RED_RUNS 4      // contiguous runs a wave streams before its lanes fall back to gathering their own rows
extern "C" __global__ void __launch_bounds__(256)
hgs_k_pair_reduce_em(View v, Layout L, const hgs_status* __restrict__ status, const SortRec* __restrict__ recs_all,
                  const float* __restrict__ pair_rows, float* __restrict__ grad_rows, uint32_t pair_cap, uint32_t R_host) {
  __shared__ float2 s_rows[4][HGS_RED_ROWS * HGS_PROW_F2];
  // R_host: the entry count of a caller that holds the forward's status (hgs_backward* refuses an overflowed one on the
  // host): the kernel's first loads then do not wait for a status round trip (~1.5 us in front of every wave's chain
  // entpair -> rows); ~0: unknown, read the device copy
  uint32_t R = R_host;
  if (R_host == 0xffffffffu) {
    if (status->overflow) return;
    R = status->num_rendered;
  }
  const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6;
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p - (uint32_t)lane >= R) return;                       // (wave-uniform)
  const bool have = p < R;
  uint2 ep = make_uint2(0u, 0u);
  if (have) ep = L.entpair[p];                                // entry id | pairs << 27, first pair id
  const uint32_t entry = ep.x & 0x7ffffffu;
  // more pairs than the scratch was sized for (hgs_k_render_bwd wrote none): no row is read, the gradient rows are NaN
  const bool poisoned = (uint32_t)L.ctr->alloc_ps > pair_cap;
  const uint32_t cnt = poisoned ? 0u : ep.x >> 27;
  const uint32_t incl = hgs_wave_incl_scan(cnt), off = incl - cnt;
  float2 s0 = make_float2(0.f, 0.f), s1 = s0, s2 = s0, s3 = s0, s4 = s0;
  // RUNS of contiguous rows: inside a tile the pair ids are entry-major, so the rows of the wave's entries of ONE tile are one
  // range (an entry with pairs sits at the run's first row + the pairs of the lanes before it); a wave whose 64 entries
  // straddle a tile boundary (one in five: tiles hold ~360 entries) has two runs.  Every run is streamed the same way.
  // (The straddling waves used to take a per-thread path - a lane loading its own rows, up to 16 dependent-in-time rounds of
  // five 8 B loads: those waves WERE the kernel's length.)  More than HGS_RED_RUNS runs (tiles of a few entries): the lanes
  // that are left gather their rows themselves.
  unsigned long long todo = __ballot(cnt != 0u);
  float2* __restrict__ sl = s_rows[w];
  for (int run = 0; run < HGS_RED_RUNS && todo != 0ull; ++run) {          // (wave-uniform)
    const int first = __builtin_ctzll(todo);
    const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)(ep.y - off), first);
    const bool mine = cnt != 0u && ep.y - off == base;
    const unsigned long long in_run = __ballot(mine);
    todo &= ~in_run;
    const int lastl = 63 - __builtin_clzll(in_run);
    // rows [r_lo, r_hi) behind `base` hold the run (and, between two lanes of it, nothing else: equality with `base` is contiguity)
    const uint32_t r_lo = (uint32_t)__builtin_amdgcn_readlane((int)off, first);
    const uint32_t r_hi = (uint32_t)__builtin_amdgcn_readlane((int)incl, lastl);
    const uint32_t total = r_hi - r_lo;
    const uint32_t moff = off - r_lo, mcnt = mine ? cnt : 0u;               // this lane's rows inside the run (moff wraps for lanes outside: mcnt = 0)
    const float2* __restrict__ src = reinterpret_cast<const float2*>(pair_rows) + (size_t)(base + r_lo) * HGS_PROW_F2;
    // double buffered through registers: the loads of tile t + 1 are in flight while tile t is summed (a wave has
    // ~3 tiles; one after the other their load latency was most of the kernel's 20 us)
    static_assert(HGS_RED_ROWS * HGS_PROW_F2 == 10 * 64, "ten float2 per lane and tile");
    float2 b0, b1, b2, b3, b4, b5, b6, b7, b8, b9;   // (named registers: an array here went to scratch memory)
    // (no branch around the loads: behind the last tile every lane re-reads element 0 - one cache line; a
    // conditional block made the compiler wait for the loads where they are issued)
#define HGS_RED_ISSUE(T0)                                                                                      \
  {                                                                                                            \
    const uint32_t t0n__ = (T0);                                                                               \
    const bool any__ = t0n__ < total;                                                                          \
    const uint32_t last__ = any__ ? min((uint32_t)HGS_RED_ROWS, total - t0n__) * HGS_PROW_F2 - 1u : 0u;        \
    const float2* p__ = src + (any__ ? (size_t)t0n__ * HGS_PROW_F2 : 0);                                       \
    b0 = p__[min((uint32_t)lane, last__)];        b1 = p__[min((uint32_t)lane + 64u, last__)];                 \
    b2 = p__[min((uint32_t)lane + 128u, last__)]; b3 = p__[min((uint32_t)lane + 192u, last__)];                \
    b4 = p__[min((uint32_t)lane + 256u, last__)]; b5 = p__[min((uint32_t)lane + 320u, last__)];                \
    b6 = p__[min((uint32_t)lane + 384u, last__)]; b7 = p__[min((uint32_t)lane + 448u, last__)];                \
    b8 = p__[min((uint32_t)lane + 512u, last__)]; b9 = p__[min((uint32_t)lane + 576u, last__)];                \
  }
    HGS_RED_ISSUE(0u);
    for (uint32_t t0 = 0; t0 < total; t0 += HGS_RED_ROWS) {
      const uint32_t nrow = min((uint32_t)HGS_RED_ROWS, total - t0);
      const uint32_t nfl = nrow * HGS_PROW_F2;
      __builtin_amdgcn_wave_barrier();               // the previous tile's LDS reads are done
      if ((uint32_t)lane < nfl) sl[lane] = b0;
      if ((uint32_t)lane + 64u < nfl) sl[lane + 64] = b1;
      if ((uint32_t)lane + 128u < nfl) sl[lane + 128] = b2;
      if ((uint32_t)lane + 192u < nfl) sl[lane + 192] = b3;
      if ((uint32_t)lane + 256u < nfl) sl[lane + 256] = b4;
      if ((uint32_t)lane + 320u < nfl) sl[lane + 320] = b5;
      if ((uint32_t)lane + 384u < nfl) sl[lane + 384] = b6;
      if ((uint32_t)lane + 448u < nfl) sl[lane + 448] = b7;
      if ((uint32_t)lane + 512u < nfl) sl[lane + 512] = b8;
      if ((uint32_t)lane + 576u < nfl) sl[lane + 576] = b9;
      HGS_RED_ISSUE(t0 + HGS_RED_ROWS);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (mcnt) {
        const uint32_t r_begin = max(moff, t0), r_end = min(moff + mcnt, t0 + nrow);
        for (uint32_t r = r_begin; r < r_end; ++r) {
          const float2* q = sl + HGS_PROW_F2 * (r - t0);
          const float2 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], a4 = q[4];
          s0.x += a0.x; s0.y += a0.y; s1.x += a1.x; s1.y += a1.y; s2.x += a2.x; s2.y += a2.y;
          s3.x += a3.x; s3.y += a3.y; s4.x += a4.x; s4.y += a4.y;
        }
      }
    }
#undef HGS_RED_ISSUE
  }
  if (todo != 0ull) {
    const bool left = ((todo >> lane) & 1ull) != 0ull;
    const float2* __restrict__ rows = reinterpret_cast<const float2*>(pair_rows) + (size_t)ep.y * HGS_PROW_F2;
    for (uint32_t r = 0; r < (left ? cnt : 0u); ++r) {
      const float2* q = rows + HGS_PROW_F2 * r;
      const float2 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], a4 = q[4];
      s0.x += a0.x; s0.y += a0.y; s1.x += a1.x; s1.y += a1.y; s2.x += a2.x; s2.y += a2.y;
      s3.x += a3.x; s3.y += a3.y; s4.x += a4.x; s4.y += a4.y;
    }
  }
  if (have) {
    if (poisoned) s0.x = s0.y = s1.x = s1.y = s2.x = s2.y = s3.x = s3.y = s4.x = s4.y = __builtin_nanf("");
    float4* dst = reinterpret_cast<float4*>(grad_rows + (size_t)entry * HGS_ROW_FLOATS);
    dst[0] = make_float4(s0.x, s0.y, s1.x, s1.y); dst[1] = make_float4(s2.x, s2.y, s3.x, s3.y);
    dst[2] = make_float4(s4.x, s4.y, 0.0f, 0.0f);
  }
}

// ------------------------------------------------------------------------------ pair reduction, CHUNK-cell-major rows (calls of >= 3 views)
// With entry-major rows every 40 B row of the blend backward is an isolated partial-line store - 5x write amplification at
// the fabric, which made the batched backward bandwidth-bound (1.74 GB in 379 us for 8 views).  In calls of >= 3 views
// (View::pairchunks) the rows of every 64-record chunk of a tile list are ONE block (entpair.y = its first row), cell by
// cell, inside a cell in list order: a 16-record batch of the backward lies in one or two chunks and writes one or two
// contiguous runs of rows.  Here a wave takes the chunks that START in its window of 64 records (chunks start at tile
// starts, not at multiples of 64); lane l = record l of the chunk.  The chunk's block [first row, + its pairs) is loaded the
// way hgs_k_pair_reduce_em loads its rows - coalesced, 128 rows per step, the next step's loads in flight - and the cell
// steps read it from LDS: the rows of cell c are block positions [cp_c, cp_c + n_c), the lane with the r-th set bit of
// cell c (ballots over the masks the sort left in the records, hgs_rec_tag) takes cp_c + r.  An entry's rows ascend with
// the cell, so taking them window by window keeps the cell order: the same sums, bit for bit, as the entry-major form.
// Chunks of the long-list sort classes keep entry-major rows (tag bit 28 clear: entpair.y = the entry's first row).
// (Gathering the rows straight from HBM - a 16 B load at 40 B stride with a quarter of the lanes live - issued ten times
// the cache-line requests of this form: 31 vs 25 us per view; a cell-major layout with a slot table for the reduction,
// round 4's first answer, cost 64 B of bin buffer per entry of capacity and 25 us more per 8-view call: EXPERIMENTS.md.)
typedef float hgs_f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));      // 16 B access at 8 B alignment (40 B pair rows)
extern "C" __global__ void __launch_bounds__(256)
hgs_k_pair_reduce_ch(View v, Layout L, const hgs_status* __restrict__ status, const SortRec* __restrict__ recs_all,
                      const float* __restrict__ pair_rows, float* __restrict__ grad_rows, uint32_t pair_cap, uint32_t R_host) {
  __shared__ float2 s_rows[4][HGS_RED_ROWS * HGS_PROW_F2];
  uint32_t R = R_host;                                     // (see hgs_k_pair_reduce_em)
  if (R_host == 0xffffffffu) {
    if (status->overflow) return;
    R = status->num_rendered;
  }
  const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6;
  const uint32_t w0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 64u;      // the wave's window of records
  if (w0 >= R) return;                                                   // (wave-uniform)
  const uint32_t* __restrict__ tags = reinterpret_cast<const uint32_t*>(recs_all) + 11;      // SortRec::pad
  const uint32_t pa = w0 + (uint32_t)lane;
  const uint32_t qa = min(pa, R - 1u), qb = min(pa + 64u, R - 1u);
  const uint32_t tagA = tags[(size_t)qa * 12u], tagB = tags[(size_t)qb * 12u];
  const uint2 epA = L.entpair[qa], epB = L.entpair[qb];
  float2* __restrict__ sl = s_rows[w];
  unsigned long long starts = __ballot(pa < R && ((tagA >> 16) & 63u) == 0u);
  while (starts) {                                                        // (wave-uniform)
    const int s = (int)__builtin_ctzll(starts);
    starts &= starts - 1ull;
    const uint32_t t_s = (uint32_t)__builtin_amdgcn_readlane((int)tagA, s);
    const uint32_t C = ((t_s >> 22) & 63u) + 1u;
    const bool chunk_rows = ((t_s >> 28) & 1u) != 0u;
    const bool have = (uint32_t)lane < C;
    const int src = (s + lane) & 63;
    const bool from_b = s + lane >= 64;
    const uint32_t tA = (uint32_t)__shfl((int)tagA, src, 64), tB = (uint32_t)__shfl((int)tagB, src, 64);
    const uint32_t xA = (uint32_t)__shfl((int)epA.x, src, 64), xB = (uint32_t)__shfl((int)epB.x, src, 64);
    const uint32_t yA = (uint32_t)__shfl((int)epA.y, src, 64), yB = (uint32_t)__shfl((int)epB.y, src, 64);
    const uint32_t tag = from_b ? tB : tA, epx = from_b ? xB : xA, epy = from_b ? yB : yA;
    // (more pairs than the scratch was sized for - hgs_k_render_bwd wrote none: no row is read, the gradient rows are NaN)
    const bool poisoned = (uint32_t)L.ctr->alloc_ps > pair_cap;
    const uint32_t mask = (have && !poisoned) ? (tag & 0xffffu) : 0u;
    const uint32_t entry = epx & 0x7ffffffu, cnt = (have && !poisoned) ? (epx >> 27) : 0u;
    float2 s0 = make_float2(0.f, 0.f), s1 = s0, s2 = s0, s3 = s0, s4 = s0;
    if (chunk_rows) {
      // block positions of this lane's rows, per cell; cells' first positions and sizes (wave-uniform)
      uint32_t rel[16], cp0[16], ncell[16];
      uint32_t total = 0;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const unsigned long long bal = __ballot(((mask >> c) & 1u) != 0u);
        rel[c] = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        cp0[c] = total;
        ncell[c] = (uint32_t)__popcll(bal);
        total += ncell[c];
      }
      const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)epy, 0);       // the chunk's first row (the same in all its lanes)
      const float2* __restrict__ srcp = reinterpret_cast<const float2*>(pair_rows) + (size_t)base * HGS_PROW_F2;
      static_assert(HGS_RED_ROWS * HGS_PROW_F2 == 10 * 64, "ten float2 per lane and window");
      float2 b0, b1, b2, b3, b4, b5, b6, b7, b8, b9;
#define HGS_REDL_ISSUE(T0)                                                                                     \
  {                                                                                                            \
    const uint32_t t0n__ = (T0);                                                                               \
    const bool any__ = t0n__ < total;                                                                          \
    const uint32_t last__ = any__ ? min((uint32_t)HGS_RED_ROWS, total - t0n__) * HGS_PROW_F2 - 1u : 0u;        \
    const float2* p__ = srcp + (any__ ? (size_t)t0n__ * HGS_PROW_F2 : 0);                                      \
    b0 = p__[min((uint32_t)lane, last__)];        b1 = p__[min((uint32_t)lane + 64u, last__)];                 \
    b2 = p__[min((uint32_t)lane + 128u, last__)]; b3 = p__[min((uint32_t)lane + 192u, last__)];                \
    b4 = p__[min((uint32_t)lane + 256u, last__)]; b5 = p__[min((uint32_t)lane + 320u, last__)];                \
    b6 = p__[min((uint32_t)lane + 384u, last__)]; b7 = p__[min((uint32_t)lane + 448u, last__)];                \
    b8 = p__[min((uint32_t)lane + 512u, last__)]; b9 = p__[min((uint32_t)lane + 576u, last__)];                \
  }
      if (total) {
        HGS_REDL_ISSUE(0u);
        for (uint32_t t0 = 0; t0 < total; t0 += HGS_RED_ROWS) {
          const uint32_t nrow = min((uint32_t)HGS_RED_ROWS, total - t0);
          const uint32_t nfl = nrow * HGS_PROW_F2;
          __builtin_amdgcn_wave_barrier();             // the previous window's LDS reads are done
          if ((uint32_t)lane < nfl) sl[lane] = b0;
          if ((uint32_t)lane + 64u < nfl) sl[lane + 64] = b1;
          if ((uint32_t)lane + 128u < nfl) sl[lane + 128] = b2;
          if ((uint32_t)lane + 192u < nfl) sl[lane + 192] = b3;
          if ((uint32_t)lane + 256u < nfl) sl[lane + 256] = b4;
          if ((uint32_t)lane + 320u < nfl) sl[lane + 320] = b5;
          if ((uint32_t)lane + 384u < nfl) sl[lane + 384] = b6;
          if ((uint32_t)lane + 448u < nfl) sl[lane + 448] = b7;
          if ((uint32_t)lane + 512u < nfl) sl[lane + 512] = b8;
          if ((uint32_t)lane + 576u < nfl) sl[lane + 576] = b9;
          HGS_REDL_ISSUE(t0 + HGS_RED_ROWS);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            if (cp0[c] < t0 + nrow && cp0[c] + ncell[c] > t0) {          // (wave-uniform: the cell has rows in this window)
              const uint32_t r = rel[c] - t0;                            // (wraps for rows of earlier windows: the range test drops them)
              if (((mask >> c) & 1u) && r < nrow) {
                const float2* q = sl + HGS_PROW_F2 * r;
                const float2 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], a4 = q[4];
                s0.x += a0.x; s0.y += a0.y; s1.x += a1.x; s1.y += a1.y; s2.x += a2.x; s2.y += a2.y;
                s3.x += a3.x; s3.y += a3.y; s4.x += a4.x; s4.y += a4.y;
              }
            }
          }
        }
      }
#undef HGS_REDL_ISSUE
    } else {
      // chunks of the long-list sort classes: entry-major rows, gathered (16 + 16 + 8 B loads at 8 B alignment)
      for (uint32_t j = 0; j < 16u; ++j) {
        const bool on = j < cnt;
        if (__ballot(on) == 0ull) break;                                  // (wave-uniform)
        const float* qq = pair_rows + (size_t)(on ? epy + j : 0u) * HGS_PROW_FLOATS;
        const hgs_f32x4_a8 a0 = *reinterpret_cast<const hgs_f32x4_a8*>(qq);
        const hgs_f32x4_a8 a1 = *reinterpret_cast<const hgs_f32x4_a8*>(qq + 4);
        const float2 a2 = *reinterpret_cast<const float2*>(qq + 8);
        if (on) {
          s0.x += a0[0]; s0.y += a0[1]; s1.x += a0[2]; s1.y += a0[3]; s2.x += a1[0]; s2.y += a1[1];
          s3.x += a1[2]; s3.y += a1[3]; s4.x += a2.x; s4.y += a2.y;
        }
      }
    }
    if (have) {
      if (poisoned) s0.x = s0.y = s1.x = s1.y = s2.x = s2.y = s3.x = s3.y = s4.x = s4.y = __builtin_nanf("");
      float4* dst = reinterpret_cast<float4*>(grad_rows + (size_t)entry * HGS_ROW_FLOATS);
      dst[0] = make_float4(s0.x, s0.y, s1.x, s1.y); dst[1] = make_float4(s2.x, s2.y, s3.x, s3.y);
      dst[2] = make_float4(s4.x, s4.y, 0.0f, 0.0f);
    }
  }
}
