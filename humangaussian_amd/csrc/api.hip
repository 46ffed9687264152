// api.hip - host side of the C ABI declared in include/hgs_rast.h: buffer carving and the
// launch sequences.  No allocation, no host synchronisation; the only process-wide state is a per-device cache of
// the CU count (read-only after its first use, filled under a mutex).
//
// Forward launch chain (one stream, no host round trip), for all B views of a call at once:
//   preprocess_fwd -> tiles -> fill (+ tile order) [status published] -> sort_{huge,large,lds} (+ cell lists,
//   backward work items) -> render_fwd (one workgroup per tile, heavy first)
// Backward: render_bwd (persistent waves, four cell-list segments each) -> pair_reduce -> preprocess_bwd.
#include "hgs_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <mutex>

// The forward kernels and the per-Gaussian backward are included here (one translation
// unit, SLP vectorisation on: the forward blend is latency-bound and profits from v_pk_*).
#ifdef HGS_TIMELINE
__device__ unsigned long long hgs_tl[HGS_TL_KERNELS][HGS_TL_SLOTS][4];
#endif
#include "preprocess.hip"
#include "binning.hip"
#include "render_fwd.hip"
#include "knn.hip"
#include "bookkeeping.hip"

// render_bwd.hip is a separate translation unit (different optimisation flags)
extern "C" __global__ void hgs_k_render_bwd(View, Layout, const hgs_status*, const SortRec*, const float*,
                                            const float*, const float*, const float*, const float*,
                                            const float*, const float*, float*, uint32_t);
extern "C" __global__ void hgs_k_pair_reduce_em(View, Layout, const hgs_status*, const SortRec*, const float*, float*, uint32_t, uint32_t);
extern "C" __global__ void hgs_k_pair_reduce_ch(View, Layout, const hgs_status*, const SortRec*, const float*, float*, uint32_t, uint32_t);

namespace {

// Experiment knobs (environment variables, rounds 3-4) exist only in -DHGS_KNOBS builds; the product library never reads the environment.
#ifdef HGS_KNOBS
inline int hgs_knob(const char* name, int dflt) { const char* e = getenv(name); return (e && atoi(e) > 0) ? atoi(e) : dflt; }
#else
inline int hgs_knob(const char*, int dflt) { return dflt; }
#endif

// CUs of the device `stream` belongs to (persistent grids are sized by it).  Cached per device; the stream's device,
// not the thread's current one.
inline int cu_count(hipStream_t stream) {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipStreamGetDevice(stream, &dev) != hipSuccess) {
    if (hipGetDevice(&dev) != hipSuccess) return 256;
  }
  if (dev < 0 || dev >= 64) return 256;
  int c = cache[dev].load(std::memory_order_relaxed);
  if (c > 0) return c;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  cache[dev].store(cus, std::memory_order_relaxed);
  return cus;
}
#define HGS_CHUNK_ROWS_MIN_VIEWS 3     // calls with at least this many views keep the backward's pair rows chunk-cell-major (binning.hip::hgs_put_pair)
#define HGS_PRE_BWD_VPAR_MIN_VIEWS 2   // calls with at least this many views run the per-Gaussian backward with one thread per
                                       // (Gaussian, view); fewer: one thread per Gaussian
constexpr size_t ALIGN = 256;
constexpr size_t HGS_LDS_BINS_MAX = 16384;   // T*4 bytes of LDS <= 64 KB
#define HGS_BIN_WGS_PER_VIEW_MAX 512   // (256 until round 5: at 500k Gaussians a workgroup then walked 8 chunks one after the other -
                                       //  preprocess_fwd 70 -> 54 us with 512, +2 us in `tiles` (twice the histogram rows); 100k: +-0)
#define HGS_BIN_WGS_TOTAL_MAX 1024
constexpr int HGS_MAX_BIN_WGS_PER_VIEW = HGS_BIN_WGS_PER_VIEW_MAX;
constexpr int HGS_BIN_WGS_TOTAL = HGS_BIN_WGS_TOTAL_MAX;      // binning workgroups of a batch (all views)

struct GeomCarve {
  size_t geom, tile_n, tile_start, tile_order, tile_rec, cell_info, items_part, fwd_cells,
      hist, tile_gbase, tile_count, chunk_sums, chunk_base, ctr, status, total;
};

inline int grid_dim(int pixels) { return (pixels + HGS_TILE - 1) / HGS_TILE; }

// binning workgroups per view / chunks per workgroup for a batch of B views of P Gaussians
inline void bin_shape(int B, int P, int& nblk, int& cpw, int& nwg) {
  nblk = (P + HGS_BLOCK - 1) / HGS_BLOCK;
  int want = HGS_BIN_WGS_TOTAL / (B > 0 ? B : 1);
  if (want > HGS_MAX_BIN_WGS_PER_VIEW) want = HGS_MAX_BIN_WGS_PER_VIEW;
  if (want < 1) want = 1;
  cpw = (nblk + want - 1) / want;
  if (cpw < 1) cpw = 1;
  nwg = (nblk + cpw - 1) / cpw;
}

GeomCarve carve_geom(int B, int P, int H, int W) {
  const size_t T = (size_t)grid_dim(W) * grid_dim(H);
  const size_t TT = T * (size_t)B;
  int nblk, cpw, nwg;
  bin_shape(B, P, nblk, cpw, nwg);
  const bool lds = T <= HGS_LDS_BINS_MAX;
  GeomCarve c;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = hgs_align_up(off + bytes, ALIGN); return o; };
  c.geom = take((size_t)B * P * sizeof(GeomRec));
  c.tile_n = take(TT * 4);
  c.tile_start = take(TT * 4);
  c.tile_order = take(TT * 4);
  c.tile_rec = take(TT * 16);
  c.cell_info = take(TT * 16 * sizeof(CellInfo));
  const size_t dcap = hgs_die_cells((int)TT);            // work tables are per die (Counters::sched)
  c.items_part = take(HGS_NXCD * 2 * dcap * sizeof(uint4));   // last (partial) segment of every cell list, by length class
  c.fwd_cells = take((size_t)HGS_NXCD * HGS_NFC * dcap * 4);    // non-empty cells by length class
  c.hist = take(lds ? (size_t)B * nwg * T * 4 : 0);
  c.tile_gbase = take(lds ? (size_t)HGS_ROW_GROUPS * TT * 4 : 0);
  c.tile_count = take(lds ? 0 : TT * 4);
  c.chunk_sums = take((size_t)B * nblk * 4);
  c.chunk_base = take((size_t)B * nblk * 4);
  c.ctr = take(sizeof(Counters));
  c.status = take(sizeof(hgs_status));
  c.total = off;
  return c;
}

struct BinCarve { size_t keys, recs, cell_list, entpair, cstate, items_full, total; };

// Pair-sized arrays hold HGS_PAIRS_PER_ENTRY slots per entry of capacity: an entry can reach all 16 cells of
// its tile (zoomed-in cameras), so no second capacity (and no second overflow path) exists.
BinCarve carve_bin(int64_t cap) {
  BinCarve c;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = hgs_align_up(off + bytes, ALIGN); return o; };
  const size_t C = (size_t)(cap > 0 ? cap : 0);
  const size_t NP = C * HGS_PAIRS_PER_ENTRY;
  c.keys = take(C * 8);
  c.recs = take(C * sizeof(SortRec));
  c.cell_list = take(NP * 8);
  c.entpair = take(C * 8);
  // a cell list of len entries has ceil(len / HGS_SEGLEN) - 1 stored states and ceil(len / HGS_SEGLEN) work items, len / HGS_SEGLEN of them full
  c.cstate = take((NP / HGS_SEGLEN + 1) * HGS_CSTATE_FLOATS * sizeof(float));
  c.items_full = take(HGS_NXCD * (NP / HGS_SEGLEN + 1) * sizeof(uint4));     // per die; one die's tiles may hold (nearly) all full segments
  c.total = off;
  return c;
}

Layout make_layout(void* geom, void* bin, void* img, int B, int P, int H, int W, int64_t cap) {
  const GeomCarve g = carve_geom(B, P, H, W);
  const BinCarve b = carve_bin(cap);
  char* gp = static_cast<char*>(geom);
  char* bp = static_cast<char*>(bin);
  Layout L;
  L.geom = reinterpret_cast<GeomRec*>(gp + g.geom);
  L.tile_n = reinterpret_cast<uint32_t*>(gp + g.tile_n);
  L.tile_start = reinterpret_cast<uint32_t*>(gp + g.tile_start);
  L.tile_order = reinterpret_cast<uint32_t*>(gp + g.tile_order);
  L.tile_rec = reinterpret_cast<uint4*>(gp + g.tile_rec);
  L.cell_info = reinterpret_cast<CellInfo*>(gp + g.cell_info);
  L.items_part = reinterpret_cast<uint4*>(gp + g.items_part);
  L.fwd_cells = reinterpret_cast<uint32_t*>(gp + g.fwd_cells);
  L.hist = reinterpret_cast<uint32_t*>(gp + g.hist);
  L.tile_gbase = reinterpret_cast<uint32_t*>(gp + g.tile_gbase);
  L.tile_count = reinterpret_cast<uint32_t*>(gp + g.tile_count);
  L.chunk_sums = reinterpret_cast<uint32_t*>(gp + g.chunk_sums);
  L.chunk_base = reinterpret_cast<uint32_t*>(gp + g.chunk_base);
  L.ctr = reinterpret_cast<Counters*>(gp + g.ctr);
  L.keys = bp ? reinterpret_cast<unsigned long long*>(bp + b.keys) : nullptr;
  L.recs = bp ? reinterpret_cast<SortRec*>(bp + b.recs) : nullptr;
  L.cell_list = bp ? reinterpret_cast<uint2*>(bp + b.cell_list) : nullptr;
  L.entpair = bp ? reinterpret_cast<uint2*>(bp + b.entpair) : nullptr;
  L.cstate = bp ? reinterpret_cast<float*>(bp + b.cstate) : nullptr;
  L.items_full = bp ? reinterpret_cast<uint4*>(bp + b.items_full) : nullptr;
  L.full_cap = (uint32_t)((size_t)(cap > 0 ? cap : 0) * HGS_PAIRS_PER_ENTRY / HGS_SEGLEN + 1);
  L.n_contrib = static_cast<uint32_t*>(img);
  return L;
}

View make_view(const hgs_settings* s, int B, int P, int M, int64_t cap, int max_tile_hint = 0, int act = 0) {
  View v;
  v.act = act;
  v.pairchunks = B >= HGS_CHUNK_ROWS_MIN_VIEWS ? 1 : 0;
  for (int b = 0; b < HGS_MAX_VIEWS; ++b) {
    const hgs_settings& sb = s[b < B ? b : 0];
    Cam& c = v.cam[b];
    c.viewmatrix = sb.viewmatrix;
    c.projmatrix = sb.projmatrix;
    c.campos = sb.campos;
    c.bg = sb.bg;
    c.tanfovx = sb.tanfovx;
    c.tanfovy = sb.tanfovy;
    c.focal_x = (float)sb.image_width / (2.0f * sb.tanfovx);
    c.focal_y = (float)sb.image_height / (2.0f * sb.tanfovy);
  }
  v.scale_modifier = s->scale_modifier;
  v.W = s->image_width;
  v.H = s->image_height;
  v.grid_x = grid_dim(v.W);
  v.grid_y = grid_dim(v.H);
  v.T = v.grid_x * v.grid_y;
  v.B = B;
  v.TT = B * v.T;
  v.P = P;
  v.M = M;
  v.D = s->sh_degree;
  v.lds_bins = (size_t)v.T <= HGS_LDS_BINS_MAX ? 1 : 0;
  bin_shape(B, P, v.nblk, v.cpw, v.nwg);
  if (!v.lds_bins) { v.cpw = 1; v.nwg = v.nblk; }
  v.entry_capacity = (uint32_t)(cap < 0 ? 0 : (cap > 0xffffffffll ? 0xffffffffll : cap));
  v.max_tile_hint = max_tile_hint;
  return v;
}

inline int hip_rc(hipError_t e) { return e == hipSuccess ? HGS_OK : -(1000 + (int)e); }

#define HGS_LAUNCH_CHECK()                       \
  do {                                           \
    hipError_t e__ = hipGetLastError();          \
    if (e__ != hipSuccess) return hip_rc(e__);   \
  } while (0)

#define HGS_STAGE(k)                                                              \
  do {                                                                            \
    if (stage_events && stage_events[k]) {                                        \
      hipError_t e__ = hipEventRecord(static_cast<hipEvent_t>(stage_events[k]), stream); \
      if (e__ != hipSuccess) return hip_rc(e__);                                  \
    }                                                                             \
  } while (0)

// The long-list sort classes (one 1024-thread workgroup per tile of more than 4 096 entries: hgs_k_sort_large / _huge) run
// BESIDE the LDS class on a per-device side stream - fork behind `fill`, join in front of the blend.  A view of 500k
// Gaussians has one or two such tiles; launched in front of hgs_k_sort_lds on the caller's stream they held the whole
// GPU for the 39 us one of them takes (8 % of that step).  The classes write disjoint tiles and share only the bump
// allocators (atomics), so they may overlap.  The side stream and its two events are created once per device; a device
// where that fails keeps the launches on the caller's stream.  Calls whose hint rules the classes out never touch it.
struct SideStream {
  std::mutex mu;
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  bool tried = false, ok = false;
};
SideStream* side_stream_for(hipStream_t stream) {
  static SideStream tab[64];
  int dev = 0;
  if (hipStreamGetDevice(stream, &dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return nullptr; }
  SideStream& t = tab[dev];
  std::lock_guard<std::mutex> lk(t.mu);
  if (!t.tried) {
    t.tried = true;
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    if (have_cur && cur != dev) (void)hipSetDevice(dev);
    t.ok = hipStreamCreateWithFlags(&t.s, hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&t.join, hipEventDisableTiming) == hipSuccess;
    if (have_cur && cur != dev) (void)hipSetDevice(cur);
    if (!t.ok) (void)hipGetLastError();
  }
  return t.ok ? &t : nullptr;
}

bool settings_ok(const hgs_settings* s) {
  return s && s->image_height > 0 && s->image_width > 0 && s->bg && s->viewmatrix &&
         s->projmatrix && s->campos && s->sh_degree >= 0 && s->sh_degree <= 3 &&
         s->image_width <= 16 * 65535 && s->image_height <= 16 * 65535;
}

// a batch: 1..HGS_MAX_VIEWS valid settings that agree in everything that is not per camera
bool batch_ok(const hgs_settings* s, int B) {
  if (!s || B < 1 || B > HGS_MAX_VIEWS) return false;
  for (int b = 0; b < B; ++b) {
    if (!settings_ok(s + b)) return false;
    if (s[b].image_height != s[0].image_height || s[b].image_width != s[0].image_width ||
        s[b].sh_degree != s[0].sh_degree || s[b].scale_modifier != s[0].scale_modifier)
      return false;
  }
  return (size_t)B * grid_dim(s->image_width) * grid_dim(s->image_height) < (1u << 30);
}

}  // namespace

#ifdef HGS_TIMELINE
// debug builds only: copy one kernel's timeline table to the host (synchronous) and clear it
extern "C" int hgs_debug_timeline_read(int kernel_id, void* host_dst) {
  if (kernel_id < 0 || kernel_id >= HGS_TL_KERNELS) return -1;
  const size_t bytes = sizeof(unsigned long long) * HGS_TL_SLOTS * 4, off = bytes * (size_t)kernel_id;
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(hgs_tl), bytes, off, hipMemcpyDeviceToHost) != hipSuccess) return -3;
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(hgs_tl)) != hipSuccess) return -4;
  return hipMemset(static_cast<char*>(p) + off, 0, bytes) == hipSuccess ? 0 : -5;
}
#endif

extern "C" {

int hgs_knn_mean_dist2(int32_t P, const float* points, float* mean_dist2, void* stream_) {
  if (P < 0) return HGS_EINVAL;
  if (P == 0) return HGS_OK;
  if (!points || !mean_dist2) return HGS_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(hgs_k_knn3, dim3((P + 255) / 256), dim3(256), 0, stream, (int)P, points, mean_dist2,
                     static_cast<const KnnGrid*>(nullptr));
  HGS_LAUNCH_CHECK();
  return HGS_OK;
}

namespace {
struct KnnCarve { size_t grid, cell_of, count, cursor, bsum, sorted, total; uint32_t nc_max; };
KnnCarve carve_knn(int32_t P) {
  KnnCarve c;
  const size_t n = (size_t)(P > 0 ? P : 0);
  const size_t want = std::max<size_t>(64, 2 * n);
  c.nc_max = (uint32_t)std::min<size_t>(want, HGS_KNN_MAX_CELLS);
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = hgs_align_up(off + bytes, ALIGN); return o; };
  c.grid = take(sizeof(KnnGrid));
  c.cell_of = take(n * 4);
  c.count = take(((size_t)c.nc_max + 1) * 4);
  c.cursor = take((size_t)c.nc_max * 4);
  c.bsum = take(((size_t)c.nc_max / 1024 + 2) * 4);
  c.sorted = take(n * 16);
  c.total = off;
  return c;
}
}  // namespace

size_t hgs_knn_scratch_bytes(int32_t P) { return carve_knn(P).total; }

int hgs_knn_mean_dist2_grid(int32_t P, const float* points, float* mean_dist2, void* scratch, void* stream_) {
  if (P < 0) return HGS_EINVAL;
  if (P == 0) return HGS_OK;
  if (!points || !mean_dist2 || !scratch) return HGS_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const KnnCarve c = carve_knn(P);
  char* sp = static_cast<char*>(scratch);
  KnnGrid* G = reinterpret_cast<KnnGrid*>(sp + c.grid);
  uint32_t* cell_of = reinterpret_cast<uint32_t*>(sp + c.cell_of);
  uint32_t* count = reinterpret_cast<uint32_t*>(sp + c.count);
  uint32_t* cursor = reinterpret_cast<uint32_t*>(sp + c.cursor);
  uint32_t* bsum = reinterpret_cast<uint32_t*>(sp + c.bsum);
  float4* sorted = reinterpret_cast<float4*>(sp + c.sorted);
  // box images: min = all ones, max = zero; cell counters zero
  hipError_t e = hipMemsetAsync(&G->bmin[0], 0xff, 12, stream);
  if (e == hipSuccess) e = hipMemsetAsync(&G->bmax[0], 0, 12, stream);
  if (e == hipSuccess) e = hipMemsetAsync(count, 0, ((size_t)c.nc_max + 1) * 4, stream);
  if (e != hipSuccess) return hip_rc(e);
  const unsigned gp = (unsigned)((P + 255) / 256), gc = (unsigned)((c.nc_max + 1023u) / 1024u);
  hipLaunchKernelGGL(hgs_k_knn_bbox, dim3(gp), dim3(256), 0, stream, (int)P, points, G);
  hipLaunchKernelGGL(hgs_k_knn_grid_setup, dim3(1), dim3(64), 0, stream, (int)P, c.nc_max, G);
  hipLaunchKernelGGL(hgs_k_knn_count, dim3(gp), dim3(256), 0, stream, (int)P, points, (const KnnGrid*)G, cell_of, count);
  hipLaunchKernelGGL(hgs_k_knn_scan1, dim3(gc), dim3(1024), 0, stream, (const KnnGrid*)G, (const uint32_t*)count, bsum);
  hipLaunchKernelGGL(hgs_k_knn_scan2, dim3(1), dim3(1024), 0, stream, G, bsum);
  hipLaunchKernelGGL(hgs_k_knn_scan3, dim3(gc), dim3(1024), 0, stream, (int)P, G, count, cursor, (const uint32_t*)bsum);
  hipLaunchKernelGGL(hgs_k_knn_scatter, dim3(gp), dim3(256), 0, stream, (int)P, points, (const uint32_t*)cell_of, cursor, sorted);
  hipLaunchKernelGGL(hgs_k_knn_search, dim3(gp), dim3(256), 0, stream, (int)P, (const KnnGrid*)G, (const uint32_t*)count,
                     (const float4*)sorted, mean_dist2);
  // degenerate clouds (decided on the device: KnnGrid::brute): the exact brute force; returns at once otherwise
  hipLaunchKernelGGL(hgs_k_knn3, dim3(gp), dim3(256), 0, stream, (int)P, points, mean_dist2, (const KnnGrid*)G);
  HGS_LAUNCH_CHECK();
  return HGS_OK;
}

int hgs_reduce_view_packs_acc(int32_t world, int64_t P, int32_t F, const float* gathered, const float* acc_in,
                              float* out, void* stream_) {
  if (world < 1 || P < 0 || F < 1) return HGS_EINVAL;
  if (P == 0) return HGS_OK;
  if (!gathered || !out) return HGS_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const long long n = (long long)P * F;
  hipLaunchKernelGGL(hgs_k_reduce_view_packs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                     (int)world, n, (int)F, gathered, acc_in, out);
  HGS_LAUNCH_CHECK();
  return HGS_OK;
}

int hgs_reduce_view_packs_unpack(int32_t world, int64_t P, int32_t M, const float* gathered, const float* acc_in,
                                 float* g_means3D, float* g_means2D, float* g_sh, float* g_opac, float* g_scales,
                                 float* g_rot, int32_t* radii, void* stream_) {
  if (world < 1 || P < 0 || M < 0) return HGS_EINVAL;
  if (P == 0) return HGS_OK;
  if (!gathered || !g_means3D || !g_means2D || (M > 0 && !g_sh) || !g_opac || !g_scales || !g_rot || !radii) return HGS_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int F = 15 + 3 * M;
  const long long n = (long long)P * F;
  hipLaunchKernelGGL(hgs_k_reduce_view_packs_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                     (int)world, n, (int)F, (int)M, gathered, acc_in, g_means3D, g_means2D, g_sh, g_opac, g_scales, g_rot, radii);
  HGS_LAUNCH_CHECK();
  return HGS_OK;
}

int hgs_reduce_view_packs(int32_t world, int64_t P, int32_t F, const float* gathered, float* out,
                          void* stream_) {
  return hgs_reduce_view_packs_acc(world, P, F, gathered, nullptr, out, stream_);
}

int hgs_pack_view_contribution(int32_t P, int32_t M, const float* g_means3D, const float* g_means2D,
                               const float* g_sh, const float* g_opac, const float* g_scales,
                               const float* g_rot, const int32_t* radii, float* out, void* stream_) {
  if (P < 0 || M < 0) return HGS_EINVAL;
  if (P == 0) return HGS_OK;
  if (!g_means3D || !g_means2D || (M > 0 && !g_sh) || !g_opac || !g_scales || !g_rot || !radii || !out)
    return HGS_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const long long n = (long long)P * (15 + 3 * M);
  hipLaunchKernelGGL(hgs_k_pack_view_contribution, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                     (int)P, (int)M, g_means3D, g_means2D, g_sh, g_opac, g_scales, g_rot, radii, out);
  HGS_LAUNCH_CHECK();
  return HGS_OK;
}

int hgs_abi_version(void) { return 16; }

size_t hgs_geom_bytes_batch(int32_t B, int32_t P, int32_t H, int32_t W) {
  if (B < 1 || B > HGS_MAX_VIEWS || P < 0 || H <= 0 || W <= 0) return 0;
  return carve_geom(B, P, H, W).total;
}
size_t hgs_geom_bytes(int32_t P, int32_t H, int32_t W) { return hgs_geom_bytes_batch(1, P, H, W); }
size_t hgs_bin_bytes(int64_t entry_capacity) { return carve_bin(entry_capacity).total; }
size_t hgs_img_bytes_batch(int32_t B, int32_t H, int32_t W) {
  if (B < 1 || B > HGS_MAX_VIEWS || H <= 0 || W <= 0) return 0;
  return hgs_align_up((size_t)B * H * W * 4, ALIGN);
}
size_t hgs_img_bytes(int32_t H, int32_t W) { return hgs_img_bytes_batch(1, H, W); }
// one gradient row per entry + one per (entry, cell) pair
size_t hgs_bwd_scratch_bytes(int64_t R) {
  return hgs_align_up((size_t)(R > 0 ? R : 0) * (HGS_ROW_FLOATS + HGS_PAIRS_PER_ENTRY * HGS_PROW_FLOATS) * sizeof(float), ALIGN);
}
size_t hgs_bwd_scratch_bytes_pairs(int64_t R, int64_t pairs) {
  if (R <= 0) return 0;
  if (pairs <= 0 || pairs > R * HGS_PAIRS_PER_ENTRY) return hgs_bwd_scratch_bytes(R);
  return hgs_align_up(((size_t)R * HGS_ROW_FLOATS + (size_t)pairs * HGS_PROW_FLOATS) * sizeof(float), ALIGN);
}

int hgs_forward_batch_act(const hgs_settings* s, int32_t B, int32_t P, int32_t M, const float* means3D,
                          const float* shs, const float* colors_precomp, const float* opacities,
                          const float* scales, const float* rotations, const float* cov3D_precomp,
                          float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                          void* geom, void* bin, int64_t entry_capacity, void* img,
                          int32_t store_bwd_state, int32_t max_tile_entries_hint, hgs_status* status_host,
                          int32_t status_host_mapped, void* status_event, void* const* stage_events,
                          int32_t activation_flags, void* stream_) {
  return hgs_forward_batch_act_leaf(s, B, P, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                    out_color, out_depth, out_alpha, radii, geom, bin, entry_capacity, img, store_bwd_state,
                                    max_tile_entries_hint, status_host, status_host_mapped, status_event, stage_events,
                                    activation_flags, nullptr, stream_);
}

int hgs_forward_batch_act_leaf(const hgs_settings* s, int32_t B, int32_t P, int32_t M, const float* means3D,
                               const float* shs, const float* colors_precomp, const float* opacities,
                               const float* scales, const float* rotations, const float* cov3D_precomp,
                               float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                               void* geom, void* bin, int64_t entry_capacity, void* img,
                               int32_t store_bwd_state, int32_t max_tile_entries_hint, hgs_status* status_host,
                               int32_t status_host_mapped, void* status_event, void* const* stage_events,
                               int32_t activation_flags, float* means2D_leaf, void* stream_) {
  if (activation_flags & ~(7 | HGS_GRAD_SCALE_TRUE_DERIVATIVE)) return HGS_EINVAL;      // (the gradient bit is the backward's: ignored here)
  if (!batch_ok(s, B) || P < 0 || !out_color || !out_depth || !out_alpha || !geom || !img ||
      entry_capacity < 0)
    return HGS_EINVAL;
  if (P > 0) {
    if (!means3D || !opacities || !radii) return HGS_EINVAL;
    if ((shs != nullptr) == (colors_precomp != nullptr)) return HGS_ESHAPE;
    const bool has_sr = scales != nullptr && rotations != nullptr;
    if ((scales != nullptr) != (rotations != nullptr)) return HGS_ESHAPE;
    if (has_sr == (cov3D_precomp != nullptr)) return HGS_ESHAPE;
    if (shs && M < (s->sh_degree + 1) * (s->sh_degree + 1)) return HGS_ESHAPE;
    if (entry_capacity > 0 && !bin) return HGS_EINVAL;
    if ((int64_t)B * P >= (1ll << 31) || P >= (1 << 28)) return HGS_EINVAL;
  }
  // entry ids travel in 27 bits (entpair.x = entry | pairs << 27): a larger list cannot be addressed
  if (entry_capacity > HGS_MAX_ENTRY_CAPACITY) return HGS_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int ncu = cu_count(stream);
  // (the gradient bit belongs to the backward: masked out, so that no forward kernel can ever branch on it)
  const View v = make_view(s, B, P, M, entry_capacity, max_tile_entries_hint > 0 ? max_tile_entries_hint : 0,
                           activation_flags & 7);
  const Layout L = make_layout(geom, bin, img, B, P, v.H, v.W, entry_capacity);
  hgs_status* status_dev =
      reinterpret_cast<hgs_status*>(static_cast<char*>(geom) + carve_geom(B, P, v.H, v.W).status);

  HGS_STAGE(0);
  hipError_t e = hipSuccess;
  if (v.nblk == 0) {       // no preprocess launch (P == 0): nobody else zeroes the counters
    e = hipMemsetAsync(L.ctr, 0, sizeof(Counters), stream);
    if (e != hipSuccess) return hip_rc(e);
  }
  const size_t lds_bytes = (size_t)v.T * 4;
  if (!v.lds_bins) {
    e = hipMemsetAsync(L.tile_count, 0, (size_t)v.TT * 4, stream);
    if (e != hipSuccess) return hip_rc(e);
  }
  if (v.nblk > 0) {
    if (v.lds_bins)
      hipLaunchKernelGGL(hgs_k_preprocess_fwd, dim3(v.B * v.nwg), dim3(HGS_BLOCK), lds_bytes, stream, v,
                         L, means3D, shs, colors_precomp, opacities, scales, rotations,
                         cov3D_precomp, radii, means2D_leaf);
    else
      hipLaunchKernelGGL(hgs_k_preprocess_fwd_ga, dim3(v.B * v.nblk), dim3(HGS_BLOCK), 0, stream, v, L,
                         means3D, shs, colors_precomp, opacities, scales, rotations,
                         cov3D_precomp, radii, means2D_leaf);
    HGS_LAUNCH_CHECK();
  }
  HGS_STAGE(1);
  // tile tables: 64 tiles per workgroup, all views in one launch
  const unsigned bpv = (unsigned)((v.T + HGS_TILES_PER_WG - 1) / HGS_TILES_PER_WG);
  hipLaunchKernelGGL(hgs_k_tiles, dim3(bpv * (unsigned)v.B), dim3(64 * HGS_ROW_GROUPS), 0, stream, v, L);
  HGS_LAUNCH_CHECK();
  HGS_STAGE(2);
  // binning workgroups scatter the keys; `order_wgs` more place the tiles into tile_order, hand out
  // the chunks' entry-id bases and publish the status (device copy, pinned host mirror)
  const unsigned order_wgs = (unsigned)((v.TT + HGS_BLOCK - 1) / HGS_BLOCK);
  hgs_status* status_mapped = status_host_mapped ? status_host : nullptr;
  const unsigned nbin = entry_capacity > 0 ? (unsigned)(v.B * (v.lds_bins ? v.nwg : v.nblk)) : 0u;
  if (v.lds_bins && nbin)
    hipLaunchKernelGGL(hgs_k_fill, dim3(nbin + order_wgs), dim3(HGS_BLOCK), lds_bytes, stream, v, L, status_dev,
                       status_mapped, (int)nbin);
  else
    hipLaunchKernelGGL(hgs_k_fill_ga, dim3(nbin + order_wgs), dim3(HGS_BLOCK), 0, stream, v, L, status_dev,
                       status_mapped, (int)nbin);
  HGS_LAUNCH_CHECK();
  // the status is final here: publish it now so the host can wait for it alone
  if (status_host && !status_host_mapped) {
    e = hipMemcpyAsync(status_host, status_dev, sizeof(hgs_status), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return hip_rc(e);
  }
  if (status_event) {
    e = hipEventRecord(static_cast<hipEvent_t>(status_event), stream);
    if (e != hipSuccess) return hip_rc(e);
  }
  if (entry_capacity > 0) {
    HGS_STAGE(3);
    // tiles are ordered heavy-first, so a class with more than LO entries per tile can only
    // occupy the first capacity/LO positions of tile_order
    auto class_grid = [&](int64_t lo) {
      const int64_t g = entry_capacity / lo + 1;
      return (unsigned)(g < v.TT ? g : v.TT);
    };
    // the caller's hint (longest tile list it has seen, with margin) lets us skip launching
    // sort classes that cannot occur; a wrong hint is caught on the device (overflow bit 2).
    const int hint = v.max_tile_hint;
    const bool need_huge = hint <= 0 || hint > 16384, need_large = hint <= 0 || hint > HGS_SORT_LDS_MAX;
    // (the side stream costs a cross-stream edge, ~11 us at the join: it is taken when a long list is EXPECTED - a hint
    // beyond 1.5 x the class boundary, i.e. a list the caller has seen, not the margin on a shorter one - or unknown)
    const bool expect_long = hint <= 0 || hint > HGS_SORT_LDS_MAX + HGS_SORT_LDS_MAX / 2 + 64;
    SideStream* side = ((need_huge || need_large) && expect_long) ? side_stream_for(stream) : nullptr;
    {
      std::unique_lock<std::mutex> side_lk;
      hipStream_t s2 = stream;
      if (side) {                                        // fork: the long-list classes wait for `fill`, nothing else
        side_lk = std::unique_lock<std::mutex>(side->mu);
        e = hipEventRecord(side->fork, stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(side->s, side->fork, 0);
        if (e != hipSuccess) return hip_rc(e);
        s2 = side->s;
      }
      if (need_huge) {
        hipLaunchKernelGGL(hgs_k_sort_huge, dim3(class_grid(16384)), dim3(1024), 0, s2, v, L, status_dev);
        HGS_LAUNCH_CHECK();
      }
      if (need_large) {
        hipLaunchKernelGGL(hgs_k_sort_large, dim3(class_grid(HGS_SORT_LDS_MAX)), dim3(1024), 0, s2, v, L, status_dev);
        HGS_LAUNCH_CHECK();
      }
      if (side) {
        e = hipEventRecord(side->join, side->s);
        if (e != hipSuccess) return hip_rc(e);
      }
      // persistent workgroups (53 KB of LDS: three per CU), tiles heavy first round-robin
      const unsigned sort_wgs = std::min<unsigned>(class_grid(1), (unsigned)(hgs_knob("HGS_SORT_WGS_PER_CU", 3) * ncu));
      if (v.pairchunks)
        hipLaunchKernelGGL(hgs_k_sort_lds_ch, dim3(sort_wgs), dim3(HGS_SORT_NT), 0, stream, v, L, status_dev);
      else
        hipLaunchKernelGGL(hgs_k_sort_lds, dim3(sort_wgs), dim3(HGS_SORT_NT), 0, stream, v, L, status_dev);
      HGS_LAUNCH_CHECK();
      if (side) {                                        // join: the blend needs every class
        e = hipStreamWaitEvent(stream, side->join, 0);
        if (e != hipSuccess) return hip_rc(e);
      }
    }
  } else {
    HGS_STAGE(3);
  }
  HGS_STAGE(4);
  // Blend: waves of four cells of one length class, longest first, then the background of the empty cells
  // (render_fwd.hip).  Non-empty cells <= min(16 B T, pairs): a capacity bound, surplus waves leave at once.
  {
    // persistent cell waves: enough to fill the chip (4 waves per block; 4 blocks per CU), never more than the cells
    const int64_t cells = std::min<int64_t>((int64_t)16 * v.TT, (int64_t)HGS_PAIRS_PER_ENTRY * entry_capacity);
    // (the snake schedule of hgs_k_render_fwd visits every item only when the block count is a multiple of 4)
    const int64_t fwd_blocks = (int64_t)hgs_knob("HGS_FWD_BLOCKS_PER_CU", 4) * ncu;
    const unsigned cell_blocks = (unsigned)std::max<int64_t>(4, std::min<int64_t>(fwd_blocks, (cells + 3) / 4) & ~int64_t(3));
    const unsigned bg_blocks = (unsigned)v.TT;
    if (store_bwd_state)
      hipLaunchKernelGGL(hgs_k_render_fwd_store, dim3(cell_blocks + bg_blocks), dim3(HGS_FWD_THREADS), 0, stream, v, L, cell_blocks,
                         status_dev, status_mapped, L.recs, L.cstate, out_color, out_depth, out_alpha);
    else
      hipLaunchKernelGGL(hgs_k_render_fwd_nostore, dim3(cell_blocks + bg_blocks), dim3(HGS_FWD_THREADS), 0, stream, v, L, cell_blocks,
                         status_dev, status_mapped, L.recs, L.cstate, out_color, out_depth, out_alpha);
  }
  HGS_LAUNCH_CHECK();
  HGS_STAGE(5);
  return HGS_OK;
}

int hgs_forward_batch(const hgs_settings* s, int32_t B, int32_t P, int32_t M, const float* means3D,
                      const float* shs, const float* colors_precomp, const float* opacities,
                      const float* scales, const float* rotations, const float* cov3D_precomp,
                      float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                      void* geom, void* bin, int64_t entry_capacity, void* img,
                      int32_t store_bwd_state, int32_t max_tile_entries_hint, hgs_status* status_host,
                      int32_t status_host_mapped, void* status_event, void* const* stage_events,
                      void* stream_) {
  return hgs_forward_batch_act(s, B, P, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                               out_color, out_depth, out_alpha, radii, geom, bin, entry_capacity, img, store_bwd_state,
                               max_tile_entries_hint, status_host, status_host_mapped, status_event, stage_events, 0,
                               stream_);
}

int hgs_forward(const hgs_settings* s, int32_t P, int32_t M, const float* means3D,
                const float* shs, const float* colors_precomp, const float* opacities,
                const float* scales, const float* rotations, const float* cov3D_precomp,
                float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                void* geom, void* bin, int64_t entry_capacity, void* img,
                int32_t store_bwd_state, int32_t max_tile_entries_hint, hgs_status* status_host,
                int32_t status_host_mapped, void* status_event, void* const* stage_events,
                void* stream_) {
  return hgs_forward_batch(s, 1, P, M, means3D, shs, colors_precomp, opacities, scales, rotations,
                           cov3D_precomp, out_color, out_depth, out_alpha, radii, geom, bin, entry_capacity,
                           img, store_bwd_state, max_tile_entries_hint, status_host, status_host_mapped,
                           status_event, stage_events, stream_);
}

}  // extern "C"

namespace {
// the backward of a batch; `pack` != nullptr: hgs_backward_batch_packed (one (P, pack_F) row-major pack instead of the six
// parameter-gradient tensors; dL_dshs / dL_dscales / dL_drotations are then non-null MARKERS for the parts the row carries)
int backward_impl(const hgs_settings* s, int32_t B, int32_t P, int32_t M, const float* means3D,
                           const float* shs, const float* colors_precomp, const float* opacities,
                           const float* scales, const float* rotations, const float* cov3D_precomp,
                           const int32_t* radii, const float* out_color, const float* out_depth,
                           const float* out_alpha, const float* dL_dout_color,
                           const float* dL_dout_depth, const float* dL_dout_alpha, const void* geom,
                           const void* bin, const void* img, const hgs_status* status,
                           int64_t entry_capacity, void* bwd_scratch, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs,
                           float* dL_dcolors_precomp, float* dL_dopacities, float* dL_dscales,
                           float* dL_drotations, float* dL_dcov3D_precomp, void* const* stage_events,
                           int32_t activation_flags, float* pack, int32_t pack_F, void* stream_) {
  (void)radii;
  if (activation_flags & ~(7 | HGS_GRAD_SCALE_TRUE_DERIVATIVE)) return HGS_EINVAL;
  if ((activation_flags & HGS_ACT_OPACITY_SIGMOID) && P > 0 && !opacities) return HGS_EINVAL;
  if (!batch_ok(s, B) || P < 0 || !geom || !img || entry_capacity < 0 || entry_capacity > HGS_MAX_ENTRY_CAPACITY) return HGS_EINVAL;
  if (status && status->overflow) return HGS_EINVAL;
  if (status && (int64_t)status->reserved[0] != entry_capacity) return HGS_EINVAL;
  if (P == 0) return HGS_OK;
  if (!means3D || !out_color || !out_depth || !out_alpha) return HGS_EINVAL;
  if ((shs != nullptr) == (colors_precomp != nullptr)) return HGS_ESHAPE;
  if ((scales != nullptr) != (rotations != nullptr)) return HGS_ESHAPE;
  if ((scales != nullptr) == (cov3D_precomp != nullptr)) return HGS_ESHAPE;
  if (shs && !dL_dshs) return HGS_EINVAL;
  const bool maybe_entries = status ? status->num_rendered > 0 : entry_capacity > 0;
  if (maybe_entries && (!bin || !bwd_scratch)) return HGS_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int64_t cap = entry_capacity;
  const View v = make_view(s, B, P, M, cap, 0, activation_flags);
  const Layout L = make_layout(const_cast<void*>(geom), const_cast<void*>(bin),
                               const_cast<void*>(img), B, P, v.H, v.W, cap);
  const hgs_status* status_dev = reinterpret_cast<const hgs_status*>(
      static_cast<const char*>(geom) + carve_geom(B, P, v.H, v.W).status);
  // gradient rows: [X entries][12] then the pair rows [pair_cap][10]; X = what the caller sized the scratch by, pair_cap =
  // status->num_pairs (hgs_bwd_scratch_bytes_pairs) or the worst case of 16 per entry (hgs_bwd_scratch_bytes).  The kernels
  // compare pair_cap with the count the sort left on the device: a stale / wrong count writes nothing out of bounds and
  // turns the gradients into NaN instead.
  const int64_t X = status ? (int64_t)status->num_rendered : cap;
  const int64_t worst = X * HGS_PAIRS_PER_ENTRY;
  const int64_t pc = (status && status->num_pairs > 0 && (int64_t)status->num_pairs < worst) ? (int64_t)status->num_pairs : worst;
  const uint32_t pair_cap = (uint32_t)(pc > 0xffffffffll ? 0xffffffffll : pc);
  float* rows = static_cast<float*>(bwd_scratch);
  float* pair_rows = rows + (size_t)X * HGS_ROW_FLOATS;
  HGS_STAGE(0);
  if (maybe_entries) {
    // persistent workgroups of HGS_BWD_BLOCK_WAVES waves: as many waves as the chip holds (LDS: 11.8 KB per wave =>
    // 12 per CU, 3 per SIMD); the waves of a workgroup draw its groups of four work items through an LDS ticket
    const int resident = hgs_knob("HGS_BWD_WAVES_PER_CU", 12) * cu_count(stream);
    hipLaunchKernelGGL(hgs_k_render_bwd, dim3((unsigned)std::max(HGS_NXCD, resident / HGS_BWD_BLOCK_WAVES / HGS_NXCD * HGS_NXCD)), dim3(64 * HGS_BWD_BLOCK_WAVES), 0, stream, v, L, status_dev, L.recs, L.cstate,
                       out_color, out_depth, out_alpha, dL_dout_color, dL_dout_depth, dL_dout_alpha, pair_rows, pair_cap);
    HGS_LAUNCH_CHECK();
    HGS_STAGE(1);
    if (X > 0) {
      // a caller that holds the status hands the reduction its entry count (no status round trip in front of its chain)
      const uint32_t R_host = status ? status->num_rendered : 0xffffffffu;
      if (v.pairchunks)
        hipLaunchKernelGGL(hgs_k_pair_reduce_ch, dim3((unsigned)((X + 255) / 256)), dim3(256), 0, stream, v, L, status_dev, L.recs,
                           pair_rows, rows, pair_cap, R_host);
      else
        hipLaunchKernelGGL(hgs_k_pair_reduce_em, dim3((unsigned)((X + 255) / 256)), dim3(256), 0, stream, v, L, status_dev, L.recs,
                           pair_rows, rows, pair_cap, R_host);
      HGS_LAUNCH_CHECK();
    }
  }
  if (!maybe_entries) HGS_STAGE(1);
  HGS_STAGE(2);
#define HGS_LAUNCH_PRE_BWD(K, GRID, THREADS, LDS)                                                     \
  hipLaunchKernelGGL(K, dim3(GRID), dim3(THREADS), LDS, stream, v, L, status_dev, rows, means3D, shs, \
                     colors_precomp, opacities, scales, rotations, cov3D_precomp, dL_dmeans3D, dL_dmeans2D, \
                     dL_dshs, dL_dcolors_precomp, dL_dopacities, dL_dscales, dL_drotations,            \
                     dL_dcov3D_precomp, pack, (int)pack_F)
  // One view: the instantiation without the loop over views (94 instead of 176 VGPRs at SH degree 0).  Several
  // views: one thread per (Gaussian, view) - a workgroup of B waves per 64 Gaussians, summed in view order
  // through LDS (preprocess.hip, mode 2); combinations whose exchange buffer would not fit: the loop.
  const int deg = shs ? v.D : 0;
  const int nc = (deg + 1) * (deg + 1);
  const unsigned thr_p = 64u * (unsigned)v.B, grid_p = (unsigned)((v.P + 63) / 64);
  const size_t lds_p = (size_t)(23 + 3 * nc) * thr_p * sizeof(float);
  // The exchange buffer may take the CU's whole LDS (160 KB on gfx950; beyond 64 KB the kernel's dynamic-LDS limit is raised
  // first): 16 views at SH degrees 0 / 1, 8 at degrees 2 / 3 (139 KB at degree 3) - the thread-per-(Gaussian, view) form
  // then covers every batch a training step makes; the loop form (d*: 256 VGPRs at degree 3, one wave per SIMD) remains
  // for what does not fit or when the limit cannot be raised.
  bool vpar_ok = v.B >= HGS_PRE_BWD_VPAR_MIN_VIEWS && v.B <= (deg >= 2 ? 8 : 16) && lds_p <= 160 * 1024;
  if (vpar_ok && lds_p > 65536) {
    // the attribute is raised ONCE per (device, kernel) to the CU's whole LDS and remembered: a driver call per backward
    // was host time on the hot path; a refusal is remembered too (and said once on stderr): the loop form then serves
    static std::atomic<int> lds_raised[64][4];          // 0: not tried, 1: raised to 160 KB, -1: refused
    int dev = 0;
    if (hipStreamGetDevice(stream, &dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int st = lds_raised[dev][deg].load(std::memory_order_relaxed);
    if (st == 0) {
      const void* kfn = deg == 0 ? reinterpret_cast<const void*>(hgs_k_preprocess_bwd_p0)
                      : deg == 1 ? reinterpret_cast<const void*>(hgs_k_preprocess_bwd_p1)
                      : deg == 2 ? reinterpret_cast<const void*>(hgs_k_preprocess_bwd_p2)
                                 : reinterpret_cast<const void*>(hgs_k_preprocess_bwd_p3);
      if (hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess) {
        st = 1;
      } else {
        (void)hipGetLastError();
        st = -1;
        fprintf(stderr, "libhgs_rast: cannot raise the dynamic LDS limit of hgs_k_preprocess_bwd_p%d on device %d: "
                        "batches that need more than 64 KB take the (slower) loop form\n", deg, dev);
      }
      lds_raised[dev][deg].store(st, std::memory_order_relaxed);
    }
    if (st < 0) vpar_ok = false;
  }
  const int mode = v.B == 1 ? 1 : (vpar_ok ? 2 : 0);
  const size_t lds_s = hgs_pre_bwd_stage_bytes(M, deg, shs != nullptr && dL_dshs != nullptr);      // (<= 53 KB)
  switch (deg + 4 * mode) {
    case 0: HGS_LAUNCH_PRE_BWD(hgs_k_preprocess_bwd_d0, v.nblk, HGS_BLOCK, 0); break;
    case 1: HGS_LAUNCH_PRE_BWD(hgs_k_preprocess_bwd_d1, v.nblk, HGS_BLOCK, 0); break;
    case 2: HGS_LAUNCH_PRE_BWD(hgs_k_preprocess_bwd_d2, v.nblk, HGS_BLOCK, 0); break;
    case 3: HGS_LAUNCH_PRE_BWD(hgs_k_preprocess_bwd_d3, v.nblk, HGS_BLOCK, 0); break;
    case 4: HGS_LAUNCH_PRE_BWD(hgs_k_preprocess_bwd_s0, v.nblk, HGS_BLOCK, 0); break;
    case 5: HGS_LAUNCH_PRE_BWD(hgs_k_preprocess_bwd_s1, v.nblk, HGS_BLOCK, lds_s); break;      // (SH blocks through LDS)
    case 6: HGS_LAUNCH_PRE_BWD(hgs_k_preprocess_bwd_s2, v.nblk, HGS_BLOCK, lds_s); break;
    case 7: HGS_LAUNCH_PRE_BWD(hgs_k_preprocess_bwd_s3, v.nblk, HGS_BLOCK, lds_s); break;
    case 8: HGS_LAUNCH_PRE_BWD(hgs_k_preprocess_bwd_p0, grid_p, thr_p, lds_p); break;
    case 9: HGS_LAUNCH_PRE_BWD(hgs_k_preprocess_bwd_p1, grid_p, thr_p, lds_p); break;
    case 10: HGS_LAUNCH_PRE_BWD(hgs_k_preprocess_bwd_p2, grid_p, thr_p, lds_p); break;
    default: HGS_LAUNCH_PRE_BWD(hgs_k_preprocess_bwd_p3, grid_p, thr_p, lds_p); break;
  }
#undef HGS_LAUNCH_PRE_BWD
  HGS_LAUNCH_CHECK();
  HGS_STAGE(3);
  return HGS_OK;
}
}  // namespace

extern "C" {

int hgs_backward_batch_act(const hgs_settings* s, int32_t B, int32_t P, int32_t M, const float* means3D,
                           const float* shs, const float* colors_precomp, const float* opacities,
                           const float* scales, const float* rotations, const float* cov3D_precomp,
                           const int32_t* radii, const float* out_color, const float* out_depth,
                           const float* out_alpha, const float* dL_dout_color,
                           const float* dL_dout_depth, const float* dL_dout_alpha, const void* geom,
                           const void* bin, const void* img, const hgs_status* status,
                           int64_t entry_capacity, void* bwd_scratch, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs,
                           float* dL_dcolors_precomp, float* dL_dopacities, float* dL_dscales,
                           float* dL_drotations, float* dL_dcov3D_precomp, void* const* stage_events,
                           int32_t activation_flags, void* stream_) {
  return backward_impl(s, B, P, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, radii, out_color,
                       out_depth, out_alpha, dL_dout_color, dL_dout_depth, dL_dout_alpha, geom, bin, img, status,
                       entry_capacity, bwd_scratch, dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dcolors_precomp, dL_dopacities,
                       dL_dscales, dL_drotations, dL_dcov3D_precomp, stage_events, activation_flags, nullptr, 0, stream_);
}

int hgs_backward_batch_packed(const hgs_settings* s, int32_t B, int32_t P, int32_t M, const float* means3D,
                              const float* shs, const float* opacities, const float* scales, const float* rotations,
                              const int32_t* radii, const float* out_color, const float* out_depth, const float* out_alpha,
                              const float* dL_dout_color, const float* dL_dout_depth, const float* dL_dout_alpha,
                              const void* geom, const void* bin, const void* img, const hgs_status* status,
                              int64_t entry_capacity, void* bwd_scratch, float* pack, float* dL_dmeans2D_views,
                              void* const* stage_events, int32_t activation_flags, void* stream_) {
  if (!pack || !shs || !scales || !rotations || M < 1) return P == 0 ? HGS_OK : HGS_EINVAL;
  float* marker = pack;        // (non-null markers: the row carries the SH, scale and rotation parts)
  return backward_impl(s, B, P, M, means3D, shs, nullptr, opacities, scales, rotations, nullptr, radii, out_color, out_depth,
                       out_alpha, dL_dout_color, dL_dout_depth, dL_dout_alpha, geom, bin, img, status, entry_capacity,
                       bwd_scratch, marker, dL_dmeans2D_views, marker, nullptr, marker, marker, marker, nullptr, stage_events,
                       activation_flags, pack, 15 + 3 * M, stream_);
}

int hgs_backward_batch(const hgs_settings* s, int32_t B, int32_t P, int32_t M, const float* means3D,
                       const float* shs, const float* colors_precomp, const float* opacities,
                       const float* scales, const float* rotations, const float* cov3D_precomp,
                       const int32_t* radii, const float* out_color, const float* out_depth,
                       const float* out_alpha, const float* dL_dout_color,
                       const float* dL_dout_depth, const float* dL_dout_alpha, const void* geom,
                       const void* bin, const void* img, const hgs_status* status,
                       int64_t entry_capacity, void* bwd_scratch, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs,
                       float* dL_dcolors_precomp, float* dL_dopacities, float* dL_dscales,
                       float* dL_drotations, float* dL_dcov3D_precomp, void* const* stage_events,
                       void* stream_) {
  return hgs_backward_batch_act(s, B, P, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                radii, out_color, out_depth, out_alpha, dL_dout_color, dL_dout_depth, dL_dout_alpha,
                                geom, bin, img, status, entry_capacity, bwd_scratch, dL_dmeans3D, dL_dmeans2D, dL_dshs,
                                dL_dcolors_precomp, dL_dopacities, dL_dscales, dL_drotations, dL_dcov3D_precomp,
                                stage_events, 0, stream_);
}

int hgs_backward(const hgs_settings* s, int32_t P, int32_t M, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* opacities,
                 const float* scales, const float* rotations, const float* cov3D_precomp,
                 const int32_t* radii, const float* out_color, const float* out_depth,
                 const float* out_alpha, const float* dL_dout_color,
                 const float* dL_dout_depth, const float* dL_dout_alpha, const void* geom,
                 const void* bin, const void* img, const hgs_status* status,
                 int64_t entry_capacity, void* bwd_scratch, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs,
                 float* dL_dcolors_precomp, float* dL_dopacities, float* dL_dscales,
                 float* dL_drotations, float* dL_dcov3D_precomp, void* const* stage_events,
                 void* stream_) {
  return hgs_backward_batch(s, 1, P, M, means3D, shs, colors_precomp, opacities, scales, rotations,
                            cov3D_precomp, radii, out_color, out_depth, out_alpha, dL_dout_color,
                            dL_dout_depth, dL_dout_alpha, geom, bin, img, status, entry_capacity, bwd_scratch,
                            dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dcolors_precomp, dL_dopacities, dL_dscales,
                            dL_drotations, dL_dcov3D_precomp, stage_events, stream_);
}

int hgs_densify_stats(int32_t B, int32_t P, const float* dL_dmeans2D, const int32_t* radii, const uint8_t* keep,
                      float* xyz_gradient_accum, float* denom, float* max_radii2D, int32_t* radii_max,
                      uint8_t* visibility, void* stream_) {
  if (B < 1 || P < 0) return HGS_EINVAL;
  if (P == 0) return HGS_OK;
  if (!dL_dmeans2D || !radii || !xyz_gradient_accum || !denom || !max_radii2D) return HGS_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(hgs_k_densify_stats, dim3((P + 255) / 256), dim3(256), 0, stream, (int)B, (int)P, dL_dmeans2D,
                     radii, keep, xyz_gradient_accum, denom, max_radii2D, radii_max, visibility);
  HGS_LAUNCH_CHECK();
  return HGS_OK;
}

int hgs_densify_masks(int32_t P, const float* xyz_gradient_accum, const float* denom, const float* scales,
                      int32_t scales_are_log, const float* opacity, int32_t opacity_is_logit,
                      const float* max_radii2D, float grad_threshold, float percent_dense, float extent,
                      float min_opacity, float max_screen_size, float size_thresh, uint8_t* clone_mask,
                      uint8_t* split_mask, uint8_t* prune_mask, uint32_t* counts, void* stream_) {
  if (P < 0) return HGS_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (counts) {
    hipError_t e = hipMemsetAsync(counts, 0, 3 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return hip_rc(e);
  }
  if (P == 0) return HGS_OK;
  if (!xyz_gradient_accum || !denom || !scales || !opacity || (max_screen_size > 0.0f && !max_radii2D)) return HGS_EINVAL;
  hipLaunchKernelGGL(hgs_k_densify_masks, dim3((P + 255) / 256), dim3(256), 0, stream, (int)P, xyz_gradient_accum, denom,
                     scales, (int)scales_are_log, opacity, (int)opacity_is_logit, max_radii2D, grad_threshold,
                     percent_dense, extent, min_opacity, max_screen_size, size_thresh, clone_mask, split_mask, prune_mask,
                     counts);
  HGS_LAUNCH_CHECK();
  return HGS_OK;
}

size_t hgs_compact_scratch_bytes(int32_t P) {
  return hgs_align_up(((size_t)(P > 0 ? P : 0) + 1023) / 1024 * 4 + 4, ALIGN);
}

int hgs_compact_index(int32_t P, const uint8_t* keep, int32_t* src_of_dst, uint32_t* num_kept, void* scratch,
                      void* stream_) {
  if (P < 0 || !num_kept) return HGS_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (P == 0) {
    hipError_t e = hipMemsetAsync(num_kept, 0, 4, stream);
    return e == hipSuccess ? HGS_OK : hip_rc(e);
  }
  if (!keep || !src_of_dst || !scratch) return HGS_EINVAL;
  const int nb = (P + 1023) / 1024;
  uint32_t* blocks = static_cast<uint32_t*>(scratch);
  hipLaunchKernelGGL(hgs_k_keep_count, dim3(nb), dim3(1024), 0, stream, (int)P, keep, blocks);
  hipLaunchKernelGGL(hgs_k_keep_scan, dim3(1), dim3(1024), 0, stream, nb, blocks, num_kept);
  hipLaunchKernelGGL(hgs_k_keep_index, dim3(nb), dim3(1024), 0, stream, (int)P, keep, blocks, src_of_dst);
  HGS_LAUNCH_CHECK();
  return HGS_OK;
}

int hgs_gather_rows(int64_t n_out, int32_t row_floats, const int32_t* src_of_dst, const float* src, float* dst,
                    void* stream_) {
  if (n_out < 0 || row_floats < 1) return HGS_EINVAL;
  if (n_out == 0) return HGS_OK;
  if (!src_of_dst || !src || !dst) return HGS_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const long long n = (long long)n_out * row_floats;
  hipLaunchKernelGGL(hgs_k_gather_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, (int)row_floats,
                     src_of_dst, src, dst);
  HGS_LAUNCH_CHECK();
  return HGS_OK;
}

int hgs_reanchor(int32_t P, const float* vertices, const int32_t* faces, const int32_t* mapping_face,
                 const float* mapping_uvw, const float* mapping_dist, float* xyz, void* stream_) {
  if (P < 0) return HGS_EINVAL;
  if (P == 0) return HGS_OK;
  if (!vertices || !faces || !mapping_face || !mapping_uvw || !mapping_dist || !xyz) return HGS_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(hgs_k_reanchor, dim3((P + 255) / 256), dim3(256), 0, stream, (int)P, vertices, faces, mapping_face,
                     mapping_uvw, mapping_dist, xyz);
  HGS_LAUNCH_CHECK();
  return HGS_OK;
}

int hgs_mark_visible(const hgs_settings* s, int32_t P, const float* means3D, uint8_t* present,
                     void* stream_) {
  if (!s || !s->viewmatrix || P < 0) return HGS_EINVAL;
  if (P == 0) return HGS_OK;
  if (!means3D || !present) return HGS_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(hgs_k_mark_visible, dim3((P + HGS_BLOCK - 1) / HGS_BLOCK), dim3(HGS_BLOCK),
                     0, stream, s->viewmatrix, (int)P, means3D, present);
  HGS_LAUNCH_CHECK();
  return HGS_OK;
}

}  // extern "C"
