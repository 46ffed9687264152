/*
 * hgs_rast.h - C ABI of libhgs_rast.so, the MI355X (gfx950) differentiable 3D Gaussian
 * Splatting rasterizer that drops in behind HumanGaussian's
 * `diff_gaussian_rasterization` extension.
 *
 * The reference has no C ABI: its boundary is the pybind11 module
 * `diff_gaussian_rasterization._C` of the un-vendored ashawkey fork, reached from
 *   /root/reference/gaussiansplatting/gaussian_renderer/__init__.py:14,36-51,86-94
 *   /root/reference/gs_renderer.py:10-13,951-966,1006-1015
 * Each entry point below names the `_C` function it replaces.  Conventions:
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says HOST;
 *   - the caller owns every buffer; the library never allocates, frees or synchronises;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no global state,
 *     so calls on distinct streams / devices may run concurrently;
 *   - return value: 0 on success, a negative HGS_E* code for argument errors, or
 *     -(1000 + hipError_t) when a launch fails.  Nothing throws across the ABI.
 *   - tensors are fp32, contiguous, row-major with the shapes of SURVEY.md section 3.3.
 */
#ifndef HGS_RAST_H
#define HGS_RAST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HGS_MAX_VIEWS 16  /* views (cameras) one batched call can take */
#define HGS_MAX_ENTRY_CAPACITY (1ll << 27)  /* entry ids travel in 27 bits: hgs_forward* / hgs_backward* return HGS_EINVAL for a larger
                                               entry_capacity (134 M tile-list entries = ~45 GB of bin buffer per call) */

#define HGS_OK 0
#define HGS_EINVAL (-1)   /* bad argument (null pointer, negative size, ...) */
#define HGS_ESHAPE (-2)   /* "exactly one of shs / colors_precomp", "scales+rotations / cov3D" */

/* Mirrors GaussianRasterizationSettings, same field order as the reference call site
 * gaussian_renderer/__init__.py:36-49.  bg / viewmatrix / projmatrix / campos stay on the
 * device (the reference hands over CUDA tensors); matrices are the row-major bytes of the
 * (4,4) tensors, i.e. viewmatrix = w2c^T, projmatrix = viewmatrix @ P^T (cameras.py:50-52). */
typedef struct hgs_settings {
  int32_t image_height;
  int32_t image_width;
  float tanfovx;
  float tanfovy;
  const float* bg;          /* [3]  */
  float scale_modifier;
  const float* viewmatrix;  /* [16] */
  const float* projmatrix;  /* [16] */
  int32_t sh_degree;        /* ACTIVE degree (0..3) */
  const float* campos;      /* [3]  */
  int32_t prefiltered;
  int32_t debug;
} hgs_settings;

/* Written by hgs_forward into the tail of the geom buffer (device) and, once the stream
 * reaches that point, mirrored to `status_host` (HOST, pinned) when it is non-NULL. */
typedef struct hgs_status {
  uint32_t num_rendered;   /* R = entries in the tile lists.  <= upstream's `num_rendered`: a    */
                           /* (Gaussian, tile) pair gets an entry only if the box of its         */
                           /* alpha >= 1/255 ellipse touches the tile (results are unchanged)    */
  uint32_t active_tiles;   /* tiles with a non-empty list                              */
  uint32_t num_pairs;      /* (entry, 4x4-pixel cell) pairs = the pair rows of the backward scratch.  Known */
                           /* only when the sort has run: 0 in what the status mirror / event delivers, set */
                           /* in the device copy and in a MAPPED host mirror when the blend forward starts  */
                           /* (a host that finds it non-zero later may size the scratch by it, see          */
                           /* hgs_bwd_scratch_bytes_pairs; 0 = not known (yet): size for the worst case).   */
                           /* LIFETIME: a mapped mirror is therefore written a SECOND time, after the ready */
                           /* word reserved[2]: it must stay valid (not freed, not reused for another call) */
                           /* until the forward's stream work has completed, not merely until the poll ends */
  uint32_t bwd_groups;     /* unused since ABI v11 (0): the blend kernels run persistent waves */
  uint32_t overflow;       /* != 0: outputs are INVALID.  bit0: R exceeded              */
                           /* entry_capacity (retry with >= num_rendered); bit1: a tile */
                           /* list exceeded max_tile_entries_hint (retry with hint 0)   */
  uint32_t reserved[3];    /* [0] = entry_capacity the bin buffer was carved with,      */
                           /* [1] = longest tile list, [2] = 1: "complete" - in the host mirror this */
                           /* word is stored LAST behind a system-scope fence, so a host that  */
                           /* cleared it before the call may POLL it instead of using an event */
} hgs_status;

/* ---- buffer sizing (host-side arithmetic, no device work) --------------------------
 * Replaces the three resize callbacks (geometry / binning / image state) that upstream's
 * rasterize_gaussians() drives through torch.  geom: per-Gaussian + per-tile state;
 * bin: per-(tile,Gaussian) entry state, sized by entry_capacity; img: per-pixel state;
 * bwd_scratch: gradient rows per entry and per (entry, 4x4-pixel cell) pair, used only inside
 * hgs_backward (48 + 16 x 40 B per entry). */
size_t hgs_geom_bytes(int32_t P, int32_t image_height, int32_t image_width);
size_t hgs_bin_bytes(int64_t entry_capacity);
size_t hgs_img_bytes(int32_t image_height, int32_t image_width);
size_t hgs_bwd_scratch_bytes(int64_t num_rendered);
/* the same with the pair rows counted instead of bounded (hgs_status.num_pairs of THAT forward call, once
 * published; ~4.4 per entry on an avatar instead of 16): 48 B per entry + 40 B per pair */
size_t hgs_bwd_scratch_bytes_pairs(int64_t num_rendered, int64_t num_pairs);
/* the same for a batch of B views (geom / img scale with B; the bin buffer and the backward
 * scratch are sized by the entry capacity / num_rendered of ALL views together) */
size_t hgs_geom_bytes_batch(int32_t B, int32_t P, int32_t image_height, int32_t image_width);
size_t hgs_img_bytes_batch(int32_t B, int32_t image_height, int32_t image_width);

/* Optional per-stage timing (measurement only; pass NULL in production): `stage_events`
 * is a HOST array of hipEvent_t handles; entry k (if non-NULL) is recorded on `stream`
 * after stage k.  Forward: 0 start, 1 preprocess, 2 tile tables, 3 fill (+ status), 4 sort, 5 blend.
 * Backward: 0 start, 1 blend backward, 2 pair reduction, 3 preprocess backward. */
#define HGS_FWD_STAGES 6
#define HGS_BWD_STAGES 4

/* ---- forward: replaces _C.rasterize_gaussians --------------------------------------
 * Exactly one of shs / colors_precomp and exactly one of {scales,rotations} /
 * cov3D_precomp must be non-NULL (HGS_ESHAPE otherwise), matching the fork's Python
 * checks.  shs is (P, M, 3); M = max coefficient count of the tensor.
 * out_color (3,H,W), out_depth (1,H,W), out_alpha (1,H,W), radii (P) int32.
 * store_bwd_state = 0 skips the pixel-state stores the backward needs (no-grad / inference calls).
 * max_tile_entries_hint: 0 = unknown; > 0 = the caller promises no tile list is longer
 * (lets the library skip launching sort classes that cannot occur); a broken promise is
 * detected on the device and reported as overflow bit 1 (value 2) - call again with 0.
 * P == 0 writes background / zeros and reports num_rendered = 0.
 * status_host_mapped != 0: `status_host` is pinned host memory that the device can address
 * with the same pointer (hipHostMalloc / torch pin_memory on ROCm); the fill launch then
 * stores the status into it directly (system-scope fence) and no copy is enqueued.
 * Otherwise the status copy to `status_host` is enqueued right after the fill stage (before
 * sort / blend); `status_event` (a hipEvent_t, may be NULL) is recorded right behind it, so a
 * host can wait for just the status - hipEventSynchronize(status_event) - while the rest of
 * the forward is still running, and knows about an overflow before it hands out outputs. */
int hgs_forward(const hgs_settings* s, int32_t P, int32_t M,
                const float* means3D, const float* shs, const float* colors_precomp,
                const float* opacities, const float* scales, const float* rotations,
                const float* cov3D_precomp,
                float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                void* geom, void* bin, int64_t entry_capacity, void* img,
                int32_t store_bwd_state, int32_t max_tile_entries_hint,
                hgs_status* status_host, int32_t status_host_mapped, void* status_event,
                void* const* stage_events, void* stream);

/* ---- batched forward: B views of the same Gaussians in ONE launch set ----------------
 * Replaces the per-view Python loop around _C.rasterize_gaussians at
 * /root/reference/threestudio/systems/GaussianDreamer.py:244-266 (8 views per training step).
 * `views` is a HOST array of B settings (1 <= B <= HGS_MAX_VIEWS) that must agree in
 * image_height / image_width / sh_degree / scale_modifier and may differ in everything a camera
 * carries (tanfov, matrices, campos, bg).  Outputs are [B][3][H][W], [B][1][H][W], [B][1][H][W]
 * and radii [B][P]; each view's result is bit-identical to a separate hgs_forward call.  One
 * status for the whole batch (num_rendered = sum over views).  hgs_forward IS this function
 * with B = 1; buffers are sized with the *_batch sizing functions. */
int hgs_forward_batch(const hgs_settings* views, int32_t B, int32_t P, int32_t M,
                      const float* means3D, const float* shs, const float* colors_precomp,
                      const float* opacities, const float* scales, const float* rotations,
                      const float* cov3D_precomp,
                      float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                      void* geom, void* bin, int64_t entry_capacity, void* img,
                      int32_t store_bwd_state, int32_t max_tile_entries_hint,
                      hgs_status* status_host, int32_t status_host_mapped, void* status_event,
                      void* const* stage_events, void* stream);

/* ---- batched backward ----------------------------------------------------------------
 * dL_dout_* are [B][..] like the outputs.  Parameter gradients are the SUM over the views, formed
 * in view order 0..B-1 inside one kernel (deterministic; what autograd accumulates over the
 * reference's loop); dL_dmeans2D stays per view, [B][P][3], because the caller consumes it per
 * view (GaussianDreamer.py:385-387).  radii is [B][P]. */
int hgs_backward_batch(const hgs_settings* views, int32_t B, int32_t P, int32_t M,
                       const float* means3D, const float* shs, const float* colors_precomp,
                       const float* opacities, const float* scales, const float* rotations,
                       const float* cov3D_precomp, const int32_t* radii,
                       const float* out_color, const float* out_depth, const float* out_alpha,
                       const float* dL_dout_color, const float* dL_dout_depth,
                       const float* dL_dout_alpha,
                       const void* geom, const void* bin, const void* img,
                       const hgs_status* status, int64_t entry_capacity, void* bwd_scratch,
                       float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs,
                       float* dL_dcolors_precomp, float* dL_dopacities, float* dL_dscales,
                       float* dL_drotations, float* dL_dcov3D_precomp,
                       void* const* stage_events, void* stream);

/* ---- fused activations -----------------------------------------------------
 * The reference feeds the rasterizer `get_opacity` = sigmoid(_opacity), `get_scaling` = exp(_scaling),
 * `get_rotation` = normalize(_rotation) (gaussiansplatting/scene/gaussian_model.py:95-115): three
 * elementwise kernels forward and three backward per render call.  With the *_act entry points the
 * caller hands over the RAW parameters and the activation runs inside the per-Gaussian kernels
 * (forward: as the value is loaded; backward: chain rule applied to the summed gradient, so
 * dL_dopacities / dL_dscales / dL_drotations are gradients w.r.t. the raw parameters).
 * activation_flags: any combination of the HGS_ACT_* bits below (+ HGS_GRAD_SCALE_TRUE_DERIVATIVE for the backward);
 * 0 = identical to the plain entry points. */
#define HGS_ACT_OPACITY_SIGMOID 1    /* opacities are logits                                  */
#define HGS_ACT_SCALE_EXP 2          /* scales are log-scales                                 */
#define HGS_ACT_ROTATION_NORMALIZE 4 /* rotations are un-normalised quaternions (w,x,y,z)      */
/* Backward only (ignored by the forward).  dL_dscales at scale_modifier != 1: the fork's backward (computeCov3D:
 * `s = mod * scale`, `dL_dscale = dot(Rt[i], dL_dMt[i])`) returns dL/d(mod * scale) - the modifier's factor is missing -
 * and that is what this library returns by default, like the extension it replaces (identical at 1.0, the only value
 * the reference passes: gaussian_renderer/__init__.py:18, gs_renderer.py:925).  With this bit set dL_dscales is the
 * true derivative dL/dscale = mod * dL/d(mod * scale). */
#define HGS_GRAD_SCALE_TRUE_DERIVATIVE 8
int hgs_forward_batch_act(const hgs_settings* views, int32_t B, int32_t P, int32_t M,
                          const float* means3D, const float* shs, const float* colors_precomp,
                          const float* opacities, const float* scales, const float* rotations,
                          const float* cov3D_precomp,
                          float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                          void* geom, void* bin, int64_t entry_capacity, void* img,
                          int32_t store_bwd_state, int32_t max_tile_entries_hint,
                          hgs_status* status_host, int32_t status_host_mapped, void* status_event,
                          void* const* stage_events, int32_t activation_flags, void* stream);
/* v16: hgs_forward_batch_act that also ZERO-FILLS the caller's screen-space leaf.  The reference creates
 * `screenspace_points = torch.zeros_like(xyz) + 0` per view (gaussian_renderer/__init__.py:26) only to receive
 * dL/dmeans2D in its .grad: a fill (and an add) launch in front of every forward.  means2D_leaf [B][P][3] (or NULL:
 * exactly hgs_forward_batch_act) is written with zeros by the per-Gaussian kernel of the forward itself - row (b, i) by
 * the thread that projects Gaussian i of view b - so the caller hands over UNINITIALISED storage and reads zeros behind
 * the call: no launch, no kernel boundary (2.6 us of a 158 us step at 100k Gaussians). */
int hgs_forward_batch_act_leaf(const hgs_settings* views, int32_t B, int32_t P, int32_t M,
                               const float* means3D, const float* shs, const float* colors_precomp,
                               const float* opacities, const float* scales, const float* rotations,
                               const float* cov3D_precomp,
                               float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                               void* geom, void* bin, int64_t entry_capacity, void* img,
                               int32_t store_bwd_state, int32_t max_tile_entries_hint,
                               hgs_status* status_host, int32_t status_host_mapped, void* status_event,
                               void* const* stage_events, int32_t activation_flags, float* means2D_leaf, void* stream);
int hgs_backward_batch_act(const hgs_settings* views, int32_t B, int32_t P, int32_t M,
                           const float* means3D, const float* shs, const float* colors_precomp,
                           const float* opacities, const float* scales, const float* rotations,
                           const float* cov3D_precomp, const int32_t* radii,
                           const float* out_color, const float* out_depth, const float* out_alpha,
                           const float* dL_dout_color, const float* dL_dout_depth,
                           const float* dL_dout_alpha,
                           const void* geom, const void* bin, const void* img,
                           const hgs_status* status, int64_t entry_capacity, void* bwd_scratch,
                           float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs,
                           float* dL_dcolors_precomp, float* dL_dopacities, float* dL_dscales,
                           float* dL_drotations, float* dL_dcov3D_precomp,
                           void* const* stage_events, int32_t activation_flags, void* stream);

/* ---- backward: replaces _C.rasterize_gaussians_backward ----------------------------
 * `status` is a HOST copy of what hgs_forward reported (read after the stream has passed
 * the forward), or NULL: then nothing about the forward's result is needed on the host -
 * the blend backward is launched with the capacity-derived upper bound of workgroups
 * (entry_capacity/64 + tiles) and every kernel reads the device-side status (an
 * overflowed forward yields all-zero gradients).  `entry_capacity` must equal the value
 * given to hgs_forward (it fixes the carve of `bin`), bwd_scratch must hold
 * hgs_bwd_scratch_bytes(num_rendered) - or (entry_capacity) when status is NULL - or, when
 * status->num_pairs is non-zero, hgs_bwd_scratch_bytes_pairs(num_rendered, status->num_pairs):
 * status->num_pairs DECLARES how many pair rows the scratch holds (0 = the worst case of 16 per
 * entry).  The kernels compare it with the count the forward left on the device; if the device
 * holds more pairs than declared (a stale or foreign status) no pair row is written and every
 * gradient of the call is NaN - loud, and in bounds (ABI v14; up to v13 this overran the scratch).
 * out_* are the forward's outputs (unmodified), dL_dout_* the incoming
 * gradients (any of them may be NULL = zeros).  Every dL_d* output that is non-NULL is
 * fully overwritten (no pre-zeroing needed, no atomics: results are deterministic);
 * dL_dmeans2D is (P,3) in NDC units with z = 0. */
int hgs_backward(const hgs_settings* s, int32_t P, int32_t M,
                 const float* means3D, const float* shs, const float* colors_precomp,
                 const float* opacities, const float* scales, const float* rotations,
                 const float* cov3D_precomp, const int32_t* radii,
                 const float* out_color, const float* out_depth, const float* out_alpha,
                 const float* dL_dout_color, const float* dL_dout_depth,
                 const float* dL_dout_alpha,
                 const void* geom, const void* bin, const void* img,
                 const hgs_status* status, int64_t entry_capacity, void* bwd_scratch,
                 float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs,
                 float* dL_dcolors_precomp, float* dL_dopacities, float* dL_dscales,
                 float* dL_drotations, float* dL_dcov3D_precomp,
                 void* const* stage_events, void* stream);

/* ---- frustum test: replaces _C.mark_visible ----------------------------------------
 * present[i] = 1 iff the view-space depth of means3D[i] exceeds 0.2. */
int hgs_mark_visible(const hgs_settings* s, int32_t P, const float* means3D,
                     uint8_t* present, void* stream);

/* Mean squared distance of every point to its 3 nearest neighbours (own index excluded,
 * duplicates count): replaces `simple_knn._C.distCUDA2(points)`, which the reference calls to
 * initialise the scales of a new cloud
 * (/root/reference/gaussiansplatting/scene/gaussian_model.py:20,134; gs_renderer.py:14,386-389;
 * kernel submodules/simple-knn/simple_knn.cu:147-183).  points: [P][3] fp32, mean_dist2: [P]. */
int hgs_knn_mean_dist2(int32_t P, const float* points, float* mean_dist2, void* stream);
/* The same result (the same three distances per point, exactly) in near-linear time: a uniform grid sized on the device
 * from the bounding box, counting sort of the points by cell, ring search around each point's cell - the role of the
 * Morton sort + 1024-point box pruning of simple_knn.cu:63-221, without a global sort and without the two host round
 * trips of SimpleKNN::knn (:187-197).  `scratch`: hgs_knn_scratch_bytes(P) bytes (~28 B per point + 8 B per cell).
 * Degenerate clouds (more than 4096 points in one cell) take the brute force above; the choice is made on the device.
 * 100k points: ~2.5 ms -> tens of us; 5 M points (a densified avatar): ~6 s -> ms. */
size_t hgs_knn_scratch_bytes(int32_t P);
int hgs_knn_mean_dist2_grid(int32_t P, const float* points, float* mean_dist2, void* scratch, void* stream);

/* View-parallel reduction behind the single all-gather (the serial accumulation it
 * reproduces: /root/reference/threestudio/systems/GaussianDreamer.py:253-256,385-391).
 * gathered: [world][P][F] fp32 packs (per-Gaussian gradient columns, radii as the LAST column);
 * out: [P][F] = sum over ranks in rank order for columns < F-1, max for column F-1. */
int hgs_reduce_view_packs(int32_t world, int64_t P, int32_t F, const float* gathered, float* out,
                          void* stream);
/* The same as one link of a CHAIN of collectives (several views per rank, one collective per round of views, each
 * overlapped with the render of the next round): out = ((acc_in + rank 0) + rank 1) + ... in exactly that order (max on
 * the radii column), so that the chain over the rounds equals the serial accumulation over the views in view order
 * (/root/reference/threestudio/systems/GaussianDreamer.py:244-266,385-391) bit for bit.  acc_in may be NULL (= the call
 * above) and may alias out. */
int hgs_reduce_view_packs_acc(int32_t world, int64_t P, int32_t F, const float* gathered, const float* acc_in,
                              float* out, void* stream);

/* Packs one rank's contribution for that all-gather: out[P][15 + 3M] =
 * [dL/dmeans3D 3 | dL/dmeans2D 3 | dL/dsh 3M | dL/dopacity 1 | dL/dscale 3 | dL/drot 4 | radii 1]. */
int hgs_pack_view_contribution(int32_t P, int32_t M, const float* g_means3D, const float* g_means2D,
                               const float* g_sh, const float* g_opac, const float* g_scales,
                               const float* g_rot, const int32_t* radii, float* out, void* stream);

/* v15: the view-parallel step without its pack pass and without its unpack kernels.
 * hgs_backward_batch_packed = hgs_backward_batch_act whose per-Gaussian kernel writes the gradients of Gaussian i as ONE
 * row of `pack` [P][15 + 3M] in the layout above (dL/dmeans2D summed over the call's B views in view order, radii = max
 * over the views, as an exact fp32 integer) instead of six tensors that hgs_pack_view_contribution would read back and
 * interleave (7.2 MB written + read + written per 100k Gaussians on the exposed path of a step: the pack IS what the
 * rank sends).  SH + scale / rotation inputs only (the configuration the reference trains: gaussian_renderer/__init__.py:
 * 57-82 with both pipe flags off); dL_dmeans2D_views [B][P][3] is optional (NULL: not written).  Same status / scratch
 * rules as hgs_backward*.
 * hgs_reduce_view_packs_unpack = hgs_reduce_view_packs_acc (acc_in may be NULL) whose result goes straight into the six
 * gradient tensors + radii [P] int32 (the step's LAST reduction: no [P][F] intermediate, no slicing / rounding kernels). */
int hgs_backward_batch_packed(const hgs_settings* views, int32_t B, int32_t P, int32_t M, const float* means3D,
                              const float* shs, const float* opacities, const float* scales, const float* rotations,
                              const int32_t* radii, const float* out_color, const float* out_depth, const float* out_alpha,
                              const float* dL_dout_color, const float* dL_dout_depth, const float* dL_dout_alpha,
                              const void* geom, const void* bin, const void* img, const hgs_status* status,
                              int64_t entry_capacity, void* bwd_scratch, float* pack, float* dL_dmeans2D_views,
                              void* const* stage_events, int32_t activation_flags, void* stream);
int hgs_reduce_view_packs_unpack(int32_t world, int64_t P, int32_t M, const float* gathered, const float* acc_in,
                                 float* g_means3D, float* g_means2D, float* g_sh, float* g_opac, float* g_scales,
                                 float* g_rot, int32_t* radii, void* stream);

/* ---- bookkeeping either side of the path ------------------------------
 * Densification statistics of one training step over B views
 * (/root/reference/threestudio/systems/GaussianDreamer.py:253-256,289,385-391 and
 * gaussiansplatting/scene/gaussian_model.py:434-438), one pass:
 *   radii_max = max_b radii[b];  visible = radii_max > 0 (&& keep[i] if keep != NULL);
 *   g = sum_b dL_dmeans2D[b] (view order);  for visible Gaussians:
 *   max_radii2D = max(max_radii2D, radii_max), xyz_gradient_accum += |g.xy|, denom += 1.
 * dL_dmeans2D [B][P][3], radii [B][P]; accum / denom / max_radii2D [P] fp32 updated in place;
 * radii_max [P] int32 and visibility [P] uint8 are optional outputs. */
int hgs_densify_stats(int32_t B, int32_t P, const float* dL_dmeans2D, const int32_t* radii, const uint8_t* keep,
                      float* xyz_gradient_accum, float* denom, float* max_radii2D, int32_t* radii_max,
                      uint8_t* visibility, void* stream);

/* Clone / split / prune masks (gaussian_model.py:359-438): grad = accum / denom (NaN -> 0),
 * big = max scale > percent_dense * extent;  clone = grad >= thr && !big;  split = grad >= thr && big;
 * prune = opacity < min_opacity || (max_screen_size > 0 && (max_radii2D > max_screen_size ||
 * max scale > 0.1 * extent)) || (size_thresh > 0 && max scale > size_thresh)  [prune_only: size_thresh].
 * scales [P][3] / opacity [P] may be the RAW parameters (log-scale / logit; flags) - the activations are
 * fused.  Masks are uint8 [P] (any may be NULL); counts (3 x uint32, optional) receives the number of
 * set clone / split / prune bits. */
int hgs_densify_masks(int32_t P, const float* xyz_gradient_accum, const float* denom, const float* scales,
                      int32_t scales_are_log, const float* opacity, int32_t opacity_is_logit,
                      const float* max_radii2D, float grad_threshold, float percent_dense, float extent,
                      float min_opacity, float max_screen_size, float size_thresh, uint8_t* clone_mask,
                      uint8_t* split_mask, uint8_t* prune_mask, uint32_t* counts, void* stream);

/* Stable compaction for the pruning of every parameter tensor and both Adam moments
 * (gaussian_model.py:283-337): hgs_compact_index turns keep[P] into the list of kept source rows
 * (order preserved) and their number; hgs_gather_rows then moves any [P][row_floats] fp32 tensor. */
size_t hgs_compact_scratch_bytes(int32_t P);
int hgs_compact_index(int32_t P, const uint8_t* keep, int32_t* src_of_dst, uint32_t* num_kept, void* scratch,
                      void* stream);
int hgs_gather_rows(int64_t n_out, int32_t row_floats, const int32_t* src_of_dst, const float* src, float* dst,
                    void* stream);

/* Re-anchoring of the Gaussians on a posed mesh (/root/reference/animation.py:384-403, a numpy pass on the
 * CPU plus an H2D copy per frame there): xyz[i] = uvw[i] . (v0,v1,v2) + dist[i] * unit normal of face
 * mapping_face[i].  vertices [V][3] fp32, faces [F][3] int32, mapping_* [P]. */
int hgs_reanchor(int32_t P, const float* vertices, const int32_t* faces, const int32_t* mapping_face,
                 const float* mapping_uvw, const float* mapping_dist, float* xyz, void* stream);

/* Library / ABI version (bumped on any signature change). */
int hgs_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HGS_RAST_H */
