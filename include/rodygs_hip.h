/*
 * rodygs_hip.h -- C-ABI of librodygs_hip.so: the MI355X (gfx950) Gaussian-rasterizer hot path of RoDyGS.
 *
 * Drop-in boundary.  The reference (pure Python) reaches this arithmetic through three un-vendored native
 * extensions; the entry points below are what a ctypes binding of those extensions' surfaces needs:
 *
 *   rdg_rasterize_forward / rdg_rasterize_backward
 *       replace  diff_gauss_pose.GaussianRasterizer(raster_settings)(means3D, means2D, shs, colors_precomp,
 *                opacities, scales, rotations, cov3Ds_precomp, viewmatrix)
 *       called at /root/reference/src/trainer/renderer.py:65,87-101,
 *                 /root/reference/src/model/rodygs_static.py:238,262-281,
 *                 /root/reference/src/evaluator/eval.py:135,157-176   (settings built at renderer.py:50-63)
 *   rdg_deform_forward / rdg_deform_backward
 *       replace the per-Gaussian part of DynRoDyGS.get_gaussian_deformation
 *                 /root/reference/src/model/rodygs_dynamic.py:122-138 (coeff @ (B(t) - B_table[birth]))
 *   rdg_dist2_knn3
 *       replaces simple_knn._C.distCUDA2 called at /root/reference/src/model/rodygs_static.py:130-133
 *   rdg_sort_pairs, rdg_preprocess_forward, rdg_bin_forward
 *       stage entry points (the same kernels rdg_rasterize_forward runs) exposed for the bit-exact
 *       tile-key / sort-order parity tests.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; all float tensors are contiguous f32;
 *   - the library never allocates or frees device memory: the caller (PyTorch caching allocator) owns every
 *     buffer, including the three opaque workspaces whose sizes come from rdg_*_bytes();
 *   - every launch goes to the hipStream_t passed as `stream` (void* here so the header is plain C);
 *     no call synchronises the host except where stated;
 *   - return value: 0 = ok, negative = error; rdg_last_error() gives a thread-local message;
 *     no C++ exception crosses the ABI;
 *   - matrices use the reference's "glm storage": viewmatrix = W2C^T, projmatrix = P^T (projection only),
 *     i.e. flat[c*4+r] = M[r][c]  (renderer.py:57,97-99).
 */
#ifndef RODYGS_HIP_H
#define RODYGS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDG_ABI_VERSION 7
#define RDG_MAX_VIEWS 16   /* cameras per step in the *_views entry points */
#define RDG_ADAM_MAX_SEGS 12 /* parameter groups per rdg_adam_step_multi launch */

/* Mirror of GaussianRasterizationSettings (renderer.py:50-63) + sizes. Host struct, passed by pointer. */
typedef struct RdgRasterSettings {
    int32_t P;               /* number of Gaussians                                         */
    int32_t M;               /* SH coefficients stored per Gaussian (shs is [P,M,3]); 0 if colors_precomp */
    int32_t sh_degree;       /* active degree 0..3                                           */
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    int32_t prefiltered;
    int32_t debug;
    int32_t enable_cov_grad; /* pose-gradient gates (SURVEY.md §7 open question 4)           */
    int32_t enable_sh_grad;
    int32_t render_normal;   /* 1: composite the normal channels (default); 0: leave them zero */
    int32_t bin_mode;        /* tile binning algorithm of THIS forward: 0 = bucket binning (default: count / scan / scatter
                              * into per-tile buckets + a per-tile sort), 1 = radix binning, depth first: the P Gaussians are
                              * sorted by depth ONCE (32-bit keys), their tile instances emitted in that order as (tile id,
                              * Gaussian id) pairs and partitioned stably by tile id (ceil(log2 tiles / 8) passes over 8 B per
                              * pair).  Same result bit for bit; bucket binning is faster on ordinary frames, the radix path has
                              * no atomics and no per-tile work, so its time does not depend on how the instances are spread
                              * over the tiles (a tile holding 200 k instances).  Callers pick it from num_rendered[1] of the
                              * previous frame.  bin_mode 1 keeps its per-Gaussian sort scratch (4 arrays of P uint32 + block
                              * sums) inside the second key buffer of the binning workspace: `capacity` must be at least
                              * 2 P + (P / 256 + 320) / 2 (the hosts of this package pass >= 4 P + 4096); a smaller capacity
                              * fails with "radix binning: capacity too small".                                            */
    int32_t num_rendered_stats; /* 1: num_rendered points to int32[2] and the binning stage also writes
                              * [1] = largest number of instances in one tile                                    */
    int32_t list_hints;      /* what the previous frame of this shape says about tile-list lengths (num_rendered[1]); speed
                              * hints, never the result:
                              *   bit 0: lists longer than 4096 instances are composited (forward AND the backward that
                              *          follows: keep the struct) by several workgroups each -- per-segment partial
                              *          composites + an ordered combine (three extra launches); 0: one workgroup walks
                              *          every list.  Same result up to the association of the transmittance product.  */
    int32_t grad_rows_zeroed; /* backward calls only: 1 = the per-Gaussian gradient rows at the head of grad_ws are already
                              * zero (the forward of this frame cleared them, zero_grad_ws below), so the backward does not
                              * launch its own fill; 0: the backward clears them itself (a repeated backward through the
                              * same graph must say 0: the first one consumed the zeros)                               */
    int32_t densify_row0;    /* backward calls only: first Gaussian row the densify_* arrays below cover (see there)   */
    void* zero_grad_ws;      /* forward calls only, optional: an rdg_grad_bytes(P) workspace whose gradient rows the
                              * compositing forward clears while it runs -- that kernel is instruction-bound and its memory
                              * pipeline idle, so the fill is free there, against a 12 us launch of its own at the head of
                              * the backward (P = 1 M).  NULL: off.                                                   */
    int32_t* num_rendered_host; /* forward calls only, optional, with num_rendered_stats = 1: a device-accessible HOST
                              * int32[2] (pinned memory) that the binning stage also writes (D, largest tile list) -- a
                              * caller that checks D after the fact (no host wait per frame) presets it to -1 and polls it;
                              * no copy, no event on the stream.  NULL: off.                                            */
    /* Backward calls only, optional (any of the three may be NULL): the per-iteration densification statistics of the
     * reference's train loop, updated by the per-Gaussian backward kernel, which holds dL/dmean2D and the radius of every
     * Gaussian in registers -- no extra pass, no host sync, no boolean-mask indexing.  For every Gaussian i with
     * radii[i] > 0 (the reference's visibility_filter, /root/reference/src/trainer/renderer.py:111) and
     * densify_row0 <= i < densify_row0 + densify_rows, with j = i - densify_row0:
     *     densify_max_radii[j]  = max(densify_max_radii[j], (float)radii[i])   (src/trainer/rodygs.py:334-338)
     *     densify_grad_accum[j] += hypot(dL/dmean2D[i].x, dL/dmean2D[i].y)       (rodygs.py:322-324, rodygs_static.py:317-318)
     *     densify_denom[j]      += 1                                            (rodygs_static.py:319)
     * The row window is the reference's slicing of the concatenated static || dynamic cloud (rodygs.py:320-332): the static
     * sub-step keeps rows [0, n_static), the dynamic one rows [n_static, P).  The arrays are float32 [densify_rows] (the
     * reference's [P,1] accumulators are the same memory).  One update per backward CALL: a caller that runs backward
     * twice through one graph (retain_graph) passes them once.  rdg_densify_stats() is the stand-alone form.          */
    float* densify_grad_accum;
    float* densify_denom;
    float* densify_max_radii;
    int32_t densify_rows;
    int32_t cull;            /* forward calls (and the backward that follows: keep the struct): which (tile, Gaussian) instances
                              * the per-Gaussian stage hands to the binning stage.
                              *   0: the reference's rule -- every tile of the square of half-width radius = ceil(3 sqrt(lambda_max))
                              *      around the splat's pixel centre.  The exported (tile | depth) key stream, the sorted
                              *      Gaussian indices, the tile ranges and D are then the reference algorithm's bit for bit
                              *      (rdg_bin_forward, the parity tests).
                              *   1: that square intersected with the tiles holding a pixel centre inside the axis-aligned box
                              *      of the splat's "alpha >= 1/255" ellipse (with a safety margin): the instances dropped
                              *      could not blend in any pixel of their tile, so images, final_T and every gradient are
                              *      those of cull = 0 (bit for bit with the deterministic backward; n_contrib counts list
                              *      positions, so it names the same last contributor at a smaller position); each tile's
                              *      list is a SUBSEQUENCE of the reference list; radii (the reference's visibility filter,
                              *      /root/reference/src/trainer/renderer.py:111) and the densification statistics are
                              *      unchanged; num_rendered_dev receives the number of instances actually binned.  27-32 %
                              *      fewer instances on the benchmark frames.                                              */
    int32_t* num_rendered_max; /* forward calls only, optional: a device int32 that receives max(itself, D) of every forward
                              * handed the same pointer -- a STICKY record for callers that do not look at every frame's
                              * num_rendered (a captured hipGraph replays the forward many times into one num_rendered_dev;
                              * a frame in the middle that outgrew the capacity was rendered empty and would go unnoticed).
                              * The caller zeroes it, reads it when it likes and compares with the capacity.  NULL: off. */
    void* aux_stream;        /* backward calls only, optional (a hipStream_t): the two small launches that finish the pose
                              * gradient (per-workgroup rows -> dL_dviewmatrix) go to THIS stream, behind an event recorded on
                              * `stream` after the per-Gaussian kernel, instead of onto `stream` itself.  The caller joins the
                              * streams before it reads dL_dviewmatrix (and keeps grad_ws / dL_dviewmatrix alive until then).
                              * Meant for a captured hipGraph, where the fork becomes two independent branches (the pose chain
                              * next to the deformation / MLP backward); in an eagerly launched step the cross-stream waits
                              * cost more than the 12 us they hide.  NULL: everything on `stream`.                          */
} RdgRasterSettings;

/* Creates the (thread-local) event RdgRasterSettings.aux_stream forks through, so that the first backward with an aux_stream can
 * run under a stream capture without creating one there.  Optional: the backward creates it on first use otherwise.          */
int rdg_pose_fork_prepare(void);

/* stage ids for rdg_stage_time_ms() */
enum {
    RDG_STAGE_PREPROCESS = 0,
    RDG_STAGE_SCAN_DUP = 1,
    RDG_STAGE_SORT = 2,
    RDG_STAGE_RANGES = 3,
    RDG_STAGE_RENDER_FWD = 4,
    RDG_STAGE_RENDER_BWD = 5,
    RDG_STAGE_PREPROCESS_BWD = 6,
    RDG_STAGE_DEFORM_FWD = 7,
    RDG_STAGE_DEFORM_BWD = 8,
    RDG_STAGE_ADAM = 9,
    RDG_STAGE_LOSS_FWD = 10,
    RDG_STAGE_LOSS_BWD = 11,
    RDG_STAGE_MLP_FWD = 12,
    RDG_STAGE_MLP_BWD = 13,
    RDG_STAGE_COUNT = 14
};

int rdg_abi_version(void);
/* sizeof(RdgRasterSettings) as this library was compiled: a binding checks its own mirror of the struct against it */
size_t rdg_settings_bytes(void);
const char* rdg_last_error(void);

/* ---- workspace sizes (bytes) -------------------------------------------------------------------------- */
size_t rdg_geom_bytes(int32_t P);                             /* per-Gaussian state kept for backward      */
size_t rdg_binning_bytes(int64_t capacity, int32_t n_tiles);  /* keys/values (double-buffered) + sort tables */
size_t rdg_image_bytes(int32_t H, int32_t W);                 /* final_T, n_contrib, tile ranges           */

/* ---- rasterizer -------------------------------------------------------------------------------------------
 * Forward.  Exactly one of shs / colors_precomp and one of (scales, rotations) / cov3D_precomp is non-NULL.
 * `capacity` is the number of (tile, Gaussian) instances binning_ws was sized for.  num_rendered_dev receives
 * the true instance count D; if D > capacity the binning and compositing stages are skipped on the device
 * (outputs untouched) and the caller must call again with capacity >= D.  The call itself never syncs.
 * Outputs: out_color[3,H,W], out_depth[1,H,W], out_normal[3,H,W], out_alpha[1,H,W], radii[P] (int32).       */
int rdg_rasterize_forward(const RdgRasterSettings* s_host, const float* bg, const float* means3D,
                          const float* shs, const float* colors_precomp, const float* opacities,
                          const float* scales, const float* rotations, const float* cov3D_precomp,
                          const float* viewmatrix, const float* projmatrix, void* geom_ws, void* binning_ws,
                          int64_t capacity, void* image_ws, float* out_color, float* out_depth,
                          float* out_normal, float* out_alpha, int32_t* radii, int32_t* num_rendered_dev,
                          void* stream);

/* Backward.  grad_out_* may be NULL (treated as zero).  All dL_* outputs must be zero-initialised by the
 * caller EXCEPT none: the library zeroes what it accumulates into.  dL_dmeans2D is [P,3] (z unused = 0),
 * dL_dviewmatrix is [16] in the same glm storage as viewmatrix.  grad_ws: rdg_grad_bytes(P) scratch.
 * grad_out_normal [3,H,W]: the per-Gaussian normals are constants of the graph, so a gradient of the normal image
 * reaches the inputs only through the compositing weights (opacity, conic, position); it needs a forward that
 * composited normals (render_normal = 1).                                                                    */
size_t rdg_grad_bytes(int32_t P);
int rdg_rasterize_backward(const RdgRasterSettings* s_host, const float* bg, const float* means3D,
                           const float* shs, const float* colors_precomp, const float* opacities,
                           const float* scales, const float* rotations, const float* cov3D_precomp,
                           const float* viewmatrix, const float* projmatrix, const int32_t* radii,
                           const void* geom_ws, const void* binning_ws, int64_t capacity, const void* image_ws,
                           const float* grad_out_color, const float* grad_out_depth,
                           const float* grad_out_alpha, const float* grad_out_normal, void* grad_ws,
                           float* dL_dmeans3D, float* dL_dmeans2D,
                           float* dL_dshs, float* dL_dcolors, float* dL_dopacities, float* dL_dscales,
                           float* dL_drotations, float* dL_dcov3D, float* dL_dviewmatrix, void* stream);

/* ---- the rasterizer in two halves (Gaussian-sharded frame-DP, rodygs_amd/sharded.py) ----------------------------
 * rdg_rasterize_forward == rdg_preprocess_forward + rdg_composite_forward, rdg_rasterize_backward ==
 * rdg_composite_backward + rdg_preprocess_backward.  The halves meet in two row formats that can cross the wire:
 *   - splat records: the first P*64 bytes of geom_ws, one 64-B row per Gaussian
 *       (px, py, conic_a, conic_b | conic_c, opacity, depth, radius as int bits | r, g, b, 1 / cov2D_yy | nx, ny, nz, -)
 *       -- 1 / cov2D_yy = conic_c - conic_b^2 / conic_a, carried because that difference cancels on needle-shaped footprints;
 *   - gradient rows: the first P*64 bytes of grad_ws, one 16-float row per Gaussian
 *       (moments of t = G dL/dG over the pixels about (w, dy), w = (px - x) + beta (py - y), beta = conic_b / conic_a of the
 *        record -- the skew coordinate the exponent is evaluated in, so that needle-shaped footprints lose nothing to
 *        cancellation: sum(t w), sum(t dy), -1/2 sum(t w^2), -sum(t w dy), -1/2 sum(t dy^2); these five divided by the
 *        Gaussian's opacity, which rdg_preprocess_backward multiplies back in when it turns them into dL/dmean2D and
 *        dL/dconic --, dL/dopacity, dL/drgb, dL/ddepth, pad).
 * A rank that owns a slice of the Gaussians runs the per-Gaussian halves for every camera of the step; the rank that
 * owns a camera gathers that camera's records into one geom_ws, calls rdg_geom_from_records (tile counts, radii and
 * the instance count D rebuilt from the records alone) and runs the compositing halves over all P records.        */
int rdg_geom_from_records(const RdgRasterSettings* s_host, void* geom_ws, int32_t* radii, int32_t* num_rendered_dev,
                          void* stream);
int rdg_composite_forward(const RdgRasterSettings* s_host, const float* bg, const void* geom_ws, const int32_t* radii,
                          void* binning_ws, int64_t capacity, void* image_ws, int32_t* num_rendered_dev,
                          float* out_color, float* out_depth, float* out_normal, float* out_alpha, void* stream);
/* zeroes the gradient rows of grad_ws, then accumulates the compositing backward into them                         */
int rdg_composite_backward(const RdgRasterSettings* s_host, const float* bg, const void* geom_ws,
                           const void* binning_ws, int64_t capacity, const void* image_ws,
                           const float* grad_out_color, const float* grad_out_depth, const float* grad_out_alpha,
                           const float* grad_out_normal, void* grad_ws, void* stream);
/* Deterministic form of rdg_composite_backward (SURVEY.md section 5b "deterministic mode: no float atomics -> bit-
 * reproducible"; the reference's un-vendored rasterizer accumulates with float atomics, whose order changes from run to
 * run): every wave STORES its per-(tile, splat) totals to its own quarter of a 256-B row per list position in det_ws
 * (rdg_det_bytes(n_instances) bytes, n_instances >= the frame's num_rendered; zeroed by this call; rows in Gaussian-major
 * order, addressed through an exclusive scan of tiles_touched kept at the end of grad_ws), then one pass adds them up per
 * Gaussian in the order of its tile rectangle.  Same gradient rows as rdg_composite_backward up to the
 * rounding of a different summation order; two calls on the same inputs give the same bits.                          */
size_t rdg_det_bytes(int64_t n_instances);
int rdg_composite_backward_det(const RdgRasterSettings* s_host, const float* bg, const void* geom_ws,
                               const void* binning_ws, int64_t capacity, const void* image_ws,
                               const float* grad_out_color, const float* grad_out_depth, const float* grad_out_alpha,
                               const float* grad_out_normal, void* grad_ws, void* det_ws, int64_t n_instances,
                               void* stream);
/* gradient rows (grad_ws) -> input gradients; geom_ws / radii are those of THIS rank's rdg_preprocess_forward      */
int rdg_preprocess_backward(const RdgRasterSettings* s_host, const float* means3D, const float* shs,
                            const float* colors_precomp, const float* opacities, const float* scales,
                            const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                            const float* projmatrix, const int32_t* radii, const void* geom_ws, void* grad_ws,
                            float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs, float* dL_dcolors,
                            float* dL_dopacities, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                            float* dL_dviewmatrix, void* stream);

/* rdg_preprocess_backward with the optimiser step of the SH features folded in ("optimizer in backward"): the kernel
 * holds a wave's 64 gradient rows of dL/dshs in LDS; instead of writing them out for rdg_adam_step_* to read back, it
 * applies the Adam update to shs [P,M,3] (IN PLACE), sh_exp_avg and sh_exp_avg_sq right there -- same arithmetic, same
 * bits as rdg_adam_step_multi on a segment (row_len = 3M, head_len, lr_head, lr_tail) fed the gradient it would have written;
 * dL/dshs itself is never materialised.  The other outputs are those of rdg_preprocess_backward.                      */
int rdg_preprocess_backward_adam(const RdgRasterSettings* s_host, const float* means3D, float* shs,
                                 const float* opacities, const float* scales, const float* rotations,
                                 const float* viewmatrix, const float* projmatrix, const int32_t* radii,
                                 const void* geom_ws, void* grad_ws, float* dL_dmeans3D, float* dL_dmeans2D,
                                 float* dL_dopacities, float* dL_dscales, float* dL_drotations, float* dL_dviewmatrix,
                                 float* sh_exp_avg, float* sh_exp_avg_sq, int32_t head_len, float lr_head, float lr_tail,
                                 double beta1, double beta2, float eps, int32_t step, void* stream);

/* The per-Gaussian halves for the nviews (<= RDG_MAX_VIEWS) cameras of one step over the SAME P Gaussians (s_host->P),
 * whose time-dependent inputs are stacked with a row stride of stride_rows (multiple of 256, >= P): means3D
 * [nviews,stride,3], rotations [nviews,stride,4]; shs [P,M,3], scales [P,3] and opacities [P] are shared;
 * viewmatrices [nviews,16].  geom_ws = rdg_geom_bytes(nviews*stride), radii [nviews*stride], grad_ws = rdg_grad_bytes(nviews*stride):
 * camera v owns rows [v*stride, v*stride + P) of each, so the records / gradient rows of all cameras are one
 * contiguous buffer that an equal-split all-to-all can send / fill.  Rows [P, stride) of a camera are never written:
 * zero the records once and they stay invisible.  Backward: dL_dshs [P,M,3] is overwritten with the SUM over the
 * cameras; the other outputs are stacked per camera (dL_dmeans3D / dL_dmeans2D / dL_dscales [nviews,stride,3],
 * dL_drotations [nviews,stride,4], dL_dopacities [nviews,stride]) and overwritten on rows [0, P) of each camera;
 * dL_dviewmatrices [nviews,16].  Both run as ONE launch that loops over the cameras per Gaussian, so the SH rows are
 * read (and their gradients written) once per step instead of once per camera.           */
int rdg_preprocess_forward_views(const RdgRasterSettings* s_host, int32_t nviews, int32_t stride_rows,
                                 const float* means3D, const float* shs, const float* opacities, const float* scales,
                                 const float* rotations, const float* viewmatrices, const float* projmatrix,
                                 void* geom_ws, int32_t* radii, void* stream);
/* The same launch restricted to rows [row0, row0 + s_host->P) of the slice (row0 a multiple of 256; every pointer is
 * that of row 0): lets the owner stage run in row chunks so that the record exchange of a chunk (all-to-all #1 of the
 * sharded step) overlaps the projection of the next.                                        */
int rdg_preprocess_forward_views_rows(const RdgRasterSettings* s_host, int32_t nviews, int32_t stride_rows,
                                      int32_t row0, const float* means3D, const float* shs, const float* opacities,
                                      const float* scales, const float* rotations, const float* viewmatrices,
                                      const float* projmatrix, void* geom_ws, int32_t* radii, void* stream);
int rdg_preprocess_backward_views(const RdgRasterSettings* s_host, int32_t nviews, int32_t stride_rows,
                                  const float* means3D, const float* shs, const float* opacities, const float* scales,
                                  const float* rotations, const float* viewmatrices, const float* projmatrix,
                                  const int32_t* radii, const void* geom_ws, void* grad_ws, float* dL_dmeans3D,
                                  float* dL_dmeans2D, float* dL_dshs, float* dL_dopacities, float* dL_dscales,
                                  float* dL_drotations, float* dL_dviewmatrices, void* stream);

/* ---- stage entry points (bit-exact parity tests) -------------------------------------------------------- */
/* Runs only the per-Gaussian stage; fills geom_ws, radii, and *num_rendered_dev.                            */
int rdg_preprocess_forward(const RdgRasterSettings* s_host, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* opacities, const float* scales,
                           const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                           const float* projmatrix, void* geom_ws, int32_t* radii, int32_t* num_rendered_dev,
                           void* stream);
/* Copies out per-Gaussian state for inspection: depth[P], xy[P,2], conic_opacity[P,4], rgb[P,3],
 * tiles_touched[P] (uint32).  Any pointer may be NULL.                                                      */
int rdg_geom_export(int32_t P, const void* geom_ws, float* depth, float* xy, float* conic_opacity, float* rgb,
                    float* normal, uint32_t* tiles_touched, void* stream);
/* Copies out the per-pixel compositing state the backward replays from: final_T[H*W] (transmittance left after
 * the last blended splat) and n_contrib[H*W] (1 + list position of the last blended splat; their sum is S, the
 * number of pixel-splat pairs the forward walked).  Either pointer may be NULL.                               */
int rdg_image_export(int32_t H, int32_t W, const void* image_ws, float* final_T, uint32_t* n_contrib, void* stream);
/* duplicateWithKeys + sort + tile ranges.  keys_unsorted/vals_unsorted/keys_sorted/vals_sorted (capacity
 * entries each) and ranges[n_tiles,2] are optional copies for the tests.                                    */
int rdg_bin_forward(const RdgRasterSettings* s_host, const void* geom_ws, const int32_t* radii, void* binning_ws,
                    int64_t capacity, void* image_ws, int32_t* num_rendered_dev, uint64_t* keys_unsorted,
                    uint32_t* vals_unsorted, uint64_t* keys_sorted, uint32_t* vals_sorted, uint32_t* ranges,
                    void* stream);
/* Stable LSD radix sort of n (key,value) pairs on key bits [0,end_bit).  n is read from *n_dev on the device
 * (clamped to capacity).  tmp_ws: rdg_sort_tmp_bytes(capacity).  Result is written back into keys/vals.     */
size_t rdg_sort_tmp_bytes(int64_t capacity);
int rdg_sort_pairs(uint64_t* keys, uint32_t* vals, int64_t capacity, const int32_t* n_dev, int32_t end_bit,
                   void* tmp_ws, void* stream);

/* ---- time deformation ----------------------------------------------------------------------------------------
 * delta[p,:] = sum_b coeff[p,b] * (basis_t[b,:] - table[time_ind[p], b, :]);  out_xyz = delta[:, :3]*spatial_scale,
 * out_rot = delta[:, 3:7].  coeff [P,B] (B = 16), basis_t [B,7], table [Tu,B,7], time_ind [P] int64.
 * table may be NULL (inverse_motion = False).                                                                */
int rdg_deform_forward(int32_t P, int32_t B, int32_t Tu, const float* coeff, const int64_t* time_ind,
                       const float* basis_t, const float* table, float spatial_scale, float* out_xyz,
                       float* out_rot, void* stream);
/* d_coeff [P,B]; d_basis_t [B,7] and d_table [Tu,B,7] are zeroed then accumulated by the library.
 * order: optional int32 [P] permutation that sorts the Gaussians by time_ind (it changes only when time_ind
 * does, so the caller caches it).  With it (and B == 16) the two dB reductions run on the matrix cores
 * (v_mfma_f32_16x16x4_f32, one per 4 Gaussians); NULL selects the order-free LDS-atomic form.
 * inv_order (int32 [P], inv_order[order[i]] = i) + sorted_ws (rdg_deform_sorted_ws_bytes(P), 16-B aligned):
 * optional; with them the gradient rows are re-laid once in birth-sorted order so that the dB reduction
 * streams them instead of gathering 12/16/8-byte pieces (either may be NULL).
 * With a table and the workspace the reduction uses NO float atomics (bit-reproducible): every wave stores its total
 * per birth index and a last stage adds them in wave order.  seg_start (int32 [Tu+1], optional, cached with `order`):
 * seg_start[u] = first position of birth index u in the sorted sequence, seg_start[Tu] = P; NULL = the library finds
 * the boundaries by binary search (slower: ~25 us at P = 1 M).                                               */
size_t rdg_deform_sorted_ws_bytes(int32_t P);
int rdg_deform_backward(int32_t P, int32_t B, int32_t Tu, const float* coeff, const int64_t* time_ind,
                        const float* basis_t, const float* table, float spatial_scale, const float* g_xyz,
                        const float* g_rot, float* d_coeff, float* d_basis_t, float* d_table,
                        const int32_t* order, const int32_t* inv_order, const int32_t* seg_start, void* sorted_ws,
                        void* stream);

/* ---- time-deformation MLP on the matrix cores (csrc/rdg_mlp.hip) --------------------------------------------
 * MLPBasisNetwork.batch_inference / the basis part of forward (/root/reference/src/model/rodygs_dynamic.py:296-327):
 * x [NR,D0] time embeddings -> timenet (D0-H-H-H/2, GELU) -> NB heads (H/2-H/4-OUT, GELU) -> out [NR,NB,OUT].
 * Weights in nn.Linear layout ([out,in]); heads stacked: hw1 [NB,H/4,H/2], hb1 [NB,H/4], hw2 [NB,OUT,H/4],
 * hb2 [NB,OUT].  ws: rdg_mlp_ws_bytes() scratch holding the activations; it must survive until backward.
 * Every product runs on v_mfma_f32_16x16x4_f32 (exact f32).  Backward overwrites all d* outputs.            */
size_t rdg_mlp_ws_bytes(int32_t NR, int32_t H, int32_t NB);
int rdg_mlp_forward(int32_t NR, int32_t D0, int32_t H, int32_t NB, int32_t OUT, const float* x, const float* W0,
                    const float* b0, const float* W1, const float* b1, const float* W2, const float* b2,
                    const float* hw1, const float* hb1, const float* hw2, const float* hb2, void* ws, float* out,
                    void* stream);
int rdg_mlp_backward(int32_t NR, int32_t D0, int32_t H, int32_t NB, int32_t OUT, const float* x, const float* W1,
                     const float* W2, const float* hw1, const float* hw2, void* ws, const float* g_out, float* dW0,
                     float* db0, float* dW1, float* db1, float* dW2, float* db2, float* dhw1, float* dhb1, float* dhw2,
                     float* dhb2, void* stream);

/* ---- fused dynamic getter: deformation + activations in one pass each way ---------------------------------------
 * What get_GS_properties (/root/reference/src/trainer/rodygs.py:68-113) assembles from get_gaussian_deformation
 * (rodygs_dynamic.py:122-138) and the model getters (rodygs_static.py:82-105), for B = 16 motion bases:
 *   means3D = xyz + spatial_scale * (coeff . dB)[0:3],  scales = exp(scaling),
 *   rots = normalize(rotation) + (coeff . dB)[3:7],     opac = sigmoid(opacity),   dB = bases[Tu] - bases[birth]
 * bases [Tu+1,16,7]: the Tu birth-time rows of the motion table followed by B(t).  rdg_dyn_getter_supported(B, Tu)
 * tells whether the table fits the kernel (else use rdg_deform_* + rdg_activate_*).  Backward overwrites the five
 * parameter gradients and d_bases [Tu+1,16,7]; g_* may be NULL (no upstream gradient); order / inv_order /
 * sorted_ws as for rdg_deform_backward (all required here), seg_start optional.                                  */
int rdg_dyn_getter_supported(int32_t B, int32_t Tu);
int rdg_dyn_getter_forward(int32_t P, int32_t Tu, const float* coeff, const int64_t* time_ind, const float* bases,
                           float spatial_scale, const float* xyz, const float* scaling, const float* rotation,
                           const float* opacity, float* means3D, float* scales, float* rots, float* opac, void* stream);
int rdg_dyn_getter_backward(int32_t P, int32_t Tu, const float* coeff, const int64_t* time_ind, const float* bases,
                            float spatial_scale, const float* scaling, const float* rotation, const float* opacity,
                            const float* g_means3D, const float* g_scales, const float* g_rots, const float* g_opac,
                            float* d_xyz, float* d_scaling, float* d_rotation, float* d_opacity, float* d_coeff,
                            float* d_bases, const int32_t* order, const int32_t* inv_order, const int32_t* seg_start,
                            void* sorted_ws, void* stream);

/* The fused getter for the nviews (<= RDG_MAX_VIEWS) camera times of one step over the same P Gaussians (sharded
 * frame-DP): bases_all [nviews,Tu+1,16,7] (table rows identical in every view); means3D / rots are stacked per view
 * with a row stride of stride_rows (>= P), scales [P,3] / opac [P] do not depend on the time and are written once.
 * Backward: g_* stacked per view [nviews,stride,*]; the five parameter gradients [P,*] are OVERWRITTEN with the sum
 * over the views; d_bases_all [nviews,Tu+1,16,7] per view; sorted_ws = rdg_deform_sorted_views_ws_bytes(P, nviews). */
int rdg_dyn_getter_views_supported(int32_t B, int32_t Tu, int32_t nviews);
size_t rdg_deform_sorted_views_ws_bytes(int32_t P, int32_t nviews);
int rdg_dyn_getter_views_forward(int32_t P, int32_t Tu, int32_t nviews, int32_t stride_rows, const float* coeff,
                                 const int64_t* time_ind, const float* bases_all, float spatial_scale, const float* xyz,
                                 const float* scaling, const float* rotation, const float* opacity, float* means3D,
                                 float* scales, float* rots, float* opac, void* stream);
int rdg_dyn_getter_views_backward(int32_t P, int32_t Tu, int32_t nviews, int32_t stride_rows, const float* coeff,
                                  const int64_t* time_ind, const float* bases_all, float spatial_scale,
                                  const float* scaling, const float* rotation, const float* opacity,
                                  const float* g_means3D, const float* g_scales, const float* g_rots,
                                  const float* g_opac, float* d_xyz, float* d_scaling, float* d_rotation,
                                  float* d_opacity, float* d_coeff, float* d_bases_all, const int32_t* order,
                                  const int32_t* inv_order, void* sorted_ws, void* stream);

/* ---- simple_knn ------------------------------------------------------------------------------------------ */
size_t rdg_knn_tmp_bytes(int32_t P);
/* out[p] = mean squared distance from points[p] to its 3 nearest other points.                              */
int rdg_dist2_knn3(int32_t P, const float* points, float* out, void* tmp_ws, void* stream);

/* ---- densify / prune row surgery (/root/reference/src/trainer/rodygs_static.py:170-319, src/trainer/utils.py:15-95) ----
 * dst[i,:] = src[idx[i],:] for i < n_new rows of row_len floats; idx[i] < 0 writes zeros (Adam moments of new rows). */
int rdg_gather_rows(int64_t n_new, int32_t row_len, const int64_t* idx, const float* src, float* dst, void* stream);
/* Split children (densify_and_split_update_attributes, rodygs_static.py:182-216): for child i of parent p = parent[i],
 * xyz_out[i] = xyz[p] + R(rotation[p] / |rotation[p]|) (exp(scaling[p]) * z[i]),
 * scaling_out[i] = log(exp(scaling[p]) / (0.8 N)); z[n,3] are standard-normal draws supplied by the caller.        */
int rdg_split_children(int64_t n, int32_t N, const int64_t* parent, const float* xyz, const float* scaling,
                       const float* rotation, const float* z, float* xyz_out, float* scaling_out, void* stream);
/* rank[i] = (number of non-zero bytes in mask[0..i]) - 1, int64 [n]: the position of row i in the selection `mask` -- what
 * the reference's boolean indexing (xyz[selected_pts_mask], rodygs_static.py:176-215, utils.py:40-63) computes inside the
 * framework; with the selection's size known, idx[rank[i]] = i over the set rows is the compacted row list.  n < 2^32;
 * ws: rdg_mask_rank_ws_bytes(n).                                                                                       */
size_t rdg_mask_rank_ws_bytes(int64_t n);
int rdg_mask_rank(int64_t n, const uint8_t* mask, int64_t* rank, void* ws, void* stream);

/* The per-iteration densification statistics as a stand-alone launch, for a caller that follows the reference's flow
 * (viewspace_point_tensor.grad and radii in hand after loss.backward(), /root/reference/src/trainer/rodygs.py:316-341,
 * add_densification_stats /root/reference/src/trainer/rodygs_static.py:317-319): for the n rows starting at row0 of
 * dL_dmeans2D [P,3] / radii [P] (the static or the dynamic part of the concatenated cloud), where radii > 0:
 * max_radii[j] = max(max_radii[j], radii), grad_accum[j] += |dL_dmeans2D[:2]|, denom[j] += 1 (j = row - row0; float32 [n]
 * each, any may be NULL).  Replaces five boolean-mask indexing ops (each a nonzero + host sync in the framework).  The
 * same update rides along in rdg_preprocess_backward* for free (RdgRasterSettings.densify_*).                          */
int rdg_densify_stats(int64_t n, int64_t row0, const float* dL_dmeans2D, const int32_t* radii, float* grad_accum,
                      float* denom, float* max_radii, void* stream);

/* Z-curve (Morton) index of every row of xyz [n,3] inside the box lo_hi = (lo_x, lo_y, lo_z, hi_x, hi_y, hi_z) (device
 * floats): `bits` (<= 21) bits per axis, x / y / z interleaved from bit 0 -- the row order a trainer built on this package
 * keeps its cloud in (rodygs_amd/layout.py; the reference keeps whatever order its point cloud and densification appends
 * produce, /root/reference/src/trainer/rodygs_static.py:218-299).  One launch where the framework expression takes ~35.   */
int rdg_morton_codes(int64_t n, const float* xyz, const float* lo_hi, int32_t bits, int64_t* codes, void* stream);

/* ThreeDGSTrainer.reset_opacity (/root/reference/src/trainer/rodygs_static.py:151-160) with the optimizer surgery of
 * replace_tensor_to_optimizer (/root/reference/src/trainer/utils.py:15-32), in place on a flat-bucket segment:
 * opacity_logit[i] = inverse_sigmoid(min(sigmoid(opacity_logit[i]), max_opacity)) (reference: 0.01), and both Adam
 * moments of the segment zeroed; the optimiser's step counter is kept, as the reference keeps state["step"].       */
int rdg_reset_opacity(int64_t n, float max_opacity, float* opacity_logit, float* exp_avg, float* exp_avg_sq,
                      void* stream);

/* Distance-preserving term of RigidityLoss (/root/reference/src/trainer/losses.py:293-358) without the [t,n,K,3]
 * intermediates, on the layout of the reference's own translation tensor (`own` [n, nt, 3], losses.py:305-306): positions
 * Gaussian-major, a group of lanes per Gaussian with lane = drawn time, every edge end one coalesced row read.
 * rdg_rigidity_pack_rows: P3 [n][nt][3] = own[order[s]] + canon[order[s]] (canon [n,3]; order NULL = identity): the sample
 * re-laid along a cache-friendly stored order.
 * rdg_rigidity_dp_rows: P3 as above; nn_idx [n,K] neighbour lists and the reverse adjacency (rev_off [n+1], rev_edge [n*K] =
 * edge ids s*K+k sorted by destination) in stored ids; orig = order (stored position -> row of the reference's sample order,
 * which defines the pairing f / nt), rank = its inverse; d2 / d_d2 [n,K] in that original order; IG: workspace of n*K*nt floats;
 * RA (optional): workspace of n*nt*2 floats -- with it (and nt >= K) d_d2 is added up in a fixed order by a third launch
 * instead of through float atomics.  Writes loss_sum[0] (f64) = sum over (tau, i, k) of sqrt((gap - d2_flat[f / nt])^2 + eps^2),
 * d_d2, and the unscaled gradient in the CALLER's layout and row order: G_own [n,nt,3] (w.r.t. own) and G_canon [n,3] (its sum
 * over the times; may be NULL).                                                                                         */
int rdg_rigidity_pack_rows(int64_t n, int32_t nt, const float* own, const float* canon, const int64_t* order, float* P3,
                           void* stream);
int rdg_rigidity_dp_rows(int64_t n, int32_t K, int32_t nt, const float* P3, const int64_t* nn_idx, const float* d2,
                         const int64_t* rev_off, const int64_t* rev_edge, const int64_t* orig, const int64_t* rank, float eps,
                         double* loss_sum, float* IG, float* RA, float* d_d2, float* G_own, float* G_canon, void* stream);

/* The other two scatters of a RigidityLoss step through the same stored-order K-NN graph (X [n,3] = the sample's positions,
 * nbr [n,K] neighbour lists, rev_off [n+1] / rev_edge [n*K] reverse adjacency -- all in the stored (curve) order; orig [n] =
 * row of stored element s in the caller's order).  Every gradient row is written once, no float atomics.
 * rdg_graph_points_backward: the backward of knn_points(p, p, K) (pytorch3d; call site /root/reference/src/trainer/losses.py:235):
 *   g_dists [n,K] and d_pts [n,3] in the caller's order; same sum as rdg_knn_points_backward with queries == targets.
 * rdg_graph_surface: the "surface" term (/root/reference/src/trainer/losses.py:241-250): loss_sum[0] (f64) = sum_i
 *   || x_i - mean_k x_nn(i,k) + 1e-6 || and d_pts [n,3] (caller's order) = its gradient; U: workspace of n*3 floats.     */
int rdg_graph_points_backward(int64_t n, int32_t K, const float* X, const int64_t* nbr, const int64_t* rev_off,
                              const int64_t* rev_edge, const int64_t* orig, const float* g_dists, float* d_pts, void* stream);
int rdg_graph_surface(int64_t n, int32_t K, const float* X, const int64_t* nbr, const int64_t* rev_off,
                      const int64_t* rev_edge, const int64_t* orig, float* U, double* loss_sum, float* d_pts, void* stream);

/* ---- Pearson depth losses (GlobalPearsonDepthLoss / LocalPearsonDepthLoss, /root/reference/src/trainer/losses.py:108-182;
 *      pearson_depth_loss, /root/reference/src/utils/loss_utils.py:100-117) -------------------------------------------
 * n_boxes boxes of bh x bw pixels of the [H,W] depth images pred / gt; row0[n_boxes], col0[n_boxes] (int64, device)
 * are the top-left corners (NULL, NULL with n_boxes = 1, bh = H, bw = W is the global loss).  mask: optional
 * [H,W] bytes (torch.bool) multiplied into both images as the reference does; a box whose mask is empty contributes
 * nothing.  loss_out[0] = weight * sum_boxes (1 - corr_box)   (weight = 1 / n_corr for the local loss).
 * ws: rdg_pearson_ws_bytes(n_boxes), kept for backward.  Backward overwrites d_pred[H,W] with g_loss[0] * dL/dpred. */
size_t rdg_pearson_ws_bytes(int32_t n_boxes);
int rdg_pearson_depth_forward(int32_t H, int32_t W, int32_t n_boxes, int32_t bh, int32_t bw, const int64_t* row0,
                              const int64_t* col0, const float* pred, const float* gt, const uint8_t* mask, float eps,
                              float weight, void* ws, float* loss_out, void* stream);
int rdg_pearson_depth_backward(int32_t H, int32_t W, int32_t n_boxes, int32_t bh, int32_t bw, const int64_t* row0,
                               const int64_t* col0, const float* pred, const float* gt, const uint8_t* mask,
                               const void* ws, const float* g_loss, float* d_pred, void* stream);

/* ---- motion L1 + sparsity regularisers (MotionL1Loss, MotionSparsityLoss, /root/reference/src/trainer/losses.py:363-384) ----
 * coeff [P,1,B] (B = 16).  Forward: sums2[0] = sum |c|, sums2[1] = sum_p sum_b |c_pb| / (max_b |c_pb| + 1e-7) (f64; the
 * losses are these divided by P*B).  Backward: d_coeff (+)= g_loss[0] * d/dcoeff (w_l1 * L1 + w_sparsity * sparsity);
 * accumulate != 0 adds to d_coeff, 0 overwrites it; g_loss may be NULL (= 1).                                       */
int rdg_motion_reg_forward(int64_t P, int32_t B, const float* coeff, double* sums2, void* stream);
int rdg_motion_reg_backward(int64_t P, int32_t B, const float* coeff, const float* g_loss, float w_l1, float w_sparsity,
                            float* d_coeff, int32_t accumulate, void* stream);

/* MotionBasisRegularizaiton (/root/reference/src/trainer/losses.py:386-525) of the motion table [Tu,16,7], value AND gradient
 * in three tiny launches: loss[0] (f64) = mean_{t,b} w_b ||d^k transl|| + mean_{t,b} w_b ||I - d^k R(q)||_F with
 * k = degree + 1 forward differences over the time axis (degree < 0 drops that term; degrees 0..2), d_table [Tu,16,7]
 * = its gradient (overwritten).  w_host: the 16 per-basis weights (HOST array).  ws: rdg_basis_reg_ws_bytes(Tu).   */
size_t rdg_basis_reg_ws_bytes(int32_t Tu);
int rdg_basis_reg(int32_t Tu, int32_t B, int32_t transl_degree, int32_t rot_degree, const float* w_host,
                  const float* table, void* ws, double* loss, float* d_table, void* stream);

/* ---- pytorch3d.ops.knn_points / knn_gather (RigidityLoss, /root/reference/src/trainer/losses.py:235-331) ---- */
/* K nearest targets of every query: dists[Pq,K] squared Euclidean, ascending; idx[Pq,K] int64 target indices.
 * tmp_ws: rdg_knn_tmp_bytes(Pt).  queries == targets (same pointer, Pq == Pt) is the self query the reference
 * uses: every point is its own first neighbour (distance 0), as in pytorch3d.  1 <= K <= 32, K <= Pt.          */
int rdg_knn_points_forward(int32_t Pq, int32_t Pt, int32_t K, const float* queries, const float* targets, float* dists,
                           int64_t* idx, void* tmp_ws, void* stream);
/* Gradient of dists w.r.t. the points: d_queries[Pq,3] and d_targets[Pt,3] are overwritten (either may be NULL;
 * pass the same pointer for both when queries and targets are one tensor: the two contributions are summed).    */
int rdg_knn_points_backward(int32_t Pq, int32_t Pt, int32_t K, const float* queries, const float* targets,
                            const int64_t* idx, const float* g_dists, float* d_queries, float* d_targets, void* stream);
/* out[r,:] = x[idx[r],:] for n_rows = Pq*K flattened index rows of U floats; backward scatter-adds g into d_x
 * [n_src_rows,U], which it overwrites.                                                                          */
int rdg_knn_gather_forward(int64_t n_rows, int32_t U, const float* x, const int64_t* idx, float* out, void* stream);
int rdg_knn_gather_backward(int64_t n_rows, int32_t U, int64_t n_src_rows, const float* g, const int64_t* idx, float* d_x,
                            void* stream);

/* ---- fused Adam over a flat f32 parameter (used by bench.py's train step) ----------
 * torch.optim.Adam semantics; beta1 / beta2 are DOUBLES, as the Python floats torch receives: 1 - beta is formed in
 * double and rounded once (1 - float(0.999) is off by 4.7e-5 relative, which would bias exp_avg_sq).           */
int rdg_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr,
                  double beta1, double beta2, float eps, int32_t step, void* stream);
/* Same, for a segment made of rows of row_len floats whose first head_len floats use lr_head and the rest lr_tail
 * (the SH features [P,16,3] kept as ONE tensor: DC at feature_lr, the rest at feature_lr/20, as the parameter
 * groups f_dc / f_rest of /root/reference/src/trainer/rodygs_static.py:106-141).                               */
int rdg_adam_step_rows(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                       int32_t row_len, int32_t head_len, float lr_head, float lr_tail, double beta1, double beta2,
                       float eps, int32_t step, void* stream);

/* All parameter groups in ONE launch (the reference steps ~8 groups per sub-step, rodygs_static.py:106-141).     */
typedef struct RdgAdamSeg {
    int64_t n;              /* floats in this segment                                   */
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    float lr_head, lr_tail; /* lr of the first head_len floats of every row / of the rest */
    int32_t row_len, head_len;   /* row_len <= 1: uniform lr_head                          */
    const float* grad2;     /* optional second gradient buffer: the step uses grad + grad2 (ABI 7).  The reference's iteration
                             * leaves a sub-step's gradient in the OTHER trainer's parameters until that trainer steps
                             * (/root/reference/src/trainer/rodygs.py:157-179, 364-369): with two gradient buffers written in
                             * turn -- every backward kernel OVERWRITES its buffer -- the stale part and the fresh part meet
                             * here, and nothing is ever accumulated or cleared.  NULL: grad alone.                          */
} RdgAdamSeg;
int rdg_adam_step_multi(int32_t nseg, const RdgAdamSeg* segs_host, double beta1, double beta2, float eps, int32_t step,
                        void* stream);

/* ---- fused photometric loss -------------------------------------------------------------
 * loss = (1-lambda) mean|img-gt| + lambda (1 - mean SSIM(img, gt)), SSIM as in
 * /root/reference/src/utils/loss_utils.py:19-100 (11x11 Gaussian window, sigma 1.5, zero padding), the combination
 * the reference builds from L1Loss + SSIMLoss (/root/reference/src/trainer/losses.py:78-107, weights in
 * configs/train/train_kubric_mrig.yaml:134-146).  img, gt: [C,H,W] f32.  ws: rdg_loss_ws_bytes() scratch that
 * must survive until backward.  loss3 receives {loss, l1_mean, ssim_mean}.  grad_loss: device scalar (NULL = 1). */
size_t rdg_loss_ws_bytes(int32_t C, int32_t H, int32_t W);
int rdg_photometric_loss_forward(int32_t C, int32_t H, int32_t W, const float* img, const float* gt, float lambda,
                                 void* ws, float* loss3, void* stream);
int rdg_photometric_loss_backward(int32_t C, int32_t H, int32_t W, const float* img, const float* gt, float lambda,
                                  const void* ws, const float* grad_loss, float* d_img, void* stream);

/* ---- parameter activations and pose -> viewmatrix ---------------------------------
 * rdg_activate_*: StaticRoDyGS.get_xyz/get_scaling/get_rotation/get_opacity/get_features
 * (/root/reference/src/model/rodygs_static.py:82-105) + the deformation add and concat of get_GS_properties
 * (/root/reference/src/trainer/rodygs.py:68-113):
 *   means3D = xyz + dxyz, scales = exp(scaling), rots = normalize(rotation) + drot, opac = sigmoid(opacity),
 *   shs[P,K,3] = cat(f_dc[P,1,3], f_rest[P,K-1,3]).   dxyz / drot may be NULL.
 * Backward OVERWRITES d_* (so they can point straight into a flat gradient bucket); g_* may be NULL (= zero).
 * d(dxyz) = g_means3D and d(drot) = g_rots are identities and are not produced.                               */
int rdg_activate_forward(int32_t P, int32_t K, const float* xyz, const float* dxyz, const float* scaling,
                         const float* rotation, const float* drot, const float* opacity, const float* f_dc,
                         const float* f_rest, float* out_means3D, float* out_scales, float* out_rots, float* out_opac,
                         float* out_shs, void* stream);
int rdg_activate_backward(int32_t P, int32_t K, const float* scaling, const float* rotation, const float* opacity,
                          const float* g_means3D, const float* g_scales, const float* g_rots, const float* g_opac,
                          const float* g_shs, float* d_xyz, float* d_scaling, float* d_rotation, float* d_opacity,
                          float* d_fdc, float* d_frest, void* stream);
/* FixedCameraTorch.world_view_transform (/root/reference/src/data/utils.py:161-170) for row `frame` of the
 * learnable cam_q[T,4] (r,i,j,k; normalised by 2/|q|^2 as graphic_utils.py:76-102) and cam_t[T,3] tables;
 * out_view16 = W2C^T (glm storage).  Backward writes full [T,4] / [T,3] gradients (zero outside `frame`).     */
int rdg_pose_view_forward(int32_t T, int32_t frame, const float* cam_q, const float* cam_t, float* out_view16,
                          void* stream);
int rdg_pose_view_backward(int32_t T, int32_t frame, const float* cam_q, const float* cam_t, const float* g_view16,
                           float* d_q, float* d_t, void* stream);
/* The same for the nviews (<= RDG_MAX_VIEWS) frames of one step: out_views [nviews,16]; backward zeroes d_q / d_t and
 * adds the rows of the frames in order (a frame may appear twice).  frames_host is a HOST array.                  */
int rdg_pose_views_forward(int32_t T, int32_t nviews, const int32_t* frames_host, const float* cam_q,
                           const float* cam_t, float* out_views, void* stream);
int rdg_pose_views_backward(int32_t T, int32_t nviews, const int32_t* frames_host, const float* cam_q,
                            const float* cam_t, const float* g_views, float* d_q, float* d_t, void* stream);

/* ---- per-step scalars in device memory (hipGraph replay of a train step, rodygs_amd/trainstep.py GraphedStep) --------
 * A captured graph bakes every by-value kernel argument.  The values of a train step that change from step to step
 * -- Adam's two bias corrections, the index of the rendered frame and (optionally) the learning rates -- can instead be
 * read from this 128-byte device struct, which the host refreshes (one small H2D copy) before each replay.  The *_dev entry points below are the
 * by-pointer forms of rdg_pose_view_forward / _backward, rdg_adam_step_multi and rdg_preprocess_backward_adam; same
 * arithmetic, same bits (the host computes the corrections exactly as the by-value forms do).                       */
typedef struct RdgStepScalars {
    float inv_bias_correction1;    /* (float)(1 / (1 - beta1^step))    */
    float sqrt_bias_correction2;   /* (float)sqrt(1 - beta2^step)      */
    int32_t frame;                 /* row of the camera tables to render */
    int32_t lr_from_table;         /* != 0: the learning rates below replace the by-value ones of the *_dev calls -- the
                                    * reference re-sets the xyz (and deform) learning rate every iteration
                                    * (/root/reference/src/trainer/rodygs_static.py:143-149); a captured graph would replay
                                    * the rates it was captured with                                                   */
    float seg_lr_head[RDG_ADAM_MAX_SEGS]; /* rdg_adam_step_multi_dev: lr_head / lr_tail of segment i of that launch     */
    float seg_lr_tail[RDG_ADAM_MAX_SEGS];
    float sh_lr_head, sh_lr_tail;  /* rdg_preprocess_backward_adam_dev                                                  */
    int32_t reserved[2];
} RdgStepScalars;                  /* 128 bytes */
int rdg_pose_view_forward_dev(int32_t T, const RdgStepScalars* dev, const float* cam_q, const float* cam_t,
                              float* out_view16, void* stream);
int rdg_pose_view_backward_dev(int32_t T, const RdgStepScalars* dev, const float* cam_q, const float* cam_t,
                               const float* g_view16, float* d_q, float* d_t, void* stream);
int rdg_adam_step_multi_dev(int32_t nseg, const RdgAdamSeg* segs_host, double beta1, double beta2, float eps,
                            const RdgStepScalars* dev, void* stream);
int rdg_preprocess_backward_adam_dev(const RdgRasterSettings* s_host, const float* means3D, float* shs,
                                     const float* opacities, const float* scales, const float* rotations,
                                     const float* viewmatrix, const float* projmatrix, const int32_t* radii,
                                     const void* geom_ws, void* grad_ws, float* dL_dmeans3D, float* dL_dmeans2D,
                                     float* dL_dopacities, float* dL_dscales, float* dL_drotations,
                                     float* dL_dviewmatrix, float* sh_exp_avg, float* sh_exp_avg_sq, int32_t head_len,
                                     float lr_head, float lr_tail, double beta1, double beta2, float eps,
                                     const RdgStepScalars* dev, void* stream);

/* ---- measurement hooks -----------------------------------------------------------------------------------
 * When enabled, every stage is bracketed by hipEvents recorded on the launch stream.  rdg_stage_time_ms()
 * synchronises on the recorded events and returns accumulated milliseconds + launch count since the last
 * reset.                                                                                                    */
int rdg_timing_enable(int32_t on);
int rdg_timing_select(uint32_t stage_mask);   /* bit i = time stage i (default all); events cost ~5 us of stream gap each */
int rdg_timing_reset(void);
int rdg_stage_time_ms(int32_t stage, double* total_ms, int64_t* count);

#ifdef __cplusplus
}
#endif
#endif /* RODYGS_HIP_H */
