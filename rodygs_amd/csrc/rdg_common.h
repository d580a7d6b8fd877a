// rdg_common.h -- shared device/host definitions for librodygs_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/rodygs_hip.h"

#define RDG_TILE 16
#define RDG_TILE_PIX 256
#define RDG_WAVE 64

// algorithm constants (SURVEY.md §7 open question 6: kept in one place)
#define RDG_NEAR_CULL 0.2f
#define RDG_FOV_CLAMP 1.3f
#define RDG_DILATION 0.3f
#define RDG_ALPHA_CAP 0.99f
#define RDG_ALPHA_MIN (1.0f / 255.0f)
#define RDG_T_STOP 0.0001f
#define RDG_LAMBDA_FLOOR 0.1f

// 64-byte per-Gaussian splat record: everything the compositing kernels gather, one L2 line-half per splat.
//   q0 = (px, py, conic_a, conic_b)   q1 = (conic_c, opacity, depth, radius as int bits)
//   q2 = (r, g, b, 1 / cov2D_yy)      q3 = (nx, ny, nz, _)
// (1 / cov2D_yy = conic_c - conic_b^2 / conic_a, the second coefficient of the completed square the compositing kernels
//  evaluate, carried because that difference cancels on needle-shaped footprints: rdg_stage_conic)
struct __attribute__((aligned(64))) RdgRec {
    float4 q0, q1, q2, q3;
};

// 64-byte per-Gaussian gradient accumulator row (one 64-B atomic request per (tile, splat)):
//   moments of t = G dL/dG about (w, dy), w = dx + beta dy, beta = conic_b / conic_a (the skew coordinate the exponent is
//   evaluated in): [0] [1] sum(t w), sum(t dy) (dL/dmean2D = -conic . (dx, dy) moments, formed by the per-Gaussian backward)
//   [2..4] -1/2 sum(t w^2), -sum(t w dy), -1/2 sum(t dy^2) (-> dL/dconic a, b, c there) -- these five DIVIDED BY THE OPACITY  [5] dL/dopacity  [6..8] dL/drgb  [9] dL/ddepth
#define RDG_GROW 16

static inline size_t rdg_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

#define RDG_PRE_BLOCK 256  // Gaussians per preprocess / duplicate block

// ---- geom workspace layout -------------------------------------------------------------------------------
struct RdgGeomLayout {
    size_t rec;            // RdgRec[P]
    size_t tiles_touched;  // uint32[P]
    size_t clamped;        // uint8[P]  (bit c set = channel c clamped at 0)
    size_t block_sums;     // uint32[nblk+1]  (exclusive-scanned in place; [nblk] = total)
    size_t rectd;          // uint4[P]: (x0 | y0 << 16, x1 | y1 << 16, view-space depth bits, tiles touched) -- the tile
                           // rectangle [x0, x1) x [y0, y1) the binning stage expands (rdg_splat_rect: the ONE place it is
                           // formed), next to the two other words that stage needs: one coalesced 16-B load per Gaussian
    size_t total;
};
static inline RdgGeomLayout rdg_geom_layout(int32_t P) {
    RdgGeomLayout L;
    size_t Pp = (size_t)(P > 0 ? P : 1);
    size_t nblk = (Pp + RDG_PRE_BLOCK - 1) / RDG_PRE_BLOCK;
    size_t o = 0;
    L.rec = o;            o = rdg_align_up(o + Pp * sizeof(RdgRec), 256);
    L.tiles_touched = o;  o = rdg_align_up(o + Pp * 4, 256);
    L.clamped = o;        o = rdg_align_up(o + Pp, 256);
    L.block_sums = o;     o = rdg_align_up(o + (nblk + 1) * 4, 256);
    L.rectd = o;          o = rdg_align_up(o + Pp * 16, 256);
    L.total = o;
    return L;
}

// ---- sort geometry ---------------------------------------------------------------------------------------
#define RDG_SORT_BITS 8
// workspace of the radix sort (rdg_radix_sort.hip): the (digit, segment) count table and the digit totals
size_t rdg_radix_sort_tmp_bytes(int64_t capacity);
struct RdgSortLayout { size_t total; };
static inline RdgSortLayout rdg_sort_layout(int64_t capacity) {
    RdgSortLayout L;
    L.total = rdg_radix_sort_tmp_bytes(capacity);
    return L;
}

// ---- heavy tiles (bucket binning): tiles with more than RDG_TSORT_LDS instances are sorted by several workgroups
// (chunk sort in LDS + a merge tree, rdg_binning.hip).  Everything is sized by the capacity alone: a heavy tile has
// > RDG_TSORT_LDS instances, so there are < cap / RDG_TSORT_LDS of them and < 2 cap / RDG_TSORT_LDS chunks in all.
#define RDG_TSORT_SMALL 1024
#define RDG_TSORT_MID 4096     // lists up to here: 1024-instance chunks sorted by a wave each, merged in LDS by one workgroup
#define RDG_TSORT_LDS 8192
struct RdgHeavyDesc { uint32_t start, n, nchunks, node_base, tile, pad0, pad1, pad2; };
struct RdgHeavyLayout {
    size_t header;   // uint32[64]: [0] = number of work items, [1] = number of heavy tiles
    size_t desc;     // RdgHeavyDesc[max_heavy]
    size_t work;     // uint2[max_work]: items of rdg_tile_sort_large_kernel: (heavy tile index, chunk index);
                     //                  (0x80000000 | tile, 0) = a tile of RDG_TSORT_SMALL + 1 .. RDG_TSORT_MID instances whose
                     //                  1024-instance chunks are sorted: merge them in LDS; (0x80000000 | tile, 0xffffffff) =
                     //                  a tile of RDG_TSORT_MID + 1 .. RDG_TSORT_LDS instances (one workgroup sorts it in LDS)
    size_t chunks;   // uint2[max_chunk_items]: (tile, c) = chunk c of 1024 instances of a mid tile, sorted in place by a
                     //                  wave of rdg_tile_sort_lanes_kernel (header[2] = their number)
    size_t nodes;    // uint32[2 * max_chunks]  arrival counters of the merge-tree nodes
    size_t total;
    uint32_t max_heavy, max_chunks, max_work, max_chunk_items;
};
static inline RdgHeavyLayout rdg_heavy_layout(int64_t capacity) {
    RdgHeavyLayout L;
    const size_t cap = (size_t)(capacity > 0 ? capacity : 1);
    L.max_heavy = (uint32_t)(cap / RDG_TSORT_LDS + 1);
    L.max_chunks = 2 * L.max_heavy;
    L.max_work = L.max_chunks + (uint32_t)(cap / RDG_TSORT_SMALL + 1);
    // chunk items of the mid tiles: every chunk but a tile's last is full, and a mid tile has > RDG_TSORT_SMALL instances
    L.max_chunk_items = 2u * (uint32_t)(cap / RDG_TSORT_SMALL + 1);
    size_t o = 0;
    L.header = o;  o = rdg_align_up(o + 256, 256);
    L.desc = o;    o = rdg_align_up(o + (size_t)L.max_heavy * sizeof(RdgHeavyDesc), 256);
    L.work = o;    o = rdg_align_up(o + (size_t)L.max_work * 8, 256);
    L.chunks = o;  o = rdg_align_up(o + (size_t)L.max_chunk_items * 8, 256);
    L.nodes = o;   o = rdg_align_up(o + (size_t)L.max_chunks * 2 * 4, 256);
    L.total = o;
    return L;
}

// ---- split path of the compositing stage (rdg_render.hip): tiles with more than RDG_SPLIT_MIN instances are cut into
// segments of RDG_SPLIT_SEG list positions, one workgroup each.  Sized by the capacity alone: a split tile has more
// than RDG_SPLIT_MIN instances, so there are < cap / RDG_SPLIT_MIN of them and < cap / RDG_SPLIT_SEG + that many segments.
#define RDG_SPLIT_MIN 4096
#define RDG_SPLIT_SEG 2048     // multiple of the staging batch (256) and of the visit-record word (64)
#define RDG_SEG_F 12           // floats per (segment, pixel): see rdg_render.hip
struct RdgSplitLayout {
    size_t header;    // uint32[64]: [0] = segments (work items), [1] = split tiles
    size_t work;      // uint4[max_seg]: (tile, segment, segments of the tile, first segment slot)
    size_t tiles;     // uint4[max_tiles]: (tile, segments, first segment slot, -)
    size_t seg_pix;   // float[max_seg][RDG_SEG_F][256]
    size_t total;
    uint32_t max_seg, max_tiles;
};
static inline RdgSplitLayout rdg_split_layout(int64_t capacity) {
    RdgSplitLayout L;
    const size_t cap = (size_t)(capacity > 0 ? capacity : 1);
    L.max_tiles = (uint32_t)(cap / RDG_SPLIT_MIN + 1);
    L.max_seg = (uint32_t)(cap / RDG_SPLIT_SEG + 1) + L.max_tiles;
    size_t o = 0;
    L.header = o;   o = rdg_align_up(o + 256, 256);
    L.work = o;     o = rdg_align_up(o + (size_t)L.max_seg * 16, 256);
    L.tiles = o;    o = rdg_align_up(o + (size_t)L.max_tiles * 16, 256);
    L.seg_pix = o;  o = rdg_align_up(o + (size_t)L.max_seg * RDG_SEG_F * RDG_TILE_PIX * 4, 256);
    L.total = o;
    return L;
}

// ---- binning workspace layout ----------------------------------------------------------------------------
struct RdgBinLayout {
    size_t keys_a, keys_b;  // uint64[cap]
    size_t vals_a, vals_b;  // uint32[cap]
    size_t sort_tmp;        // RdgSortLayout
    size_t heavy;           // RdgHeavyLayout: work list of the multi-workgroup sort of tiles > RDG_TSORT_LDS instances
    size_t hit;             // uint64[cap/64 + n_tiles + 2][4]: per 64 list slots of a tile, per quadrant, "the forward
                            // had a pixel that could blend this splat" (lets the backward skip the other visits)
    size_t split;           // RdgSplitLayout: work lists and per-segment pixel records of the split compositing path
    size_t total;
};
static inline RdgBinLayout rdg_bin_layout(int64_t capacity, int32_t n_tiles = 0) {
    RdgBinLayout L;
    size_t cap = (size_t)(capacity > 0 ? capacity : 1);
    size_t o = 0;
    L.keys_a = o;  o = rdg_align_up(o + cap * 8, 256);
    L.keys_b = o;  o = rdg_align_up(o + cap * 8, 256);
    L.vals_a = o;  o = rdg_align_up(o + cap * 4, 256);
    L.vals_b = o;  o = rdg_align_up(o + cap * 4, 256);
    L.sort_tmp = o; o = rdg_align_up(o + rdg_sort_layout(capacity).total, 256);
    L.heavy = o;   o = rdg_align_up(o + rdg_heavy_layout(capacity).total, 256);
    L.hit = o;     o = rdg_align_up(o + (cap / 64 + (size_t)(n_tiles > 0 ? n_tiles : 262144) + 2) * 32, 256);
    // LAST: its offset depends on n_tiles (callers that only need the earlier offsets pass 0), its size does not
    L.split = o;   o = rdg_align_up(o + rdg_split_layout(capacity).total, 256);
    L.total = o;
    return L;
}
static inline size_t rdg_hit_bytes(int64_t capacity, int32_t n_tiles) {
    return ((size_t)(capacity > 0 ? capacity : 1) / 64 + (size_t)n_tiles + 2) * 32;
}

// ---- image workspace layout ------------------------------------------------------------------------------
struct RdgImageLayout {
    size_t final_T;    // float[H*W]
    size_t n_contrib;  // uint32[H*W]
    size_t ranges;     // uint2[n_tiles]
    size_t tile_cnt;   // uint32[n_tiles]  instances per tile (bucket binning)
    size_t tile_fill;  // uint32[n_tiles]  scatter cursors
    size_t total;
};
// The per-tile instance counters are indexed along a Z curve over the tile grid: the 2x2 .. 3x3 tiles one Gaussian
// touches then share one or two 64-B lines, and the counting atomics of a wave merge better than with row-major rows
// 480 B apart.  The curve covers the power-of-two square around the grid.
static inline size_t rdg_cnt_entries(int32_t gx, int32_t gy) {
    size_t side = 1;
    while ((int64_t)side < gx || (int64_t)side < gy) side <<= 1;
    return side * side;
}
static inline RdgImageLayout rdg_image_layout(int32_t H, int32_t W) {
    RdgImageLayout L;
    size_t hw = (size_t)H * W;
    size_t nt = (size_t)((W + RDG_TILE - 1) / RDG_TILE) * ((H + RDG_TILE - 1) / RDG_TILE);
    const size_t nz = rdg_cnt_entries((W + RDG_TILE - 1) / RDG_TILE, (H + RDG_TILE - 1) / RDG_TILE);
    size_t o = 0;
    L.final_T = o;    o = rdg_align_up(o + hw * 4, 256);
    L.n_contrib = o;  o = rdg_align_up(o + hw * 4, 256);
    L.ranges = o;     o = rdg_align_up(o + nt * 8, 256);
    L.tile_cnt = o;   o = rdg_align_up(o + nz * 4, 256);
    L.tile_fill = o;  o = rdg_align_up(o + nt * 4, 256);
    L.total = o;
    return L;
}

static inline int rdg_key_bits(int32_t n_tiles) {
    int b = 0;
    while ((1ll << b) < (long long)n_tiles) ++b;
    return 32 + b;
}

// device-side view of the camera / settings handed to kernels by value

struct RdgDev {
    int32_t P, M, deg, H, W, gx, gy;
    float tanx, tany, fx, fy, smod;
    int32_t prefiltered, cov_grad, sh_grad, render_normal;
    int32_t bin_mode, nren_stats;
    int32_t cull;              // RdgRasterSettings.cull: 0 = the reference's tile rectangles, 1 = tight rectangles
    int32_t list_hints;        // RdgRasterSettings.list_hints: bit 0 = split compositing path for lists > RDG_SPLIT_MIN
    int32_t tile_cnt_zeroed;   // internal: the per-tile counters were cleared by the per-Gaussian stage's scan kernel
    int32_t grad_rows_zeroed;  // RdgRasterSettings.grad_rows_zeroed (backward)
    void* zero_grad_ws;        // RdgRasterSettings.zero_grad_ws (forward): gradient rows cleared by the compositing kernel
    int32_t* nren_host;        // RdgRasterSettings.num_rendered_host (forward): host mirror of num_rendered[0..1]
    int32_t* nren_max;         // RdgRasterSettings.num_rendered_max (forward): sticky maximum of D over forwards
    // RdgRasterSettings.densify_* (backward): densification statistics updated by the per-Gaussian backward kernel
    float* dn_accum; float* dn_denom; float* dn_maxr;
    int32_t dn_row0, dn_rows;
};

// ---- error + timing plumbing (rdg_api.hip) ---------------------------------------------------------------
int rdg_set_error(const char* fmt, ...);
int rdg_check_hip(hipError_t e, const char* what);
// zero `bytes` at `p` on stream `st` with a kernel (not a memset node: see rdg_api.hip)
hipError_t rdg_zero_async(void* p, size_t bytes, hipStream_t st);
void rdg_stage_begin(int stage, hipStream_t s);
void rdg_stage_end(int stage, hipStream_t s);
// begin / end as a scope: every return path between them, the error returns included, closes the stage (its timing
// bracket and its roctx range)
struct RdgStageScope {
    int stage; hipStream_t s;
    RdgStageScope(int stage_, hipStream_t s_) : stage(stage_), s(s_) { rdg_stage_begin(stage_, s_); }
    ~RdgStageScope() { rdg_stage_end(stage, s); }
    RdgStageScope(const RdgStageScope&) = delete;
    RdgStageScope& operator=(const RdgStageScope&) = delete;
};

// Adam state of the SH-feature segment handed to the per-Gaussian backward kernel ("optimizer in backward",
// rdg_preprocess_backward_adam): the kernel has a wave's 64 gradient rows in LDS -- instead of writing them out for
// the optimizer to read back, it applies the update right there.  m == nullptr: off.
struct RdgShAdam {
    float* m; float* v;
    float step_head, step_tail, b1, b2, omb1, omb2, eps, bc2_sqrt;   // omb = 1 - beta, rounded from double
    int head_len;
    // graph replay: the bias corrections come from device memory (rdg_sh_adam_resolve), lr_* are the raw rates
    const RdgStepScalars* dev;
    float lr_head, lr_tail;
};

#ifdef __HIPCC__
// by-pointer form of the per-step scalars: the same products the host forms for the by-value form
__device__ __forceinline__ RdgShAdam rdg_sh_adam_resolve(RdgShAdam ad) {
    if (ad.dev) {
        const float inv = ad.dev->inv_bias_correction1;
        const bool tab = ad.dev->lr_from_table != 0;
        ad.step_head = (tab ? ad.dev->sh_lr_head : ad.lr_head) * inv;
        ad.step_tail = (tab ? ad.dev->sh_lr_tail : ad.lr_tail) * inv;
        ad.bc2_sqrt = ad.dev->sqrt_bias_correction2;
    }
    return ad;
}
#endif

// ---- kernel launchers (one per .hip file) ----------------------------------------------------------------
int rdg_launch_preprocess_fwd(const RdgDev& d, const float* means3D, const float* shs, const float* colors,
                              const float* opac, const float* scales, const float* rots, const float* cov3D,
                              const float* view, const float* proj, void* geom_ws, int32_t* radii,
                              int32_t* num_rendered, hipStream_t s, uint32_t* zero_buf = nullptr, size_t zero_words = 0);
// rdg_densify.hip: the statistics update as its own launch (rdg_densify_stats)
int rdg_launch_densify_stats(long long n, long long row0, const float* dmeans2D, const int32_t* radii, float* accum,
                             float* denom, float* maxr, hipStream_t s);
int rdg_launch_geom_from_records(const RdgDev& d, void* geom_ws, int32_t* radii, int32_t* num_rendered,
                                 hipStream_t s);
int rdg_launch_bin(const RdgDev& d, const void* geom_ws, const int32_t* radii, void* bin_ws, int64_t capacity,
                   void* image_ws, int32_t* num_rendered, uint64_t* keys_unsorted_copy,
                   uint32_t* vals_unsorted_copy, hipStream_t s, bool export_sorted_keys = false);
int rdg_launch_sort(uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, int64_t capacity,
                    const int32_t* n_dev, int end_bit, void* sort_tmp, int* result_in_b, hipStream_t s);
// stable LSD sort on key bits [begin_bit, end_bit) of uint32 / uint64 keys with uint32 values (rdg_radix_sort.hip)
template <typename KeyT>
int rdg_launch_radix_sort(KeyT* keys_a, KeyT* keys_b, uint32_t* vals_a, uint32_t* vals_b, int64_t capacity,
                        const int32_t* n_dev, int begin_bit, int end_bit, void* tmp, int* result_in_b, hipStream_t s);
int rdg_launch_render_fwd(const RdgDev& d, const float* bg, const void* geom_ws, void* bin_ws,
                          int64_t capacity, void* image_ws, const int32_t* num_rendered, float* out_color,
                          float* out_depth, float* out_normal, float* out_alpha, hipStream_t s);
// device-side work lists of the split compositing path, from the tile ranges (rdg_render.hip)
int rdg_launch_split_build(const RdgDev& d, void* bin_ws, int64_t capacity, const void* image_ws,
                           const int32_t* num_rendered, hipStream_t s);
int rdg_launch_render_bwd(const RdgDev& d, const float* bg, const void* geom_ws, const void* bin_ws,
                          int64_t capacity, const void* image_ws, const float* g_color, const float* g_depth,
                          const float* g_alpha, float* grow, hipStream_t s, float* det = nullptr,
                          const float* g_normal = nullptr, uint32_t* det_off = nullptr, long long n_instances = 0);
int rdg_launch_preprocess_bwd(const RdgDev& d, const float* means3D, const float* shs, const float* colors,
                              const float* opac, const float* scales, const float* rots, const float* cov3D,
                              const float* view, const float* proj, const int32_t* radii, const void* geom_ws,
                              const float* grow, float* posebuf, float* dmeans3D, float* dmeans2D, float* dshs,
                              float* dcolors, float* dopac, float* dscales, float* drots, float* dcov3D,
                              float* dview, hipStream_t s, const RdgShAdam* sh_adam = nullptr);

int rdg_launch_preprocess_fwd_views(const RdgDev& d, int32_t nviews, int32_t stride, int32_t row0,
                                    const float* means3D, const float* shs, const float* opac, const float* scales,
                                    const float* rots, const float* views, const float* proj, void* geom_ws,
                                    int32_t* radii, hipStream_t s);
int rdg_launch_preprocess_bwd_views(const RdgDev& d, int32_t nviews, int32_t stride, const float* means3D,
                                    const float* shs, const float* opac, const float* scales, const float* rots,
                                    const float* views, const float* proj, const int32_t* radii, const void* geom_ws,
                                    const float* grow, float* posebuf, float* dmeans3D, float* dmeans2D, float* dshs,
                                    float* dopac, float* dscales, float* drots, hipStream_t s);
int rdg_launch_pose_reduce_views(int nviews, int view_rows, int nblk, const float* views, float* posebuf, float* part,
                                 float* dviews, hipStream_t s);

// ---- small device helpers --------------------------------------------------------------------------------
#ifdef __HIPCC__
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND = true>
__device__ __forceinline__ float rdg_dpp(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK,
                                                                 BANK_MASK, BOUND));
}
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND = true>
__device__ __forceinline__ uint32_t rdg_dpp_u(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, BANK_MASK, BOUND);
}
// Sum over the 64 lanes of a wave; the total is valid in lane 63 (DPP only, no LDS traffic).
__device__ __forceinline__ float rdg_wave_sum_to63(float x) {
    x += rdg_dpp<0xB1>(x);         // quad_perm [1,0,3,2]
    x += rdg_dpp<0x4E>(x);         // quad_perm [2,3,0,1]
    x += rdg_dpp<0x141>(x);        // row_half_mirror
    x += rdg_dpp<0x140>(x);        // row_mirror   -> every lane holds its 16-lane row total
    x += rdg_dpp<0x142, 0xa>(x);   // row_bcast:15 into rows 1,3
    x += rdg_dpp<0x143, 0xc>(x);   // row_bcast:31 into rows 2,3 -> lane 63 = total
    return x;
}
__device__ __forceinline__ float rdg_wave_sum_all(float x) {
    x = rdg_wave_sum_to63(x);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
// LDS traffic of ONE wave is executed in program order; this only pins the compiler's ordering around a hand-off
// between lanes of the same wave through LDS (no workgroup barrier involved).
__device__ __forceinline__ void rdg_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Wave-cooperative transposing copies between 64 consecutive rows of a row-major global array g[n_rows][row] and
// an LDS tile S[64][stride] (stride odd -> a lane walking its own row is bank-conflict free).  One lane per
// Gaussian reading its own 48-float SH row straight from global memory touches 64 cache lines per load
// instruction and uses 4-16 bytes of each; through here every global instruction moves 1 KB (or 256 B) contiguous.
template <bool NT = true>
__device__ __forceinline__ void rdg_rows_to_lds(const float* __restrict__ g, long long first_row, long long n_rows,
                                                int row, int stride, float* S, int lane) {
    const long long base = first_row * row, total = n_rows * row;
    const float inv_row = 1.0f / (float)row;
    if ((row & 3) == 0 && (((uintptr_t)g) & 15) == 0) {
        typedef float rdg_nt4 __attribute__((ext_vector_type(4)));
        // four 16-B loads in flight per lane before the first LDS store (the callers run at 2 waves per SIMD)
        for (int v0 = lane; v0 < 16 * row; v0 += 256) {
            rdg_nt4 val[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int v = v0 + 64 * u;
                const long long e = base + 4ll * v;
                // NT = false: the rows will be read again soon (another camera of the same step): keep them cached
                if (v < 16 * row && e < total)
                    val[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const rdg_nt4*>(g + e))
                                : *reinterpret_cast<const rdg_nt4*>(g + e);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int v = v0 + 64 * u;
                const long long e = base + 4ll * v;
                if (v < 16 * row && e < total) {
                    const int gi = (int)(((float)(4 * v) + 0.5f) * inv_row);
                    float* dst = S + gi * stride + (4 * v - gi * row);
                    dst[0] = val[u].x; dst[1] = val[u].y; dst[2] = val[u].z; dst[3] = val[u].w;
                }
            }
        }
    } else {
        for (int idx = lane; idx < 64 * row; idx += 64) {
            const long long e = base + idx;
            if (e < total) {
                const int gi = (int)(((float)idx + 0.5f) * inv_row);
                S[gi * stride + (idx - gi * row)] = g[e];
            }
        }
    }
}
__device__ __forceinline__ void rdg_lds_to_rows(float* __restrict__ g, long long first_row, long long n_rows, int row,
                                                int stride, const float* S, int lane) {
    const long long base = first_row * row, total = n_rows * row;
    const float inv_row = 1.0f / (float)row;
    if ((row & 3) == 0 && (((uintptr_t)g) & 15) == 0) {
        for (int v = lane; v < 16 * row; v += 64) {
            const long long e = base + 4ll * v;
            if (e < total) {
                const int gi = (int)(((float)(4 * v) + 0.5f) * inv_row);
                const float* src = S + gi * stride + (4 * v - gi * row);
                typedef float rdg_nt4 __attribute__((ext_vector_type(4)));
                const rdg_nt4 val = {src[0], src[1], src[2], src[3]};
                __builtin_nontemporal_store(val, reinterpret_cast<rdg_nt4*>(g + e));
            }
        }
    } else {
        for (int idx = lane; idx < 64 * row; idx += 64) {
            const long long e = base + idx;
            if (e < total) {
                const int gi = (int)(((float)idx + 0.5f) * inv_row);
                g[e] = S[gi * stride + (idx - gi * row)];
            }
        }
    }
}
// g[rows] += S (or = S when !ACC): the accumulate form of rdg_lds_to_rows with ordinary (cached) accesses, for outputs
// that several cameras of one step add into back to back.
template <bool ACC>
__device__ __forceinline__ void rdg_lds_acc_rows(float* __restrict__ g, long long first_row, long long n_rows, int row,
                                                 int stride, const float* S, int lane) {
    const long long base = first_row * row, total = n_rows * row;
    const float inv_row = 1.0f / (float)row;
    if ((row & 3) == 0 && (((uintptr_t)g) & 15) == 0) {
        for (int v = lane; v < 16 * row; v += 64) {
            const long long e = base + 4ll * v;
            if (e < total) {
                const int gi = (int)(((float)(4 * v) + 0.5f) * inv_row);
                const float* src = S + gi * stride + (4 * v - gi * row);
                float4* dst = reinterpret_cast<float4*>(g + e);
                float4 val = make_float4(src[0], src[1], src[2], src[3]);
                if (ACC) { const float4 o = *dst; val.x += o.x; val.y += o.y; val.z += o.z; val.w += o.w; }
                *dst = val;
            }
        }
    } else {
        for (int idx = lane; idx < 64 * row; idx += 64) {
            const long long e = base + idx;
            if (e < total) {
                const int gi = (int)(((float)idx + 0.5f) * inv_row);
                const float val = S[gi * stride + (idx - gi * row)];
                g[e] = ACC ? g[e] + val : val;
            }
        }
    }
}
// One element of the update, with the fused multiply-adds written out: the float4 body and the scalar tail of a segment
// must round identically, or a parameter's value would depend on where its segment happens to end (sharded vs
// replicated layouts of the same cloud differed by one ulp on the tail elements).
// omb1 / omb2 = 1 - beta evaluated in double on the host and rounded once (torch.optim.Adam takes Python doubles:
// 1 - float(0.999) would be off by 4.7e-5 relative, a systematic difference in exp_avg_sq).
__device__ __forceinline__ void rdg_adam_elem(float& p, float g, float& m, float& v, float st, float b1, float b2,
                                              float omb1, float omb2, float eps, float bc2_sqrt) {
    m = __fmaf_rn(b1, m, omb1 * g);
    v = __fmaf_rn(b2, v, (omb2 * g) * g);
    p = __fmaf_rn(-st, m / (sqrtf(v) / bc2_sqrt + eps), p);
}


// ---- factored SH gradient rows -------------------------------------------------------------------------------------
// dL/dsh[k][ch] of a Gaussian is basis_k(direction) * dL/dcolour[ch] (clamped channels: 0): 16 + 3 numbers instead of
// 48.  The per-Gaussian backward leaves them in a small LDS tile (F: 64 rows of RDG_FAC floats -- basis values at 0..15,
// the three colour gradients at 16..18) next to the tile S that still holds the SH VALUES it staged at its start, so the
// optimizer-in-backward step below needs no second trip to memory for the parameters (192 B per Gaussian at 16
// coefficients, 13 % of that kernel's bytes).  g = F[k] * F[16 + ch] is the same product the unfactored form stored.
#define RDG_FAC 20
__device__ __forceinline__ float rdg_fac_grad(const float* Frow, int idx) {
    const int k = (idx * 171) >> 9;              // idx / 3 for idx < 171 (rows hold <= 48 floats)
    return Frow[k] * Frow[16 + (idx - 3 * k)];
}
// p[rows] <- Adam(p = S (the staged values), g = factored): the update form of rdg_lds_to_rows (same chunking; head_len
// leading floats of a row take step_head, the rest step_tail).  RDG_ADAM_UNROLL chunks per trip with every load issued
// before the first use: the kernel that calls this runs two waves per SIMD, so the bytes in flight per CU -- not the
// arithmetic -- set its rate
__device__ __forceinline__ void rdg_lds_adam_rows_fac(float* __restrict__ p, const RdgShAdam ad, long long first_row,
                                                      long long n_rows, int row, int stride, const float* S,
                                                      const float* F, int lane) {
    const long long base = first_row * row, total = n_rows * row;
    const float inv_row = 1.0f / (float)row;
    typedef float rdg_nt4 __attribute__((ext_vector_type(4)));
    if ((row & 3) == 0 && ((((uintptr_t)p) | ((uintptr_t)ad.m) | ((uintptr_t)ad.v)) & 15) == 0) {
#define RDG_ADAM_UNROLL 4
        for (int v0 = lane; v0 < 16 * row; v0 += 64 * RDG_ADAM_UNROLL) {
            rdg_nt4 mm[RDG_ADAM_UNROLL], vv[RDG_ADAM_UNROLL];
#pragma unroll
            for (int u = 0; u < RDG_ADAM_UNROLL; ++u) {
                const int v = v0 + 64 * u;
                const long long e = base + 4ll * v;
                if (v < 16 * row && e < total) {
                    mm[u] = __builtin_nontemporal_load(reinterpret_cast<const rdg_nt4*>(ad.m + e));
                    vv[u] = __builtin_nontemporal_load(reinterpret_cast<const rdg_nt4*>(ad.v + e));
                }
            }
#pragma unroll
            for (int u = 0; u < RDG_ADAM_UNROLL; ++u) {
                const int v = v0 + 64 * u;
                const long long e = base + 4ll * v;
                if (v < 16 * row && e < total) {
                    const int gi = (int)(((float)(4 * v) + 0.5f) * inv_row);
                    const int col = 4 * v - gi * row;
                    const float* src = S + gi * stride + col;
                    const float* Frow = F + gi * RDG_FAC;
                    float P4[4] = {src[0], src[1], src[2], src[3]}, M4[4] = {mm[u].x, mm[u].y, mm[u].z, mm[u].w};
                    float V4[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        rdg_adam_elem(P4[c], rdg_fac_grad(Frow, col + c), M4[c], V4[c],
                                      (col + c) < ad.head_len ? ad.step_head : ad.step_tail, ad.b1, ad.b2, ad.omb1, ad.omb2,
                                      ad.eps, ad.bc2_sqrt);
                    __builtin_nontemporal_store(rdg_nt4{P4[0], P4[1], P4[2], P4[3]}, reinterpret_cast<rdg_nt4*>(p + e));
                    __builtin_nontemporal_store(rdg_nt4{M4[0], M4[1], M4[2], M4[3]}, reinterpret_cast<rdg_nt4*>(ad.m + e));
                    __builtin_nontemporal_store(rdg_nt4{V4[0], V4[1], V4[2], V4[3]}, reinterpret_cast<rdg_nt4*>(ad.v + e));
                }
            }
        }
#undef RDG_ADAM_UNROLL
    } else {
        for (int idx = lane; idx < 64 * row; idx += 64) {
            const long long e = base + idx;
            if (e < total) {
                const int gi = (int)(((float)idx + 0.5f) * inv_row);
                const int col = idx - gi * row;
                float pi = S[gi * stride + col], mi = ad.m[e], vi = ad.v[e];
                rdg_adam_elem(pi, rdg_fac_grad(F + gi * RDG_FAC, col), mi, vi, col < ad.head_len ? ad.step_head : ad.step_tail,
                              ad.b1, ad.b2, ad.omb1, ad.omb2, ad.eps, ad.bc2_sqrt);
                p[e] = pi; ad.m[e] = mi; ad.v[e] = vi;
            }
        }
    }
}
// g[rows] = factored gradient: the store form (no optimizer in the kernel)
__device__ __forceinline__ void rdg_lds_to_rows_fac(float* __restrict__ g, long long first_row, long long n_rows, int row,
                                                    const float* F, int lane) {
    const long long base = first_row * row, total = n_rows * row;
    const float inv_row = 1.0f / (float)row;
    if ((row & 3) == 0 && (((uintptr_t)g) & 15) == 0) {
        for (int v = lane; v < 16 * row; v += 64) {
            const long long e = base + 4ll * v;
            if (e < total) {
                const int gi = (int)(((float)(4 * v) + 0.5f) * inv_row);
                const int col = 4 * v - gi * row;
                const float* Frow = F + gi * RDG_FAC;
                typedef float rdg_nt4 __attribute__((ext_vector_type(4)));
                const rdg_nt4 val = {rdg_fac_grad(Frow, col), rdg_fac_grad(Frow, col + 1), rdg_fac_grad(Frow, col + 2),
                                     rdg_fac_grad(Frow, col + 3)};
                __builtin_nontemporal_store(val, reinterpret_cast<rdg_nt4*>(g + e));
            }
        }
    } else {
        for (int idx = lane; idx < 64 * row; idx += 64) {
            const long long e = base + idx;
            if (e < total) {
                const int gi = (int)(((float)idx + 0.5f) * inv_row);
                g[e] = rdg_fac_grad(F + gi * RDG_FAC, idx - gi * row);
            }
        }
    }
}
__device__ __forceinline__ uint32_t rdg_lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// inclusive prefix sum over the 64 lanes (uint32)
__device__ __forceinline__ uint32_t rdg_wave_scan_incl(uint32_t x) {
    uint32_t t;
    t = rdg_dpp_u<0x111>(x); x += t;               // row_shr:1
    t = rdg_dpp_u<0x112>(x); x += t;               // row_shr:2
    t = rdg_dpp_u<0x114>(x); x += t;               // row_shr:4
    t = rdg_dpp_u<0x118>(x); x += t;               // row_shr:8  -> row-inclusive
    t = rdg_dpp_u<0x142, 0xa>(x); x += t;          // row_bcast:15 -> rows 1,3
    t = rdg_dpp_u<0x143, 0xc>(x); x += t;          // row_bcast:31 -> rows 2,3
    return x;
}
// inclusive running maximum over the 64 lanes (uint32; lanes shifted in from outside a row read 0)
__device__ __forceinline__ uint32_t rdg_wave_scan_max_incl(uint32_t x) {
    x = max(x, rdg_dpp_u<0x111>(x));
    x = max(x, rdg_dpp_u<0x112>(x));
    x = max(x, rdg_dpp_u<0x114>(x));
    x = max(x, rdg_dpp_u<0x118>(x));
    x = max(x, rdg_dpp_u<0x142, 0xa>(x));
    x = max(x, rdg_dpp_u<0x143, 0xc>(x));
    return x;
}
#endif
