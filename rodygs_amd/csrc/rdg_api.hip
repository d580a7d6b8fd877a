// rdg_api.hip -- the extern "C" surface of librodygs_hip.so (declared in include/rodygs_hip.h).
// No torch headers, no exceptions across the ABI, no device allocation: plain pointers, sizes and a stream.
#include "rdg_common.h"
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <mutex>

static thread_local char g_err[512] = "";

int rdg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}
int rdg_check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    return rdg_set_error("%s: %s", what, hipGetErrorString(e));
}

// ---- stage timing (hipEvents on the launch stream) -----------------------------------------------------------
#define RDG_MAX_PENDING 4096
struct RdgPending { hipEvent_t a, b; int stage; };
static int g_timing = 0;
static uint32_t g_timing_mask = 0xFFFFFFFFu;
static RdgPending g_pending[RDG_MAX_PENDING];
static int g_npending = 0;
// A stage is opened and closed by the thread that launches it (begin / end pair inside one entry point), so the open
// events are per thread; the list of finished brackets and the totals are shared by the threads of the process (the
// autograd engine runs backward on a thread of its own) and guarded by one mutex -- taken only while timing is on.
static thread_local hipEvent_t g_open[RDG_STAGE_COUNT];
static double g_total_ms[RDG_STAGE_COUNT];
static int64_t g_count[RDG_STAGE_COUNT];
static std::mutex g_timing_mu;

static void rdg_timing_drain() {   // caller holds g_timing_mu
    for (int i = 0; i < g_npending; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(g_pending[i].b) == hipSuccess &&
            hipEventElapsedTime(&ms, g_pending[i].a, g_pending[i].b) == hipSuccess) {
            g_total_ms[g_pending[i].stage] += ms;
            g_count[g_pending[i].stage] += 1;
        }
        (void)hipEventDestroy(g_pending[i].a);
        (void)hipEventDestroy(g_pending[i].b);
    }
    g_npending = 0;
}
// ---- named ranges for rocprofv3 --marker-trace: RDG_ROCTX=1 in the environment ---------------
// roctx is looked up at run time (librocprofiler-sdk-roctx.so, then libroctx64.so): the library has no link-time
// dependency on a profiler.  Ranges are host-side push / pop pairs around the launches of a stage.
typedef int (*rdg_roctx_push_t)(const char*);
typedef int (*rdg_roctx_pop_t)(void);
static rdg_roctx_push_t g_roctx_push = nullptr;
static rdg_roctx_pop_t g_roctx_pop = nullptr;
static int g_roctx = -1;      // -1: not looked at yet, 0: off, 1: on
static const char* const g_stage_names[RDG_STAGE_COUNT] = {
    "rdg:preprocess", "rdg:scan_dup", "rdg:sort", "rdg:ranges", "rdg:render_fwd", "rdg:render_bwd", "rdg:preprocess_bwd",
    "rdg:deform_fwd", "rdg:deform_bwd", "rdg:adam", "rdg:loss_fwd", "rdg:loss_bwd", "rdg:mlp_fwd", "rdg:mlp_bwd"};
static bool rdg_roctx_on() {
    if (g_roctx < 0) {
        g_roctx = 0;
        const char* e = getenv("RDG_ROCTX");
        if (e && e[0] == '1') {
            void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
            if (h) {
                g_roctx_push = (rdg_roctx_push_t)dlsym(h, "roctxRangePushA");
                g_roctx_pop = (rdg_roctx_pop_t)dlsym(h, "roctxRangePop");
                if (g_roctx_push && g_roctx_pop) g_roctx = 1;
            }
        }
    }
    return g_roctx == 1;
}

void rdg_stage_begin(int stage, hipStream_t s) {
    if (rdg_roctx_on() && stage >= 0 && stage < RDG_STAGE_COUNT) (void)g_roctx_push(g_stage_names[stage]);
    if (!g_timing || stage < 0 || stage >= RDG_STAGE_COUNT || !((g_timing_mask >> stage) & 1u)) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, s);
    g_open[stage] = e;
}
void rdg_stage_end(int stage, hipStream_t s) {
    if (g_roctx == 1 && stage >= 0 && stage < RDG_STAGE_COUNT) (void)g_roctx_pop();
    if (stage < 0 || stage >= RDG_STAGE_COUNT || !g_open[stage]) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) { (void)hipEventDestroy(g_open[stage]); g_open[stage] = nullptr; return; }
    (void)hipEventRecord(e, s);
    std::lock_guard<std::mutex> lk(g_timing_mu);
    if (g_npending >= RDG_MAX_PENDING) rdg_timing_drain();
    g_pending[g_npending].a = g_open[stage];
    g_pending[g_npending].b = e;
    g_pending[g_npending].stage = stage;
    g_npending++;
    g_open[stage] = nullptr;
}

void rdg_set_pose_aux(hipStream_t s);       // rdg_preprocess_bwd.hip: RdgRasterSettings.aux_stream of the call in flight
int rdg_pose_fork_event_ready();

static int rdg_make_dev(const RdgRasterSettings* s, RdgDev* d) {
    if (!s) return rdg_set_error("settings pointer is NULL");
    if (s->P < 0 || s->image_height <= 0 || s->image_width <= 0) return rdg_set_error("bad sizes");
    if (s->sh_degree < 0 || s->sh_degree > 3) return rdg_set_error("sh_degree must be 0..3");
    d->P = s->P; d->M = s->M; d->deg = s->sh_degree; d->H = s->image_height; d->W = s->image_width;
    d->gx = (d->W + RDG_TILE - 1) / RDG_TILE; d->gy = (d->H + RDG_TILE - 1) / RDG_TILE;
    d->tanx = s->tanfovx; d->tany = s->tanfovy;
    d->fx = (float)((double)d->W / (2.0 * (double)s->tanfovx));
    d->fy = (float)((double)d->H / (2.0 * (double)s->tanfovy));
    d->smod = s->scale_modifier;
    d->prefiltered = s->prefiltered; d->cov_grad = s->enable_cov_grad; d->sh_grad = s->enable_sh_grad;
    d->render_normal = s->render_normal;
    d->cull = s->cull ? 1 : 0;
    d->bin_mode = s->bin_mode; d->nren_stats = s->num_rendered_stats; d->list_hints = s->list_hints;
    d->grad_rows_zeroed = s->grad_rows_zeroed; d->zero_grad_ws = s->zero_grad_ws;
    d->nren_host = s->num_rendered_stats ? s->num_rendered_host : nullptr;
    d->nren_max = s->num_rendered_max;
    d->dn_accum = s->densify_grad_accum; d->dn_denom = s->densify_denom; d->dn_maxr = s->densify_max_radii;
    d->dn_row0 = s->densify_row0; d->dn_rows = s->densify_rows;
    if ((d->dn_accum || d->dn_denom || d->dn_maxr) && (d->dn_row0 < 0 || d->dn_rows < 0))
        return rdg_set_error("densify_row0 / densify_rows must not be negative");
    d->tile_cnt_zeroed = 0;
    return 0;
}

static int rdg_check_inputs(const RdgRasterSettings* s, const float* shs, const float* colors, const float* scales,
                            const float* rots, const float* cov3D) {
    if ((shs == nullptr) == (colors == nullptr))
        return rdg_set_error("Please provide excatly one of either SHs or precomputed colors!");
    if (((scales == nullptr || rots == nullptr) && cov3D == nullptr) ||
        ((scales != nullptr || rots != nullptr) && cov3D != nullptr))
        return rdg_set_error("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    if (shs && s->M < (s->sh_degree + 1) * (s->sh_degree + 1))
        return rdg_set_error("shs holds %d coefficients, sh_degree %d needs %d", s->M, s->sh_degree,
                             (s->sh_degree + 1) * (s->sh_degree + 1));
    return 0;
}

// Zero fill as a KERNEL, never hipMemsetAsync: a memset node inside a captured hipGraph was not ordered against the kernel
// nodes next to it on this ROCm (the backward's gradient rows were only sometimes zero when the graph replayed: different
// trajectories from run to run, NaN after another process had left other data in the memory; found with
// torch.utils.deterministic.fill_uninitialized_memory, scripts/dbg_fill.py).  Same cost: the runtime's own memset is a
// fill kernel too.
__global__ void __launch_bounds__(256) rdg_zero16_kernel(uint4* __restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        p[i] = make_uint4(0u, 0u, 0u, 0u);
}
__global__ void __launch_bounds__(256) rdg_zero4_kernel(uint32_t* __restrict__ p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
hipError_t rdg_zero_async(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    if (((bytes | (uintptr_t)p) & 3) != 0) return hipMemsetAsync(p, 0, bytes, st);   // no caller does this
    const bool wide = ((bytes | (uintptr_t)p) & 15) == 0;
    const size_t n = wide ? bytes / 16 : bytes / 4;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (wide) hipLaunchKernelGGL(rdg_zero16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint4*)p, n);
    else hipLaunchKernelGGL(rdg_zero4_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint32_t*)p, n);
    return hipGetLastError();
}

// zero the first min(*n_dev, cap) rows of row16 * 16 bytes each: the deterministic backward's per-instance rows, of which
// only the frame's own D are ever read (the workspace is sized by the capacity when the host does not know D)
__global__ void __launch_bounds__(256) rdg_zero_rows_dev_kernel(uint4* __restrict__ p, unsigned row16,
                                                               const uint32_t* __restrict__ n_dev, long long cap) {
    long long n = (long long)*n_dev;
    if (n > cap) n = cap;
    const size_t n16 = (size_t)n * row16;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        p[i] = make_uint4(0u, 0u, 0u, 0u);
}

extern "C" {

int rdg_abi_version(void) { return RDG_ABI_VERSION; }
int rdg_pose_fork_prepare(void) { return rdg_pose_fork_event_ready(); }
size_t rdg_settings_bytes(void) { return sizeof(RdgRasterSettings); }
const char* rdg_last_error(void) { return g_err; }

size_t rdg_geom_bytes(int32_t P) { return rdg_geom_layout(P).total; }
size_t rdg_binning_bytes(int64_t capacity, int32_t n_tiles) { return rdg_bin_layout(capacity, n_tiles).total; }
size_t rdg_image_bytes(int32_t H, int32_t W) { return rdg_image_layout(H, W).total; }
// deterministic mode: first-instance index of every Gaussian (uint32 [P]), the LAST piece of the gradient workspace
static size_t rdg_det_off_bytes(int32_t P) { return rdg_align_up((size_t)(P > 0 ? P : 1) * 4, 256); }
size_t rdg_grad_bytes(int32_t P) {
    const size_t Pp = (size_t)(P > 0 ? P : 1);
    // gradient rows + one pose partial row per per-Gaussian workgroup + 32 second-level pose partial rows
    // (one set of second-level rows per camera for the *_views entry points)
    return rdg_align_up(Pp * RDG_GROW * 4, 256) + rdg_align_up(((Pp + 127) / 128) * 19 * 4, 256) + 4096 +
           (size_t)RDG_MAX_VIEWS * 32 * 19 * 4 + rdg_det_off_bytes(P);
}
size_t rdg_sort_tmp_bytes(int64_t capacity) {
    // alternate key/value buffers + tables
    size_t cap = (size_t)(capacity > 0 ? capacity : 1);
    return rdg_align_up(cap * 8, 256) + rdg_align_up(cap * 4, 256) + rdg_sort_layout(capacity).total;
}

int rdg_preprocess_forward(const RdgRasterSettings* s_host, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* opacities, const float* scales,
                           const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                           const float* projmatrix, void* geom_ws, int32_t* radii, int32_t* num_rendered_dev,
                           void* stream) {
    RdgDev d;
    if (rdg_make_dev(s_host, &d)) return -1;
    if (rdg_check_inputs(s_host, shs, colors_precomp, scales, rotations, cov3D_precomp)) return -1;
    hipStream_t st = (hipStream_t)stream;
    rdg_stage_begin(RDG_STAGE_PREPROCESS, st);
    int rc = rdg_launch_preprocess_fwd(d, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                       viewmatrix, projmatrix, geom_ws, radii, num_rendered_dev, st);
    rdg_stage_end(RDG_STAGE_PREPROCESS, st);
    return rc;
}

int rdg_geom_from_records(const RdgRasterSettings* s_host, void* geom_ws, int32_t* radii, int32_t* num_rendered_dev,
                          void* stream) {
    RdgDev d;
    if (rdg_make_dev(s_host, &d)) return -1;
    if (!geom_ws || !radii || !num_rendered_dev) return rdg_set_error("rdg_geom_from_records: NULL argument");
    return rdg_launch_geom_from_records(d, geom_ws, radii, num_rendered_dev, (hipStream_t)stream);
}

int rdg_composite_forward(const RdgRasterSettings* s_host, const float* bg, const void* geom_ws, const int32_t* radii,
                          void* binning_ws, int64_t capacity, void* image_ws, int32_t* num_rendered_dev,
                          float* out_color, float* out_depth, float* out_normal, float* out_alpha, void* stream) {
    RdgDev d;
    if (rdg_make_dev(s_host, &d)) return -1;
    hipStream_t st = (hipStream_t)stream;
    int rc = rdg_launch_bin(d, geom_ws, radii, binning_ws, capacity, image_ws, num_rendered_dev, nullptr, nullptr, st);
    if (rc) return rc;
    rdg_stage_begin(RDG_STAGE_RENDER_FWD, st);
    rc = rdg_launch_render_fwd(d, bg, geom_ws, binning_ws, capacity, image_ws, num_rendered_dev, out_color, out_depth,
                               out_normal, out_alpha, st);
    rdg_stage_end(RDG_STAGE_RENDER_FWD, st);
    return rc;
}

int rdg_rasterize_forward(const RdgRasterSettings* s_host, const float* bg, const float* means3D, const float* shs,
                          const float* colors_precomp, const float* opacities, const float* scales,
                          const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                          const float* projmatrix, void* geom_ws, void* binning_ws, int64_t capacity, void* image_ws,
                          float* out_color, float* out_depth, float* out_normal, float* out_alpha, int32_t* radii,
                          int32_t* num_rendered_dev, void* stream) {
    // the same two stages as rdg_preprocess_forward + rdg_composite_forward, with one difference: knowing the image
    // workspace, the per-Gaussian stage's scan kernel clears the binning counters on its way (no memset launch)
    RdgDev d;
    if (rdg_make_dev(s_host, &d)) return -1;
    if (rdg_check_inputs(s_host, shs, colors_precomp, scales, rotations, cov3D_precomp)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const RdgImageLayout I = rdg_image_layout(d.H, d.W);
    rdg_stage_begin(RDG_STAGE_PREPROCESS, st);
    int rc = rdg_launch_preprocess_fwd(d, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                       viewmatrix, projmatrix, geom_ws, radii, num_rendered_dev, st,
                                       (uint32_t*)((char*)image_ws + I.tile_cnt), rdg_cnt_entries(d.gx, d.gy));
    rdg_stage_end(RDG_STAGE_PREPROCESS, st);
    if (rc) return rc;
    d.tile_cnt_zeroed = 1;
    rc = rdg_launch_bin(d, geom_ws, radii, binning_ws, capacity, image_ws, num_rendered_dev, nullptr, nullptr, st);
    if (rc) return rc;
    rdg_stage_begin(RDG_STAGE_RENDER_FWD, st);
    rc = rdg_launch_render_fwd(d, bg, geom_ws, binning_ws, capacity, image_ws, num_rendered_dev, out_color, out_depth,
                               out_normal, out_alpha, st);
    rdg_stage_end(RDG_STAGE_RENDER_FWD, st);
    return rc;
}

int rdg_composite_backward(const RdgRasterSettings* s_host, const float* bg, const void* geom_ws,
                           const void* binning_ws, int64_t capacity, const void* image_ws,
                           const float* grad_out_color, const float* grad_out_depth, const float* grad_out_alpha,
                           const float* grad_out_normal, void* grad_ws, void* stream) {
    RdgDev d;
    if (rdg_make_dev(s_host, &d)) return -1;
    hipStream_t st = (hipStream_t)stream;
    float* grow = (float*)grad_ws;
    const size_t grow_bytes = rdg_align_up((size_t)(d.P > 0 ? d.P : 1) * RDG_GROW * 4, 256);
    rdg_stage_begin(RDG_STAGE_RENDER_BWD, st);
    if (!d.grad_rows_zeroed) {          // else: cleared by this frame's compositing forward (zero_grad_ws)
        hipError_t e = rdg_zero_async(grow, grow_bytes, st);
        if (e != hipSuccess) { rdg_stage_end(RDG_STAGE_RENDER_BWD, st); return rdg_check_hip(e, "grad row memset"); }
    }
    int rc = rdg_launch_render_bwd(d, bg, geom_ws, binning_ws, capacity, image_ws, grad_out_color, grad_out_depth,
                                   grad_out_alpha, grow, st, nullptr, grad_out_normal);
    rdg_stage_end(RDG_STAGE_RENDER_BWD, st);
    return rc;
}

size_t rdg_det_bytes(int64_t n_instances) {
    return rdg_align_up((size_t)(n_instances > 0 ? n_instances : 1) * 4 * RDG_GROW * 4, 256);
}

int rdg_composite_backward_det(const RdgRasterSettings* s_host, const float* bg, const void* geom_ws,
                               const void* binning_ws, int64_t capacity, const void* image_ws,
                               const float* grad_out_color, const float* grad_out_depth, const float* grad_out_alpha,
                               const float* grad_out_normal, void* grad_ws, void* det_ws, int64_t n_instances,
                               void* stream) {
    RdgDev d;
    if (rdg_make_dev(s_host, &d)) return -1;
    if (!det_ws || !grad_ws) return rdg_set_error("rdg_composite_backward_det: NULL workspace");
    if (n_instances < 0 || n_instances > capacity)
        return rdg_set_error("rdg_composite_backward_det: n_instances must be in [num_rendered, capacity]");
    hipStream_t st = (hipStream_t)stream;
    rdg_stage_begin(RDG_STAGE_RENDER_BWD, st);
    // rows of positions no wave visits stay zero; the gradient rows themselves are written (not accumulated) by the
    // reduction, every one of them
    // (only the rows of the frame's own D instances, read on the device from the per-Gaussian stage's scan total)
    {
        const RdgGeomLayout G = rdg_geom_layout(d.P);
        const int nblk = (d.P + RDG_PRE_BLOCK - 1) / RDG_PRE_BLOCK;
        const uint32_t* d_dev = (const uint32_t*)((const char*)geom_ws + G.block_sums) + (d.P > 0 ? nblk : 0);
        hipLaunchKernelGGL(rdg_zero_rows_dev_kernel, dim3(4096), dim3(256), 0, st, (uint4*)det_ws,
                           (unsigned)(4 * RDG_GROW * 4 / 16), d_dev, (long long)n_instances);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { rdg_stage_end(RDG_STAGE_RENDER_BWD, st); return rdg_check_hip(e, "det row fill"); }
    }
    uint32_t* det_off = (uint32_t*)((char*)grad_ws + rdg_grad_bytes(d.P) - rdg_det_off_bytes(d.P));
    int rc = rdg_launch_render_bwd(d, bg, geom_ws, binning_ws, capacity, image_ws, grad_out_color, grad_out_depth,
                                   grad_out_alpha, (float*)grad_ws, st, (float*)det_ws, grad_out_normal, det_off,
                                   (long long)n_instances);
    rdg_stage_end(RDG_STAGE_RENDER_BWD, st);
    return rc;
}

int rdg_preprocess_backward(const RdgRasterSettings* s_host, const float* means3D, const float* shs,
                            const float* colors_precomp, const float* opacities, const float* scales,
                            const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                            const float* projmatrix, const int32_t* radii, const void* geom_ws, void* grad_ws,
                            float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs, float* dL_dcolors,
                            float* dL_dopacities, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                            float* dL_dviewmatrix, void* stream) {
    RdgDev d;
    if (rdg_make_dev(s_host, &d)) return -1;
    if (rdg_check_inputs(s_host, shs, colors_precomp, scales, rotations, cov3D_precomp)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const size_t grow_bytes = rdg_align_up((size_t)(d.P > 0 ? d.P : 1) * RDG_GROW * 4, 256);
    float* posebuf = (float*)((char*)grad_ws + grow_bytes);
    rdg_stage_begin(RDG_STAGE_PREPROCESS_BWD, st);
    rdg_set_pose_aux((hipStream_t)s_host->aux_stream);
    int rc = rdg_launch_preprocess_bwd(d, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                       viewmatrix, projmatrix, radii, geom_ws, (const float*)grad_ws, posebuf,
                                       dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dcolors, dL_dopacities, dL_dscales,
                                       dL_drotations, dL_dcov3D, dL_dviewmatrix, st);
    rdg_stage_end(RDG_STAGE_PREPROCESS_BWD, st);
    return rc;
}

static int rdg_pre_bwd_adam(const RdgRasterSettings* s_host, const float* means3D, float* shs, const float* opacities,
                            const float* scales, const float* rotations, const float* viewmatrix,
                            const float* projmatrix, const int32_t* radii, const void* geom_ws, void* grad_ws,
                            float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacities, float* dL_dscales,
                            float* dL_drotations, float* dL_dviewmatrix, float* sh_exp_avg, float* sh_exp_avg_sq,
                            int32_t head_len, float lr_head, float lr_tail, double beta1, double beta2, float eps,
                            int32_t step, const RdgStepScalars* dev, void* stream) {
    RdgDev d;
    if (rdg_make_dev(s_host, &d)) return -1;
    if (!shs || !scales || !rotations || !sh_exp_avg || !sh_exp_avg_sq)
        return rdg_set_error("rdg_preprocess_backward_adam: shs, scales, rotations and both moment buffers are required");
    if (rdg_check_inputs(s_host, shs, nullptr, scales, rotations, nullptr)) return -1;
    if (!dev && step < 1) return rdg_set_error("adam: step must be >= 1");
    hipStream_t st = (hipStream_t)stream;
    const size_t grow_bytes = rdg_align_up((size_t)(d.P > 0 ? d.P : 1) * RDG_GROW * 4, 256);
    float* posebuf = (float*)((char*)grad_ws + grow_bytes);
    RdgShAdam ad;
    ad.m = sh_exp_avg; ad.v = sh_exp_avg_sq;
    ad.dev = dev; ad.lr_head = lr_head; ad.lr_tail = lr_tail;
    ad.step_head = ad.step_tail = ad.bc2_sqrt = 0.0f;
    if (!dev) {
        const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
        // the same scalars rdg_adam_step_multi hands its kernel (float(lr) * float(1 / bc1)), so that the fused update and
        // the separate launch produce the same bits
        const float inv_bc1 = (float)(1.0 / bc1);
        ad.step_head = lr_head * inv_bc1; ad.step_tail = lr_tail * inv_bc1;
        ad.bc2_sqrt = (float)sqrt(bc2);
    }
    ad.b1 = (float)beta1; ad.b2 = (float)beta2; ad.omb1 = (float)(1.0 - beta1); ad.omb2 = (float)(1.0 - beta2);
    ad.eps = eps;
    ad.head_len = head_len;
    rdg_stage_begin(RDG_STAGE_PREPROCESS_BWD, st);
    rdg_set_pose_aux((hipStream_t)s_host->aux_stream);
    int rc = rdg_launch_preprocess_bwd(d, means3D, shs, nullptr, opacities, scales, rotations, nullptr, viewmatrix,
                                       projmatrix, radii, geom_ws, (const float*)grad_ws, posebuf, dL_dmeans3D,
                                       dL_dmeans2D, nullptr, nullptr, dL_dopacities, dL_dscales, dL_drotations, nullptr,
                                       dL_dviewmatrix, st, &ad);
    rdg_stage_end(RDG_STAGE_PREPROCESS_BWD, st);
    return rc;
}

int rdg_preprocess_backward_adam(const RdgRasterSettings* s_host, const float* means3D, float* shs,
                                 const float* opacities, const float* scales, const float* rotations,
                                 const float* viewmatrix, const float* projmatrix, const int32_t* radii,
                                 const void* geom_ws, void* grad_ws, float* dL_dmeans3D, float* dL_dmeans2D,
                                 float* dL_dopacities, float* dL_dscales, float* dL_drotations, float* dL_dviewmatrix,
                                 float* sh_exp_avg, float* sh_exp_avg_sq, int32_t head_len, float lr_head, float lr_tail,
                                 double beta1, double beta2, float eps, int32_t step, void* stream) {
    return rdg_pre_bwd_adam(s_host, means3D, shs, opacities, scales, rotations, viewmatrix, projmatrix, radii, geom_ws,
                            grad_ws, dL_dmeans3D, dL_dmeans2D, dL_dopacities, dL_dscales, dL_drotations, dL_dviewmatrix,
                            sh_exp_avg, sh_exp_avg_sq, head_len, lr_head, lr_tail, beta1, beta2, eps, step, nullptr, stream);
}

int rdg_preprocess_backward_adam_dev(const RdgRasterSettings* s_host, const float* means3D, float* shs,
                                     const float* opacities, const float* scales, const float* rotations,
                                     const float* viewmatrix, const float* projmatrix, const int32_t* radii,
                                     const void* geom_ws, void* grad_ws, float* dL_dmeans3D, float* dL_dmeans2D,
                                     float* dL_dopacities, float* dL_dscales, float* dL_drotations,
                                     float* dL_dviewmatrix, float* sh_exp_avg, float* sh_exp_avg_sq, int32_t head_len,
                                     float lr_head, float lr_tail, double beta1, double beta2, float eps,
                                     const RdgStepScalars* dev, void* stream) {
    if (!dev) return rdg_set_error("rdg_preprocess_backward_adam_dev: NULL step scalars");
    return rdg_pre_bwd_adam(s_host, means3D, shs, opacities, scales, rotations, viewmatrix, projmatrix, radii, geom_ws,
                            grad_ws, dL_dmeans3D, dL_dmeans2D, dL_dopacities, dL_dscales, dL_drotations, dL_dviewmatrix,
                            sh_exp_avg, sh_exp_avg_sq, head_len, lr_head, lr_tail, beta1, beta2, eps, 0, dev, stream);
}

int rdg_rasterize_backward(const RdgRasterSettings* s_host, const float* bg, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* opacities, const float* scales,
                           const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                           const float* projmatrix, const int32_t* radii, const void* geom_ws, const void* binning_ws,
                           int64_t capacity, const void* image_ws, const float* grad_out_color,
                           const float* grad_out_depth, const float* grad_out_alpha, const float* grad_out_normal,
                           void* grad_ws, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs, float* dL_dcolors,
                           float* dL_dopacities, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                           float* dL_dviewmatrix, void* stream) {
    if (rdg_check_inputs(s_host, shs, colors_precomp, scales, rotations, cov3D_precomp)) return -1;
    int rc = rdg_composite_backward(s_host, bg, geom_ws, binning_ws, capacity, image_ws, grad_out_color,
                                    grad_out_depth, grad_out_alpha, grad_out_normal, grad_ws, stream);
    if (rc) return rc;
    return rdg_preprocess_backward(s_host, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   viewmatrix, projmatrix, radii, geom_ws, grad_ws, dL_dmeans3D, dL_dmeans2D, dL_dshs,
                                   dL_dcolors, dL_dopacities, dL_dscales, dL_drotations, dL_dcov3D, dL_dviewmatrix,
                                   stream);
}

static int rdg_views_args(const RdgRasterSettings* s_host, int32_t nviews, int32_t stride_rows, const float* shs,
                          const float* scales, const float* rotations) {
    if (nviews < 1 || nviews > RDG_MAX_VIEWS) return rdg_set_error("views: nviews must be 1..%d", RDG_MAX_VIEWS);
    if (stride_rows % RDG_PRE_BLOCK || stride_rows < s_host->P)
        return rdg_set_error("views: stride_rows must be a multiple of %d and >= P", RDG_PRE_BLOCK);
    if (!shs || !scales || !rotations) return rdg_set_error("views: shs, scales and rotations are required");
    if (s_host->M < (s_host->sh_degree + 1) * (s_host->sh_degree + 1)) return rdg_set_error("views: too few SH rows");
    return 0;
}

int rdg_preprocess_forward_views_rows(const RdgRasterSettings* s_host, int32_t nviews, int32_t stride_rows,
                                      int32_t row0, const float* means3D, const float* shs, const float* opacities,
                                      const float* scales, const float* rotations, const float* viewmatrices,
                                      const float* projmatrix, void* geom_ws, int32_t* radii, void* stream) {
    RdgDev d;
    if (rdg_make_dev(s_host, &d)) return -1;
    if (rdg_views_args(s_host, nviews, stride_rows, shs, scales, rotations)) return -1;
    if (row0 < 0 || row0 % RDG_PRE_BLOCK || (int64_t)row0 + s_host->P > stride_rows)
        return rdg_set_error("views: row0 must be a multiple of %d with row0 + P <= stride_rows", RDG_PRE_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    rdg_stage_begin(RDG_STAGE_PREPROCESS, st);
    int rc = rdg_launch_preprocess_fwd_views(d, nviews, stride_rows, row0, means3D, shs, opacities, scales, rotations,
                                             viewmatrices, projmatrix, geom_ws, radii, st);
    rdg_stage_end(RDG_STAGE_PREPROCESS, st);
    return rc;
}

int rdg_preprocess_forward_views(const RdgRasterSettings* s_host, int32_t nviews, int32_t stride_rows,
                                 const float* means3D, const float* shs, const float* opacities, const float* scales,
                                 const float* rotations, const float* viewmatrices, const float* projmatrix,
                                 void* geom_ws, int32_t* radii, void* stream) {
    return rdg_preprocess_forward_views_rows(s_host, nviews, stride_rows, 0, means3D, shs, opacities, scales, rotations,
                                             viewmatrices, projmatrix, geom_ws, radii, stream);
}

int rdg_preprocess_backward_views(const RdgRasterSettings* s_host, int32_t nviews, int32_t stride_rows,
                                  const float* means3D, const float* shs, const float* opacities, const float* scales,
                                  const float* rotations, const float* viewmatrices, const float* projmatrix,
                                  const int32_t* radii, const void* geom_ws, void* grad_ws, float* dL_dmeans3D,
                                  float* dL_dmeans2D, float* dL_dshs, float* dL_dopacities, float* dL_dscales,
                                  float* dL_drotations, float* dL_dviewmatrices, void* stream) {
    RdgDev d;
    if (rdg_make_dev(s_host, &d)) return -1;
    if (rdg_views_args(s_host, nviews, stride_rows, shs, scales, rotations)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const int32_t total = nviews * stride_rows;
    const size_t grow_bytes = rdg_align_up((size_t)total * RDG_GROW * 4, 256);
    float* posebuf = (float*)((char*)grad_ws + grow_bytes);
    RdgStageScope scope(RDG_STAGE_PREPROCESS_BWD, st);
    {
        int rc = rdg_launch_preprocess_bwd_views(d, nviews, stride_rows, means3D, shs, opacities, scales, rotations,
                                                 viewmatrices, projmatrix, radii, geom_ws, (const float*)grad_ws, posebuf,
                                                 dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dopacities, dL_dscales,
                                                 dL_drotations, st);
        if (rc) return rc;
    }
    {
        // the multi-camera kernel runs 128-thread workgroups: one partial pose row per 128 Gaussians
        const int view_rows = stride_rows / 128, nblk = (d.P + 127) / 128;
        float* part = (float*)((char*)posebuf + rdg_align_up(((size_t)total / 128 + 1) * 19 * 4, 256));
        int rc = rdg_launch_pose_reduce_views(nviews, view_rows, d.P > 0 ? nblk : 0, viewmatrices, posebuf, part,
                                              dL_dviewmatrices, st);
        if (rc) return rc;
    }
    return 0;
}

// ---- geom export ---------------------------------------------------------------------------------------------
__global__ void rdg_geom_export_kernel(int P, const RdgRec* __restrict__ rec, const uint32_t* __restrict__ tt,
                                       float* depth, float* xy, float* conic_opacity, float* rgb, float* normal,
                                       uint32_t* tiles_touched) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const RdgRec r = rec[i];
    if (depth) depth[i] = r.q1.z;
    if (xy) { xy[2 * i] = r.q0.x; xy[2 * i + 1] = r.q0.y; }
    if (conic_opacity) {
        conic_opacity[4 * i] = r.q0.z; conic_opacity[4 * i + 1] = r.q0.w; conic_opacity[4 * i + 2] = r.q1.x;
        conic_opacity[4 * i + 3] = r.q1.y;
    }
    if (rgb) { rgb[3 * i] = r.q2.x; rgb[3 * i + 1] = r.q2.y; rgb[3 * i + 2] = r.q2.z; }
    if (normal) { normal[3 * i] = r.q3.x; normal[3 * i + 1] = r.q3.y; normal[3 * i + 2] = r.q3.z; }
    if (tiles_touched) tiles_touched[i] = tt[i];
}

int rdg_geom_export(int32_t P, const void* geom_ws, float* depth, float* xy, float* conic_opacity, float* rgb,
                    float* normal, uint32_t* tiles_touched, void* stream) {
    if (P <= 0) return 0;
    const RdgGeomLayout G = rdg_geom_layout(P);
    const char* g = (const char*)geom_ws;
    hipLaunchKernelGGL(rdg_geom_export_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P,
                       (const RdgRec*)(g + G.rec), (const uint32_t*)(g + G.tiles_touched), depth, xy, conic_opacity,
                       rgb, normal, tiles_touched);
    return rdg_check_hip(hipGetLastError(), "geom_export launch");
}

int rdg_image_export(int32_t H, int32_t W, const void* image_ws, float* final_T, uint32_t* n_contrib, void* stream) {
    if (H <= 0 || W <= 0 || !image_ws) return rdg_set_error("rdg_image_export: bad arguments");
    const RdgImageLayout I = rdg_image_layout(H, W);
    const size_t bytes = (size_t)H * W * 4;
    hipStream_t st = (hipStream_t)stream;
    if (final_T) {
        hipError_t e = hipMemcpyAsync(final_T, (const char*)image_ws + I.final_T, bytes, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return rdg_check_hip(e, "image_export final_T");
    }
    if (n_contrib) {
        hipError_t e = hipMemcpyAsync(n_contrib, (const char*)image_ws + I.n_contrib, bytes, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return rdg_check_hip(e, "image_export n_contrib");
    }
    return 0;
}

__global__ void rdg_copy_u64_kernel(const uint64_t* __restrict__ a, uint64_t* __restrict__ b, long long cap,
                                    const int32_t* __restrict__ n_dev) {
    long long n = *n_dev; if (n > cap) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        b[i] = a[i];
}
__global__ void rdg_copy_u32_kernel(const uint32_t* __restrict__ a, uint32_t* __restrict__ b, long long cap,
                                    const int32_t* __restrict__ n_dev) {
    long long n = *n_dev; if (n > cap) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        b[i] = a[i];
}

int rdg_bin_forward(const RdgRasterSettings* s_host, const void* geom_ws, const int32_t* radii, void* binning_ws,
                    int64_t capacity, void* image_ws, int32_t* num_rendered_dev, uint64_t* keys_unsorted,
                    uint32_t* vals_unsorted, uint64_t* keys_sorted, uint32_t* vals_sorted, uint32_t* ranges,
                    void* stream) {
    RdgDev d;
    if (rdg_make_dev(s_host, &d)) return -1;
    hipStream_t st = (hipStream_t)stream;
    int rc = rdg_launch_bin(d, geom_ws, radii, binning_ws, capacity, image_ws, num_rendered_dev, keys_unsorted,
                            vals_unsorted, st, keys_sorted != nullptr);
    if (rc) return rc;
    const RdgBinLayout B = rdg_bin_layout(capacity);
    const RdgImageLayout I = rdg_image_layout(d.H, d.W);
    const int n_tiles = d.gx * d.gy;
    const int npass = (rdg_key_bits(n_tiles) + RDG_SORT_BITS - 1) / RDG_SORT_BITS;
    char* b = (char*)binning_ws;
    if (keys_sorted)
        hipLaunchKernelGGL(rdg_copy_u64_kernel, dim3(1024), dim3(256), 0, st,
                           (const uint64_t*)(b + ((npass & 1) ? B.keys_b : B.keys_a)), keys_sorted, (long long)capacity,
                           num_rendered_dev);
    if (vals_sorted)
        hipLaunchKernelGGL(rdg_copy_u32_kernel, dim3(1024), dim3(256), 0, st,
                           (const uint32_t*)(b + ((npass & 1) ? B.vals_b : B.vals_a)), vals_sorted, (long long)capacity,
                           num_rendered_dev);
    if (ranges) {
        hipError_t e = hipMemcpyAsync(ranges, (char*)image_ws + I.ranges, (size_t)n_tiles * 8, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return rdg_check_hip(e, "ranges copy");
    }
    return rdg_check_hip(hipGetLastError(), "bin_forward");
}

int rdg_sort_pairs(uint64_t* keys, uint32_t* vals, int64_t capacity, const int32_t* n_dev, int32_t end_bit,
                   void* tmp_ws, void* stream) {
    if (end_bit <= 0 || end_bit > 64) return rdg_set_error("end_bit must be 1..64");
    hipStream_t st = (hipStream_t)stream;
    size_t cap = (size_t)(capacity > 0 ? capacity : 1);
    char* t = (char*)tmp_ws;
    uint64_t* keys_b = (uint64_t*)t;
    uint32_t* vals_b = (uint32_t*)(t + rdg_align_up(cap * 8, 256));
    void* tables = t + rdg_align_up(cap * 8, 256) + rdg_align_up(cap * 4, 256);
    int in_b = 0;
    int rc = rdg_launch_sort(keys, keys_b, vals, vals_b, capacity, n_dev, end_bit, tables, &in_b, st);
    if (rc) return rc;
    if (in_b) {
        hipLaunchKernelGGL(rdg_copy_u64_kernel, dim3(1024), dim3(256), 0, st, keys_b, keys, (long long)capacity, n_dev);
        hipLaunchKernelGGL(rdg_copy_u32_kernel, dim3(1024), dim3(256), 0, st, vals_b, vals, (long long)capacity, n_dev);
    }
    return rdg_check_hip(hipGetLastError(), "sort_pairs");
}

int rdg_densify_stats(int64_t n, int64_t row0, const float* dL_dmeans2D, const int32_t* radii, float* grad_accum,
                      float* denom, float* max_radii, void* stream) {
    if (n < 0 || row0 < 0) return rdg_set_error("rdg_densify_stats: n and row0 must not be negative");
    if (n == 0) return 0;
    if (!radii || (grad_accum && !dL_dmeans2D)) return rdg_set_error("rdg_densify_stats: radii (and dL_dmeans2D) required");
    return rdg_launch_densify_stats((long long)n, (long long)row0, dL_dmeans2D, radii, grad_accum, denom, max_radii,
                                    (hipStream_t)stream);
}

int rdg_timing_enable(int32_t on) { g_timing = on ? 1 : 0; return 0; }
int rdg_timing_select(uint32_t stage_mask) { g_timing_mask = stage_mask; return 0; }
int rdg_timing_reset(void) {
    std::lock_guard<std::mutex> lk(g_timing_mu);
    rdg_timing_drain();
    memset(g_total_ms, 0, sizeof(g_total_ms));
    memset(g_count, 0, sizeof(g_count));
    return 0;
}
int rdg_stage_time_ms(int32_t stage, double* total_ms, int64_t* count) {
    if (stage < 0 || stage >= RDG_STAGE_COUNT) return rdg_set_error("bad stage id");
    std::lock_guard<std::mutex> lk(g_timing_mu);
    rdg_timing_drain();
    if (total_ms) *total_ms = g_total_ms[stage];
    if (count) *count = g_count[stage];
    return 0;
}

}  // extern "C"
