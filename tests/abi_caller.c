/*
 * abi_caller.c -- a caller of librodygs_hip.so with NO torch and NO Python in the process ("plain
 * pointers and sizes, no torch types; the .so never allocates"): plain C, the HIP runtime for hipMalloc / hipMemcpy, the
 * library through dlopen -- the way a cgo / JNI / N-API binding of include/rodygs_hip.h would reach it.
 *
 *     abi_caller <librodygs_hip.so> <dir>
 *
 * <dir>/meta.txt: "P M sh_degree H W tanfovx tanfovy cull" ; <dir>/in_<name>.bin: raw little-endian float32 inputs
 * (means3D, shs, opacities, scales, rotations, viewmatrix, projmatrix, bg) and upstream gradients (g_color, g_depth, g_alpha).
 * Runs  rdg_preprocess_forward + rdg_bin_forward  (the exported key stream),  rdg_rasterize_forward,  rdg_rasterize_backward,
 * and writes every output as <dir>/out_<name>.bin.  tests/test_gpu_round6.py starts it as a child process and compares the
 * files with the committed fixture tests/golden/rasterizer_golden_c1.npz.  Exit code 0 = every call returned 0.
 *
 * Build (cross-compiles without a GPU; __graft_entry__.build() and `make -C rodygs_amd/csrc abi_caller` do it):
 *     hipcc -x c -std=c99 -O1 -I include tests/abi_caller.c -o tests/abi_caller -ldl
 */
#define __HIP_PLATFORM_AMD__ 1
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rodygs_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static const char* g_dir;

static void* read_file(const char* name, size_t bytes) {
    char path[1024];
    snprintf(path, sizeof(path), "%s/%s", g_dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(3); }
    void* p = malloc(bytes ? bytes : 1);
    if (fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "%s: short read (%zu bytes wanted)\n", path, bytes); exit(3); }
    fclose(f);
    return p;
}
static void* to_device(const char* name, size_t bytes) {
    void* h = read_file(name, bytes);
    void* d = NULL;
    if (hipMalloc(&d, bytes ? bytes : 4) != hipSuccess) { fprintf(stderr, "hipMalloc(%zu) failed\n", bytes); exit(2); }
    if (hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "H2D %s failed\n", name); exit(2); }
    free(h);
    return d;
}
static void* dev_alloc(size_t bytes) {
    void* d = NULL;
    if (hipMalloc(&d, bytes ? bytes : 4) != hipSuccess) { fprintf(stderr, "hipMalloc(%zu) failed\n", bytes); exit(2); }
    /* poison: the library must write everything it hands back and zero what it accumulates into */
    if (hipMemset(d, 0xA5, bytes ? bytes : 4) != hipSuccess) exit(2);
    return d;
}
static void write_device(const char* name, const void* d, size_t bytes) {
    void* h = malloc(bytes ? bytes : 1);
    if (hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "D2H %s failed\n", name); exit(2); }
    char path[1024];
    snprintf(path, sizeof(path), "%s/%s", g_dir, name);
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(h, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", path); exit(3); }
    fclose(f);
    free(h);
}

/* the entry points this caller binds, typed from the header's declarations */
typedef int (*abi_version_t)(void);
typedef size_t (*settings_bytes_t)(void);
typedef const char* (*last_error_t)(void);
typedef size_t (*bytes_p_t)(int32_t);
typedef size_t (*bytes_bin_t)(int64_t, int32_t);
typedef size_t (*bytes_img_t)(int32_t, int32_t);
typedef __typeof__(&rdg_preprocess_forward) preprocess_forward_t;
typedef __typeof__(&rdg_bin_forward) bin_forward_t;
typedef __typeof__(&rdg_rasterize_forward) rasterize_forward_t;
typedef __typeof__(&rdg_rasterize_backward) rasterize_backward_t;
typedef __typeof__(&rdg_image_export) image_export_t;

#define BIND(var, type, name) type var = (type)dlsym(lib, name); if (!var) { fprintf(stderr, "missing symbol %s\n", name); return 4; }
#define CALL(expr) do { int rc_ = (expr); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, last_error()); return 5; } } while (0)

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s <librodygs_hip.so> <dir>\n", argv[0]); return 1; }
    g_dir = argv[2];
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 4; }
    BIND(abi_version, abi_version_t, "rdg_abi_version")
    BIND(settings_bytes, settings_bytes_t, "rdg_settings_bytes")
    BIND(last_error, last_error_t, "rdg_last_error")
    BIND(geom_bytes, bytes_p_t, "rdg_geom_bytes")
    BIND(grad_bytes, bytes_p_t, "rdg_grad_bytes")
    BIND(binning_bytes, bytes_bin_t, "rdg_binning_bytes")
    BIND(image_bytes, bytes_img_t, "rdg_image_bytes")
    BIND(preprocess_forward, preprocess_forward_t, "rdg_preprocess_forward")
    BIND(bin_forward, bin_forward_t, "rdg_bin_forward")
    BIND(rasterize_forward, rasterize_forward_t, "rdg_rasterize_forward")
    BIND(rasterize_backward, rasterize_backward_t, "rdg_rasterize_backward")
    BIND(image_export, image_export_t, "rdg_image_export")
    /* the header's two guards */
    if (abi_version() != RDG_ABI_VERSION) { fprintf(stderr, "ABI %d, header %d\n", abi_version(), RDG_ABI_VERSION); return 4; }
    if (settings_bytes() != sizeof(RdgRasterSettings)) { fprintf(stderr, "settings struct size differs\n"); return 4; }

    int P, M, deg, H, W, cull;
    float tanx, tany;
    {
        char path[1024];
        snprintf(path, sizeof(path), "%s/meta.txt", g_dir);
        FILE* f = fopen(path, "r");
        if (!f || fscanf(f, "%d %d %d %d %d %f %f %d", &P, &M, &deg, &H, &W, &tanx, &tany, &cull) != 8) {
            fprintf(stderr, "bad meta.txt\n"); return 3;
        }
        fclose(f);
    }
    const size_t hw = (size_t)H * W;
    const int n_tiles = ((W + 15) / 16) * ((H + 15) / 16);
    CHECK_HIP(hipSetDevice(0));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));

    float* means3D = (float*)to_device("in_means3D.bin", (size_t)P * 3 * 4);
    float* shs = (float*)to_device("in_shs.bin", (size_t)P * M * 3 * 4);
    float* opac = (float*)to_device("in_opacities.bin", (size_t)P * 4);
    float* scales = (float*)to_device("in_scales.bin", (size_t)P * 3 * 4);
    float* rots = (float*)to_device("in_rotations.bin", (size_t)P * 4 * 4);
    float* view = (float*)to_device("in_viewmatrix.bin", 64);
    float* proj = (float*)to_device("in_projmatrix.bin", 64);
    float* bg = (float*)to_device("in_bg.bin", 12);
    float* g_color = (float*)to_device("in_g_color.bin", 3 * hw * 4);
    float* g_depth = (float*)to_device("in_g_depth.bin", hw * 4);
    float* g_alpha = (float*)to_device("in_g_alpha.bin", hw * 4);

    RdgRasterSettings s;
    memset(&s, 0, sizeof(s));
    s.P = P; s.M = M; s.sh_degree = deg; s.image_height = H; s.image_width = W;
    s.tanfovx = tanx; s.tanfovy = tany; s.scale_modifier = 1.0f;
    s.enable_cov_grad = 1; s.enable_sh_grad = 1; s.render_normal = 1; s.cull = cull;

    const int64_t cap = 4 * (int64_t)P + 4096;
    void* geom = dev_alloc(geom_bytes(P));
    void* binning = dev_alloc(binning_bytes(cap, n_tiles));
    void* image = dev_alloc(image_bytes(H, W));
    void* grad_ws = dev_alloc(grad_bytes(P));
    int32_t* radii = (int32_t*)dev_alloc((size_t)P * 4);
    int32_t* nren = (int32_t*)dev_alloc(8);
    uint64_t* keys_sorted = (uint64_t*)dev_alloc((size_t)cap * 8);
    uint32_t* vals_sorted = (uint32_t*)dev_alloc((size_t)cap * 4);
    uint32_t* ranges = (uint32_t*)dev_alloc((size_t)n_tiles * 8);
    CHECK_HIP(hipMemset(ranges, 0, (size_t)n_tiles * 8));

    /* 1. the stage entry points: the exported key stream */
    CALL(preprocess_forward(&s, means3D, shs, NULL, opac, scales, rots, NULL, view, proj, geom, radii, nren, stream));
    CALL(bin_forward(&s, geom, radii, binning, cap, image, nren, NULL, NULL, keys_sorted, vals_sorted, ranges, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    int32_t D = 0;
    CHECK_HIP(hipMemcpy(&D, nren, 4, hipMemcpyDeviceToHost));
    if (D < 0 || D > cap) { fprintf(stderr, "D = %d outside [0, %lld]\n", D, (long long)cap); return 6; }
    write_device("out_nren.bin", nren, 4);
    write_device("out_radii_stage.bin", radii, (size_t)P * 4);
    write_device("out_keys_sorted.bin", keys_sorted, (size_t)D * 8);
    write_device("out_vals_sorted.bin", vals_sorted, (size_t)D * 4);
    write_device("out_ranges.bin", ranges, (size_t)n_tiles * 8);

    /* 2. the whole forward */
    float* color = (float*)dev_alloc(3 * hw * 4);
    float* depth = (float*)dev_alloc(hw * 4);
    float* normal = (float*)dev_alloc(3 * hw * 4);
    float* alpha = (float*)dev_alloc(hw * 4);
    CALL(rasterize_forward(&s, bg, means3D, shs, NULL, opac, scales, rots, NULL, view, proj, geom, binning, cap, image, color,
                           depth, normal, alpha, radii, nren, stream));
    float* final_T = (float*)dev_alloc(hw * 4);
    uint32_t* n_contrib = (uint32_t*)dev_alloc(hw * 4);
    CALL(image_export(H, W, image, final_T, n_contrib, stream));

    /* 3. the backward of the fixture's loss */
    float* d_means3D = (float*)dev_alloc((size_t)P * 3 * 4);
    float* d_means2D = (float*)dev_alloc((size_t)P * 3 * 4);
    float* d_shs = (float*)dev_alloc((size_t)P * M * 3 * 4);
    float* d_opac = (float*)dev_alloc((size_t)P * 4);
    float* d_scales = (float*)dev_alloc((size_t)P * 3 * 4);
    float* d_rots = (float*)dev_alloc((size_t)P * 4 * 4);
    float* d_view = (float*)dev_alloc(64);
    CALL(rasterize_backward(&s, bg, means3D, shs, NULL, opac, scales, rots, NULL, view, proj, radii, geom, binning, cap, image,
                            g_color, g_depth, g_alpha, NULL, grad_ws, d_means3D, d_means2D, d_shs, NULL, d_opac, d_scales,
                            d_rots, NULL, d_view, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    write_device("out_color.bin", color, 3 * hw * 4);
    write_device("out_depth.bin", depth, hw * 4);
    write_device("out_normal.bin", normal, 3 * hw * 4);
    write_device("out_alpha.bin", alpha, hw * 4);
    write_device("out_radii.bin", radii, (size_t)P * 4);
    write_device("out_final_T.bin", final_T, hw * 4);
    write_device("out_n_contrib.bin", n_contrib, hw * 4);
    write_device("out_d_means3D.bin", d_means3D, (size_t)P * 3 * 4);
    write_device("out_d_means2D.bin", d_means2D, (size_t)P * 3 * 4);
    write_device("out_d_shs.bin", d_shs, (size_t)P * M * 3 * 4);
    write_device("out_d_opacities.bin", d_opac, (size_t)P * 4);
    write_device("out_d_scales.bin", d_scales, (size_t)P * 3 * 4);
    write_device("out_d_rotations.bin", d_rots, (size_t)P * 4 * 4);
    write_device("out_d_viewmatrix.bin", d_view, 64);
    printf("abi_caller: P=%d D=%d cull=%d ok\n", P, D, cull);
    return 0;
}
