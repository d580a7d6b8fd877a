// valu_probe.hip -- wall-clock issue cost of single VALU / LDS instructions on gfx950 with the SIMDs saturated
// (4 waves per SIMD on every CU, 8 independent chains per wave).  Reported: ns per wave-instruction per SIMD and the
// ratio to v_mul_f32.  The shader-clock counter is useless here (wave 0 of a workgroup is favoured by the arbiter), and
// the chip's clock moves with the power drawn, so only wall-clock ratios inside one run mean anything.
//   hipcc -O3 --offload-arch=gfx950 valu_probe.hip -o valu_probe && ./valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP 64
#define ITERS 16384

typedef float float2v __attribute__((ext_vector_type(2)));

#define OPS(X) \
    X(0,  "v_mul_f32 (VOP2)",            "v_mul_f32 %0, %0, %2",                               a) \
    X(1,  "v_add_f32",                   "v_add_f32 %0, %0, %2",                               a) \
    X(2,  "v_sub_f32",                   "v_sub_f32 %0, %0, %2",                               a) \
    X(3,  "v_fma_f32",                   "v_fma_f32 %0, %0, %2, %3",                           a) \
    X(4,  "v_fmac_f32",                  "v_fmac_f32 %0, %2, %3",                              a) \
    X(5,  "v_min_f32",                   "v_min_f32 %0, %0, %2",                               a) \
    X(6,  "v_max_f32",                   "v_max_f32 %0, %0, %2",                               a) \
    X(7,  "v_mul_f32 literal",           "v_mul_f32 %0, 0x3f7d70a4, %0",                       a) \
    X(8,  "v_mul_f32 sgpr",              "v_mul_f32 %0, %4, %0",                               a) \
    X(9,  "v_mul_f32 e64",               "v_mul_f32_e64 %0, %0, %2",                           a) \
    X(10, "v_mov_b32",                   "v_mov_b32 %0, %2",                                   a) \
    X(11, "v_cndmask_b32 vcc",           "v_cndmask_b32 %0, %0, %2, vcc",                      a) \
    X(12, "v_cndmask_b32 e64 sgpr",      "v_cndmask_b32_e64 %0, %0, %2, s[20:21]",             a) \
    X(13, "v_cmp_lt_f32 vcc",            "v_cmp_lt_f32 vcc, %0, %2",                           a) \
    X(14, "v_cmp_lt_f32 e64 sgpr",       "v_cmp_lt_f32_e64 s[22:23], %0, %2",                  a) \
    X(15, "v_add_u32",                   "v_add_u32 %0, %0, %2",                               a) \
    X(16, "v_lshlrev_b32",               "v_lshlrev_b32 %0, 1, %0",                            a) \
    X(17, "v_and_b32",                   "v_and_b32 %0, %0, %2",                               a) \
    X(18, "v_readfirstlane_b32",         "v_readfirstlane_b32 s24, %0",                        a) \
    X(19, "v_add_f32_dpp quad_perm",     "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", a) \
    X(20, "v_add_f32_dpp row_ror:8",     "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf", a) \
    X(21, "v_add_f32_dpp half_mirror",   "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5", a) \
    X(22, "v_mov_b32_dpp quad_perm",     "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", a) \
    X(23, "v_exp_f32",                   "v_exp_f32 %0, %0",                                   a) \
    X(24, "v_rcp_f32",                   "v_rcp_f32 %0, %0",                                   a) \
    X(25, "v_log_f32",                   "v_log_f32 %0, %0",                                   a) \
    X(26, "v_pk_mul_f32",                "v_pk_mul_f32 %1, %1, %5",                            p) \
    X(27, "v_pk_add_f32",                "v_pk_add_f32 %1, %1, %5",                            p) \
    X(28, "v_pk_fma_f32",                "v_pk_fma_f32 %1, %1, %5, %6",                        p) \
    X(29, "v_med3_f32",                  "v_med3_f32 %0, %0, %2, %3",                          a) \
    X(30, "v_max3_f32",                  "v_max3_f32 %0, %0, %2, %3",                          a) \
    X(31, "v_mad_u32_u24",               "v_mad_u32_u24 %0, %0, %2, %3",                       a) \
    X(32, "v_cvt_f32_i32",               "v_cvt_f32_i32 %0, %0",                               a) \
    X(33, "v_mul_f32 + s_nop 0 pairs",   "v_mul_f32 %0, %0, %2\n\ts_nop 0",                    a) \
    X(34, "v_fma_f32 (2 distinct srcs)", "v_fma_f32 %0, %2, %3, %0",                           a) \
    X(35, "v_mul_legacy / v_mul_f32 neg","v_mul_f32_e64 %0, -%0, %2",                          a) \
    X(36, "v_bfe_u32",                   "v_bfe_u32 %0, %0, 3, 5",                             a) \
    X(37, "v_fma_f32 sgpr src",          "v_fma_f32 %0, %0, %4, %3",                           a) \
    X(38, "v_sub_f32 sgpr src",          "v_sub_f32 %0, %4, %0",                               a) \
    X(39, "v_min_f64",                   "v_min_f64 %1, %1, %5",                               p) \
    X(40, "v_max_f64",                   "v_max_f64 %1, %1, %5",                               p) \
    X(41, "v_cmp_lt_u64 vcc",            "v_cmp_lt_u64 vcc, %1, %5",                           p) \
    X(42, "v_and_or_b32 sgpr",           "v_and_or_b32 %0, %0, %4, %2",                        a) \
    X(43, "v_mov_b32 sgpr src",          "v_mov_b32 %0, %4",                                   a) \
    X(44, "v_cmp_ge_u32 e64 sgpr dst",   "v_cmp_ge_u32_e64 s[22:23], %0, %2",                  a) \
    X(45, "v_fma_f32 inline const",      "v_fma_f32 %0, %0, 0.5, %3",                          a) \
    X(46, "v_add_f64",                   "v_add_f64 %1, %1, %5",                               p) \
    X(47, "v_pk_min... v_min3_f32",      "v_min3_f32 %0, %0, %2, %3",                          a) \
    X(48, "v_mul_f32 + v_exp pairs",     "v_mul_f32 %0, %0, %2\n\tv_exp_f32 %0, %0",            a) \
    X(49, "v_permlane32_swap",           "v_permlane32_swap_b32 %0, %0",                       a)

template <int MODE>
__global__ void __launch_bounds__(1024) probe(float* out, float seed, float sc) {
    float a[8];
    float2v p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = float2v{seed + i, seed - i + threadIdx.x}; }
    const float m = 1.0001f, c = 0.0001f;
    const float2v m2 = {1.0001f, 0.9999f}, c2 = {0.0001f, 0.0002f};
    asm volatile("s_mov_b64 s[20:21], 0x5555\n\tv_cmp_lt_f32 vcc, %0, %1" :: "v"(a[0]), "v"(m) : "vcc", "s20", "s21");
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#define X(ID, NAME, TXT, KIND)                                                                                          \
                if (MODE == ID) asm volatile(TXT : "+v"(a[i]), "+v"(p[i]) : "v"(m), "v"(c), "s"(sc), "v"(m2), "v"(c2)   \
                                             : "vcc", "s22", "s23", "s24");
                OPS(X)
#undef X
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float g_ref = 0.f;

template <int MODE>
static void run(const char* name) {
    const int threads = 1024, blocks = 256;
    float* out;
    if (hipMalloc(&out, sizeof(float) * blocks * threads) != hipSuccess) exit(1);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, 1.0001f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, 1.0001f);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)ITERS * REP * 4;   // wave-instructions per SIMD
    const float ns = ms * 1e6 / n;
    if (MODE == 0) g_ref = ns;
    printf("%-32s %6.3f ns per wave-inst per SIMD   %5.2f x v_mul_f32\n", name, ns, ns / g_ref);
    (void)hipFree(out);
}

int main() {
    for (int pass = 0; pass < 2; ++pass) {
        printf("---- pass %d (pass 0 warms the power state)\n", pass);
#define X(ID, NAME, TXT, KIND) run<ID>(NAME);
        OPS(X)
#undef X
    }
    return 0;
}
