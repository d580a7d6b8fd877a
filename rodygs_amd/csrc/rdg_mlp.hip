// rdg_mlp.hip -- the time-deformation MLP of RoDyGS on the matrix cores.
//
// Reference: MLPBasisNetwork (/root/reference/src/model/rodygs_dynamic.py:243-327): time embedding [N,53] ->
// timenet 53-128-128-64 (GELU) -> 16 heads 64-32-7 (GELU) -> motion bases [N,16,7]; N = number of distinct birth
// times + 1 (<= ~100).  In the reference (and in any framework) this is ~40 launches forward and ~100 backward of
// launch-latency-bound micro-kernels.  Here every product is one launch of ONE generic kernel:
//
//   rdg_gemm16_kernel : C[M,N] = epilogue( A[M,K] . B[K,N] ), four wave64 per 16x16 output tile (split K, LDS
//   reduction), accumulating with v_mfma_f32_16x16x4_f32 (f32 in / f32 acc: each partial is a k-ordered fmaf chain).
//   Arbitrary element strides for A, B and C let the same kernel run X.W^T, dZ.W and dZ^T.A (K = batch rows) without
//   any transposed copies; blockIdx.y batches the 16 heads.  Epilogues: +bias, GELU (storing the pre-activation for
//   backward), or multiplication by GELU'(pre-activation).
//
// Forward = 5 launches, backward = 5 launches (a layer's weight gradient and input gradient share one launch; bias
// gradients ride along as an all-ones column of the dW products).  The matrices are
// tiny (<= 128x128): MFMA is used because the shape is a GEMM, not because it is the bottleneck.
#include "rdg_common.h"
#include <math.h>

typedef float rdg_v4f __attribute__((ext_vector_type(4)));

struct RdgGemm {
    int M, N, K;
    const float* A; long long sam, sak, sab;   // A(m,k) = A[b*sab + m*sam + k*sak]
    const float* B; long long sbk, sbn, sbb;   // B(k,n)
    float* C;       long long scm, scn, scb;   // C(m,n)
    const float* bias; long long sbias;        // bias[b*sbias + n] or NULL
    const float* aux;                          // epilogue 2: pre-activation, indexed like C
    float* C2;                                 // epilogue 1: activated copy, indexed like C
    int epi;                                   // 0: none, 1: C = z, C2 = gelu(z), 2: C = acc * gelu'(aux)
    float* rowsum; long long srsb;             // optional: rowsum[b*srsb + m] = sum_k A(m,k)  (bias gradients: an
                                               // extra all-ones column of B, computed by the same MFMA tiles)
};

__device__ __forceinline__ float rdg_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float rdg_gelu_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// Up to two independent products per launch (blockIdx.z): in the backward pass the weight gradient of a layer and the
// input gradient handed to the previous layer depend on the same dZ only.
struct RdgGemm2 { RdgGemm g[2]; };

// One workgroup = 4 waves per 16x16 output tile: the waves split K in interleaved 32-wide blocks (the loop is a
// dependent load -> MFMA chain, so K = 512 on one wave was 33 us), partial tiles meet in LDS, wave 0 runs the epilogue.
__global__ void __launch_bounds__(256) rdg_gemm16_kernel(RdgGemm2 p) {
    const RdgGemm& g = p.g[blockIdx.z];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int Neff = g.N + (g.rowsum ? 1 : 0);
    const int tn = (Neff + 15) >> 4;
    const int tile = blockIdx.x;
    if (tile >= tn * ((g.M + 15) >> 4)) return;
    const int m0 = (tile / tn) << 4, n0 = (tile % tn) << 4;
    const long long b = blockIdx.y;
    const float* A = g.A + b * g.sab;
    const float* B = g.B + b * g.sbb;
    const int li = lane & 15, lk = lane >> 4;
    const bool am = (m0 + li) < g.M, bn = (n0 + li) < g.N;
    const bool ones = g.rowsum && (n0 + li) == g.N;
    const float* ap = A + (long long)(am ? (m0 + li) : 0) * g.sam;
    const float* bp = B + (long long)(bn ? (n0 + li) : 0) * g.sbn;
    rdg_v4f acc = {0.f, 0.f, 0.f, 0.f};
    // 8 k-steps (32 k values) per iteration: all 16 loads first, then 8 dependent MFMAs (the loop is load-latency bound)
    for (int k0 = 32 * wv; k0 < g.K; k0 += 128) {
        float a4[8], b4[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u + lk;
            const int kc = k < g.K ? k : 0;
            const float av = ap[(long long)kc * g.sak];
            const float bv = bp[(long long)kc * g.sbk];
            const bool kv = k < g.K;
            a4[u] = (am && kv) ? av : 0.0f;
            b4[u] = kv ? (ones ? 1.0f : (bn ? bv : 0.0f)) : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u], b4[u], acc, 0, 0, 0);
    }
    __shared__ rdg_v4f red[3][64];
    if (wv) red[wv - 1][lane] = acc;
    __syncthreads();
    if (wv) return;
    acc = (acc + red[0][lane]) + (red[1][lane] + red[2][lane]);
    // D layout: column = lane & 15, row = (lane >> 4) * 4 + r
    const int n = n0 + li;
    if (n >= Neff) return;
    if (ones) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + lk * 4 + r;
            if (m < g.M) g.rowsum[b * g.srsb + m] = acc[r];
        }
        return;
    }
    const float bias = g.bias ? g.bias[b * g.sbias + n] : 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + lk * 4 + r;
        if (m >= g.M) continue;
        const long long ci = b * g.scb + (long long)m * g.scm + (long long)n * g.scn;
        float v = acc[r] + bias;
        if (g.epi == 1) { g.C[ci] = v; g.C2[ci] = rdg_gelu(v); }
        else if (g.epi == 2) { g.C[ci] = v * rdg_gelu_grad(g.aux[ci]); }
        else { g.C[ci] = v; }
    }
}

// dst[b*sdb + n] = sum_m src[b*ssb + m*ssm + n*ssn]   (bias gradients; M <= a few hundred)
__global__ void rdg_colsum_kernel(int M, int N, const float* __restrict__ src, long long ssm, long long ssn, long long ssb,
                                  float* __restrict__ dst, long long sdb) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const long long b = blockIdx.y;
    float s = 0.0f;
    for (int m = 0; m < M; ++m) s += src[b * ssb + (long long)m * ssm + (long long)n * ssn];
    dst[b * sdb + n] = s;
}

struct RdgGemmJob { RdgGemm g; int batch, tiles; };
static RdgGemmJob rdg_gemm_job(int batch, int M, int N, int K, const float* A, long long sam, long long sak,
                               long long sab, const float* B, long long sbk, long long sbn, long long sbb, float* C,
                               long long scm, long long scn, long long scb, const float* bias, long long sbias, int epi,
                               const float* aux, float* C2, float* rowsum = nullptr, long long srsb = 0) {
    RdgGemmJob j;
    RdgGemm& g = j.g;
    g.M = M; g.N = N; g.K = K;
    g.A = A; g.sam = sam; g.sak = sak; g.sab = sab;
    g.B = B; g.sbk = sbk; g.sbn = sbn; g.sbb = sbb;
    g.C = C; g.scm = scm; g.scn = scn; g.scb = scb;
    g.bias = bias; g.sbias = sbias; g.aux = aux; g.C2 = C2; g.epi = epi; g.rowsum = rowsum; g.srsb = srsb;
    j.batch = batch;
    j.tiles = ((M + 15) / 16) * ((N + (rowsum ? 1 : 0) + 15) / 16);
    return j;
}
// one launch for one product, or for two independent products with the same batch count
static void rdg_gemm_run(hipStream_t st, const RdgGemmJob& a, const RdgGemmJob* b = nullptr) {
    RdgGemm2 p;
    p.g[0] = a.g;
    p.g[1] = b ? b->g : a.g;
    const int tiles = b && b->tiles > a.tiles ? b->tiles : a.tiles;
    hipLaunchKernelGGL(rdg_gemm16_kernel, dim3(tiles, a.batch, b ? 2 : 1), dim3(256), 0, st, p);
}
static void rdg_gemm(hipStream_t st, int batch, int M, int N, int K, const float* A, long long sam, long long sak,
                     long long sab, const float* B, long long sbk, long long sbn, long long sbb, float* C, long long scm,
                     long long scn, long long scb, const float* bias, long long sbias, int epi, const float* aux,
                     float* C2, float* rowsum = nullptr, long long srsb = 0) {
    rdg_gemm_run(st, rdg_gemm_job(batch, M, N, K, A, sam, sak, sab, B, sbk, sbn, sbb, C, scm, scn, scb, bias, sbias, epi,
                                  aux, C2, rowsum, srsb));
}
static void rdg_colsum(hipStream_t st, int batch, int M, int N, const float* src, long long ssm, long long ssn,
                       long long ssb, float* dst, long long sdb) {
    hipLaunchKernelGGL(rdg_colsum_kernel, dim3((N + 63) / 64, batch), dim3(64), 0, st, M, N, src, ssm, ssn, ssb, dst, sdb);
}

// workspace layout (floats), all [NR, width] row-major
struct RdgMlpWs { size_t z1, a1, z2, a2, z3, a3, u, v, du, d1, d2, total; };
static RdgMlpWs rdg_mlp_ws(int NR, int H, int NB, int HM) {
    RdgMlpWs w; size_t o = 0; const size_t n = (size_t)NR;
    w.z1 = o; o += n * H;  w.a1 = o; o += n * H;
    w.z2 = o; o += n * H;  w.a2 = o; o += n * H;
    w.z3 = o; o += n * (H / 2);  w.a3 = o; o += n * (H / 2);
    w.u = o; o += n * NB * HM;  w.v = o; o += n * NB * HM;
    w.du = o; o += n * NB * HM;
    w.d1 = o; o += n * H;  w.d2 = o; o += n * H;
    w.total = o;
    return w;
}

extern "C" {

size_t rdg_mlp_ws_bytes(int32_t NR, int32_t H, int32_t NB) { return rdg_mlp_ws(NR, H, NB, H / 4).total * sizeof(float) + 256; }

// x [NR,D0]; timenet W0 [H,D0] b0 [H], W1 [H,H] b1 [H], W2 [H/2,H] b2 [H/2]; heads hw1 [NB,H/4,H/2] hb1 [NB,H/4],
// hw2 [NB,OUT,H/4] hb2 [NB,OUT]; out [NR,NB,OUT].  ws keeps the activations for backward.
int rdg_mlp_forward(int32_t NR, int32_t D0, int32_t H, int32_t NB, int32_t OUT, const float* x, const float* W0,
                    const float* b0, const float* W1, const float* b1, const float* W2, const float* b2,
                    const float* hw1, const float* hb1, const float* hw2, const float* hb2, void* ws, float* out,
                    void* stream) {
    if (NR <= 0) return 0;
    if (H % 4 || H < 8) return rdg_set_error("mlp: width must be a multiple of 4");
    hipStream_t st = (hipStream_t)stream;
    const int H2 = H / 2, HM = H / 4;
    const RdgMlpWs w = rdg_mlp_ws(NR, H, NB, HM);
    float* f = (float*)ws;
    rdg_stage_begin(RDG_STAGE_MLP_FWD, st);
    // z1 = x W0^T + b0
    rdg_gemm(st, 1, NR, H, D0, x, D0, 1, 0, W0, 1, D0, 0, f + w.z1, H, 1, 0, b0, 0, 1, nullptr, f + w.a1);
    rdg_gemm(st, 1, NR, H, H, f + w.a1, H, 1, 0, W1, 1, H, 0, f + w.z2, H, 1, 0, b1, 0, 1, nullptr, f + w.a2);
    rdg_gemm(st, 1, NR, H2, H, f + w.a2, H, 1, 0, W2, 1, H, 0, f + w.z3, H2, 1, 0, b2, 0, 1, nullptr, f + w.a3);
    // all heads' first layer as one product: u [NR, NB*HM] = a3 . hw1_flat^T + hb1_flat
    rdg_gemm(st, 1, NR, NB * HM, H2, f + w.a3, H2, 1, 0, hw1, 1, H2, 0, f + w.u, (long long)NB * HM, 1, 0, hb1, 0, 1,
             nullptr, f + w.v);
    // second layer, batched over heads: out[:, h, :] = v[:, h, :] . hw2[h]^T + hb2[h]
    rdg_gemm(st, NB, NR, OUT, HM, f + w.v, (long long)NB * HM, 1, HM, hw2, 1, HM, (long long)OUT * HM, out,
             (long long)NB * OUT, 1, OUT, hb2, OUT, 0, nullptr, nullptr);
    rdg_stage_end(RDG_STAGE_MLP_FWD, st);
    return rdg_check_hip(hipGetLastError(), "mlp_fwd launch");
}

int rdg_mlp_backward(int32_t NR, int32_t D0, int32_t H, int32_t NB, int32_t OUT, const float* x, const float* W1,
                     const float* W2, const float* hw1, const float* hw2, void* ws, const float* g_out, float* dW0,
                     float* db0, float* dW1, float* db1, float* dW2, float* db2, float* dhw1, float* dhb1, float* dhw2,
                     float* dhb2, void* stream) {
    if (NR <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int H2 = H / 2, HM = H / 4;
    const long long NH = (long long)NB * HM, NO = (long long)NB * OUT;
    const RdgMlpWs w = rdg_mlp_ws(NR, H, NB, HM);
    float* f = (float*)ws;
    rdg_stage_begin(RDG_STAGE_MLP_BWD, st);
    // Every layer: (weight gradient, input gradient) as ONE launch -- both read the same dZ, write disjoint buffers.
    // heads, layer 2:  dhw2[h] = g_h^T v_h [OUT,HM];  dhb2[h] = colsum g_h;  du = (g_h hw2[h]) * gelu'(u)
    {
        const RdgGemmJob a = rdg_gemm_job(NB, OUT, HM, NR, g_out, 1, NO, OUT, f + w.v, NH, 1, HM, dhw2, HM, 1,
                                          (long long)OUT * HM, nullptr, 0, 0, nullptr, nullptr, dhb2, OUT);
        const RdgGemmJob b = rdg_gemm_job(NB, NR, HM, OUT, g_out, NO, 1, OUT, hw2, HM, 1, (long long)OUT * HM, f + w.du,
                                          NH, 1, HM, nullptr, 0, 2, f + w.u, nullptr);
        rdg_gemm_run(st, a, &b);
    }
    // heads, layer 1:  dhw1_flat = du^T a3 [NB*HM, H2];  dhb1 = colsum du;  dz3 = (du hw1_flat) * gelu'(z3)
    {
        const RdgGemmJob a = rdg_gemm_job(1, (int)NH, H2, NR, f + w.du, 1, NH, 0, f + w.a3, H2, 1, 0, dhw1, H2, 1, 0,
                                          nullptr, 0, 0, nullptr, nullptr, dhb1, 0);
        const RdgGemmJob b = rdg_gemm_job(1, NR, H2, (int)NH, f + w.du, NH, 1, 0, hw1, H2, 1, 0, f + w.d1, H2, 1, 0,
                                          nullptr, 0, 2, f + w.z3, nullptr);
        rdg_gemm_run(st, a, &b);
    }
    // timenet layer 3 (H -> H2):  dW2 = dz3^T a2;  db2;  dz2 = (dz3 W2) * gelu'(z2)
    {
        const RdgGemmJob a = rdg_gemm_job(1, H2, H, NR, f + w.d1, 1, H2, 0, f + w.a2, H, 1, 0, dW2, H, 1, 0, nullptr, 0, 0,
                                          nullptr, nullptr, db2, 0);
        const RdgGemmJob b = rdg_gemm_job(1, NR, H, H2, f + w.d1, H2, 1, 0, W2, H, 1, 0, f + w.d2, H, 1, 0, nullptr, 0, 2,
                                          f + w.z2, nullptr);
        rdg_gemm_run(st, a, &b);
    }
    // layer 2 (H -> H):  dW1 = dz2^T a1;  db1;  dz1 = (dz2 W1) * gelu'(z1)
    {
        const RdgGemmJob a = rdg_gemm_job(1, H, H, NR, f + w.d2, 1, H, 0, f + w.a1, H, 1, 0, dW1, H, 1, 0, nullptr, 0, 0,
                                          nullptr, nullptr, db1, 0);
        const RdgGemmJob b = rdg_gemm_job(1, NR, H, H, f + w.d2, H, 1, 0, W1, H, 1, 0, f + w.d1, H, 1, 0, nullptr, 0, 2,
                                          f + w.z1, nullptr);
        rdg_gemm_run(st, a, &b);
    }
    // layer 1 (D0 -> H):  dW0 = dz1^T x;  db0
    rdg_gemm(st, 1, H, D0, NR, f + w.d1, 1, H, 0, x, D0, 1, 0, dW0, D0, 1, 0, nullptr, 0, 0, nullptr, nullptr, db0, 0);
    rdg_stage_end(RDG_STAGE_MLP_BWD, st);
    return rdg_check_hip(hipGetLastError(), "mlp_bwd launch");
}

}  // extern "C"
