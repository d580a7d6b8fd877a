// rdg_motionreg.hip -- the two per-Gaussian motion regularisers of the dynamic sub-step in one pass each way
// (config 5, /root/reference/configs/train/train_kubric_mrig.yaml:186-232):
//   MotionL1Loss        (/root/reference/src/trainer/losses.py:363-371):  mean |c|                over [P,1,B]
//   MotionSparsityLoss  (/root/reference/src/trainer/losses.py:374-384):  mean |c| / (max_b |c| + 1e-7)
// Through the framework these are ~6 forward and ~10 backward elementwise / reduction launches over the 64-byte
// coefficient rows (1.2 ms per step at 1 M Gaussians).  Here: forward = one read of the rows -> two block-reduced sums;
// backward = one read of the rows, the gradient of  w1 * L1 + w2 * sparsity  ADDED to (or written over) dL/dcoeff.
// The maximum's gradient goes to the FIRST largest |c| of a row, as torch.max(dim) routes it.
#include "rdg_common.h"

#define RDG_MR_B 16

__device__ __forceinline__ void rdg_mr_load(const float* __restrict__ c, long long p, float a[RDG_MR_B]) {
    const float4* r = reinterpret_cast<const float4*>(c + p * RDG_MR_B);
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float4 t = r[q]; a[4 * q] = t.x; a[4 * q + 1] = t.y; a[4 * q + 2] = t.z; a[4 * q + 3] = t.w; }
}

__global__ void __launch_bounds__(256)
rdg_motion_reg_fwd_kernel(long long P, const float* __restrict__ coeff, double* __restrict__ sums) {
    float s1 = 0.0f, s2 = 0.0f;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        float c[RDG_MR_B];
        rdg_mr_load(coeff, p, c);
        float m = 0.0f, t = 0.0f;
#pragma unroll
        for (int b = 0; b < RDG_MR_B; ++b) { const float a = fabsf(c[b]); t += a; m = fmaxf(m, a); }
        s1 += t;
        s2 += t / (m + 1e-7f);
    }
    // per-workgroup partials in f64 (a few hundred workgroups x 2 atomics)
    __shared__ float w1[4], w2[4];
    const float r1 = rdg_wave_sum_all(s1), r2 = rdg_wave_sum_all(s2);
    if ((threadIdx.x & 63) == 0) { w1[threadIdx.x >> 6] = r1; w2[threadIdx.x >> 6] = r2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&sums[0], (double)w1[0] + (double)w1[1] + (double)w1[2] + (double)w1[3]);
        atomicAdd(&sums[1], (double)w2[0] + (double)w2[1] + (double)w2[2] + (double)w2[3]);
    }
}

template <bool ACC>
__global__ void __launch_bounds__(256)
rdg_motion_reg_bwd_kernel(long long P, const float* __restrict__ coeff, const float* __restrict__ g_loss, float k1,
                          float k2, float* __restrict__ d_coeff) {
    const float g = g_loss ? g_loss[0] : 1.0f;
    const float g1 = g * k1, g2 = g * k2;                   // k = weight / (P * B)
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        float c[RDG_MR_B];
        rdg_mr_load(coeff, p, c);
        float m = 0.0f, t = 0.0f;
        int km = 0;
#pragma unroll
        for (int b = 0; b < RDG_MR_B; ++b) {
            const float a = fabsf(c[b]);
            t += a;
            if (a > m) { m = a; km = b; }                   // first maximum
        }
        const float inv = 1.0f / (m + 1e-7f);
        const float dmax = -t * inv * inv;                  // d/dm of sum_b a_b / (m + eps)
        float o[RDG_MR_B];
#pragma unroll
        for (int b = 0; b < RDG_MR_B; ++b) {
            const float sg = c[b] > 0.0f ? 1.0f : (c[b] < 0.0f ? -1.0f : 0.0f);
            o[b] = sg * (g1 + g2 * (inv + (b == km ? dmax : 0.0f)));
        }
        float4* d = reinterpret_cast<float4*>(d_coeff + p * RDG_MR_B);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
            if (ACC) { const float4 e = d[q]; v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w; }
            d[q] = v;
        }
    }
}

extern "C" {

int rdg_motion_reg_forward(int64_t P, int32_t B, const float* coeff, double* sums2, void* stream) {
    if (B != RDG_MR_B) return rdg_set_error("motion_reg: B must be %d", RDG_MR_B);
    if (((uintptr_t)coeff) & 15) return rdg_set_error("motion_reg: 16-B alignment");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = rdg_zero_async(sums2, 2 * sizeof(double), st);
    if (e != hipSuccess) return rdg_check_hip(e, "motion_reg memset");
    if (P <= 0) return 0;
    long long nb = (P + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(rdg_motion_reg_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, st, (long long)P, coeff, sums2);
    return rdg_check_hip(hipGetLastError(), "motion_reg_fwd launch");
}

int rdg_motion_reg_backward(int64_t P, int32_t B, const float* coeff, const float* g_loss, float w_l1, float w_sparsity,
                            float* d_coeff, int32_t accumulate, void* stream) {
    if (B != RDG_MR_B) return rdg_set_error("motion_reg: B must be %d", RDG_MR_B);
    if ((((uintptr_t)coeff) | ((uintptr_t)d_coeff)) & 15) return rdg_set_error("motion_reg: 16-B alignment");
    if (P <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const float k1 = (float)((double)w_l1 / ((double)P * B)), k2 = (float)((double)w_sparsity / ((double)P * B));
    long long nb = (P + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (accumulate)
        hipLaunchKernelGGL(rdg_motion_reg_bwd_kernel<true>, dim3((unsigned)nb), dim3(256), 0, st, (long long)P, coeff, g_loss,
                           k1, k2, d_coeff);
    else
        hipLaunchKernelGGL(rdg_motion_reg_bwd_kernel<false>, dim3((unsigned)nb), dim3(256), 0, st, (long long)P, coeff, g_loss,
                           k1, k2, d_coeff);
    return rdg_check_hip(hipGetLastError(), "motion_reg_bwd launch");
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// MotionBasisRegularizaiton (/root/reference/src/trainer/losses.py:386-525): smoothness of the motion table [Tu,B,7] over
// the sorted birth times -- k-th forward differences (k = degree + 1) of the translations and of the rotation MATRICES
// (the reference's recursion differences the matrices by plain subtraction), loss =
//   mean_{t,b} w_b ||d^k transl||_2  +  mean_{t,b} w_b || I - d^k R(q) ||_F .
// Through the framework this tiny tensor costs ~60 launches forward and ~100 backward (1.3 ms of host time per step,
// more than the rasterizer's); here: R(q) -> difference terms (value + gradient, float atomics into two tiny buffers)
// -> quaternion chain; the gradient is produced together with the value and only scaled in backward.
// ---------------------------------------------------------------------------------------------------------
struct RdgBasisRegArgs { int Tu, kt, kr; float w[RDG_MR_B]; float inv_nt, inv_nr; };

__device__ __forceinline__ float rdg_diff_coef(int k, int j) {          // (-1)^(k-j) C(k,j), k <= 3
    const int C[4][4] = {{1, 0, 0, 0}, {1, 1, 0, 0}, {1, 2, 1, 0}, {1, 3, 3, 1}};
    return (((k - j) & 1) ? -1.0f : 1.0f) * (float)C[k][j];
}

__global__ void rdg_basis_R_kernel(int n, const float* __restrict__ table, float* __restrict__ R_ws) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float* q = table + (size_t)e * 7 + 3;
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float s = 2.0f / (r * r + i * i + j * j + k * k);
    float* R = R_ws + (size_t)e * 9;
    R[0] = 1.f - s * (j * j + k * k); R[1] = s * (i * j - k * r); R[2] = s * (i * k + j * r);
    R[3] = s * (i * j + k * r); R[4] = 1.f - s * (i * i + k * k); R[5] = s * (j * k - i * r);
    R[6] = s * (i * k - j * r); R[7] = s * (j * k + i * r); R[8] = 1.f - s * (i * i + j * j);
}

__global__ void __launch_bounds__(256)
rdg_basis_terms_kernel(RdgBasisRegArgs A, const float* __restrict__ table, const float* __restrict__ R_ws,
                       float* __restrict__ d_table, float* __restrict__ dR_ws, double* __restrict__ loss) {
    float acc = 0.0f;
    const int n = A.Tu * RDG_MR_B;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int t = e / RDG_MR_B, b = e - t * RDG_MR_B;
        if (A.kt > 0 && t + A.kt < A.Tu) {
            float d[3] = {0.f, 0.f, 0.f};
            for (int j = 0; j <= A.kt; ++j) {
                const float cf = rdg_diff_coef(A.kt, j);
                const float* x = table + ((size_t)(t + j) * RDG_MR_B + b) * 7;
                d[0] += cf * x[0]; d[1] += cf * x[1]; d[2] += cf * x[2];
            }
            const float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            acc += nrm * A.w[b] * A.inv_nt;
            if (nrm > 0.0f) {
                const float gs = A.w[b] * A.inv_nt / nrm;
                for (int j = 0; j <= A.kt; ++j) {
                    const float cf = rdg_diff_coef(A.kt, j) * gs;
                    float* o = d_table + ((size_t)(t + j) * RDG_MR_B + b) * 7;
                    atomicAdd(o + 0, cf * d[0]); atomicAdd(o + 1, cf * d[1]); atomicAdd(o + 2, cf * d[2]);
                }
            }
        }
        if (A.kr > 0 && t + A.kr < A.Tu) {
            float E[9];
#pragma unroll
            for (int a = 0; a < 9; ++a) E[a] = (a == 0 || a == 4 || a == 8) ? 1.0f : 0.0f;
            for (int j = 0; j <= A.kr; ++j) {
                const float cf = rdg_diff_coef(A.kr, j);
                const float* R = R_ws + ((size_t)(t + j) * RDG_MR_B + b) * 9;
#pragma unroll
                for (int a = 0; a < 9; ++a) E[a] -= cf * R[a];
            }
            float n2 = 0.0f;
#pragma unroll
            for (int a = 0; a < 9; ++a) n2 += E[a] * E[a];
            const float nrm = sqrtf(n2);
            acc += nrm * A.w[b] * A.inv_nr;
            if (nrm > 0.0f) {
                const float gs = -A.w[b] * A.inv_nr / nrm;               // d||I - D|| / dD = -(I - D) / ||.||
                for (int j = 0; j <= A.kr; ++j) {
                    const float cf = rdg_diff_coef(A.kr, j) * gs;
                    float* o = dR_ws + ((size_t)(t + j) * RDG_MR_B + b) * 9;
#pragma unroll
                    for (int a = 0; a < 9; ++a) atomicAdd(o + a, cf * E[a]);
                }
            }
        }
    }
    __shared__ float ws[4];
    const float r = rdg_wave_sum_all(acc);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (double)ws[0] + (double)ws[1] + (double)ws[2] + (double)ws[3]);
}

__global__ void rdg_basis_dq_kernel(int n, const float* __restrict__ table, const float* __restrict__ dR_ws,
                                    float* __restrict__ d_table) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float* q = table + (size_t)e * 7 + 3;
    const float* dR = dR_ws + (size_t)e * 9;
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float n2 = r * r + i * i + j * j + k * k;
    const float s = 2.0f / n2;
    // R = I + s M(q):  dL/dq = s * sum dR_ab dM_ab/dq + (sum dR_ab M_ab) * ds/dq,  ds/dq = -s * 2 q / |q|^2
    const float M[9] = {-(j * j + k * k), i * j - k * r, i * k + j * r, i * j + k * r, -(i * i + k * k), j * k - i * r,
                        i * k - j * r, j * k + i * r, -(i * i + j * j)};
    float dotM = 0.f;
#pragma unroll
    for (int a = 0; a < 9; ++a) dotM += dR[a] * M[a];
    const float dsc = -2.0f * s / n2;
    const float gr = -k * dR[1] + j * dR[2] + k * dR[3] - i * dR[5] - j * dR[6] + i * dR[7];
    const float gi = j * dR[1] + k * dR[2] + j * dR[3] - 2.f * i * dR[4] - r * dR[5] + k * dR[6] + r * dR[7] - 2.f * i * dR[8];
    const float gj = -2.f * j * dR[0] + i * dR[1] + r * dR[2] + i * dR[3] + k * dR[5] - r * dR[6] + k * dR[7] - 2.f * j * dR[8];
    const float gk = -2.f * k * dR[0] - r * dR[1] + i * dR[2] + r * dR[3] - 2.f * k * dR[4] + j * dR[5] + i * dR[6] + j * dR[7];
    float* o = d_table + (size_t)e * 7 + 3;
    o[0] = s * gr + dotM * dsc * r; o[1] = s * gi + dotM * dsc * i; o[2] = s * gj + dotM * dsc * j; o[3] = s * gk + dotM * dsc * k;
}

extern "C" {

size_t rdg_basis_reg_ws_bytes(int32_t Tu) { return (size_t)(Tu > 0 ? Tu : 1) * RDG_MR_B * 9 * 4 * 2 + 256; }

int rdg_basis_reg(int32_t Tu, int32_t B, int32_t transl_degree, int32_t rot_degree, const float* w_host,
                  const float* table, void* ws, double* loss, float* d_table, void* stream) {
    if (B != RDG_MR_B) return rdg_set_error("basis_reg: B must be %d", RDG_MR_B);
    if (transl_degree > 2 || rot_degree > 2) return rdg_set_error("basis_reg: degrees above 2 are not supported");
    if (Tu < 1) return rdg_set_error("basis_reg: empty table");
    hipStream_t st = (hipStream_t)stream;
    RdgBasisRegArgs A;
    A.Tu = Tu; A.kt = transl_degree < 0 ? 0 : transl_degree + 1; A.kr = rot_degree < 0 ? 0 : rot_degree + 1;
    for (int b = 0; b < RDG_MR_B; ++b) A.w[b] = w_host[b];
    A.inv_nt = (A.kt && Tu > A.kt) ? 1.0f / (float)((Tu - A.kt) * RDG_MR_B) : 0.0f;
    A.inv_nr = (A.kr && Tu > A.kr) ? 1.0f / (float)((Tu - A.kr) * RDG_MR_B) : 0.0f;
    const int n = Tu * RDG_MR_B;
    float* R_ws = (float*)ws;
    float* dR_ws = R_ws + (size_t)n * 9;
    hipError_t e = rdg_zero_async(dR_ws, (size_t)n * 9 * 4, st);
    if (e == hipSuccess) e = rdg_zero_async(d_table, (size_t)n * 7 * 4, st);
    if (e == hipSuccess) e = rdg_zero_async(loss, sizeof(double), st);
    if (e != hipSuccess) return rdg_check_hip(e, "basis_reg memset");
    hipLaunchKernelGGL(rdg_basis_R_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, table, R_ws);
    int nb = (n + 255) / 256;
    if (nb > 64) nb = 64;
    hipLaunchKernelGGL(rdg_basis_terms_kernel, dim3(nb), dim3(256), 0, st, A, table, (const float*)R_ws, d_table, dR_ws, loss);
    hipLaunchKernelGGL(rdg_basis_dq_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, table, (const float*)dR_ws, d_table);
    return rdg_check_hip(hipGetLastError(), "basis_reg launch");
}

}  // extern "C"
