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
    hipError_t e = hipMemsetAsync(sums2, 0, 2 * sizeof(double), st);
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
