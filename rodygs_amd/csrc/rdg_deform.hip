// rdg_deform.hip -- per-Gaussian time deformation and the fused Adam step (§8f row 2).
//
// Reference: DynRoDyGS.get_gaussian_deformation, /root/reference/src/model/rodygs_dynamic.py:122-138, which
// materialises table[time_ind] as a [P,16,7] tensor (448 MB at P = 1 M) and then runs a bmm.  Here the
// difference table  diff[u] = B(t) - B_table[u]  ([Tu,16,7], 45 KB at Tu = 100) is built once per workgroup in
// LDS (row stride 113 floats -> random birth indices spread over the banks) and every Gaussian does its
// 16x7 contraction against its own row: 64 B of coefficients in, 28 B out -- the algorithmic minimum.
//
// Backward: dcoeff needs the same rows; the two small reductions dB(t) = sum_p c_p^T g_p and
// dB_table[u] = -sum_{birth(p)=u} c_p^T g_p are accumulated with LDS float atomics into a per-workgroup
// [Tu,16,7] table (ds_add_f32, random rows => few bank conflicts) and flushed once per workgroup with
// contiguous global float atomics.
#include "rdg_common.h"

#define RDG_DEF_K 7
#define RDG_DEF_MAXB 16

template <bool USE_LDS>
__global__ void __launch_bounds__(1024)
rdg_deform_fwd_kernel(int P, int B, int Tu, const float* __restrict__ coeff, const long long* __restrict__ time_ind,
                      const float* __restrict__ basis_t, const float* __restrict__ table, float scale,
                      float* __restrict__ out_xyz, float* __restrict__ out_rot) {
    extern __shared__ float smem[];
    const int row = B * RDG_DEF_K;       // 112
    const int stride = row + 1;          // 113: de-conflicts random rows
    float* sDiff = smem;                 // [Tu_eff][stride]
    float* sBt = smem;                   // non-LDS path: only basis_t
    const int Tu_eff = table ? Tu : 1;
    if (USE_LDS) {
        for (int k = threadIdx.x; k < Tu_eff * row; k += blockDim.x) {
            const int u = k / row, c = k - u * row;
            sDiff[u * stride + c] = basis_t[c] - (table ? table[(size_t)u * row + c] : 0.0f);
        }
    } else {
        for (int k = threadIdx.x; k < row; k += blockDim.x) sBt[k] = basis_t[k];
    }
    __syncthreads();
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const int u = table ? (int)time_ind[p] : 0;
        float acc[RDG_DEF_K];
#pragma unroll
        for (int k = 0; k < RDG_DEF_K; ++k) acc[k] = 0.0f;
        const float* c = coeff + (size_t)p * B;
        for (int b = 0; b < B; ++b) {
            const float cb = c[b];
#pragma unroll
            for (int k = 0; k < RDG_DEF_K; ++k) {
                float dv;
                if (USE_LDS) dv = sDiff[u * stride + b * RDG_DEF_K + k];
                else dv = sBt[b * RDG_DEF_K + k] - (table ? table[(size_t)u * row + b * RDG_DEF_K + k] : 0.0f);
                acc[k] += cb * dv;
            }
        }
        out_xyz[3 * p + 0] = acc[0] * scale; out_xyz[3 * p + 1] = acc[1] * scale; out_xyz[3 * p + 2] = acc[2] * scale;
        out_rot[4 * p + 0] = acc[3]; out_rot[4 * p + 1] = acc[4]; out_rot[4 * p + 2] = acc[5]; out_rot[4 * p + 3] = acc[6];
    }
}

template <bool USE_LDS>
__global__ void __launch_bounds__(1024)
rdg_deform_bwd_kernel(int P, int B, int Tu, const float* __restrict__ coeff, const long long* __restrict__ time_ind,
                      const float* __restrict__ basis_t, const float* __restrict__ table, float scale,
                      const float* __restrict__ g_xyz, const float* __restrict__ g_rot, float* __restrict__ d_coeff,
                      float* __restrict__ d_basis_t, float* __restrict__ d_table) {
    extern __shared__ float smem[];
    const int row = B * RDG_DEF_K;
    const int stride = row + 1;
    const int Tu_eff = table ? Tu : 1;
    float* sDiff = smem;                                  // [Tu_eff][stride]      (LDS path)
    float* sAcc = smem + (size_t)Tu_eff * stride;         // [Tu_eff][stride]      (LDS path)
    float* sBt = smem;                                    // [row]                 (global path)
    if (USE_LDS) {
        for (int k = threadIdx.x; k < Tu_eff * row; k += blockDim.x) {
            const int u = k / row, c = k - u * row;
            sDiff[u * stride + c] = basis_t[c] - (table ? table[(size_t)u * row + c] : 0.0f);
            sAcc[u * stride + c] = 0.0f;
        }
    } else {
        for (int k = threadIdx.x; k < row; k += blockDim.x) sBt[k] = basis_t[k];
    }
    __syncthreads();
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const int u = table ? (int)time_ind[p] : 0;
        float g[RDG_DEF_K];
        g[0] = g_xyz[3 * p + 0] * scale; g[1] = g_xyz[3 * p + 1] * scale; g[2] = g_xyz[3 * p + 2] * scale;
        g[3] = g_rot[4 * p + 0]; g[4] = g_rot[4 * p + 1]; g[5] = g_rot[4 * p + 2]; g[6] = g_rot[4 * p + 3];
        const float* c = coeff + (size_t)p * B;
        float* dc = d_coeff + (size_t)p * B;
        for (int b = 0; b < B; ++b) {
            const float cb = c[b];
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < RDG_DEF_K; ++k) {
                float dv;
                if (USE_LDS) dv = sDiff[u * stride + b * RDG_DEF_K + k];
                else dv = sBt[b * RDG_DEF_K + k] - (table ? table[(size_t)u * row + b * RDG_DEF_K + k] : 0.0f);
                s += g[k] * dv;
                const float v = cb * g[k];
                if (USE_LDS) {
                    atomicAdd(&sAcc[u * stride + b * RDG_DEF_K + k], v);
                } else {
                    atomicAdd(&d_basis_t[b * RDG_DEF_K + k], v);
                    if (table) atomicAdd(&d_table[(size_t)u * row + b * RDG_DEF_K + k], -v);
                }
            }
            dc[b] = s;
        }
    }
    if (USE_LDS) {
        __syncthreads();
        // dB_table[u] -= acc[u];  dB(t) += sum_u acc[u]
        for (int k = threadIdx.x; k < Tu_eff * row; k += blockDim.x) {
            const int u = k / row, c = k - u * row;
            const float v = sAcc[u * stride + c];
            if (table && v != 0.0f) atomicAdd(&d_table[(size_t)u * row + c], -v);
        }
        for (int c = threadIdx.x; c < row; c += blockDim.x) {
            float s = 0.0f;
            for (int u = 0; u < Tu_eff; ++u) s += sAcc[u * stride + c];
            if (s != 0.0f) atomicAdd(&d_basis_t[c], s);
        }
    }
}

__global__ void __launch_bounds__(256)
rdg_adam_kernel(long long n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                float* __restrict__ v, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt) {
    const long long n4 = n >> 2;
    const float step = lr / bc1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
#define RDG_ADAM1(c)                                                   \
        mm.c = b1 * mm.c + (1.0f - b1) * gg.c;                         \
        vv.c = b2 * vv.c + (1.0f - b2) * gg.c * gg.c;                  \
        pp.c -= step * (mm.c / (sqrtf(vv.c) / bc2_sqrt + eps));
        RDG_ADAM1(x) RDG_ADAM1(y) RDG_ADAM1(z) RDG_ADAM1(w)
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    // tail
    const long long t0 = n4 << 2;
    const long long i = t0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= step * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
}

extern "C" {

int rdg_deform_forward(int32_t P, int32_t B, int32_t Tu, const float* coeff, const int64_t* time_ind,
                       const float* basis_t, const float* table, float spatial_scale, float* out_xyz, float* out_rot,
                       void* stream) {
    if (B <= 0 || B > RDG_DEF_MAXB * 4) return rdg_set_error("deform: bad basis count %d", B);
    if (P <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int Tu_eff = table ? Tu : 1;
    const size_t lds = (size_t)Tu_eff * (B * RDG_DEF_K + 1) * sizeof(float);
    const int threads = 1024;
    int blocks = (P + threads - 1) / threads;
    if (blocks > 512) blocks = 512;
    rdg_stage_begin(RDG_STAGE_DEFORM_FWD, st);
    if (lds <= 64 * 1024) {
        hipLaunchKernelGGL(rdg_deform_fwd_kernel<true>, dim3(blocks), dim3(threads), lds, st, P, B, Tu, coeff,
                           (const long long*)time_ind, basis_t, table, spatial_scale, out_xyz, out_rot);
    } else {
        hipLaunchKernelGGL(rdg_deform_fwd_kernel<false>, dim3(blocks), dim3(threads), (size_t)B * RDG_DEF_K * 4, st, P,
                           B, Tu, coeff, (const long long*)time_ind, basis_t, table, spatial_scale, out_xyz, out_rot);
    }
    rdg_stage_end(RDG_STAGE_DEFORM_FWD, st);
    return rdg_check_hip(hipGetLastError(), "deform_fwd launch");
}

int rdg_deform_backward(int32_t P, int32_t B, int32_t Tu, const float* coeff, const int64_t* time_ind,
                        const float* basis_t, const float* table, float spatial_scale, const float* g_xyz,
                        const float* g_rot, float* d_coeff, float* d_basis_t, float* d_table, void* stream) {
    if (B <= 0 || B > RDG_DEF_MAXB * 4) return rdg_set_error("deform: bad basis count %d", B);
    hipStream_t st = (hipStream_t)stream;
    const int row = B * RDG_DEF_K;
    rdg_stage_begin(RDG_STAGE_DEFORM_BWD, st);
    hipError_t e = hipMemsetAsync(d_basis_t, 0, (size_t)row * 4, st);
    if (e == hipSuccess && table && d_table) e = hipMemsetAsync(d_table, 0, (size_t)Tu * row * 4, st);
    if (e != hipSuccess) return rdg_check_hip(e, "deform_bwd memset");
    if (P > 0) {
        const int Tu_eff = table ? Tu : 1;
        const size_t lds = 2 * (size_t)Tu_eff * (row + 1) * sizeof(float);
        const int threads = 1024;
        int blocks = (P + threads - 1) / threads;
        if (blocks > 256) blocks = 256;
        if (lds <= 128 * 1024) {
            if (lds > 64 * 1024) {
                hipError_t ea = hipFuncSetAttribute((const void*)rdg_deform_bwd_kernel<true>,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (ea != hipSuccess) return rdg_check_hip(ea, "deform_bwd LDS attribute");
            }
            hipLaunchKernelGGL(rdg_deform_bwd_kernel<true>, dim3(blocks), dim3(threads), lds, st, P, B, Tu, coeff,
                               (const long long*)time_ind, basis_t, table, spatial_scale, g_xyz, g_rot, d_coeff,
                               d_basis_t, d_table);
        } else {
            hipLaunchKernelGGL(rdg_deform_bwd_kernel<false>, dim3(blocks), dim3(threads), (size_t)row * 4, st, P, B, Tu,
                               coeff, (const long long*)time_ind, basis_t, table, spatial_scale, g_xyz, g_rot, d_coeff,
                               d_basis_t, d_table);
        }
    }
    rdg_stage_end(RDG_STAGE_DEFORM_BWD, st);
    return rdg_check_hip(hipGetLastError(), "deform_bwd launch");
}

int rdg_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                  float beta2, float eps, int32_t step, void* stream) {
    if (n <= 0) return 0;
    if (step < 1) return rdg_set_error("adam: step must be >= 1");
    hipStream_t st = (hipStream_t)stream;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    long long blocks = ((n >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    rdg_stage_begin(RDG_STAGE_ADAM, st);
    hipLaunchKernelGGL(rdg_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (long long)n, param, grad, exp_avg,
                       exp_avg_sq, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2));
    rdg_stage_end(RDG_STAGE_ADAM, st);
    return rdg_check_hip(hipGetLastError(), "adam launch");
}

}  // extern "C"
