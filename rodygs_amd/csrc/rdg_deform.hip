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
#include <stdlib.h>
// workgroups of the single-camera getter kernels (1024 threads, the difference table in LDS: two fit a CU);
// RDG_GETTER_GRID overrides for probes
static int rdg_getter_grid_cap() {
    static int cap = -1;
    if (cap < 0) { const char* ev = getenv("RDG_GETTER_GRID"); cap = ev ? atoi(ev) : 256; if (cap < 1) cap = 256; }
    return cap;
}

#define RDG_DEF_K 7
#define RDG_DEF_MAXB 16
#define RDG_DC_STRIDE 116   // LDS row stride of the [Tu][112] difference table (16-B aligned rows, 116 = 20 mod 32)

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

template <bool USE_LDS, bool ACC>
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
    float* sAcc = smem + (size_t)Tu_eff * stride;         // [Tu_eff][stride]      (LDS path, ACC only)
    float* sBt = smem;                                    // [row]                 (global path)
    if (USE_LDS) {
        for (int k = threadIdx.x; k < Tu_eff * row; k += blockDim.x) {
            const int u = k / row, c = k - u * row;
            sDiff[u * stride + c] = basis_t[c] - (table ? table[(size_t)u * row + c] : 0.0f);
            if (ACC) sAcc[u * stride + c] = 0.0f;
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
                if (!ACC) {
                } else if (USE_LDS) {
                    atomicAdd(&sAcc[u * stride + b * RDG_DEF_K + k], v);
                } else {
                    atomicAdd(&d_basis_t[b * RDG_DEF_K + k], v);
                    if (table) atomicAdd(&d_table[(size_t)u * row + b * RDG_DEF_K + k], -v);
                }
            }
            dc[b] = s;
        }
    }
    if (USE_LDS && ACC) {
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

// dL/dcoeff for B = 16: d_coeff[p][b] = sum_k g_p[k] * (B(t) - table[birth(p)])[b][k].  Natural (coalesced) order;
// the difference table lives in LDS with a row stride of 116 floats (16-B aligned rows, 116 = 20 mod 32: the
// b128 reads of lanes holding different birth indices spread over all banks), read as 28 float4 per Gaussian
// instead of 112 dwords, and the 64-B result row leaves as 4 float4 stores.
__global__ void __launch_bounds__(1024)
rdg_deform_dcoeff16_kernel(int P, int Tu, const long long* __restrict__ time_ind, const float* __restrict__ basis_t,
                           const float* __restrict__ table, float scale, const float* __restrict__ g_xyz,
                           const float* __restrict__ g_rot, float* __restrict__ d_coeff,
                           const int* __restrict__ inv_order, float4* __restrict__ gs) {
    extern __shared__ __attribute__((aligned(16))) float smem_dc[];
    const int Tu_eff = table ? Tu : 1;
    for (int k = threadIdx.x; k < Tu_eff * 112; k += blockDim.x) {
        const int u = k / 112, c = k - u * 112;
        smem_dc[u * RDG_DC_STRIDE + c] = basis_t[c] - (table ? table[(size_t)u * 112 + c] : 0.0f);
    }
    __syncthreads();
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const int u = table ? (int)time_ind[p] : 0;
        float g[RDG_DEF_K];
        g[0] = g_xyz[3 * p + 0] * scale; g[1] = g_xyz[3 * p + 1] * scale; g[2] = g_xyz[3 * p + 2] * scale;
        const float4 gr = reinterpret_cast<const float4*>(g_rot)[p];
        g[3] = gr.x; g[4] = gr.y; g[5] = gr.z; g[6] = gr.w;
        if (gs) {
            // compact copy of (scaled gradient, birth index) at the Gaussian's position in birth-sorted order: the
            // dB accumulation then streams 32 contiguous bytes per Gaussian instead of gathering 12 + 16 + 8 bytes
            // out of three 64-B sectors (PMC: that kernel fetched 4.8x its algorithmic bytes)
            const size_t sidx = (size_t)inv_order[p];
            gs[2 * sidx] = make_float4(g[0], g[1], g[2], g[3]);
            gs[2 * sidx + 1] = make_float4(g[4], g[5], g[6], __int_as_float(u));
        }
        const float4* r4 = reinterpret_cast<const float4*>(smem_dc + u * RDG_DC_STRIDE);
        float sacc[16];
#pragma unroll
        for (int b = 0; b < 16; ++b) sacc[b] = 0.0f;
#pragma unroll
        for (int j = 0; j < 28; ++j) {
            const float4 v = r4[j];
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = 4 * j + c;          // compile-time after unrolling: e = b * 7 + k
                sacc[e / 7] += g[e % 7] * vv[c];
            }
        }
        float4* dc = reinterpret_cast<float4*>(d_coeff + (size_t)p * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) dc[q] = make_float4(sacc[4 * q], sacc[4 * q + 1], sacc[4 * q + 2], sacc[4 * q + 3]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Fused "dynamic getter": deformation + activations of the dynamic Gaussians in one pass each way, i.e. what
// /root/reference/src/trainer/rodygs.py:68-113 (get_GS_properties) assembles from get_gaussian_deformation
// (rodygs_dynamic.py:122-138) and the model getters (rodygs_static.py:82-105):
//   means3D = xyz + scale * (c . dB)[0:3]        scales = exp(scaling)
//   rots    = normalize(rotation) + (c . dB)[3:7]   opac = sigmoid(opacity)         dB = B(t) - table[birth]
// Separate kernels write the 28-B deformation per Gaussian only for the next kernel to read it back (and the same
// for its gradient on the way back); fused, a Gaussian's parameters are read once and its activated values written
// once.  B = 16; the difference table sits in LDS exactly as in rdg_deform_dcoeff16_kernel.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rdg_diff16_to_lds(float* sm, int Tu, const float* __restrict__ bases) {
    const float* bt = bases + (size_t)Tu * 112;          // packed bases: Tu table rows, then B(t)
    // four elements per thread and trip, every load issued before the first LDS store: written as the plain loop the
    // compiler waited for each pair of loads in turn (11 dependent L2 round trips at Tu = 100 before the kernel began)
    const int n = Tu * 112;
    for (int k0 = threadIdx.x; k0 < n; k0 += 4 * blockDim.x) {
        float a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = min(k0 + q * (int)blockDim.x, n - 1);
            const int u = k / 112, c = k - u * 112;
            a[q] = bt[c]; b[q] = bases[(size_t)u * 112 + c];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = k0 + q * (int)blockDim.x;
            if (k < n) { const int u = k / 112, c = k - u * 112; sm[u * RDG_DC_STRIDE + c] = a[q] - b[q]; }
        }
    }
}

// The coefficient-gradient rows (64 B per Gaussian) leave the backward kernel through a wave-private LDS stage: written
// by a lane directly, one instruction stores 16 B of each of 64 different rows (256 write requests of 16 B per wave);
// through the stage every store instruction moves 1 KB contiguous (-11 us at P = 1 M).  The forward's coefficient READS
// gained nothing from the same stage (+4 us: the lines of a row stay in the L1 between its four loads) and go direct.
// Stage row stride 20 floats: 16-B aligned, and 8 consecutive rows cover all 32 banks.
#define RDG_DG_STAGE 20
#define RDG_DG_STAGE_BYTES (64 * RDG_DG_STAGE * 4)                  // per wave
__device__ __forceinline__ void rdg_stage_rows16_out(float* __restrict__ dst, long long first_row, long long n_rows,
                                                     float* stage, int lane, const float4 in[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(stage + lane * RDG_DG_STAGE + 4 * i) = in[i];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long r = first_row + 16 * i + (lane >> 2);
        const float4 v = *reinterpret_cast<const float4*>(stage + (16 * i + (lane >> 2)) * RDG_DG_STAGE + 4 * (lane & 3));
        if (r < n_rows) reinterpret_cast<float4*>(dst)[r * 4 + (lane & 3)] = v;
    }
    __builtin_amdgcn_wave_barrier();
}

// Single-camera forward (the 1-GPU train step): the difference table sits in LDS, so a Gaussian costs 28 row reads and
// nothing else.  Same operations in the same order as rdg_dyn_getter_views_fwd_kernel: the results are bit-identical.
__global__ void __launch_bounds__(1024)
rdg_dyn_getter_fwd_kernel(int P, int Tu, const float* __restrict__ coeff, const long long* __restrict__ time_ind,
                          const float* __restrict__ bases, float scale, const float* __restrict__ xyz,
                          const float* __restrict__ scaling, const float* __restrict__ rotation,
                          const float* __restrict__ opacity, float* __restrict__ means3D, float* __restrict__ scales,
                          float* __restrict__ rots, float* __restrict__ opac) {
    extern __shared__ __attribute__((aligned(16))) float smem_dg[];
    rdg_diff16_to_lds(smem_dg, Tu, bases);
    __syncthreads();
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        // every global load of the Gaussian before the first use (one round trip to memory per Gaussian: written in
        // program order "load, use, store, load ..." the compiler kept that order across the stores)
        const int u = (int)time_ind[p];
        const float4* c4 = reinterpret_cast<const float4*>(coeff + (size_t)p * 16);
        const float4 c0 = c4[0], c1 = c4[1], c2 = c4[2], c3 = c4[3];
        float x0 = xyz[3 * p], x1 = xyz[3 * p + 1], x2 = xyz[3 * p + 2];
        float s0 = scaling[3 * p], s1 = scaling[3 * p + 1], s2 = scaling[3 * p + 2];
        float4 q = reinterpret_cast<const float4*>(rotation)[p];
        float op = opacity[p];
        asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w), "+v"(op), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(s0), "+v"(s1),
                     "+v"(s2));
        const float c[16] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w};
        const float4* r4 = reinterpret_cast<const float4*>(smem_dg + u * RDG_DC_STRIDE);
        float acc[RDG_DEF_K];
#pragma unroll
        for (int k = 0; k < RDG_DEF_K; ++k) acc[k] = 0.0f;
#pragma unroll
        for (int j = 0; j < 28; ++j) {
            const float4 v = r4[j];
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const int e = 4 * j + e4;          // compile-time: e = b * 7 + k
                acc[e % 7] = __fmaf_rn(c[e / 7], vv[e4], acc[e % 7]);   // written out: same bits as the views kernel
            }
        }
        means3D[3 * p] = __fmaf_rn(acc[0], scale, x0);
        means3D[3 * p + 1] = __fmaf_rn(acc[1], scale, x1);
        means3D[3 * p + 2] = __fmaf_rn(acc[2], scale, x2);
        scales[3 * p] = __expf(s0); scales[3 * p + 1] = __expf(s1); scales[3 * p + 2] = __expf(s2);
        const float inv = 1.0f / fmaxf(sqrtf(__fmaf_rn(q.w, q.w, __fmaf_rn(q.z, q.z, __fmaf_rn(q.y, q.y, q.x * q.x)))), 1e-12f);
        reinterpret_cast<float4*>(rots)[p] = make_float4(__fmaf_rn(q.x, inv, acc[3]), __fmaf_rn(q.y, inv, acc[4]),
                                                         __fmaf_rn(q.z, inv, acc[5]), __fmaf_rn(q.w, inv, acc[6]));
        opac[p] = 1.0f / (1.0f + __expf(-op));
    }
}

__global__ void __launch_bounds__(1024)
rdg_dyn_getter_bwd_kernel(int P, int Tu, const long long* __restrict__ time_ind, const float* __restrict__ bases,
                          float scale, const float* __restrict__ scaling, const float* __restrict__ rotation,
                          const float* __restrict__ opacity, const float* __restrict__ g_means3D,
                          const float* __restrict__ g_scales, const float* __restrict__ g_rots,
                          const float* __restrict__ g_opac, float* __restrict__ d_xyz, float* __restrict__ d_scaling,
                          float* __restrict__ d_rotation, float* __restrict__ d_opacity, float* __restrict__ d_coeff,
                          const int* __restrict__ inv_order, float4* __restrict__ gs, uint32_t* __restrict__ zero_out,
                          int n_zero, const float* __restrict__ coeff_abl = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float smem_dg[];
    // the finished-workgroup counter of the dB reduction's last stage is cleared here (one memset launch less)
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n_zero; k += gridDim.x * blockDim.x) zero_out[k] = 0u;
    rdg_diff16_to_lds(smem_dg, Tu, bases);
    __syncthreads();
    // an absent upstream gradient reads the parameter of the same shape instead (any valid address) and is replaced by
    // zero afterwards: no branch between the loads, so all of a Gaussian's loads are in flight together (with the four
    // null tests as branches the loop made ten dependent round trips to memory per Gaussian: 2.7 TB/s)
    const bool has_m = g_means3D != nullptr, has_s = g_scales != nullptr, has_r = g_rots != nullptr, has_o = g_opac != nullptr;
    const float* pm = has_m ? g_means3D : scaling;
    const float* ps = has_s ? g_scales : scaling;
    const float4* pr = reinterpret_cast<const float4*>(has_r ? g_rots : rotation);
    const float* po = has_o ? g_opac : opacity;
    const int lane = threadIdx.x & 63;
    float* stage = smem_dg + Tu * RDG_DC_STRIDE + (threadIdx.x >> 6) * (64 * RDG_DG_STAGE);
    for (int p0 = (blockIdx.x * blockDim.x + threadIdx.x) - lane; p0 < P; p0 += gridDim.x * blockDim.x) {
        const int p = p0 + lane;
        const bool live = p < P;
        const int pc = live ? p : P - 1;
        const int u = (int)time_ind[pc];
        const size_t sidx = (size_t)inv_order[pc];
        float gm[3], gsc[3], sc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { gm[k] = pm[3 * pc + k]; gsc[k] = ps[3 * pc + k]; sc[k] = scaling[3 * pc + k]; }
        const float4 q = reinterpret_cast<const float4*>(rotation)[pc];
        float4 gr = pr[pc];
        const float opv = opacity[pc];
        float gop = po[pc];
        // loaded here, not behind a branch further down
        asm volatile("" : "+v"(sc[0]), "+v"(sc[1]), "+v"(sc[2]), "+v"(gr.x), "+v"(gr.y), "+v"(gr.z), "+v"(gr.w), "+v"(gop));
        float qx = q.x, qy = q.y, qz = q.z, qw = q.w;
        asm volatile("" : "+v"(qx), "+v"(qy), "+v"(qz), "+v"(qw));
#pragma unroll
        for (int k = 0; k < 3; ++k) { gm[k] = has_m ? gm[k] : 0.0f; gsc[k] = has_s ? gsc[k] : 0.0f; }
        gr = has_r ? gr : make_float4(0.f, 0.f, 0.f, 0.f);
        gop = has_o ? gop : 0.0f;
        float g[RDG_DEF_K];
        // ---- activations backward (as rdg_activate_bwd_kernel) ----
#pragma unroll
        for (int k = 0; k < 3; ++k) g[k] = gm[k] * scale;
        g[3] = gr.x; g[4] = gr.y; g[5] = gr.z; g[6] = gr.w;
        float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_r) {
            const float nn = sqrtf(qx * qx + qy * qy + qz * qz + qw * qw);
            if (nn > 1e-12f) {
                const float inv = 1.0f / nn;
                const float yx = qx * inv, yy = qy * inv, yz = qz * inv, yw = qw * inv;
                const float dot = yx * gr.x + yy * gr.y + yz * gr.z + yw * gr.w;
                dq = make_float4((gr.x - yx * dot) * inv, (gr.y - yy * dot) * inv, (gr.z - yz * dot) * inv,
                                 (gr.w - yw * dot) * inv);
            } else {
                dq = make_float4(gr.x * 1e12f, gr.y * 1e12f, gr.z * 1e12f, gr.w * 1e12f);
            }
        }
        const float sg = 1.0f / (1.0f + __expf(-opv));
        // ---- the stores that do not wait for the table row: issued first, their registers are free for the row loop ----
        if (live) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d_xyz[3 * p + k] = gm[k];
                d_scaling[3 * p + k] = has_s ? gsc[k] * __expf(sc[k]) : 0.0f;
            }
            reinterpret_cast<float4*>(d_rotation)[p] = dq;
            d_opacity[p] = has_o ? gop * sg * (1.0f - sg) : 0.0f;
#ifdef RDG_ABL_FUSE_DB
            // ablation build (results wrong, timing only; profiles/r05_experiments.txt): what a dB reduction fused into this
            // kernel would move at the least -- no birth-sorted copy written, the coefficient row read here instead
            {
                const float4* cr = reinterpret_cast<const float4*>(coeff_abl) + 4 * (size_t)p;
                float4 c0 = cr[0], c1 = cr[1], c2 = cr[2], c3 = cr[3];
                asm volatile("" :: "v"(c0.x), "v"(c1.y), "v"(c2.z), "v"(c3.w), "v"(c0.w), "v"(c1.x), "v"(c2.x), "v"(c3.x));
            }
#else
            // sorted compact copy (scaled gradient, birth index) for the dB reduction
            gs[2 * sidx] = make_float4(g[0], g[1], g[2], g[3]);
            gs[2 * sidx + 1] = make_float4(g[4], g[5], g[6], __int_as_float(u));
#endif
        }
        // ---- deformation backward: dL/dcoeff ----
        const float4* r4 = reinterpret_cast<const float4*>(smem_dg + u * RDG_DC_STRIDE);
        float sacc[16];
#pragma unroll
        for (int b = 0; b < 16; ++b) sacc[b] = 0.0f;
#pragma unroll
        for (int j = 0; j < 28; ++j) {
            const float4 v = r4[j];
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const int e = 4 * j + e4;
                sacc[e / 7] += g[e % 7] * vv[e4];
            }
            if ((j & 3) == 3) asm volatile("" ::: "memory");   // at most four row reads hoisted: 112 live registers otherwise (spills)
        }
        float4 dc[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) dc[q4] = make_float4(sacc[4 * q4], sacc[4 * q4 + 1], sacc[4 * q4 + 2], sacc[4 * q4 + 3]);
        rdg_stage_rows16_out(d_coeff, p0, P, stage, lane, dc);
    }
}

// ---------------------------------------------------------------------------------------------------------
// dB accumulation on the matrix cores (the one GEMM-shaped piece of this path: K = number of Gaussians).
// acc[u] = sum_{birth(p)=u} c_p^T g_p  is a [16 x n_u] x [n_u x 7] product per birth index u.  Gaussians are
// visited in BIRTH-SORTED order (`order`, a permutation the host caches: birth indices only change at
// densification), so a wave's run of Gaussians shares one u and the wave keeps a 16x16 f32 accumulator in
// registers: one v_mfma_f32_16x16x4_f32 per 4 Gaussians (A = coefficients [16 x 4], B = gradients [4 x 16],
// columns 7..15 zero), flushed with 112 float atomics only when u changes (about once per wave).
// f32-input MFMA is bit-for-bit a k-ordered fmaf chain, so numerics equal the VALU form.
// ---------------------------------------------------------------------------------------------------------
typedef float rdg_f32x4 __attribute__((ext_vector_type(4)));
#define RDG_DEF_G 8  // groups of 4 Gaussians gathered per loop iteration
#define RDG_DEF_ACC_BLOCKS 2048   // workgroups (of 4 waves) of the dB accumulation with a table
#define RDG_DEF_PART_MAX_TU 1024  // the partial-slot form covers tables of up to this many birth times
// slots of the deterministic partial sums: (waves + birth indices) rows of 112 floats, kept behind the sorted copy
#define RDG_DEF_PART_BYTES (256 + (size_t)(RDG_DEF_ACC_BLOCKS * 4 + RDG_DEF_PART_MAX_TU) * 16 * RDG_DEF_K * 4)   // counter, slots
__host__ __device__ __forceinline__ int rdg_deform_rows_per_wave(int P, int nwaves) {
    const int per = (P + nwaves - 1) / nwaves;
    return (per + 4 * RDG_DEF_G - 1) / (4 * RDG_DEF_G) * (4 * RDG_DEF_G);
}
static inline size_t rdg_deform_gs_bytes(int32_t P) { return rdg_align_up((size_t)(P > 0 ? P : 1) * 32 + 256, 256); }

// part != nullptr (deterministic form, only with a table): the wave STORES its total for birth index u to the slot
// (wave + u) of `part` -- along the birth-sorted sequence either the wave or the birth index advances from one
// (wave, u) pair to the next, so the sum is a unique slot number -- and rdg_deform_part_finalize_kernel adds the slots
// of a birth index in wave order.  No float atomics: the same bits on every run.
__device__ __forceinline__ void rdg_deform_flush(rdg_f32x4 acc, int u, int lane, bool has_table,
                                                 float* __restrict__ d_basis_t, float* __restrict__ d_table,
                                                 float* __restrict__ part = nullptr, int wave = 0) {
    const int k = lane & 15;
    if (u < 0 || k >= RDG_DEF_K) return;
    if (part) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            part[(size_t)(wave + u) * (16 * RDG_DEF_K) + ((lane >> 4) * 4 + r) * RDG_DEF_K + k] = acc[r];
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = (lane >> 4) * 4 + r;
        const float v = acc[r];
        if (v != 0.0f) {
            // with a table, dB(t) = -sum_u dB_table[u] is formed afterwards by rdg_deform_dbt_kernel: thousands of
            // waves adding into the SAME 112 floats ran at the contended-atomic rate (0.64 ms at P = 1 M)
            if (has_table) atomicAdd(&d_table[(size_t)u * (16 * RDG_DEF_K) + b * RDG_DEF_K + k], -v);
            else atomicAdd(&d_basis_t[b * RDG_DEF_K + k], v);
        }
    }
}

__global__ void __launch_bounds__(256)
rdg_deform_bwd_acc_mfma_kernel(int P, const float* __restrict__ coeff, const long long* __restrict__ time_ind,
                               const int* __restrict__ order, const float* __restrict__ g_xyz,
                               const float* __restrict__ g_rot, float scale, int has_table,
                               float* __restrict__ d_basis_t, float* __restrict__ d_table,
                               const float* __restrict__ gs, float* __restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int per = rdg_deform_rows_per_wave(P, nwaves);
    const int beg = min(P, wave * per), end = min(P, beg + per);
    const int slot = lane >> 4, j = lane & 15;
    rdg_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int cur_u = -1;
    // The partial-slot form is only right for a sequence that really is sorted by birth index (`order` comes from a host
    // cache): every wave checks its own stretch -- and the row before it -- and raises a flag the last stage turns into
    // NaN gradients.  A stale permutation must fail loudly, not train on garbage.
    int prev_u = -1;
    if (part && beg > 0 && beg < end)
        prev_u = gs ? __float_as_int(gs[(size_t)(beg - 1) * 8 + 7]) : (int)time_ind[order[beg - 1]];
    bool unsorted = false;
    const int jx = min(j, 2), jr = min(max(j - 3, 0), 3);
    const float wx = j < 3 ? scale : 0.0f, wr = (j >= 3 && j < RDG_DEF_K) ? 1.0f : 0.0f;
    for (int base = beg; base < end; base += 4 * RDG_DEF_G) {
        // RDG_DEF_G groups of 4 Gaussians.  Branch-free gathers (indices clamped, values masked afterwards) so
        // that all first-level loads (order[]) and then all second-level loads are in flight together.
        int p4[RDG_DEF_G], u4[RDG_DEF_G]; float a4[RDG_DEF_G], g4[RDG_DEF_G]; bool v4[RDG_DEF_G];
#pragma unroll
        for (int q = 0; q < RDG_DEF_G; ++q) {
            const int idx = base + 4 * q + slot;
            v4[q] = idx < end;
            p4[q] = order[min(idx, end - 1)];
        }
        if (gs) {
            // sorted compact copy available: (g[0..6], birth index) are 8 contiguous floats at the sorted position
#pragma unroll
            for (int q = 0; q < RDG_DEF_G; ++q) {
                const size_t p = (size_t)p4[q];
                const size_t sidx = (size_t)min(base + 4 * q + slot, end - 1);
                const float av = coeff[p * 16 + j];
                const float gv = gs[sidx * 8 + min(j, 6)];
                const int tu = __float_as_int(gs[sidx * 8 + 7]);
                u4[q] = v4[q] ? (has_table ? tu : 0) : -1;
                a4[q] = v4[q] ? av : 0.0f;
                g4[q] = (v4[q] && j < RDG_DEF_K) ? gv : 0.0f;
            }
        } else {
#pragma unroll
            for (int q = 0; q < RDG_DEF_G; ++q) {
                const size_t p = (size_t)p4[q];
                const long long tu = time_ind[p];
                const float av = coeff[p * 16 + j];
                const float gx = g_xyz[p * 3 + jx], gr = g_rot[p * 4 + jr];
                u4[q] = v4[q] ? (has_table ? (int)tu : 0) : -1;
                a4[q] = v4[q] ? av : 0.0f;
                g4[q] = v4[q] ? (gx * wx + gr * wr) : 0.0f;   // arithmetic blend: keeps both loads unconditional
            }
        }
#pragma unroll
        for (int q = 0; q < RDG_DEF_G; ++q) {
            const int u = u4[q];
            const float a = a4[q], gv = g4[q];
            if (__ballot(v4[q] && u != cur_u) == 0ull) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, gv, acc, 0, 0, 0);
            } else {
                // a birth-index boundary inside this group of 4 (or the first group): one slot at a time
#pragma unroll 1
                for (int sl = 0; sl < 4; ++sl) {
                    const int us = __builtin_amdgcn_readlane(u, sl * 16);
                    if (us < 0) continue;
                    if (us != cur_u) {
                        unsorted |= us < max(cur_u, prev_u);
                        rdg_deform_flush(acc, cur_u, lane, has_table != 0, d_basis_t, d_table, part, wave);
                        acc = rdg_f32x4{0.f, 0.f, 0.f, 0.f};
                        cur_u = us;
                    }
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, slot == sl ? gv : 0.0f, acc, 0, 0, 0);
                }
            }
        }
    }
    rdg_deform_flush(acc, cur_u, lane, has_table != 0, d_basis_t, d_table, part, wave);
    // the flag sits in the word after the finished-workgroup counter, 256 B before the slots
    if (part && unsorted && lane == 0) atomicOr(reinterpret_cast<unsigned int*>(part) - 63, 1u);
}

// sum over u of d_table[u][c] in a fixed order (eight interleaved partial sums, then a tree)
__device__ __forceinline__ float rdg_dbt_sum(const float* __restrict__ d_table, int Tu, int row, int c) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int u = 0;
    for (; u + 7 < Tu; u += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] += d_table[(size_t)(u + q) * row + c];
    }
    for (; u < Tu; ++u) a[0] += d_table[(size_t)u * row + c];
    return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

// dB_table[u] = -(sum over the waves whose row range meets birth index u, in wave order, of their partial totals);
// the workgroup that finishes last then forms dB(t) = -sum_u dB_table[u] (one launch less; `counter` must be zero on
// entry and is left zero).  Where the birth-sorted sequence switches to index u: seg_start[u] (int32 [Tu + 1], the
// host caches it with the permutation) or, without it, a binary search on the sequence itself (the sorted compact copy
// carries the index in its 8th float; without the copy: time_ind[order[.]]) -- 20 dependent loads, 25 us at P = 1 M.
#define RDG_DEF_FIN_GROUPS 8
__global__ void __launch_bounds__(128 * RDG_DEF_FIN_GROUPS)
rdg_deform_part_finalize_kernel(int P, int Tu, int per, const int* __restrict__ seg_start,
                                const float* __restrict__ gs, const long long* __restrict__ time_ind,
                                const int* __restrict__ order, const float* __restrict__ part,
                                float* __restrict__ d_table, float* __restrict__ d_basis_t,
                                uint32_t* __restrict__ counter) {
    // 8 groups of 128 threads: group g adds the slots w0 + g, w0 + g + 8, ... of element e, then the eight group totals
    // are added in group order -- a fixed order (what makes the sum reproducible) with eight loads in flight per element
    const int u = blockIdx.x, e = threadIdx.x & 127, grp = threadIdx.x >> 7;
    __shared__ uint32_t sOld;
    __shared__ float sPart[RDG_DEF_FIN_GROUPS][128];
    float acc = 0.0f;
    if (e < 16 * RDG_DEF_K) {
        int bound[2];
        if (seg_start) {
            bound[0] = seg_start[u]; bound[1] = seg_start[u + 1];
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) {           // first sorted position whose birth index is >= u + q
                int lo = 0, hi = P;
                while (lo < hi) {
                    const int mid = lo + ((hi - lo) >> 1);
                    const int um = gs ? __float_as_int(gs[(size_t)mid * 8 + 7]) : (int)time_ind[order[mid]];
                    if (um < u + q) lo = mid + 1; else hi = mid;
                }
                bound[q] = lo;
            }
        }
        if (bound[1] > bound[0]) {
            const int w0 = bound[0] / per, w1 = (bound[1] - 1) / per;
            for (int w = w0 + grp; w <= w1; w += RDG_DEF_FIN_GROUPS) acc += part[(size_t)(w + u) * (16 * RDG_DEF_K) + e];
        }
    }
    sPart[grp][e] = acc;
    __syncthreads();
    if (grp == 0 && e < 16 * RDG_DEF_K) {
        float tot = sPart[0][e];
#pragma unroll
        for (int g = 1; g < RDG_DEF_FIN_GROUPS; ++g) tot += sPart[g][e];
        if (counter[1] != 0u) tot = __int_as_float(0x7fc00000);   // `order` was not sorted by birth index: poison, loudly
        d_table[(size_t)u * (16 * RDG_DEF_K) + e] = -tot;
    }
    // publish this row (agent-scope release), count the finished workgroups; the last one acquires and reduces over u
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sOld = atomicAdd(counter, 1u);
    }
    __syncthreads();
    if (sOld != (uint32_t)(Tu - 1)) return;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *counter = 0u;
    }
    __syncthreads();
    if (grp == 0 && e < 16 * RDG_DEF_K) d_basis_t[e] = -rdg_dbt_sum(d_table, Tu, 16 * RDG_DEF_K, e);
}

// ---------------------------------------------------------------------------------------------------------
// Multi-view dynamic getter (Gaussian-sharded frame-DP, rodygs_amd/sharded.py): the same Gaussians deformed to the
// times of the NV cameras of one step.  A Gaussian's parameters and coefficients are read once; scales / opacities
// do not depend on the time and are written once; backward sums the gradients over the views in registers, so the
// five parameter gradients leave the kernel final (no per-view copies, no reduction pass).
// bases_all [NV][Tu+1][112]: every view's packed bases (table rows, then B(t_v)); the table rows are identical in
// all views and are taken from view 0.  LDS: raw table [Tu][RDG_DC_STRIDE] + B(t_v) [NV][112].
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rdg_views_to_lds(float* sm, int Tu, int NV, const float* __restrict__ bases_all) {
    for (int k = threadIdx.x; k < Tu * 112; k += blockDim.x) {
        const int u = k / 112, c = k - u * 112;
        sm[u * RDG_DC_STRIDE + c] = bases_all[(size_t)u * 112 + c];
    }
    float* bt = sm + Tu * RDG_DC_STRIDE;
    for (int k = threadIdx.x; k < NV * 112; k += blockDim.x) {
        const int v = k / 112, c = k - v * 112;
        bt[k] = bases_all[((size_t)v * (Tu + 1) + Tu) * 112 + c];
    }
}

// The table row of a lane is the expensive LDS operand (64 different rows per instruction): both kernels read it once
// per Gaussian into registers and reuse it for every view (re-reading it per view was LDS-bound: 137 us for 125 k
// Gaussians x 8 views); B(t_v) is a broadcast read.  The arithmetic is the reference's -- differences first, then the
// dot product (evaluating c.B(t_v) - c.table[birth] instead loses digits when the motion basis is smooth in time) --
// and the single-view getter performs the same operations in the same order, so both frame-DP modes produce the same bits.
__global__ void __launch_bounds__(512)
rdg_dyn_getter_views_fwd_kernel(int P, int Tu, int NV, int stride, const float* __restrict__ coeff,
                                const long long* __restrict__ time_ind, const float* __restrict__ bases_all,
                                float scale, const float* __restrict__ xyz, const float* __restrict__ scaling,
                                const float* __restrict__ rotation, const float* __restrict__ opacity,
                                float* __restrict__ means3D, float* __restrict__ scales, float* __restrict__ rots,
                                float* __restrict__ opac) {
    extern __shared__ __attribute__((aligned(16))) float smem_dg[];
    rdg_views_to_lds(smem_dg, Tu, NV, bases_all);
    __syncthreads();
    const float* sbt = smem_dg + Tu * RDG_DC_STRIDE;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const int u = (int)time_ind[p];
        const float4* c4 = reinterpret_cast<const float4*>(coeff + (size_t)p * 16);
        float c[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float4 t = c4[q]; c[4 * q] = t.x; c[4 * q + 1] = t.y; c[4 * q + 2] = t.z; c[4 * q + 3] = t.w; }
        const float4* r4 = reinterpret_cast<const float4*>(smem_dg + u * RDG_DC_STRIDE);
        const float x0 = xyz[3 * p], x1 = xyz[3 * p + 1], x2 = xyz[3 * p + 2];
        const float4 q = reinterpret_cast<const float4*>(rotation)[p];
        const float inv = 1.0f / fmaxf(sqrtf(__fmaf_rn(q.w, q.w, __fmaf_rn(q.z, q.z, __fmaf_rn(q.y, q.y, q.x * q.x)))), 1e-12f);
#pragma unroll
        for (int k = 0; k < 3; ++k) scales[3 * p + k] = __expf(scaling[3 * p + k]);
        opac[p] = 1.0f / (1.0f + __expf(-opacity[p]));
        // the birth row is the expensive LDS operand (64 different rows per instruction): read ONCE, kept in registers
        // for all views; B(t_v) is a broadcast read.  Arithmetic = the reference's: differences first, then the dot.
        float4 trow[28];
#pragma unroll
        for (int j = 0; j < 28; ++j) trow[j] = r4[j];
        for (int v = 0; v < NV; ++v) {
            const float4* b4 = reinterpret_cast<const float4*>(sbt + v * 112);
            float acc[RDG_DEF_K];
#pragma unroll
            for (int k = 0; k < RDG_DEF_K; ++k) acc[k] = 0.0f;
#pragma unroll
            for (int j = 0; j < 28; ++j) {
                const float4 t = trow[j], bb = b4[j];
                const float vv[4] = {bb.x - t.x, bb.y - t.y, bb.z - t.z, bb.w - t.w};
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const int e = 4 * j + e4;
                    acc[e % 7] = __fmaf_rn(c[e / 7], vv[e4], acc[e % 7]);
                }
                // compiler barrier: without it all 28 broadcast reads of a view are hoisted (112 more live registers
                // on top of the 112 of the row) and the kernel spills
                asm volatile("" ::: "memory");
            }
            const size_t row = (size_t)v * stride + p;
            means3D[3 * row] = __fmaf_rn(acc[0], scale, x0);
            means3D[3 * row + 1] = __fmaf_rn(acc[1], scale, x1);
            means3D[3 * row + 2] = __fmaf_rn(acc[2], scale, x2);
            reinterpret_cast<float4*>(rots)[row] = make_float4(__fmaf_rn(q.x, inv, acc[3]), __fmaf_rn(q.y, inv, acc[4]),
                                                               __fmaf_rn(q.z, inv, acc[5]), __fmaf_rn(q.w, inv, acc[6]));
        }
    }
}

__global__ void __launch_bounds__(512)
rdg_dyn_getter_views_bwd_kernel(int P, int Tu, int NV, int stride, const long long* __restrict__ time_ind,
                                const float* __restrict__ bases_all, float scale, const float* __restrict__ scaling,
                                const float* __restrict__ rotation, const float* __restrict__ opacity,
                                const float* __restrict__ g_means3D, const float* __restrict__ g_scales,
                                const float* __restrict__ g_rots, const float* __restrict__ g_opac,
                                float* __restrict__ d_xyz, float* __restrict__ d_scaling,
                                float* __restrict__ d_rotation, float* __restrict__ d_opacity,
                                float* __restrict__ d_coeff, const int* __restrict__ inv_order,
                                float4* __restrict__ gs) {
    extern __shared__ __attribute__((aligned(16))) float smem_dg[];
    rdg_views_to_lds(smem_dg, Tu, NV, bases_all);
    __syncthreads();
    const float* sbt = smem_dg + Tu * RDG_DC_STRIDE;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const int u = (int)time_ind[p];
        const size_t sidx = (size_t)inv_order[p];
        float gsum[RDG_DEF_K] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float sm3[3] = {0.f, 0.f, 0.f}, ssc[3] = {0.f, 0.f, 0.f}, sop = 0.f;
        float sacc[16];
#pragma unroll
        for (int b = 0; b < 16; ++b) sacc[b] = 0.0f;
        // birth row of the table: read once, kept in registers for all views (as in the forward kernel)
        const float4* r4 = reinterpret_cast<const float4*>(smem_dg + u * RDG_DC_STRIDE);
        float4 trow[28];
#pragma unroll
        for (int j = 0; j < 28; ++j) trow[j] = r4[j];
        for (int v = 0; v < NV; ++v) {
            const size_t row = (size_t)v * stride + p;
            const float m0 = g_means3D[3 * row], m1 = g_means3D[3 * row + 1], m2 = g_means3D[3 * row + 2];
            const float4 gr = reinterpret_cast<const float4*>(g_rots)[row];
            sm3[0] += m0; sm3[1] += m1; sm3[2] += m2;
#pragma unroll
            for (int k = 0; k < 3; ++k) ssc[k] += g_scales[3 * row + k];
            sop += g_opac[row];
            const float g[RDG_DEF_K] = {m0 * scale, m1 * scale, m2 * scale, gr.x, gr.y, gr.z, gr.w};
#pragma unroll
            for (int k = 0; k < RDG_DEF_K; ++k) gsum[k] += g[k];
            // compact copy at the birth-sorted position, views interleaved: [sorted Gaussian][view][g0..g6, birth]
            gs[(sidx * NV + v) * 2] = make_float4(g[0], g[1], g[2], g[3]);
            gs[(sidx * NV + v) * 2 + 1] = make_float4(g[4], g[5], g[6], __int_as_float(u));
            const float4* b4 = reinterpret_cast<const float4*>(sbt + v * 112);
#pragma unroll
            for (int j = 0; j < 28; ++j) {
                const float4 b = b4[j], t = trow[j];
                // dL/dcoeff += g_v . (B(t_v) - table[birth]): differences first, as the reference (and the single-view
                // kernel) evaluate it
                const float vv[4] = {b.x - t.x, b.y - t.y, b.z - t.z, b.w - t.w};
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const int e = 4 * j + e4;
                    sacc[e / 7] = __fmaf_rn(g[e % 7], vv[e4], sacc[e / 7]);
                }
                asm volatile("" ::: "memory");     // keep the broadcast reads next to their use (register pressure)
            }
        }
        float4* dc = reinterpret_cast<float4*>(d_coeff + (size_t)p * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) dc[q] = make_float4(sacc[4 * q], sacc[4 * q + 1], sacc[4 * q + 2], sacc[4 * q + 3]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d_xyz[3 * p + k] = sm3[k];
            d_scaling[3 * p + k] = ssc[k] * __expf(scaling[3 * p + k]);
        }
        const float4 q = reinterpret_cast<const float4*>(rotation)[p];
        const float4 gr = make_float4(gsum[3], gsum[4], gsum[5], gsum[6]);
        const float nn = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
        float4 dq;
        if (nn > 1e-12f) {
            const float inv = 1.0f / nn;
            const float yx = q.x * inv, yy = q.y * inv, yz = q.z * inv, yw = q.w * inv;
            const float dot = yx * gr.x + yy * gr.y + yz * gr.z + yw * gr.w;
            dq = make_float4((gr.x - yx * dot) * inv, (gr.y - yy * dot) * inv, (gr.z - yz * dot) * inv,
                             (gr.w - yw * dot) * inv);
        } else {
            dq = make_float4(gr.x * 1e12f, gr.y * 1e12f, gr.z * 1e12f, gr.w * 1e12f);
        }
        reinterpret_cast<float4*>(d_rotation)[p] = dq;
        const float sg = 1.0f / (1.0f + __expf(-opacity[p]));
        d_opacity[p] = sop * sg * (1.0f - sg);
    }
}

// dB accumulation for all views in one pass over the birth-sorted Gaussians: the coefficient fragment of a group of 4
// Gaussians is gathered once and multiplied with the gradient rows of two views per MFMA (columns 0-6 and 8-14 of the
// 16-column B operand), NPAIR accumulators per wave; flushed per birth index into that view's d_table (negated, as in
// the single-view kernel), from which rdg_deform_dbt_views_kernel forms dB(t_v).
#define RDG_DEFV_G 4
template <int NPAIR>
__global__ void __launch_bounds__(256)
rdg_deform_bwd_acc_views_kernel(int P, int NV, int Tu, const float* __restrict__ coeff, const int* __restrict__ order,
                                float* __restrict__ d_bases_all, const float* __restrict__ gs) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    int per = (P + nwaves - 1) / nwaves;
    per = (per + 4 * RDG_DEFV_G - 1) / (4 * RDG_DEFV_G) * (4 * RDG_DEFV_G);
    const int beg = min(P, wave * per), end = min(P, beg + per);
    const int slot = lane >> 4, j = lane & 15;
    const int jh = j >> 3, jk = j & 7;                  // which view of the pair, which gradient component
    rdg_f32x4 acc[NPAIR];
#pragma unroll
    for (int q = 0; q < NPAIR; ++q) acc[q] = rdg_f32x4{0.f, 0.f, 0.f, 0.f};
    int cur_u = -1;
    const size_t vrow = (size_t)(Tu + 1) * 112;

    auto flush = [&](int u) {
        if (u < 0 || jk >= RDG_DEF_K) return;
#pragma unroll
        for (int q = 0; q < NPAIR; ++q) {
            const int v = 2 * q + jh;
            if (v >= NV) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = (lane >> 4) * 4 + r;
                const float val = acc[q][r];
                if (val != 0.0f) atomicAdd(&d_bases_all[v * vrow + (size_t)u * 112 + b * RDG_DEF_K + jk], -val);
            }
        }
    };

    for (int base = beg; base < end; base += 4 * RDG_DEFV_G) {
        int u4[RDG_DEFV_G]; float a4[RDG_DEFV_G]; float g4[RDG_DEFV_G][NPAIR]; bool v4[RDG_DEFV_G];
#pragma unroll
        for (int q = 0; q < RDG_DEFV_G; ++q) {
            const int idx = base + 4 * q + slot;
            v4[q] = idx < end;
            const size_t sidx = (size_t)min(idx, end - 1);
            const size_t p = (size_t)order[sidx];
            a4[q] = v4[q] ? coeff[p * 16 + j] : 0.0f;
            const float* row = gs + sidx * NV * 8;
            u4[q] = v4[q] ? __float_as_int(row[7]) : -1;
#pragma unroll
            for (int w = 0; w < NPAIR; ++w) {
                const int v = 2 * w + jh;
                const float gv = row[min(v, NV - 1) * 8 + jk];
                g4[q][w] = (v4[q] && v < NV && jk < RDG_DEF_K) ? gv : 0.0f;
            }
        }
#pragma unroll
        for (int q = 0; q < RDG_DEFV_G; ++q) {
            const int u = u4[q];
            if (__ballot(v4[q] && u != cur_u) == 0ull) {
#pragma unroll
                for (int w = 0; w < NPAIR; ++w) acc[w] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q], g4[q][w], acc[w], 0, 0, 0);
            } else {
#pragma unroll 1
                for (int sl = 0; sl < 4; ++sl) {
                    const int us = __builtin_amdgcn_readlane(u, sl * 16);
                    if (us < 0) continue;
                    if (us != cur_u) {
                        flush(cur_u);
#pragma unroll
                        for (int w = 0; w < NPAIR; ++w) acc[w] = rdg_f32x4{0.f, 0.f, 0.f, 0.f};
                        cur_u = us;
                    }
#pragma unroll
                    for (int w = 0; w < NPAIR; ++w)
                        acc[w] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q], slot == sl ? g4[q][w] : 0.0f, acc[w], 0, 0, 0);
                }
            }
        }
    }
    flush(cur_u);
}

// dB(t_v)[c] = -sum_u d_table_v[u][c] for every view (blockIdx.y = view)
__global__ void rdg_deform_dbt_views_kernel(int Tu, float* __restrict__ d_bases_all) {
    const int c = threadIdx.x;
    if (c >= 112) return;
    float* dv = d_bases_all + (size_t)blockIdx.y * (Tu + 1) * 112;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    int u = 0;
    for (; u + 3 < Tu; u += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] += dv[(size_t)(u + q) * 112 + c];
    }
    for (; u < Tu; ++u) a[0] += dv[(size_t)u * 112 + c];
    dv[(size_t)Tu * 112 + c] = -((a[0] + a[1]) + (a[2] + a[3]));
}

// dB(t)[c] = -sum_u dB_table[u][c]   (fixed order: deterministic given dB_table)
__global__ void rdg_deform_dbt_kernel(int Tu, int row, const float* __restrict__ d_table, float* __restrict__ d_basis_t) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= row) return;
    d_basis_t[c] = -rdg_dbt_sum(d_table, Tu, row, c);
}

// Fused Adam over a flat f32 segment.  row_len > 1 gives the segment a row structure whose first head_len floats
// take step_head and the rest step_tail: the SH features [P,16,3] are ONE tensor whose DC coefficient trains 20x
// faster than the rest (feature_lr vs feature_lr/20), so no cat/split of f_dc/f_rest is ever needed.
typedef float rdg_f4 __attribute__((ext_vector_type(4)));
template <int VAR>
__device__ __forceinline__ float4 rdg_ld4(const float* base, long long i) {
    if (VAR & 1) {
        const rdg_f4 t = __builtin_nontemporal_load(reinterpret_cast<const rdg_f4*>(base) + i);
        return make_float4(t.x, t.y, t.z, t.w);
    }
    return reinterpret_cast<const float4*>(base)[i];
}
template <int VAR>
__device__ __forceinline__ void rdg_st4(float* base, long long i, float4 v) {
    if (VAR & 1) {
        rdg_f4 t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<rdg_f4*>(base) + i);
    } else {
        reinterpret_cast<float4*>(base)[i] = v;
    }
}

template <int VAR>
__device__ __forceinline__ void
rdg_adam_segment(long long n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                 float* __restrict__ v, float step_head, float step_tail, int row_len, int head_len, float b1, float b2, float omb1, float omb2,
                 float eps, float bc2_sqrt, const float* __restrict__ g2 = nullptr) {
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 pp = rdg_ld4<VAR>(p, i);
        float4 gg = rdg_ld4<VAR>(g, i);
        if (g2) {      // RdgAdamSeg.grad2: the gradient is the sum of two buffers (one float add per element, as AccumulateGrad's)
            const float4 hh = rdg_ld4<VAR>(g2, i);
            gg.x += hh.x; gg.y += hh.y; gg.z += hh.z; gg.w += hh.w;
        }
        float4 mm = rdg_ld4<VAR>(m, i);
        float4 vv = rdg_ld4<VAR>(v, i);
        float s0 = step_tail, s1 = step_tail, s2 = step_tail, s3 = step_tail;
        if (row_len > 1) {
            const unsigned r = (unsigned)((i * 4) % row_len);   // row_len is a multiple of 4 or handled per lane below
            s0 = (int)(r % row_len) < head_len ? step_head : step_tail;
            s1 = (int)((r + 1) % row_len) < head_len ? step_head : step_tail;
            s2 = (int)((r + 2) % row_len) < head_len ? step_head : step_tail;
            s3 = (int)((r + 3) % row_len) < head_len ? step_head : step_tail;
        } else {
            s0 = s1 = s2 = s3 = step_head;
        }
#define RDG_ADAM1(c, st) rdg_adam_elem(pp.c, gg.c, mm.c, vv.c, st, b1, b2, omb1, omb2, eps, bc2_sqrt);
        RDG_ADAM1(x, s0) RDG_ADAM1(y, s1) RDG_ADAM1(z, s2) RDG_ADAM1(w, s3)
        rdg_st4<VAR>(p, i, pp);
        rdg_st4<VAR>(m, i, mm);
        rdg_st4<VAR>(v, i, vv);
    }
    // tail
    const long long t0 = n4 << 2;
    const long long i = t0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float st = (row_len > 1) ? ((int)(i % row_len) < head_len ? step_head : step_tail) : step_head;
        float pi = p[i], mi = m[i], vi = v[i];
        rdg_adam_elem(pi, g2 ? g[i] + g2[i] : g[i], mi, vi, st, b1, b2, omb1, omb2, eps, bc2_sqrt);
        m[i] = mi; v[i] = vi; p[i] = pi;
    }
}

__global__ void __launch_bounds__(256)
rdg_adam_kernel(long long n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                float* __restrict__ v, float step_head, float step_tail, int row_len, int head_len, float b1, float b2, float omb1, float omb2,
                float eps, float bc2_sqrt) {
    rdg_adam_segment<0>(n, p, g, m, v, step_head, step_tail, row_len, head_len, b1, b2, omb1, omb2, eps, bc2_sqrt);
}

// all parameter groups of a model in ONE launch: blockIdx.y selects the segment
struct RdgAdamSegs { RdgAdamSeg s[RDG_ADAM_MAX_SEGS]; };
template <int VAR>
__global__ void __launch_bounds__(256)
rdg_adam_multi_kernel(RdgAdamSegs segs, float inv_bc1, float b1, float b2, float omb1, float omb2, float eps,
                      float bc2_sqrt, const RdgStepScalars* __restrict__ dev) {
    const RdgAdamSeg sg = segs.s[blockIdx.y];
    float lr_head = sg.lr_head, lr_tail = sg.lr_tail;
    if (dev) {                                                  // graph replay
        inv_bc1 = dev->inv_bias_correction1; bc2_sqrt = dev->sqrt_bias_correction2;
        if (dev->lr_from_table) { lr_head = dev->seg_lr_head[blockIdx.y]; lr_tail = dev->seg_lr_tail[blockIdx.y]; }
    }
    rdg_adam_segment<VAR>(sg.n, sg.param, sg.grad, sg.exp_avg, sg.exp_avg_sq, lr_head * inv_bc1, lr_tail * inv_bc1,
                     sg.row_len, sg.head_len, b1, b2, omb1, omb2, eps, bc2_sqrt, sg.grad2);
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

size_t rdg_deform_sorted_ws_bytes(int32_t P) { return rdg_deform_gs_bytes(P) + RDG_DEF_PART_BYTES; }
size_t rdg_deform_sorted_views_ws_bytes(int32_t P, int32_t nviews) {
    return (size_t)(P > 0 ? P : 1) * 32 * (size_t)(nviews > 0 ? nviews : 1) + 256;
}

int rdg_deform_backward(int32_t P, int32_t B, int32_t Tu, const float* coeff, const int64_t* time_ind,
                        const float* basis_t, const float* table, float spatial_scale, const float* g_xyz,
                        const float* g_rot, float* d_coeff, float* d_basis_t, float* d_table, const int32_t* order,
                        const int32_t* inv_order, const int32_t* seg_start, void* sorted_ws, void* stream) {
    if (B <= 0 || B > RDG_DEF_MAXB * 4) return rdg_set_error("deform: bad basis count %d", B);
    hipStream_t st = (hipStream_t)stream;
    const int row = B * RDG_DEF_K;
    rdg_stage_begin(RDG_STAGE_DEFORM_BWD, st);
    hipError_t e = rdg_zero_async(d_basis_t, (size_t)row * 4, st);
    if (e == hipSuccess && table && d_table) e = rdg_zero_async(d_table, (size_t)Tu * row * 4, st);
    if (e != hipSuccess) return rdg_check_hip(e, "deform_bwd memset");
    if (P > 0) {
        const int Tu_eff = table ? Tu : 1;
        const bool mfma = order != nullptr && B == 16;   // birth-sorted order given: dB on the matrix cores
        const size_t lds = (mfma ? 1 : 2) * (size_t)Tu_eff * (row + 1) * sizeof(float);
        const int threads = 1024;
        int blocks = (P + threads - 1) / threads;
        if (blocks > (mfma ? 512 : 256)) blocks = mfma ? 512 : 256;
#define RDG_DEF_BWD(USE, ACC, LDSB)                                                                              \
        hipLaunchKernelGGL((rdg_deform_bwd_kernel<USE, ACC>), dim3(blocks), dim3(threads), LDSB, st, P, B, Tu, coeff, \
                           (const long long*)time_ind, basis_t, table, spatial_scale, g_xyz, g_rot, d_coeff,      \
                           d_basis_t, d_table)
        const size_t lds16 = (size_t)Tu_eff * RDG_DC_STRIDE * sizeof(float);
        bool use_gs = false;
        const bool dc16 = mfma && lds16 <= 64 * 1024 && (((uintptr_t)g_rot | (uintptr_t)d_coeff) & 15) == 0;
        if (dc16) {
            int nb = (P + 1023) / 1024;
            if (nb > 256) nb = 256;
            use_gs = inv_order != nullptr && sorted_ws != nullptr && (((uintptr_t)sorted_ws) & 15) == 0;
            hipLaunchKernelGGL(rdg_deform_dcoeff16_kernel, dim3(nb), dim3(1024), lds16, st, P, Tu, (const long long*)time_ind,
                               basis_t, table, spatial_scale, g_xyz, g_rot, d_coeff, (const int*)inv_order,
                               use_gs ? (float4*)sorted_ws : (float4*)nullptr);
        } else if (lds <= 128 * 1024) {
            if (lds > 64 * 1024) {
                hipError_t ea = mfma ? hipFuncSetAttribute((const void*)rdg_deform_bwd_kernel<true, false>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                     : hipFuncSetAttribute((const void*)rdg_deform_bwd_kernel<true, true>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (ea != hipSuccess) return rdg_check_hip(ea, "deform_bwd LDS attribute");
            }
            if (mfma) RDG_DEF_BWD(true, false, lds); else RDG_DEF_BWD(true, true, lds);
        } else {
            if (mfma) RDG_DEF_BWD(false, false, (size_t)row * 4); else RDG_DEF_BWD(false, true, (size_t)row * 4);
        }
        if (mfma) {
            // without a table every wave flushes into the same 112 floats: keep the wave count low there
            // with a table and a workspace: per-(wave, birth index) partial totals + a fixed-order sum (no float atomics)
            uint32_t* counter = (table && d_table && sorted_ws && Tu >= 1 && Tu <= RDG_DEF_PART_MAX_TU)
                                    ? (uint32_t*)((char*)sorted_ws + rdg_deform_gs_bytes(P)) : nullptr;
            float* part = counter ? (float*)((char*)counter + 256) : nullptr;
            if (counter) {
                hipError_t ec = rdg_zero_async(counter, 8, st);
                if (ec != hipSuccess) return rdg_check_hip(ec, "deform_bwd counter memset");
            }
            hipLaunchKernelGGL(rdg_deform_bwd_acc_mfma_kernel, dim3(table ? RDG_DEF_ACC_BLOCKS : 64), dim3(256), 0, st, P,
                               coeff, (const long long*)time_ind, (const int*)order, g_xyz, g_rot, spatial_scale,
                               table ? 1 : 0, d_basis_t, d_table, use_gs ? (const float*)sorted_ws : (const float*)nullptr,
                               part);
            if (part)
                hipLaunchKernelGGL(rdg_deform_part_finalize_kernel, dim3(Tu), dim3(128 * RDG_DEF_FIN_GROUPS), 0, st, P, Tu,
                                   rdg_deform_rows_per_wave(P, RDG_DEF_ACC_BLOCKS * 4), (const int*)seg_start,
                                   use_gs ? (const float*)sorted_ws : (const float*)nullptr, (const long long*)time_ind,
                                   (const int*)order, (const float*)part, d_table, d_basis_t, counter);
            else if (table)
                hipLaunchKernelGGL(rdg_deform_dbt_kernel, dim3((row + 127) / 128), dim3(128), 0, st, Tu, row, d_table,
                                   d_basis_t);
        }
    }
    rdg_stage_end(RDG_STAGE_DEFORM_BWD, st);
    return rdg_check_hip(hipGetLastError(), "deform_bwd launch");
}


int rdg_dyn_getter_supported(int32_t B, int32_t Tu) { return B == 16 && Tu >= 1 && ((size_t)Tu * RDG_DC_STRIDE + 112) * 4 <= 64 * 1024; }
// LDS of the single-camera backward kernel: the difference table + one coefficient-gradient stage per wave (above the
// 64 KB a kernel gets without asking: the attribute is set once)
static size_t rdg_getter_lds(int32_t Tu) { return (size_t)Tu * RDG_DC_STRIDE * 4 + 16 * RDG_DG_STAGE_BYTES; }
static hipError_t rdg_getter_lds_attr(const void* fn) {
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rdg_getter_lds(141));
}

int rdg_dyn_getter_forward(int32_t P, int32_t Tu, const float* coeff, const int64_t* time_ind, const float* bases,
                           float spatial_scale, const float* xyz, const float* scaling, const float* rotation,
                           const float* opacity, float* means3D, float* scales, float* rots, float* opac, void* stream) {
    if (!rdg_dyn_getter_supported(16, Tu)) return rdg_set_error("dyn_getter: unsupported table size Tu = %d", Tu);
    if ((((uintptr_t)coeff | (uintptr_t)rotation | (uintptr_t)rots)) & 15) return rdg_set_error("dyn_getter: 16-B alignment");
    if (P <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int nb = (P + 1023) / 1024;
    if (nb > rdg_getter_grid_cap()) nb = rdg_getter_grid_cap();
    rdg_stage_begin(RDG_STAGE_DEFORM_FWD, st);
    hipLaunchKernelGGL(rdg_dyn_getter_fwd_kernel, dim3(nb), dim3(1024), (size_t)Tu * RDG_DC_STRIDE * 4, st, P, Tu, coeff,
                       (const long long*)time_ind, bases, spatial_scale, xyz, scaling, rotation, opacity, means3D, scales,
                       rots, opac);
    rdg_stage_end(RDG_STAGE_DEFORM_FWD, st);
    return rdg_check_hip(hipGetLastError(), "dyn_getter_fwd launch");
}

int rdg_dyn_getter_backward(int32_t P, int32_t Tu, const float* coeff, const int64_t* time_ind, const float* bases,
                            float spatial_scale, const float* scaling, const float* rotation, const float* opacity,
                            const float* g_means3D, const float* g_scales, const float* g_rots, const float* g_opac,
                            float* d_xyz, float* d_scaling, float* d_rotation, float* d_opacity, float* d_coeff,
                            float* d_bases, const int32_t* order, const int32_t* inv_order, const int32_t* seg_start,
                            void* sorted_ws, void* stream) {
    if (!rdg_dyn_getter_supported(16, Tu)) return rdg_set_error("dyn_getter: unsupported table size Tu = %d", Tu);
    if (!order || !inv_order || !sorted_ws || (((uintptr_t)sorted_ws) & 15))
        return rdg_set_error("dyn_getter_backward needs order, inv_order and a 16-B aligned sorted workspace");
    if ((((uintptr_t)coeff | (uintptr_t)rotation | (uintptr_t)d_rotation | (uintptr_t)d_coeff |
          (uintptr_t)(g_rots ? g_rots : coeff))) & 15)
        return rdg_set_error("dyn_getter: 16-B alignment");
    hipStream_t st = (hipStream_t)stream;
    float* d_table = d_bases;
    float* d_basis_t = d_bases + (size_t)Tu * 112;
    rdg_stage_begin(RDG_STAGE_DEFORM_BWD, st);
    if (P <= 0) {
        hipError_t e = rdg_zero_async(d_bases, (size_t)(Tu + 1) * 112 * 4, st);
        if (e != hipSuccess) return rdg_check_hip(e, "dyn_getter_bwd memset");
    }
    if (P > 0) {
        int nb = (P + 1023) / 1024;
        if (nb > rdg_getter_grid_cap()) nb = rdg_getter_grid_cap();
        uint32_t* counter = (uint32_t*)((char*)sorted_ws + rdg_deform_gs_bytes(P));
        // once per device of this process (the attribute belongs to the function ON a device)
        static bool attr_set[64] = {};
        int dev_id = 0;
        (void)hipGetDevice(&dev_id);
        if (dev_id < 0 || dev_id >= 64 || !attr_set[dev_id]) {
            const hipError_t attr = rdg_getter_lds_attr((const void*)rdg_dyn_getter_bwd_kernel);
            if (attr != hipSuccess) return rdg_check_hip(attr, "dyn_getter_bwd LDS attribute");
            if (dev_id >= 0 && dev_id < 64) attr_set[dev_id] = true;
        }
        hipLaunchKernelGGL(rdg_dyn_getter_bwd_kernel, dim3(nb), dim3(1024), rdg_getter_lds(Tu), st, P, Tu,
                           (const long long*)time_ind, bases, spatial_scale, scaling, rotation, opacity, g_means3D,
                           g_scales, g_rots, g_opac, d_xyz, d_scaling, d_rotation, d_opacity, d_coeff,
                           (const int*)inv_order, (float4*)sorted_ws, counter, 2, coeff);
        float* part = (float*)((char*)counter + 256);
#ifndef RDG_ABL_FUSE_DB
        hipLaunchKernelGGL(rdg_deform_bwd_acc_mfma_kernel, dim3(RDG_DEF_ACC_BLOCKS), dim3(256), 0, st, P, coeff,
                           (const long long*)time_ind, (const int*)order, (const float*)nullptr, (const float*)nullptr,
                           spatial_scale, 1, d_basis_t, d_table, (const float*)sorted_ws, part);
        hipLaunchKernelGGL(rdg_deform_part_finalize_kernel, dim3(Tu), dim3(128 * RDG_DEF_FIN_GROUPS), 0, st, P, Tu,
                           rdg_deform_rows_per_wave(P, RDG_DEF_ACC_BLOCKS * 4), (const int*)seg_start,
                           (const float*)sorted_ws, (const long long*)time_ind, (const int*)order, (const float*)part,
                           d_table, d_basis_t, counter);
#else
        (void)part; (void)order; (void)seg_start;
        // (a zero basis gradient instead of the reduction: the step stays finite, the timing is what is read)
        if (rdg_zero_async(d_bases, (size_t)(Tu + 1) * 112 * 4, st) != hipSuccess) return rdg_set_error("ablation memset");
#endif
    }
    rdg_stage_end(RDG_STAGE_DEFORM_BWD, st);
    return rdg_check_hip(hipGetLastError(), "dyn_getter_bwd launch");
}

int rdg_dyn_getter_views_supported(int32_t B, int32_t Tu, int32_t nviews) {
    return B == 16 && Tu >= 1 && nviews >= 1 && nviews <= RDG_MAX_VIEWS &&
           ((size_t)Tu * RDG_DC_STRIDE + (size_t)nviews * 112) * 4 <= 64 * 1024;
}

int rdg_dyn_getter_views_forward(int32_t P, int32_t Tu, int32_t nviews, int32_t stride_rows, const float* coeff,
                                 const int64_t* time_ind, const float* bases_all, float spatial_scale, const float* xyz,
                                 const float* scaling, const float* rotation, const float* opacity, float* means3D,
                                 float* scales, float* rots, float* opac, void* stream) {
    if (!rdg_dyn_getter_views_supported(16, Tu, nviews))
        return rdg_set_error("dyn_getter_views: unsupported sizes Tu = %d, nviews = %d", Tu, nviews);
    if (stride_rows < P) return rdg_set_error("dyn_getter_views: stride_rows < P");
    if ((((uintptr_t)coeff | (uintptr_t)rotation | (uintptr_t)rots)) & 15) return rdg_set_error("dyn_getter_views: 16-B alignment");
    if (P <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int nb = (P + 511) / 512;
    if (nb > 768) nb = 768;
    const size_t lds = ((size_t)Tu * RDG_DC_STRIDE + (size_t)nviews * 112) * 4;
    rdg_stage_begin(RDG_STAGE_DEFORM_FWD, st);
    hipLaunchKernelGGL(rdg_dyn_getter_views_fwd_kernel, dim3(nb), dim3(512), lds, st, P, Tu, nviews, stride_rows, coeff,
                       (const long long*)time_ind, bases_all, spatial_scale, xyz, scaling, rotation, opacity, means3D,
                       scales, rots, opac);
    rdg_stage_end(RDG_STAGE_DEFORM_FWD, st);
    return rdg_check_hip(hipGetLastError(), "dyn_getter_views_fwd launch");
}

int rdg_dyn_getter_views_backward(int32_t P, int32_t Tu, int32_t nviews, int32_t stride_rows, const float* coeff,
                                  const int64_t* time_ind, const float* bases_all, float spatial_scale,
                                  const float* scaling, const float* rotation, const float* opacity,
                                  const float* g_means3D, const float* g_scales, const float* g_rots,
                                  const float* g_opac, float* d_xyz, float* d_scaling, float* d_rotation,
                                  float* d_opacity, float* d_coeff, float* d_bases_all, const int32_t* order,
                                  const int32_t* inv_order, void* sorted_ws, void* stream) {
    if (!rdg_dyn_getter_views_supported(16, Tu, nviews))
        return rdg_set_error("dyn_getter_views: unsupported sizes Tu = %d, nviews = %d", Tu, nviews);
    if (!order || !inv_order || !sorted_ws || (((uintptr_t)sorted_ws) & 15))
        return rdg_set_error("dyn_getter_views_backward needs order, inv_order and a 16-B aligned sorted workspace");
    if (!g_means3D || !g_scales || !g_rots || !g_opac) return rdg_set_error("dyn_getter_views_backward: NULL gradient");
    if ((((uintptr_t)coeff | (uintptr_t)rotation | (uintptr_t)d_rotation | (uintptr_t)d_coeff | (uintptr_t)g_rots)) & 15)
        return rdg_set_error("dyn_getter_views: 16-B alignment");
    hipStream_t st = (hipStream_t)stream;
    rdg_stage_begin(RDG_STAGE_DEFORM_BWD, st);
    hipError_t e = rdg_zero_async(d_bases_all, (size_t)nviews * (Tu + 1) * 112 * 4, st);
    if (e != hipSuccess) return rdg_check_hip(e, "dyn_getter_views_bwd memset");
    if (P > 0) {
        int nb = (P + 511) / 512;
        if (nb > 768) nb = 768;
        const size_t lds = ((size_t)Tu * RDG_DC_STRIDE + (size_t)nviews * 112) * 4;
        hipLaunchKernelGGL(rdg_dyn_getter_views_bwd_kernel, dim3(nb), dim3(512), lds, st, P, Tu, nviews, stride_rows,
                           (const long long*)time_ind, bases_all, spatial_scale, scaling, rotation, opacity, g_means3D,
                           g_scales, g_rots, g_opac, d_xyz, d_scaling, d_rotation, d_opacity, d_coeff,
                           (const int*)inv_order, (float4*)sorted_ws);
        const int npair = (nviews + 1) / 2;
        // ~128 birth-sorted Gaussians per wave (rocprofv3 at 125 k Gaussians x 8 views: 128 workgroups 98 us, 256: 52 us,
        // 1024: 57 us -- fewer leaves the gathers latency-bound, more multiplies the flush atomics)
        static int accv_grid = -1;
        if (accv_grid < 0) { const char* ev = getenv("RDG_ACCV_GRID"); accv_grid = ev ? atoi(ev) : 0; }
        int accv = accv_grid > 0 ? accv_grid : (P + 511) / 512;
        if (accv_grid <= 0) accv = accv < 256 ? 256 : (accv > 2048 ? 2048 : accv);
        const dim3 grid(accv), block(256);
        if (npair <= 1)
            hipLaunchKernelGGL(rdg_deform_bwd_acc_views_kernel<1>, grid, block, 0, st, P, nviews, Tu, coeff,
                               (const int*)order, d_bases_all, (const float*)sorted_ws);
        else if (npair <= 2)
            hipLaunchKernelGGL(rdg_deform_bwd_acc_views_kernel<2>, grid, block, 0, st, P, nviews, Tu, coeff,
                               (const int*)order, d_bases_all, (const float*)sorted_ws);
        else if (npair <= 4)
            hipLaunchKernelGGL(rdg_deform_bwd_acc_views_kernel<4>, grid, block, 0, st, P, nviews, Tu, coeff,
                               (const int*)order, d_bases_all, (const float*)sorted_ws);
        else
            hipLaunchKernelGGL(rdg_deform_bwd_acc_views_kernel<8>, grid, block, 0, st, P, nviews, Tu, coeff,
                               (const int*)order, d_bases_all, (const float*)sorted_ws);
        hipLaunchKernelGGL(rdg_deform_dbt_views_kernel, dim3(1, nviews), dim3(128), 0, st, Tu, d_bases_all);
    }
    rdg_stage_end(RDG_STAGE_DEFORM_BWD, st);
    return rdg_check_hip(hipGetLastError(), "dyn_getter_views_bwd launch");
}

static int rdg_adam_launch(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int row_len,
                           int head_len, float lr_head, float lr_tail, double beta1, double beta2, float eps, int32_t step,
                           void* stream) {
    if (n <= 0) return 0;
    if (step < 1) return rdg_set_error("adam: step must be >= 1");
    hipStream_t st = (hipStream_t)stream;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    long long blocks = ((n >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    rdg_stage_begin(RDG_STAGE_ADAM, st);
    hipLaunchKernelGGL(rdg_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (long long)n, param, grad, exp_avg,
                       exp_avg_sq, (float)(lr_head / bc1), (float)(lr_tail / bc1), row_len, head_len, (float)beta1, (float)beta2,
                       (float)(1.0 - beta1), (float)(1.0 - beta2), eps,
                       (float)sqrt(bc2));
    rdg_stage_end(RDG_STAGE_ADAM, st);
    return rdg_check_hip(hipGetLastError(), "adam launch");
}

int rdg_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, double beta1,
                  double beta2, float eps, int32_t step, void* stream) {
    return rdg_adam_launch(n, param, grad, exp_avg, exp_avg_sq, 1, 1, lr, lr, beta1, beta2, eps, step, stream);
}

static int rdg_adam_multi_launch(int32_t nseg, const RdgAdamSeg* segs_host, double beta1, double beta2, float eps,
                                 int32_t step, const RdgStepScalars* dev, void* stream) {
    if (nseg <= 0) return 0;
    if (nseg > RDG_ADAM_MAX_SEGS) return rdg_set_error("adam: at most %d segments per launch", RDG_ADAM_MAX_SEGS);
    if (!dev && step < 1) return rdg_set_error("adam: step must be >= 1");
    hipStream_t st = (hipStream_t)stream;
    RdgAdamSegs segs;
    long long nmax = 0;
    for (int i = 0; i < nseg; ++i) {
        segs.s[i] = segs_host[i];
        if (segs.s[i].row_len < 1) segs.s[i].row_len = 1;
        if (segs.s[i].n > nmax) nmax = segs.s[i].n;
    }
    const double bc1 = dev ? 1.0 : 1.0 - pow(beta1, (double)step);
    const double bc2 = dev ? 1.0 : 1.0 - pow(beta2, (double)step);
    // measured at 75 M parameters (scripts/adam_probe.py): streaming (nontemporal) loads/stores + 16 k workgroups per
    // segment 367 us = 5.7 TB/s; cached accesses + 2 k workgroups 416 us.  RDG_ADAM_VAR=0 / RDG_ADAM_BLOCKS override.
    static int var = -1, cap = 16384;
    if (var < 0) {
        const char* ev = getenv("RDG_ADAM_VAR"); var = ev ? atoi(ev) : 1;
        const char* ec = getenv("RDG_ADAM_BLOCKS"); if (ec) cap = atoi(ec);
    }
    long long blocks = ((nmax >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > cap) blocks = cap;
    rdg_stage_begin(RDG_STAGE_ADAM, st);
    if (var & 1)
        hipLaunchKernelGGL(rdg_adam_multi_kernel<1>, dim3((unsigned)blocks, nseg), dim3(256), 0, st, segs,
                           (float)(1.0 / bc1), (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps,
                           (float)sqrt(bc2), dev);
    else
        hipLaunchKernelGGL(rdg_adam_multi_kernel<0>, dim3((unsigned)blocks, nseg), dim3(256), 0, st, segs,
                           (float)(1.0 / bc1), (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps,
                           (float)sqrt(bc2), dev);
    rdg_stage_end(RDG_STAGE_ADAM, st);
    return rdg_check_hip(hipGetLastError(), "adam multi launch");
}

int rdg_adam_step_multi(int32_t nseg, const RdgAdamSeg* segs_host, double beta1, double beta2, float eps, int32_t step,
                        void* stream) {
    return rdg_adam_multi_launch(nseg, segs_host, beta1, beta2, eps, step, nullptr, stream);
}

int rdg_adam_step_multi_dev(int32_t nseg, const RdgAdamSeg* segs_host, double beta1, double beta2, float eps,
                            const RdgStepScalars* dev, void* stream) {
    if (!dev) return rdg_set_error("rdg_adam_step_multi_dev: NULL step scalars");
    return rdg_adam_multi_launch(nseg, segs_host, beta1, beta2, eps, 0, dev, stream);
}

int rdg_adam_step_rows(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int32_t row_len,
                       int32_t head_len, float lr_head, float lr_tail, double beta1, double beta2, float eps, int32_t step,
                       void* stream) {
    if (row_len < 1 || head_len < 0 || head_len > row_len) return rdg_set_error("adam: bad row structure");
    return rdg_adam_launch(n, param, grad, exp_avg, exp_avg_sq, row_len, head_len, lr_head, lr_tail, beta1, beta2, eps,
                           step, stream);
}

}  // extern "C"
