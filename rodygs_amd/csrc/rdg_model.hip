// rdg_model.hip -- the per-Gaussian glue that sits between the optimiser's raw parameters and the rasterizer
// (SURVEY.md §8a row a11 and §8b "camera conventions"), fused so the train step does not pay ~150 tiny framework
// kernels for it:
//   rdg_activate_forward/backward : StaticRoDyGS.get_xyz / get_scaling / get_rotation / get_opacity / get_features
//       (/root/reference/src/model/rodygs_static.py:82-105) + the deformation add of get_GS_properties
//       (/root/reference/src/trainer/rodygs.py:68-113):  means3D = xyz + dxyz, scales = exp(s),
//       rots = normalize(q) + dq (NOT re-normalised), opacity = sigmoid(o),
//       shs = cat(f_dc, f_rest) -- the concat is one flat coalesced copy instead of a cat + split-copies.
//   rdg_pose_view_forward/backward: FixedCameraTorch.world_view_transform
//       (/root/reference/src/data/utils.py:161-170, quaternion_to_matrix graphic_utils.py:76-102) for one frame of
//       the learnable [T,4] quaternion / [T,3] translation tables, returned in glm storage (W2C^T).
// All HBM-bound elementwise work; one thread per Gaussian (or per float for the SH copy).
#include "rdg_common.h"

__global__ void __launch_bounds__(256)
rdg_activate_fwd_kernel(int P, const float* __restrict__ xyz, const float* __restrict__ dxyz,
                        const float* __restrict__ scaling, const float* __restrict__ rotation,
                        const float* __restrict__ drot, const float* __restrict__ opacity, float* __restrict__ means3D,
                        float* __restrict__ scales, float* __restrict__ rots, float* __restrict__ opac) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        means3D[3 * p + c] = xyz[3 * p + c] + (dxyz ? dxyz[3 * p + c] : 0.0f);
        scales[3 * p + c] = __expf(scaling[3 * p + c]);
    }
    const float4 q = reinterpret_cast<const float4*>(rotation)[p];
    const float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);  // F.normalize eps
    const float inv = 1.0f / n;
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (drot) d = reinterpret_cast<const float4*>(drot)[p];
    reinterpret_cast<float4*>(rots)[p] = make_float4(q.x * inv + d.x, q.y * inv + d.y, q.z * inv + d.z, q.w * inv + d.w);
    opac[p] = 1.0f / (1.0f + __expf(-opacity[p]));
}

// shs[p][k][c], k = 0 from f_dc[p][0][c], k >= 1 from f_rest[p][k-1][c]; one float per thread-iteration
__global__ void __launch_bounds__(256)
rdg_sh_concat_kernel(long long n, int row, const float* __restrict__ f_dc, const float* __restrict__ f_rest,
                     float* __restrict__ shs) {
    const int rest = row - 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long p = i / row;
        const int r = (int)(i - p * row);
        shs[i] = r < 3 ? f_dc[p * 3 + r] : f_rest[p * rest + (r - 3)];
    }
}
__global__ void __launch_bounds__(256)
rdg_sh_split_kernel(long long n, int row, const float* __restrict__ g_shs, float* __restrict__ d_fdc,
                    float* __restrict__ d_frest) {
    const int rest = row - 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long p = i / row;
        const int r = (int)(i - p * row);
        const float v = g_shs[i];
        if (r < 3) d_fdc[p * 3 + r] = v; else d_frest[p * rest + (r - 3)] = v;
    }
}

__global__ void __launch_bounds__(256)
rdg_activate_bwd_kernel(int P, const float* __restrict__ scaling, const float* __restrict__ rotation,
                        const float* __restrict__ opacity, const float* __restrict__ g_means3D,
                        const float* __restrict__ g_scales, const float* __restrict__ g_rots,
                        const float* __restrict__ g_opac, float* __restrict__ d_xyz, float* __restrict__ d_scaling,
                        float* __restrict__ d_rotation, float* __restrict__ d_opacity) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        d_xyz[3 * p + c] = g_means3D ? g_means3D[3 * p + c] : 0.0f;
        d_scaling[3 * p + c] = g_scales ? g_scales[3 * p + c] * __expf(scaling[3 * p + c]) : 0.0f;
    }
    float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g_rots) {
        const float4 q = reinterpret_cast<const float4*>(rotation)[p];
        const float4 g = reinterpret_cast<const float4*>(g_rots)[p];
        const float nn = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
        if (nn > 1e-12f) {
            const float inv = 1.0f / nn;
            const float yx = q.x * inv, yy = q.y * inv, yz = q.z * inv, yw = q.w * inv;
            const float dot = yx * g.x + yy * g.y + yz * g.z + yw * g.w;
            dq = make_float4((g.x - yx * dot) * inv, (g.y - yy * dot) * inv, (g.z - yz * dot) * inv, (g.w - yw * dot) * inv);
        } else {
            dq = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);  // clamp_min branch of F.normalize
        }
    }
    reinterpret_cast<float4*>(d_rotation)[p] = dq;
    const float s = 1.0f / (1.0f + __expf(-opacity[p]));
    d_opacity[p] = g_opac ? g_opac[p] * s * (1.0f - s) : 0.0f;
}

// ---- pose -> viewmatrix ------------------------------------------------------------------------------------------
__device__ __forceinline__ void rdg_quat_to_R(const float q[4], float R[9], float& two_s) {
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    two_s = 2.0f / (r * r + i * i + j * j + k * k);
    R[0] = 1.f - two_s * (j * j + k * k); R[1] = two_s * (i * j - k * r); R[2] = two_s * (i * k + j * r);
    R[3] = two_s * (i * j + k * r); R[4] = 1.f - two_s * (i * i + k * k); R[5] = two_s * (j * k - i * r);
    R[6] = two_s * (i * k - j * r); R[7] = two_s * (j * k + i * r); R[8] = 1.f - two_s * (i * i + j * j);
}

__device__ void rdg_pose_view_row(int frame, const float* __restrict__ cam_q, const float* __restrict__ cam_t,
                                  float* __restrict__ view) {
    float q[4], t[3], R[9], s;
    for (int c = 0; c < 4; ++c) q[c] = cam_q[4 * frame + c];
    for (int c = 0; c < 3; ++c) t[c] = cam_t[3 * frame + c];
    rdg_quat_to_R(q, R, s);
    // W2C[r][c] = R[c][r] (r,c < 3);  W2C[r][3] = -sum_c R[c][r] t[c];  glm storage: view[c*4+r] = W2C[r][c]
    for (int r = 0; r < 3; ++r) {
        float tw = 0.f;
        for (int c = 0; c < 3; ++c) { view[c * 4 + r] = R[c * 3 + r]; tw += R[c * 3 + r] * t[c]; }
        view[12 + r] = -tw;
        view[r * 4 + 3] = 0.0f;
    }
    view[15] = 1.0f;
}

// dev != nullptr: the frame index comes from device memory (graph replay), clamped to the table
__global__ void rdg_pose_view_fwd_kernel(int T, int frame, const RdgStepScalars* __restrict__ dev,
                                         const float* __restrict__ cam_q, const float* __restrict__ cam_t,
                                         float* __restrict__ view) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (dev) frame = min(max(dev->frame, 0), T - 1);
    rdg_pose_view_row(frame, cam_q, cam_t, view);
}

// several frames of one step (Gaussian-sharded frame-DP: every rank needs the cameras of all ranks)
struct RdgFrames { int32_t f[RDG_MAX_VIEWS]; };
__global__ void rdg_pose_views_fwd_kernel(int nviews, RdgFrames fr, const float* __restrict__ cam_q,
                                          const float* __restrict__ cam_t, float* __restrict__ views) {
    if (threadIdx.x < nviews) rdg_pose_view_row(fr.f[threadIdx.x], cam_q, cam_t, views + 16 * threadIdx.x);
}

template <bool ADD>
__device__ void rdg_pose_view_bwd_row(int frame, const float* __restrict__ cam_q, const float* __restrict__ cam_t,
                                      const float* __restrict__ g_view, float* __restrict__ d_q,
                                      float* __restrict__ d_t) {
    float q[4], t[3], R[9], s;
    for (int c = 0; c < 4; ++c) q[c] = cam_q[4 * frame + c];
    for (int c = 0; c < 3; ++c) t[c] = cam_t[3 * frame + c];
    rdg_quat_to_R(q, R, s);
    // G[r][c] = dL/dW2C[r][c] = g_view[c*4+r];  dL/dR[c][r] = G[r][c] - G[r][3] t[c];  dL/dt[c] = -sum_r R[c][r] G[r][3]
    float dR[9];
    for (int c = 0; c < 3; ++c) {
        float dt = 0.f;
        for (int r = 0; r < 3; ++r) {
            dR[c * 3 + r] = g_view[c * 4 + r] - g_view[12 + r] * t[c];
            dt += R[c * 3 + r] * g_view[12 + r];
        }
        d_t[3 * frame + c] = (ADD ? d_t[3 * frame + c] : 0.0f) - dt;
    }
    // R = I + two_s * M(q):  dL/dq_m = two_s * sum dR_ab dM_ab/dq_m + (sum dR_ab M_ab) * d(two_s)/dq_m
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float M[9] = {-(j * j + k * k), i * j - k * r, i * k + j * r, i * j + k * r, -(i * i + k * k), j * k - i * r,
                        i * k - j * r, j * k + i * r, -(i * i + j * j)};
    float dotM = 0.f;
    for (int a = 0; a < 9; ++a) dotM += dR[a] * M[a];
    const float n2 = r * r + i * i + j * j + k * k;
    const float ds_scale = -2.0f * s / n2;  // d(two_s)/dq_m = -2 * two_s * q_m / |q|^2
    const float gr = -k * dR[1] + j * dR[2] + k * dR[3] - i * dR[5] - j * dR[6] + i * dR[7];
    const float gi = j * dR[1] + k * dR[2] + j * dR[3] - 2.f * i * dR[4] - r * dR[5] + k * dR[6] + r * dR[7] - 2.f * i * dR[8];
    const float gj = -2.f * j * dR[0] + i * dR[1] + r * dR[2] + i * dR[3] + k * dR[5] - r * dR[6] + k * dR[7] - 2.f * j * dR[8];
    const float gk = -2.f * k * dR[0] - r * dR[1] + i * dR[2] + r * dR[3] - 2.f * k * dR[4] + j * dR[5] + i * dR[6] + j * dR[7];
    const float o0 = ADD ? d_q[4 * frame + 0] : 0.0f, o1 = ADD ? d_q[4 * frame + 1] : 0.0f;
    const float o2 = ADD ? d_q[4 * frame + 2] : 0.0f, o3 = ADD ? d_q[4 * frame + 3] : 0.0f;
    d_q[4 * frame + 0] = o0 + (s * gr + dotM * ds_scale * r);
    d_q[4 * frame + 1] = o1 + (s * gi + dotM * ds_scale * i);
    d_q[4 * frame + 2] = o2 + (s * gj + dotM * ds_scale * j);
    d_q[4 * frame + 3] = o3 + (s * gk + dotM * ds_scale * k);
}

__global__ void rdg_pose_view_bwd_kernel(int T, int frame, const RdgStepScalars* __restrict__ dev,
                                         const float* __restrict__ cam_q,
                                         const float* __restrict__ cam_t, const float* __restrict__ g_view,
                                         float* __restrict__ d_q, float* __restrict__ d_t) {
    if (dev) frame = min(max(dev->frame, 0), T - 1);
    // zero the other frames' rows, then thread 0 writes the rendered frame's row
    for (int k = threadIdx.x; k < T * 4; k += blockDim.x) if (k / 4 != frame) d_q[k] = 0.0f;
    for (int k = threadIdx.x; k < T * 3; k += blockDim.x) if (k / 3 != frame) d_t[k] = 0.0f;
    if (threadIdx.x != 0) return;
    rdg_pose_view_bwd_row<false>(frame, cam_q, cam_t, g_view, d_q, d_t);
}

// zero both tables, then one thread adds the rows of the step's frames in order (a frame may repeat)
__global__ void rdg_pose_views_bwd_kernel(int T, int nviews, RdgFrames fr, const float* __restrict__ cam_q,
                                          const float* __restrict__ cam_t, const float* __restrict__ g_views,
                                          float* __restrict__ d_q, float* __restrict__ d_t) {
    for (int k = threadIdx.x; k < T * 4; k += blockDim.x) d_q[k] = 0.0f;
    for (int k = threadIdx.x; k < T * 3; k += blockDim.x) d_t[k] = 0.0f;
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int v = 0; v < nviews; ++v) rdg_pose_view_bwd_row<true>(fr.f[v], cam_q, cam_t, g_views + 16 * v, d_q, d_t);
}

extern "C" {

int rdg_activate_forward(int32_t P, int32_t K, const float* xyz, const float* dxyz, const float* scaling,
                         const float* rotation, const float* drot, const float* opacity, const float* f_dc,
                         const float* f_rest, float* out_means3D, float* out_scales, float* out_rots, float* out_opac,
                         float* out_shs, void* stream) {
    if (P <= 0) return 0;
    if (K < 1) return rdg_set_error("activate: K must be >= 1");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(rdg_activate_fwd_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, xyz, dxyz, scaling,
                       rotation, drot, opacity, out_means3D, out_scales, out_rots, out_opac);
    const long long n = (long long)P * K * 3;
    if (f_dc && out_shs)   // features kept as one [P,K,3] tensor by the caller: nothing to concatenate
        hipLaunchKernelGGL(rdg_sh_concat_kernel, dim3(4096), dim3(256), 0, st, n, K * 3, f_dc, f_rest, out_shs);
    return rdg_check_hip(hipGetLastError(), "activate_fwd launch");
}

int rdg_activate_backward(int32_t P, int32_t K, const float* scaling, const float* rotation, const float* opacity,
                          const float* g_means3D, const float* g_scales, const float* g_rots, const float* g_opac,
                          const float* g_shs, float* d_xyz, float* d_scaling, float* d_rotation, float* d_opacity,
                          float* d_fdc, float* d_frest, void* stream) {
    if (P <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(rdg_activate_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, scaling, rotation, opacity,
                       g_means3D, g_scales, g_rots, g_opac, d_xyz, d_scaling, d_rotation, d_opacity);
    const long long n = (long long)P * K * 3;
    if (!d_fdc) {
    } else if (g_shs) {
        hipLaunchKernelGGL(rdg_sh_split_kernel, dim3(4096), dim3(256), 0, st, n, K * 3, g_shs, d_fdc, d_frest);
    } else {
        hipError_t e = rdg_zero_async(d_fdc, (size_t)P * 3 * 4, st);
        if (e == hipSuccess && K > 1) e = rdg_zero_async(d_frest, (size_t)P * (K - 1) * 3 * 4, st);
        if (e != hipSuccess) return rdg_check_hip(e, "activate_bwd memset");
    }
    return rdg_check_hip(hipGetLastError(), "activate_bwd launch");
}

int rdg_pose_view_forward(int32_t T, int32_t frame, const float* cam_q, const float* cam_t, float* out_view16,
                          void* stream) {
    if (frame < 0 || frame >= T) return rdg_set_error("pose: frame %d out of range [0,%d)", frame, T);
    hipLaunchKernelGGL(rdg_pose_view_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, T, frame,
                       (const RdgStepScalars*)nullptr, cam_q, cam_t, out_view16);
    return rdg_check_hip(hipGetLastError(), "pose_view_fwd launch");
}

int rdg_pose_view_forward_dev(int32_t T, const RdgStepScalars* dev, const float* cam_q, const float* cam_t,
                              float* out_view16, void* stream) {
    if (!dev || T < 1) return rdg_set_error("pose_view_forward_dev: NULL step scalars / empty table");
    hipLaunchKernelGGL(rdg_pose_view_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, T, 0, dev, cam_q, cam_t,
                       out_view16);
    return rdg_check_hip(hipGetLastError(), "pose_view_fwd launch");
}

int rdg_pose_view_backward(int32_t T, int32_t frame, const float* cam_q, const float* cam_t, const float* g_view16,
                           float* d_q, float* d_t, void* stream) {
    if (frame < 0 || frame >= T) return rdg_set_error("pose: frame %d out of range [0,%d)", frame, T);
    hipLaunchKernelGGL(rdg_pose_view_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, T, frame,
                       (const RdgStepScalars*)nullptr, cam_q, cam_t, g_view16, d_q, d_t);
    return rdg_check_hip(hipGetLastError(), "pose_view_bwd launch");
}

int rdg_pose_view_backward_dev(int32_t T, const RdgStepScalars* dev, const float* cam_q, const float* cam_t,
                               const float* g_view16, float* d_q, float* d_t, void* stream) {
    if (!dev || T < 1) return rdg_set_error("pose_view_backward_dev: NULL step scalars / empty table");
    hipLaunchKernelGGL(rdg_pose_view_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, T, 0, dev, cam_q, cam_t,
                       g_view16, d_q, d_t);
    return rdg_check_hip(hipGetLastError(), "pose_view_bwd launch");
}

static int rdg_frames_arg(int32_t T, int32_t nviews, const int32_t* frames_host, RdgFrames* fr) {
    if (nviews < 1 || nviews > RDG_MAX_VIEWS) return rdg_set_error("pose: nviews must be 1..%d", RDG_MAX_VIEWS);
    for (int v = 0; v < nviews; ++v) {
        if (frames_host[v] < 0 || frames_host[v] >= T)
            return rdg_set_error("pose: frame %d out of range [0,%d)", frames_host[v], T);
        fr->f[v] = frames_host[v];
    }
    return 0;
}

int rdg_pose_views_forward(int32_t T, int32_t nviews, const int32_t* frames_host, const float* cam_q,
                           const float* cam_t, float* out_views, void* stream) {
    RdgFrames fr;
    if (rdg_frames_arg(T, nviews, frames_host, &fr)) return -1;
    hipLaunchKernelGGL(rdg_pose_views_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, nviews, fr, cam_q, cam_t,
                       out_views);
    return rdg_check_hip(hipGetLastError(), "pose_views_fwd launch");
}

int rdg_pose_views_backward(int32_t T, int32_t nviews, const int32_t* frames_host, const float* cam_q,
                            const float* cam_t, const float* g_views, float* d_q, float* d_t, void* stream) {
    RdgFrames fr;
    if (rdg_frames_arg(T, nviews, frames_host, &fr)) return -1;
    hipLaunchKernelGGL(rdg_pose_views_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, T, nviews, fr, cam_q,
                       cam_t, g_views, d_q, d_t);
    return rdg_check_hip(hipGetLastError(), "pose_views_bwd launch");
}

}  // extern "C"
