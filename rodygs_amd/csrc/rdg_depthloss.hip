// rdg_depthloss.hip -- Pearson-correlation depth losses.
//
// Reference: pearson_depth_loss (/root/reference/src/utils/loss_utils.py:100-117), GlobalPearsonDepthLoss and
// LocalPearsonDepthLoss (/root/reference/src/trainer/losses.py:108-182).  For a box of n pixels with
// a = pred * mask, b = gt * mask:
//     loss = 1 - mean( (a - mean a) / (std a + eps) * (b - mean b) / (std b + eps) ),  std unbiased (n - 1).
// The local loss is the mean over ~60 random 128x128 boxes; the reference walks them in a Python loop of small
// kernels with a host synchronisation per box.  Here all boxes of one call (same size, corners in device memory)
// take three launches: per-box sums (f64 accumulation: one pass, no cancellation problem), a finalize kernel that
// turns the sums into the loss and into the two coefficients of the gradient,
//     dL/da_i = alpha (b_i - mean b) + beta (a_i - mean a),
// and a backward kernel that adds those into the depth-image gradient (boxes overlap: float atomics over
// contiguous row segments).  A box whose mask is empty is skipped, as the reference does (losses.py:155-163).
#include "rdg_common.h"

#define RDG_PD_NSTAT 6      // sum a, a^2, b, b^2, ab, mask
#define RDG_PD_CHUNK 8192   // pixels per workgroup

__device__ __forceinline__ double rdg_wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void __launch_bounds__(256)
rdg_pearson_stats_kernel(int H, int W, int bh, int bw, const int64_t* __restrict__ row0, const int64_t* __restrict__ col0,
                         const float* __restrict__ pred, const float* __restrict__ gt,
                         const uint8_t* __restrict__ mask, double* __restrict__ stats) {
    const int box = blockIdx.x;
    const long long r0 = row0 ? row0[box] : 0, c0 = col0 ? col0[box] : 0;
    const int n = bh * bw;
    const int beg = blockIdx.y * RDG_PD_CHUNK, end = min(n, beg + RDG_PD_CHUNK);
    double s[RDG_PD_NSTAT] = {0, 0, 0, 0, 0, 0};
    for (int k = beg + threadIdx.x; k < end; k += 256) {
        const int r = k / bw, c = k - r * bw;
        const size_t pix = (size_t)(r0 + r) * W + (size_t)(c0 + c);
        const float m = mask ? (mask[pix] ? 1.0f : 0.0f) : 1.0f;
        const double a = (double)(pred[pix] * m), b = (double)(gt[pix] * m);
        s[0] += a; s[1] += a * a; s[2] += b; s[3] += b * b; s[4] += a * b; s[5] += (double)m;
    }
    __shared__ double sh[4][RDG_PD_NSTAT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < RDG_PD_NSTAT; ++q) {
        const double t = rdg_wave_sum_f64(s[q]);
        if (lane == 0) sh[wv][q] = t;
    }
    __syncthreads();
    if (threadIdx.x < RDG_PD_NSTAT)
        atomicAdd(&stats[(size_t)box * RDG_PD_NSTAT + threadIdx.x],
                  (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]));
}

// coef[box] = (alpha, beta, mean a, mean b); loss_out[0] = weight * sum over boxes of (1 - corr)
__global__ void __launch_bounds__(256)
rdg_pearson_finalize_kernel(int n_boxes, int n, float eps, float weight, int has_mask, const double* __restrict__ stats,
                            float4* __restrict__ coef, float* __restrict__ loss_out) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int box = threadIdx.x; box < n_boxes; box += 256) {
        const double* s = stats + (size_t)box * RDG_PD_NSTAT;
        const double dn = (double)n;
        float4 cf = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(has_mask && s[5] == 0.0)) {
            const double ma = s[0] / dn, mb = s[2] / dn;
            const double va = fmax((s[1] - dn * ma * ma) / (dn - 1.0), 0.0), vb = fmax((s[3] - dn * mb * mb) / (dn - 1.0), 0.0);
            const double sa = sqrt(va), sb = sqrt(vb);
            const double cab = s[4] - dn * ma * mb;                     // sum (a - ma)(b - mb)
            const double da = sa + (double)eps, db = sb + (double)eps;
            acc += 1.0 - cab / (da * db) / dn;
            const double alpha = -(double)weight / (dn * da * db);
            const double beta = sa > 0.0 ? (double)weight * cab / (db * dn * (dn - 1.0) * sa * da * da) : 0.0;
            cf = make_float4((float)alpha, (float)beta, (float)ma, (float)mb);
        }
        coef[box] = cf;
    }
    acc = rdg_wave_sum_f64(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss_out[0] = (float)(((sh[0] + sh[1]) + (sh[2] + sh[3])) * (double)weight);
}

__global__ void __launch_bounds__(256)
rdg_pearson_bwd_kernel(int H, int W, int bh, int bw, const int64_t* __restrict__ row0, const int64_t* __restrict__ col0,
                       const float* __restrict__ pred, const float* __restrict__ gt, const uint8_t* __restrict__ mask,
                       const float4* __restrict__ coef, const float* __restrict__ g_loss, float* __restrict__ d_pred) {
    const int box = blockIdx.x;
    const float4 cf = coef[box];
    if (cf.x == 0.0f && cf.y == 0.0f) return;
    const float g = g_loss[0];
    const long long r0 = row0 ? row0[box] : 0, c0 = col0 ? col0[box] : 0;
    const int n = bh * bw;
    const int beg = blockIdx.y * RDG_PD_CHUNK, end = min(n, beg + RDG_PD_CHUNK);
    for (int k = beg + threadIdx.x; k < end; k += 256) {
        const int r = k / bw, c = k - r * bw;
        const size_t pix = (size_t)(r0 + r) * W + (size_t)(c0 + c);
        const float m = mask ? (mask[pix] ? 1.0f : 0.0f) : 1.0f;
        if (m == 0.0f) continue;
        const float a = pred[pix], b = gt[pix];
        atomicAdd(&d_pred[pix], g * (cf.x * (b - cf.w) + cf.y * (a - cf.z)));
    }
}

extern "C" {

size_t rdg_pearson_ws_bytes(int32_t n_boxes) {
    const size_t nb = (size_t)(n_boxes > 0 ? n_boxes : 1);
    return rdg_align_up(nb * RDG_PD_NSTAT * sizeof(double), 256) + rdg_align_up(nb * sizeof(float4), 256);
}

static int rdg_pearson_check(int32_t H, int32_t W, int32_t n_boxes, int32_t bh, int32_t bw) {
    if (H <= 0 || W <= 0 || n_boxes < 0 || bh < 2 || bw < 1 || bh > H || bw > W || (long long)bh * bw < 2)
        return rdg_set_error("pearson depth loss: bad sizes (H %d W %d boxes %d of %d x %d)", H, W, n_boxes, bh, bw);
    return 0;
}

int rdg_pearson_depth_forward(int32_t H, int32_t W, int32_t n_boxes, int32_t bh, int32_t bw, const int64_t* row0,
                              const int64_t* col0, const float* pred, const float* gt, const uint8_t* mask, float eps,
                              float weight, void* ws, float* loss_out, void* stream) {
    if (rdg_pearson_check(H, W, n_boxes, bh, bw)) return -1;
    hipStream_t st = (hipStream_t)stream;
    if (n_boxes == 0) {
        hipError_t e = rdg_zero_async(loss_out, 4, st);
        return rdg_check_hip(e, "pearson memset");
    }
    double* stats = (double*)ws;
    float4* coef = (float4*)((char*)ws + rdg_align_up((size_t)n_boxes * RDG_PD_NSTAT * sizeof(double), 256));
    hipError_t e = rdg_zero_async(stats, (size_t)n_boxes * RDG_PD_NSTAT * sizeof(double), st);
    if (e != hipSuccess) return rdg_check_hip(e, "pearson memset");
    const int chunks = (bh * bw + RDG_PD_CHUNK - 1) / RDG_PD_CHUNK;
    hipLaunchKernelGGL(rdg_pearson_stats_kernel, dim3(n_boxes, chunks), dim3(256), 0, st, H, W, bh, bw, row0, col0, pred,
                       gt, mask, stats);
    hipLaunchKernelGGL(rdg_pearson_finalize_kernel, dim3(1), dim3(256), 0, st, n_boxes, bh * bw, eps, weight,
                       mask ? 1 : 0, (const double*)stats, coef, loss_out);
    return rdg_check_hip(hipGetLastError(), "pearson forward launch");
}

int rdg_pearson_depth_backward(int32_t H, int32_t W, int32_t n_boxes, int32_t bh, int32_t bw, const int64_t* row0,
                               const int64_t* col0, const float* pred, const float* gt, const uint8_t* mask,
                               const void* ws, const float* g_loss, float* d_pred, void* stream) {
    if (rdg_pearson_check(H, W, n_boxes, bh, bw)) return -1;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = rdg_zero_async(d_pred, (size_t)H * W * 4, st);
    if (e != hipSuccess) return rdg_check_hip(e, "pearson bwd memset");
    if (n_boxes == 0) return 0;
    const float4* coef = (const float4*)((const char*)ws + rdg_align_up((size_t)n_boxes * RDG_PD_NSTAT * sizeof(double), 256));
    const int chunks = (bh * bw + RDG_PD_CHUNK - 1) / RDG_PD_CHUNK;
    hipLaunchKernelGGL(rdg_pearson_bwd_kernel, dim3(n_boxes, chunks), dim3(256), 0, st, H, W, bh, bw, row0, col0, pred, gt,
                       mask, coef, g_loss, d_pred);
    return rdg_check_hip(hipGetLastError(), "pearson backward launch");
}

}  // extern "C"
