// rdg_densify.hip -- row compaction / extension of the Gaussian tensors for densify-and-prune.
//
// Reference: ThreeDGSTrainer.densify_and_clone / densify_and_split / prune_points
// (/root/reference/src/trainer/rodygs_static.py:170-319) with the optimizer surgery of
// /root/reference/src/trainer/utils.py:15-95.  There every parameter and both Adam moments are rebuilt tensor by
// tensor with boolean indexing and torch.cat, three times per densification (clone, split, prune).  Here the three
// steps are composed into ONE source-row list on the host side (rodygs_amd/densify.py) and every buffer is rebuilt
// by one gather:  dst[i,:] = src[idx[i],:]  (idx < 0 -> zeros: the Adam moments of new Gaussians), plus one small
// kernel that places the split children: xyz = parent + R(q/|q|) (exp(s) * z), scaling = log(exp(s) / (0.8 N)).
#include "rdg_common.h"

__global__ void rdg_gather_rows_kernel(long long n_new, int row_len, const long long* __restrict__ idx,
                                       const float* __restrict__ src, float* __restrict__ dst) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_new * row_len) return;
    const long long r = e / row_len;
    const int c = (int)(e - r * row_len);
    const long long s = idx[r];
    dst[e] = s < 0 ? 0.0f : src[s * row_len + c];
}

__global__ void rdg_split_children_kernel(long long n, const long long* __restrict__ parent, float inv_shrink,
                                          const float* __restrict__ xyz, const float* __restrict__ scaling,
                                          const float* __restrict__ rotation, const float* __restrict__ z,
                                          float* __restrict__ xyz_out, float* __restrict__ scaling_out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long p = parent[i];
    const float s0 = expf(scaling[3 * p]), s1 = expf(scaling[3 * p + 1]), s2 = expf(scaling[3 * p + 2]);
    const float qr0 = rotation[4 * p], qx0 = rotation[4 * p + 1], qy0 = rotation[4 * p + 2], qz0 = rotation[4 * p + 3];
    const float nrm = sqrtf(qr0 * qr0 + qx0 * qx0 + qy0 * qy0 + qz0 * qz0);
    const float r = qr0 / nrm, x = qx0 / nrm, y = qy0 / nrm, w = qz0 / nrm;
    const float v0 = s0 * z[3 * i], v1 = s1 * z[3 * i + 1], v2 = s2 * z[3 * i + 2];   // sample ~ N(0, diag(s)^2)
    const float R00 = 1.f - 2.f * (y * y + w * w), R01 = 2.f * (x * y - r * w), R02 = 2.f * (x * w + r * y);
    const float R10 = 2.f * (x * y + r * w), R11 = 1.f - 2.f * (x * x + w * w), R12 = 2.f * (y * w - r * x);
    const float R20 = 2.f * (x * w - r * y), R21 = 2.f * (y * w + r * x), R22 = 1.f - 2.f * (x * x + y * y);
    xyz_out[3 * i] = (R00 * v0 + R01 * v1 + R02 * v2) + xyz[3 * p];
    xyz_out[3 * i + 1] = (R10 * v0 + R11 * v1 + R12 * v2) + xyz[3 * p + 1];
    xyz_out[3 * i + 2] = (R20 * v0 + R21 * v1 + R22 * v2) + xyz[3 * p + 2];
    scaling_out[3 * i] = logf(s0 * inv_shrink);
    scaling_out[3 * i + 1] = logf(s1 * inv_shrink);
    scaling_out[3 * i + 2] = logf(s2 * inv_shrink);
}

// reset_opacity (/root/reference/src/trainer/rodygs_static.py:151-160 + replace_tensor_to_optimizer,
// src/trainer/utils.py:15-32): logit <- inverse_sigmoid(min(sigmoid(logit), cap)), both Adam moments <- 0.
__global__ void rdg_reset_opacity_kernel(long long n, float cap, float* __restrict__ logit, float* __restrict__ m,
                                         float* __restrict__ v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float o = 1.0f / (1.0f + expf(-logit[i]));
    const float c = fminf(o, cap);
    logit[i] = logf(c / (1.0f - c));
    m[i] = 0.0f;
    v[i] = 0.0f;
}

// Densification statistics of one iteration as their own launch (rdg_densify_stats; the per-Gaussian backward does the
// same update in passing, rdg_preprocess_bwd.hip): rows [row0, row0 + n) of the concatenated cloud, visible = radius > 0
// (/root/reference/src/trainer/rodygs.py:316-341, rodygs_static.py:317-319).
__global__ void __launch_bounds__(256)
rdg_densify_stats_kernel(long long n, long long row0, const float* __restrict__ dm2, const int32_t* __restrict__ radii,
                         float* __restrict__ accum, float* __restrict__ denom, float* __restrict__ maxr) {
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const long long i = row0 + j;
    const int32_t r = radii[i];
    if (r <= 0) return;
    if (maxr) maxr[j] = fmaxf(maxr[j], (float)r);
    if (accum) { const float gx = dm2[3 * i], gy = dm2[3 * i + 1]; accum[j] += sqrtf(gx * gx + gy * gy); }
    if (denom) denom[j] += 1.0f;
}

// Z-curve index of a row inside the cloud's bounding box (rodygs_amd/layout.py::morton_codes, same arithmetic in double):
// q = round((x - lo) / max(hi - lo, 1e-30) * (2^bits - 1)) per axis, bits of x / y / z interleaved from bit 0.
__device__ __forceinline__ unsigned long long rdg_spread3(unsigned long long v) {
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x1F00000000FFFFull;
    v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
__global__ void __launch_bounds__(256)
rdg_morton_codes_kernel(long long n, const float* __restrict__ xyz, const float* __restrict__ lo_hi, int bits,
                        long long* __restrict__ codes) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double top = (double)((1ll << bits) - 1);
    unsigned long long q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double lo = (double)lo_hi[k], hi = (double)lo_hi[3 + k];
        const double ext = fmax(hi - lo, 1e-30);
        q[k] = (unsigned long long)(long long)rint(((double)xyz[3 * i + k] - lo) / ext * top);
    }
    codes[i] = (long long)(rdg_spread3(q[0]) | (rdg_spread3(q[1]) << 1) | (rdg_spread3(q[2]) << 2));
}

// ---- rank of the set entries of a byte mask: rank[i] = (number of non-zero mask bytes in [0, i]) - 1 ---------------------
// What densify.py needs to compact a selection with its size already known on the host (the row lists of clone / split /
// prune, the free-row list of the in-place form) -- the one place the product path still went through the framework's
// scan.  Three launches: per-block counts (4096 entries per 256-thread block), a one-workgroup exclusive scan of the
// block counts, the in-block ranks.
#define RDG_RANK_PER_THREAD 16
#define RDG_RANK_BLOCK (256 * RDG_RANK_PER_THREAD)

__device__ __forceinline__ uint32_t rdg_mask_bits16(const uint8_t* __restrict__ mask, long long i0, long long n) {
    uint32_t bits = 0;
    if (i0 + RDG_RANK_PER_THREAD <= n && ((((uintptr_t)mask) + (uintptr_t)i0) & 15) == 0) {
        const uint4 v = *(const uint4*)(mask + i0);
        const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int b = 0; b < 4; ++b) bits |= (((wds[k] >> (8 * b)) & 0xffu) != 0u ? 1u : 0u) << (4 * k + b);
    } else {
        for (int k = 0; k < RDG_RANK_PER_THREAD; ++k)
            if (i0 + k < n && mask[i0 + k] != 0) bits |= 1u << k;
    }
    return bits;
}

__global__ void __launch_bounds__(256)
rdg_mask_block_count_kernel(long long n, const uint8_t* __restrict__ mask, uint32_t* __restrict__ bsum) {
    __shared__ uint32_t wsum[4];
    const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * RDG_RANK_PER_THREAD;
    const uint32_t c = i0 < n ? (uint32_t)__popc(rdg_mask_bits16(mask, i0, n)) : 0u;
    const uint32_t inc = rdg_wave_scan_incl(c);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// exclusive scan of v[0..n) in place by one workgroup (a run of ceil(n / 1024) block counts per thread)
__global__ void __launch_bounds__(1024) rdg_mask_scan_blocks_kernel(uint32_t* __restrict__ v, int n) {
    __shared__ uint32_t wtot[16];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int per = (n + 1023) / 1024;
    const int t0 = min(n, (int)threadIdx.x * per), t1 = min(n, t0 + per);
    uint32_t mine = 0;
    for (int c = t0; c < t1; ++c) mine += v[c];
    const uint32_t inc = rdg_wave_scan_incl(mine);
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    uint32_t run = inc - mine;
    for (uint32_t k = 0; k < w; ++k) run += wtot[k];
    for (int c = t0; c < t1; ++c) { const uint32_t x = v[c]; v[c] = run; run += x; }
}

__global__ void __launch_bounds__(256)
rdg_mask_rank_kernel(long long n, const uint8_t* __restrict__ mask, const uint32_t* __restrict__ bsum,
                     long long* __restrict__ rank) {
    __shared__ uint32_t wsum[4];
    const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * RDG_RANK_PER_THREAD;
    const uint32_t bits = i0 < n ? rdg_mask_bits16(mask, i0, n) : 0u;
    const uint32_t c = (uint32_t)__popc(bits);
    const uint32_t inc = rdg_wave_scan_incl(c);
    const uint32_t w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t before = bsum[blockIdx.x] + inc - c;
    for (uint32_t k = 0; k < w; ++k) before += wsum[k];
    if (i0 >= n) return;
    const int cnt = (int)min((long long)RDG_RANK_PER_THREAD, n - i0);
    for (int k = 0; k < cnt; ++k)
        rank[i0 + k] = (long long)before + (long long)__popc(bits & ((2u << k) - 1u)) - 1;
}

int rdg_launch_densify_stats(long long n, long long row0, const float* dmeans2D, const int32_t* radii, float* accum,
                             float* denom, float* maxr, hipStream_t s) {
    hipLaunchKernelGGL(rdg_densify_stats_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, row0, dmeans2D,
                       radii, accum, denom, maxr);
    return rdg_check_hip(hipGetLastError(), "densify_stats launch");
}

extern "C" {

int rdg_reset_opacity(int64_t n, float max_opacity, float* opacity_logit, float* exp_avg, float* exp_avg_sq,
                      void* stream) {
    if (n <= 0) return 0;
    if (!(max_opacity > 0.0f && max_opacity < 1.0f)) return rdg_set_error("reset_opacity: max_opacity must be in (0, 1)");
    if (!opacity_logit || !exp_avg || !exp_avg_sq) return rdg_set_error("reset_opacity: NULL buffer");
    hipLaunchKernelGGL(rdg_reset_opacity_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (long long)n, max_opacity, opacity_logit, exp_avg, exp_avg_sq);
    return rdg_check_hip(hipGetLastError(), "reset_opacity launch");
}

int rdg_morton_codes(int64_t n, const float* xyz, const float* lo_hi, int32_t bits, int64_t* codes, void* stream) {
    if (n <= 0) return 0;
    if (bits < 1 || bits > 21) return rdg_set_error("morton_codes: bits must be in [1, 21]");
    if (!xyz || !lo_hi || !codes) return rdg_set_error("morton_codes: NULL buffer");
    hipLaunchKernelGGL(rdg_morton_codes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (long long)n, xyz, lo_hi, (int)bits, (long long*)codes);
    return rdg_check_hip(hipGetLastError(), "morton_codes launch");
}

int rdg_gather_rows(int64_t n_new, int32_t row_len, const int64_t* idx, const float* src, float* dst, void* stream) {
    if (n_new <= 0 || row_len <= 0) return 0;
    const long long n = n_new * row_len;
    hipLaunchKernelGGL(rdg_gather_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (long long)n_new, row_len, (const long long*)idx, src, dst);
    return rdg_check_hip(hipGetLastError(), "gather_rows launch");
}

int rdg_split_children(int64_t n, int32_t N, const int64_t* parent, const float* xyz, const float* scaling,
                       const float* rotation, const float* z, float* xyz_out, float* scaling_out, void* stream) {
    if (n <= 0) return 0;
    if (N < 1) return rdg_set_error("split_children: N must be >= 1");
    hipLaunchKernelGGL(rdg_split_children_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (long long)n, (const long long*)parent, 1.0f / (0.8f * (float)N), xyz, scaling, rotation, z,
                       xyz_out, scaling_out);
    return rdg_check_hip(hipGetLastError(), "split_children launch");
}

size_t rdg_mask_rank_ws_bytes(int64_t n) {
    return (size_t)((n + RDG_RANK_BLOCK - 1) / RDG_RANK_BLOCK + 1) * 4 + 256;
}

int rdg_mask_rank(int64_t n, const uint8_t* mask, int64_t* rank, void* ws, void* stream) {
    if (n <= 0) return 0;
    if (!mask || !rank || !ws) return rdg_set_error("mask_rank: NULL buffer");
    if (n >= ((int64_t)1 << 32)) return rdg_set_error("mask_rank: n must be below 2^32");
    const long long nblk = (n + RDG_RANK_BLOCK - 1) / RDG_RANK_BLOCK;
    uint32_t* bsum = (uint32_t*)ws;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(rdg_mask_block_count_kernel, dim3((unsigned)nblk), dim3(256), 0, st, (long long)n, mask, bsum);
    hipLaunchKernelGGL(rdg_mask_scan_blocks_kernel, dim3(1), dim3(1024), 0, st, bsum, (int)nblk);
    hipLaunchKernelGGL(rdg_mask_rank_kernel, dim3((unsigned)nblk), dim3(256), 0, st, (long long)n, mask, bsum,
                       (long long*)rank);
    return rdg_check_hip(hipGetLastError(), "mask_rank launch");
}

}  // extern "C"
