// rdg_loss.hip -- fused photometric loss of the RoDyGS train step:
//     loss = (1 - lambda) * mean|x - y| + lambda * (1 - mean SSIM(x, y))
// with the reference's SSIM definition (/root/reference/src/utils/loss_utils.py:19-100: 11x11 Gaussian window,
// sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2) and call sites /root/reference/src/trainer/losses.py:78-107.
//
// The reference runs five 121-tap depth-wise convolutions forward and their transposes backward through the
// framework's conv library (3.6 ms per step at 1080p on MI355X, more than the whole rasterizer).  Here:
//   forward : one pass per 32x32 tile -- the 42x42 halo tile of x and y goes to LDS once, the windowed moments
//             (FOUR maps: x, y, x^2 + y^2, xy -- E[x^2] and E[y^2] are only needed as a sum) are produced separably
//             (11 + 11 taps, register-blocked 4 outputs per work item so that an output costs ~22 LDS reads instead
//             of ~90) from LDS, and the kernel stores, per pixel, the three partial derivatives dm/dmu1, dm/dE[x^2],
//             dm/dE[xy] of the SSIM map; |x-y| and the map are block-reduced into two floats per tile.
//   backward: dL/dx(p) = g * (conv(dm/dmu1) + 2 x(p) conv(dm/dE11) + y(p) conv(dm/dE12))(p) + L1 term, i.e. the
//             same separable filter over the three stored maps (the window is symmetric).
// Forward reads 2 and writes 3 images, backward reads 5 and writes 1; the backward runs at 4.9 TB/s, the forward is bound by
// VALU issue (round 6: profiles/r06_experiments.txt 3).
#include "rdg_common.h"
#include <math.h>

#define LTX 32                 // output tile: 32 x 32 pixels per 256-thread workgroup
#define LTY 32
#define LH 5
#define LWX (LTX + 2 * LH)    // 42: tile + halo
#define LWY (LTY + 2 * LH)
#define RDG_LOSS_NL ((LWY * LWX + 255) / 256)   // halo-tile elements per thread

struct RdgWin { float w[11]; };

// Both kernels are VALU-issue bound (not HBM-bound: profiles/r06_experiments.txt), and a third of the issue slots went to
// address arithmetic: 64-bit address chains per load / store and a division by 42 per halo element.  So: uniform base
// pointer (SGPRs) + one unsigned 32-bit BYTE offset per access (the global_load saddr form; the launchers refuse images
// of 2^30 pixels or more), and a walk over the halo tile that advances (row, column) without dividing.
__device__ __forceinline__ float rdg_ldg(const float* base, unsigned byte_off) {
    return *(const float*)((const char*)base + byte_off);
}
__device__ __forceinline__ void rdg_stg(float* base, unsigned byte_off, float v) {
    *(float*)((char*)base + byte_off) = v;
}
// The halo tile is loaded by 6 x 42 = 252 threads, thread (hr0, hc) taking the rows hr0 + 6 it, it = 0..6 (7 x 6 = 42
// rows exactly): the column test is made once, a row costs one compare and one add, and the LDS offsets are immediates.
#define RDG_HALO_RPI (256 / LWX)
static_assert(RDG_HALO_RPI * RDG_LOSS_NL == LWY, "the halo walk covers the tile with no remainder");
#define RDG_HALO_WALK(hc, hact, colok, gy, g, lo)                                        \
    const int hr0_ = tid / LWX, hc = tid - hr0_ * LWX;                                   \
    const bool hact = tid < RDG_HALO_RPI * LWX;                                          \
    const bool colok = hact && (unsigned)(ox + hc - LH) < (unsigned)Wd;                  \
    int gy = oy + hr0_ - LH;                                                             \
    unsigned g = (unsigned)(gy * Wd + ox + hc - LH) * 4u;                                \
    const int lo = hr0_ * (LWX + 1) + hc;

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so every XCD gets a
// contiguous band of tiles: the 5-pixel halos a tile shares with its neighbours are then L2 hits instead of a second
// trip to memory (PMC: the backward kernel fetched 2.8x its algorithmic bytes with the plain 3-D grid).
__device__ __forceinline__ bool rdg_loss_tile(int gx, int gy, int C, int& ox, int& oy, int& c, size_t& bid) {
    const int n = gx * gy * C, per = (n + 7) >> 3;
    const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || t >= n) return false;
    c = t / (gx * gy);
    const int r = t - c * gx * gy;
    oy = (r / gx) * LTY; ox = (r - (r / gx) * gx) * LTX;
    bid = (size_t)t;
    return true;
}

static RdgWin rdg_make_window() {
    RdgWin W;
    float g[11], sum = 0.f;
    for (int i = 0; i < 11; ++i) {
        g[i] = (float)exp(-((double)((i - 5) * (i - 5))) / (2.0 * 1.5 * 1.5));
        sum += g[i];
    }
    for (int i = 0; i < 11; ++i) W.w[i] = g[i] / sum;
    return W;
}

// Register blocking: a work item of the horizontal pass produces RB adjacent outputs of a row from RB + 10 LDS
// reads per input (instead of 11 each), a thread of the vertical pass RB vertically adjacent pixels from RB + 10 rows.
#define RB 4

// The window weights arrive as kernel arguments (SGPRs); a VALU instruction with a scalar source issues at half the
// rate of one with vector sources on gfx950 (profiles/r02_valu_issue_cost.txt), and the filters are nothing but such
// multiply-adds: keep the eleven weights in VGPRs.
#define RDG_LOSS_WEIGHTS(w, win)                                      \
    float w[11];                                                      \
    _Pragma("unroll") for (int k_ = 0; k_ < 11; ++k_) { w[k_] = (win).w[k_]; asm volatile("" : "+v"(w[k_])); }

#define RDG_LOSS_HITEMS ((LWY * (LTX / RB) + 255) / 256)   // horizontal-pass work items per thread (2)

// Tiles per workgroup of the forward kernel: a workgroup takes RDG_LOSS_NT consecutive tiles of its XCD band and issues the global
// loads of the next tile's halo (14 registers) before it filters the current one, so only the first tile of a workgroup waits
// for memory.
#ifndef RDG_LOSS_NT
#define RDG_LOSS_NT 2
#endif

__device__ __forceinline__ bool rdg_loss_tile_nt(int gx, int gy, int C, int rep, int& ox, int& oy, int& c, size_t& bid) {
    const int n = gx * gy * C, per = (n + 7) >> 3;
    const int j = (int)(blockIdx.x >> 3) * RDG_LOSS_NT + rep;
    const int t = (int)(blockIdx.x & 7) * per + j;
    if (j >= per || t >= n) return false;
    c = t / (gx * gy);
    const int r = t - c * gx * gy;
    oy = (r / gx) * LTY; ox = (r - (r / gx) * gx) * LTX;
    bid = (size_t)t;
    return true;
}

// four waves per SIMD (<= 128 registers): with the 35 KB of LDS that is four workgroups per CU
__global__ void __launch_bounds__(256, 4)
rdg_loss_fwd_kernel(int H, int Wd, int C, RdgWin win, const float* __restrict__ img, const float* __restrict__ gt,
                    float* __restrict__ maps, float* __restrict__ sums) {
    // FOUR filtered maps, not five: the SSIM map and the three derivatives the backward needs (d/dmu1, d/dE[x^2], d/dE[xy]: the
    // ground truth gets no gradient) contain E[x^2] and E[y^2] only through sigma1^2 + sigma2^2, so x^2 + y^2 is filtered as ONE
    // map.  35.1 KB of LDS: four workgroups per CU (the phases of a workgroup are separated by barriers; others fill the gaps)
    __shared__ float sx[LWY][LWX + 1], sy[LWY][LWX + 1];
    __shared__ float sh4[4][LWY][LTX];
    __shared__ float sred[2][4];
    const int tid = threadIdx.x;
    const int gxt = (Wd + LTX - 1) / LTX, gyt = (H + LTY - 1) / LTY;
    int ox, oy, c; size_t bid;
    if (!rdg_loss_tile_nt(gxt, gyt, C, 0, ox, oy, c, bid)) return;
    RDG_LOSS_WEIGHTS(w, win)
    const size_t hw = (size_t)H * Wd;
    const size_t stride = (size_t)C * hw;  // maps layout: [3 maps][C][H][W]
    // this thread's element of the halo walk (6 x 42 threads, 7 rows each) and of the vertical pass (column tx, RB rows from ty0)
    const int hr0 = tid / LWX, hc = tid - hr0 * LWX;
    const bool hact = tid < RDG_HALO_RPI * LWX;
    const int lo = hr0 * (LWX + 1) + hc;
    const unsigned gstep = (unsigned)(RDG_HALO_RPI * Wd) * 4u;
    const int tx = tid % LTX, ty0 = (tid / LTX) * RB;
    // the whole halo tile in ONE round of loads per thread (7 elements x 2 images in flight before the first LDS store)
    float xv[RDG_LOSS_NL], yv[RDG_LOSS_NL];
#define RDG_LOSS_ISSUE_LOADS(ox_, oy_, c_)                                                               \
    {                                                                                                    \
        const float* X_ = img + (c_) * hw;                                                               \
        const float* Y_ = gt + (c_) * hw;                                                                \
        const bool colok_ = hact && (unsigned)((ox_) + hc - LH) < (unsigned)Wd;                          \
        int gy_ = (oy_) + hr0 - LH;                                                                      \
        unsigned g_ = (unsigned)(gy_ * Wd + (ox_) + hc - LH) * 4u;                                       \
        _Pragma("unroll") for (int it = 0; it < RDG_LOSS_NL; ++it) {                                     \
            xv[it] = 0.f; yv[it] = 0.f;                                                                  \
            if (colok_ && (unsigned)gy_ < (unsigned)H) { xv[it] = rdg_ldg(X_, g_); yv[it] = rdg_ldg(Y_, g_); } \
            gy_ += RDG_HALO_RPI; g_ += gstep;                                                            \
        }                                                                                                \
    }
    RDG_LOSS_ISSUE_LOADS(ox, oy, c)
#pragma unroll 1
    for (int rep = 0; rep < RDG_LOSS_NT; ++rep) {
        // the barrier in front of these stores also ends the previous tile's vertical pass (it reads sh4, which the horizontal
        // pass below overwrites); sx / sy were last read before the previous tile's second barrier
        if (rep > 0) __syncthreads();
        if (hact) {
#pragma unroll
            for (int it = 0; it < RDG_LOSS_NL; ++it) {
                (&sx[0][0])[lo + it * RDG_HALO_RPI * (LWX + 1)] = xv[it];
                (&sy[0][0])[lo + it * RDG_HALO_RPI * (LWX + 1)] = yv[it];
            }
        }
        __syncthreads();
        // the next tile's halo: in flight while this one is filtered
        int nox = 0, noy = 0, nc = 0; size_t nbid = 0;
        const bool more = rep + 1 < RDG_LOSS_NT && rdg_loss_tile_nt(gxt, gyt, C, rep + 1, nox, noy, nc, nbid);
        if (more) RDG_LOSS_ISSUE_LOADS(nox, noy, nc)
        const int px = ox + tx;
        // the |x - y| terms of the thread's pixels are taken now, while the halo tile of x is still in LDS
        float l1v[RB];
#pragma unroll
        for (int o = 0; o < RB; ++o) l1v[o] = fabsf(sx[ty0 + o + LH][tx + LH] - sy[ty0 + o + LH][tx + LH]);
        // horizontal pass: LWY rows x (LTX / RB) groups of RB outputs
#pragma unroll
        for (int hi = 0; hi < RDG_LOSS_HITEMS; ++hi) {
            const int item = tid + 256 * hi;
            if (item < LWY * (LTX / RB)) {
                const int r = item / (LTX / RB), c0 = (item - r * (LTX / RB)) * RB;
                float xs[RB + 10], ys[RB + 10];
#pragma unroll
                for (int k = 0; k < RB + 10; ++k) { xs[k] = sx[r][c0 + k]; ys[k] = sy[r][c0 + k]; }
                // the products once per element (not once per tap): an output then costs 4 multiply-adds per tap
                float ss[RB + 10], xy[RB + 10];
#pragma unroll
                for (int k = 0; k < RB + 10; ++k) { ss[k] = xs[k] * xs[k] + ys[k] * ys[k]; xy[k] = xs[k] * ys[k]; }
#pragma unroll
                for (int o = 0; o < RB; ++o) {
                    float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f;
#pragma unroll
                    for (int k = 0; k < 11; ++k) {
                        h0 += w[k] * xs[o + k]; h1 += w[k] * ys[o + k]; h2 += w[k] * ss[o + k]; h3 += w[k] * xy[o + k];
                    }
                    sh4[0][r][c0 + o] = h0; sh4[1][r][c0 + o] = h1; sh4[2][r][c0 + o] = h2; sh4[3][r][c0 + o] = h3;
                }
            }
            asm volatile("" ::: "memory");      // one work item's reads at a time (the second item's 28 loads hoisted: +28 registers)
        }
        __syncthreads();
        float l1 = 0.f, ms = 0.f;
        float mu1[RB], mu2[RB], ess[RB], e12[RB];
#pragma unroll
        for (int o = 0; o < RB; ++o) { mu1[o] = 0.f; mu2[o] = 0.f; ess[o] = 0.f; e12[o] = 0.f; }
#pragma unroll
        for (int rr = 0; rr < RB + 10; ++rr) {
            const float a0 = sh4[0][ty0 + rr][tx], a1 = sh4[1][ty0 + rr][tx], a2 = sh4[2][ty0 + rr][tx], a3 = sh4[3][ty0 + rr][tx];
            if ((rr & 1) == 1) asm volatile("" ::: "memory");   // eight row reads in flight, not fifty-six (registers: occupancy 4)
#pragma unroll
            for (int o = 0; o < RB; ++o) {
                const int k = rr - o;
                if (k >= 0 && k < 11) { mu1[o] += w[k] * a0; mu2[o] += w[k] * a1; ess[o] += w[k] * a2; e12[o] += w[k] * a3; }
            }
        }
        float* M0 = maps + c * hw;
        float* M1 = M0 + stride;
        float* M2 = M1 + stride;
#pragma unroll
        for (int o = 0; o < RB; ++o) {
            const int py = oy + ty0 + o;
            if (py < H && px < Wd) {
                const float C1 = 0.0001f, C2 = 0.0009f;
                const float m1 = mu1[o], m2 = mu2[o];
                const float mu1s = m1 * m1, mu2s = m2 * m2, mu12 = m1 * m2;
                const float s12 = e12[o] - mu12;
                const float B1 = mu1s + mu2s + C1;
                const float A1 = 2.f * mu12 + C1, A2 = 2.f * s12 + C2, B2 = (ess[o] - B1) + (C1 + C2);   // sigma1^2 + sigma2^2 + C2
                // v_rcp_f32 (1 ulp) instead of the IEEE division sequence (ten instructions each, eight of them per thread):
                // B1, B2 >= C1, C2 > 0, and the parity bar of the loss is 1e-4
                const float iB1 = __builtin_amdgcn_rcpf(B1), iB2 = __builtin_amdgcn_rcpf(B2);
                const float m = A1 * A2 * iB1 * iB2;
                const float dmu1 = 2.f * m2 * (A2 - A1) * iB1 * iB2 - 2.f * m1 * m * (iB1 - iB2);
                const float de11 = -m * iB2;
                const float de12 = 2.f * A1 * iB1 * iB2;
                const unsigned p = (unsigned)(py * Wd + px) * 4u;
                rdg_stg(M0, p, dmu1); rdg_stg(M1, p, de11); rdg_stg(M2, p, de12);
                ms += m;
                l1 += l1v[o];
            }
        }
        l1 = rdg_wave_sum_to63(l1);
        ms = rdg_wave_sum_to63(ms);
        if ((tid & 63) == 63) { sred[0][tid >> 6] = l1; sred[1][tid >> 6] = ms; }
        __syncthreads();
        // one partial pair per TILE, summed in fixed order by the finalize kernel: 24 k workgroups adding into the
        // same two floats ran at the contended-atomic rate (0.3 ms of a 0.33 ms kernel) and were not deterministic
        if (tid < 2) sums[2 * bid + tid] = (sred[tid][0] + sred[tid][1]) + (sred[tid][2] + sred[tid][3]);
        if (!more) break;
        ox = nox; oy = noy; c = nc; bid = nbid;
    }
#undef RDG_LOSS_ISSUE_LOADS
}

__global__ void __launch_bounds__(1024)
rdg_loss_finalize_kernel(const float* __restrict__ sums, int nblk, float inv_n, float lambda, float* __restrict__ loss) {
    __shared__ float s0[16], s1[16];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 1024) { a += sums[2 * i]; b += sums[2 * i + 1]; }
    a = rdg_wave_sum_to63(a);
    b = rdg_wave_sum_to63(b);
    if ((threadIdx.x & 63) == 63) { s0[threadIdx.x >> 6] = a; s1[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float l1 = 0.f, ss = 0.f;
        for (int k = 0; k < 16; ++k) { l1 += s0[k]; ss += s1[k]; }
        l1 *= inv_n; ss *= inv_n;
        loss[0] = (1.0f - lambda) * l1 + lambda * (1.0f - ss);
        loss[1] = l1;
        loss[2] = ss;
    }
}

__global__ void __launch_bounds__(256)
rdg_loss_bwd_kernel(int H, int Wd, int C, RdgWin win, const float* __restrict__ img, const float* __restrict__ gt,
                    const float* __restrict__ maps, const float* __restrict__ grad_loss, float inv_n, float lambda,
                    float* __restrict__ d_img) {
    // one LDS buffer: the halo tiles of the three maps, then (registers in between) their horizontally filtered rows
    __shared__ float smem[3 * LWY * (LWX + 1)];
    float (*sa)[LWY][LWX + 1] = (float (*)[LWY][LWX + 1])smem;
    float (*sh)[LWY][LTX] = (float (*)[LWY][LTX])smem;
    const int tid = threadIdx.x;
    int ox, oy, c; size_t bid;
    if (!rdg_loss_tile((Wd + LTX - 1) / LTX, (H + LTY - 1) / LTY, C, ox, oy, c, bid)) return;
    RDG_LOSS_WEIGHTS(w, win)
    const size_t hw = (size_t)H * Wd;
    const size_t stride = (size_t)C * hw;
    const float* M0 = maps + c * hw;
    const float* M1 = M0 + stride;
    const float* M2 = M1 + stride;
    {
        float a[RDG_LOSS_NL], b[RDG_LOSS_NL], d[RDG_LOSS_NL];
        RDG_HALO_WALK(hc, hact, colok, gy, g, lo)
        const unsigned gstep = (unsigned)(RDG_HALO_RPI * Wd) * 4u;
#pragma unroll
        for (int it = 0; it < RDG_LOSS_NL; ++it) {
            a[it] = 0.f; b[it] = 0.f; d[it] = 0.f;
            if (colok && (unsigned)gy < (unsigned)H) { a[it] = rdg_ldg(M0, g); b[it] = rdg_ldg(M1, g); d[it] = rdg_ldg(M2, g); }
            gy += RDG_HALO_RPI; g += gstep;
        }
        if (hact) {
#pragma unroll
            for (int it = 0; it < RDG_LOSS_NL; ++it) {
                (&sa[0][0][0])[lo + it * RDG_HALO_RPI * (LWX + 1)] = a[it];
                (&sa[1][0][0])[lo + it * RDG_HALO_RPI * (LWX + 1)] = b[it];
                (&sa[2][0][0])[lo + it * RDG_HALO_RPI * (LWX + 1)] = d[it];
            }
        }
    }
    __syncthreads();
    float hreg[RDG_LOSS_HITEMS][RB][3];
#pragma unroll
    for (int hi = 0; hi < RDG_LOSS_HITEMS; ++hi) {
        const int item = tid + 256 * hi;
        if (item < LWY * (LTX / RB)) {
            const int r = item / (LTX / RB), c0 = (item - r * (LTX / RB)) * RB;
            float v0[RB + 10], v1[RB + 10], v2[RB + 10];
#pragma unroll
            for (int k = 0; k < RB + 10; ++k) { v0[k] = sa[0][r][c0 + k]; v1[k] = sa[1][r][c0 + k]; v2[k] = sa[2][r][c0 + k]; }
#pragma unroll
            for (int o = 0; o < RB; ++o) {
                float h0 = 0.f, h1 = 0.f, h2 = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) { h0 += w[k] * v0[o + k]; h1 += w[k] * v1[o + k]; h2 += w[k] * v2[o + k]; }
                hreg[hi][o][0] = h0; hreg[hi][o][1] = h1; hreg[hi][o][2] = h2;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int hi = 0; hi < RDG_LOSS_HITEMS; ++hi) {
        const int item = tid + 256 * hi;
        if (item < LWY * (LTX / RB)) {
            const int r = item / (LTX / RB), c0 = (item - r * (LTX / RB)) * RB;
#pragma unroll
            for (int o = 0; o < RB; ++o) { sh[0][r][c0 + o] = hreg[hi][o][0]; sh[1][r][c0 + o] = hreg[hi][o][1]; sh[2][r][c0 + o] = hreg[hi][o][2]; }
        }
    }
    __syncthreads();
    const int tx = tid % LTX, ty0 = (tid / LTX) * RB;
    const int px = ox + tx;
    float u0[RB], u1[RB], u2[RB];
#pragma unroll
    for (int o = 0; o < RB; ++o) { u0[o] = 0.f; u1[o] = 0.f; u2[o] = 0.f; }
#pragma unroll
    for (int rr = 0; rr < RB + 10; ++rr) {
        const float a0 = sh[0][ty0 + rr][tx], a1 = sh[1][ty0 + rr][tx], a2 = sh[2][ty0 + rr][tx];
#pragma unroll
        for (int o = 0; o < RB; ++o) {
            const int k = rr - o;
            if (k >= 0 && k < 11) { u0[o] += w[k] * a0; u1[o] += w[k] * a1; u2[o] += w[k] * a2; }
        }
    }
    const float go = grad_loss ? grad_loss[0] : 1.0f;
    const float gs = -lambda * inv_n * go;           // d loss / d ssim_map(q)
    const float gl = (1.0f - lambda) * inv_n * go;   // d loss / d |x - y|
    const float* X = img + c * hw;
    const float* Y = gt + c * hw;
    float* DX = d_img + c * hw;
#pragma unroll
    for (int o = 0; o < RB; ++o) {
        const int py = oy + ty0 + o;
        if (py < H && px < Wd) {
            const unsigned p = (unsigned)(py * Wd + px) * 4u;
            const float x = rdg_ldg(X, p), y = rdg_ldg(Y, p);
            const float d = x - y;
            const float sgn = d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.0f);
            rdg_stg(DX, p, gs * (u0[o] + 2.0f * x * u1[o] + y * u2[o]) + gl * sgn);
        }
    }
}

extern "C" {

size_t rdg_loss_ws_bytes(int32_t C, int32_t H, int32_t W) {
    const size_t nblk = (size_t)((W + LTX - 1) / LTX) * ((H + LTY - 1) / LTY) * C;
    return (size_t)3 * C * H * W * 4 + nblk * 8 + 256;
}

int rdg_photometric_loss_forward(int32_t C, int32_t H, int32_t W, const float* img, const float* gt, float lambda,
                                 void* ws, float* loss3, void* stream) {
    if (C <= 0 || H <= 0 || W <= 0 || (size_t)H * W >= ((size_t)1 << 30)) return rdg_set_error("loss: bad image size");
    hipStream_t st = (hipStream_t)stream;
    float* maps = (float*)ws;
    float* sums = (float*)((char*)ws + (size_t)3 * C * H * W * 4);
    rdg_stage_begin(RDG_STAGE_LOSS_FWD, st);
    const RdgWin win = rdg_make_window();
    const int n_tiles = ((W + LTX - 1) / LTX) * ((H + LTY - 1) / LTY) * C;
#ifndef RDG_ABL_LOSS_LDSPAD     // measurement builds: extra (unused) dynamic LDS bytes, i.e. fewer workgroups per CU
#define RDG_ABL_LOSS_LDSPAD 0
#endif
    const int per_band = (n_tiles + 7) / 8;      // tiles per XCD band; RDG_LOSS_NT consecutive ones per workgroup
    hipLaunchKernelGGL(rdg_loss_fwd_kernel, dim3(((per_band + RDG_LOSS_NT - 1) / RDG_LOSS_NT) * 8), dim3(256), RDG_ABL_LOSS_LDSPAD, st,
                       H, W, C, win, img, gt, maps, sums);
    hipLaunchKernelGGL(rdg_loss_finalize_kernel, dim3(1), dim3(1024), 0, st, sums, n_tiles,
                       1.0f / ((float)C * H * W), lambda, loss3);
    rdg_stage_end(RDG_STAGE_LOSS_FWD, st);
    return rdg_check_hip(hipGetLastError(), "loss_fwd launch");
}

int rdg_photometric_loss_backward(int32_t C, int32_t H, int32_t W, const float* img, const float* gt, float lambda,
                                  const void* ws, const float* grad_loss, float* d_img, void* stream) {
    if (C <= 0 || H <= 0 || W <= 0 || (size_t)H * W >= ((size_t)1 << 30)) return rdg_set_error("loss: bad image size");
    hipStream_t st = (hipStream_t)stream;
    const RdgWin win = rdg_make_window();
    const int n_tiles = ((W + LTX - 1) / LTX) * ((H + LTY - 1) / LTY) * C;
    rdg_stage_begin(RDG_STAGE_LOSS_BWD, st);
    hipLaunchKernelGGL(rdg_loss_bwd_kernel, dim3(((n_tiles + 7) / 8) * 8), dim3(256), 0, st, H, W, C, win, img, gt,
                       (const float*)ws, grad_loss,
                       1.0f / ((float)C * H * W), lambda, d_img);
    rdg_stage_end(RDG_STAGE_LOSS_BWD, st);
    return rdg_check_hip(hipGetLastError(), "loss_bwd launch");
}

}  // extern "C"
