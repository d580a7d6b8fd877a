// rdg_render.hip -- per-tile alpha compositing, forward and backward.
//
// Geometry: one 256-thread workgroup per 16x16 tile = 4 wave64, each wave owning one 8x8 pixel quadrant
// (lane -> (lane&7, lane>>3)), so early termination and "does any pixel of my quadrant see this splat" are
// decided per wave, not per 32-wide warp.  Splat records of a tile are staged 256 at a time through LDS with one
// coalesced 64-B gather per thread; inside the loop every lane reads the same LDS address (broadcast).
//
// Wave-level occupancy masks (the wave64 replacement for per-thread rejection): while staging, the thread that
// holds splat j tests the axis-aligned box of the splat's "alpha >= 1/255" ellipse against each of the four
// quadrants; a ballot per quadrant turns the 256 verdicts into four 64-bit masks per wave in LDS.  A wave then
// walks only the set bits of its own masks with scalar bit ops (s_ff1 / s_andn2) -- on the bench scene 60 % of the
// (wave, splat) pairs are never touched.  The test is conservative (power margin 0.01 + inflated box), so the
// result is identical to visiting every splat; order inside the list is preserved.
//
// Backward: per (wave, splat) the 10 partial derivatives are reduced with a TRANSPOSED DPP reduction (the value
// index is folded into the lane index: 14 DPP adds instead of 60) down to one total per 16-lane row; the row totals
// are parked with plain LDS stores in a ring PRIVATE to the wave (no LDS atomics -- ds_add_f32 into a table shared
// by the 4 waves was a third of the kernel), and every 16 splats the wave flushes its ring with 64-B-row global
// float atomics: 16 consecutive lanes cover the 16 floats of a Gaussian's accumulator row, 4 Gaussians per
// instruction, which is the access shape the global float-atomic unit runs at full rate for (MI355X_MICROARCH.md
// "Global float atomics").  Atomics are therefore per (wave, splat) that a pixel actually blended.
//
// The kernels are VALU/transcendental bound, not HBM bound.
#include "rdg_common.h"

#define RDG_BATCH 256

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give every XCD one contiguous
// band of tiles -- neighbouring tiles share splats, and a band's splat records then stay in that XCD's L2.
__device__ __forceinline__ int rdg_tile_of_block(int bid, int n_tiles) {
    const int per = (n_tiles + 7) >> 3;
    return (bid & 7) * per + (bid >> 3);
}

// 4-bit mask: bit q set = the splat may reach quadrant q (q = qy*2+qx) of the tile whose top-left pixel is
// (X0, Y0).  A pixel blends the splat only where alpha >= 1/255 and power <= 0, i.e. where the quadratic form
// Q(d) = a dx^2 + 2 b dx dy + c dy^2 <= r2 = 2 ln(255 o).  The test is EXACT for the quadrant's rectangle of pixel
// centres (not the axis-aligned box of the ellipse, which lets diagonal splats through): the minimum of the convex Q
// over a box is 0 if the centre is inside, otherwise it sits on one of the four edges, where it is a clamped 1-D
// parabola.  r2 carries a margin (0.02 in the log + 1e-3 relative) that dwarfs the rounding of this evaluation, so no
// blending pixel is ever skipped; the result is identical to visiting every splat.  Anything unusual (indefinite
// conic, NaN) keeps the splat for every quadrant.
__device__ __forceinline__ float rdg_edge_min(float a, float b2, float c, float inv_c, float ue, float v0, float v1) {
    // min over v in [v0, v1] of a ue^2 + 2 b ue v + c v^2   (b2 = 2 b)
    const float t = b2 * ue;
    const float vs = __builtin_amdgcn_fmed3f(-0.5f * t * inv_c, v0, v1);
    return fmaf(fmaf(c, vs, t), vs, a * ue * ue);
}
__device__ __forceinline__ uint32_t rdg_quadrant_bits(const float4 q0, const float4 q1, float X0, float Y0) {
    const float a = q0.z, b = q0.w, c = q1.x, o = q1.y;
    const float t255 = 255.0f * o;
    if (t255 < 0.99f) return 0u;  // alpha can never reach 1/255 (margin below)
    const float det = a * c - b * b;
    if (!(det > 0.0f) || !(a > 0.0f) || !(c > 0.0f)) return 0xFu;
    const float r2 = 2.0f * (__logf(t255) + 0.02f) * 1.001f;
    const float inv_a = 1.0f / a, inv_c = 1.0f / c, b2 = 2.0f * b;
    uint32_t bits = 0u;
#pragma unroll
    for (int qy = 0; qy < 2; ++qy) {
        const float v1 = q0.y - (Y0 + 8.0f * qy), v0 = v1 - 7.0f;       // v = py - y over the quadrant's pixel rows
#pragma unroll
        for (int qx = 0; qx < 2; ++qx) {
            const float u1 = q0.x - (X0 + 8.0f * qx), u0 = u1 - 7.0f;
            const bool inside = u0 <= 0.0f && u1 >= 0.0f && v0 <= 0.0f && v1 >= 0.0f;
            const float m = fminf(fminf(rdg_edge_min(a, b2, c, inv_c, u0, v0, v1), rdg_edge_min(a, b2, c, inv_c, u1, v0, v1)),
                                  fminf(rdg_edge_min(c, b2, a, inv_a, v0, u0, u1), rdg_edge_min(c, b2, a, inv_a, v1, u0, u1)));
            // written so that a NaN keeps the quadrant
            bits |= (uint32_t)(inside || !(m > r2)) << (qy * 2 + qx);
        }
    }
    return bits;
}

// Staged form of a splat's conic: (A2, B, C2) = -log2(e) * (a, b, c), so that
//   log2(G) = 0.5 (A2 dx^2 + C2 dy^2) + B dx dy      and      G = v_exp_f32(log2 G)  with no extra multiply,
// and the products t = A2 dx, v = C2 dy are reused by the backward (dG/ddx ~ t + B dy, dG/ddy ~ v + B dx).  Forward
// and backward evaluate exactly this expression, so they take identical blend / skip decisions.
#define RDG_NEG_LOG2E (-1.4426950408889634f)
__device__ __forceinline__ float rdg_log2_gauss(float A2, float B, float C2, float dx, float dy, float& t, float& v) {
    t = A2 * dx;
    v = C2 * dy;
    const float u = fmaf(v, dy, t * dx);
    return fmaf(0.5f, u, (B * dx) * dy);
}

// wave votes straight from the ballot (hip's __any/__all go through an int compare per lane)
__device__ __forceinline__ bool rdg_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ bool rdg_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }

__device__ __forceinline__ unsigned long long rdg_uniform_u64(unsigned long long v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

__global__ void __launch_bounds__(256)
rdg_render_fwd_kernel(int W, int H, int gx, int n_tiles, int render_normal, const float* __restrict__ bg,
                      const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      const RdgRec* __restrict__ rec, long long capacity, const int32_t* __restrict__ num_rendered,
                      float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
                      float* __restrict__ out_depth, float* __restrict__ out_normal, float* __restrict__ out_alpha,
                      unsigned long long* __restrict__ hitbits) {
    // on capacity overflow the binning stage has emptied every tile range: this kernel then renders the background
    const int tile = rdg_tile_of_block(blockIdx.x, n_tiles);
    if (tile >= n_tiles) return;
    __shared__ float4 sQ0[RDG_BATCH], sQ1[RDG_BATCH], sQ2[RDG_BATCH], sQ3[RDG_BATCH];
    __shared__ unsigned long long sMask[4][4];  // [consumer quadrant][staging wave]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int tx = tile % gx, ty = tile / gx;
    const int pxi = tx * RDG_TILE + (wv & 1) * 8 + (lane & 7);
    const int pyi = ty * RDG_TILE + (wv >> 1) * 8 + (lane >> 3);
    const bool inside = pxi < W && pyi < H;
    float pixx = (float)pxi, pixy = (float)pyi;
    // opaque to the compiler: it otherwise re-converts the integer coordinate inside the visit loop (one VALU
    // instruction per visit to save one register)
    asm volatile("" : "+v"(pixx), "+v"(pixy));
    const float X0 = (float)(tx * RDG_TILE), Y0 = (float)(ty * RDG_TILE);
    const uint2 range = ranges[tile];
    const int todo_total = (int)(range.y - range.x);
    const int rounds = (todo_total + RDG_BATCH - 1) / RDG_BATCH;
    // visit record for the backward: word w of this tile covers list slots [64 w, 64 w + 63]; the words of a tile
    // start at (range.x / 64 + tile), which cannot overlap the next tile's; zeroed by the launcher
    unsigned long long* const hit = hitbits + ((size_t)(range.x >> 6) + (size_t)tile) * 4;

    // "this pixel has stopped" is folded into its alpha threshold (+inf once stopped / outside the image): the hot
    // test is two compares, and no loop-carried lane mask has to be re-canonicalised every iteration
    float amin = inside ? RDG_ALPHA_MIN : __builtin_inff();
    float T = 1.0f;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
    uint32_t last_contributor = 0;
    for (int r = 0; r < rounds; ++r) {
        if (__syncthreads_count(amin > 1.0f) == 256) break;
        const int k = r * RDG_BATCH + tid;
        uint32_t qbits = 0;
        if (k < todo_total) {
            const uint32_t id = point_list[range.x + k];
            const RdgRec* p = rec + id;
            const float4 q0 = p->q0, q1 = p->q1;
            sQ0[tid] = make_float4(q0.x, q0.y, RDG_NEG_LOG2E * q0.z, RDG_NEG_LOG2E * q0.w);
            sQ1[tid] = make_float4(RDG_NEG_LOG2E * q1.x, q1.y, q1.z, 0.0f);
            sQ2[tid] = p->q2;
            if (render_normal) sQ3[tid] = p->q3;
            qbits = rdg_quadrant_bits(q0, q1, X0, Y0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned long long m = __ballot((qbits >> q) & 1u);
            if (lane == 0) sMask[q][wv] = m;
        }
        __syncthreads();
        const uint32_t base_idx = (uint32_t)(r * RDG_BATCH);
#pragma unroll 1
        for (int s = 0; s < 4; ++s) {
            if (rdg_all(amin > 1.0f)) break;   // once per 64 staged splats; inside the walk only after a pixel stops
            unsigned long long mask = rdg_uniform_u64(sMask[wv][s]);
            unsigned long long seen = 0ull;    // splats of this 64-slot word that some pixel of the quadrant may blend
            while (mask) {
                const int jb = __builtin_ctzll(mask);
                const int j = s * 64 + jb;
                mask &= mask - 1;
                const float4 q0 = sQ0[j];
                const float4 q1 = sQ1[j];
                const float dx = q0.x - pixx, dy = q0.y - pixy;
                float t_, v_;
                const float power = rdg_log2_gauss(q0.z, q0.w, q1.x, dx, dy, t_, v_);
                const float alpha = fminf(RDG_ALPHA_CAP, q1.y * __builtin_amdgcn_exp2f(power));
                const bool cand = power <= 0.0f && alpha >= amin;
                if (!rdg_any(cand)) continue;
                seen |= 1ull << jb;
                const float test_T = T * (1.0f - alpha);
                const bool stop = cand && test_T < RDG_T_STOP;
                const bool hit = cand && !stop;
                const float wgt = hit ? alpha * T : 0.0f;
                const float4 q2 = sQ2[j];
                C0 += wgt * q2.x; C1 += wgt * q2.y; C2 += wgt * q2.z;
                Dp += wgt * q1.z;
                if (render_normal) {
                    const float4 q3 = sQ3[j];
                    N0 += wgt * q3.x; N1 += wgt * q3.y; N2 += wgt * q3.z;
                }
                if (hit) { T = test_T; last_contributor = base_idx + (uint32_t)j + 1u; }
                if (rdg_any(stop)) {
                    amin = stop ? __builtin_inff() : amin;
                    if (rdg_all(amin > 1.0f)) { mask = 0ull; }
                }
            }
            if (lane == 0 && seen) hit[(size_t)(r * 4 + s) * 4 + wv] = seen;
        }
    }
    if (inside) {
        const size_t hw = (size_t)H * W;
        const size_t pid = (size_t)pyi * W + pxi;
        final_T[pid] = T;
        n_contrib[pid] = last_contributor;
        out_color[pid] = C0 + T * bg[0];
        out_color[hw + pid] = C1 + T * bg[1];
        out_color[2 * hw + pid] = C2 + T * bg[2];
        out_depth[pid] = Dp;
        out_alpha[pid] = 1.0f - T;
        out_normal[pid] = N0; out_normal[hw + pid] = N1; out_normal[2 * hw + pid] = N2;
    }
}

int rdg_launch_render_fwd(const RdgDev& d, const float* bg, const void* geom_ws, void* bin_ws,
                          int64_t capacity, void* image_ws, const int32_t* num_rendered, float* out_color,
                          float* out_depth, float* out_normal, float* out_alpha, hipStream_t s) {
    const RdgGeomLayout G = rdg_geom_layout(d.P);
    const RdgBinLayout B = rdg_bin_layout(capacity);
    const RdgImageLayout I = rdg_image_layout(d.H, d.W);
    const int n_tiles = d.gx * d.gy;
    const int npass = (rdg_key_bits(n_tiles) + RDG_SORT_BITS - 1) / RDG_SORT_BITS;
    const char* b = (const char*)bin_ws;
    const uint32_t* plist = (const uint32_t*)(b + ((npass & 1) ? B.vals_b : B.vals_a));
    char* im = (char*)image_ws;
    const int nblk = ((n_tiles + 7) / 8) * 8;
    // zeroed by the binning stage (rdg_launch_bin), which always runs before this launch
    unsigned long long* hitbits = (unsigned long long*)((char*)bin_ws + B.hit);
    hipLaunchKernelGGL(rdg_render_fwd_kernel, dim3(nblk), dim3(256), 0, s, d.W, d.H, d.gx, n_tiles, d.render_normal,
                       bg, (const uint2*)(im + I.ranges), plist, (const RdgRec*)((const char*)geom_ws + G.rec),
                       (long long)capacity, num_rendered, (float*)(im + I.final_T), (uint32_t*)(im + I.n_contrib),
                       out_color, out_depth, out_normal, out_alpha, hitbits);
    return rdg_check_hip(hipGetLastError(), "render_fwd launch");
}

#define RDG_RING 16
// Flush a wave's ring: 16 consecutive lanes = the 64-B accumulator row of one Gaussian, 4 entries per instruction,
// which is the access shape the global float-atomic unit runs at full rate for.
// `scale` = this lane's constant factor for component (lane & 15), see the derivative block of the kernel.
__device__ __forceinline__ void rdg_ring_flush(float (*ring)[4][12], const uint32_t* ids, int n, int lane, float scale,
                                               float* __restrict__ grow) {
    // LDS operations of one wave execute in program order; the fences only pin the compiler's ordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int c = lane & 15;
    for (int e0 = 0; e0 < n; e0 += 4) {
        const int e = e0 + (lane >> 4);
        if (e < n && c < 10) {
            const float* r = &ring[e][0][c];
            const float v = ((r[0] + r[12]) + (r[24] + r[36])) * scale;
            if (v != 0.0f) atomicAdd(grow + (size_t)ids[e] * RDG_GROW + c, v);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------
// HAS_DEPTH = false: no upstream gradient for the depth image (photometric-only losses) -- the depth channel drops out
// of the per-pair arithmetic and of the reduction.
template <bool HAS_DEPTH>
__global__ void __launch_bounds__(256)
rdg_render_bwd_kernel(int W, int H, int gx, int n_tiles, const float* __restrict__ bg,
                      const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      const RdgRec* __restrict__ rec, const float* __restrict__ final_T,
                      const uint32_t* __restrict__ n_contrib, const float* __restrict__ g_color,
                      const float* __restrict__ g_depth, const float* __restrict__ g_alpha,
                      float* __restrict__ grow, const unsigned long long* __restrict__ hitbits) {
    const int tile = rdg_tile_of_block(blockIdx.x, n_tiles);
    if (tile >= n_tiles) return;
    __shared__ float4 sQ0[RDG_BATCH], sQ1[RDG_BATCH], sQ2[RDG_BATCH];
    __shared__ uint32_t sId[RDG_BATCH];
    __shared__ float sRing[4][RDG_RING][4][12];   // per wave: [entry][16-lane row][component] partial sums
    __shared__ uint32_t sRingId[4][RDG_RING];
    __shared__ unsigned long long sMask[4][4];
    __shared__ int sMax[4];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int tx = tile % gx, ty = tile / gx;
    const int pxi = tx * RDG_TILE + (wv & 1) * 8 + (lane & 7);
    const int pyi = ty * RDG_TILE + (wv >> 1) * 8 + (lane >> 3);
    const bool inside = pxi < W && pyi < H;
    const float pixx = (float)pxi, pixy = (float)pyi;
    const uint2 range = ranges[tile];
    const size_t hw = (size_t)H * W;
    const size_t pid = (size_t)pyi * W + pxi;

    const unsigned long long* const hit = hitbits + ((size_t)(range.x >> 6) + (size_t)tile) * 4;
    const float T_final = inside ? final_T[pid] : 0.0f;
    float T = T_final;
    const int last_contributor = inside ? (int)n_contrib[pid] : 0;
    float dLp0 = 0.f, dLp1 = 0.f, dLp2 = 0.f, dLd = 0.f, dLa = 0.f;
    if (inside) {
        if (g_color) { dLp0 = g_color[pid]; dLp1 = g_color[hw + pid]; dLp2 = g_color[2 * hw + pid]; }
        if (HAS_DEPTH) dLd = g_depth[pid];
        if (g_alpha) dLa = g_alpha[pid];
    }
    const float bgdot = bg[0] * dLp0 + bg[1] * dLp1 + bg[2] * dLp2;
    const float tail = dLa - bgdot;  // d(out)/dT_final chain: alpha_out = 1 - T_final, colour += T_final*bg
    const float Ttail = T_final * tail;   // per-pixel constant of the dL/dalpha recurrence

    // wave / block maxima of last_contributor: splats at list positions >= max are skipped wholesale
    int wmax = last_contributor;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o));
    if (lane == 0) sMax[wv] = wmax;
    __syncthreads();
    const int m0 = sMax[0], m1 = sMax[1], m2 = sMax[2], m3 = sMax[3];
    const int kmax = max(max(m0, m1), max(m2, m3));
    const int rounds = (kmax + RDG_BATCH - 1) / RDG_BATCH;

    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, accd = 0.f;
    int ring_n = 0;   // wave-uniform fill level of this wave's ring
    const int fc = lane & 15;
    // components 0, 1 are the raw first moments sum(G dL/dG dx), sum(G dL/dG dy): the per-Gaussian backward turns them
    // into dL/dmean2D with the conic it has anyway (two multiplies and two fused multiply-adds per pair less here)
    const float flush_scale = (fc == 2 || fc == 4) ? -0.5f : fc == 3 ? -1.0f : (fc == 9 && !HAS_DEPTH) ? 0.0f : 1.0f;

    for (int r = 0; r < rounds; ++r) {
        const int kbase = kmax - 1 - r * RDG_BATCH;  // list position of slot 0 of this batch
        uint32_t qbits = 0;
        {
            const int k = kbase - tid;
            if (k >= 0) {
                // a quadrant whose pixels all stopped before this list position never needs the splat ...
                qbits = (uint32_t)(k < m0) | ((uint32_t)(k < m1) << 1) | ((uint32_t)(k < m2) << 2) |
                        ((uint32_t)(k < m3) << 3);
                // ... and neither does one in which the forward found no pixel that could blend it (same staged
                // values, same expression: every backward hit was a forward candidate)
                const unsigned long long* hw = hit + (size_t)(k >> 6) * 4;
                const int hb = k & 63;
                qbits &= (uint32_t)((hw[0] >> hb) & 1ull) | ((uint32_t)((hw[1] >> hb) & 1ull) << 1) |
                         ((uint32_t)((hw[2] >> hb) & 1ull) << 2) | ((uint32_t)((hw[3] >> hb) & 1ull) << 3);
                if (qbits) {   // nobody will look at the other slots: skip their record gathers
                    const uint32_t id = point_list[range.x + k];
                    const RdgRec* p = rec + id;
                    const float4 q0 = p->q0, q1 = p->q1;
                    sId[tid] = id;
                    sQ0[tid] = make_float4(q0.x, q0.y, RDG_NEG_LOG2E * q0.z, RDG_NEG_LOG2E * q0.w);
                    sQ1[tid] = make_float4(RDG_NEG_LOG2E * q1.x, q1.y, q1.z, 0.0f);
                    sQ2[tid] = p->q2;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned long long m = __ballot((qbits >> q) & 1u);
            if (lane == 0) sMask[q][wv] = m;
        }
        __syncthreads();
#pragma unroll 1
        for (int s = 0; s < 4; ++s) {
            unsigned long long mask = rdg_uniform_u64(sMask[wv][s]);
            while (mask) {
                const int j = s * 64 + __builtin_ctzll(mask);
                mask &= mask - 1;
                const int k = kbase - j;  // list position of this splat
                const float4 q0 = sQ0[j];
                const float4 q1 = sQ1[j];
                const float dx = q0.x - pixx, dy = q0.y - pixy;
                float t_, v_;
                const float power = rdg_log2_gauss(q0.z, q0.w, q1.x, dx, dy, t_, v_);
                const float G = __builtin_amdgcn_exp2f(power);
                const float alpha = fminf(RDG_ALPHA_CAP, q1.y * G);
                const bool hit = (k < last_contributor) && power <= 0.0f && alpha >= RDG_ALPHA_MIN;
                if (!rdg_any(hit)) continue;
                const float4 q2 = sQ2[j];
                // Branch-free per-pixel derivatives.  A lane that does not blend this splat runs the same instructions
                // with alpha_eff = 0, which leaves every piece of its state unchanged (T/(1-0) = T, B += 0*(c - B)) and
                // zeroes its contributions, so no per-variable selects are needed.  B* is the colour accumulated
                // BEHIND the current splat (back-to-front recurrence B <- alpha c + (1 - alpha) B), i.e. upstream's
                // accum_rec evaluated eagerly instead of through last_alpha / last_color.
                const float aeff = hit ? alpha : 0.0f;
                const float inv1ma = __builtin_amdgcn_rcpf(1.0f - aeff);
                T = T * inv1ma;
                const float dch = aeff * T;
                const float e0 = q2.x - acc0, e1 = q2.y - acc1, e2 = q2.z - acc2;
                float dL_dalpha = e0 * dLp0 + e1 * dLp1 + e2 * dLp2;
                acc0 += aeff * e0; acc1 += aeff * e1; acc2 += aeff * e2;
                if (HAS_DEPTH) {
                    const float ed = q1.z - accd;
                    dL_dalpha += ed * dLd;
                    accd += aeff * ed;
                }
                dL_dalpha = fmaf(dL_dalpha, T, Ttail * inv1ma);
                dL_dalpha = hit ? dL_dalpha : 0.0f;
                // The constant factors of the conic derivatives (-0.5, -1, -0.5) are applied once per flushed row total
                // (rdg_ring_flush), not per pixel-splat pair.
                const float g5 = G * dL_dalpha;                     // dL/dopacity; G dL/dG = opacity * g5 -- the opacity is a
                //                                                     per-splat constant: applied by the per-Gaussian backward
                const float g0 = g5 * dx, g1 = g5 * dy;             // first moments / opacity (-> dL/d(mean2D) per Gaussian)
                const float g2 = g0 * dx;                           // ~ dL/d(conic a)
                const float g3 = g0 * dy;                           // ~ dL/d(conic b)
                const float g4 = g1 * dy;                           // ~ dL/d(conic c)
                const float g6 = dch * dLp0, g7 = dch * dLp1, g8 = dch * dLp2, g9 = HAS_DEPTH ? dch * dLd : 0.0f;
                // Transposed wave reduction: instead of ten 6-step butterflies (60 DPP adds), fold the VALUE index into
                // the lane index while reducing.  DPP write masks work on quads (bank_mask: 4 lanes) and rows, so the
                // folding steps come FIRST and act across quads: a rotate-by-8 exchange turns two values into one (the
                // low half-row keeps the pair sum of the first value, the high half-row of the second), a half-row
                // mirror does it again between neighbouring quads, and two plain quad butterflies finish.  Each
                // "two values -> one" step is TWO instructions (one DPP add per bank_mask into the same register),
                // which the compiler cannot express (it needs two selects + one DPP add), hence the hand-scheduled
                // block: 21 VALU for the whole 10-value reduction.  DPP needs two wait states after the VALU write of
                // the register it reads: the order below keeps at least two instructions between every producer and
                // its DPP consumer; s_nop 1 covers the inputs.
                // Result: every lane of quad q (h = q >> 1, p = q & 1) holds the 16-lane row total of component
                // 2p + h (y0), 4 + 2p + h (y1), 8 + h (y2).
                float x0, x1, x2, x3, x4, y0, y1, y2;
                if (HAS_DEPTH) {
                    asm volatile(
                        "s_nop 1\n\t"
                        "v_add_f32_dpp %[x4], %[g8], %[g8] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                        "v_add_f32_dpp %[x4], %[g9], %[g9] row_ror:8 row_mask:0xf bank_mask:0xc"
                        : [x4] "=&v"(x4) : [g8] "v"(g8), [g9] "v"(g9));
                } else {
                    // both half-rows carry component 8 (the flush multiplies component 9 by zero)
                    asm volatile(
                        "s_nop 1\n\t"
                        "v_add_f32_dpp %[x4], %[g8], %[g8] row_ror:8 row_mask:0xf bank_mask:0xf"
                        : [x4] "=&v"(x4) : [g8] "v"(g8));
                }
                asm volatile(
                    "s_nop 1\n\t"
                    "v_add_f32_dpp %[x0], %[g0], %[g0] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                    "v_add_f32_dpp %[x0], %[g1], %[g1] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                    "v_add_f32_dpp %[x1], %[g2], %[g2] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                    "v_add_f32_dpp %[x1], %[g3], %[g3] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                    "v_add_f32_dpp %[x2], %[g4], %[g4] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                    "v_add_f32_dpp %[x2], %[g5], %[g5] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                    "v_add_f32_dpp %[x3], %[g6], %[g6] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                    "v_add_f32_dpp %[x3], %[g7], %[g7] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                    "v_add_f32_dpp %[y0], %[x0], %[x0] row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                    "v_add_f32_dpp %[y0], %[x1], %[x1] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                    "v_add_f32_dpp %[y1], %[x2], %[x2] row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                    "v_add_f32_dpp %[y1], %[x3], %[x3] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                    "v_add_f32_dpp %[y2], %[x4], %[x4] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                    "v_add_f32_dpp %[y0], %[y0], %[y0] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                    "v_add_f32_dpp %[y1], %[y1], %[y1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                    "v_add_f32_dpp %[y2], %[y2], %[y2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                    "v_add_f32_dpp %[y0], %[y0], %[y0] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                    "v_add_f32_dpp %[y1], %[y1], %[y1] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                    "v_add_f32_dpp %[y2], %[y2], %[y2] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                    : [x0] "=&v"(x0), [x1] "=&v"(x1), [x2] "=&v"(x2), [x3] "=&v"(x3),
                      [y0] "=&v"(y0), [y1] "=&v"(y1), [y2] "=&v"(y2)
                    : [g0] "v"(g0), [g1] "v"(g1), [g2] "v"(g2), [g3] "v"(g3), [g4] "v"(g4), [g5] "v"(g5),
                      [g6] "v"(g6), [g7] "v"(g7), [x4] "v"(x4));
                {
                    // One lane per quad parks the row totals in this wave's private ring with PLAIN LDS stores (LDS
                    // float atomics into a table shared by the 4 waves cost a third of the kernel); the 4 row partials
                    // are added when the ring is flushed.
                    if ((lane & 3) == 0) {
                        const int par = (lane >> 2) & 1, hh = (lane >> 3) & 1;
                        float* gr = &sRing[wv][ring_n][lane >> 4][2 * par + hh];
                        gr[0] = y0;
                        gr[4] = y1;
                        if (!par) gr[8] = y2;
                        if (lane == 0) sRingId[wv][ring_n] = sId[j];
                    }
                    if (++ring_n == RDG_RING) { rdg_ring_flush(sRing[wv], sRingId[wv], ring_n, lane, flush_scale, grow); ring_n = 0; }
                }
            }
        }
        __syncthreads();   // every wave is done with this round's staged records
    }
    rdg_ring_flush(sRing[wv], sRingId[wv], ring_n, lane, flush_scale, grow);
}

int rdg_launch_render_bwd(const RdgDev& d, const float* bg, const void* geom_ws, const void* bin_ws,
                          int64_t capacity, const void* image_ws, const float* g_color, const float* g_depth,
                          const float* g_alpha, float* grow, hipStream_t s) {
    const RdgGeomLayout G = rdg_geom_layout(d.P);
    const RdgBinLayout B = rdg_bin_layout(capacity);
    const RdgImageLayout I = rdg_image_layout(d.H, d.W);
    const int n_tiles = d.gx * d.gy;
    const int npass = (rdg_key_bits(n_tiles) + RDG_SORT_BITS - 1) / RDG_SORT_BITS;
    const char* b = (const char*)bin_ws;
    const uint32_t* plist = (const uint32_t*)(b + ((npass & 1) ? B.vals_b : B.vals_a));
    const char* im = (const char*)image_ws;
    const int nblk = ((n_tiles + 7) / 8) * 8;
#define RDG_BWD_LAUNCH(DEPTH)                                                                                      \
    hipLaunchKernelGGL(rdg_render_bwd_kernel<DEPTH>, dim3(nblk), dim3(256), 0, s, d.W, d.H, d.gx, n_tiles, bg,    \
                       (const uint2*)(im + I.ranges), plist, (const RdgRec*)((const char*)geom_ws + G.rec),        \
                       (const float*)(im + I.final_T), (const uint32_t*)(im + I.n_contrib), g_color, g_depth,      \
                       g_alpha, grow, (const unsigned long long*)(b + B.hit))
    if (g_depth) RDG_BWD_LAUNCH(true); else RDG_BWD_LAUNCH(false);
#undef RDG_BWD_LAUNCH
    return rdg_check_hip(hipGetLastError(), "render_bwd launch");
}
