// rdg_knn.hip -- simple_knn.distCUDA2 replacement: mean squared distance to the 3
// nearest other points.  Call site: /root/reference/src/model/rodygs_static.py:130-133 (init only).
//
// Exact 3-NN: points are ordered along a 48-bit Morton curve with the library's own radix sort and cut into leaf
// boxes of 16 consecutive points with their AABBs, grouped 16 by 16 into coarser levels (256, 4096 and -- for the K-NN walk --
// 65 536 points).
// A thread (one point, in Morton order, so a wave's 64 points are spatial neighbours and take the same branches)
// seeds its best-3 from its curve neighbours and then descends only into the boxes whose AABB is closer than its
// current 3rd-best distance.  The top level is streamed through LDS; lower levels are uniform (broadcast) loads.
#include "rdg_common.h"
#include <float.h>

#ifndef RDG_KNN_BOX
#define RDG_KNN_BOX 16    // points per leaf box (consecutive along the Morton curve)
#endif
#define RDG_KNN_SUP 16    // children per box of the two upper pruning levels (512 and 8192 points)
#define RDG_KNN_MMB 256   // workgroups of the bounding-box reduction
#define RDG_KNN_PBATCH 8  // leaf points fetched ahead of their distance tests

struct RdgKnnLayout {
    size_t minmax;      // float[8]
    size_t n_dev;       // int32
    size_t keys_a, keys_b, vals_a, vals_b, sort_tmp;
    size_t sorted;      // float4[P]
    size_t boxes;       // float4[2*nbox]
    size_t sboxes;      // float4[2*nsb]   AABBs of RDG_KNN_SUP consecutive boxes
    size_t tboxes;      // float4[2*ntb]   AABBs of RDG_KNN_SUP consecutive super-boxes
    size_t qboxes;      // float4[2*nqb]   AABBs of RDG_KNN_SUP consecutive top boxes (65 536 points)
    size_t mm_part;     // float[RDG_KNN_MMB][8]
    size_t total;
};
static RdgKnnLayout rdg_knn_layout(int32_t P) {
    RdgKnnLayout L;
    size_t Pp = (size_t)(P > 0 ? P : 1);
    size_t nbox = (Pp + RDG_KNN_BOX - 1) / RDG_KNN_BOX;
    size_t o = 0;
    L.minmax = o;   o = rdg_align_up(o + 32, 256);
    L.n_dev = o;    o = rdg_align_up(o + 4, 256);
    L.keys_a = o;   o = rdg_align_up(o + Pp * 8, 256);
    L.keys_b = o;   o = rdg_align_up(o + Pp * 8, 256);
    L.vals_a = o;   o = rdg_align_up(o + Pp * 4, 256);
    L.vals_b = o;   o = rdg_align_up(o + Pp * 4, 256);
    L.sort_tmp = o; o = rdg_align_up(o + rdg_sort_layout((int64_t)Pp).total, 256);
    L.sorted = o;   o = rdg_align_up(o + Pp * 16, 256);
    L.boxes = o;    o = rdg_align_up(o + nbox * 32, 256);
    const size_t nsb = (nbox + RDG_KNN_SUP - 1) / RDG_KNN_SUP;
    L.sboxes = o;   o = rdg_align_up(o + nsb * 32, 256);
    L.tboxes = o;   o = rdg_align_up(o + ((nsb + RDG_KNN_SUP - 1) / RDG_KNN_SUP) * 32, 256);
    L.qboxes = o;   o = rdg_align_up(o + ((((nsb + RDG_KNN_SUP - 1) / RDG_KNN_SUP) + RDG_KNN_SUP - 1) / RDG_KNN_SUP) * 32, 256);
    L.mm_part = o;  o = rdg_align_up(o + (size_t)RDG_KNN_MMB * 32, 256);
    L.total = o;
    return L;
}

// bounding box of the cloud: RDG_KNN_MMB workgroups write partial boxes, one small workgroup finishes
__global__ void __launch_bounds__(256) rdg_knn_minmax_partial_kernel(int P, const float* __restrict__ pts,
                                                                     float* __restrict__ part) {
    __shared__ float s[6][4];
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = pts[3 * i + c];
            mn[c] = fminf(mn[c], v); mx[c] = fmaxf(mx[c], v);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o));
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        for (int c = 0; c < 3; ++c) { s[c][w] = mn[c]; s[3 + c][w] = mx[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        part[blockIdx.x * 8 + c] = fminf(fminf(s[c][0], s[c][1]), fminf(s[c][2], s[c][3]));
        part[blockIdx.x * 8 + 4 + c] = fmaxf(fmaxf(s[3 + c][0], s[3 + c][1]), fmaxf(s[3 + c][2], s[3 + c][3]));
    }
}
__global__ void __launch_bounds__(64) rdg_knn_minmax_final_kernel(int P, int nblk, const float* __restrict__ part,
                                                                  float* __restrict__ minmax, int32_t* n_dev) {
    const int lane = threadIdx.x;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int b = lane; b < nblk; b += 64) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { mn[c] = fminf(mn[c], part[b * 8 + c]); mx[c] = fmaxf(mx[c], part[b * 8 + 4 + c]); }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o));
        }
    }
    if (lane == 0) {
        for (int c = 0; c < 3; ++c) { minmax[c] = mn[c]; minmax[4 + c] = mx[c]; }
        *n_dev = P;
    }
}

// Curve code: RDG_KNN_BITS bits per axis over the cloud's bounding box.  16 bits (48-bit keys, six 8-bit sort passes): with
// the 10 bits of rounds 1-3 a cloud whose bounding box is set by a few far points -- or that has a dense core -- put
// thousands of points into ONE cell of the grid, in arbitrary order, and every leaf of such a cell overlaps every other
// (2 M points with a dense core: 77 ms against 4.4 ms for a uniform cloud).
#define RDG_KNN_BITS 16
__device__ __forceinline__ uint64_t rdg_expand21(uint64_t v) {      // bit k of v (k < 21) -> bit 3 k
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x001f00000000ffffull;
    v = (v | (v << 16)) & 0x001f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
__device__ __forceinline__ uint64_t rdg_knn_code(float x, float y, float z, const float* __restrict__ minmax) {
    const float p[3] = {x, y, z};
    const float top = (float)((1u << RDG_KNN_BITS) - 1u);
    uint64_t code = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float lo = minmax[c], hi = minmax[4 + c];
        const float ext = hi - lo;
        float t = ext > 0.f ? (p[c] - lo) / ext : 0.f;
        t = fminf(fmaxf(t * top, 0.0f), top);
        code |= rdg_expand21((uint64_t)(uint32_t)t) << (2 - c);
    }
    return code;
}

__global__ void rdg_knn_morton_kernel(int P, const float* __restrict__ pts, const float* __restrict__ minmax,
                                      uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    keys[i] = rdg_knn_code(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], minmax);
    vals[i] = (uint32_t)i;
}

// sorted[i] = (x, y, z, original index) in curve order; one AABB per RDG_KNN_BOX consecutive points
__global__ void __launch_bounds__(256)
rdg_knn_gather_box_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order,
                          float4* __restrict__ sorted, float4* __restrict__ boxes) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (i < P) {
        const uint32_t src = order[i];
        const float x = pts[3 * src], y = pts[3 * src + 1], z = pts[3 * src + 2];
        sorted[i] = make_float4(x, y, z, __uint_as_float(src));
        mn[0] = mx[0] = x; mn[1] = mx[1] = y; mn[2] = mx[2] = z;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = RDG_KNN_BOX / 2; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o));
        }
    }
    if ((threadIdx.x & (RDG_KNN_BOX - 1)) == 0 && i < P) {
        const int b = i / RDG_KNN_BOX;
        boxes[2 * b] = make_float4(mn[0], mn[1], mn[2], 0.f);
        boxes[2 * b + 1] = make_float4(mx[0], mx[1], mx[2], 0.f);
    }
}

__global__ void rdg_knn_superbox_kernel(int nbox, const float4* __restrict__ boxes, float4* __restrict__ sboxes) {
    const int sb = blockIdx.x * blockDim.x + threadIdx.x;
    const int b0 = sb * RDG_KNN_SUP;
    if (b0 >= nbox) return;
    float4 lo = boxes[2 * b0], hi = boxes[2 * b0 + 1];
    for (int b = b0 + 1; b < min(nbox, b0 + RDG_KNN_SUP); ++b) {
        const float4 l = boxes[2 * b], h = boxes[2 * b + 1];
        lo.x = fminf(lo.x, l.x); lo.y = fminf(lo.y, l.y); lo.z = fminf(lo.z, l.z);
        hi.x = fmaxf(hi.x, h.x); hi.y = fmaxf(hi.y, h.y); hi.z = fmaxf(hi.z, h.z);
    }
    sboxes[2 * sb] = lo; sboxes[2 * sb + 1] = hi;
}

__device__ __forceinline__ float rdg_knn_box_d2(const float4 lo, const float4 hi, float x, float y, float z) {
    const float ex = fmaxf(fmaxf(lo.x - x, x - hi.x), 0.f);
    const float ey = fmaxf(fmaxf(lo.y - y, y - hi.y), 0.f);
    const float ez = fmaxf(fmaxf(lo.z - z, z - hi.z), 0.f);
    return ex * ex + ey * ey + ez * ez;
}

__device__ __forceinline__ void rdg_knn_insert(float d, float& b0, float& b1, float& b2) {
    if (d < b2) {
        if (d < b1) {
            b2 = b1;
            if (d < b0) { b1 = b0; b0 = d; } else { b1 = d; }
        } else {
            b2 = d;
        }
    }
}

__global__ void __launch_bounds__(256)
rdg_knn_search_kernel(int P, int nbox, const float4* __restrict__ sorted, const float4* __restrict__ boxes,
                      const float4* __restrict__ sboxes, const float4* __restrict__ tboxes, float* __restrict__ out) {
    __shared__ float4 sBox[2 * 256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool act = i < P;
    float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    if (act) {
        me = sorted[i];
        for (int j = max(0, i - 3); j <= min(P - 1, i + 3); ++j) {
            if (j == i) continue;
            const float4 o = sorted[j];
            const float dx = me.x - o.x, dy = me.y - o.y, dz = me.z - o.z;
            rdg_knn_insert(dx * dx + dy * dy + dz * dz, b0, b1, b2);
        }
    }
    const int nsb = (nbox + RDG_KNN_SUP - 1) / RDG_KNN_SUP, ntb = (nsb + RDG_KNN_SUP - 1) / RDG_KNN_SUP;
    for (int base = 0; base < ntb; base += 256) {
        __syncthreads();
        const int nb = min(256, ntb - base);
        if ((int)threadIdx.x < nb) {
            sBox[2 * threadIdx.x] = tboxes[2 * (base + threadIdx.x)];
            sBox[2 * threadIdx.x + 1] = tboxes[2 * (base + threadIdx.x) + 1];
        }
        __syncthreads();
        for (int k = 0; k < nb; ++k) {
            const bool near_top = act && rdg_knn_box_d2(sBox[2 * k], sBox[2 * k + 1], me.x, me.y, me.z) <= b2;
            if (__builtin_amdgcn_ballot_w64(near_top) == 0ull) continue;
            const int sx0 = (base + k) * RDG_KNN_SUP, sx1 = min(nsb, sx0 + RDG_KNN_SUP);
            for (int sx = sx0; sx < sx1; ++sx) {
                const bool near_sup = act && rdg_knn_box_d2(sboxes[2 * sx], sboxes[2 * sx + 1], me.x, me.y, me.z) <= b2;
                if (__builtin_amdgcn_ballot_w64(near_sup) == 0ull) continue;
                const int bx0 = sx * RDG_KNN_SUP, bx1 = min(nbox, bx0 + RDG_KNN_SUP);
                for (int bx = bx0; bx < bx1; ++bx) {
                    const bool visit = act && rdg_knn_box_d2(boxes[2 * bx], boxes[2 * bx + 1], me.x, me.y, me.z) <= b2;
                    if (__builtin_amdgcn_ballot_w64(visit) == 0ull) continue;
                    if (visit) {
                        const int s0 = bx * RDG_KNN_BOX, s1 = min(P, s0 + RDG_KNN_BOX);
                        for (int j = s0; j < s1; ++j) {
                            if (j == i || (j >= i - 3 && j <= i + 3)) continue;  // seeds already counted
                            const float4 o = sorted[j];
                            const float dx = me.x - o.x, dy = me.y - o.y, dz = me.z - o.z;
                            rdg_knn_insert(dx * dx + dy * dy + dz * dz, b0, b1, b2);
                        }
                    }
                }
            }
        }
    }
    if (act) out[__float_as_uint(me.w)] = (b0 + b1 + b2) / 3.0f;
}

// ---------------------------------------------------------------------------------------------------------
// K nearest neighbours + gather: the pytorch3d.ops.knn_points / knn_gather pair RigidityLoss is built on
// (/root/reference/src/trainer/losses.py:235-331; pytorch3d is the third un-vendored native dependency,
// .gitmodules:11-13).  Same machinery as above -- targets Morton-sorted into leaf boxes with AABBs under two coarser levels -- with a
// register-resident sorted best-K list per query.  Self mode (queries ARE the targets, the only form the
// reference uses) walks the queries in Morton order so a wave's 64 queries prune the same boxes; the general
// mode locates each query on the targets' curve by binary search over the sorted codes and seeds from there.
// Distances are squared Euclidean, ascending; a self query returns itself first (distance 0), as pytorch3d does.
// ---------------------------------------------------------------------------------------------------------
// Inserts (d, id) into the ascending best list (the caller has checked d < bd[KM - 1]): the new element sinks from the front,
// every slot keeps the smaller of itself and what arrives and hands on the larger; an element equal to a kept one stays behind
// it (first found, first listed).  One compare and four selects per slot on named registers -- the form "append, then swap
// upwards" compiled into chains of selects over the WHOLE array per swap (register arrays indexed by a run-time position).
template <int KM>
__device__ __forceinline__ void rdg_knn_push(float d, uint32_t id, float (&bd)[KM], uint32_t (&bi)[KM]) {
#pragma unroll
    for (int s = 0; s < KM; ++s) {
        const bool lt = d < bd[s];
        const float td = lt ? bd[s] : d;
        const uint32_t ti = lt ? bi[s] : id;
        bd[s] = lt ? d : bd[s];
        bi[s] = lt ? id : bi[s];
        d = td; id = ti;
    }
}

// squared distance (the compiler contracts the sum into fused multiply-adds, as the reference's CUDA build does with its
// own accumulation: neither is the other's bits; neighbours closer than a rounding error apart may swap places)
__device__ __forceinline__ float rdg_knn_d2(float dx, float dy, float dz) { return dx * dx + dy * dy + dz * dz; }

template <int KM>
__global__ void __launch_bounds__(256)
rdg_knn_points_kernel(int Pq, int Pt, int nbox, int K, int self_mode, const float* __restrict__ queries,
                      const float4* __restrict__ sorted, const uint64_t* __restrict__ tkeys,
                      const float* __restrict__ minmax, const float4* __restrict__ boxes,
                      const float4* __restrict__ sboxes, const float4* __restrict__ tboxes,
                      const float4* __restrict__ qboxes, float* __restrict__ dists, long long* __restrict__ idx) {
    __shared__ float4 sBox[2 * 256];
    __shared__ float4 sQ[2 * (256 / RDG_KNN_SUP)];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool act = i < Pq;
    float mx = 0.f, my = 0.f, mz = 0.f;
    long long out_row = 0;
    int pos = 0;
    float bd[KM]; uint32_t bi[KM];
#pragma unroll
    for (int s = 0; s < KM; ++s) { bd[s] = FLT_MAX; bi[s] = 0u; }
    int s_lo = 0, s_hi = -1;
    if (act) {
        if (self_mode) {
            const float4 me = sorted[i];
            mx = me.x; my = me.y; mz = me.z;
            out_row = (long long)__float_as_uint(me.w);
            pos = i;
        } else {
            mx = queries[3 * i]; my = queries[3 * i + 1]; mz = queries[3 * i + 2];
            out_row = i;
            const uint64_t code = rdg_knn_code(mx, my, mz, minmax);
            int lo = 0, hi = Pt;          // first sorted target whose code is >= the query's
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (tkeys[mid] < code) lo = mid + 1; else hi = mid; }
            pos = min(lo, Pt - 1);
        }
        if (!self_mode) {
            s_lo = max(0, pos - KM); s_hi = min(Pt - 1, pos + KM);
            for (int j = s_lo; j <= s_hi; ++j) {
                const float4 o = sorted[j];
                const float d = rdg_knn_d2(mx - o.x, my - o.y, mz - o.z);
                if (d < bd[KM - 1]) rdg_knn_push<KM>(d, __float_as_uint(o.w), bd, bi);
            }
        }
    }
    if (self_mode) {
        // Self mode: the wave's 64 queries ARE 64 consecutive points of the curve.  Their own leaves and 32 points on either
        // side are scanned first (wave-uniform addresses: scalar loads, fetched in batches), so the walk below starts with a
        // K-th distance that is already close to final and prunes from its first box on (seeded from 2 K curve neighbours it
        // descended into everything near the START of the curve with a loose bound); those leaves are then skipped.
        const int w0 = __builtin_amdgcn_readfirstlane(i & ~63);
        if (w0 < Pt) {
            constexpr int PAD = RDG_KNN_BOX >= 32 ? 1 : 32 / RDG_KNN_BOX;      // leaves on either side: 32 points
            const int l0 = w0 / RDG_KNN_BOX, l1 = min(Pt - 1, w0 + 63) / RDG_KNN_BOX;
            s_lo = max(0, (l0 - PAD) * RDG_KNN_BOX); s_hi = min(Pt, (l1 + 1 + PAD) * RDG_KNN_BOX) - 1;
            for (int j0 = s_lo; j0 <= s_hi; j0 += RDG_KNN_PBATCH) {
                float4 o[RDG_KNN_PBATCH];
#pragma unroll
                for (int w = 0; w < RDG_KNN_PBATCH; ++w) o[w] = sorted[min(j0 + w, Pt - 1)];
#pragma unroll
                for (int w = 0; w < RDG_KNN_PBATCH; ++w) {
                    const float d = rdg_knn_d2(mx - o[w].x, my - o[w].y, mz - o[w].z);
                    if (act && j0 + w <= s_hi && d < bd[KM - 1]) rdg_knn_push<KM>(d, __float_as_uint(o[w].w), bd, bi);
                }
            }
        }
    }
    const int nsb = (nbox + RDG_KNN_SUP - 1) / RDG_KNN_SUP, ntb = (nsb + RDG_KNN_SUP - 1) / RDG_KNN_SUP;
    const int nqb = (ntb + RDG_KNN_SUP - 1) / RDG_KNN_SUP;
    for (int base = 0; base < ntb; base += 256) {
        __syncthreads();
        const int nb = min(256, ntb - base);
        if ((int)threadIdx.x < nb) {
            sBox[2 * threadIdx.x] = tboxes[2 * (base + threadIdx.x)];
            sBox[2 * threadIdx.x + 1] = tboxes[2 * (base + threadIdx.x) + 1];
        }
        // a fourth level over the top boxes (16 of them = 65 536 points): at 2 M points a wave tested all 488 top boxes,
        // a fifth of its instructions
        if ((int)threadIdx.x < 256 / RDG_KNN_SUP && base / RDG_KNN_SUP + (int)threadIdx.x < nqb) {
            sQ[2 * threadIdx.x] = qboxes[2 * (base / RDG_KNN_SUP + threadIdx.x)];
            sQ[2 * threadIdx.x + 1] = qboxes[2 * (base / RDG_KNN_SUP + threadIdx.x) + 1];
        }
        __syncthreads();
        for (int k = 0; k < nb; ++k) {
            if ((k & (RDG_KNN_SUP - 1)) == 0) {
                const int q = k / RDG_KNN_SUP;
                const bool near_q = act && rdg_knn_box_d2(sQ[2 * q], sQ[2 * q + 1], mx, my, mz) <= bd[KM - 1];
                if (__builtin_amdgcn_ballot_w64(near_q) == 0ull) { k += RDG_KNN_SUP - 1; continue; }
            }
            const bool near_top = act && rdg_knn_box_d2(sBox[2 * k], sBox[2 * k + 1], mx, my, mz) <= bd[KM - 1];
            if (__builtin_amdgcn_ballot_w64(near_top) == 0ull) continue;
            // The AABBs of a node's children and the points of a leaf sit at wave-uniform addresses (scalar loads).  One load,
            // one wait, one test per child and per point made the walk a chain of scalar-cache round trips (3.9 ms at 2 M
            // points, K = 8): the next child's AABB is fetched while the current one is tested and scanned, a leaf's points
            // RDG_KNN_PBATCH at a time.
            const int sx0 = (base + k) * RDG_KNN_SUP, sx1 = min(nsb, sx0 + RDG_KNN_SUP);
            float4 slo = sboxes[2 * sx0], shi = sboxes[2 * sx0 + 1];
            for (int sx = sx0; sx < sx1; ++sx) {
                const int sxn = min(sx + 1, nsb - 1);
                const float4 slo_n = sboxes[2 * sxn], shi_n = sboxes[2 * sxn + 1];
                const bool near_sup = act && rdg_knn_box_d2(slo, shi, mx, my, mz) <= bd[KM - 1];
                slo = slo_n; shi = shi_n;
                if (__builtin_amdgcn_ballot_w64(near_sup) == 0ull) continue;
                const int bx0 = sx * RDG_KNN_SUP, bx1 = min(nbox, bx0 + RDG_KNN_SUP);
                float4 blo = boxes[2 * bx0], bhi = boxes[2 * bx0 + 1];
                for (int bx = bx0; bx < bx1; ++bx) {
                    const int bxn = min(bx + 1, nbox - 1);
                    const float4 blo_n = boxes[2 * bxn], bhi_n = boxes[2 * bxn + 1];
                    const int b0 = bx * RDG_KNN_BOX, b1 = min(Pt, b0 + RDG_KNN_BOX);
                    const bool done = self_mode && b0 >= s_lo && b1 - 1 <= s_hi;       // scanned before the walk
                    const bool visit = act && !done && rdg_knn_box_d2(blo, bhi, mx, my, mz) <= bd[KM - 1];
                    blo = blo_n; bhi = bhi_n;
                    if (__builtin_amdgcn_ballot_w64(visit) == 0ull) continue;
                    if (visit) {
                        for (int j0 = b0; j0 < b1; j0 += RDG_KNN_PBATCH) {
                            float4 o[RDG_KNN_PBATCH];
#pragma unroll
                            for (int w = 0; w < RDG_KNN_PBATCH; ++w) o[w] = sorted[min(j0 + w, Pt - 1)];
#pragma unroll
                            for (int w = 0; w < RDG_KNN_PBATCH; ++w) {
                                const int j = j0 + w;
                                const float d = rdg_knn_d2(mx - o[w].x, my - o[w].y, mz - o[w].z);
                                // not a seed (already taken), inside the leaf, and better than the current K-th
                                if (j < b1 && (j < s_lo || j > s_hi) && d < bd[KM - 1])
                                    rdg_knn_push<KM>(d, __float_as_uint(o[w].w), bd, bi);
                            }
                        }
                    }
                }
            }
        }
    }
    if (act) {
#pragma unroll
        for (int s = 0; s < KM; ++s) {
            if (s < K) { dists[out_row * K + s] = bd[s]; idx[out_row * K + s] = (long long)bi[s]; }
        }
    }
}

// dL/dp1[q] += sum_k g[q,k] * 2 (p1[q] - p2[idx[q,k]]);  dL/dp2[idx[q,k]] -= the same (float atomics).
__global__ void rdg_knn_points_bwd_kernel(int Pq, int K, const float* __restrict__ p1, const float* __restrict__ p2,
                                          const long long* __restrict__ idx, const float* __restrict__ g,
                                          float* __restrict__ d_p1, float* __restrict__ d_p2) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Pq) return;
    const float x = p1[3 * q], y = p1[3 * q + 1], z = p1[3 * q + 2];
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int k = 0; k < K; ++k) {
        const long long t = idx[(long long)q * K + k];
        const float w = 2.0f * g[(long long)q * K + k];
        const float dx = w * (x - p2[3 * t]), dy = w * (y - p2[3 * t + 1]), dz = w * (z - p2[3 * t + 2]);
        ax += dx; ay += dy; az += dz;
        if (d_p2) { atomicAdd(&d_p2[3 * t], -dx); atomicAdd(&d_p2[3 * t + 1], -dy); atomicAdd(&d_p2[3 * t + 2], -dz); }
    }
    if (d_p1) { atomicAdd(&d_p1[3 * q], ax); atomicAdd(&d_p1[3 * q + 1], ay); atomicAdd(&d_p1[3 * q + 2], az); }
}

// out[q,k,:] = x[idx[q,k],:]  (U floats per row); backward: d_x[idx[q,k],:] += g[q,k,:]
__global__ void rdg_knn_gather_kernel(long long n_rows, int U, const float* __restrict__ x,
                                      const long long* __restrict__ idx, float* __restrict__ out) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_rows * U) return;
    const long long r = e / U; const int u = (int)(e - r * U);
    out[e] = x[idx[r] * U + u];
}
__global__ void rdg_knn_gather_bwd_kernel(long long n_rows, int U, const float* __restrict__ g,
                                          const long long* __restrict__ idx, float* __restrict__ d_x) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_rows * U) return;
    const long long r = e / U; const int u = (int)(e - r * U);
    atomicAdd(&d_x[idx[r] * U + u], g[e]);
}

extern "C" {

size_t rdg_knn_tmp_bytes(int32_t P) { return rdg_knn_layout(P).total; }

/* Sorts the targets along the Morton curve and builds the box AABBs (shared by both entry points). */
static int rdg_knn_prepare(int32_t P, const float* points, char* t, const RdgKnnLayout& L, int* in_b, hipStream_t st) {
    float* minmax = (float*)(t + L.minmax);
    int32_t* n_dev = (int32_t*)(t + L.n_dev);
    uint64_t* keys_a = (uint64_t*)(t + L.keys_a); uint64_t* keys_b = (uint64_t*)(t + L.keys_b);
    uint32_t* vals_a = (uint32_t*)(t + L.vals_a); uint32_t* vals_b = (uint32_t*)(t + L.vals_b);
    const int nbox = (P + RDG_KNN_BOX - 1) / RDG_KNN_BOX;
    int mmb = (P + 255) / 256;
    if (mmb > RDG_KNN_MMB) mmb = RDG_KNN_MMB;
    hipLaunchKernelGGL(rdg_knn_minmax_partial_kernel, dim3(mmb), dim3(256), 0, st, P, points, (float*)(t + L.mm_part));
    hipLaunchKernelGGL(rdg_knn_minmax_final_kernel, dim3(1), dim3(64), 0, st, P, mmb, (const float*)(t + L.mm_part),
                       minmax, n_dev);
    hipLaunchKernelGGL(rdg_knn_morton_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, points, minmax, keys_a, vals_a);
    int rc = rdg_launch_sort(keys_a, keys_b, vals_a, vals_b, (int64_t)P, n_dev, 3 * RDG_KNN_BITS, t + L.sort_tmp, in_b, st);
    if (rc) return rc;
    hipLaunchKernelGGL(rdg_knn_gather_box_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, points,
                       *in_b ? vals_b : vals_a, (float4*)(t + L.sorted), (float4*)(t + L.boxes));
    const int nsb = (nbox + RDG_KNN_SUP - 1) / RDG_KNN_SUP, ntb = (nsb + RDG_KNN_SUP - 1) / RDG_KNN_SUP;
    hipLaunchKernelGGL(rdg_knn_superbox_kernel, dim3((nsb + 63) / 64), dim3(64), 0, st, nbox,
                       (const float4*)(t + L.boxes), (float4*)(t + L.sboxes));
    hipLaunchKernelGGL(rdg_knn_superbox_kernel, dim3((ntb + 63) / 64), dim3(64), 0, st, nsb,
                       (const float4*)(t + L.sboxes), (float4*)(t + L.tboxes));
    const int nqb = (ntb + RDG_KNN_SUP - 1) / RDG_KNN_SUP;
    hipLaunchKernelGGL(rdg_knn_superbox_kernel, dim3((nqb + 63) / 64), dim3(64), 0, st, ntb,
                       (const float4*)(t + L.tboxes), (float4*)(t + L.qboxes));
    return 0;
}

int rdg_dist2_knn3(int32_t P, const float* points, float* out, void* tmp_ws, void* stream) {
    if (P <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const RdgKnnLayout L = rdg_knn_layout(P);
    char* t = (char*)tmp_ws;
    int in_b = 0;
    int rc = rdg_knn_prepare(P, points, t, L, &in_b, st);
    if (rc) return rc;
    const int nbox = (P + RDG_KNN_BOX - 1) / RDG_KNN_BOX;
    hipLaunchKernelGGL(rdg_knn_search_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, nbox,
                       (const float4*)(t + L.sorted), (const float4*)(t + L.boxes), (const float4*)(t + L.sboxes),
                       (const float4*)(t + L.tboxes), out);
    return rdg_check_hip(hipGetLastError(), "knn launch");
}

int rdg_knn_points_forward(int32_t Pq, int32_t Pt, int32_t K, const float* queries, const float* targets, float* dists,
                           int64_t* idx, void* tmp_ws, void* stream) {
    if (Pq <= 0) return 0;
    if (K < 1 || K > 32) return rdg_set_error("knn_points: K must be 1..32 (got %d)", K);
    if (Pt < K) return rdg_set_error("knn_points: fewer targets (%d) than K (%d)", Pt, K);
    hipStream_t st = (hipStream_t)stream;
    const RdgKnnLayout L = rdg_knn_layout(Pt);
    char* t = (char*)tmp_ws;
    int in_b = 0;
    int rc = rdg_knn_prepare(Pt, targets, t, L, &in_b, st);
    if (rc) return rc;
    const int self_mode = (queries == targets && Pq == Pt) ? 1 : 0;
    const int nbox = (Pt + RDG_KNN_BOX - 1) / RDG_KNN_BOX;
    const uint64_t* tkeys = (const uint64_t*)(t + (in_b ? L.keys_b : L.keys_a));
#define RDG_KNN_LAUNCH(KM)                                                                                        \
    hipLaunchKernelGGL(rdg_knn_points_kernel<KM>, dim3((Pq + 255) / 256), dim3(256), 0, st, Pq, Pt, nbox, K, self_mode, \
                       queries, (const float4*)(t + L.sorted), tkeys, (const float*)(t + L.minmax),                \
                       (const float4*)(t + L.boxes), (const float4*)(t + L.sboxes), (const float4*)(t + L.tboxes),                        \
                       (const float4*)(t + L.qboxes), dists, \
                       (long long*)idx)
    if (K <= 4) RDG_KNN_LAUNCH(4); else if (K <= 8) RDG_KNN_LAUNCH(8); else if (K <= 16) RDG_KNN_LAUNCH(16);
    else RDG_KNN_LAUNCH(32);
#undef RDG_KNN_LAUNCH
    return rdg_check_hip(hipGetLastError(), "knn_points launch");
}

int rdg_knn_points_backward(int32_t Pq, int32_t Pt, int32_t K, const float* queries, const float* targets,
                            const int64_t* idx, const float* g_dists, float* d_queries, float* d_targets, void* stream) {
    if (Pq <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if (d_queries) e = rdg_zero_async(d_queries, (size_t)Pq * 12, st);
    if (e == hipSuccess && d_targets && d_targets != d_queries) e = rdg_zero_async(d_targets, (size_t)Pt * 12, st);
    if (e != hipSuccess) return rdg_check_hip(e, "knn_points_bwd memset");
    hipLaunchKernelGGL(rdg_knn_points_bwd_kernel, dim3((Pq + 255) / 256), dim3(256), 0, st, Pq, K, queries, targets,
                       (const long long*)idx, g_dists, d_queries, d_targets);
    return rdg_check_hip(hipGetLastError(), "knn_points_bwd launch");
}

int rdg_knn_gather_forward(int64_t n_rows, int32_t U, const float* x, const int64_t* idx, float* out, void* stream) {
    if (n_rows <= 0 || U <= 0) return 0;
    const long long n = n_rows * U;
    hipLaunchKernelGGL(rdg_knn_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (long long)n_rows, U, x, (const long long*)idx, out);
    return rdg_check_hip(hipGetLastError(), "knn_gather launch");
}

int rdg_knn_gather_backward(int64_t n_rows, int32_t U, int64_t n_src_rows, const float* g, const int64_t* idx, float* d_x,
                            void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n_src_rows > 0 && U > 0) {
        hipError_t e = rdg_zero_async(d_x, (size_t)n_src_rows * U * 4, st);
        if (e != hipSuccess) return rdg_check_hip(e, "knn_gather_bwd memset");
    }
    if (n_rows <= 0 || U <= 0) return 0;
    const long long n = n_rows * U;
    hipLaunchKernelGGL(rdg_knn_gather_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (long long)n_rows,
                       U, g, (const long long*)idx, d_x);
    return rdg_check_hip(hipGetLastError(), "knn_gather_bwd launch");
}

}  // extern "C"

// f / d for 0 <= f < 2^52 through one double multiply and a correction step: the 64-bit integer division the compiler emits is
// ~100 instructions, and the kernels below need one per (Gaussian, time) pair (the reference's f / nt row pairing)
__device__ __forceinline__ long long rdg_div_pos(long long f, long long d, double inv_d) {
    long long q = (long long)((double)f * inv_d);
    const long long r = f - q * d;
    if (r < 0) --q; else if (r >= d) ++q;
    return q;
}


// ---------------------------------------------------------------------------------------------------------
// Distance-preserving term of RigidityLoss (/root/reference/src/trainer/losses.py:293-358), the part that the
// reference materialises as several [t, n, K, 3] tensors (1.2 GB each at n = 500 k, t = 25): for every sampled
// Gaussian i, neighbour k and drawn time tau,
//     gap = || (canon_nn + own_nn(tau)) - (canon_i + own_i(tau)) ||,
//     term = sqrt((gap - y)^2 + eps^2),   y = d2_flat[f / nt],  f = (tau * n + i) * K + k
// (the reference views the [t, n, K] block of gaps as rows of nt consecutive entries and compares row r with the
// r-th squared neighbour distance; kept).  Outputs are the SUM of the terms and its unscaled gradients.
//
// The positions are stored GAUSSIAN-major: P3 [n][nt][3], the nt positions of a sampled Gaussian in one contiguous row
// (300 B at nt = 25) -- which is how the caller's translation tensor [n, nt, 3] is laid out anyway.
// A group of LPG lanes (LPG = 8 ... 64, the power of two >= nt) owns one Gaussian, lane = drawn time: an edge end is then ONE
// coalesced row read for the group.  (Rounds 1-4 kept time-major slabs [nt][n] float4 with a thread per (time, Gaussian):
// one 16-B gather per (time, edge end), 0.8 G gathers at n = 2 M, nt = 25, served at the rate the L2 hands out 64-B
// sectors: 9.0 ms against 4.7 ms for the three launches below on the same cloud.)  Launches:
//   out : the K edges leaving every Gaussian: the terms, the d2 gradient, -u on this end; the factor ig = s / gap of every
//         (edge, time) is kept (IG [n*K][nt], one coalesced row per edge);
//   in  : the edges arriving at every Gaussian (reverse adjacency): +ig (p - q) with the stored factor: one row of the
//         source's positions and one row of IG per edge -- no d2 gather, no division, no square root, and nothing to
//         balance: a group's trip count is its Gaussian's in-degree, groups are independent.
// The gradient leaves in the caller's layout and row order (G_own [n][nt][3], written by `out`, completed by `in`), with its
// sum over the times (the canonical position's gradient, G_canon [n][3]).
// ---------------------------------------------------------------------------------------------------------
#ifndef RDG_RIGR_ITERS
#define RDG_RIGR_ITERS 4       // Gaussians a lane group handles one after the other
#endif

// RA (optional, needs nt >= K): the K terms of a (Gaussian, time) pair fall into at most two consecutive rows of d2; their two
// partial d2 gradients are then STORED (RA [n][nt][2], coalesced) and rdg_rigidity_d2grad_kernel adds up every row's handful
// of contributions in a fixed order -- instead of 1.3 scattered float atomics per pair (66 M at n = 2 M, nt = 25: 1.5 of
// the 3.6 ms of this kernel, measured with them removed).
template <int LPG, int KT>
__global__ void __launch_bounds__(256)
rdg_rigidity_rows_out_kernel(long long n, int K_rt, int nt, const float* __restrict__ P3, const long long* __restrict__ nn_idx,
                             const float* __restrict__ d2, const long long* __restrict__ orig, float eps2,
                             double* __restrict__ loss_sum, float* __restrict__ IG, float2* __restrict__ RA,
                             float* __restrict__ G_own, float* __restrict__ d_d2) {
    constexpr int GPB = 256 / LPG;                 // lane groups per workgroup
    const int tl = threadIdx.x % LPG;
    const int K = KT > 0 ? KT : K_rt;
    const double inv_nt = 1.0 / (double)nt;
    const long long n_rows = n * K;
    const bool two_rows = nt >= K;                 // K consecutive flat entries touch at most two rows of d2
    double local = 0.0;
    for (int it = 0; it < RDG_RIGR_ITERS; ++it) {
        const long long s = ((long long)blockIdx.x * RDG_RIGR_ITERS + it) * GPB + threadIdx.x / LPG;
        if (s >= n) break;
        const long long i_row = orig ? orig[s] : s;
        long long nbr[KT > 0 ? KT : 1];
        if (KT > 0) {
#pragma unroll
            for (int k = 0; k < KT; ++k) nbr[k] = nn_idx[s * K + k];
        }
        for (int tau = tl; tau < nt; tau += LPG) {
            const float* pp = P3 + (s * nt + tau) * 3;
            const float px = pp[0], py = pp[1], pz = pp[2];
            const long long f0 = ((long long)tau * n + i_row) * K;
            long long row = rdg_div_pos(f0, nt, inv_nt);
            int rem = (int)(f0 - row * nt);
            float ax = 0.f, ay = 0.f, az = 0.f;
            float* ig_row = IG + (s * K) * nt + tau;
            if (two_rows) {
                const float y0 = d2[row], y1 = d2[row + 1 < n_rows ? row + 1 : row];
                const int k1 = nt - rem;                          // entries k >= k1 belong to the next row
                float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
                for (int k = 0; k < (KT > 0 ? KT : K); ++k) {
                    const float* qp = P3 + ((KT > 0 ? nbr[k] : nn_idx[s * K + k]) * nt + tau) * 3;
                    const float dx = qp[0] - px, dy = qp[1] - py, dz = qp[2] - pz;
                    const float gap = sqrtf(dx * dx + dy * dy + dz * dz);
                    const bool nx = k >= k1;
                    const float diff = gap - (nx ? y1 : y0);
                    const float term = sqrtf(diff * diff + eps2);
                    local += (double)term;
                    const float sg = diff / term;                 // d term / d gap ;  d term / d y = -sg
                    acc0 -= nx ? 0.f : sg; acc1 -= nx ? sg : 0.f;
                    const float ig = gap > 0.f ? sg / gap : 0.f;  // torch.norm backward is 0 at the origin
                    ig_row[(long long)k * nt] = ig;
                    ax -= ig * dx; ay -= ig * dy; az -= ig * dz;
                }
                if (RA) RA[s * nt + tau] = make_float2(acc0, acc1);
                else {
                    atomicAdd(&d_d2[row], acc0);
                    if (k1 < K) atomicAdd(&d_d2[row + 1], acc1);
                }
            } else {
                long long row_prev = -1;
                float row_acc = 0.f;
                for (int k = 0; k < K; ++k) {
                    const float* qp = P3 + ((KT > 0 ? nbr[k] : nn_idx[s * K + k]) * nt + tau) * 3;
                    const float dx = qp[0] - px, dy = qp[1] - py, dz = qp[2] - pz;
                    const float gap = sqrtf(dx * dx + dy * dy + dz * dz);
                    const float diff = gap - d2[row];
                    const float term = sqrtf(diff * diff + eps2);
                    local += (double)term;
                    const float sg = diff / term;
                    if (row != row_prev) {
                        if (row_prev >= 0) atomicAdd(&d_d2[row_prev], row_acc);
                        row_prev = row; row_acc = 0.f;
                    }
                    row_acc -= sg;
                    const float ig = gap > 0.f ? sg / gap : 0.f;
                    ig_row[(long long)k * nt] = ig;
                    ax -= ig * dx; ay -= ig * dy; az -= ig * dz;
                    if (++rem == nt) { rem = 0; ++row; }
                }
                if (row_prev >= 0) atomicAdd(&d_d2[row_prev], row_acc);
            }
            float* go = G_own + (i_row * nt + tau) * 3;
            go[0] = ax; go[1] = ay; go[2] = az;
        }
    }
    __shared__ double sh[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, (sh[0] + sh[1]) + (sh[2] + sh[3]));
}

// d_d2[r] = the stored partial gradients of the (time, Gaussian) pairs whose K flat entries [m K, (m + 1) K) meet row r =
// [r nt, (r + 1) nt): nt / K + 1 pairs at most, consecutive in the ORIGINAL sample order (m = tau n + i_row); a pair's first
// partial belongs to the row its first entry falls into, the second to the next row.  One thread per row, fixed order.
__global__ void __launch_bounds__(256)
rdg_rigidity_d2grad_kernel(long long n, int K, int nt, const long long* __restrict__ rank, const float2* __restrict__ RA,
                           float* __restrict__ d_d2) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n * K) return;
    const long long f_lo = r * nt;
    const long long m_lo = rdg_div_pos(f_lo, K, 1.0 / (double)K);
    const int cnt = (int)((f_lo + nt - 1 - m_lo * K) / K) + 1;          // pairs that meet the row (small numbers: 32-bit division)
    // the first pair's time, Gaussian and first row; the following pairs by increments
    long long tau = rdg_div_pos(m_lo, n, 1.0 / (double)n);
    long long i_row = m_lo - tau * n;
    // pair m's first entry lies in row (m K) / nt: r - 1 or r for the first pair (it meets row r), r for all the others
    bool first_in_prev = m_lo * K < f_lo;
    float acc = 0.f;
    for (int c0 = 0; c0 < cnt; c0 += 4) {
        long long sidx[4]; long long tt[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            tt[u] = tau;
            sidx[u] = c0 + u < cnt ? (rank ? rank[i_row] : i_row) : 0;
            if (++i_row == n) { i_row = 0; ++tau; }
        }
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = c0 + u < cnt ? RA[sidx[u] * nt + tt[u]] : make_float2(0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc += (c0 + u == 0 && first_in_prev) ? v[u].y : v[u].x;
        }
    }
    d_d2[r] = acc;
}

template <int LPG, int KT>
__global__ void __launch_bounds__(256)
rdg_rigidity_rows_in_kernel(long long n, int K_rt, int nt, const float* __restrict__ P3, const long long* __restrict__ rev_off,
                            const long long* __restrict__ rev_edge, const long long* __restrict__ orig,
                            const float* __restrict__ IG, float* __restrict__ G_own, float* __restrict__ G_canon) {
    constexpr int GPB = 256 / LPG;
    const int tl = threadIdx.x % LPG;
    const int K = KT > 0 ? KT : K_rt;
    for (int it = 0; it < RDG_RIGR_ITERS; ++it) {
        const long long s = ((long long)blockIdx.x * RDG_RIGR_ITERS + it) * GPB + threadIdx.x / LPG;
        if (s >= n) break;
        const long long i_row = orig ? orig[s] : s;
        const long long e0 = rev_off[s], e1 = rev_off[s + 1];
        float cx = 0.f, cy = 0.f, cz = 0.f;
        for (int tau = tl; tau < nt; tau += LPG) {
            const float* pp = P3 + (s * nt + tau) * 3;
            const float px = pp[0], py = pp[1], pz = pp[2];
            float* go = G_own + (i_row * nt + tau) * 3;
            float gx = go[0], gy = go[1], gz = go[2];
            for (long long e = e0; e < e1; e += 4) {
                long long edge[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) edge[u] = rev_edge[e + u < e1 ? e + u : e1 - 1];
                float q[4][3], ig[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long src = KT > 0 ? edge[u] / KT : edge[u] / K;
                    const float* qp = P3 + (src * nt + tau) * 3;
                    q[u][0] = qp[0]; q[u][1] = qp[1]; q[u][2] = qp[2];
                    ig[u] = e + u < e1 ? IG[edge[u] * nt + tau] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    gx += ig[u] * (px - q[u][0]); gy += ig[u] * (py - q[u][1]); gz += ig[u] * (pz - q[u][2]);
                }
            }
            go[0] = gx; go[1] = gy; go[2] = gz;
            cx += gx; cy += gy; cz += gz;
        }
        if (G_canon) {
#pragma unroll
            for (int o = LPG >> 1; o > 0; o >>= 1) {
                cx += __shfl_xor(cx, o); cy += __shfl_xor(cy, o); cz += __shfl_xor(cz, o);
            }
            if (tl == 0) { float* gc = G_canon + i_row * 3; gc[0] = cx; gc[1] = cy; gc[2] = cz; }
        }
    }
}

// P3[s][tau] = own[order[s]][tau] + canon[order[s]]: the sample's positions at the drawn times, rows re-laid along the
// curve order.  One thread per output float (whole rows move: coalesced on both sides, no idle lanes whatever nt is)
__global__ void __launch_bounds__(256)
rdg_rigidity_pack_rows_kernel(long long n, int nt, const float* __restrict__ own, const float* __restrict__ canon,
                              const long long* __restrict__ order, float* __restrict__ P3) {
    const int len = nt * 3;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * len) return;
    const long long s = rdg_div_pos(e, len, 1.0 / (double)len);
    const int j = (int)(e - s * len);
    const long long r = order ? order[s] : s;
    P3[e] = own[r * len + j] + canon[r * 3 + j % 3];
}

template <int LPG>
static int rdg_rigidity_rows_launch(int64_t n, int32_t K, int32_t nt, const float* P3, const int64_t* nn_idx, const float* d2,
                                    const int64_t* rev_off, const int64_t* rev_edge, const int64_t* orig, float eps,
                                    double* loss_sum, float* IG, float2* RA, float* d_d2, float* G_own, float* G_canon,
                                    hipStream_t st) {
    const long long per_wg = (256 / LPG) * RDG_RIGR_ITERS;
    const dim3 grid((unsigned)((n + per_wg - 1) / per_wg));
    if (K == 8) {
        hipLaunchKernelGGL((rdg_rigidity_rows_out_kernel<LPG, 8>), grid, dim3(256), 0, st, (long long)n, K, nt, P3,
                           (const long long*)nn_idx, d2, (const long long*)orig, eps * eps, loss_sum, IG, RA, G_own, d_d2);
        hipLaunchKernelGGL((rdg_rigidity_rows_in_kernel<LPG, 8>), grid, dim3(256), 0, st, (long long)n, K, nt, P3,
                           (const long long*)rev_off, (const long long*)rev_edge, (const long long*)orig, IG, G_own, G_canon);
    } else {
        hipLaunchKernelGGL((rdg_rigidity_rows_out_kernel<LPG, 0>), grid, dim3(256), 0, st, (long long)n, K, nt, P3,
                           (const long long*)nn_idx, d2, (const long long*)orig, eps * eps, loss_sum, IG, RA, G_own, d_d2);
        hipLaunchKernelGGL((rdg_rigidity_rows_in_kernel<LPG, 0>), grid, dim3(256), 0, st, (long long)n, K, nt, P3,
                           (const long long*)rev_off, (const long long*)rev_edge, (const long long*)orig, IG, G_own, G_canon);
    }
    return rdg_check_hip(hipGetLastError(), "rigidity_rows launch");
}

static inline int rdg_rig_lpg(int nt) { return nt <= 8 ? 8 : nt <= 16 ? 16 : nt <= 32 ? 32 : 64; }

extern "C" int rdg_rigidity_pack_rows(int64_t n, int32_t nt, const float* own, const float* canon, const int64_t* order,
                                      float* P3, void* stream) {
    if (n <= 0 || nt <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)n * nt * 3;
    hipLaunchKernelGGL(rdg_rigidity_pack_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (long long)n, nt,
                       own, canon, (const long long*)order, P3);
    return rdg_check_hip(hipGetLastError(), "rigidity_pack_rows launch");
}

extern "C" int rdg_rigidity_dp_rows(int64_t n, int32_t K, int32_t nt, const float* P3, const int64_t* nn_idx, const float* d2,
                                    const int64_t* rev_off, const int64_t* rev_edge, const int64_t* orig, const int64_t* rank,
                                    float eps, double* loss_sum, float* IG, float* RA, float* d_d2, float* G_own,
                                    float* G_canon, void* stream) {
    if (n <= 0 || K <= 0 || nt <= 0) return rdg_set_error("rigidity_dp_rows: bad sizes");
    if (!IG || !G_own || !d_d2 || !loss_sum) return rdg_set_error("rigidity_dp_rows: IG, G_own, d_d2 and loss_sum are required");
    if (orig && RA && !rank) return rdg_set_error("rigidity_dp_rows: RA with a stored order needs rank (the inverse of orig)");
    if (((uintptr_t)RA) & 7) return rdg_set_error("rigidity_dp_rows: RA must be 8-B aligned");
    hipStream_t st = (hipStream_t)stream;
    float2* ra = nt >= K ? (float2*)RA : nullptr;      // the two-row form of the d2 gradient needs nt >= K
    hipError_t e = rdg_zero_async(loss_sum, 8, st);
    if (e == hipSuccess && !ra) e = rdg_zero_async(d_d2, (size_t)n * K * 4, st);
    if (e != hipSuccess) return rdg_check_hip(e, "rigidity_dp_rows memset");
    int rc;
    switch (rdg_rig_lpg(nt)) {
        case 8:  rc = rdg_rigidity_rows_launch<8>(n, K, nt, P3, nn_idx, d2, rev_off, rev_edge, orig, eps, loss_sum, IG, ra, d_d2, G_own, G_canon, st); break;
        case 16: rc = rdg_rigidity_rows_launch<16>(n, K, nt, P3, nn_idx, d2, rev_off, rev_edge, orig, eps, loss_sum, IG, ra, d_d2, G_own, G_canon, st); break;
        case 32: rc = rdg_rigidity_rows_launch<32>(n, K, nt, P3, nn_idx, d2, rev_off, rev_edge, orig, eps, loss_sum, IG, ra, d_d2, G_own, G_canon, st); break;
        default: rc = rdg_rigidity_rows_launch<64>(n, K, nt, P3, nn_idx, d2, rev_off, rev_edge, orig, eps, loss_sum, IG, ra, d_d2, G_own, G_canon, st); break;
    }
    if (rc || !ra) return rc;
    const long long rows = (long long)n * K;
    hipLaunchKernelGGL(rdg_rigidity_d2grad_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, (long long)n, K, nt,
                       (const long long*)rank, (const float2*)ra, d_d2);
    return rdg_check_hip(hipGetLastError(), "rigidity_d2grad launch");
}

// ---------------------------------------------------------------------------------------------------------
// The other two scatters of a rigidity step, through the same stored-order graph (nbr [n][K] neighbour lists, rev_off / rev_edge
// reverse adjacency, X [n][3] the sample's positions, all along the curve order; orig[s] = row of stored element s in the
// caller's order): every gradient row is written once by the thread that owns it, its arriving edges read from rows that are
// close in memory.
//   rdg_graph_points_backward : knn_points' backward for p1 is p2 (rdg_knn_points_backward: 27 float atomics per query)
//   rdg_graph_surface_*       : RigidityLoss "surface" (/root/reference/src/trainer/losses.py:241-250): mean_i || x_i - mean_k x_nn(i,k) + 1e-6 ||
//                               (F.pairwise_distance adds its eps to the difference), forward + the unit vectors its backward needs
// ---------------------------------------------------------------------------------------------------------
template <int KT>
__global__ void __launch_bounds__(256)
rdg_graph_points_bwd_kernel(long long n, int K_rt, const float* __restrict__ X, const long long* __restrict__ nbr,
                            const long long* __restrict__ rev_off, const long long* __restrict__ rev_edge,
                            const long long* __restrict__ orig, const float* __restrict__ g, float* __restrict__ d_pts) {
    const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const int K = KT > 0 ? KT : K_rt;
    const long long i_row = orig[s];
    const float x = X[3 * s], y = X[3 * s + 1], z = X[3 * s + 2];
    float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
    for (int k = 0; k < (KT > 0 ? KT : K); ++k) {
        const long long t = nbr[s * K + k];
        const float w = 2.0f * g[i_row * K + k];
        ax += w * (x - X[3 * t]); ay += w * (y - X[3 * t + 1]); az += w * (z - X[3 * t + 2]);
    }
    const long long e1 = rev_off[s + 1];
    for (long long e = rev_off[s]; e < e1; ++e) {
        const long long edge = rev_edge[e];
        const long long src = KT > 0 ? edge / KT : edge / K;
        const float w = 2.0f * g[orig[src] * K + (edge - src * K)];
        ax -= w * (X[3 * src] - x); ay -= w * (X[3 * src + 1] - y); az -= w * (X[3 * src + 2] - z);
    }
    d_pts[3 * i_row] = ax; d_pts[3 * i_row + 1] = ay; d_pts[3 * i_row + 2] = az;
}

template <int KT>
__global__ void __launch_bounds__(256)
rdg_graph_surface_fwd_kernel(long long n, int K_rt, const float* __restrict__ X, const long long* __restrict__ nbr,
                             float* __restrict__ U, double* __restrict__ loss_sum) {
    const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
    const int K = KT > 0 ? KT : K_rt;
    double local = 0.0;
    if (s < n) {
        float cx = 0.f, cy = 0.f, cz = 0.f;
#pragma unroll
        for (int k = 0; k < (KT > 0 ? KT : K); ++k) {
            const long long t = nbr[s * K + k];
            cx += X[3 * t]; cy += X[3 * t + 1]; cz += X[3 * t + 2];
        }
        const float fk = (float)K;
        const float dx = X[3 * s] - cx / fk + 1e-6f, dy = X[3 * s + 1] - cy / fk + 1e-6f, dz = X[3 * s + 2] - cz / fk + 1e-6f;
        const float r = sqrtf(dx * dx + dy * dy + dz * dz);
        local = (double)r;
        const float ir = r > 0.f ? 1.0f / r : 0.f;
        U[3 * s] = dx * ir; U[3 * s + 1] = dy * ir; U[3 * s + 2] = dz * ir;
    }
    __shared__ double sh[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, (sh[0] + sh[1]) + (sh[2] + sh[3]));
}

// d(sum_i r_i) / d x_s = u_s - (1 / K) sum over the edges arriving at s of u_src
template <int KT>
__global__ void __launch_bounds__(256)
rdg_graph_surface_bwd_kernel(long long n, int K_rt, const float* __restrict__ U, const long long* __restrict__ rev_off,
                             const long long* __restrict__ rev_edge, const long long* __restrict__ orig,
                             float* __restrict__ d_pts) {
    const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const int K = KT > 0 ? KT : K_rt;
    float ax = 0.f, ay = 0.f, az = 0.f;
    const long long e1 = rev_off[s + 1];
    for (long long e = rev_off[s]; e < e1; ++e) {
        const long long edge = rev_edge[e];
        const long long src = KT > 0 ? edge / KT : edge / K;
        ax += U[3 * src]; ay += U[3 * src + 1]; az += U[3 * src + 2];
    }
    const float fk = (float)K;
    const long long i_row = orig[s];
    d_pts[3 * i_row] = U[3 * s] - ax / fk; d_pts[3 * i_row + 1] = U[3 * s + 1] - ay / fk; d_pts[3 * i_row + 2] = U[3 * s + 2] - az / fk;
}

extern "C" int rdg_graph_points_backward(int64_t n, int32_t K, const float* X, const int64_t* nbr, const int64_t* rev_off,
                                         const int64_t* rev_edge, const int64_t* orig, const float* g_dists, float* d_pts,
                                         void* stream) {
    if (n <= 0 || K <= 0) return 0;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (K == 8)
        hipLaunchKernelGGL(rdg_graph_points_bwd_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, (long long)n, K, X,
                           (const long long*)nbr, (const long long*)rev_off, (const long long*)rev_edge, (const long long*)orig,
                           g_dists, d_pts);
    else
        hipLaunchKernelGGL(rdg_graph_points_bwd_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, (long long)n, K, X,
                           (const long long*)nbr, (const long long*)rev_off, (const long long*)rev_edge, (const long long*)orig,
                           g_dists, d_pts);
    return rdg_check_hip(hipGetLastError(), "graph_points_backward launch");
}

extern "C" int rdg_graph_surface(int64_t n, int32_t K, const float* X, const int64_t* nbr, const int64_t* rev_off,
                                 const int64_t* rev_edge, const int64_t* orig, float* U, double* loss_sum, float* d_pts,
                                 void* stream) {
    if (n <= 0 || K <= 0) return rdg_set_error("graph_surface: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = rdg_zero_async(loss_sum, 8, st);
    if (e != hipSuccess) return rdg_check_hip(e, "graph_surface memset");
    const dim3 grid((unsigned)((n + 255) / 256));
    if (K == 8) {
        hipLaunchKernelGGL(rdg_graph_surface_fwd_kernel<8>, grid, dim3(256), 0, st, (long long)n, K, X, (const long long*)nbr, U,
                           loss_sum);
        hipLaunchKernelGGL(rdg_graph_surface_bwd_kernel<8>, grid, dim3(256), 0, st, (long long)n, K, U, (const long long*)rev_off,
                           (const long long*)rev_edge, (const long long*)orig, d_pts);
    } else {
        hipLaunchKernelGGL(rdg_graph_surface_fwd_kernel<0>, grid, dim3(256), 0, st, (long long)n, K, X, (const long long*)nbr, U,
                           loss_sum);
        hipLaunchKernelGGL(rdg_graph_surface_bwd_kernel<0>, grid, dim3(256), 0, st, (long long)n, K, U, (const long long*)rev_off,
                           (const long long*)rev_edge, (const long long*)orig, d_pts);
    }
    return rdg_check_hip(hipGetLastError(), "graph_surface launch");
}
