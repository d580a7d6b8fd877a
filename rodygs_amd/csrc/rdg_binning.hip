// rdg_binning.hip -- tile binning, two algorithms with the same result bit for bit:
//   * bucket binning (bin_mode 0, second half of the file): count / scan / scatter into per-tile buckets + per-tile sort;
//   * radix binning (bin_mode 1, "depth first", end of the file): the Gaussians are sorted by depth once (P keys), their
//     tile instances emitted in that order and then stably partitioned by tile with the radix sort of
//     rdg_radix_sort.hip -- two 8-bit passes over (tile id, Gaussian id) pairs instead of six over 64-bit (tile | depth)
//     keys, and no per-tile work at all, so its time does not depend on how the instances are spread over the tiles.
//
// Integer/byte work, HBM-bound.  Nothing here needs the host to know D (= num_rendered): every kernel is
// launched with a D-independent grid and reads D from device memory, so the forward pass has no D2H stall.
#include "rdg_common.h"
#include <stdlib.h>

// ---------------------------------------------------------------------------------------------------------
// duplicateWithKeys: load-balanced expansion.  A block owns 256 consecutive Gaussians; its output slots
// [block_base, block_base + block_total) are dealt round-robin to the 256 threads, each of which finds its
// Gaussian by binary search in the block's LDS offsets -> perfectly coalesced 8-B key / 4-B value stores
// regardless of how skewed tiles_touched is.
// ---------------------------------------------------------------------------------------------------------
// every kernel below takes a Gaussian's tile rectangle, depth bits and tile count from the per-Gaussian `rectd` words the
// per-Gaussian stage wrote (rdg_splat_rect, rdg_preprocess_fwd.hip): one coalesced 16-B load, and ONE definition of the
// rectangle for the reference rule and the tight one (RdgRasterSettings.cull)
__global__ void __launch_bounds__(RDG_PRE_BLOCK)
rdg_duplicate_kernel(int P, int gx, int gy, const uint4* __restrict__ rectd,
                     const uint32_t* __restrict__ block_sums, uint64_t* __restrict__ keys,
                     uint32_t* __restrict__ vals, long long capacity, const int32_t* __restrict__ num_rendered) {
    if ((long long)(*num_rendered) > capacity) return;
    __shared__ uint32_t sOff[RDG_PRE_BLOCK];
    __shared__ uint32_t sDepth[RDG_PRE_BLOCK];
    __shared__ uint16_t sX0[RDG_PRE_BLOCK], sY0[RDG_PRE_BLOCK], sW[RDG_PRE_BLOCK];
    __shared__ uint32_t wsum[RDG_PRE_BLOCK / RDG_WAVE];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * RDG_PRE_BLOCK + tid;
    uint32_t t = 0;
    if (i < P) {
        const uint4 rd = rectd[i];
        t = rd.w;
        if (t > 0) {
            sX0[tid] = (uint16_t)(rd.x & 0xffffu); sY0[tid] = (uint16_t)(rd.x >> 16);
            sW[tid] = (uint16_t)((rd.y & 0xffffu) - (rd.x & 0xffffu));
            sDepth[tid] = rd.z;
        }
    }
    const uint32_t inc = rdg_wave_scan_incl(t);
    const uint32_t lane = tid & 63, w = tid >> 6;
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t k = 0; k < w; ++k) woff += wsum[k];
    sOff[tid] = woff + inc - t;
    __syncthreads();
    const uint32_t base = block_sums[blockIdx.x];
    const uint32_t total = block_sums[blockIdx.x + 1] - base;
    for (uint32_t k = tid; k < total; k += RDG_PRE_BLOCK) {
        // last g with sOff[g] <= k
        int lo = 0, hi = RDG_PRE_BLOCK - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (sOff[mid] <= k) lo = mid; else hi = mid - 1;
        }
        const uint32_t j = k - sOff[lo];
        const uint32_t wd = sW[lo];
        const uint32_t ry = j / wd, rx = j - ry * wd;
        const uint64_t tile = (uint64_t)(sY0[lo] + ry) * (uint64_t)gx + (uint64_t)(sX0[lo] + rx);
        keys[base + k] = (tile << 32) | (uint64_t)sDepth[lo];
        vals[base + k] = (uint32_t)(blockIdx.x * RDG_PRE_BLOCK + lo);
    }
}

// ---------------------------------------------------------------------------------------------------------
// radix sort: the stable LSD radix sort of rdg_radix_sort.hip (one kernel per 8-bit pass + one histogram kernel)
// ---------------------------------------------------------------------------------------------------------
int rdg_launch_sort(uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, int64_t capacity,
                    const int32_t* n_dev, int end_bit, void* sort_tmp, int* result_in_b, hipStream_t s) {
    return rdg_launch_radix_sort<uint64_t>(keys_a, keys_b, vals_a, vals_b, capacity, n_dev, 0, end_bit, sort_tmp, result_in_b, s);
}

// ---------------------------------------------------------------------------------------------------------
// identifyTileRanges
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
rdg_tile_ranges_kernel(const uint64_t* __restrict__ keys, long long capacity, const int32_t* __restrict__ n_dev,
                       uint2* __restrict__ ranges) {
    const long long n = *n_dev;
    if (n > capacity) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const uint32_t tile = (uint32_t)(keys[i] >> 32);
        if (i == 0) {
            ranges[tile].x = 0;
        } else {
            const uint32_t prev = (uint32_t)(keys[i - 1] >> 32);
            if (prev != tile) {
                ranges[prev].y = (uint32_t)i;
                ranges[tile].x = (uint32_t)i;
            }
        }
        if (i == n - 1) ranges[tile].y = (uint32_t)n;
    }
}

__global__ void rdg_copy_pairs_kernel(const uint64_t* __restrict__ k, const uint32_t* __restrict__ v,
                                      uint64_t* __restrict__ ko, uint32_t* __restrict__ vo, long long capacity,
                                      const int32_t* __restrict__ n_dev) {
    const long long n = *n_dev;
    if (n > capacity) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        if (ko) ko[i] = k[i];
        if (vo) vo[i] = v[i];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Bucket binning (default): the 45-47-bit LSD radix sort moves every (key, value) pair 6 times through HBM with
// a random scatter each time.  The key is (tile | depth), so the same order is reached with far less traffic:
//   1. count   : per-tile instance counts (load-balanced expansion + integer atomics); the value each atomic
//                returns is the instance's rank inside its tile and is kept
//   2. scan    : exclusive scan over tiles  ->  the tile ranges themselves (no identifyTileRanges pass)
//   3. scatter : every instance goes to slot (tile start + rank) as the 64-bit composite
//                (depth_bits << 32 | gaussian_id), no atomics; the order inside a tile is arbitrary ...
//   4. sort    : ... and is fixed by a per-tile bitonic sort of the composites in LDS.  Composites are unique,
//                so the result is deterministic and identical to the stable sort on (tile | depth): equal depths
//                come out in increasing Gaussian index = emission order.  Bit-exact against the oracle.
// Tiles with more than RDG_TSORT_LDS instances go through a multi-workgroup chunk sort + merge tree (below).
// ---------------------------------------------------------------------------------------------------------
// position of tile (x, y) on the Z curve of the counter array (rdg_cnt_entries)
__device__ __forceinline__ uint32_t rdg_zidx(uint32_t x, uint32_t y) {
    x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu; x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    y = (y | (y << 8)) & 0x00FF00FFu; y = (y | (y << 4)) & 0x0F0F0F0Fu; y = (y | (y << 2)) & 0x33333333u;
    y = (y | (y << 1)) & 0x55555555u;
    return x | (y << 1);
}

// Count pass, wave-aggregated: the 64 instance slots a wave handles per step are grouped by tile with a ballot
// "match" over the bits of the counter index (the same idiom as the radix scatter); the lowest lane of every group
// adds the group's size with ONE returning atomic and the others take (base + their position in the group).  The
// rank an instance keeps is therefore exactly what a per-instance atomic would have handed out, with as many atomics
// as there are DISTINCT tiles among the wave's 64 slots: spatially coherent clouds (Morton-ordered scenes, densified
// clusters, one tile holding 200 k instances) issue a fraction of the atomics and no longer serialise on one address.
#ifndef RDG_BUCKET_PIPE
#define RDG_BUCKET_PIPE 1
#endif
template <int MODE>  // 0 = count, 1 = scatter
__global__ void __launch_bounds__(RDG_PRE_BLOCK)
rdg_tile_bucket_kernel(int P, int gx, int gy, int zbits, const uint4* __restrict__ rectd,
#ifdef RDG_ABL_EXACT_CULL   // measurement build (scripts/build_variant.sh): the exact ellipse-vs-tile test per INSTANCE in the count pass
                       const RdgRec* __restrict__ rec_abl,
#endif
                       const uint32_t* __restrict__ block_sums, uint32_t* __restrict__ tile_cnt,
                       const uint2* __restrict__ ranges, uint32_t* __restrict__ rank_buf,
                       uint64_t* __restrict__ comp, long long capacity, const int32_t* __restrict__ num_rendered,
                       uint4* __restrict__ zero16, long long n_zero16) {
    // scatter pass: also clears the compositing stage's visit record (one launch and one stream boundary less than a
    // memset of its own); must happen on the overflow path too -- the forward then renders an empty scene
    if (MODE == 1)
        for (long long i = (long long)blockIdx.x * RDG_PRE_BLOCK + threadIdx.x; i < n_zero16;
             i += (long long)gridDim.x * RDG_PRE_BLOCK)
            zero16[i] = make_uint4(0u, 0u, 0u, 0u);
    if ((long long)(*num_rendered) > capacity) return;
    // Load-balanced expansion, one WAVE at a time: the wave's 64 Gaussians own T_w consecutive instance slots (their
    // rectangles' tiles, row by row), handled 64 slots per step.  The owner of a slot is found without a search: every
    // Gaussian drops its lane number at the slot its run starts on (one LDS store per Gaussian and frame), a running
    // maximum along the 64 slots fills the runs in (DPP scan), and the owner's rectangle comes over by ds_bpermute.
    // (The earlier form was a binary search over the workgroup's 256 offsets, eight dependent LDS reads per slot; this one
    // is shorter and 2-3 us faster in the count pass on the bench frame -- what the passes are bound by: see below.)
    // Marks of earlier steps need no clearing: they name lanes <= the owner carried over from the previous step, which
    // the maximum ignores.
    __shared__ uint32_t sMark[RDG_PRE_BLOCK];
    __shared__ uint32_t wsum[RDG_PRE_BLOCK / RDG_WAVE];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * RDG_PRE_BLOCK + tid;
    uint32_t t = 0, xy0 = 0, wd0 = 1, dep = 0;
    if (i < P) {
        const uint4 rd = rectd[i];
        t = rd.w;
        if (t > 0) { xy0 = rd.x; wd0 = (rd.y & 0xffffu) - (rd.x & 0xffffu); dep = rd.z; }
    }
#ifdef RDG_ABL_EXACT_CULL
    // per-Gaussian constants of the quadratic form the compositing kernels evaluate (rdg_quadrant_bits): Q <= r2 somewhere
    // in the tile's rectangle of pixel centres, or the instance gets no slot in its tile's list
    float e_px = 0.f, e_py = 0.f, e_a = 1.f, e_b2 = 0.f, e_c = 1.f, e_r2 = -1.f;
    if (MODE == 0 && i < P && t > 0) {
        const float4 q0 = rec_abl[i].q0, q1 = rec_abl[i].q1;
        const float icyy = rec_abl[i].q2.w;
        e_px = q0.x; e_py = q0.y; e_a = q0.z; e_b2 = 2.0f * q0.w; e_c = (q0.w * q0.w) / q0.z + icyy;
        e_r2 = 2.0f * (__logf(255.0f * q1.y) + 0.02f) * 1.001f;
    }
#endif
    const uint32_t inc = rdg_wave_scan_incl(t);
    const uint32_t lane = tid & 63, w = tid >> 6;
    const uint32_t excl = inc - t;
    sMark[tid] = 0u;
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t k = 0; k < w; ++k) woff += wsum[k];
    const uint32_t total = wsum[w];                                // wave-uniform trip count: the ballots need every lane
    const uint32_t first = block_sums[blockIdx.x] + woff;          // the wave's first slot in the frame's instance order
    uint32_t* __restrict__ mark = sMark + (w << 6);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t carry = 0;
    // U steps can be taken together (their returning atomics / loads all in flight before the first result is needed).
    // Measured on the bench frame: U = 2 and 4 change nothing (count + scan + scatter 87-88 us in every form; the ISA
    // at U = 4 does hold four atomics in flight) -- with 8 workgroups per CU the round trips of different waves
    // already overlap.  The atomics are not the bound either: PMC (scripts/atomic_probe.sh) counts 0.33 M atomic
    // requests per frame for 10 M instances (the wave aggregation on a Z-ordered cloud).  Nor is the tile matching:
    // matching on the index bits that differ inside the wave only was bit-exact and not a microsecond faster.  A wave
    // of the count pass lives 12.5 us for ~1 000 vector instructions (SQ counters): the pass is each workgroup's chain
    // of dependent memory trips (tile count -> record -> scan -> per step: atomic -> store) at ~100 MB / 36 us; the
    // scatter pass moves ~170 MB, 80 MB of it as scattered 8-B stores, in 33 us (5 TB/s).  U stays 1.  
    constexpr int U = RDG_BUCKET_PIPE;
    for (uint32_t k0 = 0; k0 < total; k0 += RDG_WAVE * U) {
        bool act[U];
#ifdef RDG_ABL_EXACT_CULL
        bool dead_slot[U];
#pragma unroll
        for (int u = 0; u < U; ++u) dead_slot[u] = false;
#endif
        uint32_t kk[U], tx[U], ty[U], own_lane[U], odep[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t kb = k0 + (uint32_t)(RDG_WAVE * u);
            kk[u] = kb + lane;
            act[u] = kk[u] < total;
            if (t > 0 && excl - kb < (uint32_t)RDG_WAVE) mark[excl - kb] = lane + 1u;  // (excl < kb wraps: no store)
            rdg_wave_lds_sync();
            const uint32_t own = max(rdg_wave_scan_max_incl(mark[lane]), carry);       // 1 + lane of the slot's Gaussian
            rdg_wave_lds_sync();
            carry = (uint32_t)__builtin_amdgcn_readlane((int)own, 63);
            own_lane[u] = (own - 1u) & 63u;
            const int src = (int)own_lane[u] << 2;
            const uint32_t oxy = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)xy0);
            const uint32_t wd = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)wd0);
            const uint32_t oex = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)excl);
            odep[u] = MODE == 1 ? (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)dep) : 0u;
            const uint32_t j = act[u] ? kk[u] - oex : 0u;
            const uint32_t ry = j / wd, rx = j - ry * wd;
            tx[u] = (oxy & 0xffffu) + rx; ty[u] = (oxy >> 16) + ry;
#ifdef RDG_ABL_EXACT_CULL
            if (MODE == 0) {
                const float opx = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, e_px)));
                const float opy = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, e_py)));
                const float oa = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, e_a)));
                const float ob2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, e_b2)));
                const float oc = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, e_c)));
                const float or2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, e_r2)));
                const float u1 = opx - (float)(tx[u] * RDG_TILE), u0 = u1 - 15.0f;
                const float v1 = opy - (float)(ty[u] * RDG_TILE), v0 = v1 - 15.0f;
                const bool inside = u0 <= 0.0f && u1 >= 0.0f && v0 <= 0.0f && v1 >= 0.0f;
                auto edge = [](float a_, float b2_, float c_, float ue, float w0, float w1) {
                    const float tt = b2_ * ue;
                    const float vs = __builtin_amdgcn_fmed3f(-0.5f * tt / c_, w0, w1);
                    return fmaf(fmaf(c_, vs, tt), vs, a_ * ue * ue);
                };
                const float m = fminf(fminf(edge(oa, ob2, oc, u0, v0, v1), edge(oa, ob2, oc, u1, v0, v1)),
                                      fminf(edge(oc, ob2, oa, v0, u0, u1), edge(oc, ob2, oa, v1, u0, u1)));
                dead_slot[u] = act[u] && !(inside || !(m > or2));
            }
#endif
        }
        if (MODE == 0) {
            uint32_t below[U], base[U];
            int leader[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t z = rdg_zidx(tx[u], ty[u]);
#ifdef RDG_ABL_EXACT_CULL
                const bool slot_in = act[u];
                if (dead_slot[u]) act[u] = false;            // no slot in the tile's list ...
                if (slot_in && dead_slot[u]) rank_buf[first + kk[u]] = 0xffffffffu;   // ... and the scatter pass skips it
#endif
                unsigned long long m = __ballot(act[u]);
                for (int bit = 0; bit < zbits; ++bit) {
                    const bool bset = (z >> bit) & 1u;
                    const unsigned long long bal = __ballot(act[u] && bset);
                    m &= bset ? bal : ~bal;
                }
                below[u] = (uint32_t)__popcll(m & lt_mask);
                base[u] = 0;
                if (act[u] && below[u] == 0) base[u] = atomicAdd(&tile_cnt[z], (uint32_t)__popcll(m));
                leader[u] = act[u] ? __ffsll((long long)m) - 1 : 0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t bs = (uint32_t)__shfl((int)base[u], leader[u]);
                // coalesced 4-B store in emission order; the scatter pass needs no second round of atomics
                if (act[u]) rank_buf[first + kk[u]] = bs + below[u];
            }
        } else {
            uint32_t st[U], rk[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                st[u] = 0u; rk[u] = 0u;
                if (act[u]) { st[u] = ranges[ty[u] * (uint32_t)gx + tx[u]].x; rk[u] = rank_buf[first + kk[u]]; }
#ifdef RDG_ABL_EXACT_CULL
                if (rk[u] == 0xffffffffu) act[u] = false;
#endif
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (act[u])
                    comp[st[u] + rk[u]] = ((uint64_t)odep[u] << 32) |
                                          (uint64_t)(blockIdx.x * RDG_PRE_BLOCK + (w << 6) + own_lane[u]);
        }
    }
}

// exclusive scan of tile_cnt -> ranges (untouched tiles stay (0,0), as identifyTileRanges leaves them); clears cursors.
// Tiles with more than RDG_TSORT_LDS instances are entered in the heavy work list (one work item per chunk of
// RDG_TSORT_LDS instances) for the multi-workgroup sort; the arrival counters of their merge trees are zeroed here.
__global__ void __launch_bounds__(1024)
rdg_tile_scan_kernel(int n_tiles, int gx, const uint32_t* __restrict__ tile_cnt, uint2* __restrict__ ranges,
                     uint32_t* __restrict__ tile_fill, long long capacity, const int32_t* __restrict__ num_rendered,
                     uint32_t* __restrict__ hv_header, RdgHeavyDesc* __restrict__ hv_desc, uint2* __restrict__ hv_work,
                     uint32_t* __restrict__ hv_nodes, uint32_t max_heavy, uint32_t max_chunks, uint32_t max_work,
                     int32_t* __restrict__ max_tile_out, uint2* __restrict__ hv_chunk_work, uint32_t max_chunk_items,
                     int32_t* __restrict__ host_mirror) {
    if ((long long)(*num_rendered) > capacity) {
        if (threadIdx.x == 0 && max_tile_out) *max_tile_out = 0;
        if (threadIdx.x == 0 && host_mirror) { host_mirror[1] = 0; __threadfence_system(); host_mirror[0] = *num_rendered; }
        // capacity overflow: leave EVERY tile empty, so the compositing kernels (forward and backward) see a valid,
        // empty scene (background image, zero gradients) instead of stale ranges; the host detects D > capacity
        for (int i = threadIdx.x; i < n_tiles; i += 1024) { ranges[i] = make_uint2(0u, 0u); tile_fill[i] = 0u; }
        if (threadIdx.x == 0) { hv_header[0] = 0u; hv_header[1] = 0u; hv_header[2] = 0u; }
        return;
    }
    // Every WAVE owns a segment of consecutive tiles and walks it 64 tiles at a time, lane l on tile (step base + l):
    // the stores of a step are then contiguous (512 B of ranges, 256 B of cursors per instruction) and its counter
    // loads touch 16 lines of the Z-ordered array instead of 64.  (With a run of 8 consecutive tiles per THREAD every
    // load and store instruction of this single workgroup touched 64 different cache lines -- ~32 k line transactions
    // through one CU's address path, most of the kernel's 14 us on the critical path of every frame.)
    // One pass for the segment totals, ONE barrier, one pass that scans (DPP) and writes.  The work-list and chunk-item
    // slots come out of the same scans (they were LDS atomics on two addresses); only the rare > RDG_TSORT_LDS tiles
    // still use atomics.
    __shared__ uint32_t wtot[3][16];
    __shared__ uint32_t sHeavy, sChunks, sMaxTile;
    if (threadIdx.x == 0) { sHeavy = 0u; sChunks = 0u; sMaxTile = 0u; }
    for (uint32_t i = threadIdx.x; i < 2u * max_chunks; i += 1024) hv_nodes[i] = 0u;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int seg = (((n_tiles + 15) / 16) + 63) & ~63;
    const int s0 = min(n_tiles, (int)w * seg), s1 = min(n_tiles, s0 + seg);
    // (x, y) of the lane's first tile by ONE division; 64 tiles further by stepping along the rows
    const uint32_t x_first = (uint32_t)((s0 + (int)lane) % gx), y_first = (uint32_t)((s0 + (int)lane) / gx);
    uint32_t mine = 0, big = 0, my_work = 0, my_items = 0;
    {
        uint32_t x = x_first, y = y_first;
        for (int c0 = s0; c0 < s1; c0 += 64 * 8) {           // 8 independent loads in flight per lane
            uint32_t vv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                vv[q] = c0 + 64 * q + (int)lane < s1 ? tile_cnt[rdg_zidx(x, y)] : 0u;
                x += 64u;
                while (x >= (uint32_t)gx) { x -= (uint32_t)gx; ++y; }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t v = vv[q];
                mine += v; big = max(big, v);
                if (v > RDG_TSORT_LDS) my_work += (v + RDG_TSORT_LDS - 1) / RDG_TSORT_LDS;
                else if (v > RDG_TSORT_MID) my_work += 1u;
                else if (v > RDG_TSORT_SMALL) { my_work += 1u; my_items += (v + RDG_TSORT_SMALL - 1) / RDG_TSORT_SMALL; }
            }
        }
    }
    {
        const uint32_t inc = rdg_wave_scan_incl(mine);
        const uint32_t inc_w = rdg_wave_scan_incl(my_work), inc_i = rdg_wave_scan_incl(my_items);
        if (lane == 63) { wtot[0][w] = inc; wtot[1][w] = inc_w; wtot[2][w] = inc_i; }
    }
    __syncthreads();
    if (max_tile_out) {
        // wave maximum first (1024 threads on one LDS atomic serialise)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) big = max(big, (uint32_t)__shfl_xor((int)big, o));
        if (lane == 0) atomicMax(&sMaxTile, big);
    }
    uint32_t run0 = 0, work0 = 0, item0 = 0;                  // the segment's first slots (wave-uniform)
    for (uint32_t k = 0; k < w; ++k) { run0 += wtot[0][k]; work0 += wtot[1][k]; item0 += wtot[2][k]; }
    uint32_t x2 = x_first, y2 = y_first;
    for (int c0 = s0; c0 < s1; c0 += 64 * 8) {
      uint32_t vv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
          vv[q] = c0 + 64 * q + (int)lane < s1 ? tile_cnt[rdg_zidx(x2, y2)] : 0u;
          x2 += 64u;
          while (x2 >= (uint32_t)gx) { x2 -= (uint32_t)gx; ++y2; }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (c0 + 64 * q >= s1) break;                         // wave-uniform
        const int i = c0 + 64 * q + (int)lane;
        const uint32_t v = vv[q];
        uint32_t wk = 0, it = 0;
        if (v > RDG_TSORT_LDS) wk = (v + RDG_TSORT_LDS - 1) / RDG_TSORT_LDS;
        else if (v > RDG_TSORT_MID) wk = 1u;
        else if (v > RDG_TSORT_SMALL) { wk = 1u; it = (v + RDG_TSORT_SMALL - 1) / RDG_TSORT_SMALL; }
        const uint32_t sc_v = rdg_wave_scan_incl(v), sc_w = rdg_wave_scan_incl(wk), sc_i = rdg_wave_scan_incl(it);
        const uint32_t run = run0 + sc_v - v, wb = work0 + sc_w - wk, cb_items = item0 + sc_i - it;
        run0 += (uint32_t)__builtin_amdgcn_readlane((int)sc_v, 63);
        work0 += (uint32_t)__builtin_amdgcn_readlane((int)sc_w, 63);
        item0 += (uint32_t)__builtin_amdgcn_readlane((int)sc_i, 63);
        if (i >= s1) continue;
        ranges[i] = v ? make_uint2(run, run + v) : make_uint2(0u, 0u);
        tile_fill[i] = 0u;
        if (v > RDG_TSORT_LDS) {
            const uint32_t nch = wk;
            const uint32_t h = atomicAdd(&sHeavy, 1u);
            const uint32_t cb = atomicAdd(&sChunks, nch);
            // sized by construction (rdg_heavy_layout): the three bounds always hold
            if (h < max_heavy && cb + nch <= max_chunks && wb + nch <= max_work) {
                RdgHeavyDesc d; d.start = run; d.n = v; d.nchunks = nch; d.node_base = 2u * cb; d.tile = (uint32_t)i;
                d.pad0 = d.pad1 = d.pad2 = 0u;
                hv_desc[h] = d;
                for (uint32_t c = 0; c < nch; ++c) hv_work[wb + c] = make_uint2(h, c);
            }
        } else if (v > RDG_TSORT_MID) {
            if (wb < max_work) hv_work[wb] = make_uint2(0x80000000u | (uint32_t)i, 0xffffffffu);
        } else if (v > RDG_TSORT_SMALL) {
            // chunk items for the waves of the register-block sort, ONE merge item for the workgroup kernel after it
            const uint32_t nch = it;
            if (cb_items + nch <= max_chunk_items && wb < max_work) {
                for (uint32_t c = 0; c < nch; ++c) hv_chunk_work[cb_items + c] = make_uint2((uint32_t)i, c);
                hv_work[wb] = make_uint2(0x80000000u | (uint32_t)i, 0u);
            }
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t n_work = 0, n_items = 0;
        for (int k = 0; k < 16; ++k) { n_work += wtot[1][k]; n_items += wtot[2][k]; }
        hv_header[0] = min(n_work, max_work); hv_header[1] = min(sHeavy, max_heavy);
        hv_header[2] = min(n_items, max_chunk_items);
        if (max_tile_out) *max_tile_out = (int32_t)sMaxTile;
        // host mirror (RdgRasterSettings.num_rendered_host): [1] first, [0] last -- the host polls [0]
        if (host_mirror) { host_mirror[1] = (int32_t)sMaxTile; __threadfence_system(); host_mirror[0] = *num_rendered; }
    }
}

// ascending compare-exchange network without directions over N2 = 2^m slots (slots >= n must hold +inf = ~0ull, which
// no real composite equals: depth bits of a positive float are < 0x7f800000), 256 threads,
// comparator idx = tid + 256 m.  Most stages never leave a wave: with partner distance j <= 64 (and the mirrored first
// step of a merge of size k <= 128) the comparators idx in [64 w + 256 m, 64 w + 256 m + 64) read and write only
// elements [2 (64 w + 256 m), + 128) -- the same 128 for every such stage.  Between two wave-local stages the LDS
// traffic of a wave is already ordered (one wave's LDS operations execute in program order), so the workgroup barrier is
// needed only around the stages that do cross waves: 5 of the 45 stages of a 512-element sort, 21 of 91 at 8192.
// (A version with the elements in registers and DPP / ds_bpermute exchanges was SLOWER, 127 vs 71 us on the bench
// frame: every lane then evaluates its own side of each comparator, twice the compares and selects.)
// Call with the data in place and visible (barrier after the load); ends with a barrier.
// The whole network is generated at compile time for each power-of-two size (templates on the merge size K and the
// partner distance J): no loop counters, no stage dispatch, index arithmetic folded to two instructions, and the two
// elements of a half-cleaner comparator (J*8 bytes apart) move with ONE ds_read2_b64 / ds_write2_b64.  With run-time
// loops the kernel spent 70 % of its instructions in the scalar unit (1 087 of 1 560 per wave and tile, PMC), and the
// scalar unit is shared by the four SIMDs of a CU: that, not LDS or VALU, was what the sort was waiting for.
// THREADS = 256: a workgroup sorts one list, comparator idx = tid + 256 m (tid = thread in the workgroup).
// What bounds this sort is the LDS WRITE path (about 80 B/clk per CU for 8-byte stores, MI355X_MICROARCH.md §LDS): 45-55
// stages x 8 B written per element and stage = 2.6 GB of LDS stores per 1080p frame = ~55 us chip-wide, whatever the
// instruction count -- run-time loops, the compile-time network below, barriers on every stage or only on the
// cross-wave ones, one WAVE per tile without any barrier (THREADS = 64: 86 us, fewer waves to cover the latency) and
// elements held in registers with DPP / ds_bpermute exchanges (127 us: every lane evaluates its own side of each
// comparator) all landed at 70 us or worse.  Next step, not taken: 16 consecutive elements per lane so that the stages
// with partner distance <= 8 run in registers between ONE LDS read and write (28 instead of 55 LDS round trips).
template <bool PREV_LOCAL, bool LOCAL>
__device__ __forceinline__ void rdg_stage_sync() {
    if constexpr (PREV_LOCAL && LOCAL) rdg_wave_lds_sync(); else __syncthreads();
}

template <int J, int N2, int THREADS, typename ARR>
__device__ __forceinline__ void rdg_half_stage(ARR a, uint32_t tid) {
    for (uint32_t idx = tid; idx < (uint32_t)(N2 >> 1); idx += THREADS) {
        const uint32_t i = idx + (idx & ~(uint32_t)(J - 1)), p = i + J;
        const uint64_t x = a[i], y = a[p];
        const bool sw = x > y;            // branch-free: both slots are always rewritten (exec-mask juggling per
        a[i] = sw ? y : x;                // comparator costs more scalar instructions than the stores it saves)
        a[p] = sw ? x : y;
    }
}

template <int K, int N2, int THREADS, typename ARR>
__device__ __forceinline__ void rdg_flip_stage(ARR a, uint32_t tid) {
    constexpr uint32_t hk = K >> 1;
    for (uint32_t idx = tid; idx < (uint32_t)(N2 >> 1); idx += THREADS) {
        const uint32_t off = idx & (hk - 1), base = (idx - off) << 1;
        const uint32_t i = base + off, p = base + K - 1 - off;
        const uint64_t x = a[i], y = a[p];
        const bool sw = x > y;
        a[i] = sw ? y : x;
        a[p] = sw ? x : y;
    }
}

// half-cleaners J, J/2, ..., 1; PREV_LOCAL = the stage before was wave-local
template <int J, int N2, int THREADS, bool PREV_LOCAL, typename ARR>
__device__ __forceinline__ void rdg_half_stages(ARR a, uint32_t tid) {
    if constexpr (J >= 1) {
        constexpr bool local = THREADS == 64 || J <= 64;
        rdg_stage_sync<PREV_LOCAL, local>();
        rdg_half_stage<J, N2, THREADS>(a, tid);
        rdg_half_stages<J / 2, N2, THREADS, local>(a, tid);
    }
}

// merges of size K, 2K, ..., N2 (each: mirrored first step, then half-cleaners K/4 ... 1)
template <int K, int N2, int THREADS, typename ARR>
__device__ __forceinline__ void rdg_merge_stages(ARR a, uint32_t tid) {
    if constexpr (K <= N2) {
        constexpr bool local = THREADS == 64 || (K >> 1) <= 64;
        if constexpr (K > 2) rdg_stage_sync<true, local>();     // the stage before a merge is the half-cleaner J = 1
        rdg_flip_stage<K, N2, THREADS>(a, tid);
        rdg_half_stages<K / 4, N2, THREADS, local>(a, tid);
        rdg_merge_stages<2 * K, N2, THREADS>(a, tid);
    }
}

// workgroup version (256 threads): data in place and visible (barrier after the load); ends with a barrier
template <typename ARR>
__device__ __forceinline__ void rdg_bitonic_sort(ARR a, uint32_t n, uint32_t N2, uint32_t tid, uint32_t nthreads) {
    (void)nthreads; (void)n;
    switch (N2) {
        case 2: rdg_merge_stages<2, 2, 256>(a, tid); break;
        case 4: rdg_merge_stages<2, 4, 256>(a, tid); break;
        case 8: rdg_merge_stages<2, 8, 256>(a, tid); break;
        case 16: rdg_merge_stages<2, 16, 256>(a, tid); break;
        case 32: rdg_merge_stages<2, 32, 256>(a, tid); break;
        case 64: rdg_merge_stages<2, 64, 256>(a, tid); break;
        case 128: rdg_merge_stages<2, 128, 256>(a, tid); break;
        case 256: rdg_merge_stages<2, 256, 256>(a, tid); break;
        case 512: rdg_merge_stages<2, 512, 256>(a, tid); break;
        case 1024: rdg_merge_stages<2, 1024, 256>(a, tid); break;
        case 2048: rdg_merge_stages<2, 2048, 256>(a, tid); break;
        case 4096: rdg_merge_stages<2, 4096, 256>(a, tid); break;
        default: rdg_merge_stages<2, 8192, 256>(a, tid); break;
    }
    __syncthreads();
}

// ---- lists of up to 1024 instances: one WAVE per tile, E = N2 / 64 consecutive elements per LANE ------------------
// The LDS network above moves every element through LDS once per stage (45-55 stages) and is bound by the LDS store
// path.  Here a lane keeps E consecutive elements of the list in registers: every comparator whose partner distance is
// below E is a register-to-register compare-exchange, and only the stages that pair different lanes go through LDS --
// the wave writes its elements transposed (element e of lane L at [e][L]: conflict-free), and every lane reads the E
// elements of its partner lane (L ^ mask; mirrored element order for the first step of a merge).  21 LDS round trips
// instead of 45 (512 slots) or 55 (1024).  One wave owns the whole list, so there is no workgroup barrier at all.
// Same ascending network (mirrored first step + half-cleaners), slots >= n hold +inf.
// In-register compare-exchange, min to x.  The composites are (depth bits << 32 | id) with depth a positive finite
// float (bits < 0x7f800000), and the padding value is the largest finite double: read as IEEE doubles they are all
// positive, finite, normal numbers, whose order is the order of their bit patterns -- so v_min_f64 / v_max_f64 (which
// return one operand bit for bit) do in 2 instructions what compare + 4 selects do in 5.  The sort is VALU-bound.
#define RDG_SORT_PAD 0x7FEFFFFFFFFFFFFFull
__device__ __forceinline__ void rdg_cx(uint64_t& x, uint64_t& y) {
    double a = __builtin_bit_cast(double, x), b = __builtin_bit_cast(double, y), lo, hi;
    asm("v_min_f64 %0, %2, %3\n\tv_max_f64 %1, %2, %3" : "=&v"(lo), "=&v"(hi) : "v"(a), "v"(b));
    x = __builtin_bit_cast(uint64_t, lo);
    y = __builtin_bit_cast(uint64_t, hi);
}

// merge of size K (K <= E) inside every lane: mirrored first step, then half-cleaners K/4 ... 1
template <int K, int E>
__device__ __forceinline__ void rdg_lane_merge_reg(uint64_t (&v)[E]) {
#pragma unroll
    for (int b = 0; b < E; b += K)
#pragma unroll
        for (int o = 0; o < K / 2; ++o) rdg_cx(v[b + o], v[b + K - 1 - o]);
#pragma unroll
    for (int J = K / 4; J >= 1; J >>= 1)
#pragma unroll
        for (int idx = 0; idx < E / 2; ++idx) {
            const int i = idx + (idx & ~(J - 1));
            rdg_cx(v[i], v[i + J]);
        }
}

// half-cleaners J = JTOP ... 1 inside every lane (JTOP < E)
template <int JTOP, int E>
__device__ __forceinline__ void rdg_lane_halves_reg(uint64_t (&v)[E]) {
#pragma unroll
    for (int J = JTOP; J >= 1; J >>= 1)
#pragma unroll
        for (int idx = 0; idx < E / 2; ++idx) {
            const int i = idx + (idx & ~(J - 1));
            rdg_cx(v[i], v[i + J]);
        }
}

// one stage that pairs lane L with lane L ^ M through LDS; MIRROR: partner element E - 1 - e (first step of a merge)
template <int M, bool MIRROR, int E>
__device__ __forceinline__ void rdg_lane_cross(uint64_t (&v)[E], uint64_t* a, uint32_t lane, bool lower) {
    rdg_wave_lds_sync();                 // every lane is done reading the previous exchange
#pragma unroll
    for (int e = 0; e < E; ++e) a[e * 64 + lane] = v[e];
    rdg_wave_lds_sync();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint64_t p = a[(MIRROR ? E - 1 - e : e) * 64 + (lane ^ (uint32_t)M)];
        const bool take = lower ? (p < v[e]) : (p > v[e]);
        v[e] = take ? p : v[e];
    }
}

// half-cleaners with partner distance J = JE * E >= E (lane distance JE), down to lane distance 1
template <int JE, int E>
__device__ __forceinline__ void rdg_lane_halves_cross(uint64_t (&v)[E], uint64_t* a, uint32_t lane) {
    if constexpr (JE >= 1) {
        rdg_lane_cross<JE, false, E>(v, a, lane, (lane & (uint32_t)JE) == 0u);
        rdg_lane_halves_cross<JE / 2, E>(v, a, lane);
    }
}

// merges whose size KE * E spans several lanes (KE = 2, 4, ..., 64)
template <int KE, int E>
__device__ __forceinline__ void rdg_lane_merges_cross(uint64_t (&v)[E], uint64_t* a, uint32_t lane) {
    if constexpr (KE <= 64) {
        rdg_lane_cross<KE - 1, true, E>(v, a, lane, (lane & (uint32_t)(KE / 2)) == 0u);
        rdg_lane_halves_cross<KE / 4, E>(v, a, lane);
        if constexpr (E >= 2) rdg_lane_halves_reg<E / 2, E>(v);
        rdg_lane_merges_cross<2 * KE, E>(v, a, lane);
    }
}

template <int K, int E>
__device__ __forceinline__ void rdg_lane_merges_reg(uint64_t (&v)[E]) {
    if constexpr (K <= E) {
        rdg_lane_merge_reg<K, E>(v);
        rdg_lane_merges_reg<2 * K, E>(v);
    }
}

// INPLACE: the sorted composites go back where they came from (a chunk of a longer list: rdg_tile_merge_chunks follows)
template <int E, bool INPLACE = false>
__device__ __forceinline__ void rdg_tile_sort_lanes(uint64_t* __restrict__ g, uint32_t n, uint32_t first,
                                                    uint32_t tile, uint64_t* a, uint32_t lane,
                                                    uint32_t* __restrict__ vals_out, uint64_t* __restrict__ keys_full_out) {
    uint64_t v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { const uint32_t i = lane * E + e; v[e] = i < n ? g[i] : RDG_SORT_PAD; }
    rdg_lane_merges_reg<2, E>(v);            // merges of 2 .. E elements: inside the lanes
    rdg_lane_merges_cross<2, E>(v, a, lane); // merges of 2 E .. 64 E elements
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = lane * E + e;
        if (i < n) {
            if (INPLACE) { g[i] = v[e]; continue; }
            vals_out[first + i] = (uint32_t)v[e];
            if (keys_full_out) keys_full_out[first + i] = ((uint64_t)tile << 32) | (v[e] >> 32);
        }
    }
}

template <bool INPLACE>
__device__ __forceinline__ void rdg_tile_sort_lanes_n(uint64_t* __restrict__ g, uint32_t n, uint32_t first, uint32_t tile,
                                                      uint64_t* a, uint32_t lane, uint32_t* __restrict__ vals_out,
                                                      uint64_t* __restrict__ keys_full_out) {
    if (n <= 64) rdg_tile_sort_lanes<1, INPLACE>(g, n, first, tile, a, lane, vals_out, keys_full_out);
    else if (n <= 128) rdg_tile_sort_lanes<2, INPLACE>(g, n, first, tile, a, lane, vals_out, keys_full_out);
    else if (n <= 256) rdg_tile_sort_lanes<4, INPLACE>(g, n, first, tile, a, lane, vals_out, keys_full_out);
    else if (n <= 512) rdg_tile_sort_lanes<8, INPLACE>(g, n, first, tile, a, lane, vals_out, keys_full_out);
    else rdg_tile_sort_lanes<16, INPLACE>(g, n, first, tile, a, lane, vals_out, keys_full_out);
}

// Besides its own tile every wave takes its share of the chunk items of the work list: 1024-instance chunks of lists of
// RDG_TSORT_SMALL + 1 .. RDG_TSORT_MID instances, sorted in place (rdg_tile_sort_large_kernel merges them).  The list is
// empty on an ordinary frame: one load per wave.
__global__ void __launch_bounds__(256)
rdg_tile_sort_lanes_kernel(int n_tiles, const uint2* __restrict__ ranges, uint64_t* __restrict__ comp,
                           uint32_t* __restrict__ vals_out, uint64_t* __restrict__ keys_full_out, long long capacity,
                           const int32_t* __restrict__ num_rendered, const uint32_t* __restrict__ hv_header,
                           const uint2* __restrict__ hv_work) {
    if ((long long)(*num_rendered) > capacity) return;
    __shared__ uint64_t sA[4][RDG_TSORT_SMALL];
#ifdef RDG_ABL_SORT_PAD       // ablation (timing only; profiles/r05_experiments.txt 3): 32 KB more LDS per workgroup (5 -> 2 per CU)
    __shared__ uint32_t sPad[8192];
    if (num_rendered[0] == -12345) sPad[threadIdx.x] = 1u, vals_out[0] = sPad[(threadIdx.x * 7) & 8191];
#endif
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint64_t* a = sA[wv];
    const int tile = blockIdx.x * 4 + (int)wv;
    if (tile < n_tiles) {
        const uint2 rg = ranges[tile];
        const uint32_t n = rg.y - rg.x;
        if (n != 0 && n <= RDG_TSORT_SMALL)
            rdg_tile_sort_lanes_n<false>(comp + rg.x, n, rg.x, (uint32_t)tile, a, lane, vals_out, keys_full_out);
    }
    const uint32_t n_work = hv_header[2];
    for (uint32_t wi = blockIdx.x * 4 + wv; wi < n_work; wi += gridDim.x * 4) {
        const uint2 item = hv_work[wi];           // (tile, chunk)
        const uint2 rg = ranges[item.x];
        const uint32_t lo = item.y * RDG_TSORT_SMALL;
        const uint32_t nc = min((uint32_t)RDG_TSORT_SMALL, rg.y - rg.x - lo);
        rdg_tile_sort_lanes_n<true>(comp + rg.x + lo, nc, 0u, 0u, a, lane, nullptr, nullptr);
    }
}

// ---- heavy tiles: chunk sort + merge tree over several workgroups --------------------------------------------
// A tile with n > RDG_TSORT_LDS instances is cut into chunks of RDG_TSORT_LDS; one workgroup sorts each chunk in LDS,
// then the chunks are merged pairwise up a binary tree.  The tree is climbed WITHOUT waiting: every finished run
// bumps the arrival counter of its parent node; the workgroup whose bump finds the sibling already there does the
// parent's merge (and climbs on), the other one exits.  Nothing ever spins, so no residency assumption is made.
// Runs alternate between the composite buffer and the spare key buffer (`alt`); hand-offs between workgroups follow
// the producer / consumer recipe for non-coherent L2s: every wave drains its stores, workgroup barrier, one lane
// releases at agent scope and bumps the counter; the merging workgroup acquires at agent scope before its first load.
// A merge streams its two runs through LDS in output blocks of RDG_MERGE_OB elements: the block boundaries come from
// a merge-path search over the two runs in global memory (one thread per boundary), each thread then merges
// RDG_MERGE_PER consecutive outputs from LDS (odd stride: conflict-free starts), and the block is written coalesced.
#define RDG_MERGE_OB 4096
#define RDG_MERGE_PER 17

// number of elements of A among the first d outputs of merge(A, B); keys are unique
template <typename PA, typename PB>
__device__ __forceinline__ uint32_t rdg_merge_path(PA A, uint32_t nA, PB B, uint32_t nB, uint32_t d) {
    uint32_t lo = d > nB ? d - nB : 0u, hi = min(d, nA);
    while (lo < hi) {
        const uint32_t m = (lo + hi) >> 1;
        if (A[m] < B[d - 1 - m]) lo = m + 1; else hi = m;
    }
    return lo;
}

// the 16 elements a thread brings in for one output block (elements tid + 256 r of the block's [A-part | B-part]):
// all 16 loads are issued back to back, so a block costs ONE global-memory latency, and the next block's loads are in
// flight while the current one is merged
#define RDG_MERGE_LD (RDG_MERGE_OB / 256)
__device__ __forceinline__ void rdg_merge_fetch(const uint64_t* __restrict__ A, const uint64_t* __restrict__ B,
                                                uint32_t a0, uint32_t la, uint32_t b0, uint32_t lt, uint32_t tid,
                                                uint64_t (&v)[RDG_MERGE_LD]) {
#pragma unroll
    for (int r = 0; r < RDG_MERGE_LD; ++r) {
        const uint32_t i = tid + 256u * r;
        v[r] = i < lt ? (i < la ? A[a0 + i] : B[b0 + (i - la)]) : ~0ull;
    }
}

__device__ void rdg_merge_runs(const uint64_t* __restrict__ A, uint32_t nA, const uint64_t* __restrict__ B, uint32_t nB,
                               uint64_t* __restrict__ O, uint64_t* sIn, uint64_t* sOut, uint32_t* sSplit, uint32_t tid) {
    const uint32_t n = nA + nB;
    const uint32_t nblk = (n + RDG_MERGE_OB - 1) / RDG_MERGE_OB;
    for (uint32_t sb = 0; sb < nblk; sb += 256) {
        const uint32_t nb = min(256u, nblk - sb);
        for (uint32_t t = tid; t <= nb; t += 256) {
            const uint32_t d = min((sb + t) * RDG_MERGE_OB, n);
            sSplit[t] = rdg_merge_path(A, nA, B, nB, d);
        }
        __syncthreads();
        uint64_t v[RDG_MERGE_LD];
        {
            const uint32_t d0 = min(sb * RDG_MERGE_OB, n), d1 = min((sb + 1) * RDG_MERGE_OB, n);
            rdg_merge_fetch(A, B, sSplit[0], sSplit[1] - sSplit[0], d0 - sSplit[0], d1 - d0, tid, v);
        }
        for (uint32_t b = 0; b < nb; ++b) {
            const uint32_t d0 = min((sb + b) * RDG_MERGE_OB, n), d1 = min((sb + b + 1) * RDG_MERGE_OB, n);
            const uint32_t la = sSplit[b + 1] - sSplit[b], lt = d1 - d0, lb = lt - la;
#pragma unroll
            for (int r = 0; r < RDG_MERGE_LD; ++r) sIn[tid + 256u * r] = v[r];
            __syncthreads();
            if (b + 1 < nb) {
                const uint32_t e0 = d1, e1 = min((sb + b + 2) * RDG_MERGE_OB, n);
                rdg_merge_fetch(A, B, sSplit[b + 1], sSplit[b + 2] - sSplit[b + 1], e0 - sSplit[b + 1], e1 - e0, tid, v);
            }
            const uint32_t o0 = tid * RDG_MERGE_PER;
            if (o0 < lt) {
                uint32_t ia = rdg_merge_path(sIn, la, sIn + la, lb, o0), ib = o0 - ia;
                uint64_t va = ia < la ? sIn[ia] : ~0ull, vb = ib < lb ? sIn[la + ib] : ~0ull;
                const uint32_t o1 = min(o0 + RDG_MERGE_PER, lt);
                for (uint32_t o = o0; o < o1; ++o) {
                    if (va < vb) { sOut[o] = va; ++ia; va = ia < la ? sIn[ia] : ~0ull; }
                    else         { sOut[o] = vb; ++ib; vb = ib < lb ? sIn[la + ib] : ~0ull; }
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < RDG_MERGE_LD; ++r) {
                const uint32_t i = tid + 256u * r;
                if (i < lt) O[d0 + i] = sOut[i];
            }
        }
        __syncthreads();   // sSplit is rewritten by the next super-batch
    }
}

// one heavy work item: chunk `chunk` of heavy tile d (leaf sort, then as far up the merge tree as this workgroup gets)
__device__ void rdg_heavy_item(const RdgHeavyDesc d, uint32_t chunk, uint64_t* __restrict__ comp, uint64_t* __restrict__ alt,
                               uint32_t* __restrict__ vals_out, uint64_t* __restrict__ keys_full_out,
                               uint32_t* __restrict__ hv_nodes, uint64_t* sK, uint32_t* sSplit, uint32_t* sOld,
                               uint32_t tid) {
    uint64_t* src = comp + d.start;
    uint64_t* dst = alt + d.start;
    {   // leaf: sort my chunk in LDS, in place
        const uint32_t lo = chunk * (uint32_t)RDG_TSORT_LDS;
        const uint32_t nc = min((uint32_t)RDG_TSORT_LDS, d.n - lo);
        uint32_t N2 = 2;
        while (N2 < nc) N2 <<= 1;
        for (uint32_t i = tid; i < N2; i += 256) sK[i] = i < nc ? src[lo + i] : ~0ull;
        __syncthreads();
        if (nc > 1) rdg_bitonic_sort(sK, nc, N2, tid, 256u);
        for (uint32_t i = tid; i < nc; i += 256) src[lo + i] = sK[i];
    }
    uint32_t idx = chunk;                     // my run's index at the current level
    uint32_t nrun = d.nchunks;                // runs at the current level
    uint32_t len = (uint32_t)RDG_TSORT_LDS;   // nominal run length at the current level
    uint32_t* cnt = hv_nodes + d.node_base;   // counters of the next level's nodes
    while (nrun > 1) {
        const uint32_t parent = idx >> 1;
        const uint32_t lo = parent * 2u * len;
        const uint32_t mid = min(lo + len, d.n), hi = min(lo + 2u * len, d.n);
        if ((idx ^ 1u) < nrun) {
            // publish my run, then see whether the sibling is already there
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                *sOld = atomicAdd(&cnt[parent], 1u);
            }
            __syncthreads();
            if (*sOld == 0u) return;   // first of the two: the sibling's workgroup merges
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            rdg_merge_runs(src + lo, mid - lo, src + mid, hi - mid, dst + lo, sK, sK + RDG_MERGE_OB, sSplit, tid);
        } else {
            // no sibling at this level: carry the run over to the other buffer
            for (uint32_t i = lo + tid; i < hi; i += 256) dst[i] = src[i];
        }
        uint64_t* t = src; src = dst; dst = t;
        cnt += (nrun + 1u) >> 1;
        idx = parent;
        nrun = (nrun + 1u) >> 1;
        len <<= 1;
    }
    // the whole tile is one sorted run in `src`, written by this workgroup: its own stores are visible to it
    __syncthreads();
    for (uint32_t i = tid; i < d.n; i += 256) {
        const uint64_t k = src[i];
        vals_out[d.start + i] = (uint32_t)k;
        // keys_full_out may BE the spare buffer the runs alternate through: element i is read before it is written
        if (keys_full_out) keys_full_out[d.start + i] = ((uint64_t)d.tile << 32) | (k >> 32);
    }
}

// A list of RDG_TSORT_SMALL + 1 .. RDG_TSORT_MID instances whose 1024-instance chunks are sorted (by
// rdg_tile_sort_lanes_kernel, the launch before): one workgroup merges them in LDS -- run length 1024, 2048 -- ping-pong
// between the two halves of sK; every thread produces n / 256 consecutive outputs of a level from a merge-path split.
// (The workgroup-wide bitonic network this replaces pads to a power of two and moves every element through LDS in each of
// its 66-78 stages: a scene with 2 000 instances per tile spent 1.2 ms there, three quarters of its binning time.)
__device__ __forceinline__ void rdg_tile_merge_chunks(const uint64_t* __restrict__ g, uint32_t n, uint32_t first,
                                                      uint32_t tile, uint64_t* sK, uint32_t tid,
                                                      uint32_t* __restrict__ vals_out,
                                                      uint64_t* __restrict__ keys_full_out) {
    // (A merge by RANK -- every element binary-searches the other run, 16 searches side by side per thread -- was built
    // and measured: 426 us on the dense scene against 162 us for this merge path; its 12 x 16 random 8-byte LDS reads per
    // thread and level cost more than the dependent chain they replace.)
    uint64_t* src = sK;
    uint64_t* dst = sK + RDG_TSORT_MID;
    for (uint32_t i = tid; i < n; i += 256) src[i] = g[i];
    __syncthreads();
    const uint32_t per = (n + 255) / 256;
    for (uint32_t len = RDG_TSORT_SMALL; len < n; len <<= 1) {
        const uint32_t d0 = min(tid * per, n), d1 = min(d0 + per, n);
        if (d0 < d1) {
            // the pair of runs output d0 falls into; a thread's outputs may run over into the next pair
            uint32_t base = d0 / (2 * len) * (2 * len);
            uint32_t nA = min(len, n - base), nB = min(len, n - base - nA);
            uint32_t i = rdg_merge_path(src + base, nA, src + base + nA, nB, d0 - base), j = d0 - base - i;
            // the heads of the two runs stay in registers: one LDS read per output (the consumed side's next element)
            uint64_t x = i < nA ? src[base + i] : ~0ull, y = j < nB ? src[base + nA + j] : ~0ull;
            for (uint32_t d = d0; d < d1; ++d) {
                if (d - base == nA + nB) {        // next pair
                    base += 2 * len;
                    nA = min(len, n - base); nB = min(len, n - base - nA);
                    i = 0; j = 0;
                    x = nA ? src[base] : ~0ull; y = nB ? src[base + nA] : ~0ull;
                }
                const bool ta = x < y;
                dst[d] = ta ? x : y;
                if (ta) { ++i; x = i < nA ? src[base + i] : ~0ull; }
                else { ++j; y = j < nB ? src[base + nA + j] : ~0ull; }
            }
        }
        __syncthreads();
        uint64_t* t = src; src = dst; dst = t;
    }
    for (uint32_t i = tid; i < n; i += 256) {
        const uint64_t k = src[i];
        vals_out[first + i] = (uint32_t)k;
        if (keys_full_out) keys_full_out[first + i] = ((uint64_t)tile << 32) | (k >> 32);
    }
}

// Lists of more than RDG_TSORT_SMALL instances: a fixed grid of workgroups walks the device-side work list the scan
// kernel wrote (tiles of up to RDG_TSORT_LDS instances: one LDS sort each; heavier tiles: one item per chunk, see
// above).  On an ordinary frame the list is empty and the launch costs one load per workgroup.
__global__ void __launch_bounds__(256)
rdg_tile_sort_large_kernel(const uint2* __restrict__ ranges, uint64_t* __restrict__ comp, uint32_t* __restrict__ vals_out,
                           uint64_t* __restrict__ keys_full_out, long long capacity,
                           const int32_t* __restrict__ num_rendered, uint64_t* __restrict__ alt,
                           const uint32_t* __restrict__ hv_header, const RdgHeavyDesc* __restrict__ hv_desc,
                           const uint2* __restrict__ hv_work, uint32_t* __restrict__ hv_nodes) {
    if ((long long)(*num_rendered) > capacity) return;
    __shared__ uint64_t sK[RDG_TSORT_LDS];
    __shared__ uint32_t sSplit[257];
    __shared__ uint32_t sOld;
    const uint32_t tid = threadIdx.x;
    const uint32_t n_work = hv_header[0];
    for (uint32_t wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
        __syncthreads();                       // the previous item is done with sK / sSplit / sOld
        const uint2 item = hv_work[wi];
        if (item.x & 0x80000000u) {
            const uint32_t tile = item.x & 0x7fffffffu;
            const uint2 rg = ranges[tile];
            const uint32_t n = rg.y - rg.x;
            if (item.y != 0xffffffffu) {
                // a mid list whose chunks the launch before has sorted: merge them
                rdg_tile_merge_chunks(comp + rg.x, n, rg.x, tile, sK, tid, vals_out, keys_full_out);
                continue;
            }
            uint32_t N2 = 2;
            while (N2 < n) N2 <<= 1;
            const uint64_t* g = comp + rg.x;
            for (uint32_t i = tid; i < N2; i += 256) sK[i] = i < n ? g[i] : ~0ull;
            __syncthreads();
            rdg_bitonic_sort(sK, n, N2, tid, 256u);
            for (uint32_t i = tid; i < n; i += 256) {
                const uint64_t k = sK[i];
                vals_out[rg.x + i] = (uint32_t)k;
                if (keys_full_out) keys_full_out[rg.x + i] = ((uint64_t)tile << 32) | (k >> 32);
            }
        } else {
            rdg_heavy_item(hv_desc[item.x], item.y, comp, alt, vals_out, keys_full_out, hv_nodes, sK, sSplit, &sOld, tid);
        }
    }
}

// largest tile list of the radix path (the bucket path gets it from its scan kernel): num_rendered[1]
__global__ void __launch_bounds__(1024)
rdg_tile_max_kernel(int n_tiles, const uint2* __restrict__ ranges, long long capacity, int32_t* __restrict__ num_rendered,
                    int32_t* __restrict__ host_mirror) {
    __shared__ uint32_t sMax;
    if (threadIdx.x == 0) sMax = 0u;
    __syncthreads();
    uint32_t m = 0;
    if ((long long)num_rendered[0] <= capacity)
        for (int i = threadIdx.x; i < n_tiles; i += 1024) { const uint2 r = ranges[i]; m = max(m, r.y - r.x); }
    atomicMax(&sMax, m);
    __syncthreads();
    if (threadIdx.x == 0) {
        num_rendered[1] = (int32_t)sMax;
        if (host_mirror) { host_mirror[1] = (int32_t)sMax; __threadfence_system(); host_mirror[0] = num_rendered[0]; }
    }
}

// ---------------------------------------------------------------------------------------------------------
// radix binning, depth first: helper kernels (see rdg_launch_bin)
// ---------------------------------------------------------------------------------------------------------
// depth key of every Gaussian (its view-space depth's bits: positive floats order like their bits; Gaussians that touch no
// tile go last), value = its index; n_p[0] = P for the sort
__global__ void __launch_bounds__(RDG_PRE_BLOCK)
rdg_depth_keys_kernel(int P, const uint4* __restrict__ rectd,
                      uint32_t* __restrict__ dkey, uint32_t* __restrict__ dval, int32_t* __restrict__ n_p) {
    const int i = blockIdx.x * RDG_PRE_BLOCK + threadIdx.x;
    if (i == 0) n_p[0] = P;
    if (i >= P) return;
    const uint4 rd = rectd[i];
    dkey[i] = rd.w > 0 ? rd.z : 0xFFFFFFFFu;
    dval[i] = (uint32_t)i;
}

// psum[b] = instances of the 256 Gaussians at depth ranks [256 b, 256 b + 256)
__global__ void __launch_bounds__(RDG_PRE_BLOCK)
rdg_perm_block_sums_kernel(int P, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ tiles_touched,
                           uint32_t* __restrict__ psum) {
    __shared__ uint32_t wsum[RDG_PRE_BLOCK / RDG_WAVE];
    const int j = blockIdx.x * RDG_PRE_BLOCK + threadIdx.x;
    const uint32_t t = j < P ? tiles_touched[perm[j]] : 0u;
    const uint32_t inc = rdg_wave_scan_incl(t);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0) psum[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// exclusive scan of v[0..n) in place by one workgroup, v[n] = total (n <= 16 k block sums: runs of <= 16 per thread)
__global__ void __launch_bounds__(1024) rdg_scan_u32_kernel(uint32_t* __restrict__ v, int n) {
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
    if (threadIdx.x == 1023) v[n] = run;
}

// rdg_duplicate_kernel over the Gaussians in DEPTH order (slot j <-> Gaussian perm[j]): (tile id, Gaussian id) pairs, a
// Gaussian's tiles row by row; slots dealt round-robin to the threads of a block (coalesced stores however skewed the
// footprints are)
__global__ void __launch_bounds__(RDG_PRE_BLOCK)
rdg_duplicate_sorted_kernel(int P, int gx, int gy, const uint32_t* __restrict__ perm, const uint4* __restrict__ rectd,
                            const uint32_t* __restrict__ psum, uint32_t* __restrict__ tkey, uint32_t* __restrict__ vals,
                            long long capacity, const int32_t* __restrict__ num_rendered) {
    if ((long long)(*num_rendered) > capacity) return;
    __shared__ uint32_t sOff[RDG_PRE_BLOCK], sId[RDG_PRE_BLOCK];
    __shared__ uint16_t sX0[RDG_PRE_BLOCK], sY0[RDG_PRE_BLOCK], sW[RDG_PRE_BLOCK];
    __shared__ uint32_t wsum[RDG_PRE_BLOCK / RDG_WAVE];
    const int tid = threadIdx.x;
    const int j = blockIdx.x * RDG_PRE_BLOCK + tid;
    uint32_t t = 0;
    if (j < P) {
        const uint32_t i = perm[j];
        sId[tid] = i;
        const uint4 rd = rectd[i];
        t = rd.w;
        if (t > 0) {
            sX0[tid] = (uint16_t)(rd.x & 0xffffu); sY0[tid] = (uint16_t)(rd.x >> 16);
            sW[tid] = (uint16_t)((rd.y & 0xffffu) - (rd.x & 0xffffu));
        }
    }
    const uint32_t inc = rdg_wave_scan_incl(t);
    const uint32_t lane = tid & 63, w = tid >> 6;
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t k = 0; k < w; ++k) woff += wsum[k];
    sOff[tid] = woff + inc - t;
    __syncthreads();
    const uint32_t base = psum[blockIdx.x];
    const uint32_t total = psum[blockIdx.x + 1] - base;
    for (uint32_t k = tid; k < total; k += RDG_PRE_BLOCK) {
        int lo = 0, hi = RDG_PRE_BLOCK - 1;        // last slot owner with sOff <= k
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (sOff[mid] <= k) lo = mid; else hi = mid - 1;
        }
        const uint32_t r = k - sOff[lo];
        const uint32_t wd = sW[lo];
        const uint32_t ry = r / wd, rx = r - ry * wd;
        tkey[base + k] = (uint32_t)(sY0[lo] + ry) * (uint32_t)gx + (uint32_t)(sX0[lo] + rx);
        vals[base + k] = sId[lo];
    }
}

// identifyTileRanges on sorted 32-bit tile ids
__global__ void __launch_bounds__(256)
rdg_tile_ranges32_kernel(const uint32_t* __restrict__ tk, long long capacity, const int32_t* __restrict__ n_dev,
                         uint2* __restrict__ ranges) {
    const long long n = *n_dev;
    if (n > capacity) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t tile = tk[i];
        if (i == 0) ranges[tile].x = 0;
        else {
            const uint32_t prev = tk[i - 1];
            if (prev != tile) { ranges[prev].y = (uint32_t)i; ranges[tile].x = (uint32_t)i; }
        }
        if (i == n - 1) ranges[tile].y = (uint32_t)n;
    }
}

// tests: the 64-bit (tile | depth) key of every sorted instance
__global__ void __launch_bounds__(256)
rdg_rebuild_keys_kernel(const uint32_t* __restrict__ tk, const uint32_t* __restrict__ vals, const RdgRec* __restrict__ rec,
                        uint64_t* __restrict__ keys, long long capacity, const int32_t* __restrict__ n_dev) {
    const long long n = *n_dev;
    if (n > capacity) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        keys[i] = ((uint64_t)tk[i] << 32) | (uint64_t)__float_as_uint(rec[vals[i]].q1.z);
}

int rdg_launch_bin(const RdgDev& d, const void* geom_ws, const int32_t* radii, void* bin_ws, int64_t capacity,
                   void* image_ws, int32_t* num_rendered, uint64_t* keys_unsorted_copy,
                   uint32_t* vals_unsorted_copy, hipStream_t s, bool radix_export_keys) {
    const RdgGeomLayout G = rdg_geom_layout(d.P);
    const RdgBinLayout B = rdg_bin_layout(capacity);
    const RdgImageLayout I = rdg_image_layout(d.H, d.W);
    const char* g = (const char*)geom_ws;
    char* b = (char*)bin_ws;
    uint64_t* keys_a = (uint64_t*)(b + B.keys_a);
    uint64_t* keys_b = (uint64_t*)(b + B.keys_b);
    uint32_t* vals_a = (uint32_t*)(b + B.vals_a);
    uint32_t* vals_b = (uint32_t*)(b + B.vals_b);
    const int n_tiles = d.gx * d.gy;
    uint2* ranges = (uint2*)((char*)image_ws + I.ranges);

    // RdgRasterSettings.bin_mode alone picks the algorithm (the Python host turns RDG_BIN_MODE=radix into bin_mode = 1
    // when it loads: no process-global switch in the library)
    if (d.bin_mode != 1) {
        const int npass = (rdg_key_bits(n_tiles) + RDG_SORT_BITS - 1) / RDG_SORT_BITS;
        uint32_t* vals_out = (npass & 1) ? vals_b : vals_a;      // where the compositing kernels look
        uint64_t* keys_out = (npass & 1) ? keys_b : keys_a;      // full (tile | depth) keys, tests only
        uint64_t* comp = (npass & 1) ? keys_a : keys_b;          // composites live in the other key buffer
        uint32_t* tile_cnt = (uint32_t*)((char*)image_ws + I.tile_cnt);
        uint32_t* tile_fill = (uint32_t*)((char*)image_ws + I.tile_fill);
        uint32_t* rank_buf = (npass & 1) ? vals_a : vals_b;      // the vals buffer nobody reads afterwards
        const bool want_keys = keys_unsorted_copy != nullptr || vals_unsorted_copy != nullptr || radix_export_keys;
        const int nblk = (d.P + RDG_PRE_BLOCK - 1) / RDG_PRE_BLOCK;
        if (want_keys && d.P > 0 && (keys_unsorted_copy || vals_unsorted_copy)) {
            // emission-order (key, value) stream for the parity tests: the radix path's duplicate kernel
            uint32_t* vscr = (npass & 1) ? vals_a : vals_b;
            hipLaunchKernelGGL(rdg_duplicate_kernel, dim3(nblk), dim3(RDG_PRE_BLOCK), 0, s, d.P, d.gx, d.gy,
                               (const uint4*)(g + G.rectd),
                               (const uint32_t*)(g + G.block_sums), keys_out, vscr, (long long)capacity, num_rendered);
            hipLaunchKernelGGL(rdg_copy_pairs_kernel, dim3(1024), dim3(256), 0, s, keys_out, vscr, keys_unsorted_copy,
                               vals_unsorted_copy, (long long)capacity, num_rendered);
        }
        const RdgHeavyLayout HL = rdg_heavy_layout(capacity);
        char* hv = b + B.heavy;
        uint32_t* hv_header = (uint32_t*)(hv + HL.header);
        RdgHeavyDesc* hv_desc = (RdgHeavyDesc*)(hv + HL.desc);
        uint2* hv_work = (uint2*)(hv + HL.work);
        uint32_t* hv_nodes = (uint32_t*)(hv + HL.nodes);
        {
        RdgStageScope scope(RDG_STAGE_SCAN_DUP, s);
        int zbits = 0;
        while (((size_t)1 << zbits) < rdg_cnt_entries(d.gx, d.gy)) ++zbits;
        if (!d.tile_cnt_zeroed) {
            hipError_t em = rdg_zero_async(tile_cnt, rdg_cnt_entries(d.gx, d.gy) * 4, s);
            if (em != hipSuccess) return rdg_check_hip(em, "tile_cnt memset");
        }
        if (d.P > 0)
            hipLaunchKernelGGL(rdg_tile_bucket_kernel<0>, dim3(nblk), dim3(RDG_PRE_BLOCK), 0, s, d.P, d.gx, d.gy, zbits,
                               (const uint4*)(g + G.rectd),
#ifdef RDG_ABL_EXACT_CULL
                               (const RdgRec*)(g + G.rec),
#endif
                               (const uint32_t*)(g + G.block_sums), tile_cnt, ranges, rank_buf, comp,
                               (long long)capacity, num_rendered, (uint4*)nullptr, 0ll);
        hipLaunchKernelGGL(rdg_tile_scan_kernel, dim3(1), dim3(1024), 0, s, n_tiles, d.gx, tile_cnt, ranges, tile_fill,
                           (long long)capacity, num_rendered, hv_header, hv_desc, hv_work, hv_nodes, HL.max_heavy,
                           HL.max_chunks, HL.max_work, d.nren_stats ? num_rendered + 1 : nullptr,
                           (uint2*)(hv + HL.chunks), HL.max_chunk_items, d.nren_host);
        if (d.P > 0)
            hipLaunchKernelGGL(rdg_tile_bucket_kernel<1>, dim3(nblk), dim3(RDG_PRE_BLOCK), 0, s, d.P, d.gx, d.gy, zbits,
                               (const uint4*)(g + G.rectd),
#ifdef RDG_ABL_EXACT_CULL
                               (const RdgRec*)(g + G.rec),
#endif
                               (const uint32_t*)(g + G.block_sums), tile_cnt, ranges, rank_buf, comp,
                               (long long)capacity, num_rendered, (uint4*)(b + B.hit),
                               (long long)(rdg_hit_bytes(capacity, n_tiles) / 16));
        else {
            hipError_t eh = rdg_zero_async(b + B.hit, rdg_hit_bytes(capacity, n_tiles), s);
            if (eh != hipSuccess) return rdg_check_hip(eh, "hit bits memset");
        }
        }
        RdgStageScope scope(RDG_STAGE_SORT, s);
        uint64_t* kfull = radix_export_keys ? keys_out : nullptr;
        hipLaunchKernelGGL(rdg_tile_sort_lanes_kernel, dim3((n_tiles + 3) / 4), dim3(256), 0, s, n_tiles, ranges, comp,
                           vals_out, kfull, (long long)capacity, num_rendered, hv_header, (const uint2*)(hv + HL.chunks));
        // lists above 1024 instances: a fixed grid walks the device-side work list (empty on an ordinary frame)
        hipLaunchKernelGGL(rdg_tile_sort_large_kernel, dim3(HL.max_work < 1024u ? HL.max_work : 1024u), dim3(256), 0, s,
                           ranges, comp, vals_out, kfull, (long long)capacity, num_rendered, keys_out, hv_header, hv_desc,
                           hv_work, hv_nodes);
        return rdg_check_hip(hipGetLastError(), "bucket bin launch");
    }

    // ---- radix binning, depth first (bin_mode 1) ---------------------------------------------------------------------
    // (1) sort the Gaussians by depth: P 32-bit keys, four passes;  (2) emit their tile instances IN THAT ORDER as
    // (tile id, Gaussian id) pairs;  (3) partition the pairs stably by tile: ceil(log2 tiles / 8) passes over 8 B
    // per pair (two at 1080p and 4K) -- the (tile, depth, Gaussian id) order of the 64-bit (tile | depth) key sort,
    // reached with a third of its passes on a third of its bytes; equal depths keep the order of their Gaussian
    // indices (both sorts are stable).
    {
        hipError_t eh = rdg_zero_async(b + B.hit, rdg_hit_bytes(capacity, n_tiles), s);
        if (eh != hipSuccess) return rdg_check_hip(eh, "hit bits memset");
    }
    const int nblk = (d.P + RDG_PRE_BLOCK - 1) / RDG_PRE_BLOCK;
    const int tile_bits = rdg_key_bits(n_tiles) - 32;
    const int npass = (tile_bits + RDG_SORT_BITS - 1) / RDG_SORT_BITS;       // same parity as the 64-bit sort's pass count
    uint32_t* tkey_a = (uint32_t*)keys_a;
    uint32_t* tkey_b = (uint32_t*)keys_b;
    if ((keys_unsorted_copy || vals_unsorted_copy) && d.P > 0) {
        // emission-order (key, value) stream of the reference algorithm, for the parity tests only
        hipLaunchKernelGGL(rdg_duplicate_kernel, dim3(nblk), dim3(RDG_PRE_BLOCK), 0, s, d.P, d.gx, d.gy,
                           (const uint4*)(g + G.rectd),
                           (const uint32_t*)(g + G.block_sums), keys_b, vals_b, (long long)capacity, num_rendered);
        hipLaunchKernelGGL(rdg_copy_pairs_kernel, dim3(1024), dim3(256), 0, s, keys_b, vals_b, keys_unsorted_copy,
                           vals_unsorted_copy, (long long)capacity, num_rendered);
    }
    {
        RdgStageScope scope(RDG_STAGE_SCAN_DUP, s);
        if (d.P > 0) {
            // scratch for the per-Gaussian sort lives in the second key buffer (8 cap >= 32 P + ...: cap >= 4 P + 4096)
            const size_t Pp = rdg_align_up((size_t)d.P * 4, 256);
            char* q = (char*)keys_b;
            uint32_t* dk_a = (uint32_t*)q;              uint32_t* dk_b = (uint32_t*)(q + Pp);
            uint32_t* dv_a = (uint32_t*)(q + 2 * Pp);   uint32_t* dv_b = (uint32_t*)(q + 3 * Pp);
            uint32_t* psum = (uint32_t*)(q + 4 * Pp);                               // nblk + 1 block sums, sorted order
            int32_t* n_p = (int32_t*)(q + 4 * Pp + rdg_align_up((size_t)(nblk + 1) * 4, 256));   // [0] = P, [1] = scratch
            if ((size_t)((char*)(n_p + 2) - q) > (size_t)capacity * 8)
                return rdg_set_error("radix binning: capacity %lld too small for %d Gaussians", (long long)capacity, d.P);
            hipLaunchKernelGGL(rdg_depth_keys_kernel, dim3(nblk), dim3(RDG_PRE_BLOCK), 0, s, d.P,
                               (const uint4*)(g + G.rectd), dk_a, dv_a, n_p);
            int in_b = 0;
            int rc = rdg_launch_radix_sort<uint32_t>(dk_a, dk_b, dv_a, dv_b, (int64_t)d.P, n_p, 0, 32, b + B.sort_tmp, &in_b, s);
            if (rc) return rc;
            const uint32_t* perm = in_b ? dv_b : dv_a;
            hipLaunchKernelGGL(rdg_perm_block_sums_kernel, dim3(nblk), dim3(RDG_PRE_BLOCK), 0, s, d.P, perm,
                               (const uint32_t*)(g + G.tiles_touched), psum);
            hipLaunchKernelGGL(rdg_scan_u32_kernel, dim3(1), dim3(1024), 0, s, psum, nblk);
            hipLaunchKernelGGL(rdg_duplicate_sorted_kernel, dim3(nblk), dim3(RDG_PRE_BLOCK), 0, s, d.P, d.gx, d.gy, perm,
                               (const uint4*)(g + G.rectd), psum, tkey_a,
                               vals_a, (long long)capacity, num_rendered);
        }
    }
    int in_b = 0;
    {
        RdgStageScope scope(RDG_STAGE_SORT, s);
        int rc = rdg_launch_radix_sort<uint32_t>(tkey_a, tkey_b, vals_a, vals_b, capacity, num_rendered, 0, tile_bits,
                                               b + B.sort_tmp, &in_b, s);
        if (rc) return rc;
    }
    RdgStageScope scope(RDG_STAGE_RANGES, s);
    hipError_t e = rdg_zero_async(ranges, (size_t)n_tiles * sizeof(uint2), s);
    if (e != hipSuccess) return rdg_check_hip(e, "ranges memset");
    const uint32_t* tk_sorted = in_b ? tkey_b : tkey_a;
    hipLaunchKernelGGL(rdg_tile_ranges32_kernel, dim3(2048), dim3(256), 0, s, tk_sorted, (long long)capacity, num_rendered,
                       ranges);
    if (d.nren_stats)
        hipLaunchKernelGGL(rdg_tile_max_kernel, dim3(1), dim3(1024), 0, s, n_tiles, ranges, (long long)capacity,
                           num_rendered, d.nren_host);
    if (radix_export_keys) {
        // tests only: the sorted 64-bit (tile | depth) keys, where rdg_bin_forward looks for them (the buffer the sorted
        // tile ids are in: built in the other one, then copied over)
        uint64_t* kx = in_b ? keys_b : keys_a;
        uint64_t* ky = in_b ? keys_a : keys_b;
        hipLaunchKernelGGL(rdg_rebuild_keys_kernel, dim3(2048), dim3(256), 0, s, tk_sorted, in_b ? vals_b : vals_a,
                           (const RdgRec*)(g + G.rec), ky, (long long)capacity, num_rendered);
        hipLaunchKernelGGL(rdg_copy_pairs_kernel, dim3(1024), dim3(256), 0, s, ky, (const uint32_t*)nullptr, kx,
                           (uint32_t*)nullptr, (long long)capacity, num_rendered);
    }
    (void)npass;
    return rdg_check_hip(hipGetLastError(), "bin launch");
}
