// rdg_radix_sort.hip -- stable LSD radix sort of (key, value) pairs on 8-bit digits, written for wave64 / gfx950.
//
// Three launches per pass, no communication between workgroups inside a launch (nothing spins, nothing can deadlock):
//   count   : the input is cut into up to 8192 contiguous SEGMENTS of whole 4096-pair tiles; a workgroup counts the digits
//             of its segment -> table[segment][digit];
//   scan    : one workgroup per digit: exclusive prefix over the segments + the digit's total;
//   scatter : the workgroup walks its segment tile by tile: ranks the tile's 4096 pairs (stable), RE-ORDERS THE TILE IN LDS
//             and writes it out as runs of equal digits -- coalesced stores (a run of the tile's 16 pairs per digit on
//             uniform digits, whole kilobytes on clustered ones such as tile ids), where the sort this replaces stored
//             every pair on its own (1.7-2.3 TB/s on 32 B per pair and pass).
// Keys are uint32 or uint64 (template), values uint32: 4 + 8 + 8 (or 8 + 12 + 12) bytes per pair and pass.  n is read
// from device memory (no host round trip); a count above the capacity makes every kernel return.
//
// Stability: a tile's pairs are held wave-striped (wave w, item i, lane l <-> index w*64*ITEMS + i*64 + l) and ranked item by
// item with a ballot "match" over the digit's bits; waves in order, tiles in order, segments in order.
//
// A chained-scan ("onesweep") form of the pass -- one launch, the tile's place in the output taken from its predecessors
// through a decoupled look-back -- was built first and measured (round 4): bit-exact, and 2.3 TB/s, because with 8 XCDs
// the agent-scope status loads of the look-back cost a microsecond apiece and a tile of the first dispatch wave has
// hundreds of predecessors to walk: 120 us per pass + a 50 us histogram launch per sort at 17 M pairs against 37 + 11 + 78 us
// (count, scan, scatter) for the three launches here at that time, 13 + 12 + 85 us since the count pass uses plain LDS
// atomics; 18 against 22 us at 1 M keys (profiles/r04_experiments.txt).  The three-launch form is kept.
#include "rdg_common.h"

#define RDG_RS_THREADS 256
// pairs per thread and tile: 16 (tiles of 4096 pairs) for large inputs; 4 (tiles of 1024) for small ones, where 4096-pair
// tiles leave most of the chip idle (1 M keys = 245 tiles on 256 CUs, each a 20 us chain of dependent steps)
#define RDG_RS_SMALL_BELOW (1ll << 22)
static inline int rdg_rs_items(int64_t capacity) { return capacity < RDG_RS_SMALL_BELOW ? 4 : 16; }
#define RDG_RS_MAXSEG 8192     // segments: one tile each up to 33 M pairs (4 M with the small tiles) -- many short workgroups

static inline int rdg_rs_nseg(int64_t capacity) {
    const long long tile = (long long)RDG_RS_THREADS * rdg_rs_items(capacity);
    long long t = ((capacity > 0 ? capacity : 1) + tile - 1) / tile;
    if (t > RDG_RS_MAXSEG) t = RDG_RS_MAXSEG;
    return (int)t;
}

size_t rdg_radix_sort_tmp_bytes(int64_t capacity) {
    // table[n_seg][256] + totals[256], for a sort of `capacity` pairs OR ANY SMALLER NUMBER in the same workspace (the
    // binning stage sorts P depth keys and D tile ids in one; a K-NN workspace serves every sample size up to its own):
    // the segment count is not monotone in the capacity -- below 4 M pairs the tiles are 1024 pairs, above 4096 -- so the
    // table is sized for the larger of the two counts a capacity <= this one can produce
    const int64_t small_cap = capacity < RDG_RS_SMALL_BELOW ? capacity : RDG_RS_SMALL_BELOW - 1;
    int nseg = rdg_rs_nseg(capacity);
    if (rdg_rs_nseg(small_cap) > nseg) nseg = rdg_rs_nseg(small_cap);
    return rdg_align_up((size_t)256 * (size_t)nseg * 4, 256) + 1024;
}

// tiles [t0, t1) of segment `seg` when n pairs are cut into nseg segments of whole tiles
template <int ITEMS>
__device__ __forceinline__ void rdg_rs_bounds(long long n, int nseg, int seg, long long& t0, long long& t1) {
    const long long tiles = (n + RDG_RS_THREADS * ITEMS - 1) / (RDG_RS_THREADS * ITEMS);
    const long long per = (tiles + nseg - 1) / nseg;
    t0 = (long long)seg * per; t1 = t0 + per;
    if (t0 > tiles) t0 = tiles;
    if (t1 > tiles) t1 = tiles;
}

// lanes of the wave whose `digit` equals mine (act lanes only)
__device__ __forceinline__ unsigned long long rdg_rs_match(uint32_t digit, bool act) {
    unsigned long long m = __ballot(act);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
        const bool bset = (digit >> bit) & 1u;
        const unsigned long long bal = __ballot(act && bset);
        m &= bset ? bal : ~bal;
    }
    return m;
}

template <typename KeyT, int RDG_RS_ITEMS>
__global__ void __launch_bounds__(RDG_RS_THREADS)
rdg_rs_count_kernel(const KeyT* __restrict__ keys, long long capacity, const int32_t* __restrict__ n_dev, int shift,
                    int nseg, uint32_t* __restrict__ table) {
    constexpr int RDG_RS_TILE = RDG_RS_THREADS * RDG_RS_ITEMS;
    const long long n = *n_dev;
    if (n > capacity) return;
    __shared__ uint32_t sCnt[4][256];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int w = 0; w < 4; ++w) sCnt[w][tid] = 0u;
    __syncthreads();
    long long t0, t1;
    rdg_rs_bounds<RDG_RS_ITEMS>(n, nseg, blockIdx.x, t0, t1);
    // Plain LDS atomics on the wave's own copy of the counters (no return value: one instruction per key; equal digits in
    // a wave serialise on their bank, at worst 64 cycles -- a third of what ranking the wave by ballot "match" cost in
    // instructions: 37 -> 13 us per pass at 17 M 32-bit keys.  The scatter pass needs the match for its stable ranks; a count does not)
    uint32_t* cw = sCnt[wv];
    for (long long t = t0; t < t1; ++t) {
        const long long base = t * RDG_RS_TILE + wv * (RDG_RS_ITEMS * 64) + lane;
        KeyT key[RDG_RS_ITEMS];
#pragma unroll
        for (int i = 0; i < RDG_RS_ITEMS; ++i) key[i] = (base + i * 64 < n) ? keys[base + i * 64] : (KeyT)0;
#pragma unroll
        for (int i = 0; i < RDG_RS_ITEMS; ++i) {
            if (base + i * 64 < n) atomicAdd(&cw[(uint32_t)(key[i] >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    table[(size_t)blockIdx.x * 256 + tid] = (sCnt[0][tid] + sCnt[1][tid]) + (sCnt[2][tid] + sCnt[3][tid]);   // one 1-KB row
}

// one workgroup per digit row: exclusive scan over the segments (in place) + the row's total
__global__ void __launch_bounds__(1024)
rdg_rs_scan_kernel(uint32_t* __restrict__ table, int nseg, uint32_t* __restrict__ totals, long long capacity,
                   const int32_t* __restrict__ n_dev) {
    if ((long long)(*n_dev) > capacity) return;
    __shared__ uint32_t wtot[16];
    // table[segment][digit]: the count and scatter kernels touch whole 1-KB rows; this kernel (256 workgroups on a table
    // that sits in the L2) walks a column
    uint32_t* col = table + blockIdx.x;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // every thread owns a run of consecutive segments: one block scan of the 1024 run totals (nseg <= 8192: runs of <= 8)
    const int per = (nseg + 1023) / 1024;
    const int s0 = min(nseg, (int)threadIdx.x * per), s1 = min(nseg, s0 + per);
    uint32_t v[8];
    uint32_t mine = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) { v[c] = (c < per && s0 + c < s1) ? col[(size_t)(s0 + c) * 256] : 0u; mine += v[c]; }
    const uint32_t inc = rdg_wave_scan_incl(mine);
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    uint32_t run = inc - mine;
    for (uint32_t k = 0; k < w; ++k) run += wtot[k];
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (c < per && s0 + c < s1) { col[(size_t)(s0 + c) * 256] = run; run += v[c]; }
    if (threadIdx.x == 1023) totals[blockIdx.x] = run;
}

template <typename KeyT, int RDG_RS_ITEMS>
__global__ void __launch_bounds__(RDG_RS_THREADS)
rdg_rs_scatter_kernel(const KeyT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, KeyT* __restrict__ keys_out,
                      uint32_t* __restrict__ vals_out, long long capacity, const int32_t* __restrict__ n_dev, int shift,
                      int nseg, const uint32_t* __restrict__ table, const uint32_t* __restrict__ totals) {
    constexpr int RDG_RS_TILE = RDG_RS_THREADS * RDG_RS_ITEMS;
    const long long n = *n_dev;
    if (n > capacity) return;
    __shared__ KeyT sKey[RDG_RS_TILE];
    __shared__ uint32_t sVal[RDG_RS_TILE];
    __shared__ uint32_t sCnt[4][256];            // per wave: running digit counts, then the wave's offset inside the digit's run
    __shared__ uint32_t sLocal[256];             // first position of the digit's run inside the sorted tile
    __shared__ long long sGlob[256];             // output index of the run's first pair, minus sLocal
    __shared__ uint32_t sWtot[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    long long t0, t1;
    rdg_rs_bounds<RDG_RS_ITEMS>(n, nseg, blockIdx.x, t0, t1);
    if (t0 >= t1) return;
    // thread = digit: where this segment's first pair of that digit goes
    long long cursor;
    {
        const uint32_t v = totals[tid];
        const uint32_t inc = rdg_wave_scan_incl(v);
        if (lane == 63) sWtot[wv] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int k = 0; k < wv; ++k) woff += sWtot[k];
        cursor = (long long)(woff + inc - v) + (long long)table[(size_t)blockIdx.x * 256 + tid];
        __syncthreads();
    }
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    volatile uint32_t* cw = sCnt[wv];
    for (long long t = t0; t < t1; ++t) {
        const long long base = t * RDG_RS_TILE;
        const int n_tile = (int)((n - base) < RDG_RS_TILE ? (n - base) : RDG_RS_TILE);
#pragma unroll
        for (int w = 0; w < 4; ++w) sCnt[w][tid] = 0u;
        // ---- load (wave-striped) and rank ------------------------------------------------------------------------
        KeyT key[RDG_RS_ITEMS];
        uint32_t val[RDG_RS_ITEMS], rank[RDG_RS_ITEMS];
#pragma unroll
        for (int i = 0; i < RDG_RS_ITEMS; ++i) {
            const int loc = wv * (RDG_RS_ITEMS * 64) + i * 64 + lane;
            key[i] = (KeyT)0; val[i] = 0u;
            if (loc < n_tile) { key[i] = keys_in[base + loc]; val[i] = vals_in[base + loc]; }
        }
        __syncthreads();                         // counters cleared (and the previous tile's LDS copies read)
#pragma unroll
        for (int i = 0; i < RDG_RS_ITEMS; ++i) {
            const int loc = wv * (RDG_RS_ITEMS * 64) + i * 64 + lane;
            const bool act = loc < n_tile;
            const uint32_t d = (uint32_t)(key[i] >> shift) & 255u;
            const unsigned long long m = rdg_rs_match(d, act);
            const uint32_t below = (uint32_t)__popcll(m & lt_mask);
            uint32_t prev = 0u;
            if (act) prev = cw[d];               // every lane of a group reads the count before its lowest lane moves it on
            rdg_wave_lds_sync();
            if (act && below == 0u) cw[d] = prev + (uint32_t)__popcll(m);
            rdg_wave_lds_sync();
            rank[i] = prev + below;
        }
        __syncthreads();
        // ---- thread = digit: the waves' offsets inside the digit's run, the run's place in the tile and in the output ---
        const uint32_t c0 = sCnt[0][tid], c1 = sCnt[1][tid], c2 = sCnt[2][tid], c3 = sCnt[3][tid];
        const uint32_t total = c0 + c1 + c2 + c3;
        sCnt[0][tid] = 0u; sCnt[1][tid] = c0; sCnt[2][tid] = c0 + c1; sCnt[3][tid] = c0 + c1 + c2;
        const uint32_t inc_t = rdg_wave_scan_incl(total);
        if (lane == 63) sWtot[wv] = inc_t;
        __syncthreads();
        uint32_t woff_t = 0;
        for (int k = 0; k < wv; ++k) woff_t += sWtot[k];
        const uint32_t local_base = woff_t + inc_t - total;
        sLocal[tid] = local_base;
        sGlob[tid] = cursor - (long long)local_base;
        cursor += total;
        __syncthreads();
        // ---- re-order the tile in LDS, write it out as runs ----------------------------------------------------------
#pragma unroll
        for (int i = 0; i < RDG_RS_ITEMS; ++i) {
            const int loc = wv * (RDG_RS_ITEMS * 64) + i * 64 + lane;
            if (loc < n_tile) {
                const uint32_t d = (uint32_t)(key[i] >> shift) & 255u;
                const uint32_t pos = sLocal[d] + sCnt[wv][d] + rank[i];
                sKey[pos] = key[i]; sVal[pos] = val[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RDG_RS_ITEMS; ++j) {
            const int p = j * RDG_RS_THREADS + tid;
            if (p < n_tile) {
                const KeyT k = sKey[p];
                const uint32_t d = (uint32_t)(k >> shift) & 255u;
                const long long o = sGlob[d] + p;
                keys_out[o] = k; vals_out[o] = sVal[p];
            }
        }
        __syncthreads();                         // sKey / sVal / sGlob are rewritten by the next tile
    }
}

// Sorts n = *n_dev (<= capacity) pairs on key bits [begin_bit, end_bit) (whole 8-bit passes from begin_bit); ping-pongs
// between (a) and (b), *result_in_b tells where the result is.  tmp: rdg_radix_sort_tmp_bytes(capacity).
template <typename KeyT>
int rdg_launch_radix_sort(KeyT* keys_a, KeyT* keys_b, uint32_t* vals_a, uint32_t* vals_b, int64_t capacity,
                        const int32_t* n_dev, int begin_bit, int end_bit, void* tmp, int* result_in_b, hipStream_t s) {
    const int npass = (end_bit - begin_bit + 7) / 8;
    if (result_in_b) *result_in_b = npass & 1;
    if (npass <= 0) return 0;
    const int nseg = rdg_rs_nseg(capacity);
    const bool small = rdg_rs_items(capacity) == 4;
    uint32_t* table = (uint32_t*)tmp;
    uint32_t* totals = (uint32_t*)((char*)tmp + rdg_align_up((size_t)256 * (size_t)nseg * 4, 256));   // right behind THIS sort's table
    KeyT* kin = keys_a; KeyT* kout = keys_b;
    uint32_t* vin = vals_a; uint32_t* vout = vals_b;
    for (int p = 0; p < npass; ++p) {
        const int shift = begin_bit + 8 * p;
        if (small) {
            hipLaunchKernelGGL((rdg_rs_count_kernel<KeyT, 4>), dim3(nseg), dim3(RDG_RS_THREADS), 0, s, kin, (long long)capacity,
                               n_dev, shift, nseg, table);
            hipLaunchKernelGGL(rdg_rs_scan_kernel, dim3(256), dim3(1024), 0, s, table, nseg, totals, (long long)capacity, n_dev);
            hipLaunchKernelGGL((rdg_rs_scatter_kernel<KeyT, 4>), dim3(nseg), dim3(RDG_RS_THREADS), 0, s, kin, vin, kout, vout,
                               (long long)capacity, n_dev, shift, nseg, table, totals);
        } else {
            hipLaunchKernelGGL((rdg_rs_count_kernel<KeyT, 16>), dim3(nseg), dim3(RDG_RS_THREADS), 0, s, kin, (long long)capacity,
                               n_dev, shift, nseg, table);
            hipLaunchKernelGGL(rdg_rs_scan_kernel, dim3(256), dim3(1024), 0, s, table, nseg, totals, (long long)capacity, n_dev);
            hipLaunchKernelGGL((rdg_rs_scatter_kernel<KeyT, 16>), dim3(nseg), dim3(RDG_RS_THREADS), 0, s, kin, vin, kout, vout,
                               (long long)capacity, n_dev, shift, nseg, table, totals);
        }
        KeyT* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }
    return rdg_check_hip(hipGetLastError(), "sort launch");
}

template int rdg_launch_radix_sort<uint32_t>(uint32_t*, uint32_t*, uint32_t*, uint32_t*, int64_t, const int32_t*, int, int,
                                           void*, int*, hipStream_t);
template int rdg_launch_radix_sort<uint64_t>(uint64_t*, uint64_t*, uint32_t*, uint32_t*, int64_t, const int32_t*, int, int,
                                           void*, int*, hipStream_t);
