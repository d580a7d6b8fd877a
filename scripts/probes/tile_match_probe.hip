// Probe for the binning count pass (rdg_tile_bucket_kernel<0>): the tile-matching step groups the 64 instance slots of a
// wave by their counter index with one ballot round per index bit (14 rounds at 1080p: ~240 mostly scalar instructions
// per 64 slots; PMC says the atomics are few, 0.33 M requests per 10 M instances, and 2-4 steps in flight change nothing).
// This probe times three forms of the step on a synthetic Z-ordered instance stream and checks that they hand out the
// same groups:
//   A  all index bits (what the kernel does)
//   B  only the bits that differ inside the wave (wave OR / AND by DPP, then a scalar loop over the set bits)
//   C  as B, without the atomics and rank stores (what the matching alone costs)
// build: hipcc -O3 --offload-arch=gfx950 tile_match_probe.hip -o /tmp/tmp_probe ; run: /tmp/tmp_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ uint32_t zidx(uint32_t x, uint32_t y) {
    x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu; x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    y = (y | (y << 8)) & 0x00FF00FFu; y = (y | (y << 4)) & 0x0F0F0F0Fu; y = (y | (y << 2)) & 0x33333333u;
    y = (y | (y << 1)) & 0x55555555u;
    return x | (y << 1);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_u(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, true);
}
// OR over the 64 lanes, result in lane 63 (prefix form: lanes shifted in from outside a row read 0)
__device__ __forceinline__ uint32_t wave_or_to63(uint32_t x) {
    x |= dpp_u<0x111>(x); x |= dpp_u<0x112>(x); x |= dpp_u<0x114>(x); x |= dpp_u<0x118>(x);
    x |= dpp_u<0x142, 0xa>(x); x |= dpp_u<0x143, 0xc>(x);
    return x;
}

// MODE 0 = A, 1 = B, 2 = C
template <int MODE>
__global__ void __launch_bounds__(256) match_kernel(const uint32_t* __restrict__ z_in, long long n, int zbits,
                                                    uint32_t* __restrict__ tile_cnt, uint32_t* __restrict__ rank,
                                                    unsigned long long* __restrict__ groups) {
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    const long long steps = (n + 63) / 64;
    // a wave takes 10 consecutive steps at a time (as a wave of the real kernel walks its own Gaussians' slots)
    for (long long s0 = wave * 10; s0 < steps; s0 += nwaves * 10) {
        for (long long s = s0; s < s0 + 10 && s < steps; ++s) {
            const long long k = s * 64 + lane;
            const bool act = k < n;
            const uint32_t z = act ? z_in[k] : 0u;
            unsigned long long m = __ballot(act);
            if (MODE == 0) {
                for (int bit = 0; bit < zbits; ++bit) {
                    const bool bset = (z >> bit) & 1u;
                    const unsigned long long bal = __ballot(act && bset);
                    m &= bset ? bal : ~bal;
                }
            } else {
                const uint32_t o = wave_or_to63(act ? z : 0u), a = wave_or_to63(act ? ~z : 0u);   // AND = ~OR(~z)
                uint32_t vary = (uint32_t)__builtin_amdgcn_readlane((int)o, 63) &
                                (uint32_t)__builtin_amdgcn_readlane((int)a, 63);                    // set somewhere, clear somewhere
                while (vary) {
                    const int bit = __builtin_ctz(vary);
                    vary &= vary - 1u;
                    const bool bset = (z >> bit) & 1u;
                    const unsigned long long bal = __ballot(act && bset);
                    m &= bset ? bal : ~bal;
                }
            }
            const uint32_t below = (uint32_t)__popcll(m & lt_mask);
            if (MODE == 2) {
                if (act) groups[k] = m;
                continue;
            }
            uint32_t base = 0;
            if (act && below == 0) base = atomicAdd(&tile_cnt[z], (uint32_t)__popcll(m));
            base = (uint32_t)__shfl((int)base, act ? __ffsll((long long)m) - 1 : 0);
            if (act) { rank[k] = base + below; groups[k] = m; }
        }
    }
}

int main() {
    const int gx = 120, gy = 68, zbits = 14;
    const long long n = 10 * 1000 * 1000;
    // synthetic stream: runs of 6-20 slots (one Gaussian's rectangle, row by row) around a tile position that drifts slowly
    // (a Z-ordered cloud: consecutive Gaussians project to neighbouring tiles)
    std::vector<uint32_t> z(n);
    uint32_t seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    long long k = 0;
    int cx = 60, cy = 34;
    while (k < n) {
        cx += (int)(rnd() % 3) - 1; cy += (int)(rnd() % 3) - 1;
        if (rnd() % 64 == 0) { cx = rnd() % gx; cy = rnd() % gy; }
        cx = cx < 0 ? 0 : (cx > gx - 4 ? gx - 4 : cx); cy = cy < 0 ? 0 : (cy > gy - 5 ? gy - 5 : cy);
        const int w = 2 + rnd() % 3, h = 2 + rnd() % 4;
        for (int yy = 0; yy < h && k < n; ++yy)
            for (int xx = 0; xx < w && k < n; ++xx) {
                uint32_t x = cx + xx, y = cy + yy, zz;
                x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu; x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
                y = (y | (y << 8)) & 0x00FF00FFu; y = (y | (y << 4)) & 0x0F0F0F0Fu; y = (y | (y << 2)) & 0x33333333u; y = (y | (y << 1)) & 0x55555555u;
                zz = x | (y << 1);
                z[k++] = zz;
            }
    }
    uint32_t *dz, *dcnt, *drank; unsigned long long *dgA, *dgB;
    hipMalloc(&dz, n * 4); hipMalloc(&dcnt, (1u << zbits) * 4); hipMalloc(&drank, n * 4);
    hipMalloc(&dgA, n * 8); hipMalloc(&dgB, n * 8);
    hipMemcpy(dz, z.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 3907;                                 // the bench frame's grid (1 M Gaussians / 256)
    auto run = [&](int mode, unsigned long long* g) {
        float best = 1e9f;
        for (int rep = 0; rep < 10; ++rep) {
            hipMemset(dcnt, 0, (1u << zbits) * 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(match_kernel<0>, dim3(grid), dim3(256), 0, 0, dz, n, zbits, dcnt, drank, g);
            if (mode == 1) hipLaunchKernelGGL(match_kernel<1>, dim3(grid), dim3(256), 0, 0, dz, n, zbits, dcnt, drank, g);
            if (mode == 2) hipLaunchKernelGGL(match_kernel<2>, dim3(grid), dim3(256), 0, 0, dz, n, zbits, dcnt, drank, g);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        return best;
    };
    const float tA = run(0, dgA), tB = run(1, dgB);
    std::vector<unsigned long long> hA(n), hB(n);
    hipMemcpy(hA.data(), dgA, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hB.data(), dgB, n * 8, hipMemcpyDeviceToHost);
    long long diff = 0;
    for (long long i = 0; i < n; ++i) diff += hA[i] != hB[i];
    const float tC = run(2, dgB);
    printf("10 M slots, %d workgroups:  A all %d bits %.1f us   B varying bits only %.1f us   C matching alone (B) %.1f us   "
           "groups that differ A vs B: %lld\n", grid, zbits, tA * 1e3f, tB * 1e3f, tC * 1e3f, diff);
    return diff != 0;
}
