// Probe: cost of a grid-wide barrier (agent-scope release / acquire around an atomic counter) inside one launch,
// against the cost of a kernel boundary.   build: hipcc -O3 --offload-arch=gfx950 grid_barrier_probe.hip -o /tmp/gbp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void grid_barrier(unsigned int* counter, unsigned int target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) chain_kernel(unsigned int* counter, float* buf, int nbar, int n) {
    // every stage: each workgroup writes its slice, barrier, reads the NEXT workgroup's slice (cross-XCD visibility check)
    float v = (float)blockIdx.x;
    for (int s = 0; s < nbar; ++s) {
        for (int i = threadIdx.x; i < n; i += 256) buf[(size_t)blockIdx.x * n + i] = v + s;
        grid_barrier(counter, (unsigned int)(s + 1) * gridDim.x);
        const int nb = (blockIdx.x + 1) % gridDim.x;
        float t = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) t += __builtin_nontemporal_load(&buf[(size_t)nb * n + i]);
        v = t / n - s - nb + blockIdx.x;          // stays blockIdx.x when the neighbour's writes were seen
        grid_barrier(counter + 32, (unsigned int)(s + 1) * gridDim.x);
    }
    if (threadIdx.x == 0) buf[(size_t)gridDim.x * n + blockIdx.x] = v;
}
__global__ void tiny_kernel(float* buf) { if (threadIdx.x == 0 && blockIdx.x == 0) buf[0] += 1.0f; }

int main() {
    const int n = 1024;
    for (int grid : {8, 32, 64, 128, 224}) {
        unsigned int* counter; float* buf;
        hipMalloc(&counter, 256); hipMalloc(&buf, (size_t)(grid * n + grid) * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int nbar : {1, 5, 21}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(counter, 0, 256);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                hipLaunchKernelGGL(chain_kernel, dim3(grid), dim3(256), 0, 0, counter, buf, nbar, n);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            float* h = (float*)malloc(grid * 4);
            hipMemcpy(h, buf + (size_t)grid * n, grid * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int b = 0; b < grid; ++b) bad += h[b] != (float)b;
            printf("grid %3d  stages %2d (2 barriers each): %.2f us  -> %.2f us per barrier pair   wrong workgroups: %d\n", grid, nbar,
                   best * 1e3f, best * 1e3f / nbar, bad);
            free(h);
        }
        // reference: 10 dependent tiny launches
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, 0, buf);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("   10 dependent tiny launches: %.2f us each\n", ms * 1e2f);
        hipFree(counter); hipFree(buf);
    }
    return 0;
}
