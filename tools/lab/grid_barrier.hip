// Cost of a grid-wide barrier inside one persistent launch on MI355X (256 workgroups, one per CU, 8 XCDs whose L2s are not coherent with
// each other): the building block of a fused decoder-layer kernel (phases separated by barriers instead of kernel boundaries).
//   variant 0: one counter, agent-scope release add + acquire spin by thread 0 of every workgroup (what cooperative groups does)
//   variant 1: the same with the counter in fine-grained... (not built: hipMalloc memory is what the engine uses)
//   variant 2: two-level: an XCD-local counter (8 of them, 32 arrivals each), then the 8 leaders on the global one
// and, for comparison, the same number of empty kernel launches back to back on one stream.
// Also checks that data written before the barrier by one workgroup is read correctly after it by a workgroup of ANOTHER XCD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned *ctr, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // every wave: no stale L1 / L2 lines behind the barrier
}

// lower bound: relaxed agent-scope atomics, no cache maintenance at all (NOT a correct barrier for data: what the counter traffic alone costs)
__device__ __forceinline__ void grid_barrier_relaxed(unsigned *ctr, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {}
    }
    __syncthreads();
}

__device__ __forceinline__ void grid_barrier2(unsigned *xc, unsigned *gc, unsigned epoch, unsigned per_xcd, unsigned xcd)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(&xc[xcd * 32], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (old == epoch * per_xcd - 1) __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // the XCD's last arrival
        while (__hip_atomic_load(gc, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch * 8u) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__global__ __launch_bounds__(256) void k_barriers(unsigned *ctr, unsigned *xc, int n, int variant, float *buf, int *bad, unsigned long long *cycles)
{
    const unsigned G = gridDim.x, b = blockIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 1; i <= n; i++) {
        // every workgroup publishes a value, the barrier, then reads its neighbour's (another XCD: blocks go to XCDs round-robin)
        if (threadIdx.x < 64) buf[b * 64 + threadIdx.x] = (float)(i * 1000 + (int)b);
        if (variant == 0) grid_barrier(ctr, (unsigned)i * 2u * G - G);
        else if (variant == 3) grid_barrier_relaxed(ctr, (unsigned)i * 2u * G - G);
        else grid_barrier2(xc, ctr, (unsigned)(2 * i - 1), G / 8, b & 7);
        const unsigned nb = (b + 1) % G;
        if (threadIdx.x < 64 && buf[nb * 64 + threadIdx.x] != (float)(i * 1000 + (int)nb)) atomicAdd(bad, 1);
        if (variant == 0) grid_barrier(ctr, (unsigned)i * 2u * G);      // (nobody overwrites before everybody has read)
        else if (variant == 3) grid_barrier_relaxed(ctr, (unsigned)i * 2u * G);
        else grid_barrier2(xc, ctr, (unsigned)(2 * i), G / 8, b & 7);
    }
    if (b == 0 && threadIdx.x == 0) *cycles = __builtin_amdgcn_s_memrealtime() - t0;   // 100 MHz
}
__global__ void k_empty(int *x) { if (x && threadIdx.x == 9999) *x = 1; }

int main()
{
    unsigned *ctr, *xc; float *buf; int *bad; unsigned long long *cyc;
    hipMalloc(&ctr, 256); hipMalloc(&xc, 8 * 32 * 4); hipMalloc(&buf, 256 * 64 * 4); hipMalloc(&bad, 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant : {0, 2, 3, 0, 2, 3}) {
        const int n = 500;
        hipMemset(ctr, 0, 256); hipMemset(xc, 0, 8 * 32 * 4); hipMemset(bad, 0, 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_barriers, dim3(256), dim3(256), 0, 0, ctr, xc, n, variant, buf, bad, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        int hb; unsigned long long hc; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        printf("variant %d: %d x 2 barriers in %.3f ms = %.2f us per barrier (in-kernel clock: %.2f us); stale reads: %d\n", variant, n, ms,
               ms * 1e3 / (2 * n), (double)hc / 100.0 / (2 * n), hb);
    }
    for (int threads : {64, 256}) {
        const int n = 1000;
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_empty, dim3(256), dim3(threads), 0, 0, (int *)nullptr);
        hipEventRecord(e0);
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_empty, dim3(256), dim3(threads), 0, 0, (int *)nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("empty kernels (256 x %d threads) back to back: %.2f us per launch\n", threads, ms * 1e3 / n);
    }
    return 0;
}
