// microbenchmark: operand delivery rate into a CU (LDS-DMA vs VGPR loads; L2-resident vs streaming)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned int u32;
typedef __attribute__((ext_vector_type(4))) u32 u32x4;

// MODE 0: global_load_lds b128; MODE 1: global_load_dwordx4 to VGPR (+xor sink); DEPTH = loads per wave between waits
template <int MODE, int DEPTH>
__global__ __launch_bounds__(512, 4) void k_bw(const char *__restrict__ src, size_t wg_stride, size_t span, int iters, u32 *__restrict__ out, unsigned long long *ticks)
{
    __shared__ __attribute__((aligned(1024))) char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const char *base = src + (size_t)blockIdx.x * wg_stride;
    u32x4 sink = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    size_t off = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const char *p = base + off + (size_t)(d * 8 + wv) * 1024 + lane * 16;
            if (MODE == 0)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                                 (__attribute__((address_space(3))) void *)(lds + ((d * 8 + wv) & 63) * 1024), 16, 0, 0);
            else {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(p);
                sink ^= v;
            }
        }
        off += (size_t)DEPTH * 8192; if (off >= span) off = 0;
        if (MODE == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) sink[0] = reinterpret_cast<u32 *>(lds)[tid];
    out[(size_t)blockIdx.x * 512 + tid] = sink[0] ^ sink[1] ^ sink[2] ^ sink[3];
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE, int DEPTH>
void run(const char *name, const char *buf, size_t wg_stride, size_t span, int grid, int iters, u32 *out, unsigned long long *ticks)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_bw<MODE, DEPTH>), dim3(grid), dim3(512), 0, 0, buf, wg_stride, span, iters, out, ticks);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_bw<MODE, DEPTH>), dim3(grid), dim3(512), 0, 0, buf, wg_stride, span, iters, out, ticks);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid);
    CK(hipMemcpy(h.data(), ticks, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += (double)v; avg /= grid;
    const double bytes = (double)grid * iters * DEPTH * 8192.0;
    // ticks are 100 MHz on some parts, shader clock on others: print both views
    printf("%-34s depth %d grid %4d: %8.3f ms  %7.2f TB/s  %6.1f B/us/WG-pair... ticks/iter %.1f\n", name, DEPTH, grid, ms, bytes / ms / 1e9,
           bytes / ms / 1e3 / 256.0, avg / iters);
}

int main()
{
    const size_t big = (size_t)4 << 30;
    char *buf; CK(hipMalloc(&buf, big)); CK(hipMemset(buf, 1, big));
    u32 *out; CK(hipMalloc(&out, 4096 * 512 * 4));
    unsigned long long *ticks; CK(hipMalloc(&ticks, 4096 * 8));
    const int grid = 512;                 // 2 WGs per CU
    // L2-resident: every WG cycles over the same 2 MB (shared), 64 KB apart per WG
    run<0, 4>("dma  shared 2MB (L2)", buf, 0, (size_t)2 << 20, grid, 2000, out, ticks);
    run<0, 8>("dma  shared 2MB (L2)", buf, 0, (size_t)2 << 20, grid, 1000, out, ticks);
    run<1, 4>("vgpr shared 2MB (L2)", buf, 0, (size_t)2 << 20, grid, 2000, out, ticks);
    run<1, 8>("vgpr shared 2MB (L2)", buf, 0, (size_t)2 << 20, grid, 1000, out, ticks);
    // MALL-resident: shared 128 MB
    run<0, 4>("dma  shared 128MB (MALL)", buf, 4096, (size_t)128 << 20, grid, 2000, out, ticks);
    run<1, 4>("vgpr shared 128MB (MALL)", buf, 4096, (size_t)128 << 20, grid, 2000, out, ticks);
    // streaming: private 8 MB per WG (4 GB total)
    run<0, 4>("dma  private 8MB/WG (HBM)", buf, (size_t)8 << 20, (size_t)8 << 20, grid, 250, out, ticks);
    run<0, 8>("dma  private 8MB/WG (HBM)", buf, (size_t)8 << 20, (size_t)8 << 20, grid, 125, out, ticks);
    run<1, 4>("vgpr private 8MB/WG (HBM)", buf, (size_t)8 << 20, (size_t)8 << 20, grid, 250, out, ticks);
    run<1, 8>("vgpr private 8MB/WG (HBM)", buf, (size_t)8 << 20, (size_t)8 << 20, grid, 125, out, ticks);
    // tiny: 32 KB per WG private (L1/L2)
    run<0, 4>("dma  private 32KB (TCP/L2)", buf, 65536, 32768, grid, 2000, out, ticks);
    run<1, 4>("vgpr private 32KB (TCP/L2)", buf, 65536, 32768, grid, 2000, out, ticks);
    return 0;
}
