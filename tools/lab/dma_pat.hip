// microbenchmark 2: LDS-DMA of GEMM-like operand tiles: ROWS rows x SEG bytes per stage at a row stride of LD bytes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned int u32;
// a stage = 32 KB per WG per step: ROWS = 32768 / SEG rows.  8 waves x 4 DMAs x 1 KB.
template <int SEG, int D>
__global__ __launch_bounds__(512, 4) void k_pat(const char *__restrict__ src, int ld, int steps_per_block, int blocks, int share, u32 *__restrict__ out, int lag)
{
    __shared__ __attribute__((aligned(1024))) char lds[65536];
    constexpr int ROWS = 32768 / SEG, LPR = SEG / 16, RPI = 64 / LPR;        // lanes per row, rows per instruction
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int par = 0;
    if (lag && ((blockIdx.x >> 3) % share) != 0) { const unsigned long long t0 = __builtin_amdgcn_s_memtime(); while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)lag) {} }
    for (int b = 0; b < blocks; b++) {
        // row block of this WG: siblings (share consecutive WGs) read the same rows
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const size_t blk = ((size_t)(xcd * ((int)(gridDim.x >> 3) / share) + slot / share) * blocks + b);
        const char *base = src + blk * (size_t)ROWS * ld;
        for (int s = 0; s < steps_per_block; s++) {
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const int row = (d * 8 + wv) * RPI + lane / LPR;
                const char *p = base + (size_t)row * ld + (size_t)s * SEG + (lane % LPR) * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                                 (__attribute__((address_space(3))) void *)(lds + par * 32768 + (d * 8 + wv) * 1024), 16, 0, 0);
            }
            par ^= 1;
            if (D == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (D == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (D == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    out[(size_t)blockIdx.x * 512 + tid] = reinterpret_cast<u32 *>(lds)[tid];
}
template <int SEG, int D> void run(const char *buf, int ld, int share, u32 *out, int lag = 0, int grid = 512, int hog = 0)
{
    const int  steps = ld / SEG, rows = 32768 / SEG;
    // total rows available: 4 GB / ld
    const size_t total_rows = ((size_t)3 << 30) / ld;
    int blocks = (int)(total_rows / rows / (grid / share)); blocks = blocks * 512 / grid > 128 ? 128 : blocks * 512 / grid; 
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        hipFuncSetAttribute((const void *)k_pat<SEG, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipLaunchKernelGGL((k_pat<SEG, D>), dim3(grid), dim3(512), hog, 0, buf, ld, steps, blocks, share, out, lag);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)grid * blocks * steps * 32768.0;
    printf("grid %d hog %d lag %5d D %d SEG %4d B  ld %5d  share %2d  blocks %3d: %7.3f ms  %6.2f TB/s into LDS  (%5.2f TB/s distinct)\n", grid, hog, lag, D, SEG, ld, share, blocks, ms, bytes / ms / 1e9,
           bytes / share / ms / 1e9);
}
int main()
{
    const size_t big = (size_t)3 << 30;
    char *buf; CK(hipMalloc(&buf, big + (1 << 20))); CK(hipMemset(buf, 1, big));
    u32 *out; CK(hipMalloc(&out, 512 * 512 * 4));
    // row pitch sweep (do equal offsets in every row camp on a subset of the L2 channels?): 128-B segments, 8 siblings, 3 stages in flight
    for (int ld : {1536, 1664, 1792, 2048, 2176, 3072, 3200, 6144, 6272, 128})
        run<128, 3>(buf, ld, 8, out, 0, 512, 0);
    for (int ld : {1536, 1664, 6144, 6272}) run<128, 3>(buf, ld, 4, out, 0, 512, 0);
    for (int ld : {1536, 1664, 6144, 6272}) run<128, 3>(buf, ld, 16, out, 0, 512, 0);
    for (int ld : {1536, 1664, 256}) run<256, 3>(buf, ld, 8, out, 0, 256, 65536);
    for (int ld : {1536, 1664}) run<64, 3>(buf, ld, 8, out, 0, 512, 0);
    return 0;
}
