#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512, 4) void k(unsigned *out)
{
    __shared__ char lds[65536];
    unsigned x, h;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
    lds[threadIdx.x] = 1;
    // linger so that the grid is co-resident
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 200000) {}
    if (threadIdx.x == 0) { out[2 * (blockIdx.y * gridDim.x + blockIdx.x)] = x; out[2 * (blockIdx.y * gridDim.x + blockIdx.x) + 1] = h; }
}
int main()
{
    unsigned *d; hipMalloc(&d, 8 * 4096);
    for (int shape = 0; shape < 2; shape++) {
        dim3 grid = shape == 0 ? dim3(1024) : dim3(12, 96);
        hipLaunchKernelGGL(k, grid, dim3(512), 0, 0, d);
        hipDeviceSynchronize();
        std::vector<unsigned> h(2 * 1152);
        hipMemcpy(h.data(), d, 8 * 1152, hipMemcpyDeviceToHost);
        printf("grid shape %d: xcc of linear block 0..63:\n", shape);
        for (int i = 0; i < 64; i++) printf("%u ", h[2 * i] & 0xf);
        printf("\n cu/se (hw_id>>8 &0xf cu, >>13 &7 se) of blocks 0..15: ");
        for (int i = 0; i < 16; i++) printf("[x%u se%u cu%u] ", h[2 * i] & 0xf, (h[2 * i + 1] >> 13) & 7, (h[2 * i + 1] >> 8) & 0xf);
        int ok = 0; for (int i = 0; i < 1024; i++) ok += ((h[2 * i] & 0xf) == (unsigned)(i & 7));
        printf("\n blocks with xcc == lin %% 8: %d of 1024\n", ok);
    }
    return 0;
}
