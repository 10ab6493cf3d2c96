// microbenchmark 3: what does the C write stream of the persistent GEMM cost by itself?  Every workgroup (one per CU, 8 waves) writes
// 256 x 256 bf16 tiles of C [M][N] with the product kernel's store pattern (16 buffer_store_dwordx4 per wave and tile: 16 rows x 64 B
// per instruction), nothing else; then the same with a compute-like delay between the tiles, and as plain full-row streaming stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k_store(unsigned short *C, int M, int N, int delay)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wr = wv >> 2, wc = wv & 3, fr = lane & 15, fq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)((unsigned)M * (unsigned)N * 2u), 0x00020000);
    const int tiles_n = N / 256, tiles = (M / 256) * tiles_n;
    const u32x4 v = {(unsigned)tid, 2u, 3u, 4u};
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int m0 = (t / tiles_n) * 256, n0 = (t % tiles_n) * 256;
        if (MODE == 0) {
            const int voff = (fr * N + fq * 8) * 2;
#pragma unroll
            for (int hA = 0; hA < 2; hA++)
#pragma unroll
                for (int hB = 0; hB < 2; hB++)
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, ((m0 + hA * 128 + wr * 64 + i * 16) * N + n0 + hB * 128 + wc * 32) * 2, 0);
        } else {                   // full 512-byte row segments: a wave-instruction writes 2 rows x 512 B
            const int voff = ((lane >> 5) * N + (lane & 31) * 8) * 2;
#pragma unroll
            for (int i = 0; i < 16; i++)
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, ((m0 + wv * 32 + i * 2) * N + n0) * 2, 0);
        }
        if (delay) { const unsigned long long t0 = __builtin_amdgcn_s_memtime(); while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)delay) __builtin_amdgcn_s_sleep(4); }
    }
}
int main(int argc, char **argv)
{
    const int M = 384000;
    for (int N : {768, 3072}) {
        unsigned short *C; CK(hipMalloc(&C, (size_t)M * N * 2));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto run = [&](const char *what, auto launch) {
            for (int i = 0; i < 3; i++) launch();
            CK(hipEventRecord(e0)); for (int i = 0; i < 10; i++) launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
            printf("N %4d %-44s %.3f ms  %.2f TB/s\n", N, what, ms, (double)M * N * 2 / ms / 1e9);
        };
        run("GEMM store pattern, back to back", [&] { hipLaunchKernelGGL((k_store<0>), dim3(256), dim3(512), 0, 0, C, M, N, 0); });
        run("GEMM store pattern, 2 waves of workgroups/CU", [&] { hipLaunchKernelGGL((k_store<0>), dim3(512), dim3(512), 0, 0, C, M, N, 0); });
        run("full rows (2 x 512 B per instruction)", [&] { hipLaunchKernelGGL((k_store<1>), dim3(256), dim3(512), 0, 0, C, M, N, 0); });
        for (int d : {5000, 10000, 20000, 40000}) {
            char b[96]; snprintf(b, sizeof b, "GEMM store pattern + %d idle cycles per tile", d);
            run(b, [&] { hipLaunchKernelGGL((k_store<0>), dim3(256), dim3(512), 0, 0, C, M, N, d); });
        }
        CK(hipFree(C));
    }
    return 0;
}
