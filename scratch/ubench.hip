#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// per-SIMD throughput and latency of fp64 ops: W waves per SIMD (block = 256*W/... ) measured with wall clock
template <int OP, int CHAINS>
__global__ void k(double *out, int iters, double seed)
{
    double a[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x * 1e-3;
    const double m = 1.0000001, b = 1e-9;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (OP == 0) a[c] = fma(a[c], m, b);
            else if (OP == 1) a[c] = a[c] * m;
            else if (OP == 2) a[c] = a[c] + b;
            else if (OP == 3) a[c] = __builtin_amdgcn_rcp(a[c]);
            else if (OP == 4) a[c] = (a[c] > 1.5) ? a[c] * m : a[c] + b;     // cmp + cndmask*2 + ...
            else if (OP == 5) { float f = (float)a[c]; f = fmaf(f, 1.0000001f, 1e-9f); a[c] = f; }
            else if (OP == 6) a[c] = a[c] / (m + c);
            else if (OP == 7) { float f = __builtin_bit_cast(float, (int)__double2loint(a[c])); f = fmaf(f, 1.0000001f, 1e-9f); a[c] = __hiloint2double(__double2hiint(a[c]), __builtin_bit_cast(int, f)); }
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP, int CHAINS> void run(const char *name, int wavesPerSimd)
{
    double *d; hipMalloc(&d, 8 * 256 * 4096);
    const int iters = 65536;
    int blocks = 256 * wavesPerSimd;      // 256 CUs, block = 256 threads = 4 waves = 1 per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, CHAINS>), dim3(blocks), dim3(256), 0, 0, d, 16, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, CHAINS>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)iters * CHAINS * wavesPerSimd;
    double cycles = ms * 1e-3 * 2.4e9;
    printf("%-28s chains %2d waves/SIMD %d : %.2f cycles per wave-instruction per SIMD (%.3f ms)\n", name, CHAINS, wavesPerSimd, cycles / instr_per_simd, ms);
    hipFree(d);
}
int main()
{
    { double *d; hipMalloc(&d, 8 * 256 * 4096); for (int r = 0; r < 40; r++) hipLaunchKernelGGL((k<0, 8>), dim3(1024), dim3(256), 0, 0, d, 65536, 1.0); hipDeviceSynchronize(); hipFree(d); }
    run<7, 8>("fma_f32 (calibration)", 2);
    run<0, 1>("fma_f64 dependent", 1);
    run<0, 8>("fma_f64", 1);
    run<0, 8>("fma_f64", 2);
    run<0, 8>("fma_f64", 4);
    run<1, 8>("mul_f64", 2);
    run<2, 8>("add_f64", 2);
    run<3, 1>("rcp_f64 dependent", 1);
    run<3, 8>("rcp_f64", 2);
    run<4, 8>("cmp+select+mul+add f64", 2);
    run<5, 8>("cvt f64->f32, fmaf, cvt back", 2);
    run<6, 4>("div f64", 2);
    return 0;
}
