// microbenchmark: what a read-only int16 reduction can stream from HBM on this part, at the sizes the framing kernels run at
// (82 MB = the C2 batch, inside the 256 MB Infinity Cache when launches repeat; 400 MB = the C4 shard per GPU; 1.6 GB).
// Three shapes of the same sum:  A  one 32 KB chunk per 256-thread workgroup, all eight 16-byte loads of a lane issued before the
// first is used (k_energy's shape);  B  the same with non-temporal loads;  C  a persistent grid (CUs x 8 workgroups) striding over
// the buffer, eight loads in flight per lane.  Reported: GB/s of HIP-event time over 20 launches after 3 warm-ups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef int i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int fold(i4 v) { return v.x ^ v.y ^ v.z ^ v.w; }
template <bool NT>
__global__ __launch_bounds__(256) void k_chunk(const i4 *__restrict__ src, size_t n16, int *__restrict__ out)
{
    const size_t base = (size_t)blockIdx.x * 2048 + threadIdx.x;
    i4 v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const size_t p = base + (size_t)i * 256;
        v[i] = p < n16 ? (NT ? __builtin_nontemporal_load(src + p) : src[p]) : (i4){0, 0, 0, 0};
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += fold(v[i]);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0 && s == 0x12345678) out[blockIdx.x] = s;          // (never true: keeps the loads)
}
// k_energy's packed route on the chunk shape; MASK bits switch parts on: 1 sum of squares (64-bit), 2 sum, 4 wrapped squares,
// 8 extrema, 16 loud count (per-half compares), 32 loud count (packed: saturating subtract, sign bits, packed add)
template <int MASK>
__global__ __launch_bounds__(256) void k_compute(const i4 *__restrict__ src, size_t n16, int thr, unsigned long long *__restrict__ out)
{
    typedef short s2 __attribute__((ext_vector_type(2)));
    typedef unsigned short u2 __attribute__((ext_vector_type(2)));
    const size_t base = (size_t)blockIdx.x * 2048 + threadIdx.x;
    i4 v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { const size_t p = base + (size_t)i * 256; v[i] = p < n16 ? __builtin_nontemporal_load(src + p) : (i4){0, 0, 0, 0}; }
    unsigned long long s_sq = 0; int s_sum = 0, s_wrap = 0, n_loud = 0;
    const s2 ones = {1, 1};
    s2 pmax = {-32768, -32768}, pmin = {32767, 32767};
    const s2 thr2 = {(short)thr, (short)thr};
    u2 loud2 = {0, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int words[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const s2 xv = __builtin_bit_cast(s2, words[k]);
            if (MASK & 1) s_sq += (unsigned long long)(unsigned int)__builtin_amdgcn_sdot2(xv, xv, 0, false);
            if (MASK & 2) s_sum = __builtin_amdgcn_sdot2(xv, ones, s_sum, false);
            if (MASK & 4) s_wrap = __builtin_amdgcn_sdot2(xv * xv, ones, s_wrap, false);
            if (MASK & 8) { pmax = __builtin_elementwise_max(pmax, xv); pmin = __builtin_elementwise_min(pmin, xv); }
            if (MASK & 48) {
                const s2 ab = __builtin_elementwise_max(xv, (s2){0, 0} - xv);
                if (MASK & 16) n_loud += ((int)ab.x > thr) + ((int)ab.y > thr);
                if (MASK & 32) {
                    const s2 d = __builtin_elementwise_sub_sat(thr2, ab);          // negative exactly where ab > thr (|-32768| = -32768 saturates to +)
                    loud2 += __builtin_bit_cast(u2, d) >> (u2){15, 15};
                }
            }
        }
    }
    unsigned long long r = s_sq + (unsigned)s_sum + (unsigned)s_wrap + (unsigned)n_loud + (unsigned)(pmax.x + pmax.y + pmin.x + pmin.y) + loud2.x + loud2.y;
    for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + (blockIdx.x & 1023), r);
}

__global__ __launch_bounds__(256) void k_persist(const i4 *__restrict__ src, size_t n16, int *__restrict__ out)
{
    int s = 0;
    const size_t stride = (size_t)gridDim.x * 2048;
    for (size_t base = (size_t)blockIdx.x * 2048 + threadIdx.x; base < n16; base += stride) {
        i4 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { const size_t p = base + (size_t)i * 256; v[i] = p < n16 ? __builtin_nontemporal_load(src + p) : (i4){0, 0, 0, 0}; }
#pragma unroll
        for (int i = 0; i < 8; i++) s += fold(v[i]);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0 && s == 0x12345678) out[blockIdx.x] = s;
}

template <class F> static double time_it(F launch)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; i++) launch();
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 20.0;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t sizes[] = {81920000, 400000000, 1600000000};
    char *buf; CK(hipMalloc(&buf, sizes[2] + 4096)); CK(hipMemset(buf, 1, sizes[2] + 4096));
    int *out; CK(hipMalloc(&out, 1 << 22));
    for (size_t bytes : sizes) {
        const size_t n16 = bytes / 16;
        const unsigned grid = (unsigned)((n16 + 2047) / 2048);
        const double a = time_it([&] { hipLaunchKernelGGL(k_chunk<false>, dim3(grid), dim3(256), 0, 0, (const i4 *)buf, n16, out); });
        const double b = time_it([&] { hipLaunchKernelGGL(k_chunk<true>, dim3(grid), dim3(256), 0, 0, (const i4 *)buf, n16, out); });
        const double c = time_it([&] { hipLaunchKernelGGL(k_persist, dim3((unsigned)cus * 8), dim3(256), 0, 0, (const i4 *)buf, n16, out); });
        unsigned long long *o64 = reinterpret_cast<unsigned long long *>(out);
#define CV(M) { const double t = time_it([&] { hipLaunchKernelGGL(k_compute<M>, dim3(grid), dim3(256), 0, 0, (const i4 *)buf, n16, 500, o64); }); \
                printf("      compute mask %2d: %6.1f us = %5.0f GB/s\n", M, t * 1e3, bytes / t / 1e6); }
        CV(0) CV(1) CV(2) CV(4) CV(8) CV(16) CV(32) CV(31) CV(47) CV(46) CV(15)
        printf("%7.1f MB   chunk %6.1f us = %5.0f GB/s   chunk nt %6.1f us = %5.0f GB/s   persistent nt %6.1f us = %5.0f GB/s\n", bytes / 1e6,
               a * 1e3, bytes / a / 1e6, b * 1e3, bytes / b / 1e6, c * 1e3, bytes / c / 1e6);
    }
    return 0;
}
