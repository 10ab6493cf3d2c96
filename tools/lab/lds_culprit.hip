// One instruction class in a tight loop, to be run BESIDE a libpce process whose log-mel transform is being repeat-tested
// (tools/lab/race_matrix.sh culprit_<mode>): which instruction of k_attention_lean16 -- the one kernel whose presence on the same compute
// unit makes another process's LDS-resident transforms glitch (profiles/r06/multiprocess_glitch.txt) -- does it?
// usage: lds_culprit <mode> [seconds = 30]
//   bperm    ds_bpermute_b32 only (the wave shuffles of the softmax maxima)
//   dma      buffer_load_dwordx4 ... lds into a 48 KiB ring + counted waits + barriers (the K / V^T staging)
//   read128  ds_read_b128 of its own LDS
//   mfma     v_mfma_f32_16x16x32_f16 chains
//   barrier  s_barrier only
//   exp      v_exp_f32 chains
//   all      everything above in one loop
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) float f4;

template <int MODE>
__global__ __launch_bounds__(256, 3) void k_culprit(const _Float16 *__restrict__ src, float *__restrict__ sink, int iters)
{
    __shared__ __attribute__((aligned(1024))) _Float16 smem[3 * 2 * 64 * 64];       // 48 KiB, as the attention kernel's ring
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 3 * 2 * 64 * 64 / 2; i += 256) reinterpret_cast<unsigned *>(smem)[i] = 0x3c003c00u;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(src), 0, 1 << 22, 0x00020000);
    float acc = (float)lane;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    h8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(0.01f * (lane + j)); b[j] = (_Float16)(0.02f * (lane - j)); }
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || MODE == 6) {
#pragma unroll
            for (int k = 0; k < 8; k++) acc = fmaxf(acc, __int_as_float(__builtin_amdgcn_ds_bpermute(((lane ^ (16 << (k & 1))) << 2), __float_as_int(acc + 1.0f))));
        }
        if (MODE == 1 || MODE == 6) {
            _Float16 *slot = smem + (it % 3) * (2 * 64 * 64);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(slot + (wv + 4 * i) * 8 * 64), 16, lane * 16, ((it * 8 + i) & 1023) * 1024, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(slot + 64 * 64 + (wv + 4 * i) * 8 * 64), 16, lane * 16, ((it * 8 + 4 + i) & 1023) * 1024, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0x0F70 | 0);
            __builtin_amdgcn_s_barrier();
        }
        if (MODE == 2 || MODE == 6) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const h8 v = *reinterpret_cast<const h8 *>(&smem[((it + k) % 3) * 8192 + ((lane * 8 + k * 512) & 8191)]);
                a[0] += v[0]; b[1] += v[7];
            }
        }
        if (MODE == 3 || MODE == 6) {
#pragma unroll
            for (int k = 0; k < 8; k++) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
        }
        if (MODE == 4) __builtin_amdgcn_s_barrier();
        if (MODE == 5 || MODE == 6) {
#pragma unroll
            for (int k = 0; k < 8; k++) acc = __builtin_amdgcn_exp2f(acc * 0.001f) + 1.0f;
        }
    }
    if (MODE == 1 || MODE == 6) __builtin_amdgcn_s_waitcnt(0x0F70 | 0);
    if (acc + c[0] + (float)a[0] + (float)b[1] == 12345.678f) sink[tid] = acc;
}

int main(int argc, char **argv)
{
    const char *names[] = {"bperm", "dma", "read128", "mfma", "barrier", "exp", "all"};
    int mode = -1;
    for (int i = 0; i < 7; i++) if (argc > 1 && !strcmp(argv[1], names[i])) mode = i;
    if (mode < 0) { fprintf(stderr, "usage: lds_culprit bperm|dma|read128|mfma|barrier|exp|all [seconds]\n"); return 2; }
    const double seconds = argc > 2 ? atof(argv[2]) : 30.0;
    _Float16 *src; float *sink;
    CK(hipMalloc(&src, 1 << 22)); CK(hipMemset(src, 0, 1 << 22)); CK(hipMalloc(&sink, 4096));
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int r = 0; r < 20; r++) {
            const dim3 g(768), b(256);
            switch (mode) {
                case 0: hipLaunchKernelGGL(k_culprit<0>, g, b, 0, 0, src, sink, 400); break;
                case 1: hipLaunchKernelGGL(k_culprit<1>, g, b, 0, 0, src, sink, 100); break;
                case 2: hipLaunchKernelGGL(k_culprit<2>, g, b, 0, 0, src, sink, 400); break;
                case 3: hipLaunchKernelGGL(k_culprit<3>, g, b, 0, 0, src, sink, 200); break;
                case 4: hipLaunchKernelGGL(k_culprit<4>, g, b, 0, 0, src, sink, 400); break;
                case 5: hipLaunchKernelGGL(k_culprit<5>, g, b, 0, 0, src, sink, 400); break;
                default: hipLaunchKernelGGL(k_culprit<6>, g, b, 0, 0, src, sink, 60); break;
            }
        }
        CK(hipDeviceSynchronize());
        launches += 20;
    }
    printf("culprit %s: %ld launches in %.1f s (%.1f us each)\n", names[mode], launches, seconds, seconds / launches * 1e6);
    return 0;
}
