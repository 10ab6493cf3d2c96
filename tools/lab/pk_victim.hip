// Which instruction class of a wave goes wrong when ANOTHER wave on the same SIMD runs dense MFMA (tools/lab/lds_culprit.hip mfma), and does it
// take another PROCESS or just another kernel?  Every wave repeats one deterministic computation `reps` times and compares each result with the
// first one in-kernel; a mismatch is reported with the lane and the repetition.
// usage: pk_victim <class> [seconds = 20] [same_process_mfma = 0|1]
//   pkfma    v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 chains on registers only
//   fma32    the same arithmetic with scalar v_fma_f32 (no packed instructions)
//   fma64    v_fma_f64 chains
//   lds64    ds_write_b64 -> ds_read_b64 round trips (no arithmetic)
//   lds32    ds_write_b32 -> ds_read_b32 round trips
//   dft      the log-mel kernel's stage B shape: ds_read_b64 of a table row + packed complex multiply-accumulate
// same_process_mfma = 1: this process also runs the MFMA kernel, on a second stream (one process, two kernels on the same SIMDs)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(2))) float f2;
struct Report { unsigned wg, lane, rep, kind; float got, want; };
constexpr int MAXREP = 1024;

__global__ __launch_bounds__(256, 3) void k_mfma(float *__restrict__ sink, int iters)
{
    __shared__ __attribute__((aligned(1024))) _Float16 smem[3 * 2 * 64 * 64];
    const int lane = threadIdx.x & 63;
    smem[threadIdx.x] = (_Float16)1.0f;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    h8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(0.01f * (lane + j)); b[j] = (_Float16)(0.02f * (lane - j)); }
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int k = 0; k < 8; k++) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (c[0] == 12345.678f) sink[threadIdx.x] = c[0] + (float)smem[lane];
}

template <int CLS>
__global__ __launch_bounds__(256) void k_victim(Report *rep, int *nrep, int reps, float seed)
{
    __shared__ f2 tab[4][25 * 16];
    __shared__ f2 w25[25];
    __shared__ float tab32[4][512];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 4 * 400; i += 256) (&tab[0][0])[i] = f2{__sinf(0.37f * i + seed), __cosf(0.11f * i + seed)};
    for (int i = tid; i < 25; i += 256) w25[i] = f2{__cosf(0.2513274f * i), -__sinf(0.2513274f * i)};
    __syncthreads();
    float first = 0.f;
    for (int r = 0; r < reps; r++) {
        float res;
        if (CLS == 0 || CLS == 1) {
            f2 x = {0.5f + 0.001f * lane + seed, 0.25f - 0.002f * lane};
            f2 acc = {0.f, 0.f};
            asm volatile("" : "+v"(x), "+v"(acc));
#pragma unroll
            for (int k = 0; k < 64; k++) {
                if (CLS == 0) {
                    const f2 m = {1.0009765625f, 0.9990234375f};
                    acc = __builtin_elementwise_fma(x, m, acc);             // v_pk_fma_f32
                    x = x * m + f2{1e-3f, -1e-3f};                         // v_pk_mul_f32 + v_pk_add_f32 (contraction is off)
                } else {
                    acc[0] = fmaf(x[0], 1.0009765625f, acc[0]); acc[1] = fmaf(x[1], 0.9990234375f, acc[1]);
                    x[0] = x[0] * 1.0009765625f + 1e-3f; x[1] = x[1] * 0.9990234375f - 1e-3f;
                }
            }
            res = acc[0] + acc[1];
        } else if (CLS == 2) {
            double x = 0.5 + 0.001 * lane + seed, acc = 0.0;
            asm volatile("" : "+v"(x), "+v"(acc));
#pragma unroll
            for (int k = 0; k < 64; k++) { acc = fma(x, 1.0009765625, acc); x = fma(x, 0.9990234375, 1e-3); }
            res = (float)acc;
        } else if (CLS == 3) {
            float acc = 0.f;
#pragma unroll 4
            for (int k = 0; k < 25; k++) {
                f2 v = {seed + 1.0f * lane + k, 2.0f * lane - k};
                tab[wv][k * 16 + (lane & 15)] = v;                        // 4 quarters write the same 16 cells ...
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                tab[wv][((k + 7) % 25) * 16 + ((lane + 48) & 15)] = v;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                const f2 y = tab[wv][k * 16 + (lane & 15)];               // ... and read them back (the last writer's value)
                acc += y[0] - y[1];
            }
            res = acc;
        } else if (CLS == 4) {
            float acc = 0.f;
#pragma unroll 4
            for (int k = 0; k < 25; k++) {
                tab32[wv][(k * 64 + lane) & 511] = seed + lane + k;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                acc += tab32[wv][(k * 64 + (lane ^ 1)) & 511];
            }
            res = acc;
        } else if (CLS == 6) {
            // complex multiply-accumulate on registers only: the packed instructions with op_sel / neg modifiers the compiler forms for (a + ib)(c + id)
            f2 y = {0.5f + 0.001f * lane + seed, 0.25f - 0.002f * lane}, w = {0.9990482f, -0.0436194f};
            float re = 0.f, im = 0.f;
            asm volatile("" : "+v"(y), "+v"(w));
#pragma unroll
            for (int n2 = 0; n2 < 50; n2++) {
                re += y[0] * w[0] - y[1] * w[1]; im += y[0] * w[1] + y[1] * w[0];
                const f2 t = {w[0] * 0.9990482f - w[1] * -0.0436194f, w[0] * -0.0436194f + w[1] * 0.9990482f};
                w = t;
            }
            res = re * re + im * im;
        } else if (CLS >= 27 && CLS <= 31) {
            // packed 16-bit forms (two halves of ONE 32-bit register): is the op_sel:[0,1] effect a property of the 64-bit packed fp32 path only?
            unsigned x = 0x3C003C00u + (unsigned)(lane & 7) + (((unsigned)(lane & 3)) << 16), w = 0x3BFF3C01u, acc = 0u;     // (1 + small, 1 + small), (1 - 2^-11.., 1 + 2^-10)
            asm volatile("" : "+v"(x), "+v"(w));
#pragma unroll
            for (int k = 0; k < 64; k++) {
                unsigned d;
                if (CLS == 27)      asm volatile("v_pk_mul_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(x), "v"(w));
                else if (CLS == 28) asm volatile("v_pk_add_f16 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(x), "v"(w));
                else if (CLS == 29) asm volatile("v_pk_fma_f16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(x), "v"(w), "v"(acc));
                else if (CLS == 30) asm volatile("v_pk_mul_lo_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(x), "v"(w));
                else                asm volatile("v_pk_max_i16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(x), "v"(w));
                acc = (acc << 1 | acc >> 31) ^ d;
                asm volatile("" : "+v"(acc));
                x = (CLS == 28) ? (d & 0x3FFF3FFFu) | 0x3C003C00u : d;
                if (CLS >= 30) x = (d & 0x0FFF0FFFu) | 0x00010001u;
            }
            res = __uint_as_float((acc & 0x007FFFFFu) | 0x3F800000u);
        } else if (CLS >= 8) {
            // ONE packed instruction form per class, written in assembly: which encoding is it?
            f2 x = {0.5f + 0.001f * lane + seed, 0.25f - 0.002f * lane}, w = {1.0009765625f, 0.9990234375f}, acc = {0.f, 0.f};
            asm volatile("" : "+v"(x), "+v"(w), "+v"(acc));
#pragma unroll
            for (int k = 0; k < 64; k++) {
                f2 d;
                if (CLS == 8)       asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(x), "v"(w));
                else if (CLS == 9)  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(x), "v"(w));
                else if (CLS == 10) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(w));
                else if (CLS == 11) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(x), "v"(w));
                else if (CLS == 12) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(x), "v"(w), "v"(acc));
                else if (CLS == 13) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(w));
                else if (CLS == 14) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(x), "v"(w));
                else if (CLS == 15) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(d) : "v"(x), "v"(w));
                else if (CLS == 16) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(x), "v"(w));               // src0 swapped
                else if (CLS == 17) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(x), "v"(w));                                // src1 hi to both
                else if (CLS == 18) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(x), "v"(w));               // src0 hi to both, src1 swapped
                else if (CLS == 19) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]" : "=v"(d) : "v"(x), "v"(w));  // src0 swapped add
                else if (CLS == 20) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=v"(d) : "v"(x), "v"(w));               // both swapped
                else if (CLS == 32 || CLS == 33) {                                    // v_pk_mov_b32 with the other selections, then a plain multiply
                    f2 t;
                    if (CLS == 32) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(t) : "v"(w), "v"(x));
                    else           asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(t) : "v"(w), "v"(x));
                    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(t));
                    d[0] = d[0] * 0.5f + 0.25f; d[1] = d[1] * 0.5f + 0.25f;
                }
                else if (CLS == 21) { f2 t; asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]" : "=v"(t) : "v"(w)); asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(t)); }   // swap by v_pk_mov_b32, plain multiply
                else if (CLS == 23) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(x), "v"(w), "v"(acc));    // src2 swapped
                else if (CLS == 24) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(x), "v"(w), "v"(acc));    // src0 swapped
                else if (CLS == 25) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(x), "v"(w));                                    // hi lane: src0 low, src1 high
                else if (CLS == 26) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(x), "v"(w), "v"(acc));    // src0 and src2 swapped
                else if (CLS == 22) { f2 t = {w[1], w[0]}; asm volatile("" : "+v"(t)); asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(t)); }       // swap by two v_mov_b32, plain multiply
                acc[0] += d[0]; acc[1] -= d[1];
                asm volatile("" : "+v"(acc));
                x = d;
            }
            res = acc[0] + acc[1];
        } else if (CLS == 7) {
            // divergent ds_read_b64 (every lane its own table entry, as w25[m]) summed with scalar adds: no packed arithmetic
            float acc = 0.f;
            int m = 0;
            const int k2 = (lane >> 4) + 4 * (r & 3);
            for (int n2 = 0; n2 < 25; n2++) {
                const f2 w = w25[m];
                float a0 = w[0], a1 = w[1];
                asm volatile("" : "+v"(a0), "+v"(a1));
                acc += a0; acc -= a1;
                m += k2; if (m >= 25) m -= 25;
            }
            res = (r & 3) == 0 ? acc : first;
        } else {
            // stage B of k_logmel_frames: X[k1 + 16 k2] = sum_{n2 < 25} Y[n2][k1] W25^{n2 k2}
            float total = 0.f;
            for (int k = lane; k < 201; k += 64) {
                const int k1 = k & 15, k2 = k >> 4;
                float re = 0.f, im = 0.f;
                int m = 0;
                for (int n2 = 0; n2 < 25; n2++) {
                    const f2 y = tab[wv][n2 * 16 + k1];
                    const f2 w = w25[m];
                    re += y[0] * w[0] - y[1] * w[1]; im += y[0] * w[1] + y[1] * w[0];
                    m += k2; if (m >= 25) m -= 25;
                }
                total += re * re + im * im;
            }
            res = total;
        }
        if (r == 0) first = res;
        else if (res != first) {
            const int s = atomicAdd(nrep, 1);
            if (s < MAXREP) rep[s] = Report{blockIdx.x, (unsigned)tid, (unsigned)r, (unsigned)CLS, res, first};
        }
    }
}

int main(int argc, char **argv)
{
    const char *names[] = {"pkfma", "fma32", "fma64", "lds64", "lds32", "dft", "cmul", "ldsdiv", "mul_sel01_hi10", "mul_hi10", "add_neg", "add_sel01_hi10", "fma_sel", "mul_plain", "mul_sel10", "add_neglo", "mul_src0swap", "mul_src1hi", "mul_cmul2", "add_src0swap", "mul_bothswap", "pkmov_swap", "vmov_swap", "fma_src2swap", "fma_src0swap", "mul_hi01", "fma_src02swap", "pk_mul_f16_sel01", "pk_add_f16_sel01", "pk_fma_f16_sel01", "pk_mul_lo_u16_sel01", "pk_max_i16_sel01", "pkmov_sel01", "pkmov_sel11"};
    int cls = -1;
    for (int i = 0; i < 34; i++) if (argc > 1 && !strcmp(argv[1], names[i])) cls = i;
    if (cls < 0) { fprintf(stderr, "usage: pk_victim pkfma|fma32|fma64|lds64|lds32|dft [seconds] [same_process_mfma]\n"); return 2; }
    const double seconds = argc > 2 ? atof(argv[2]) : 20.0;
    const int same = argc > 3 ? atoi(argv[3]) : 0;
    Report *rep; int *nrep; float *sink;
    CK(hipMalloc(&rep, sizeof(Report) * MAXREP)); CK(hipMalloc(&nrep, 4)); CK(hipMemset(nrep, 0, 4)); CK(hipMalloc(&sink, 4096));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0; int total = 0; int quarter[4] = {0, 0, 0, 0};
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int r = 0; r < 10; r++) {
            const dim3 g(1024), b(256);
            const float seed = 0.001f * (launches % 7);
            if (same) hipLaunchKernelGGL(k_mfma, dim3(768), dim3(256), 0, sb, sink, 200);
            switch (cls) {
                case 0: hipLaunchKernelGGL(k_victim<0>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 1: hipLaunchKernelGGL(k_victim<1>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 2: hipLaunchKernelGGL(k_victim<2>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 3: hipLaunchKernelGGL(k_victim<3>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 4: hipLaunchKernelGGL(k_victim<4>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 6: hipLaunchKernelGGL(k_victim<6>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 7: hipLaunchKernelGGL(k_victim<7>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 8: hipLaunchKernelGGL(k_victim<8>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 9: hipLaunchKernelGGL(k_victim<9>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 10: hipLaunchKernelGGL(k_victim<10>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 11: hipLaunchKernelGGL(k_victim<11>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 12: hipLaunchKernelGGL(k_victim<12>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 13: hipLaunchKernelGGL(k_victim<13>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 14: hipLaunchKernelGGL(k_victim<14>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 15: hipLaunchKernelGGL(k_victim<15>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 16: hipLaunchKernelGGL(k_victim<16>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 17: hipLaunchKernelGGL(k_victim<17>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 18: hipLaunchKernelGGL(k_victim<18>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 19: hipLaunchKernelGGL(k_victim<19>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 20: hipLaunchKernelGGL(k_victim<20>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 21: hipLaunchKernelGGL(k_victim<21>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 22: hipLaunchKernelGGL(k_victim<22>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 23: hipLaunchKernelGGL(k_victim<23>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 24: hipLaunchKernelGGL(k_victim<24>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 25: hipLaunchKernelGGL(k_victim<25>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 26: hipLaunchKernelGGL(k_victim<26>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 27: hipLaunchKernelGGL(k_victim<27>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 28: hipLaunchKernelGGL(k_victim<28>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 29: hipLaunchKernelGGL(k_victim<29>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 30: hipLaunchKernelGGL(k_victim<30>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 31: hipLaunchKernelGGL(k_victim<31>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 32: hipLaunchKernelGGL(k_victim<32>, g, b, 0, sa, rep, nrep, 40, seed); break;
                case 33: hipLaunchKernelGGL(k_victim<33>, g, b, 0, sa, rep, nrep, 40, seed); break;
                default: hipLaunchKernelGGL(k_victim<5>, g, b, 0, sa, rep, nrep, 20, seed); break;
            }
            launches++;
        }
        CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
        int n; CK(hipMemcpy(&n, nrep, 4, hipMemcpyDeviceToHost));
        if (n) {
            std::vector<Report> h(n < MAXREP ? n : MAXREP);
            CK(hipMemcpy(h.data(), rep, sizeof(Report) * h.size(), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < h.size(); i++) {
                quarter[(h[i].lane & 63) >> 4]++;
                if (total + (int)i < 12) printf("  %s wg %u wave %u lane %u repetition %u got %.9g want %.9g\n", names[cls], h[i].wg, h[i].lane >> 6, h[i].lane & 63, h[i].rep, h[i].got, h[i].want);
            }
            total += n;
            CK(hipMemset(nrep, 0, 4));
        }
    }
    printf("victim %s%s: %ld launches x 1024 workgroups x 4 waves in %.1f s: %d wrong results; by lane quarter (0-15, 16-31, 32-47, 48-63): %d %d %d %d\n", names[cls],
           same ? " + MFMA kernel of the SAME process on a second stream" : "", launches, seconds, total, quarter[0], quarter[1], quarter[2], quarter[3]);
    return 0;
}
