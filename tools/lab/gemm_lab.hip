// GEMM laboratory: C[M][N] (bf16) = A[M][K] * B[N][K]^T, variants of the 128x256 tile kernel with ablation switches
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int BM = 128, BN = 256, BK = 32, THREADS = 512;
constexpr int STAGE = (BM + BN) * BK;
constexpr int TLD = 128 + 4;
__device__ __forceinline__ int swz32(int row, int chunk) { return chunk ^ ((row >> 2) & 3); }
template <int ROWS>
__device__ __forceinline__ void stage_rows32(const bf16 *__restrict__ G, int64_t ld, int row0, int k0, bf16 *lds_tile, int wv, int lane)
{
#pragma unroll
    for (int i = 0; i < ROWS / 128; i++) {
        const int r0 = (wv + 8 * i) * 16;
        const int row = r0 + (lane >> 2);
        const int c = swz32(row, lane & 3);
        const bf16 *src = G + (int64_t)(row0 + row) * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds_tile + r0 * BK), 16, 0, 0);
    }
}
// ABL bit 0: skip MFMA; bit 1: skip ds_reads (fragments loaded once); bit 2: skip global loads after the prologue; bit 3: skip the epilogue store
template <int ABL>
__global__ __launch_bounds__(THREADS, 4) void k_wide(const bf16 *__restrict__ A, const bf16 *__restrict__ B, int M, int N, int K, bf16 *__restrict__ C, int sn_tiles, int sm_tiles, unsigned long long *stamps = nullptr)
{
    unsigned long long st0 = 0, sr0 = 0;
    if (stamps) { st0 = __builtin_amdgcn_s_memtime(); sr0 = __builtin_amdgcn_s_memrealtime(); }
    constexpr int SMEM_ELEMS = (2 * STAGE * 2 > BM * TLD * 4 ? 2 * STAGE : BM * TLD * 2);
    __shared__ __attribute__((aligned(1024))) bf16 smem[SMEM_ELEMS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 2, wc = wv & 3;
    const int tiles_n = (int)gridDim.x, tiles_m_pad = (int)gridDim.y;
    int lin = (int)blockIdx.y * tiles_n + (int)blockIdx.x;
    const int total = tiles_n * tiles_m_pad;
    if ((total & 7) == 0) lin = (lin & 7) * (total >> 3) + (lin >> 3);
    const int per = sm_tiles * sn_tiles, sup = lin / per, r = lin - sup * per;
    const int n_sn = tiles_n / sn_tiles;
    const int tm = (sup / n_sn) * sm_tiles + r / sn_tiles, tn = (sup % n_sn) * sn_tiles + r % sn_tiles;
    const int m0 = tm * BM, n0 = tn * BN;
    if (m0 >= M) return;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    const int nk = K / BK;
    stage_rows32<BM>(A, K, m0, 0, smem, wv, lane);
    stage_rows32<BN>(B, K, n0, 0, smem + BM * BK, wv, lane);
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { a[i] = bf16x8{}; b[i] = bf16x8{}; }
    for (int kt = 0; kt < nk; kt++) {
        const bf16 *sA = smem + (kt & 1) * STAGE, *sB = sA + BM * BK;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk && !(ABL & 4)) {
            bf16 *nA = smem + ((kt + 1) & 1) * STAGE;
            stage_rows32<BM>(A, K, m0, (kt + 1) * BK, nA, wv, lane);
            stage_rows32<BN>(B, K, n0, (kt + 1) * BK, nA + BM * BK, wv, lane);
        }
        if (!(ABL & 2) || kt == 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = wr * 64 + i * 16 + fr;
                a[i] = *reinterpret_cast<const bf16x8 *>(&sA[row * BK + swz32(row, fq) * 8]);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = wc * 64 + j * 16 + fr;
                b[j] = *reinterpret_cast<const bf16x8 *>(&sB[row * BK + swz32(row, fq) * 8]);
            }
        }
        if (!(ABL & 1)) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i][i][0] += (float)a[i][0] + (float)b[i][0];
        }
    }
    float *tile = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int p = 0; p < 2; p++) {
        __syncthreads();
        if ((wc >> 1) == p) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int r4 = 0; r4 < 4; r4++)
                        tile[(wr * 64 + i * 16 + fq * 4 + r4) * TLD + (wc & 1) * 64 + j * 16 + fr] = acc[i][j][r4];
        }
        __syncthreads();
        const int cx = (tid & 15) * 8, ry = tid >> 4;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = ry + 32 * q;
            const float4 v0 = *reinterpret_cast<const float4 *>(&tile[row * TLD + cx]);
            const float4 v1 = *reinterpret_cast<const float4 *>(&tile[row * TLD + cx + 4]);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = (bf16)v[e];
            if (!(ABL & 8) || v[0] == 12345.678f)
                __builtin_nontemporal_store(o, reinterpret_cast<bf16x8 *>(C + (int64_t)(m0 + row) * N + n0 + p * 128 + cx));
        }
    }
    if (stamps && threadIdx.x == 0) { unsigned long long *o = stamps + 4 * (blockIdx.y * gridDim.x + blockIdx.x); o[0] = st0; o[1] = sr0; o[2] = __builtin_amdgcn_s_memtime(); o[3] = __builtin_amdgcn_s_memrealtime(); }
}

#include "gemm_lab_v2.inc"

template <class F> float time_ms(F f, int reps)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int M = 192000, N = argc > 1 ? atoi(argv[1]) : 3072, K = argc > 2 ? atoi(argv[2]) : 768;
    bf16 *A, *B, *C, *C2;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2)); CK(hipMalloc(&C2, (size_t)M * N * 2));
    {
        std::vector<bf16> h((size_t)M * K);
        unsigned s = 12345;
        for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (bf16)(((int)(s >> 20) % 17 - 8) * 0.125f); }
        CK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        std::vector<bf16> w((size_t)N * K);
        for (auto &v : w) { s = s * 1664525u + 1013904223u; v = (bf16)(((int)(s >> 20) % 13 - 6) * 0.0625f); }
        CK(hipMemcpy(B, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    }
    const double tf = 2.0 * M * N * K / 1e9;
    const int wt = N / BN; int wsn = 1;
    for (int cand : {4, 3, 2}) if (wt % cand == 0) { wsn = cand; break; }
    const int wsm = 16;
    dim3 grid(wt, ((M + BM * wsm - 1) / (BM * wsm)) * wsm);
#define RUN(ABL) { float ms = time_ms([&] { hipLaunchKernelGGL((k_wide<ABL>), grid, dim3(THREADS), 0, 0, A, B, M, N, K, C, wsn, wsm); }, 5); \
                   printf("wide abl %2d: %.3f ms  %.0f TFLOP/s\n", ABL, ms, tf / ms); }
    RUN(0)
    {
        unsigned long long *stamps; const size_t nb = (size_t)grid.x * grid.y;
        CK(hipMalloc(&stamps, nb * 32));
        auto clock_of = [&](const char *name, auto launch) {
            for (int i = 0; i < 400; i++) launch();
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(nb * 4);
            CK(hipMemcpy(h.data(), stamps, nb * 32, hipMemcpyDeviceToHost));
            std::vector<double> f;
            for (size_t b = 0; b < nb; b++) if (h[4 * b + 3] > h[4 * b + 1]) f.push_back((double)(h[4 * b + 2] - h[4 * b]) / (double)(h[4 * b + 3] - h[4 * b + 1]) * 100.0);
            std::sort(f.begin(), f.end());
            printf("in-kernel clock, %s: median %.0f MHz (p10 %.0f, p90 %.0f) over %zu workgroups\n", name, f[f.size() / 2], f[f.size() / 10], f[f.size() * 9 / 10], f.size());
        };
        clock_of("wide full", [&] { hipLaunchKernelGGL((k_wide<0>), grid, dim3(THREADS), 0, 0, A, B, M, N, K, C, wsn, wsm, stamps); });
        clock_of("wide no-MFMA (loads, reads, stores)", [&] { hipLaunchKernelGGL((k_wide<1>), grid, dim3(THREADS), 0, 0, A, B, M, N, K, C, wsn, wsm, stamps); });
        clock_of("wide MFMA only", [&] { hipLaunchKernelGGL((k_wide<14>), grid, dim3(THREADS), 0, 0, A, B, M, N, K, C, wsn, wsm, stamps); });
        float ms = time_ms([&] { hipLaunchKernelGGL((k_wide<0>), grid, dim3(THREADS), 0, 0, A, B, M, N, K, C, wsn, wsm); }, 50);
        printf("wide full after 1 s of load: %.3f ms  %.0f TFLOP/s\n", ms, tf / ms);
        ms = time_ms([&] { hipLaunchKernelGGL((k_wide<14>), grid, dim3(THREADS), 0, 0, A, B, M, N, K, C, wsn, wsm); }, 50);
        printf("wide MFMA only, sustained: %.3f ms  %.0f TFLOP/s\n", ms, tf / ms);
        ms = time_ms([&] { hipLaunchKernelGGL((k_wide<11>), grid, dim3(THREADS), 0, 0, A, B, M, N, K, C, wsn, wsm); }, 50);
        printf("wide loads only, sustained: %.3f ms\n", ms);
    }
    run_v2(A, B, C, C2, M, N, K, tf);
    return 0;
}
