// GEMM laboratory, round 3: the experiment round 2 left open -- a 256 x 256 persistent tile computed by FOUR waves (one per SIMD,
// 128 x 128 outputs each, the accumulators in the 256 AGPRs) instead of eight (two per SIMD, 128 x 64 each):
//   * no two waves compete for a SIMD's matrix pipe (the stamped K-steps of the 8-wave kernel showed the partner waves waiting 600 cycles
//     at each of the two barriers of a K-step for one another);
//   * 128 KB of fragment reads per K-step instead of 192 KB (a wave re-uses each A fragment against 8 B fragments, not 4);
//   * one barrier per K-step among four waves.
// What it gives up: nobody issues MFMAs while a wave is held at a DMA issue or a counted wait -- the partner wave did.
// C[M][N] (bf16) = A[M][K] B[N][K]^T, M any, N % 256 == 0, K % 64 == 0.  Same ring as pce_gemm256.inc: a K-step is four 16 KB half-tiles
// (A rows 0-127 | A rows 128-255 | B rows 0-127 | B rows 128-255), two K-steps of slots (128 KB), LDS-DMA through buffer resources,
// chunks XOR-swizzled, B rows permuted inside 32-row groups so that the swapped MFMA (D^T = B A^T) leaves a lane 8 consecutive columns.
// K-step s (parity p): phase 1 = 64 MFMAs on the fragments of (s, k 0..31) while those of (s, k 32..63) are read; ONE barrier (every
// wave is done reading step s, every wave's DMA of step s + 1 has landed); phase 2 = 64 MFMAs on (s, k 32..63) while the fragments of
// (s + 1, k 0..31) are read and the 16 DMA instructions per wave of step s + 2 are issued into the slots of step s.
// build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -o bin/gemm_w4 gemm_w4.hip        run: bin/gemm_w4 [N K M]
// (without the flag the allocator spills 316 B per lane -- 256 live accumulators leave the AGPR class no spare tuple -- and every scratch
//  reload is a vector-memory load in front of whose use the compiler waits vmcnt(0): the DMA queue drained twice per K-step, 310 TFLOP/s)
//
// RESULT (MI355X, uniform random operands; gpurun_out/r3k, copied to profiles/r03/gemm_w4.txt): correct, and SLOWER than the 8-wave kernel:
//     N 3072, K 768 (fc1):  0.993 ms = 912 TFLOP/s   (8 waves: 1 090-1 130 in this laboratory)
//     N 768, K 3072 (fc2):  1.806 ms = 1 004          (8 waves: 1 230)
//     N 768, K 768 (out):   0.535 ms = 847            (8 waves: ~1 000)
// Ablations on the fc1 shape: without the MFMAs 0.737 ms, without the DMA after the prologue 0.838, without fragment reads 0.963.
// The operand stream alone (64 KB per K-step at the ~25 B/clk a CU takes in through LDS-DMA) costs three quarters of the kernel, and a
// single wave per SIMD adds it to its MFMA time instead of hiding one behind the other: 0.74 + 0.36-0.42 ~ 0.99.  That IS what was given up:
// in the 8-wave kernel the partner wave issues MFMAs while a wave sits at a DMA issue.  Conclusion: at K-step 64 a 256 x 256 tile needs
// 32 B/clk/CU of operands against ~25 delivered -- the tile shape is operand-delivery bound at ~78 % of the MFMA rate even with perfect
// overlap, the 8-wave schedule realises 50 % of the peak, and neither wave count changes the bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int T = 256, BK = 64, THREADS = 256;
constexpr int HALF = 128 * BK;                                 // elements of a half-tile (16 KB)
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
__device__ __forceinline__ int perm32(int p) { return ((p >> 2) & 3) * 8 + (p >> 4) * 4 + (p & 3); }

struct Sched {
    int tiles_n, sm, sn, per_xcd, xcd, slot, wper, my_tiles;
    __device__ void init(int M, int N, int sm_, int sn_)
    {
        tiles_n = N / T; sm = sm_; sn = sn_;
        const int tiles_m = (M + T - 1) / T, tiles_m_pad = ((tiles_m + sm - 1) / sm) * sm;
        const int total = tiles_m_pad * tiles_n;
        per_xcd = (total + 7) >> 3; xcd = blockIdx.x & 7; slot = blockIdx.x >> 3; wper = gridDim.x >> 3;
        my_tiles = per_xcd > slot ? (per_xcd - slot + wper - 1) / wper : 0;
    }
    __device__ void tile(int it, int &m0, int &n0) const
    {
        const int lin = xcd * per_xcd + slot + it * wper;
        const int per = sm * sn, sup = lin / per, r = lin - sup * per;
        const int n_sn = tiles_n / sn;
        m0 = ((sup / n_sn) * sm + r / sn) * T; n0 = ((sup % n_sn) * sn + r % sn) * T;
    }
};
struct Cursor { int kofs, m0, n0; };
template <int N> __device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt(((N >> 4) << 14) | 0x0F70 | (N & 15)); }
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }

// ABL bit 0: no MFMA; bit 1: no fragment reads after the prologue; bit 2: no DMA after the prologue; bit 3: no stores
template <int ABL>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_w4(const bf16 *__restrict__ A, const bf16 *__restrict__ B, int M, int N, int K, bf16 *__restrict__ C, int sm, int sn)
{
    extern __shared__ __attribute__((aligned(1024))) bf16 dsm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 1, wc = wv & 1;
    const int fr = lane & 15, fq = lane >> 4;
    Sched S; S.init(M, N, sm, sn);
    const int nk = K / BK;
    if (S.my_tiles == 0) return;
    // staging: a half-tile = 16 wave-instructions of 8 rows x 128 B; a wave issues 4 of them per half-tile (LDS rows w*8 + 32 i)
    int voffA[4], voffB[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = (wv + 4 * i) * 8 + (lane >> 3), c8 = swz(row, lane & 7) * 8;
        const int prow = (row & ~31) + perm32(row & 31);
        voffA[i] = (row * K + c8) * 2; voffB[i] = (prow * K + c8) * 2;
    }
    // fragment addresses inside a half-tile: row block i adds i * 16 * BK (the swizzle term (row >> 1) & 7 is the same for row + 16 i)
    int ra[2], rb[2];
    {
        ra[0] = fr * BK + swz(fr, fq) * 8; ra[1] = fr * BK + swz(fr, 4 + fq) * 8;
        rb[0] = ra[0]; rb[1] = ra[1];
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(A), 0, (int)((unsigned)M * (unsigned)K * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(B), 0, (int)((unsigned)N * (unsigned)K * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)((unsigned)M * (unsigned)N * 2u), 0x00020000);
    const int last_tile = S.my_tiles - 1;
    auto set_tile = [&](Cursor &c, int it) { S.tile(it < last_tile ? it : last_tile, c.m0, c.n0); };
    auto advance = [&](Cursor &c, int &it, int &kt) { if (++kt == nk) { kt = 0; ++it; set_tile(c, it); } c.kofs = kt * BK; };
    // slot of (parity, kind): kind 0 = A rows 0-127, 1 = A rows 128-255, 2 = B rows 0-127, 3 = B rows 128-255
    auto slot_of = [&](int par, int kind) { return dsm + (kind * 2 + par) * HALF; };
    auto issue_half = [&](const Cursor &c, int kind, int par) {
        if (ABL & 4) return;
        const bool is_a = kind < 2;
        const unsigned soff = (unsigned)((is_a ? c.m0 : c.n0) + (kind & 1) * 128) * (unsigned)K * 2u + (unsigned)c.kofs * 2u;
        bf16 *s = slot_of(par, kind);
#pragma unroll
        for (int i = 0; i < 4; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(is_a ? rsA : rsB, (__attribute__((address_space(3))) void *)(s + (wv + 4 * i) * 8 * BK), 16,
                                                     is_a ? voffA[i] : voffB[i], (int)soff, 0, 0);
    };
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment registers: B is double buffered (every row group of a phase uses all eight B fragments), A is not: the A fragments of the NEXT
    // k half are read into a row pair's registers right after that pair's last MFMA of this half (96 VGPRs instead of 128: the first
    // version spilled, and a scratch reload is a vector-memory load -- the compiler then waits vmcnt(0) in front of its use, which drains
    // the DMA queue twice per K-step: 310 TFLOP/s)
    bf16x8 FA[8], FB[2][8];
    auto read_B = [&](int set, int kk, int par) {
        const bf16 *sb = slot_of(par, 2 + wc);
#pragma unroll
        for (int j = 0; j < 8; j++) FB[set][j] = *reinterpret_cast<const bf16x8 *>(sb + rb[kk] + j * 16 * BK);
    };
    auto read_A2 = [&](int i0, int kk, int par) {
        const bf16 *sa = slot_of(par, wr);
        FA[i0] = *reinterpret_cast<const bf16x8 *>(sa + ra[kk] + i0 * 16 * BK);
        FA[i0 + 1] = *reinterpret_cast<const bf16x8 *>(sa + ra[kk] + (i0 + 1) * 16 * BK);
    };
    auto mma_rows = [&](int set, int i0) {                      // row blocks i0, i0 + 1: 16 MFMAs
        if (ABL & 1) { acc[i0][0][0] += (float)FA[i0][0] + (float)FB[set][i0][0]; return; }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[set][j], FA[i0 + i], acc[i0 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int voffC = (fr * N + fq * 8) * 2;
    auto store_tile = [&](int m0, int n0) {
        if (ABL & 8) return;
        // lane: row fr of row block i; accumulator column blocks 2 jp, 2 jp + 1 hold the 8 consecutive columns fq * 8 .. + 7 of the 32-column group jp
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int jp = 0; jp < 4; jp++) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = (bf16)acc[i][2 * jp + (e >> 2)][e & 3];
                const unsigned soff = ((unsigned)(m0 + wr * 128 + i * 16) * (unsigned)N + (unsigned)(n0 + wc * 128 + jp * 32)) * 2u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsC, voffC, (int)soff, 0);
                asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");          // (store-data hazard: see pce_gemm256.inc)
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    // ---- prologue: all of K-step 0 and K-step 1; fragments (0, k 0..31)
    Cursor c0; c0.kofs = 0; set_tile(c0, 0);
    int it1 = 0, kt1 = 0; Cursor pc1 = c0; advance(pc1, it1, kt1);
    int it2 = it1, kt2 = kt1; Cursor pc2 = pc1; advance(pc2, it2, kt2);
    for (int kind = 0; kind < 4; kind++) issue_half(c0, kind, 0);
    for (int kind = 0; kind < 4; kind++) issue_half(pc1, kind, 1);
    wait_vm<16>();
    __builtin_amdgcn_s_barrier();
    read_B(0, 0, 0);
    read_A2(0, 0, 0); read_A2(2, 0, 0); read_A2(4, 0, 0); read_A2(6, 0, 0);
    wait_lgkm0();
    int c_m0, c_n0; S.tile(0, c_m0, c_n0);
    int par = 0;
    bool stores_pending = false;
    for (int it = 0; it < S.my_tiles; it++) {
        for (int kt = 0; kt < nk; kt++) {
            // phase 1: MFMAs on (s, k 0..31); B of (s, k 32..63) into the other set, A of (s, k 32..63) pair by pair behind its pair's MFMAs
            if (!(ABL & 2)) read_B(1, 1, par);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i0 = 0; i0 < 8; i0 += 2) {
                mma_rows(0, i0);
                if (!(ABL & 2)) read_A2(i0, 1, par);
                __builtin_amdgcn_sched_barrier(0);
            }
            // every DMA of step s + 1 (issued during phase 2 of step s - 1; a tile's 32 stores may sit behind them in the queue)
            if (stores_pending) wait_vm<32>(); else wait_vm<0>();
            stores_pending = false;
            wait_lgkm0();
            __builtin_amdgcn_s_barrier();
            // phase 2: MFMAs on (s, k 32..63); fragments of (s + 1, k 0..31); step s + 2 issued into the slots of step s
            if (!(ABL & 2)) read_B(0, 0, par ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i0 = 0; i0 < 8; i0 += 2) {
                issue_half(pc2, i0 >> 1, par);
                __builtin_amdgcn_sched_barrier(0);
                mma_rows(1, i0);
                if (!(ABL & 2)) read_A2(i0, 0, par ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            pc1 = pc2; it1 = it2; kt1 = kt2; advance(pc2, it2, kt2);
            if (kt == nk - 1) {
                store_tile(c_m0, c_n0);
                stores_pending = true;
#pragma unroll
                for (int i = 0; i < 8; i++)
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            wait_lgkm0();
            par ^= 1;
        }
        if (it + 1 < S.my_tiles) S.tile(it + 1, c_m0, c_n0);
    }
    wait_vm<0>();
}

template <class F> float time_ms(F f, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; i++) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}
__global__ void k_ref(const bf16 *A, const bf16 *B, int N, int K, const int *rows, int n_rows, float *out)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (n >= N) return;
    const bf16 *a = A + (size_t)rows[r] * K, *b = B + (size_t)n * K;
    float s = 0.f;
    for (int k = 0; k < K; k++) s += (float)a[k] * (float)b[k];
    out[(size_t)r * N + n] = s;
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 768, K = argc > 2 ? atoi(argv[2]) : 768, M = argc > 3 ? atoi(argv[3]) : 384000;
    bf16 *A, *B, *C;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
    {
        std::vector<bf16> h((size_t)M * K);
        unsigned s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 9) & 0x7fff) / 16384.0f - 1.0f; };
        for (auto &v : h) v = (bf16)rnd();
        CK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        std::vector<bf16> w((size_t)N * K);
        for (auto &v : w) v = (bf16)(rnd() * 0.125f);
        CK(hipMemcpy(B, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    }
    const double tf = 2.0 * M * N * K / 1e9;
    int n_cu = 256;
    { hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); n_cu = p.multiProcessorCount; }
    const int tiles_n = N / T;
    int sn = 1;
    for (int cand : {4, 3, 2}) if (tiles_n % cand == 0) { sn = cand; break; }
    const int lds = 8 * HALF * 2;
    printf("k_w4: M %d N %d K %d, %d CUs, sn %d, LDS %d\n", M, N, K, n_cu, sn, lds);
    CK(hipMemset(C, 0xff, (size_t)M * N * 2));
#define RUN(ABL) { CK(hipFuncSetAttribute((const void *)k_w4<ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        float ms = time_ms([&] { hipLaunchKernelGGL((k_w4<ABL>), dim3(n_cu), dim3(THREADS), lds, 0, A, B, M, N, K, C, 16, sn); }, 30); \
        CK(hipGetLastError()); printf("w4 abl %2d: %.3f ms  %.0f TFLOP/s\n", ABL, ms, tf / ms); fflush(stdout); }
    RUN(0)
    {
        std::vector<int> rows = {0, 1, 63, 64, 127, 128, 255, 256, 257, M / 2, M / 2 + 129, M - 257, M - 2, M - 1};
        unsigned s = 99; for (int i = 0; i < 18; i++) { s = s * 1664525u + 1013904223u; rows.push_back((int)(s % (unsigned)M)); }
        int *d_rows; float *d_ref; CK(hipMalloc(&d_rows, rows.size() * 4)); CK(hipMalloc(&d_ref, rows.size() * (size_t)N * 4));
        CK(hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_ref, dim3((N + 255) / 256, (unsigned)rows.size()), dim3(256), 0, 0, A, B, N, K, d_rows, (int)rows.size(), d_ref);
        std::vector<float> ref(rows.size() * (size_t)N); CK(hipMemcpy(ref.data(), d_ref, ref.size() * 4, hipMemcpyDeviceToHost));
        std::vector<bf16> got(N);
        double worst = 0; size_t bad = 0;
        for (size_t r = 0; r < rows.size(); r++) {
            CK(hipMemcpy(got.data(), C + (size_t)rows[r] * N, (size_t)N * 2, hipMemcpyDeviceToHost));
            for (int n = 0; n < N; n++) {
                const double d = std::abs((double)(float)got[n] - ref[r * N + n]), tol = 0.02 + 0.01 * std::abs(ref[r * N + n]);
                worst = std::max(worst, d); bad += !(d <= tol);
            }
        }
        printf("check: %zu rows x %d cols, worst abs diff %.4f, %zu outside tolerance\n", rows.size(), N, worst, bad);
    }
    RUN(0) RUN(1) RUN(2) RUN(4) RUN(8) RUN(14)
    return 0;
}
