// GEMM laboratory, round 2: C[M][N] (bf16) = A[M][K] * B[N][K]^T with a PERSISTENT 256 x 256 tile kernel whose operand
// stream never drains: a workgroup walks its output tiles as one flat sequence of 64-deep K-steps; every K-step is four
// 16 KB half-tiles (A rows 0-127, B rows 0-127, B rows 128-255, A rows 128-255) in an 8-slot LDS ring (128 KB); one
// half-tile is issued per phase, five phases ahead of its first MFMA; one barrier per phase; fragments for phase g+1 are read
// into a second register set during phase g.  A wave owns 64 rows of each A half and 32 columns of each B half, so that a
// phase (one 64 x 32 quadrant per wave, 16 MFMAs) touches ONE new half-tile: half-tiles retire in the order they arrive.
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

constexpr int FT = 256, FK = 64, FTHREADS = 512;
constexpr int HALF_ELEMS = 128 * FK;                  // 16 KB
constexpr int NSLOT = 8;
__device__ __forceinline__ int swz64(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
__device__ __forceinline__ int perm32(int p) { return ((p >> 2) & 3) * 8 + (p >> 4) * 4 + (p & 3); }   // LDS row of a 32-row group -> B row

struct Cursor { int kofs, m0, n0; };
// ring layout v2 (k_flat4): the two parities of a kind are neighbours and A0 | A1 | B0 | B1 follow each other, so that every A read is
// within a 16-bit immediate of one base register and every B read of another
__device__ __forceinline__ constexpr int slot2(int par, int kind) { return ((kind == 0 ? 0 : kind == 3 ? 1 : kind == 1 ? 2 : 3) * 2 + par) * HALF_ELEMS; }

// tile order: XCD x owns a contiguous chunk of the supertile-ordered list; its 32 workgroups take consecutive entries
struct Sched {
    int tiles_n, sm, sn, per_xcd, xcd, slot, wper, my_tiles;
    __device__ void init(int M, int N, int sm_, int sn_)
    {
        tiles_n = N / FT; sm = sm_; sn = sn_;
        const int tiles_m = (M + FT - 1) / FT, tiles_m_pad = ((tiles_m + sm - 1) / sm) * sm;
        const int total = tiles_m_pad * tiles_n;
        per_xcd = (total + 7) >> 3; xcd = blockIdx.x & 7; slot = blockIdx.x >> 3; wper = gridDim.x >> 3;
        my_tiles = per_xcd > slot ? (per_xcd - slot + wper - 1) / wper : 0;
    }
    __device__ void tile(int it, int &m0, int &n0) const
    {
        const int lin = xcd * per_xcd + slot + it * wper;
        const int per = sm * sn, sup = lin / per, r = lin - sup * per;
        const int n_sn = tiles_n / sn;
        m0 = ((sup / n_sn) * sm + r / sn) * FT; n0 = ((sup % n_sn) * sn + r % sn) * FT;
    }
};

// per-thread constants of the staging pattern: a half-tile is 128 LDS rows x 128 B = 16 wave-instructions of 8 rows, 2 per wave
struct StageIdx { int row_a, row_b, c8; };   // LDS row of instruction 0 (instruction 1: + 64), its B-row image, source chunk * 8
template <bool IS_B>
__device__ __forceinline__ void issue_half(const bf16 *__restrict__ G, int ld, int row0, int row_max, int kofs, bf16 *slot, int wv, const StageIdx &X)
{
#pragma unroll
    for (int i = 0; i < 2; i++) {
        int gr = row0 + (IS_B ? X.row_b : X.row_a) + 64 * i;      // (perm32 permutes inside 32-row groups: + 64 commutes with it)
        gr = gr > row_max ? row_max : gr;
        const bf16 *src = G + ((unsigned)gr * (unsigned)ld + (unsigned)(kofs + X.c8));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(slot + (wv + 8 * i) * 8 * FK), 16, 0, 0);
    }
}

// s_waitcnt through the builtin (simm16: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14): the compiler's own counter
// tracking sees these waits; after an inline-asm wait it does not, and re-waits lgkmcnt(0) in front of the next MFMA group
template <int N> __device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt(((N >> 4) << 14) | 0x0F70 | (N & 15)); }
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }
template <int N> __device__ __forceinline__ void wait_lgkm() { __builtin_amdgcn_s_waitcnt(0xC07F | (N << 8)); }   // all but the N youngest LDS reads

// fragment registers: the A sub-tile in use (by k half), B0 (kept for quadrants (0,0) and (1,0)) and B1
struct Frags { bf16x8 RA[2][4], RB0[2][2], RB1[2][2]; };
struct RdIdx { int a[2], b[2]; };             // element offsets inside a half-tile of this lane's A / B fragment reads, kk = 0, 1

template <int KK, int HALF, int PAR>
__device__ __forceinline__ void read_A(Frags &F, const bf16 *dsm, const RdIdx &R)
{
    const bf16 *s = dsm + (PAR * 4 + (HALF ? 3 : 0)) * HALF_ELEMS;
#pragma unroll
    for (int i = 0; i < 4; i++) F.RA[KK][i] = *reinterpret_cast<const bf16x8 *>(s + R.a[KK] + i * 16 * FK);
}
template <int KK, int HALF, int PAR>
__device__ __forceinline__ void read_B(Frags &F, const bf16 *dsm, const RdIdx &R)
{
    const bf16 *s = dsm + (PAR * 4 + (HALF ? 2 : 1)) * HALF_ELEMS;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const bf16x8 v = *reinterpret_cast<const bf16x8 *>(s + R.b[KK] + j * 16 * FK);
        if (HALF) F.RB1[KK][j] = v; else F.RB0[KK][j] = v;
    }
}
template <int ABL, int HA, int HB, int KK, bool ZERO = false>
__device__ __forceinline__ void mma8(f32x4 (&acc)[2][2][4][2], const Frags &F)
{
    if constexpr (ABL & 1) { acc[HA][HB][0][0][0] += (float)F.RA[KK][0][0] + (float)(HB ? F.RB1[KK][0][0] : F.RB0[KK][0][0]); return; }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
            acc[HA][HB][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(HB ? F.RB1[KK][j] : F.RB0[KK][j], F.RA[KK][i],
                                                                        ZERO ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[HA][HB][i][j], 0, 0, 0);
}

// ABL bit 0: no MFMA; bit 1: no ds_reads after the first; bit 2: no global loads after the prologue; bit 3: no stores
template <int ABL>
__global__ __launch_bounds__(FTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_flat256(const bf16 *__restrict__ A, const bf16 *__restrict__ B, int M, int N, int K, bf16 *__restrict__ C, int sm, int sn, int stagger)
{
    extern __shared__ __attribute__((aligned(1024))) bf16 dsm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 2, wc = wv & 3;
    const int fr = lane & 15, fq = lane >> 4;
    Sched S; S.init(M, N, sm, sn);
    const int nk = K / FK;
    const int total_steps = S.my_tiles * nk;            // flat K-steps of this workgroup
    if (total_steps == 0) return;
    if (stagger > 0) {                                   // spread the epilogues (HBM write bursts) of the workgroups over a tile time
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), d = (unsigned long long)stagger * (unsigned)(S.slot * 8 + S.xcd) / 256u;
        while (__builtin_amdgcn_s_memtime() - t0 < d) __builtin_amdgcn_s_sleep(8);
    }
    StageIdx X;
    {
        const int row = wv * 8 + (lane >> 3);
        X.row_a = row; X.row_b = (row & ~31) + perm32(row & 31); X.c8 = swz64(row, lane & 7) * 8;   // (row + 64 has the same swizzle: (64 >> 1) & 7 == 0)
    }
    RdIdx R;
    {
        const int ra = wr * 64 + fr, rb = wc * 32 + fr;                       // + 16 i: the swizzle term (row >> 1) & 7 does not change
        R.a[0] = ra * FK + swz64(ra, fq) * 8; R.a[1] = ra * FK + swz64(ra, 4 + fq) * 8;
        R.b[0] = rb * FK + swz64(rb, fq) * 8; R.b[1] = rb * FK + swz64(rb, 4 + fq) * 8;
    }
    // ---- producer.  K-step s: halves A0, B0 are issued in the second half of step s - 2, halves B1, A1 in the first half of step s - 1
    Cursor pc1, pc2;                                       // K-steps s + 1 and s + 2
    int it1, kt1, it2, kt2;
    auto cursor_at = [&](int step, Cursor &c, int &it, int &kt) {
        it = step / nk; kt = step - it * nk;
        c.kofs = kt * FK;
        if (it < S.my_tiles) S.tile(it, c.m0, c.n0); else { c.m0 = 0; c.n0 = 0; }
    };
    auto advance = [&](Cursor &c, int &it, int &kt) {
        if (++kt == nk) { kt = 0; ++it; if (it < S.my_tiles) S.tile(it, c.m0, c.n0); }
        c.kofs = kt * FK;
    };
    auto issue = [&](int step, const Cursor &c, int kind, int par) -> int {   // kind 0 A0, 1 B0, 2 B1, 3 A1; -> 1 if issued
        if (step >= total_steps) return 0;
        if ((ABL & 4) && step >= 2) return 0;
        bf16 *slot = dsm + (par * 4 + kind) * HALF_ELEMS;
        if (kind == 0) issue_half<false>(A, K, c.m0, M - 1, c.kofs, slot, wv, X);
        else if (kind == 3) issue_half<false>(A, K, c.m0 + 128, M - 1, c.kofs, slot, wv, X);
        else if (kind == 1) issue_half<true>(B, K, c.n0, N - 1, c.kofs, slot, wv, X);
        else issue_half<true>(B, K, c.n0 + 128, N - 1, c.kofs, slot, wv, X);
        return 1;
    };

    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    Frags F;

    auto store_tile = [&](int m0, int n0) {
#pragma unroll
        for (int hA = 0; hA < 2; hA++)
#pragma unroll
            for (int hB = 0; hB < 2; hB++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int m = m0 + hA * 128 + wr * 64 + i * 16 + fr;
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; e++) o[e] = (bf16)acc[hA][hB][i][e >> 2][e & 3];
                    if (m < M && (!(ABL & 8) || acc[hA][hB][i][0][0] == 12345.678f))
                        __builtin_nontemporal_store(o, reinterpret_cast<bf16x8 *>(C + (int64_t)m * N + n0 + hB * 128 + wc * 32 + fq * 8));
                    acc[hA][hB][i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[hA][hB][i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
    };
    // end of a half K-step: everything issued before this half must have landed; what this half issued (`fly` halves, 2 DMAs each)
    // and, for two half-steps after an epilogue, its 16 stores may stay in flight
    auto end_wait = [&](int fly, bool stores) {
        if (stores) { if (fly == 2) wait_vm<20>(); else if (fly == 1) wait_vm<18>(); else wait_vm<16>(); }
        else if (fly == 2) wait_vm<4>(); else if (fly == 1) wait_vm<2>(); else wait_vm<0>();
        wait_lgkm0();
    };

    // ---- prologue: all of step 0, A0 / B0 of step 1; then the first fragments
    {
        Cursor c0; int it0, kt0;
        cursor_at(0, c0, it0, kt0); cursor_at(1, pc1, it1, kt1); cursor_at(2, pc2, it2, kt2);
        issue(0, c0, 0, 0); issue(0, c0, 1, 0); issue(0, c0, 2, 0); issue(0, c0, 3, 0);
        const int f = issue(1, pc1, 0, 1) + issue(1, pc1, 1, 1);
        if (f == 2) wait_vm<4>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        read_A<0, 0, 0>(F, dsm, R); read_B<0, 0, 0>(F, dsm, R);
        wait_lgkm0();
    }
    int c_kt = 0, c_it = 0, c_m0, c_n0;
    S.tile(0, c_m0, c_n0);
    int stores_recent = 0;

    // one K-step of parity U (ring half U): groups G0..G7 of 8 MFMAs; reads issued in a group fill registers the previous group used last
#define KSTEP(U)                                                                                                       \
    {                                                                                                                  \
        int fly;                                                                                                       \
        __builtin_amdgcn_s_barrier();                                                                                  \
        fly = issue(ss + 1, pc1, 2, U ^ 1);                                          /* B1(s+1) */                     \
        if (!(ABL & 2)) { read_A<1, 0, U>(F, dsm, R); read_B<1, 0, U>(F, dsm, R); }  /* A0[kk1], B0[kk1] */            \
        wait_lgkm<6>(); __builtin_amdgcn_s_setprio(1); mma8<ABL, 0, 0, 0>(acc, F); __builtin_amdgcn_s_setprio(0);                      \
        if (!(ABL & 2)) read_B<0, 1, U>(F, dsm, R);                                  /* B1[kk0] */                     \
        wait_lgkm<2>(); __builtin_amdgcn_s_setprio(1); mma8<ABL, 0, 0, 1>(acc, F); __builtin_amdgcn_s_setprio(0);                      \
        fly += issue(ss + 1, pc1, 3, U ^ 1);                                         /* A1(s+1) */                     \
        if (!(ABL & 2)) read_B<1, 1, U>(F, dsm, R);                                  /* B1[kk1] */                     \
        wait_lgkm<2>(); __builtin_amdgcn_s_setprio(1); mma8<ABL, 0, 1, 0>(acc, F); __builtin_amdgcn_s_setprio(0);                      \
        if (!(ABL & 2)) read_A<0, 1, U>(F, dsm, R);                                  /* A1[kk0] */                     \
        wait_lgkm<4>(); __builtin_amdgcn_s_setprio(1); mma8<ABL, 0, 1, 1>(acc, F); __builtin_amdgcn_s_setprio(0);                      \
        end_wait(fly, stores_recent > 0);                                                                              \
        if (stores_recent > 0) stores_recent--;                                                                        \
        __builtin_amdgcn_s_barrier();                                                                                  \
        fly = issue(ss + 2, pc2, 0, U);                                              /* A0(s+2) */                     \
        if (!(ABL & 2)) read_A<1, 1, U>(F, dsm, R);                                  /* A1[kk1] */                     \
        wait_lgkm<4>(); __builtin_amdgcn_s_setprio(1); mma8<ABL, 1, 1, 0>(acc, F); __builtin_amdgcn_s_setprio(0);                      \
        wait_lgkm<0>(); __builtin_amdgcn_s_setprio(1); mma8<ABL, 1, 1, 1>(acc, F); __builtin_amdgcn_s_setprio(0);                      \
        fly += issue(ss + 2, pc2, 1, U);                                             /* B0(s+2) */                     \
        wait_lgkm<0>(); __builtin_amdgcn_s_setprio(1); mma8<ABL, 1, 0, 0>(acc, F); __builtin_amdgcn_s_setprio(0);                      \
        if (!(ABL & 2) && ss + 1 < total_steps) { read_A<0, 0, U ^ 1>(F, dsm, R); read_B<0, 0, U ^ 1>(F, dsm, R); }   /* A0(s+1)[kk0], B0(s+1)[kk0] */ \
        wait_lgkm<6>(); __builtin_amdgcn_s_setprio(1); mma8<ABL, 1, 0, 1>(acc, F); __builtin_amdgcn_s_setprio(0);                      \
        pc1 = pc2; it1 = it2; kt1 = kt2; advance(pc2, it2, kt2);                                                       \
        if (++c_kt == nk) {                                                                                            \
            store_tile(c_m0, c_n0); stores_recent = 2;                                                                 \
            c_kt = 0; ++c_it;                                                                                          \
            if (c_it < S.my_tiles) S.tile(c_it, c_m0, c_n0);                                                           \
        }                                                                                                              \
        end_wait(fly, stores_recent > 0);                                                                              \
        if (stores_recent > 0) stores_recent--;                                                                        \
    }
    for (int s = 0; s < total_steps; s += 2) {
        { const int ss = s; KSTEP(0) }
        if (s + 1 < total_steps) { const int ss = s + 1; KSTEP(1) }
    }
#undef KSTEP
}

// ---------------------------------------------------------------------------------------------------------------
// flat256, branch-free main path (v4): K a multiple of 128 (even number of K-steps per tile); the producer never stops
// (past the end of the stream it re-issues the last K-step's addresses into ring slots nobody reads), so every wait is a
// compile-time count; the three K-step flavours (first of a tile / middle / last of a tile with the epilogue) are separate
// code copies.
// ---------------------------------------------------------------------------------------------------------------

template <int KK, int HALF, int PAR>
__device__ __forceinline__ void read_A2(Frags &F, const bf16 *dsm, const RdIdx &R)
{
    const bf16 *s = dsm + slot2(PAR, HALF ? 3 : 0);
#pragma unroll
    for (int i = 0; i < 4; i++) F.RA[KK][i] = *reinterpret_cast<const bf16x8 *>(s + R.a[KK] + i * 16 * FK);
}
template <int KK, int HALF, int PAR>
__device__ __forceinline__ void read_B2(Frags &F, const bf16 *dsm, const RdIdx &R)
{
    const bf16 *s = dsm + slot2(PAR, HALF ? 2 : 1) - 4 * HALF_ELEMS;     // R.b already points into the B region
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const bf16x8 v = *reinterpret_cast<const bf16x8 *>(s + R.b[KK] + j * 16 * FK);
        if (HALF) F.RB1[KK][j] = v; else F.RB0[KK][j] = v;
    }
}

template <int KK, int HALF>
__device__ __forceinline__ void read_A3(Frags &F, const bf16 *dsm, const RdIdx &R, int par)
{
    const bf16 *s = dsm + ((HALF ? 1 : 0) * 2 + par) * HALF_ELEMS;
#pragma unroll
    for (int i = 0; i < 4; i++) F.RA[KK][i] = *reinterpret_cast<const bf16x8 *>(s + R.a[KK] + i * 16 * FK);
}
template <int KK, int HALF>
__device__ __forceinline__ void read_B3(Frags &F, const bf16 *dsm, const RdIdx &R, int par)
{
    const bf16 *s = dsm + ((HALF ? 1 : 0) * 2 + par) * HALF_ELEMS;           // R.b already points into the B region
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const bf16x8 v = *reinterpret_cast<const bf16x8 *>(s + R.b[KK] + j * 16 * FK);
        if (HALF) F.RB1[KK][j] = v; else F.RB0[KK][j] = v;
    }
}

// LDS-DMA through a buffer resource: the per-lane part of the address is ONE 32-bit offset that never changes (row * ld + swizzled
// chunk), everything that moves (tile origin, K offset, the second instruction's 64 rows) is a scalar offset; rows past the
// end of the matrix read as zeros (range check of the resource) -- no address arithmetic in the loop, no 64-bit pointers in VGPRs
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void issue_half_buf(__amdgpu_buffer_rsrc_t rs, int voff, int soff, int ld2_64, bf16 *slot, int wv)
{
#pragma unroll
    for (int i = 0; i < 2; i++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(slot + (wv + 8 * i) * 8 * FK), 16, voff, soff + i * ld2_64, 0, 0);
}

template <int ABL>
__global__ __launch_bounds__(FTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_flat4(const bf16 *__restrict__ A, const bf16 *__restrict__ B, int M, int N, int K, bf16 *__restrict__ C, int sm, int sn, int stagger, unsigned long long *stamps = nullptr)
{
    unsigned long long st0 = 0, sr0 = 0;
    if (stamps) { st0 = __builtin_amdgcn_s_memtime(); sr0 = __builtin_amdgcn_s_memrealtime(); }
    extern __shared__ __attribute__((aligned(1024))) bf16 dsm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 2, wc = wv & 3;
    const int fr = lane & 15, fq = lane >> 4;
    Sched S; S.init(M, N, sm, sn);
    const int nk = K / FK;
    if (S.my_tiles == 0) return;
    if (stagger > 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), d = (unsigned long long)stagger * (unsigned)(S.slot * 8 + S.xcd) / 256u;
        while (__builtin_amdgcn_s_memtime() - t0 < d) __builtin_amdgcn_s_sleep(8);
    }
    StageIdx X;
    {
        const int row = wv * 8 + (lane >> 3);
        X.row_a = row; X.row_b = (row & ~31) + perm32(row & 31); X.c8 = swz64(row, lane & 7) * 8;
    }
    RdIdx R;
    {
        const int ra = wr * 64 + fr, rb = wc * 32 + fr;
        R.a[0] = ra * FK + swz64(ra, fq) * 8; R.a[1] = ra * FK + swz64(ra, 4 + fq) * 8;
        R.b[0] = rb * FK + swz64(rb, fq) * 8 + 4 * HALF_ELEMS; R.b[1] = rb * FK + swz64(rb, 4 + fq) * 8 + 4 * HALF_ELEMS;   // (B region: ring bytes 64 K .. 128 K)
    }
    Cursor pc1, pc2;
    int it1, kt1, it2, kt2;
    const int last_tile = S.my_tiles - 1;
    auto set_tile = [&](Cursor &c, int it) { S.tile(it < last_tile ? it : last_tile, c.m0, c.n0); };
    auto advance = [&](Cursor &c, int &it, int &kt) {
        if (++kt == nk) { kt = 0; ++it; set_tile(c, it); }
        c.kofs = kt * FK;
    };
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(A), 0, (int)((unsigned)M * (unsigned)K * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(B), 0, (int)((unsigned)N * (unsigned)K * 2u), 0x00020000);
    const int voffA = (X.row_a * K + X.c8) * 2, voffB = (X.row_b * K + X.c8) * 2, ld2_64 = 64 * K * 2;
    auto issue = [&](const Cursor &c, int kind, int par) {   // kind 0 A0, 1 B0, 2 B1, 3 A1
        if (ABL & 4) return;
        bf16 *slot = dsm + ((kind == 0 ? 0 : kind == 3 ? 1 : kind == 1 ? 2 : 3) * 2 + par) * HALF_ELEMS;
        if (ABL & 512) {
            // ablation: operands as if stored half-tile by half-tile ([rows / 128][K / 64][128 x 64] images, 16 KB contiguous each): every DMA
            // instruction copies 1 KB of consecutive bytes (timing only: the data is not what the product needs)
            const bool is_a = kind == 0 || kind == 3;
            const unsigned row0 = (unsigned)(is_a ? c.m0 : c.n0) + ((kind == 3 || kind == 2) ? 128u : 0u);
            const unsigned img = ((row0 >> 7) * (unsigned)(K / FK) + (unsigned)(c.kofs / FK)) * 16384u;
#pragma unroll
            for (int i = 0; i < 2; i++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(is_a ? rsA : rsB, (__attribute__((address_space(3))) void *)(slot + (wv + 8 * i) * 8 * FK), 16,
                                                         lane * 16, (int)(img + (unsigned)(wv + 8 * i) * 1024u), 0, 0);
            return;
        }
        if (kind == 0) issue_half_buf(rsA, voffA, (c.m0 * K + c.kofs) * 2, ld2_64, slot, wv);
        else if (kind == 3) issue_half_buf(rsA, voffA, ((c.m0 + 128) * K + c.kofs) * 2, ld2_64, slot, wv);
        else if (kind == 1) issue_half_buf(rsB, voffB, (c.n0 * K + c.kofs) * 2, ld2_64, slot, wv);
        else issue_half_buf(rsB, voffB, ((c.n0 + 128) * K + c.kofs) * 2, ld2_64, slot, wv);
    };
    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    Frags F;
    // epilogue: bf16 rows through a buffer resource; per lane ONE constant offset (row fr, 16-byte column group fq), everything else
    // scalar; rows past M are dropped by the resource's range check.  The accumulators are not cleared: the next tile's first
    // K-step starts them from the zero constant.
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)((unsigned)M * (unsigned)N * 2u), 0x00020000);
    const int voffC = (fr * N + fq * 8) * 2;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    auto store_tile = [&](int m0, int n0) {
#pragma unroll
        for (int hA = 0; hA < 2; hA++)
#pragma unroll
            for (int hB = 0; hB < 2; hB++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; e++) o[e] = (bf16)acc[hA][hB][i][e >> 2][e & 3];
                    const int soff = ((m0 + hA * 128 + wr * 64 + i * 16) * N + n0 + hB * 128 + wc * 32) * 2;
                    if (ABL & 16) {     // ablation: the same stores into a 128 KB window per workgroup that stays in L2
                        const int mm = blockIdx.x * 256 + hA * 128 + wr * 64 + i * 16 + fr;
                        *reinterpret_cast<bf16x8 *>(C + (int64_t)mm * 256 + hB * 128 + wc * 32 + fq * 8) = o;
                    } else if (ABL & 32) {   // ablation: the same bytes as 8 rows x 128 B (full lines) per instruction; placement is NOT that of the product
                        const int voff2 = ((fr & 7) * N + (fq + 4 * (fr >> 3)) * 8) * 2;
                        const int soff2 = ((m0 + hA * 128 + wr * 64 + i * 16 + hB * 8) * N + n0 + wc * 64) * 2;
                        if (ABL & 128) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsC, voff2, soff2, 2);
                        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsC, voff2, soff2, 0);
                    } else if (ABL & 128) {  // ablation: non-temporal stores
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsC, voffC, soff, 2);
                    } else if (!(ABL & 8))
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsC, voffC, soff, 0);
                }
    };
    // ---- prologue: all of step 0, A0 / B0 / B1 of step 1; then the fragments the "second half of step -1" would have read
    {
        Cursor c0; c0.kofs = 0; set_tile(c0, 0);
        it1 = 0; kt1 = 0; pc1 = c0; advance(pc1, it1, kt1);
        it2 = it1; kt2 = kt1; pc2 = pc1; advance(pc2, it2, kt2);
        issue(c0, 0, 0); issue(c0, 1, 0); issue(c0, 2, 0); issue(c0, 3, 0);
        issue(pc1, 0, 1); issue(pc1, 1, 1); issue(pc1, 2, 1);
        wait_vm<6>();
        __builtin_amdgcn_s_barrier();
        read_A3<0, 0>(F, dsm, R, 0); read_B3<0, 0>(F, dsm, R, 0); read_B3<0, 1>(F, dsm, R, 0); read_B3<1, 1>(F, dsm, R, 0);
        wait_lgkm0();
    }
    int c_m0, c_n0;
    S.tile(0, c_m0, c_n0);
    // K-step s of ring parity U.  Fragment reads are issued two MFMA groups (or a barrier) before their first use:
    //   first half:  (0,0)k0 | (0,1)k0 | (0,0)k1 | (0,1)k1      second half:  (1,1)k0 | (1,0)k0 | (1,1)k1 | (1,0)k1
    // DMA: first half issues A1(s+1); second half A0(s+2), B0(s+2), B1(s+2).  FIRST: the previous tile's 16 stores are still among
    // the youngest operations at the end of the first half; LAST: the epilogue's stores at the end of the second half.
#define KSTEP4(FIRST, LAST)                                                                                            \
    {                                                                                                                  \
        __builtin_amdgcn_s_barrier();                                                                                  \
        __builtin_amdgcn_sched_barrier(0); issue(pc1, 3, par ^ 1); __builtin_amdgcn_sched_barrier(0);  /* A1(s+1) */   \
        if (!(ABL & 2)) { read_A3<1, 0>(F, dsm, R, par); read_B3<1, 0>(F, dsm, R, par); }              /* A0k1, B0k1 */ \
        wait_lgkm<6>(); __builtin_amdgcn_s_setprio(1); if (FIRST) mma8<ABL, 0, 0, 0, true>(acc, F); else mma8<ABL, 0, 0, 0>(acc, F); GEND                                \
        __builtin_amdgcn_s_setprio(1); if (FIRST) mma8<ABL, 0, 1, 0, true>(acc, F); else mma8<ABL, 0, 1, 0>(acc, F); GEND                                                \
        if (!(ABL & 2)) read_A3<0, 1>(F, dsm, R, par);                                                 /* A1k0 */      \
        wait_lgkm<4>(); __builtin_amdgcn_s_setprio(1); mma8<ABL, 0, 0, 1>(acc, F); GEND                                \
        __builtin_amdgcn_s_setprio(1); mma8<ABL, 0, 1, 1>(acc, F); GEND                                                \
        if (FIRST) wait_vm<18>(); else wait_vm<2>();                                                                   \
        wait_lgkm0();                                                                                                  \
        __builtin_amdgcn_s_barrier();                                                                                  \
        __builtin_amdgcn_sched_barrier(0); issue(pc2, 0, par); __builtin_amdgcn_sched_barrier(0);      /* A0(s+2) */   \
        if (!(ABL & 2)) read_A3<1, 1>(F, dsm, R, par);                                                 /* A1k1 */      \
        wait_lgkm<4>(); __builtin_amdgcn_s_setprio(1); if (FIRST) mma8<ABL, 1, 1, 0, true>(acc, F); else mma8<ABL, 1, 1, 0>(acc, F); GEND                                \
        __builtin_amdgcn_sched_barrier(0); issue(pc2, 1, par); __builtin_amdgcn_sched_barrier(0);      /* B0(s+2) */   \
        if (!(ABL & 2)) read_B3<0, 1>(F, dsm, R, par ^ 1);                                             /* B1k0(s+1) */ \
        wait_lgkm<2>(); __builtin_amdgcn_s_setprio(1); if (FIRST) mma8<ABL, 1, 0, 0, true>(acc, F); else mma8<ABL, 1, 0, 0>(acc, F); GEND                                \
        __builtin_amdgcn_sched_barrier(0); issue(pc2, 2, par); __builtin_amdgcn_sched_barrier(0);      /* B1(s+2) */   \
        if (!(ABL & 2)) { read_A3<0, 0>(F, dsm, R, par ^ 1); read_B3<0, 0>(F, dsm, R, par ^ 1); }      /* A0k0(s+1), B0k0(s+1) */ \
        wait_lgkm<6>(); __builtin_amdgcn_s_setprio(1); mma8<ABL, 1, 1, 1>(acc, F); GEND                                \
        if (!(ABL & 2)) read_B3<1, 1>(F, dsm, R, par ^ 1);                                             /* B1k1(s+1) */ \
        wait_lgkm<2>(); __builtin_amdgcn_s_setprio(1); mma8<ABL, 1, 0, 1>(acc, F); GEND                                \
        pc1 = pc2; it1 = it2; kt1 = kt2; advance(pc2, it2, kt2);                                                       \
        if (LAST) { unsigned long long ts0 = 0; if (ABL & 256) ts0 = __builtin_amdgcn_s_memtime();                      \
                    store_tile(c_m0, c_n0); if (ABL & 256) ph[4] += __builtin_amdgcn_s_memtime() - ts0; wait_vm<22>(); } else wait_vm<6>(); \
        wait_lgkm0();                                                                                                  \
        par ^= 1;                                                                                                      \
        if (ABL & 256) { const unsigned long long tn = __builtin_amdgcn_s_memtime();                                   \
            const int cls = (FIRST) ? 0 : (kt == 1) ? 1 : (LAST) ? 3 : 2; ph[cls] += tn - tprev; if (cls == 2) ph[5]++; tprev = tn; } \
    }
#define GEND __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0);
    int par = 0;
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();   // cycles in: first / second / middle / last K-steps, store issue; middle count
    for (int it = 0; it < S.my_tiles; it++) {
        for (int kt = 0; kt < nk; kt++) KSTEP4(kt == 0, kt == nk - 1)
        if (it + 1 < S.my_tiles) S.tile(it + 1, c_m0, c_n0);
    }
#undef GEND
#undef KSTEP4
    if (stamps && threadIdx.x == 0) { unsigned long long *o = stamps + 4 * blockIdx.x; o[0] = st0; o[1] = sr0; o[2] = __builtin_amdgcn_s_memtime(); o[3] = __builtin_amdgcn_s_memrealtime(); }
    if ((ABL & 256) && stamps && threadIdx.x == 0) { unsigned long long *o = stamps + 4 * 256 + 8 * blockIdx.x; for (int i = 0; i < 6; i++) o[i] = ph[i]; o[6] = (unsigned long long)S.my_tiles; }
}

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

// plain reference: one thread per output element (only for a sample of rows)
__global__ void k_ref(const bf16 *A, const bf16 *B, int N, int K, const int *rows, int n_rows, float *out)
{
    const int r = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows || n >= N) return;
    const bf16 *a = A + (int64_t)rows[r] * K, *b = B + (int64_t)n * K;
    float s = 0.f;
    for (int k = 0; k < K; k++) s += (float)a[k] * (float)b[k];
    out[(int64_t)r * N + n] = s;
}

int main(int argc, char **argv)
{
    const int M = argc > 3 ? atoi(argv[3]) : 192000, N = argc > 1 ? atoi(argv[1]) : 3072, K = argc > 2 ? atoi(argv[2]) : 768;
    bf16 *A, *B, *C;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
    {
        std::vector<bf16> h((size_t)M * K);
        unsigned s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 9) & 0x7fff) / 16384.0f - 1.0f; };     // uniform [-1, 1): random data (DVFS-honest)
        for (auto &v : h) v = (bf16)rnd();
        CK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        std::vector<bf16> w((size_t)N * K);
        for (auto &v : w) v = (bf16)(rnd() * 0.125f);
        CK(hipMemcpy(B, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    }
    const double tf = 2.0 * M * N * K / 1e9;
    int n_cu = 256;
    { hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); n_cu = p.multiProcessorCount; }
    const int tiles_n = N / FT;
    int sn = 1;
    for (int cand : {4, 3, 2}) if (tiles_n % cand == 0) { sn = cand; break; }
    const int lds = NSLOT * HALF_ELEMS * 2;
    printf("M %d N %d K %d, %d CUs, sn %d, LDS %d\n", M, N, K, n_cu, sn, lds);
    if (argc > 4) {   // profiling mode: only the kernels to be counted (argv[4]: bit mask of variants)
        const int which = atoi(argv[4]);
        CK(hipFuncSetAttribute((const void *)k_flat4<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CK(hipFuncSetAttribute((const void *)k_flat4<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CK(hipFuncSetAttribute((const void *)k_flat4<10>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CK(hipFuncSetAttribute((const void *)k_flat4<12>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        for (int rep = 0; rep < 3; rep++) {
            if (which & 1) hipLaunchKernelGGL((k_flat4<0>), dim3(n_cu), dim3(FTHREADS), lds, 0, A, B, M, N, K, C, 8, sn, 0, nullptr);
            if (which & 2) hipLaunchKernelGGL((k_flat4<8>), dim3(n_cu), dim3(FTHREADS), lds, 0, A, B, M, N, K, C, 8, sn, 0, nullptr);
            if (which & 4) hipLaunchKernelGGL((k_flat4<10>), dim3(n_cu), dim3(FTHREADS), lds, 0, A, B, M, N, K, C, 8, sn, 0, nullptr);
            if (which & 8) hipLaunchKernelGGL((k_flat4<12>), dim3(n_cu), dim3(FTHREADS), lds, 0, A, B, M, N, K, C, 8, sn, 0, nullptr);
        }
        CK(hipDeviceSynchronize());
        return 0;
    }
    CK(hipMemset(C, 0xff, (size_t)M * N * 2));
#define RUNF(ABL, SM, STG) { CK(hipFuncSetAttribute((const void *)k_flat256<ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        float ms = time_ms([&] { hipLaunchKernelGGL((k_flat256<ABL>), dim3(n_cu), dim3(FTHREADS), lds, 0, A, B, M, N, K, C, SM, sn, STG); }, 30); \
        CK(hipGetLastError()); printf("flat256 abl %2d sm %2d stagger %6d: %.3f ms  %.0f TFLOP/s\n", ABL, SM, STG, ms, tf / ms); fflush(stdout); }
    RUNF(0, 8, 0)
    auto check_rows = [&](const char *what) {   // correctness on a sample of rows (first, last, tile borders, random)
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
        printf("check %s: %zu rows x %d cols, worst abs diff %.4f, %zu outside tolerance\n", what, rows.size(), N, worst, bad);
    };
    check_rows("flat256");
    RUNF(0, 4, 0) RUNF(0, 16, 0) RUNF(0, 8, 20000) RUNF(0, 8, 40000) RUNF(0, 8, 80000)
    RUNF(1, 8, 0) RUNF(3, 8, 0) RUNF(14, 8, 0) RUNF(8, 8, 0) RUNF(4, 8, 0) RUNF(11, 8, 0)
#define RUNP(ABL, SM, STG) { CK(hipFuncSetAttribute((const void *)k_flat4<ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        float ms = time_ms([&] { hipLaunchKernelGGL((k_flat4<ABL>), dim3(n_cu), dim3(FTHREADS), lds, 0, A, B, M, N, K, C, SM, sn, STG); }, 30); \
        CK(hipGetLastError()); printf("flat4 abl %2d sm %2d stagger %6d: %.3f ms  %.0f TFLOP/s\n", ABL, SM, STG, ms, tf / ms); fflush(stdout); }
    CK(hipMemset(C, 0xff, (size_t)M * N * 2));
    RUNP(0, 8, 0)
    check_rows("flat4");
    RUNP(0, 16, 0) RUNP(0, 16, 40000) RUNP(0, 8, 40000)
    // (tried: the A operand three K-steps deep -- 10 slots, 160 KB, B half-tiles queued before the step's A0 so that the in-order vmcnt waits leave
    //  the A loads outstanding: +2 % / -1 % / +4 % on the out-proj / fc1 / fc2 shapes: operand latency is not what bounds the K-steps)

    RUNP(32, 8, 40000) RUNP(128, 8, 40000) RUNP(160, 8, 40000) RUNP(8, 8, 40000) RUNP(512, 8, 40000) RUNP(512, 8, 0) RUNP(520, 8, 40000)
    RUNP(1, 8, 0) RUNP(14, 8, 0) RUNP(8, 8, 0) RUNP(4, 8, 0) RUNP(11, 8, 0) RUNP(16, 8, 0) RUNP(16, 8, 40000) RUNP(64, 8, 40000) RUNP(10, 8, 0) RUNP(12, 8, 0) RUNP(9, 8, 0)
    {
        unsigned long long *stamps; CK(hipMalloc(&stamps, 256 * 32 + 256 * 64)); CK(hipMemset(stamps, 0, 256 * 32 + 256 * 64));
        auto clock_of = [&](const char *name, auto launch) {
            for (int i = 0; i < 300; i++) launch();
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(n_cu * 4);
            CK(hipMemcpy(h.data(), stamps, n_cu * 32, hipMemcpyDeviceToHost));
            std::vector<double> f, cyc;
            for (int b = 0; b < n_cu; b++) if (h[4 * b + 3] > h[4 * b + 1]) { f.push_back((double)(h[4 * b + 2] - h[4 * b]) / (double)(h[4 * b + 3] - h[4 * b + 1]) * 100.0); cyc.push_back((double)(h[4 * b + 2] - h[4 * b])); }
            std::sort(f.begin(), f.end()); std::sort(cyc.begin(), cyc.end());
            printf("in-kernel clock, %s: median %.0f MHz (p10 %.0f, p90 %.0f); median workgroup lifetime %.0f cycles\n", name, f[f.size() / 2], f[f.size() / 10], f[f.size() * 9 / 10], cyc[cyc.size() / 2]);
        };
        CK(hipFuncSetAttribute((const void *)k_flat4<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CK(hipFuncSetAttribute((const void *)k_flat4<14>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CK(hipFuncSetAttribute((const void *)k_flat4<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        {   // where a workgroup's time goes (wave 0's view): K-steps by position in the tile, and the time spent ISSUING the epilogue's stores
            CK(hipFuncSetAttribute((const void *)k_flat4<256>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k_flat4<256>), dim3(n_cu), dim3(FTHREADS), lds, 0, A, B, M, N, K, C, 8, sn, 40000, stamps);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(n_cu * 8);
            CK(hipMemcpy(h.data(), stamps + 4 * 256, n_cu * 64, hipMemcpyDeviceToHost));
            double a[6] = {0, 0, 0, 0, 0, 0}, tiles = 0;
            for (int b = 0; b < n_cu; b++) { for (int i = 0; i < 6; i++) a[i] += (double)h[8 * b + i]; tiles += (double)h[8 * b + 6]; }
            printf("per tile (cycles, mean over workgroups): first K-step %.0f, second %.0f, a middle one %.0f, last (incl. epilogue) %.0f of which store issue %.0f; tile total %.0f\n",
                   a[0] / tiles, a[1] / tiles, a[2] / a[5], a[3] / tiles, a[4] / tiles, (a[0] + a[1] + a[2] + a[3]) / tiles);
        }
        clock_of("flat4 full", [&] { hipLaunchKernelGGL((k_flat4<0>), dim3(n_cu), dim3(FTHREADS), lds, 0, A, B, M, N, K, C, 8, sn, 0, stamps); });
        clock_of("flat4 MFMA only", [&] { hipLaunchKernelGGL((k_flat4<14>), dim3(n_cu), dim3(FTHREADS), lds, 0, A, B, M, N, K, C, 8, sn, 0, stamps); });
        clock_of("flat4 no stores", [&] { hipLaunchKernelGGL((k_flat4<8>), dim3(n_cu), dim3(FTHREADS), lds, 0, A, B, M, N, K, C, 8, sn, 0, stamps); });
    }
    {   // sustained (clocks settled)
        for (int i = 0; i < 300; i++) hipLaunchKernelGGL((k_flat256<0>), dim3(n_cu), dim3(FTHREADS), lds, 0, A, B, M, N, K, C, 8, sn, 40000);
        CK(hipDeviceSynchronize());
        float ms = time_ms([&] { hipLaunchKernelGGL((k_flat256<0>), dim3(n_cu), dim3(FTHREADS), lds, 0, A, B, M, N, K, C, 8, sn, 40000); }, 50);
        printf("flat256 sustained: %.3f ms  %.0f TFLOP/s\n", ms, tf / ms);
    }
    return 0;
}
