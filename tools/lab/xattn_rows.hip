// The encoder-output form of cross-attention (see xattn_absorb.hip, csrc/pce_xattn.inc) for a TEACHER-FORCED pass: T tokens per clip, each a 16-row
// block (its heads), RB tokens per workgroup sharing every tile of E: S^T = E Q'^T and U^T = E^T P^T cost RB times the MFMAs of the one-token kernel
// per tile while the LDS traffic and the DMA stay.  A costing experiment for DESIGN.md section 9 (nothing in the library uses it): how fast can the
// alignment's cross K|V projection + cross-attention (14.6 ms per alignment today) be replaced?  E is re-streamed ceil(T / RB) times per clip.
//   hipcc -O3 --offload-arch=gfx950 [-DRBN=3] [-DLO=0] -o bin/xattn_rows xattn_rows.hip ;  bin/xattn_rows [clips] [d] [tokens]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 op_t;
typedef __attribute__((ext_vector_type(8))) op_t opx8;
typedef __attribute__((ext_vector_type(4))) op_t opx4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ f32x4 mfma16(opx8 a, opx8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16k16(opx4 a, opx4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }

// LDS image of a 16-frame tile of E: the rows as they lie in memory (row pitch 2 d bytes, a multiple of 256), the sixteen 16-byte chunks of every
// 256-byte group XOR-swizzled by the row (cdna_hip_programming.md T10, image (b)): serves the row reads of S^T = E Q'^T and the transposed reads of
// U^T = E^T P^T, and a DMA instruction (64 consecutive chunks of the image) still reads 1 KB of CONTIGUOUS memory.
__device__ __forceinline__ int xa_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
template <int D> __device__ __forceinline__ int xa_off(int row, int col8 /* 16-byte chunk of the row: column / 8 */)
{
    return 2 * D * row + 16 * (col8 ^ xa_swz(row));          // bytes
}

__device__ __forceinline__ s16x4 tr_read(unsigned addr) { s16x4 r; asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr)); return r; }

#ifndef RBN
#define RBN 3           // tokens (16-row blocks) per workgroup
#endif
#ifndef LO
#define LO 0            // 1: Q' and P as hi + lo pairs (two MFMAs each), 0: single-rounded operands
#endif
#ifndef CPOL
#define CPOL 0         // cache policy bits of the LDS-DMA loads (gfx950: 1 sc0, 2 nt, 16 sc1)
#endif
#ifndef ABL
#define ABL 0          // ablation bits (timing only, results wrong): 1 no lo-part MFMAs, 2 no U^T product, 4 no DMA, 8 no compute (DMA + barriers only)
#endif

struct XaArgs {
    const op_t *E; int64_t e_clip; int e_ld;        // E + clip * e_clip + t * e_ld (+ column)
    const op_t *qp_hi, *qp_lo;                      // [clip][T][16][D]: Q' of every token (scaled by log2 e / sqrt(head dim)), rows >= heads zero
    const int *k_len;
    float *u_out;                                   // [clip][T][16][D]: normalised U
    int heads, n, T;
};

// one workgroup = one (clip, group of RB tokens): 16-frame tiles of E through a ring of NSLOT LDS slots, 4 waves; wave w owns d / 4 of the reduction
// axis of S^T = E Q'^T (partials summed through LDS) and d / 4 of the columns of U^T = E^T P^T, for all RB tokens
template <int D, int NSLOT, int RB>
__device__ __forceinline__ void xattn_rows_body(const XaArgs &A)
{
    constexpr int TF = 16, KS = D / 128, CB = D / 64, LPW = D / 128, TILE = TF * D * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];                          // NSLOT tiles | RB x 4 x 1 KB of partial S^T
    const int groups = (A.T + RB - 1) / RB;
    const int clip = (int)blockIdx.x / groups, tg = (int)blockIdx.x - clip * groups;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n16 = lane & 15, g = lane >> 4;
    const int Sk = A.k_len[clip];
    const int nt = (Sk + TF - 1) / TF;
    const op_t *eb = A.E + (int64_t)clip * A.e_clip;
    const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(const_cast<op_t *>(eb), 0, ((Sk - 1) * A.e_ld + D) * 2, 0x00020000);
    opx8 qh[RB][KS];
#if LO
    opx8 ql[RB][KS];
#endif
#pragma unroll
    for (int rb = 0; rb < RB; rb++) {
        const int tok = min(tg * RB + rb, A.T - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int64_t o = (((int64_t)clip * A.T + tok) * 16 + n16) * D + 32 * (wv * KS + ks) + 8 * g;
            qh[rb][ks] = *reinterpret_cast<const opx8 *>(A.qp_hi + o);
#if LO
            ql[rb][ks] = *reinterpret_cast<const opx8 *>(A.qp_lo + o);
#endif
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(qh[rb][ks]));
#if LO
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(ql[rb][ks]));
#endif
        }
    int voff[LPW];
#pragma unroll
    for (int i = 0; i < LPW; i++) {
        const int X = 64 * (wv + 4 * i) + lane, row = X / (D / 8), ch = (X % (D / 8)) ^ xa_swz(row);
        voff[i] = (row * A.e_ld + 8 * ch) * 2;
    }
    auto stage = [&](int t) {
        char *slot = smem + (t % NSLOT) * TILE;
#pragma unroll
        for (int i = 0; i < LPW; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsE, (__attribute__((address_space(3))) void *)(slot + 1024 * (wv + 4 * i)), 16, voff[i], t * TF * A.e_ld * 2, 0, CPOL);
    };
    int raddr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) raddr[ks] = xa_off<D>(n16, 4 * (wv * KS + ks) + g);
    const int q4 = n16 >> 2, p4 = n16 & 3;
    float *xs = reinterpret_cast<float *>(smem + NSLOT * TILE);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 u[RB][CB];
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
#pragma unroll
        for (int cb = 0; cb < CB; cb++) u[rb][cb] = z4;
    float m_run[RB], l_part[RB];
#pragma unroll
    for (int rb = 0; rb < RB; rb++) { m_run[rb] = -1e30f; l_part[rb] = 0.f; }
    for (int t = 0; t < NSLOT - 1 && t < nt; t++) stage(t);
    for (int t = 0; t < nt; t++) {
        if (NSLOT >= 3 && t + 1 < nt) __builtin_amdgcn_s_waitcnt(0x0F70 | ((NSLOT - 2) * LPW)); else __builtin_amdgcn_s_waitcnt(0x0F70 | 0);
        __builtin_amdgcn_s_barrier();
        if (t + NSLOT - 1 < nt) stage(t + NSLOT - 1);
        const char *sC = smem + (t % NSLOT) * TILE;
        const unsigned sbase = lds0 + (unsigned)((t % NSLOT) * TILE);
        {   // ---- partial S^T of every token over this wave's quarter of d: the E fragments are read once
            f32x4 a[RB];
#pragma unroll
            for (int rb = 0; rb < RB; rb++) a[rb] = z4;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const opx8 ef = *reinterpret_cast<const opx8 *>(sC + raddr[ks]);
#pragma unroll
                for (int rb = 0; rb < RB; rb++) {
                    a[rb] = mfma16(ef, qh[rb][ks], a[rb]);
#if LO
                    a[rb] = mfma16(ef, ql[rb][ks], a[rb]);
#endif
                }
            }
#pragma unroll
            for (int rb = 0; rb < RB; rb++) *reinterpret_cast<f32x4 *>(xs + ((rb * 4 + wv) * 64 + lane) * 4) = a[rb];
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        opx4 ph[RB];
#if LO
        opx4 pl[RB];
#endif
        const int f0 = t * TF;
#pragma unroll
        for (int rb = 0; rb < RB; rb++) {
            f32x4 sc = z4;
#pragma unroll
            for (int w2 = 0; w2 < 4; w2++) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(xs + ((rb * 4 + w2) * 64 + lane) * 4);
                sc[0] += v[0]; sc[1] += v[1]; sc[2] += v[2]; sc[3] += v[3];
            }
            if (f0 + TF > Sk) {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (f0 + 4 * g + i >= Sk) sc[i] = -1e30f;
            }
            float m = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            const float mx = fmaxf(m_run[rb], m);
            if (__builtin_amdgcn_ballot_w64(mx > m_run[rb]) != 0) {
                const float corr = __builtin_amdgcn_exp2f(m_run[rb] - mx);
                l_part[rb] *= corr;
#pragma unroll
                for (int cb = 0; cb < CB; cb++) { u[rb][cb][0] *= corr; u[rb][cb][1] *= corr; u[rb][cb][2] *= corr; u[rb][cb][3] *= corr; }
                m_run[rb] = mx;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float pv = __builtin_amdgcn_exp2f(sc[i] - m_run[rb]);
                l_part[rb] += pv;
                const op_t h = (op_t)pv;
                ph[rb][i] = h;
#if LO
                pl[rb][i] = (op_t)(pv - (float)h);
#endif
            }
        }
        // ---- U^T += E^T P^T: every transposed E^T block is read once and multiplies all RB tokens' probabilities
        constexpr int GR = (CB % 4 == 0) ? 4 : 2;
#pragma unroll
        for (int cb0 = 0; cb0 < CB; cb0 += GR) {
            s16x4 r[GR];
#pragma unroll
            for (int j = 0; j < GR; j++) {
                const int c8 = 2 * (wv * CB + cb0 + j) + (p4 >> 1);
                r[j] = tr_read(sbase + xa_off<D>(4 * g + q4, c8) + 8 * (p4 & 1));
            }
            if (GR == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]));
#pragma unroll
            for (int j = 0; j < GR; j++) {
                union { s16x4 s; opx4 v; } ea;
                ea.s = r[j];
#pragma unroll
                for (int rb = 0; rb < RB; rb++) {
                    u[rb][cb0 + j] = mfma16k16(ea.v, ph[rb], u[rb][cb0 + j]);
#if LO
                    u[rb][cb0 + j] = mfma16k16(ea.v, pl[rb], u[rb][cb0 + j]);
#endif
                }
            }
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; rb++) {
        const int tok = tg * RB + rb;
        float l = l_part[rb];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        if (tok < A.T && n16 < A.heads) {
            float *up = A.u_out + (((int64_t)clip * A.T + tok) * 16 + n16) * D;
#pragma unroll
            for (int cb = 0; cb < CB; cb++) {
                f32x4 v = u[rb][cb]; v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv;
                *reinterpret_cast<f32x4 *>(up + 16 * (wv * CB + cb) + 4 * g) = v;
            }
        }
    }
}

template <int D, int NSLOT, int RB>
__global__ __launch_bounds__(256, 1) void k_xattn_rows(XaArgs A) { xattn_rows_body<D, NSLOT, RB>(A); }

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 256, D = argc > 2 ? atoi(argv[2]) : 768, T = argc > 3 ? atoi(argv[3]) : 36, H = D / 64, F = 1500;
    if (D != 768) { printf("this experiment is built for d = 768\n"); return 1; }
    std::vector<op_t> E((size_t)n * F * D), qh((size_t)n * T * 16 * D), ql(qh.size());
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f; };
    for (auto &v : E) v = (op_t)(rnd() * 1.5f);
    for (size_t i = 0; i < qh.size(); i++) {
        const int row = (int)((i / D) % 16);
        const float v = row < H ? rnd() * 0.12f : 0.f;
        qh[i] = (op_t)v; ql[i] = (op_t)(LO ? v - (float)qh[i] : 0.f);
    }
    std::vector<int> klen(n, F);
    if (n > 1) klen[1] = 1473;
    op_t *dE, *dqh, *dql; float *du; int *dk;
    const size_t u_n = (size_t)n * T * 16 * D;
    CK(hipMalloc(&dE, E.size() * 2)); CK(hipMalloc(&dqh, qh.size() * 2)); CK(hipMalloc(&dql, ql.size() * 2));
    CK(hipMalloc(&du, u_n * 4)); CK(hipMalloc(&dk, n * 4));
    CK(hipMemcpy(dE, E.data(), E.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dqh, qh.data(), qh.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dql, ql.data(), ql.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dk, klen.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(du, 0, u_n * 4));
    XaArgs a{dE, (int64_t)F * D, D, dqh, dql, dk, du, H, n, T};
    hipStream_t st; CK(hipStreamCreate(&st));
    constexpr int NSLOT = 3;
    const size_t lds = (size_t)NSLOT * 16 * 768 * 2 + 4096 * RBN;
    CK(hipFuncSetAttribute((const void *)k_xattn_rows<768, NSLOT, RBN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int groups = (T + RBN - 1) / RBN;
    const dim3 grid(n * groups);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_xattn_rows<768, NSLOT, RBN>), grid, dim3(256), lds, st, a);
    CK(hipEventRecord(e0, st));
    const int reps = 10;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_xattn_rows<768, NSLOT, RBN>), grid, dim3(256), lds, st, a);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    std::vector<float> U(u_n);
    CK(hipMemcpy(U.data(), du, u_n * 4, hipMemcpyDeviceToHost));
    double worst = 0; int bad = 0;
    const int cc[] = {0, 1, n - 1}, tt[] = {0, T / 2, T - 1};
    for (int ci = 0; ci < 3; ci++) for (int ti = 0; ti < 3; ti++) {
        const int c = cc[ci], tok = tt[ti]; if (c < 0 || c >= n) continue;
        const int Sk = klen[c];
        for (int h = 0; h < H; h += 5) {
            const size_t qo = (((size_t)c * T + tok) * 16 + h) * D;
            std::vector<double> sc(Sk), R(D, 0.0);
            double mx = -1e300;
            for (int t = 0; t < Sk; t++) {
                double a2 = 0;
                for (int j = 0; j < D; j++) a2 += ((double)(float)qh[qo + j] + (double)(float)ql[qo + j]) * (double)(float)E[((size_t)c * F + t) * D + j];
                sc[t] = a2; mx = std::max(mx, a2);
            }
            double l = 0;
            for (int t = 0; t < Sk; t++) { sc[t] = std::exp2(sc[t] - mx); l += sc[t]; }
            for (int t = 0; t < Sk; t++) { const double p = sc[t] / l; for (int j = 0; j < D; j++) R[j] += p * (double)(float)E[((size_t)c * F + t) * D + j]; }
            double num = 0, den = 0;
            for (int j = 0; j < D; j++) { const double got = U[qo + j]; num += (got - R[j]) * (got - R[j]); den += R[j] * R[j]; }
            const double rel = std::sqrt(num / std::max(den, 1e-30));
            worst = std::max(worst, rel);
            if (!(rel < 5e-3)) { if (bad < 8) printf("clip %d token %d head %d: relative L2 error %.3e\n", c, tok, h, rel); bad++; }
        }
    }
    const double flops = 2.0 * 2.0 * (double)n * T * 16 * D * F * (LO ? 2 : 1);      // S^T and U^T, 16-row blocks (12 of 16 rows live at 12 heads)
    printf("d %d clips %d tokens %d, %d tokens per workgroup (%d passes over E per clip), lo-parts %d: %.1f us per launch = %.2f PFLOP/s of issued MFMA work; "
           "E re-streamed: %.2f GB per launch; worst relative L2 error of U %.2e, %d bad\n", D, n, T, RBN, groups, LO, ms * 1e3, flops / (ms * 1e-3) / 1e15,
           (double)n * groups * F * D * 2 / 1e9, worst, bad);
    return bad ? 1 : 0;
}
