// Cross-attention of ONE query position per (clip, head) computed from the ENCODER OUTPUT itself instead of from projected K / V^T:
//   scores_h[t] = q_h . (Wk_h E[t]) = (q_h Wk_h) . E[t] = Q'_h . E[t]           Q' = [heads][d], one row per head
//   out_h       = sum_t p_h[t] (Wv_h E[t] + bv_h) = Wv_h (sum_t p_h[t] E[t]) + bv_h = Wv_h U_h + bv_h
// so a step reads E (1 500 x d, 2.3 MB per clip at d = 768) ONCE per layer instead of K and V^T (2 x 2.3 MB): half the HBM bytes of the
// kernel that is 72 % of a decoding step, at 12 x the (negligible) MFMA work.  This file is the development bench of the main kernel
// (S^T = E Q'^T and U^T = E^T P^T in one pass over E, online softmax): correctness against a double-precision host computation, then the
// achieved bandwidth for 256 clips.   hipcc -O3 --offload-arch=gfx950 -o bin/xattn_absorb xattn_absorb.hip ;  bin/xattn_absorb [clips] [d]
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

#ifndef CPOL
#define CPOL 0         // cache policy bits of the LDS-DMA loads (gfx950: 1 sc0, 2 nt, 16 sc1)
#endif
#ifndef ABL
#define ABL 0          // ablation bits (timing only, results wrong): 1 no lo-part MFMAs, 2 no U^T product, 4 no DMA, 8 no compute (DMA + barriers only)
#endif

struct XaArgs {
    const op_t *E; int64_t e_clip; int e_ld;        // E + clip * e_clip + t * e_ld (+ column)
    const op_t *qp_hi, *qp_lo;                      // [clip][16][D]: Q' (scaled by log2 e / sqrt(head dim)), rows >= heads zero
    const int *k_len, *skip;
    float *u_part;                                  // [clip][nsplit][16][D]: unnormalised sum_t 2^(s_t - m) E[t] of the split's frames
    float *ml_part;                                 // [clip][nsplit][16][2]: the split's reference m and sum l
    int heads, n, flip, nsplit;
};

// one workgroup = one (clip, split): 16-frame tiles of E through a ring of NSLOT LDS slots, 4 waves; wave w owns d / 4 of the reduction axis of
// S^T = E Q'^T (partials summed through LDS) and d / 4 of the columns of U^T = E^T P^T
template <int D, int NSLOT, int G2>
__device__ __forceinline__ void xattn_absorbed_body(const XaArgs &A)
{
    constexpr int TF = 16, KS = D / 128, CB = D / 64, LPW = D / 128, TILE = TF * D * 2;   // k-steps / column blocks / DMA instructions per wave; tile bytes
    extern __shared__ __attribute__((aligned(16))) char smem[];                          // NSLOT tiles | 4 x 1 KB of partial S^T
    const int bid = A.flip ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int nsw = A.nsplit / G2;                              // splits at workgroup level; a workgroup's G2 groups of 4 waves take alternate tiles of its range
    const int clip = bid / nsw, wsplit = bid - clip * nsw;
    if (A.skip && A.skip[clip]) return;
    const int lane = threadIdx.x & 63, wv8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), grp = wv8 >> 2, wv = wv8 & 3;
    const int split = wsplit * G2 + grp;
    const int n16 = lane & 15, g = lane >> 4;
    const int Sk = A.k_len[clip];
    const int nt_all = (Sk + TF - 1) / TF, per = (nt_all + nsw - 1) / nsw;
#ifdef INTERLEAVE                 // (measured: the splits of a clip taking every nsplit-th tile instead of contiguous ranges changes nothing, 124 against 122 us)
    const int t_lo = split, t_st = A.nsplit, nt = nt_all > split ? (nt_all - split + A.nsplit - 1) / A.nsplit : 0; (void)per;
#else
    const int t_lo = wsplit * per, t_st = 1, t_hi = min(nt_all, t_lo + per), nt = max(t_hi - t_lo, 0);
#endif
    const op_t *eb = A.E + (int64_t)clip * A.e_clip;
    const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(const_cast<op_t *>(eb), 0, ((Sk - 1) * A.e_ld + D) * 2, 0x00020000);
    // Q' fragments of this wave's slice of the d axis (B operand: lane (n16 = head row, g) holds d = 32 kk + 8 g .. + 7)
    opx8 qh[KS], ql[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        const int64_t o = ((int64_t)clip * 16 + n16) * D + 32 * (wv * KS + ks) + 8 * g;
        qh[ks] = *reinterpret_cast<const opx8 *>(A.qp_hi + o);
        ql[ks] = *reinterpret_cast<const opx8 *>(A.qp_lo + o);
    }
    // Q' must have ARRIVED before the ring starts: left to itself the compiler waits for these loads at their first use inside the loop with a
    // vmcnt count that also drains the tile prefetched behind them -- every iteration
#pragma unroll
    for (int ks = 0; ks < KS; ks++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(qh[ks]), "+v"(ql[ks]));
    int voff[LPW];
#pragma unroll
    for (int i = 0; i < LPW; i++) {
        const int X = 64 * (wv8 + 4 * G2 * i) + lane, row = X / (D / 8), ch = (X % (D / 8)) ^ xa_swz(row);
        voff[i] = (row * A.e_ld + 8 * ch) * 2;
    }
    constexpr int NIT = NSLOT / G2;                              // ring depth in iterations (an iteration = G2 consecutive tiles, one per group)
    auto stage = [&](int it) {
        char *slot = smem + (it % NIT) * G2 * TILE;
#pragma unroll
        for (int i = 0; i < LPW; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsE, (__attribute__((address_space(3))) void *)(slot + 1024 * (wv8 + 4 * G2 * i)), 16, voff[i], (t_lo + it * G2) * TF * A.e_ld * 2, 0, CPOL);
    };
    int raddr[KS];                                               // row reads: frame n16, d chunk of k-step wv KS + ks
#pragma unroll
    for (int ks = 0; ks < KS; ks++) raddr[ks] = xa_off<D>(n16, 4 * (wv * KS + ks) + g);
    const int q4 = n16 >> 2, p4 = n16 & 3;
    float *xs = reinterpret_cast<float *>(smem + NSLOT * TILE) + grp * 1024;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 u[CB];
#pragma unroll
    for (int cb = 0; cb < CB; cb++) u[cb] = z4;
    float m_run = -1e30f, l_part = 0.f;
    const int nit = (nt + G2 - 1) / G2;
    for (int it = 0; it < NIT - 1 && it < nit; it++) stage(it);
    for (int it = 0; it < nit; it++) {
        if (NIT >= 3 && it + 1 < nit) __builtin_amdgcn_s_waitcnt(0x0F70 | ((NIT - 2) * LPW)); else __builtin_amdgcn_s_waitcnt(0x0F70 | 0);
        __builtin_amdgcn_s_barrier();
        if (!(ABL & 4) && it + NIT - 1 < nit) stage(it + NIT - 1);
        if (ABL & 8) continue;
        const int t = it * G2 + grp;
        const bool live = t < nt;                                // (wave-uniform; a group without a tile in the last iteration still meets the barriers)
        const char *sC = smem + (it % NIT) * G2 * TILE + grp * TILE;
        const unsigned sbase = lds0 + (unsigned)((it % NIT) * G2 * TILE + grp * TILE);
        // ---- partial S^T over this wave's quarter of d
        if (live) {
            f32x4 a = z4;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const opx8 ef = *reinterpret_cast<const opx8 *>(sC + raddr[ks]);
                a = mfma16(ef, qh[ks], a);
                if (!(ABL & 1)) a = mfma16(ef, ql[ks], a);
            }
            *reinterpret_cast<f32x4 *>(xs + (wv * 64 + lane) * 4) = a;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                      // the partials are in LDS (lgkmcnt 0; the DMA of later tiles stays in flight)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (!live) continue;
        f32x4 sc = z4;
#pragma unroll
        for (int w2 = 0; w2 < 4; w2++) {                         // (every wave adds in the same order: the same bits everywhere)
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xs + (w2 * 64 + lane) * 4);
            sc[0] += v[0]; sc[1] += v[1]; sc[2] += v[2]; sc[3] += v[3];
        }
        const int f0 = (t_lo + t * t_st) * TF;
        if (f0 + TF > Sk) {
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (f0 + 4 * g + i >= Sk) sc[i] = -1e30f;
        }
        float m = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float mx = fmaxf(m_run, m);
        if (__builtin_amdgcn_ballot_w64(mx > m_run) != 0) {
            const float corr = __builtin_amdgcn_exp2f(m_run - mx);
            l_part *= corr;
#pragma unroll
            for (int cb = 0; cb < CB; cb++) { u[cb][0] *= corr; u[cb][1] *= corr; u[cb][2] *= corr; u[cb][3] *= corr; }
            m_run = mx;
        }
        opx4 ph, pl;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float pv = __builtin_amdgcn_exp2f(sc[i] - m_run);
            l_part += pv;
            const op_t h = (op_t)pv;
            ph[i] = h; pl[i] = (op_t)(pv - (float)h);
        }
        // ---- U^T += E^T P^T over this wave's quarter of the columns on v_mfma_f32_16x16x16: A = one transposed 4-frame x 16-column block per column
        // block (lane (column n16, g): frames 4 g .. 4 g + 3), B = this lane's four probabilities as they lie.
        // (ds_read_b64_tr_b16 through asm: the builtin makes the compiler wait for ALL outstanding LDS-DMA -- vmcnt(0) -- before the read, which
        //  would drain the ring every tile; the reads of a group are waited for by hand, the wait carrying them as operands so that no MFMA moves above it)
        constexpr int GR = (CB % 4 == 0) ? 4 : 2;
#pragma unroll
        for (int cb0 = 0; cb0 < ((ABL & 2) ? 0 : CB); cb0 += GR) {
            s16x4 r[GR];
#pragma unroll
            for (int j = 0; j < GR; j++) {
                const int c8 = 2 * (wv * CB + cb0 + j) + (p4 >> 1);   // 16-byte chunk that holds columns 16 (wv CB + cb) + 4 p4 .. + 3
                r[j] = tr_read(sbase + xa_off<D>(4 * g + q4, c8) + 8 * (p4 & 1));
            }
            if (GR == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]));
#pragma unroll
            for (int j = 0; j < GR; j++) {
                union { s16x4 s; opx4 v; } ea;
                ea.s = r[j];
                u[cb0 + j] = mfma16k16(ea.v, ph, u[cb0 + j]);
                if (!(ABL & 1)) u[cb0 + j] = mfma16k16(ea.v, pl, u[cb0 + j]);
            }
        }
    }
    float l = l_part;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (n16 < A.heads) {
        float *up = A.u_part + (((int64_t)clip * A.nsplit + split) * 16 + n16) * D;
#pragma unroll
        for (int cb = 0; cb < CB; cb++) *reinterpret_cast<f32x4 *>(up + 16 * (wv * CB + cb) + 4 * g) = u[cb];
        if (wv == 0 && g == 0) { float *mp = A.ml_part + (((int64_t)clip * A.nsplit + split) * 16 + n16) * 2; mp[0] = m_run; mp[1] = l; }
    }
}

template <int D, int NSLOT, int G2>
__global__ __launch_bounds__(256 * G2, G2 == 1 ? 2 : 1) void k_xattn_absorbed(XaArgs A) { xattn_absorbed_body<D, NSLOT, G2>(A); }

template <int D, int NSLOT, int G2 = 1>
static float run(const XaArgs &a, int reps, hipStream_t st)
{
    const size_t lds = (size_t)NSLOT * 16 * D * 2 + 4096 * G2;
    CK(hipFuncSetAttribute((const void *)k_xattn_absorbed<D, NSLOT, G2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    XaArgs b = a;
    const dim3 grid(a.n * a.nsplit / G2);
    for (int i = 0; i < 3; i++) { b.flip = (i & 1) & a.flip; hipLaunchKernelGGL((k_xattn_absorbed<D, NSLOT, G2>), grid, dim3(256 * G2), lds, st, b); }
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; i++) { b.flip = (i & 1) & a.flip; hipLaunchKernelGGL((k_xattn_absorbed<D, NSLOT, G2>), grid, dim3(256 * G2), lds, st, b); }
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    b.flip = 0; hipLaunchKernelGGL((k_xattn_absorbed<D, NSLOT, G2>), grid, dim3(256 * G2), lds, st, b); CK(hipStreamSynchronize(st));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 256, D = argc > 2 ? atoi(argv[2]) : 768, H = D / 64, T = 1500;
    const int nsplit = argc > 3 ? atoi(argv[3]) : 2, flipmode = argc > 4 ? atoi(argv[4]) : 1;
    std::vector<op_t> E((size_t)n * T * D), qh((size_t)n * 16 * D), ql((size_t)n * 16 * D);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f; };
    for (auto &v : E) v = (op_t)(rnd() * 1.5f);
    for (size_t i = 0; i < qh.size(); i++) {
        const int row = (int)((i / D) % 16);
        const float v = row < H ? rnd() * 0.12f : 0.f;           // scores of a few units: a peaked softmax
        qh[i] = (op_t)v; ql[i] = (op_t)(v - (float)qh[i]);
    }
    std::vector<int> klen(n, T);
    if (n > 3) { klen[1] = 1473; klen[2] = 31; klen[3] = 1; }
    op_t *dE, *dqh, *dql; float *dup, *dml; int *dk;
    const size_t up_n = (size_t)n * nsplit * 16 * D, ml_n = (size_t)n * nsplit * 16 * 2;
    CK(hipMalloc(&dE, E.size() * 2)); CK(hipMalloc(&dqh, qh.size() * 2)); CK(hipMalloc(&dql, ql.size() * 2));
    CK(hipMalloc(&dup, up_n * 4)); CK(hipMalloc(&dml, ml_n * 4)); CK(hipMalloc(&dk, n * 4));
    CK(hipMemcpy(dE, E.data(), E.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dqh, qh.data(), qh.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dql, ql.data(), ql.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dk, klen.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dup, 0, up_n * 4)); CK(hipMemset(dml, 0, ml_n * 4));
    XaArgs a{dE, (int64_t)T * D, D, dqh, dql, dk, nullptr, dup, dml, H, n, flipmode, nsplit};
    hipStream_t st; CK(hipStreamCreate(&st));
    const int reps = 40;
    float ms = 0;
    if (D == 384) ms = run<384, 3>(a, reps, st);
    else if (D == 512) ms = run<512, 3>(a, reps, st);
#ifdef GRP2
    else if (D == 768) ms = run<768, 6, 2>(a, reps, st);
#elif defined(NSL)
    else if (D == 768) ms = run<768, NSL>(a, reps, st);
#else
    else if (D == 768) ms = run<768, 3>(a, reps, st);
#endif
    else if (D == 1024) ms = run<1024, 2>(a, reps, st);
    else { printf("d = %d not built\n", D); return 1; }
    std::vector<float> up(up_n), ml(ml_n);
    CK(hipMemcpy(up.data(), dup, up_n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ml.data(), dml, ml_n * 4, hipMemcpyDeviceToHost));
    // host check on a few clips (the merge of the splits is what the consumer of the partials does)
    double worst = 0; int bad = 0;
    const int check[] = {0, 1, 2, 3, n - 1};
    for (int ci = 0; ci < 5; ci++) {
        const int c = check[ci]; if (c < 0 || c >= n) continue;
        const int Sk = klen[c];
        for (int h = 0; h < H; h++) {
            std::vector<double> sc(Sk), U(D, 0.0);
            double mx = -1e300;
            for (int t = 0; t < Sk; t++) {
                double a2 = 0;
                for (int j = 0; j < D; j++) a2 += ((double)(float)qh[((size_t)c * 16 + h) * D + j] + (double)(float)ql[((size_t)c * 16 + h) * D + j]) * (double)(float)E[((size_t)c * T + t) * D + j];
                sc[t] = a2; mx = std::max(mx, a2);
            }
            double l = 0;
            for (int t = 0; t < Sk; t++) { sc[t] = std::exp2(sc[t] - mx); l += sc[t]; }
            for (int t = 0; t < Sk; t++) { const double p = sc[t] / l; for (int j = 0; j < D; j++) U[j] += p * (double)(float)E[((size_t)c * T + t) * D + j]; }
            double mm = -1e300, lt = 0;
            for (int sp = 0; sp < nsplit; sp++) mm = std::max(mm, (double)ml[(((size_t)c * nsplit + sp) * 16 + h) * 2]);
            for (int sp = 0; sp < nsplit; sp++) { const float *q = &ml[(((size_t)c * nsplit + sp) * 16 + h) * 2]; lt += (double)q[1] * std::exp2((double)q[0] - mm); }
            double num = 0, den = 0;
            for (int j = 0; j < D; j++) {
                double got = 0;
                for (int sp = 0; sp < nsplit; sp++) got += (double)up[(((size_t)c * nsplit + sp) * 16 + h) * D + j] * std::exp2((double)ml[(((size_t)c * nsplit + sp) * 16 + h) * 2] - mm);
                got /= lt;
                num += (got - U[j]) * (got - U[j]); den += U[j] * U[j];
            }
            const double rel = std::sqrt(num / std::max(den, 1e-30));
            worst = std::max(worst, rel);
            if (!(rel < 2e-3)) { if (bad < 8) printf("clip %d head %d: relative L2 error %.3e (|U| %.3e)\n", c, h, rel, std::sqrt(den)); bad++; }
        }
    }
    double live = 0; for (int c = 0; c < n; c++) live += (double)klen[c] * D * 2;
    printf("d %d heads %d clips %d splits %d: %.2f us per launch, %.1f MB of E per launch -> %.2f TB/s; worst relative L2 error of U %.2e, %d bad\n", D, H, n, nsplit, ms * 1e3,
           live / 1e6, live / (ms * 1e-3) / 1e12, worst, bad);
    return bad ? 1 : 0;
}
