// pce_whisper.hip -- the transformer work of the alignment step (R8) on gfx950: log-mel spectrogram, Whisper audio encoder,
// text decoder (teacher-forced forced alignment with cross-attention DTW; free-running greedy decoding with a K / V
// cache and openai-whisper's logit filters), and the break-prediction BERT forward, which shares the same kernels.
//
// Replaces the device work of whisper_timestamped.transcribe
// (Code/Aligners/use_whisper_timestamped.py:139,150-163), as openai-whisper==20240930 defines it (third-party,
// restated from its published architecture; the torch fp32 restatement in oracle/ is pinned against the installed
// transformers port of the same network: tests/test_whisper_hf_crosscheck.py):
//   log_mel_spectrogram: Hann(400) STFT hop 160 (centre, reflect padding), |.|^2, 80 Slaney
//     mel bands, log10(max(.,1e-10)), max(., max-8), (.+4)/4, 30 s window = 3000 frames
//   AudioEncoder: conv1d(80->d,k3,p1)+GELU, conv1d(d->d,k3,s2,p1)+GELU, + sinusoid positions,
//     L x { x += proj(MHA(LN(x)));  x += W2 gelu(W1 LN(x)) },  LN
//
// Execution plan:
//   k_logmel_frames  one wavefront per frame, 400-point DFT as 25 x 16 Cooley-Tukey in LDS (fp32),
//                    sparse mel bands, per-clip maximum by ordered-int atomicMax
//   k_logmel_norm    clamp/scale, writes float [80][3000] and the bf16 time-major padded image
//                    [3002][80] the first convolution reads as an implicit im2col GEMM operand
//   k_gemm_bf16      C = A B^T on v_mfma_f32_16x16x32_bf16: 128x128x64 tiles, 4 waves x (64x64),
//                    fp32 accumulate; operands stream global -> LDS with global_load_lds (16 B/lane),
//                    double buffered behind the MFMAs, XOR-swizzled chunks (conflict-free
//                    ds_read_b128), XCD-aware tile order, fused epilogues (bias, exact GELU,
//                    positional add, residual accumulate).
//                    Both convolutions are this GEMM with overlapping A rows (lda < K): the
//                    activations are time-major with zero pad rows, so "im2col" is just a stride.
//   k_layernorm      one wavefront per row, fp32 statistics
//   k_attention      flash-style forward per (clip, head, 64 queries): QK^T and PV on MFMA,
//                    online softmax in registers with DPP row reductions, K / V^T tiles in LDS
// The residual stream is fp32, every GEMM operand bf16.  Roofline: MFMA (dense bf16).
#include "pce_internal.h"
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __bf16 bf16;

constexpr int W_NFFT = 400, W_HOP = 160, W_BINS = 201, W_FRAMES = 3000, W_SAMPLES = 480000;
constexpr int W_CTX = 1500;
constexpr double W_PI = 3.14159265358979323846;

// ---------------------------------------------------------------------------
// log-mel
// ---------------------------------------------------------------------------
struct MelTables {            // device pointers
    const float2 *w16;        // [16]      exp(-2 pi i m / 16)
    const float2 *w400;       // [25][16]  exp(-2 pi i n2 k1 / 400)
    const float2 *w25;        // [25]      exp(-2 pi i m / 25)
    const float *window;      // [400]     periodic Hann
    const int *mel_lo;        // [n_mels]  first bin of band
    const int *mel_n;         // [n_mels]  number of bins
    const int *mel_off;       // [n_mels]  offset into mel_w
    const float *mel_w;       // packed weights
};

__device__ __forceinline__ unsigned int f32_order_key(float f)
{
    const unsigned int b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_from_key(unsigned int k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// frame0 == nullptr: the window is frames 0..2999 of the clip trimmed to 30 s (pad_or_trim + log_mel_spectrogram).
// frame0 != nullptr: whisper.transcribe's slicing of the log-mel of the WHOLE clip followed by 30 s of zeros: the window is
// frames frame0[clip] .. +2999, samples beyond 30 s exist, and (scan != 0) a first launch reduces the maximum over every
// frame that overlaps audio (the frames of pure padding are at the -10 floor) without writing anything.
__global__ __launch_bounds__(256) void k_logmel_frames(const int16_t *__restrict__ pcm, const int64_t *__restrict__ clip_off, int n_mels,
                                                      MelTables T, float *__restrict__ logspec /* [clip][n_mels][3000] */,
                                                      unsigned int *__restrict__ clip_max, const int64_t *__restrict__ frame0, int scan)
{
    __shared__ float xs[4][W_NFFT];
    __shared__ float2 ys[4][25 * 16];
    __shared__ float pw[4][W_BINS + 3];
    __shared__ float2 s_w16[16], s_w25[25], s_w400[25 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 16; i += 256) s_w16[i] = T.w16[i];
    for (int i = tid; i < 25; i += 256) s_w25[i] = T.w25[i];
    for (int i = tid; i < 400; i += 256) s_w400[i] = T.w400[i];
    __syncthreads();
    const int clip = blockIdx.y;
    const int64_t base = clip_off[clip], full = clip_off[clip + 1] - base;
    const int64_t len = frame0 ? full : min<int64_t>(full, (int64_t)W_SAMPLES);
    const int64_t f_begin = (frame0 && !scan) ? frame0[clip] : 0;
    const int64_t n_fr = scan ? (full + W_NFFT / 2) / W_HOP + 1 : W_FRAMES;       // scan: every frame that can see a sample
    float vmax = -1e30f;
    for (int64_t fl = blockIdx.x * 4 + wv; fl < n_fr; fl += gridDim.x * 4) {
        const int64_t frame = f_begin + fl;
        if (frame * W_HOP - W_NFFT / 2 >= len) {
            // a frame of the zero padding behind the audio (two thirds of the 30 s window of a 10 s clip): every sample is an exact zero, so are
            // the spectrum and the mel energies, and the value is log10 of the 1e-10 clamp -- written without running the transform
            const float lg = log10f(fmaxf(0.f, 1e-10f));
            for (int b = lane; b < n_mels; b += 64)
                if (!scan) logspec[((int64_t)clip * n_mels + b) * W_FRAMES + fl] = lg;
            vmax = fmaxf(vmax, lg);
            continue;
        }
        // windowed frame, centre = frame*160, reflect padding at the start, zeros past the audio
        for (int n = lane; n < W_NFFT; n += 64) {
            int64_t i = frame * W_HOP - W_NFFT / 2 + n;
            if (i < 0) i = -i;
            const float v = (i < len) ? (float)pcm[base + i] * (1.0f / 32768.0f) : 0.0f;
            xs[wv][n] = v * T.window[n];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        // stage A: Y[n2][k1] = sum_{n1<16} x[25 n1 + n2] W16^{n1 k1}, then twiddle W400^{n2 k1}
        for (int o = lane; o < 400; o += 64) {
            const int n2 = o >> 4, k1 = o & 15;
            float re = 0.f, im = 0.f;
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                const float x = xs[wv][25 * n1 + n2];
                const float2 w = s_w16[(n1 * k1) & 15];
                re = fmaf(x, w.x, re); im = fmaf(x, w.y, im);
            }
            const float2 t = s_w400[o];
            ys[wv][o] = make_float2(re * t.x - im * t.y, re * t.y + im * t.x);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        // stage B: X[k1 + 16 k2] = sum_{n2<25} Y[n2][k1] W25^{n2 k2}; only k <= 200 is needed
        for (int k = lane; k < W_BINS; k += 64) {
            const int k1 = k & 15, k2 = k >> 4;
            float re = 0.f, im = 0.f;
            int m = 0;
            for (int n2 = 0; n2 < 25; n2++) {
                const float2 y = ys[wv][n2 * 16 + k1];
                const float2 w = s_w25[m];
                re += y.x * w.x - y.y * w.y; im += y.x * w.y + y.y * w.x;
                m += k2; if (m >= 25) m -= 25;
            }
            pw[wv][k] = re * re + im * im;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        for (int b = lane; b < n_mels; b += 64) {
            const int lo = T.mel_lo[b], cnt = T.mel_n[b];
            const float *w = T.mel_w + T.mel_off[b];
            float acc = 0.f;
            for (int i = 0; i < cnt; i++) acc = fmaf(w[i], pw[wv][lo + i], acc);
            const float lg = log10f(fmaxf(acc, 1e-10f));
            if (!scan) logspec[((int64_t)clip * n_mels + b) * W_FRAMES + fl] = lg;
            vmax = fmaxf(vmax, lg);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
    for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
    if (lane == 0) atomicMax(clip_max + clip, f32_order_key(vmax));
}

// out_f32 [clip][n_mels][3000];  out_tm bf16 [clip][3002][n_mels] (rows 0 and 3001 stay zero)
__global__ __launch_bounds__(256) void k_logmel_norm(float *__restrict__ logspec, const unsigned int *__restrict__ clip_max, int n_mels,
                                                    bf16 *__restrict__ out_tm)
{
    __shared__ float tile[64][65];
    const int clip = blockIdx.z, f0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    const float floor_v = f32_from_key(clip_max[clip]) - 8.0f;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int b = b0 + r, f = f0 + tx;
        if (b < n_mels && f < W_FRAMES) {
            float *p = logspec + ((int64_t)clip * n_mels + b) * W_FRAMES + f;
            const float v = (fmaxf(*p, floor_v) + 4.0f) / 4.0f;
            *p = v; tile[r][tx] = v;
        }
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int f = f0 + r, b = b0 + tx;
        if (b < n_mels && f < W_FRAMES) out_tm[((int64_t)clip * (W_FRAMES + 2) + f + 1) * n_mels + b] = (bf16)tile[tx][r];
    }
}

// ---------------------------------------------------------------------------
// GEMM  C[M x N] = A[M x K] * B[N x K]^T  (bf16 in, fp32 accumulate)
// ---------------------------------------------------------------------------
enum { EPI_BF16 = 0, EPI_GELU_BF16 = 1, EPI_GELU_POS_F32 = 2, EPI_RESID_F32 = 3, EPI_QKV = 4 };
constexpr int AT_SP = 1536;               // padded key axis of the transposed V image (multiple of the 64-key tile)
constexpr int G_BM = 128, G_BN = 128, G_BK = 64;

// GELU(x) = x/2 (1 + erf(x/sqrt 2)) = x/2 + |x|/2 erf(|x|/sqrt 2) (erf is odd: no sign select); erf by Abramowitz-Stegun 7.1.26,
// erf(z) = 1 - t P(t) exp(-z^2), t = 1/(1 + p z), |error| < 1.5e-7 -- far below the bf16 step of the outputs (libm's erff costs more than
// the tile's MFMAs in a K = 768 epilogue).  With z' = z sqrt(log2 e) the exponential is a bare v_exp_f32 of -z'^2 and p z = (p / sqrt(log2 e)) z'.
// gelu_exact8 is the same operation sequence on eight values, written level by level: the eight dependency chains interleave, so the packed
// fp32 instructions the compiler forms need no wait states between them (the one-value form in a loop cost 18 issue slots per value, 4 of
// them s_nop; this costs 11).
constexpr float GELU_C = 0.70710678118654752440f * 1.2011224087864498f;     // |x| -> z' = |x| / sqrt 2 * sqrt(log2 e)
constexpr float GELU_P = 0.3275911f / 1.2011224087864498f;
__device__ __forceinline__ float gelu_exact(float x)
{
    const float zp = fabsf(x) * GELU_C;
    const float t = __builtin_amdgcn_rcpf(fmaf(zp, GELU_P, 1.0f));         // v_rcp_f32 (1 ulp); __frcp_rn expands to a 10-instruction IEEE division
    const float e = __builtin_amdgcn_exp2f(zp * -zp);
    const float poly = fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float erf_abs = fmaf(-(t * poly), e, 1.0f);
    const float h = 0.5f * x;
    return fmaf(fabsf(h), erf_abs, h);
}
__device__ __forceinline__ void gelu_exact8(float (&x)[8])
{
    float zp[8], t[8], e[8], poly[8], h[8];
#pragma unroll
    for (int i = 0; i < 8; i++) zp[i] = fabsf(x[i]) * GELU_C;
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = __builtin_amdgcn_rcpf(fmaf(zp[i], GELU_P, 1.0f));
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = __builtin_amdgcn_exp2f(zp[i] * -zp[i]);
#pragma unroll
    for (int i = 0; i < 8; i++) poly[i] = fmaf(t[i], 1.061405429f, -1.453152027f);
#pragma unroll
    for (int i = 0; i < 8; i++) poly[i] = fmaf(t[i], poly[i], 1.421413741f);
#pragma unroll
    for (int i = 0; i < 8; i++) poly[i] = fmaf(t[i], poly[i], -0.284496736f);
#pragma unroll
    for (int i = 0; i < 8; i++) poly[i] = fmaf(t[i], poly[i], 0.254829592f);
#pragma unroll
    for (int i = 0; i < 8; i++) poly[i] = t[i] * poly[i];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = fmaf(-poly[i], e[i], 1.0f);
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = 0.5f * x[i];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = fmaf(fabsf(h[i]), e[i], h[i]);
}

// LDS image of one operand tile: 128 rows x 64 bf16 (128 B = eight 16-byte chunks per row), rows
// contiguous (what global_load_lds writes: wave-uniform base + lane * 16 B).  To keep the fragment
// reads (16 rows x one chunk per ds_read_b128 lane group) conflict-free the chunk index is
// XOR-swizzled with (row >> 1) & 7: applied to the GLOBAL source address when staging and to the
// LDS address when reading (the same involution on both sides).
__device__ __forceinline__ int swz_chunk(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

constexpr int G_THREADS = 512;
constexpr int G_TLD = G_BN + 4;           // fp32 epilogue tile row (pad keeps the accumulator scatter conflict-free)

// one operand tile (128 rows x 64 k): 16 wave-instructions of 1 KiB (8 rows each); 8 waves -> 2 each
__device__ __forceinline__ void stage_tile(const bf16 *__restrict__ G, int64_t ld, int row0, int row_max, int k0, bf16 *lds_tile, int wv, int lane)
{
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int r0 = wv * 16 + i * 8;
        const int row = r0 + (lane >> 3);
        const int c = swz_chunk(row, lane & 7);
        int gr = row0 + row; if (gr > row_max) gr = row_max;
        const bf16 *src = G + (int64_t)gr * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds_tile + r0 * G_BK), 16, 0, 0);
    }
}

template <int EPI>
__device__ __forceinline__ void gemm_fetch_bias(const float *__restrict__ bias, int n0, int tid, float (&bv)[8], float &bv_col)
{
    constexpr bool OUT_BF16 = (EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_QKV);
    const int cx0 = OUT_BF16 ? (tid & 15) * 8 : (tid & 31) * 4;
#pragma unroll
    for (int e = 0; e < 8; e++) bv[e] = (bias && e < (OUT_BF16 ? 8 : 4)) ? bias[n0 + cx0 + e] : 0.f;
    bv_col = (EPI == EPI_QKV && bias) ? bias[n0 + (tid & 127)] : 0.f;      // transposed-V path: one column per thread
}

// Output stage shared by the GEMM kernels: a [128][G_TLD] fp32 tile in LDS (rows m0.., columns n0..) leaves as full
// 256-byte row segments with the fused bias / GELU / positional add / residual accumulate / transposed-V write.
// bv / bv_col: this thread's bias values for the tile's columns, fetched by the caller before its K loop.
template <int EPI>
__device__ __forceinline__ void gemm_store_tile(const float *tile, int tid, int m0, int n0, int M, int N, const float (&bv)[8], float bv_col,
                                                void *__restrict__ Cv, int64_t ldc, int64_t cbase, const float *__restrict__ pos, int pos_T,
                                                int v_col0, int vt_sp)
{
    if (EPI == EPI_QKV && n0 >= v_col0) {
        // V columns: written transposed, vt[clip][head][d][key], so the attention kernel can stage V^T tiles
        // (8 keys contiguous per d) without an LDS transpose.  `pos` carries the vt pointer, pos_T = S.
        bf16 *vt = reinterpret_cast<bf16 *>(const_cast<float *>(pos));
        const int dmodel = N - v_col0, S = pos_T;                   // V is the last d_model columns
        const int cc = tid & 127, rg = tid >> 7;                  // one column, 8-row groups
        const int n = n0 + cc - v_col0, head = n >> 6, dd = n & 63;
        const float bv = bv_col;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int row = rg * 8 + 32 * p;
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {                      // halves of 4 tokens never straddle a clip (S % 4 == 0)
                const int m = m0 + row + 4 * hf;
                if (m >= M) continue;
                const int clip = m / S, t = m - clip * S;
                typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = (bf16)(tile[(row + 4 * hf + e) * G_TLD + cc] + bv);
                *reinterpret_cast<bf16x4 *>(vt + (((int64_t)clip * (dmodel >> 6) + head) * 64 + dd) * vt_sp + t) = o;
            }
        }
    } else if (EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_QKV) {
        // 16 threads x 8 columns per row, 32 rows per pass
        const int cx = (tid & 15) * 8, ry = tid >> 4;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int row = ry + 32 * p;
            if (m0 + row >= M) continue;
            const float4 v0 = *reinterpret_cast<const float4 *>(&tile[row * G_TLD + cx]);
            const float4 v1 = *reinterpret_cast<const float4 *>(&tile[row * G_TLD + cx + 4]);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float t = v[e] + bv[e];
                o[e] = (bf16)(EPI == EPI_GELU_BF16 ? gelu_exact(t) : t);
            }
            // written once, read by the next kernel: keep it out of the way of the operand tiles in L2
            __builtin_nontemporal_store(o, reinterpret_cast<bf16x8 *>(reinterpret_cast<bf16 *>(Cv) + cbase + (int64_t)(m0 + row) * ldc + n0 + cx));
        }
    } else {
        // fp32 outputs: 32 threads x 4 columns per row, 16 rows per pass
        const int cx = (tid & 31) * 4, ry = tid >> 5;
        float4 oldv[8];
        if (EPI == EPI_RESID_F32) {
#pragma unroll
            for (int p = 0; p < 8; p++) {                         // all eight residual loads in flight together
                const int row = ry + 16 * p;
                oldv[p] = m0 + row < M ? *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(Cv) + cbase + (int64_t)(m0 + row) * ldc + n0 + cx)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const int row = ry + 16 * p;
            if (m0 + row >= M) continue;
            const float4 v = *reinterpret_cast<const float4 *>(&tile[row * G_TLD + cx]);
            float4 *dst = reinterpret_cast<float4 *>(reinterpret_cast<float *>(Cv) + cbase + (int64_t)(m0 + row) * ldc + n0 + cx);
            float4 o;
            if (EPI == EPI_GELU_POS_F32) {
                const float4 pe = *reinterpret_cast<const float4 *>(pos + (int64_t)((m0 + row) % pos_T) * N + n0 + cx);
                o = make_float4(gelu_exact(v.x + bv[0]) + pe.x, gelu_exact(v.y + bv[1]) + pe.y, gelu_exact(v.z + bv[2]) + pe.z,
                                gelu_exact(v.w + bv[3]) + pe.w);
            } else {
                const float4 old = oldv[p];
                o = make_float4(old.x + v.x + bv[0], old.y + v.y + bv[1], old.z + v.z + bv[2], old.w + v.w + bv[3]);
            }
            *dst = o;
        }
    }
}

// 8 wavefronts as 2 (M) x 4 (N), each owning a 64 x 32 slice of the 128 x 128 tile.  Three LDS stages:
// while tile kt is multiplied, tiles kt+1 and kt+2 are in flight as LDS-DMA (4 instructions per wave and
// tile), so a tile has two full K-steps to arrive.  The barrier is a raw s_barrier behind a COUNTED
// s_waitcnt vmcnt(4): __syncthreads() would drain the DMA queue (vmcnt(0)) and serialise the ring.
// G_STAGES = 2: two workgroups per CU cover each other's waits (big grids).  G_STAGES = 4 (round 3): the few-row launches of an
// incremental decoding step (M = clips: 2 x N/128 workgroups on 256 CUs) are alone on their CU and each K-step waited out a full
// L2 / HBM round trip (12 K-steps x 1.5 us = the 24 us such a launch took); with three tiles in flight the K loop runs at the
// DMA issue rate instead.
template <int EPI, int G_STAGES = 2>
__global__ __launch_bounds__(G_THREADS, G_STAGES == 2 ? 4 : 2) void k_gemm_bf16(const bf16 *__restrict__ A, int64_t lda, int64_t a_batch,
                                                        const bf16 *__restrict__ B, int M, int N, int K,
                                                        const float *__restrict__ bias, void *__restrict__ Cv, int64_t ldc, int64_t c_batch,
                                                        const float *__restrict__ pos, int pos_T, int v_col0, int vt_sp, int sn_tiles, int sm_tiles, unsigned long long *trace)
{
    // operand ring [stage][A|B][128][64] bf16, re-used as the fp32 epilogue tile [128][G_TLD]
    constexpr int SMEM_ELEMS = (G_STAGES * 2 * G_BM * G_BK * 2 > G_BM * G_TLD * 4 ? G_STAGES * 2 * G_BM * G_BK : G_BM * G_TLD * 2);
    __shared__ __attribute__((aligned(1024))) bf16 smem[SMEM_ELEMS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 2, wc = wv & 3;
    // XCD-aware, L2-sized tile order.  Workgroups go to the 8 XCDs round-robin, so XCD x is given a contiguous range of
    // the tile sequence, and the ~64 workgroups resident on an XCD at a time are 64 consecutive tiles of it.  The
    // sequence walks SUPERTILES of sm (M) x sn (N) tiles: their operands (sm A row blocks + sn B row blocks of 128 x K)
    // fit the XCD's 4 MB L2 and every block is re-read 8 times from it.  (A plain row-major sequence keeps 2-3 A blocks
    // and ALL of B live: at N = 3072 that is 4.7 MB of weights, which evicts itself; measured L2 hit rate 50 %.)
    const int tiles_n = (int)gridDim.x, tiles_m_pad = (int)gridDim.y;          // gridDim.y is padded to a multiple of 8
    int lin = (int)blockIdx.y * tiles_n + (int)blockIdx.x;
    const int total = tiles_n * tiles_m_pad;
    if ((total & 7) == 0) lin = (lin & 7) * (total >> 3) + (lin >> 3);
    int m0, n0;
    {
        const int per = sm_tiles * sn_tiles, sup = lin / per, r = lin - sup * per;
        const int n_sn = tiles_n / sn_tiles;
        const int tm = (sup / n_sn) * sm_tiles + r / sn_tiles, tn = (sup % n_sn) * sn_tiles + r % sn_tiles;
        m0 = tm * G_BM; n0 = tn * G_BN;
        if (m0 >= M) return;                                                     // padding tile
    }
    const unsigned long long t_start = trace ? __builtin_amdgcn_s_memtime() : 0;
    // the epilogue's bias values are fetched now: after the K loop their load latency (1-2 us) was fully exposed
    float bv[8]; float bv_col;
    gemm_fetch_bias<EPI>(bias, n0, tid, bv, bv_col);
    A += (int64_t)blockIdx.z * a_batch;
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    const int nk = K / G_BK;
    constexpr int STAGE = 2 * G_BM * G_BK;
    // prologue: tiles 0 .. G_STAGES - 2 (every tile is 4 DMA instructions per wave; a tile past the end of K is issued anyway -- the last
    // tile again, into a slot nobody reads -- so that the counted waits below are the same in every iteration)
#pragma unroll
    for (int t = 0; t < G_STAGES - 1; t++) {
        const int tt = t < nk ? t : nk - 1;
        stage_tile(A, lda, m0, M - 1, tt * G_BK, smem + t * STAGE, wv, lane);
        stage_tile(B, K, n0, N - 1, tt * G_BK, smem + t * STAGE + G_BM * G_BK, wv, lane);
    }
    for (int kt = 0; kt < nk; kt++) {
        const bf16 *sA = smem + (kt % G_STAGES) * STAGE, *sB = sA + G_BM * G_BK;
        // tile kt has landed once at most the DMAs of the G_STAGES - 2 younger tiles of this wave are outstanding
        __builtin_amdgcn_s_waitcnt(0x0F70 | (((4 * (G_STAGES - 2)) >> 4) << 14) | ((4 * (G_STAGES - 2)) & 15));
        __builtin_amdgcn_s_barrier();
        if (G_STAGES > 2 || kt + G_STAGES - 1 < nk) {        // refill the stage tile kt-1 has just released (deep ring: always, see the prologue)
            const int tn = kt + G_STAGES - 1 < nk ? kt + G_STAGES - 1 : nk - 1;
            bf16 *nA = smem + ((kt + G_STAGES - 1) % G_STAGES) * STAGE;
            stage_tile(A, lda, m0, M - 1, tn * G_BK, nA, wv, lane);
            stage_tile(B, K, n0, N - 1, tn * G_BK, nA + G_BM * G_BK, wv, lane);
        }
#pragma unroll
        for (int kk = 0; kk < G_BK; kk += 32) {
            bf16x8 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = wr * 64 + i * 16 + fr;
                a[i] = *reinterpret_cast<const bf16x8 *>(&sA[row * G_BK + swz_chunk(row, (kk >> 3) + fq) * 8]);
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int row = wc * 32 + j * 16 + fr;
                b[j] = *reinterpret_cast<const bf16x8 *>(&sB[row * G_BK + swz_chunk(row, (kk >> 3) + fq) * 8]);
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    const unsigned long long t_loop = trace ? __builtin_amdgcn_s_memtime() : 0;
    // epilogue.  The accumulator layout (col = lane & 15, row = (lane >> 4) * 4 + reg) would store 2-byte
    // elements 32 B at a time; measured, such an epilogue cost more than the whole K loop.  The tile
    // goes through LDS instead (the operand ring is free now) and leaves as full 256-B row segments.
    if (G_STAGES > 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the surplus refills of the deep ring must have landed before the ring is reused
    __builtin_amdgcn_s_barrier();                           // every wave is done reading the operand stages
    float *tile = reinterpret_cast<float *>(smem);           // [128][G_TLD] fp32 = 66 KiB
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                tile[(wr * 64 + i * 16 + fq * 4 + r) * G_TLD + wc * 32 + j * 16 + fr] = acc[i][j][r];
    __syncthreads();
    const unsigned long long t_tile = trace ? __builtin_amdgcn_s_memtime() : 0;
    gemm_store_tile<EPI>(tile, tid, m0, n0, M, N, bv, bv_col, Cv, ldc, (int64_t)blockIdx.z * c_batch, pos, pos_T, v_col0, vt_sp);
    if (trace && tid == 0) {
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        unsigned long long *o = trace + 4 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        o[0] = t_start; o[1] = t_loop; o[2] = t_tile; o[3] = t_end;
    }
}

// ---------------------------------------------------------------------------
// 128 x 256 tile variant for wide outputs (N >= 2048: QKV, fc1).  Skipping the A-operand loads of the 128 x 128 kernel
// speeds the encoder up by 36 %, skipping the B loads by 12 % (ablation): the ACTIVATION stream is what these launches
// wait for.  A tile twice as wide reads every A row block half as often; the weights (B) stay L2-resident anyway.
// 8 waves as 2 (M) x 4 (N), 64 x 64 per wave, 32-deep K-steps (16 MFMAs per wave between barriers, as before), two
// 24 KB stages so that two workgroups still share a CU, and the output leaves through the same 128 x 128 LDS tile
// in two column passes.
// ---------------------------------------------------------------------------
constexpr int W_BN = 256, W_BK = 32, W_STAGES = 2;   // a third stage (72 KB, still two workgroups per CU) measured +0.2 %: not latency bound
constexpr int W_STAGE = (G_BM + W_BN) * W_BK;                   // bf16 elements per stage (24 KB)
__device__ __forceinline__ int swz32(int row, int chunk) { return chunk ^ ((row >> 2) & 3); }   // 64-byte rows (see swz_chunk)
// ROWS x 32 k operand tile: wave-instructions of 1 KiB (16 rows each)
template <int ROWS>
__device__ __forceinline__ void stage_rows32(const bf16 *__restrict__ G, int64_t ld, int row0, int row_max, int k0, bf16 *lds_tile, int wv, int lane)
{
#pragma unroll
    for (int i = 0; i < ROWS / 128; i++) {
        const int r0 = (wv + 8 * i) * 16;
        const int row = r0 + (lane >> 2);
        const int c = swz32(row, lane & 3);
        int gr = row0 + row; if (gr > row_max) gr = row_max;
        const bf16 *src = G + (int64_t)gr * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds_tile + r0 * W_BK), 16, 0, 0);
    }
}

template <int EPI>
__global__ __launch_bounds__(G_THREADS, 4) void k_gemm_wide(const bf16 *__restrict__ A, int64_t lda, int64_t a_batch,
                                                        const bf16 *__restrict__ B, int M, int N, int K,
                                                        const float *__restrict__ bias, void *__restrict__ Cv, int64_t ldc, int64_t c_batch,
                                                        const float *__restrict__ pos, int pos_T, int v_col0, int vt_sp, int sn_tiles, int sm_tiles)
{
    constexpr int SMEM_ELEMS = (W_STAGES * W_STAGE * 2 > G_BM * G_TLD * 4 ? W_STAGES * W_STAGE : G_BM * G_TLD * 2);
    __shared__ __attribute__((aligned(1024))) bf16 smem[SMEM_ELEMS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 2, wc = wv & 3;                         // 2 x 4 waves, 64 x 64 each
    const int tiles_n = (int)gridDim.x, tiles_m_pad = (int)gridDim.y;
    int lin = (int)blockIdx.y * tiles_n + (int)blockIdx.x;
    const int total = tiles_n * tiles_m_pad;
    if ((total & 7) == 0) lin = (lin & 7) * (total >> 3) + (lin >> 3);
    int m0, n0;
    {
        const int per = sm_tiles * sn_tiles, sup = lin / per, r = lin - sup * per;
        const int n_sn = tiles_n / sn_tiles;
        const int tm = (sup / n_sn) * sm_tiles + r / sn_tiles, tn = (sup % n_sn) * sn_tiles + r % sn_tiles;
        m0 = tm * G_BM; n0 = tn * W_BN;
        if (m0 >= M) return;
    }
    float bv0[8], bv1[8]; float bc0, bc1;
    gemm_fetch_bias<EPI>(bias, n0, tid, bv0, bc0);
    gemm_fetch_bias<EPI>(bias, n0 + 128, tid, bv1, bc1);
    A += (int64_t)blockIdx.z * a_batch;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    const int nk = K / W_BK;
    // the bias loads above are older than every DMA: a counted wait on the DMAs retires them too
#pragma unroll
    for (int s = 0; s < W_STAGES - 1; s++)
        if (s < nk) {
            stage_rows32<G_BM>(A, lda, m0, M - 1, s * W_BK, smem + s * W_STAGE, wv, lane);
            stage_rows32<W_BN>(B, K, n0, N - 1, s * W_BK, smem + s * W_STAGE + G_BM * W_BK, wv, lane);
        }
    int cur = 0;
    for (int kt = 0; kt < nk; kt++) {
        const bf16 *sA = smem + cur * W_STAGE, *sB = sA + G_BM * W_BK;
        // K-step kt has landed once at most the DMAs of the one younger stage (3 per wave) are outstanding
        if (W_STAGES > 2 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + W_STAGES - 1 < nk) {                            // refill the stage K-step kt-1 has released
            int nx = cur + W_STAGES - 1; if (nx >= W_STAGES) nx -= W_STAGES;
            bf16 *nA = smem + nx * W_STAGE;
            stage_rows32<G_BM>(A, lda, m0, M - 1, (kt + W_STAGES - 1) * W_BK, nA, wv, lane);
            stage_rows32<W_BN>(B, K, n0, N - 1, (kt + W_STAGES - 1) * W_BK, nA + G_BM * W_BK, wv, lane);
        }
        bf16x8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = wr * 64 + i * 16 + fr;
            a[i] = *reinterpret_cast<const bf16x8 *>(&sA[row * W_BK + swz32(row, fq) * 8]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int row = wc * 64 + j * 16 + fr;
            b[j] = *reinterpret_cast<const bf16x8 *>(&sB[row * W_BK + swz32(row, fq) * 8]);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        cur = cur + 1 == W_STAGES ? 0 : cur + 1;
    }
    float *tile = reinterpret_cast<float *>(smem);               // [128][G_TLD] fp32: one half of the columns per pass
#pragma unroll
    for (int p = 0; p < 2; p++) {
        __syncthreads();                                         // operand reads (p = 0) / the first pass's tile reads are done
        if ((wc >> 1) == p) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        tile[(wr * 64 + i * 16 + fq * 4 + r) * G_TLD + (wc & 1) * 64 + j * 16 + fr] = acc[i][j][r];
        }
        __syncthreads();
        if (p == 0) gemm_store_tile<EPI>(tile, tid, m0, n0, M, N, bv0, bc0, Cv, ldc, (int64_t)blockIdx.z * c_batch, pos, pos_T, v_col0, vt_sp);
        else gemm_store_tile<EPI>(tile, tid, m0, n0 + 128, M, N, bv1, bc1, Cv, ldc, (int64_t)blockIdx.z * c_batch, pos, pos_T, v_col0, vt_sp);
    }
}

// ---------------------------------------------------------------------------
// LayerNorm over the last dimension (one wavefront per row)
// ---------------------------------------------------------------------------
// The row (d <= 1280 floats) is read once with 16-byte loads and kept in registers (<= 5 float4 per lane);
// statistics in fp32, two-pass (mean, then centred second moment) as torch does.
template <class OUT> __device__ __forceinline__ void ln_store4(OUT *p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void ln_store4<float>(float *p, float a, float b, float c, float d)
{
    *reinterpret_cast<float4 *>(p) = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void ln_store4<bf16>(bf16 *p, float a, float b, float c, float d)
{
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4 v; v[0] = (bf16)a; v[1] = (bf16)b; v[2] = (bf16)c; v[3] = (bf16)d;
    *reinterpret_cast<bf16x4 *>(p) = v;
}

template <class OUT>
__global__ __launch_bounds__(256) void k_layernorm(const float *x, const float *__restrict__ w, const float *__restrict__ b,
                                                  int64_t rows, int d, OUT *__restrict__ out, float eps = 1e-5f, float *out2 = nullptr)
{   // out2 (optional, may alias x): the same values in fp32 -- the post-LN residual stream of the BERT layers
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float4 *xr = reinterpret_cast<const float4 *>(x + row * d);
    const int nv = d >> 2;                               // float4 per row, d % 4 == 0
    float4 v[5];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int idx = lane + 64 * i;
        v[i] = idx < nv ? xr[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        if (lane + 64 * i < nv) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
    const float inv = rsqrtf(q / (float)d + eps);
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float4 ww = reinterpret_cast<const float4 *>(w)[idx], bb = reinterpret_cast<const float4 *>(b)[idx];
            const float y0 = (v[i].x - mean) * inv * ww.x + bb.x, y1 = (v[i].y - mean) * inv * ww.y + bb.y,
                        y2 = (v[i].z - mean) * inv * ww.z + bb.z, y3 = (v[i].w - mean) * inv * ww.w + bb.w;
            ln_store4<OUT>(out + row * d + 4 * idx, y0, y1, y2, y3);
            if (out2) ln_store4<float>(out2 + row * d + 4 * idx, y0, y1, y2, y3);
        }
    }
}

#include "pce_gemm256.inc"

// ---------------------------------------------------------------------------
// attention forward, transposed formulation on v_mfma_f32_32x32x16_bf16.
//   S^T = K Q^T   (keys on the accumulator rows, queries on its columns = lanes)
//   O^T = V^T P^T (P^T is consumed straight out of the S^T accumulator registers as the B operand:
//                  a 32x32 result has its column on the lane and its rows in the 16 registers, which is
//                  exactly a B fragment summed over the row index; no LDS round trip, no cross-lane
//                  softmax reductions except one exchange between the two lane halves)
// The S^T accumulator row rho of lane-half h, register 4g+i is rho = i + 8g + 4h.  The k-step s of the
// second product reads registers 8s..8s+7, i.e. rows 16s + 8a + 4h + b (j = 4a + b).  Softmax does not
// care about the order of the keys inside a tile, so lane r loads K row pi(r) (pi swaps bits 2 and 3):
// then slot (h, j) of k-step s is key 16s + 8h + j and the V^T fragment is one contiguous 16-byte read.
// One workgroup = 4 waves x 32 queries of one (clip, head); K [64 keys][64 d] and V^T [64 d][64 keys]
// tiles stream through LDS by DMA (global_load_lds), double buffered, XOR-swizzled like the GEMM operands.
// ---------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int AT_QB = 128;                  // queries per workgroup

struct AttnArgs {
    const bf16 *q; int64_t q_ld;            // query row i of clip c: q + (q_row0[c] + i) * q_ld + head * 64
    const bf16 *k; int64_t k_ld;            // key row j:             k + (k_row0[c] + j) * k_ld + head * 64
    const bf16 *vt; int64_t vt_clip; int vt_sp;   // V^T: vt + c * vt_clip + (head * 64 + d) * vt_sp + j
    const int *q_row0, *q_len, *k_row0, *k_len;   // per clip
    bf16 *out; int64_t out_ld;              // out + (q_row0[c] + i) * out_ld + head * 64
    int causal;                             // key j visible to query i only if j <= i
    int *fell_back;                         // (self-test only, else null) counts the workgroups of k_attention_lean that re-ran on the exact path
};

__device__ __forceinline__ void stage_kv(const bf16 *__restrict__ kbase, int64_t kld, int key0, int key_max,
                                         const bf16 *__restrict__ vtbase, int vt_sp, bf16 *sK, bf16 *sVt, int wv, int lane)
{
    // K tile: 64 rows (keys) x 128 B; V^T tile: 64 rows (d) x 128 B; 8 wave-instructions each, 2 per wave
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int r0 = wv * 16 + i * 8, row = r0 + (lane >> 3);
        const int c = swz_chunk(row, lane & 7);
        int kr = key0 + row; if (kr > key_max) kr = key_max;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(kbase + (int64_t)kr * kld + c * 8),
                                         (__attribute__((address_space(3))) void *)(sK + r0 * 64), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vtbase + (int64_t)row * vt_sp + key0 + c * 8),
                                         (__attribute__((address_space(3))) void *)(sVt + r0 * 64), 16, 0, 0);
    }
}

__global__ __launch_bounds__(256) void k_attention(AttnArgs A)
{
    constexpr int AT_STAGES = 3;                                                 // two K/V^T tiles in flight behind the one being consumed
    __shared__ __attribute__((aligned(1024))) bf16 smem[AT_STAGES * 2 * 64 * 64];        // [stage][K | V^T][64][64] = 48 KiB
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, clip = blockIdx.z;
    const int Sq = A.q_len[clip], Sk = A.k_len[clip];
    if ((int)blockIdx.x * AT_QB >= Sq) return;
    const int q0 = blockIdx.x * AT_QB + wv * 32;
    const bf16 *qbase = A.q + (int64_t)A.q_row0[clip] * A.q_ld + head * 64;
    const bf16 *kbase = A.k + (int64_t)A.k_row0[clip] * A.k_ld + head * 64;
    const bf16 *vtbase = A.vt + (int64_t)clip * A.vt_clip + (int64_t)head * 64 * A.vt_sp;
    // Q^T fragments (B operand): lane holds Q[q r][16 s + 8 h + j]
    bf16x8 qf[4];
    {
        int qr = q0 + r; if (qr >= Sq) qr = Sq - 1;
        const bf16 *qp = qbase + (int64_t)qr * A.q_ld;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) qf[s4] = *reinterpret_cast<const bf16x8 *>(qp + 16 * s4 + 8 * h);
    }
    f32x16 o[2];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) o[t][e] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const float sl2 = 0.125f * 1.4426950408889634f;               // softmax scale * log2(e): exp(x) = exp2(x log2 e)
    const int pr = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);   // pi(r): swap bits 2 and 3
    int k_need = Sk;
    if (A.causal) k_need = min(Sk, (int)blockIdx.x * AT_QB + AT_QB);               // keys beyond the block's last query are masked
    const int nt = (k_need + 63) / 64;
    const int my_q = q0 + r;
    stage_kv(kbase, A.k_ld, 0, Sk - 1, vtbase, A.vt_sp, smem, smem + 64 * 64, wv, lane);
    if (nt > 1) stage_kv(kbase, A.k_ld, 64, Sk - 1, vtbase, A.vt_sp, smem + 2 * 64 * 64, smem + 3 * 64 * 64, wv, lane);
    for (int kt = 0; kt < nt; kt++) {
        const bf16 *sK = smem + (kt % AT_STAGES) * (2 * 64 * 64), *sVt = sK + 64 * 64;
        // tile kt must have LANDED before anyone reads it: the compiler does not order LDS-DMA against the barrier
        // (the ISA had no vmcnt wait inside this loop; a leaner loop body exposed the race as NaNs), so wait here
        // Each wave issues 4 DMA instructions per tile, in tile order: tile kt has landed once at most the 4
        // youngest (tile kt+1) are outstanding.
        if (kt + 1 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                          // tile kt landed for every wave, tile kt-1 consumed
        if (kt + 2 < nt) {                                        // refill the stage tile kt-1 has just released
            bf16 *nK = smem + ((kt + 2) % AT_STAGES) * (2 * 64 * 64);
            stage_kv(kbase, A.k_ld, (kt + 2) * 64, Sk - 1, vtbase, A.vt_sp, nK, nK + 64 * 64, wv, lane);
        }
        // S^T for the two 32-key halves of the tile
        f32x16 st[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
#pragma unroll
            for (int e = 0; e < 16; e++) st[u][e] = 0.f;
            const int row = u * 32 + pr;
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(&sK[row * 64 + swz_chunk(row, 2 * s4 + h) * 8]);
                st[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s4], st[u], 0, 0, 0);
            }
        }
        // online softmax over this lane's query column: rows (keys) live in the registers of the two lane halves.
        // The softmax arithmetic, not the MFMAs, bounded this kernel (PMC: 44 VALU instructions per MFMA), so
        // interior tiles take a lean path: no visibility tests, the scale folded into the exponent's FMA
        // (exp2(s c - m)), raw v_exp_f32, and the accumulator rescale only when some lane's maximum moved.
        const bool edge = (kt * 64 + 63 >= Sk) || (A.causal && kt * 64 + 63 > q0);      // wave-uniform
        float mx = -1e30f;
        if (!edge) {
            float m0 = st[0][0], m1 = st[0][1], m2 = st[1][0], m3 = st[1][1];
#pragma unroll
            for (int e = 2; e < 16; e += 2) {
                m0 = fmaxf(m0, st[0][e]); m1 = fmaxf(m1, st[0][e + 1]); m2 = fmaxf(m2, st[1][e]); m3 = fmaxf(m3, st[1][e + 1]);
            }
            mx = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)) * sl2;
        } else {
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int rho = (e & 3) + 8 * (e >> 2) + 4 * h;                       // accumulator row
                    const int key = kt * 64 + u * 32 + ((rho & ~12) | ((rho & 4) << 1) | ((rho & 8) >> 1));
                    const bool vis = key < Sk && (!A.causal || key <= my_q);
                    const float v = vis ? st[u][e] : -1e30f;                              // (unscaled; -1e30 c is still hugely negative)
                    st[u][e] = v; mx = fmaxf(mx, v);
                }
            mx = mx > -1e29f ? mx * sl2 : -1e30f;
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
            const float corr = __builtin_amdgcn_exp2f(m_run - m_new);                      // exactly 1 where the maximum stayed
            l_run *= corr;
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int e = 0; e < 16; e++) o[t][e] *= corr;
            m_run = m_new;
        }
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (!edge) {
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const float p0 = __builtin_amdgcn_exp2f(fmaf(st[u][e], sl2, -m_run)), p1 = __builtin_amdgcn_exp2f(fmaf(st[u][e + 1], sl2, -m_run));
                    st[u][e] = p0; st[u][e + 1] = p1;
                    if (u == 0) { s0 += p0; s1 += p1; } else { s2 += p0; s3 += p1; }
                }
        } else {
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    // a masked key must add nothing, also when the whole tile is masked (m_run still -1e30)
                    const float p0 = st[u][e] > -1e29f ? __builtin_amdgcn_exp2f(fmaf(st[u][e], sl2, -m_run)) : 0.f;
                    const float p1 = st[u][e + 1] > -1e29f ? __builtin_amdgcn_exp2f(fmaf(st[u][e + 1], sl2, -m_run)) : 0.f;
                    st[u][e] = p0; st[u][e + 1] = p1;
                    if (u == 0) { s0 += p0; s1 += p1; } else { s2 += p0; s3 += p1; }
                }
        }
        float sum = (s0 + s1) + (s2 + s3);
        sum += __shfl_xor(sum, 32, 64);
        l_run += sum;
        // O^T += V^T P^T : k-step (u, s2) covers keys u*32 + 16 s2 .. +15 (in pi order); B = registers 8 s2 .. 8 s2 + 7
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; j++) pf[j] = (bf16)st[u][8 * s2 + j];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int row = t * 32 + r;                                           // d
                    const int chunk = (u * 32 + 16 * s2 + 8 * h) >> 3;                    // keys u*32+16 s2+8h .. +7
                    const bf16x8 vf = *reinterpret_cast<const bf16x8 *>(&sVt[row * 64 + swz_chunk(row, chunk) * 8]);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[t], 0, 0, 0);
                }
            }
    }
    // O^T[d][q]: this lane owns query q0 + r; d = 32 t + (e & 3) + 8 (e >> 2) + 4 h -> runs of 4 consecutive d
    if (my_q < Sq) {
        const float inv = 1.0f / l_run;
        bf16 *op = A.out + ((int64_t)A.q_row0[clip] + my_q) * A.out_ld + head * 64;
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                bf16x4 v4;
#pragma unroll
                for (int i = 0; i < 4; i++) v4[i] = (bf16)(o[t][4 * g + i] * inv);
                *reinterpret_cast<bf16x4 *>(op + 32 * t + 8 * g + 4 * h) = v4;
            }
    }
}

// ---------------------------------------------------------------------------
// k_attention_lean: the same product with less than half of k_attention's vector instructions per tile.  Per 64-key tile a wave issues
// 16 MFMAs and, in k_attention, about 200 VALU instructions; MFMAs and VALU instructions of all waves of a SIMD share one issue port
// (v_fma 4 cycles, v_exp 8, an MFMA 8 of its 32: MI355X_MICROARCH.md "vector-instruction ISSUE cost"), and with three waves per SIMD the
// PMC counters put k_attention's issue port at 96 % busy (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES = 0.32 per wave): it is instruction-count
// bound.  Removed from the tile loop:
//   * the running maximum.  exp2(s c - m) needs SOME reference m per query, not the maximum: softmax is invariant under the choice and
//     fp32 / bf16 share an 8-bit exponent, so with m fixed at the query's maximum over TILE 0 nothing is lost until a later score exceeds
//     m by 127 / c.  That is detected (a non-finite row sum) and the workgroup then runs again on the exact path (EXACT = true: running
//     maximum, accumulator rescale, VALU row sums) -- never on Whisper's logits; tests drive it with synthetic ones and force it;
//   * the row sums: a 17th..20th MFMA per tile against a constant "row 0 = ones" fragment accumulates sum_j P[q][j] in the matrix pipe
//     (which has slack) instead of 32 v_add / 16 v_pk_add (which has none); the sums are over the bf16 P the second product consumes;
//   * address arithmetic: K / V^T tiles arrive by LDS-DMA through buffer resources (constant per-lane offsets, one scalar offset per
//     tile; rows past the last key read as zeros and are masked like any invisible key), fragment addresses are loop invariants.
// Same workgroup shape, ring and register budget as k_attention (4 waves x 32 queries, 3 slots = 48 KB, three workgroups per CU).
// Measured, 256 clips x 12 heads x 1500^2: 2.56 -> 2.25 ms per launch.  Tried on the way and dropped: the next tile's S^T MFMAs
// interleaved with this tile's exponentials inside one wave (two score sets in registers, 4 slots, two workgroups per CU): 2.59 ms with or
// without the instruction diet -- waves then wait on LDS-DMA / barriers (SQ_WAIT_ANY 0.41) with too few partners to cover them; the same
// with eight waves per workgroup (half the DMA instructions per wave): 3.0 ms.
// ---------------------------------------------------------------------------
template <bool EXACT>
__device__ __forceinline__ bool attn_block(const AttnArgs &A, bf16 *smem, int bx, int head, int clip)
{
    constexpr int NS = 3, DPW = 4;                               // ring slots; DMA wave-instructions per wave and tile (16 pieces of 1 KB, 4 waves)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int Sq = A.q_len[clip], Sk = A.k_len[clip];
    const int q0 = bx * AT_QB + wv * 32;
    const bf16 *qbase = A.q + (int64_t)A.q_row0[clip] * A.q_ld + head * 64;
    const bf16 *kbase = A.k + (int64_t)A.k_row0[clip] * A.k_ld + head * 64;
    const bf16 *vtbase = A.vt + (int64_t)clip * A.vt_clip + (int64_t)head * 64 * A.vt_sp;
    const int kld = (int)A.k_ld, vsp = A.vt_sp;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(kbase), 0, ((Sk - 1) * kld + 64) * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(vtbase), 0, 64 * vsp * 2, 0x00020000);
    // a tile = 8 pieces of K (8 keys x 128 B each) + 8 pieces of V^T (8 rows of d); wave w moves pieces w and w + 4 of both
    int voffK[2], voffV[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int row = (wv + 4 * i) * 8 + (lane >> 3), c8 = swz_chunk(row, lane & 7) * 8;
        voffK[i] = (row * kld + c8) * 2; voffV[i] = (row * vsp + c8) * 2;
    }
    auto stage = [&](int t) {                                    // tile t -> slot t % NS
        bf16 *sK = smem + (t % NS) * (2 * 64 * 64), *sV = sK + 64 * 64;
        const int key0 = t * 64;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (__attribute__((address_space(3))) void *)(sK + (wv + 4 * i) * 8 * 64), 16, voffK[i], key0 * kld * 2, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (__attribute__((address_space(3))) void *)(sV + (wv + 4 * i) * 8 * 64), 16, voffV[i], key0 * 2, 0, 0);
        }
    };
    bf16x8 qf[4];
    {
        int qr = q0 + r; if (qr >= Sq) qr = Sq - 1;
        const bf16 *qp = qbase + (int64_t)qr * A.q_ld;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) qf[s4] = *reinterpret_cast<const bf16x8 *>(qp + 16 * s4 + 8 * h);
    }
    auto fzero = [] { f32x16 z;
#pragma unroll
        for (int e = 0; e < 16; e++) z[e] = 0.f;
        return z; };
    bf16x8 ones;                                                 // A fragment "row 0 = ones, rows 1..31 = 0" of the row-sum MFMA
#pragma unroll
    for (int j = 0; j < 8; j++) ones[j] = (bf16)(r == 0 ? 1.0f : 0.0f);
    f32x16 o[2], osum = fzero();
    o[0] = fzero(); o[1] = fzero();
    float m_run = -1e30f, l_run = 0.f;
    const float sl2 = 0.125f * 1.4426950408889634f;               // softmax scale * log2(e): exp(x) = exp2(x log2 e)
    const int pr = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);   // pi(r): swap bits 2 and 3 (see k_attention)
    int k_need = Sk;
    if (A.causal) k_need = min(Sk, bx * AT_QB + AT_QB);
    const int nt = (k_need + 63) / 64;
    const int my_q = q0 + r;
    int kaddr[2][4], vaddr[2][4];                                // LDS addresses of this lane's fragments inside a slot (elements)
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            const int row = u * 32 + pr;
            kaddr[u][s4] = row * 64 + swz_chunk(row, 2 * s4 + h) * 8;
        }
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int c = 0; c < 4; c++) {                             // c = 2 u + s2: keys u*32 + 16 s2 + 8 h .. + 7
            const int row = t * 32 + r;
            vaddr[t][c] = 64 * 64 + row * 64 + swz_chunk(row, 2 * c + h) * 8;
        }
    stage(0);
    if (nt > 1) stage(1);
    for (int kt = 0; kt < nt; kt++) {
        // tile kt landed (each wave issues DPW instructions per tile, in tile order); everyone is done with tile kt-1, whose slot tile kt+2 takes
        if (kt + 1 < nt) __builtin_amdgcn_s_waitcnt(0x0F70 | DPW); else __builtin_amdgcn_s_waitcnt(0x0F70 | 0);
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nt) stage(kt + 2);
        const bf16 *sC = smem + (kt % NS) * (2 * 64 * 64);
        f32x16 sc[2];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++)
                sc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&sC[kaddr[u][s4]]), qf[s4], s4 == 0 ? fzero() : sc[u], 0, 0, 0);
        // edge tiles (last keys of the clip, the causal diagonal): invisible scores become -1e30 (unscaled), and exp2(-1e30 c - m) = 0 for
        // any finite m; tile 0 always holds a visible key for every query, so m is finite from then on
        if ((kt * 64 + 63 >= Sk) || (A.causal && kt * 64 + 63 > q0)) {                          // wave-uniform
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int rho = (e & 3) + 8 * (e >> 2) + 4 * h;
                    const int key = kt * 64 + u * 32 + ((rho & ~12) | ((rho & 4) << 1) | ((rho & 8) >> 1));
                    const bool vis = key < Sk && (!A.causal || key <= my_q);
                    sc[u][e] = vis ? sc[u][e] : -1e30f;
                }
        }
        if (EXACT || kt == 0) {
            float m0 = sc[0][0], m1 = sc[0][1], m2 = sc[1][0], m3 = sc[1][1];
#pragma unroll
            for (int e = 2; e < 16; e += 2) { m0 = fmaxf(m0, sc[0][e]); m1 = fmaxf(m1, sc[0][e + 1]); m2 = fmaxf(m2, sc[1][e]); m3 = fmaxf(m3, sc[1][e + 1]); }
            float mx = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
            mx = mx > -1e29f ? mx * sl2 : -1e30f;
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
                const float corr = __builtin_amdgcn_exp2f(m_run - m_new);          // (tile 0: everything it scales is still zero)
                l_run *= corr;
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int e = 0; e < 16; e++) o[t][e] *= corr;
                m_run = m_new;
            }
        }
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const float p0 = __builtin_amdgcn_exp2f(fmaf(sc[u][e], sl2, -m_run)), p1 = __builtin_amdgcn_exp2f(fmaf(sc[u][e + 1], sl2, -m_run));
                sc[u][e] = p0; sc[u][e + 1] = p1;
                if (EXACT) { if (u == 0) { s0 += p0; s1 += p1; } else { s2 += p0; s3 += p1; } }
            }
        if (EXACT) {
            float sum = (s0 + s1) + (s2 + s3);
            sum += __shfl_xor(sum, 32, 64);
            l_run += sum;
        }
        // O^T += V^T P^T (and the row sums): k-step c = 2 u + s2 covers keys u*32 + 16 s2 .. +15 (in pi order); B = registers 8 s2 .. 8 s2 + 7
        bf16x8 vf[2][4];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int c = 0; c < 4; c++) vf[t][c] = *reinterpret_cast<const bf16x8 *>(&sC[vaddr[t][c]]);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; j++) pf[j] = (bf16)sc[c >> 1][8 * (c & 1) + j];
#pragma unroll
            for (int t = 0; t < 2; t++) o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[t][c], pf, o[t], 0, 0, 0);
            if (!EXACT) osum = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, osum, 0, 0, 0);
        }
    }
    if (!EXACT) l_run = __shfl(osum[0], r, 64);                  // row 0 of the row-sum product sits in register 0 of the lower lane half
    // (the accumulators are sums of p * v, up to l_run * max|v|: a row sum near FLT_MAX can pass a bare finiteness test while O has
    // already overflowed, so the fast path keeps eight decades of headroom; any finite reference is equally exact, the bound costs nothing)
    const bool ok = EXACT || (l_run > 0.f && l_run < 1.0e30f) || my_q >= Sq;
    if (!EXACT && __syncthreads_or(!ok)) return false;           // some row overflowed its fixed reference: the workgroup runs again, exactly
    // O^T[d][q]: this lane owns query q0 + r; d = 32 t + (e & 3) + 8 (e >> 2) + 4 h -> runs of 4 consecutive d
    if (my_q < Sq) {
        const float inv = 1.0f / l_run;
        bf16 *op = A.out + ((int64_t)A.q_row0[clip] + my_q) * A.out_ld + head * 64;
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                bf16x4 v4;
#pragma unroll
                for (int i = 0; i < 4; i++) v4[i] = (bf16)(o[t][4 * g + i] * inv);
                *reinterpret_cast<bf16x4 *>(op + 32 * t + 8 * g + 4 * h) = v4;
            }
    }
    return true;
}

__global__ __launch_bounds__(256, 3) void k_attention_lean(AttnArgs A, int force_exact)
{
    __shared__ __attribute__((aligned(1024))) bf16 smem[3 * 2 * 64 * 64];               // [slot][K | V^T][64][64] = 48 KiB
    // Workgroups go to the 8 XCDs round robin by linear id, and the query blocks of one (clip, head) share its K / V^T through L2: give every XCD
    // a CONTIGUOUS range of (clip, head, query block) triples, so that the blocks that share keys meet in one L2 instead of eight
    const unsigned nx = gridDim.x, ny = gridDim.y, total = nx * ny * gridDim.z;
    const unsigned lin = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z), xcd = lin & 7u, per = total >> 3, rem = total & 7u;
    const unsigned logical = xcd * per + (xcd < rem ? xcd : rem) + (lin >> 3);
    const int bx = (int)(logical % nx), head = (int)((logical / nx) % ny), clip = (int)(logical / (nx * ny));
    if (bx * AT_QB >= A.q_len[clip]) return;
    if (!force_exact && attn_block<false>(A, smem, bx, head, clip)) return;
    if (!force_exact && A.fell_back && threadIdx.x == 0) atomicAdd(A.fell_back, 1);
    __syncthreads();
    attn_block<true>(A, smem, bx, head, clip);
}

static void launch_attention(pce_ctx *c, dim3 grid, const AttnArgs &a, double flops = 0.0)
{
    // flops > 0: a launch the profiler brackets on its own (the encoder's); the decoder's small launches stay inside their composite entry
    if (c->attn_mode == 0) {
        KernelTimer kt(c, PCE_K_ATTENTION, nullptr, flops);
        hipLaunchKernelGGL(k_attention, grid, dim3(256), 0, c->stream, a);
    } else {
        KernelTimer kt(c, PCE_K_ATTENTION_LEAN, nullptr, flops);
        hipLaunchKernelGGL(k_attention_lean, grid, dim3(256), 0, c->stream, a, c->attn_mode == 2 ? 1 : 0);
    }
}

// ---------------------------------------------------------------------------
// Text decoder (teacher forced) and cross-attention alignment  -- openai-whisper timing.py find_alignment
// ---------------------------------------------------------------------------
__global__ void k_embed_tokens(const int *__restrict__ tokens /* [clips][T_pad] */, const float *__restrict__ tok_emb,
                               const float *__restrict__ pos_emb, int T_pad, int n_ctx, int d, int64_t rows, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * d) return;
    const int64_t m = i / d; const int col = (int)(i - m * d);
    int t = (int)(m % T_pad); if (t >= n_ctx) t = n_ctx - 1;      // pad rows beyond the clip's tokens: any finite value
    out[i] = tok_emb[(int64_t)tokens[m] * d + col] + pos_emb[(int64_t)t * d + col];
}

// last position of every sequence: resid row (clip * T_pad + len - 1) -> out row clip
__global__ void k_gather_last(const float *__restrict__ resid, const int *__restrict__ t_len, int T_pad, int d, int n, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * d) return;
    const int clip = (int)(i / d), col = (int)(i - (int64_t)clip * d);
    out[i] = resid[((int64_t)clip * T_pad + t_len[clip] - 1) * d + col];
}

// self-attention cache maintenance.  k_cache_k: K rows of a prefix run, [clip * T_pad + t][2d] (second half) -> cache
// [clip][T_cap][d]; k_append_kv: the new position of an incremental step, compact [clip][3d] (q | k | v) -> K row and V^T column.
__global__ void k_cache_k(const bf16 *__restrict__ qk, int T_pad, const int *__restrict__ t_len, int d, int T_cap, int n, bf16 *__restrict__ ck)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * T_pad * d) return;
    const int col = (int)(i % d); const int64_t r = i / d; const int t = (int)(r % T_pad), clip = (int)(r / T_pad);
    if (t < t_len[clip]) ck[((int64_t)clip * T_cap + t) * d + col] = qk[r * 2 * d + d + col];
}
__global__ void k_append_kv(const bf16 *__restrict__ qkv, const int *__restrict__ pos_of, int d, int T_cap, int sp, int n, bf16 *__restrict__ ck,
                            bf16 *__restrict__ cvt)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * d) return;
    const int clip = (int)(i / d), col = (int)(i - (int64_t)clip * d);
    const int pos = pos_of[clip];
    ck[((int64_t)clip * T_cap + pos) * d + col] = qkv[(int64_t)clip * 3 * d + d + col];
    cvt[((int64_t)clip * d + col) * sp + pos] = qkv[(int64_t)clip * 3 * d + 2 * d + col];
}
__global__ void k_embed_one(const int *__restrict__ tok, const float *__restrict__ tok_emb, const float *__restrict__ pos_emb,
                            const int *__restrict__ pos_of, int d, int n, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * d) return;
    const int clip = (int)(i / d), col = (int)(i - (int64_t)clip * d);
    out[i] = tok_emb[(int64_t)tok[clip] * d + col] + pos_emb[(int64_t)pos_of[clip] * d + col];
}

// ---- device-resident decoding loop (round 3) -----------------------------------------------------------------------------------
// After the logit filters have chosen `next` for every sequence, k_step_advance appends it to the device-resident token table,
// refreshes the per-sequence tables the NEXT step's kernels read (new token, position, self-attention key count, "ended" flag),
// records the step's outputs and counts the sequences whose last token is end-of-text.  Nothing of a step passes through the host.
// tables ct[8 n]: new token | q_row0 | q_len | k_row0 | k_len | a_row0 | a_len | position   (the layout pce_whisper_decode_step_ex uploads)
__global__ __launch_bounds__(256) void k_step_advance(int n, int T_cap, int eot, int max_new, int *__restrict__ tok_table, int *__restrict__ len,
                                                     const int *__restrict__ next, const float *__restrict__ next_lp, int *__restrict__ ct,
                                                     int *__restrict__ ended, int *__restrict__ out_tok, float *__restrict__ out_lp,
                                                     int *__restrict__ step_ctr, int *__restrict__ n_ended)
{   // ONE workgroup (a few hundred sequences)
    const int step = *step_ctr;
    __syncthreads();
    int cnt = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int L = len[i], t = next[i];
        if (step < max_new) { out_tok[(size_t)i * max_new + step] = t; out_lp[(size_t)i * max_new + step] = next_lp[i]; }
        if (L < T_cap) { tok_table[(size_t)i * T_cap + L] = t; len[i] = L + 1; }
        ct[i] = t; ct[4 * n + i] = (L < T_cap ? L : T_cap - 1) + 1; ct[7 * n + i] = L < T_cap ? L : T_cap - 1;
        const int e = t == eot;
        ended[i] = e; cnt += e;
    }
    if (cnt) atomicAdd(n_ended + (step < max_new ? step : max_new - 1), cnt);
    if (threadIdx.x == 0) *step_ctr = step + 1;
}

// Attention of ONE query per (clip, head) over a long key axis: the cross-attention (1 500 audio positions) and the cached
// self-attention of an incremental decoding step.  The MFMA attention kernels spend a 32-query tile on it and walk the keys as 24
// dependent LDS-DMA tiles per workgroup; this is a streaming kernel: HBM-bound by construction (one pass over K, one over V^T:
// 14 GB per step for 256 clips at Whisper-small size, the roofline of free-running decoding).
//   phase 1: scores.  8 lanes per key row (16 B = 8 dims each: a wave-instruction covers eight whole 128-byte lines), eight rows per lane
//            in flight, 8-lane sums by DPP -> fp32 scores in LDS
//   phase 2: softmax (fp32, max-subtracted) in LDS
//   phase 3: O[d] = sum_t p[t] V^T[d][t]: a wave streams one V^T row as 1 KB pieces (lane = 8 consecutive keys, p in registers), 16 rows per wave
// `skip[clip]` != 0: the sequence has ended, its output is never used (the filters return end-of-text for it): no bytes are read.
struct Attn1Args {
    const bf16 *q; int64_t q_ld;              // query of clip c: q + c * q_ld + head * 64
    const bf16 *k; int64_t k_ld;              // key row j:       k + (k_row0[c] + j) * k_ld + head * 64
    const bf16 *vt; int64_t vt_clip; int vt_sp;   // V^T:        vt + c * vt_clip + (head * 64 + d) * vt_sp + j   (columns >= k_len hold finite values)
    const int *k_row0, *k_len;                // per clip; k_len <= 1536
    const int *skip;                          // per clip or null
    bf16 *out; int64_t out_ld;                // out + c * out_ld + head * 64
};
// 128 threads: 16 workgroups fit a CU, so the 3 072 (clip, head) pairs of 256 clips at Whisper-small size are ONE resident round
// (with 256 threads 2 048 run at a time and the second round leaves half the chip idle)
constexpr int A1_T = 128, A1_W = A1_T / 64;
__global__ __launch_bounds__(A1_T) void k_cross_attn1(Attn1Args A)
{
    __shared__ float sp[1536];
    __shared__ float red[2 * A1_W];
    __shared__ float so[64];
    const int head = blockIdx.x, clip = blockIdx.y;
    if (A.skip && A.skip[clip]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int Sk = A.k_len[clip];
    const int g = lane >> 3, ch = lane & 7;
    float qf[8];
    {
        const bf16x8 qv = *reinterpret_cast<const bf16x8 *>(A.q + (int64_t)clip * A.q_ld + head * 64 + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; e++) qf[e] = (float)qv[e] * 0.125f;
    }
    const bf16 *kb = A.k + (int64_t)A.k_row0[clip] * A.k_ld + head * 64 + ch * 8;
    constexpr int U = 8;
    for (int base = wv * 8; base < Sk; base += 8 * A1_W * U) {
        bf16x8 kv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int r = base + u * 8 * A1_W + g;
            kv[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(kb + (int64_t)(r < Sk ? r : Sk - 1) * A.k_ld));      // read once per step: streamed
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) d = fmaf((float)kv[u][e], qf[e], d);
            d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
            const int r = base + u * 8 * A1_W + g;
            if (ch == 0 && r < Sk) sp[r] = d;
        }
    }
    __syncthreads();
    float m = -3.0e38f;
    for (int t = tid; t < Sk; t += A1_T) m = fmaxf(m, sp[t]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) red[wv] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int u = 1; u < A1_W; u++) m = fmaxf(m, red[u]);
    float sum = 0.f;
    const int Sp = (Sk + 511) & ~511;                            // p is read in 512-key pieces: zero beyond the last key
    for (int t = tid; t < Sp; t += A1_T) {
        const float pv = t < Sk ? __expf(sp[t] - m) : 0.f;
        sp[t] = pv; sum += pv;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) red[A1_W + wv] = sum;
    __syncthreads();
    float tot = red[A1_W];
#pragma unroll
    for (int u = 1; u < A1_W; u++) tot += red[A1_W + u];
    const float inv = 1.0f / tot;
    const int np = Sp >> 9;                                      // 1..3 pieces
    float pr[3][8];
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int e = 0; e < 8; e++) pr[j][e] = j < np ? sp[j * 512 + lane * 8 + e] : 0.f;
    constexpr int RW = 64 / A1_W;                                // V^T rows (output dims) per wave
    const bf16 *vb = A.vt + (int64_t)clip * A.vt_clip + (int64_t)(head * 64 + wv * RW) * A.vt_sp + lane * 8;
#pragma unroll 2
    for (int r4 = 0; r4 < RW; r4 += 4) {
        bf16x8 vv[4][3];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                if (j < np) vv[r][j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(vb + (int64_t)(r4 + r) * A.vt_sp + j * 512));
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 3; j++)
                if (j < np)
#pragma unroll
                    for (int e = 0; e < 8; e++) a = fmaf(pr[j][e], (float)vv[r][j][e], a);
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
            if (lane == 0) so[wv * RW + r4 + r] = a * inv;
        }
    }
    __syncthreads();
    if (tid < 64) A.out[(int64_t)clip * A.out_ld + head * 64 + tid] = (bf16)so[tid];
}

// openai-whisper decoding.py at temperature 0, one workgroup per sequence: SuppressBlank, SuppressTokens,
// ApplyTimestampRules on the logits of the last position, then the GreedyDecoder's arg-max (first maximum) and its
// "once end-of-text, always end-of-text" rule.  vmask: bit 0 = always suppressed (suppress list, no_timestamps),
// bit 1 = suppressed at the first sampled position (blank, end of text).
struct DecRules { int eot, ts_begin, n_vocab, ld, sample_begin, max_initial_ts; float temperature; unsigned seed_lo, seed_hi; int probe; };
// uniform in (0, 1) from (seed, clip, position, token): two rounds of the splitmix64 finaliser over the packed counter
__device__ __forceinline__ float dec_uniform(unsigned seed_lo, unsigned seed_hi, int clip, int pos, int v)
{
    unsigned long long z = ((unsigned long long)seed_hi << 32 | seed_lo) + 0x9E3779B97F4A7C15ull * ((unsigned long long)(unsigned)clip << 40 ^ (unsigned long long)(unsigned)pos << 20 ^ (unsigned)v);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    // 23 random bits + 1/2: every value is exact in float and lies strictly inside (0, 1) (24 bits + 1/2 rounds to 1.0 at the top)
    return ((float)(unsigned)(z >> 41) + 0.5f) * (1.0f / 8388608.0f);
}
__global__ __launch_bounds__(256) void k_decode_rules(float *__restrict__ logits, const int *__restrict__ tokens, const int *__restrict__ t_len,
                                                     int T_pad, const unsigned char *__restrict__ vmask, DecRules R, const int *__restrict__ sample_begin_of,
                                                     int *__restrict__ next, float *__restrict__ next_logprob, float *__restrict__ probe_prob)
{
    __shared__ float r_f[2][4]; __shared__ int r_i[4]; __shared__ float s_bcast[2]; __shared__ int s_flags[4];
    const int clip = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int L = t_len[clip];
    const int *seq = tokens + (size_t)clip * T_pad;
    float *x = logits + (size_t)clip * R.ld;
    if (sample_begin_of) R.sample_begin = sample_begin_of[clip];
    if (R.probe >= 0 && probe_prob) {
        // softmax of the unfiltered logits at one token (no_speech_prob when the prefix ends at <|startoftranscript|>)
        float m = -__builtin_huge_valf();
        for (int v = tid; v < R.n_vocab; v += 256) m = fmaxf(m, x[v]);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) r_f[0][wv] = m;
        __syncthreads();
        m = fmaxf(fmaxf(r_f[0][0], r_f[0][1]), fmaxf(r_f[0][2], r_f[0][3]));
        float se = 0.f;
        for (int v = tid; v < R.n_vocab; v += 256) se += __expf(x[v] - m);
        for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
        if (lane == 0) r_f[1][wv] = se;
        __syncthreads();
        if (tid == 0) probe_prob[clip] = __expf(x[R.probe] - m) / ((r_f[1][0] + r_f[1][1]) + (r_f[1][2] + r_f[1][3]));
        __syncthreads();
    }
    if (tid == 0) {
        const int ns = L - R.sample_begin;                         // sampled tokens so far
        const bool last_ts = ns >= 1 && seq[L - 1] >= R.ts_begin;
        const bool pen_ts = ns < 2 || seq[L - 2] >= R.ts_begin;
        int last_stamp = -1;
        for (int t = R.sample_begin; t < L; t++) if (seq[t] >= R.ts_begin) last_stamp = seq[t];
        int ts_floor = R.ts_begin;                                 // timestamps below this are forbidden
        if (last_stamp >= 0) ts_floor = (last_ts && !pen_ts) ? last_stamp : last_stamp + 1;
        s_flags[0] = last_ts ? (pen_ts ? 1 : 2) : 0;               // 1: no timestamp may follow, 2: no text token may follow
        s_flags[1] = ts_floor;
        s_flags[2] = ns == 0;
        s_flags[3] = L > 0 && seq[L - 1] == R.eot;
    }
    __syncthreads();
    const int pair = s_flags[0], ts_floor = s_flags[1]; const bool first = s_flags[2] != 0, done = s_flags[3] != 0;
    const float NEG = -__builtin_huge_valf();
    // pass 1: masks; maximum over the text ids and over the timestamp ids
    float m_text = NEG, m_ts = NEG;
    for (int v = tid; v < R.n_vocab; v += 256) {
        float a = x[v];
        const unsigned char mk = vmask[v];
        bool off = (mk & 1) || (first && (mk & 2));
        if (pair == 1 && v >= R.ts_begin) off = true;
        if (pair == 2 && v < R.eot) off = true;
        if (v >= R.ts_begin && v < ts_floor) off = true;
        if (first && (v < R.ts_begin || (R.max_initial_ts >= 0 && v > R.ts_begin + R.max_initial_ts))) off = true;
        if (off) a = NEG;
        x[v] = a;
        if (v < R.ts_begin) m_text = fmaxf(m_text, a); else m_ts = fmaxf(m_ts, a);
    }
    for (int o = 32; o > 0; o >>= 1) { m_text = fmaxf(m_text, __shfl_xor(m_text, o, 64)); m_ts = fmaxf(m_ts, __shfl_xor(m_ts, o, 64)); }
    if (lane == 0) { r_f[0][wv] = m_text; r_f[1][wv] = m_ts; }
    __syncthreads();
    m_text = fmaxf(fmaxf(r_f[0][0], r_f[0][1]), fmaxf(r_f[0][2], r_f[0][3]));
    m_ts = fmaxf(fmaxf(r_f[1][0], r_f[1][1]), fmaxf(r_f[1][2], r_f[1][3]));
    __syncthreads();
    // pass 2: log-sum-exp over the timestamps; "if the probability mass of the timestamps exceeds every text token, sample a timestamp"
    float se = 0.f;
    if (m_ts > NEG)
        for (int v = R.ts_begin + tid; v < R.n_vocab; v += 256) se += __expf(x[v] - m_ts);
    for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
    if (lane == 0) r_f[0][wv] = se;
    __syncthreads();
    if (tid == 0) {
        const float tot = (r_f[0][0] + r_f[0][1]) + (r_f[0][2] + r_f[0][3]);
        const float lse_ts = m_ts > NEG ? m_ts + __logf(tot) : NEG;
        s_bcast[0] = (lse_ts > m_text) ? 1.f : 0.f;
    }
    __syncthreads();
    const bool only_ts = s_bcast[0] != 0.f;
    // pass 3: arg-max (first maximum); at a temperature the arg-max of logits / temperature + Gumbel noise, which is one
    // draw from softmax(logits / temperature) (Categorical(logits = logits / temperature).sample() of GreedyDecoder.update)
    float best = NEG; int bi = 0x7fffffff;
    const bool sampling = R.temperature > 0.f;
    const float inv_t = sampling ? 1.0f / R.temperature : 1.0f;
    for (int v = (only_ts ? R.ts_begin : 0) + tid; v < R.n_vocab; v += 256) {
        float a = x[v];
        if (sampling && a > NEG) a = a * inv_t - __logf(-__logf(dec_uniform(R.seed_lo, R.seed_hi, clip, L, v)));
        if (a > best) { best = a; bi = v; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { r_f[0][wv] = best; r_i[wv] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int u = 1; u < 4; u++) if (r_f[0][u] > best || (r_f[0][u] == best && r_i[u] < bi)) { best = r_f[0][u]; bi = r_i[u]; }
        if (bi == 0x7fffffff) bi = only_ts ? R.ts_begin : 0;       // every candidate is -inf: torch.argmax returns the first index
        next[clip] = done ? R.eot : bi;
        s_bcast[1] = sampling ? x[bi] : best;                      // the choice's own (filtered, unscaled) logit
    }
    __syncthreads();
    // log-probability of the choice under the filtered distribution (GreedyDecoder.update adds it to sum_logprobs unless
    // the sequence had already ended): log_softmax over what is left after the filters
    {
        const float top = s_bcast[1];                              // (a sampled choice need not be the maximum: exp(x - top) may exceed 1, the sum stays exact enough in float)
        float se2 = 0.f;
        if (top > NEG)
            for (int v = (only_ts ? R.ts_begin : 0) + tid; v < R.n_vocab; v += 256) se2 += __expf(x[v] - top);
        for (int o = 32; o > 0; o >>= 1) se2 += __shfl_xor(se2, o, 64);
        __syncthreads();
        if (lane == 0) r_f[1][wv] = se2;
        __syncthreads();
        if (tid == 0) {
            const float tot = (r_f[1][0] + r_f[1][1]) + (r_f[1][2] + r_f[1][3]);
            next_logprob[clip] = (done || !(top > NEG)) ? 0.f : -__logf(tot);     // x[best] - logsumexp = -log(sum exp(x - best))
        }
    }
}

// BERT embeddings: word[id] + token_type[0] + position[t] (the LayerNorm follows as its own launch)
__global__ void k_bert_embed(const int *__restrict__ tokens /* [seqs][T_pad] */, const float *__restrict__ word, const float *__restrict__ pos,
                             const float *__restrict__ type0, int T_pad, int n_pos, int d, int64_t rows, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * d) return;
    const int64_t m = i / d; const int col = (int)(i - m * d);
    int t = (int)(m % T_pad); if (t >= n_pos) t = n_pos - 1;        // pad rows: any finite value
    out[i] = (word[(int64_t)tokens[m] * d + col] + type0[col]) + pos[(int64_t)t * d + col];
}

// softmax over the audio frames of the (scaled) cross-attention logits of one alignment head:
// w[clip][sel][t][s] = softmax_s(q_t . k_s * 0.125 * qk_scale), s < F_c.  16 tokens per workgroup, 4 waves x 16
// keys per 64-key tile on v_mfma_f32_16x16x32_bf16, two passes (row max / sum, then normalised write).
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_max16(float v)      // over the 16 lanes of a DPP row, result in every lane
{
    v = fmaxf(v, dpp_f32<0xB1>(v)); v = fmaxf(v, dpp_f32<0x4E>(v)); v = fmaxf(v, dpp_f32<0x141>(v)); v = fmaxf(v, dpp_f32<0x140>(v));
    return v;
}
__device__ __forceinline__ float row_sum16f(float v)
{
    v += dpp_f32<0xB1>(v); v += dpp_f32<0x4E>(v); v += dpp_f32<0x141>(v); v += dpp_f32<0x140>(v);
    return v;
}
struct AlignArgs {
    const bf16 *q; int64_t q_ld;             // decoder cross-attention queries [clips * T_pad][d]
    const bf16 *k; int64_t k_ld;             // audio keys of this layer      [clips * 1500][d]
    const int *t_len, *f_len;                // per clip: tokens, frames considered (num_frames // 2)
    const int *heads;                        // head index of every selected head of this layer
    float *w; int sel0, n_sel_total, T_pad, F_pad;
    float scale;
};
__global__ __launch_bounds__(256) void k_align_scores(AlignArgs A)
{
    __shared__ float red_m[4][16], red_s[4][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int clip = blockIdx.z, sel = blockIdx.y, t0 = blockIdx.x * 16;
    const int T = A.t_len[clip], F = A.f_len[clip];
    if (t0 >= T || F <= 0) return;
    const int head = A.heads[sel];
    const bf16 *qb = A.q + ((int64_t)clip * A.T_pad) * A.q_ld + head * 64;
    const bf16 *kb = A.k + ((int64_t)clip * W_CTX) * A.k_ld + head * 64;
    bf16x8 qf[2];
    {
        int tr = t0 + fr; if (tr >= T) tr = T - 1;
        qf[0] = *reinterpret_cast<const bf16x8 *>(qb + (int64_t)tr * A.q_ld + fq * 8);
        qf[1] = *reinterpret_cast<const bf16x8 *>(qb + (int64_t)tr * A.q_ld + 32 + fq * 8);
    }
    float *wb = A.w + (((int64_t)clip * A.n_sel_total + A.sel0 + sel) * A.T_pad) * (int64_t)A.F_pad;
    float m_run[4], l_run[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) { m_run[r4] = -1e30f; l_run[r4] = 0.f; }
    for (int pass = 0; pass < 2; pass++) {
        for (int s0 = wv * 16; s0 < F; s0 += 64) {
            int kr = s0 + fr; if (kr >= F) kr = F - 1;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(kb + (int64_t)kr * A.k_ld + kk * 32 + fq * 8);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[kk], kf, acc, 0, 0, 0);
            }
            const bool valid = (s0 + fr) < F;                  // column = key s0 + fr, rows = tokens fq*4 + r4
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) {
                const float v = valid ? acc[r4] * A.scale : -1e30f;
                if (pass == 0) {
                    const float mx = row_max16(v);
                    const float m_new = fmaxf(m_run[r4], mx);
                    const float e = valid ? __expf(v - m_new) : 0.f;
                    l_run[r4] = l_run[r4] * __expf(m_run[r4] - m_new) + row_sum16f(e);
                    m_run[r4] = m_new;
                } else if (valid && t0 + fq * 4 + r4 < T) {
                    wb[(int64_t)(t0 + fq * 4 + r4) * A.F_pad + s0 + fr] = __expf(v - m_run[r4]) / l_run[r4];
                }
            }
        }
        if (pass == 0) {                                       // merge the four waves' partial (max, sum) per token row
            if (fr == 0)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) { red_m[wv][fq * 4 + r4] = m_run[r4]; red_s[wv][fq * 4 + r4] = l_run[r4]; }
            __syncthreads();
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) {
                const int row = fq * 4 + r4;
                float m = fmaxf(fmaxf(red_m[0][row], red_m[1][row]), fmaxf(red_m[2][row], red_m[3][row]));
                float l = 0.f;
                for (int u = 0; u < 4; u++) l += red_s[u][row] * __expf(red_m[u][row] - m);
                m_run[r4] = m; l_run[r4] = l;
            }
        }
    }
}

// std/mean normalisation over the token axis (torch.std_mean(dim=-2, unbiased=False)), in place
__global__ __launch_bounds__(256) void k_align_colnorm(float *__restrict__ w, const int *__restrict__ t_len, const int *__restrict__ f_len,
                                                      int n_sel, int T_pad, int F_pad)
{
    const int clip = blockIdx.z, sel = blockIdx.y, s = blockIdx.x * blockDim.x + threadIdx.x;
    const int T = t_len[clip], F = f_len[clip];
    if (s >= F) return;
    float *col = w + (((int64_t)clip * n_sel + sel) * T_pad) * (int64_t)F_pad + s;
    if (T <= 64) {                                               // the usual case (a window's tokens): the column lives in registers, one read and one write
        float v[64];
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 64; t++) { v[t] = t < T ? __builtin_nontemporal_load(col + (int64_t)t * F_pad) : 0.f; sum += v[t]; }      // (same order of additions as the loop below)
        const float mean = sum / (float)T;
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < 64; t++) if (t < T) { const float a = v[t] - mean; q += a * a; }
        const float sd = sqrtf(q / (float)T);
#pragma unroll
        for (int t = 0; t < 64; t++) if (t < T) col[(int64_t)t * F_pad] = (v[t] - mean) / sd;
        return;
    }
    float sum = 0.f;
    for (int t = 0; t < T; t++) sum += col[(int64_t)t * F_pad];
    const float mean = sum / (float)T;
    float q = 0.f;
    for (int t = 0; t < T; t++) { const float a = col[(int64_t)t * F_pad] - mean; q += a * a; }
    const float sd = sqrtf(q / (float)T);
    for (int t = 0; t < T; t++) col[(int64_t)t * F_pad] = (col[(int64_t)t * F_pad] - mean) / sd;
}

// median filter along time (reflect padding), mean over the selected heads, rows [sot_len, T-1) -> cost = -mean (fp64).
// WIDTH 7 (whisper-timestamped's default) selects with a 13-exchange network in registers; WIDTH 0 is the generic
// insertion sort (<= 15 taps, a scratch array: 20 x slower, measured 6.1 ms vs 0.3 ms per 256 clips).
__device__ __forceinline__ void cswap(float &a, float &b) { const bool sw = a > b; const float lo = sw ? b : a, hi = sw ? a : b; a = lo; b = hi; }
template <int WIDTH>
__global__ __launch_bounds__(256) void k_align_cost(const float *__restrict__ w, const int *__restrict__ t_len, const int *__restrict__ f_len,
                                                   int n_sel, int T_pad, int F_pad, int sot_len, int width, int N_max,
                                                   double *__restrict__ cost /* [clips][N_max][F_pad] */)
{
    const int clip = blockIdx.z, row = blockIdx.y, s = blockIdx.x * blockDim.x + threadIdx.x;
    const int T = t_len[clip], F = f_len[clip];
    const int t = row + sot_len;
    if (t >= T - 1 || s >= F) return;
    const int pad = width / 2;
    float acc = 0.f;
    if (WIDTH == 7 && F > 3) {
        int idx[7];
#pragma unroll
        for (int u = 0; u < 7; u++) {
            int i = s - 3 + u;
            if (i < 0) i = -i;
            if (i >= F) i = 2 * (F - 1) - i;
            idx[u] = i;
        }
        for (int sel = 0; sel < n_sel; sel++) {
            const float *rp = w + ((((int64_t)clip * n_sel + sel) * T_pad) + t) * (int64_t)F_pad;
            float p0 = rp[idx[0]], p1 = rp[idx[1]], p2 = rp[idx[2]], p3 = rp[idx[3]], p4 = rp[idx[4]], p5 = rp[idx[5]], p6 = rp[idx[6]];
            cswap(p0, p5); cswap(p0, p3); cswap(p1, p6); cswap(p2, p4); cswap(p0, p1); cswap(p3, p5); cswap(p2, p6);
            cswap(p2, p3); cswap(p3, p6); cswap(p4, p5); cswap(p1, p4); cswap(p1, p3); cswap(p3, p4);
            acc += p3;
        }
    } else {
        for (int sel = 0; sel < n_sel; sel++) {
            const float *rp = w + ((((int64_t)clip * n_sel + sel) * T_pad) + t) * (int64_t)F_pad;
            float v;
            if (F <= pad) v = rp[s];                               // torch skips the filter for very short rows
            else {
                float win[15];
                for (int u = 0; u < width; u++) {
                    int idx = s - pad + u;
                    if (idx < 0) idx = -idx;
                    if (idx >= F) idx = 2 * (F - 1) - idx;
                    win[u] = rp[idx];
                }
                for (int a = 1; a < width; a++) {                  // insertion sort of <= 15 values
                    const float key = win[a]; int b = a - 1;
                    while (b >= 0 && win[b] > key) { win[b + 1] = win[b]; b--; }
                    win[b + 1] = key;
                }
                v = win[pad];
            }
            acc += v;
        }
    }
    cost[((int64_t)clip * N_max + row) * F_pad + s] = -(double)(acc / (float)n_sel);
}

__global__ void k_f32_to_bf16(const float *__restrict__ in, bf16 *__restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (bf16)in[i];
}

// librosa.filters.mel(sr=16000, n_fft=400, n_mels) (Slaney scale and normalisation), float32
void mel_filterbank(int n_mels, std::vector<float> &dense)
{
    const double sr = 16000.0, f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    auto hz_to_mel = [&](double f) { return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp; };
    auto mel_to_hz = [&](double m) { return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m; };
    std::vector<double> mel_f((size_t)n_mels + 2), fft_f(W_BINS);
    const double m_lo = hz_to_mel(0.0), m_hi = hz_to_mel(sr / 2);
    for (int i = 0; i < n_mels + 2; i++) mel_f[(size_t)i] = mel_to_hz(m_lo + (m_hi - m_lo) * i / (n_mels + 1));
    for (int k = 0; k < W_BINS; k++) fft_f[(size_t)k] = (sr / 2) * k / (W_BINS - 1);
    dense.assign((size_t)n_mels * W_BINS, 0.f);
    for (int i = 0; i < n_mels; i++) {
        const double enorm = 2.0 / (mel_f[(size_t)i + 2] - mel_f[(size_t)i]);
        for (int k = 0; k < W_BINS; k++) {
            const double lower = -(mel_f[(size_t)i] - fft_f[(size_t)k]) / (mel_f[(size_t)i + 1] - mel_f[(size_t)i]);
            const double upper = (mel_f[(size_t)i + 2] - fft_f[(size_t)k]) / (mel_f[(size_t)i + 2] - mel_f[(size_t)i + 1]);
            const double w = std::max(0.0, std::min(lower, upper));
            dense[(size_t)i * W_BINS + k] = (float)(w * enorm);
        }
    }
}

struct WhisperState {
    pce_whisper_dims dims{};
    bool loaded = false;
    DevBuf tables, logspec, clipmax, mel_tm, mel_start, w_bf16, w_f32, pos;
    DevBuf c1_out, resid, ln_out, qkv, vt, attn, hidden, final_out, enc_tab, delta, delta2;
    size_t vt_elems_zeroed = 0;
    // text decoder
    pce_whisper_text_dims tdims{};
    bool dec_loaded = false;
    DevBuf dw_bf16, dw_f32, d_tok_emb, d_pos_emb;
    DevBuf d_tab, d_tokens, d_resid, d_ln, d_qk, d_vt, d_attn, d_q, d_hidden, d_enc_bf16, d_aw, d_cost, d_trace, d_pi, d_pj, d_pl, d_heads;
    int enc_bf16_clips = -1;             // d_enc_bf16 holds the bf16 copy of final_out for this many clips (written by the encoder's last LayerNorm)
    // free-running decoding: tied output projection in bf16 (rows padded to 128), cross K / V of every layer, last-position buffers
    DevBuf g_emb_bf16, g_xk, g_xvt, g_last, g_lastln, g_logits, g_mask, g_next;
    int g_xkv_clips = -1;            // clips the cross K / V cache was computed for (-1: stale)
    // self-attention K / V of the sequences decoded so far: K rows [layer][clip][T_cap][d], V^T [layer][clip][d][512]
    DevBuf g_sk, g_svt, g_c_resid, g_c_ln, g_c_qkv, g_c_attn, g_c_q, g_c_hidden, g_c_tab, g_loop;
    std::vector<int> g_cache_tok;     // host copy of the cached prefixes [clip][T_cap]
    int g_cache_len = -1, g_cache_n = -1;   // g_cache_len: < 0 = nothing cached (new encoded batch); per-sequence lengths in g_cache_lens
    std::vector<int> g_cache_lens;
    struct DLayer { size_t ln1_w, ln1_b, qkv_w, qkv_b, out_w, out_b, lnx_w, lnx_b, xq_w, xq_b, xkv_w, xkv_b, xout_w, xout_b,
                    ln2_w, ln2_b, m1_w, m1_b, m2_w, m2_b; };
    std::vector<DLayer> dlayers;
    size_t dln_w = 0, dln_b = 0;
    int al_n = -1, al_Nmax = 0, al_Fpad = 0, al_Mmax = 0;
    std::vector<int> al_rows, al_cols;
    MelTables mt{};
    int mel_nmels = 0;
    int32_t n_clips_mel = -1, n_clips_enc = -1, enc_tab_clips = -1;
    // break-prediction token classifier (BertForTokenClassification)
    struct Bert {
        pce_bert_dims dims{};
        bool loaded = false;
        DevBuf w_bf16, w_f32, word, pos, type0;
        DevBuf tab, tokens, resid, ln, qk, vt, attn, hidden, logits;
        struct Layer { size_t qkv_w, qkv_b, out_w, out_b, ln1_w, ln1_b, m1_w, m1_b, m2_w, m2_b, ln2_w, ln2_b; };
        std::vector<Layer> layers;
        size_t lne_w = 0, lne_b = 0, cls_w = 0, cls_b = 0;
        int n_seq = -1, T_pad = 0;
        std::vector<int> lens;
    } bert;
    // offsets (elements) into w_bf16 / w_f32
    struct Layer { size_t ln1_w, ln1_b, qkv_w, qkv_b, out_w, out_b, ln2_w, ln2_b, m1_w, m1_b, m2_w, m2_b; };
    size_t c1_w = 0, c1_b = 0, c2_w = 0, c2_b = 0, lnp_w = 0, lnp_b = 0;
    std::vector<Layer> layers;
};

WhisperState *ws_of(pce_ctx *c)
{
    if (!c->whisper) c->whisper = new WhisperState();
    return static_cast<WhisperState *>(c->whisper);
}

int mel_setup(pce_ctx *c, WhisperState *w, int n_mels)
{
    if (w->mel_nmels == n_mels) return PCE_OK;
    std::vector<float> dense; mel_filterbank(n_mels, dense);
    std::vector<int> lo((size_t)n_mels), cnt((size_t)n_mels), off((size_t)n_mels);
    std::vector<float> packed;
    for (int i = 0; i < n_mels; i++) {
        int a = 0, b = W_BINS;
        while (a < W_BINS && dense[(size_t)i * W_BINS + a] == 0.f) a++;
        while (b > a && dense[(size_t)i * W_BINS + b - 1] == 0.f) b--;
        lo[(size_t)i] = a; cnt[(size_t)i] = b - a; off[(size_t)i] = (int)packed.size();
        for (int k = a; k < b; k++) packed.push_back(dense[(size_t)i * W_BINS + k]);
    }
    std::vector<float> w16(32), w25(50), w400(800), win(W_NFFT);
    for (int m = 0; m < 16; m++) { w16[2 * (size_t)m] = (float)std::cos(2 * W_PI * m / 16); w16[2 * (size_t)m + 1] = (float)-std::sin(2 * W_PI * m / 16); }
    for (int m = 0; m < 25; m++) { w25[2 * (size_t)m] = (float)std::cos(2 * W_PI * m / 25); w25[2 * (size_t)m + 1] = (float)-std::sin(2 * W_PI * m / 25); }
    for (int n2 = 0; n2 < 25; n2++)
        for (int k1 = 0; k1 < 16; k1++) {
            const double a = 2 * W_PI * (n2 * k1) / 400.0;
            w400[2 * (size_t)(n2 * 16 + k1)] = (float)std::cos(a); w400[2 * (size_t)(n2 * 16 + k1) + 1] = (float)-std::sin(a);
        }
    for (int n = 0; n < W_NFFT; n++) win[(size_t)n] = (float)(0.5 - 0.5 * std::cos(2 * W_PI * n / W_NFFT));
    // one buffer: [w16 | w400 | w25 | window | lo | cnt | off | packed]
    const size_t bytes = sizeof(float) * (32 + 800 + 50 + W_NFFT + packed.size()) + sizeof(int) * 3 * (size_t)n_mels + 64;
    PCE_HIP(c, w->tables.reserve(bytes));
    char *d = w->tables.as<char>(); size_t o = 0;
    auto put = [&](const void *src, size_t n) -> const void * {
        (void)hipMemcpyAsync(d + o, src, n, hipMemcpyHostToDevice, c->stream);
        const void *p = d + o; o += (n + 15) & ~(size_t)15; return p;
    };
    w->mt.w16 = (const float2 *)put(w16.data(), sizeof(float) * 32);
    w->mt.w400 = (const float2 *)put(w400.data(), sizeof(float) * 800);
    w->mt.w25 = (const float2 *)put(w25.data(), sizeof(float) * 50);
    w->mt.window = (const float *)put(win.data(), sizeof(float) * W_NFFT);
    w->mt.mel_lo = (const int *)put(lo.data(), sizeof(int) * (size_t)n_mels);
    w->mt.mel_n = (const int *)put(cnt.data(), sizeof(int) * (size_t)n_mels);
    w->mt.mel_off = (const int *)put(off.data(), sizeof(int) * (size_t)n_mels);
    w->mt.mel_w = (const float *)put(packed.data(), sizeof(float) * packed.size());
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    w->mel_nmels = n_mels;
    return PCE_OK;
}

// Does every byte offset k_gemm_flat forms stay below 2^32?  rows_pad = the rows its padded tile list reaches.
static bool gemm_flat_offsets_fit(int64_t M, int64_t N, int64_t K, int64_t ldc, int64_t S, int64_t vt_sp)
{
    const int64_t tiles_m = (M + F_T - 1) / F_T, rows_pad = ((tiles_m + 15) / 16) * 16 * F_T, lim = 1ll << 32;
    if (rows_pad * K * 2 >= lim || N * K * 2 >= lim) return false;
    if (S > 0) return (rows_pad / S + 2) * N * vt_sp * 2 < lim;           // V^T image: one clip block per S rows (+ the clip a tile may spill into)
    return rows_pad * ldc * 2 < lim;
}

// Persistent 256 x 256 GEMM (pce_gemm256.inc) for the big projections: A dense [M][K], C bf16.  Returns false when the shape does not fit
// (the caller then takes the tiled kernels).
template <int EPI>
bool launch_gemm_flat(pce_ctx *c, const bf16 *A, const bf16 *B, const float *bias, bf16 *C, int M, int N, int K, int ldc, int S = 1, int vt_sp = 0,
                      int prof_id = PCE_K_GEMM_FLAT, bf16 *C2 = nullptr, int vt_n0 = 0)
{
    // 32-bit byte offsets: the kernel addresses the row tiles up to the PADDED tile count (16 row tiles per supertile); rows past M must
    // fall outside the buffer resources (loads read zeros, stores are dropped), which only holds while their offsets do not wrap
    constexpr bool HAS_VT = EPI == FEPI_VT || EPI == FEPI_SPLIT, HAS_RM = EPI != FEPI_VT;
    if (!c->gemm_flat || N % F_T || K % F_K || M < 2048 || N > 6144 || (HAS_VT && (S % 4 || S < F_T))) return false;
    if (EPI == FEPI_SPLIT && (!C2 || vt_n0 <= 0 || vt_n0 >= N || vt_n0 % F_T)) return false;
    if (HAS_RM && !gemm_flat_offsets_fit(M, N, K, ldc, 0, 0)) return false;
    if (HAS_VT && !gemm_flat_offsets_fit(M, N - vt_n0, K, 0, S, vt_sp)) return false;
    FArgs P{};
    P.A = A; P.B = B; P.bias = bias; P.C = C; P.M = M; P.N = N; P.K = K; P.ldc = ldc; P.S = S; P.vt_sp = vt_sp; P.C2 = C2; P.vt_n0 = vt_n0;
    const int tiles_n = N / F_T;
    P.sn = 1;
    for (int cand : {4, 3, 2}) if (tiles_n % cand == 0) { P.sn = cand; break; }
    P.sm = 16; P.stagger = 20000;
    const int lds = F_RING_BYTES + N * (int)sizeof(float);
    if (!c->gemm_flat_attr[EPI]) {                           // once per context (the attribute is per device) and epilogue: the ring + the widest bias vector the shape test admits
        if (hipFuncSetAttribute((const void *)k_gemm_flat<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, F_RING_BYTES + 6144 * (int)sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        c->gemm_flat_attr[EPI] = true;
    }
    const int grid = ((c->cu_count > 0 ? c->cu_count : 256) / 8) * 8;
    KernelTimer kt(c, prof_id, nullptr, 2.0 * M * (double)N * K);
    hipLaunchKernelGGL((k_gemm_flat<EPI>), dim3((unsigned)grid), dim3(F_THREADS), lds, c->stream, P);
    return true;
}

static unsigned long long *g_gemm_trace = nullptr;          // debugging aid: per-workgroup s_memtime stamps (tools only)
template <int EPI>
void launch_gemm(pce_ctx *c, const bf16 *A, int64_t lda, int64_t a_batch, const bf16 *B, int M, int N, int K, const float *bias,
                 void *C, int64_t ldc, int64_t c_batch, int batch, const float *pos = nullptr, int pos_T = 1, int v_col0 = 0,
                 int vt_sp = AT_SP)
{
    const int tiles_n = N / G_BN;
    int sn = 1;
    for (int cand : {8, 6, 4, 3, 2}) if (tiles_n % cand == 0) { sn = cand; break; }        // supertile width (divides the N tiles)
    const int sm_env = c->gemm_sm, sn_env = c->gemm_sn, wide_env = c->gemm_wide;       // diagnostics, read once at pce_create
    const bool wide_ok = N % W_BN == 0 && K % W_BK == 0 && (int64_t)M * batch >= 4096 && (EPI != EPI_QKV || v_col0 % W_BN == 0);
    if (wide_ok && (wide_env > 0 || (wide_env < 0 && N >= 1536))) {      // wide outputs (QKV, fc1; from N = 1536 so that the tiny model exercises it in the tests)
        const int wt = N / W_BN;
        int wsn = 1;
        for (int cand : {4, 3, 2}) if (wt % cand == 0) { wsn = cand; break; }
        const int wsm = 16;
        dim3 wgrid((unsigned)wt, (unsigned)(div_up(M, G_BM * wsm) * wsm), (unsigned)batch);
        KernelTimer kt(c, PCE_K_GEMM_WIDE, nullptr, 2.0 * M * (double)N * K * batch);
        hipLaunchKernelGGL((k_gemm_wide<EPI>), wgrid, dim3(G_THREADS), 0, c->stream, A, lda, a_batch, B, M, N, K, bias, C, ldc, c_batch, pos, pos_T,
                           v_col0, vt_sp, wsn, wsm);
        return;
    }
    int sm = sm_env > 0 ? sm_env : 16;                     // 16 x sn measured best by a hair (645-656 TFLOP/s over 4..16 x 2..8)
    if (sn_env > 0 && tiles_n % sn_env == 0) sn = sn_env;
    dim3 grid((unsigned)tiles_n, (unsigned)(div_up(M, G_BM * sm) * sm), (unsigned)batch);
    const bool want_trace = c->gemm_trace;
    if (want_trace) {
        const size_t nblk = (size_t)grid.x * grid.y;
        (void)hipMalloc(&g_gemm_trace, nblk * 32); (void)hipMemsetAsync(g_gemm_trace, 0, nblk * 32, c->stream);
    }
    {
        KernelTimer kt(c, PCE_K_GEMM128, nullptr, 2.0 * M * (double)N * K * batch);
        if ((int64_t)M * batch <= 1024 && K >= 4 * G_BK && !want_trace)      // a few rows (incremental decoding step): few workgroups, each alone on its CU
            hipLaunchKernelGGL((k_gemm_bf16<EPI, 4>), grid, dim3(G_THREADS), 0, c->stream, A, lda, a_batch, B, M, N, K, bias, C, ldc, c_batch, pos, pos_T,
                               v_col0, vt_sp, sn, sm, g_gemm_trace);
        else
            hipLaunchKernelGGL((k_gemm_bf16<EPI>), grid, dim3(G_THREADS), 0, c->stream, A, lda, a_batch, B, M, N, K, bias, C, ldc, c_batch, pos, pos_T,
                               v_col0, vt_sp, sn, sm, g_gemm_trace);
    }
    if (want_trace) {
        const size_t nblk = (size_t)grid.x * grid.y;
        std::vector<unsigned long long> h(nblk * 4);
        (void)hipStreamSynchronize(c->stream);
        (void)hipMemcpy(h.data(), g_gemm_trace, nblk * 32, hipMemcpyDeviceToHost);
        double a = 0, b = 0, e = 0; size_t n = 0; unsigned long long lo = ~0ull, hi = 0;
        for (size_t i = 0; i < nblk; i++) {
            if (!h[4 * i + 3]) continue;
            a += (double)(h[4 * i + 1] - h[4 * i]); b += (double)(h[4 * i + 2] - h[4 * i + 1]); e += (double)(h[4 * i + 3] - h[4 * i + 2]); n++;
            if (h[4 * i] < lo) lo = h[4 * i];
            if (h[4 * i + 3] > hi) hi = h[4 * i + 3];
        }
        if (n) fprintf(stderr, "gemm EPI %d M %d N %d K %d: %zu tiles, ticks/tile: k-loop %.0f, acc->lds %.0f, store %.0f; span %llu ticks\n", EPI, M, N, K, n,
                       a / n, b / n, e / n, hi - lo);
        (void)hipFree(g_gemm_trace); g_gemm_trace = nullptr;
    }
}

} // namespace

void pce_whisper_free(pce_ctx *c)
{
    if (!c->whisper) return;
    WhisperState *w = static_cast<WhisperState *>(c->whisper);
    DevBuf *bufs[] = {&w->tables, &w->logspec, &w->clipmax, &w->mel_tm, &w->mel_start, &w->w_bf16, &w->w_f32, &w->pos, &w->c1_out, &w->resid,
                      &w->ln_out, &w->qkv, &w->vt, &w->attn, &w->enc_tab, &w->delta, &w->delta2, &w->dw_bf16, &w->dw_f32, &w->d_tok_emb, &w->d_pos_emb, &w->d_tab,
                      &w->d_tokens, &w->d_resid, &w->d_ln, &w->d_qk, &w->d_vt, &w->d_attn, &w->d_q, &w->d_hidden, &w->d_enc_bf16, &w->d_aw,
                      &w->d_cost, &w->d_trace, &w->d_pi, &w->d_pj, &w->d_pl, &w->d_heads, &w->hidden, &w->final_out,
                      &w->g_sk, &w->g_svt, &w->g_c_resid, &w->g_c_ln, &w->g_c_qkv, &w->g_c_attn, &w->g_c_q, &w->g_c_hidden, &w->g_c_tab, &w->g_loop,
                      &w->g_emb_bf16, &w->g_xk, &w->g_xvt, &w->g_last, &w->g_lastln, &w->g_logits, &w->g_mask, &w->g_next,
                      &w->bert.w_bf16, &w->bert.w_f32, &w->bert.word, &w->bert.pos, &w->bert.type0, &w->bert.tab, &w->bert.tokens, &w->bert.resid,
                      &w->bert.ln, &w->bert.qk, &w->bert.vt, &w->bert.attn, &w->bert.hidden, &w->bert.logits};
    for (auto b : bufs) b->release();
    delete w;
    c->whisper = nullptr;
}

extern "C" {

static int logmel_run_impl(pce_ctx *c, int32_t n_mels, const int64_t *start_frames);
int pce_logmel_run(pce_ctx *c, int32_t n_mels) { return logmel_run_impl(c, n_mels, nullptr); }
int pce_logmel_run_at(pce_ctx *c, int32_t n_mels, const int64_t *start_frames)
{
    if (!start_frames) return PCE_E_INVALID;
    return logmel_run_impl(c, n_mels, start_frames);
}
static int logmel_run_impl(pce_ctx *c, int32_t n_mels, const int64_t *start_frames)
{
    if (!c) return PCE_E_INVALID;
    if (!c->d_pcm) return pce_fail(c, PCE_E_STATE, "no batch uploaded");
    if (c->rate != 16000) return pce_fail(c, PCE_E_INVALID, "log-mel needs 16 kHz audio (got %d Hz): resample on the host first", c->rate);
    if (n_mels <= 0 || n_mels > 128) return pce_fail(c, PCE_E_INVALID, "n_mels %d out of range", n_mels);
    PCE_HIP(c, hipSetDevice(c->device));
    WhisperState *w = ws_of(c);
    int rc = mel_setup(c, w, n_mels);
    if (rc) return rc;
    const int32_t n = c->n_clips;
    const size_t tm_elems = (size_t)n * (W_FRAMES + 2) * (size_t)n_mels + 512;
    PCE_HIP(c, w->logspec.reserve(sizeof(float) * (size_t)n * (size_t)n_mels * W_FRAMES));
    PCE_HIP(c, w->clipmax.reserve(sizeof(unsigned int) * (size_t)n));
    PCE_HIP(c, w->mel_tm.reserve(sizeof(bf16) * tm_elems));
    PCE_HIP(c, hipMemsetAsync(w->clipmax.p, 0, sizeof(unsigned int) * (size_t)n, c->stream));
    PCE_HIP(c, hipMemsetAsync(w->mel_tm.p, 0, sizeof(bf16) * tm_elems, c->stream));
    const int64_t *d_start = nullptr;
    if (start_frames) {
        int64_t longest = 0;
        for (int32_t i = 0; i < n; i++) {
            const int64_t len = c->clip_off[(size_t)i + 1] - c->clip_off[(size_t)i];
            if (start_frames[i] < 0 || start_frames[i] * W_HOP > len) return pce_fail(c, PCE_E_INVALID, "clip %d: window start %lld frames is past the audio", i, (long long)start_frames[i]);
            longest = std::max(longest, len);
        }
        PCE_HIP(c, w->mel_start.reserve(sizeof(int64_t) * (size_t)(n > 0 ? n : 1)));
        PCE_HIP(c, hipMemcpyAsync(w->mel_start.p, start_frames, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, c->stream));
        PCE_HIP(c, hipStreamSynchronize(c->stream));
        d_start = w->mel_start.as<int64_t>();
        if (n > 0) {       // the clamp of log_mel_spectrogram uses the maximum over the whole recording
            const int64_t scan_frames = (longest + W_NFFT / 2) / W_HOP + 1;
            hipLaunchKernelGGL(k_logmel_frames, dim3((unsigned)std::min<int64_t>(div_up(scan_frames, 4 * 5), 4096), (unsigned)n), dim3(256), 0, c->stream, c->d_pcm,
                               c->d_clip_off.as<int64_t>(), (int)n_mels, w->mt, w->logspec.as<float>(), w->clipmax.as<unsigned int>(), d_start, 1);
        }
    }
    {
        KernelTimer t(c, PCE_K_LOGMEL);
        hipLaunchKernelGGL(k_logmel_frames, dim3(W_FRAMES / 4 / 5, (unsigned)n), dim3(256), 0, c->stream, c->d_pcm,
                           c->d_clip_off.as<int64_t>(), (int)n_mels, w->mt, w->logspec.as<float>(), w->clipmax.as<unsigned int>(), d_start, 0);
    }
    {
        KernelTimer t(c, PCE_K_LOGMEL_NORM);
        hipLaunchKernelGGL(k_logmel_norm, dim3(div_up(W_FRAMES, 64), div_up(n_mels, 64), (unsigned)n), dim3(256), 0, c->stream,
                           w->logspec.as<float>(), w->clipmax.as<unsigned int>(), (int)n_mels, w->mel_tm.as<bf16>());
    }
    PCE_HIP(c, hipGetLastError());
    w->n_clips_mel = n;
    return PCE_OK;
}

int pce_logmel_fetch(pce_ctx *c, int32_t clip, float *out)
{
    if (!c || !out) return PCE_E_INVALID;
    WhisperState *w = ws_of(c);
    if (w->n_clips_mel < 0) return pce_fail(c, PCE_E_STATE, "pce_logmel_fetch before pce_logmel_run");
    if (clip < 0 || clip >= w->n_clips_mel) return pce_fail(c, PCE_E_INVALID, "clip out of range");
    const size_t per = (size_t)w->mel_nmels * W_FRAMES;
    PCE_HIP(c, hipMemcpyAsync(out, w->logspec.as<float>() + per * (size_t)clip, sizeof(float) * per, hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    pce_profile_collect(c);
    return PCE_OK;
}

int pce_whisper_load(pce_ctx *c, const pce_whisper_dims *dims, const float *weights, int64_t n_floats)
{
    if (!c || !dims || !weights) return PCE_E_INVALID;
    const int d = dims->n_state, L = dims->n_layer, nm = dims->n_mels;
    if (d <= 0 || d % 128 || dims->n_head * 64 != d || L <= 0 || dims->n_ctx != W_CTX || nm <= 0 || nm > 128 || (nm % 8))
        return pce_fail(c, PCE_E_LIMIT, "unsupported encoder dims (need n_state %% 128 == 0, head size 64, n_ctx 1500, n_mels %% 8 == 0)");
    const int64_t per_layer = 2LL * d + 4LL * d * d + 3LL * d + 2LL * d + 8LL * d * d + 5LL * d;
    const int64_t expect = (int64_t)d * nm * 3 + d + 3LL * d * d + d + L * per_layer + 2LL * d;
    if (n_floats != expect) return pce_fail(c, PCE_E_INVALID, "weight blob has %lld floats, expected %lld", (long long)n_floats, (long long)expect);
    PCE_HIP(c, hipSetDevice(c->device));
    WhisperState *w = ws_of(c);
    w->dims = *dims; w->loaded = false; w->layers.assign((size_t)L, {});
    // host repack: bf16-bound matrices (as fp32, converted on the device) and fp32 vectors
    std::vector<float> mats, vecs;
    auto add_vec = [&](const float *p, size_t n) { size_t o = vecs.size(); vecs.insert(vecs.end(), p, p + n); return o; };
    auto add_zero_vec = [&](size_t n) { size_t o = vecs.size(); vecs.insert(vecs.end(), n, 0.f); return o; };
    const float *p = weights;
    {   // conv1 [d][nm][3] -> [d][K1p] with k-major columns (k*nm + ci), zero padded to a multiple of 64
        const int K1 = 3 * nm, K1p = (int)div_up(K1, 64) * 64;
        w->c1_w = mats.size(); mats.resize(mats.size() + (size_t)d * K1p, 0.f);
        for (int co = 0; co < d; co++)
            for (int ci = 0; ci < nm; ci++)
                for (int k = 0; k < 3; k++) mats[w->c1_w + (size_t)co * K1p + (size_t)k * nm + ci] = p[((size_t)co * nm + ci) * 3 + k];
        p += (size_t)d * nm * 3;
        w->c1_b = add_vec(p, (size_t)d); p += d;
    }
    {   // conv2 [d][d][3] -> [d][3 d]
        w->c2_w = mats.size(); mats.resize(mats.size() + (size_t)d * 3 * d, 0.f);
        for (int co = 0; co < d; co++)
            for (int ci = 0; ci < d; ci++)
                for (int k = 0; k < 3; k++) mats[w->c2_w + (size_t)co * 3 * d + (size_t)k * d + ci] = p[((size_t)co * d + ci) * 3 + k];
        p += (size_t)d * d * 3;
        w->c2_b = add_vec(p, (size_t)d); p += d;
    }
    for (int l = 0; l < L; l++) {
        WhisperState::Layer &ly = w->layers[(size_t)l];
        ly.ln1_w = add_vec(p, (size_t)d); p += d; ly.ln1_b = add_vec(p, (size_t)d); p += d;
        // q.w q.b k.w v.w v.b -> fused [3d][d], bias [q.b | 0 | v.b]
        ly.qkv_w = mats.size(); mats.resize(mats.size() + (size_t)3 * d * d);
        const float *qw = p, *qb = p + (size_t)d * d, *kw = qb + d, *vw = kw + (size_t)d * d, *vb = vw + (size_t)d * d;
        std::copy(qw, qw + (size_t)d * d, mats.begin() + (ptrdiff_t)ly.qkv_w);
        std::copy(kw, kw + (size_t)d * d, mats.begin() + (ptrdiff_t)(ly.qkv_w + (size_t)d * d));
        std::copy(vw, vw + (size_t)d * d, mats.begin() + (ptrdiff_t)(ly.qkv_w + 2 * (size_t)d * d));
        ly.qkv_b = add_vec(qb, (size_t)d); add_zero_vec((size_t)d); add_vec(vb, (size_t)d);
        p = vb + d;
        ly.out_w = mats.size(); mats.insert(mats.end(), p, p + (size_t)d * d); p += (size_t)d * d;
        ly.out_b = add_vec(p, (size_t)d); p += d;
        ly.ln2_w = add_vec(p, (size_t)d); p += d; ly.ln2_b = add_vec(p, (size_t)d); p += d;
        ly.m1_w = mats.size(); mats.insert(mats.end(), p, p + (size_t)4 * d * d); p += (size_t)4 * d * d;
        ly.m1_b = add_vec(p, (size_t)4 * d); p += 4 * d;
        ly.m2_w = mats.size(); mats.insert(mats.end(), p, p + (size_t)4 * d * d); p += (size_t)4 * d * d;
        ly.m2_b = add_vec(p, (size_t)d); p += d;
    }
    w->lnp_w = add_vec(p, (size_t)d); p += d; w->lnp_b = add_vec(p, (size_t)d); p += d;
    // sinusoid positions [1500][d]
    std::vector<float> pos((size_t)W_CTX * d);
    {
        const int half = d / 2;
        const double inc = std::log(10000.0) / (half - 1);
        for (int t = 0; t < W_CTX; t++)
            for (int i = 0; i < half; i++) {
                const float inv = (float)std::exp(-inc * i);
                const float a = (float)t * inv;
                pos[(size_t)t * d + i] = std::sin(a); pos[(size_t)t * d + half + i] = std::cos(a);
            }
    }
    DevBuf tmp;
    PCE_HIP(c, tmp.reserve(sizeof(float) * mats.size()));
    PCE_HIP(c, w->w_bf16.reserve(sizeof(bf16) * mats.size() + 256));
    PCE_HIP(c, w->w_f32.reserve(sizeof(float) * vecs.size()));
    PCE_HIP(c, w->pos.reserve(sizeof(float) * pos.size()));
    PCE_HIP(c, hipMemcpyAsync(tmp.p, mats.data(), sizeof(float) * mats.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(w->w_f32.p, vecs.data(), sizeof(float) * vecs.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(w->pos.p, pos.data(), sizeof(float) * pos.size(), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_f32_to_bf16, dim3((unsigned)div_up((int64_t)mats.size(), 256)), dim3(256), 0, c->stream, tmp.as<float>(),
                       w->w_bf16.as<bf16>(), (int64_t)mats.size());
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    tmp.release();
    w->loaded = true;
    return PCE_OK;
}

int pce_whisper_encode_run(pce_ctx *c)
{
    if (!c) return PCE_E_INVALID;
    WhisperState *w = ws_of(c);
    if (!w->loaded) return pce_fail(c, PCE_E_STATE, "pce_whisper_encode_run before pce_whisper_load");
    if (w->n_clips_mel < 0 || w->mel_nmels != w->dims.n_mels) return pce_fail(c, PCE_E_STATE, "run pce_logmel_run(n_mels of the model) first");
    PCE_HIP(c, hipSetDevice(c->device));
    const int d = w->dims.n_state, L = w->dims.n_layer, nm = w->dims.n_mels, H = w->dims.n_head;
    const int n = w->n_clips_mel;
    const int64_t M = (int64_t)n * W_CTX;
    if (M > INT32_MAX) return pce_fail(c, PCE_E_LIMIT, "too many clips for one encoder batch");
    const int K1p = (int)div_up(3 * nm, 64) * 64;
    const size_t c1_elems = (size_t)n * (W_FRAMES + 2) * (size_t)d + 4096;
    PCE_HIP(c, w->c1_out.reserve(sizeof(bf16) * c1_elems));
    PCE_HIP(c, w->resid.reserve(sizeof(float) * (size_t)M * d));
    PCE_HIP(c, w->ln_out.reserve(sizeof(bf16) * (size_t)M * d));
    PCE_HIP(c, w->qkv.reserve(sizeof(bf16) * (size_t)M * 2 * d));            // q | k, row-major
    const size_t vt_elems = (size_t)n * (size_t)d * AT_SP + 64;          // V^T [clip][head][64][AT_SP], pad keys stay zero
    {
        const size_t before = w->vt.cap;
        PCE_HIP(c, w->vt.reserve(sizeof(bf16) * vt_elems));
        if (w->vt.cap != before || w->vt_elems_zeroed < vt_elems) {
            PCE_HIP(c, hipMemsetAsync(w->vt.p, 0, sizeof(bf16) * vt_elems, c->stream));
            w->vt_elems_zeroed = vt_elems;
        }
    }
    PCE_HIP(c, w->attn.reserve(sizeof(bf16) * (size_t)M * d));
    PCE_HIP(c, w->hidden.reserve(sizeof(bf16) * (size_t)M * 4 * d));
    PCE_HIP(c, w->final_out.reserve(sizeof(float) * (size_t)M * d));
    const bf16 *Wb = w->w_bf16.as<bf16>();
    const float *Wf = w->w_f32.as<float>();
    if (w->enc_tab_clips != n) {   // per-clip row tables of the attention descriptor: clip c owns rows [1500 c, 1500 c + 1500); they only depend on the
                                   // batch size, so a steady stream of equal batches uploads (and waits for) them once
        std::vector<int> tab((size_t)2 * n);
        for (int i = 0; i < n; i++) { tab[(size_t)i] = i * W_CTX; tab[(size_t)n + i] = W_CTX; }
        PCE_HIP(c, w->enc_tab.reserve(sizeof(int) * tab.size()));
        PCE_HIP(c, hipMemcpyAsync(w->enc_tab.p, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice, c->stream));
        PCE_HIP(c, hipStreamSynchronize(c->stream));
        w->enc_tab_clips = n;
    }
    KernelTimer t(c, PCE_K_WHISPER_ENC);
    PCE_HIP(c, hipMemsetAsync(w->c1_out.p, 0, sizeof(bf16) * c1_elems, c->stream));
    // conv1: per clip, A row t starts at padded row t (= t-1 unpadded), K = 3 n_mels (padded to K1p with zero weights)
    launch_gemm<EPI_GELU_BF16>(c, w->mel_tm.as<bf16>(), nm, (int64_t)(W_FRAMES + 2) * nm, Wb + w->c1_w, W_FRAMES, d, K1p, Wf + w->c1_b,
                               w->c1_out.as<bf16>() + d, d, (int64_t)(W_FRAMES + 2) * d, n);
    // conv2 (stride 2): A row t' starts at padded row 2 t', K = 3 d, lda = 2 d; epilogue adds the positional embedding
    launch_gemm<EPI_GELU_POS_F32>(c, w->c1_out.as<bf16>(), 2 * (int64_t)d, (int64_t)(W_FRAMES + 2) * d, Wb + w->c2_w, W_CTX, d, 3 * d,
                                  Wf + w->c2_b, w->resid.as<float>(), d, (int64_t)W_CTX * d, n, w->pos.as<float>(), W_CTX);
    // The big projections run on the persistent 256 x 256 kernel when the shape allows it (n_state % 256 == 0, a batch of at least a few
    // clips).  On that path a branch (attention projection, MLP) leaves its output as a bf16 row block and the residual add is fused
    // into the LayerNorm that follows (k_add_layernorm): one pass over the residual stream instead of the GEMM's read-modify-write
    // plus the LayerNorm's read.
    const bool flat = c->gemm_flat && d % F_T == 0 && M >= 2048 && gemm_flat_offsets_fit(M, 4 * d, 4 * d, 4 * d, 0, 0) &&
                      gemm_flat_offsets_fit(M, d, d, 0, W_CTX, AT_SP);
    if (flat) {
        PCE_HIP(c, w->delta.reserve(sizeof(bf16) * (size_t)M * d)); PCE_HIP(c, w->delta2.reserve(sizeof(bf16) * (size_t)M * d));
        PCE_HIP(c, w->d_enc_bf16.reserve(sizeof(bf16) * (size_t)M * d + 4096));
    }
    // mid = the pass after the attention projection (x = resid + delta feeds ln2, the stream is not written), else the pass at the end of
    // the layer (x = (resid + delta) + delta2 is written back and feeds the next layer's ln1, or ln_post -> the fp32 encoder output)
    auto add_ln = [&](size_t w_off, size_t b_off, bool mid, bool last) {
        KernelTimer kt(c, PCE_K_ADD_LAYERNORM);
        const bf16 *d2 = mid ? nullptr : w->delta2.as<bf16>();
        if (last)
            hipLaunchKernelGGL((k_add_layernorm<float>), dim3((unsigned)div_up(M, 4)), dim3(256), 0, c->stream, w->resid.as<float>(), w->delta.as<bf16>(), d2, 1,
                               Wf + w_off, Wf + b_off, M, d, w->final_out.as<float>(), 1e-5f, w->d_enc_bf16.as<bf16>());   // + the bf16 copy the decoder's cross K / V projections read
        else
            hipLaunchKernelGGL((k_add_layernorm<bf16>), dim3((unsigned)div_up(M, 4)), dim3(256), 0, c->stream, w->resid.as<float>(), w->delta.as<bf16>(), d2,
                               mid ? 0 : 1, Wf + w_off, Wf + b_off, M, d, w->ln_out.as<bf16>(), 1e-5f);
    };
    for (int l = 0; l < L; l++) {
        const WhisperState::Layer &ly = w->layers[(size_t)l];
        if (!flat || l == 0) {
            KernelTimer kt(c, PCE_K_LAYERNORM);
            hipLaunchKernelGGL((k_layernorm<bf16>), dim3((unsigned)div_up(M, 4)), dim3(256), 0, c->stream, w->resid.as<float>(), Wf + ly.ln1_w,
                               Wf + ly.ln1_b, M, d, w->ln_out.as<bf16>());
        }
        // Q | K go to the row-major [M][2d] buffer, V is written transposed per head: one launch sweeps the LayerNorm output once
        const bool done = flat && launch_gemm_flat<FEPI_SPLIT>(c, w->ln_out.as<bf16>(), Wb + ly.qkv_w, Wf + ly.qkv_b, w->qkv.as<bf16>(), (int)M, 3 * d, d, 2 * d,
                                                                W_CTX, AT_SP, PCE_K_GEMM_FLAT_QKV, w->vt.as<bf16>(), 2 * d);
        if (flat && !done) return pce_fail(c, PCE_E_LIMIT, "Q | K | V projection does not fit the 256 x 256 kernel");
        if (!done)
            launch_gemm<EPI_QKV>(c, w->ln_out.as<bf16>(), d, 0, Wb + ly.qkv_w, (int)M, 3 * d, d, Wf + ly.qkv_b, w->qkv.as<bf16>(), 2 * d, 0, 1,
                                 reinterpret_cast<const float *>(w->vt.as<bf16>()), W_CTX, 2 * d, AT_SP);
        {
            AttnArgs a{};
            a.q = w->qkv.as<bf16>(); a.q_ld = 2 * d; a.k = w->qkv.as<bf16>() + d; a.k_ld = 2 * d;
            a.vt = w->vt.as<bf16>(); a.vt_clip = (int64_t)d * AT_SP; a.vt_sp = AT_SP;
            a.q_row0 = a.k_row0 = w->enc_tab.as<int>(); a.q_len = a.k_len = w->enc_tab.as<int>() + n;
            a.out = w->attn.as<bf16>(); a.out_ld = d; a.causal = 0;
            launch_attention(c, dim3((unsigned)div_up(W_CTX, AT_QB), (unsigned)H, (unsigned)n), a, 4.0 * W_CTX * (double)W_CTX * d * n);
        }
        if (flat) {
            if (!launch_gemm_flat<FEPI_BF16>(c, w->attn.as<bf16>(), Wb + ly.out_w, Wf + ly.out_b, w->delta.as<bf16>(), (int)M, d, d, d, 1, 0, PCE_K_GEMM_FLAT_OUT))
                return pce_fail(c, PCE_E_LIMIT, "attention projection does not fit the 256 x 256 kernel");
            add_ln(ly.ln2_w, ly.ln2_b, true, false);
            if (!launch_gemm_flat<FEPI_GELU>(c, w->ln_out.as<bf16>(), Wb + ly.m1_w, Wf + ly.m1_b, w->hidden.as<bf16>(), (int)M, 4 * d, d, 4 * d, 1, 0, PCE_K_GEMM_FLAT_FC1) ||
                !launch_gemm_flat<FEPI_BF16>(c, w->hidden.as<bf16>(), Wb + ly.m2_w, Wf + ly.m2_b, w->delta2.as<bf16>(), (int)M, d, 4 * d, d, 1, 0, PCE_K_GEMM_FLAT_FC2))
                return pce_fail(c, PCE_E_LIMIT, "MLP does not fit the 256 x 256 kernel");
            if (l + 1 < L) add_ln(w->layers[(size_t)l + 1].ln1_w, w->layers[(size_t)l + 1].ln1_b, false, false);
            else add_ln(w->lnp_w, w->lnp_b, false, true);
            continue;
        }
        launch_gemm<EPI_RESID_F32>(c, w->attn.as<bf16>(), d, 0, Wb + ly.out_w, (int)M, d, d, Wf + ly.out_b, w->resid.as<float>(), d, 0, 1);
        {
            KernelTimer kt(c, PCE_K_LAYERNORM);
            hipLaunchKernelGGL((k_layernorm<bf16>), dim3((unsigned)div_up(M, 4)), dim3(256), 0, c->stream, w->resid.as<float>(), Wf + ly.ln2_w,
                               Wf + ly.ln2_b, M, d, w->ln_out.as<bf16>());
        }
        launch_gemm<EPI_GELU_BF16>(c, w->ln_out.as<bf16>(), d, 0, Wb + ly.m1_w, (int)M, 4 * d, d, Wf + ly.m1_b, w->hidden.as<bf16>(), 4 * d, 0, 1);
        launch_gemm<EPI_RESID_F32>(c, w->hidden.as<bf16>(), 4 * d, 0, Wb + ly.m2_w, (int)M, d, 4 * d, Wf + ly.m2_b, w->resid.as<float>(), d, 0, 1);
    }
    if (!flat) {
        KernelTimer kt(c, PCE_K_LAYERNORM);
        hipLaunchKernelGGL((k_layernorm<float>), dim3((unsigned)div_up(M, 4)), dim3(256), 0, c->stream, w->resid.as<float>(), Wf + w->lnp_w,
                           Wf + w->lnp_b, M, d, w->final_out.as<float>());
    }
    PCE_HIP(c, hipGetLastError());
    w->n_clips_enc = n; w->g_xkv_clips = -1; w->g_cache_len = -1; w->enc_bf16_clips = flat ? n : -1;
    return PCE_OK;
}

// Cross-attention keys (rows, [Ma][d]) and values (transposed per clip, key axis padded to AT_SP) of one decoder layer from the
// encoded audio: W = [K weights | V weights] (2d rows of d).  On the persistent 256 x 256 kernel when the shape fits.
static void project_cross_kv(pce_ctx *c, const bf16 *enc, int Ma, int d, const bf16 *W, const float *bias, bf16 *xk, bf16 *xvt)
{
    // one launch: K columns leave row-major, V columns as the transposed image (the 590 MB of encoded audio are swept once)
    if (launch_gemm_flat<FEPI_SPLIT>(c, enc, W, bias, xk, Ma, 2 * d, d, d, W_CTX, AT_SP, PCE_K_GEMM_FLAT_XKV, xvt, d)) return;
    launch_gemm<EPI_QKV>(c, enc, d, 0, W, Ma, 2 * d, d, bias, xk, d, 0, 1, reinterpret_cast<const float *>(xvt), W_CTX, d, AT_SP);
}

// V rows [clips][k_len][heads * 64] -> the V^T image the attention kernels read: [clip][head * 64 + d][sp] (key axis padded with zeros)
__global__ void k_selftest_vt(const bf16 *__restrict__ v, int k_len, int hd, int sp, int64_t n, bf16 *__restrict__ vt)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int col = (int)(i % hd); const int64_t r = i / hd; const int t = (int)(r % k_len); const int64_t clip = r / k_len;
    vt[(clip * hd + col) * sp + t] = v[i];
}

// Self-test hook of the attention kernels (see pce.h)
int pce_selftest_attention(pce_ctx *c, const uint16_t *q, const uint16_t *k, const uint16_t *v, int32_t clips, int32_t heads, int32_t q_len,
                           int32_t k_len, int32_t causal, int32_t mode, uint16_t *out, int32_t *fell_back)
{
    if (!c || !q || !k || !v || !out || clips <= 0 || heads <= 0 || q_len <= 0 || k_len <= 0 || mode < 0 || mode > 2) return PCE_E_INVALID;
    PCE_HIP(c, hipSetDevice(c->device));
    const int hd = heads * 64, sp = div_up(k_len, 64) * 64;
    const size_t nq = (size_t)clips * q_len * hd, nk = (size_t)clips * k_len * hd, nvt = (size_t)clips * hd * sp;
    DevBuf dq, dk, dv, dvt, dout, dtab, dcnt;
    PCE_HIP(c, dq.reserve(nq * 2 + 64)); PCE_HIP(c, dk.reserve(nk * 2 + 64)); PCE_HIP(c, dv.reserve(nk * 2 + 64)); PCE_HIP(c, dvt.reserve(nvt * 2 + 128));
    PCE_HIP(c, dout.reserve(nq * 2 + 64)); PCE_HIP(c, dtab.reserve(sizeof(int) * 4 * (size_t)clips)); PCE_HIP(c, dcnt.reserve(sizeof(int)));
    std::vector<int> tab((size_t)4 * clips);
    for (int i = 0; i < clips; i++) { tab[(size_t)i] = i * q_len; tab[(size_t)clips + i] = q_len; tab[(size_t)2 * clips + i] = i * k_len; tab[(size_t)3 * clips + i] = k_len; }
    PCE_HIP(c, hipMemcpyAsync(dq.p, q, nq * 2, hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(dk.p, k, nk * 2, hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(dv.p, v, nk * 2, hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(dtab.p, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemsetAsync(dvt.p, 0, nvt * 2 + 128, c->stream));
    PCE_HIP(c, hipMemsetAsync(dout.p, 0, nq * 2, c->stream));
    PCE_HIP(c, hipMemsetAsync(dcnt.p, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(k_selftest_vt, dim3((unsigned)div_up((int64_t)nk, 256)), dim3(256), 0, c->stream, dv.as<bf16>(), k_len, hd, sp, (int64_t)nk, dvt.as<bf16>());
    AttnArgs a{};
    a.q = dq.as<bf16>(); a.q_ld = hd; a.k = dk.as<bf16>(); a.k_ld = hd; a.vt = dvt.as<bf16>(); a.vt_clip = (int64_t)hd * sp; a.vt_sp = sp;
    a.q_row0 = dtab.as<int>(); a.q_len = a.q_row0 + clips; a.k_row0 = a.q_row0 + 2 * clips; a.k_len = a.q_row0 + 3 * clips;
    a.out = dout.as<bf16>(); a.out_ld = hd; a.causal = causal; a.fell_back = dcnt.as<int>();
    const dim3 grid((unsigned)div_up(q_len, AT_QB), (unsigned)heads, (unsigned)clips);
    if (mode == 2) hipLaunchKernelGGL(k_attention, grid, dim3(256), 0, c->stream, a);
    else hipLaunchKernelGGL(k_attention_lean, grid, dim3(256), 0, c->stream, a, mode);
    PCE_HIP(c, hipGetLastError());
    int cnt = 0;
    PCE_HIP(c, hipMemcpyAsync(out, dout.p, nq * 2, hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipMemcpyAsync(&cnt, dcnt.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    if (fell_back) *fell_back = cnt;
    return PCE_OK;
}

// Self-test hook of the persistent 256 x 256 GEMM: C = epilogue(A B^T + bias) on host arrays (bf16 bit patterns in, bf16 bit patterns out).
// epilogue 0: bias, 1: bias + GELU, 2: bias, written transposed per clip (rows_per_clip rows, key axis padded to vt_sp): out[(clip N + n) vt_sp + t]
int pce_selftest_gemm(pce_ctx *c, const uint16_t *A, const uint16_t *B, const float *bias, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                      int32_t rows_per_clip, int32_t vt_sp, uint16_t *out)
{
    if (!c || !A || !B || !out || M <= 0 || N <= 0 || K <= 0) return PCE_E_INVALID;
    PCE_HIP(c, hipSetDevice(c->device));
    const size_t n_out = epilogue == 2 ? (size_t)(M / rows_per_clip) * N * vt_sp
                         : epilogue >= 256 ? (size_t)M * epilogue + (size_t)(M / rows_per_clip) * (N - epilogue) * vt_sp : (size_t)M * N;
    DevBuf dA, dB, dC, dbias;
    PCE_HIP(c, dA.reserve((size_t)M * K * 2)); PCE_HIP(c, dB.reserve((size_t)N * K * 2)); PCE_HIP(c, dC.reserve(n_out * 2)); PCE_HIP(c, dbias.reserve((size_t)N * 4));
    PCE_HIP(c, hipMemcpyAsync(dA.p, A, (size_t)M * K * 2, hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(dB.p, B, (size_t)N * K * 2, hipMemcpyHostToDevice, c->stream));
    if (bias) PCE_HIP(c, hipMemcpyAsync(dbias.p, bias, (size_t)N * 4, hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemsetAsync(dC.p, 0, n_out * 2, c->stream));
    bool ok = false;
    const float *bp = bias ? dbias.as<float>() : nullptr;
    if (epilogue == 0) ok = launch_gemm_flat<FEPI_BF16>(c, dA.as<bf16>(), dB.as<bf16>(), bp, dC.as<bf16>(), M, N, K, N);
    else if (epilogue == 1) ok = launch_gemm_flat<FEPI_GELU>(c, dA.as<bf16>(), dB.as<bf16>(), bp, dC.as<bf16>(), M, N, K, N);
    else if (epilogue == 2) ok = launch_gemm_flat<FEPI_VT>(c, dA.as<bf16>(), dB.as<bf16>(), bp, dC.as<bf16>(), M, N, K, 0, rows_per_clip, vt_sp);
    else if (epilogue >= 256 && epilogue % 256 == 0 && epilogue < N)      // split launch: columns [0, epilogue) row-major [M][epilogue], then the V^T image of the rest
        ok = launch_gemm_flat<FEPI_SPLIT>(c, dA.as<bf16>(), dB.as<bf16>(), bp, dC.as<bf16>(), M, N, K, epilogue, rows_per_clip, vt_sp, PCE_K_GEMM_FLAT,
                                          dC.as<bf16>() + (size_t)M * epilogue, epilogue);
    int rc = PCE_OK;
    if (!ok) rc = pce_fail(c, PCE_E_LIMIT, "shape not handled by the 256 x 256 kernel (N %% 256, K %% 64, M >= 2048)");
    else {
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(out, dC.p, n_out * 2, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = pce_fail(c, PCE_E_DEVICE, "selftest gemm: %s", hipGetErrorString(e));
    }
    (void)hipStreamSynchronize(c->stream);
    dA.release(); dB.release(); dC.release(); dbias.release();
    pce_profile_collect(c);
    return rc;
}

int pce_whisper_encode_fetch(pce_ctx *c, int32_t clip, float *out)
{
    if (!c || !out) return PCE_E_INVALID;
    WhisperState *w = ws_of(c);
    if (w->n_clips_enc < 0) return pce_fail(c, PCE_E_STATE, "pce_whisper_encode_fetch before pce_whisper_encode_run");
    if (clip < 0 || clip >= w->n_clips_enc) return pce_fail(c, PCE_E_INVALID, "clip out of range");
    const size_t per = (size_t)W_CTX * (size_t)w->dims.n_state;
    PCE_HIP(c, hipMemcpyAsync(out, w->final_out.as<float>() + per * (size_t)clip, sizeof(float) * per, hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    pce_profile_collect(c);
    return PCE_OK;
}

} // extern "C"

int pce_dtw_launch(pce_ctx *c, const double *d_x, int64_t x_stride, int ld, const int *d_rows, const int *d_cols, int N_max, int M_max,
                   int batch, unsigned char *d_trace, int *d_pi, int *d_pj, int *d_pl);

extern "C" {

int pce_whisper_decoder_load(pce_ctx *c, const pce_whisper_text_dims *dims, const float *weights, int64_t n_floats)
{
    if (!c || !dims || !weights) return PCE_E_INVALID;
    const int d = dims->n_state, L = dims->n_layer, V = dims->n_vocab, TC = dims->n_text_ctx;
    if (d <= 0 || d % 128 || dims->n_head * 64 != d || L <= 0 || V <= 0 || TC <= 0 || TC > 448)
        return pce_fail(c, PCE_E_LIMIT, "unsupported decoder dims (need n_state %% 128 == 0, head size 64, n_text_ctx <= 448)");
    const int64_t per_layer = 2LL * d + (4LL * d * d + 3LL * d) + 2LL * d + (4LL * d * d + 3LL * d) + 2LL * d + 8LL * d * d + 5LL * d;
    const int64_t expect = (int64_t)V * d + (int64_t)TC * d + L * per_layer + 2LL * d;
    if (n_floats != expect) return pce_fail(c, PCE_E_INVALID, "decoder weight blob has %lld floats, expected %lld", (long long)n_floats, (long long)expect);
    PCE_HIP(c, hipSetDevice(c->device));
    WhisperState *w = ws_of(c);
    w->tdims = *dims; w->dec_loaded = false; w->dlayers.assign((size_t)L, {});
    std::vector<float> mats, vecs;
    auto add_vec = [&](const float *p, size_t n) { size_t o = vecs.size(); vecs.insert(vecs.end(), p, p + n); return o; };
    auto add_zero = [&](size_t n) { size_t o = vecs.size(); vecs.insert(vecs.end(), n, 0.f); return o; };
    auto add_mat = [&](const float *p, size_t n) { size_t o = mats.size(); mats.insert(mats.end(), p, p + n); return o; };
    const float *p = weights;
    const float *tok = p; p += (size_t)V * d;
    const float *pos = p; p += (size_t)TC * d;
    const size_t dd = (size_t)d * d;
    for (int l = 0; l < L; l++) {
        WhisperState::DLayer &ly = w->dlayers[(size_t)l];
        ly.ln1_w = add_vec(p, (size_t)d); p += d; ly.ln1_b = add_vec(p, (size_t)d); p += d;
        {   // self attention: q.w q.b k.w v.w v.b out.w out.b -> fused [3d][d]
            const float *qw = p, *qb = qw + dd, *kw = qb + d, *vw = kw + dd, *vb = vw + dd;
            ly.qkv_w = add_mat(qw, dd); add_mat(kw, dd); add_mat(vw, dd);
            ly.qkv_b = add_vec(qb, (size_t)d); add_zero((size_t)d); add_vec(vb, (size_t)d);
            p = vb + d;
            ly.out_w = add_mat(p, dd); p += dd; ly.out_b = add_vec(p, (size_t)d); p += d;
        }
        ly.lnx_w = add_vec(p, (size_t)d); p += d; ly.lnx_b = add_vec(p, (size_t)d); p += d;
        {   // cross attention: q from text, k | v from audio -> [d][d] and fused [2d][d]
            const float *qw = p, *qb = qw + dd, *kw = qb + d, *vw = kw + dd, *vb = vw + dd;
            ly.xq_w = add_mat(qw, dd); ly.xq_b = add_vec(qb, (size_t)d);
            ly.xkv_w = add_mat(kw, dd); add_mat(vw, dd);
            ly.xkv_b = add_zero((size_t)d); add_vec(vb, (size_t)d);
            p = vb + d;
            ly.xout_w = add_mat(p, dd); p += dd; ly.xout_b = add_vec(p, (size_t)d); p += d;
        }
        ly.ln2_w = add_vec(p, (size_t)d); p += d; ly.ln2_b = add_vec(p, (size_t)d); p += d;
        ly.m1_w = add_mat(p, 4 * dd); p += 4 * dd; ly.m1_b = add_vec(p, (size_t)4 * d); p += 4 * d;
        ly.m2_w = add_mat(p, 4 * dd); p += 4 * dd; ly.m2_b = add_vec(p, (size_t)d); p += d;
    }
    w->dln_w = add_vec(p, (size_t)d); p += d; w->dln_b = add_vec(p, (size_t)d); p += d;
    DevBuf tmp;
    PCE_HIP(c, tmp.reserve(sizeof(float) * mats.size()));
    PCE_HIP(c, w->dw_bf16.reserve(sizeof(bf16) * mats.size() + 256));
    PCE_HIP(c, w->dw_f32.reserve(sizeof(float) * vecs.size()));
    PCE_HIP(c, w->d_tok_emb.reserve(sizeof(float) * (size_t)V * d));
    PCE_HIP(c, w->d_pos_emb.reserve(sizeof(float) * (size_t)TC * d));
    PCE_HIP(c, hipMemcpyAsync(tmp.p, mats.data(), sizeof(float) * mats.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(w->dw_f32.p, vecs.data(), sizeof(float) * vecs.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(w->d_tok_emb.p, tok, sizeof(float) * (size_t)V * d, hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(w->d_pos_emb.p, pos, sizeof(float) * (size_t)TC * d, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_f32_to_bf16, dim3((unsigned)div_up((int64_t)mats.size(), 256)), dim3(256), 0, c->stream, tmp.as<float>(),
                       w->dw_bf16.as<bf16>(), (int64_t)mats.size());
    {   // tied output projection for free-running decoding: the token embedding in bf16, rows padded to the 128-column GEMM tile
        const size_t Vp = (size_t)div_up(V, 128) * 128;
        PCE_HIP(c, w->g_emb_bf16.reserve(sizeof(bf16) * Vp * (size_t)d + 256));
        PCE_HIP(c, hipMemsetAsync(w->g_emb_bf16.p, 0, sizeof(bf16) * Vp * (size_t)d, c->stream));
        hipLaunchKernelGGL(k_f32_to_bf16, dim3((unsigned)div_up((int64_t)V * d, 256)), dim3(256), 0, c->stream, w->d_tok_emb.as<float>(),
                           w->g_emb_bf16.as<bf16>(), (int64_t)V * d);
    }
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    tmp.release();
    w->dec_loaded = true; w->g_xkv_clips = -1; w->g_cache_len = -1;
    return PCE_OK;
}

int pce_whisper_align_run(pce_ctx *c, const int32_t *tokens, const int32_t *token_offsets, const int32_t *num_frames, int32_t sot_len,
                          const uint8_t *head_mask, int32_t medfilt_width, float qk_scale)
{
    if (!c || !tokens || !token_offsets || !num_frames) return PCE_E_INVALID;
    WhisperState *w = ws_of(c);
    if (!w->dec_loaded) return pce_fail(c, PCE_E_STATE, "pce_whisper_align_run before pce_whisper_decoder_load");
    if (w->n_clips_enc < 0) return pce_fail(c, PCE_E_STATE, "run pce_whisper_encode_run first");
    if (w->tdims.n_state != w->dims.n_state) return pce_fail(c, PCE_E_INVALID, "decoder and encoder widths differ");
    if (medfilt_width < 1 || medfilt_width > 15 || !(medfilt_width & 1)) return pce_fail(c, PCE_E_INVALID, "median filter width must be odd, <= 15");
    PCE_HIP(c, hipSetDevice(c->device));
    const int n = w->n_clips_enc, d = w->tdims.n_state, H = w->tdims.n_head, L = w->tdims.n_layer, V = w->tdims.n_vocab;
    // ---- shapes
    int T_max = 0, F_max = 0, N_max = 0;
    std::vector<int> t_len((size_t)n), f_len((size_t)n);
    for (int i = 0; i < n; i++) {
        const int T = token_offsets[i + 1] - token_offsets[i];
        if (T < sot_len + 2 || T > w->tdims.n_text_ctx) return pce_fail(c, PCE_E_INVALID, "clip %d: %d tokens (need %d..%d)", i, T, sot_len + 2, w->tdims.n_text_ctx);
        int F = num_frames[i] / 2; if (F > W_CTX) F = W_CTX; if (F < 1) return pce_fail(c, PCE_E_INVALID, "clip %d: no audio frames", i);
        t_len[(size_t)i] = T; f_len[(size_t)i] = F;
        T_max = std::max(T_max, T); F_max = std::max(F_max, F); N_max = std::max(N_max, T - sot_len - 1);
    }
    const int T_pad = (int)div_up(T_max, 64) * 64, F_pad = (int)div_up(F_max, 64) * 64, SPD = 512;
    const int64_t Mt = (int64_t)n * T_pad, Ma = (int64_t)n * W_CTX;
    // selected heads (default: every head of the last half of the layers, as whisper's Whisper.__init__ sets alignment_heads)
    std::vector<int> heads; std::vector<int> layer_first((size_t)L + 1, 0);
    for (int l = 0; l < L; l++) {
        layer_first[(size_t)l] = (int)heads.size();
        for (int hh = 0; hh < H; hh++)
            if (head_mask ? head_mask[l * H + hh] != 0 : l >= L / 2) heads.push_back(hh);
    }
    layer_first[(size_t)L] = (int)heads.size();
    const int n_sel = (int)heads.size();
    if (n_sel == 0) return pce_fail(c, PCE_E_INVALID, "no alignment head selected");
    // ---- tables: [q_row0 | q_len | a_row0 | a_len | f_len | n_rows(dtw)] and padded tokens
    std::vector<int> tab((size_t)6 * n), tok((size_t)Mt, 0);
    for (int i = 0; i < n; i++) {
        tab[(size_t)i] = i * T_pad; tab[(size_t)n + i] = t_len[(size_t)i]; tab[(size_t)2 * n + i] = i * W_CTX; tab[(size_t)3 * n + i] = W_CTX;
        tab[(size_t)4 * n + i] = f_len[(size_t)i]; tab[(size_t)5 * n + i] = t_len[(size_t)i] - sot_len - 1;
        for (int t = 0; t < t_len[(size_t)i]; t++) {
            const int v = tokens[token_offsets[i] + t];
            if (v < 0 || v >= V) return pce_fail(c, PCE_E_INVALID, "token %d out of the vocabulary", v);
            tok[(size_t)i * T_pad + t] = v;
        }
    }
    w->al_rows.assign(tab.begin() + 5 * n, tab.begin() + 6 * n); w->al_cols = f_len;
    PCE_HIP(c, w->d_tab.reserve(sizeof(int) * tab.size()));
    PCE_HIP(c, w->d_tokens.reserve(sizeof(int) * tok.size()));
    PCE_HIP(c, w->d_heads.reserve(sizeof(int) * heads.size()));
    PCE_HIP(c, hipMemcpyAsync(w->d_tab.p, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(w->d_tokens.p, tok.data(), sizeof(int) * tok.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(w->d_heads.p, heads.data(), sizeof(int) * heads.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    const int *T0 = w->d_tab.as<int>(), *TL = T0 + n, *A0 = T0 + 2 * n, *AL = T0 + 3 * n, *FL = T0 + 4 * n, *NR = T0 + 5 * n;
    // ---- buffers
    PCE_HIP(c, w->d_resid.reserve(sizeof(float) * (size_t)Mt * d));
    PCE_HIP(c, w->d_ln.reserve(sizeof(bf16) * (size_t)Mt * d + 4096));
    PCE_HIP(c, w->d_qk.reserve(sizeof(bf16) * (size_t)Mt * 2 * d + 4096));
    PCE_HIP(c, w->d_attn.reserve(sizeof(bf16) * (size_t)Mt * d + 4096));
    // k_attention writes the rows of real tokens only: the pad rows (T..T_pad of every clip) must not hold stale bits, a NaN
    // there would reach the next layer's V^T and poison valid queries through 0 * NaN (seen as an intermittent NaN cost matrix)
    PCE_HIP(c, hipMemsetAsync(w->d_attn.p, 0, sizeof(bf16) * (size_t)Mt * d + 4096, c->stream));
    PCE_HIP(c, w->d_q.reserve(sizeof(bf16) * (size_t)Mt * d + 4096));
    PCE_HIP(c, w->d_hidden.reserve(sizeof(bf16) * (size_t)Mt * 4 * d + 4096));
    PCE_HIP(c, w->d_enc_bf16.reserve(sizeof(bf16) * (size_t)Ma * d + 4096));
    const size_t dvt_elems = (size_t)n * (size_t)d * SPD + 64;
    PCE_HIP(c, w->d_vt.reserve(sizeof(bf16) * dvt_elems));
    PCE_HIP(c, hipMemsetAsync(w->d_vt.p, 0, sizeof(bf16) * dvt_elems, c->stream));
    PCE_HIP(c, w->d_aw.reserve(sizeof(float) * (size_t)n * n_sel * T_pad * (size_t)F_pad));
    PCE_HIP(c, w->d_cost.reserve(sizeof(double) * (size_t)n * N_max * (size_t)F_pad));
    PCE_HIP(c, w->d_trace.reserve((size_t)n * (size_t)(N_max + 1) * (size_t)(F_max + 1)));
    PCE_HIP(c, w->d_pi.reserve(sizeof(int) * (size_t)n * (size_t)(N_max + F_max)));
    PCE_HIP(c, w->d_pj.reserve(sizeof(int) * (size_t)n * (size_t)(N_max + F_max)));
    PCE_HIP(c, w->d_pl.reserve(sizeof(int) * (size_t)n));
    // the audio keys/values reuse the encoder's q|k and V^T buffers (the encoder is finished)
    bf16 *xk = w->qkv.as<bf16>(), *xvt = w->vt.as<bf16>();
    // when a decoding step has already projected this encoded batch (pce_whisper_decode_step keeps the cross K / V of all
    // layers), the 12 projections (15 of this call's 27 ms at 256 clips) are read from there
    const bool xkv_cached = w->g_xkv_clips == n;
    const size_t xk_cl = (size_t)Ma * d, xvt_cl = (size_t)n * (size_t)d * AT_SP;
    const bf16 *Wb = w->dw_bf16.as<bf16>();
    const float *Wf = w->dw_f32.as<float>();
    KernelTimer timer(c, PCE_K_WHISPER_ALIGN);
    if (w->enc_bf16_clips != n)                                   // (the persistent-GEMM encoder path has already written it)
        hipLaunchKernelGGL(k_f32_to_bf16, dim3((unsigned)div_up(Ma * d, 256)), dim3(256), 0, c->stream, w->final_out.as<float>(),
                               w->d_enc_bf16.as<bf16>(), Ma * d);
    hipLaunchKernelGGL(k_embed_tokens, dim3((unsigned)div_up(Mt * d, 256)), dim3(256), 0, c->stream, w->d_tokens.as<int>(),
                       w->d_tok_emb.as<float>(), w->d_pos_emb.as<float>(), T_pad, w->tdims.n_text_ctx, d, Mt, w->d_resid.as<float>());
    auto attn = [&](const bf16 *q, int64_t q_ld, const bf16 *k, int64_t k_ld, const bf16 *vt, int64_t vt_clip, int vt_sp,
                    const int *k0, const int *kl, int causal) {
        AttnArgs a{};
        a.q = q; a.q_ld = q_ld; a.k = k; a.k_ld = k_ld; a.vt = vt; a.vt_clip = vt_clip; a.vt_sp = vt_sp;
        a.q_row0 = T0; a.q_len = TL; a.k_row0 = k0; a.k_len = kl; a.out = w->d_attn.as<bf16>(); a.out_ld = d; a.causal = causal;
        launch_attention(c, dim3((unsigned)div_up(T_pad, AT_QB), (unsigned)H, (unsigned)n), a);
    };
    for (int l = 0; l < L; l++) {
        const WhisperState::DLayer &ly = w->dlayers[(size_t)l];
        // masked self attention
        hipLaunchKernelGGL((k_layernorm<bf16>), dim3((unsigned)div_up(Mt, 4)), dim3(256), 0, c->stream, w->d_resid.as<float>(), Wf + ly.ln1_w,
                           Wf + ly.ln1_b, Mt, d, w->d_ln.as<bf16>());
        launch_gemm<EPI_QKV>(c, w->d_ln.as<bf16>(), d, 0, Wb + ly.qkv_w, (int)Mt, 3 * d, d, Wf + ly.qkv_b, w->d_qk.as<bf16>(), 2 * d, 0, 1,
                             reinterpret_cast<const float *>(w->d_vt.as<bf16>()), T_pad, 2 * d, SPD);
        attn(w->d_qk.as<bf16>(), 2 * d, w->d_qk.as<bf16>() + d, 2 * d, w->d_vt.as<bf16>(), (int64_t)d * SPD, SPD, T0, TL, 1);
        launch_gemm<EPI_RESID_F32>(c, w->d_attn.as<bf16>(), d, 0, Wb + ly.out_w, (int)Mt, d, d, Wf + ly.out_b, w->d_resid.as<float>(), d, 0, 1);
        // cross attention over the audio features
        hipLaunchKernelGGL((k_layernorm<bf16>), dim3((unsigned)div_up(Mt, 4)), dim3(256), 0, c->stream, w->d_resid.as<float>(), Wf + ly.lnx_w,
                           Wf + ly.lnx_b, Mt, d, w->d_ln.as<bf16>());
        launch_gemm<EPI_BF16>(c, w->d_ln.as<bf16>(), d, 0, Wb + ly.xq_w, (int)Mt, d, d, Wf + ly.xq_b, w->d_q.as<bf16>(), d, 0, 1);
        if (xkv_cached) { xk = w->g_xk.as<bf16>() + xk_cl * (size_t)l; xvt = w->g_xvt.as<bf16>() + xvt_cl * (size_t)l; }
        else
            project_cross_kv(c, w->d_enc_bf16.as<bf16>(), (int)Ma, d, Wb + ly.xkv_w, Wf + ly.xkv_b, xk, xvt);
        attn(w->d_q.as<bf16>(), d, xk, d, xvt, (int64_t)d * AT_SP, AT_SP, A0, AL, 0);
        const int ns_l = layer_first[(size_t)l + 1] - layer_first[(size_t)l];
        if (ns_l > 0) {
            AlignArgs g{};
            g.q = w->d_q.as<bf16>(); g.q_ld = d; g.k = xk; g.k_ld = d; g.t_len = TL; g.f_len = FL;
            g.heads = w->d_heads.as<int>() + layer_first[(size_t)l];
            g.w = w->d_aw.as<float>(); g.sel0 = layer_first[(size_t)l]; g.n_sel_total = n_sel; g.T_pad = T_pad; g.F_pad = F_pad;
            g.scale = 0.125f * qk_scale;
            hipLaunchKernelGGL(k_align_scores, dim3((unsigned)div_up(T_pad, 16), (unsigned)ns_l, (unsigned)n), dim3(256), 0, c->stream, g);
        }
        launch_gemm<EPI_RESID_F32>(c, w->d_attn.as<bf16>(), d, 0, Wb + ly.xout_w, (int)Mt, d, d, Wf + ly.xout_b, w->d_resid.as<float>(), d, 0, 1);
        // MLP
        hipLaunchKernelGGL((k_layernorm<bf16>), dim3((unsigned)div_up(Mt, 4)), dim3(256), 0, c->stream, w->d_resid.as<float>(), Wf + ly.ln2_w,
                           Wf + ly.ln2_b, Mt, d, w->d_ln.as<bf16>());
        launch_gemm<EPI_GELU_BF16>(c, w->d_ln.as<bf16>(), d, 0, Wb + ly.m1_w, (int)Mt, 4 * d, d, Wf + ly.m1_b, w->d_hidden.as<bf16>(), 4 * d, 0, 1);
        launch_gemm<EPI_RESID_F32>(c, w->d_hidden.as<bf16>(), 4 * d, 0, Wb + ly.m2_w, (int)Mt, d, 4 * d, Wf + ly.m2_b, w->d_resid.as<float>(), d, 0, 1);
    }
    // alignment matrix: normalise over tokens, median filter over time, mean over heads, DTW
    hipLaunchKernelGGL(k_align_colnorm, dim3((unsigned)div_up(F_pad, 256), (unsigned)n_sel, (unsigned)n), dim3(256), 0, c->stream,
                       w->d_aw.as<float>(), TL, FL, n_sel, T_pad, F_pad);
    if (medfilt_width == 7 && !c->generic_median)
        hipLaunchKernelGGL((k_align_cost<7>), dim3((unsigned)div_up(F_pad, 256), (unsigned)N_max, (unsigned)n), dim3(256), 0, c->stream,
                           w->d_aw.as<float>(), TL, FL, n_sel, T_pad, F_pad, (int)sot_len, (int)medfilt_width, N_max, w->d_cost.as<double>());
    else
        hipLaunchKernelGGL((k_align_cost<0>), dim3((unsigned)div_up(F_pad, 256), (unsigned)N_max, (unsigned)n), dim3(256), 0, c->stream,
                           w->d_aw.as<float>(), TL, FL, n_sel, T_pad, F_pad, (int)sot_len, (int)medfilt_width, N_max, w->d_cost.as<double>());
    int rc = pce_dtw_launch(c, w->d_cost.as<double>(), (int64_t)N_max * F_pad, F_pad, NR, FL, N_max, F_max, n, w->d_trace.as<unsigned char>(),
                            w->d_pi.as<int>(), w->d_pj.as<int>(), w->d_pl.as<int>());
    if (rc) return rc;
    PCE_HIP(c, hipGetLastError());
    w->al_n = n; w->al_Nmax = N_max; w->al_Fpad = F_pad; w->al_Mmax = F_max;
    return PCE_OK;
}

int pce_whisper_align_fetch(pce_ctx *c, int32_t clip, int32_t *text_idx, int32_t *time_idx, int32_t *path_len, double *cost)
{
    if (!c || !path_len) return PCE_E_INVALID;
    WhisperState *w = ws_of(c);
    if (w->al_n < 0) return pce_fail(c, PCE_E_STATE, "pce_whisper_align_fetch before pce_whisper_align_run");
    if (clip < 0 || clip >= w->al_n) return pce_fail(c, PCE_E_INVALID, "clip out of range");
    PCE_HIP(c, hipSetDevice(c->device));
    const size_t stride = (size_t)(w->al_Nmax + w->al_Mmax);
    int n = 0;
    PCE_HIP(c, hipMemcpyAsync(&n, w->d_pl.as<int>() + clip, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    *path_len = n;
    if (text_idx) PCE_HIP(c, hipMemcpyAsync(text_idx, w->d_pi.as<int>() + stride * (size_t)clip, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (time_idx) PCE_HIP(c, hipMemcpyAsync(time_idx, w->d_pj.as<int>() + stride * (size_t)clip, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (cost) {
        const int N = w->al_rows[(size_t)clip], F = w->al_cols[(size_t)clip];
        PCE_HIP(c, hipMemcpy2DAsync(cost, sizeof(double) * (size_t)F, w->d_cost.as<double>() + (size_t)clip * w->al_Nmax * (size_t)w->al_Fpad,
                                    sizeof(double) * (size_t)w->al_Fpad, sizeof(double) * (size_t)F, (size_t)N, hipMemcpyDeviceToHost, c->stream));
    }
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    pce_profile_collect(c);
    return PCE_OK;
}

int pce_whisper_align_shape(pce_ctx *c, int32_t clip, int32_t *n_rows, int32_t *n_cols)
{
    if (!c) return PCE_E_INVALID;
    WhisperState *w = ws_of(c);
    if (w->al_n < 0) return pce_fail(c, PCE_E_STATE, "pce_whisper_align_shape before pce_whisper_align_run");
    if (clip < 0 || clip >= w->al_n) return pce_fail(c, PCE_E_INVALID, "clip out of range");
    if (n_rows) *n_rows = w->al_rows[(size_t)clip];
    if (n_cols) *n_cols = w->al_cols[(size_t)clip];
    return PCE_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------
// Free-running decoding, one step: the text decoder over the sequences so far (cross K / V of every layer computed once
// per encoded batch and kept), logits of the last position through the tied output projection, openai-whisper's logit
// filters and greedy choice.  The loop over steps, the prompt and the stopping rule are host logic
// (Aligners/decoding.py); every step re-runs the decoder over the whole prefix (a per-step K / V cache for the
// self-attention is the next refinement).
// ---------------------------------------------------------------------------
extern "C" int pce_whisper_decode_step(pce_ctx *c, const int32_t *tokens, const int32_t *token_offsets, int32_t sample_begin,
                                       const pce_whisper_decode_rules *rules, const uint8_t *vocab_mask, int32_t *next_tokens,
                                       float *next_logprobs)
{
    pce_whisper_decode_opts o{};
    o.sample_begin = nullptr; o.sample_begin_all = sample_begin; o.temperature = 0.f; o.probe_token = -1;
    return pce_whisper_decode_step_ex(c, tokens, token_offsets, rules, vocab_mask, &o, next_tokens, next_logprobs, nullptr);
}

// launches of ONE incremental step (one new position per sequence) from the tables ct[8 n] (see k_step_advance) -- host-uploaded by
// pce_whisper_decode_step_ex, device-maintained inside pce_whisper_decode_loop -- down to the final LayerNorm of the new position
static void decode_incremental_launches(pce_ctx *c, WhisperState *w, int n, const int *CT, const int *ended)
{
    const int d = w->tdims.n_state, H = w->tdims.n_head, L = w->tdims.n_layer, SPD = 512, T_cap = w->tdims.n_text_ctx;
    const int64_t Ma = (int64_t)n * W_CTX;
    const size_t xk_l = (size_t)Ma * d, xvt_l = (size_t)n * (size_t)d * AT_SP, sk_l = (size_t)n * T_cap * d, svt_l = (size_t)n * (size_t)d * SPD;
    const bf16 *Wb = w->dw_bf16.as<bf16>();
    const float *Wf = w->dw_f32.as<float>();
    const int *Q0 = CT + n, *QL = CT + 2 * n, *K0 = CT + 3 * n, *KL = CT + 4 * n, *X0 = CT + 5 * n, *XL = CT + 6 * n, *POS = CT + 7 * n;
    hipLaunchKernelGGL(k_embed_one, dim3((unsigned)div_up((int64_t)n * d, 256)), dim3(256), 0, c->stream, CT, w->d_tok_emb.as<float>(),
                       w->d_pos_emb.as<float>(), POS, d, n, w->g_c_resid.as<float>());
    auto cattn = [&](const bf16 *q, int64_t q_ld, const bf16 *k, int64_t k_ld, const bf16 *vt, int64_t vt_clip, int vt_sp, const int *k0, const int *kl) {
        if (c->attn1) {                                           // streaming single-query kernel (PCE_ATTN1=0: the MFMA attention kernel with one live query)
            Attn1Args a{};
            a.q = q; a.q_ld = q_ld; a.k = k; a.k_ld = k_ld; a.vt = vt; a.vt_clip = vt_clip; a.vt_sp = vt_sp; a.k_row0 = k0; a.k_len = kl; a.skip = ended;
            a.out = w->g_c_attn.as<bf16>(); a.out_ld = d;
            KernelTimer kt(c, PCE_K_CROSS_ATTN1);
            hipLaunchKernelGGL(k_cross_attn1, dim3((unsigned)H, (unsigned)n), dim3(A1_T), 0, c->stream, a);
            return;
        }
        AttnArgs a{};
        a.q = q; a.q_ld = q_ld; a.k = k; a.k_ld = k_ld; a.vt = vt; a.vt_clip = vt_clip; a.vt_sp = vt_sp;
        a.q_row0 = Q0; a.q_len = QL; a.k_row0 = k0; a.k_len = kl; a.out = w->g_c_attn.as<bf16>(); a.out_ld = d; a.causal = 0;
        launch_attention(c, dim3(1u, (unsigned)H, (unsigned)n), a);
    };
    auto cln = [&](size_t w_off, size_t b_off) {
        hipLaunchKernelGGL((k_layernorm<bf16>), dim3((unsigned)div_up(n, 4)), dim3(256), 0, c->stream, w->g_c_resid.as<float>(), Wf + w_off, Wf + b_off,
                           (int64_t)n, d, w->g_c_ln.as<bf16>());
    };
    for (int l = 0; l < L; l++) {
        const WhisperState::DLayer &ly = w->dlayers[(size_t)l];
        cln(ly.ln1_w, ly.ln1_b);
        launch_gemm<EPI_BF16>(c, w->g_c_ln.as<bf16>(), d, 0, Wb + ly.qkv_w, n, 3 * d, d, Wf + ly.qkv_b, w->g_c_qkv.as<bf16>(), 3 * d, 0, 1);
        hipLaunchKernelGGL(k_append_kv, dim3((unsigned)div_up((int64_t)n * d, 256)), dim3(256), 0, c->stream, w->g_c_qkv.as<bf16>(), POS, d, T_cap, SPD, n,
                           w->g_sk.as<bf16>() + sk_l * (size_t)l, w->g_svt.as<bf16>() + svt_l * (size_t)l);
        cattn(w->g_c_qkv.as<bf16>(), 3 * d, w->g_sk.as<bf16>() + sk_l * (size_t)l, d, w->g_svt.as<bf16>() + svt_l * (size_t)l, (int64_t)d * SPD, SPD, K0, KL);
        launch_gemm<EPI_RESID_F32>(c, w->g_c_attn.as<bf16>(), d, 0, Wb + ly.out_w, n, d, d, Wf + ly.out_b, w->g_c_resid.as<float>(), d, 0, 1);
        cln(ly.lnx_w, ly.lnx_b);
        launch_gemm<EPI_BF16>(c, w->g_c_ln.as<bf16>(), d, 0, Wb + ly.xq_w, n, d, d, Wf + ly.xq_b, w->g_c_q.as<bf16>(), d, 0, 1);
        cattn(w->g_c_q.as<bf16>(), d, w->g_xk.as<bf16>() + xk_l * (size_t)l, d, w->g_xvt.as<bf16>() + xvt_l * (size_t)l, (int64_t)d * AT_SP, AT_SP, X0, XL);
        launch_gemm<EPI_RESID_F32>(c, w->g_c_attn.as<bf16>(), d, 0, Wb + ly.xout_w, n, d, d, Wf + ly.xout_b, w->g_c_resid.as<float>(), d, 0, 1);
        cln(ly.ln2_w, ly.ln2_b);
        launch_gemm<EPI_GELU_BF16>(c, w->g_c_ln.as<bf16>(), d, 0, Wb + ly.m1_w, n, 4 * d, d, Wf + ly.m1_b, w->g_c_hidden.as<bf16>(), 4 * d, 0, 1);
        launch_gemm<EPI_RESID_F32>(c, w->g_c_hidden.as<bf16>(), 4 * d, 0, Wb + ly.m2_w, n, d, 4 * d, Wf + ly.m2_b, w->g_c_resid.as<float>(), d, 0, 1);
    }
    hipLaunchKernelGGL((k_layernorm<bf16>), dim3((unsigned)div_up(n, 4)), dim3(256), 0, c->stream, w->g_c_resid.as<float>(), Wf + w->dln_w, Wf + w->dln_b,
                       (int64_t)n, d, w->g_lastln.as<bf16>());
}
static int decode_incremental_reserve(pce_ctx *c, WhisperState *w, int n)
{
    const int d = w->tdims.n_state;
    PCE_HIP(c, w->g_c_tab.reserve(sizeof(int) * 8 * (size_t)n));
    PCE_HIP(c, w->g_c_resid.reserve(sizeof(float) * (size_t)n * d));
    PCE_HIP(c, w->g_c_ln.reserve(sizeof(bf16) * (size_t)(n + 128) * d + 4096));
    PCE_HIP(c, w->g_c_qkv.reserve(sizeof(bf16) * (size_t)(n + 128) * 3 * d + 4096));
    {
        const size_t before = w->g_c_attn.cap;
        PCE_HIP(c, w->g_c_attn.reserve(sizeof(bf16) * (size_t)(n + 128) * d + 4096));
        if (w->g_c_attn.cap != before) PCE_HIP(c, hipMemsetAsync(w->g_c_attn.p, 0, w->g_c_attn.cap, c->stream));   // rows of ended sequences are skipped: never uninitialised bits
    }
    PCE_HIP(c, w->g_c_q.reserve(sizeof(bf16) * (size_t)(n + 128) * d + 4096));
    PCE_HIP(c, w->g_c_hidden.reserve(sizeof(bf16) * (size_t)(n + 128) * 4 * d + 4096));
    return PCE_OK;
}

// one decoding step; results stay on the device in g_next (next token | log-probability | probe probability, n each).  `resident`: the
// caller (pce_whisper_decode_loop) keeps the tables of an incremental step on the device: `tokens` is then only consulted for step 0.
static int decode_step_device(pce_ctx *c, const int32_t *tokens, const int32_t *token_offsets, const pce_whisper_decode_rules *rules,
                              const uint8_t *vocab_mask, const pce_whisper_decode_opts *opts, bool want_probe);

extern "C" int pce_whisper_decode_step_ex(pce_ctx *c, const int32_t *tokens, const int32_t *token_offsets, const pce_whisper_decode_rules *rules,
                                          const uint8_t *vocab_mask, const pce_whisper_decode_opts *opts, int32_t *next_tokens,
                                          float *next_logprobs, float *probe_prob)
{
    if (!c || !tokens || !token_offsets || !rules || !vocab_mask || !next_tokens || !opts) return PCE_E_INVALID;
    const int rc = decode_step_device(c, tokens, token_offsets, rules, vocab_mask, opts, probe_prob != nullptr);
    if (rc) return rc;
    WhisperState *w = ws_of(c);
    const int n = w->n_clips_enc;
    if (n == 0) return PCE_OK;
    PCE_HIP(c, hipMemcpyAsync(next_tokens, w->g_next.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (next_logprobs) PCE_HIP(c, hipMemcpyAsync(next_logprobs, w->g_next.as<int>() + n, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (probe_prob && opts->probe_token >= 0)
        PCE_HIP(c, hipMemcpyAsync(probe_prob, w->g_next.as<int>() + 2 * n, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    return PCE_OK;
}

static int decode_step_device(pce_ctx *c, const int32_t *tokens, const int32_t *token_offsets, const pce_whisper_decode_rules *rules,
                              const uint8_t *vocab_mask, const pce_whisper_decode_opts *opts, bool want_probe)
{
    float probe_dummy = 0.f;
    float *probe_prob = want_probe ? &probe_dummy : nullptr;      // (only its non-nullness is consulted below)
    const int32_t sample_begin = opts->sample_begin_all;
    WhisperState *w = ws_of(c);
    if (!w->dec_loaded) return pce_fail(c, PCE_E_STATE, "pce_whisper_decode_step before pce_whisper_decoder_load");
    if (w->n_clips_enc < 0) return pce_fail(c, PCE_E_STATE, "run pce_whisper_encode_run first");
    if (w->tdims.n_state != w->dims.n_state) return pce_fail(c, PCE_E_INVALID, "decoder and encoder widths differ");
    PCE_HIP(c, hipSetDevice(c->device));
    const int n = w->n_clips_enc, d = w->tdims.n_state, H = w->tdims.n_head, L = w->tdims.n_layer, V = w->tdims.n_vocab, SPD = 512;
    if (rules->eot < 0 || rules->eot >= V || rules->timestamp_begin <= rules->eot || rules->timestamp_begin > V || (!opts->sample_begin && sample_begin < 1))
        return pce_fail(c, PCE_E_INVALID, "decoding rules: need 0 <= eot < timestamp_begin <= n_vocab, sample_begin >= 1");
    if (!(opts->temperature >= 0.f) || opts->probe_token >= V) return pce_fail(c, PCE_E_INVALID, "decoding options: temperature >= 0, probe_token < n_vocab");
    if (n == 0) return PCE_OK;
    int T_max = 0;
    std::vector<int> t_len((size_t)n);
    for (int i = 0; i < n; i++) {
        const int T = token_offsets[i + 1] - token_offsets[i];
        const int sb = opts->sample_begin ? opts->sample_begin[i] : sample_begin;
        if (sb < 1 || T < sb || T > w->tdims.n_text_ctx) return pce_fail(c, PCE_E_INVALID, "clip %d: %d tokens (need %d..%d)", i, T, sb, w->tdims.n_text_ctx);
        t_len[(size_t)i] = T; T_max = std::max(T_max, T);
    }
    const int T_pad = (int)div_up(T_max, 64) * 64;
    const int64_t Mt = (int64_t)n * T_pad, Ma = (int64_t)n * W_CTX;
    const int64_t Vp = div_up(V, 128) * 128;
    std::vector<int> tab((size_t)5 * n), tok((size_t)Mt, 0);
    for (int i = 0; i < n; i++) {
        tab[(size_t)i] = i * T_pad; tab[(size_t)n + i] = t_len[(size_t)i]; tab[(size_t)2 * n + i] = i * W_CTX; tab[(size_t)3 * n + i] = W_CTX;
        tab[(size_t)4 * n + i] = opts->sample_begin ? opts->sample_begin[i] : sample_begin;
        for (int t = 0; t < t_len[(size_t)i]; t++) {
            const int v = tokens[token_offsets[i] + t];
            if (v < 0 || v >= V) return pce_fail(c, PCE_E_INVALID, "token %d out of the vocabulary", v);
            tok[(size_t)i * T_pad + t] = v;
        }
    }
    PCE_HIP(c, w->d_tab.reserve(sizeof(int) * tab.size()));
    PCE_HIP(c, w->d_tokens.reserve(sizeof(int) * tok.size()));
    PCE_HIP(c, w->d_resid.reserve(sizeof(float) * (size_t)Mt * d));
    PCE_HIP(c, w->d_ln.reserve(sizeof(bf16) * (size_t)Mt * d + 4096));
    PCE_HIP(c, w->d_qk.reserve(sizeof(bf16) * (size_t)Mt * 2 * d + 4096));
    PCE_HIP(c, w->d_attn.reserve(sizeof(bf16) * (size_t)Mt * d + 4096));
    PCE_HIP(c, w->d_q.reserve(sizeof(bf16) * (size_t)Mt * d + 4096));
    PCE_HIP(c, w->d_hidden.reserve(sizeof(bf16) * (size_t)Mt * 4 * d + 4096));
    PCE_HIP(c, w->g_last.reserve(sizeof(float) * (size_t)n * d));
    PCE_HIP(c, w->g_lastln.reserve(sizeof(bf16) * (size_t)(n + 128) * d + 4096));
    PCE_HIP(c, w->g_logits.reserve(sizeof(float) * (size_t)n * (size_t)Vp));
    PCE_HIP(c, w->g_mask.reserve((size_t)V + 64));
    PCE_HIP(c, w->g_next.reserve((sizeof(int) + 2 * sizeof(float)) * (size_t)n));
    PCE_HIP(c, hipMemcpyAsync(w->d_tab.p, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(w->d_tokens.p, tok.data(), sizeof(int) * tok.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(w->g_mask.p, vocab_mask, (size_t)V, hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    const int *T0 = w->d_tab.as<int>(), *TL = T0 + n, *A0 = T0 + 2 * n, *AL = T0 + 3 * n;
    const bf16 *Wb = w->dw_bf16.as<bf16>();
    const float *Wf = w->dw_f32.as<float>();
    KernelTimer timer(c, PCE_K_WHISPER_DECODE);
    // ---- cross K / V of every layer, once per encoded batch
    const size_t xk_l = (size_t)Ma * d, xvt_l = (size_t)n * (size_t)d * AT_SP;
    if (w->g_xkv_clips != n) {
        PCE_HIP(c, w->g_xk.reserve(sizeof(bf16) * xk_l * (size_t)L + 4096));
        PCE_HIP(c, w->g_xvt.reserve(sizeof(bf16) * xvt_l * (size_t)L + 4096));
        PCE_HIP(c, w->d_enc_bf16.reserve(sizeof(bf16) * (size_t)Ma * d + 4096));
        PCE_HIP(c, hipMemsetAsync(w->g_xvt.p, 0, sizeof(bf16) * xvt_l * (size_t)L, c->stream));      // V^T columns 1500..AT_SP are read as zeros
        if (w->enc_bf16_clips != n)
            hipLaunchKernelGGL(k_f32_to_bf16, dim3((unsigned)div_up(Ma * d, 256)), dim3(256), 0, c->stream, w->final_out.as<float>(),
                               w->d_enc_bf16.as<bf16>(), Ma * d);
        for (int l = 0; l < L; l++) {
            const WhisperState::DLayer &ly = w->dlayers[(size_t)l];
            project_cross_kv(c, w->d_enc_bf16.as<bf16>(), (int)Ma, d, Wb + ly.xkv_w, Wf + ly.xkv_b, w->g_xk.as<bf16>() + xk_l * (size_t)l,
                             w->g_xvt.as<bf16>() + xvt_l * (size_t)l);
        }
        w->g_xkv_clips = n;
    }
    // ---- self-attention K / V cache [layer][clip][T_cap] (rows) / [layer][clip][d][512] (transposed)
    const int T_cap = w->tdims.n_text_ctx;
    const size_t sk_l = (size_t)n * T_cap * d, svt_l = (size_t)n * (size_t)d * SPD;
    if (w->g_cache_n != n || !w->g_sk.p) {
        PCE_HIP(c, w->g_sk.reserve(sizeof(bf16) * sk_l * (size_t)L + 4096));
        PCE_HIP(c, w->g_svt.reserve(sizeof(bf16) * svt_l * (size_t)L + 4096));
        PCE_HIP(c, hipMemsetAsync(w->g_sk.p, 0, sizeof(bf16) * sk_l * (size_t)L, c->stream));
        PCE_HIP(c, hipMemsetAsync(w->g_svt.p, 0, sizeof(bf16) * svt_l * (size_t)L, c->stream));
        w->g_cache_n = n; w->g_cache_len = -1; w->g_cache_tok.assign((size_t)n * T_cap, 0); w->g_cache_lens.assign((size_t)n, -1);
    }
    // incremental: every sequence extends what the cache holds for it by exactly one token (lengths may differ between sequences)
    bool incremental = w->g_cache_len >= 0 && !(opts->flags & 1);
    for (int i = 0; incremental && i < n; i++) {
        const int Li = t_len[(size_t)i];
        incremental = Li >= 2 && w->g_cache_lens[(size_t)i] == Li - 1 &&
                      memcmp(&w->g_cache_tok[(size_t)i * T_cap], &tokens[token_offsets[i]], sizeof(int) * (size_t)(Li - 1)) == 0;
    }
    const bf16 *last_ln = nullptr;                                // [n][d] bf16: the final LayerNorm of the last position
    if (incremental) {
        // ---- one new position per sequence: every GEMM has M = clips rows, the attention one query per (clip, head)
        std::vector<int> ct((size_t)8 * n);                     // new token | q_row0 | q_len | k_row0 | k_len | a_row0 | a_len | position
        for (int i = 0; i < n; i++) {
            const int pos_i = t_len[(size_t)i] - 1;
            ct[(size_t)i] = tokens[token_offsets[i] + pos_i]; ct[(size_t)n + i] = i; ct[(size_t)2 * n + i] = 1; ct[(size_t)3 * n + i] = i * T_cap;
            ct[(size_t)4 * n + i] = pos_i + 1; ct[(size_t)5 * n + i] = i * W_CTX; ct[(size_t)6 * n + i] = W_CTX; ct[(size_t)7 * n + i] = pos_i;
        }
        { const int rc = decode_incremental_reserve(c, w, n); if (rc) return rc; }
        PCE_HIP(c, hipMemcpyAsync(w->g_c_tab.p, ct.data(), sizeof(int) * ct.size(), hipMemcpyHostToDevice, c->stream));
        PCE_HIP(c, hipStreamSynchronize(c->stream));
        decode_incremental_launches(c, w, n, w->g_c_tab.as<int>(), nullptr);
        last_ln = w->g_lastln.as<bf16>();
    } else {
    PCE_HIP(c, hipMemsetAsync(w->d_attn.p, 0, sizeof(bf16) * (size_t)Mt * d + 4096, c->stream));      // pad rows: no stale bits (see pce_whisper_align_run)
    hipLaunchKernelGGL(k_embed_tokens, dim3((unsigned)div_up(Mt * d, 256)), dim3(256), 0, c->stream, w->d_tokens.as<int>(),
                       w->d_tok_emb.as<float>(), w->d_pos_emb.as<float>(), T_pad, w->tdims.n_text_ctx, d, Mt, w->d_resid.as<float>());
    auto attn = [&](const bf16 *q, int64_t q_ld, const bf16 *k, int64_t k_ld, const bf16 *vt, int64_t vt_clip, int vt_sp,
                    const int *k0, const int *kl, int causal) {
        AttnArgs a{};
        a.q = q; a.q_ld = q_ld; a.k = k; a.k_ld = k_ld; a.vt = vt; a.vt_clip = vt_clip; a.vt_sp = vt_sp;
        a.q_row0 = T0; a.q_len = TL; a.k_row0 = k0; a.k_len = kl; a.out = w->d_attn.as<bf16>(); a.out_ld = d; a.causal = causal;
        launch_attention(c, dim3((unsigned)div_up(T_pad, AT_QB), (unsigned)H, (unsigned)n), a);
    };
    auto ln = [&](size_t w_off, size_t b_off) {
        hipLaunchKernelGGL((k_layernorm<bf16>), dim3((unsigned)div_up(Mt, 4)), dim3(256), 0, c->stream, w->d_resid.as<float>(), Wf + w_off, Wf + b_off,
                           Mt, d, w->d_ln.as<bf16>());
    };
    for (int l = 0; l < L; l++) {
        const WhisperState::DLayer &ly = w->dlayers[(size_t)l];
        ln(ly.ln1_w, ly.ln1_b);
        bf16 *svt = w->g_svt.as<bf16>() + svt_l * (size_t)l;       // the prefix run fills the cache: V^T straight from the epilogue, K rows copied
        launch_gemm<EPI_QKV>(c, w->d_ln.as<bf16>(), d, 0, Wb + ly.qkv_w, (int)Mt, 3 * d, d, Wf + ly.qkv_b, w->d_qk.as<bf16>(), 2 * d, 0, 1,
                             reinterpret_cast<const float *>(svt), T_pad, 2 * d, SPD);
        hipLaunchKernelGGL(k_cache_k, dim3((unsigned)div_up(Mt * d, 256)), dim3(256), 0, c->stream, w->d_qk.as<bf16>(), T_pad, TL, d, T_cap, n,
                           w->g_sk.as<bf16>() + sk_l * (size_t)l);
        attn(w->d_qk.as<bf16>(), 2 * d, w->d_qk.as<bf16>() + d, 2 * d, svt, (int64_t)d * SPD, SPD, T0, TL, 1);
        launch_gemm<EPI_RESID_F32>(c, w->d_attn.as<bf16>(), d, 0, Wb + ly.out_w, (int)Mt, d, d, Wf + ly.out_b, w->d_resid.as<float>(), d, 0, 1);
        ln(ly.lnx_w, ly.lnx_b);
        launch_gemm<EPI_BF16>(c, w->d_ln.as<bf16>(), d, 0, Wb + ly.xq_w, (int)Mt, d, d, Wf + ly.xq_b, w->d_q.as<bf16>(), d, 0, 1);
        attn(w->d_q.as<bf16>(), d, w->g_xk.as<bf16>() + xk_l * (size_t)l, d, w->g_xvt.as<bf16>() + xvt_l * (size_t)l, (int64_t)d * AT_SP, AT_SP, A0, AL, 0);
        launch_gemm<EPI_RESID_F32>(c, w->d_attn.as<bf16>(), d, 0, Wb + ly.xout_w, (int)Mt, d, d, Wf + ly.xout_b, w->d_resid.as<float>(), d, 0, 1);
        ln(ly.ln2_w, ly.ln2_b);
        launch_gemm<EPI_GELU_BF16>(c, w->d_ln.as<bf16>(), d, 0, Wb + ly.m1_w, (int)Mt, 4 * d, d, Wf + ly.m1_b, w->d_hidden.as<bf16>(), 4 * d, 0, 1);
        launch_gemm<EPI_RESID_F32>(c, w->d_hidden.as<bf16>(), 4 * d, 0, Wb + ly.m2_w, (int)Mt, d, 4 * d, Wf + ly.m2_b, w->d_resid.as<float>(), d, 0, 1);
    }
    // ---- last position -> ln -> logits = hidden . E^T (fp32, zero-initialised accumulator)
    hipLaunchKernelGGL(k_gather_last, dim3((unsigned)div_up((int64_t)n * d, 256)), dim3(256), 0, c->stream, w->d_resid.as<float>(), TL, T_pad, d, n,
                       w->g_last.as<float>());
    hipLaunchKernelGGL((k_layernorm<bf16>), dim3((unsigned)div_up(n, 4)), dim3(256), 0, c->stream, w->g_last.as<float>(), Wf + w->dln_w, Wf + w->dln_b,
                       (int64_t)n, d, w->g_lastln.as<bf16>());
    last_ln = w->g_lastln.as<bf16>();
    }
    // the cache now holds every position of these prefixes
    for (int i = 0; i < n; i++) {
        memcpy(&w->g_cache_tok[(size_t)i * T_cap], &tokens[token_offsets[i]], sizeof(int) * (size_t)t_len[(size_t)i]);
        w->g_cache_lens[(size_t)i] = t_len[(size_t)i];
    }
    w->g_cache_len = 0;
    PCE_HIP(c, hipMemsetAsync(w->g_logits.p, 0, sizeof(float) * (size_t)n * (size_t)Vp, c->stream));
    launch_gemm<EPI_RESID_F32>(c, last_ln, d, 0, w->g_emb_bf16.as<bf16>(), n, (int)Vp, d, nullptr, w->g_logits.as<float>(), Vp, 0, 1);
    DecRules R{rules->eot, rules->timestamp_begin, V, (int)Vp, sample_begin, rules->max_initial_timestamp_index, opts->temperature, opts->seed_lo,
               opts->seed_hi, probe_prob ? opts->probe_token : -1};
    hipLaunchKernelGGL(k_decode_rules, dim3((unsigned)n), dim3(256), 0, c->stream, w->g_logits.as<float>(), w->d_tokens.as<int>(), TL, T_pad,
                       w->g_mask.as<unsigned char>(), R, T0 + 4 * n, w->g_next.as<int>(), reinterpret_cast<float *>(w->g_next.as<int>() + n),
                       reinterpret_cast<float *>(w->g_next.as<int>() + 2 * n));
    PCE_HIP(c, hipGetLastError());
    return PCE_OK;
}

// ---------------------------------------------------------------------------
// Free-running decoding with the loop on the device (round 3).  pce_whisper_decode_step_ex round-trips every step through the
// host (tokens up, next token down, a synchronisation each way); here the prompts go up once, every later step reads what it
// needs -- the token it embeds, its position, the self-attention key count, which sequences have ended -- from tables the previous
// step's k_step_advance left on the device, and the host only looks at an "ended" counter every `check_every` steps.
// Same kernels, same filters, same counter-based sampling keys (seed, clip, position): token for token what the host-driven loop
// of Aligners/decoding.py produces (tests/test_gpu_aligner.py).
// ---------------------------------------------------------------------------
extern "C" int pce_whisper_decode_loop(pce_ctx *c, const int32_t *tokens, const int32_t *token_offsets, const pce_whisper_decode_rules *rules,
                                       const uint8_t *vocab_mask, const pce_whisper_decode_opts *opts, int32_t max_new, int32_t check_every,
                                       int32_t *out_tokens, float *out_logprobs, int32_t *out_steps, float *probe_prob)
{
    if (!c || !tokens || !token_offsets || !rules || !vocab_mask || !opts || !out_tokens || !out_steps) return PCE_E_INVALID;
    if (max_new < 1) return pce_fail(c, PCE_E_INVALID, "decode loop: max_new must be >= 1");
    if (check_every < 1) check_every = 4;
    WhisperState *w = ws_of(c);
    if (!w->dec_loaded) return pce_fail(c, PCE_E_STATE, "pce_whisper_decode_loop before pce_whisper_decoder_load");
    if (w->n_clips_enc < 0) return pce_fail(c, PCE_E_STATE, "run pce_whisper_encode_run first");
    const int n = w->n_clips_enc, d = w->tdims.n_state, V = w->tdims.n_vocab, T_cap = w->tdims.n_text_ctx;
    *out_steps = 0;
    if (n == 0) return PCE_OK;
    for (int i = 0; i < n; i++) {
        const int T = token_offsets[i + 1] - token_offsets[i];
        if (T + max_new > T_cap) return pce_fail(c, PCE_E_INVALID, "clip %d: %d prompt tokens + %d new ones exceed the text context (%d)", i, T, max_new, T_cap);
    }
    PCE_HIP(c, hipSetDevice(c->device));
    KernelTimer loop_timer(c, PCE_K_DECODE_LOOP);
    // ---- step 0: the prompts (prefix run, or one more position when the cache already holds them); next token stays in g_next
    {
        const int rc = decode_step_device(c, tokens, token_offsets, rules, vocab_mask, opts, probe_prob != nullptr && opts->probe_token >= 0);
        if (rc) return rc;
    }
    { const int rc = decode_incremental_reserve(c, w, n); if (rc) return rc; }
    const int64_t Vp = div_up(V, 128) * 128;
    // ---- device-resident loop state
    const size_t n_state_ints = (size_t)n * T_cap + 3 * (size_t)n + (size_t)n * max_new + 8 + (size_t)max_new;
    PCE_HIP(c, w->g_loop.reserve(sizeof(int) * n_state_ints + sizeof(float) * (size_t)n * max_new + 256));
    int *tok_table = w->g_loop.as<int>(), *len = tok_table + (size_t)n * T_cap, *ended = len + n, *sb = ended + n, *out_tok = sb + n,
        *ctr = out_tok + (size_t)n * max_new, *n_ended = ctr + 8;
    float *out_lp = reinterpret_cast<float *>(n_ended + max_new);
    std::vector<int> h_tab((size_t)n * T_cap, 0), h_len((size_t)n), h_sb((size_t)n), ct((size_t)8 * n, 0);
    for (int i = 0; i < n; i++) {
        const int T = token_offsets[i + 1] - token_offsets[i];
        memcpy(&h_tab[(size_t)i * T_cap], &tokens[token_offsets[i]], sizeof(int) * (size_t)T);
        h_len[(size_t)i] = T; h_sb[(size_t)i] = opts->sample_begin ? opts->sample_begin[i] : opts->sample_begin_all;
        ct[(size_t)n + i] = i; ct[(size_t)2 * n + i] = 1; ct[(size_t)3 * n + i] = i * T_cap; ct[(size_t)5 * n + i] = i * W_CTX; ct[(size_t)6 * n + i] = W_CTX;
    }
    PCE_HIP(c, hipMemsetAsync(ended, 0, sizeof(int) * (n_state_ints - (size_t)n * T_cap - (size_t)n) + sizeof(float) * (size_t)n * max_new, c->stream));
    PCE_HIP(c, hipMemcpyAsync(tok_table, h_tab.data(), sizeof(int) * h_tab.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(len, h_len.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(sb, h_sb.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(w->g_c_tab.p, ct.data(), sizeof(int) * ct.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));                 // (the host vectors are done with; the ONE upload of the window)
    int *CT = w->g_c_tab.as<int>();
    const int *next = w->g_next.as<int>();
    const float *next_lp = reinterpret_cast<const float *>(w->g_next.as<int>() + n);
    auto advance = [&]() {
        hipLaunchKernelGGL(k_step_advance, dim3(1), dim3(256), 0, c->stream, n, T_cap, (int)rules->eot, (int)max_new, tok_table, len, next, next_lp, CT, ended,
                           out_tok, out_lp, ctr, n_ended);
    };
    auto all_ended = [&](int step, bool *yes) -> int {
        int cnt = 0;
        PCE_HIP(c, hipMemcpyAsync(&cnt, n_ended + step, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        PCE_HIP(c, hipStreamSynchronize(c->stream));
        *yes = cnt >= n;
        return PCE_OK;
    };
    advance();
    int steps = 1;
    bool stop = false;
    DecRules R{rules->eot, rules->timestamp_begin, V, (int)Vp, opts->sample_begin_all, rules->max_initial_timestamp_index, opts->temperature, opts->seed_lo,
               opts->seed_hi, -1};
    if (max_new > 1) { const int rc = all_ended(0, &stop); if (rc) return rc; }
    while (!stop && steps < max_new) {
        decode_incremental_launches(c, w, n, CT, ended);
        PCE_HIP(c, hipMemsetAsync(w->g_logits.p, 0, sizeof(float) * (size_t)n * (size_t)Vp, c->stream));
        launch_gemm<EPI_RESID_F32>(c, w->g_lastln.as<bf16>(), d, 0, w->g_emb_bf16.as<bf16>(), n, (int)Vp, d, nullptr, w->g_logits.as<float>(), Vp, 0, 1);
        hipLaunchKernelGGL(k_decode_rules, dim3((unsigned)n), dim3(256), 0, c->stream, w->g_logits.as<float>(), tok_table, len, T_cap,
                           w->g_mask.as<unsigned char>(), R, sb, w->g_next.as<int>(), reinterpret_cast<float *>(w->g_next.as<int>() + n),
                           static_cast<float *>(nullptr));
        advance();
        steps++;
        if (steps % check_every == 0 && steps < max_new) { const int rc = all_ended(steps - 1, &stop); if (rc) return rc; }
    }
    PCE_HIP(c, hipGetLastError());
    // ---- the ONE download of the window
    std::vector<int> h_out((size_t)n * max_new);
    std::vector<float> h_lp((size_t)n * max_new);
    PCE_HIP(c, hipMemcpyAsync(h_out.data(), out_tok, sizeof(int) * h_out.size(), hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipMemcpyAsync(h_lp.data(), out_lp, sizeof(float) * h_lp.size(), hipMemcpyDeviceToHost, c->stream));
    if (probe_prob && opts->probe_token >= 0)
        PCE_HIP(c, hipMemcpyAsync(probe_prob, w->g_next.as<int>() + 2 * n, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; i++)
        for (int sidx = 0; sidx < max_new; sidx++) {
            // steps that never ran read as "already ended": end-of-text with a zero log-probability, what further steps would have produced
            out_tokens[(size_t)i * max_new + sidx] = sidx < steps ? h_out[(size_t)i * max_new + sidx] : rules->eot;
            if (out_logprobs) out_logprobs[(size_t)i * max_new + sidx] = sidx < steps ? h_lp[(size_t)i * max_new + sidx] : 0.f;
        }
    *out_steps = steps;
    w->g_cache_len = -1;                                         // the host's copy of the cached prefixes is stale: the next host-driven step re-runs its prefixes
    return PCE_OK;
}

// ---------------------------------------------------------------------------
// Break-prediction token classifier: BertForTokenClassification forward (post-LN encoder layers on the same GEMM /
// attention / LayerNorm kernels as the Whisper encoder; Code/baseline_models/pause_bert.py:127-132).
// ---------------------------------------------------------------------------
extern "C" {

int pce_bert_load(pce_ctx *c, const pce_bert_dims *dims, const float *weights, int64_t n_floats)
{
    if (!c || !dims || !weights) return PCE_E_INVALID;
    const int d = dims->n_state, L = dims->n_layer, V = dims->n_vocab, P = dims->n_pos, TY = dims->n_type, NL = dims->n_labels;
    if (d <= 0 || d % 128 || dims->n_head * 64 != d || L <= 0 || V <= 0 || P <= 0 || P > 512 || TY <= 0 || NL <= 0 || NL > 128)
        return pce_fail(c, PCE_E_LIMIT, "unsupported BERT dims (need n_state %% 128 == 0, head size 64, n_pos <= 512, n_labels <= 128)");
    const int64_t dd = (int64_t)d * d;
    const int64_t per_layer = 4 * (dd + d) + 2LL * d + (4 * dd + 4LL * d) + (4 * dd + d) + 2LL * d;
    const int64_t expect = ((int64_t)V + P + TY) * d + 2LL * d + L * per_layer + (int64_t)NL * (d + 1);
    if (n_floats != expect) return pce_fail(c, PCE_E_INVALID, "BERT weight blob has %lld floats, expected %lld", (long long)n_floats, (long long)expect);
    PCE_HIP(c, hipSetDevice(c->device));
    WhisperState::Bert &b = ws_of(c)->bert;
    b.dims = *dims; b.loaded = false; b.n_seq = -1; b.layers.assign((size_t)L, {});
    std::vector<float> mats, vecs;
    auto add_vec = [&](const float *p, size_t n) { size_t o = vecs.size(); vecs.insert(vecs.end(), p, p + n); return o; };
    auto add_mat = [&](const float *p, size_t n) { size_t o = mats.size(); mats.insert(mats.end(), p, p + n); return o; };
    const float *p = weights;
    const float *word = p; p += (size_t)V * d;
    const float *pos = p; p += (size_t)P * d;
    const float *type = p; p += (size_t)TY * d;
    b.lne_w = add_vec(p, (size_t)d); p += d; b.lne_b = add_vec(p, (size_t)d); p += d;
    for (int l = 0; l < L; l++) {
        WhisperState::Bert::Layer &ly = b.layers[(size_t)l];
        const float *qw = p, *qb = qw + dd, *kw = qb + d, *kb = kw + dd, *vw = kb + d, *vb = vw + dd;
        ly.qkv_w = add_mat(qw, (size_t)dd); add_mat(kw, (size_t)dd); add_mat(vw, (size_t)dd);          // fused [3d][d]
        ly.qkv_b = add_vec(qb, (size_t)d); add_vec(kb, (size_t)d); add_vec(vb, (size_t)d);
        p = vb + d;
        ly.out_w = add_mat(p, (size_t)dd); p += dd; ly.out_b = add_vec(p, (size_t)d); p += d;
        ly.ln1_w = add_vec(p, (size_t)d); p += d; ly.ln1_b = add_vec(p, (size_t)d); p += d;
        ly.m1_w = add_mat(p, (size_t)(4 * dd)); p += 4 * dd; ly.m1_b = add_vec(p, (size_t)4 * d); p += 4 * d;
        ly.m2_w = add_mat(p, (size_t)(4 * dd)); p += 4 * dd; ly.m2_b = add_vec(p, (size_t)d); p += d;
        ly.ln2_w = add_vec(p, (size_t)d); p += d; ly.ln2_b = add_vec(p, (size_t)d); p += d;
    }
    {   // classifier [n_labels][d] padded with zero rows to one 128-column GEMM tile
        std::vector<float> cw((size_t)128 * d, 0.f), cb(128, 0.f);
        memcpy(cw.data(), p, sizeof(float) * (size_t)NL * d); p += (size_t)NL * d;
        memcpy(cb.data(), p, sizeof(float) * (size_t)NL); p += NL;
        b.cls_w = add_mat(cw.data(), cw.size()); b.cls_b = add_vec(cb.data(), cb.size());
    }
    DevBuf tmp;
    PCE_HIP(c, tmp.reserve(sizeof(float) * mats.size()));
    PCE_HIP(c, b.w_bf16.reserve(sizeof(bf16) * mats.size() + 256));
    PCE_HIP(c, b.w_f32.reserve(sizeof(float) * vecs.size()));
    PCE_HIP(c, b.word.reserve(sizeof(float) * (size_t)V * d));
    PCE_HIP(c, b.pos.reserve(sizeof(float) * (size_t)P * d));
    PCE_HIP(c, b.type0.reserve(sizeof(float) * (size_t)d));
    PCE_HIP(c, hipMemcpyAsync(tmp.p, mats.data(), sizeof(float) * mats.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(b.w_f32.p, vecs.data(), sizeof(float) * vecs.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(b.word.p, word, sizeof(float) * (size_t)V * d, hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(b.pos.p, pos, sizeof(float) * (size_t)P * d, hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(b.type0.p, type, sizeof(float) * (size_t)d, hipMemcpyHostToDevice, c->stream));      // token_type_ids = 0
    hipLaunchKernelGGL(k_f32_to_bf16, dim3((unsigned)div_up((int64_t)mats.size(), 256)), dim3(256), 0, c->stream, tmp.as<float>(),
                       b.w_bf16.as<bf16>(), (int64_t)mats.size());
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    tmp.release();
    b.loaded = true;
    return PCE_OK;
}

int pce_bert_run(pce_ctx *c, const int32_t *input_ids, const int32_t *offsets, int32_t n_seq)
{
    if (!c || !input_ids || !offsets || n_seq < 0) return PCE_E_INVALID;
    WhisperState::Bert &b = ws_of(c)->bert;
    if (!b.loaded) return pce_fail(c, PCE_E_STATE, "pce_bert_run before pce_bert_load");
    PCE_HIP(c, hipSetDevice(c->device));
    const int d = b.dims.n_state, H = b.dims.n_head, L = b.dims.n_layer, V = b.dims.n_vocab, n = n_seq, SPD = 512;
    b.n_seq = -1;
    b.lens.assign((size_t)n, 0);
    int T_max = 0;
    for (int i = 0; i < n; i++) {
        const int T = offsets[i + 1] - offsets[i];
        if (T < 1 || T > b.dims.n_pos) return pce_fail(c, PCE_E_INVALID, "sequence %d: %d tokens (need 1..%d)", i, T, b.dims.n_pos);
        b.lens[(size_t)i] = T; T_max = std::max(T_max, T);
    }
    if (n == 0) { b.n_seq = 0; return PCE_OK; }
    const int T_pad = (int)div_up(T_max, 64) * 64;
    const int64_t M = (int64_t)n * T_pad;
    std::vector<int> tab((size_t)2 * n), tok((size_t)M, 0);
    for (int i = 0; i < n; i++) {
        tab[(size_t)i] = i * T_pad; tab[(size_t)n + i] = b.lens[(size_t)i];
        for (int t = 0; t < b.lens[(size_t)i]; t++) {
            const int v = input_ids[offsets[i] + t];
            if (v < 0 || v >= V) return pce_fail(c, PCE_E_INVALID, "token id %d out of the vocabulary", v);
            tok[(size_t)i * T_pad + t] = v;
        }
    }
    PCE_HIP(c, b.tab.reserve(sizeof(int) * tab.size()));
    PCE_HIP(c, b.tokens.reserve(sizeof(int) * tok.size()));
    PCE_HIP(c, b.resid.reserve(sizeof(float) * (size_t)M * d));
    PCE_HIP(c, b.ln.reserve(sizeof(bf16) * (size_t)M * d + 4096));
    PCE_HIP(c, b.qk.reserve(sizeof(bf16) * (size_t)M * 2 * d + 4096));
    PCE_HIP(c, b.attn.reserve(sizeof(bf16) * (size_t)M * d + 4096));
    PCE_HIP(c, b.hidden.reserve(sizeof(bf16) * (size_t)M * 4 * d + 4096));
    PCE_HIP(c, b.logits.reserve(sizeof(float) * (size_t)M * 128));
    const size_t vt_elems = (size_t)n * (size_t)d * SPD + 64;
    PCE_HIP(c, b.vt.reserve(sizeof(bf16) * vt_elems));
    PCE_HIP(c, hipMemcpyAsync(b.tab.p, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(b.tokens.p, tok.data(), sizeof(int) * tok.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    // the attention kernel writes the rows of real tokens only and V^T columns beyond T_pad are never written: no stale bits
    PCE_HIP(c, hipMemsetAsync(b.attn.p, 0, sizeof(bf16) * (size_t)M * d + 4096, c->stream));
    PCE_HIP(c, hipMemsetAsync(b.vt.p, 0, sizeof(bf16) * vt_elems, c->stream));
    PCE_HIP(c, hipMemsetAsync(b.logits.p, 0, sizeof(float) * (size_t)M * 128, c->stream));
    const int *T0 = b.tab.as<int>(), *TL = T0 + n;
    const bf16 *Wb = b.w_bf16.as<bf16>();
    const float *Wf = b.w_f32.as<float>();
    const float eps = 1e-12f;                                    // BertConfig.layer_norm_eps
    KernelTimer timer(c, PCE_K_BERT);
    hipLaunchKernelGGL(k_bert_embed, dim3((unsigned)div_up(M * d, 256)), dim3(256), 0, c->stream, b.tokens.as<int>(), b.word.as<float>(),
                       b.pos.as<float>(), b.type0.as<float>(), T_pad, b.dims.n_pos, d, M, b.resid.as<float>());
    auto ln = [&](size_t w_off, size_t b_off) {                  // resid <- LN(resid) (fp32, in place) and its bf16 copy
        hipLaunchKernelGGL((k_layernorm<bf16>), dim3((unsigned)div_up(M, 4)), dim3(256), 0, c->stream, b.resid.as<float>(), Wf + w_off, Wf + b_off,
                           M, d, b.ln.as<bf16>(), eps, b.resid.as<float>());
    };
    ln(b.lne_w, b.lne_b);
    for (int l = 0; l < L; l++) {
        const WhisperState::Bert::Layer &ly = b.layers[(size_t)l];
        launch_gemm<EPI_QKV>(c, b.ln.as<bf16>(), d, 0, Wb + ly.qkv_w, (int)M, 3 * d, d, Wf + ly.qkv_b, b.qk.as<bf16>(), 2 * d, 0, 1,
                             reinterpret_cast<const float *>(b.vt.as<bf16>()), T_pad, 2 * d, SPD);
        {
            AttnArgs a{};
            a.q = b.qk.as<bf16>(); a.q_ld = 2 * d; a.k = b.qk.as<bf16>() + d; a.k_ld = 2 * d;
            a.vt = b.vt.as<bf16>(); a.vt_clip = (int64_t)d * SPD; a.vt_sp = SPD;
            a.q_row0 = a.k_row0 = T0; a.q_len = a.k_len = TL;    // keys beyond the sequence length are masked (right padding)
            a.out = b.attn.as<bf16>(); a.out_ld = d; a.causal = 0;
            launch_attention(c, dim3((unsigned)div_up(T_pad, AT_QB), (unsigned)H, (unsigned)n), a);
        }
        launch_gemm<EPI_RESID_F32>(c, b.attn.as<bf16>(), d, 0, Wb + ly.out_w, (int)M, d, d, Wf + ly.out_b, b.resid.as<float>(), d, 0, 1);
        ln(ly.ln1_w, ly.ln1_b);
        launch_gemm<EPI_GELU_BF16>(c, b.ln.as<bf16>(), d, 0, Wb + ly.m1_w, (int)M, 4 * d, d, Wf + ly.m1_b, b.hidden.as<bf16>(), 4 * d, 0, 1);
        launch_gemm<EPI_RESID_F32>(c, b.hidden.as<bf16>(), 4 * d, 0, Wb + ly.m2_w, (int)M, d, 4 * d, Wf + ly.m2_b, b.resid.as<float>(), d, 0, 1);
        ln(ly.ln2_w, ly.ln2_b);
    }
    // classifier: logits[M][128] = 0 + hidden . W^T + b (the first n_labels columns are real)
    launch_gemm<EPI_RESID_F32>(c, b.ln.as<bf16>(), d, 0, Wb + b.cls_w, (int)M, 128, d, Wf + b.cls_b, b.logits.as<float>(), 128, 0, 1);
    PCE_HIP(c, hipGetLastError());
    b.n_seq = n; b.T_pad = T_pad;
    return PCE_OK;
}

int pce_bert_fetch(pce_ctx *c, int32_t seq, float *logits, int32_t *labels)
{
    if (!c) return PCE_E_INVALID;
    WhisperState::Bert &b = ws_of(c)->bert;
    if (b.n_seq < 0) return pce_fail(c, PCE_E_STATE, "pce_bert_fetch before pce_bert_run");
    if (seq < 0 || seq >= b.n_seq) return pce_fail(c, PCE_E_INVALID, "sequence out of range");
    PCE_HIP(c, hipSetDevice(c->device));
    const int T = b.lens[(size_t)seq], NL = b.dims.n_labels;
    std::vector<float> rows((size_t)T * 128);
    PCE_HIP(c, hipMemcpyAsync(rows.data(), b.logits.as<float>() + (size_t)seq * b.T_pad * 128, sizeof(float) * rows.size(), hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    pce_profile_collect(c);
    for (int t = 0; t < T; t++) {
        int best = 0;
        for (int k = 0; k < NL; k++) {
            const float v = rows[(size_t)t * 128 + k];
            if (logits) logits[(size_t)t * NL + k] = v;
            if (v > rows[(size_t)t * 128 + best]) best = k;          // first maximum, as argmax
        }
        if (labels) labels[t] = best;
    }
    return PCE_OK;
}

} // extern "C"

