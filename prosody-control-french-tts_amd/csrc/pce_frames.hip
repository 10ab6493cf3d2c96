// pce_frames.hip -- k_frame_energy: exact integer short-time energy of every analysis window of every clip.
//
// Reference step replaced: the energy detector of the "auditok" VAD that the aligner asks for
// (Code/Aligners/use_whisper_timestamped.py:152 `"vad": "auditok"` -> whisper-timestamped get_vad_segments ->
// auditok.split(energy_threshold=50): 50 ms analysis windows, 20*log10(sqrt(mean(x^2))) per window, third-party and
// absent from /root/reference: restated from the packages' published behaviour, parity unpinned).  The window sums are
// exact integers, so the few floating-point finishing operations (mean, sqrt, log10, threshold) are host logic, as for
// k_energy.  `requantize` applies, per sample, the float32 round trip whisper-timestamped performs before the VAD
// (int16 / 32768 as whisper.load_audio leaves it, then (audio * 32767).astype(int16), truncation toward zero).
//
// Frames: frame k of a clip of n samples covers [k * hop, min(k * hop + window, n)), k = 0 .. ceil(n / hop) - 1; with
// hop == window these are auditok's blocks (the last one short, not padded).
//
// Roofline: HBM-bound, 2 bytes per sample read once when hop == window, 12 bytes written per frame.
// One wavefront per frame: 16-byte loads (8 samples per lane and load), register sums, wave shuffle reduction.
#include "pce_internal.h"

namespace {

constexpr int FR_THREADS = 256;
constexpr int FR_FPW = 8;                 // frames per wavefront and trip on the common path (windows of at most 1 024 samples)

// wave-wide integer sum without LDS round trips (see pce_energy.hip): DPP inside the 16-lane rows, v_readlane across them
template <int CTRL> __device__ __forceinline__ int fr_dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ int fr_wave_sum(int v)
{
    v += fr_dpp<0xB1>(v); v += fr_dpp<0x4E>(v); v += fr_dpp<0x141>(v); v += fr_dpp<0x140>(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// FR_FPW frames per wavefront on the common path (a window of at most 1 024 samples = two 16-byte loads per lane): the 2 FR_FPW loads of a
// lane are all requested before the first is used and a wavefront lives FPW times longer; one frame per wavefront (rounds 1-3a) left
// 1.6 KB in flight per wave and spent most of a wave's life on launch and the 64-bit shuffle reduction: 92 us at 400 MB (4.3 TB/s).
// NT: non-temporal loads for a batch beyond the Infinity Cache.
template <bool NT, bool REQ>
__global__ __launch_bounds__(FR_THREADS) void k_frame_energy(const int16_t *__restrict__ pcm, const int64_t *__restrict__ clip_off,
                                                            const int64_t *__restrict__ frame_off, int n_clips, int window, int hop,
                                                            long long *__restrict__ sum_sq, int *__restrict__ count)
{
    typedef int i4 __attribute__((ext_vector_type(4)));
    constexpr int FPW = FR_FPW;
    constexpr bool requantize = REQ;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    auto ld = [&](const int16_t *p) -> i4 { return NT ? __builtin_nontemporal_load(reinterpret_cast<const i4 *>(p)) : *reinterpret_cast<const i4 *>(p); };
    // r = index of the load's first sample within the frame (-7 ..), nfr = samples in the frame.  Branch-free: samples outside the frame
    // are masked to zero in the packed words (two compares, two selects per word), then two samples are squared per instruction
    // (v_dot2_i32_i16); the per-sample route with its range tests cost 220 registers once eight frames were in flight per wave.
    auto consume = [&](const i4 v, int r, int nfr, unsigned long long &s) {
        const int words[4] = {v.x, v.y, v.z, v.w};
        typedef short s2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned keep = ((unsigned)(r + 2 * q) < (unsigned)nfr ? 0x0000FFFFu : 0u) | ((unsigned)(r + 2 * q + 1) < (unsigned)nfr ? 0xFFFF0000u : 0u);
            const int wq = words[q] & (int)keep;
            if (!requantize) {
                const s2 xv = __builtin_bit_cast(s2, wq);
                s += (unsigned long long)(unsigned int)__builtin_amdgcn_sdot2(xv, xv, 0, false);   // <= 2^31: exact as unsigned
            } else {
                const int x0 = __float2int_rz(((float)(int)(short)(wq & 0xFFFF) * (1.0f / 32768.0f)) * 32767.0f);
                const int x1 = __float2int_rz(((float)(wq >> 16) * (1.0f / 32768.0f)) * 32767.0f);
                s += (unsigned long long)(unsigned int)(x0 * x0) + (unsigned long long)(unsigned int)(x1 * x1);
            }
        }
    };
    // a lane's share of a frame is below 2^40 whenever the frame has at most 2^16 samples: low 20 bits and the rest both sum to < 2^31 over the wave
    auto wave_total = [&](unsigned long long s) -> unsigned long long {
        const int lo = fr_wave_sum((int)(s & 0xFFFFFull)), hi = fr_wave_sum((int)(s >> 20));
        return ((unsigned long long)(unsigned int)hi << 20) + (unsigned long long)(unsigned int)lo;
    };
    for (int clip = blockIdx.y; clip < n_clips; clip += gridDim.y) {
        const int64_t c0 = clip_off[clip], len = clip_off[clip + 1] - c0;
        const int64_t f0 = frame_off[clip], nf = frame_off[clip + 1] - f0;
        if (window <= 1024) {
            for (int64_t kb = ((int64_t)blockIdx.x * (FR_THREADS / 64) + wv) * FPW; kb < nf; kb += (int64_t)gridDim.x * (FR_THREADS / 64) * FPW) {
                // one wave-uniform base (aligned down to a 16-byte group) and unsigned 32-bit lane offsets: no 64-bit address per load
                const int64_t gbase = c0 + kb * hop, abase = gbase & ~(int64_t)7;
                const int16_t *pb = pcm + abase;
                const int left = (int)(nf - kb < FPW ? nf - kb : FPW);                          // frames of this trip
                const int avail = (int)(len - kb * hop < (int64_t)FPW * window ? len - kb * hop : (int64_t)FPW * window);   // samples of the clip from frame kb on (capped)
                i4 v[FPW][2]; int rr[FPW], nn[FPW];
#pragma unroll
                for (int f = 0; f < FPW; f++) {
                    const int start = (int)(gbase - abase) + f * hop;                           // of frame kb + f, relative to abase (>= 0)
                    const int astart = start & ~7;
                    int n = avail - f * hop; if (n > window) n = window;
                    nn[f] = f < left ? n : 0;
                    rr[f] = astart - start + lane * 8;
                    const unsigned uo = (unsigned)(astart + lane * 8);
                    v[f][0] = (i4){0, 0, 0, 0}; v[f][1] = (i4){0, 0, 0, 0};
                    if (rr[f] < nn[f]) v[f][0] = ld(pb + uo);
                    if (rr[f] + 512 < nn[f]) v[f][1] = ld(pb + uo + 512u);
                }
#pragma unroll
                for (int f = 0; f < FPW; f++) {
                    unsigned long long s = 0;
                    // (the range masks do not depend on the data: left alone the compiler builds all 64 of them while the loads are in
                    //  flight -- 256 registers; tying the frame's offset to its first loaded word keeps them behind the wait)
                    int rf = rr[f];
                    asm volatile("" : "+v"(rf) : "v"(v[f][0].x));
                    consume(v[f][0], rf, nn[f], s);              // (a load that was not made holds zeros)
                    consume(v[f][1], rf + 512, nn[f], s);
                    const unsigned long long t = wave_total(s);
                    if (lane == 0 && f < left) { sum_sq[f0 + kb + f] = (long long)t; count[f0 + kb + f] = nn[f]; }
                }
            }
            continue;
        }
        for (int64_t k = (int64_t)blockIdx.x * (FR_THREADS / 64) + wv; k < nf; k += (int64_t)gridDim.x * (FR_THREADS / 64)) {
            const int64_t b = k * hop;
            const int64_t e = b + window < len ? b + window : len;
            const int64_t g0 = c0 + b;
            unsigned long long s = 0;
            const int nfr = (int)(e - b);                       // samples in this frame (<= window)
            // two 16-byte loads per lane in flight
            const int64_t a0 = g0 & ~(int64_t)7;
            const int r0 = (int)(a0 - g0);
            for (int r = r0 + lane * 8; r < nfr; r += 2 * 64 * 8) {
                const int r1 = r + 64 * 8;
                const i4 v0 = ld(pcm + g0 + r);
                const i4 v1 = r1 < nfr ? ld(pcm + g0 + r1) : (i4){0, 0, 0, 0};
                consume(v0, r, nfr, s);
                consume(v1, r1, nfr, s);
            }
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0) { sum_sq[f0 + k] = (long long)s; count[f0 + k] = (int)(e - b); }
        }
    }
}

} // namespace

extern "C" {

int pce_frame_energy_run(pce_ctx *c, int32_t window, int32_t hop, int32_t requantize)
{
    if (!c) return PCE_E_INVALID;
    if (!c->d_pcm) return pce_fail(c, PCE_E_STATE, "no batch uploaded");
    if (window < 1 || hop < 1 || hop > window) return pce_fail(c, PCE_E_INVALID, "frame energy: need 1 <= hop <= window");
    PCE_HIP(c, hipSetDevice(c->device));
    c->fr_off.assign((size_t)c->n_clips + 1, 0);
    int64_t max_frames = 0;
    for (int32_t i = 0; i < c->n_clips; i++) {
        const int64_t len = c->clip_off[(size_t)i + 1] - c->clip_off[(size_t)i];
        const int64_t nf = (len + hop - 1) / hop;
        c->fr_off[(size_t)i + 1] = c->fr_off[(size_t)i] + nf;
        if (nf > max_frames) max_frames = nf;
    }
    const int64_t total = c->fr_off[(size_t)c->n_clips];
    PCE_HIP(c, c->fr_doff.reserve(sizeof(int64_t) * ((size_t)c->n_clips + 1)));
    PCE_HIP(c, c->fr_sum.reserve(sizeof(long long) * (size_t)(total > 0 ? total : 1)));
    PCE_HIP(c, c->fr_cnt.reserve(sizeof(int) * (size_t)(total > 0 ? total : 1)));
    PCE_HIP(c, hipMemcpyAsync(c->fr_doff.p, c->fr_off.data(), sizeof(int64_t) * ((size_t)c->n_clips + 1), hipMemcpyHostToDevice, c->stream));
    if (total > 0) {
        const unsigned gy = (unsigned)(c->n_clips < 65535 ? c->n_clips : 65535);
        const int per_wg = (FR_THREADS / 64) * (window <= 1024 ? FR_FPW : 1);     // frames per workgroup and trip
        int64_t gx = (max_frames + per_wg - 1) / per_wg;
        if (gx > 4096) gx = 4096;
        const bool nt = c->clip_off.back() * 2 > ((int64_t)256 << 20);      // the batch does not fit the Infinity Cache
        KernelTimer t(c, PCE_K_FRAME_ENERGY);
        auto launch = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3((unsigned)gx, gy), dim3(FR_THREADS), 0, c->stream, c->d_pcm, c->d_clip_off.as<int64_t>(),
                               c->fr_doff.as<int64_t>(), (int)c->n_clips, (int)window, (int)hop, c->fr_sum.as<long long>(), c->fr_cnt.as<int>());
        };
        if (requantize) { if (nt) launch(k_frame_energy<true, true>); else launch(k_frame_energy<false, true>); }
        else { if (nt) launch(k_frame_energy<true, false>); else launch(k_frame_energy<false, false>); }
        PCE_HIP(c, hipGetLastError());
    }
    PCE_HIP(c, hipStreamSynchronize(c->stream));                 // fr_off (host vector) was the source of an async copy
    c->fr_ran = true;
    return PCE_OK;
}

int pce_frame_energy_shape(pce_ctx *c, int32_t clip, int64_t *n_frames)
{
    if (!c || !n_frames) return PCE_E_INVALID;
    if (!c->fr_ran) return pce_fail(c, PCE_E_STATE, "pce_frame_energy_shape before pce_frame_energy_run");
    if (clip < 0 || clip >= c->n_clips) return pce_fail(c, PCE_E_INVALID, "clip out of range");
    *n_frames = c->fr_off[(size_t)clip + 1] - c->fr_off[(size_t)clip];
    return PCE_OK;
}

int pce_frame_energy_fetch(pce_ctx *c, int32_t clip, int64_t *sum_sq, int32_t *count)
{
    if (!c) return PCE_E_INVALID;
    if (!c->fr_ran) return pce_fail(c, PCE_E_STATE, "pce_frame_energy_fetch before pce_frame_energy_run");
    if (clip < 0 || clip >= c->n_clips) return pce_fail(c, PCE_E_INVALID, "clip out of range");
    PCE_HIP(c, hipSetDevice(c->device));
    const int64_t f0 = c->fr_off[(size_t)clip], nf = c->fr_off[(size_t)clip + 1] - f0;
    if (nf > 0 && sum_sq)
        PCE_HIP(c, hipMemcpyAsync(sum_sq, c->fr_sum.as<long long>() + f0, sizeof(int64_t) * (size_t)nf, hipMemcpyDeviceToHost, c->stream));
    if (nf > 0 && count)
        PCE_HIP(c, hipMemcpyAsync(count, c->fr_cnt.as<int>() + f0, sizeof(int32_t) * (size_t)nf, hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    pce_profile_collect(c);
    return PCE_OK;
}

} // extern "C"
