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

__global__ __launch_bounds__(FR_THREADS) void k_frame_energy(const int16_t *__restrict__ pcm, const int64_t *__restrict__ clip_off,
                                                            const int64_t *__restrict__ frame_off, int n_clips, int window, int hop,
                                                            int requantize, long long *__restrict__ sum_sq, int *__restrict__ count)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int clip = blockIdx.y; clip < n_clips; clip += gridDim.y) {
        const int64_t c0 = clip_off[clip], len = clip_off[clip + 1] - c0;
        const int64_t f0 = frame_off[clip], nf = frame_off[clip + 1] - f0;
        for (int64_t k = (int64_t)blockIdx.x * (FR_THREADS / 64) + wv; k < nf; k += (int64_t)gridDim.x * (FR_THREADS / 64)) {
            const int64_t b = k * hop;
            const int64_t e = b + window < len ? b + window : len;
            const int64_t g0 = c0 + b;
            unsigned long long s = 0;
            const int nfr = (int)(e - b);                       // samples in this frame (<= window)
            auto consume = [&](const int4 v, int r) {            // r = index of the load's first sample within the frame (-7 ..)
                const int words[4] = {v.x, v.y, v.z, v.w};
                const bool inside = r >= 0 && r + 8 <= nfr;        // interior loads skip the per-sample range tests
                if (inside && !requantize) {                       // ... and square two samples per instruction (v_dot2_i32_i16)
                    typedef short s2 __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const s2 xv = __builtin_bit_cast(s2, words[q]);
                        s += (unsigned long long)(unsigned int)__builtin_amdgcn_sdot2(xv, xv, 0, false);
                    }
                    return;
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    int x0 = (int)(short)(words[q] & 0xFFFF), x1 = words[q] >> 16;
                    if (!inside) {
                        if ((unsigned)(r + 2 * q) >= (unsigned)nfr) x0 = 0;
                        if ((unsigned)(r + 2 * q + 1) >= (unsigned)nfr) x1 = 0;
                    }
                    if (requantize) {
                        x0 = __float2int_rz(((float)x0 * (1.0f / 32768.0f)) * 32767.0f);
                        x1 = __float2int_rz(((float)x1 * (1.0f / 32768.0f)) * 32767.0f);
                    }
                    s += (unsigned long long)(unsigned int)(x0 * x0) + (unsigned long long)(unsigned int)(x1 * x1);
                }
            };
            // two 16-byte loads per lane in flight (a 50 ms window at 16 kHz is 1.6 KB: one round)
            const int64_t a0 = g0 & ~(int64_t)7;
            const int r0 = (int)(a0 - g0);
            for (int r = r0 + lane * 8; r < nfr; r += 2 * 64 * 8) {
                const int r1 = r + 64 * 8;
                const int4 v0 = *reinterpret_cast<const int4 *>(pcm + g0 + r);
                const int4 v1 = r1 < nfr ? *reinterpret_cast<const int4 *>(pcm + g0 + r1) : make_int4(0, 0, 0, 0);
                consume(v0, r);
                consume(v1, r1);
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
        int64_t gx = (max_frames + FR_THREADS / 64 - 1) / (FR_THREADS / 64);
        // one wave per frame: a resident grid (8 workgroups per CU, several frames per wave) measured 10 % slower
        if (gx > 4096) gx = 4096;
        KernelTimer t(c, PCE_K_FRAME_ENERGY);
        hipLaunchKernelGGL(k_frame_energy, dim3((unsigned)gx, gy), dim3(FR_THREADS), 0, c->stream, c->d_pcm, c->d_clip_off.as<int64_t>(),
                           c->fr_doff.as<int64_t>(), (int)c->n_clips, (int)window, (int)hop, (int)(requantize != 0), c->fr_sum.as<long long>(),
                           c->fr_cnt.as<int>());
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
