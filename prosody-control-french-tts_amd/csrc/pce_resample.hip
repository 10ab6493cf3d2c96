// pce_resample.hip -- rational-rate polyphase resampler on the resident batch.
//
// The reference decodes every file to 16 kHz with ffmpeg inside whisper.load_audio
// (Code/Aligners/use_whisper_timestamped.py:139; the demo recordings are 44.1 kHz).  ffmpeg's
// swresample is not reproducible here, so the engine defines its own spec and the oracle pins
// it: y = upfirdn(h, x, up, down) with the windowed-sinc low-pass h the host supplies
// (hostrules.resample_filter: scipy.signal.resample_poly's design, Kaiser beta 5, 10 * max(up, down)
// half length), fp64 accumulation in tap order, round-half-even to int16 with saturation.
//
// One thread per output sample, about len(h)/up (55 for 160/441) taps each: the batch is a few
// hundred MB at most and the kernel is a pure gather/FMA stream out of L2.
#include "pce_internal.h"
#include <cmath>

namespace {

__global__ __launch_bounds__(256) void k_resample(const int16_t *__restrict__ in, const int64_t *__restrict__ in_off,
                                                 const int64_t *__restrict__ out_off, int n_clips, int up, int down,
                                                 const double *__restrict__ h, int n_taps, int64_t n_pre_remove, int16_t *__restrict__ out)
{
    const int clip = blockIdx.y;
    const int64_t n_in = in_off[clip + 1] - in_off[clip], n_out = out_off[clip + 1] - out_off[clip];
    const int16_t *x = in + in_off[clip];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = (i + n_pre_remove) * down;          // position on the up-sampled grid
        // y[n] = sum_m h[n - m up] x[m], 0 <= n - m up < n_taps, 0 <= m < n_in
        int64_t m_hi = n / up; if (m_hi > n_in - 1) m_hi = n_in - 1;
        int64_t m_lo = (n - n_taps + 1 + up - 1) / up; if (n - n_taps + 1 < 0) m_lo = 0; if (m_lo < 0) m_lo = 0;
        double acc = 0.0;
        for (int64_t m = m_lo; m <= m_hi; m++) acc = fma(h[n - m * up], (double)x[m], acc);
        double r = rint(acc);                                  // round half to even, like np.rint
        r = r > 32767.0 ? 32767.0 : (r < -32768.0 ? -32768.0 : r);
        out[out_off[clip] + i] = (int16_t)r;
    }
}

} // namespace

extern "C" {

int pce_resample_run(pce_ctx *c, int32_t up, int32_t down, const double *taps, int32_t n_taps, int64_t n_pre_remove)
{
    if (!c || !taps || up <= 0 || down <= 0 || n_taps <= 0 || n_pre_remove < 0) return PCE_E_INVALID;
    if (!c->d_pcm) return pce_fail(c, PCE_E_STATE, "no batch uploaded");
    { int rc = pce_join_aux(c); if (rc) return rc; }           // side-stream kernels may still read the batch this call replaces
    if (((int64_t)c->rate * up) % down) return pce_fail(c, PCE_E_INVALID, "rate %d * %d / %d is not an integer", c->rate, up, down);
    PCE_HIP(c, hipSetDevice(c->device));
    const int32_t n = c->n_clips;
    std::vector<int64_t> new_off((size_t)n + 1, 0);
    for (int32_t i = 0; i < n; i++) {
        const int64_t n_in = c->clip_off[(size_t)i + 1] - c->clip_off[(size_t)i];
        const int64_t t = n_in * up;
        new_off[(size_t)i + 1] = new_off[(size_t)i] + (t / down + (t % down ? 1 : 0));      // resample_poly's n_out
    }
    const size_t total = (size_t)new_off[(size_t)n];
    DevBuf d_taps, d_newoff, d_out;
    PCE_HIP(c, d_taps.reserve(sizeof(double) * (size_t)n_taps));
    PCE_HIP(c, d_newoff.reserve(sizeof(int64_t) * (size_t)(n + 1)));
    PCE_HIP(c, d_out.reserve(total * 2 + 64));
    PCE_HIP(c, hipMemcpyAsync(d_taps.p, taps, sizeof(double) * (size_t)n_taps, hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(d_newoff.p, new_off.data(), sizeof(int64_t) * (size_t)(n + 1), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemsetAsync((char *)d_out.p + total * 2, 0, 64, c->stream));
    {
        KernelTimer t(c, PCE_K_RESAMPLE);
        int64_t longest = 0;
        for (int32_t i = 0; i < n; i++) longest = std::max<int64_t>(longest, new_off[(size_t)i + 1] - new_off[(size_t)i]);
        const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(div_up(longest, 256), 4096));
        hipLaunchKernelGGL(k_resample, dim3(gx, (unsigned)n), dim3(256), 0, c->stream, c->d_pcm, c->d_clip_off.as<int64_t>(),
                           d_newoff.as<int64_t>(), (int)n, (int)up, (int)down, d_taps.as<double>(), (int)n_taps, n_pre_remove,
                           d_out.as<int16_t>());
    }
    PCE_HIP(c, hipGetLastError());
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    // the resampled batch becomes the resident batch
    c->pcm_own.release();
    c->pcm_own = d_out; d_out.p = nullptr; d_out.cap = 0;
    c->d_pcm = c->pcm_own.as<const int16_t>();
    c->d_clip_off.release();
    c->d_clip_off = d_newoff; d_newoff.p = nullptr; d_newoff.cap = 0;
    c->clip_off = new_off;
    c->rate = (int32_t)(((int64_t)c->rate * up) / down);
    c->en_n = c->lu_n = c->pi_n = -1; c->st_nfft = 0; c->st_ran = false; c->fr_ran = false; c->py_ran = false;
    c->en_cache.drop(); c->lu_cache.drop(); c->pi_cache.drop();
    d_taps.release();
    return PCE_OK;
}

int pce_download_pcm_s16(pce_ctx *c, int16_t *pcm, int64_t *offsets, int32_t *sample_rate)
{
    if (!c) return PCE_E_INVALID;
    if (!c->d_pcm) return pce_fail(c, PCE_E_STATE, "no batch uploaded");
    PCE_HIP(c, hipSetDevice(c->device));
    if (offsets) memcpy(offsets, c->clip_off.data(), sizeof(int64_t) * (size_t)(c->n_clips + 1));
    if (sample_rate) *sample_rate = c->rate;
    if (pcm) {
        PCE_HIP(c, hipMemcpyAsync(pcm, c->d_pcm, (size_t)c->clip_off[(size_t)c->n_clips] * 2, hipMemcpyDeviceToHost, c->stream));
        PCE_HIP(c, hipStreamSynchronize(c->stream));
    }
    return PCE_OK;
}

} // extern "C"
