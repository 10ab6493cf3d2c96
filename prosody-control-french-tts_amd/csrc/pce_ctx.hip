// pce_ctx.hip -- context, batch residency, profiling brackets of libpce.so.
#include "pce_internal.h"
#include <cstdlib>
#include <cstring>
#include <cstdarg>

int pce_fail(pce_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

KernelTimer::KernelTimer(pce_ctx *ctx, int kid, hipStream_t on, double work_flops) : c(ctx), id(kid), s(on ? on : ctx->stream), flops(work_flops)
{
    if (!c->prof) return;
    auto take = [&]() -> hipEvent_t {
        if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    };
    a = take(); b = take();
    if (a) (void)hipEventRecord(a, s);
}
KernelTimer::~KernelTimer()
{
    if (!c->prof || !a || !b) return;
    (void)hipEventRecord(b, s);
    c->pending.push_back({id, a, b, flops});
}
int pce_side_join(pce_ctx *c, int which)
{
    pce_ctx::Side &sd = c->side[which];
    if (sd.pending) {
        PCE_HIP(c, hipStreamWaitEvent(c->stream, sd.join, 0));
        sd.pending = false;
    }
    return PCE_OK;
}
int pce_join_aux(pce_ctx *c)
{
    for (int i = 0; i < pce_ctx::SIDE_COUNT; i++) { int rc = pce_side_join(c, i); if (rc) return rc; }
    return PCE_OK;
}
int pce_side_begin(pce_ctx *c, int which, hipStream_t *out)
{
    *out = c->stream;
    if (c->no_side) return PCE_OK;
    pce_ctx::Side &sd = c->side[which];
    PCE_HIP(c, hipEventRecord(sd.fork, c->stream));
    PCE_HIP(c, hipStreamWaitEvent(sd.s, sd.fork, 0));
    *out = sd.s;
    return PCE_OK;
}
int pce_side_end(pce_ctx *c, int which, hipStream_t used)
{
    if (used == c->stream) return PCE_OK;
    pce_ctx::Side &sd = c->side[which];
    PCE_HIP(c, hipEventRecord(sd.join, used));
    sd.pending = true;
    return PCE_OK;
}

void pce_profile_collect(pce_ctx *ctx, bool wait)
{
    size_t kept = 0;
    for (auto &p : ctx->pending) {
        if (!wait && hipEventQuery(p.b) != hipSuccess) { ctx->pending[kept++] = p; continue; }   // still running
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            ctx->prof_ms[p.id] += ms;
            ctx->prof_n[p.id] += 1;
            ctx->prof_flops[p.id] += p.flops;
        }
        ctx->ev_pool.push_back(p.a);
        ctx->ev_pool.push_back(p.b);
    }
    ctx->pending.resize(kept);
}

extern "C" {

int pce_api_version(void) { return PCE_API_VERSION; }
int pce_api_minor(void) { return PCE_API_MINOR; }

pce_ctx *pce_create(int device, void *stream, char *err, size_t errlen)
{
    auto fail = [&](const char *what, hipError_t e) -> pce_ctx * {
        if (err && errlen) snprintf(err, errlen, "%s: %s", what, hipGetErrorString(e));
        return nullptr;
    };
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        if (err && errlen) snprintf(err, errlen, "no HIP device available (libpce has no CPU fallback): %s", hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= n) {
        if (err && errlen) snprintf(err, errlen, "device %d out of range (0..%d)", device, n - 1);
        return nullptr;
    }
    if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return fail("hipGetDeviceProperties", e);
    pce_ctx *c = new (std::nothrow) pce_ctx();
    if (!c) { if (err && errlen) snprintf(err, errlen, "out of host memory"); return nullptr; }
    c->device = device;
    c->cu_count = prop.multiProcessorCount;
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else {
        if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) { delete c; return fail("hipStreamCreate", e); }
        c->own_stream = true;
    }
    c->no_side = getenv("PCE_NO_AUX") != nullptr;
    c->stft_two_fft = getenv("PCE_STFT_TWO_FFT") != nullptr;
    c->generic_median = getenv("PCE_ALIGN_GENERIC_MEDIAN") != nullptr;
    c->gemm_flat = !(getenv("PCE_GEMM_FLAT") && atoi(getenv("PCE_GEMM_FLAT")) == 0);
    c->en_cpb = getenv("PCE_EN_CPB") ? atoi(getenv("PCE_EN_CPB")) : 0;
    c->gemm_skinny = !(getenv("PCE_GEMM_SKINNY") && atoi(getenv("PCE_GEMM_SKINNY")) == 0);
    // attention on v_mfma_f32_16x16x32 (round 5, default: 30.5 -> 29.6 ms per C3 step on one box, profiles/r05); PCE_ATTN_M16=0: the 32x32x16 kernel
    c->attn_m16 = !(getenv("PCE_ATTN_M16") && atoi(getenv("PCE_ATTN_M16")) == 0);
    c->self_rows = !(getenv("PCE_SELF_ROWS") && atoi(getenv("PCE_SELF_ROWS")) == 0);
    c->attn_nt = !(getenv("PCE_ATTN_NT") && atoi(getenv("PCE_ATTN_NT")) == 0);
    c->xattn_absorb = !(getenv("PCE_XATTN_ABSORB") && atoi(getenv("PCE_XATTN_ABSORB")) == 0);
    {
        const char *ops = getenv("PCE_WHISPER_OPERANDS");
        // default (round 4): fp16 operands AND the fp16 residual stream, the reference's own arithmetic end to end (openai-whisper fp16=True);
        // measured against the fp32 restatement at Whisper-small depth (tools/operand_precision.py, profiles/r04): encoder rel-L2 1.0e-3,
        // every word boundary identical, no greedy flip in 2 112 steps; "fp16" keeps the fp32 stream (4.6e-4), "bf16" the round-1 / 2 arithmetic
        if (ops && *ops && strcmp(ops, "bf16") && strcmp(ops, "fp16") && strcmp(ops, "fp16-resid16")) {
            // a misspelt mode must not silently select the default
            if (err && errlen) snprintf(err, errlen, "PCE_WHISPER_OPERANDS=%s: bf16, fp16 or fp16-resid16", ops);
            if (c->own_stream) (void)hipStreamDestroy(c->stream);
            delete c;
            return nullptr;
        }
        c->whisper_ops = (ops && !strcmp(ops, "bf16")) ? 0 : (ops && !strcmp(ops, "fp16")) ? 1 : 2;
        c->resid16 = c->whisper_ops == 2;
    }
    c->pitch_refine_praat = getenv("PCE_PITCH_REFINE") && !strcmp(getenv("PCE_PITCH_REFINE"), "praat");
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    for (int si = 0; si < pce_ctx::SIDE_COUNT; si++) {
        pce_ctx::Side &sd = c->side[si];
        // the STFT normalisation pass streams 0.4 GB beside the next batch's first kernels and the statistics copies:
        // lowest priority, so that those (small, latency critical) are dispatched first
        const int prio = si == pce_ctx::SIDE_STFT ? prio_least : 0;     // (measured: 3.02 -> 3.00 ms per step)
        if ((e = hipStreamCreateWithPriority(&sd.s, hipStreamNonBlocking, prio)) != hipSuccess) { delete c; return fail("hipStreamCreate", e); }
        if ((e = hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming)) != hipSuccess || (e = hipEventCreateWithFlags(&sd.join, hipEventDisableTiming)) != hipSuccess) { delete c; return fail("hipEventCreate", e); }
    }
    return c;
}

void pce_destroy(pce_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)pce_join_aux(c);
    (void)hipStreamSynchronize(c->stream);
    for (auto &sd : c->side) if (sd.s) (void)hipStreamSynchronize(sd.s);
    pce_profile_collect(c);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    for (auto &sd : c->side) {
        if (sd.fork) (void)hipEventDestroy(sd.fork);
        if (sd.join) (void)hipEventDestroy(sd.join);
        if (sd.s) (void)hipStreamDestroy(sd.s);
    }
    DevBuf *bufs[] = {&c->pcm_own, &c->d_clip_off, &c->en_work, &c->en_out,
                      &c->lu_meta, &c->lu_chunks, &c->lu_blocks, &c->lu_pow, &c->lu_state_end, &c->lu_state_init,
                      &c->lu_energy, &c->lu_zbuf, &c->lu_out, &c->lu_en_work, &c->lu_en_acc,
                      &c->pi_meta, &c->pi_window, &c->pi_windowR, &c->pi_work, &c->pi_cand, &c->pi_gpeak,
                      &c->pi_psi, &c->pi_f0, &c->pi_strength, &c->pi_summary, &c->pi_peakwork, &c->pi_acc, &c->pi_rr, &c->pi_items, &c->pi_tw, &c->pi_dl, &c->pi_runs, &c->pi_fslice, &c->pi_blob,
                      &c->st_out, &c->st_max, &c->st_off, &c->st_window, &c->st_twiddle, &c->st_work, &c->st_stage,
                      &c->fr_doff, &c->fr_sum, &c->fr_cnt,
                      &c->py_doff, &c->py_tab, &c->py_hdr, &c->py_bin, &c->py_lp, &c->py_ptr, &c->py_states};
    for (auto b : bufs) b->release();
    for (auto &st : c->stat) {
        if (st.host) (void)hipHostFree(st.host);
        if (st.ev) (void)hipEventDestroy(st.ev);
        if (st.ev_main) (void)hipEventDestroy(st.ev_main);
    }
    pce_whisper_free(c);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *pce_last_error(const pce_ctx *c) { return c ? c->err.c_str() : "null context"; }

int pce_sync(pce_ctx *c)
{
    if (!c) return PCE_E_INVALID;
    { int rc = pce_join_aux(c); if (rc) return rc; }
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    pce_profile_collect(c);
    return PCE_OK;
}

int pce_stats_enqueue(pce_ctx *c, int32_t slot)
{
    if (!c || slot < 0 || slot > 1) return PCE_E_INVALID;
    PCE_HIP(c, hipSetDevice(c->device));
    pce_ctx::StatSlot &st = c->stat[slot];
    const size_t b_en = pce_energy_stage_bytes(c), b_lu = c->lu_n > 0 ? sizeof(double) * (size_t)c->lu_n : 0, b_pi = pce_pitch_stage_bytes(c);
    auto up = [](size_t v) { return (v + 63) & ~(size_t)63; };
    const size_t need = up(b_en) + up(b_lu) + up(b_pi) + 64;
    if (need > st.cap) {
        if (st.host) { PCE_HIP(c, hipHostFree(st.host)); st.host = nullptr; st.cap = 0; }
        PCE_HIP(c, hipHostMalloc(&st.host, need + (need >> 2), hipHostMallocDefault));
        st.cap = need + (need >> 2);
    }
    if (!st.ev) PCE_HIP(c, hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
    char *base = static_cast<char *>(st.host);
    st.off_lu = up(b_en); st.off_pi = st.off_lu + up(b_lu);
    st.en_n = c->en_n; st.lu_n = c->lu_n; st.pi_n = c->pi_n;
    // The LUFS values and the pitch summaries are copied on the side stream that produced them, right behind the producing
    // kernel (enqueued on `stream` at the end of a step the same copies ran beside the STFT normalisation pass, up to 90 us
    // each), and `stream` itself does NOT wait for them: the slot's event is recorded on the last stream to finish (the
    // pitch tail) after it has waited for the other copies, so the next batch's first kernels start while the tail of
    // this one is still running.  The side streams stay `pending`: whoever reuses their buffers joins them as before.
    // The STFT pass (SIDE_STFT) is not involved: the statistics do not depend on it.
    pce_ctx::Side &sl = c->side[pce_ctx::SIDE_LUFS], &stl = c->side[pce_ctx::SIDE_TAIL];
    if (c->en_n >= 0) { int rc = pce_energy_stage_enqueue(c, base, st.en_len); if (rc) return rc; }
    if (c->lu_n > 0) PCE_HIP(c, hipMemcpyAsync(base + st.off_lu, c->lu_out.p, b_lu, hipMemcpyDeviceToHost, sl.pending ? sl.s : c->stream));
    if (sl.pending) PCE_HIP(c, hipEventRecord(sl.join, sl.s));                  // the join now covers the copy
    if (c->lu_n >= 0) st.lu_status = c->lu_host_status;
    if (c->pi_n >= 0) {
        int rc = pce_pitch_stage_enqueue(c, base + st.off_pi, stl.pending ? stl.s : c->stream); if (rc) return rc;
        st.pi_frames.resize((size_t)c->pi_n);
        for (int32_t i = 0; i < c->pi_n; i++) st.pi_frames[(size_t)i] = c->pi_frame_off[(size_t)i + 1] - c->pi_frame_off[(size_t)i];
        st.pi_t1 = c->pi_t1; st.pi_status = c->pi_status;
    }
    if (stl.pending) {
        if (!st.ev_main) PCE_HIP(c, hipEventCreateWithFlags(&st.ev_main, hipEventDisableTiming));
        PCE_HIP(c, hipEventRecord(st.ev_main, c->stream));                     // behind the copies made on `stream`
        PCE_HIP(c, hipStreamWaitEvent(stl.s, st.ev_main, 0));
        if (sl.pending) PCE_HIP(c, hipStreamWaitEvent(stl.s, sl.join, 0));
        PCE_HIP(c, hipEventRecord(st.ev, stl.s));
        PCE_HIP(c, hipEventRecord(stl.join, stl.s));                           // later joins cover the copy and the waits
    } else {
        { int rc = pce_side_join(c, pce_ctx::SIDE_LUFS); if (rc) return rc; }
        PCE_HIP(c, hipEventRecord(st.ev, c->stream));
    }
    st.armed = true;
    return PCE_OK;
}

int pce_stats_wait(pce_ctx *c, int32_t slot, pce_energy *energy, double *lufs, int32_t *lufs_status, pce_pitch_summary *pitch)
{
    if (!c || slot < 0 || slot > 1) return PCE_E_INVALID;
    pce_ctx::StatSlot &st = c->stat[slot];
    if (!st.armed) return pce_fail(c, PCE_E_STATE, "pce_stats_wait: slot %d has no enqueued fetch", slot);
    if ((energy && st.en_n < 0) || ((lufs || lufs_status) && st.lu_n < 0) || (pitch && st.pi_n < 0))
        return pce_fail(c, PCE_E_STATE, "pce_stats_wait: a requested statistic was not computed before pce_stats_enqueue");
    PCE_HIP(c, hipSetDevice(c->device));
    PCE_HIP(c, hipEventSynchronize(st.ev));
    st.armed = false;
    pce_profile_collect(c, false);
    const char *base = static_cast<const char *>(st.host);
    if (energy) pce_energy_stage_unpack(base, st.en_len, energy);
    if (lufs && st.lu_n > 0) memcpy(lufs, base + st.off_lu, sizeof(double) * (size_t)st.lu_n);
    if (lufs_status) for (int32_t i = 0; i < st.lu_n; i++) lufs_status[i] = st.lu_status[(size_t)i];
    if (pitch) {
        pce_pitch_stage_unpack(base + st.off_pi, st.pi_n, pitch);
        for (int32_t i = 0; i < st.pi_n; i++) {
            pitch[i].n_frames = st.pi_frames[(size_t)i]; pitch[i].t1 = st.pi_t1[(size_t)i]; pitch[i].status = st.pi_status[(size_t)i];
            pitch[i].reserved = 0;
        }
    }
    return PCE_OK;
}

int pce_device_info(pce_ctx *c, char *name, size_t namelen, int32_t *cus, int64_t *hbm)
{
    if (!c) return PCE_E_INVALID;
    hipDeviceProp_t prop;
    PCE_HIP(c, hipGetDeviceProperties(&prop, c->device));
    if (name && namelen) snprintf(name, namelen, "%s (%s)", prop.name, prop.gcnArchName);
    if (cus) *cus = prop.multiProcessorCount;
    if (hbm) *hbm = (int64_t)prop.totalGlobalMem;
    return PCE_OK;
}

static int set_offsets(pce_ctx *c, const int64_t *offsets, int32_t n_clips, int32_t rate)
{
    if (!offsets || n_clips <= 0 || rate <= 0) return pce_fail(c, PCE_E_INVALID, "bad batch description");
    if (offsets[0] != 0) return pce_fail(c, PCE_E_INVALID, "offsets[0] must be 0");
    for (int32_t i = 0; i < n_clips; i++)
        if (offsets[i + 1] < offsets[i]) return pce_fail(c, PCE_E_INVALID, "offsets must be non-decreasing");
    c->clip_off.assign(offsets, offsets + n_clips + 1);
    c->n_clips = n_clips; c->rate = rate;
    PCE_HIP(c, c->d_clip_off.reserve(sizeof(int64_t) * (size_t)(n_clips + 1)));
    PCE_HIP(c, hipMemcpyAsync(c->d_clip_off.p, offsets, sizeof(int64_t) * (size_t)(n_clips + 1), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    c->en_n = c->lu_n = c->pi_n = -1; c->st_nfft = 0; c->st_ran = false; c->fr_ran = false; c->py_ran = false;
    c->en_cache.drop(); c->lu_cache.drop(); c->pi_cache.drop();
    return PCE_OK;
}

int pce_upload_pcm_s16(pce_ctx *c, const int16_t *pcm, const int64_t *offsets, int32_t n_clips, int32_t rate)
{
    if (!c || !pcm) return PCE_E_INVALID;
    PCE_HIP(c, hipSetDevice(c->device));
    { int rc = pce_join_aux(c); if (rc) return rc; }           // side-stream kernels may still read the old batch
    int st = set_offsets(c, offsets, n_clips, rate);
    if (st) return st;
    size_t total = (size_t)offsets[n_clips];
    // 64 bytes of zero slack so that 16-byte vector loads may run past the last sample
    PCE_HIP(c, c->pcm_own.reserve(total * 2 + 64));
    PCE_HIP(c, hipMemsetAsync((char *)c->pcm_own.p + total * 2, 0, 64, c->stream));
    PCE_HIP(c, hipMemcpyAsync(c->pcm_own.p, pcm, total * 2, hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    c->d_pcm = c->pcm_own.as<const int16_t>();
    return PCE_OK;
}

int pce_bind_pcm_s16_device(pce_ctx *c, const void *d_pcm, const int64_t *offsets, int32_t n_clips, int32_t rate)
{
    if (!c || !d_pcm) return PCE_E_INVALID;
    if (((uintptr_t)d_pcm) & 15) return pce_fail(c, PCE_E_INVALID, "device PCM pointer must be 16-byte aligned");
    PCE_HIP(c, hipSetDevice(c->device));
    { int rc = pce_join_aux(c); if (rc) return rc; }
    int st = set_offsets(c, offsets, n_clips, rate);
    if (st) return st;
    c->d_pcm = (const int16_t *)d_pcm;
    return PCE_OK;
}

int pce_num_clips(const pce_ctx *c) { return c ? c->n_clips : 0; }

int pce_profile_enable(pce_ctx *c, int on) { if (!c) return PCE_E_INVALID; c->prof = on != 0; return PCE_OK; }
int pce_profile_reset(pce_ctx *c)
{
    if (!c) return PCE_E_INVALID;
    pce_profile_collect(c);
    for (int i = 0; i < PCE_K_COUNT; i++) { c->prof_ms[i] = 0; c->prof_n[i] = 0; c->prof_flops[i] = 0; }
    return PCE_OK;
}
int pce_profile_get(pce_ctx *c, int id, double *ms, int64_t *n)
{
    if (!c || id < 0 || id >= PCE_K_COUNT) return PCE_E_INVALID;
    pce_profile_collect(c);
    if (ms) *ms = c->prof_ms[id];
    if (n) *n = c->prof_n[id];
    return PCE_OK;
}
int pce_profile_get_work(pce_ctx *c, int id, double *flops)
{
    if (!c || id < 0 || id >= PCE_K_COUNT) return PCE_E_INVALID;
    pce_profile_collect(c);
    if (flops) *flops = c->prof_flops[id];
    return PCE_OK;
}
const char *pce_kernel_name(int id)
{
    static const char *names[PCE_K_COUNT] = {
        "k_energy", "k_lufs_pass1", "k_lufs_scan", "k_lufs_pass2", "k_lufs_gate",
        "k_pitch_refine", "k_pitch_frames", "k_pitch_path", "k_pitch_median", "k_pitch_delta",
        "k_stft_max", "k_stft_db", "k_logmel_frames", "whisper_encoder", "k_resample", "k_dtw", "whisper_align", "k_nw", "k_stft_norm", "k_frame_energy", "bert_forward", "k_pyin_frames", "k_pyin_viterbi", "whisper_decode_step",
        "k_gemm_bf16", "k_gemm_wide", "k_attention", "k_layernorm", "k_gemm_flat",
        "k_add_layernorm", "k_stft_raw", "k_logmel_norm", "k_attention_lean",
        "k_gemm_flat:qkv", "k_gemm_flat:out", "k_gemm_flat:fc1", "k_gemm_flat:fc2", "k_gemm_flat:xkv",
        "whisper_decode_loop", "k_cross_attn1", "k_gemm_skinny", "k_levenshtein"};
    return (id >= 0 && id < PCE_K_COUNT) ? names[id] : "?";
}

} // extern "C"
