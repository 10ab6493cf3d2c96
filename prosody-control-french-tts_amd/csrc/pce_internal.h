// pce_internal.h -- shared between the translation units of libpce.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "pce.h"

// A kernel marked PCE_NO_PK_F32 is compiled without packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32).  Why (round 6,
// profiles/r06/multiprocess_glitch.txt, tools/lab/pk_victim.hip): on the MI355X boxes of this pool a packed fp32 instruction whose low result lane
// reads src0's LOW half and src1's HIGH half (op_sel:[0,1]) returns wrong values in lanes 48..63 while another wave on the same SIMD executes MFMA.
// The compiler forms exactly that selection for complex products and for pair sums; tools/isa_guard.py refuses a library that holds one.
#if defined(__HIP_DEVICE_COMPILE__)
#define PCE_NO_PK_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define PCE_NO_PK_F32                                    /* (the host pass of the same source: an x86 target knows no such feature) */
#endif

// Growable device buffer owned by the context.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + (bytes >> 3) + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Last slice list an op was planned for: a repeated *_run with the same slices
// (the steady state of a batch pipeline, and of bench.py) skips the host plan.
struct SliceCache {
    std::vector<pce_slice> v;
    bool valid = false;
    bool same(const pce_slice *s, int32_t n) const {
        return valid && (size_t)n == v.size() && (n == 0 || memcmp(v.data(), s, sizeof(pce_slice) * (size_t)n) == 0);
    }
    void store(const pce_slice *s, int32_t n) { v.assign(s, s + n); valid = true; }
    void drop() { valid = false; v.clear(); }
};

struct pce_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // side streams: work that leaves most of the machine idle (or is independent of what the caller launches next) is
    // forked from `stream` onto one of these and joined back by whoever consumes its results:
    //   SIDE_TAIL  path finder + median of the pitch analysis (latency bound, ~0.4 ms with few CUs busy)
    //   SIDE_LUFS  the LUFS chain (sequential IIR per lane: few waves, long dependent chains)
    //   SIDE_STFT  the HBM-bound normalisation pass of the STFT-dB (the FFT pass itself stays on `stream`: launched
    //              beside the pitch kernels it only competed for VALU issue, 3.33 -> 3.54 ms)
    enum { SIDE_TAIL = 0, SIDE_LUFS = 1, SIDE_STFT = 2, SIDE_COUNT = 3 };
    struct Side { hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; bool pending = false; } side[SIDE_COUNT];
    bool no_side = false;                // PCE_NO_AUX at pce_create: everything on `stream`
    bool generic_median = false;         // PCE_ALIGN_GENERIC_MEDIAN at pce_create: the insertion-sort median filter for every width
    bool pitch_refine_praat = false;     // PCE_PITCH_REFINE=praat at pce_create: the candidate refinement replays NUMminimize_brent's own iterates (round 1 / 2 behaviour)
    // dynamic-LDS opt-ins (hipFuncSetAttribute) done on this context's device, per operand-type build: the implementation file is compiled
    // twice, so every kernel below exists as two distinct functions
    bool attn1w_attr[2] = {false, false};
    unsigned xattn_attr[2] = {0u, 0u};    // k_xattn_absorbed<d, slots>: bit per instantiation whose dynamic LDS size has been set (per operand build)
    bool gemm_flat_attr[2][4] = {{false, false, false, false}, {false, false, false, false}};   // k_gemm_flat<EPI>
    bool gemm_few_rows = false;          // set by the incremental decoding step around its launches: k_gemm_skinny is eligible
    bool gemm_skinny = true, gemm_skinny_attr[2][4] = {{false, false, false, false}, {false, false, false, false}};   // PCE_GEMM_SKINNY=0 at pce_create: few-row launches stay on the 128 x 128 kernel
    bool gemm_flat = true;               // PCE_GEMM_FLAT=0 at pce_create: the encoder's projections stay on the 128 x 128 / 128 x 256 tile kernels
    bool stft_two_fft = false;           // PCE_STFT_TWO_FFT at pce_create: traffic-minimal STFT-dB (the FFT runs twice)
    std::string err;
    int cu_count = 0;

    // resident batch
    DevBuf pcm_own;                 // used by pce_upload_pcm_s16
    const int16_t *d_pcm = nullptr; // device pointer to the concatenated clips
    DevBuf d_clip_off;              // int64[n_clips+1]
    std::vector<int64_t> clip_off;  // host copy
    int32_t n_clips = 0;
    int32_t rate = 0;

    // energy
    int en_cpb = 0;                 // PCE_EN_CPB: chunks per k_energy workgroup (0 = by batch size)
    DevBuf en_work, en_out;
    SliceCache en_cache;
    int64_t en_n_work = 0;
    int32_t en_n = -1;

    // lufs
    DevBuf lu_meta, lu_chunks, lu_blocks, lu_pow, lu_state_end, lu_state_init, lu_energy, lu_zbuf, lu_out, lu_en_work, lu_en_acc;
    SliceCache lu_cache;
    int64_t lu_n_chunks = 0, lu_n_blocks = 0, lu_n_energy_work = 0;
    double lu_coef[13] = {0};
    int32_t lu_n = -1;
    int32_t lu_meter_rate = 0;       // pce_lufs_set_meter_rate: 0 = the batch's own rate
    bool lu_reads_en_out = false;    // the running LUFS chain takes its peaks from pce_energy_run's accumulators (same slices)
    std::vector<int32_t> lu_host_status;

    // pitch
    DevBuf pi_meta, pi_window, pi_windowR, pi_work, pi_cand, pi_gpeak, pi_psi, pi_f0, pi_strength, pi_summary, pi_peakwork, pi_acc, pi_rr, pi_items, pi_tw, pi_dl, pi_runs, pi_fslice, pi_blob;
    double pi_P[32] = {0};          // PiParams image
    int64_t pi_n_work = 0, pi_n_energy_work = 0;
    int pi_np2 = 1;
    bool xattn_absorb = true;       // incremental decoding steps: cross-attention from the encoder output (pce_xattn.inc); PCE_XATTN_ABSORB=0: from the K / V^T cache
    bool self_rows = true;          // incremental steps' self-attention on row-major K / V caches (k_self_attn1w); PCE_SELF_ROWS=0: k_cross_attn1w on K rows + V^T
    bool attn_nt = true;            // single-query-block attention launches stream K / V^T with the non-temporal policy (PCE_ATTN_NT=0: default policy)
    bool attn_m16 = false;          // attention on v_mfma_f32_16x16x32 (k_attention_lean16) instead of 32x32x16
    bool pi_long_slices = false;    // some slice has more frames than the in-LDS median sort holds (k_pitch_median_long takes those)
    SliceCache pi_cache;
    pce_pitch_params pi_params;
    bool pi_params_valid = false;
    int32_t pi_n = -1;
    int64_t pi_total_frames = 0;
    std::vector<int64_t> pi_frame_off;
    std::vector<int32_t> pi_status;
    std::vector<double> pi_t1;

    // stft
    DevBuf st_out, st_max, st_off, st_window, st_twiddle, st_work, st_stage;   // st_stage: one clip's finished values on their way to the host (pce_stft_db_fetch)
    int32_t st_nfft = 0, st_hop = 0;
    int64_t st_n_tiles = 0;
    bool st_ran = false;
    bool st_final = false;            // st_out holds finished values (ref = max applied): after the two-FFT form or pce_stft_db_device; raw dB + st_max otherwise
    std::vector<int64_t> st_off_host;   // float offsets per clip (n_clips+1)
    std::vector<int32_t> st_frames;

    // frame energy (analysis windows of the energy VAD)
    DevBuf fr_doff, fr_sum, fr_cnt;
    std::vector<int64_t> fr_off;        // frames before clip i (n_clips+1)
    bool fr_ran = false;

    // probabilistic YIN
    DevBuf py_doff, py_tab, py_hdr, py_bin, py_lp, py_ptr, py_states;
    std::vector<int64_t> py_off;
    bool py_ran = false;

    // whisper / BERT state, opaque (pce_whisper_impl.inc): one slot per operand-type build (0: bf16, 1: fp16); whisper_ops selects the build the
    // entry points of include/pce.h forward to (pce_whisper_set_operands, or PCE_WHISPER_OPERANDS=fp16 at pce_create)
    void *whisper_slot[2] = {nullptr, nullptr};
    int whisper_ops = 2;                 // (pce_create: 2 = the fp16 build + resid16 unless PCE_WHISPER_OPERANDS=bf16 / fp16)
    bool resid16 = true;                 // the batched encoder path keeps its residual stream in 16 bits (fp16 build only)

    // asynchronous statistics fetch (pce_stats_enqueue / pce_stats_wait)
    struct StatSlot {
        void *host = nullptr; size_t cap = 0; hipEvent_t ev = nullptr, ev_main = nullptr; bool armed = false;
        int32_t en_n = -1, lu_n = -1, pi_n = -1;
        size_t off_lu = 0, off_pi = 0;
        std::vector<int64_t> en_len, pi_frames;
        std::vector<int32_t> lu_status, pi_status;
        std::vector<double> pi_t1;
    } stat[2];

    // profiling
    bool prof = false;
    double prof_ms[PCE_K_COUNT] = {0};
    int64_t prof_n[PCE_K_COUNT] = {0};
    double prof_flops[PCE_K_COUNT] = {0};
    struct Pending { int id; hipEvent_t a, b; double flops; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> ev_pool;
};

int pce_fail(pce_ctx *ctx, int code, const char *fmt, ...);
#define PCE_HIP(ctx, call)                                                                      \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return pce_fail((ctx), PCE_E_DEVICE, "%s: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

// Kernel-launch bracket for the profiler: records events around the launch when enabled.
struct KernelTimer {
    pce_ctx *c; int id; hipEvent_t a = nullptr, b = nullptr; hipStream_t s; double flops;
    KernelTimer(pce_ctx *ctx, int kid, hipStream_t on = nullptr, double work_flops = 0.0);
    ~KernelTimer();
};
void pce_profile_collect(pce_ctx *ctx, bool wait = true);
int pce_join_aux(pce_ctx *c);                                // make `stream` wait for ALL pending side-stream work
int pce_side_join(pce_ctx *c, int which);                    // ... for one side stream
int pce_side_begin(pce_ctx *c, int which, hipStream_t *out); // fork: *out = side stream ordered behind `stream` (or `stream` itself: PCE_NO_AUX)
int pce_side_end(pce_ctx *c, int which, hipStream_t used);   // record the join point   // wait = false: only the launches that have completed

// staged fetch helpers of the modules (pce_stats_*): bytes needed, enqueue the copy into pinned memory, unpack it
size_t pce_energy_stage_bytes(const pce_ctx *c);
int pce_energy_stage_enqueue(pce_ctx *c, void *pinned, std::vector<int64_t> &lens);
void pce_energy_stage_unpack(const void *pinned, const std::vector<int64_t> &lens, pce_energy *out);
size_t pce_pitch_stage_bytes(const pce_ctx *c);
int pce_pitch_stage_enqueue(pce_ctx *c, void *pinned, hipStream_t on = nullptr);
void pce_pitch_stage_unpack(const void *pinned, int32_t n, pce_pitch_summary *out);

// Host-side copy of the sizes Praat derives before its frame loop (see pce_pitch.hip).
struct PitchPlan {
    double dt, t1, ceiling, dt_window;
    int64_t n_frames, nsamp_period, halfnsamp_period, nsamp_window, halfnsamp_window;
    int64_t maximum_lag, brent_ixmax, max_candidates;
};
int pitch_plan_make(int64_t nx, double dx, double x1, const pce_pitch_params *p, PitchPlan *pl);

void pce_whisper_free(pce_ctx *c);
static inline int64_t div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }
