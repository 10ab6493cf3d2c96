/*
 * pce.h -- C ABI of the MI355X prosody-extraction engine ("pce").
 *
 * Drop-in boundary for the prosody + alignment hot path of
 * hi-paris/Prosody-Control-French-TTS.  The reference has no FFI of its own
 * (its boundary is Python call signatures); every entry
 * point below names the reference call it replaces.  Plain C: pointers and
 * sizes only, no C++/torch types.  The library (libpce.so) is built by hipcc
 * for gfx950 only and has NO CPU fallback: pce_create fails when no GPU is
 * present.
 *
 * Conventions
 *   - return value: 0 (PCE_OK) or a negative pce_status; text via pce_last_error.
 *   - one context per process per GPU, not thread-safe, one call in flight.
 *   - the caller owns every host buffer; the library owns device memory only.
 *   - *_run calls only enqueue kernels on the context's HIP stream; *_fetch
 *     calls synchronise that stream and copy results to host buffers.
 *   - audio is 16-bit PCM, mono; clips are concatenated, clip i occupying
 *     samples [offsets[i], offsets[i+1]) of the buffer.
 *   - a slice addresses samples [begin, end) of one clip in clip coordinates;
 *     indices outside [0, clip length) are "virtual" samples equal to zero,
 *     which is what both Praat's Sound_extractPart and pydub's silence padding
 *     produce.  Converting (t0, t1) seconds or ms to sample indices is host
 *     logic (pydub / Praat rules) and lives in the Python shim.
 */
#ifndef PCE_H
#define PCE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libpce.so is built with -fvisibility=hidden: the functions declared between this pragma and its pop are the library's ONLY exported
 * symbols (tests/test_abi_and_shard.py compares `nm -D` with this header, both ways). */
#pragma GCC visibility push(default)

/* PCE_API_VERSION changes when an existing entry point changes its signature or is removed; PCE_API_MINOR counts additive changes and
 * changes of observable defaults:  1 = round 4's default operand mode (fp16 operands + fp16 residual stream), decode loop without the
 * n_text_ctx pre-check;  2 = round 5: pce_levenshtein, pce_whisper_align_paths_enqueue / _wait, no frame limit in pce_pitch_run, the 256 x 256
 * GEMM for every batch size (a clip's Whisper results no longer depend on what it is batched with), unknown PCE_WHISPER_OPERANDS rejected,
 * pce_whisper_sample_keys (temperature sampling keyed by the caller's clip ids instead of batch positions);  3 = round 6: the encoder-output
 * cross-attention of a decoding step always cuts a clip's frames into the same four ranges (minor 2's promise now holds for every batch size: the
 * round-5 kernel cut them by the number of workgroups per clip, which followed the batch size), pce_whisper_align_paths_enqueue on a slot that
 * still holds an un-waited fetch is PCE_E_STATE, and no kernel carries the packed fp32 operand selection that returns wrong lanes beside MFMA waves
 * of other streams / contexts / processes (results no longer depend on what else runs on the device: tools/isa_guard.py); pce_selftest_xattn. */
#define PCE_API_VERSION 1
#define PCE_API_MINOR 3

typedef struct pce_ctx pce_ctx;

enum pce_status {
    PCE_OK = 0,
    PCE_E_INVALID = -1,   /* bad argument                                  */
    PCE_E_DEVICE = -2,    /* HIP runtime error (text in pce_last_error)    */
    PCE_E_NOMEM = -3,     /* host or device allocation failed              */
    PCE_E_STATE = -4,     /* call order violated (fetch before run, ...)   */
    PCE_E_LIMIT = -5      /* input exceeds a documented engine limit       */
};

/* per-slice status codes written to int32 status arrays */
enum pce_slice_status {
    PCE_SLICE_OK = 0,
    PCE_SLICE_TOO_SHORT = 1,  /* Praat would throw / pyloudnorm raises ValueError */
    PCE_SLICE_EMPTY = 2       /* zero samples                                      */
};

typedef struct pce_slice {
    int32_t clip;     /* index into the uploaded batch                               */
    int32_t flags;    /* reserved, 0                                                 */
    int64_t begin;    /* first sample (clip coordinates, may be < 0)                 */
    int64_t end;      /* one past the last sample (may exceed the clip length)       */
    double  x1;       /* time of sample `begin` in seconds (Praat Sound.x1); only
                         the pitch analysis reads it                                 */
} pce_slice;

/* ---- context ---------------------------------------------------------- */

/* stream: a hipStream_t to enqueue on (e.g. torch.cuda.current_stream().cuda_stream)
 * or NULL for a private stream.  Returns NULL on failure with text in err. */
pce_ctx *pce_create(int device, void *stream, char *err, size_t errlen);
void pce_destroy(pce_ctx *ctx);
const char *pce_last_error(const pce_ctx *ctx);
int pce_sync(pce_ctx *ctx);
int pce_api_version(void);
int pce_api_minor(void);
int pce_device_info(pce_ctx *ctx, char *name, size_t namelen, int32_t *compute_units, int64_t *hbm_bytes);

/* ---- batch residency ---------------------------------------------------
 * Replaces the per-call full-file decode of the reference closures
 * (AudioSegment.from_file / parselmouth.Sound(path) in
 * Code/audioPipeline.py:319,327,340,361): a batch is decoded once by the host,
 * uploaded once and stays resident in HBM for every measurement. */
int pce_upload_pcm_s16(pce_ctx *ctx, const int16_t *pcm, const int64_t *offsets, int32_t n_clips, int32_t sample_rate);
/* zero-copy variant: d_pcm is a device pointer that stays valid until the next
 * upload/bind/destroy; it must be 16-byte aligned and followed by >= 16 readable bytes. */
int pce_bind_pcm_s16_device(pce_ctx *ctx, const void *d_pcm, const int64_t *offsets, int32_t n_clips, int32_t sample_rate);
int pce_num_clips(const pce_ctx *ctx);

/* ---- sample-rate conversion of the resident batch -------
 * Stands in for the ffmpeg decode-to-16-kHz inside whisper.load_audio
 * (Code/Aligners/use_whisper_timestamped.py:139): y = upfirdn(taps, x, up, down)[n_pre_remove : n_pre_remove + n_out],
 * n_out = ceil(n_in * up / down), fp64 accumulation, round-half-even to int16.  The low-pass
 * `taps` (already scaled by `up`, zero padded as scipy.signal.resample_poly does) is host logic
 * (hostrules.resample_filter).  The resampled batch REPLACES the resident batch. */
int pce_resample_run(pce_ctx *ctx, int32_t up, int32_t down, const double *taps, int32_t n_taps, int64_t n_pre_remove);
/* copy the resident batch back: pcm may be NULL to query offsets[n_clips+1] / sample_rate only */
int pce_download_pcm_s16(pce_ctx *ctx, int16_t *pcm, int64_t *offsets, int32_t *sample_rate);

/* ---- frame-level short-time energy (the energy detector of the aligner's VAD) ----
 * Replaces the per-window energy of auditok.split(energy_threshold=50), which whisper-timestamped runs because the
 * aligner asks for `"vad": "auditok"` (Code/Aligners/use_whisper_timestamped.py:152; packages absent from
 * /root/reference: restated from their published behaviour, parity unpinned).  Frame k of a clip of n samples covers
 * [k*hop, min(k*hop + window, n)), k = 0 .. ceil(n/hop)-1: with hop == window these are auditok's analysis windows
 * (0.05 s; the last one short).  sum_sq[k] = exact integer sum of squares, count[k] = samples in the frame; the
 * mean / sqrt / 20 log10 / threshold are host logic (Aligners/vad.py).  requantize != 0 first applies, per sample,
 * whisper-timestamped's float32 round trip: (int16)((float32)(x / 32768) * 32767), truncated toward zero. */
int pce_frame_energy_run(pce_ctx *ctx, int32_t window, int32_t hop, int32_t requantize);
int pce_frame_energy_shape(pce_ctx *ctx, int32_t clip, int64_t *n_frames);
int pce_frame_energy_fetch(pce_ctx *ctx, int32_t clip, int64_t *sum_sq /* [n_frames] or NULL */, int32_t *count /* [n_frames] or NULL */);

/* ---- R3 / R7: short-time energy, peak, silence gate --------------------
 * Replaces _calculate_loudness (Code/Pipeline/compute_loudness_adjustments.py:8-25),
 * _check_audio_content (Code/Aligners/use_whisper_timestamped.py:197-229 and its
 * inline copy :581-599) and the np.abs(samples).max() of get_lufs
 * (Code/audioPipeline.py:349).  All fields are exact integers; the few
 * floating-point finishing operations (sqrt/log10) are host logic. */
typedef struct pce_energy {
    int64_t n;              /* samples in the slice, virtual zeros included           */
    int64_t sum_sq;         /* sum of x*x (true integer squares)                      */
    int64_t sum_sq_wrap16;  /* sum of (int16)(x*x): numpy int16 ** 2 wraparound (R3)  */
    int64_t n_loud;         /* count of (int16)abs(x) > loud_threshold  (R7; abs(-32768) wraps to -32768) */
    int32_t peak_abs;       /* max |x| as a true integer (0..32768)                   */
    int32_t reserved;
} pce_energy;
int pce_energy_run(pce_ctx *ctx, const pce_slice *slices, int32_t n_slices, int32_t loud_threshold);
int pce_energy_fetch(pce_ctx *ctx, pce_energy *out /* n_slices */);

/* ---- R4: BS.1770 integrated loudness -----------------------------------
 * Replaces pyloudnorm.Meter(rate).integrated_loudness(samples / peak) as called
 * by get_lufs (Code/audioPipeline.py:338-358): peak normalisation, K-weighting
 * biquads designed for the batch's sample rate, 400 ms / 75 % blocks, -70 LKFS and
 * -10 LU gates.  status[i] = PCE_SLICE_TOO_SHORT where pyloudnorm raises
 * ValueError (n < 0.4 * rate); the whole-file fallback is the caller's.
 * pce_lufs_set_meter_rate: the reference builds ONE meter from the natural recording's frame rate
 * (Code/audioPipeline.py:372,493: pyln.Meter(AudioSegment.from_file(wav).frame_rate)) and measures the raw synthesis with
 * it as well, whatever that file's own rate is (Azure's default RIFF output is 16 kHz, the recordings are 44.1 kHz):
 * filter design, the 0.4 s block length in samples and the too-short test all use the METER's rate.  rate > 0 makes
 * the following pce_lufs_run calls do the same; 0 (default) = the batch's rate. */
int pce_lufs_set_meter_rate(pce_ctx *ctx, int32_t rate);
int pce_lufs_run(pce_ctx *ctx, const pce_slice *slices, int32_t n_slices);
int pce_lufs_fetch(pce_ctx *ctx, double *lufs /* n_slices */, int32_t *status /* n_slices */);

/* ---- R1 / R2: Praat autocorrelation pitch ------------------------------
 * Replaces parselmouth Sound.to_pitch(pitch_floor, pitch_ceiling) followed by
 * selected_array["frequency"] and the voiced median (get_median_pitch,
 * Code/audioPipeline.py:326-335) or the voiced geometric mean
 * (calculate_pitch_segment, Code/Pipeline/compute_pitch_adjustments.py:167-208). */
typedef struct pce_pitch_params {
    double  time_step;            /* <= 0: 0.75 / pitch_floor                    */
    double  pitch_floor;
    double  periods_per_window;   /* 3.0                                         */
    int32_t max_candidates;       /* 15                                          */
    int32_t reserved;
    double  silence_threshold;    /* 0.03 */
    double  voicing_threshold;    /* 0.45 */
    double  octave_cost;          /* 0.01 */
    double  octave_jump_cost;     /* 0.35 */
    double  voiced_unvoiced_cost; /* 0.14 */
    double  pitch_ceiling;
} pce_pitch_params;

typedef struct pce_pitch_summary {
    int64_t n_frames;     /* 0 when status != PCE_SLICE_OK                             */
    int64_t n_voiced;     /* frames with f0 > 0                                        */
    double  median_f0;    /* np.median of voiced f0, 0.0 when none                     */
    double  mean_log_f0;  /* mean of ln(f0) over voiced frames (exp -> geometric mean) */
    double  t1;           /* centre time of the first frame                            */
    int32_t status;       /* pce_slice_status                                          */
    int32_t reserved;
} pce_pitch_summary;

/* Host-only sizing: frame_offsets[n_slices+1] (prefix sum of frame counts). */
int pce_pitch_plan(pce_ctx *ctx, const pce_pitch_params *p, const pce_slice *slices, int32_t n_slices,
                   int64_t *frame_offsets, int32_t *status);
int pce_pitch_run(pce_ctx *ctx, const pce_pitch_params *p, const pce_slice *slices, int32_t n_slices);
/* How k_pitch_refine searches a candidate's maximum (Praat: NUMimproveMaximum -> NUMminimize_brent on the sinc-interpolated
 * autocorrelation).  PCE_REFINE_SEEDED (default): successive parabolic interpolation seeded with the three samples around the peak,
 * Praat's own iterates only where the two could differ (within 0.03 lag of a sample, or when a safeguard trips): candidates agree with
 * Praat's to ~1e-6 relative.  PCE_REFINE_PRAAT: NUMminimize_brent's iterates replayed for every candidate (iterate for iterate; about
 * 2.5 x the refinement time).  The environment variable PCE_PITCH_REFINE=praat at pce_create selects the second as the context's default. */
enum { PCE_REFINE_SEEDED = 0, PCE_REFINE_PRAAT = 1 };
int pce_pitch_set_refine(pce_ctx *ctx, int32_t mode);
/* f0 / strength: ragged [frame_offsets[n_slices]], either may be NULL. */
int pce_pitch_fetch(pce_ctx *ctx, double *f0, double *strength, pce_pitch_summary *summary /* n_slices */);

/* ---- R10: STFT magnitude in dB -----------------------------------------
 * Replaces librosa.amplitude_to_db(np.abs(librosa.stft(y, n_fft, hop_length)),
 * ref=np.max) (Code/visualisation/app.py:69-72, acoustic_analysis.py:76-90):
 * periodic Hann, centred frames with zero padding, float32, amin 1e-5, top_db 80.
 * One [n_fft/2+1, 1+n/hop] row-major float32 matrix per uploaded clip. */
int pce_stft_db_run(pce_ctx *ctx, int32_t n_fft, int32_t hop);
int pce_stft_db_shape(pce_ctx *ctx, int32_t clip, int32_t *n_bins, int32_t *n_frames);
int pce_stft_db_fetch(pce_ctx *ctx, int32_t clip, float *out);
/* device pointer + byte size of the resident result (all clips, concatenated) */
int pce_stft_db_device(pce_ctx *ctx, const void **d_ptr, int64_t *bytes);

/* ---- R10: probabilistic YIN -------------------------------------------------
 * Replaces librosa.pyin(audio, sr=sr, fmin=60, fmax=2000, hop_length=256) of the reference's viewers
 * (Code/visualisation/app.py:74-78, acoustic_analysis.py:76-94, visualisation_abtest/app.py:108-111): frame_length 2048,
 * win_length 1024, centred zero-padded frames, 100 thresholds with the beta(2, 18) prior, Boltzmann(2) trough prior,
 * 0.1-semitone pitch bins, triangular local transitions, switch probability 0.01, dense Viterbi (first maximum).
 * librosa is third party and absent: restated from its published implementation, parity unpinned.  The plan (periods,
 * bin counts, log constants) and the constant tables are host logic (visualisation/acoustic_analysis.py builds them
 * with numpy); the difference function is exact integer arithmetic on the int16 samples.  One decoded state per frame
 * (state < n_pitch_bins: voiced, f0 = fmin 2^(state / bins_per_octave)) and the voiced probability. */
typedef struct pce_pyin_plan {
    int32_t frame_length, hop_length, min_period, max_period, n_pitch_bins, trans_width, n_thresholds, reserved;
    double sr, fmin, bins_per_octave, no_trough_prob, log_tiny, log_p_init, tiny;
} pce_pyin_plan;
int pce_pyin_run(pce_ctx *ctx, const pce_pyin_plan *plan, const double *tables, int64_t n_tables);
int pce_pyin_shape(pce_ctx *ctx, int32_t clip, int64_t *n_frames);
/* states [n_frames] / voiced_prob [n_frames] / status (bit 0: a frame had more troughs than the engine keeps) may be NULL */
int pce_pyin_fetch(pce_ctx *ctx, int32_t clip, int32_t *states, double *voiced_prob, int32_t *status);

/* ---- R8: log-mel spectrogram + Whisper audio encoder --------------------
 * Replaces the device work of whisper_timestamped.transcribe before decoding
 * (Code/Aligners/use_whisper_timestamped.py:139,150-163; openai-whisper==20240930):
 * whisper.log_mel_spectrogram on the 30 s window starting at sample 0 of every
 * uploaded clip (16 kHz; shorter clips are zero padded, as whisper.pad_or_trim
 * does), then AudioEncoder.forward.  Matmuls run on MFMA in the context's
 * operand mode (pce_whisper_set_operands; default: fp16 operands and an fp16
 * residual stream, openai-whisper's own fp16 arithmetic) with fp32 accumulation
 * and fp32 LayerNorm / softmax statistics.
 *
 * Weight blob (float32, this order; Linear/Conv weights in PyTorch layout):
 *   conv1.weight[d][n_mels][3] conv1.bias[d] conv2.weight[d][d][3] conv2.bias[d]
 *   per layer: attn_ln.w[d] attn_ln.b[d] query.w[d][d] query.b[d] key.w[d][d]
 *              value.w[d][d] value.b[d] out.w[d][d] out.b[d] mlp_ln.w[d] mlp_ln.b[d]
 *              mlp.0.w[4d][d] mlp.0.b[4d] mlp.2.w[d][4d] mlp.2.b[d]
 *   ln_post.w[d] ln_post.b[d]
 * The sinusoidal positional embedding is generated inside the library. */
typedef struct pce_whisper_dims {
    int32_t n_mels;    /* 80 (128 for large-v3)          */
    int32_t n_ctx;     /* 1500                            */
    int32_t n_state;   /* d: 384/512/768/1024/1280        */
    int32_t n_head;    /* d / 64                          */
    int32_t n_layer;
} pce_whisper_dims;
int pce_logmel_run(pce_ctx *ctx, int32_t n_mels);
/* window variant (whisper.transcribe slices the log-mel of the WHOLE recording, followed by 30 s of zeros, at its seek
 * position): frames start_frames[clip] .. +2999 of that spectrogram, samples beyond 30 s included, the max - 8 clamp from
 * the maximum over the whole recording.  start_frames[clip] * 160 must not exceed the clip length. */
int pce_logmel_run_at(pce_ctx *ctx, int32_t n_mels, const int64_t *start_frames /* [clips] */);
int pce_logmel_fetch(pce_ctx *ctx, int32_t clip, float *out /* [n_mels][3000] */);
int pce_whisper_load(pce_ctx *ctx, const pce_whisper_dims *dims, const float *weights, int64_t n_floats);
int pce_whisper_encode_run(pce_ctx *ctx);
/* Self-test of the GEMM kernel the encoder's projections run on (persistent 256 x 256 tiles, csrc/pce_gemm256.inc) on host arrays of bf16 bit
 * patterns: C = epilogue(A[M][K] B[N][K]^T + bias).  epilogue 0: bias; 1: bias + exact GELU; 2: bias, written transposed per clip
 * (rows_per_clip rows each, key axis padded to vt_sp): out[(clip N + n) vt_sp + t].  N % 256 == 0, K % 64 == 0. */
int pce_selftest_gemm(pce_ctx *ctx, const uint16_t *A, const uint16_t *B, const float *bias, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                      int32_t rows_per_clip, int32_t vt_sp, uint16_t *out);
/* Self-test of the attention kernel (64-wide heads; csrc/pce_whisper.hip k_attention_lean) on host arrays of bf16 bit patterns: clips x
 * heads independent problems, q [clips][q_len][heads * 64], k and v [clips][k_len][heads * 64], out like q; softmax(q k^T / 8) v with the
 * causal mask when causal != 0.  mode 0: the kernel as the engine runs it (fixed softmax reference, exact fallback), 1: its exact
 * path only.  *fell_back (may be NULL): number of workgroups that had to take the exact path (mode 0). */
int pce_selftest_attention(pce_ctx *ctx, const uint16_t *q, const uint16_t *k, const uint16_t *v, int32_t clips, int32_t heads, int32_t q_len,
                           int32_t k_len, int32_t causal, int32_t mode, uint16_t *out, int32_t *fell_back);
/* Self-test of the cross-attention of an incremental decoding step from the ENCODER OUTPUT (csrc/pce_xattn.inc: LayerNorm + query projection + Q' = q Wk,
 * one streaming pass over E with an online softmax per leaf of frames, the merge tree, out = Wv U + bv) for ONE layer on host arrays: resid [n][d] fp32,
 * ln_w / ln_b / bq / bv [d] fp32, wq / wk / wv [d][d] and E [n][k_cap][d] as 16-bit patterns of the context's operand type (rows of the weights = output
 * features; Whisper's key projection has no bias), k_len[n] valid frames per clip.  d in {128, 256, 384, 512, 768, 1024}, heads = d / 64.
 * workgroups_per_clip: 0 = what the batch size selects, or 1 / 2 / 4 -- the result must not depend on it (the frames are always cut into the same four
 * leaves and merged in the same tree).  out [n][d]: 16-bit patterns. */
int pce_selftest_xattn(pce_ctx *ctx, const float *resid, const float *ln_w, const float *ln_b, const uint16_t *wq, const float *bq, const uint16_t *wk,
                       const uint16_t *wv, const float *bv, const uint16_t *E, const int32_t *k_len, int32_t n, int32_t k_cap, int32_t d, int32_t heads,
                       int32_t workgroups_per_clip, uint16_t *out);
int pce_whisper_encode_fetch(pce_ctx *ctx, int32_t clip, float *out /* [1500][n_state] */);

/* ---- R8: forced alignment of known text tokens (teacher-forced decoder + cross-attention DTW) ----
 * openai-whisper timing.py find_alignment, the mechanism whisper_timestamped's word timestamps rest on
 * (Code/Aligners/use_whisper_timestamped.py:163): TextDecoder.forward over the given token sequence
 * (sot sequence, no-timestamps, text tokens, eot), cross-attention logits of the alignment heads ->
 * softmax over the first num_frames/2 audio positions -> std/mean normalisation over tokens -> median filter
 * over time -> mean over heads -> DTW of -matrix[sot_len:-1].  Token ids -> words is tokenizer (host) logic.
 * Needs pce_whisper_encode_run on the same batch.  Decoder weight blob (float32, PyTorch layouts):
 *   token_embedding[n_vocab][d] positional_embedding[n_text_ctx][d]
 *   per layer: attn_ln.w,b  attn.{query.w,query.b,key.w,value.w,value.b,out.w,out.b}
 *              cross_attn_ln.w,b  cross_attn.{query.w,query.b,key.w,value.w,value.b,out.w,out.b}
 *              mlp_ln.w,b  mlp.0.w,b  mlp.2.w,b
 *   ln.w,b */
typedef struct pce_whisper_text_dims {
    int32_t n_vocab /* <= 52224 */, n_text_ctx /* <= 448 */, n_state, n_head, n_layer;
} pce_whisper_text_dims;
int pce_whisper_decoder_load(pce_ctx *ctx, const pce_whisper_text_dims *dims, const float *weights, int64_t n_floats);
/* tokens: concatenated per clip (token_offsets[n_clips+1]); num_frames: mel frames of real audio per clip;
 * head_mask: [n_layer * n_head] bytes or NULL (all heads of the last half of the layers, whisper's default) */
int pce_whisper_align_run(pce_ctx *ctx, const int32_t *tokens, const int32_t *token_offsets, const int32_t *num_frames,
                          int32_t sot_len, const uint8_t *head_mask, int32_t medfilt_width, float qk_scale);
int pce_whisper_align_shape(pce_ctx *ctx, int32_t clip, int32_t *n_rows, int32_t *n_cols);
/* path arrays hold up to n_rows + n_cols entries; cost (nullable) is the [n_rows][n_cols] fp64 DTW input */
int pce_whisper_align_fetch(pce_ctx *ctx, int32_t clip, int32_t *text_idx, int32_t *time_idx, int32_t *path_len, double *cost);
/* Every clip's path of the last pce_whisper_align_run at once, asynchronously (the form a batch pipeline uses: what
 * whisper_timestamped hands back per segment, Code/Aligners/use_whisper_timestamped.py:163, for all utterances of the batch).
 * _enqueue queues the device-to-host copies behind the alignment on the context's stream into pinned staging memory and returns at
 * once (*n_clips, *path_stride = max rows + max columns of the batch: the row pitch of the index arrays); _wait blocks on those
 * copies only and writes path_len[n_clips] and the first path_len[i] entries of text_idx / time_idx [n_clips][path_stride] (either may
 * be NULL).  slot is 0 or 1 (two batches in flight); _enqueue on a slot whose previous fetch has not been waited for is PCE_E_STATE (its
 * copies may still be landing in the staging buffer).  The same indices as pce_whisper_align_fetch clip by clip. */
int pce_whisper_align_paths_enqueue(pce_ctx *ctx, int32_t slot, int32_t *n_clips, int32_t *path_stride);
int pce_whisper_align_paths_wait(pce_ctx *ctx, int32_t slot, int32_t *path_len, int32_t *text_idx, int32_t *time_idx);

/* ---- R8: free-running decoding, one step -------------------------------------
 * openai-whisper decoding.py at temperature 0 (a default DecodingTask with a GreedyDecoder, what whisper_timestamped's
 * transcribe runs first: Code/Aligners/use_whisper_timestamped.py:150-163): the text decoder over the sequences so far,
 * the logits of the last position through the tied output projection, the logit filters SuppressBlank / SuppressTokens /
 * ApplyTimestampRules and the arg-max (first maximum; a sequence whose last token is end-of-text stays there).
 * vocab_mask[n_vocab]: bit 0 = always suppressed (the suppress list and <|notimestamps|>), bit 1 = suppressed at the
 * first sampled position (the blank token and end-of-text).  The prompt (<|startoftranscript|><|fr|><|transcribe|>),
 * the loop and the stopping rule are host logic (Aligners/decoding.py); token ids in, token ids out (text needs the
 * checkpoint's vocabulary).  The cross-attention K / V of all layers are computed at the first step after
 * pce_whisper_encode_run and kept. */
typedef struct pce_whisper_decode_rules { int32_t eot, timestamp_begin, max_initial_timestamp_index /* < 0: none */, reserved; } pce_whisper_decode_rules;
int pce_whisper_decode_step(pce_ctx *ctx, const int32_t *tokens, const int32_t *token_offsets /* [clips + 1] */, int32_t sample_begin,
                            const pce_whisper_decode_rules *rules, const uint8_t *vocab_mask, int32_t *next_tokens /* [clips] */,
                            float *next_logprobs /* [clips] or NULL: log-probability of the choice under the filtered distribution (sum_logprobs) */);
/* The same step with the per-sequence controls whisper.transcribe needs around it (transcribe.py decode_with_fallback,
 * decoding.py DecodingTask): every sequence has its own prompt length (condition_on_previous_text prepends
 * <|startofprev|> + the previous windows' text), a temperature > 0 draws ONE sample from softmax(filtered logits /
 * temperature) as GreedyDecoder does (Gumbel-max over a counter-based generator keyed by (seed, clip, position): the
 * same call gives the same draw), and probe_token >= 0 also returns softmax(UNFILTERED logits of the last
 * position)[probe_token] -- run on the prefix that ends at <|startoftranscript|> this is no_speech_prob.
 * next_logprobs stays the log-probability at temperature 1 under the filtered distribution (sum_logprobs).
 * Sequences of different lengths keep their self-attention K / V cache: a call whose every prefix extends the
 * previous call's by exactly one token appends one position per sequence. */
typedef struct pce_whisper_decode_opts {
    const int32_t *sample_begin;   /* [clips] first sampled position of every sequence, or NULL: sample_begin_all for all */
    int32_t sample_begin_all;
    float temperature;             /* 0: arg-max (first maximum) */
    uint32_t seed_lo, seed_hi;
    int32_t probe_token;           /* < 0: no probe */
    int32_t flags;                 /* bit 0: do not use the self-attention K / V cache (re-run the decoder over the whole prefix: the check of the cache) */
} pce_whisper_decode_opts;
int pce_whisper_decode_step_ex(pce_ctx *ctx, const int32_t *tokens, const int32_t *token_offsets /* [clips + 1] */,
                               const pce_whisper_decode_rules *rules, const uint8_t *vocab_mask, const pce_whisper_decode_opts *opts,
                               int32_t *next_tokens /* [clips] */, float *next_logprobs /* [clips] or NULL */,
                               float *probe_prob /* [clips] or NULL */);

/* What the sampling noise of a clip is keyed by.  The draw at temperature > 0 is a function of (seed, key, position, token); without this
 * call the key is the clip's position in the encoded batch, so the same recording samples differently when it is batched with other
 * clips (another shard of a multi-rank run, another batch size).  keys[i]: any int32 that names clip i wherever it is batched (the mirror's
 * transcribe() passes a hash of the clip's samples and its window start); they hold for the batch now encoded -- the next
 * pce_whisper_encode_run drops them -- and n must be that batch's clip count (n = 0: drop them now).  PCE_E_STATE before an encoder run. */
int pce_whisper_sample_keys(pce_ctx *ctx, const int32_t *keys /* [n] */, int32_t n);

/* Operand type of every Whisper / BERT matrix product (round 3).  Default since round 4: PCE_OPERANDS_F16_RESID16 (below; PCE_WHISPER_OPERANDS=fp16
 * or bf16 in the environment at pce_create selects another default).  PCE_OPERANDS_FP16: fp16 operands, the reference's own
 * arithmetic (openai-whisper runs in half precision: transcribe's fp16=True default behind
 * Code/Aligners/use_whisper_timestamped.py:163).  PCE_OPERANDS_BF16 (PCE_WHISPER_OPERANDS=bf16 in the environment at pce_create, or
 * this call): bf16 operands, 8 instead of 11 significand bits, about 3 % faster end to end (the MFMA rate is the same; the clock is not).
 * fp32 accumulation, fp32 LayerNorm / softmax statistics and an fp32 residual stream in both.  The two builds keep separate state:
 * call this BEFORE pce_whisper_load / pce_whisper_decoder_load / pce_bert_load / pce_logmel_run, and load again after switching.
 * PCE_OPERANDS_F16_RESID16 (round 4): the fp16 build with the encoder's residual stream kept in fp16 as well -- openai-whisper's own
 * `x = x + attn(...)`, `x = x + mlp(...)` are fp16 + fp16 -> fp16, only LayerNorm computes in fp32 (model.py) -- which cuts the bytes of the
 * residual / LayerNorm passes from 22 to 16 per element and layer on the batched encoder path; same state slot as PCE_OPERANDS_FP16 (no
 * reload needed between the two), pce_whisper_get_operands returns the value that was set.  The 16-bit stream belongs to the 256 x 256 GEMM path,
 * which since round 5 runs for EVERY batch size when n_state is a multiple of 256 (base, small, medium, large); a model whose width is not (tiny:
 * 384; the miniatures of the tests) takes the tiled kernels with the fp32 stream in every mode, for every batch size alike. */
enum { PCE_OPERANDS_BF16 = 0, PCE_OPERANDS_FP16 = 1, PCE_OPERANDS_F16_RESID16 = 2 };
int pce_whisper_set_operands(pce_ctx *ctx, int32_t operand_type);
int pce_whisper_get_operands(pce_ctx *ctx);

/* The whole free-running loop on the device (round 3): what whisper.decoding.DecodingTask._main_loop does for a batch
 * (Code/Aligners/use_whisper_timestamped.py:150-163 -> whisper_timestamped.transcribe -> whisper.decode).  The prompts are uploaded
 * once; every later step takes the token it embeds, its position and the "ended" flags from device memory written by the previous
 * step; the host synchronises every `check_every` steps (<= 0: 4) to read ONE counter and once at the end for the results.
 * A sequence that has produced end-of-text keeps receiving it (log-probability 0), as GreedyDecoder.update pads finished
 * sequences; the loop stops when every sequence has ended or after max_new steps.  out_tokens / out_logprobs: [clips][max_new]
 * (entries of steps that did not run: eot / 0); *out_steps = steps run.  opts / probe_prob as pce_whisper_decode_step_ex (the probe
 * belongs to step 0).  Token for token the sequence of pce_whisper_decode_step_ex calls it replaces. */
int pce_whisper_decode_loop(pce_ctx *ctx, const int32_t *tokens, const int32_t *token_offsets /* [clips + 1] */,
                            const pce_whisper_decode_rules *rules, const uint8_t *vocab_mask, const pce_whisper_decode_opts *opts,
                            int32_t max_new, int32_t check_every, int32_t *out_tokens, float *out_logprobs /* or NULL */,
                            int32_t *out_steps, float *probe_prob /* [clips] or NULL */);

/* ---- R8: dynamic time warping (alignment indices) ------------------------
 * The DTW of openai-whisper's timing.py (dtw_cpu) that whisper_timestamped's word alignment rests on
 * (Code/Aligners/use_whisper_timestamped.py:163): x is `batch` row-major [n_rows][n_cols] fp64 cost matrices
 * (tokens x frames, n_rows <= 1024); path_i / path_j receive up to n_rows + n_cols index pairs per matrix
 * (stride n_rows + n_cols), path_len their count.  The accumulated cost is float32, as dtw_cpu keeps it (every cell the
 * float64 sum of the input and the chosen predecessor, rounded to float32): indices bit-identical to that recurrence. */
int pce_dtw(pce_ctx *ctx, const double *x, int32_t n_rows, int32_t n_cols, int32_t batch, int32_t *path_i, int32_t *path_j,
            int32_t *path_len);

/* ---- batched Needleman-Wunsch word alignment ------------------------------
 * Replaces needleman_wunsch (Code/Pipeline/NeedlemanWunschAlignement.py:27-81) for a batch of sequence pairs.
 * Pair b aligns a_ids[a_off[b] .. a_off[b+1]) (rows; any number since round 5) with b_ids[b_off[b] .. b_off[b+1]); the ids are
 * the host's integer codes of the normalised tokens (:43-47), equal ids = equal tokens.  Scores as the reference's
 * keyword arguments (match 1, mismatch -1, gap -1).  Output for pair b starts at element
 * sum_{p<b} (len_a[p] + len_b[p]) of out_i / out_j and has out_len[b] steps in alignment order: (i, j) = a
 * diagonal step, (i, -1) a gap in the second sequence, (-1, j) a gap in the first; ties resolve diagonal > up >
 * left as the reference's trace-back does (:69-80).  Integer arithmetic: identical alignments. */
int pce_nw_align(pce_ctx *ctx, const int32_t *a_ids, const int64_t *a_off, const int32_t *b_ids, const int64_t *b_off, int32_t batch,
                 int32_t match, int32_t mismatch, int32_t gap, int32_t *out_i, int32_t *out_j, int32_t *out_len);

/* ---- batched Levenshtein distance --------------------------
 * Replaces levenshtein_distance (Code/Aligners/levenshtein_dist_align_txtgrids.py:43-70) for a batch of string pairs.  Pair b
 * compares a_chars[a_off[b] .. a_off[b+1]) with b_chars[b_off[b] .. b_off[b+1]); the characters are Unicode code points (what a
 * Python str iterates over).  Unit insertion / deletion / substitution costs; out_dist[b] = the distance (len of the other string
 * when one is empty, :57-58).  Integer arithmetic: identical distances; no length limit.  The caller of the reference's function,
 * main()'s merge loop (:98-158), clamps its cursors (`min(i + 1, n1 - 1)`, :113) under `while i < n1 and j < n2` and therefore
 * never terminates: it has no output to reproduce and is not part of this interface. */
int pce_levenshtein(pce_ctx *ctx, const uint32_t *a_chars, const int64_t *a_off, const uint32_t *b_chars, const int64_t *b_off,
                    int32_t batch, int32_t *out_dist);

/* ---- break-prediction token classifier --------------------
 * Forward pass of transformers.BertForTokenClassification, the model Code/baseline_models/pause_bert.py:127-132 trains
 * (bert-base-multilingual-uncased, num_labels = 2, MAX_LENGTH = 128; the reference has training code only: this is the
 * inference path a pipeline step would call).  weights: the float32 state_dict flattened in the order of
 * prosody-control-french-tts_amd/bert_weights.py:tensor_order.  Sequences are token ids (tokenisation is host logic and
 * needs the checkpoint's vocabulary); token_type_ids = 0, right padding is implicit in the offsets.  MFMA operands of
 * the context's operand mode (pce_whisper_set_operands: fp16 by default, bf16 on request), fp32 accumulation, LayerNorm /
 * residual stream / logits in fp32. */
typedef struct pce_bert_dims { int32_t n_vocab, n_pos, n_type, n_state, n_head, n_layer, n_labels; } pce_bert_dims;
int pce_bert_load(pce_ctx *ctx, const pce_bert_dims *dims, const float *weights, int64_t n_floats);
int pce_bert_run(pce_ctx *ctx, const int32_t *input_ids, const int32_t *offsets /* [n_seq + 1] */, int32_t n_seq);
/* logits: [len][n_labels] or NULL; labels: [len] argmax (first maximum) or NULL */
int pce_bert_fetch(pce_ctx *ctx, int32_t seq, float *logits, int32_t *labels);

/* ---- asynchronous statistics fetch ---------------------------------------
 * The per-slice numbers of the last pce_energy_run / pce_lufs_run / pce_pitch_run are what the reference's
 * driver consumes per utterance (Code/audioPipeline.py:380-400) and what the sharded driver all-gathers
 * (SURVEY.md 8e).  pce_stats_enqueue queues their device-to-host copies behind those runs into pinned
 * staging memory and returns at once; pce_stats_wait blocks only on those copies and unpacks them, so the
 * next batch's kernels can already be running.  slot is 0 or 1 (two batches in flight).  Any output pointer
 * may be NULL; a non-NULL pointer whose run did not precede the enqueue is an error (PCE_E_STATE). */
int pce_stats_enqueue(pce_ctx *ctx, int32_t slot);
int pce_stats_wait(pce_ctx *ctx, int32_t slot, pce_energy *energy, double *lufs, int32_t *lufs_status, pce_pitch_summary *pitch);

/* ---- measurement -------------------------------------------------------
 * With profiling on, every kernel launch is bracketed by HIP events on the
 * context's stream; pce_profile_get returns the accumulated device time. */
enum pce_kernel_id {
    PCE_K_ENERGY = 0,
    PCE_K_LUFS_PASS1, PCE_K_LUFS_SCAN, PCE_K_LUFS_PASS2, PCE_K_LUFS_GATE,
    PCE_K_PITCH_REFINE, PCE_K_PITCH_FRAMES, PCE_K_PITCH_PATH, PCE_K_PITCH_MEDIAN, PCE_K_PITCH_DELTA,
    PCE_K_STFT_MAX, PCE_K_STFT_DB, PCE_K_LOGMEL, PCE_K_WHISPER_ENC, PCE_K_RESAMPLE, PCE_K_DTW, PCE_K_WHISPER_ALIGN, PCE_K_NW, PCE_K_STFT_NORM,
    PCE_K_FRAME_ENERGY, PCE_K_BERT, PCE_K_PYIN_FRAMES, PCE_K_PYIN_VITERBI, PCE_K_WHISPER_DECODE,
    /* the launches inside the composite entries above (whisper_encoder, whisper_align, bert_forward, whisper_decode_step),
     * each bracketed on its own so that a roofline figure divides one kernel's work by that kernel's own duration */
    PCE_K_GEMM128, PCE_K_GEMM_WIDE, PCE_K_ATTENTION, PCE_K_LAYERNORM, PCE_K_GEMM_FLAT,
    /* round 3: one id per device kernel name (what rocprofv3 --kernel-trace prints), and the persistent GEMM per encoder shape
     * ("k_gemm_flat:<shape>"; PCE_K_GEMM_FLAT keeps the launches no shape is named for) */
    PCE_K_ADD_LAYERNORM, PCE_K_STFT_RAW, PCE_K_LOGMEL_NORM, PCE_K_ATTENTION_LEAN,
    PCE_K_GEMM_FLAT_QKV, PCE_K_GEMM_FLAT_OUT, PCE_K_GEMM_FLAT_FC1, PCE_K_GEMM_FLAT_FC2, PCE_K_GEMM_FLAT_XKV,
    PCE_K_DECODE_LOOP, PCE_K_CROSS_ATTN1, PCE_K_GEMM_SKINNY, PCE_K_LEVENSHTEIN, PCE_K_COUNT
};
int pce_profile_enable(pce_ctx *ctx, int on);
int pce_profile_reset(pce_ctx *ctx);
int pce_profile_get(pce_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches);
/* algorithmic work of the bracketed launches of a kernel id since the last reset: floating-point operations
 * (2 M N K of a GEMM launch, 4 T^2 d of an attention launch; 0 for ids that do not count) */
int pce_profile_get_work(pce_ctx *ctx, int kernel_id, double *flops);
const char *pce_kernel_name(int kernel_id);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* PCE_H */
