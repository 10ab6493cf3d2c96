// pce_whisper_dispatch.hip -- the Whisper / BERT entry points of include/pce.h: each forwards to the build of pce_whisper_impl.inc
// (bf16 or fp16 operands) the context has selected.  The two builds keep separate state (weights, buffers): switching the operand
// type means loading the weights again.
#include "pce_internal.h"

#define PCE_BOTH(ret, name, params)  extern "C" { ret name##_bf16 params; ret name##_f16 params; }
#define PCE_FWD(name, ...) ((c && c->whisper_ops >= 1) ? name##_f16(__VA_ARGS__) : name##_bf16(__VA_ARGS__))

PCE_BOTH(int, pce_logmel_run, (pce_ctx *, int32_t))
PCE_BOTH(int, pce_logmel_run_at, (pce_ctx *, int32_t, const int64_t *))
PCE_BOTH(int, pce_logmel_fetch, (pce_ctx *, int32_t, float *))
PCE_BOTH(int, pce_whisper_load, (pce_ctx *, const pce_whisper_dims *, const float *, int64_t))
PCE_BOTH(int, pce_whisper_encode_run, (pce_ctx *))
PCE_BOTH(int, pce_whisper_encode_fetch, (pce_ctx *, int32_t, float *))
PCE_BOTH(int, pce_selftest_attention, (pce_ctx *, const uint16_t *, const uint16_t *, const uint16_t *, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, uint16_t *, int32_t *))
PCE_BOTH(int, pce_selftest_gemm, (pce_ctx *, const uint16_t *, const uint16_t *, const float *, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, uint16_t *))
PCE_BOTH(int, pce_selftest_xattn, (pce_ctx *, const float *, const float *, const float *, const uint16_t *, const float *, const uint16_t *, const uint16_t *, const float *, const uint16_t *, const int32_t *, int32_t, int32_t, int32_t, int32_t, int32_t, uint16_t *))
PCE_BOTH(int, pce_whisper_decoder_load, (pce_ctx *, const pce_whisper_text_dims *, const float *, int64_t))
PCE_BOTH(int, pce_whisper_align_run, (pce_ctx *, const int32_t *, const int32_t *, const int32_t *, int32_t, const uint8_t *, int32_t, float))
PCE_BOTH(int, pce_whisper_align_fetch, (pce_ctx *, int32_t, int32_t *, int32_t *, int32_t *, double *))
PCE_BOTH(int, pce_whisper_align_shape, (pce_ctx *, int32_t, int32_t *, int32_t *))
PCE_BOTH(int, pce_whisper_align_paths_enqueue, (pce_ctx *, int32_t, int32_t *, int32_t *))
PCE_BOTH(int, pce_whisper_align_paths_wait, (pce_ctx *, int32_t, int32_t *, int32_t *, int32_t *))
PCE_BOTH(int, pce_whisper_sample_keys, (pce_ctx *, const int32_t *, int32_t))
PCE_BOTH(int, pce_whisper_decode_step, (pce_ctx *, const int32_t *, const int32_t *, int32_t, const pce_whisper_decode_rules *, const uint8_t *, int32_t *, float *))
PCE_BOTH(int, pce_whisper_decode_step_ex, (pce_ctx *, const int32_t *, const int32_t *, const pce_whisper_decode_rules *, const uint8_t *, const pce_whisper_decode_opts *, int32_t *, float *, float *))
PCE_BOTH(int, pce_whisper_decode_loop, (pce_ctx *, const int32_t *, const int32_t *, const pce_whisper_decode_rules *, const uint8_t *, const pce_whisper_decode_opts *, int32_t, int32_t, int32_t *, float *, int32_t *, float *))
PCE_BOTH(int, pce_bert_load, (pce_ctx *, const pce_bert_dims *, const float *, int64_t))
PCE_BOTH(int, pce_bert_run, (pce_ctx *, const int32_t *, const int32_t *, int32_t))
PCE_BOTH(int, pce_bert_fetch, (pce_ctx *, int32_t, float *, int32_t *))
void pce_whisper_free_bf16(pce_ctx *c);
void pce_whisper_free_f16(pce_ctx *c);

void pce_whisper_free(pce_ctx *c) { pce_whisper_free_bf16(c); pce_whisper_free_f16(c); }

extern "C" {

int pce_whisper_set_operands(pce_ctx *c, int32_t operand_type)
{
    if (!c) return PCE_E_INVALID;
    if (operand_type != PCE_OPERANDS_BF16 && operand_type != PCE_OPERANDS_FP16 && operand_type != PCE_OPERANDS_F16_RESID16)
        return pce_fail(c, PCE_E_INVALID, "operand type %d (0 = bf16, 1 = fp16, 2 = fp16 with the fp16 residual stream)", operand_type);
    c->whisper_ops = operand_type;
    c->resid16 = operand_type == PCE_OPERANDS_F16_RESID16;
    return PCE_OK;
}
int pce_whisper_get_operands(pce_ctx *c) { return c ? c->whisper_ops : PCE_E_INVALID; }

int pce_logmel_run(pce_ctx *c, int32_t n_mels) { return PCE_FWD(pce_logmel_run, c, n_mels); }
int pce_logmel_run_at(pce_ctx *c, int32_t n_mels, const int64_t *start_frames) { return PCE_FWD(pce_logmel_run_at, c, n_mels, start_frames); }
int pce_logmel_fetch(pce_ctx *c, int32_t clip, float *out) { return PCE_FWD(pce_logmel_fetch, c, clip, out); }
int pce_whisper_load(pce_ctx *c, const pce_whisper_dims *dims, const float *weights, int64_t n_floats) { return PCE_FWD(pce_whisper_load, c, dims, weights, n_floats); }
int pce_whisper_encode_run(pce_ctx *c) { return PCE_FWD(pce_whisper_encode_run, c); }
int pce_whisper_encode_fetch(pce_ctx *c, int32_t clip, float *out) { return PCE_FWD(pce_whisper_encode_fetch, c, clip, out); }
int pce_selftest_attention(pce_ctx *c, const uint16_t *q, const uint16_t *k, const uint16_t *v, int32_t clips, int32_t heads, int32_t q_len, int32_t k_len,
                           int32_t causal, int32_t mode, uint16_t *out, int32_t *fell_back)
{
    return PCE_FWD(pce_selftest_attention, c, q, k, v, clips, heads, q_len, k_len, causal, mode, out, fell_back);
}
int pce_selftest_gemm(pce_ctx *c, const uint16_t *A, const uint16_t *B, const float *bias, int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t rows_per_clip,
                      int32_t vt_sp, uint16_t *out)
{
    return PCE_FWD(pce_selftest_gemm, c, A, B, bias, M, N, K, epilogue, rows_per_clip, vt_sp, out);
}
int pce_selftest_xattn(pce_ctx *c, const float *resid, const float *ln_w, const float *ln_b, const uint16_t *wq, const float *bq, const uint16_t *wk, const uint16_t *wv,
                       const float *bv, const uint16_t *E, const int32_t *k_len, int32_t n, int32_t k_cap, int32_t d, int32_t heads, int32_t workgroups_per_clip, uint16_t *out)
{
    return PCE_FWD(pce_selftest_xattn, c, resid, ln_w, ln_b, wq, bq, wk, wv, bv, E, k_len, n, k_cap, d, heads, workgroups_per_clip, out);
}
int pce_whisper_decoder_load(pce_ctx *c, const pce_whisper_text_dims *dims, const float *weights, int64_t n_floats)
{
    return PCE_FWD(pce_whisper_decoder_load, c, dims, weights, n_floats);
}
int pce_whisper_align_run(pce_ctx *c, const int32_t *tokens, const int32_t *token_offsets, const int32_t *num_frames, int32_t sot_len, const uint8_t *head_mask,
                          int32_t medfilt_width, float qk_scale)
{
    return PCE_FWD(pce_whisper_align_run, c, tokens, token_offsets, num_frames, sot_len, head_mask, medfilt_width, qk_scale);
}
int pce_whisper_align_fetch(pce_ctx *c, int32_t clip, int32_t *text_idx, int32_t *time_idx, int32_t *path_len, double *cost)
{
    return PCE_FWD(pce_whisper_align_fetch, c, clip, text_idx, time_idx, path_len, cost);
}
int pce_whisper_align_shape(pce_ctx *c, int32_t clip, int32_t *n_rows, int32_t *n_cols) { return PCE_FWD(pce_whisper_align_shape, c, clip, n_rows, n_cols); }
int pce_whisper_align_paths_enqueue(pce_ctx *c, int32_t slot, int32_t *n_clips, int32_t *path_stride) { return PCE_FWD(pce_whisper_align_paths_enqueue, c, slot, n_clips, path_stride); }
int pce_whisper_align_paths_wait(pce_ctx *c, int32_t slot, int32_t *path_len, int32_t *text_idx, int32_t *time_idx)
{
    return PCE_FWD(pce_whisper_align_paths_wait, c, slot, path_len, text_idx, time_idx);
}
int pce_whisper_sample_keys(pce_ctx *c, const int32_t *keys, int32_t n) { return PCE_FWD(pce_whisper_sample_keys, c, keys, n); }
int pce_whisper_decode_step(pce_ctx *c, const int32_t *tokens, const int32_t *token_offsets, int32_t sample_begin, const pce_whisper_decode_rules *rules,
                            const uint8_t *vocab_mask, int32_t *next_tokens, float *next_logprobs)
{
    return PCE_FWD(pce_whisper_decode_step, c, tokens, token_offsets, sample_begin, rules, vocab_mask, next_tokens, next_logprobs);
}
int pce_whisper_decode_step_ex(pce_ctx *c, const int32_t *tokens, const int32_t *token_offsets, const pce_whisper_decode_rules *rules, const uint8_t *vocab_mask,
                               const pce_whisper_decode_opts *opts, int32_t *next_tokens, float *next_logprobs, float *probe_prob)
{
    return PCE_FWD(pce_whisper_decode_step_ex, c, tokens, token_offsets, rules, vocab_mask, opts, next_tokens, next_logprobs, probe_prob);
}
int pce_whisper_decode_loop(pce_ctx *c, const int32_t *tokens, const int32_t *token_offsets, const pce_whisper_decode_rules *rules, const uint8_t *vocab_mask,
                            const pce_whisper_decode_opts *opts, int32_t max_new, int32_t check_every, int32_t *out_tokens, float *out_logprobs, int32_t *out_steps,
                            float *probe_prob)
{
    return PCE_FWD(pce_whisper_decode_loop, c, tokens, token_offsets, rules, vocab_mask, opts, max_new, check_every, out_tokens, out_logprobs, out_steps, probe_prob);
}
int pce_bert_load(pce_ctx *c, const pce_bert_dims *dims, const float *weights, int64_t n_floats) { return PCE_FWD(pce_bert_load, c, dims, weights, n_floats); }
int pce_bert_run(pce_ctx *c, const int32_t *input_ids, const int32_t *offsets, int32_t n_seq) { return PCE_FWD(pce_bert_run, c, input_ids, offsets, n_seq); }
int pce_bert_fetch(pce_ctx *c, int32_t seq, float *logits, int32_t *labels) { return PCE_FWD(pce_bert_fetch, c, seq, logits, labels); }

} // extern "C"
