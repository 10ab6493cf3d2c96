"""Generates tests/golden/transcribe_hf_long.json: LONG-FORM transcription (recordings longer than one 30 s window) of a random-init
two-layer Whisper by the INSTALLED transformers implementation -- ``WhisperForConditionalGeneration.generate`` with
``return_segments=True``: transformers' own port of ``whisper.transcribe``'s window loop (seek from the timestamp tokens,
``condition_on_prev_tokens`` = openai's ``condition_on_previous_text``, the no-speech / log-probability skip rule).  The fixture pins
``Aligners/transcribe.py``'s restatement of that loop (tests/test_transcribe_flow_hf.py) against an implementation its author did
not write.  The vocabulary layout is the real one (``WhisperTokenizer.toy``: 256 byte tokens + the special tokens + 1 501 timestamps).
Run in the build container:  python tests/golden/make_goldens_transcribe_hf.py"""
import json
import os
import sys

import numpy as np
import torch
from transformers import GenerationConfig, WhisperConfig, WhisperForConditionalGeneration

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import whisper_oracle as WO  # noqa: E402
from prosody_control_french_tts_amd import synth, whisper_weights as WW  # noqa: E402
from prosody_control_french_tts_amd.Aligners.tokenizer import WhisperTokenizer  # noqa: E402

TOK = WhisperTokenizer.toy()
EDIMS = dict(n_mels=80, n_ctx=1500, n_state=128, n_head=2, n_layer=2)
TDIMS = dict(n_vocab=TOK.n_vocab, n_text_ctx=96, n_state=128, n_head=2, n_layer=2)
SEEDS = (77, 79)
CLIPS = {"a": [10 + k for k in range(13)], "b": [30 + k for k in range(22)]}     # 39 s and 66 s of 3 s synthetic pieces
SAMPLE_LEN = 40
T = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def weights():
    return WW.synthetic_weights(EDIMS, seed=SEEDS[0]), WW.greedy_test_decoder_weights(TDIMS, seed=SEEDS[1])


def clip(name):
    return np.concatenate([synth.synth_clip(k, seconds=3.0) for k in CLIPS[name]])


def mel_full(pcm):
    """log-mel of the whole recording with openai's normalisation (global maximum), one column per 10 ms of content."""
    n = len(pcm) // 160
    return np.concatenate([WO.log_mel_window(pcm, s, 80) for s in range(0, n + 3000, 3000)], axis=1)[:, :n]


def build_model(We, Wd):
    V = TDIMS["n_vocab"]
    cfg = WhisperConfig(vocab_size=V, num_mel_bins=80, d_model=128, encoder_layers=2, encoder_attention_heads=2, decoder_layers=2,
                        decoder_attention_heads=2, encoder_ffn_dim=512, decoder_ffn_dim=512, max_source_positions=1500, max_target_positions=96,
                        activation_function="gelu", dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, pad_token_id=TOK.eot, bos_token_id=TOK.eot,
                        eos_token_id=TOK.eot, decoder_start_token_id=TOK.sot, suppress_tokens=None, begin_suppress_tokens=None, attn_implementation="eager")
    model = WhisperForConditionalGeneration(cfg).eval()
    sd = {}

    def attn(dst, src, W):
        sd[dst + "q_proj.weight"] = T(W[src + "query.weight"]); sd[dst + "q_proj.bias"] = T(W[src + "query.bias"])
        sd[dst + "k_proj.weight"] = T(W[src + "key.weight"])
        sd[dst + "v_proj.weight"] = T(W[src + "value.weight"]); sd[dst + "v_proj.bias"] = T(W[src + "value.bias"])
        sd[dst + "out_proj.weight"] = T(W[src + "out.weight"]); sd[dst + "out_proj.bias"] = T(W[src + "out.bias"])
    sd["model.encoder.conv1.weight"] = T(We["conv1.weight"]); sd["model.encoder.conv1.bias"] = T(We["conv1.bias"])
    sd["model.encoder.conv2.weight"] = T(We["conv2.weight"]); sd["model.encoder.conv2.bias"] = T(We["conv2.bias"])
    sd["model.encoder.embed_positions.weight"] = T(WO.sinusoids(1500, 128))
    sd["model.encoder.layer_norm.weight"] = T(We["ln_post.weight"]); sd["model.encoder.layer_norm.bias"] = T(We["ln_post.bias"])
    for l in range(2):
        h, o = f"model.encoder.layers.{l}.", f"blocks.{l}."
        attn(h + "self_attn.", o + "attn.", We)
        for a, b in (("self_attn_layer_norm", "attn_ln"), ("final_layer_norm", "mlp_ln"), ("fc1", "mlp.0"), ("fc2", "mlp.2")):
            sd[h + a + ".weight"] = T(We[o + b + ".weight"]); sd[h + a + ".bias"] = T(We[o + b + ".bias"])
        h = f"model.decoder.layers.{l}."
        attn(h + "self_attn.", o + "attn.", Wd); attn(h + "encoder_attn.", o + "cross_attn.", Wd)
        for a, b in (("self_attn_layer_norm", "attn_ln"), ("encoder_attn_layer_norm", "cross_attn_ln"), ("final_layer_norm", "mlp_ln"),
                     ("fc1", "mlp.0"), ("fc2", "mlp.2")):
            sd[h + a + ".weight"] = T(Wd[o + b + ".weight"]); sd[h + a + ".bias"] = T(Wd[o + b + ".bias"])
    sd["model.decoder.embed_tokens.weight"] = T(Wd["token_embedding.weight"])
    sd["model.decoder.embed_positions.weight"] = T(Wd["positional_embedding"])
    sd["model.decoder.layer_norm.weight"] = T(Wd["ln.weight"]); sd["model.decoder.layer_norm.bias"] = T(Wd["ln.bias"])
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and not res.missing_keys, res
    rules = TOK.decoding_rules()
    model.generation_config = GenerationConfig(
        eos_token_id=TOK.eot, pad_token_id=TOK.eot, bos_token_id=TOK.eot, decoder_start_token_id=TOK.sot, no_timestamps_token_id=TOK.no_timestamps,
        prev_sot_token_id=TOK.sot_prev, is_multilingual=True, lang_to_id={"<|fr|>": TOK.language_token("fr")}, task_to_id={"transcribe": TOK.transcribe},
        max_initial_timestamp_index=rules["max_initial_timestamp_index"], suppress_tokens=sorted(rules["suppress_tokens"]),
        begin_suppress_tokens=list(rules["blank_tokens"]), max_length=96, return_timestamps=True)
    return model


def run(model, pcm, **kw):
    feats = torch.from_numpy(mel_full(pcm))[None]
    am = torch.ones(1, feats.shape[-1], dtype=torch.long)
    with torch.no_grad():
        out = model.generate(input_features=feats, attention_mask=am, return_timestamps=True, return_segments=True, language="fr", task="transcribe",
                             temperature=(0.0,), max_new_tokens=SAMPLE_LEN, **kw)
    return [{"start": float(s["start"]), "end": float(s["end"]), "tokens": [int(t) for t in s["tokens"]]} for s in out["segments"][0]]


def main():
    We, Wd = weights()
    model = build_model(We, Wd)
    cases = []
    for name in CLIPS:
        pcm = clip(name)
        for cond in (True, False):
            segs = run(model, pcm, condition_on_prev_tokens=cond, logprob_threshold=None, compression_ratio_threshold=None, no_speech_threshold=None)
            cases.append({"clip": name, "condition_on_previous_text": cond, "no_speech_threshold": None, "logprob_threshold": None, "segments": segs})
            print(name, cond, len(segs), [s["tokens"] for s in segs[:3]])
    # the skip rule of whisper.transcribe: a window whose no-speech probability exceeds the threshold is dropped unless its average
    # log-probability clears logprob_threshold (thresholds chosen between this model's per-window values so that SOME windows go)
    for nst, lpt in ((0.001, -1.0), (0.0006, -1.0), (0.001, -5.0)):
        segs = run(model, clip("b"), condition_on_prev_tokens=True, logprob_threshold=lpt, compression_ratio_threshold=None, no_speech_threshold=nst)
        cases.append({"clip": "b", "condition_on_previous_text": True, "no_speech_threshold": nst, "logprob_threshold": lpt, "segments": segs})
        print("b", nst, lpt, len(segs), [s["start"] for s in segs])
    json.dump({"made_by": "transformers " + __import__("transformers").__version__ + " WhisperForConditionalGeneration.generate(return_segments=True)",
               "sample_len": SAMPLE_LEN, "cases": cases}, open(os.path.join(HERE, "transcribe_hf_long.json"), "w"), indent=0)
    print("wrote transcribe_hf_long.json")


if __name__ == "__main__":
    main()
