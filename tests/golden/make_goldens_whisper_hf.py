"""Generates tests/golden/whisper_hf_tiny.npz with the INSTALLED transformers implementation of Whisper
(WhisperForConditionalGeneration: an independent port of openai-whisper's model.py, the network behind
Code/Aligners/use_whisper_timestamped.py:163): random-init two-layer model carrying the package's fixed-seed numpy
weights, one synthetic clip; stores sampled rows of the encoder output, the teacher-forced decoder logits and one
cross-attention map.  Run in the build container:  python tests/golden/make_goldens_whisper_hf.py"""
import os
import sys

import numpy as np
import torch
from transformers import WhisperConfig, WhisperFeatureExtractor, WhisperForConditionalGeneration

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import whisper_oracle as WO  # noqa: E402
from prosody_control_french_tts_amd import synth, whisper_weights as WW  # noqa: E402

edims = dict(n_mels=80, n_ctx=1500, n_state=128, n_head=2, n_layer=2)
tdims = dict(n_vocab=300, n_text_ctx=96, n_state=128, n_head=2, n_layer=2)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def build_model(We, Wd):
    """transformers WhisperForConditionalGeneration (eval) carrying the given openai-named numpy weights."""
    cfg = WhisperConfig(vocab_size=300, num_mel_bins=80, d_model=128, encoder_layers=2, encoder_attention_heads=2, decoder_layers=2,
                        decoder_attention_heads=2, encoder_ffn_dim=512, decoder_ffn_dim=512, max_source_positions=1500, max_target_positions=96,
                        activation_function="gelu", dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, pad_token_id=0, bos_token_id=1,
                        eos_token_id=2, decoder_start_token_id=1, suppress_tokens=None, begin_suppress_tokens=None, attn_implementation="eager")
    model = WhisperForConditionalGeneration(cfg).eval()
    sd = {}
    sd["model.encoder.conv1.weight"] = T(We["conv1.weight"]); sd["model.encoder.conv1.bias"] = T(We["conv1.bias"])
    sd["model.encoder.conv2.weight"] = T(We["conv2.weight"]); sd["model.encoder.conv2.bias"] = T(We["conv2.bias"])
    sd["model.encoder.embed_positions.weight"] = T(WO.sinusoids(1500, 128))
    sd["model.encoder.layer_norm.weight"] = T(We["ln_post.weight"]); sd["model.encoder.layer_norm.bias"] = T(We["ln_post.bias"])

    def attn(dst, src, W):
        sd[dst + "q_proj.weight"] = T(W[src + "query.weight"]); sd[dst + "q_proj.bias"] = T(W[src + "query.bias"])
        sd[dst + "k_proj.weight"] = T(W[src + "key.weight"])
        sd[dst + "v_proj.weight"] = T(W[src + "value.weight"]); sd[dst + "v_proj.bias"] = T(W[src + "value.bias"])
        sd[dst + "out_proj.weight"] = T(W[src + "out.weight"]); sd[dst + "out_proj.bias"] = T(W[src + "out.bias"])

    for l in range(2):
        h, o = f"model.encoder.layers.{l}.", f"blocks.{l}."
        attn(h + "self_attn.", o + "attn.", We)
        for a, b in (("self_attn_layer_norm", "attn_ln"), ("final_layer_norm", "mlp_ln"), ("fc1", "mlp.0"), ("fc2", "mlp.2")):
            sd[h + a + ".weight"] = T(We[o + b + ".weight"]); sd[h + a + ".bias"] = T(We[o + b + ".bias"])
        h = f"model.decoder.layers.{l}."
        attn(h + "self_attn.", o + "attn.", Wd)
        attn(h + "encoder_attn.", o + "cross_attn.", Wd)
        for a, b in (("self_attn_layer_norm", "attn_ln"), ("encoder_attn_layer_norm", "cross_attn_ln"), ("final_layer_norm", "mlp_ln"),
                     ("fc1", "mlp.0"), ("fc2", "mlp.2")):
            sd[h + a + ".weight"] = T(Wd[o + b + ".weight"]); sd[h + a + ".bias"] = T(Wd[o + b + ".bias"])
    sd["model.decoder.embed_tokens.weight"] = T(Wd["token_embedding.weight"])
    sd["model.decoder.embed_positions.weight"] = T(Wd["positional_embedding"])
    sd["model.decoder.layer_norm.weight"] = T(Wd["ln.weight"]); sd["model.decoder.layer_norm.bias"] = T(Wd["ln.bias"])
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and not res.missing_keys, res
    return model


def main():
    We, Wd = WW.synthetic_weights(edims, seed=77), WW.synthetic_decoder_weights(tdims, seed=78)
    model = build_model(We, Wd)
    clip = synth.synth_clip(5, seconds=4.0)
    mel = WO.log_mel(clip, 80)                                                 # [80][3000]
    tokens = np.random.default_rng(6).integers(3, 300, size=21).tolist()
    with torch.no_grad():
        out = model(input_features=torch.from_numpy(mel)[None], decoder_input_ids=torch.tensor([tokens]), output_attentions=True)
    enc = out.encoder_last_hidden_state[0].numpy()
    rows = np.r_[0:8, 100:1500:100, 1492:1500]
    # the log-mel front end by transformers' own feature extractor (its defaults are Whisper's: 80 bins, 400 / 160, 30 s)
    fe = WhisperFeatureExtractor(feature_size=80, sampling_rate=16000, hop_length=160, chunk_length=30, n_fft=400)
    hf_mel = fe(clip.astype(np.float32) / 32768.0, sampling_rate=16000, return_tensors="np")["input_features"][0]
    mel_cols = np.r_[0:400:7, 400:3000:97]
    np.savez_compressed(os.path.join(HERE, "whisper_hf_tiny.npz"), clip_index=np.array([5]), seconds=np.array([4.0]), tokens=np.array(tokens, dtype=np.int32),
                        mel_cols=mel_cols.astype(np.int32), hf_mel=hf_mel[:, mel_cols].astype(np.float32), rows=rows.astype(np.int32), enc_rows=enc[rows].astype(np.float32), logits=out.logits[0].numpy().astype(np.float32),
                        cross_attn=out.cross_attentions[1][0, 1].numpy().astype(np.float16))       # layer 1, head 1: softmax over the 1500 frames
    print("wrote whisper_hf_tiny.npz", enc.shape, out.logits.shape, out.cross_attentions[1].shape)


if __name__ == "__main__":
    main()
