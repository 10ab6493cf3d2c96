"""The Whisper restatement (oracle/whisper_oracle.py: log-mel -> audio encoder -> teacher-forced text decoder with its
cross-attention logits) against outputs of an independent implementation of the same network: the installed
transformers WhisperForConditionalGeneration carrying the same fixed-seed weights (tests/golden/whisper_hf_tiny.npz,
made by tests/golden/make_goldens_whisper_hf.py).  fp32 on both sides."""
import os

import numpy as np
import torch

from oracle import whisper_oracle as WO
from prosody_control_french_tts_amd import synth, whisper_weights as WW

GOLD = os.path.join(os.path.dirname(__file__), "golden", "whisper_hf_tiny.npz")


def test_encoder_decoder_and_cross_attention_match_transformers():
    g = np.load(GOLD)
    edims = dict(n_mels=80, n_ctx=1500, n_state=128, n_head=2, n_layer=2)
    tdims = dict(n_vocab=300, n_text_ctx=96, n_state=128, n_head=2, n_layer=2)
    We, Wd = WW.synthetic_weights(edims, seed=77), WW.synthetic_decoder_weights(tdims, seed=78)
    clip = synth.synth_clip(int(g["clip_index"][0]), seconds=float(g["seconds"][0]))
    mel = WO.log_mel(clip, 80)
    # transformers' WhisperFeatureExtractor (observed: identical on the build container's CPU, 1.7e-5 on the GPU box's host: torch's CPU FFT differs by host)
    assert np.max(np.abs(mel[:, g["mel_cols"]] - g["hf_mel"])) <= 5e-5
    enc = WO.encoder_forward(mel, We, edims)
    assert enc.shape == (1500, 128)
    assert np.max(np.abs(enc[g["rows"]] - g["enc_rows"])) <= 2e-4, float(np.max(np.abs(enc[g["rows"]] - g["enc_rows"])))
    tokens = g["tokens"].tolist()
    hm = np.ones((2, 2), dtype=bool)
    inner = WO.find_alignment(tokens, enc, Wd, tdims, 400, 3, head_mask=hm, want_internal=True)
    assert inner["logits"].shape == g["logits"].shape == (len(tokens), 300)
    assert np.max(np.abs(inner["logits"] - g["logits"])) <= 2e-3 * max(1.0, float(np.abs(g["logits"]).max()))
    # cross-attention of layer 1, head 1: softmax over all 1500 frames of the restatement's logits
    qk = torch.from_numpy(inner["cross_qk"][1 * 2 + 1])
    att = torch.softmax(qk.float(), dim=-1).numpy()
    assert att.shape == (len(tokens), 1500)
    assert np.max(np.abs(att - g["cross_attn"].astype(np.float32))) <= 2e-3        # the fixture stores float16


def test_dtw_and_median_filter_match_transformers_ports():
    """transformers carries ports of openai-whisper's ``dtw_cpu`` / ``backtrace`` and ``median_filter``
    (generation_whisper._dynamic_time_warping, ._median_filter): the cost array is float32 there, in the restatement and in
    k_dtw alike (every cell the float64 sum rounded to float32): same paths, also on exact ties."""
    import transformers.models.whisper.generation_whisper as G
    rng = np.random.default_rng(0)
    for _ in range(12):
        n, m = int(rng.integers(5, 40)), int(rng.integers(50, 300))
        t = np.linspace(0, 1, m)[None, :]
        c = np.sort(rng.uniform(0, 1, size=(n, 1)), axis=0)
        x = -np.exp(-0.5 * ((t - c) / 0.03) ** 2) + 0.05 * rng.standard_normal((n, m))
        ti, tj = WO.dtw_path(x)
        hi, hj = G._dynamic_time_warping(x)
        assert np.array_equal(ti, hi) and np.array_equal(tj, hj)
    x = rng.integers(0, 3, size=(20, 55)).astype(np.float64)
    ti, tj = WO.dtw_path(x)
    hi, hj = G._dynamic_time_warping(x)
    assert np.array_equal(ti, hi) and np.array_equal(tj, hj)
    w = torch.randn(3, 4, 17, 40)
    for width in (3, 7, 9):
        assert torch.equal(WO.median_filter(w, width), G._median_filter(w, width))


def test_word_timings_follow_the_first_arrival_rule():
    from prosody_control_french_tts_amd.Aligners import use_whisper_timestamped as UW
    ti = [0, 0, 1, 1, 1, 2, 3, 3]; tj = [0, 1, 2, 3, 4, 5, 6, 7]
    got = UW.word_timings(ti, tj, [" bon", " jour", ""], [2, 1, 1])
    assert got == [{"text": " bon", "start": 0.0, "end": 0.1}, {"text": " jour", "start": 0.1, "end": 0.12}]
    res = UW.transcription_result(got)
    assert res["text"] == "bon jour" and [w["text"] for w in res["segments"][0]["words"]] == [" bon", " jour"]
    assert UW.transcription_result([]) == {"text": "", "segments": [], "language": "fr"}
    # the same first-arrival times from transformers' port of the token-level rule (generation_whisper: jumps / jump_times)
    jumps = np.pad(np.diff(np.array(ti)), (1, 0), constant_values=1).astype(bool)
    assert np.array_equal(np.array(tj)[jumps], [0, 2, 5, 6])


def _greedy_gold():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "whisper_greedy_tiny.npz"))
    eot, no_ts, tsb, mi = (int(v) for v in g["layout"])
    rules = dict(eot=eot, no_timestamps=no_ts, timestamp_begin=tsb, suppress_tokens=g["suppress"].tolist(), blank_tokens=g["blank"].tolist(),
                 max_initial_timestamp_index=mi)
    return g, rules


def test_greedy_decoding_matches_transformers_model_and_logit_processors():
    """Free-running decoding at temperature 0: the restatement of openai-whisper's DecodingTask loop and logit filters
    against token sequences produced by the installed transformers model driven through transformers' own ports of those
    filters (tests/golden/make_goldens_whisper_greedy.py): timestamps in pairs, monotone, text in between."""
    g, rules = _greedy_gold()
    edims = dict(n_mels=80, n_ctx=1500, n_state=128, n_head=2, n_layer=2)
    tdims = dict(n_vocab=300, n_text_ctx=96, n_state=128, n_head=2, n_layer=2)
    We, Wd = WW.synthetic_weights(edims, seed=77), WW.greedy_test_decoder_weights(tdims, seed=79)
    for ci in g["clips"]:
        clip = synth.synth_clip(int(ci), seconds=4.0)
        enc = WO.encoder_forward(WO.log_mel(clip, 80), We, edims)
        want = g[f"tokens_{int(ci)}"].tolist()
        got = WO.greedy_decode(enc, Wd, tdims, g["initial"].tolist(), rules, sample_len=len(want) - len(g["initial"]))
        assert got == want, (int(ci), got, want)
        new = want[len(g["initial"]):]
        assert new[0] >= rules["timestamp_begin"] and any(t < rules["eot"] for t in new)
