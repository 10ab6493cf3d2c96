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


def test_find_alignment_chain_matches_transformers_token_timestamps():
    """The middle of ``find_alignment`` -- alignment-head selection -> softmax -> std / mean normalisation over the token axis -> median
    filter -> mean over heads -- and the DTW path / token jump times after it, against vectors computed by transformers'
    ``_extract_token_timestamps`` and its DTW port on the transformers model's own cross-attentions (tests/golden/whisper_hf_align.npz;
    the generating script explains the row-crop difference between the two recipes and how it is handled).  The restatement starts from
    ITS OWN decoder (already cross-checked above), so this covers the whole alignment leg: same matrix (fp32 noise), same path, same
    times, for the default upper-half heads and for a picked head list, on a 4 s and a 9 s clip."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "whisper_hf_align.npz"))
    edims = dict(n_mels=80, n_ctx=1500, n_state=128, n_head=2, n_layer=2)
    tdims = dict(n_vocab=300, n_text_ctx=96, n_state=128, n_head=2, n_layer=2)
    We, Wd = WW.synthetic_weights(edims, seed=77), WW.synthetic_decoder_weights(tdims, seed=78)
    sot_len = int(g["sot_len"][0])
    for ci, seconds in g["clips"]:
        enc = WO.encoder_forward(WO.log_mel(synth.synth_clip(int(ci), seconds=float(seconds)), 80), We, edims)
        for name in ("upper_half", "picked"):
            key = f"c{int(ci)}_{name}"
            tokens = g[key + "_tokens"].tolist()
            hm = np.zeros((2, 2), dtype=bool)
            for l, h in g[key + "_heads"]:
                hm[l, h] = True
            cost, ti, tj, full = WO.find_alignment(tokens, enc, Wd, tdims, 3000, sot_len, head_mask=hm, want_matrix=True)
            want = g[key + "_matrix"]
            assert full.shape == want.shape == (len(tokens), 1500)
            assert np.max(np.abs(full - want)) <= 2e-3, (key, float(np.max(np.abs(full - want))))
            assert np.array_equal(ti, g[key + "_text_idx"]) and np.array_equal(tj, g[key + "_time_idx"]), key
            jumps = np.pad(np.diff(ti), (1, 0), constant_values=1).astype(bool)
            assert np.array_equal(tj[jumps] * 0.02, g[key + "_jump_times"])
            # the function's own return value (DTW over ALL rows: prompt rows and the last row included) from the same matrix
            ai, aj = WO.dtw_path(-full.astype(np.float64))
            aj_times = aj[np.pad(np.diff(ai), (1, 0), constant_values=1).astype(bool)] * 0.02
            assert np.allclose(np.r_[aj_times, aj_times[-1]], g[key + "_all_rows_times"], atol=1e-6)
