"""GPU parity of the Whisper front end (R8): log-mel and audio encoder vs a torch fp32 CPU
restatement (oracle/whisper_oracle.py), fixed-seed synthetic weights.

Tolerances: log-mel <= 2e-3 absolute in Whisper's (x+4)/4 units (fp32 DFT by 25x16
decomposition vs torch's FFT; values at the max-8 clamp agree to the same band); encoder
output relative L2 error <= 2e-2 per clip and <= 6e-2 max-abs on unit-variance outputs
(bf16 MFMA operands, fp32 accumulation and residual stream)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import whisper_oracle as WO
from prosody_control_french_tts_amd import synth, whisper_weights as WW

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def clips():
    rng = np.random.default_rng(3)
    return [synth.synth_clip(0, seconds=10.0), synth.synth_clip(1, seconds=3.3),
            (rng.standard_normal(16000 * 31) * 2500).astype(np.int16),            # longer than the 30 s window
            np.zeros(16000, dtype=np.int16)]


def test_logmel_matches_torch(engine, clips):
    engine.upload(clips, 16000)
    engine.logmel_run(80)
    for i, c in enumerate(clips):
        got = engine.logmel_fetch(i)
        want = WO.log_mel(c[:WO.N_SAMPLES], 80)
        assert got.shape == want.shape == (80, 3000)
        assert np.max(np.abs(got - want)) <= 2e-3, (i, np.max(np.abs(got - want)))


def test_mel_filterbank_known_values():
    w = WO.mel_filters(80)
    assert w.shape == (80, 201) and abs(float(w[0, 1]) - 0.02486259) < 1e-7 and (w > 0).sum() == 391


@pytest.mark.parametrize("dims", [dict(n_mels=80, n_ctx=1500, n_state=128, n_head=2, n_layer=2), WW.DIMS["tiny"],
                                  # the widths of "small" (BASELINE C3) and "medium" (the reference's default, config.yaml:15),
                                  # two layers each: the 128 x 256 GEMM with the fused QKV / V-transpose epilogue only runs at these
                                  dict(WW.DIMS["small"], n_layer=2), dict(WW.DIMS["medium"], n_layer=2),
                                  # 512 and 1280: the other widths the persistent 256 x 256 GEMM takes (2 / 5 column tiles per projection)
                                  dict(WW.DIMS["base"], n_layer=2), dict(n_mels=80, n_ctx=1500, n_state=1280, n_head=20, n_layer=1)])
def test_encoder_matches_torch(engine, clips, dims, ops):
    """Every width the GEMM / attention routing distinguishes, per operand type, against the fp32 restatement.  Bounds = what the
    operand rounding allows after a stack of layers (measured, tools/operand_precision.py / DESIGN section 4: fp16 4.6e-4 rel-L2 at
    Whisper-small depth, bf16 5.9e-3), with a margin of 3: fp16 1.5e-3 / 1.2e-2 sigma max, bf16 2e-2 / 6e-2 sigma max."""
    W = WW.synthetic_weights(dims)
    engine.upload(clips[:2], 16000)
    engine.logmel_run(dims["n_mels"])
    engine.whisper_load(dims, WW.pack(W, dims))
    engine.whisper_encode_run()
    l2_bound, max_bound = ops["enc_l2"], ops["enc_max"]
    for i in range(2):
        got = engine.whisper_encode_fetch(i)
        want = WO.encoder_forward(WO.log_mel(clips[i], dims["n_mels"]), W, dims)
        assert got.shape == want.shape == (1500, dims["n_state"])
        rel = np.linalg.norm(got - want) / np.linalg.norm(want)
        assert rel <= l2_bound, (ops["name"], rel)
        worst = np.max(np.abs(got - want)) / max(1.0, float(np.std(want)))
        assert worst <= max_bound, (ops["name"], worst)


def test_dtw_indices_bit_exact(engine):
    """Alignment indices: the GPU wavefront DTW returns exactly the path of the CPU recurrence."""
    rng = np.random.default_rng(9)
    mats = []
    for (n, m) in [(1, 1), (1, 40), (37, 1), (5, 7), (64, 200)]:
        mats.append(rng.standard_normal((n, m)))
    # ties everywhere (integer costs) and a hand-computable diagonal
    mats.append(rng.integers(0, 3, size=(20, 55)).astype(np.float64))
    mats.append(-np.eye(12, 12))
    for x in mats:
        (gi, gj), = engine.dtw(x)
        wi, wj = WO.dtw_path(x)
        assert np.array_equal(gi, wi) and np.array_equal(gj, wj), x.shape
    # batch of Whisper-sized matrices (tokens x 1500 frames), attention-like costs
    t = np.linspace(0, 1, 1500)[None, :]
    centers = np.sort(rng.uniform(0, 1, size=(3, 90, 1)), axis=1)
    cost = -np.exp(-0.5 * ((t - centers) / 0.02) ** 2) + 0.05 * rng.standard_normal((3, 90, 1500))
    for x, (gi, gj) in zip(cost, engine.dtw(cost)):
        wi, wj = WO.dtw_path(x)
        assert np.array_equal(gi, wi) and np.array_equal(gj, wj)
        assert gi[0] == 0 and gj[0] == 0 and gi[-1] == 89 and gj[-1] == 1499 and np.all(np.diff(gi) >= 0) and np.all(np.diff(gj) >= 0)


@pytest.mark.parametrize("dims", [dict(WW.DIMS["small"], n_layer=4), dict(WW.DIMS["base"], n_layer=3)])
def test_encoder_with_the_fp16_residual_stream(engine, clips, dims):
    """``PCE_OPERANDS_F16_RESID16``: fp16 operands and the encoder's residual stream in fp16 (openai-whisper's own arithmetic: fp16 + fp16 ->
    fp16 residual adds, fp32 LayerNorm), on the batched path (n_state % 256 == 0, two clips = 3 000 rows).  Against the fp32 restatement
    within the fp16 bounds of test_encoder_matches_torch, and close to the fp32-stream result (the stream's 24 extra roundings)."""
    W = WW.synthetic_weights(dims)
    engine.upload(clips[:2], 16000)
    outs = {}
    try:
        for kind in ("fp16", "fp16-resid16"):
            engine.whisper_set_operands(kind)
            assert engine.whisper_operands == kind
            engine.logmel_run(dims["n_mels"])
            engine.whisper_load(dims, WW.pack(W, dims))
            engine.whisper_encode_run()
            outs[kind] = [engine.whisper_encode_fetch(i) for i in range(2)]
    finally:
        engine.whisper_set_operands("fp16")
    from tests.conftest import OPERANDS
    for i in range(2):
        want = WO.encoder_forward(WO.log_mel(clips[i], dims["n_mels"]), W, dims)
        for kind in outs:
            got = outs[kind][i]
            rel = np.linalg.norm(got - want) / np.linalg.norm(want)
            assert rel <= OPERANDS["fp16"]["enc_l2"], (kind, rel)
            assert np.max(np.abs(got - want)) / max(1.0, float(np.std(want))) <= OPERANDS["fp16"]["enc_max"], kind
        a, b = outs["fp16"][i], outs["fp16-resid16"][i]
        assert not np.array_equal(a, b)                                          # the mode really changes the arithmetic
        assert np.linalg.norm(a - b) / np.linalg.norm(a) <= 1.5e-3


@pytest.mark.parametrize("width,heads", [(128, 2), (768, 12)])
def test_forced_alignment_matches_torch(engine, clips, width, heads):
    """Teacher-forced decoder + cross-attention alignment (openai-whisper find_alignment up to the DTW path).
    The cost matrix agrees with the torch fp32 restatement within bf16 tolerance; the GPU DTW path is EXACTLY the
    CPU recurrence's path on the GPU's own cost matrix (alignment indices bit-exact given identical costs)."""
    edims = dict(n_mels=80, n_ctx=1500, n_state=width, n_head=heads, n_layer=2)
    tdims = dict(n_vocab=300, n_text_ctx=96, n_state=width, n_head=heads, n_layer=4)     # 768: the width of "small" (wide cross K/V GEMM)
    We, Wd = WW.synthetic_weights(edims), WW.synthetic_decoder_weights(tdims)
    use = clips[:2]
    engine.upload(use, 16000)
    engine.logmel_run(80)
    engine.whisper_load(edims, WW.pack(We, edims))
    engine.whisper_encode_run()
    engine.whisper_decoder_load(tdims, WW.pack_decoder(Wd, tdims))
    rng = np.random.default_rng(21)
    sot_len = 3
    toks = [rng.integers(0, 300, size=n).tolist() for n in (37, 70)]
    num_frames = [len(c) // 160 for c in use]                      # mel frames of the real audio
    res = engine.whisper_align(toks, num_frames, sot_len, want_cost=True)
    for i in range(2):
        enc = engine.whisper_encode_fetch(i)                       # same audio features for both sides
        cost, ti, tj = WO.find_alignment(toks[i], enc, Wd, tdims, num_frames[i], sot_len)
        got = res[i]
        assert got["cost"].shape == cost.shape == (len(toks[i]) - sot_len - 1, num_frames[i] // 2)
        assert np.max(np.abs(got["cost"] - cost)) <= 0.08                                    # observed 0.035 at std 0.4
        assert np.linalg.norm(got["cost"] - cost) / np.linalg.norm(cost) <= 3e-2            # observed 9e-3 (bf16 operands)
        wi, wj = WO.dtw_path(got["cost"])
        assert np.array_equal(got["text_indices"], wi) and np.array_equal(got["time_indices"], wj)
        # and the path agrees with the torch path almost everywhere (costs differ by bf16 rounding)
        n = min(len(ti), len(got["text_indices"]))
        jumps_g = got["time_indices"][np.r_[True, np.diff(got["text_indices"]) > 0]]
        jumps_w = tj[np.r_[True, np.diff(ti) > 0]]
        assert len(jumps_g) == len(jumps_w) and np.mean(np.abs(jumps_g - jumps_w) <= 1) >= 0.95          # observed: identical
    # the batched asynchronous fetch (pce_whisper_align_paths_enqueue / _wait: what a batch pipeline and bench.py's timed step use) hands over
    # the same indices as the clip-by-clip fetch; two slots in flight; a wait without an enqueue is an error
    engine.whisper_align_run(toks, num_frames, sot_len)
    engine.whisper_align_paths_enqueue(0); engine.whisper_align_paths_enqueue(1)
    for slot in (1, 0):
        pl, pi, pj = engine.whisper_align_paths_wait(slot)
        assert pi.shape == pj.shape == (2, max(len(t) - sot_len - 1 for t in toks) + max(num_frames) // 2)
        for i in range(2):
            assert pl[i] == len(res[i]["text_indices"])
            assert np.array_equal(pi[i, :pl[i]], res[i]["text_indices"]) and np.array_equal(pj[i, :pl[i]], res[i]["time_indices"])
    with pytest.raises(Exception):
        engine.whisper_align_paths_wait(0)
    # a slot that still holds a fetch nobody waited for is refused (its copies may still be landing in the staging buffer), then usable again
    engine.whisper_align_paths_enqueue(0)
    with pytest.raises(Exception, match="still holds"):
        engine.whisper_align_paths_enqueue(0)
    assert np.array_equal(engine.whisper_align_paths_wait(0)[0], pl)


def test_forced_alignment_matches_transformers_token_timestamps(engine, ops):
    """``pce_whisper_align_run`` against vectors computed by transformers' ``_extract_token_timestamps`` and its DTW port on the
    transformers model's own cross-attentions (tests/golden/whisper_hf_align.npz, see its generating script): the cost matrix the DTW
    runs on is the golden's normalised / median-filtered / head-averaged matrix (rows ``[sot_len:-1]``, negated) within the operand
    type's rounding, and the alignment indices -- the DTW path and the token jump times -- are the golden's: identical with fp16
    operands (the reference's arithmetic), at least 95 % within one frame (20 ms) with bf16.  Both head selections, both clips."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "whisper_hf_align.npz"))
    edims = dict(n_mels=80, n_ctx=1500, n_state=128, n_head=2, n_layer=2)
    tdims = dict(n_vocab=300, n_text_ctx=96, n_state=128, n_head=2, n_layer=2)
    We, Wd = WW.synthetic_weights(edims, seed=77), WW.synthetic_decoder_weights(tdims, seed=78)
    sot_len = int(g["sot_len"][0])
    use = [synth.synth_clip(int(ci), seconds=float(sec)) for ci, sec in g["clips"]]
    engine.upload(use, 16000)
    engine.logmel_run(80)
    engine.whisper_load(edims, WW.pack(We, edims))
    engine.whisper_encode_run()
    engine.whisper_decoder_load(tdims, WW.pack_decoder(Wd, tdims))
    for name in ("upper_half", "picked"):
        keys = [f"c{int(ci)}_{name}" for ci, _ in g["clips"]]
        toks = [g[k + "_tokens"].tolist() for k in keys]
        hm = np.zeros((2, 2), dtype=bool)
        for l, h in g[keys[0] + "_heads"]:
            hm[l, h] = True
        res = engine.whisper_align(toks, [3000, 3000], sot_len, head_mask=hm, want_cost=True)
        for k, got in zip(keys, res):
            want_cost = -g[k + "_matrix"][sot_len:-1].astype(np.float64)
            assert got["cost"].shape == want_cost.shape
            rel = np.linalg.norm(got["cost"] - want_cost) / np.linalg.norm(want_cost)
            assert rel <= (8e-3 if ops["name"] != "bf16" else 4e-2), (k, ops["name"], rel)
            jumps = np.r_[True, np.diff(got["text_indices"]) > 0]
            jt = got["time_indices"][jumps] * 0.02
            wt = g[k + "_jump_times"]
            assert len(jt) == len(wt) == len(toks[keys.index(k)]) - sot_len - 1
            if ops["name"] != "bf16":
                assert np.array_equal(got["text_indices"], g[k + "_text_idx"]) and np.array_equal(got["time_indices"], g[k + "_time_idx"]), k
                assert np.array_equal(jt, wt), k
            else:
                assert np.mean(np.abs(jt - wt) <= 0.02 + 1e-9) >= 0.95, (k, float(np.mean(np.abs(jt - wt) <= 0.02 + 1e-9)))


def test_median_filter_network_equals_generic_sort(engine, clips, monkeypatch):
    """Width 7 (the default) selects its median with a 13-exchange network in registers; a context created with
    PCE_ALIGN_GENERIC_MEDIAN=1 runs the insertion sort used for every other width.  A median is a selection: the
    cost matrices must be bit-identical.  Width 5 goes through the generic path and is checked against torch."""
    from prosody_control_french_tts_amd import ProsodyEngine
    edims = dict(n_mels=80, n_ctx=1500, n_state=128, n_head=2, n_layer=2)
    tdims = dict(n_vocab=300, n_text_ctx=96, n_state=128, n_head=2, n_layer=4)
    We, Wd = WW.synthetic_weights(edims), WW.synthetic_decoder_weights(tdims)
    use = clips[:2]
    rng = np.random.default_rng(5)
    toks = [rng.integers(0, 300, size=n).tolist() for n in (20, 41)]
    num_frames = [len(c) // 160 for c in use]

    def run(eng, width):
        eng.upload(use, 16000); eng.logmel_run(80)
        eng.whisper_load(edims, WW.pack(We, edims)); eng.whisper_encode_run()
        eng.whisper_decoder_load(tdims, WW.pack_decoder(Wd, tdims))
        return eng.whisper_align(toks, num_frames, 3, medfilt_width=width, want_cost=True)

    monkeypatch.setenv("PCE_ALIGN_GENERIC_MEDIAN", "1")
    with ProsodyEngine(0) as generic:
        ref = run(generic, 7)
    monkeypatch.delenv("PCE_ALIGN_GENERIC_MEDIAN")
    got = run(engine, 7)
    for g, r in zip(got, ref):
        assert g["cost"].tobytes() == r["cost"].tobytes()
        assert np.array_equal(g["time_indices"], r["time_indices"])
    got5 = run(engine, 5)
    for i in range(2):
        enc = engine.whisper_encode_fetch(i)
        cost, _, _ = WO.find_alignment(toks[i], enc, Wd, tdims, num_frames[i], 3, medfilt_width=5)
        assert np.linalg.norm(got5[i]["cost"] - cost) / np.linalg.norm(cost) <= 3e-2


def _greedy_setup(engine):
    from tests.test_whisper_hf_crosscheck import _greedy_gold
    g, rules = _greedy_gold()
    edims = dict(n_mels=80, n_ctx=1500, n_state=128, n_head=2, n_layer=2)
    tdims = dict(n_vocab=300, n_text_ctx=96, n_state=128, n_head=2, n_layer=2)
    We, Wd = WW.synthetic_weights(edims, seed=77), WW.greedy_test_decoder_weights(tdims, seed=79)
    use = [synth.synth_clip(int(ci), seconds=4.0) for ci in g["clips"]]
    engine.upload(use, 16000)
    engine.logmel_run(80)
    engine.whisper_load(edims, WW.pack(We, edims))
    engine.whisper_encode_run()
    engine.whisper_decoder_load(tdims, WW.pack_decoder(Wd, tdims))
    return g, rules, tdims, Wd, use


def test_decode_step_decisions_match_the_restatement(engine):
    """Free-running decoding, step by step on the golden prefixes (the transformers-driven sequences of
    tests/golden/whisper_greedy_tiny.npz): the GPU step must pick the golden next token wherever the fp32 restatement's
    filtered logits leave a margin of 0.05 between the best two (bf16 operands), and never break a timestamp rule."""
    from prosody_control_french_tts_amd.Aligners import decoding as DEC
    g, rules, tdims, Wd, use = _greedy_setup(engine)
    mask = DEC.vocab_mask(tdims["n_vocab"], rules["suppress_tokens"], rules["blank_tokens"], rules["no_timestamps"])
    gold = [g[f"tokens_{int(ci)}"].tolist() for ci in g["clips"]]
    begin = len(g["initial"])
    encs = [engine.whisper_encode_fetch(i) for i in range(len(use))]
    n_checked = n_same = 0
    for L in list(range(begin, begin + 14)) + [begin + 30, begin + 44]:
        prefixes = [t[:L] for t in gold]
        nxt = engine.whisper_decode_step(prefixes, begin, rules["eot"], rules["timestamp_begin"], mask, rules["max_initial_timestamp_index"])
        for i, p in enumerate(prefixes):
            logits = WO.find_alignment(p, encs[i], Wd, tdims, 2, 0, want_internal=True)["logits"][-1]
            f = WO.apply_decoding_rules(logits, p, begin, rules)
            assert np.isfinite(f[int(nxt[i])]), (L, i, int(nxt[i]))               # never a suppressed / rule-breaking token
            lp = f - (np.max(f) + np.log(np.sum(np.exp(f[np.isfinite(f)] - np.max(f)))))       # log_softmax of the filtered logits
            assert abs(float(engine.last_decode_logprobs[i]) - float(lp[int(nxt[i])])) <= 0.05, (L, i)
            top = np.sort(f[np.isfinite(f)])[::-1]
            n_checked += 1
            if len(top) < 2 or top[0] - top[1] > 0.05:
                assert int(nxt[i]) == int(np.argmax(f)) == gold[i][L], (L, i)
                n_same += 1
    assert n_same >= 0.8 * n_checked


def test_greedy_decoding_runs_free_and_obeys_the_rules(engine):
    from prosody_control_french_tts_amd.Aligners import decoding as DEC
    g, rules, tdims, Wd, use = _greedy_setup(engine)
    out = DEC.greedy_decode(engine, tdims["n_vocab"], g["initial"].tolist(), rules, sample_len=40)
    tsb, eot = rules["timestamp_begin"], rules["eot"]
    for seq in out:
        new = seq[len(g["initial"]):]
        assert 1 <= len(new) <= 40 and new[0] >= tsb and new[0] <= tsb + rules["max_initial_timestamp_index"]
        assert not (set(new) & set(rules["suppress_tokens"])) and rules["no_timestamps"] not in new
        stamps = [t for t in new if t >= tsb]
        assert all(b >= a for a, b in zip(stamps, stamps[1:]))                     # timestamps never decrease
        for a, b, c in zip(new, new[1:], new[2:]):
            if a >= tsb and b >= tsb:
                assert c < tsb                                                    # after a pair: text (or end of text)
        assert eot not in new[:-1]
    # the first tokens agree with the transformers-driven golden sequences (margins are comfortable there)
    gold = [g[f"tokens_{int(ci)}"].tolist() for ci in g["clips"]]
    assert all(o[:len(g["initial"]) + 3] == w[:len(g["initial"]) + 3] for o, w in zip(out, gold))


def test_cached_and_uncached_decoding_agree(engine):
    """The per-step K / V cache (one new position per step) against re-running the decoder over the whole prefix
    (``no_cache``): same tokens, except where bf16 rounding decides a near tie (then the sequences part).  Prompts of
    DIFFERENT lengths per clip (``condition_on_previous_text``): the cache appends one position per sequence wherever it
    stands."""
    from prosody_control_french_tts_amd.Aligners import decoding as DEC
    g, rules, tdims, Wd, use = _greedy_setup(engine)
    init = g["initial"].tolist()
    n = len(use)
    for prompts in ([list(init) for _ in range(n)],
                    [[7, 11 + i, 13][: i % 3 + 1] * (i + 1) + list(init) for i in range(n)]):           # ragged: 3, 5 ... tokens of previous text
        begins = [len(p) for p in prompts]
        cached, lp_c, _ = DEC.decode_batch(engine, tdims["n_vocab"], prompts, begins, rules, sample_len=30)
        plain, lp_p, _ = DEC.decode_batch(engine, tdims["n_vocab"], prompts, begins, rules, sample_len=30, no_cache=True)
        agree = [next((k for k, (a, b) in enumerate(zip(x, y)) if a != b), min(len(x), len(y))) for x, y in zip(cached, plain)]
        assert min(agree) >= 6 and sum(a == min(len(x), len(y)) for a, x, y in zip(agree, cached, plain)) >= 2, agree
        for a, x, y in zip(agree, lp_c, lp_p):
            assert np.allclose(x[:a], y[:a], atol=0.05)
    # the uniform-prompt route equals the plain greedy loop of the first API
    old = DEC.greedy_decode(engine, tdims["n_vocab"], init, rules, sample_len=30)
    new, _, _ = DEC.decode_batch(engine, tdims["n_vocab"], [list(init)] * n, [len(init)] * n, rules, sample_len=30)
    for o, w in zip(old, new):
        body = o[len(init):]
        assert (body[:-1] if body and body[-1] == rules["eot"] else body) == w


def test_cached_decoding_with_long_prompts_agrees(engine):
    """Prompts of 130-230 tokens (``condition_on_previous_text`` at its longest) on a decoder with the real 448-token text context: the
    self-attention of an incremental step then takes the kernel's general path (more than 128 keys: V^T rows read in 512-key pieces by
    the lanes that hold live keys, the new key and value attended to from registers) instead of the few-key one; cached and uncached
    decoding agree as for short prompts."""
    from tests.test_whisper_hf_crosscheck import _greedy_gold
    from prosody_control_french_tts_amd.Aligners import decoding as DEC
    _, rules = _greedy_gold()
    edims = dict(n_mels=80, n_ctx=1500, n_state=128, n_head=2, n_layer=2)
    tdims = dict(n_vocab=300, n_text_ctx=448, n_state=128, n_head=2, n_layer=2)
    We, Wd = WW.synthetic_weights(edims, seed=77), WW.greedy_test_decoder_weights(tdims, seed=79)
    use = [synth.synth_clip(i, seconds=4.0) for i in range(4)]
    engine.upload(use, 16000)
    engine.logmel_run(80)
    engine.whisper_load(edims, WW.pack(We, edims))
    engine.whisper_encode_run()
    engine.whisper_decoder_load(tdims, WW.pack_decoder(Wd, tdims))
    init = _greedy_gold()[0]["initial"].tolist()
    n = len(use)
    prompts = [[(17 * k + 3 * i) % 200 + 5 for k in range(130 + 33 * (i % 4))] + list(init) for i in range(n)]
    begins = [len(p) for p in prompts]
    cached, lp_c, _ = DEC.decode_batch(engine, tdims["n_vocab"], prompts, begins, rules, sample_len=14)
    plain, lp_p, _ = DEC.decode_batch(engine, tdims["n_vocab"], prompts, begins, rules, sample_len=14, no_cache=True)
    agree = [next((k for k, (a, b) in enumerate(zip(x, y)) if a != b), min(len(x), len(y))) for x, y in zip(cached, plain)]
    assert min(agree) >= 4 and sum(a == min(len(x), len(y)) for a, x, y in zip(agree, cached, plain)) >= 2, agree
    for a, x, y in zip(agree, lp_c, lp_p):
        assert np.allclose(x[:a], y[:a], atol=0.05)


@pytest.mark.parametrize("d,heads,n,ops", [(128, 2, 5, None), (256, 4, 3, None), (384, 6, 2, None), (768, 12, 3, None), (1024, 16, 2, None), (256, 4, 3, "bf16")])
def test_cross_attention_from_the_encoder_output_against_the_kv_path_and_the_restatement(d, heads, n, ops):
    """Incremental decoding steps compute their cross-attention from the encoder output (``pce_xattn.inc``: Q' = q Wk, one pass over E,
    out = Wv U + bv) instead of from the projected K / V^T cache (``PCE_XATTN_ABSORB=0``: the round-3 kernel).  Two contexts, one per form, the
    same free-running loop (ragged prompts, a clip that is inactive from the start): the same tokens as far as the restatement's margins
    decide them, log-probabilities within 0.02 of each other, and -- step by step against the fp32 restatement's filtered log-softmax --
    the new form at least as close as the K / V form (its operands enter the MFMAs as hi + lo pairs: nothing is rounded that the
    reference's own fp16 arithmetic keeps)."""
    import os
    import prosody_control_french_tts_amd as P
    from tests.test_whisper_hf_crosscheck import _greedy_gold
    from prosody_control_french_tts_amd.Aligners import decoding as DEC
    _, rules = _greedy_gold()
    edims = dict(n_mels=80, n_ctx=1500, n_state=d, n_head=heads, n_layer=2)
    tdims = dict(n_vocab=300, n_text_ctx=96, n_state=d, n_head=heads, n_layer=2)
    We, Wd = WW.synthetic_weights(edims, seed=177), WW.greedy_test_decoder_weights(tdims, seed=179)
    use = [synth.synth_clip(40 + i, seconds=3.0 + i) for i in range(n)]
    init = _greedy_gold()[0]["initial"].tolist()
    prompts = [[7, 11 + i, 13][: i % 3 + 1] * (i + 1) + list(init) for i in range(n)]
    begins = [len(p) for p in prompts]
    active = [i != 1 for i in range(n)]
    mask = DEC.vocab_mask(tdims["n_vocab"], rules["suppress_tokens"], rules["blank_tokens"], rules["no_timestamps"])
    runs = {}
    for form in ("1", "0"):
        old = os.environ.get("PCE_XATTN_ABSORB")
        os.environ["PCE_XATTN_ABSORB"] = form
        try:
            eng = P.ProsodyEngine(0)
        finally:
            if old is None:
                os.environ.pop("PCE_XATTN_ABSORB", None)
            else:
                os.environ["PCE_XATTN_ABSORB"] = old
        try:
            if ops:
                eng.whisper_set_operands(ops)
            eng.upload(use, 16000); eng.logmel_run(80)
            eng.whisper_load(edims, WW.pack(We, edims)); eng.whisper_encode_run()
            eng.whisper_decoder_load(tdims, WW.pack_decoder(Wd, tdims))
            toks, lps, _ = DEC.decode_batch(eng, tdims["n_vocab"], prompts, begins, rules, sample_len=12, active=active)
            # teacher-forced on ONE sequence of tokens (the new form's): each step's log-probability of the token that follows
            encs = [eng.whisper_encode_fetch(i) for i in range(n)]
            runs[form] = (toks, lps, encs)
        finally:
            eng.close()
    (ta, la, encs), (tb, lb, _) = runs["1"], runs["0"]
    assert ta[1] == [] and tb[1] == []
    err = {"1": [], "0": []}
    for i in range(n):
        if not active[i]:
            continue
        agree = next((k for k, (x, y) in enumerate(zip(ta[i], tb[i])) if x != y), min(len(ta[i]), len(tb[i])))
        assert agree >= 3, (i, ta[i], tb[i])
        assert np.allclose(la[i][:agree], lb[i][:agree], atol=0.1 if ops == "bf16" else 0.02), (i, la[i][:agree], lb[i][:agree])
        seq = list(prompts[i])
        for k in range(agree):
            logits = WO.find_alignment(seq, encs[i], Wd, tdims, 2, 0, want_internal=True)["logits"][-1]
            f = WO.apply_decoding_rules(logits, seq, begins[i], rules)
            lsm = f - (np.max(f) + np.log(np.sum(np.exp(f[np.isfinite(f)] - np.max(f)))))
            err["1"].append(abs(la[i][k] - lsm[ta[i][k]])); err["0"].append(abs(lb[i][k] - lsm[ta[i][k]]))
            seq.append(ta[i][k])
    tol = 0.25 if ops == "bf16" else 0.05
    assert max(err["1"]) <= tol and max(err["0"]) <= tol, (max(err["1"]), max(err["0"]))
    # (the two forms' mean errors against the fp32 restatement are the same size: a few dozen samples of operand-rounding noise each -- 1.25 x for fp16
    # operands; 1.5 x for bf16, whose 0.004-0.005 means have been seen 1.31 apart)
    assert np.mean(err["1"]) <= (1.5 if ops == "bf16" else 1.25) * np.mean(err["0"]) + 1e-4, (np.mean(err["1"]), np.mean(err["0"]))


@pytest.mark.parametrize("n", [130, 520])
def test_frame_splits_of_the_encoder_output_cross_attention(n):
    """``k_xattn_absorbed`` runs 4 / 2 / 1 workgroups per clip (fewer than 256 / 512 clips, more; always the same four leaves of frames): the 4- and the
    1-workgroup launches at batch sizes the other tests do not reach, against the K / V-form kernels on the same batch -- the same tokens, log-probabilities
    within 0.02."""
    import os
    import prosody_control_french_tts_amd as P
    from tests.test_whisper_hf_crosscheck import _greedy_gold
    from prosody_control_french_tts_amd.Aligners import decoding as DEC
    _, rules = _greedy_gold()
    edims = dict(n_mels=80, n_ctx=1500, n_state=128, n_head=2, n_layer=1)
    tdims = dict(n_vocab=300, n_text_ctx=96, n_state=128, n_head=2, n_layer=2)
    We, Wd = WW.synthetic_weights(edims, seed=277), WW.greedy_test_decoder_weights(tdims, seed=279)
    base = [synth.synth_clip(80 + i, seconds=1.0 + 0.25 * (i % 5)) for i in range(13)]
    use = [base[i % 13] for i in range(n)]
    init = _greedy_gold()[0]["initial"].tolist()
    runs = {}
    for form in ("1", "0"):
        old = {k: os.environ.get(k) for k in ("PCE_XATTN_ABSORB", "PCE_SELF_ROWS")}
        os.environ["PCE_XATTN_ABSORB"] = form; os.environ["PCE_SELF_ROWS"] = form
        try:
            eng = P.ProsodyEngine(0)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        try:
            eng.upload(use, 16000); eng.logmel_run(80)
            eng.whisper_load(edims, WW.pack(We, edims)); eng.whisper_encode_run()
            eng.whisper_decoder_load(tdims, WW.pack_decoder(Wd, tdims))
            runs[form] = DEC.decode_batch(eng, tdims["n_vocab"], [list(init)] * n, [len(init)] * n, rules, sample_len=8)[:2]
        finally:
            eng.close()
    (ta, la), (tb, lb) = runs["1"], runs["0"]
    same = 0
    for i in range(n):
        agree = next((k for k, (x, y) in enumerate(zip(ta[i], tb[i])) if x != y), min(len(ta[i]), len(tb[i])))
        assert agree >= 3, (i, ta[i], tb[i])
        assert np.allclose(la[i][:agree], lb[i][:agree], atol=0.02), (i, la[i][:agree], lb[i][:agree])
        same += ta[i] == tb[i]
        assert ta[i] == ta[i % 13] and np.allclose(la[i], la[i % 13], atol=1e-6)      # the same recording decodes the same wherever it stands in the batch
    assert same >= 0.9 * n


def test_a_clips_decoding_does_not_depend_on_the_batch_size():
    """include/pce.h, minor 2: "a clip's Whisper results no longer depend on what it is batched with".  The encoder-output cross-attention runs 4 / 2 / 1
    workgroups per clip by batch size (fewer than 256 / 512 clips, more; 8 / 4 / 2 / 1 in round 5); until round 6 each workgroup ran its own online softmax over its share of the
    frames, so the partition -- and with it a clip's bits -- followed from the batch size.  Now the frames are always cut into the same four leaves
    and merged in leaf order.  The same 13 recordings decoded alone, in a batch of 130, of 260 and of 520 (every workgroup count): IDENTICAL tokens and
    log-probabilities, bit for bit (the next test forces the workgroup count at a fixed batch size)."""
    import prosody_control_french_tts_amd as P
    from tests.test_whisper_hf_crosscheck import _greedy_gold
    from prosody_control_french_tts_amd.Aligners import decoding as DEC
    _, rules = _greedy_gold()
    edims = dict(n_mels=80, n_ctx=1500, n_state=128, n_head=2, n_layer=1)
    tdims = dict(n_vocab=300, n_text_ctx=96, n_state=128, n_head=2, n_layer=2)
    We, Wd = WW.synthetic_weights(edims, seed=277), WW.greedy_test_decoder_weights(tdims, seed=279)
    base = [synth.synth_clip(80 + i, seconds=1.0 + 0.25 * (i % 5)) for i in range(13)]
    init = _greedy_gold()[0]["initial"].tolist()

    def decode(n):
        eng = P.ProsodyEngine(0)
        try:
            eng.upload([base[i % 13] for i in range(n)], 16000); eng.logmel_run(80)
            eng.whisper_load(edims, WW.pack(We, edims)); eng.whisper_encode_run()
            eng.whisper_decoder_load(tdims, WW.pack_decoder(Wd, tdims))
            return DEC.decode_batch(eng, tdims["n_vocab"], [list(init)] * n, [len(init)] * n, rules, sample_len=8)[:2]
        finally:
            eng.close()

    ref_t, ref_l = decode(13)                                # 4 workgroups per clip
    assert len({np.asarray(x, dtype=np.float32).tobytes() for x in ref_l}) > 6      # (the recordings really decode differently: their log-probabilities do)
    for n in (1, 130, 260, 520):                             # 4, 4, 2, 1 workgroups per clip
        t, l = decode(n)
        for i in range(n):
            assert t[i] == ref_t[i % 13], (n, i, t[i], ref_t[i % 13])
            assert np.asarray(l[i], dtype=np.float32).tobytes() == np.asarray(ref_l[i % 13], dtype=np.float32).tobytes(), (n, i, l[i], ref_l[i % 13])


@pytest.mark.parametrize("d,heads", [(128, 2), (768, 12), (1024, 16)])
def test_cross_attention_bits_do_not_depend_on_the_workgroups_per_clip(d, heads, tmp_path):
    """The same decode with 4, 2 and 1 workgroups per clip forced at one batch size (``PCE_XATTN_WPC``, read once per process: fresh processes), at the
    head sizes of Whisper tiny-like, small and medium: tokens and log-probabilities bit-identical."""
    import subprocess, sys
    script = tmp_path / "wpc.py"
    script.write_text(f"""
import sys, hashlib
import numpy as np
sys.path.insert(0, {ROOT!r})
import prosody_control_french_tts_amd as P
from prosody_control_french_tts_amd import synth, whisper_weights as WW
from prosody_control_french_tts_amd.Aligners import decoding as DEC
from tests.test_whisper_hf_crosscheck import _greedy_gold
_, rules = _greedy_gold()
init = _greedy_gold()[0]["initial"].tolist()
edims = dict(n_mels=80, n_ctx=1500, n_state={d}, n_head={heads}, n_layer=1)
tdims = dict(n_vocab=300, n_text_ctx=96, n_state={d}, n_head={heads}, n_layer=2)
We, Wd = WW.synthetic_weights(edims, seed=31), WW.greedy_test_decoder_weights(tdims, seed=33)
use = [synth.synth_clip(60 + i, seconds=1.5 + 0.5 * (i % 3)) for i in range(5)]
with P.ProsodyEngine(0) as eng:
    eng.upload(use, 16000); eng.logmel_run(80)
    eng.whisper_load(edims, WW.pack(We, edims)); eng.whisper_encode_run()
    eng.whisper_decoder_load(tdims, WW.pack_decoder(Wd, tdims))
    t, l = DEC.decode_batch(eng, 300, [list(init)] * 5, [len(init)] * 5, rules, sample_len=6)[:2]
h = hashlib.sha1()
for a, b in zip(t, l):
    h.update(np.asarray(a, dtype=np.int64).tobytes()); h.update(np.asarray(b, dtype=np.float32).tobytes())
print("HASH", h.hexdigest(), [len(a) for a in t])
""")
    seen = {}
    for wpc in (4, 2, 1):
        env = dict(os.environ, PCE_XATTN_WPC=str(wpc))
        r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        seen[wpc] = [ln for ln in r.stdout.splitlines() if ln.startswith("HASH")][0]
    assert len(set(seen.values())) == 1, seen


@pytest.mark.parametrize("d", [128, 256, 384, 512, 768, 1024])
def test_encoder_output_cross_attention_against_a_float64_restatement(engine, d):
    """One layer of the cross-attention of a decoding step (``k_xq_fused`` -> ``k_xattn_absorbed`` -> ``k_uv_absorb``) on its own, for every width it is
    built for and every workgroup count per clip (the decode-level tests allow 0.02 in a log-probability; a systematic error of a few 1e-3 in
    one head -- the unrounded K / V, a hi / lo split mistake -- would pass them).  The restatement is float64 on the values the kernels really multiply
    (16-bit weights and E; the LayerNorm output and the query rounded to 16 bits where the reference's fp16 Linear rounds them): what is left is the
    rounding of Q' = q Wk, of the probabilities and of U to hi + lo pairs (2^-22) and of the OUTPUT to 16 bits, so the bound is one 16-bit step of the
    output plus 1e-3 of its scale; the bits must not depend on the workgroup count (1, 2, 4) nor on the batch (the first clips alone)."""
    engine.whisper_set_operands("fp16")
    heads = d // 64
    rng = np.random.default_rng(1000 + d)
    n, k_cap = 5, 1500
    k_len = np.array([1500, 1499, 700, 33, 1], dtype=np.int32)
    resid = rng.standard_normal((n, d)).astype(np.float32) * 1.5 + 0.2
    ln_w = (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32); ln_b = (0.05 * rng.standard_normal(d)).astype(np.float32)
    sc = 1.0 / np.sqrt(d)
    wq, wk, wv = (rng.standard_normal((d, d)).astype(np.float32) * sc * g for g in (1.6, 1.6, 1.0))
    bq, bv = (0.1 * rng.standard_normal(d)).astype(np.float32), (0.1 * rng.standard_normal(d)).astype(np.float32)
    E = rng.standard_normal((n, k_cap, d)).astype(np.float32)
    E[:, :, : d // 2] += rng.standard_normal((n, 1, d // 2)).astype(np.float32)          # (a per-clip offset: the attention is not uniform)
    outs = {}
    for wpc in (4, 2, 1, 0):
        outs[wpc], rq, rk, rv, rE = engine.selftest_xattn(resid, ln_w, ln_b, wq, bq, wk, wv, bv, E, k_len, heads, wpc)
    for wpc in (2, 1, 0):
        assert outs[wpc].tobytes() == outs[4].tobytes(), wpc
    solo = engine.selftest_xattn(resid[:2], ln_w, ln_b, wq, bq, wk, wv, bv, E[:2], k_len[:2], heads, 0)[0]
    assert solo.tobytes() == outs[4][:2].tobytes()
    # float64 restatement
    r16 = lambda x: x.astype(np.float16).astype(np.float64)
    x = resid.astype(np.float64)
    mu = x.mean(axis=1, keepdims=True); var = ((x - mu) ** 2).mean(axis=1, keepdims=True)
    ln = r16((x - mu) / np.sqrt(var + 1e-5) * ln_w + ln_b)                                # the kernel parks the LayerNorm output as op_t
    q = r16(ln @ rq.astype(np.float64).T + bq)                                            # rounded as the reference's fp16 Linear does
    want = np.zeros((n, d))
    for c in range(n):
        Ec = rE[c, : k_len[c]].astype(np.float64)
        for h in range(heads):
            sl = slice(64 * h, 64 * h + 64)
            kh = Ec @ rk[sl].astype(np.float64).T                                         # [t][64]   (never rounded: the kernels fold Wk into the query)
            s_ = (kh @ q[c, sl]) * 0.125
            p = np.exp(s_ - s_.max()); p /= p.sum()
            u = p @ Ec                                                                    # [d]
            want[c, sl] = rv[sl].astype(np.float64) @ u + bv[sl]
    got = outs[4].astype(np.float64)
    scale = np.abs(want).max()
    err = np.abs(got - want)
    assert np.all(err <= np.abs(want) * 2.0 ** -10 + 1.0e-3 * scale), (d, float(err.max()), float(scale))
    assert np.sqrt(np.mean(err ** 2)) <= 4.0e-4 * scale, (d, float(np.sqrt(np.mean(err ** 2))), float(scale))
    engine.whisper_set_operands("fp16-resid16")


def test_one_context_decodes_models_of_different_widths_in_turn(engine):
    """``k_xattn_absorbed<d, slots>`` is one function per width, each with its own dynamic-LDS attribute (68 KB at d = 1024: above the default
    limit): a context that has decoded with one width must still be able to decode with another (the attribute is tracked per instantiation)."""
    from tests.test_whisper_hf_crosscheck import _greedy_gold
    from prosody_control_french_tts_amd.Aligners import decoding as DEC
    _, rules = _greedy_gold()
    init = _greedy_gold()[0]["initial"].tolist()
    use = [synth.synth_clip(60 + i, seconds=2.0) for i in range(2)]
    for d, heads in ((128, 2), (1024, 16), (256, 4), (768, 12)):
        edims = dict(n_mels=80, n_ctx=1500, n_state=d, n_head=heads, n_layer=1)
        tdims = dict(n_vocab=300, n_text_ctx=96, n_state=d, n_head=heads, n_layer=1)
        engine.upload(use, 16000); engine.logmel_run(80)
        engine.whisper_load(edims, WW.pack(WW.synthetic_weights(edims, seed=5), edims)); engine.whisper_encode_run()
        engine.whisper_decoder_load(tdims, WW.pack_decoder(WW.greedy_test_decoder_weights(tdims, seed=6), tdims))
        toks, lps, _ = DEC.decode_batch(engine, tdims["n_vocab"], [list(init)] * 2, [len(init)] * 2, rules, sample_len=5)
        assert all(1 <= len(t) <= 5 for t in toks) and all(np.isfinite(l).all() for l in lps)


def test_logmel_windows_of_a_long_recording(engine):
    """Recordings longer than 30 s (segment_ph6 of the demo data runs 37.2 s): the window whisper.transcribe takes at a
    seek position is a slice of the log-mel of the WHOLE recording, clamped with the global maximum."""
    rng = np.random.default_rng(12)
    t = np.arange(16000 * 37) / 16000.0
    long_clip = np.round(6000 * np.sin(2 * np.pi * (200 + 3 * t) * t) * (np.sin(2 * np.pi * 0.4 * t) > -0.3) + 200 * rng.standard_normal(len(t))).astype(np.int16)
    long_clip[16000 * 33:] //= 40                                   # a quiet tail: its frames sit below the global clamp
    short = synth.synth_clip(2, seconds=4.0)
    clips = [long_clip, short, long_clip]
    engine.upload(clips, 16000)
    starts = [1500, 0, 3000]
    engine.logmel_run_at(80, starts)
    for i, (c, s0) in enumerate(zip(clips, starts)):
        got = engine.logmel_fetch(i)
        want = WO.log_mel_window(c, s0, 80)
        assert got.shape == want.shape == (80, 3000)
        assert np.max(np.abs(got - want)) <= 2e-3, (i, float(np.max(np.abs(got - want))))
    engine.logmel_run_at(80, [0, 0, len(long_clip) // 160])           # a window that starts at the very end: all padding
    assert np.max(np.abs(engine.logmel_fetch(2) - WO.log_mel_window(long_clip, len(long_clip) // 160, 80))) <= 2e-3
    assert np.max(np.abs(engine.logmel_fetch(1) - WO.log_mel(short, 80))) <= 2e-3        # <= 30 s at start 0: the plain front end
    from prosody_control_french_tts_amd import PceError
    with pytest.raises(PceError):
        engine.logmel_run_at(80, [0, 500, 0])                         # past the end of the 4 s clip


def test_windowed_transcription_loop_terminates_and_advances(engine):
    """Aligners.decoding.transcribe_tokens on a 37 s and a 4 s recording (tiny random-init model): every window is decoded
    at its seek position, the seek only moves forward, segments carry absolute times within the recording."""
    import warnings
    from prosody_control_french_tts_amd.Aligners import decoding as DEC
    g, rules, tdims, Wd, _ = _greedy_setup(engine)
    # The golden's 300-token vocabulary has 50 timestamp tokens (0 .. 0.98 s) and a random-init decoder closes a segment on an early one in
    # every window, so the seek crawls and the 37 s recording used to end on the max_windows cap.  Here the timestamp rows of the (tied)
    # embedding are scaled down and the text rows up (checked on the CPU restatement: only the forced initial timestamp appears), a window
    # has no consecutive timestamp pair and is consumed whole (whisper.transcribe's `seek += segment_size`), and the loop leaves through
    # its natural exit: end of audio on every clip.
    Wd = dict(Wd); emb = Wd["token_embedding.weight"].copy(); emb[rules["timestamp_begin"]:] *= 0.02; emb[:rules["eot"]] *= 4.0
    Wd["token_embedding.weight"] = emb
    engine.whisper_decoder_load(tdims, WW.pack_decoder(Wd, tdims))
    rng = np.random.default_rng(3)
    t = np.arange(16000 * 37) / 16000.0
    long_clip = np.round(5000 * np.sin(2 * np.pi * (180 + 2 * t) * t) + 300 * rng.standard_normal(len(t))).astype(np.int16)
    clips = [long_clip, synth.synth_clip(4, seconds=4.0)]
    engine.upload(clips, 16000)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)                 # "stopped after max_windows with audio left" fails the test
        segs = DEC.transcribe_tokens(engine, 80, tdims["n_vocab"], g["initial"].tolist(), rules, sample_len=30, max_windows=8)
    assert len(segs) == 2 and len(segs[0]) >= 2 and len(segs[1]) >= 1
    for i, c in enumerate(clips):
        starts = [s["start"] for s in segs[i]]
        assert all(b >= a for a, b in zip(starts, starts[1:])) and starts[0] >= 0.0
        assert all(s["end"] >= s["start"] for s in segs[i]) and segs[i][-1]["start"] <= len(c) / 16000.0
    assert segs[0][-1]["end"] > 30.0                                   # the long recording was decoded past the first window


def test_alignment_reads_the_cross_kv_a_decoding_step_left(engine):
    """pce_whisper_align_run projects the encoder output to cross-attention K / V itself unless a decoding step has already
    done so for this encoded batch: the two routes must give the same cost matrices bit for bit."""
    from prosody_control_french_tts_amd.Aligners import decoding as DEC
    g, rules, tdims, Wd, use = _greedy_setup(engine)
    toks = [g[f"tokens_{int(ci)}"].tolist()[:20] for ci in g["clips"]]
    frames = [len(c) // 160 for c in use]
    alone = engine.whisper_align(toks, frames, 3, want_cost=True)
    mask = DEC.vocab_mask(tdims["n_vocab"], rules["suppress_tokens"], rules["blank_tokens"], rules["no_timestamps"])
    engine.whisper_decode_step([t[:5] for t in toks], 3, rules["eot"], rules["timestamp_begin"], mask, rules["max_initial_timestamp_index"])
    after = engine.whisper_align(toks, frames, 3, want_cost=True)
    for a, b in zip(alone, after):
        assert a["cost"].tobytes() == b["cost"].tobytes() and np.array_equal(a["time_indices"], b["time_indices"])


# ---------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[2] ("C3") at its real size: Whisper-small, 12 + 12 layers, and the 256-clip batch
# ---------------------------------------------------------------------------------------------------------------
FULL_DEPTH_BOUNDS = {   # measured at 12 + 12 layers (tools/operand_precision.py, profiles/r03): fp16 4.6e-4 / 1.4e-3 / 100 %; bf16 3.7e-3 / 1.1e-2 / 90 % (92 % within a frame)
    "fp16": dict(enc_l2=1.5e-3, enc_max=1e-2, cost_l2=5e-3, identical=0.97, within1=0.99),
    "bf16": dict(enc_l2=8e-3, enc_max=8e-2, cost_l2=3e-2, identical=0.75, within1=0.85),
    # round 4 (profiles/r04/operand_precision.json): 1.04e-3 / 2.5e-3 / 100 % with the fp16 residual stream
    "fp16-resid16": dict(enc_l2=2.5e-3, enc_max=2.5e-2, cost_l2=8e-3, identical=0.97, within1=0.99)}


def test_c3_whisper_small_full_depth_matches_the_restatement(engine, ops):
    """The 12-layer Whisper-small encoder and the 12-layer teacher-forced decoder / alignment on 4 ten-second clips against
    the float32 restatement, per operand type, plus how the rounding error grows with depth (the same weights truncated to
    2 / 6 / 12 layers).  Bounds (FULL_DEPTH_BOUNDS, about 3 x what is observed): fp16 operands -- encoder relative L2 <= 1.5e-3,
    alignment cost matrix <= 5e-3, >= 97 % of the word-boundary frames IDENTICAL to the fp32 path; bf16 operands -- 8e-3, 3e-2,
    >= 85 % within one 20 ms frame.  In both the GPU path is EXACTLY the recurrence's path on the GPU's own cost matrix."""
    B = FULL_DEPTH_BOUNDS[ops["name"]]
    dims, tdims = WW.DIMS["small"], dict(WW.TEXT_DIMS["small"], n_vocab=2048)        # (a 51865-row embedding adds nothing to this check)
    W, Wd = WW.synthetic_weights(dims), WW.synthetic_decoder_weights(tdims)
    clips4 = [synth.synth_clip(40 + i, seconds=10.0) for i in range(4)]
    engine.upload(clips4, 16000)
    growth = {}
    for depth in (2, 6, 12):
        dd = dict(dims, n_layer=depth)
        engine.logmel_run(80)
        engine.whisper_load(dd, WW.pack(W, dd))
        engine.whisper_encode_run()
        errs = []
        for i in range(4 if depth == 12 else 1):
            got = engine.whisper_encode_fetch(i)
            want = WO.encoder_forward(WO.log_mel(clips4[i], 80), W, dd)
            errs.append(float(np.linalg.norm(got - want) / np.linalg.norm(want)))
            assert np.max(np.abs(got - want)) <= B["enc_max"] * max(1.0, float(np.std(want))), (depth, i)
        growth[depth] = max(errs)
        assert growth[depth] <= B["enc_l2"], growth
    print("encoder relative L2 error by depth:", growth)
    assert growth[12] <= 4 * growth[2] + B["enc_l2"] / 4           # grows slowly with depth, does not blow up
    # forced alignment at full depth on the 12-layer encoder output
    engine.whisper_decoder_load(tdims, WW.pack_decoder(Wd, tdims))
    rng = np.random.default_rng(33)
    toks = [rng.integers(0, tdims["n_vocab"], size=int(n)).tolist() for n in (24, 31, 40, 47)]
    frames = [len(c) // 160 for c in clips4]
    res = engine.whisper_align(toks, frames, 3, want_cost=True)
    same = near = total = 0
    for i in range(4):
        enc = engine.whisper_encode_fetch(i)
        cost, ti, tj = WO.find_alignment(toks[i], enc, Wd, tdims, frames[i], 3)
        got = res[i]
        assert got["cost"].shape == cost.shape
        assert np.linalg.norm(got["cost"] - cost) / np.linalg.norm(cost) <= B["cost_l2"], i
        wi, wj = WO.dtw_path(got["cost"])
        assert np.array_equal(got["text_indices"], wi) and np.array_equal(got["time_indices"], wj)
        jumps_g = got["time_indices"][np.r_[True, np.diff(got["text_indices"]) > 0]]
        jumps_w = tj[np.r_[True, np.diff(ti) > 0]]
        assert len(jumps_g) == len(jumps_w), i
        same += int(np.sum(jumps_g == jumps_w)); near += int(np.sum(np.abs(jumps_g - jumps_w) <= 1)); total += len(jumps_g)
    assert same >= B["identical"] * total and near >= B["within1"] * total, (same, near, total)


def test_c3_batch_of_256_is_clip_independent(engine):
    """The C3 batch (256 clips x 10 s, Whisper-small, full depth): a clip's encoder output and DTW path inside the batch are
    bit-identical to the same clip processed in a batch of 4 -- no tile, supertile or XCD mapping leaks between clips."""
    import hashlib
    dims, tdims = WW.DIMS["small"], dict(WW.TEXT_DIMS["small"], n_vocab=2048)
    W, Wd = WW.synthetic_weights(dims), WW.synthetic_decoder_weights(tdims)
    n = 256
    clips = synth.synth_batch(n, 10.0, 16000, first=0)
    rng = np.random.default_rng(8)
    toks = [rng.integers(0, tdims["n_vocab"], size=int(rng.integers(24, 48))).tolist() for _ in range(n)]
    frames = [len(c) // 160 for c in clips]
    engine.whisper_load(dims, WW.pack(W, dims))
    engine.whisper_decoder_load(tdims, WW.pack_decoder(Wd, tdims))
    pick = [0, 97, 200, 255]

    def run(idx):
        engine.upload([clips[i] for i in idx], 16000)
        engine.logmel_run(80)
        engine.whisper_encode_run()
        paths = engine.whisper_align([toks[i] for i in idx], [frames[i] for i in idx], 3)
        return paths

    paths_all = run(list(range(n)))
    h_all = {i: hashlib.sha256(engine.whisper_encode_fetch(i).tobytes()).hexdigest() for i in pick}
    assert np.isfinite(engine.whisper_encode_fetch(128)).all()
    paths_4 = run(pick)
    for k, i in enumerate(pick):
        assert hashlib.sha256(engine.whisper_encode_fetch(k).tobytes()).hexdigest() == h_all[i], i
        assert np.array_equal(paths_4[k]["text_indices"], paths_all[i]["text_indices"])
        assert np.array_equal(paths_4[k]["time_indices"], paths_all[i]["time_indices"])
        assert paths_all[i]["text_indices"][-1] == len(toks[i]) - 3 - 2 and paths_all[i]["time_indices"][-1] == frames[i] // 2 - 1


@pytest.mark.parametrize("shape", [(3000, 1536, 768, 0), (3000, 3072, 768, 1), (6000, 768, 3072, 0), (3000, 768, 768, 2), (4100, 256, 64, 0),
                                   (2100, 5120, 128, 0)])             # (the widest bias vector after narrower launches: Whisper-large fc1)
def test_persistent_256_gemm_against_torch(engine, ops, shape):
    """The persistent 256 x 256 GEMM of the big encoder projections (pce_gemm256.inc) alone, through its self-test entry
    point: C = epilogue(A B^T + bias) against torch fp32 on the bf16-rounded operands.  Tolerance: the bf16 rounding of
    the OUTPUT (relative 2^-8 per element; 4e-3 relative L2 over the matrix, 2^-7 |x| + 1e-2 per element).  Shapes: the
    four encoder projections of Whisper-small (bias, bias + exact GELU, K = 3072, transposed-V image with the key axis
    padded to 1536), ragged M (last row tile partial) and the smallest legal N, K."""
    import torch
    M, N, K, epi = shape
    rng = np.random.default_rng(M + N + K + epi)
    A = rng.standard_normal((M, K)).astype(np.float32); B = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    dt = getattr(torch, ops["torch"])
    a = torch.from_numpy(A).to(dt).float(); b = torch.from_numpy(B).to(dt).float()
    want = a @ b.T + torch.from_numpy(bias)
    if epi == 1: want = torch.nn.functional.gelu(want)
    want = want.numpy()
    S = 1500 if epi == 2 else 1
    got = engine.selftest_gemm(A, B, bias, epi, S, 1536)
    if epi == 2:
        assert got.shape == (M // S, N, 1536)
        assert not got[:, :, S:].any()                                 # the padding of the key axis is never written
        got = np.concatenate([got[c, :, :S].T for c in range(M // S)])
    assert np.isfinite(got).all()
    err = np.abs(got - want)
    assert np.linalg.norm(err) / np.linalg.norm(want) <= ops["l2"]
    assert (err <= np.abs(want) * ops["rel"] + ops["abs"]).all(), np.argwhere(err > np.abs(want) * ops["rel"] + ops["abs"])[:8]


@pytest.mark.parametrize("shape", [(3000, 2304, 768, 1536), (4500, 1536, 768, 768), (6000, 512, 128, 256)])
def test_persistent_256_gemm_split_launch(engine, ops, shape):
    """Round 3: ONE launch for Q | K | V (N = 3 d, the last d columns as the transposed image) and for the cross-attention K | V
    (N = 2 d): the row-major part and the V^T image both equal what the two separate launches (epilogues 0 and 2) write, BIT FOR BIT
    (same fragments, same summation order: the transposed tiles are computed as the transposed problem), and match torch fp32 on the
    bf16-rounded operands.  M = 4 500 / 6 000: tiles that straddle a clip boundary (1 500 rows per clip), also in the middle of a
    lane's eight key positions."""
    import torch
    M, N, K, split = shape
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32); B = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    S = 1500
    rm, vt = engine.selftest_gemm(A, B, bias, split, S, 1536)
    assert rm.shape == (M, split) and vt.shape == (M // S, N - split, 1536)
    assert np.array_equal(rm, engine.selftest_gemm(A, B[:split], bias[:split], 0))
    assert np.array_equal(vt, engine.selftest_gemm(A, B[split:], bias[split:], 2, S, 1536))
    assert not vt[:, :, S:].any()
    dt = getattr(torch, ops["torch"])
    a = torch.from_numpy(A).to(dt).float(); b = torch.from_numpy(B).to(dt).float()
    want = (a @ b.T + torch.from_numpy(bias)).numpy()
    got = np.concatenate([rm, np.concatenate([vt[c, :, :S].T for c in range(M // S)])], axis=1)
    err = np.abs(got - want)
    assert np.linalg.norm(err) / np.linalg.norm(want) <= ops["l2"]
    assert (err <= np.abs(want) * ops["rel"] + ops["abs"]).all()


@pytest.mark.parametrize("shape", [(256, 768, 768), (256, 3072, 768), (256, 768, 3072), (200, 2304, 768), (3, 768, 768)])
def test_few_row_gemm_against_torch(engine, ops, shape):
    """k_gemm_skinny (the projections of an incremental decoding step: M = sequences, one row each) through the self-test entry point:
    bias, bias + GELU (16-bit outputs) and accumulate-into-fp32 epilogues against torch fp32 on the rounded operands, at the shapes of a
    Whisper-small decoder layer, a ragged row count and a three-sequence batch."""
    import torch
    M, N, K = shape
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32); B = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    dt = getattr(torch, ops["torch"])
    a = torch.from_numpy(A).to(dt).float(); b = torch.from_numpy(B).to(dt).float()
    want = a @ b.T + torch.from_numpy(bias)
    for epi, ref in ((16, want), (17, torch.nn.functional.gelu(want)), (19, want)):
        got = engine.selftest_gemm(A, B, bias, epi)
        err = np.abs(got - ref.numpy())
        if epi == 19:
            assert np.max(err) <= 2e-4                                   # fp32 out: only the summation order differs from torch
        else:
            assert np.linalg.norm(err) / np.linalg.norm(ref.numpy()) <= ops["l2"] and (err <= np.abs(ref.numpy()) * ops["rel"] + ops["abs"]).all()


def test_persistent_256_gemm_is_deterministic_and_row_count_independent(engine):
    """Each output element is one lane's fixed-order sum: a row block gives the same BITS whether 6 000 or 96 000 rows are in
    the product (other tile order, other workgroup, other ring phase), run after run.  (Two scheduling bugs showed up as
    exactly this kind of difference: a store overtaken by the next conversion, and the first K-step of a workgroup's first tile
    read before it had landed.)"""
    rng = np.random.default_rng(0)
    for (N, K, epi) in [(768, 768, 0), (768, 3072, 0), (1536, 768, 1), (768, 768, 2)]:
        M1, M2 = 6000, 96000
        A = rng.standard_normal((M2, K), dtype=np.float32); B = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        small = engine.selftest_gemm(A[:M1], B, bias, epi, 1500, 1536)
        head = (lambda x: x[:4]) if epi == 2 else (lambda x: x[:M1])
        for _ in range(3):
            assert np.array_equal(head(engine.selftest_gemm(A, B, bias, epi, 1500, 1536)), small), (N, K, epi)
        assert np.array_equal(engine.selftest_gemm(A[:M1], B, bias, epi, 1500, 1536), small)


def _attention_reference(q, k, v, causal, dtype="bfloat16"):
    """torch fp32 softmax(q k^T / 8) v per (clip, head) on the operands rounded to the engine's 16-bit type."""
    import torch
    tq, tk, tv = (torch.from_numpy(x).to(getattr(torch, dtype)).float() for x in (q, k, v))
    clips, q_len, hd = tq.shape
    heads = hd // 64
    tq, tk, tv = (x.view(clips, -1, heads, 64).transpose(1, 2) for x in (tq, tk, tv))
    s = tq @ tk.transpose(-1, -2) / 8.0
    if causal:
        s = s + torch.full((q_len, tk.shape[2]), float("-inf")).triu(1)
    return (s.softmax(-1) @ tv).transpose(1, 2).reshape(clips, q_len, hd).numpy()


@pytest.mark.parametrize("shape", [(2, 2, 1500, 1500, False), (3, 1, 77, 77, True), (2, 2, 40, 1500, False), (1, 1, 130, 65, False), (1, 1, 1, 1, True)])
@pytest.mark.parametrize("mode", [0, 1])
def test_attention_kernel_against_torch(engine, ops, shape, mode):
    """The attention kernel alone (self-test entry point): encoder shape, causal decoder prefix, decoder-over-audio cross attention,
    ragged lengths (a key tile with one key, a query block with two queries), a single token.  Modes: the kernel as the engine runs it
    (softmax reference fixed after the first key tile) and its exact running-maximum path.  Tolerance: P and the
    output are bf16 (2^-8 relative), the sums fp32: 1e-2 absolute on outputs that are convex combinations of unit-variance values."""
    clips, heads, q_len, k_len, causal = shape
    rng = np.random.default_rng(q_len * 7 + k_len)
    q = rng.standard_normal((clips, q_len, heads * 64)).astype(np.float32) * 1.5
    k = rng.standard_normal((clips, k_len, heads * 64)).astype(np.float32) * 1.5
    v = rng.standard_normal((clips, k_len, heads * 64)).astype(np.float32)
    got, fell_back = engine.selftest_attention(q, k, v, causal, mode)
    want = _attention_reference(q, k, v, causal, ops["torch"])
    assert np.isfinite(got).all() and fell_back == 0
    assert np.max(np.abs(got - want)) <= ops["attn_abs"], np.max(np.abs(got - want))
    assert np.linalg.norm(got - want) / np.linalg.norm(want) <= ops["attn_l2"]


def test_attention_fixed_reference_overflow_takes_the_exact_path(engine, ops):
    """Scores that outgrow the first key tile's maximum by more than the operand type's exponent range (bf16: 2^127 after the log2(e)/8
    scale; fp16: 2^20 with the reference placed 4 octaves above that maximum): the fast path sees a non-finite row sum, the workgroup
    runs again with the running maximum, and the result is the exact softmax (here: one dominant key per query, so the output is that
    key's value row).  Moderately larger scores (bf16: 2^46, fp16: 2^12) must NOT fall back."""
    rng = np.random.default_rng(5)
    q_len = k_len = 320
    q = rng.standard_normal((1, q_len, 64)).astype(np.float32)
    k = rng.standard_normal((1, k_len, 64)).astype(np.float32) * 0.1
    v = rng.standard_normal((1, k_len, 64)).astype(np.float32)
    cases = ((10.0, False), (400.0, True)) if ops["name"] == "bf16" else ((2.5, False), (400.0, True))   # the loved key scores about N(0, 8 boost): 2^46 / 2^1800 at 3 sigma
    for boost, expect_fallback in cases:
        k2 = k.copy()
        k2[0, 200] = np.sign(q[0].mean(axis=0)) * boost         # a key in the FOURTH tile that every query with a positive projection loves
        got, fell_back = engine.selftest_attention(q, k2, v, False, 0)
        want = _attention_reference(q, k2, v, False, ops["torch"])
        assert np.isfinite(got).all()
        assert (fell_back > 0) == expect_fallback, (boost, fell_back)
        assert np.max(np.abs(got - want)) <= 2e-2, (boost, np.max(np.abs(got - want)))
        exact, _ = engine.selftest_attention(q, k2, v, False, 1)
        assert np.max(np.abs(exact - want)) <= 2e-2


# ---------------------------------------------------------------------------------------------------------------
# fp16 operands: the reference's own arithmetic (openai-whisper fp16=True), same kernels, same MFMA rate
# ---------------------------------------------------------------------------------------------------------------
def test_fp16_operands_full_depth_encoder_and_alignment(engine):
    """Whisper-small at its 12 + 12 layers on fp16 operands against the float32 restatement: encoder output relative L2 <= 1.5e-3
    (bf16 operands: 3.3e-3 observed, bound 2e-2), alignment cost <= 1e-2, the DTW path IS the recurrence's path on the engine's own
    cost matrix, and the bf16 build still gives its own answer afterwards (separate state per operand type)."""
    eng = engine
    eng.whisper_set_operands("fp16")                            # (fp16 operands, fp32 residual stream)
    dims, tdims = WW.DIMS["small"], dict(WW.TEXT_DIMS["small"], n_vocab=2048)
    W, Wd = WW.synthetic_weights(dims), WW.synthetic_decoder_weights(tdims)
    clips2 = [synth.synth_clip(40 + i, seconds=10.0) for i in range(2)]
    eng.upload(clips2, 16000)
    eng.logmel_run(80)
    eng.whisper_load(dims, WW.pack(W, dims))
    eng.whisper_encode_run()
    errs = []
    for i in range(2):
        got = eng.whisper_encode_fetch(i)
        want = WO.encoder_forward(WO.log_mel(clips2[i], 80), W, dims)
        errs.append(float(np.linalg.norm(got - want) / np.linalg.norm(want)))
    print("fp16 encoder relative L2:", errs)
    assert max(errs) <= 1.5e-3, errs
    eng.whisper_decoder_load(tdims, WW.pack_decoder(Wd, tdims))
    rng = np.random.default_rng(33)
    toks = [rng.integers(0, tdims["n_vocab"], size=int(n)).tolist() for n in (24, 31)]
    frames = [len(c) // 160 for c in clips2]
    res = eng.whisper_align(toks, frames, 3, want_cost=True)
    for i in range(2):
        cost, ti, tj = WO.find_alignment(toks[i], eng.whisper_encode_fetch(i), Wd, tdims, frames[i], 3)
        assert np.linalg.norm(res[i]["cost"] - cost) / np.linalg.norm(cost) <= 1e-2, i
        wi, wj = WO.dtw_path(res[i]["cost"])
        assert np.array_equal(res[i]["text_indices"], wi) and np.array_equal(res[i]["time_indices"], wj)
