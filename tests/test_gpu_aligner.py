"""The aligner end to end on the GPU (R8, SURVEY.md 8b "Aligner entry"): ``Aligners.use_whisper_timestamped.main`` and the
two Whisper steps of ``AudioPipeline`` over directories of WAV files, with a random-init miniature Whisper written as a
checkpoint file and a miniature vocabulary file (no trained weights exist offline).  Checked: the reference's directory
contract (golden G8 for the scripted part, structure + internal consistency for the free-running part), the decoding
controls of ``pce_whisper_decode_step_ex`` against the float32 restatement, and the transcription flow's invariants."""
import base64
import json
import logging
import wave
from pathlib import Path

import numpy as np
import pytest

from oracle import whisper_oracle as WO
from prosody_control_french_tts_amd import engine as E
from prosody_control_french_tts_amd import synth, textgrid_io as TG, whisper_weights as WW
from prosody_control_french_tts_amd.Aligners import checkpoint as CK
from prosody_control_french_tts_amd.Aligners import decoding as DEC
from prosody_control_french_tts_amd.Aligners import transcribe as TR
from prosody_control_french_tts_amd.Aligners import use_whisper_timestamped as A
from prosody_control_french_tts_amd.Aligners.tokenizer import WhisperTokenizer

pytestmark = pytest.mark.gpu

EDIMS = dict(n_mels=80, n_ctx=1500, n_state=128, n_head=2, n_layer=2)
MERGES = [b" b", b"on", b" bon", b"jo", b"ur", b" le", b" m", b"nd", b"\xc3\xa9", b" \xc3", b" la", b" de", b"es", b" p", b"ar", b" v", b"oi"]


def toy_tokenizer():
    return WhisperTokenizer.toy(MERGES, language="fr")


def write_wav(path, pcm, rate=16000):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(rate); w.writeframes(np.asarray(pcm, dtype="<i2").tobytes())


def write_model_dir(root: Path, name="medium", seed=79, merges=None, word_gain=1.0, eot_gain=1.0):
    """A checkpoint directory as ``PCE_WHISPER_DIR`` expects it: <name>.npz + multilingual.tiktoken (toy sizes).
    ``merges``: the multi-byte tokens of the vocabulary (default: MERGES); ``word_gain`` / ``eot_gain`` scale the embedding rows (= the
    tied output projection) of the multi-byte tokens / of end-of-text: a random-init model that prefers whole words to single bytes and
    does not stop at once produces TextGrids with many words and pauses, which the end-to-end tests need (tests/test_gpu_c5.py)."""
    tk = toy_tokenizer() if merges is None else WhisperTokenizer.toy(merges, language="fr")
    tdims = dict(n_vocab=tk.n_vocab, n_text_ctx=128, n_state=128, n_head=2, n_layer=2)
    enc, dec = WW.synthetic_weights(EDIMS, seed=77), WW.greedy_test_decoder_weights(tdims, seed=seed)
    if word_gain != 1.0 or eot_gain != 1.0:
        emb = dec["token_embedding.weight"].copy()
        emb[256:tk.eot] *= word_gain; emb[tk.eot] *= eot_gain
        dec["token_embedding.weight"] = emb
    root.mkdir(parents=True, exist_ok=True)
    np.savez(root / f"{name}.npz", **{"encoder." + k: v for k, v in enc.items()}, **{"decoder." + k: v for k, v in dec.items()})
    with open(root / "multilingual.tiktoken", "wb") as f:
        for tok, rank in tk.ranks.items():
            f.write(base64.b64encode(tok) + b" " + str(rank).encode() + b"\n")
    return tk, tdims, enc, dec


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    root = tmp_path_factory.mktemp("whisper_dir")
    return (root,) + write_model_dir(root)


# ----------------------------------------------------------------------------------------------------- golden G8, real gate
def test_main_file_contract_with_the_gpu_gate(engine, tmp_path, monkeypatch):
    """tests/test_aligner_host.py's G8 check of main() with the gate's sums coming from k_energy instead of numpy."""
    from tests.test_aligner_host import GOLD, run_main_on_fixture
    from prosody_control_french_tts_amd import tagger as T
    with open(GOLD / "aligner.json", encoding="utf-8") as f:
        g8 = json.load(f)
    A.set_nlp(T.TablePosTagger(table=g8["pos_table"]))
    try:
        got = run_main_on_fixture(g8, tmp_path, monkeypatch, engine)
    finally:
        A.set_nlp(T.TablePosTagger())
    assert got == g8["main"]["files"]


# ----------------------------------------------------------------------------------------------------- decode step controls
def _encode_clips(engine, model_dir, clips):
    root, tk, tdims, enc, dec = model_dir
    model = CK.load_model("medium", str(root)).load_into(engine)
    engine.upload(clips, 16000)
    engine.logmel_run(80)
    engine.whisper_encode_run()
    return model, tk, tdims, dec


def test_decode_step_ex_prompts_probe_and_sampling(engine, model_dir):
    clips = [synth.synth_clip(i, seconds=3.0) for i in range(3)]
    model, tk, tdims, dec = _encode_clips(engine, model_dir, clips)
    rules = tk.decoding_rules()
    mask = DEC.vocab_mask(tdims["n_vocab"], rules["suppress_tokens"], rules["blank_tokens"], rules["no_timestamps"])
    sot = list(tk.sot_sequence())
    prev = tk.encode(" bonjour le monde")
    prompts = [sot, [tk.sot_prev] + prev + sot, [tk.sot_prev] + prev[:2] + sot]
    begins = [len(p) for p in prompts]
    encs = [engine.whisper_encode_fetch(i) for i in range(3)]
    nxt, lp, _ = engine.whisper_decode_step_ex(prompts, begins, rules["eot"], rules["timestamp_begin"], mask, rules["max_initial_timestamp_index"])
    for i, p in enumerate(prompts):
        logits = WO.find_alignment(p, encs[i], dec, tdims, 2, 0, want_internal=True)["logits"][-1]
        f = WO.apply_decoding_rules(logits, p, begins[i], rules)
        assert np.isfinite(f[int(nxt[i])]) and int(nxt[i]) >= tk.timestamp_begin            # first sampled token: a timestamp, whatever the prompt length
        top = np.sort(f[np.isfinite(f)])[::-1]
        if top[0] - top[1] > 0.05:
            assert int(nxt[i]) == int(np.argmax(f))
        lsm = f - (np.max(f) + np.log(np.sum(np.exp(f[np.isfinite(f)] - np.max(f)))))
        assert abs(float(lp[i]) - float(lsm[int(nxt[i])])) <= 0.05
    # no_speech probe: softmax of the UNFILTERED logits at the <|startoftranscript|> position
    sot_index = [p.index(tk.sot) for p in prompts]
    got = DEC.no_speech_probs(engine, tdims["n_vocab"], prompts, sot_index, rules, tk.no_speech)
    for i, p in enumerate(prompts):
        logits = WO.find_alignment(p[:sot_index[i] + 1], encs[i], dec, tdims, 2, 0, want_internal=True)["logits"][-1].astype(np.float64)
        want = np.exp(logits[tk.no_speech] - np.max(logits)) / np.sum(np.exp(logits - np.max(logits)))
        assert abs(got[i] - want) <= 0.02 * max(want, 1e-3) + 1e-5, (i, got[i], want)
    # temperature: reproducible for a seed, different across seeds, never a filtered token, and at a tiny temperature the arg-max
    a1, _, _ = engine.whisper_decode_step_ex(prompts, begins, rules["eot"], rules["timestamp_begin"], mask, rules["max_initial_timestamp_index"], temperature=1.0, seed=5)
    a2, _, _ = engine.whisper_decode_step_ex(prompts, begins, rules["eot"], rules["timestamp_begin"], mask, rules["max_initial_timestamp_index"], temperature=1.0, seed=5)
    assert np.array_equal(a1, a2)
    draws = np.stack([engine.whisper_decode_step_ex(prompts, begins, rules["eot"], rules["timestamp_begin"], mask, rules["max_initial_timestamp_index"],
                                                    temperature=1.0, seed=s)[0] for s in range(40)])
    assert all(len(set(draws[:, i].tolist())) > 1 for i in range(3))                          # it does sample
    assert draws.min() >= tk.timestamp_begin and draws.max() <= tk.timestamp_begin + rules["max_initial_timestamp_index"]
    cold, _, _ = engine.whisper_decode_step_ex(prompts, begins, rules["eot"], rules["timestamp_begin"], mask, rules["max_initial_timestamp_index"], temperature=1e-4, seed=9)
    assert np.array_equal(cold, nxt)
    # sampled frequencies follow softmax(filtered logits / T): compare the empirical mass of the arg-max token with its probability
    i = 0
    logits = WO.find_alignment(prompts[i], encs[i], dec, tdims, 2, 0, want_internal=True)["logits"][-1]
    f = WO.apply_decoding_rules(logits, prompts[i], begins[i], rules).astype(np.float64)
    p = np.exp(f - np.max(f)); p /= p.sum()
    many = np.array([engine.whisper_decode_step_ex(prompts, begins, rules["eot"], rules["timestamp_begin"], mask, rules["max_initial_timestamp_index"],
                                                   temperature=1.0, seed=1000 + s)[0][i] for s in range(300)])
    k = int(np.argmax(p))
    assert abs(np.mean(many == k) - p[k]) <= 4 * np.sqrt(p[k] * (1 - p[k]) / 300) + 0.03


def test_device_resident_decode_loop_equals_the_host_driven_loop(engine, model_dir):
    """pce_whisper_decode_loop (prompts up once, tables of every later step kept on the device, results down once) against the loop
    it replaces (one pce_whisper_decode_step_ex round trip per step): the same tokens and the same log-probabilities, step by step,
    for prompts of different lengths, greedy and at a temperature (the sampling keys are (seed, clip, position) in both), with a
    sequence that starts out ended, and when the loop stops early because every sequence has produced end-of-text."""
    clips = [synth.synth_clip(i, seconds=3.0) for i in range(4)]
    model, tk, tdims, dec = _encode_clips(engine, model_dir, clips)
    rules = tk.decoding_rules()
    V = tdims["n_vocab"]
    sot = list(tk.sot_sequence())
    prev = tk.encode(" bonjour le monde")
    prompts = [sot, [tk.sot_prev] + prev + sot, [tk.sot_prev] + prev[:2] + sot, sot]
    begins = [len(p) for p in prompts]
    for temp, seed, active in [(0.0, 0, None), (0.7, 11, None), (0.0, 0, [True, False, True, True])]:
        host = DEC.decode_batch(engine, V, prompts, begins, rules, 14, temperature=temp, seed=seed, active=active, device_loop=False)
        dev = DEC.decode_batch(engine, V, prompts, begins, rules, 14, temperature=temp, seed=seed, active=active, device_loop=True)
        assert dev[0] == host[0], (temp, active)
        for a, b in zip(dev[1], host[1]):
            assert np.allclose(a, b, rtol=0, atol=2e-3)          # (the streaming single-query attention sums in fp32 where the MFMA kernel rounds P to bf16)
        assert np.allclose(dev[2], host[2], atol=2e-2)
        if active:
            assert dev[0][1] == [] and dev[2][1] == 0.0
    # raw entry point: column s equals the s-th host-driven step; steps that did not run read as end-of-text
    mask = DEC.vocab_mask(V, rules["suppress_tokens"], rules["blank_tokens"], rules["no_timestamps"])
    toks, lps, _ = engine.whisper_decode_loop(prompts, begins, rules["eot"], rules["timestamp_begin"], mask, 9, rules["max_initial_timestamp_index"], check_every=2)
    seqs = [list(p) for p in prompts]
    for step in range(toks.shape[1]):
        nxt, lp, _ = engine.whisper_decode_step_ex(seqs, begins, rules["eot"], rules["timestamp_begin"], mask, rules["max_initial_timestamp_index"])
        assert np.array_equal(nxt, toks[:, step]), step
        for s_, t in zip(seqs, nxt):
            s_.append(int(t))
    # every sequence already ended: ONE step, all end-of-text
    ended = [p + [rules["eot"]] for p in prompts]
    toks, lps, _ = engine.whisper_decode_loop(ended, begins, rules["eot"], rules["timestamp_begin"], mask, 12, rules["max_initial_timestamp_index"])
    assert toks.shape[1] == 1 and (toks == rules["eot"]).all() and (lps == 0).all()


def test_decode_loop_stops_a_sequence_at_the_text_context(engine, model_dir):
    """A prompt of n_text_ctx // 2 + 3 tokens (``condition_on_previous_text``: up to 227 of 448 at full size, 67 of 128 here) plus
    sample_len = n_text_ctx // 2 sampled ones exceeds the text context.  openai-whisper's loop breaks when ``tokens.shape[-1] > n_ctx``
    (decoding.py, DecodingTask._main_loop) and keeps what was sampled; the device loop does the same PER SEQUENCE (the other clips of
    the batch go on) instead of refusing the window, also in a temperature-fallback round where the clip with the longest prompt is
    already accepted (inactive: prompt + end-of-text).  End-of-text is suppressed here so that every sequence runs into a limit."""
    clips = [synth.synth_clip(i, seconds=3.0) for i in range(3)]
    model, tk, tdims, dec = _encode_clips(engine, model_dir, clips)
    n_ctx = tdims["n_text_ctx"]
    rules = tk.decoding_rules()
    rules = dict(rules, suppress_tokens=sorted(set(rules["suppress_tokens"]) | {rules["eot"]}))
    V = tdims["n_vocab"]
    sot = list(tk.sot_sequence())
    prev = (tk.encode(" bonjour le monde") * 40)[: n_ctx // 2 - 1]
    long_prompt = [tk.sot_prev] + prev + sot
    assert len(long_prompt) == n_ctx // 2 + len(sot)
    prompts = [long_prompt, sot, [tk.sot_prev] + prev[:5] + sot]
    begins = [len(p) for p in prompts]
    sample_len = n_ctx // 2
    assert begins[0] + sample_len > n_ctx
    for active in (None, [False, True, True]):
        host = DEC.decode_batch(engine, V, prompts, begins, rules, sample_len, active=active, device_loop=False, n_text_ctx=n_ctx)
        dev = DEC.decode_batch(engine, V, prompts, begins, rules, sample_len, active=active, device_loop=True, n_text_ctx=n_ctx)
        assert dev[0] == host[0], active
        assert np.allclose(dev[2], host[2], atol=2e-2)
        if active is None:
            assert len(dev[0][0]) == n_ctx + 1 - begins[0]               # the forward pass over the full context still yields its token
        else:
            assert dev[0][0] == []
        assert len(dev[0][1]) == sample_len and len(dev[0][2]) == sample_len      # the clips with room were not cut short


# ----------------------------------------------------------------------------------------------------- transcription flow
def test_transcribe_batch_invariants(engine, model_dir):
    """Free-running flow on a 4 s, a 33 s (two windows) and a 0.3 s clip: result shape of whisper_timestamped, times inside the
    recording and monotone, words inside their segment, tokens <-> text, confidences in (0, 1], VAD cuts honoured."""
    root, tk, tdims, enc, dec = model_dir
    model = CK.load_model("medium", str(root)).load_into(engine)
    rng = np.random.default_rng(5)
    long_clip = np.concatenate([synth.synth_clip(10 + k, seconds=3.0) for k in range(11)])
    clips = [synth.synth_clip(1, seconds=4.0), long_clip, (rng.standard_normal(4800) * 3000).astype(np.int16)]
    opts = TR.TranscribeOptions(sample_len=24, logprob_threshold=None, no_speech_threshold=None, compression_ratio_threshold=None)
    class _Catch(logging.Handler):
        def __init__(self): super().__init__(); self.msgs = []
        def emit(self, record): self.msgs.append((record.levelno, record.getMessage()))
    lg = logging.getLogger("t"); catch = _Catch(); lg.addHandler(catch)
    res = TR.transcribe_batch(engine, model, tk, clips, opts, lg)
    lg.removeHandler(catch)
    # the one option of the reference's call the engine cannot honour is announced at WARNING level, every call
    assert any(lv >= logging.WARNING and "detect_disfluencies" in m for lv, m in catch.msgs)
    assert len(res) == 3
    for r, c in zip(res, clips):
        dur = len(c) / 16000.0
        assert set(r) == {"text", "segments", "language"} and r["language"] == "fr"
        assert r["text"] == "".join(s["text"] for s in r["segments"])
        last = 0.0
        for k, s in enumerate(r["segments"]):
            assert s["id"] == k and {"seek", "start", "end", "text", "tokens", "temperature", "avg_logprob", "compression_ratio", "no_speech_prob",
                                      "confidence", "words"} <= set(s)
            assert 0.0 <= s["start"] <= s["end"] <= dur + 0.02 and s["start"] >= last - 1e-9
            last = s["start"]
            assert tk.decode(s["tokens"]) == s["text"]
            assert 0.0 < s["confidence"] <= 1.0 and s["avg_logprob"] <= 0.0
            for w in s["words"]:
                assert s["start"] - 1e-9 <= w["start"] <= w["end"] <= s["end"] + 1e-9 and w["text"] == w["text"].strip()
                assert 0.0 < w["confidence"] <= 1.0
    assert len(res[1]["segments"]) >= 2 and res[1]["segments"][-1]["seek"] > 0                # the long recording went past its first window
    # no VAD: same shape; with VAD every time still lies inside the recording (mapped back through the cuts)
    res2 = TR.transcribe_batch(engine, model, tk, clips[:1], TR.TranscribeOptions(vad=None, sample_len=24, logprob_threshold=None,
                                                                                 no_speech_threshold=None, compression_ratio_threshold=None))
    assert res2[0]["segments"] and res2[0]["segments"][0]["start"] >= 0.0
    # a clip the VAD finds silent gives an empty result (the caller turns it into the "..." placeholder)
    quiet = TR.transcribe_batch(engine, model, tk, [np.zeros(32000, np.int16)], opts)
    assert quiet[0]["segments"] == [] and quiet[0]["text"] == ""


def test_temperature_ladder_and_thresholds(engine, model_dir):
    """decode_with_fallback: a log-probability threshold nothing can meet walks the whole ladder (the last temperature is
    kept); no_speech_threshold 0 with that threshold skips every window (no segments)."""
    root, tk, tdims, enc, dec = model_dir
    model = CK.load_model("medium", str(root)).load_into(engine)
    clips = [synth.synth_clip(2, seconds=4.0)]
    opts = TR.TranscribeOptions(vad=None, sample_len=12, temperature=(0.0, 0.4, 1.0), logprob_threshold=0.0, no_speech_threshold=None,
                                compression_ratio_threshold=None)
    r = TR.transcribe_batch(engine, model, tk, clips, opts)[0]
    assert r["segments"] and all(s["temperature"] == 1.0 for s in r["segments"])
    opts = TR.TranscribeOptions(vad=None, sample_len=12, logprob_threshold=0.0, no_speech_threshold=0.0, compression_ratio_threshold=None)
    assert TR.transcribe_batch(engine, model, tk, clips, opts)[0]["segments"] == []


def test_sampled_tokens_do_not_depend_on_the_batch(engine, model_dir):
    """``pce_whisper_sample_keys``: the noise of a sampled token is keyed by the caller's id of the clip, so a clip draws the same tokens
    wherever it stands in a batch and whatever it is batched with -- without keys the batch position is the key and the draws move.
    Then the same through ``transcribe_batch`` with every window pushed to the last temperature of the ladder (the keys are a hash of
    the clip's samples and the window start): a recording's result is the same alone, first of three and last of three."""
    root, tk, tdims, enc, dec = model_dir
    model = CK.load_model("medium", str(root)).load_into(engine)
    rules = tk.decoding_rules()
    V = tdims["n_vocab"]
    clips = [synth.synth_clip(k, seconds=3.0 + k) for k in (1, 2, 3)]
    sot = list(tk.sot_sequence())

    def sample(order, keys):
        engine.upload([clips[k] for k in order], 16000)
        engine.logmel_run(model.dims["n_mels"]); engine.whisper_encode_run()
        if keys is not None:
            engine.whisper_sample_keys([keys[k] for k in order])
        toks, _, _ = DEC.decode_batch(engine, V, [sot] * len(order), [len(sot)] * len(order), rules, 10, temperature=1.0, seed=11)
        return {k: toks[i] for i, k in enumerate(order)}

    keys = {0: 123456789, 1: 7, 2: 2**31 - 1}
    a, b, c = sample([0, 1, 2], keys), sample([2, 0, 1], keys), sample([1], keys)
    assert a == b and c[1] == a[1]
    p, q = sample([0, 1, 2], None), sample([2, 0, 1], None)
    assert p != q                                                    # (10 draws from a 50 000-way softmax at temperature 1, three clips)
    assert sample([0, 1, 2], {0: 0, 1: 1, 2: 2}) == p               # the default key IS the batch position
    # keys are dropped by the next encoder run, refused for another clip count, and before any encoder run
    engine.whisper_sample_keys([5, 6, 7])
    with pytest.raises(E.PceError):
        engine.whisper_sample_keys([5, 6])
    engine.whisper_sample_keys(None)

    opts = TR.TranscribeOptions(vad=None, sample_len=10, temperature=(0.0, 1.0), logprob_threshold=0.0, no_speech_threshold=None,
                                compression_ratio_threshold=None, max_windows=2)
    strip = lambda r: [(s["tokens"], s["temperature"], [w["text"] for w in s["words"]]) for s in r["segments"]]
    alone = strip(TR.transcribe_batch(engine, model, tk, [clips[1]], opts)[0])
    first = strip(TR.transcribe_batch(engine, model, tk, [clips[1], clips[0], clips[2]], opts)[0])
    last = strip(TR.transcribe_batch(engine, model, tk, [clips[2], clips[0], clips[1]], opts)[2])
    assert alone and all(t == 1.0 for _, t, _ in alone)
    assert alone == first == last


# ----------------------------------------------------------------------------------------------------- main() and the pipeline steps
def _assert_aligner_outputs(audio_dir: Path, out: Path, names, gated=()):
    for n in names:
        tg = TG.read_textgrid(out / f"{n}.TextGrid")
        assert len(tg.tiers) == 1 and tg.tiers[0].name == "words" and tg.tiers[0].intervals
        txt = (Path(str(out) + "_transcription") / f"{n}.txt").read_text(encoding="utf-8")
        if n in gated:
            assert [m for _, _, m in tg.tiers[0].intervals] == ["..."] and txt == "..."
            assert not (Path(str(out) + "_raw_json") / f"{n}.raw.json").exists()
            continue
        raw = json.loads((Path(str(out) + "_raw_json") / f"{n}.raw.json").read_text(encoding="utf-8"))
        final = json.loads((Path(str(out) + "_transcription") / f"{n}.json").read_text(encoding="utf-8"))
        assert set(raw) == {"text", "segments", "language"} and len(final["segments"]) == len(raw["segments"])
        assert (Path(str(out) + "_textgrid_raw") / f"{n}.TextGrid").exists()
        words = [w for s in final["segments"] for w in s["words"]]
        marks = [m for _, _, m in tg.tiers[0].intervals if m != " "]
        assert marks == [w["text"].replace("[*]", " ") for w in words]
        ivs = tg.tiers[0].intervals
        assert all(abs(ivs[k][1] - ivs[k + 1][0]) < 1e-9 for k in range(len(ivs) - 1)) and ivs[0][0] == 0.0


def test_main_end_to_end_from_checkpoint_files(engine, model_dir, tmp_path, monkeypatch):
    """Audio in, TextGrids out: gate -> resample (44.1 kHz input) -> VAD -> log-mel -> encoder -> greedy decoding -> forced
    alignment -> JSON / txt / TextGrid + the raw copies + "..." placeholders, weights and vocabulary read from files."""
    root = model_dir[0]
    monkeypatch.setenv("PCE_WHISPER_DIR", str(root))
    A.set_model_source()
    E.set_default_engine(engine)
    voice = tmp_path / "Data" / "V9"
    audio = voice / "audio"; audio.mkdir(parents=True)
    write_wav(audio / "segment_ph1.wav", synth.synth_clip(3, seconds=5.0))
    write_wav(audio / "segment_ph2.wav", np.zeros(20000, np.int16))                                  # gated: silence
    up = np.repeat(synth.synth_clip(4, seconds=3.0), 3)[:132300]                                   # crude 44.1 kHz material, 3 s
    write_wav(audio / "segment_ph3.wav", up, 44100)
    write_wav(audio / "segment_ph4.wav", np.concatenate([synth.synth_clip(20 + k, seconds=3.0) for k in range(11)]))   # 33 s: two windows
    out = voice / "WhisperTS_textgrid_files"
    try:
        A.main(str(audio), str(out), whisper_model="medium", device="cuda:0", logger=logging.getLogger("t"))
    finally:
        E.set_default_engine(None)
    # the toy model's output is arbitrary text: a file with fewer than 3 words / 10 characters legitimately becomes "..."
    names = ["segment_ph1", "segment_ph2", "segment_ph3", "segment_ph4"]
    for n in names:
        assert (out / f"{n}.TextGrid").exists() and (Path(str(out) + "_transcription") / f"{n}.txt").exists()
    _assert_aligner_outputs(audio, out, names, gated=("segment_ph2",))
    raw4 = json.loads((Path(str(out) + "_raw_json") / "segment_ph4.raw.json").read_text(encoding="utf-8"))
    if raw4["text"] != "...":
        assert max(s["end"] for s in raw4["segments"]) > 30.0


def test_audio_pipeline_whisper_steps(engine, model_dir, tmp_path):
    """config.yaml's first step and the final one no longer raise: "Align+Transcribe" leaves WhisperTS_textgrid_files,
    transcription, transcription_raw and the raw JSON folder; "Final Transcribe" leaves OUT.TextGrid / OUT.txt beside
    OUT.wav; used_config.yaml is written."""
    from prosody_control_french_tts_amd import audio_pipeline as AP
    root = model_dir[0]
    cfg = {"data_dir": "Data", "out_dir": "Out", "azure_voice_name": "fr-FR-HenriNeural", "whisper_device": "cuda:0", "whisper_model": "medium",
           "whisper_dir": str(root), "steps_to_run": ["Align+Transcribe", "Final Transcribe"]}
    voice = tmp_path / "Data" / "v1"
    (voice / "audio").mkdir(parents=True)
    write_wav(voice / "audio" / "segment_ph1.wav", synth.synth_clip(5, seconds=4.0))
    write_wav(voice / "audio" / "segment_ph2.wav", np.zeros(16000, np.int16))
    res = tmp_path / "Out" / "results" / "v1"; res.mkdir(parents=True)
    write_wav(res / "OUT.wav", synth.synth_clip(6, seconds=6.0))
    AP.AudioPipeline("v1", cfg, base=tmp_path, engine=engine).run()
    for n in ("segment_ph1", "segment_ph2"):
        assert (voice / "WhisperTS_textgrid_files" / f"{n}.TextGrid").exists()
        assert (voice / "transcription" / f"{n}.txt").exists() and (voice / "transcription_raw" / f"{n}.txt").exists()
    assert (voice / "transcription_raw" / "segment_ph2.txt").read_text(encoding="utf-8") == "..."
    assert (res / "OUT.TextGrid").exists() and (res / "OUT.txt").exists() and (res / "used_config.yaml").exists()
    assert not list((res / "final_whisper" / "WhisperTS_textgrid_files").glob("*.TextGrid"))           # moved next to OUT.wav
