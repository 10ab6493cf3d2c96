"""GPU edge cases through the C ABI: empty and ragged inputs, virtual samples, too-short slices,
call-order and limit errors, repeated runs on a cached plan, full-size property checks."""
import numpy as np
import pytest

import prosody_control_french_tts_amd as pkg
from oracle import oracle as O
from prosody_control_french_tts_amd import engine as E
from prosody_control_french_tts_amd import synth

pytestmark = pytest.mark.gpu


def test_call_order_and_argument_errors():
    with pkg.ProsodyEngine(0) as eng:
        for fn in (lambda: eng._check(eng._lib.pce_energy_run(eng._ctx, None, 0, 500)),
                   lambda: eng.lufs_run(E.make_slices([0], [0], [10])),
                   lambda: eng.stft_db_run(1024, 256)):
            with pytest.raises(pkg.PceError, match="no batch uploaded"):
                fn()
        eng.upload([np.zeros(100, dtype=np.int16)], 16000)
        with pytest.raises(pkg.PceError, match="before"):
            eng._en_n = 1; eng.energy_fetch()
        with pytest.raises(pkg.PceError, match="clip 3 out of range"):
            eng.energy(E.make_slices([3], [0], [10]))
        with pytest.raises(pkg.PceError, match="end < begin"):
            eng.energy(E.make_slices([0], [10], [5]))
        with pytest.raises(pkg.PceError, match="n_fft 512 unsupported"):
            eng.stft_db_run(512, 128)
        p = E.PitchParams.praat(20.0, 600.0)            # ceiling/floor = 30 candidates > 16
        with pytest.raises(pkg.PceError, match="candidates"):
            eng.upload([synth.synth_clip(0, seconds=1.0)], 16000); eng.pitch(eng.whole_clip_slices(), p)
        with pytest.raises(pkg.PceError, match="16 kHz"):
            eng.upload([np.zeros(4410, dtype=np.int16)], 44100); eng.logmel_run(80)


def test_empty_ragged_and_virtual_slices(engine):
    clips = [np.zeros(0, dtype=np.int16), np.array([1234], dtype=np.int16), synth.synth_clip(2, seconds=0.5),
             np.full(7, -32768, dtype=np.int16)]
    engine.upload(clips, 16000)
    # no slices at all
    assert len(engine.energy(E.make_slices([], [], []))) == 0
    lu, st = engine.lufs(E.make_slices([], [], []))
    assert len(lu) == 0
    assert len(engine.pitch(E.make_slices([], [], []), E.PitchParams.praat(150.0, 600.0))["summary"]) == 0
    # slices entirely or partly outside their clip are virtual zeros
    sl = E.make_slices([0, 1, 1, 2, 2, 3], [0, -5, 0, -100, 7990, 0], [0, 6, 1, 50, 8100, 7])
    en = engine.energy(sl)
    assert list(en["n"]) == [0, 11, 1, 150, 110, 7]
    assert list(en["sum_sq"]) == [0, 1234 * 1234, 1234 * 1234, int(np.sum(clips[2][:50].astype(np.int64) ** 2)),
                                  int(np.sum(clips[2][7990:].astype(np.int64) ** 2)), 7 * 32768 * 32768]
    assert en["peak_abs"][5] == 32768 and en["n_loud"][5] == 0          # abs(-32768) wraps in int16: never "loud"
    # every slice here is too short for LUFS and for a 150 Hz pitch floor
    lu, st = engine.lufs(sl)
    assert list(st) == [E.SLICE_EMPTY] + [E.SLICE_TOO_SHORT] * 5 and np.all(np.isnan(lu))
    res = engine.pitch(sl, E.PitchParams.praat(150.0, 600.0))
    assert list(res["summary"]["status"]) == [E.SLICE_EMPTY] + [E.SLICE_TOO_SHORT] * 5 and res["frame_offsets"][-1] == 0
    # STFT of empty and one-sample clips: one centred frame each
    engine.stft_db_run(1024, 256)
    assert engine.stft_db_fetch(0).shape == (513, 1) and np.all(engine.stft_db_fetch(0) == 0.0)
    assert np.allclose(engine.stft_db_fetch(1), O.stft_db(clips[1].astype(np.float32) / 32768.0), atol=2e-2)


def test_pitch_slice_with_virtual_tail_matches_oracle(engine):
    """Praat's extract_part beyond the end of the sound pads with zeros: t1 past the file."""
    c = synth.synth_clip(5, seconds=1.0)
    engine.upload([c], 16000)
    sl = E.make_slices([0, 0], [12000, -800], [20000, 4000], [0.5 / 16000 + 12000 / 16000, 0.5 / 16000 - 800 / 16000])
    res = engine.pitch(sl, E.PitchParams.praat(150.0, 600.0))
    for k, (b, e) in enumerate([(12000, 20000), (-800, 4000)]):
        x = np.zeros(e - b); lo, hi = max(b, 0), min(e, len(c)); x[lo - b:hi - b] = c[lo:hi] / 32768.0
        want = O.pitch_ac(x, 1 / 16000, float(sl[k]["x1"]), O.praat_params(150.0, 600.0))["f0"]
        got = res["f0"][res["frame_offsets"][k]:res["frame_offsets"][k + 1]]
        assert np.array_equal(got > 0, want > 0)
        v = want > 0
        assert not v.any() or np.max(np.abs(got[v] - want[v]) / want[v]) <= 1e-6


def test_repeated_runs_are_bitwise_reproducible(engine, synth16k):
    engine.upload(synth16k, 16000)
    sl = engine.whole_clip_slices()
    p = E.PitchParams.praat(150.0, 600.0)
    a = engine.pitch(sl, p, want_strength=True); la, _ = engine.lufs(sl); ea = engine.energy(sl)
    engine.stft_db_run(1024, 256); sa = engine.stft_db_fetch(0)
    for _ in range(3):
        b = engine.pitch(sl, p, want_strength=True); lb, _ = engine.lufs(sl); eb = engine.energy(sl)
        engine.stft_db_run(1024, 256)
        assert np.array_equal(a["f0"], b["f0"]) and np.array_equal(a["strength"], b["strength"])
        assert np.array_equal(a["summary"], b["summary"]) and np.array_equal(la, lb, equal_nan=True) and np.array_equal(ea, eb)
        assert np.array_equal(sa, engine.stft_db_fetch(0))


def test_full_size_batch_properties(engine):
    """BASELINE.json C2 size (256 x 10 s): size-independent properties instead of a full CPU comparison."""
    clips = synth.synth_batch(256, 10.0)
    engine.upload(clips, 16000)
    sl = engine.whole_clip_slices()
    en = engine.energy(sl)
    idx = [0, 17, 128, 255]
    for i in idx:                                                     # spot checks against numpy on single clips
        assert en[i]["sum_sq"] == int(np.sum(clips[i].astype(np.int64) ** 2)) and en[i]["n"] == 160000
    assert int(en["sum_sq"].sum()) == sum(int(np.sum(c.astype(np.int64) ** 2)) for c in clips)    # checksum of checksums
    # additivity: the energy of a clip equals the sum over any partition of it into slices
    cuts = [0, 1, 12345, 80000, 159999, 160000]
    parts = engine.energy(E.make_slices([7] * 5, cuts[:-1], cuts[1:]))
    assert int(parts["sum_sq"].sum()) == int(en[7]["sum_sq"]) and int(parts["n_loud"].sum()) == int(en[7]["n_loud"])
    assert int(parts["peak_abs"].max()) == int(en[7]["peak_abs"])
    res = engine.pitch(sl, E.PitchParams.praat(150.0, 600.0))
    assert np.all(res["summary"]["n_frames"] == 1997) and res["frame_offsets"][-1] == 256 * 1997
    f0 = res["f0"].reshape(256, 1997)
    assert np.all((f0 == 0) | ((f0 > 100.0) & (f0 < 600.0)))
    v = f0 > 0
    assert np.array_equal(res["summary"]["n_voiced"], v.sum(axis=1))
    med = np.array([np.median(r[r > 0]) if (r > 0).any() else 0.0 for r in f0])
    assert np.array_equal(res["summary"]["median_f0"], med)          # the GPU median is an exact order statistic
    for i in idx:
        want = O.pitch_ac(clips[i] / 32768.0, 1 / 16000, 0.5 / 16000, O.praat_params(150.0, 600.0))["f0"]
        assert np.array_equal(f0[i] > 0, want > 0) and np.max(np.abs(f0[i][want > 0] - want[want > 0]) / want[want > 0]) <= 1e-6
    lu, st = engine.lufs(sl)
    assert np.all(st == 0) and np.all(np.isfinite(lu))
    for i in idx:
        assert abs(lu[i] - O.lufs_c(clips[i].astype(float), 16000)) <= 1e-6
    # loudness is invariant to a power-of-two gain (peak normalisation) -- checked on a scaled copy
    engine.upload([clips[3], (clips[3].astype(np.int32) // 2 * 2 // 2).astype(np.int16)], 16000)
    engine.stft_db_run(1024, 256)
    s0 = engine.stft_db_fetch(0)
    assert s0.shape == (513, 626) and s0.max() == 0.0 and s0.min() >= -80.0


@pytest.mark.gpu
def test_async_stats_match_the_synchronous_fetches(engine, synth16k):
    """pce_stats_enqueue / pce_stats_wait return the numbers of pce_energy_fetch / pce_lufs_fetch /
    pce_pitch_fetch, also when a second batch of launches is already queued behind the copies."""
    import prosody_control_french_tts_amd as pkg
    eng = engine
    eng.upload(synth16k, 16000)
    sl = eng.whole_clip_slices()
    params = pkg.PitchParams.praat(150.0, 600.0)
    eng.energy_run(sl, 500); eng.lufs_run(sl); eng.pitch_run(sl, params)
    eng.stats_enqueue(0)
    eng.energy_run(sl, 500); eng.lufs_run(sl); eng.pitch_run(sl, params)      # next batch already in flight
    eng.stats_enqueue(1)
    a = eng.stats_wait(0); b = eng.stats_wait(1)
    en = eng.energy_fetch(); lu, st = eng.lufs_fetch(); pi = eng.pitch_fetch(want_f0=False)["summary"]
    for r in (a, b):
        assert r["energy"].tobytes() == en.tobytes()
        assert r["lufs"][0].tobytes() == lu.tobytes() and (r["lufs"][1] == st).all()
        assert r["pitch"].tobytes() == pi.tobytes()


@pytest.mark.gpu
def test_nw_align_gpu_reproduces_the_reference_alignments(engine, tmp_path):
    """pce_nw_align (one launch for the whole batch) gives the text of the reference's own runs (golden G2),
    plus ragged / empty / tie-heavy pairs against the host restatement."""
    import json, os
    from prosody_control_french_tts_amd.Pipeline import NeedlemanWunschAlignement as NW
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "needleman_wunsch.json")))
    tup = lambda rows: [(r["PhraseID"], r["Text"], float(r["Start"]), float(r["End"]), float(r["Duration"])) for r in rows]
    pairs = [(tup(c["seq1"]), tup(c["seq2"])) for c in cases]
    got = NW.needleman_wunsch_batch(pairs, engine)
    for c, g in zip(cases, got):
        assert NW.format_alignment(g) == c["expected_text"]
    rng = np.random.default_rng(5)
    words = ["le", "la", "chat", "Chat.", "mange", "souris,", "une", "et", "ß", "oui?"]
    extra = []
    for n, m in [(0, 0), (0, 5), (7, 0), (1, 1), (40, 37), (300, 280), (64, 65), (1024, 1000), (1025, 30), (2300, 2250)]:   # (beyond 1 024 rows: striped)
        mk = lambda k: [(str(i), words[int(rng.integers(len(words)))], float(i), float(i) + 0.5, 0.5) for i in range(k)]
        extra.append((mk(n), mk(m)))
    for (a, b), g in zip(extra, NW.needleman_wunsch_batch(extra, engine)):
        if a and b:
            assert NW.format_alignment(g) == NW.format_alignment(NW.needleman_wunsch(a, b))
        else:       # the reference indexes seq[-1] of an empty list (IndexError); the kernel aligns everything to gaps
            assert len(g[0]) == len(a) + len(b) and [w for w in g[0] if w[0] != "-"] == a and [w for w in g[1] if w[0] != "-"] == b


@pytest.mark.gpu
def test_side_streams_do_not_change_results(engine, synth16k, monkeypatch):
    """The LUFS chain and the pitch tail run on side streams; a context created with PCE_NO_AUX=1 (single stream)
    must produce bit-identical results, also with a second batch queued behind the first."""
    clips = synth16k[:4]
    p = pkg.PitchParams.praat(150.0, 600.0)

    def run(eng):
        eng.upload(clips, 16000); sl = eng.whole_clip_slices()
        for _ in range(2):
            eng.energy_run(sl, 500); eng.lufs_run(sl); eng.pitch_run(sl, p); eng.stft_db_run(1024, 256)
        r = eng.pitch_fetch(want_f0=True, want_strength=True)
        return [eng.lufs_fetch()[0].tobytes(), r["f0"].tobytes(), r["strength"].tobytes(), r["summary"].tobytes(), eng.stft_db_fetch(1).tobytes()]

    monkeypatch.setenv("PCE_NO_AUX", "1")
    with pkg.ProsodyEngine(0) as single:
        ref = run(single)
    monkeypatch.delenv("PCE_NO_AUX")
    assert run(engine) == ref


@pytest.mark.gpu
@pytest.mark.parametrize("cpb", ["1", "3", "8"])
def test_energy_chunks_per_workgroup_do_not_change_results(monkeypatch, cpb):
    """k_energy streams `cpb` consecutive 32 KB chunks per workgroup and adds to a slice's accumulators once per (workgroup, slice):
    forced to 1 / 3 / 8 chunks on a batch whose slices start and end anywhere (several slices inside one chunk, slices that span
    many chunks, unaligned ends, empty and out-of-clip ranges) the seven integer statistics equal the numpy restatement exactly."""
    rng = np.random.default_rng(123)
    clips = [rng.integers(-32768, 32768, size=n).astype(np.int16) for n in (200001, 16384 * 3, 5, 70000, 16383, 16385)]
    idx, b, e = [], [], []
    for ci, c in enumerate(clips):
        n = len(c)
        idx += [ci] * 6
        b += [0, 1, n // 3, max(n - 9, 0), -50, n // 2]
        e += [n, min(n, 17), n // 3 + 40000, n, 10, n // 2]
    sl = E.make_slices(idx, b, e)
    monkeypatch.setenv("PCE_EN_CPB", cpb)
    with pkg.ProsodyEngine(0) as eng:
        eng.upload(clips, 16000)
        got = eng.energy(sl, 500)
    for k, (ci, b0, e0) in enumerate(zip(idx, b, e)):
        x = clips[ci][max(b0, 0):max(min(e0, len(clips[ci])), 0)].astype(np.int64)
        assert got["n"][k] == e0 - b0
        assert got["sum_sq"][k] == int(np.sum(x * x)), (cpb, k)
        assert got["sum_sq_wrap16"][k] == int(np.sum((x * x).astype(np.int16).astype(np.int64))), (cpb, k)
        assert got["n_loud"][k] == int(np.sum(np.abs(x.astype(np.int16)) > 500)), (cpb, k)
        assert got["peak_abs"][k] == (int(np.max(np.abs(x))) if len(x) else 0), (cpb, k)


@pytest.mark.gpu
def test_stft_one_fft_and_two_fft_forms_agree_bitwise(engine, synth16k, monkeypatch):
    """Default: one FFT pass writing unnormalised dB + the clip's maximum, normalised where the values are consumed (round 6; an eager in-place
    pass on a side stream before).  PCE_STFT_TWO_FFT=1: maximum pass, then the dB pass (each byte moved once).  Same float operations in the
    same order on every bin: the matrices must be identical."""
    clips = synth16k
    monkeypatch.setenv("PCE_STFT_TWO_FFT", "1")
    with pkg.ProsodyEngine(0) as two:
        two.upload(clips, 16000); two.stft_db_run(1024, 256)
        ref = [two.stft_db_fetch(i).tobytes() for i in range(len(clips))]
    monkeypatch.delenv("PCE_STFT_TWO_FFT")
    engine.upload(clips, 16000)
    engine.stft_db_run(1024, 256); engine.stft_db_run(1024, 256)        # a second run queued behind the first one's side-stream pass
    assert [engine.stft_db_fetch(i).tobytes() for i in range(len(clips))] == ref


@pytest.mark.gpu
def test_stft_db_is_finished_where_it_is_consumed(engine, synth16k, monkeypatch):
    """Round 6: ``pce_stft_db_run`` leaves the matrix as the FFT pass wrote it (raw dB + the clip's maximum: one write instead of write, read,
    rewrite); ``ref = max`` / the -80 dB floor are applied by whoever takes the values.  A fetch finishes ONE clip through a staging buffer and leaves
    the resident matrix raw (fetching twice, in any order, gives the same bytes); ``pce_stft_db_device`` finishes the whole matrix once, in place, and
    fetches after it read the finished values as they are.  All three are the two-FFT form's bytes."""
    import ctypes as C
    clips = synth16k
    monkeypatch.setenv("PCE_STFT_TWO_FFT", "1")
    with pkg.ProsodyEngine(0) as two:
        two.upload(clips, 16000); two.stft_db_run(1024, 256)
        ref = [two.stft_db_fetch(i) for i in range(len(clips))]
    monkeypatch.delenv("PCE_STFT_TWO_FFT")
    engine.upload(clips, 16000)
    engine.stft_db_run(1024, 256)
    order = [3, 0, 3, len(clips) - 1, 1, 0]
    for i in order:                                             # lazy, per clip, repeatable
        assert engine.stft_db_fetch(i).tobytes() == ref[i].tobytes(), i
    ptr, nbytes = engine.stft_db_device()                       # the finished matrix for a device-side consumer
    assert nbytes == 4 * sum(r.size for r in ref)
    hip = C.CDLL("libamdhip64.so")
    host = np.empty(nbytes // 4, dtype=np.float32)
    engine.sync()
    assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(ptr), C.c_size_t(nbytes), 2) == 0        # hipMemcpyDeviceToHost
    assert host.tobytes() == b"".join(r.tobytes() for r in ref)
    ptr2, _ = engine.stft_db_device()                           # a second request does not normalise twice
    assert ptr2 == ptr
    for i in (0, len(clips) - 1, 2):
        assert engine.stft_db_fetch(i).tobytes() == ref[i].tobytes(), i
    assert (ref[0] <= 0.0).all() and ref[0].max() == 0.0 and ref[0].min() >= -80.0


@pytest.mark.gpu
def test_pitch_reuses_the_energy_pass_only_for_the_same_slices(engine, synth16k):
    """pce_pitch_run reads the slice sums / extrema from pce_energy_run's accumulators when that call covered the very
    same slice list (one stream over the batch saved); any other order or slice list takes its own pass.  Same bits."""
    clips = synth16k
    p = pkg.PitchParams.praat(150.0, 600.0)
    engine.upload(clips, 16000)
    sl = engine.whole_clip_slices()

    def pitch():
        engine.pitch_run(sl, p)
        r = engine.pitch_fetch(want_f0=True, want_strength=True)
        return r["f0"].tobytes(), r["strength"].tobytes(), r["summary"].tobytes()

    alone = pitch()                                            # no energy pass before: own accumulators
    engine.energy_run(sl, 500)
    assert pitch() == alone                                    # reused
    half = sl.copy(); half["end"] = half["end"] // 2
    engine.energy_run(half, 500)
    assert pitch() == alone                                    # different slice list in the energy pass: not reused
    engine.energy_run(sl, 123)                                 # another gate threshold does not matter for sums / extrema
    assert pitch() == alone
    assert engine.energy_fetch()["sum_sq"].tolist() == [int((c.astype(np.int64) ** 2).sum()) for c in clips]


@pytest.mark.gpu
def test_c4_shard_size_batch_equals_its_sub_batches():
    """BASELINE.json C4: 10 000 clips over 8 GPUs = 1 250 ten-second clips per rank in ONE resident batch.  Every
    per-utterance result must equal what five sub-batches of 250 give (utterances are independent: the batch size,
    the work lists and the XCD tile order may not leak into the numbers)."""
    import hashlib
    n, part = 1250, 250
    clips = synth.synth_batch(n, 10.0, 16000, first=5000)
    p = pkg.PitchParams.praat(150.0, 600.0)

    def measure(eng, batch):
        eng.upload(batch, 16000); sl = eng.whole_clip_slices()
        eng.energy_run(sl, 500); eng.lufs_run(sl); eng.pitch_run(sl, p); eng.stft_db_run(1024, 256); eng.frame_energy_run(800)
        en = eng.energy_fetch(); lu = eng.lufs_fetch()[0]; pi = eng.pitch_fetch(want_f0=True)
        picks = range(0, len(batch), 50)
        stft = [hashlib.sha256(eng.stft_db_fetch(i).tobytes()).hexdigest() for i in picks]
        fr = [eng.frame_energy_fetch(i)[0].tobytes() for i in picks]
        off = pi["frame_offsets"]
        f0 = [pi["f0"][off[i]:off[i + 1]].tobytes() for i in range(len(batch))]
        return en.tobytes(), lu.tobytes(), pi["summary"]["median_f0"].tobytes(), pi["summary"]["n_voiced"].tobytes(), f0, stft, fr

    with pkg.ProsodyEngine(0) as eng:
        whole = measure(eng, clips)
        parts = [measure(eng, clips[k:k + part]) for k in range(0, n, part)]
    for j in range(4):
        assert whole[j] == b"".join(pt[j] for pt in parts), j
    for j in (4, 5, 6):
        assert whole[j] == [x for pt in parts for x in pt[j]], j


@pytest.mark.gpu
def test_environment_switches_of_pce_create(engine, monkeypatch):
    """Every environment variable pce_create still reads selects code a test runs (PCE_NO_AUX, PCE_STFT_TWO_FFT, PCE_EN_CPB,
    PCE_ALIGN_GENERIC_MEDIAN have their own tests); here the remaining four:
    PCE_WHISPER_OPERANDS=bf16 / PCE_PITCH_REFINE=praat -> the context's defaults for what pce_whisper_set_operands / pce_pitch_set_refine
    set; PCE_GEMM_FLAT=0 -> the encoder's projections on the 128 x 128 / 128 x 256 tile kernels instead of the persistent 256 x 256 one
    (same products: outputs agree to the rounding of a 16-bit store); PCE_GEMM_SKINNY=0 -> the few-row GEMMs of an incremental decoding
    step on the tiled kernel (the same tokens, log-probabilities within 2e-3, from the device-resident loop)."""
    from prosody_control_french_tts_amd import synth, whisper_weights as WW
    from prosody_control_french_tts_amd.Aligners import decoding as DEC
    from prosody_control_french_tts_amd.Aligners.tokenizer import WhisperTokenizer
    monkeypatch.setenv("PCE_WHISPER_OPERANDS", "bf16"); monkeypatch.setenv("PCE_PITCH_REFINE", "praat")
    clips = [synth.synth_clip(i, seconds=3.0) for i in range(2)]
    p = pkg.PitchParams.praat(150.0, 600.0)
    with pkg.ProsodyEngine(0) as eng:
        assert eng.whisper_operands == "bf16"
        eng.upload(clips, 16000)
        praat = eng.pitch(eng.whole_clip_slices(), p, want_f0=True)["f0"]
    monkeypatch.delenv("PCE_WHISPER_OPERANDS"); monkeypatch.delenv("PCE_PITCH_REFINE")
    assert engine.whisper_operands == "fp16-resid16"           # the default: fp16 operands + fp16 residual stream
    engine.upload(clips, 16000)
    engine.pitch_set_refine("praat")
    try:
        assert engine.pitch(engine.whole_clip_slices(), p, want_f0=True)["f0"].tobytes() == praat.tobytes()      # the env default IS the API's mode
    finally:
        engine.pitch_set_refine("seeded")
    # ---- the two GEMM routing switches, on a model wide enough for the persistent kernel (n_state % 256 == 0, M = 2 x 1500 rows)
    edims = dict(n_mels=80, n_ctx=1500, n_state=256, n_head=4, n_layer=2)
    tk = WhisperTokenizer.toy([b" b", b"on", b" bon", b"jo", b"ur"], language="fr")
    tdims = dict(n_vocab=tk.n_vocab, n_text_ctx=64, n_state=256, n_head=4, n_layer=2)
    We, Wd = WW.synthetic_weights(edims, seed=3), WW.greedy_test_decoder_weights(tdims, seed=4)
    rules = tk.decoding_rules()
    prompts = [list(tk.sot_sequence())] * 2

    def run(eng):
        eng.upload(clips, 16000); eng.logmel_run(80)
        eng.whisper_load(edims, WW.pack(We, edims)); eng.whisper_encode_run()
        enc = [eng.whisper_encode_fetch(i) for i in range(2)]
        eng.whisper_decoder_load(tdims, WW.pack_decoder(Wd, tdims))
        toks, lps, sums = DEC.decode_batch(eng, tdims["n_vocab"], prompts, [len(q) for q in prompts], rules, 12)
        return enc, toks, lps

    base = run(engine)
    for var in ("PCE_GEMM_FLAT", "PCE_GEMM_SKINNY"):
        monkeypatch.setenv(var, "0")
        with pkg.ProsodyEngine(0) as other:
            got = run(other)
        monkeypatch.delenv(var)
        for a, b in zip(got[0], base[0]):
            assert np.isfinite(a).all() and np.linalg.norm(a - b) / np.linalg.norm(b) <= 2e-3, var
        if var == "PCE_GEMM_SKINNY":
            # (the two kernels sum identically; the compiler schedules the inlined GELU differently in them, and a few 1e-5 of its fp16 outputs
            # round to the neighbouring value: the same tokens, log-probabilities to the last digits)
            assert got[1] == base[1] and all(np.allclose(x, y, rtol=0, atol=2e-3) for x, y in zip(got[2], base[2]))


@pytest.mark.gpu
def test_levenshtein_gpu_reproduces_the_reference_distances(engine):
    """pce_levenshtein (one wave per pair, the whole batch in one launch) against golden G9 = the outputs of the reference's own
    levenshtein_distance (Code/Aligners/levenshtein_dist_align_txtgrids.py:43-70): bit-exact, both argument orders, plus seeded random
    pairs of every stripe count against the CPU restatement, a 5 000-character pair (79 stripes), and the merge loop on the GPU."""
    import json, os
    from oracle import oracle as O
    from prosody_control_french_tts_amd.Aligners import levenshtein_dist_align_txtgrids as LV
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "levenshtein.json"), encoding="utf-8"))["cases"]
    pairs = [(c["s1"], c["s2"]) for c in cases]
    got = engine.levenshtein(pairs)
    assert got.dtype == np.int32 and got.tolist() == [c["distance"] for c in cases]
    assert engine.levenshtein([(b, a) for a, b in pairs]).tolist() == [c["distance"] for c in cases]
    assert LV.levenshtein_distance("voilà", "voila", engine) == 1 and LV.levenshtein_distances([], engine) == []
    rng = np.random.default_rng(11)
    extra = []
    for n in list(range(0, 9)) + [62, 63, 64, 65, 66, 127, 128, 129, 191, 192, 193, 320, 1000]:
        for m in (0, 1, 5, 64, 65, 130, n):
            for alpha in ("ab", "abcdefghéà "):
                extra.append(("".join(rng.choice(list(alpha), n)) if n else "", "".join(rng.choice(list(alpha), m)) if m else ""))
    got = engine.levenshtein(extra)
    assert got.tolist() == [O.levenshtein(a, b) for a, b in extra]
    big = "".join(rng.choice(list("abcde"), 5000)); big2 = big[:1700] + "xyz" + big[1750:4000] + big[4100:]
    assert engine.levenshtein([(big, big2), (big2, big), (big, big)]).tolist() == [O.levenshtein(big, big2)] * 2 + [0]
    # the (terminating) merge loop with the distances from the GPU = with the distances from the CPU restatement
    class CpuDistances:
        def levenshtein(self, ps):
            return np.array([O.levenshtein(a, b) for a, b in ps], dtype=np.int32)
    words = ["bon", "jour", "bonjour", "le", "la", "monde", "mondes", " ", "", "très", "tres", "bien", "oui", "non", "peut", "être", "peut-être"]
    def tier(k):
        t, out = 0.0, []
        for _ in range(k):
            d = float(rng.integers(5, 40)) / 100.0
            out.append((t, t + d, str(rng.choice(words)))); t += d
        return out
    tiers = [(tier(int(rng.integers(1, 30))), tier(int(rng.integers(1, 30)))) for _ in range(40)]
    assert LV.merge_word_tiers(tiers, engine) == LV.merge_word_tiers(tiers, CpuDistances())


@pytest.mark.gpu
def test_whole_recording_of_two_minutes_has_no_frame_limit(engine):
    """get_median_pitch takes any recording (Code/audioPipeline.py:326-335: no length limit).  A 120 s clip = 23 997 frames at floor 150,
    more than the in-LDS median sort holds (16 384): the slice goes to k_pitch_median_long (radix selection in global memory).  Track,
    voiced median (np.median) and mean log against the oracle; short slices of the same batch still take the LDS kernel; odd and even
    voiced counts both occur."""
    from prosody_control_french_tts_amd import synth, PitchParams
    from oracle import oracle as O
    long_clip = np.concatenate([synth.synth_clip(900 + i, seconds=10.0) for i in range(12)])
    short = synth.synth_clip(3, seconds=3.0)
    engine.upload([long_clip, short], 16000)
    from prosody_control_french_tts_amd.engine import make_slices
    # whole long clip, the short clip, the long clip minus its first 7 ms (another voiced count: the other parity with luck, checked below)
    sl = make_slices([0, 1, 0, 0], [0, 0, 112, 0], [len(long_clip), len(short), len(long_clip), 90 * 16000], [0.5 / 16000, 0.5 / 16000, 112.5 / 16000, 0.5 / 16000])
    res = engine.pitch(sl, PitchParams.praat(150.0, 600.0))
    summ, off = res["summary"], res["frame_offsets"]
    assert off[1] - off[0] == 23997 and off[1] - off[0] > 16384
    parities = set()
    for k, (clip, b, e) in enumerate([(long_clip, 0, len(long_clip)), (short, 0, len(short)), (long_clip, 112, len(long_clip)), (long_clip, 0, 90 * 16000)]):
        want = O.pitch_ac(clip[b:e] / 32768.0, 1 / 16000, float(sl[k]["x1"]), O.praat_params(150.0, 600.0))["f0"]
        got = res["f0"][off[k]:off[k + 1]]
        assert got.shape == want.shape and np.array_equal(got > 0, want > 0)
        v = want > 0
        assert np.max(np.abs(got[v] - want[v]) / want[v]) < 1e-6
        assert summ[k]["n_voiced"] == int(v.sum())
        # the median is an ELEMENT (or the mean of two) of the GPU's own track: exact against np.median of that track
        assert summ[k]["median_f0"] == float(np.median(got[got > 0]))
        assert abs(summ[k]["median_f0"] - float(np.median(want[v]))) <= 1e-6 * float(np.median(want[v]))
        assert abs(summ[k]["mean_log_f0"] - float(np.mean(np.log(got[got > 0])))) <= 1e-12 * abs(float(np.mean(np.log(got[got > 0]))))
        if k != 1:
            parities.add(int(v.sum()) & 1)
    assert parities == {0, 1}, parities


@pytest.mark.gpu
def test_round5_edge_cases_of_the_unlimited_paths(engine):
    """Corners of the paths that lost their limits in round 5: a long slice with NO voiced frame (k_pitch_median_long with nothing to select), an
    exact multiple of the Needleman-Wunsch stripe height against a single column, 20 000 word pairs in one Levenshtein launch, lone surrogates."""
    from prosody_control_french_tts_amd import PitchParams
    from prosody_control_french_tts_amd.engine import make_slices
    from prosody_control_french_tts_amd.Pipeline import NeedlemanWunschAlignement as NW
    from oracle import oracle as O
    rng = np.random.default_rng(2)
    quiet = np.zeros(100 * 16000, dtype=np.int16)                       # 100 s of digital silence: 19 997 frames, all unvoiced
    quiet[::4001] = 3                                                    # (not identically zero: the frames are analysed, none passes the voicing threshold)
    engine.upload([quiet], 16000)
    res = engine.pitch(make_slices([0], [0], [len(quiet)], [0.5 / 16000]), PitchParams.praat(150.0, 600.0))
    assert res["frame_offsets"][1] == 19997 and res["summary"][0]["n_voiced"] == 0
    assert res["summary"][0]["median_f0"] == 0.0 and res["summary"][0]["mean_log_f0"] == 0.0 and not (res["f0"] > 0).any()
    words = ["a", "b", "c", "d."]
    mk = lambda k: [(str(i), words[int(rng.integers(len(words)))], float(i), float(i) + 0.5, 0.5) for i in range(k)]
    for n, m in [(2048, 1), (1, 2048), (1024, 1024)]:
        a, b = mk(n), mk(m)
        (g,) = NW.needleman_wunsch_batch([(a, b)], engine)
        assert NW.format_alignment(g) == NW.format_alignment(NW.needleman_wunsch(a, b)), (n, m)
    vocab = ["le", "la", "les", "monde", "mondes", "bonjour", "bon", "jour", "été", "etait", "cœur", "coeur", ""]
    pairs = [(vocab[int(i)], vocab[int(j)]) for i, j in rng.integers(0, len(vocab), size=(20000, 2))]
    got = engine.levenshtein(pairs)
    table = {(a, b): O.levenshtein(a, b) for a in vocab for b in vocab}
    assert got.tolist() == [table[p] for p in pairs]
    lone = [("a\ud800b", "ab"), ("\udfff", ""), ("x\U0001F600y", "x\ud83dy")]     # surrogates are code points like any other to Python's str
    assert engine.levenshtein(lone).tolist() == [O.levenshtein(a, b) for a, b in lone] == [1, 1, 1]
