"""GPU parity tests: libpce (HIP, through the C ABI) vs the CPU oracle on the same inputs.

Tolerances (stated per test): integer energy/gate fields bit-exact; R3 RMS-dB bit-exact;
F0 <= 1e-6 relative with identical voiced/unvoiced decisions; LUFS <= 1e-6 LU (fp64 with a
different summation order than numpy's pairwise sum); STFT-dB <= 2e-2 dB where the value
is above the -80 dB floor (float32 FFT), floor membership identical within that band.
"""
import numpy as np
import pytest

from oracle import oracle as O
from prosody_control_french_tts_amd import engine as E
from prosody_control_french_tts_amd import hostrules as H

pytestmark = pytest.mark.gpu


def _slices_for(clips, rate, rng, per_clip=6):
    cl, b, e = [], [], []
    for i, c in enumerate(clips):
        n = len(c)
        cl += [i]; b += [0]; e += [n]
        for _ in range(per_clip):
            x0 = int(rng.integers(-200, n)); x1 = int(rng.integers(x0, n + 400))
            cl.append(i); b.append(x0); e.append(x1)
        cl.append(i); b.append(5); e.append(5)            # empty
    return E.make_slices(cl, b, e)


def _materialise(clip, b, e):
    out = np.zeros(e - b, dtype=np.int16)
    lo, hi = max(b, 0), min(e, len(clip))
    if hi > lo:
        out[lo - b:hi - b] = clip[lo:hi]
    return out


def test_energy_bit_exact(engine, synth16k):
    rng = np.random.default_rng(1)
    engine.upload(synth16k, 16000)
    sl = _slices_for(synth16k, 16000, rng)
    got = engine.energy(sl, 500)
    for s, g in zip(sl, got):
        x = _materialise(synth16k[s["clip"]], int(s["begin"]), int(s["end"]))
        x64 = x.astype(np.int64)
        assert g["n"] == len(x)
        assert g["sum_sq"] == int(np.sum(x64 * x64))
        with np.errstate(over="ignore"):
            assert g["sum_sq_wrap16"] == int(np.sum((x ** 2).astype(np.int64)))
            assert g["n_loud"] == int(np.sum(np.abs(x) > 500))
        assert g["peak_abs"] == (int(np.max(np.abs(x64))) if len(x) else 0)
        if len(x):
            # R3 finishing math is bit-exact, R7 verdict identical
            assert H.rms_db_from_wrapped(int(g["sum_sq_wrap16"]), len(x)) == O.rms_db_int16_wrapped(x) or \
                (np.isnan(H.rms_db_from_wrapped(int(g["sum_sq_wrap16"]), len(x))) and np.isnan(O.rms_db_int16_wrapped(x)))
            rms, ratio, ok = H.gate_from_counts(int(g["sum_sq"]), int(g["n_loud"]), len(x))
            orms, oratio, ook = O.gate_check(x)
            assert ratio == oratio and ok == ook
            assert abs(float(rms) - float(orms)) <= 2e-6 * max(1.0, float(orms))


def test_lufs_matches_oracle(engine, synth16k):
    rng = np.random.default_rng(2)
    engine.upload(synth16k, 16000)
    sl = _slices_for(synth16k, 16000, rng)
    got, st = engine.lufs(sl)
    n_checked = 0
    for s, g, code in zip(sl, got, st):
        x = _materialise(synth16k[s["clip"]], int(s["begin"]), int(s["end"])).astype(float)
        if len(x) == 0:
            assert code == E.SLICE_EMPTY
            continue
        try:
            want = O.lufs_numpy(x, 16000)
        except ValueError:
            assert code == E.SLICE_TOO_SHORT
            continue
        assert code == E.SLICE_OK
        if np.isinf(want):
            assert np.isinf(g) and g < 0
        else:
            assert abs(g - want) <= 1e-6, (g, want)       # LU
        n_checked += 1
    assert n_checked >= 8


@pytest.mark.parametrize("floor,ceiling", [(150.0, 600.0), (75.0, 600.0), (100.0, 600.0), (200.0, 600.0)])
def test_pitch_matches_oracle(engine, synth16k, floor, ceiling):
    rng = np.random.default_rng(3)
    rate = 16000
    engine.upload(synth16k, rate)
    cl, b, e, x1 = [], [], [], []
    for i, c in enumerate(synth16k):
        cl.append(i); b.append(0); e.append(len(c)); x1.append(0.5 / rate)
        for _ in range(3):
            t0 = float(rng.uniform(0, len(c) / rate * 0.8)); t1 = t0 + float(rng.uniform(0.005, 1.2))
            bb, ee, xx = H.praat_part_frames(len(c), rate, t0, t1, preserve_times=True)
            cl.append(i); b.append(bb); e.append(ee); x1.append(xx)
    sl = E.make_slices(cl, b, e, x1)
    res = engine.pitch(sl, E.PitchParams.praat(floor, ceiling), want_f0=True, want_strength=True)
    off = res["frame_offsets"]; summ = res["summary"]
    n_ok = 0
    for k, s in enumerate(sl):
        x = _materialise(synth16k[s["clip"]], int(s["begin"]), int(s["end"])).astype(np.float64) / 32768.0
        try:
            want = O.pitch_ac(x, 1.0 / rate, float(s["x1"]), O.praat_params(floor, ceiling))
        except O.PraatError:
            assert summ[k]["status"] == E.SLICE_TOO_SHORT and off[k + 1] == off[k]
            continue
        assert summ[k]["status"] == E.SLICE_OK
        f0 = res["f0"][off[k]:off[k + 1]]; sg = res["strength"][off[k]:off[k + 1]]
        assert len(f0) == len(want["f0"]) == summ[k]["n_frames"]
        assert summ[k]["t1"] == want["plan"].t1
        assert np.array_equal(f0 > 0, want["f0"] > 0), "voiced/unvoiced decisions differ"
        v = want["f0"] > 0
        if v.any():
            assert np.max(np.abs(f0[v] - want["f0"][v]) / want["f0"][v]) <= 1e-6
            assert np.max(np.abs(sg[v] - want["strength"][v])) <= 1e-6
            med = float(np.median(want["f0"][v]))
            assert abs(summ[k]["median_f0"] - med) <= 1e-6 * med
            assert abs(summ[k]["mean_log_f0"] - float(np.mean(np.log(want["f0"][v])))) <= 1e-6   # = the F0 tolerance
        else:
            assert summ[k]["median_f0"] == 0.0
        assert summ[k]["n_voiced"] == int(v.sum())
        n_ok += 1
    assert n_ok >= len(synth16k)


def test_stft_db_matches_oracle(engine, synth16k):
    engine.upload(synth16k, 16000)
    engine.stft_db_run(1024, 256)
    for i, c in enumerate(synth16k):
        got = engine.stft_db_fetch(i)
        want = O.stft_db(c.astype(np.float32) / 32768.0)
        assert got.shape == want.shape == (513, 1 + len(c) // 256)
        assert got.max() <= 1e-6 and got.min() >= -80.0
        live = (want > -79.9) & (got > -79.9)
        assert np.max(np.abs(got[live] - want[live])) <= 2e-2
        # values at the floor agree up to the same band
        assert np.max(np.abs(got - want)) <= 0.25


def test_c1_real_speech_clip_all_measurements(engine):
    """BASELINE.json configs[0] (C1): the reference's demo recording segment_ph9 at 16 kHz, 5.000 s
    (tests/golden/c1_segment_ph9_16k.npz: resampled with the engine's own polyphase spec, zero padded).
    Real speech through every kernel of the prosody path, against the oracle at the tolerances above."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "c1_segment_ph9_16k.npz"))
    pcm, rate = z["pcm"], int(z["rate"])
    assert rate == 16000 and len(pcm) == 80000
    engine.upload([pcm], rate)
    sl = engine.whole_clip_slices()
    en = engine.energy(sl, 500)[0]
    x64 = pcm.astype(np.int64)
    assert en["sum_sq"] == int(np.sum(x64 * x64)) and en["peak_abs"] == int(np.max(np.abs(x64)))
    rms, ratio, ok = H.gate_from_counts(int(en["sum_sq"]), int(en["n_loud"]), len(pcm))
    assert (ratio, ok) == O.gate_check(pcm)[1:]
    lu, st = engine.lufs(sl)
    assert st[0] == E.SLICE_OK and abs(lu[0] - O.lufs_numpy(pcm.astype(float), rate)) <= 1e-6
    res = engine.pitch(sl, E.PitchParams.praat(150.0, 600.0), want_f0=True)
    want = O.pitch_ac(pcm.astype(np.float64) / 32768.0, 1.0 / rate, 0.5 / rate, O.praat_params(150.0, 600.0))
    assert len(res["f0"]) == len(want["f0"]) == 997           # 997 frames at 5 s
    v = want["f0"] > 0
    assert np.array_equal(res["f0"] > 0, v) and v.sum() > 100
    assert np.max(np.abs(res["f0"][v] - want["f0"][v]) / want["f0"][v]) <= 1e-6
    assert abs(res["summary"][0]["median_f0"] - float(np.median(want["f0"][v]))) <= 1e-6 * float(np.median(want["f0"][v]))
    engine.stft_db_run(1024, 256)
    got = engine.stft_db_fetch(0); ws = O.stft_db(pcm.astype(np.float32) / 32768.0)
    assert got.shape == ws.shape == (513, 313)                # 513 x 313 at 5 s
    live = (ws > -79.9) & (got > -79.9)
    assert np.max(np.abs(got[live] - ws[live])) <= 2e-2


@pytest.mark.parametrize("rate,floor", [(48000, 150.0), (22050, 75.0), (44100, 100.0), (8000, 75.0)])
def test_pitch_other_rates_match_oracle(engine, rate, floor):
    """The transform path depends on N = nsampFFT: 2048 (register path, 16 points per lane) at 48 kHz / 150 Hz,
    22.05 kHz / 75 Hz and 44.1 kHz / 100 Hz; 512 at 8 kHz / 75 Hz.  Same tolerance as the 16 kHz test."""
    from prosody_control_french_tts_amd import synth
    clips = [synth.synth_clip(i, 2.0, rate) for i in range(3)]
    engine.upload(clips, rate)
    res = engine.pitch(engine.whole_clip_slices(), E.PitchParams.praat(floor, 600.0), want_f0=True, want_strength=True)
    off = res["frame_offsets"]
    for i, c in enumerate(clips):
        want = O.pitch_ac(c.astype(np.float64) / 32768.0, 1.0 / rate, 0.5 / rate, O.praat_params(floor, 600.0))
        f0 = res["f0"][off[i]:off[i + 1]]; sg = res["strength"][off[i]:off[i + 1]]
        v = want["f0"] > 0
        assert len(f0) == len(want["f0"]) and np.array_equal(f0 > 0, v) and v.sum() > 10
        assert np.max(np.abs(f0[v] - want["f0"][v]) / want["f0"][v]) <= 1e-6
        assert np.max(np.abs(sg[v] - want["strength"][v])) <= 1e-6


@pytest.mark.parametrize("rate", [48000, 16000])
def test_lufs_ebu_tech_3341_cases_on_the_gpu(engine, rate):
    """The five EBU Tech 3341 integrated-loudness cases (tests/test_oracle_kat.py: schedule, published targets and their
    translation to this mono, peak-normalised path) through pce_lufs_run: within the document's +-0.1 LU of the target and
    within 1e-6 LU of the oracle."""
    from tests.test_oracle_kat import EBU_3341, ebu_3341_signal
    sigs = [ebu_3341_signal(c, rate) for c in sorted(EBU_3341)]
    engine.upload([s for s, _ in sigs], rate)
    vals, st = engine.lufs(engine.whole_clip_slices())
    for (x, want), v, code in zip(sigs, vals, st):
        assert code == 0 and abs(v - want) <= 0.1, (rate, v, want)
        assert abs(v - O.lufs_c(x.astype(np.float64), rate)) <= 1e-6


def test_refine_modes_give_the_same_f0_tracks(engine, synth16k):
    """``pce_pitch_set_refine``: the default candidate search (seeded parabolic interpolation, Praat's iterates only where the two could
    differ) against NUMminimize_brent's own iterates for every candidate (PCE_REFINE_PRAAT), on real speech (C1, the 44.1 kHz demo
    excerpts) and the synthetic set.  Compared are the FULL F0 tracks after the path finder -- which candidate wins every frame, not
    only the candidates' values: the same voiced / unvoiced decision in every frame, F0 within 1e-6 relative, strengths within 1e-6,
    the medians the tagger consumes within 1e-6 relative; and the praat mode itself against the oracle."""
    import os
    g = os.path.join(os.path.dirname(__file__), "golden")
    c1 = np.load(os.path.join(g, "c1_segment_ph9_16k.npz"))["pcm"]
    z = np.load(os.path.join(g, "demo_excerpts.npz"))
    sets = [([c1] + list(synth16k[:4]), 16000), ([z[k] for k in sorted(z.files) if k != "rate"][:5], int(z["rate"]))]
    try:
        for clips, rate in sets:
            engine.upload(clips, rate)
            sl = engine.whole_clip_slices()
            tracks = {}
            for mode in ("seeded", "praat"):
                engine.pitch_set_refine(mode)
                tracks[mode] = engine.pitch(sl, E.PitchParams.praat(150.0, 600.0), want_f0=True, want_strength=True)
            a, b = tracks["seeded"], tracks["praat"]
            assert np.array_equal(a["frame_offsets"], b["frame_offsets"])
            va, vb = a["f0"] > 0, b["f0"] > 0
            assert np.array_equal(va, vb) and va.sum() > 200                       # the path finder took the same decisions
            assert np.max(np.abs(a["f0"][va] - b["f0"][va]) / b["f0"][va]) <= 1e-6
            assert np.max(np.abs(a["strength"][va] - b["strength"][va])) <= 1e-6
            for sa, sb in zip(a["summary"], b["summary"]):
                assert sa["n_voiced"] == sb["n_voiced"] and abs(sa["median_f0"] - sb["median_f0"]) <= 1e-6 * max(sb["median_f0"], 1.0)
            want = O.pitch_ac(clips[0].astype(np.float64) / 32768.0, 1.0 / rate, 0.5 / rate, O.praat_params(150.0, 600.0))["f0"]
            got = b["f0"][b["frame_offsets"][0]:b["frame_offsets"][1]]
            v = want > 0
            assert np.array_equal(got > 0, v) and np.max(np.abs(got[v] - want[v]) / want[v]) <= 1e-6
    finally:
        engine.pitch_set_refine("seeded")
    with pytest.raises(Exception):
        engine._check(engine._lib.pce_pitch_set_refine(engine._ctx, 7))
