"""GPU parity of the probabilistic YIN (pce_pyin_*) with the numpy restatement.  The kernel evaluates the
difference function exactly (integer arithmetic on the int16 samples): against the restatement's exact mode the decoded
states must agree on >= 99.5 % of the frames and the voiced probabilities within 1e-9 (observed: all frames, 1e-15);
against librosa's float32 FFT route (the restatement's default mode) >= 97 % / 0.1."""
import numpy as np
import pytest

from oracle import pyin_oracle as PO
from prosody_control_french_tts_amd import synth
from prosody_control_french_tts_amd.visualisation import acoustic_analysis as AA

pytestmark = pytest.mark.gpu


def _clips(rate):
    rng = np.random.default_rng(5)
    t = np.arange(int(1.5 * rate)) / rate
    chirp = np.round(9000 * np.sin(2 * np.pi * (150 + 60 * t) * t) + 300 * rng.standard_normal(len(t))).astype(np.int16)
    return [synth.synth_clip(0, seconds=2.0, rate=rate) if rate != 16000 else synth.synth_clip(0, seconds=2.0), chirp,
            np.zeros(3000, dtype=np.int16), (rng.standard_normal(5000) * 2000).astype(np.int16), np.array([1, -2, 3], dtype=np.int16)]


def _compare(got, want, min_same, vp_tol):
    f0, voiced, vp = got
    wf0, wvoiced, wvp = want
    assert len(f0) == len(wf0)
    same = ((f0 == wf0) | (np.isnan(f0) & np.isnan(wf0))) & (voiced == wvoiced)
    assert same.mean() >= min_same, same.mean()
    assert np.max(np.abs(vp - wvp)) <= vp_tol, float(np.max(np.abs(vp - wvp)))


def test_pyin_matches_restatement_16k(engine):
    rate = 16000
    clips = _clips(rate)
    engine.upload(clips, rate)
    res = AA.pyin_batch(engine)
    for c, got in zip(clips, res):
        y = (c.astype(np.float32) / np.float32(32768.0))
        _compare(got, PO.pyin(y, rate, exact=True), 0.995, 1e-9)
        _compare(got, PO.pyin(y, rate), 0.97, 0.1)
    t, f0 = AA.compute_pitch(engine, clip=1)
    assert len(t) == len(f0) and np.nanmedian(f0) > 150


def test_pyin_at_44k1(engine):
    rate = 44100
    rng = np.random.default_rng(9)
    t = np.arange(int(0.7 * rate)) / rate
    tone = np.round(8000 * np.sin(2 * np.pi * 196.0 * t) + 3000 * np.sin(2 * np.pi * 392.0 * t) + 100 * rng.standard_normal(len(t))).astype(np.int16)
    clips = [tone, (rng.standard_normal(9000) * 1500).astype(np.int16)]
    engine.upload(clips, rate)
    res = AA.pyin_batch(engine)
    for c, got in zip(clips, res):
        _compare(got, PO.pyin(c.astype(np.float32) / np.float32(32768.0), rate, exact=True), 0.995, 1e-9)
    f0 = res[0][0]
    assert abs(12 * np.log2(np.nanmedian(f0) / 196.0)) < 0.06


def test_pyin_errors(engine):
    from prosody_control_french_tts_amd import PceError
    engine.upload([np.zeros(1000, np.int16)], 16000)
    plan, tables, _ = AA.pyin_plan(16000)
    with pytest.raises(PceError):
        engine.pyin_run(plan, tables[:-1])
    engine.pyin_run(plan, tables)
    st, vp, status = engine.pyin_fetch(0)
    assert len(st) == 1 + 1000 // 256 and status == 0 and np.all(st >= plan.n_pitch_bins)
    engine.upload([np.zeros(10, np.int16)], 16000)
    with pytest.raises(PceError):
        engine.pyin_fetch(0)


@pytest.mark.parametrize("rate", [8000, 22050])
def test_pyin_other_rates_take_the_generic_lag_grouping(engine, rate):
    rng = np.random.default_rng(rate)
    t = np.arange(int(0.6 * rate)) / rate
    tone = np.round(7000 * np.sin(2 * np.pi * (140.0 + 30 * t) * t) + 150 * rng.standard_normal(len(t))).astype(np.int16)
    clips = [tone, (rng.standard_normal(3000) * 900).astype(np.int16)]
    engine.upload(clips, rate)
    for c, got in zip(clips, AA.pyin_batch(engine)):
        _compare(got, PO.pyin(c.astype(np.float32) / np.float32(32768.0), rate, exact=True), 0.995, 1e-9)
