"""GPU parity of k_frame_energy (the energy detector of the aligner's auditok VAD): exact integer window sums
against the numpy restatement, ragged and degenerate clips, overlap, the float32 round trip; then the whole VAD
(GPU sums + host tokenizer) against the host-only path."""
import numpy as np
import pytest

from oracle import oracle as O
from prosody_control_french_tts_amd import synth
from prosody_control_french_tts_amd.Aligners import vad

pytestmark = pytest.mark.gpu


def _clips():
    rng = np.random.default_rng(11)
    return [synth.synth_clip(0, seconds=3.0), synth.synth_clip(1, seconds=1.2345),
            rng.integers(-32768, 32768, size=12345).astype(np.int16),             # full-scale noise, ragged length
            np.zeros(0, dtype=np.int16), np.array([-32768], dtype=np.int16),        # empty clip, single sample
            np.full(799, 32767, dtype=np.int16), np.full(801, -32768, dtype=np.int16),
            rng.integers(-3, 4, size=4001).astype(np.int16)]


@pytest.mark.parametrize("window,hop,requant", [(800, 800, False), (800, 800, True), (2205, 2205, False), (400, 160, False), (1, 1, True), (5000, 3, False)])
def test_frame_energy_bit_exact(engine, window, hop, requant):
    clips = _clips()
    if hop == 3:
        clips = [c[:600] for c in clips]
    engine.upload(clips, 16000)
    engine.frame_energy_run(window, hop, requantize=requant)
    for i, c in enumerate(clips):
        ss, cnt = engine.frame_energy_fetch(i)
        wss, wcnt = O.frame_energy(c, window, hop, requantize=requant)
        assert ss.dtype == np.int64 and np.array_equal(ss, wss), (i, window, hop)
        assert np.array_equal(cnt, wcnt)


def test_frame_energy_argument_errors(engine):
    from prosody_control_french_tts_amd import PceError
    engine.upload([np.zeros(100, np.int16)], 16000)
    for w, h in ((0, 1), (10, 0), (10, 11)):
        with pytest.raises(PceError):
            engine.frame_energy_run(w, h)
    engine.frame_energy_run(10, 10)
    assert engine.frame_energy_fetch(0)[1].tolist() == [10] * 10
    engine.upload([np.zeros(50, np.int16)], 16000)                                # a new batch invalidates the result
    with pytest.raises(PceError):
        engine.frame_energy_fetch(0)


def test_vad_through_the_gpu_equals_host_only_path(engine):
    rate = 16000
    clips = [synth.synth_clip(i, seconds=10.0) for i in range(6)] + [np.zeros(rate, np.int16)]
    engine.upload(clips, rate)
    got = vad.get_vad_segments(engine, rate=rate)
    assert len(got) == len(clips) and got[-1] == []
    for c, g in zip(clips, got):
        ss, cnt = O.frame_energy(c, 800, requantize=True)
        want = vad.vad_segments_from_energy(ss, cnt, len(c), rate)
        assert g == want
        db = O.frame_energy_db(c, 800, requantize=True)
        assert np.array_equal(vad.energy_db(ss, cnt) >= 50.0, db >= 50.0)
    assert sum(len(g) for g in got[:-1]) >= 6                                     # the synthetic clips hold speech-like bursts


def test_full_size_batch_checksum(engine):
    """BASELINE.json C2 size (256 x 10 s): the window sums add up to the slice energy k_energy reports (two kernels,
    one exact integer), window counts add up to the clip lengths."""
    clips = synth.synth_batch(256, 10.0, 16000, first=0)
    engine.upload(clips, 16000)
    sl = engine.whole_clip_slices()
    engine.energy_run(sl, 500)
    en = engine.energy_fetch()
    engine.frame_energy_run(800)
    for i in (0, 1, 17, 128, 255):
        ss, cnt = engine.frame_energy_fetch(i)
        assert len(ss) == 200 and int(cnt.sum()) == len(clips[i]) and int(ss.sum()) == int(en["sum_sq"][i])
