"""Energy VAD host logic (Aligners/vad.py) and its oracle (third party behaviour restated: parity unpinned).
Known-answer cases computable by hand; the GPU side of the same path is in tests/test_gpu_vad.py."""
import numpy as np
import pytest

from oracle import oracle as O
from prosody_control_french_tts_amd.Aligners import vad


def test_tokenizer_hand_cases():
    v = [0, 1, 1, 0, 0, 0, 1, 1, 1, 1, 0, 1, 0, 0, 0, 0]
    # two tolerated silent windows stay inside a token unless trailing silence is dropped
    assert vad.tokenize(v, 2, 100, 2) == [(1, 4), (6, 13)]
    assert vad.tokenize(v, 2, 100, 2, drop_trailing_silence=True) == [(1, 2), (6, 11)]
    # truncation at max_length: the remainder is contiguous and exempt from min_length ...
    assert vad.tokenize([1] * 10, 2, 4, 1) == [(0, 3), (4, 7), (8, 9)]
    assert vad.tokenize([1] * 9, 2, 4, 1) == [(0, 3), (4, 7), (8, 8)]
    # ... unless strict
    assert vad.tokenize([1] * 9, 2, 4, 1, strict_min_length=True) == [(0, 3), (4, 7)]
    # no silence tolerated: every valid window alone
    assert vad.tokenize([1, 0, 1, 0, 1], 1, 10, 0) == [(0, 0), (2, 2), (4, 4)]
    # a lone valid window shorter than min_length is dropped
    assert vad.tokenize([1, 0, 0, 1], 2, 10, 1) == [(0, 1)]
    assert vad.tokenize([0] * 7, 1, 10, 2) == [] and vad.tokenize([], 1, 10, 2) == []
    # a token that is only silence after truncation is not delivered
    assert vad.tokenize([1, 1, 1, 0, 0, 0, 0], 1, 3, 2) == [(0, 2)]
    with pytest.raises(ValueError):
        vad.tokenize([1], 3, 2, 0)
    with pytest.raises(ValueError):
        vad.tokenize([1], 1, 2, 2)


def test_energy_db_closed_form():
    rate, block = 16000, 800
    t = np.arange(rate) / rate
    for amp in (100.0, 1000.0, 20000.0):
        x = np.round(amp * np.sin(2 * np.pi * 400.0 * t)).astype(np.int16)     # 400 Hz: 20 periods per 50 ms window
        ss, cnt = O.frame_energy(x, block)
        db = vad.energy_db(ss, cnt)
        assert db.shape == (20,) and np.all(cnt == block)
        assert np.allclose(db, 20 * np.log10(amp / np.sqrt(2)), atol=0.02)
        assert np.array_equal(db, O.frame_energy_db(x, block))                  # integer sums -> the same float64 values
    ss, cnt = O.frame_energy(np.zeros(1000, np.int16), block)
    assert list(cnt) == [800, 200] and np.allclose(vad.energy_db(ss, cnt), -200.0)
    # the float32 round trip of whisper-timestamped: x * 32767 / 32768, truncated toward zero
    x = np.array([1000, -1000, 32767, -32768, 1, -1, 0], dtype=np.int16)
    ss, _ = O.frame_energy(x, 7, requantize=True)
    assert ss[0] == 999 ** 2 * 2 + 32766 ** 2 + 32767 ** 2


def test_vad_segments_on_a_constructed_clip():
    rate = 16000
    x = np.zeros(2 * rate, dtype=np.int16)
    t = np.arange(int(0.7 * rate)) / rate
    x[int(0.5 * rate):int(1.2 * rate)] = np.round(3000 * np.sin(2 * np.pi * 200 * t)).astype(np.int16)
    ss, cnt = O.frame_energy(x, 800, requantize=True)
    assert vad.auditok_split(ss, cnt, rate, min_dur=0.1, max_dur=2.0, max_silence=0.1, drop_trailing_silence=True) == \
        [(pytest.approx(0.5), pytest.approx(1.2))]
    segs = vad.vad_segments_from_energy(ss, cnt, len(x), rate)
    assert segs == [{"start": pytest.approx(0.0), "end": pytest.approx(1.7)}]            # dilated by 0.5 s, clamped at 0
    assert vad.vad_segments_from_energy(ss, cnt, len(x), rate, output_sample=True) == [{"start": 0, "end": 27200}]
    # two bursts 1.2 s apart stay separate; 0.9 s apart they merge after dilatation
    for gap, n_seg in ((1.2, 2), (0.9, 1)):
        y = np.zeros(4 * rate, dtype=np.int16)
        burst = np.round(3000 * np.sin(2 * np.pi * 200 * np.arange(int(0.4 * rate)) / rate)).astype(np.int16)
        y[int(0.5 * rate):int(0.9 * rate)] = burst
        b2 = int((0.9 + gap) * rate)
        y[b2:b2 + len(burst)] = burst
        ss, cnt = O.frame_energy(y, 800, requantize=True)
        assert len(vad.vad_segments_from_energy(ss, cnt, len(y), rate)) == n_seg
    # quiet audio (below 50 dB) -> nothing
    q = np.round(100 * np.sin(2 * np.pi * 200 * np.arange(rate) / rate)).astype(np.int16)
    ss, cnt = O.frame_energy(q, 800, requantize=True)
    assert vad.vad_segments_from_energy(ss, cnt, len(q), rate) == []


def test_short_audio_raises_the_message_the_aligner_retries_on():
    # Code/Aligners/use_whisper_timestamped.py:165-171: `if "max_silence" in str(e)` -> transcribe again without VAD
    x = np.full(1600, 5000, dtype=np.int16)                                      # 0.1 s: max_dur 2 windows, max_silence 0.095 s -> 1
    ss, cnt = O.frame_energy(x, 800)
    assert vad.vad_segments_from_energy(ss, cnt, len(x), 16000) == [{"start": 0.0, "end": 0.1}]
    x = np.full(800, 5000, dtype=np.int16)                                       # 0.05 s: one window, min_dur needs two
    ss, cnt = O.frame_energy(x, 800)
    with pytest.raises(ValueError):
        vad.vad_segments_from_energy(ss, cnt, len(x), 16000)
    x = np.full(2400, 5000, dtype=np.int16)                                      # 0.15 s: max_silence 2 windows, max_dur 3
    ss, cnt = O.frame_energy(x, 800)
    assert len(vad.vad_segments_from_energy(ss, cnt, len(x), 16000)) == 1
    with pytest.raises(ValueError, match="max_silence"):
        vad.auditok_split(ss, cnt, 16000, min_dur=0.05, max_dur=0.1, max_silence=0.1)
