"""Host-side rules the reference applies *around* its numeric kernels.

These are integer / index conversions and a few scalar finishing operations: WAV
decoding, pydub's millisecond slicing, Praat's ``extract_part`` sample selection,
and the sqrt/log10 that turn the engine's exact integer sums into the reference's
floats.  No array arithmetic happens here: that is the GPU's job (``libpce.so``).
"""
from __future__ import annotations

import math
import os
import wave

import numpy as np


class CouldntDecodeError(Exception):
    """Mirror of ``pydub.exceptions.CouldntDecodeError`` (caught at Code/audioPipeline.py:385,506)."""


class PraatError(RuntimeError):
    """Mirror of ``parselmouth.PraatError``: raised where Praat refuses the analysis."""


def decode_wav(path):
    """Decode a 16-bit PCM RIFF/WAVE file -> (frame_rate, int16 mono samples).

    Stands in for ``AudioSegment.from_file`` / ``parselmouth.Sound(path)`` /
    ``scipy.io.wavfile.read`` (Code/audioPipeline.py:319,327,340;
    Code/Aligners/use_whisper_timestamped.py:199).  Multi-channel files keep the first
    channel, as the aligner's gate does (use_whisper_timestamped.py:201-202)."""
    try:
        with wave.open(os.fspath(path), "rb") as w:
            if w.getsampwidth() != 2 or w.getcomptype() != "NONE":
                raise CouldntDecodeError(f"{path}: only 16-bit PCM WAV is supported")
            rate, ch, n = w.getframerate(), w.getnchannels(), w.getnframes()
            data = np.frombuffer(w.readframes(n), dtype="<i2")
    except (wave.Error, EOFError, OSError) as e:
        raise CouldntDecodeError(f"{path}: {e}") from e
    if ch > 1:
        data = data.reshape(-1, ch)[:, 0]
    return rate, np.ascontiguousarray(data, dtype=np.int16)


# ---------------------------------------------------------------- pydub 0.25.1
def pydub_len_ms(n_frames: int, rate: int) -> int:
    return round(1000 * (float(n_frames) / rate))


def pydub_slice_frames(n_frames: int, rate: int, start_ms, stop_ms):
    """``AudioSegment[start_ms:stop_ms]`` as a sample range -> (begin, end) with
    ``end - begin`` samples of which those beyond ``n_frames`` are silence.

    pydub: ms are clamped to ``len(seg)``, converted with ``int(ms * (rate / 1000.0))``,
    and up to 2 ms of missing frames are filled with silence when at least one real
    frame exists."""
    L = pydub_len_ms(n_frames, rate)
    start = 0 if start_ms is None else start_ms
    end = L if stop_ms is None else stop_ms
    start, end = min(start, L), min(end, L)

    def pos(v):
        if v < 0:
            v = L - abs(v)
        return int(v * (rate / 1000.0))

    s, e = pos(start), pos(end)
    s_c = min(max(s, 0), n_frames) if s >= 0 else max(n_frames + s, 0)
    e_c = min(max(e, 0), n_frames) if e >= 0 else max(n_frames + e, 0)
    n_real = max(e_c - s_c, 0)
    missing = (e - s) - n_real
    pad = 0
    if missing:
        if missing > 2 * (rate / 1000.0):
            raise ValueError("TooManyMissingFrames")
        if n_real > 0:
            pad = max(missing, 0)
    return s_c, s_c + n_real + pad


def seconds_slice_frames(n_frames: int, rate: int, t0: float, t1):
    """The ``audio[int(t0*1000):int(t1*1000)]`` of Code/audioPipeline.py:321,342."""
    if t1 is None:
        return 0, n_frames
    return pydub_slice_frames(n_frames, rate, int(t0 * 1000), int(t1 * 1000))


# ---------------------------------------------------------------- Praat extract_part
def praat_part_frames(n_frames: int, rate: int, t0: float, t1: float, preserve_times: bool):
    """Sample selection of ``Sound.extract_part(t0, t1, preserve_times=...)`` with the
    default rectangular window -> (begin, end, x1): 0-based [begin, end) in file
    coordinates (virtual samples outside the file are zero) and the time of sample
    ``begin`` in the extracted Sound."""
    dx = 1.0 / rate
    x1 = 0.5 * dx
    if t0 == t1:
        t0, t1 = 0.0, n_frames * dx
    ix1 = 1 + math.ceil((t0 - x1) / dx)
    ix2 = 1 + math.floor((t1 - x1) / dx)
    if ix2 < ix1:
        raise PraatError("Extracted Sound would contain no samples.")
    x1n = x1 + (ix1 - 1) * dx
    if not preserve_times:
        x1n -= t0
    return ix1 - 1, ix2, x1n


# ---------------------------------------------------------------- finishing math
def rms_db_from_wrapped(sum_sq_wrap16: int, n: int) -> float:
    """Code/Pipeline/compute_loudness_adjustments.py:19-24 from the exact sum of
    int16-wrapped squares: ``20*log10(sqrt(abs(mean(S))))`` (numpy semantics, -inf for 0,
    nan for an empty slice)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        mean = np.float64(sum_sq_wrap16) / np.float64(n)
        return float(20 * np.log10(np.sqrt(np.abs(mean))))


def gate_from_counts(sum_sq: int, n_loud: int, n: int):
    """Code/Aligners/use_whisper_timestamped.py:204-210 from exact integers:
    (rms as float32, silence_ratio, ok)."""
    rms = np.float32(math.sqrt(sum_sq / n)) if n else np.float32("nan")
    ratio = 1.0 - (n_loud / n) if n else float("nan")
    ok = not (ratio > 0.95 or rms < 100)
    return rms, ratio, ok


# ---------------------------------------------------------------- resampling filter (host-side design)
def resample_filter(rate_in: int, rate_out: int, beta: float = 5.0):
    """Low-pass of ``scipy.signal.resample_poly(x, up, down)`` (window ('kaiser', 5.0)): returns
    (up, down, taps, n_pre_remove) with ``taps`` scaled by ``up`` and zero padded exactly as scipy does,
    so that y = upfirdn(taps, x, up, down)[n_pre_remove : n_pre_remove + ceil(n * up / down)]."""
    g = math.gcd(int(rate_in), int(rate_out))
    up, down = int(rate_out) // g, int(rate_in) // g
    max_rate = max(up, down)
    f_c = 1.0 / max_rate
    half_len = 10 * max_rate
    numtaps = 2 * half_len + 1
    m = np.arange(numtaps) - (numtaps - 1) / 2.0
    h = f_c * np.sinc(f_c * m)                                   # firwin: ideal low-pass, cutoff f_c (Nyquist = 1)
    n = np.arange(numtaps)
    alpha = (numtaps - 1) / 2.0
    w = np.i0(beta * np.sqrt(np.maximum(0.0, 1.0 - ((n - alpha) / alpha) ** 2))) / np.i0(beta)
    h = h * w
    h = h / np.sum(h)                                            # unit DC gain
    h = h * up
    n_pre_pad = down - half_len % down
    n_pre_remove = (half_len + n_pre_pad) // down
    # scipy adds post padding until the filtered length covers n_out + n_pre_remove; harmless extra zeros here
    taps = np.concatenate([np.zeros(n_pre_pad), h, np.zeros(down)])
    return up, down, taps, int(n_pre_remove)
