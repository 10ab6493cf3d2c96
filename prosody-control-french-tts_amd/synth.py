"""Deterministic synthetic 16 kHz French-speech-like clips.

Per clip ``i`` with ``rng = np.random.default_rng(1234 + i)``: a schedule of 150-400 ms
voiced / unvoiced segments and 100-600 ms silences (exact zeros); voiced = glottal-like
harmonic complex (20 harmonics, a_k ~ 1/k) on an f0 random walk inside [150, 400] Hz,
amplitude 0.1-0.5 FS; unvoiced = white noise at -30 dBFS; int16 rounding.
"""
from __future__ import annotations

import numpy as np

BASE_SEED = 1234


def synth_clip(index: int, seconds: float = 10.0, rate: int = 16000) -> np.ndarray:
    rng = np.random.default_rng(BASE_SEED + int(index))
    n = int(round(seconds * rate))
    y = np.zeros(n, dtype=np.float64)
    pos = 0
    f0 = rng.uniform(150.0, 400.0)
    phase = 0.0
    while pos < n:
        kind = rng.choice(3, p=[0.55, 0.2, 0.25])          # voiced / unvoiced / silence
        if kind == 2:
            dur = int(rng.uniform(0.100, 0.600) * rate)
        else:
            dur = int(rng.uniform(0.150, 0.400) * rate)
        dur = min(dur, n - pos)
        if dur <= 0:
            break
        if kind == 0:
            steps = rng.normal(0.0, 0.15, size=dur)
            f = np.clip(f0 + np.cumsum(steps), 150.0, 400.0)
            f0 = float(f[-1])
            ph = phase + 2.0 * np.pi * np.cumsum(f) / rate
            phase = float(ph[-1] % (2.0 * np.pi))
            amp = rng.uniform(0.1, 0.5)
            seg = np.zeros(dur)
            for k in range(1, 21):
                seg += (1.0 / k) * np.sin(k * ph)
            seg *= amp / np.max(np.abs(seg))
            ramp = min(dur // 2, int(0.005 * rate))
            if ramp > 0:
                env = np.ones(dur); env[:ramp] = np.linspace(0, 1, ramp); env[-ramp:] = np.linspace(1, 0, ramp)
                seg *= env
            y[pos:pos + dur] = seg
        elif kind == 1:
            y[pos:pos + dur] = rng.normal(0.0, 10 ** (-30 / 20), size=dur)
        pos += dur
    return np.round(np.clip(y, -1.0, 32767.0 / 32768.0) * 32768.0).astype(np.int16)


def synth_batch(count: int, seconds: float = 10.0, rate: int = 16000, first: int = 0):
    return [synth_clip(first + i, seconds, rate) for i in range(count)]
