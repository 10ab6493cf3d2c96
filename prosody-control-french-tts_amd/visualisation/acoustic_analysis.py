"""``compute_pitch`` / ``compute_spectrogram`` of the reference's viewers (Code/visualisation/acoustic_analysis.py:76-113,
app.py:69-78, visualisation_abtest/app.py:102-111) on the engine's resident batch.

``librosa.pyin(audio, sr=sr, fmin=60, fmax=2000, hop_length=256)``: the per-frame YIN analysis and the HMM decoding run
on the GPU (``pce_pyin_*``); what is host logic here is the PLAN -- periods, bin counts and the constant tables
(thresholds, beta(2, 18) probabilities, Boltzmann factors, log transition rows), built with numpy so that every
constant carries the bits librosa's own numpy expressions give -- and the final state -> Hz mapping."""
from __future__ import annotations

import ctypes as C

import numpy as np

MAX_TROUGHS = 512            # PY_MAXTR of csrc/pce_pyin.hip
_TINY = np.finfo(np.float64).tiny


class PyinPlan(C.Structure):
    _fields_ = [("frame_length", C.c_int32), ("hop_length", C.c_int32), ("min_period", C.c_int32), ("max_period", C.c_int32),
                ("n_pitch_bins", C.c_int32), ("trans_width", C.c_int32), ("n_thresholds", C.c_int32), ("reserved", C.c_int32),
                ("sr", C.c_double), ("fmin", C.c_double), ("bins_per_octave", C.c_double), ("no_trough_prob", C.c_double),
                ("log_tiny", C.c_double), ("log_p_init", C.c_double), ("tiny", C.c_double)]


def _triang(m):
    """scipy.signal.windows.triang(m) (symmetric)."""
    n = np.arange(1, (m + 1) // 2 + 1)
    if m % 2 == 0:
        w = (2 * n - 1.0) / m
        return np.r_[w, w[::-1]]
    w = 2 * n / (m + 1.0)
    return np.r_[w, w[-2::-1]]


def pyin_plan(sr, fmin=60.0, fmax=2000.0, frame_length=2048, hop_length=256, n_thresholds=100, boltzmann_parameter=2.0,
              resolution=0.1, max_transition_rate=35.92, switch_prob=0.01, no_trough_prob=0.01):
    """-> (PyinPlan, tables float64, freqs): everything ``librosa.pyin`` derives from its arguments before touching audio."""
    win_length = frame_length // 2
    min_period = int(np.floor(sr / fmax))
    max_period = min(int(np.ceil(sr / fmin)), frame_length - win_length - 1)
    n_bins_per_semitone = int(np.ceil(1.0 / resolution))
    n_pitch_bins = int(np.floor(12 * n_bins_per_semitone * np.log2(fmax / fmin))) + 1
    max_semitones = round(max_transition_rate * 12 * hop_length / sr)
    width = max_semitones * n_bins_per_semitone + 1
    half = width // 2
    thresholds = np.linspace(0, 1, n_thresholds + 1)
    x = thresholds
    beta_cdf = 1.0 - (1.0 - x) ** 19 - 19.0 * x * (1.0 - x) ** 18          # scipy.stats.beta.cdf(x, 2, 18) in closed form
    beta_probs = np.diff(beta_cdf)
    bprefix = np.array([np.sum(beta_probs[:n]) for n in range(n_thresholds + 1)])
    lam = boltzmann_parameter
    nn = np.arange(MAX_TROUGHS + 1, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        bfac = (1 - np.exp(-lam)) / (1 - np.exp(-lam * nn))
    bfac[0] = 0.0
    bexp = np.exp(-lam * nn)
    # librosa.sequence.transition_local(n_pitch_bins, width, window="triangle", wrap=False), built the way librosa builds
    # it (pad, roll, clip, divide every row by its own numpy sum: the row sums differ by an ulp from row to row), then
    # the voicing switch of np.kron(transition_loop(2, 1 - switch_prob), .) and log(p + tiny) as sequence.viterbi takes it
    tri = _triang(width)
    trans = np.zeros((n_pitch_bins, n_pitch_bins), dtype=np.float64)
    lpad = (n_pitch_bins - width) // 2
    for i in range(n_pitch_bins):
        r = np.zeros(n_pitch_bins)
        r[lpad:lpad + width] = tri
        r = np.roll(r, n_pitch_bins // 2 + i + 1)
        r[min(n_pitch_bins, i + width // 2 + 1):] = 0
        r[:max(0, i - width // 2)] = 0
        trans[i] = r
    trans /= trans.sum(axis=1, keepdims=True)
    def log_table(p):                                            # [e + half][k] = log(p * trans[k][k + e] + tiny)
        out = np.full((width, n_pitch_bins), np.log(_TINY))
        scaled = np.log(p * trans + _TINY)
        for e in range(-half, half + 1):
            ks = np.arange(max(0, -e), min(n_pitch_bins, n_pitch_bins - e))
            out[e + half, ks] = scaled[ks, ks + e]
        return out
    lt_same = log_table(1 - switch_prob)
    lt_sw = log_table(switch_prob)
    head = np.concatenate([thresholds[1:], beta_probs, bprefix, bfac, bexp])
    if len(head) % 2:                                            # the (same, switch) pairs are read as 16-byte values
        head = np.concatenate([head, [0.0]])
    tables = np.concatenate([head, np.stack([lt_same, lt_sw], axis=-1).reshape(-1)]).astype(np.float64)
    plan = PyinPlan(frame_length, hop_length, min_period, max_period, n_pitch_bins, width, n_thresholds, 0, float(sr), float(fmin),
                    float(12 * n_bins_per_semitone), float(no_trough_prob), float(np.log(_TINY)),
                    float(np.log(1.0 / (2 * n_pitch_bins) + _TINY)), float(_TINY))
    freqs = fmin * 2 ** (np.arange(n_pitch_bins) / (12 * n_bins_per_semitone))
    return plan, tables, freqs


def pyin_batch(engine, fmin=60.0, fmax=2000.0, hop_length=256, fill_na=np.nan, **kw):
    """``librosa.pyin`` of every clip of the engine's resident batch -> [(f0, voiced_flag, voiced_prob), ...]."""
    plan, tables, freqs = pyin_plan(engine.rate, fmin, fmax, hop_length=hop_length, **kw)
    engine.pyin_run(plan, tables)
    out = []
    for i in range(len(engine.clip_lengths)):
        states, vp, _ = engine.pyin_fetch(i)
        voiced = states < plan.n_pitch_bins
        f0 = freqs[states % plan.n_pitch_bins]
        if fill_na is not None:
            f0 = np.where(voiced, f0, fill_na)
        out.append((f0, voiced, vp))
    return out


def compute_pitch(engine, clip=0, fmin=60.0, fmax=2000.0, hop_length=256):
    """Code/visualisation/acoustic_analysis.py:76-94: -> (time_f0, f0) of one resident clip."""
    f0, _, _ = pyin_batch(engine, fmin, fmax, hop_length)[clip]
    return np.arange(len(f0)) * hop_length / engine.rate, f0


def compute_spectrogram(engine, clip=0, n_fft=1024, hop_length=256):
    """Code/visualisation/acoustic_analysis.py:98-113: amplitude_to_db(|stft|, ref=np.max) of one resident clip."""
    engine.stft_db_run(n_fft, hop_length)
    return engine.stft_db_fetch(clip)
