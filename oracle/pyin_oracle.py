"""oracle/pyin_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of ``librosa.pyin(y, sr=sr, fmin=60, fmax=2000, hop_length=256)`` as the reference's viewers call it
(Code/visualisation/app.py:74-78, acoustic_analysis.py:76-94, visualisation_abtest/app.py:108-111).
librosa (0.11.0 in tts-env.yml) is absent from /root/reference and from this image: the algorithm below follows the
published implementation (``librosa/core/pitch.py``: ``_cumulative_mean_normalized_difference``,
``_parabolic_interpolation``, ``__pyin_helper``; ``librosa/sequence.py``: ``transition_local``, ``transition_loop``,
``viterbi``) step by step -- **parity unpinned**, checked by analytic cases (tests/test_pyin.py).

Only tests/ may import this module."""
from __future__ import annotations

import numpy as np

TINY64 = np.finfo(np.float64).tiny


def _tiny(x):
    return np.finfo(x.dtype).tiny if np.issubdtype(x.dtype, np.floating) else np.finfo(np.float32).tiny


def frames_of(y, frame_length, hop_length):
    """center=True, pad_mode="constant": (frame_length, n_frames) view of the zero-padded signal."""
    y = np.asarray(y)
    yp = np.pad(y, (frame_length // 2, frame_length // 2), mode="constant")
    n_frames = 1 + (len(yp) - frame_length) // hop_length
    idx = np.arange(frame_length)[:, None] + hop_length * np.arange(n_frames)[None, :]
    return yp[idx]


def cmnd(y_frames, frame_length, win_length, min_period, max_period, exact=False):
    """Cumulative mean normalised difference function, rows min_period..max_period.
    exact=False: librosa's arithmetic (FFT cross-correlation and cumulative-sum energies in the input precision).
    exact=True: the same terms (cross-correlation, window energies, the 1e-6 clamps) evaluated in float64, exact for
    int16-valued samples -- what the GPU kernel computes; the two differ by the float32 rounding noise of the FFT
    and cumulative-sum route."""
    if exact:
        # the same terms without rounding noise: cross-correlation and window energies in float64 (exact for
        # int16-valued samples), librosa's |v| < 1e-6 -> 0 clamps, then the same combination and normalisation
        yf = y_frames.astype(np.float64)
        base = yf[1:win_length + 1, :]
        acf = np.zeros((max_period + 1, yf.shape[1])); energy = np.zeros_like(acf)
        for tau in range(max_period + 1):
            seg = yf[1 + tau:win_length + 1 + tau, :]
            acf[tau] = np.sum(base * seg, axis=0)
            energy[tau] = np.sum(seg * seg, axis=0)
        acf[np.abs(acf) < 1e-6] = 0
        energy[np.abs(energy) < 1e-6] = 0
        yin = energy[:1, :] + energy - 2 * acf
        num = yin[min_period:max_period + 1, :]
        tau = np.arange(1, max_period + 1)[:, None]
        den = (np.cumsum(yin[1:max_period + 1, :], axis=0) / tau)[min_period - 1:max_period, :]
        return num / (den + TINY64)
    a = np.fft.rfft(y_frames, frame_length, axis=-2)
    b = np.fft.rfft(y_frames[win_length:0:-1, :], frame_length, axis=-2)
    acf = np.fft.irfft(a * b, frame_length, axis=-2)[win_length:, :]
    acf[np.abs(acf) < 1e-6] = 0
    energy = np.cumsum(y_frames ** 2, axis=-2)
    energy = energy[win_length:, :] - energy[:-win_length, :]
    energy[np.abs(energy) < 1e-6] = 0
    yin = energy[:1, :] + energy - 2 * acf
    num = yin[min_period:max_period + 1, :]
    tau = np.arange(1, max_period + 1)[:, None]
    cummean = np.cumsum(yin[1:max_period + 1, :], axis=-2) / tau
    den = cummean[min_period - 1:max_period, :]
    return num / (den + _tiny(den))


def parabolic_shifts(x):
    """librosa ``_parabolic_interpolation`` along axis 0."""
    shifts = np.zeros_like(x)
    a = x[2:] + x[:-2] - 2 * x[1:-1]
    b = (x[2:] - x[:-2]) / 2
    with np.errstate(divide="ignore", invalid="ignore"):
        s = -b / a
    s[np.abs(b) >= np.abs(a)] = 0
    shifts[1:-1] = s
    return shifts


def localmin(x):
    """librosa.util.localmin along axis 0 of a 1-D array (edge padding)."""
    xp = np.pad(x, 1, mode="edge")
    return (x < xp[:-2]) & (x <= xp[2:])


def beta_cdf_2_18(x):
    """scipy.stats.beta.cdf(x, 2, 18) in closed form: 1 - (1 - x)^19 - 19 x (1 - x)^18."""
    x = np.asarray(x, dtype=np.float64)
    return 1.0 - (1.0 - x) ** 19 - 19.0 * x * (1.0 - x) ** 18


def boltzmann_pmf(k, lam, n):
    """scipy.stats.boltzmann.pmf(k, lam, n) for 0 <= k < n (0 elsewhere)."""
    k = np.asarray(k, dtype=np.float64); n = np.asarray(n, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        p = (1 - np.exp(-lam)) / (1 - np.exp(-lam * n)) * np.exp(-lam * k)
    return np.where((k >= 0) & (k < n), p, 0.0)


def transition_local_triangle(n_states, width):
    """librosa.sequence.transition_local(n_states, width, window="triangle", wrap=False)."""
    m = width
    nn = np.arange(1, (m + 1) // 2 + 1)
    w = (2 * nn) / (m + 1.0) if m % 2 else (2 * nn - 1.0) / m          # scipy.signal.windows.triang
    tri = np.r_[w, w[-2::-1]] if m % 2 else np.r_[w, w[::-1]]
    T = np.zeros((n_states, n_states))
    for i in range(n_states):
        row = np.zeros(n_states)
        lpad = (n_states - m) // 2
        row[lpad:lpad + m] = tri
        row = np.roll(row, n_states // 2 + i + 1)
        row[min(n_states, i + m // 2 + 1):] = 0
        row[:max(0, i - m // 2)] = 0
        T[i] = row
    return T / T.sum(axis=1, keepdims=True)


def viterbi(prob, transition, p_init):
    """librosa.sequence.viterbi: prob (n_states, n_steps) -> states (n_steps)."""
    n_states, n_steps = prob.shape
    log_trans = np.log(transition + TINY64)
    log_prob = np.log(prob.T + TINY64)
    value = np.zeros((n_steps, n_states)); ptr = np.zeros((n_steps, n_states), dtype=np.int64)
    value[0] = log_prob[0] + np.log(p_init + TINY64)
    lt = log_trans.T.copy()
    for t in range(1, n_steps):
        trans_out = value[t - 1][None, :] + lt                         # [j][k]
        ptr[t] = np.argmax(trans_out, axis=1)
        value[t] = log_prob[t] + trans_out[np.arange(n_states), ptr[t]]
    states = np.zeros(n_steps, dtype=np.int64)
    states[-1] = np.argmax(value[-1])
    for t in range(n_steps - 2, -1, -1):
        states[t] = ptr[t + 1, states[t + 1]]
    return states


def pyin(y, sr, fmin=60.0, fmax=2000.0, frame_length=2048, hop_length=256, n_thresholds=100, boltzmann_parameter=2.0,
         resolution=0.1, max_transition_rate=35.92, switch_prob=0.01, no_trough_prob=0.01, want_intermediate=False, exact=False):
    """-> (f0 with nan for unvoiced, voiced_flag, voiced_prob)."""
    y = np.asarray(y, dtype=np.float32)
    win_length = frame_length // 2
    y_frames = frames_of(y, frame_length, hop_length)
    min_period = int(np.floor(sr / fmax))
    max_period = min(int(np.ceil(sr / fmin)), frame_length - win_length - 1)
    yin = cmnd(y_frames, frame_length, win_length, min_period, max_period, exact=exact)
    shifts = parabolic_shifts(yin)
    thresholds = np.linspace(0, 1, n_thresholds + 1)
    beta_probs = np.diff(beta_cdf_2_18(thresholds))
    n_bins_per_semitone = int(np.ceil(1.0 / resolution))
    n_pitch_bins = int(np.floor(12 * n_bins_per_semitone * np.log2(fmax / fmin))) + 1
    n_frames = yin.shape[1]
    yin_probs = np.zeros_like(yin)
    for i in range(n_frames):
        fr = yin[:, i]
        is_trough = localmin(fr)
        is_trough[0] = fr[0] < fr[1]
        (idx,) = np.nonzero(is_trough)
        if len(idx) == 0:
            continue
        heights = fr[idx]
        below = np.less.outer(heights, thresholds[1:])
        positions = np.cumsum(below, axis=0) - 1
        n_troughs = np.count_nonzero(below, axis=0)
        prior = boltzmann_pmf(positions, boltzmann_parameter, n_troughs)
        prior[~below] = 0
        probs = prior.dot(beta_probs)
        gmin = np.argmin(heights)
        n_below_min = np.count_nonzero(~below[gmin, :])
        probs[gmin] += no_trough_prob * np.sum(beta_probs[:n_below_min])
        yin_probs[idx, i] = probs
    yin_period, frame_index = np.nonzero(yin_probs)
    period = min_period + yin_period
    period = period + shifts[yin_period, frame_index]
    f0_cand = sr / period
    bin_index = 12 * n_bins_per_semitone * np.log2(f0_cand / fmin)
    bin_index = np.clip(np.round(bin_index), 0, n_pitch_bins).astype(int)
    obs = np.zeros((2 * n_pitch_bins, n_frames))
    obs[bin_index, frame_index] = yin_probs[yin_period, frame_index]
    voiced_prob = np.clip(np.sum(obs[:n_pitch_bins, :], axis=0, keepdims=True), 0, 1)
    obs[n_pitch_bins:, :] = (1 - voiced_prob) / n_pitch_bins
    max_semitones = round(max_transition_rate * 12 * hop_length / sr)
    width = max_semitones * n_bins_per_semitone + 1
    trans = transition_local_triangle(n_pitch_bins, width)
    t_switch = np.array([[1 - switch_prob, switch_prob], [switch_prob, 1 - switch_prob]])
    transition = np.kron(t_switch, trans)
    p_init = np.ones(2 * n_pitch_bins) / (2 * n_pitch_bins)
    states = viterbi(obs, transition, p_init)
    freqs = fmin * 2 ** (np.arange(n_pitch_bins) / (12 * n_bins_per_semitone))
    f0 = freqs[states % n_pitch_bins]
    voiced = states < n_pitch_bins
    f0 = np.where(voiced, f0, np.nan)
    if want_intermediate:
        return f0, voiced, voiced_prob[0], dict(yin=yin, shifts=shifts, obs=obs, states=states, width=width, n_pitch_bins=n_pitch_bins,
                                                min_period=min_period, max_period=max_period, trans=trans)
    return f0, voiced, voiced_prob[0]
