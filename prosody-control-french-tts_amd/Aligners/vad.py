"""Energy VAD of the alignment step: ``"vad": "auditok"`` (Code/Aligners/use_whisper_timestamped.py:152).

whisper-timestamped (``get_vad_segments(method="auditok")``) hands the 16 kHz waveform to ``auditok.split`` and
dilates / merges the detected regions.  Both packages are third party and absent from /root/reference; this module
restates their published behaviour (auditok's energy validator and stream tokenizer, whisper-timestamped's
post-processing) -- **parity unpinned**: checked by hand-computable cases only (tests/test_vad.py).

Array work is on the GPU: the per-window sums of squares come from ``k_frame_energy``
(``ProsodyEngine.frame_energy_run``) as exact integers; what happens here is the scalar finishing
(mean, sqrt, 20 log10, threshold) and the tokenizer, a sequential state machine over ~20 windows per second.
"""
from __future__ import annotations

import math

import numpy as np

ANALYSIS_WINDOW = 0.05          # auditok's default analysis window (seconds)
ENERGY_THRESHOLD = 50.0         # whisper-timestamped: energy_threshold=50
_EPS = 1e-10                    # floor of the root mean square before the logarithm


def energy_db(sum_sq, count):
    """20 log10(sqrt(mean(x^2))) per window from the exact integer sums (float64, numpy semantics)."""
    ss = np.asarray(sum_sq, dtype=np.float64)
    n = np.asarray(count, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        rms = np.sqrt(ss / n)
    rms = np.clip(rms, _EPS, None)
    return 20.0 * np.log10(rms)


def _nb_windows(duration, window, rounder, eps=0.0):
    """auditok's duration -> number of analysis windows (``_duration_to_nb_windows``)."""
    if duration < 0 or window <= 0:
        raise ValueError("negative duration or non-positive analysis window")
    if duration == 0:
        return 0
    return int(rounder(duration / window + eps))


_SILENCE, _POSSIBLE_SILENCE, _POSSIBLE_NOISE, _NOISE = range(4)


def tokenize(valid, min_length, max_length, max_continuous_silence, *, init_min=0, init_max_silence=0,
             drop_trailing_silence=False, strict_min_length=False):
    """auditok's stream tokenizer over a sequence of window verdicts -> [(first_window, last_window), ...].

    A token opens on a valid window, tolerates runs of up to ``max_continuous_silence`` invalid windows, is cut at
    ``max_length`` windows (the next token is then contiguous and exempt from ``min_length`` unless
    ``strict_min_length``) and is dropped when shorter than ``min_length``."""
    if min_length <= 0 or max_length <= 0 or min_length > max_length:
        raise ValueError("need 0 < min_length <= max_length")
    if max_continuous_silence >= max_length:
        raise ValueError("max_continuous_silence must be < max_length")
    if init_min >= max_length:
        raise ValueError("init_min must be < max_length")
    tokens = []
    state = _SILENCE
    length = 0            # windows held by the open token
    silence = 0           # trailing invalid windows inside it
    start = 0
    init_count = 0
    contiguous = False
    cur = -1

    def close(truncated):
        nonlocal length, silence, start, contiguous
        n = length
        if not truncated and drop_trailing_silence and silence > 0:
            n -= silence
        if n >= min_length or (n > 0 and not strict_min_length and contiguous):
            tokens.append((start, start + n - 1))
            if truncated:
                start = cur + 1
                contiguous = True
            else:
                contiguous = False
        else:
            contiguous = False
        length = 0

    for v in valid:
        cur += 1
        v = bool(v)
        if state == _SILENCE:
            if v:
                init_count = 1; silence = 0; start = cur; length = 1
                if init_count >= init_min:
                    state = _NOISE
                    if length >= max_length:
                        close(True)
                else:
                    state = _POSSIBLE_NOISE
        elif state == _POSSIBLE_NOISE:
            if v:
                silence = 0; init_count += 1; length += 1
                if init_count >= init_min:
                    state = _NOISE
                    if length >= max_length:
                        close(True)
            else:
                silence += 1
                if silence > init_max_silence or length + 1 >= max_length:
                    length = 0; state = _SILENCE
                else:
                    length += 1
        elif state == _NOISE:
            if v:
                length += 1
                if length >= max_length:
                    close(True)
            elif max_continuous_silence <= 0:
                state = _SILENCE
                close(False)
            else:
                silence = 1; length += 1; state = _POSSIBLE_SILENCE
                if length == max_length:
                    close(True)
        else:  # _POSSIBLE_SILENCE
            if v:
                length += 1; silence = 0; state = _NOISE
                if length >= max_length:
                    close(True)
            elif silence >= max_continuous_silence:
                state = _SILENCE
                if silence < length:
                    close(False)
                length = 0; silence = 0
            else:
                length += 1; silence += 1
                if length >= max_length:
                    close(True)
    if state in (_NOISE, _POSSIBLE_SILENCE) and length > 0 and length > silence:
        close(False)
    return tokens


def auditok_split(sum_sq, count, rate, *, min_dur=0.2, max_dur=5.0, max_silence=0.3, drop_trailing_silence=False,
                  strict_min_dur=False, analysis_window=ANALYSIS_WINDOW, energy_threshold=ENERGY_THRESHOLD):
    """``auditok.split`` on per-window energies -> [(start_s, end_s), ...] (the regions' ``meta.start`` / ``meta.end``)."""
    block = int(analysis_window * rate)
    window = block / rate                                   # auditok re-derives the window from the integer block size
    min_length = _nb_windows(min_dur, window, math.ceil)
    max_length = _nb_windows(max_dur, window, math.floor, 1e-10)
    max_cont = _nb_windows(max_silence, window, math.floor, 1e-10)
    if min_length > max_length:
        raise ValueError(f"'min_dur' ({min_dur} sec.) results in {min_length} analysis window(s), more than the "
                         f"{max_length} of 'max_dur' ({max_dur} sec.)")
    if max_cont >= max_length:                              # the aligner retries without VAD on this message (:166)
        raise ValueError(f"'max_silence' ({max_silence} sec.) results in {max_cont} analysis window(s), which must be "
                         f"fewer than the {max_length} of 'max_dur' ({max_dur} sec.)")
    count = np.asarray(count, dtype=np.int64)
    valid = energy_db(sum_sq, count) >= energy_threshold
    out = []
    starts = np.concatenate([[0], np.cumsum(count)])      # samples before window k
    for a, b in tokenize(valid, min_length, max_length, max_cont, drop_trailing_silence=drop_trailing_silence,
                         strict_min_length=strict_min_dur):
        start = a * window
        duration = float(starts[b + 1] - starts[a]) / rate
        out.append((start, start + duration))
    return out


def vad_segments_from_energy(sum_sq, count, n_samples, rate=16000, *, min_speech_duration=0.1, min_silence_duration=0.1,
                             dilatation=0.5, output_sample=False):
    """whisper-timestamped ``get_vad_segments(method="auditok")`` from the window sums of one clip ->
    [{"start": s, "end": e}, ...] in seconds (or samples)."""
    dur = n_samples / rate
    regs = auditok_split(sum_sq, count, rate, min_dur=min_speech_duration, max_dur=dur,
                         max_silence=min(dur * 0.95, min_silence_duration), drop_trailing_silence=True)
    segs = [{"start": s * rate, "end": e * rate} for s, e in regs]
    if dilatation > 0:
        d = round(dilatation * rate)
        merged = []
        for sg in segs:
            ns = {"start": max(0, sg["start"] - d), "end": min(n_samples, sg["end"] + d)}
            if merged and merged[-1]["end"] >= ns["start"]:
                merged[-1]["end"] = ns["end"]
            else:
                merged.append(ns)
        segs = merged
    if output_sample:
        return [{"start": round(sg["start"]), "end": round(sg["end"])} for sg in segs]
    return [{"start": sg["start"] / rate, "end": sg["end"] / rate} for sg in segs]


def get_vad_segments(engine, *, rate=16000, **kw):
    """VAD of every clip of the engine's resident 16 kHz batch: one ``k_frame_energy`` launch (50 ms windows, the
    float32 round trip whisper-timestamped applies before auditok), then the host tokenizer per clip."""
    block = int(ANALYSIS_WINDOW * rate)
    engine.frame_energy_run(block, block, requantize=True)
    lens = engine.clip_lengths
    out = []
    for i, n in enumerate(lens):
        ss, cnt = engine.frame_energy_fetch(i)
        out.append(vad_segments_from_energy(ss, cnt, int(n), rate, **kw) if n > 0 else [])
    return out
