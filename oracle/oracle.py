"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference hot path (hi-paris/Prosody-Control-French-TTS,
``Code/audioPipeline.py`` + ``Code/Pipeline`` + ``Code/Aligners``).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.  The product (``prosody-control-french-tts_amd``) never
does and has no CPU fallback.

Parity status per function (see DESIGN.md "Oracle"):

* pinned by reference-owned code run in the build container (goldens under
  ``tests/golden``): ``rms_db_int16_wrapped`` (R3), ``gate_check`` (R7),
  ``rate_metrics`` (R6 legacy), ``needleman_wunsch`` (legacy NW),
  ``prosody_adjustments`` (R6 modern).
* **parity unpinned** (third-party arithmetic absent from /root/reference and
  not installable here; restated from the published algorithm, checked by
  analytic known-answer tests): ``pitch_ac`` (Praat, praat-parselmouth==0.4.5),
  ``lufs`` (pyloudnorm, unpinned in tts-env.yml), ``pydub_slice`` (pydub==0.25.1),
  ``stft_db`` (librosa==0.11.0), ``log_mel`` (openai-whisper==20240930),
  ``frame_energy`` / ``frame_energy_db`` (auditok==0.3.0 energy validator behind
  whisper-timestamped==1.15.8 ``get_vad_segments(method="auditok")``).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
import wave

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    """Compile oracle/pce_oracle.c -> oracle/libpce_oracle.so (gcc)."""
    so = os.path.join(_HERE, "libpce_oracle.so")
    src = os.path.join(_HERE, "pce_oracle.c")
    if (not os.path.exists(so)) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "libpce_oracle.so"], stdout=subprocess.DEVNULL)
    return so


class PitchParams(C.Structure):
    _fields_ = [
        ("time_step", C.c_double), ("pitch_floor", C.c_double), ("periods_per_window", C.c_double),
        ("max_candidates", C.c_int32), ("reserved", C.c_int32),
        ("silence_threshold", C.c_double), ("voicing_threshold", C.c_double), ("octave_cost", C.c_double),
        ("octave_jump_cost", C.c_double), ("voiced_unvoiced_cost", C.c_double), ("pitch_ceiling", C.c_double),
    ]


class PitchPlan(C.Structure):
    _fields_ = [
        ("dt", C.c_double), ("t1", C.c_double), ("ceiling", C.c_double), ("dt_window", C.c_double),
        ("n_frames", C.c_long), ("nsamp_period", C.c_long), ("halfnsamp_period", C.c_long),
        ("nsamp_window", C.c_long), ("halfnsamp_window", C.c_long), ("maximum_lag", C.c_long),
        ("nsamp_fft", C.c_long), ("brent_ixmax", C.c_long), ("max_candidates", C.c_long),
    ]


def praat_params(pitch_floor=75.0, pitch_ceiling=600.0, time_step=0.0) -> PitchParams:
    """Parameter set of parselmouth ``Sound.to_pitch(time_step, pitch_floor, pitch_ceiling)``
    = Praat ``Sound_to_Pitch`` -> ``Sound_to_Pitch_ac(dt, floor, 3.0, 15, false, 0.03, 0.45,
    0.01, 0.35, 0.14, ceiling)``."""
    return PitchParams(time_step or 0.0, pitch_floor, 3.0, 15, 0, 0.03, 0.45, 0.01, 0.35, 0.14, pitch_ceiling)


def _lib():
    global _LIB
    if _LIB is None:
        lib = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        lib.por_pitch_plan_make.argtypes = [C.c_long, C.c_double, C.c_double, C.POINTER(PitchParams), C.POINTER(PitchPlan)]
        lib.por_pitch_plan_make.restype = C.c_int
        lib.por_pitch_ac.argtypes = [dp, C.c_long, C.c_double, C.c_double, C.POINTER(PitchParams), dp, dp, dp, dp, dp,
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
        lib.por_pitch_ac.restype = C.c_int
        lib.por_window_autocorr.argtypes = [C.c_long, C.c_double, C.c_double, C.POINTER(PitchParams), dp, C.c_long]
        lib.por_window_autocorr.restype = None
        lib.por_lufs.argtypes = [dp, C.c_long, C.c_double, dp, dp]
        lib.por_lufs.restype = C.c_int
        lib.por_lufs_num_blocks.argtypes = [C.c_long, C.c_double]
        lib.por_lufs_num_blocks.restype = C.c_long
        lib.por_kweight_coeffs.argtypes = [C.c_double, dp, dp, dp, dp]
        lib.por_kweight_coeffs.restype = None
        _LIB = lib
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class PraatError(RuntimeError):
    """What parselmouth raises (``parselmouth.PraatError``) for too-short sounds."""


# --------------------------------------------------------------------------
# decode + slicing (pydub==0.25.1 / Praat semantics)
# --------------------------------------------------------------------------
def read_wav(path):
    """(rate, int16 ndarray [n_frames] or [n_frames, ch]) of a PCM16 RIFF/WAVE file."""
    with wave.open(str(path), "rb") as w:
        if w.getsampwidth() != 2:
            raise ValueError("only 16-bit PCM is supported")
        rate, ch, n = w.getframerate(), w.getnchannels(), w.getnframes()
        data = np.frombuffer(w.readframes(n), dtype="<i2")
    if ch > 1:
        data = data.reshape(-1, ch)
    return rate, data


def pydub_len_ms(n_frames: int, rate: int) -> int:
    """``len(AudioSegment)`` = round(1000 * frame_count / frame_rate) (pydub/audio_segment.py)."""
    return round(1000 * (float(n_frames) / rate))


def pydub_slice(n_frames: int, rate: int, start_ms, stop_ms):
    """``AudioSegment[start_ms:stop_ms]`` -> (begin_frame, n_real, n_pad) of the result.

    Follows pydub 0.25.1 ``__getitem__``/``_parse_position``: ms clamp to len(),
    frame = int(ms * (rate/1000.0)), missing frames (<= 2 ms worth) padded with
    silence, only if at least one real frame exists.  Used by the reference at
    Code/audioPipeline.py:321,342 and Code/Pipeline/compute_loudness_adjustments.py:14.
    """
    L = pydub_len_ms(n_frames, rate)
    start = start_ms if start_ms is not None else 0
    end = stop_ms if stop_ms is not None else L
    start = min(start, L)
    end = min(end, L)

    def parse(val):
        if val < 0:
            val = L - abs(val)
        return int(val * (rate / 1000.0))

    s, e = parse(start), parse(end)
    # python bytes slicing semantics on the frame axis
    s_c = min(max(s, 0), n_frames) if s >= 0 else max(n_frames + s, 0)
    e_c = min(max(e, 0), n_frames) if e >= 0 else max(n_frames + e, 0)
    n_real = max(e_c - s_c, 0)
    expected = e - s
    missing = expected - n_real
    n_pad = 0
    if missing:
        if missing > 2 * (rate / 1000.0):
            raise ValueError("TooManyMissingFrames")
        if n_real > 0:
            n_pad = missing
    return s_c, n_real, max(n_pad, 0)


def pydub_samples(pcm: np.ndarray, rate: int, start_ms=None, stop_ms=None) -> np.ndarray:
    """int16 samples of ``AudioSegment[start_ms:stop_ms].get_array_of_samples()`` (mono)."""
    if start_ms is None and stop_ms is None:
        return pcm
    b, n, pad = pydub_slice(len(pcm), rate, start_ms, stop_ms)
    out = pcm[b:b + n]
    if pad:
        out = np.concatenate([out, np.zeros(pad, dtype=pcm.dtype)])
    return out


def praat_extract_part(n: int, rate: float, t0: float, t1: float, preserve_times=True):
    """Index math of Praat ``Sound_extractPart`` (rectangular window, relativeWidth 1):
    returns (ix1, ix2, x1_new) with 1-based inclusive virtual sample indices; samples
    outside 1..n are zero.  Sound from a file: xmin=0, dx=1/rate, x1=dx/2."""
    dx = 1.0 / rate
    x1 = 0.5 * dx
    xmin, xmax = 0.0, n * dx
    if t0 == t1:
        t0, t1 = xmin, xmax
    ix1 = 1 + math.ceil((t0 - x1) / dx)
    ix2 = 1 + math.floor((t1 - x1) / dx)
    if ix2 < ix1:
        raise PraatError("Extracted Sound would contain no samples.")
    x1_new = x1 + (ix1 - 1) * dx
    if not preserve_times:
        x1_new -= t0
    return ix1, ix2, x1_new


def praat_part_samples(pcm: np.ndarray, rate, t0, t1, preserve_times=True):
    """float64 samples (int16/32768) of ``Sound(path).extract_part(t0, t1, preserve_times)`` and its x1."""
    ix1, ix2, x1n = praat_extract_part(len(pcm), rate, t0, t1, preserve_times)
    out = np.zeros(ix2 - ix1 + 1, dtype=np.float64)
    lo, hi = max(ix1, 1), min(ix2, len(pcm))
    if hi >= lo:
        out[lo - ix1: hi - ix1 + 1] = pcm[lo - 1: hi].astype(np.float64) / 32768.0
    return out, x1n


# --------------------------------------------------------------------------
# R1 / R2: Praat autocorrelation pitch
# --------------------------------------------------------------------------
def pitch_plan(nx: int, dx: float, x1: float, params: PitchParams) -> PitchPlan:
    pl = PitchPlan()
    st = _lib().por_pitch_plan_make(nx, dx, x1, C.byref(params), C.byref(pl))
    if st != 0:
        raise PraatError(f"Sound too short for pitch analysis (status {st})")
    return pl


def pitch_ac(z: np.ndarray, dx: float, x1: float, params: PitchParams, want_candidates=False):
    """Praat Sound_to_Pitch_ac + Pitch_pathFinder on float64 samples ``z``.

    Returns dict(f0, strength, intensity, plan[, cand_f, cand_s, ncand, evals]).
    Raises PraatError where Praat throws (sound shorter than 3 periods of the floor)."""
    z = np.ascontiguousarray(z, dtype=np.float64)
    pl = pitch_plan(len(z), dx, x1, params)
    nF, mc = pl.n_frames, pl.max_candidates
    f0 = np.zeros(nF); st_ = np.zeros(nF); it = np.zeros(nF)
    cf = np.zeros(nF * mc); cs = np.zeros(nF * mc); nc = np.zeros(nF, dtype=np.int32)
    stats = np.zeros(2, dtype=np.int64)
    st = _lib().por_pitch_ac(_dp(z), len(z), dx, x1, C.byref(params), _dp(f0), _dp(st_), _dp(it), _dp(cf), _dp(cs),
                             nc.ctypes.data_as(C.POINTER(C.c_int32)), stats.ctypes.data_as(C.POINTER(C.c_int64)))
    if st != 0:
        raise PraatError(f"pitch analysis failed (status {st})")
    out = dict(f0=f0, strength=st_, intensity=it, plan=pl, evals=int(stats[0]), n_cands=int(stats[1]))
    if want_candidates:
        out.update(cand_f=cf.reshape(nF, mc), cand_s=cs.reshape(nF, mc), ncand=nc)
    return out


def window_autocorr(nx, dx, x1, params, n_out):
    out = np.zeros(n_out)
    _lib().por_window_autocorr(nx, dx, x1, C.byref(params), _dp(out), n_out)
    return out


def median_pitch(pcm: np.ndarray, rate, t0=0.0, t1=None, floor=150.0, ceiling=600.0) -> float:
    """``get_median_pitch`` closure, Code/audioPipeline.py:326-335."""
    if t1 is None:
        z, x1 = pcm.astype(np.float64) / 32768.0, 0.5 / rate
    else:
        z, x1 = praat_part_samples(pcm, rate, t0, t1, preserve_times=True)
    f = pitch_ac(z, 1.0 / rate, x1, praat_params(floor, ceiling))["f0"]
    v = f[f > 0]
    return float(np.median(v)) if v.size > 0 else 0.0


def legacy_pitch_segment(pcm: np.ndarray, rate, start, end) -> float:
    """``calculate_pitch_segment``, Code/Pipeline/compute_pitch_adjustments.py:167-208
    (file checks excluded): floors 75,100,150,200, geometric mean of voiced frames."""
    import statistics
    n = len(pcm)
    total = n / rate                                   # snd.get_total_duration() = xmax - xmin
    if start < 0 or end > total or start >= end:
        return 0
    try:
        z, x1 = praat_part_samples(pcm, rate, start, end, preserve_times=False)
    except PraatError:
        return 0
    for fl in (75, 100, 150, 200):
        try:
            f = pitch_ac(z, 1.0 / rate, x1, praat_params(float(fl), 600.0))["f0"]
        except PraatError:
            continue
        v = f[f > 0]
        if len(v) > 0:
            return statistics.geometric_mean(v)
    return 0


# --------------------------------------------------------------------------
# R4: integrated loudness (pyloudnorm semantics)
# --------------------------------------------------------------------------
def kweight_coeffs(rate):
    b1 = np.zeros(3); a1 = np.zeros(3); b2 = np.zeros(3); a2 = np.zeros(3)
    _lib().por_kweight_coeffs(float(rate), _dp(b1), _dp(a1), _dp(b2), _dp(a2))
    return (b1, a1), (b2, a2)


def lufs_c(samples: np.ndarray, rate) -> float:
    """C restatement; raises ValueError where pyloudnorm does."""
    x = np.ascontiguousarray(samples, dtype=np.float64)
    out = C.c_double()
    st = _lib().por_lufs(_dp(x), len(x), float(rate), C.byref(out), None)
    if st != 0:
        raise ValueError("Audio must have length greater than the block size.")
    return out.value


def lufs_numpy(samples: np.ndarray, rate) -> float:
    """numpy/scipy restatement of ``peak-normalise -> pyln.Meter(rate).integrated_loudness``
    (Code/audioPipeline.py:349-352) using scipy.signal.lfilter, as pyloudnorm itself does."""
    import scipy.signal
    data = np.asarray(samples, dtype=float)
    peak = np.abs(data).max() or 1.0
    data = data / peak
    if data.shape[0] < 0.400 * rate:
        raise ValueError("Audio must have length greater than the block size.")
    G, Q, fc = 4.0, 1 / np.sqrt(2), 1500.0
    A = 10 ** (G / 40.0); w0 = 2.0 * np.pi * (fc / rate); alpha = np.sin(w0) / (2.0 * Q)
    b0 = A * ((A + 1) + (A - 1) * np.cos(w0) + 2 * np.sqrt(A) * alpha)
    b1 = -2 * A * ((A - 1) + (A + 1) * np.cos(w0))
    b2 = A * ((A + 1) + (A - 1) * np.cos(w0) - 2 * np.sqrt(A) * alpha)
    a0 = (A + 1) - (A - 1) * np.cos(w0) + 2 * np.sqrt(A) * alpha
    a1 = 2 * ((A - 1) - (A + 1) * np.cos(w0))
    a2 = (A + 1) - (A - 1) * np.cos(w0) - 2 * np.sqrt(A) * alpha
    data = scipy.signal.lfilter(np.array([b0, b1, b2]) / a0, np.array([a0, a1, a2]) / a0, data)
    Q, fc = 0.5, 38.0
    w0 = 2.0 * np.pi * (fc / rate); alpha = np.sin(w0) / (2.0 * Q)
    b0 = (1 + np.cos(w0)) / 2; b1 = -(1 + np.cos(w0)); b2 = (1 + np.cos(w0)) / 2
    a0 = 1 + alpha; a1 = -2 * np.cos(w0); a2 = 1 - alpha
    data = scipy.signal.lfilter(np.array([b0, b1, b2]) / a0, np.array([a0, a1, a2]) / a0, data)

    T_g, Gamma_a, step = 0.400, -70.0, 1.0 - 0.75
    n = data.shape[0]
    T = n / rate
    nb = int(np.round(((T - T_g) / (T_g * step))) + 1)
    z = np.zeros(max(nb, 0))
    for j in range(nb):
        l = int(T_g * (j * step) * rate); u = int(T_g * (j * step + 1) * rate)
        z[j] = (1.0 / (T_g * rate)) * np.sum(np.square(data[l:u]))
    with np.errstate(divide="ignore", invalid="ignore"):
        l_ = [-0.691 + 10.0 * np.log10(zj) for zj in z]
        J = [j for j, lj in enumerate(l_) if lj >= Gamma_a]
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            zavg = np.mean([z[j] for j in J])
            Gamma_r = -0.691 + 10.0 * np.log10(zavg) - 10.0
            J = [j for j, lj in enumerate(l_) if (lj > Gamma_r and lj > Gamma_a)]
            zavg = np.nan_to_num(np.mean([z[j] for j in J]))
        return float(-0.691 + 10.0 * np.log10(zavg))


def get_lufs(pcm: np.ndarray, rate, t0=0.0, t1=None, impl=lufs_c, meter_rate=None) -> float:
    """``get_lufs`` closure, Code/audioPipeline.py:338-358.  ``rate``: the file's own frame rate (pydub's slicing);
    ``meter_rate``: the rate of the ``pyln.Meter`` handed in (:372,:493 build it from the NATURAL recording, also for the raw
    synthesis whose rate differs) -- pyloudnorm never sees the data's rate, only the meter's; default: the file's."""
    meter_rate = rate if meter_rate is None else meter_rate
    if t1 is not None:
        s = pydub_samples(pcm, rate, int(t0 * 1000), int(t1 * 1000))
    else:
        s = pcm
    if s.size == 0:
        s = pcm
    try:
        return impl(s.astype(float), meter_rate)
    except ValueError:
        return impl(pcm.astype(float), meter_rate)


def part_duration(n_frames, rate, t0=0.0, t1=None) -> float:
    """``get_part_duration`` / ``get_duration`` closures, Code/audioPipeline.py:314-323, 360-361."""
    if t1 is not None:
        _, n, pad = pydub_slice(n_frames, rate, int(t0 * 1000), int(t1 * 1000))
        return ((n + pad) / rate if rate else 0.0) or 1e-4
    return (n_frames / rate if rate else 0.0) or 1e-4


# --------------------------------------------------------------------------
# R3 / R7: reference-owned energy arithmetic (pinned by goldens G3 / G4)
# --------------------------------------------------------------------------
def rms_db_int16_wrapped(samples_i16: np.ndarray) -> float:
    """``_calculate_loudness`` arithmetic, Code/Pipeline/compute_loudness_adjustments.py:17-24.
    ``np.array(array('h')) ** 2`` stays int16 and wraps mod 2**16."""
    s = np.asarray(samples_i16, dtype=np.int16)
    with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
        S = s ** 2
        rms = np.sqrt(np.abs(np.mean(S)))
        return float(20 * np.log10(rms))


def gate_check(data_i16: np.ndarray):
    """``_check_audio_content`` arithmetic, Code/Aligners/use_whisper_timestamped.py:204-210:
    (rms float32, silence_ratio float64, is_ok)."""
    data = np.asarray(data_i16)
    if data.ndim > 1:
        data = data[:, 0]
    rms = np.sqrt(np.mean(np.square(data.astype(np.float32))))
    non_silence = np.sum(np.abs(data) > 500)
    ratio = 1.0 - (non_silence / len(data))
    ok = not (ratio > 0.95 or rms < 100)
    return rms, ratio, ok


# --------------------------------------------------------------------------
# R10: librosa STFT-dB (librosa==0.11.0 defaults), float32
# --------------------------------------------------------------------------
def stft_db(y_f32: np.ndarray, n_fft=1024, hop=256, amin=1e-5, top_db=80.0) -> np.ndarray:
    """``librosa.amplitude_to_db(np.abs(librosa.stft(y, n_fft, hop)), ref=np.max)``
    (Code/visualisation/app.py:69-72): periodic Hann, center=True zero padding, complex64."""
    y = np.asarray(y_f32, dtype=np.float32)
    w = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)).astype(np.float32)   # get_window('hann', fftbins=True)
    yp = np.pad(y, n_fft // 2, mode="constant")
    nfr = 1 + (len(yp) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(nfr)[:, None]
    S = np.fft.rfft(yp[idx] * w[None, :], axis=1).astype(np.complex64).T        # [1+n_fft/2, nfr]
    mag = np.abs(S).astype(np.float32)
    ref = np.max(mag) if mag.size else np.float32(0)
    power = np.square(mag, dtype=np.float32)
    log_spec = 10.0 * np.log10(np.maximum(np.float32(amin * amin), power))
    log_spec -= 10.0 * np.log10(np.maximum(np.float32(amin * amin), np.float32(ref) ** 2))
    return np.maximum(log_spec, log_spec.max() - top_db).astype(np.float32)


# --------------------------------------------------------------------------
# sample-rate conversion spec (ffmpeg's resampler is not reproducible here)
# --------------------------------------------------------------------------
def resample_int16(pcm_i16: np.ndarray, rate_in: int, rate_out: int) -> np.ndarray:
    """The engine's resampling spec stated with scipy: resample_poly (Kaiser 5.0 low-pass, zero
    padding), round half to even, saturate to int16."""
    import scipy.signal
    g = math.gcd(rate_in, rate_out)
    y = scipy.signal.resample_poly(np.asarray(pcm_i16, dtype=np.float64), rate_out // g, rate_in // g)
    return np.clip(np.rint(y), -32768, 32767).astype(np.int16)


# ---------------------------------------------------------------- energy VAD front end (3P, parity unpinned)
def frame_energy(pcm_i16: np.ndarray, window: int, hop: int = None, requantize: bool = False):
    """Per analysis window: (exact integer sum of squares, samples in the window).

    Follows the call chain behind ``"vad": "auditok"`` (Code/Aligners/use_whisper_timestamped.py:152):
    whisper.load_audio leaves int16 / 32768 as float32; whisper-timestamped's ``get_vad_segments`` feeds auditok
    ``(audio * 32767).astype(np.int16)`` (``requantize``); auditok reads blocks of ``int(0.05 * rate)`` samples,
    the last one short.  Frame k = [k*hop, min(k*hop + window, n)), k < ceil(n / hop)."""
    x = np.asarray(pcm_i16, dtype=np.int16)
    if requantize:
        a = x.astype(np.float32) / np.float32(32768.0)
        x = (a * 32767).astype(np.int16)
    hop = window if hop is None else hop
    n = len(x)
    nf = -(-n // hop)
    ss = np.zeros(nf, dtype=np.int64); cnt = np.zeros(nf, dtype=np.int32)
    for k in range(nf):
        seg = x[k * hop:k * hop + window].astype(np.int64)
        ss[k] = int(np.sum(seg * seg)); cnt[k] = len(seg)
    return ss, cnt


def frame_energy_db(pcm_i16: np.ndarray, window: int, requantize: bool = False) -> np.ndarray:
    """auditok's window energy: 20 log10(max(sqrt(mean(x^2)), 1e-10)) on float64 samples."""
    x = np.asarray(pcm_i16, dtype=np.int16)
    if requantize:
        x = ((x.astype(np.float32) / np.float32(32768.0)) * 32767).astype(np.int16)
    out = []
    for k in range(-(-len(x) // window)):
        seg = x[k * window:(k + 1) * window].astype(np.float64)
        out.append(20.0 * np.log10(max(np.sqrt(np.mean(seg ** 2)), 1e-10)))
    return np.array(out)



# ----------------------------------------------------------------------------------------------------------------------------
# Levenshtein distance (legacy aligner, Code/Aligners/levenshtein_dist_align_txtgrids.py:43-70).  PINNED: golden G9
# (tests/golden/levenshtein.json) holds the reference function's own outputs.
# ----------------------------------------------------------------------------------------------------------------------------
def levenshtein(s1: str, s2: str) -> int:
    """The reference's two-row recurrence over Python characters: swap so that s2 is the shorter (:54-55), len(s1) when s2 is empty
    (:57-58), then per character of s1 a new row ``min(previous[j + 1] + 1, current[j] + 1, previous[j] + (c1 != c2))`` (:62-68).
    The sequential ``current[j] + 1`` term is resolved with a running minimum (numpy), which is the same integer arithmetic."""
    if len(s1) < len(s2):
        s1, s2 = s2, s1
    if len(s2) == 0:
        return len(s1)
    b = np.frombuffer(s2.encode("utf-32-le", "surrogatepass"), dtype=np.uint32)
    prev = np.arange(len(s2) + 1, dtype=np.int64)
    idx = np.arange(len(s2) + 1, dtype=np.int64)
    for i, c1 in enumerate(np.frombuffer(s1.encode("utf-32-le", "surrogatepass"), dtype=np.uint32)):
        cand = np.empty(len(s2) + 1, dtype=np.int64)
        cand[0] = i + 1
        cand[1:] = np.minimum(prev[1:] + 1, prev[:-1] + (b != c1))
        # current[j] = min(cand[j], current[j - 1] + 1)  ==  min over k <= j of cand[k] + (j - k)
        prev = np.minimum.accumulate(cand - idx) + idx
    return int(prev[-1])
