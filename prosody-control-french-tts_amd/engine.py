"""ctypes binding of ``libpce.so`` (C ABI: ``include/pce.h``).

:class:`ProsodyEngine` owns one ``pce_ctx`` (one process, one GPU).  A batch of clips is
uploaded once and stays resident in HBM; every measurement the reference takes by
re-decoding the file (Code/audioPipeline.py:314-361) becomes a slice of that batch.
There is no CPU path: without the built library or without a GPU, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpce.so")


class PceError(RuntimeError):
    pass


def native_library_path() -> str:
    """The library this process binds: ``libpce.so`` beside this file, or the file ``PCE_LIBRARY`` names (laboratory builds are
    selected this way -- ``tools/ab_*.sh`` -- instead of being copied over the product's library)."""
    return os.environ.get("PCE_LIBRARY") or _LIB_PATH


def build_native(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into ``libpce.so`` (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", csrc, "-j4"], stdout=subprocess.DEVNULL)
    if not os.path.exists(_LIB_PATH):
        raise PceError("build did not produce libpce.so")
    return _LIB_PATH


class Slice(C.Structure):
    _fields_ = [("clip", C.c_int32), ("flags", C.c_int32), ("begin", C.c_int64), ("end", C.c_int64), ("x1", C.c_double)]


class Energy(C.Structure):
    _fields_ = [("n", C.c_int64), ("sum_sq", C.c_int64), ("sum_sq_wrap16", C.c_int64), ("n_loud", C.c_int64),
                ("peak_abs", C.c_int32), ("reserved", C.c_int32)]


class PitchParams(C.Structure):
    _fields_ = [("time_step", C.c_double), ("pitch_floor", C.c_double), ("periods_per_window", C.c_double),
                ("max_candidates", C.c_int32), ("reserved", C.c_int32), ("silence_threshold", C.c_double),
                ("voicing_threshold", C.c_double), ("octave_cost", C.c_double), ("octave_jump_cost", C.c_double),
                ("voiced_unvoiced_cost", C.c_double), ("pitch_ceiling", C.c_double)]

    @classmethod
    def praat(cls, pitch_floor=75.0, pitch_ceiling=600.0, time_step=0.0):
        """parselmouth ``Sound.to_pitch(time_step, pitch_floor, pitch_ceiling)`` defaults."""
        return cls(time_step or 0.0, float(pitch_floor), 3.0, 15, 0, 0.03, 0.45, 0.01, 0.35, 0.14, float(pitch_ceiling))


class PitchSummary(C.Structure):
    _fields_ = [("n_frames", C.c_int64), ("n_voiced", C.c_int64), ("median_f0", C.c_double), ("mean_log_f0", C.c_double),
                ("t1", C.c_double), ("status", C.c_int32), ("reserved", C.c_int32)]


class WhisperDims(C.Structure):
    _fields_ = [("n_mels", C.c_int32), ("n_ctx", C.c_int32), ("n_state", C.c_int32), ("n_head", C.c_int32), ("n_layer", C.c_int32)]


class WhisperTextDims(C.Structure):
    _fields_ = [("n_vocab", C.c_int32), ("n_text_ctx", C.c_int32), ("n_state", C.c_int32), ("n_head", C.c_int32), ("n_layer", C.c_int32)]


class WhisperDecodeRules(C.Structure):
    _fields_ = [("eot", C.c_int32), ("timestamp_begin", C.c_int32), ("max_initial_timestamp_index", C.c_int32), ("reserved", C.c_int32)]


class WhisperDecodeOpts(C.Structure):
    _fields_ = [("sample_begin", C.c_void_p), ("sample_begin_all", C.c_int32), ("temperature", C.c_float), ("seed_lo", C.c_uint32),
                ("seed_hi", C.c_uint32), ("probe_token", C.c_int32), ("flags", C.c_int32)]


class BertDims(C.Structure):
    _fields_ = [("n_vocab", C.c_int32), ("n_pos", C.c_int32), ("n_type", C.c_int32), ("n_state", C.c_int32), ("n_head", C.c_int32),
                ("n_layer", C.c_int32), ("n_labels", C.c_int32)]


SLICE_DTYPE = np.dtype([("clip", "<i4"), ("flags", "<i4"), ("begin", "<i8"), ("end", "<i8"), ("x1", "<f8")])
ENERGY_DTYPE = np.dtype([("n", "<i8"), ("sum_sq", "<i8"), ("sum_sq_wrap16", "<i8"), ("n_loud", "<i8"),
                         ("peak_abs", "<i4"), ("reserved", "<i4")])
SUMMARY_DTYPE = np.dtype([("n_frames", "<i8"), ("n_voiced", "<i8"), ("median_f0", "<f8"), ("mean_log_f0", "<f8"),
                          ("t1", "<f8"), ("status", "<i4"), ("reserved", "<i4")])
assert SLICE_DTYPE.itemsize == C.sizeof(Slice) and ENERGY_DTYPE.itemsize == C.sizeof(Energy)
assert SUMMARY_DTYPE.itemsize == C.sizeof(PitchSummary)

SLICE_OK, SLICE_TOO_SHORT, SLICE_EMPTY = 0, 1, 2

KERNEL_IDS = ["k_energy", "k_lufs_pass1", "k_lufs_scan", "k_lufs_pass2", "k_lufs_gate",
              "k_pitch_refine", "k_pitch_frames", "k_pitch_path", "k_pitch_median", "k_pitch_delta", "k_stft_max", "k_stft_db", "k_logmel_frames", "whisper_encoder", "k_resample", "k_dtw", "whisper_align", "k_nw", "k_stft_norm", "k_frame_energy", "bert_forward", "k_pyin_frames", "k_pyin_viterbi", "whisper_decode_step",
              "k_gemm_bf16", "k_gemm_wide", "k_attention", "k_layernorm", "k_gemm_flat",
              "k_add_layernorm", "k_stft_raw", "k_logmel_norm", "k_attention_lean",
              "k_gemm_flat:qkv", "k_gemm_flat:out", "k_gemm_flat:fc1", "k_gemm_flat:fc2", "k_gemm_flat:xkv",
              "whisper_decode_loop", "k_cross_attn1", "k_gemm_skinny", "k_levenshtein"]     # = pce_kernel_name(id) for every id (tests/test_abi_and_shard.py)

# every symbol include/pce.h declares
EXPORTS = ["pce_create", "pce_destroy", "pce_last_error", "pce_sync", "pce_api_version", "pce_api_minor", "pce_device_info",
           "pce_upload_pcm_s16", "pce_bind_pcm_s16_device", "pce_num_clips",
           "pce_energy_run", "pce_energy_fetch", "pce_lufs_set_meter_rate", "pce_lufs_run", "pce_lufs_fetch",
           "pce_frame_energy_run", "pce_frame_energy_shape", "pce_frame_energy_fetch", "pce_pyin_run", "pce_pyin_shape", "pce_pyin_fetch",
           "pce_pitch_plan", "pce_pitch_run", "pce_pitch_set_refine", "pce_pitch_fetch",
           "pce_stft_db_run", "pce_stft_db_shape", "pce_stft_db_fetch", "pce_stft_db_device",
           "pce_resample_run", "pce_download_pcm_s16",
           "pce_dtw", "pce_nw_align", "pce_levenshtein", "pce_whisper_decoder_load", "pce_whisper_align_run", "pce_whisper_align_shape", "pce_whisper_align_fetch", "pce_whisper_align_paths_enqueue", "pce_whisper_align_paths_wait", "pce_whisper_sample_keys", "pce_whisper_decode_step", "pce_whisper_decode_step_ex", "pce_whisper_decode_loop", "pce_whisper_set_operands", "pce_whisper_get_operands", "pce_selftest_xattn",
           "pce_logmel_run", "pce_logmel_run_at", "pce_logmel_fetch", "pce_whisper_load", "pce_whisper_encode_run", "pce_selftest_gemm", "pce_selftest_attention", "pce_whisper_encode_fetch",
           "pce_stats_enqueue", "pce_stats_wait", "pce_bert_load", "pce_bert_run", "pce_bert_fetch",
           "pce_profile_enable", "pce_profile_reset", "pce_profile_get", "pce_profile_get_work", "pce_kernel_name"]


def load_library() -> C.CDLL:
    path = native_library_path()
    if not os.path.exists(path):
        raise PceError(f"{path} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                       "this engine has no CPU fallback")
    lib = C.CDLL(path)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.pce_create.argtypes = [C.c_int, vp, C.c_char_p, C.c_size_t]; lib.pce_create.restype = vp
    lib.pce_destroy.argtypes = [vp]; lib.pce_destroy.restype = None
    lib.pce_last_error.argtypes = [vp]; lib.pce_last_error.restype = C.c_char_p
    lib.pce_sync.argtypes = [vp]
    lib.pce_api_version.argtypes = []
    lib.pce_api_minor.argtypes = []
    lib.pce_device_info.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(i32), C.POINTER(i64)]
    lib.pce_upload_pcm_s16.argtypes = [vp, vp, vp, i32, i32]
    lib.pce_bind_pcm_s16_device.argtypes = [vp, vp, vp, i32, i32]
    lib.pce_num_clips.argtypes = [vp]
    lib.pce_energy_run.argtypes = [vp, vp, i32, i32]
    lib.pce_energy_fetch.argtypes = [vp, vp]
    lib.pce_lufs_set_meter_rate.argtypes = [vp, i32]
    lib.pce_pitch_set_refine.argtypes = [vp, i32]
    lib.pce_lufs_run.argtypes = [vp, vp, i32]
    lib.pce_frame_energy_run.argtypes = [vp, i32, i32, i32]
    lib.pce_pyin_run.argtypes = [vp, vp, vp, i64]
    lib.pce_pyin_shape.argtypes = [vp, i32, C.POINTER(i64)]
    lib.pce_pyin_fetch.argtypes = [vp, i32, vp, vp, C.POINTER(i32)]
    lib.pce_frame_energy_shape.argtypes = [vp, i32, C.POINTER(i64)]
    lib.pce_frame_energy_fetch.argtypes = [vp, i32, vp, vp]
    lib.pce_lufs_fetch.argtypes = [vp, vp, vp]
    lib.pce_pitch_plan.argtypes = [vp, C.POINTER(PitchParams), vp, i32, vp, vp]
    lib.pce_pitch_run.argtypes = [vp, C.POINTER(PitchParams), vp, i32]
    lib.pce_pitch_fetch.argtypes = [vp, vp, vp, vp]
    lib.pce_stats_enqueue.argtypes = [vp, i32]
    lib.pce_nw_align.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]
    lib.pce_levenshtein.argtypes = [vp, vp, vp, vp, vp, i32, vp]
    lib.pce_stats_wait.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.pce_stft_db_run.argtypes = [vp, i32, i32]
    lib.pce_stft_db_shape.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32)]
    lib.pce_stft_db_fetch.argtypes = [vp, i32, vp]
    lib.pce_stft_db_device.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    lib.pce_resample_run.argtypes = [vp, i32, i32, vp, i32, i64]
    lib.pce_download_pcm_s16.argtypes = [vp, vp, vp, C.POINTER(i32)]
    lib.pce_dtw.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    lib.pce_whisper_decoder_load.argtypes = [vp, C.POINTER(WhisperTextDims), vp, i64]
    lib.pce_whisper_align_run.argtypes = [vp, vp, vp, vp, i32, vp, i32, C.c_float]
    lib.pce_whisper_align_shape.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32)]
    lib.pce_whisper_decode_step.argtypes = [vp, vp, vp, i32, C.POINTER(WhisperDecodeRules), vp, vp, vp]
    lib.pce_whisper_decode_step_ex.argtypes = [vp, vp, vp, C.POINTER(WhisperDecodeRules), vp, C.POINTER(WhisperDecodeOpts), vp, vp, vp]
    lib.pce_whisper_set_operands.argtypes = [vp, i32]
    lib.pce_whisper_get_operands.argtypes = [vp]
    lib.pce_whisper_decode_loop.argtypes = [vp, vp, vp, C.POINTER(WhisperDecodeRules), vp, C.POINTER(WhisperDecodeOpts), i32, i32, vp, vp, vp, vp]
    lib.pce_whisper_align_fetch.argtypes = [vp, i32, vp, vp, C.POINTER(i32), vp]
    lib.pce_whisper_align_paths_enqueue.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32)]
    lib.pce_whisper_align_paths_wait.argtypes = [vp, i32, vp, vp, vp]
    lib.pce_whisper_sample_keys.argtypes = [vp, vp, i32]
    lib.pce_bert_load.argtypes = [vp, C.POINTER(BertDims), vp, i64]
    lib.pce_bert_run.argtypes = [vp, vp, vp, i32]
    lib.pce_bert_fetch.argtypes = [vp, i32, vp, vp]
    lib.pce_logmel_run.argtypes = [vp, i32]
    lib.pce_logmel_run_at.argtypes = [vp, i32, vp]
    lib.pce_logmel_fetch.argtypes = [vp, i32, vp]
    lib.pce_whisper_load.argtypes = [vp, C.POINTER(WhisperDims), vp, i64]
    lib.pce_whisper_encode_run.argtypes = [vp]
    lib.pce_selftest_gemm.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.pce_selftest_attention.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.pce_selftest_xattn.argtypes = [vp] * 11 + [i32] * 5 + [vp]
    lib.pce_whisper_encode_fetch.argtypes = [vp, i32, vp]
    lib.pce_profile_enable.argtypes = [vp, C.c_int]
    lib.pce_profile_reset.argtypes = [vp]
    lib.pce_profile_get.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(i64)]
    lib.pce_profile_get_work.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    lib.pce_kernel_name.argtypes = [C.c_int]; lib.pce_kernel_name.restype = C.c_char_p
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int and name not in ("pce_create",):
            fn.restype = C.c_int
    return lib


def make_slices(clips, begins, ends, x1=None) -> np.ndarray:
    """Pack parallel arrays into the ``pce_slice`` layout."""
    n = len(clips)
    s = np.zeros(n, dtype=SLICE_DTYPE)
    s["clip"] = clips; s["begin"] = begins; s["end"] = ends
    if x1 is not None:
        s["x1"] = x1
    return s


class ProsodyEngine:
    """One GPU context.  ``device``: HIP device index; ``stream``: optional ``hipStream_t`` handle
    (e.g. ``torch.cuda.current_stream().cuda_stream``) to enqueue on."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._lib = load_library()
        err = C.create_string_buffer(512)
        self._ctx = self._lib.pce_create(int(device), C.c_void_p(stream) if stream else None, err, len(err))
        if not self._ctx:
            raise PceError(f"pce_create failed: {err.value.decode(errors='replace')}")
        self.device = int(device)
        self.rate = 0
        self.clip_lengths = np.zeros(0, dtype=np.int64)
        self._keep = []

    # ---------------------------------------------------------------- plumbing
    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.pce_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int):
        if rc != 0:
            raise PceError(f"libpce status {rc}: {self._lib.pce_last_error(self._ctx).decode(errors='replace')}")

    def sync(self):
        self._check(self._lib.pce_sync(self._ctx))

    def device_info(self):
        name = C.create_string_buffer(256); cus = C.c_int32(); hbm = C.c_int64()
        self._check(self._lib.pce_device_info(self._ctx, name, len(name), C.byref(cus), C.byref(hbm)))
        return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value}

    # ---------------------------------------------------------------- residency
    def upload(self, clips, rate: int):
        """Upload a batch: ``clips`` is a list of int16 1-D arrays (one per utterance)."""
        clips = [np.ascontiguousarray(c, dtype=np.int16).reshape(-1) for c in clips]
        lens = np.array([len(c) for c in clips], dtype=np.int64)
        offsets = np.zeros(len(clips) + 1, dtype=np.int64)
        np.cumsum(lens, out=offsets[1:])
        pcm = np.concatenate(clips) if clips else np.zeros(0, dtype=np.int16)
        if pcm.size == 0:
            pcm = np.zeros(1, dtype=np.int16)
        self._check(self._lib.pce_upload_pcm_s16(self._ctx, pcm.ctypes.data, offsets.ctypes.data, len(clips), int(rate)))
        self.rate = int(rate); self.clip_lengths = lens; self.offsets = offsets
        return self

    def bind_device(self, device_ptr: int, offsets, rate: int, keepalive=None):
        """Zero-copy: adopt int16 PCM already resident in HBM (e.g. a torch tensor's ``data_ptr()``)."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self._check(self._lib.pce_bind_pcm_s16_device(self._ctx, C.c_void_p(device_ptr), offsets.ctypes.data,
                                                       len(offsets) - 1, int(rate)))
        self.rate = int(rate); self.clip_lengths = np.diff(offsets); self.offsets = offsets
        self._keep = [keepalive]
        return self

    def whole_clip_slices(self) -> np.ndarray:
        n = len(self.clip_lengths)
        return make_slices(np.arange(n), np.zeros(n, dtype=np.int64), self.clip_lengths, np.full(n, 0.5 / self.rate))

    # ---------------------------------------------------------------- ops
    @staticmethod
    def _slices(s) -> np.ndarray:
        s = np.ascontiguousarray(s, dtype=SLICE_DTYPE)
        return s

    def energy_run(self, slices, loud_threshold: int = 500):
        s = self._slices(slices); self._en_n = len(s)
        self._check(self._lib.pce_energy_run(self._ctx, s.ctypes.data, len(s), int(loud_threshold)))

    def energy_fetch(self) -> np.ndarray:
        out = np.zeros(self._en_n, dtype=ENERGY_DTYPE)
        self._check(self._lib.pce_energy_fetch(self._ctx, out.ctypes.data))
        return out

    def energy(self, slices, loud_threshold: int = 500) -> np.ndarray:
        self.energy_run(slices, loud_threshold)
        return self.energy_fetch()

    def lufs_set_meter_rate(self, rate: int = 0):
        """``pyln.Meter(rate)`` of the following ``lufs`` calls (0: the batch's own rate)."""
        self._check(self._lib.pce_lufs_set_meter_rate(self._ctx, int(rate)))

    def pitch_set_refine(self, mode: str = "seeded"):
        """How candidate maxima are refined (``pce_pitch_set_refine``): ``"seeded"`` (default: parabolic search seeded with the samples
        around the peak, Praat's own iterates only where the two could differ) or ``"praat"`` (NUMminimize_brent's iterates for every
        candidate).  Additive config key of ``AudioPipeline``: ``pitch_refine``."""
        self._check(self._lib.pce_pitch_set_refine(self._ctx, {"seeded": 0, "praat": 1}[str(mode).lower()]))

    def lufs_run(self, slices):
        s = self._slices(slices); self._lu_n = len(s)
        self._check(self._lib.pce_lufs_run(self._ctx, s.ctypes.data, len(s)))

    def lufs_fetch(self):
        out = np.zeros(self._lu_n, dtype=np.float64); st = np.zeros(self._lu_n, dtype=np.int32)
        self._check(self._lib.pce_lufs_fetch(self._ctx, out.ctypes.data, st.ctypes.data))
        return out, st

    def lufs(self, slices):
        self.lufs_run(slices)
        return self.lufs_fetch()

    def pitch_plan(self, slices, params: PitchParams):
        s = self._slices(slices)
        off = np.zeros(len(s) + 1, dtype=np.int64); st = np.zeros(len(s), dtype=np.int32)
        self._check(self._lib.pce_pitch_plan(self._ctx, C.byref(params), s.ctypes.data, len(s), off.ctypes.data, st.ctypes.data))
        return off, st

    def pitch_run(self, slices, params: PitchParams):
        s = self._slices(slices); self._pi_slices = s; self._pi_params = params
        self._check(self._lib.pce_pitch_run(self._ctx, C.byref(params), s.ctypes.data, len(s)))

    def pitch_fetch(self, want_f0=True, want_strength=False):
        s = self._pi_slices
        off, _ = self.pitch_plan(s, self._pi_params)
        total = int(off[-1])
        f0 = np.zeros(total, dtype=np.float64) if want_f0 else None
        sg = np.zeros(total, dtype=np.float64) if want_strength else None
        summ = np.zeros(len(s), dtype=SUMMARY_DTYPE)
        self._check(self._lib.pce_pitch_fetch(self._ctx, f0.ctypes.data if want_f0 else None,
                                              sg.ctypes.data if want_strength else None, summ.ctypes.data))
        return {"frame_offsets": off, "f0": f0, "strength": sg, "summary": summ}

    def pitch(self, slices, params: PitchParams, want_f0=True, want_strength=False):
        self.pitch_run(slices, params)
        return self.pitch_fetch(want_f0, want_strength)

    def stats_enqueue(self, slot: int = 0):
        """Queue the device-to-host copies of the last energy / LUFS / pitch-summary results behind their
        runs and return at once; ``stats_wait(slot)`` collects them.  Two slots: the next batch can be
        launched before this one's numbers are read."""
        self._stat_n = getattr(self, "_stat_n", {})
        self._stat_n[slot] = (getattr(self, "_en_n", None), getattr(self, "_lu_n", None),
                              len(self._pi_slices) if getattr(self, "_pi_slices", None) is not None else None)
        self._check(self._lib.pce_stats_enqueue(self._ctx, int(slot)))

    def stats_wait(self, slot: int = 0):
        """-> dict(energy=ENERGY_DTYPE[n] | None, lufs=(values, status) | None, pitch=SUMMARY_DTYPE[n] | None)."""
        en_n, lu_n, pi_n = self._stat_n[slot]
        en = np.zeros(en_n, dtype=ENERGY_DTYPE) if en_n is not None else None
        lu = np.zeros(lu_n, dtype=np.float64) if lu_n is not None else None
        st = np.zeros(lu_n, dtype=np.int32) if lu_n is not None else None
        pi = np.zeros(pi_n, dtype=SUMMARY_DTYPE) if pi_n is not None else None
        self._check(self._lib.pce_stats_wait(self._ctx, int(slot), en.ctypes.data if en is not None else None,
                                             lu.ctypes.data if lu is not None else None, st.ctypes.data if st is not None else None,
                                             pi.ctypes.data if pi is not None else None))
        return {"energy": en, "lufs": (lu, st) if lu is not None else None, "pitch": pi}

    def stft_db_run(self, n_fft: int = 1024, hop: int = 256):
        self._check(self._lib.pce_stft_db_run(self._ctx, int(n_fft), int(hop)))

    def stft_db_fetch(self, clip: int) -> np.ndarray:
        nb = C.c_int32(); nf = C.c_int32()
        self._check(self._lib.pce_stft_db_shape(self._ctx, int(clip), C.byref(nb), C.byref(nf)))
        out = np.zeros((nb.value, nf.value), dtype=np.float32)
        self._check(self._lib.pce_stft_db_fetch(self._ctx, int(clip), out.ctypes.data))
        return out

    def stft_db_device(self):
        p = C.c_void_p(); n = C.c_int64()
        self._check(self._lib.pce_stft_db_device(self._ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    # ---------------------------------------------------------------- sample-rate conversion
    def resample(self, target_rate: int):
        """Resample the resident batch to ``target_rate`` (polyphase, scipy.signal.resample_poly's filter design)."""
        from .hostrules import resample_filter
        if target_rate == self.rate:
            return self
        up, down, taps, n_pre_remove = resample_filter(self.rate, target_rate)
        taps = np.ascontiguousarray(taps, dtype=np.float64)
        self._check(self._lib.pce_resample_run(self._ctx, up, down, taps.ctypes.data, len(taps), n_pre_remove))
        off = np.zeros(len(self.clip_lengths) + 1, dtype=np.int64); rate = C.c_int32()
        self._check(self._lib.pce_download_pcm_s16(self._ctx, None, off.ctypes.data, C.byref(rate)))
        self.rate, self.offsets, self.clip_lengths = rate.value, off, np.diff(off)
        return self

    def download(self):
        """The resident batch as a list of int16 arrays."""
        pcm = np.zeros(max(int(self.offsets[-1]), 1), dtype=np.int16)
        self._check(self._lib.pce_download_pcm_s16(self._ctx, pcm.ctypes.data, None, None))
        return [pcm[self.offsets[i]:self.offsets[i + 1]].copy() for i in range(len(self.clip_lengths))]

    # ---------------------------------------------------------------- whisper front end
    def logmel_run(self, n_mels: int = 80):
        self._n_mels = int(n_mels)
        self._check(self._lib.pce_logmel_run(self._ctx, int(n_mels)))

    def logmel_run_at(self, n_mels: int, start_frames):
        """The 30 s window that starts ``start_frames[clip]`` frames (10 ms each) into every clip, as ``whisper.transcribe``
        slices the log-mel of the whole recording at its seek position."""
        sf = np.ascontiguousarray(start_frames, dtype=np.int64)
        self._check(self._lib.pce_logmel_run_at(self._ctx, int(n_mels), sf.ctypes.data))
        self._n_mels = int(n_mels)

    def logmel_fetch(self, clip: int) -> np.ndarray:
        out = np.zeros((self._n_mels, 3000), dtype=np.float32)
        self._check(self._lib.pce_logmel_fetch(self._ctx, int(clip), out.ctypes.data))
        return out

    # ---------------------------------------------------------------- operand type of the Whisper / BERT products
    def whisper_set_operands(self, kind: str):
        """``"fp16"`` (default: the reference's own arithmetic, openai-whisper's fp16=True) or ``"bf16"``.  The two builds keep
        separate state: select BEFORE loading weights / running the log-mel, and load again after switching."""
        code = {"bf16": 0, "fp16": 1, "fp16-resid16": 2}[str(kind).lower()]       # fp16-resid16: fp16 operands AND an fp16 residual stream in the batched encoder
        self._check(self._lib.pce_whisper_set_operands(self._ctx, code))

    @property
    def whisper_operands(self) -> str:
        code = self._lib.pce_whisper_get_operands(self._ctx)
        if code < 0:
            self._check(code)
        return {0: "bf16", 1: "fp16", 2: "fp16-resid16"}.get(code, f"mode-{code}")

    def _op_dtype(self):
        import torch
        return torch.bfloat16 if self.whisper_operands == "bf16" else torch.float16

    def whisper_load(self, dims: dict, weights: np.ndarray):
        """``dims``: n_mels, n_ctx, n_state, n_head, n_layer; ``weights``: float32 blob in the order of include/pce.h."""
        w = np.ascontiguousarray(weights, dtype=np.float32)
        self._wdims = WhisperDims(dims["n_mels"], dims["n_ctx"], dims["n_state"], dims["n_head"], dims["n_layer"])
        self._check(self._lib.pce_whisper_load(self._ctx, C.byref(self._wdims), w.ctypes.data, w.size))

    def whisper_encode_run(self):
        self._check(self._lib.pce_whisper_encode_run(self._ctx))
        self._n_encoded = len(self.clip_lengths)

    def whisper_num_encoded(self) -> int:
        return getattr(self, "_n_encoded", 0)

    def whisper_sample_keys(self, keys=None):
        """``pce_whisper_sample_keys``: the ids temperature sampling keys its noise by, one per clip of the encoded batch (None: back to
        batch positions).  They last until the next :meth:`whisper_encode_run`."""
        if keys is None or len(keys) == 0:
            self._check(self._lib.pce_whisper_sample_keys(self._ctx, None, 0)); return
        k = np.ascontiguousarray(np.asarray(keys, dtype=np.int64) & 0x7FFFFFFF, dtype=np.int32)
        self._check(self._lib.pce_whisper_sample_keys(self._ctx, k.ctypes.data, int(k.size)))

    def selftest_gemm(self, A, B, bias=None, epilogue: int = 0, rows_per_clip: int = 1, vt_sp: int = 0):
        """C = epilogue(A B^T + bias) on the persistent 256 x 256 GEMM kernel; A [M][K], B [N][K] float arrays (rounded to bf16 here) ->
        float32 result decoded from bf16 ([M][N], or [clips][N][vt_sp] for the transposed epilogue 2).  ``epilogue`` >= 256 (a multiple of 256, < N)
        is the split launch: returns (row-major [M][epilogue], transposed image [clips][N - epilogue][vt_sp])."""
        import torch
        a = torch.from_numpy(np.ascontiguousarray(A, dtype=np.float32)).to(self._op_dtype()).contiguous()
        b = torch.from_numpy(np.ascontiguousarray(B, dtype=np.float32)).to(self._op_dtype()).contiguous()
        M, K = a.shape; N = b.shape[0]
        if 16 <= epilogue <= 19:                                    # the tiled / few-row kernels behind launch_gemm (see pce_selftest_gemm)
            f32 = epilogue == 19
            outb = np.zeros(M * N, dtype=np.float32) if f32 else torch.zeros(M * N, dtype=self._op_dtype())
            bvv = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
            self._check(self._lib.pce_selftest_gemm(self._ctx, a.view(torch.int16).numpy().ctypes.data, b.view(torch.int16).numpy().ctypes.data,
                                                    bvv.ctypes.data if bvv is not None else None, M, N, K, int(epilogue), 1, 0,
                                                    outb.ctypes.data if f32 else outb.view(torch.int16).numpy().ctypes.data))
            return outb.reshape(M, N) if f32 else outb.float().numpy().reshape(M, N)
        split = epilogue if epilogue >= 256 else 0
        n_out = (M // rows_per_clip) * N * vt_sp if epilogue == 2 else M * split + (M // rows_per_clip) * (N - split) * vt_sp if split else M * N
        out = torch.zeros(n_out, dtype=self._op_dtype())
        bv = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
        self._check(self._lib.pce_selftest_gemm(self._ctx, a.view(torch.int16).numpy().ctypes.data, b.view(torch.int16).numpy().ctypes.data,
                                                bv.ctypes.data if bv is not None else None, M, N, K, int(epilogue), int(rows_per_clip), int(vt_sp),
                                                out.view(torch.int16).numpy().ctypes.data))
        res = out.float().numpy()
        if split:
            return res[:M * split].reshape(M, split), res[M * split:].reshape(M // rows_per_clip, N - split, vt_sp)
        return res.reshape(M // rows_per_clip, N, vt_sp) if epilogue == 2 else res.reshape(M, N)

    def selftest_attention(self, q, k, v, causal: bool = False, mode: int = 0):
        """softmax(q k^T / 8) v per (clip, head) on the attention kernel; q [clips][q_len][heads*64], k / v [clips][k_len][heads*64] float
        arrays (rounded to bf16 here).  mode 0: as the engine runs it, 1: exact path only.  Returns (out float32 decoded
        from bf16, number of workgroups that fell back to the exact path)."""
        import torch
        tq, tk, tv = (torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(self._op_dtype()).contiguous() for x in (q, k, v))
        clips, q_len, hd = tq.shape
        k_len = tk.shape[1]
        assert hd % 64 == 0 and tk.shape == tv.shape == (clips, k_len, hd)
        out = torch.zeros_like(tq)
        fb = C.c_int32(0)
        self._check(self._lib.pce_selftest_attention(self._ctx, tq.view(torch.int16).numpy().ctypes.data, tk.view(torch.int16).numpy().ctypes.data,
                                                     tv.view(torch.int16).numpy().ctypes.data, clips, hd // 64, q_len, k_len, int(bool(causal)), int(mode),
                                                     out.view(torch.int16).numpy().ctypes.data, C.addressof(fb)))
        return out.float().numpy(), int(fb.value)

    def selftest_xattn(self, resid, ln_w, ln_b, wq, bq, wk, wv, bv, E, k_len, heads: int, workgroups_per_clip: int = 0):
        """One layer of the encoder-output cross-attention of a decoding step (``pce_selftest_xattn``): resid [n][d], E [n][k_cap][d], weights [d][d]
        float arrays (weights and E rounded to the context's operand type here) -> (out [n][d] float32 decoded from the 16-bit result, and the
        ROUNDED wq / wk / wv / E as float32: what the kernels really multiplied, for an exact restatement)."""
        import torch
        dt = self._op_dtype()
        r16 = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dt).contiguous()
        tq, tk, tv, tE = r16(wq), r16(wk), r16(wv), r16(E)
        n, k_cap, d = tE.shape
        f32 = lambda x: np.ascontiguousarray(x, dtype=np.float32)
        res, lw, lb, vq, vv = f32(resid), f32(ln_w), f32(ln_b), f32(bq), f32(bv)
        kl = np.ascontiguousarray(k_len, dtype=np.int32)
        out = torch.zeros((n, d), dtype=dt)
        ptr = lambda t: t.view(torch.int16).numpy().ctypes.data
        self._check(self._lib.pce_selftest_xattn(self._ctx, res.ctypes.data, lw.ctypes.data, lb.ctypes.data, ptr(tq), vq.ctypes.data, ptr(tk), ptr(tv), vv.ctypes.data,
                                                 ptr(tE), kl.ctypes.data, int(n), int(k_cap), int(d), int(heads), int(workgroups_per_clip), ptr(out)))
        return out.float().numpy(), tq.float().numpy(), tk.float().numpy(), tv.float().numpy(), tE.float().numpy()

    def whisper_encode_fetch(self, clip: int) -> np.ndarray:
        out = np.zeros((1500, self._wdims.n_state), dtype=np.float32)
        self._check(self._lib.pce_whisper_encode_fetch(self._ctx, int(clip), out.ctypes.data))
        return out

    def whisper_decoder_load(self, dims: dict, weights: np.ndarray):
        w = np.ascontiguousarray(weights, dtype=np.float32)
        self._tdims = WhisperTextDims(dims["n_vocab"], dims["n_text_ctx"], dims["n_state"], dims["n_head"], dims["n_layer"])
        self._check(self._lib.pce_whisper_decoder_load(self._ctx, C.byref(self._tdims), w.ctypes.data, w.size))

    def whisper_align_run(self, token_lists, num_frames, sot_len: int, head_mask=None, medfilt_width: int = 7, qk_scale: float = 1.0):
        """Enqueue the forced alignment (decoder, cross-attention weights, DTW) without fetching anything."""
        toks = np.concatenate([np.asarray(t, dtype=np.int32) for t in token_lists]).astype(np.int32)
        off = np.zeros(len(token_lists) + 1, dtype=np.int32); np.cumsum([len(t) for t in token_lists], out=off[1:])
        nf = np.ascontiguousarray(num_frames, dtype=np.int32)
        hm = None if head_mask is None else np.ascontiguousarray(head_mask, dtype=np.uint8)
        self._check(self._lib.pce_whisper_align_run(self._ctx, toks.ctypes.data, off.ctypes.data, nf.ctypes.data, int(sot_len),
                                                    hm.ctypes.data if hm is not None else None, int(medfilt_width), float(qk_scale)))

    def whisper_align_paths_enqueue(self, slot: int = 0):
        """Queue the device-to-host copies of every clip's DTW path of the last ``whisper_align_run`` (pinned staging, returns at once)."""
        n = C.c_int32(); stride = C.c_int32()
        self._check(self._lib.pce_whisper_align_paths_enqueue(self._ctx, int(slot), C.byref(n), C.byref(stride)))
        self._al_slot = getattr(self, "_al_slot", {}); self._al_slot[int(slot)] = (n.value, stride.value)

    def whisper_align_paths_wait(self, slot: int = 0):
        """-> (path_len[n], text_idx[n][stride], time_idx[n][stride]); row i holds path_len[i] steps."""
        n, stride = self._al_slot[int(slot)]
        pl = np.zeros(n, dtype=np.int32); ti = np.zeros((n, stride), dtype=np.int32); tj = np.zeros((n, stride), dtype=np.int32)
        self._check(self._lib.pce_whisper_align_paths_wait(self._ctx, int(slot), pl.ctypes.data, ti.ctypes.data, tj.ctypes.data))
        return pl, ti, tj

    def whisper_align(self, token_lists, num_frames, sot_len: int, head_mask=None, medfilt_width: int = 7, qk_scale: float = 1.0,
                      want_cost: bool = False):
        """Forced alignment of the given token sequences (one per clip, specials included) against the encoded audio.
        Returns per clip a dict(text_indices, time_indices[, cost]) -- openai-whisper ``find_alignment`` up to the DTW."""
        toks = np.concatenate([np.asarray(t, dtype=np.int32) for t in token_lists]).astype(np.int32)
        off = np.zeros(len(token_lists) + 1, dtype=np.int32); np.cumsum([len(t) for t in token_lists], out=off[1:])
        nf = np.ascontiguousarray(num_frames, dtype=np.int32)
        hm = None if head_mask is None else np.ascontiguousarray(head_mask, dtype=np.uint8)
        self._check(self._lib.pce_whisper_align_run(self._ctx, toks.ctypes.data, off.ctypes.data, nf.ctypes.data, int(sot_len),
                                                    hm.ctypes.data if hm is not None else None, int(medfilt_width), float(qk_scale)))
        out = []
        for i in range(len(token_lists)):
            nr = C.c_int32(); nc = C.c_int32()
            self._check(self._lib.pce_whisper_align_shape(self._ctx, i, C.byref(nr), C.byref(nc)))
            ti = np.zeros(nr.value + nc.value, dtype=np.int32); tj = np.zeros(nr.value + nc.value, dtype=np.int32); pl = C.c_int32()
            cost = np.zeros((nr.value, nc.value), dtype=np.float64) if want_cost else None
            self._check(self._lib.pce_whisper_align_fetch(self._ctx, i, ti.ctypes.data, tj.ctypes.data, C.byref(pl),
                                                          cost.ctypes.data if want_cost else None))
            r = {"text_indices": ti[:pl.value].copy(), "time_indices": tj[:pl.value].copy()}
            if want_cost:
                r["cost"] = cost
            out.append(r)
        return out

    def dtw(self, cost: np.ndarray):
        """DTW paths of a batch of [n_rows, n_cols] fp64 cost matrices -> list of (text_indices, time_indices)."""
        x = np.ascontiguousarray(cost, dtype=np.float64)
        if x.ndim == 2:
            x = x[None]
        b, n, m = x.shape
        pi = np.zeros((b, n + m), dtype=np.int32); pj = np.zeros((b, n + m), dtype=np.int32); pl = np.zeros(b, dtype=np.int32)
        self._check(self._lib.pce_dtw(self._ctx, x.ctypes.data, n, m, b, pi.ctypes.data, pj.ctypes.data, pl.ctypes.data))
        return [(pi[k, :pl[k]].copy(), pj[k, :pl[k]].copy()) for k in range(b)]

    def frame_energy_run(self, window: int, hop: int = None, requantize: bool = False):
        """Exact integer energy of every analysis window of every clip (k_frame_energy): frame k covers
        [k*hop, min(k*hop + window, n)); hop defaults to window (auditok's blocks)."""
        self._check(self._lib.pce_frame_energy_run(self._ctx, int(window), int(window if hop is None else hop), 1 if requantize else 0))

    def frame_energy_fetch(self, clip: int):
        """-> (sum_sq int64[n_frames], count int32[n_frames]) of one clip."""
        nf = C.c_int64()
        self._check(self._lib.pce_frame_energy_shape(self._ctx, int(clip), C.byref(nf)))
        ss = np.zeros(nf.value, dtype=np.int64); cnt = np.zeros(nf.value, dtype=np.int32)
        self._check(self._lib.pce_frame_energy_fetch(self._ctx, int(clip), ss.ctypes.data, cnt.ctypes.data))
        return ss, cnt

    def whisper_decode_step(self, token_lists, sample_begin: int, eot: int, timestamp_begin: int, vocab_mask, max_initial_timestamp_index=None):
        """One step of free-running decoding for every encoded clip -> next token ids (int32 [clips]).  ``vocab_mask``:
        uint8 [n_vocab], bit 0 = always suppressed, bit 1 = suppressed at the first sampled position."""
        toks = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.int32) for t in token_lists]), dtype=np.int32)
        off = np.zeros(len(token_lists) + 1, dtype=np.int32); np.cumsum([len(t) for t in token_lists], out=off[1:])
        rules = WhisperDecodeRules(int(eot), int(timestamp_begin), -1 if max_initial_timestamp_index is None else int(max_initial_timestamp_index), 0)
        vm = np.ascontiguousarray(vocab_mask, dtype=np.uint8)
        nxt = np.zeros(len(token_lists), dtype=np.int32); lp = np.zeros(len(token_lists), dtype=np.float32)
        self._check(self._lib.pce_whisper_decode_step(self._ctx, toks.ctypes.data, off.ctypes.data, int(sample_begin), C.byref(rules),
                                                      vm.ctypes.data, nxt.ctypes.data, lp.ctypes.data))
        self.last_decode_logprobs = lp
        return nxt

    def whisper_decode_step_ex(self, token_lists, sample_begin, eot: int, timestamp_begin: int, vocab_mask, max_initial_timestamp_index=None,
                               temperature: float = 0.0, seed: int = 0, probe_token: int = -1, no_cache: bool = False):
        """``whisper_decode_step`` with a prompt length per sequence (``sample_begin``: int or one per clip), sampling at a
        temperature (one draw from softmax(filtered logits / temperature), reproducible for a given ``seed``) and an
        optional probe of the unfiltered distribution at one token (``probe_token``: no_speech_prob when the prefixes end at
        <|startoftranscript|>).  -> (next ids int32 [clips], log-probabilities float32 [clips], probe float32 [clips] | None)."""
        toks = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.int32) for t in token_lists]), dtype=np.int32)
        off = np.zeros(len(token_lists) + 1, dtype=np.int32); np.cumsum([len(t) for t in token_lists], out=off[1:])
        rules = WhisperDecodeRules(int(eot), int(timestamp_begin), -1 if max_initial_timestamp_index is None else int(max_initial_timestamp_index), 0)
        vm = np.ascontiguousarray(vocab_mask, dtype=np.uint8)
        n = len(token_lists)
        sb = None if np.isscalar(sample_begin) else np.ascontiguousarray(sample_begin, dtype=np.int32)
        if sb is not None and sb.shape != (n,):
            raise ValueError("sample_begin: one prompt length per sequence")
        opts = WhisperDecodeOpts(sb.ctypes.data if sb is not None else None, int(sample_begin) if sb is None else 0, float(temperature),
                                 int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF, int(probe_token), 1 if no_cache else 0)
        nxt = np.zeros(n, dtype=np.int32); lp = np.zeros(n, dtype=np.float32)
        pr = np.zeros(n, dtype=np.float32) if probe_token >= 0 else None
        self._check(self._lib.pce_whisper_decode_step_ex(self._ctx, toks.ctypes.data, off.ctypes.data, C.byref(rules), vm.ctypes.data, C.byref(opts),
                                                         nxt.ctypes.data, lp.ctypes.data, pr.ctypes.data if pr is not None else None))
        return nxt, lp, pr

    def whisper_decode_loop(self, token_lists, sample_begin, eot: int, timestamp_begin: int, vocab_mask, max_new: int,
                            max_initial_timestamp_index=None, temperature: float = 0.0, seed: int = 0, probe_token: int = -1,
                            no_cache: bool = False, check_every: int = 4):
        """The free-running loop on the device: prompts up once, at most ``max_new`` steps, results down once (the host only reads an
        "ended" counter every ``check_every`` steps).  -> (tokens int32 [clips][steps run], log-probabilities float32 [clips][steps run],
        probe float32 [clips] | None): what ``steps run`` consecutive ``whisper_decode_step_ex`` calls return, column by column."""
        toks = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.int32) for t in token_lists]), dtype=np.int32)
        off = np.zeros(len(token_lists) + 1, dtype=np.int32); np.cumsum([len(t) for t in token_lists], out=off[1:])
        rules = WhisperDecodeRules(int(eot), int(timestamp_begin), -1 if max_initial_timestamp_index is None else int(max_initial_timestamp_index), 0)
        vm = np.ascontiguousarray(vocab_mask, dtype=np.uint8)
        n = len(token_lists)
        sb = None if np.isscalar(sample_begin) else np.ascontiguousarray(sample_begin, dtype=np.int32)
        if sb is not None and sb.shape != (n,):
            raise ValueError("sample_begin: one prompt length per sequence")
        opts = WhisperDecodeOpts(sb.ctypes.data if sb is not None else None, int(sample_begin) if sb is None else 0, float(temperature),
                                 int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF, int(probe_token), 1 if no_cache else 0)
        out = np.zeros((n, int(max_new)), dtype=np.int32); lp = np.zeros((n, int(max_new)), dtype=np.float32)
        pr = np.zeros(n, dtype=np.float32) if probe_token >= 0 else None
        steps = C.c_int32()
        self._check(self._lib.pce_whisper_decode_loop(self._ctx, toks.ctypes.data, off.ctypes.data, C.byref(rules), vm.ctypes.data, C.byref(opts),
                                                      int(max_new), int(check_every), out.ctypes.data, lp.ctypes.data, C.byref(steps),
                                                      pr.ctypes.data if pr is not None else None))
        return out[:, :steps.value], lp[:, :steps.value], pr

    # ---------------------------------------------------------------- probabilistic YIN (viewers)
    def pyin_run(self, plan, tables):
        """``plan`` / ``tables`` from ``visualisation.acoustic_analysis.pyin_plan``."""
        t = np.ascontiguousarray(tables, dtype=np.float64)
        self._check(self._lib.pce_pyin_run(self._ctx, C.byref(plan), t.ctypes.data, t.size))

    def pyin_fetch(self, clip: int):
        """-> (states int32 [n_frames], voiced_prob float64 [n_frames], status)."""
        nf = C.c_int64()
        self._check(self._lib.pce_pyin_shape(self._ctx, int(clip), C.byref(nf)))
        st = np.zeros(nf.value, dtype=np.int32); vp = np.zeros(nf.value, dtype=np.float64); status = C.c_int32()
        self._check(self._lib.pce_pyin_fetch(self._ctx, int(clip), st.ctypes.data, vp.ctypes.data, C.byref(status)))
        return st, vp, status.value

    # ---------------------------------------------------------------- break-prediction token classifier
    def bert_load(self, dims: dict, weights: np.ndarray):
        """``weights``: float32 blob in ``bert_weights.tensor_order`` (``bert_weights.pack(model.state_dict(), dims)``)."""
        w = np.ascontiguousarray(weights, dtype=np.float32)
        self._bdims = BertDims(*(dims[k] for k in ("n_vocab", "n_pos", "n_type", "n_state", "n_head", "n_layer", "n_labels")))
        self._check(self._lib.pce_bert_load(self._ctx, C.byref(self._bdims), w.ctypes.data, w.size))

    def bert_run(self, token_lists):
        toks = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.int32) for t in token_lists] + [np.zeros(0, np.int32)]), dtype=np.int32)
        off = np.zeros(len(token_lists) + 1, dtype=np.int32); np.cumsum([len(t) for t in token_lists], out=off[1:])
        self._bert_lens = [len(t) for t in token_lists]
        self._check(self._lib.pce_bert_run(self._ctx, toks.ctypes.data, off.ctypes.data, len(token_lists)))

    def bert_fetch(self, seq: int):
        """-> (logits float32 [len][n_labels], labels int32 [len])."""
        n = self._bert_lens[seq]
        logits = np.zeros((n, self._bdims.n_labels), dtype=np.float32); labels = np.zeros(n, dtype=np.int32)
        self._check(self._lib.pce_bert_fetch(self._ctx, int(seq), logits.ctypes.data, labels.ctypes.data))
        return logits, labels

    def bert_token_classify(self, token_lists):
        self.bert_run(token_lists)
        return [self.bert_fetch(i) for i in range(len(token_lists))]

    def levenshtein(self, pairs):
        """Levenshtein distances of a batch of string pairs in one launch (``pce_levenshtein``): ``pairs`` = [(s1, s2), ...] of ``str``
        -> int32 array.  Characters are code points, as ``for c in s`` of Code/Aligners/levenshtein_dist_align_txtgrids.py:62."""
        if not len(pairs):
            return np.zeros(0, dtype=np.int32)

        def pack(strings):
            cps = [np.frombuffer(s.encode("utf-32-le", "surrogatepass"), dtype=np.uint32) for s in strings]
            off = np.zeros(len(cps) + 1, dtype=np.int64); np.cumsum([len(x) for x in cps], out=off[1:])
            return np.ascontiguousarray(np.concatenate(cps + [np.zeros(0, np.uint32)])), off
        a, ao = pack([p[0] for p in pairs]); b, bo = pack([p[1] for p in pairs])
        out = np.zeros(len(pairs), dtype=np.int32)
        self._check(self._lib.pce_levenshtein(self._ctx, a.ctypes.data if a.size else None, ao.ctypes.data, b.ctypes.data if b.size else None,
                                              bo.ctypes.data, len(pairs), out.ctypes.data))
        return out

    def nw_align(self, pairs, match=1, mismatch=-1, gap=-1):
        """Batched Needleman-Wunsch over integer token ids: ``pairs`` = [(ids_a, ids_b), ...] ->
        [(i_idx, j_idx), ...] with -1 marking a gap (alignment order)."""
        la = np.array([len(a) for a, _ in pairs], dtype=np.int64); lb = np.array([len(b) for _, b in pairs], dtype=np.int64)
        ao = np.zeros(len(pairs) + 1, dtype=np.int64); bo = np.zeros(len(pairs) + 1, dtype=np.int64)
        np.cumsum(la, out=ao[1:]); np.cumsum(lb, out=bo[1:])
        a = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.int32) for x, _ in pairs] + [np.zeros(0, np.int32)]))
        b = np.ascontiguousarray(np.concatenate([np.asarray(y, dtype=np.int32) for _, y in pairs] + [np.zeros(0, np.int32)]))
        oo = np.zeros(len(pairs) + 1, dtype=np.int64); np.cumsum(la + lb, out=oo[1:])
        oi = np.zeros(max(int(oo[-1]), 1), dtype=np.int32); oj = np.zeros_like(oi); ol = np.zeros(len(pairs), dtype=np.int32)
        self._check(self._lib.pce_nw_align(self._ctx, a.ctypes.data, ao.ctypes.data, b.ctypes.data, bo.ctypes.data, len(pairs),
                                           int(match), int(mismatch), int(gap), oi.ctypes.data, oj.ctypes.data, ol.ctypes.data))
        return [(oi[oo[k]:oo[k] + ol[k]].copy(), oj[oo[k]:oo[k] + ol[k]].copy()) for k in range(len(pairs))]

    # ---------------------------------------------------------------- the reference's measurement closures
    def closures(self):
        """``(get_part_duration, get_median_pitch, get_lufs, get_duration)`` with the signatures of the closures inside
        ``AudioPipeline.measure_prosody_and_build_ssml`` (Code/audioPipeline.py:314-361), answered by this engine
        (``audio_pipeline.ProsodySeam``: path-keyed, batched when the queries are announced with ``prefetch``)."""
        from .audio_pipeline import ProsodySeam
        return ProsodySeam(self).closures()

    # ---------------------------------------------------------------- measurement
    def profile_enable(self, on=True):
        self._check(self._lib.pce_profile_enable(self._ctx, 1 if on else 0))

    def profile_reset(self):
        self._check(self._lib.pce_profile_reset(self._ctx))

    def profile(self) -> dict:
        out = {}
        for i, name in enumerate(KERNEL_IDS):
            ms = C.c_double(); n = C.c_int64()
            self._check(self._lib.pce_profile_get(self._ctx, i, C.byref(ms), C.byref(n)))
            if n.value:
                fl = C.c_double()
                self._check(self._lib.pce_profile_get_work(self._ctx, i, C.byref(fl)))
                out[name] = {"total_ms": ms.value, "launches": n.value, "flops": fl.value}
        return out


_DEFAULT_ENGINE = None


def get_default_engine(device: int = None) -> ProsodyEngine:
    """Process-wide engine used by the module-level drop-in functions (``Pipeline.compute_*``).  ONE context per process (the
    reference's process owns one Whisper model on one GPU, config.yaml:58): ``device=None`` takes whatever engine exists -- when none
    does, THIS RANK's device (``shard.local_device()``: ``LOCAL_RANK`` under a one-process-per-GPU launcher, 0 without one; a device-less
    first caller such as ``Pipeline.compute_*`` must not pin rank 3 to GPU 0); naming a device other than the existing engine's is an
    error rather than a silent run on the wrong GPU."""
    global _DEFAULT_ENGINE
    if _DEFAULT_ENGINE is None:
        if device is None:
            from . import shard
            device = shard.local_device()
        _DEFAULT_ENGINE = ProsodyEngine(device)
    elif device is not None and getattr(_DEFAULT_ENGINE, "device", device) != device:
        raise RuntimeError(f"this process's engine lives on device {_DEFAULT_ENGINE.device}; device {device} was asked for "
                           "(one context per process: start one process per GPU, or close the engine first)")
    return _DEFAULT_ENGINE


def set_default_engine(engine: ProsodyEngine):
    global _DEFAULT_ENGINE
    _DEFAULT_ENGINE = engine
