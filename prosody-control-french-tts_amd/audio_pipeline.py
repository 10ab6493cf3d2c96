"""Drop-in for the hot step of ``Code/audioPipeline.py``: ``"Measure & Build SSML"``.

Same constructor contract (``AudioPipeline(name, cfg)``), same ``config.yaml`` keys and step
names (Code/audioPipeline.py:84-153, :1076-1103), same output artefacts
(``BDD_ssml.csv``, ``BDD_syntagme_ssml.csv``, ``BDD_syntagme_for_synth.csv``).  What changes is
*how* the measurements are taken: the reference re-decodes a WAV for every closure call
(O(#syntagmes x file size), :314-361); here every file of the voice is decoded once, uploaded
once, and every whole-file and per-syntagme pitch / loudness query becomes one slice of three
batched GPU passes (:class:`EngineMeasurements`).  The decision logic lives in :mod:`.tagger`.

Steps that are not on the hot path (Azure synthesis, demucs, JSON export, break comparison)
are not reimplemented: selecting them raises ``NotImplementedError`` naming the step.
"""
from __future__ import annotations

import logging
import sys
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import hostrules as H
from .engine import PitchParams, ProsodyEngine, SLICE_OK, SLICE_TOO_SHORT, make_slices
from .tagger import MeasurementSource, ProsodySettings, SegmentInput, SsmlTagger, segment_sort_key
from .textgrid_io import read_textgrid

STEP_NAMES = ["Preprocess", "Align+Transcribe", "Raw Synthesis", "Measure & Build SSML", "Synthesize+Merge",
              "Export JSON", "Final Transcribe", "Compare Breaks"]                    # Code/audioPipeline.py:1077-1086


class EngineMeasurements(MeasurementSource):
    """Answers the tagger's queries from batched GPU passes over a resident batch.

    ``files``: {(kind, segment): path} with kind "nat"/"syn".  Every decodable file becomes a
    clip; ``plan`` collects the (segment, t0, t1) queries the tagger will make, ``run`` executes
    them (pitch: Praat ``extract_part`` sample rule; loudness: pydub ms rule + the reference's
    whole-file fallbacks, decided on the host from the slice length)."""

    def __init__(self, engine: ProsodyEngine, files: Dict[Tuple[str, str], Path], pitch_floor=150.0, pitch_ceiling=600.0):
        self.eng = engine
        self.floor, self.ceiling = pitch_floor, pitch_ceiling
        self.clip_of: Dict[Tuple[str, str], int] = {}
        self.n_frames: Dict[Tuple[str, str], int] = {}
        self.undecodable = set()
        clips, rates = [], set()
        for key, path in files.items():
            try:
                rate, pcm = H.decode_wav(path)
            except H.CouldntDecodeError:
                self.undecodable.add(key)
                continue
            self.clip_of[key] = len(clips); self.n_frames[key] = len(pcm)
            clips.append(pcm); rates.add(rate)
        if len(rates) > 1:
            raise ValueError(f"all files of a voice must share one sample rate, found {sorted(rates)}")
        self.rate = rates.pop() if rates else 0
        if clips:
            engine.upload(clips, self.rate)
        self._pitch: Dict[Tuple[str, Optional[float], Optional[float]], float] = {}
        self._lufs: Dict[Tuple[str, str, Optional[int], Optional[int]], float] = {}
        self._pq: List[Tuple[str, Optional[float], Optional[float]]] = []
        self._lq: List[Tuple[str, str, Optional[int], Optional[int]]] = []

    # ------------------------------------------------------------------ planning
    def plan_pitch(self, segment, t0=0.0, t1=None):
        self._pq.append((segment, None, None) if t1 is None else (segment, t0, t1))

    def plan_lufs(self, kind, segment, t0=0.0, t1=None):
        if (kind, segment) in self.clip_of:
            self._lq.append((kind, segment, None, None) if t1 is None else (kind, segment, int(t0 * 1000), int(t1 * 1000)))

    def _lufs_frames(self, kind, segment, a, b):
        """Slice to measure for ``audio[a:b]`` including the reference's fallbacks (None = whole file)."""
        n = self.n_frames[(kind, segment)]
        if a is None:
            return 0, n
        lo, hi = H.pydub_slice_frames(n, self.rate, a, b)
        if hi - lo == 0 or hi - lo < 0.4 * self.rate:       # empty slice, or pyloudnorm ValueError -> full segment
            return 0, n
        return lo, hi

    def run(self):
        eng, rate = self.eng, self.rate
        # ---- pitch: Praat extract_part(preserve_times=True) then to_pitch(floor, ceiling)
        pq = list(dict.fromkeys(self._pq))
        if pq:
            cl, b, e, x1 = [], [], [], []
            for seg, t0, t1 in pq:
                n = self.n_frames[("nat", seg)]
                if t1 is None:
                    bb, ee, xx = 0, n, 0.5 / rate
                else:
                    bb, ee, xx = H.praat_part_frames(n, rate, t0, t1, preserve_times=True)
                cl.append(self.clip_of[("nat", seg)]); b.append(bb); e.append(ee); x1.append(xx)
            res = eng.pitch(make_slices(cl, b, e, x1), PitchParams.praat(self.floor, self.ceiling), want_f0=False)
            for q, s in zip(pq, res["summary"]):
                if s["status"] == SLICE_TOO_SHORT:
                    # parselmouth raises PraatError here and the reference does not catch it
                    raise H.PraatError(f"{q[0]}[{q[1]}, {q[2]}]: sound shorter than 3 periods of the pitch floor")
                self._pitch[q] = float(s["median_f0"])
        # ---- loudness
        lq = list(dict.fromkeys(self._lq))
        if lq:
            spans = {}
            for kind, seg, a, b in lq:
                spans[(kind, seg, a, b)] = (self.clip_of[(kind, seg)],) + self._lufs_frames(kind, seg, a, b)
            uniq = list(dict.fromkeys(spans.values()))
            vals, st = eng.lufs(make_slices([u[0] for u in uniq], [u[1] for u in uniq], [u[2] for u in uniq]))
            table = {}
            for u, v, code in zip(uniq, vals, st):
                if code != SLICE_OK:
                    raise ValueError("Audio must have length greater than the block size.")   # whole file < 0.4 s
                table[u] = float(v)
            for q, u in spans.items():
                self._lufs[q] = table[u]
        self._pq.clear(); self._lq.clear()

    # ------------------------------------------------------------------ MeasurementSource
    def _need(self, kind, segment):
        if (kind, segment) not in self.clip_of:
            raise H.CouldntDecodeError(f"{kind}:{segment}")

    def median_pitch(self, segment, t0=0.0, t1=None):
        return self._pitch[(segment, None, None) if t1 is None else (segment, t0, t1)]

    def lufs(self, kind, segment, t0=0.0, t1=None):
        self._need(kind, segment)
        return self._lufs[(kind, segment, None, None) if t1 is None else (kind, segment, int(t0 * 1000), int(t1 * 1000))]

    def duration(self, kind, segment):
        self._need(kind, segment)
        return (self.n_frames[(kind, segment)] / self.rate) or 1e-4

    def part_duration(self, kind, segment, t0=0.0, t1=None):
        self._need(kind, segment)
        n = self.n_frames[(kind, segment)]
        lo, hi = H.seconds_slice_frames(n, self.rate, t0, t1)
        return ((hi - lo) / self.rate) or 1e-4


class _Planner(MeasurementSource):
    """First pass of the tagger: records the queries, answers with neutral values."""

    def __init__(self, em: EngineMeasurements):
        self.em = em

    def median_pitch(self, segment, t0=0.0, t1=None):
        self.em.plan_pitch(segment, t0, t1); return 200.0

    def lufs(self, kind, segment, t0=0.0, t1=None):
        self.em._need(kind, segment); self.em.plan_lufs(kind, segment, t0, t1); return -23.0

    def duration(self, kind, segment):
        return self.em.duration(kind, segment)

    def part_duration(self, kind, segment, t0=0.0, t1=None):
        return self.em.part_duration(kind, segment, t0, t1)


class AudioPipeline:
    def __init__(self, name, cfg, base: Optional[Path] = None, engine: Optional[ProsodyEngine] = None, nlp=None):
        self.name, self.cfg = name, cfg
        base = Path(base) if base is not None else Path.cwd()
        self.data_dir = base / cfg["data_dir"]
        self.out_dir = base / cfg["out_dir"]
        self.voice_dir = self.data_dir / name
        self.raw_synth_dir = self.data_dir / f"{name}_raw"
        self.results_dir = self.out_dir / "results" / name
        self.textgrid_dir = self.voice_dir / "WhisperTS_textgrid_files"
        self.transcription_dir = self.voice_dir / "transcription"
        self.raw_audio_dir = self.raw_synth_dir / "audio"
        self.bdd_ssml_csv = self.results_dir / "BDD_ssml.csv"
        self.bdd_syntagme_ssml_csv = self.results_dir / "BDD_syntagme_ssml.csv"
        self.bdd_syntagme_synth_csv = self.results_dir / "BDD_syntagme_for_synth.csv"
        self.azure_voice = cfg.get("azure_voice_name", "fr-FR-HenriNeural")
        self.whisper_device = cfg.get("whisper_device", "cuda")
        self.whisper_model = cfg.get("whisper_model", "turbo")
        self.settings = ProsodySettings.from_config(cfg.get("prosody_settings", {}))
        self.device_index = int(str(self.whisper_device).split(":")[1]) if ":" in str(self.whisper_device) else 0
        self._engine, self._nlp = engine, nlp
        self.results_dir.mkdir(parents=True, exist_ok=True)

    def _get_engine(self) -> ProsodyEngine:
        if self._engine is None:
            self._engine = ProsodyEngine(self.device_index)
        return self._engine

    # ------------------------------------------------------------------ the hot step
    def measure_prosody_and_build_ssml(self):
        logging.info(">>> Measure Prosody & Build SSML")
        wavs = sorted(self.voice_dir.joinpath("audio").glob("*.wav"), key=lambda p: segment_sort_key(p.stem))
        if not wavs:
            logging.error("No audio segments found!")
            return None
        files, segments = {}, []
        for w in wavs:
            files[("nat", w.stem)] = w
            files[("syn", w.stem)] = self.raw_audio_dir / f"{w.stem}.wav"
            tg = read_textgrid(self.textgrid_dir / f"{w.stem}.TextGrid")
            segments.append(SegmentInput(w.stem, tg.tiers[0].intervals))
        em = EngineMeasurements(self._get_engine(), files)
        tagger = SsmlTagger(self.settings, self.azure_voice, nlp=self._nlp)
        tagger.run(segments, _Planner(em))         # pass 1: collect every query (they depend on the TextGrids only)
        em.run()                                   # three batched GPU passes
        res = tagger.run(segments, em)             # pass 2: the real numbers
        res.bdd_ssml.to_csv(self.bdd_ssml_csv, index=False)
        res.bdd_syntagme_ssml.to_csv(self.bdd_syntagme_ssml_csv, index=False)
        res.bdd_syntagme_for_synth.to_csv(self.bdd_syntagme_synth_csv, index=False)
        return res

    # ------------------------------------------------------------------ step table
    def _not_on_hot_path(self, step):
        def f():
            raise NotImplementedError(f'step "{step}" is outside the accelerated hot path (SURVEY.md section 8): '
                                      "run it with the reference implementation")
        return f

    def run(self):
        steps = {n: self._not_on_hot_path(n) for n in STEP_NAMES}
        steps["Measure & Build SSML"] = self.measure_prosody_and_build_ssml
        wanted = self.cfg.get("steps_to_run")
        order = [n for n in STEP_NAMES if (not wanted or n in wanted)]
        for n in order:
            try:
                steps[n]()
            except Exception:
                logging.exception(f"Failed at step: {n}")
                sys.exit(1)
