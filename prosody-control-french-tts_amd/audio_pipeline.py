"""Drop-in for the hot step of ``Code/audioPipeline.py``: ``"Measure & Build SSML"``.

Same constructor contract (``AudioPipeline(name, cfg)``), same ``config.yaml`` keys and step
names (Code/audioPipeline.py:84-153, :1076-1103), same output artefacts
(``BDD_ssml.csv``, ``BDD_syntagme_ssml.csv``, ``BDD_syntagme_for_synth.csv``).  What changes is
*how* the measurements are taken: the reference re-decodes a WAV for every closure call
(O(#syntagmes x file size), :314-361); here every file of the voice is decoded once, uploaded
once, and every whole-file and per-syntagme pitch / loudness query becomes one slice of three
batched GPU passes (:class:`EngineMeasurements`).  The decision logic lives in :mod:`.tagger`.

The two Whisper steps (``"Align+Transcribe"``, ``"Final Transcribe"``, :179-241 and :856-892) run the batched
aligner of :mod:`.Aligners.use_whisper_timestamped` and leave the reference's folders.  Steps that are not on the hot
path (demucs, Azure synthesis, JSON export, break comparison: SURVEY.md section 8 marks them out of scope) are not
reimplemented: ``run()`` skips them with a warning unless ``strict_steps: true`` is set (additive config key), in
which case selecting one fails at construction.
"""
from __future__ import annotations

import json
import logging
import shutil
import sys
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import hostrules as H
from .engine import PitchParams, ProsodyEngine, SLICE_OK, SLICE_TOO_SHORT, make_slices
from .tagger import MeasurementSource, ProsodySettings, SegmentInput, SsmlTagger, segment_sort_key
from .textgrid_io import read_textgrid

STEP_NAMES = ["Preprocess", "Align+Transcribe", "Raw Synthesis", "Measure & Build SSML", "Synthesize+Merge",
              "Export JSON", "Final Transcribe", "Compare Breaks"]                    # Code/audioPipeline.py:1077-1086
ACCELERATED_STEPS = ("Align+Transcribe", "Measure & Build SSML", "Final Transcribe")


class BatchedMeasurements:
    """Decoded files (any hashable key) + planned pitch / loudness queries -> batched GPU passes.

    Files are uploaded as one resident batch per sample rate.  A loudness query names the rate of the ``pyln.Meter`` the
    reference would hand to ``get_lufs``: filter design, block length and the too-short test follow the METER's rate, not
    the data's (``ProsodyEngine.lufs_set_meter_rate``).  Queries that were not planned are answered on demand (a batch of
    one): correct, but the point of planning is that a voice costs a handful of launches."""

    def __init__(self, engine: ProsodyEngine, pitch_floor=150.0, pitch_ceiling=600.0):
        self.eng = engine
        self.floor, self.ceiling = pitch_floor, pitch_ceiling
        self.n_frames: Dict[object, int] = {}
        self.rate_of: Dict[object, int] = {}
        self._pcm: Dict[object, np.ndarray] = {}
        self.undecodable = set()
        self._pitch: Dict[tuple, float] = {}
        self._lufs: Dict[tuple, float] = {}
        self._pq: List[tuple] = []
        self._lq: List[tuple] = []

    def add_file(self, key, path) -> bool:
        if key in self.rate_of or key in self.undecodable:
            return key in self.rate_of
        try:
            rate, pcm = H.decode_wav(path)
        except H.CouldntDecodeError:
            self.undecodable.add(key)
            return False
        self.n_frames[key] = len(pcm); self.rate_of[key] = rate; self._pcm[key] = pcm
        return True

    # ------------------------------------------------------------------ planning
    @staticmethod
    def _pq_key(key, t0, t1):
        return (key, None, None) if t1 is None else (key, t0, t1)

    @staticmethod
    def _lq_key(key, t0, t1, meter):
        return (key, None, None, int(meter)) if t1 is None else (key, int(t0 * 1000), int(t1 * 1000), int(meter))

    def plan_pitch_key(self, key, t0=0.0, t1=None):
        self._pq.append(self._pq_key(key, t0, t1))

    def plan_lufs_key(self, key, t0, t1, meter_rate):
        if key in self.rate_of:
            self._lq.append(self._lq_key(key, t0, t1, meter_rate))

    def _lufs_frames(self, key, a, b, meter_rate):
        """Slice to measure for ``audio[a:b]`` including the reference's fallbacks (Code/audioPipeline.py:345-358)."""
        n = self.n_frames[key]
        if a is None:
            return 0, n
        lo, hi = H.pydub_slice_frames(n, self.rate_of[key], a, b)
        if hi - lo == 0 or hi - lo < 0.4 * meter_rate:       # empty slice, or pyloudnorm ValueError -> full segment
            return 0, n
        return lo, hi

    def run(self):
        eng = self.eng
        pq = [q for q in dict.fromkeys(self._pq) if q not in self._pitch]
        lq = [q for q in dict.fromkeys(self._lq) if q not in self._lufs]
        lspan = {q: (q[0], q[3]) + self._lufs_frames(q[0], q[1], q[2], q[3]) for q in lq}
        used = {q[0] for q in pq} | {q[0] for q in lq}
        for rate in sorted({self.rate_of[k] for k in used}):
            keys = [k for k in self.rate_of if self.rate_of[k] == rate and k in used]
            clip = {k: i for i, k in enumerate(keys)}
            eng.upload([self._pcm[k] for k in keys], rate)
            # ---- pitch: Praat extract_part(preserve_times=True) then to_pitch(floor, ceiling)
            mine = [q for q in pq if self.rate_of[q[0]] == rate]
            if mine:
                cl, b, e, x1 = [], [], [], []
                for key, t0, t1 in mine:
                    n = self.n_frames[key]
                    if t1 is None:
                        bb, ee, xx = 0, n, 0.5 / rate
                    else:
                        bb, ee, xx = H.praat_part_frames(n, rate, t0, t1, preserve_times=True)
                    cl.append(clip[key]); b.append(bb); e.append(ee); x1.append(xx)
                res = eng.pitch(make_slices(cl, b, e, x1), PitchParams.praat(self.floor, self.ceiling), want_f0=False)
                for q, sm in zip(mine, res["summary"]):
                    if sm["status"] == SLICE_TOO_SHORT:
                        # parselmouth raises PraatError here and the reference does not catch it
                        raise H.PraatError(f"{q[0]}[{q[1]}, {q[2]}]: sound shorter than 3 periods of the pitch floor")
                    self._pitch[q] = float(sm["median_f0"])
            # ---- loudness: one pass per meter rate over this batch
            for meter in sorted({v[1] for v in lspan.values() if self.rate_of[v[0]] == rate}):
                spans = {q: (clip[v[0]], v[2], v[3]) for q, v in lspan.items() if self.rate_of[v[0]] == rate and v[1] == meter}
                uniq = list(dict.fromkeys(spans.values()))
                eng.lufs_set_meter_rate(0 if meter == rate else meter)
                try:
                    vals, st = eng.lufs(make_slices([u[0] for u in uniq], [u[1] for u in uniq], [u[2] for u in uniq]))
                finally:
                    eng.lufs_set_meter_rate(0)
                table = {}
                for u, v, code in zip(uniq, vals, st):
                    table[u] = float(v) if code == SLICE_OK else None      # None: the whole file is shorter than 0.4 s of the meter's rate
                for q, u in spans.items():
                    self._lufs[q] = table[u]
        self._pq.clear(); self._lq.clear()

    # ------------------------------------------------------------------ answers
    def _need(self, key):
        if key not in self.rate_of:
            raise H.CouldntDecodeError(str(key))

    def pitch_key(self, key, t0=0.0, t1=None) -> float:
        self._need(key)
        q = self._pq_key(key, t0, t1)
        if q not in self._pitch:
            self._pq.append(q); self.run()
        return self._pitch[q]

    def lufs_key(self, key, t0, t1, meter_rate) -> float:
        self._need(key)
        q = self._lq_key(key, t0, t1, meter_rate)
        if q not in self._lufs:
            self._lq.append(q); self.run()
        v = self._lufs[q]
        if v is None:
            raise ValueError("Audio must have length greater than the block size.")      # pyloudnorm, uncaught by the reference's fallback
        return v

    def duration_key(self, key) -> float:
        self._need(key)
        return (self.n_frames[key] / self.rate_of[key]) or 1e-4

    def part_duration_key(self, key, t0=0.0, t1=None) -> float:
        self._need(key)
        rate = self.rate_of[key]
        lo, hi = H.seconds_slice_frames(self.n_frames[key], rate, t0, t1)
        return ((hi - lo) / rate) or 1e-4


class EngineMeasurements(BatchedMeasurements, MeasurementSource):
    """The tagger's ``MeasurementSource`` over :class:`BatchedMeasurements`.  ``files``: {(kind, segment): path} with kind
    "nat"/"syn".

    Sample rates: the natural recordings are 44.1 kHz and the raw synthesis is whatever the service returned (Azure's
    default RIFF output is 16 kHz, Code/Preprocessing/get_synth.py:46-51), so a voice normally mixes rates.  Loudness
    follows the reference's meters exactly: the segment-level numbers use ONE ``pyln.Meter`` built at the FIRST natural
    file's rate for natural and synthetic files alike (Code/audioPipeline.py:372), the syntagme-level numbers a meter at
    the segment's own natural file's rate (:493)."""

    def __init__(self, engine: ProsodyEngine, files: Dict[Tuple[str, str], Path], pitch_floor=150.0, pitch_ceiling=600.0,
                 first_nat_rate: Optional[int] = None):
        BatchedMeasurements.__init__(self, engine, pitch_floor, pitch_ceiling)
        for key, path in files.items():
            self.add_file(key, path)
        nat = sorted((k for k in self.rate_of if k[0] == "nat"), key=lambda k: segment_sort_key(k[1]))
        # the segment-level meter (:372) is built at the rate of the voice's FIRST natural file; a rank that holds only a block of
        # the voice is told that rate (``first_nat_rate``), it does not follow from the rank's own files
        self.first_nat_rate = int(first_nat_rate) if first_nat_rate else (self.rate_of[nat[0]] if nat else 0)

    def _meter_rate(self, segment, t1) -> int:
        if t1 is None:
            return self.first_nat_rate
        return self.rate_of.get(("nat", segment), self.first_nat_rate)

    def plan_pitch(self, segment, t0=0.0, t1=None):
        self.plan_pitch_key(("nat", segment), t0, t1)

    def plan_lufs(self, kind, segment, t0=0.0, t1=None):
        self.plan_lufs_key((kind, segment), t0, t1, self._meter_rate(segment, t1))

    def median_pitch(self, segment, t0=0.0, t1=None):
        return self.pitch_key(("nat", segment), t0, t1)

    def lufs(self, kind, segment, t0=0.0, t1=None):
        return self.lufs_key((kind, segment), t0, t1, self._meter_rate(segment, t1))

    def duration(self, kind, segment):
        return self.duration_key((kind, segment))

    def part_duration(self, kind, segment, t0=0.0, t1=None):
        return self.part_duration_key((kind, segment), t0, t1)


class ProsodySeam(BatchedMeasurements):
    """The four measurement closures of ``AudioPipeline.measure_prosody_and_build_ssml`` (Code/audioPipeline.py:314-361)
    with the reference's own signatures, keyed by WAV path, so that the reference's method can be rebound to the engine
    without this package's ``AudioPipeline``:

        seam = ProsodySeam(engine)
        get_part_duration, get_median_pitch, get_lufs, get_duration = seam.closures()

    Units and sentinels are the reference's: seconds in, Hz / LUFS / seconds out; ``0.0`` = no voiced frame (:335);
    durations floored at ``1e-4`` (:322-323,:361); ``peak or 1.0`` (:349); a slice shorter than 0.4 s of the meter's rate,
    or empty, falls back to the whole file (:345-358); an undecodable file raises ``CouldntDecodeError``.  ``meter`` is
    anything with a ``rate`` attribute (``pyloudnorm.Meter``) or the rate itself.  ``prefetch`` batches queries known in
    advance (every query of a voice follows from its TextGrids); anything else is computed on first use."""

    def _key(self, wav_path):
        key = str(wav_path)
        self.add_file(key, key)
        return key

    def prefetch(self, pitch_queries=(), lufs_queries=()):
        """pitch_queries: (path, t0, t1 | None); lufs_queries: (path, meter_rate, t0, t1 | None)."""
        for path, t0, t1 in pitch_queries:
            k = self._key(path)
            if k in self.rate_of:
                self.plan_pitch_key(k, t0, t1)
        for path, meter, t0, t1 in lufs_queries:
            k = self._key(path)
            self.plan_lufs_key(k, t0, t1, getattr(meter, "rate", meter))
        self.run()

    def closures(self):
        def get_part_duration(wav_path, t0=0.0, t1=None):
            return self.part_duration_key(self._key(wav_path), t0, t1)

        def get_median_pitch(wav_path, t0=0.0, t1=None):
            return self.pitch_key(self._key(wav_path), t0, t1)

        def get_lufs(wav_path, meter, t0=0.0, t1=None):
            return self.lufs_key(self._key(wav_path), t0, t1, getattr(meter, "rate", meter))

        def get_duration(wav_path):
            return self.duration_key(self._key(wav_path))

        return get_part_duration, get_median_pitch, get_lufs, get_duration


class _Planner(MeasurementSource):
    """First pass of the tagger: records the queries, answers with neutral values."""

    def __init__(self, em: EngineMeasurements):
        self.em = em

    def median_pitch(self, segment, t0=0.0, t1=None):
        self.em.plan_pitch(segment, t0, t1); return 200.0

    def lufs(self, kind, segment, t0=0.0, t1=None):
        self.em._need((kind, segment)); self.em.plan_lufs(kind, segment, t0, t1); return -23.0

    def duration(self, kind, segment):
        return self.em.duration(kind, segment)

    def part_duration(self, kind, segment, t0=0.0, t1=None):
        return self.em.part_duration(kind, segment, t0, t1)


class _FailedSource(MeasurementSource):
    """Stands in for the measurements of a rank whose local stage (decode / plan / GPU passes) raised: every query re-raises that
    error, inside ``SsmlTagger.run_sharded``'s guarded section, which still enters the collective."""

    def __init__(self, error):
        self.error = error

    def median_pitch(self, segment, t0=0.0, t1=None): raise self.error
    def lufs(self, kind, segment, t0=0.0, t1=None): raise self.error
    def duration(self, kind, segment): raise self.error
    def part_duration(self, kind, segment, t0=0.0, t1=None): raise self.error


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
        from . import shard
        # "cuda:3" names the device; a bare "cuda" under a one-process-per-GPU launcher means this rank's own GPU (LOCAL_RANK)
        self.device_index = (int(str(self.whisper_device).split(":")[1]) if ":" in str(self.whisper_device)
                             else shard.local_device())
        self._engine, self._nlp, self._refine_set = engine, nlp, False
        self.results_dir.mkdir(parents=True, exist_ok=True)
        wanted = cfg.get("steps_to_run") or STEP_NAMES
        unknown = [n for n in wanted if n not in STEP_NAMES]
        if unknown:
            raise ValueError(f"steps_to_run names unknown steps {unknown}; known: {STEP_NAMES}")
        outside = [n for n in wanted if n not in ACCELERATED_STEPS]
        if outside and cfg.get("strict_steps"):
            raise NotImplementedError(f"steps {outside} are outside the accelerated hot path (SURVEY.md section 8): run them with the "
                                      "reference implementation, or drop strict_steps to skip them")

    def _get_engine(self) -> ProsodyEngine:
        if self._engine is None:
            from .engine import get_default_engine
            self._engine = get_default_engine(self.device_index)             # ONE context per process: the aligner's steps use the same one
        if not self._refine_set:
            # additive key: "seeded" (default) | "praat" (Praat's own iterates).  The context is shared by every pipeline of the process, so
            # the mode is set by every pipeline, default included: a voice without the key does not inherit the previous voice's choice
            if hasattr(self._engine, "pitch_set_refine"):
                self._engine.pitch_set_refine(self.cfg.get("pitch_refine") or "seeded")
            self._refine_set = True
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
        tagger = SsmlTagger(self.settings, self.azure_voice, nlp=self._nlp)
        from . import shard
        rank, world = shard.rank_world()
        if world > 1 or self.cfg.get("force_sharded_path"):
            # One process per GPU (torch.distributed initialised by the launcher): this rank decodes, uploads and measures only its
            # contiguous block of the segment-sorted list; ONE all-gather of the per-segment / per-syntagme records (RCCL over xGMI
            # under the nccl backend); baselines, adjustments, the EMA over all syntagmes and the SSML strings on every rank
            # (Code/audioPipeline.py:364-424, :592-602).  Rank 0 writes the three tables.
            lo, hi = shard.shard_range(len(segments), rank, world)
            mine = segments[lo:hi]
            try:
                local_files = {k: v for k, v in files.items() if k[1] in {sg.name for sg in mine}}
                first_rate = None
                try:
                    first_rate = H.decode_wav(wavs[0])[0]                    # the voice's first natural file fixes the segment-level meter (:372)
                except H.CouldntDecodeError:
                    pass
                em = EngineMeasurements(self._get_engine(), local_files, first_nat_rate=first_rate)
                planner = _Planner(em)
                tagger.segment_statistics(mine, planner); tagger.syntagme_measurements(mine, planner)     # pass 1 over the local block: collect the queries
                em.run()
            except Exception as e:                                           # noqa: BLE001
                em = _FailedSource(e)                                        # run_sharded flags this rank's block in the ONE collective: all ranks fail together
            res = tagger.run_sharded(segments, em, rank, world, shard.allgather_records)
            with shard.agreed(only_rank=0) as sec:                           # the tables exist (or every rank knows they do not) when any rank returns
                if sec.mine:
                    res.bdd_ssml.to_csv(self.bdd_ssml_csv, index=False)
                    res.bdd_syntagme_ssml.to_csv(self.bdd_syntagme_ssml_csv, index=False)
                    res.bdd_syntagme_for_synth.to_csv(self.bdd_syntagme_synth_csv, index=False)
            return res
        em = EngineMeasurements(self._get_engine(), files)
        tagger.run(segments, _Planner(em))         # pass 1: collect every query (they depend on the TextGrids only)
        em.run()                                   # three batched GPU passes
        res = tagger.run(segments, em)             # pass 2: the real numbers
        res.bdd_ssml.to_csv(self.bdd_ssml_csv, index=False)
        res.bdd_syntagme_ssml.to_csv(self.bdd_syntagme_ssml_csv, index=False)
        res.bdd_syntagme_for_synth.to_csv(self.bdd_syntagme_synth_csv, index=False)
        return res

    # ------------------------------------------------------------------ the Whisper steps
    def _whisper_main(self, audio_folder, out_folder):
        from .Aligners import use_whisper_timestamped as A
        if self.cfg.get("whisper_dir"):
            A.set_model_source(model_dir=str(self.cfg["whisper_dir"]))
        if self._nlp is not None:
            A.set_nlp(self._nlp)
        if self._engine is not None:
            from .engine import set_default_engine
            set_default_engine(self._engine)
        A.main(str(audio_folder), str(out_folder), whisper_model=self.whisper_model, device=self.whisper_device, logger=logging.getLogger())

    def align_and_transcribe(self):
        """Code/audioPipeline.py:179-241: fresh output folders, the aligner over ``<voice>/audio``, raw transcriptions from
        the raw JSONs ("..." where a file produced none), cleaned transcriptions from the TextGrids."""
        logging.info(">>> Align & Transcribe: WhisperTS")
        from .Aligners.use_whisper_timestamped import remove_spurious_commas
        from .Pipeline.utils import save_clean_transcriptions_from_textgrids
        audio_folder = self.voice_dir / "audio"
        tg_folder = self.voice_dir / "WhisperTS_textgrid_files"
        txt_folder = self.voice_dir / "transcription"
        txt_raw_folder = self.voice_dir / "transcription_raw"
        raw_json_dir = Path(str(tg_folder) + "_raw_json")
        from . import shard
        # Under one process per GPU the folders are shared: rank 0 alone resets them BEFORE any rank's aligner writes into them, and rank 0
        # alone post-processes AFTER every rank's files exist (A.main ends on a status barrier); each section is left through
        # ``shard.agreed`` so that a failure on one rank fails the step on all of them.
        with shard.agreed(only_rank=0) as sec:
            if sec.mine:
                for d in (tg_folder, txt_folder, self.voice_dir / "WhisperTS_textgrid_files_transcription", txt_raw_folder, raw_json_dir):
                    shutil.rmtree(d, ignore_errors=True)
                for d in (tg_folder, txt_folder, txt_raw_folder, raw_json_dir):
                    d.mkdir(parents=True, exist_ok=True)
        self._whisper_main(audio_folder, tg_folder)
        with shard.agreed(only_rank=0) as sec:
            if sec.mine:
                for js in raw_json_dir.glob("*.raw.json"):
                    data = json.loads(js.read_text(encoding="utf-8"))
                    (txt_raw_folder / js.name.replace(".raw.json", ".txt")).write_text(" ".join(seg["text"] for seg in data["segments"]), encoding="utf-8")
                for wav in Path(audio_folder).glob("*.wav"):
                    raw_txt = txt_raw_folder / f"{wav.stem}.txt"
                    if not raw_txt.exists():
                        raw_txt.write_text("...", encoding="utf-8")
                save_clean_transcriptions_from_textgrids(tg_folder, txt_folder)
                for txt in Path(txt_folder).glob("*.txt"):
                    txt.write_text(remove_spurious_commas(txt.read_text(encoding="utf-8")), encoding="utf-8")

    def final_transcribe(self):
        """Code/audioPipeline.py:856-892: the aligner over ``results/<voice>/OUT.wav``, results moved next to it."""
        logging.info(">>> Final Transcribe: WhisperTS on OUT.wav")
        from .Pipeline.utils import save_clean_transcriptions_from_textgrids
        out_wav = self.results_dir / "OUT.wav"
        if not out_wav.exists():
            logging.error(f"No OUT.wav found at {out_wav}")
            return
        from . import shard
        temp_dir = self.results_dir / "final_whisper"
        tg_dir, txt_dir = temp_dir / "WhisperTS_textgrid_files", temp_dir / "transcription_final"
        with shard.agreed(only_rank=0) as sec:
            if sec.mine:
                tg_dir.mkdir(parents=True, exist_ok=True); txt_dir.mkdir(parents=True, exist_ok=True)
        self._whisper_main(out_wav.parent, tg_dir)
        with shard.agreed(only_rank=0) as sec:                               # (every rank would rename the same files: rank 0 does, the others wait)
            if sec.mine:
                save_clean_transcriptions_from_textgrids(tg_dir, txt_dir)
                for tg in tg_dir.glob("*.TextGrid"):
                    tg.rename(self.results_dir / tg.name)
                for txt in txt_dir.glob("*.txt"):
                    txt.rename(self.results_dir / txt.name)
        logging.info(f"Final transcription files saved in {self.results_dir}")

    # ------------------------------------------------------------------ break prediction (BASELINE.json configs[4]: additive, not a reference step)
    def predict_breaks(self, word_piecer=None, weights=None, dims=None, cls_id: int = 101, sep_id: int = 102):
        """Break prediction for every segment of the voice with the token classifier the reference trains
        (Code/baseline_models/pause_bert.py:14-21,127-132; the reference has no inference step: this is the forward a pipeline would
        call).  One sentence per segment = the words of its cleaned transcription (``<voice>/transcription/<segment>.txt``).

        ``word_piecer(word) -> [sub-token ids]``: the checkpoint's WordPiece split (``transformers.BertTokenizer(vocab_file)`` offline;
        additive config key ``break_bert_vocab``); ``weights`` / ``dims``: a ``BertForTokenClassification`` state dict and its dims
        (additive key ``break_bert_weights``: ``.npz`` / ``.safetensors``), loaded into the engine once.

        Sharding: sentences are independent, so under an initialised ``torch.distributed`` rank r classifies a contiguous block of the
        segment-sorted list and ONE all-gather of the 0 / 1 labels (padded to the longest sentence) gives every rank the whole table;
        rank 0 writes ``results/<voice>/BDD_breaks.csv`` (segment, word_index, word, break)."""
        from . import bert_weights as BW, shard
        from .Preprocessing import break_bert as BB
        txts = sorted(self.transcription_dir.glob("*.txt"), key=lambda p: segment_sort_key(p.stem))
        words = [t.read_text(encoding="utf-8").split() for t in txts]
        rank, world = shard.rank_world()
        lo, hi = shard.shard_range(len(txts), rank, world)
        width = max([len(ws) for ws in words] + [1])
        counts = [b - a for a, b in (shard.shard_range(len(txts), r, world) for r in range(world))]
        local_error = None
        try:                                                                 # everything that can fail on ONE rank (device, weight file, vocabulary)
            eng = self._get_engine()
            if weights is None and self.cfg.get("break_bert_weights"):
                path = str(self.cfg["break_bert_weights"])
                if path.endswith(".npz"):
                    with np.load(path) as z:
                        weights = {k: z[k] for k in z.files}
                else:
                    from safetensors.numpy import load_file
                    weights = load_file(path)
            if weights is not None:
                if dims is None:
                    d = weights["bert.embeddings.word_embeddings.weight"].shape
                    n_layer = 1 + max(int(k.split(".")[3]) for k in weights if k.startswith("bert.encoder.layer."))
                    dims = dict(n_vocab=int(d[0]), n_pos=int(weights["bert.embeddings.position_embeddings.weight"].shape[0]),
                                n_type=int(weights["bert.embeddings.token_type_embeddings.weight"].shape[0]), n_state=int(d[1]), n_head=int(d[1]) // 64,
                                n_layer=n_layer, n_labels=int(weights["classifier.weight"].shape[0]))
                eng.bert_load(dims, BW.pack(weights, dims))
            if word_piecer is None:
                vocab = self.cfg.get("break_bert_vocab")
                if not vocab:
                    raise FileNotFoundError('break prediction needs the checkpoint\'s WordPiece vocabulary: set "break_bert_vocab" (vocab.txt) or pass word_piecer')
                from transformers import BertTokenizer
                tok = BertTokenizer(str(vocab), do_lower_case=True)
                cls_id, sep_id = tok.cls_token_id, tok.sep_token_id
                word_piecer = lambda w: tok.convert_tokens_to_ids(tok.tokenize(w))
            mine = [[word_piecer(w) for w in ws] for ws in words[lo:hi]]
            max_len = min(BW.MAX_LENGTH, int(dims["n_pos"])) if dims else BW.MAX_LENGTH       # (pause_bert.py:16; a smaller checkpoint truncates at its own table)
            local = BB.predict_breaks(eng, mine, cls_id, sep_id, max_len) if mine else []
            rec = np.full((len(local), width), -1.0)
            for i, lab in enumerate(local):
                rec[i, :len(lab)] = lab
        except Exception as e:                                               # noqa: BLE001  (the rank still enters the ONE collective, flagged)
            local_error, rec = e, np.zeros((0, width))
        try:
            allrec = shard.allgather_records(rec, counts, failed=local_error is not None)
        except shard.PeerFailure:
            if local_error is not None:
                raise local_error
            raise
        labels = [[int(v) for v in allrec[i, :len(ws)]] for i, ws in enumerate(words)]
        with shard.agreed(only_rank=0) as sec:
            if sec.mine:
                import pandas as pd
                rows = [{"segment": t.stem, "word_index": k, "word": w, "break": b} for t, ws, lab in zip(txts, words, labels) for k, (w, b) in enumerate(zip(ws, lab))]
                pd.DataFrame(rows, columns=["segment", "word_index", "word", "break"]).to_csv(self.results_dir / "BDD_breaks.csv", index=False)
        return dict(zip([t.stem for t in txts], labels))

    # ------------------------------------------------------------------ step table
    def run(self):
        """Code/audioPipeline.py:1076-1103: the selected steps in the fixed order, ``sys.exit(1)`` when one raises, then
        ``used_config.yaml`` in the results folder.  Steps outside the hot path are skipped with a warning."""
        steps = {"Align+Transcribe": self.align_and_transcribe, "Measure & Build SSML": self.measure_prosody_and_build_ssml,
                 "Final Transcribe": self.final_transcribe}
        wanted = self.cfg.get("steps_to_run") or STEP_NAMES
        for n in [n for n in STEP_NAMES if n in wanted]:
            if n not in steps:
                logging.warning(f'step "{n}" is outside the accelerated hot path (SURVEY.md section 8): skipped, run it with the reference implementation')
                continue
            try:
                steps[n]()
            except Exception:
                logging.exception(f"Failed step {n}")
                sys.exit(1)
        import yaml
        from . import shard
        config_path = self.results_dir / "used_config.yaml"
        with shard.agreed(only_rank=0) as sec:
            if sec.mine:
                with open(config_path, "w", encoding="utf-8") as f:
                    yaml.dump(self.cfg, f, default_flow_style=False, allow_unicode=True)
        logging.info(f"Config saved to {config_path}")


# ---------------------------------------------------------------------------------------------------------------
# the driver of Code/audioPipeline.py:1105-1160, mapped onto one process per GPU
# ---------------------------------------------------------------------------------------------------------------
def run_pipeline_for_voice(args):
    """``run_pipeline_for_voice((name, cfg))`` of the reference (:1105-1119): (success, name)."""
    name, cfg = args
    base = cfg.get("_base")
    logging.info(f"--- Starting pipeline for: {name} ---")
    try:
        AudioPipeline(name, {k: v for k, v in cfg.items() if k != "_base"}, base=base).run()
        logging.info(f"--- Finished pipeline for: {name} ---")
        return True, name
    except SystemExit as e:                                                  # run() exits on a failed step, as the reference does
        logging.error(f"--- Pipeline failed for: {name} --- (exit {e.code})")
        return False, name
    except Exception as e:                                                   # noqa: BLE001
        logging.error(f"--- Pipeline failed for: {name} ---")
        logging.exception(e)
        return False, name


def run_all(cfg, base=None):
    """The ``__main__`` block of the reference (:1121-1160).  The reference parallelises over VOICES with a ``spawn`` pool of
    ``num_processes`` workers, each owning a Whisper model on the one GPU (config.yaml:58).  Here the unit of parallelism is the
    GPU: launched as one process per GPU (``python -m torch.distributed.run --nproc-per-node 8 ... -m prosody_control_french_tts_amd.audio_pipeline
    config.yaml``), every voice is processed by ALL ranks together -- each step shards the voice's utterances over the ranks
    (``measure_prosody_and_build_ssml``: one all-gather; the aligner: none; ``predict_breaks``: one all-gather) -- and the voices
    follow one another.  ``multiprocessing`` / ``num_processes`` are accepted and ignored under a process group; without one this is
    the reference's sequential loop on one GPU.  -> the list of voices that failed."""
    from . import shard
    names = cfg.get("voice_names")
    if not names:
        logging.error("Missing 'voice_names' in config.yaml")
        sys.exit(1)
    if isinstance(names, str):
        names = [names]
    elif not isinstance(names, list):
        logging.error("'voice_names' in config.yaml must be a string or a list")
        sys.exit(1)
    rank, world = shard.rank_world()
    logging.info(f"Processing voices: {', '.join(names)}" + (f" (rank {rank} of {world}: utterances of every voice sharded over the ranks)" if world > 1 else ""))
    if cfg.get("multiprocessing") and world > 1:
        logging.info("multiprocessing / num_processes ignored: parallelism comes from the one-process-per-GPU launch")
    failed = []
    for name in names:
        ok, _ = run_pipeline_for_voice((name, dict(cfg, _base=base)))
        if not ok:
            failed.append(name)
    if failed:
        logging.error(f"Some pipelines failed: {', '.join(failed)}")
    return failed


def main(argv=None):
    """``python -m prosody_control_french_tts_amd.audio_pipeline [config.yaml]``; initialises torch.distributed (RCCL; gloo with
    ``PCE_DIST_BACKEND=gloo``, see ``shard.init_from_env``) when started by a launcher that sets WORLD_SIZE > 1."""
    import yaml
    from . import shard
    argv = sys.argv[1:] if argv is None else argv
    path = Path(argv[0]) if argv else Path("config.yaml")
    with open(path, encoding="utf-8") as f:
        cfg = yaml.safe_load(f)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
    _, world, _ = shard.init_from_env()
    failed = run_all(cfg, base=path.resolve().parent)
    if world > 1:
        import torch.distributed as dist
        dist.barrier(); dist.destroy_process_group()
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
