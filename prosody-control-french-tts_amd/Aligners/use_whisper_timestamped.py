"""Drop-in for ``Code/Aligners/use_whisper_timestamped.py``: WAV directory in, TextGrids + transcriptions out.

Same entry point and side effects as the reference (``main(audio_path, out_path, whisper_model, device, logger)``,
:501-728): ``out_path/<n>.TextGrid``, ``out_path + "_transcription"/<n>.{json,txt}`` (:552,:573-574),
``out_path + "_raw_json"/<n>.raw.json`` (:627-631), ``out_path + "_textgrid_raw"/<n>.TextGrid`` (:634-638), "..."
placeholders for gated and failed files (:604-621,:666-701), matching placeholder grids between a voice and its
``_microsoft`` sibling (:703-722), ``sys.exit(1)`` on a missing directory or a fatal error (:516-518,:725-728),
``sys.exit(0)`` when there is no WAV (:546-548).

What differs is how the work is done: the reference loads, gates and transcribes one file after the other; here the
directory is decoded once, gated in one ``k_energy`` launch and transcribed as resident batches
(``Aligners/transcribe.py``: VAD, log-mel, encoder, greedy decoding, forced alignment -- all ``libpce.so`` kernels).
The trained weights and the vocabulary come from a directory the caller names (``PCE_WHISPER_DIR`` or
``set_model_source``), because nothing can be downloaded by this build; without them ``load_model`` raises
``FileNotFoundError`` naming what is missing, and ``main`` exits 1 like the reference does on a failed model load.
"""
from __future__ import annotations

import json
import logging
import os
import re
import sys
import traceback
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np

from .. import hostrules as H
from .. import shard
from ..engine import ProsodyEngine, get_default_engine
from ..tagger import FORBIDDEN_POS, TablePosTagger
from ..textgrid_io import IntervalTier, TextGrid, read_textgrid, words_to_textgrid, write_textgrid  # noqa: F401  (re-exported)
from . import checkpoint as CK
from . import transcribe as TR

_PAUSE_MARKERS = {"[*]"}
_nlp = TablePosTagger()                  # the reference loads spaCy fr_core_news_sm at import (:29); pass a spaCy pipeline to set_nlp()
_MODEL_SOURCE: Dict[str, Any] = {"dir": None, "model": None, "tokenizer": None}
BATCH_CLIPS = 64                         # files transcribed per resident batch
TOKENS_PER_SECOND = 50                   # openai-whisper audio.py: 20 ms per audio token


def set_nlp(nlp):
    """Use a spaCy pipeline (``spacy.load("fr_core_news_sm", disable=["ner"])``) instead of the closed-class table."""
    global _nlp
    _nlp = nlp


def set_model_source(model_dir=None, model=None, tokenizer=None):
    """Where ``WhisperTranscriber.load_model`` finds its weights: a checkpoint directory (see ``Aligners/checkpoint.py``) or
    objects built by the caller (a ``checkpoint.WhisperModel`` and a ``tokenizer.WhisperTokenizer``)."""
    _MODEL_SOURCE.update(dir=model_dir, model=model, tokenizer=tokenizer)


def remove_spurious_commas(text: str) -> str:
    """Drop "," / "." / "[*]" right after a DET/ADP/CCONJ/SCONJ/PART/PRON token (:33-52)."""
    out = []
    for tok in _nlp(text):
        if tok.text in ({",", "."} | _PAUSE_MARKERS) and out and out[-1].pos_ in FORBIDDEN_POS:
            continue
        out.append(tok)
    return "".join(t.text_with_ws for t in out)


def check_audio_content_batch(paths, engine=None):
    """[(ok, message)] per file: too small (< 1000 bytes), > 95 % silence (|x| <= 500) or RMS < 100 (:197-229)."""
    eng = engine or get_default_engine()
    decoded, out = {}, [None] * len(paths)
    for i, p in enumerate(paths):
        try:
            decoded[i] = H.decode_wav(p)
        except H.CouldntDecodeError as e:
            out[i] = (True, f"Unable to check the audio: {e}")          # the reference lets undecodable files through
    for i, (ok, msg, _, _) in _gate(eng, decoded, paths).items():
        out[i] = (ok, msg)
    return out


def _gate(eng, decoded: Dict[int, tuple], paths) -> Dict[int, tuple]:
    """index -> (ok, message, rms, silence_ratio) for the decoded files: one ``k_energy`` launch per sample rate."""
    res = {}
    by_rate: Dict[int, List[int]] = {}
    for i, (rate, _) in decoded.items():
        by_rate.setdefault(rate, []).append(i)
    for rate, idxs in by_rate.items():
        eng.upload([decoded[i][1] for i in idxs], rate)
        for i, e in zip(idxs, eng.energy(eng.whole_clip_slices(), 500)):
            rms, ratio, _ = H.gate_from_counts(int(e["sum_sq"]), int(e["n_loud"]), int(e["n"]))
            size = os.path.getsize(paths[i])
            if size < 1000:
                res[i] = (False, f"File too small ({size} octets)", rms, ratio)
            elif ratio > 0.95:
                res[i] = (False, f"File mainly contains silence ({ratio:.2f})", rms, ratio)
            elif rms < 100:
                res[i] = (False, f"Very low audio level (RMS={rms})", rms, ratio)
            else:
                res[i] = (True, "Audio valide", rms, ratio)
    return res


class WhisperTranscriber:
    """Same surface as the reference class (:56-328): ``load_model``, ``process_audio``, ``_check_audio_content``,
    ``_is_empty_result``, ``_create_empty_result``, ``clean_text``, ``save_results``, ``display_results``; plus
    ``process_audio_batch`` which is what ``main`` uses."""

    def __init__(self, model_size: str = "medium", device: Optional[str] = None, language: str = "fr", logger: Optional[logging.Logger] = None,
                 engine: Optional[ProsodyEngine] = None):
        self.model_size = model_size
        self.device = device or "cuda"
        self.language = language
        self.model = None
        self.tokenizer = None
        self.logger = logger if logger else logging.getLogger(__name__)
        self._engine = engine
        self.logger.info(f"Initialisation with model={model_size}, device={self.device}")

    # ------------------------------------------------------------------ engine / model
    @property
    def engine(self) -> ProsodyEngine:
        if self._engine is None:
            dev = str(self.device)
            # "cuda:3" names the device; a bare "cuda" under a one-process-per-GPU launcher is this rank's own GPU
            self._engine = get_default_engine(int(dev.split(":")[1]) if ":" in dev else shard.local_device())
        return self._engine

    def load_model(self) -> None:
        try:
            self.logger.info(f"Loading the model  {self.model_size}")
            model = _MODEL_SOURCE["model"] or CK.load_model(self.model_size, _MODEL_SOURCE["dir"])
            tok = _MODEL_SOURCE["tokenizer"] or CK.load_tokenizer(_MODEL_SOURCE["dir"], self.language, model.text_dims["n_vocab"])
            CK.pad_vocab(None, model.text_dims, tok.n_vocab)
            tok.language = self.language
            model.load_into(self.engine)
            self.model, self.tokenizer = model, tok
            self.logger.info("Model successfully loaded")
        except Exception as e:
            self.logger.error(f"Error during model loading: {e}")
            self.logger.error(traceback.format_exc())
            raise

    # ------------------------------------------------------------------ one file / many files
    def process_audio(self, audio_path: str) -> Dict[str, Any]:
        res = self.process_audio_batch([audio_path])[0]
        if isinstance(res, Exception):
            raise res
        return res

    def transcription_config(self) -> Dict[str, Any]:
        return {"language": self.language, "vad": "auditok", "compute_word_confidence": True, "detect_disfluencies": True,
                "trust_whisper_timestamps": True}                                # :150-156

    def process_audio_batch(self, audio_paths: List[str]) -> List[Any]:
        """``process_audio`` (:113-195) for many files: per file a result dict, the "..." result for gated / nearly
        empty transcriptions, or the exception that file raised."""
        out: List[Any] = [None] * len(audio_paths)
        decoded: Dict[int, tuple] = {}
        for i, p in enumerate(audio_paths):
            try:
                if not Path(p).exists():
                    self.logger.error(f"Audio file not found: {p}")
                    raise FileNotFoundError(f"Audio file not found: {p}")
                self.logger.info(f"Processing File: {p}")
                decoded[i] = H.decode_wav(p)
            except Exception as e:                                           # noqa: BLE001 (per-file failures are results)
                out[i] = e
        gate = {}
        try:
            gate = _gate(self.engine, decoded, audio_paths)
        except Exception as e:                                               # noqa: BLE001
            self.logger.error(f"Error during audio check: {e}")             # the reference lets the file through (:227-229)
        todo = []
        for i in decoded:
            if i in gate and not gate[i][0]:
                self.logger.warning(f"The audio file seems problematic: {gate[i][1]}")
                out[i] = self._create_empty_result()
            else:
                todo.append(i)
        if todo and self.model is None:
            try:
                self.load_model()
            except Exception as e:                                           # noqa: BLE001
                for i in todo:                                               # in the reference every file then fails on its own (:146-147)
                    out[i] = e
                return out
        opts = TR.TranscribeOptions(**self.transcription_config())

        def run(idxs):
            clips = self._load_audio_16k([decoded[i] for i in idxs])
            self.logger.info("Début de la transcription")
            results = TR.transcribe_batch(self.engine, self.model, self.tokenizer, clips, opts, self.logger)
            self.logger.info("Transcription ended")
            for i, r in zip(idxs, results):
                if self._is_empty_result(r):
                    self.logger.warning(f"Very little content detected in the audio file: {audio_paths[i]}")
                    r = self._create_empty_result()
                out[i] = r

        for k in range(0, len(todo), BATCH_CLIPS):
            idxs = todo[k:k + BATCH_CLIPS]
            try:
                run(idxs)
            except Exception as e:                                           # noqa: BLE001
                # in the reference one file's failure never touches another file (:640-660): retry the batch file by file
                self.logger.error(f"Erreur lors du traitement: {e}")
                for i in idxs:
                    try:
                        run([i])
                    except Exception as e1:                                  # noqa: BLE001
                        self.logger.error(f"Erreur lors du traitement: {e1}")
                        self.logger.error(traceback.format_exc())
                        out[i] = e1
        return out

    def _load_audio_16k(self, decoded: List[tuple]) -> List[np.ndarray]:
        """``whisper.load_audio`` (:139): 16 kHz mono.  Files at other rates go through the engine's polyphase
        resampler, one launch per source rate."""
        clips: List[Optional[np.ndarray]] = [None] * len(decoded)
        by_rate: Dict[int, List[int]] = {}
        for i, (rate, pcm) in enumerate(decoded):
            by_rate.setdefault(rate, []).append(i)
        for rate, idxs in by_rate.items():
            if rate == TR.SAMPLE_RATE:
                for i in idxs:
                    clips[i] = decoded[i][1]
                continue
            self.engine.upload([decoded[i][1] for i in idxs], rate)
            self.engine.resample(TR.SAMPLE_RATE)
            for i, c in zip(idxs, self.engine.download()):
                clips[i] = c
        return clips

    # ------------------------------------------------------------------ the reference's helpers
    def _check_audio_content(self, audio_path):
        return check_audio_content_batch([audio_path], self.engine)[0]

    def _is_empty_result(self, result) -> bool:                              # :231-242
        if not result["segments"]:
            return True
        if sum(len(seg["words"]) for seg in result["segments"]) < 3:
            return True
        return len(" ".join(seg["text"] for seg in result["segments"]).strip()) < 10

    def _create_empty_result(self) -> Dict[str, Any]:                        # :244-261
        return {"text": "...",
                "segments": [{"id": 0, "start": 0.0, "end": 1.0, "text": "...",
                              "words": [{"start": 0.0, "end": 1.0, "text": "...", "confidence": 0.0}]}],
                "language": self.language}

    def clean_text(self, text: str) -> str:                                  # :263-295
        original = text
        text = re.sub(r"\s+", " ", text).strip()
        text = remove_spurious_commas(text)
        fw = r"\b(?:que|et|ou|mais|donc|car|ni|où|dont|à|de|du|au|aux|en|par|pour|avec|sans|sur|sous)\b"
        text = re.sub(rf"({fw})\s*[,\.]+", lambda m: m.group(1), text, flags=re.IGNORECASE)
        text = re.sub(rf"({fw})\s*\[\*\]\s*", lambda m: m.group(1), text, flags=re.IGNORECASE)
        text = text.replace(";", "")
        if original != text:
            self.logger.debug(f"Texte nettoyé: '{original}' -> '{text}'")
        return text

    def save_results(self, result: Dict[str, Any], output_path: str) -> None:    # :297-314 (cleans again, as the reference does)
        for segment in result["segments"]:
            segment["text"] = self.clean_text(segment["text"])
            for word in segment["words"]:
                word["text"] = self.clean_text(word["text"])
        with open(output_path, "w", encoding="utf-8") as f:
            json.dump(result, f, ensure_ascii=False, indent=2)
        self.logger.info(f"Results saved: {output_path}")

    def display_results(self, result: Dict[str, Any]) -> None:               # :316-328
        for segment in result["segments"]:
            print(f"\nSegment {segment['start']:.2f}s -> {segment['end']:.2f}s:")
            print(f"Texte: {segment['text']}")
            print("Mots détaillés:")
            for word in segment["words"]:
                print(f"  {word['start']:.2f}s -> {word['end']:.2f}s : {word['text']} (conf: {word['confidence']:.2f})")


def json_to_textgrid(json_file, logger=None) -> TextGrid:                    # :330-395
    logger = logger or logging.getLogger(__name__)
    try:
        if not os.path.exists(json_file):
            logger.error(f"Fichier JSON non trouvé: {json_file}")
            raise FileNotFoundError(f"Fichier JSON non trouvé: {json_file}")
        with open(json_file, "r", encoding="utf-8") as f:
            data = json.load(f)
        return words_to_textgrid(data)
    except Exception as e:
        logger.error(f"Erreur lors de la conversion JSON → TextGrid: {e}")
        raise


def _placeholder_grid(max_time: float = 1.0) -> TextGrid:
    """``IntervalTier(name="words", minTime=0, maxTime=max_time)`` with one "..." interval (:608-611, :687-691, :462-466)."""
    tier = IntervalTier("words", tier_min=0.0, tier_max=max_time)
    tier.add(0.0, max_time, "...")
    return TextGrid([tier], 0.0, 0.0)                           # textgrid.TextGrid(): no maxTime of its own, the writer takes the tier's


def create_matching_textgrids(natural_dir, synthetic_dir, logger=None):      # :425-498
    """Every TextGrid name present in one directory exists in the other: a missing one is written as a single "..."
    interval spanning the other side's ``maxTime`` (1.0 if that grid cannot be read)."""
    logger = logger or logging.getLogger(__name__)
    os.makedirs(natural_dir, exist_ok=True)
    os.makedirs(synthetic_dir, exist_ok=True)
    nat = {f for f in os.listdir(natural_dir) if f.endswith(".TextGrid")}
    syn = {f for f in os.listdir(synthetic_dir) if f.endswith(".TextGrid")}
    for missing, src_dir, dst_dir in ((nat - syn, natural_dir, synthetic_dir), (syn - nat, synthetic_dir, natural_dir)):
        for name in sorted(missing):
            try:
                max_time = read_textgrid(os.path.join(src_dir, name)).max_time
            except Exception as e:                                           # noqa: BLE001
                logger.warning(f"Erreur lors de la lecture du TextGrid {os.path.join(src_dir, name)}: {e}")
                max_time = 1.0
            write_textgrid(_placeholder_grid(max_time), os.path.join(dst_dir, name))
            logger.info(f"TextGrid vide créé: {os.path.join(dst_dir, name)}")


def word_timings(text_indices, time_indices, words, word_token_counts):
    """The tail of openai-whisper's ``find_alignment`` (timing.py): DTW path -> one (start, end) per word.

    ``word_token_counts``: tokens per word for ``text_tokens + [eot]`` as ``tokenizer.split_to_word_tokens`` groups
    them (the last group is the end-of-text token); ``words``: their texts, same length.  A token's time is the frame
    at which the path first reaches it, a word runs from its first token's time to the next word's first token's time.
    -> [{"text", "start", "end"}, ...] for every word but the final end-of-text group."""
    text_indices = np.asarray(text_indices); time_indices = np.asarray(time_indices)
    jumps = np.pad(np.diff(text_indices), (1, 0), constant_values=1).astype(bool)
    jump_times = time_indices[jumps] / TOKENS_PER_SECOND
    counts = list(word_token_counts)
    boundaries = np.pad(np.cumsum(counts[:-1]), (1, 0))
    starts, ends = jump_times[boundaries[:-1]], jump_times[boundaries[1:]]
    return [{"text": w, "start": float(a), "end": float(b)} for w, a, b in zip(words, starts, ends)]


def transcription_result(word_items, language="fr"):
    """Word timings -> the dict shape of ``whisper_timestamped.transcribe`` that ``json_to_textgrid`` and the rest of the
    pipeline read (:244-261, :330-395): one segment holding the words."""
    words = [{"text": w["text"], "start": round(w["start"], 2), "end": round(w["end"], 2), "confidence": w.get("confidence", 1.0)} for w in word_items]
    text = "".join(w["text"] for w in word_items).strip()
    seg = {"id": 0, "start": words[0]["start"] if words else 0.0, "end": words[-1]["end"] if words else 0.0, "text": text, "words": words}
    return {"text": text, "segments": [seg] if words else [], "language": language}


def main(audio_path, out_path, whisper_model="medium", device=None, logger=None):
    """Entry point (:501-728).  Files are processed as a batch; per-file outcomes and the files written are the
    reference's."""
    if logger is None:
        logger = logging.getLogger(__name__)
        if not logger.hasHandlers():
            logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
            logger.info("No logger provided, using basic configuration.")
    logger.debug(f"Démarrage du script avec audio_path={audio_path}, out_path={out_path}")
    try:
        if not os.path.exists(audio_path):
            logger.error(f"Le dossier audio n'existe pas: {audio_path}")
            sys.exit(1)
        transcriber = WhisperTranscriber(model_size=whisper_model, device="cuda" if device is None else device, language="fr", logger=logger)
        try:
            names = [n[:-4] for n in os.listdir(audio_path) if n.endswith(".wav")]
        except Exception as e:                                               # noqa: BLE001
            logger.error(f"Erreur lors de la lecture du dossier audio: {e}")
            sys.exit(1)
        N = len(names)
        logger.info(f"Nombre de fichiers .wav trouvés: {N}")
        if N == 0:
            logger.warning(f"Aucun fichier .wav trouvé dans {audio_path}")
            sys.exit(0)
        # one process per GPU (torch.distributed initialised by the launcher): files are independent, so rank r takes a contiguous
        # block of the SORTED names and writes those files' outputs; nothing is exchanged (no collective on this step)
        from .. import shard
        rank, world = shard.rank_world()
        if world > 1:
            names = sorted(names)
            lo, hi = shard.shard_range(N, rank, world)
            names = names[lo:hi]
            N = len(names)
            logger.info(f"rank {rank}/{world}: fichiers {lo}..{hi - 1}")
        textgrid_dir = out_path
        # the rank-local part: whatever happens in it, every rank leaves it through the same status barrier, so that every rank's
        # TextGrids exist before the folders are matched and a failure on one rank fails the call on all of them (sys.exit(1) below)
        with shard.agreed():
            OP = os.path.join(out_path + "_transcription")
            os.makedirs(OP, exist_ok=True)
            os.makedirs(textgrid_dir, exist_ok=True)

            # ---- the inline noise gate (:579-621) for all files in one pass
            paths = [os.path.join(audio_path, f"{n}.wav") for n in names]
            decoded = {}
            for i, p in enumerate(paths):
                try:
                    decoded[i] = H.decode_wav(p)
                except Exception as e:                                           # noqa: BLE001
                    logger.error(f"Erreur lors de l'analyse audio de {names[i]}: {e}")     # the reference continues normally
            try:
                gate = _gate(transcriber.engine, decoded, paths)
            except Exception as e:                                               # noqa: BLE001
                logger.error(f"Erreur lors de l'analyse audio: {e}")
                gate = {}
            processed, count, todo = [], 0, []
            for i, n in enumerate(names):
                g = gate.get(i)
                if g is not None and (g[3] > 0.95 or g[2] < 100):
                    logger.warning(f"Fichier {n} détecté comme bruit/silence (RMS={g[2]}, silence_ratio={g[3]:.2f})")
                    with open(os.path.join(OP, f"{n}.txt"), "w", encoding="utf-8") as f:
                        f.write("...")
                    write_textgrid(_placeholder_grid(1.0), os.path.join(textgrid_dir, f"{n}.TextGrid"))
                    processed.append(n); count += 1
                else:
                    todo.append(i)

            # ---- transcription of everything that passed, as resident batches
            results = transcriber.process_audio_batch([paths[i] for i in todo]) if todo else []
            for i, result in zip(todo, results):
                n = names[i]
                json_file, txt_file = os.path.join(OP, f"{n}.json"), os.path.join(OP, f"{n}.txt")
                try:
                    if isinstance(result, Exception):
                        raise result
                    raw_json_dir = Path(out_path + "_raw_json"); raw_json_dir.mkdir(parents=True, exist_ok=True)
                    raw_json_path = raw_json_dir / f"{n}.raw.json"
                    raw_json_path.write_text(json.dumps(result, ensure_ascii=False, indent=2), encoding="utf-8")
                    raw_tg_dir = Path(out_path + "_textgrid_raw"); raw_tg_dir.mkdir(parents=True, exist_ok=True)
                    write_textgrid(json_to_textgrid(str(raw_json_path), logger), str(raw_tg_dir / f"{n}.TextGrid"))
                    for segment in result["segments"]:
                        segment["text"] = transcriber.clean_text(segment["text"])
                        for word in segment["words"]:
                            word["text"] = transcriber.clean_text(word["text"])
                    transcriber.save_results(result, json_file)
                    clean = transcriber.clean_text(" ".join(seg["text"] for seg in result["segments"]))
                    with open(txt_file, "w", encoding="utf-8") as f:
                        f.write(clean)
                    write_textgrid(json_to_textgrid(json_file, logger), os.path.join(textgrid_dir, f"{n}.TextGrid"))
                    logger.info(f"TextGrid created: {os.path.join(textgrid_dir, f'{n}.TextGrid')}")
                    processed.append(n)
                except Exception as e:                                           # noqa: BLE001
                    logger.error(f"Error during file processing {n}: {e}")
                    logger.error("".join(traceback.format_exception(type(e), e, e.__traceback__)))
                    logger.warning("Moving to the next file...")
                    continue
                count += 1
                logger.info(f"Progression: {count}/{N} Files processed successfully")
            logger.info(f"Processing completed: {count}/{N} ")

            problematic = [n for n in names if n not in processed]
            if problematic:
                logger.warning(f"{len(problematic)}  Problematic files identified:")
                for n in problematic:
                    logger.warning(f"  - {n}.wav")
                    write_textgrid(_placeholder_grid(1.0), os.path.join(textgrid_dir, f"{n}.TextGrid"))
                    with open(os.path.join(OP, f"{n}.txt"), "w", encoding="utf-8") as f:
                        f.write("...")

        with shard.agreed(only_rank=0) as sec:                               # (ranks other than 0 leave when the folders are matched, not before)
            if sec.mine:
                base_path = os.path.dirname(audio_path)
                if "_microsoft" in base_path:
                    natural_dir = os.path.join(os.path.dirname(base_path), os.path.basename(base_path).replace("_microsoft", ""), "WhisperTS_textgrid_files")
                    if os.path.exists(natural_dir):
                        create_matching_textgrids(natural_dir, textgrid_dir, logger)
                else:
                    synthetic_dir = os.path.join(os.path.dirname(base_path), os.path.basename(base_path) + "_microsoft", "WhisperTS_textgrid_files")
                    if os.path.exists(synthetic_dir):
                        create_matching_textgrids(textgrid_dir, synthetic_dir, logger)
    except Exception as e:
        logger.error(f"Error in principal code: {e}")
        logger.error(traceback.format_exc())
        sys.exit(1)


def cli_main():
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s", handlers=[logging.StreamHandler(sys.stdout)])
    logger = logging.getLogger(__name__)
    if len(sys.argv) != 3:
        logger.error(f"Arguments incorrects: {sys.argv}")
        print("Usage: python use_whisper_timestamped.py <audio_path> <out_path>")
        sys.exit(1)
    main(sys.argv[1], sys.argv[2], logger=logger)
    logger.info("Script end of execution.")


if __name__ == "__main__":
    cli_main()
