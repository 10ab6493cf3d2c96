"""Noise / silence gate of the aligner (``Code/Aligners/use_whisper_timestamped.py:197-229`` and its
inline copy :581-599) on the GPU, plus the JSON -> TextGrid conversion of :330-395.

What runs on the engine: the gate, the 16 kHz resampler, log-mel, the audio encoder and the forced
alignment of *given* token ids (teacher-forced text decoder, alignment-head cross-attention, DTW:
``ProsodyEngine.whisper_align``), free-running greedy decoding at token level (``Aligners/decoding.py``).  Turning
audio files into TextGrids end to end additionally needs the trained checkpoint and the tiktoken vocabulary, neither of
which is available offline, so ``main()`` raises ``NotImplementedError``."""
import os

from .. import hostrules as H
from ..engine import get_default_engine
from ..textgrid_io import words_to_textgrid, write_textgrid  # noqa: F401  (re-exported)


def check_audio_content_batch(paths, engine=None):
    """[(ok, message)] per file: too small (< 1000 bytes), > 95 % silence (|x| <= 500) or RMS < 100."""
    eng = engine or get_default_engine()
    decoded, out = {}, [None] * len(paths)
    for i, p in enumerate(paths):
        try:
            decoded[i] = H.decode_wav(p)
        except H.CouldntDecodeError as e:
            out[i] = (True, f"Unable to check the audio: {e}")          # the reference lets undecodable files through
    by_rate = {}
    for i, (rate, _) in decoded.items():
        by_rate.setdefault(rate, []).append(i)
    for rate, idxs in by_rate.items():
        eng.upload([decoded[i][1] for i in idxs], rate)
        for i, e in zip(idxs, eng.energy(eng.whole_clip_slices(), 500)):
            rms, ratio, _ = H.gate_from_counts(int(e["sum_sq"]), int(e["n_loud"]), int(e["n"]))
            size = os.path.getsize(paths[i])
            if size < 1000:
                out[i] = (False, f"File too small ({size} octets)")
            elif ratio > 0.95:
                out[i] = (False, f"File mainly contains silence ({ratio:.2f})")
            elif rms < 100:
                out[i] = (False, f"Very low audio level (RMS={rms})")
            else:
                out[i] = (True, "Audio valide")
    return out


class WhisperTranscriber:
    def __init__(self, model_size="medium", device=None, language="fr", logger=None):
        self.model_size, self.device, self.language, self.logger = model_size, device or "cuda", language, logger

    def _check_audio_content(self, audio_path):
        return check_audio_content_batch([audio_path])[0]


def json_to_textgrid(json_file, logger=None):
    import json
    if not os.path.exists(json_file):
        raise FileNotFoundError(f"Fichier JSON non trouvé: {json_file}")
    with open(json_file, "r", encoding="utf-8") as f:
        return words_to_textgrid(json.load(f))


TOKENS_PER_SECOND = 50          # openai-whisper audio.py: 20 ms per audio token


def word_timings(text_indices, time_indices, words, word_token_counts):
    """The tail of openai-whisper's ``find_alignment`` (timing.py): DTW path -> one (start, end) per word.

    ``word_token_counts``: tokens per word for ``text_tokens + [eot]`` as ``tokenizer.split_to_word_tokens`` groups
    them (the last group is the end-of-text token); ``words``: their texts, same length.  A token's time is the frame
    at which the path first reaches it, a word runs from its first token's time to the next word's first token's time.
    -> [{"text", "start", "end"}, ...] for every word but the final end-of-text group."""
    import numpy as np
    text_indices = np.asarray(text_indices); time_indices = np.asarray(time_indices)
    jumps = np.pad(np.diff(text_indices), (1, 0), constant_values=1).astype(bool)
    jump_times = time_indices[jumps] / TOKENS_PER_SECOND
    counts = list(word_token_counts)
    boundaries = np.pad(np.cumsum(counts[:-1]), (1, 0))
    starts, ends = jump_times[boundaries[:-1]], jump_times[boundaries[1:]]
    return [{"text": w, "start": float(a), "end": float(b)} for w, a, b in zip(words, starts, ends)]


def transcription_result(word_items, language="fr"):
    """Word timings -> the dict shape of ``whisper_timestamped.transcribe`` that ``json_to_textgrid`` and the rest of the
    pipeline read (Code/Aligners/use_whisper_timestamped.py:244-261, 330-395): one segment holding the words."""
    words = [{"text": w["text"], "start": round(w["start"], 2), "end": round(w["end"], 2), "confidence": w.get("confidence", 1.0)} for w in word_items]
    text = "".join(w["text"] for w in word_items).strip()
    seg = {"id": 0, "start": words[0]["start"] if words else 0.0, "end": words[-1]["end"] if words else 0.0, "text": text, "words": words}
    return {"text": text, "segments": [seg] if words else [], "language": language}


def main(audio_path, out_path, whisper_model="medium", device=None, logger=None):
    raise NotImplementedError("free-running Whisper transcription (checkpoint + tokenizer) is not available in this build; the gate, "
                              "resampler, log-mel, encoder and token-level forced alignment (ProsodyEngine.whisper_align) are")
