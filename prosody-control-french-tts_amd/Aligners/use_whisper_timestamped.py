"""Noise / silence gate of the aligner (``Code/Aligners/use_whisper_timestamped.py:197-229`` and its
inline copy :581-599) on the GPU, plus the JSON -> TextGrid conversion of :330-395.

What runs on the engine: the gate, the 16 kHz resampler, log-mel, the audio encoder and the forced
alignment of *given* token ids (teacher-forced text decoder, alignment-head cross-attention, DTW:
``ProsodyEngine.whisper_align``).  Free-running transcription needs the trained checkpoint and the
tokenizer vocabulary, neither of which is available offline, so ``main()`` raises ``NotImplementedError``."""
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


def main(audio_path, out_path, whisper_model="medium", device=None, logger=None):
    raise NotImplementedError("free-running Whisper transcription (checkpoint + tokenizer) is not available in this build; the gate, "
                              "resampler, log-mel, encoder and token-level forced alignment (ProsodyEngine.whisper_align) are")
