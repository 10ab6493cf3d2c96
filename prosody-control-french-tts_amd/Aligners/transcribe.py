"""``whisper_timestamped.transcribe(model, audio, language="fr", vad="auditok", compute_word_confidence=True,
detect_disfluencies=True, trust_whisper_timestamps=True)`` (Code/Aligners/use_whisper_timestamped.py:150-170) for a
BATCH of recordings on the engine.

The reference transcribes one file at a time; here every file of a directory is one clip of a resident batch and each
stage is one set of launches over all of them:

    energy VAD (k_frame_energy) -> speech-only audio -> log-mel window at each clip's seek (k_logmel_*)
    -> audio encoder -> greedy decoding step by step (pce_whisper_decode_step_ex) -> whisper.transcribe's segment / seek
    rules -> forced alignment of the window's text tokens (teacher-forced decoder, cross-attention, DTW:
    pce_whisper_align_run) -> word timings, confidences -> times mapped back through the VAD cuts.

What is restated, from the published sources of packages that are absent here (**parity unpinned**, hand-made cases
in tests/test_aligner_host.py, GPU flow in tests/test_gpu_aligner.py):
  * openai-whisper==20240930 ``transcribe.py`` (window loop, prompts with ``condition_on_previous_text``, no-speech /
    log-probability / compression-ratio thresholds, the temperature ladder) and ``timing.py`` (``find_alignment``,
    ``merge_punctuations``, the distribution of words over segments in ``add_word_timestamps``);
  * whisper-timestamped==1.15.8: its defaults (temperature 0.0 only, no ``best_of`` / beam: the "efficient" settings the
    reference's call leaves in place), the speech-only gluing of the VAD and the mapping of times back, word confidence
    = exp(mean log-probability of the word's tokens), times rounded to 10 ms, words clamped into their segment when
    ``trust_whisper_timestamps``.  NOT restated: its own variant of the attention post-processing (what runs is
    openai-whisper's recipe on the same cross-attention logits) and the ``[*]`` disfluency marks (``detect_disfluencies``
    is accepted WITH A WARNING on every call and produces none; everything downstream handles the mark: json_to_textgrid,
    clean_text).
"""
from __future__ import annotations

import logging
import zlib
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import decoding as DEC
from . import vad as VAD

SAMPLE_RATE = 16000
HOP = 160
N_FRAMES = DEC.N_FRAMES
TOKENS_PER_SECOND = 50
PREPEND_PUNCTUATIONS = "\"'“¿([{-"
APPEND_PUNCTUATIONS = "\"'.。,，!！?？:：”)]}、"


@dataclass
class TranscribeOptions:
    language: str = "fr"
    vad: Optional[str] = "auditok"
    compute_word_confidence: bool = True
    detect_disfluencies: bool = True
    trust_whisper_timestamps: bool = True
    condition_on_previous_text: bool = True
    temperature: Tuple[float, ...] = (0.0,)
    compression_ratio_threshold: Optional[float] = 2.4
    logprob_threshold: Optional[float] = -1.0
    no_speech_threshold: Optional[float] = 0.6
    sample_len: Optional[int] = None
    max_windows: int = 400
    seed: int = 0
    min_word_duration: float = 0.02        # whisper_timestamped.transcribe's parameter of that name (0.02 s since its 1.11), see ensure_increasing_positions


def ensure_increasing_positions(items: List[dict], min_duration: float = 0.0) -> None:
    """Word (or segment) times in increasing order, every item at least ``min_duration`` long: whisper-timestamped's function of that
    name, applied to the words with ``min_word_duration`` (third party, absent: restated from its published behaviour, parity
    unpinned).  An item that starts before its predecessor ends is moved to the middle of the overlap (the predecessor is shortened) unless
    that would leave the predecessor shorter than ``min_duration``, in which case it starts where the predecessor ends; an item not longer
    than ``min_duration`` is stretched to it.  A DTW path that gives several tokens the same frame -- zero-length words the TextGrid writer
    of the reference rejects (``textgrid.IntervalTier.add`` raises on overlap, use_whisper_timestamped.py:368-374) -- comes out as a run
    of ``min_duration`` words."""
    for _ in range(len(items) + 1):
        modified_backward = False
        previous_end = 0.0
        for i, it in enumerate(items):
            if it["start"] < previous_end and i > 0:
                new_start = round((previous_end + it["start"]) / 2, 2)
                if new_start < items[i - 1]["start"] + min_duration:
                    new_start = previous_end
                else:
                    items[i - 1]["end"] = new_start
                    modified_backward = True
                it["start"] = new_start
            if it["end"] <= it["start"] + min_duration:
                it["end"] = it["start"] + min_duration
            previous_end = it["end"]
        if not modified_backward:
            return


def compression_ratio(text: str) -> float:
    b = text.encode("utf-8")
    return len(b) / len(zlib.compress(b))


# ---------------------------------------------------------------------------------------------------------------
# VAD: keep the speech, remember where it came from
# ---------------------------------------------------------------------------------------------------------------
@dataclass
class SpeechCuts:
    """Speech regions of one recording in samples; the transcribed audio is their concatenation."""
    segments: List[Tuple[int, int]]
    n_original: int

    @property
    def glued_length(self) -> int:
        return sum(b - a for a, b in self.segments)

    def glue(self, pcm: np.ndarray) -> np.ndarray:
        return np.concatenate([pcm[a:b] for a, b in self.segments]) if self.segments else pcm[:0]

    def to_original(self, t: float, is_end: bool = False) -> float:
        """Seconds on the glued timeline -> seconds in the recording.  A time on a cut belongs to the region before
        it when it ends something and to the region after it when it starts something."""
        x = t * SAMPLE_RATE
        acc = 0
        for k, (a, b) in enumerate(self.segments):
            n = b - a
            last = k == len(self.segments) - 1
            if x < acc + n or (is_end and x <= acc + n) or last:
                return (a + min(max(x - acc, 0), n)) / SAMPLE_RATE
            acc += n
        return t


def vad_cuts(engine, clips: Sequence[np.ndarray], logger=None) -> List[Optional[SpeechCuts]]:
    """``get_vad_segments(audio, method="auditok")`` + ``remove_non_speech`` for every clip of the batch (one
    ``k_frame_energy`` launch).  A clip too short for auditok's ``max_silence`` rule raises ValueError inside
    whisper-timestamped; the reference catches it and transcribes without VAD (use_whisper_timestamped.py:164-170):
    such clips come back as None."""
    block = int(VAD.ANALYSIS_WINDOW * SAMPLE_RATE)
    engine.upload(list(clips), SAMPLE_RATE)
    engine.frame_energy_run(block, block, requantize=True)
    out: List[Optional[SpeechCuts]] = []
    for i, c in enumerate(clips):
        n = len(c)
        if n == 0:
            out.append(SpeechCuts([], 0)); continue
        ss, cnt = engine.frame_energy_fetch(i)
        try:
            segs = VAD.vad_segments_from_energy(ss, cnt, n, SAMPLE_RATE, output_sample=True)
        except ValueError as e:
            if "max_silence" in str(e):
                if logger:
                    logger.warning("Auditok VAD failed (short audio); retrying transcription without VAD splitting")
                out.append(None); continue
            raise
        out.append(SpeechCuts([(int(s["start"]), int(s["end"])) for s in segs if s["end"] > s["start"]], n))
    return out


# ---------------------------------------------------------------------------------------------------------------
# words
# ---------------------------------------------------------------------------------------------------------------
def merge_punctuations(words: List[dict], prepended: str = PREPEND_PUNCTUATIONS, appended: str = APPEND_PUNCTUATIONS):
    """timing.py ``merge_punctuations``: opening marks join the following word, closing marks the preceding one
    (tokens move with them; the emptied entries stay in the list with no text)."""
    i, j = len(words) - 2, len(words) - 1
    while i >= 0:
        prev, nxt = words[i], words[j]
        if prev["word"].startswith(" ") and prev["word"].strip() in prepended:
            nxt["word"] = prev["word"] + nxt["word"]; nxt["tokens"] = prev["tokens"] + nxt["tokens"]
            nxt["logprobs"] = prev["logprobs"] + nxt["logprobs"]
            prev["word"] = ""; prev["tokens"] = []; prev["logprobs"] = []
        else:
            j = i
        i -= 1
    i, j = 0, 1
    while j < len(words):
        prev, nxt = words[i], words[j]
        if not prev["word"].endswith(" ") and nxt["word"] in appended:
            prev["word"] = prev["word"] + nxt["word"]; prev["tokens"] = prev["tokens"] + nxt["tokens"]
            prev["logprobs"] = prev["logprobs"] + nxt["logprobs"]
            nxt["word"] = ""; nxt["tokens"] = []; nxt["logprobs"] = []
        else:
            i = j
        j += 1


def words_from_path(tokenizer, text_tokens: List[int], token_logprobs: List[float], text_indices, time_indices) -> List[dict]:
    """The tail of timing.py ``find_alignment``: DTW path over (text tokens + end-of-text) x frames -> words with
    start / end on the window's own timeline."""
    words, word_tokens = tokenizer.split_to_word_tokens(list(text_tokens) + [tokenizer.eot])
    if len(word_tokens) <= 1:
        return []
    text_indices, time_indices = np.asarray(text_indices), np.asarray(time_indices)
    boundaries = np.pad(np.cumsum([len(t) for t in word_tokens[:-1]]), (1, 0))
    jumps = np.pad(np.diff(text_indices), (1, 0), constant_values=1).astype(bool)
    jump_times = time_indices[jumps] / TOKENS_PER_SECOND
    starts, ends = jump_times[boundaries[:-1]], jump_times[boundaries[1:]]
    lp = list(token_logprobs) + [0.0] * (len(text_tokens) - len(token_logprobs))
    out = []
    for w, toks, a, b, i0, i1 in zip(words, word_tokens, starts, ends, boundaries[:-1], boundaries[1:]):
        out.append({"word": w, "tokens": list(toks), "start": float(a), "end": float(b), "logprobs": lp[int(i0):int(i1)]})
    return out


def _confidence(logprobs: Sequence[float]) -> float:
    return round(float(np.exp(np.mean(logprobs))), 3) if len(logprobs) else 0.0


# ---------------------------------------------------------------------------------------------------------------
# the window loop
# ---------------------------------------------------------------------------------------------------------------
def transcribe_batch(engine, model, tokenizer, clips: Sequence[np.ndarray], options: Optional[TranscribeOptions] = None, logger=None) -> List[dict]:
    """Transcribe every clip (int16, 16 kHz mono) -> one ``whisper_timestamped`` result dict per clip:
    ``{"text", "segments": [{"id", "seek", "start", "end", "text", "tokens", "temperature", "avg_logprob",
    "compression_ratio", "no_speech_prob", "confidence", "words": [{"text", "start", "end", "confidence"}]}], "language"}``.
    ``model``: ``checkpoint.WhisperModel`` already loaded into ``engine``."""
    opt = options or TranscribeOptions()
    log = logger or logging.getLogger(__name__)
    n = len(clips)
    clips = [np.ascontiguousarray(c, dtype=np.int16).reshape(-1) for c in clips]
    if opt.detect_disfluencies:
        # the one option of the reference's call (use_whisper_timestamped.py:154) this engine cannot honour: say so, loudly, every call
        log.warning('detect_disfluencies=True: this engine produces no "[*]" disfluency marks (whisper-timestamped\'s rule is not restated: '
                    "its source is absent); a hesitation the model did not transcribe stays inside the neighbouring words' intervals "
                    "instead of becoming a gap interval in the TextGrid")
    # ---- 1. VAD
    cuts: List[Optional[SpeechCuts]] = [None] * n
    if opt.vad:
        if opt.vad != "auditok":
            raise ValueError(f'vad="{opt.vad}": only the energy VAD the reference selects ("auditok") is built')
        cuts = vad_cuts(engine, clips, log)
    audio = [c if k is None else k.glue(c) for c, k in zip(clips, cuts)]
    results = [{"text": "", "segments": [], "language": opt.language} for _ in range(n)]
    if n == 0:
        return results
    # ---- 2. windows
    engine.upload([a if len(a) else np.zeros(1, np.int16) for a in audio], SAMPLE_RATE)
    n_vocab = model.text_dims["n_vocab"]
    n_text_ctx = model.text_dims["n_text_ctx"]
    sample_len = opt.sample_len or n_text_ctx // 2
    rules = tokenizer.decoding_rules()
    sot_seq = list(tokenizer.sot_sequence(opt.language))
    content = [len(a) // HOP for a in audio]                                # mel frames of content per clip
    seeks = [0] * n
    clip_ids = [zlib.crc32(np.ascontiguousarray(a, dtype=np.int16).tobytes()) for a in audio]      # (what names a recording: its samples)
    all_tokens: List[List[int]] = [[] for _ in range(n)]                   # text of the previous windows (prompt material)
    prompt_reset = [0] * n
    n_mels = model.dims["n_mels"]
    for _ in range(opt.max_windows):
        active = [seeks[i] < content[i] for i in range(n)]
        if not any(active):
            break
        engine.logmel_run_at(n_mels, [min(seeks[i], content[i]) for i in range(n)])
        engine.whisper_encode_run()
        # prompts: <|startofprev|> + the tail of the previous text + the sot sequence (DecodingTask._get_initial_tokens)
        prompts, sot_index = [], []
        for i in range(n):
            prev = all_tokens[i][prompt_reset[i]:] if (opt.condition_on_previous_text and active[i]) else []
            if prev:
                prev = prev[-(n_text_ctx // 2 - 1):]
                prompts.append([tokenizer.sot_prev] + prev + sot_seq); sot_index.append(1 + len(prev))
            else:
                prompts.append(list(sot_seq)); sot_index.append(0)
        begins = [len(p) for p in prompts]
        nsp = (DEC.no_speech_probs(engine, n_vocab, prompts, sot_index, rules, tokenizer.no_speech)
               if opt.no_speech_threshold is not None else np.zeros(n))
        # decode_with_fallback: walk the temperature ladder for the clips whose result fails a threshold
        final = [None] * n
        todo = list(active)
        # the noise of a sampled token is keyed by (seed, clip key, position, token): the key names the recording and its window, so a
        # clip draws the same tokens whichever clips share its batch (another shard of the run, another batch size)
        engine.whisper_sample_keys([(clip_ids[i] ^ (2654435761 * (seeks[i] + 1))) & 0x7FFFFFFF for i in range(n)])
        for t in opt.temperature:
            toks, lps, sums = DEC.decode_batch(engine, n_vocab, prompts, begins, rules, sample_len, temperature=float(t),
                                               seed=opt.seed + int(round(t * 1000)), active=todo, n_text_ctx=n_text_ctx)
            for i in range(n):
                if not todo[i]:
                    continue
                text = tokenizer.decode(toks[i]).strip()
                avg_lp = float(sums[i]) / (len(toks[i]) + 1)
                cr = compression_ratio(text) if text else 0.0
                final[i] = {"tokens": toks[i], "logprobs": lps[i], "avg_logprob": avg_lp, "compression_ratio": cr, "temperature": float(t),
                            "no_speech_prob": float(nsp[i])}
                needs = False
                if opt.compression_ratio_threshold is not None and cr > opt.compression_ratio_threshold:
                    needs = True
                if opt.logprob_threshold is not None and avg_lp < opt.logprob_threshold:
                    needs = True
                if opt.no_speech_threshold is not None and nsp[i] > opt.no_speech_threshold:
                    needs = False                                           # silence: a higher temperature will not help
                todo[i] = needs
            if not any(todo):
                break
        # segments and seek per clip; collect what has to be aligned in this window
        win_segments: List[List[dict]] = [[] for _ in range(n)]
        align_tokens: List[List[int]] = []
        align_frames: List[int] = []
        for i in range(n):
            seg_size = min(N_FRAMES, max(content[i] - seeks[i], 0))
            minimal = sot_seq + [tokenizer.no_timestamps, tokenizer.eot]
            if not active[i]:
                align_tokens.append(minimal); align_frames.append(64); continue
            r = final[i]
            skip = False
            if opt.no_speech_threshold is not None:
                skip = r["no_speech_prob"] > opt.no_speech_threshold
                if opt.logprob_threshold is not None and r["avg_logprob"] > opt.logprob_threshold:
                    skip = False
            new = list(r["tokens"])
            if skip or not new:
                seeks[i] += seg_size
                align_tokens.append(minimal); align_frames.append(64); continue
            segs, new_seek = DEC.segments_and_seek(new, tokenizer.timestamp_begin, seeks[i], seg_size)
            # log-probabilities travel with their tokens (the slices of segments_and_seek are contiguous and in order)
            pos = 0
            for sg in segs:
                k = len(sg["tokens"])
                sg["_lp"] = r["logprobs"][pos:pos + k]; pos += k
                sg.update(seek=seeks[i], temperature=r["temperature"], avg_logprob=r["avg_logprob"], compression_ratio=r["compression_ratio"],
                          no_speech_prob=r["no_speech_prob"])
            text_tokens = [t for sg in segs for t in sg["tokens"] if t < tokenizer.eot]
            text_lps = [l for sg in segs for t, l in zip(sg["tokens"], sg["_lp"]) if t < tokenizer.eot]
            win_segments[i] = segs
            for sg in segs:
                sg["_window"] = (seeks[i], seg_size)
            if text_tokens:
                align_tokens.append(sot_seq + [tokenizer.no_timestamps] + text_tokens + [tokenizer.eot]); align_frames.append(max(seg_size, 2))
                win_segments[i][0]["_align"] = (text_tokens, text_lps)
            else:
                align_tokens.append(minimal); align_frames.append(64)
            all_tokens[i].extend(t for sg in segs for t in sg["tokens"])
            if not opt.condition_on_previous_text or r["temperature"] > 0.5:
                prompt_reset[i] = len(all_tokens[i])                         # do not feed the prompt tokens if a high temperature was used
            seeks[i] = new_seek
        # ---- 3. forced alignment of every clip's window text in one pass
        if any("_align" in s[0] for s in win_segments if s):
            paths = engine.whisper_align(align_tokens, align_frames, len(sot_seq), head_mask=model.alignment_heads)
            for i in range(n):
                segs = win_segments[i]
                if not segs or "_align" not in segs[0]:
                    continue
                text_tokens, text_lps = segs[0].pop("_align")
                words = words_from_path(tokenizer, text_tokens, text_lps, paths[i]["text_indices"], paths[i]["time_indices"])
                merge_punctuations(words)
                offset = segs[0]["_window"][0] * HOP / SAMPLE_RATE
                wi = 0
                for sg in segs:
                    saved = 0
                    sg["words"] = []
                    want = sum(1 for t in sg["tokens"] if t < tokenizer.eot)
                    while wi < len(words) and saved < want:
                        w = words[wi]
                        if w["word"]:
                            sg["words"].append({"text": w["word"].strip(), "start": offset + w["start"], "end": offset + w["end"],
                                                "confidence": _confidence(w["logprobs"]) if opt.compute_word_confidence else 1.0})
                        saved += len(w["tokens"]); wi += 1
        for i in range(n):
            for sg in win_segments[i]:
                sg.setdefault("words", [])
                results[i]["segments"].append(sg)
    else:
        if any(seeks[i] < content[i] for i in range(n)):
            log.warning("transcription stopped after %d windows with audio left (max_windows)", opt.max_windows)
    # ---- 4. finish: clamp, map back through the VAD cuts, round, texts
    for i in range(n):
        for sg in results[i]["segments"]:
            for w in sg["words"]:
                if opt.trust_whisper_timestamps:
                    w["start"] = min(max(w["start"], sg["start"]), sg["end"]); w["end"] = min(max(w["end"], w["start"]), sg["end"])
        # back through the VAD cuts first (a cut maps everything behind the last speech region onto its end), THEN the minimum duration: the
        # times the TextGrid writer sees are the ones that must not collide
        k = cuts[i]
        conv = (lambda t, e=False: k.to_original(t, e)) if k is not None else (lambda t, e=False: t)
        for sg in results[i]["segments"]:
            for w in sg["words"]:
                w["start"], w["end"] = conv(w["start"]), conv(w["end"], True)
        if opt.min_word_duration is not None:
            ensure_increasing_positions([w for sg in results[i]["segments"] for w in sg["words"]],
                                        opt.min_word_duration if opt.trust_whisper_timestamps else 0.0)
        out_segments = []
        for sid, sg in enumerate(results[i]["segments"]):
            sg.pop("_window", None)
            lps = sg.pop("_lp", [])
            toks = sg["tokens"]
            text = tokenizer.decode(toks)
            if not text.strip() and not sg["words"]:
                continue
            a, b = conv(sg["start"]), conv(sg["end"], True)
            if sg["words"]:
                if not opt.trust_whisper_timestamps:
                    a, b = sg["words"][0]["start"], sg["words"][-1]["end"]
                a, b = min(a, sg["words"][0]["start"]), max(b, sg["words"][-1]["end"])      # (words stretched to the minimum duration stay inside their segment)
            seg = {"id": len(out_segments), "seek": sg["seek"], "start": round(a, 2), "end": round(b, 2), "text": text,
                   "tokens": list(toks), "temperature": sg["temperature"], "avg_logprob": sg["avg_logprob"],
                   "compression_ratio": sg["compression_ratio"], "no_speech_prob": sg["no_speech_prob"],
                   "confidence": _confidence([l for t, l in zip(toks, lps) if t < tokenizer.eot]) if opt.compute_word_confidence else 1.0,
                   "words": [{"text": w["text"], "start": round(w["start"], 2), "end": round(w["end"], 2), "confidence": w["confidence"]}
                             for w in sg["words"]]}
            out_segments.append(seg)
        results[i]["segments"] = out_segments
        results[i]["text"] = "".join(s["text"] for s in out_segments)
    return results
