"""Free-running Whisper decoding at temperature 0: the host side of ``whisper.decoding.DecodingTask`` with a
``GreedyDecoder`` (what ``whisper_timestamped.transcribe`` runs first, Code/Aligners/use_whisper_timestamped.py:150-163).

The array work of a step -- text decoder, output projection, the logit filters SuppressBlank / SuppressTokens /
ApplyTimestampRules and the arg-max -- is ``ProsodyEngine.whisper_decode_step`` (libpce.so).  Here: the prompt, the
vocabulary mask the filters read, the loop and its stopping rule.  Token ids in and out: ``Aligners/tokenizer.py`` turns ids into text given the checkpoint's
vocabulary file, ``Aligners/transcribe.py`` drives the windows of a recording (prompts, thresholds, word timings)."""
from __future__ import annotations

import numpy as np


def vocab_mask(n_vocab: int, suppress_tokens, blank_tokens, no_timestamps: int) -> np.ndarray:
    """uint8 [n_vocab]: bit 0 = suppressed at every step (SuppressTokens + <|notimestamps|>, which ApplyTimestampRules
    removes), bit 1 = suppressed at the first sampled position (SuppressBlank: the blank token and end-of-text)."""
    m = np.zeros(n_vocab, dtype=np.uint8)
    m[np.asarray(list(suppress_tokens), dtype=np.int64)] |= 1
    m[int(no_timestamps)] |= 1
    m[np.asarray(list(blank_tokens), dtype=np.int64)] |= 2
    return m


def greedy_decode(engine, n_vocab: int, initial_tokens, rules: dict, sample_len: int):
    """Decode every clip the engine has encoded (``whisper_encode_run``) -> list of token lists (prompt included, cut
    after the first end-of-text).  rules: eot, no_timestamps, timestamp_begin, suppress_tokens, blank_tokens,
    max_initial_timestamp_index.  As DecodingTask._main_loop: at most ``sample_len`` steps, stop when every sequence has
    produced end-of-text (finished sequences are padded with it meanwhile)."""
    n = engine.whisper_num_encoded()
    seqs = [list(initial_tokens) for _ in range(n)]
    begin = len(initial_tokens)
    mask = vocab_mask(n_vocab, rules["suppress_tokens"], rules["blank_tokens"], rules["no_timestamps"])
    sum_logprobs = np.zeros(n, dtype=np.float64)
    for _ in range(sample_len):
        nxt = engine.whisper_decode_step(seqs, begin, rules["eot"], rules["timestamp_begin"], mask, rules.get("max_initial_timestamp_index"))
        sum_logprobs += engine.last_decode_logprobs                 # (0 for sequences that had already ended: GreedyDecoder.update)
        for s, t in zip(seqs, nxt):
            s.append(int(t))
        if all(s[-1] == rules["eot"] for s in seqs):
            break
    out = []
    for s in seqs:
        new = s[begin:]
        cut = new.index(rules["eot"]) + 1 if rules["eot"] in new else len(new)
        out.append(s[:begin + cut])
    greedy_decode.last_sum_logprobs = sum_logprobs                  # avg_logprob of transcribe = sum / (tokens + 1)
    return out


def decode_batch(engine, n_vocab: int, prompts, sample_begins, rules: dict, sample_len: int, temperature: float = 0.0, seed: int = 0,
                 active=None, no_cache: bool = False, device_loop: bool = True, n_text_ctx: int = None):
    """``greedy_decode`` with one prompt per clip (``condition_on_previous_text`` gives every recording its own) and an
    optional temperature (GreedyDecoder at temperature t: one sample per step from softmax(filtered logits / t)).
    ``active[i]`` False: the clip is left alone (it reads as ended from the first step on).
    -> (sampled tokens per clip, end-of-text cut off; their log-probabilities per clip; sum_logprobs [clips]).
    ``device_loop`` (default): the step loop runs inside the engine (``pce_whisper_decode_loop``: prompts up once, results down
    once); False: one ``whisper_decode_step_ex`` round trip per step -- the form the device loop is checked against.
    A sequence that fills the text context (``n_text_ctx``; prompt of up to 227 tokens + 224 sampled ones can exceed 448) stops there and keeps
    what it has, as DecodingTask._main_loop breaks on ``tokens.shape[-1] > n_ctx``: the other sequences of the batch go on."""
    n = engine.whisper_num_encoded()
    eot = rules["eot"]
    active = [True] * n if active is None else list(active)
    seqs = [list(p) if a else list(p) + [eot] for p, a in zip(prompts, active)]
    begins = np.asarray([b if a else b for b, a in zip(sample_begins, active)], dtype=np.int32)
    mask = vocab_mask(n_vocab, rules["suppress_tokens"], rules["blank_tokens"], rules["no_timestamps"])
    lps = [[] for _ in range(n)]
    if device_loop and not no_cache and sample_len >= 1:            # (no_cache asks for the prefix re-run every step: only the host-driven form has it)
        toks, lp, _ = engine.whisper_decode_loop(seqs, begins, eot, rules["timestamp_begin"], mask, int(sample_len), rules.get("max_initial_timestamp_index"),
                                                 temperature=temperature, seed=seed, no_cache=no_cache)
        for step in range(toks.shape[1]):
            for i in range(n):
                ended = seqs[i][-1] == eot and len(seqs[i]) > begins[i]
                seqs[i].append(int(toks[i, step]))
                if not ended:
                    lps[i].append(float(lp[i, step]))
        sample_len = 0                                              # (the loop below has nothing left to do)
    for _ in range(sample_len):
        # (a sequence beyond the context is handed over as ended: its last entry reads end-of-text, exactly what the device loop's table holds)
        send = seqs if n_text_ctx is None else [s if len(s) <= n_text_ctx else s[:n_text_ctx - 1] + [eot] for s in seqs]
        nxt, lp, _ = engine.whisper_decode_step_ex(send, begins, eot, rules["timestamp_begin"], mask, rules.get("max_initial_timestamp_index"),
                                                   temperature=temperature, seed=seed, no_cache=no_cache)
        for i in range(n):
            ended = seqs[i][-1] == eot and len(seqs[i]) > begins[i]
            seqs[i].append(int(nxt[i]))
            if not ended:
                lps[i].append(float(lp[i]))
        if all(s[-1] == eot for s in seqs):
            break
    out_t, out_lp = [], []
    for i in range(n):
        new = seqs[i][int(begins[i]):] if active[i] else []
        cut = new.index(eot) if eot in new else len(new)
        out_t.append(new[:cut])
        out_lp.append(lps[i][:cut + (1 if eot in new else 0)] if active[i] else [])        # (the end-of-text token's own term is part of sum_logprobs)
    sums = np.array([float(np.sum(l)) if l else 0.0 for l in out_lp])
    return out_t, [l[:len(t)] for l, t in zip(out_lp, out_t)], sums


def no_speech_probs(engine, n_vocab: int, prompts, sot_index, rules: dict, no_speech_token: int):
    """``probs_at_sot[:, tokenizer.no_speech]`` of DecodingTask._main_loop: the decoder is causal, so the distribution at the
    <|startoftranscript|> position of a prompt equals the last position of the prefix that ends there."""
    mask = vocab_mask(n_vocab, rules["suppress_tokens"], rules["blank_tokens"], rules["no_timestamps"])
    prefixes = [list(p[:k + 1]) for p, k in zip(prompts, sot_index)]
    _, _, probe = engine.whisper_decode_step_ex(prefixes, [len(p) for p in prefixes], rules["eot"], rules["timestamp_begin"], mask,
                                                rules.get("max_initial_timestamp_index"), probe_token=int(no_speech_token))
    return np.asarray(probe, dtype=np.float64)


# ---------------------------------------------------------------------------------------------------------------
# Windows of a long recording: the seek logic of whisper.transcribe (openai-whisper transcribe.py, restated; the
# package is absent: parity unpinned, hand-made cases in tests/test_decoding_host.py)
# ---------------------------------------------------------------------------------------------------------------
N_FRAMES = 3000                 # mel frames per 30 s window
INPUT_STRIDE = 2                # mel frames per audio token
TIME_PRECISION = 0.02           # seconds per timestamp step


def segments_and_seek(tokens, timestamp_begin: int, seek: int, segment_size: int = N_FRAMES):
    """One decoded window (sampled tokens, end-of-text removed) -> (segments, new_seek).

    As whisper.transcribe: consecutive timestamp pairs cut the window into segments; when the window ends on a single
    timestamp the whole window is consumed, otherwise the seek moves to the last closed timestamp; a window without
    consecutive timestamps is one segment and is consumed whole.  segments: dicts with start / end (seconds, absolute)
    and the token slice."""
    tokens = [int(t) for t in tokens]
    time_offset = seek * 0.01
    is_ts = [t >= timestamp_begin for t in tokens]
    single_timestamp_ending = is_ts[-2:] == [False, True]
    consecutive = [i + 1 for i in range(len(tokens) - 1) if is_ts[i] and is_ts[i + 1]]
    segments = []
    if consecutive:
        slices = list(consecutive)
        if single_timestamp_ending:
            slices.append(len(tokens))
        last = 0
        for cur in slices:
            sl = tokens[last:cur]
            segments.append({"start": time_offset + (sl[0] - timestamp_begin) * TIME_PRECISION,
                             "end": time_offset + (sl[-1] - timestamp_begin) * TIME_PRECISION, "tokens": sl})
            last = cur
        if single_timestamp_ending:
            seek += segment_size
        else:
            seek += (tokens[last - 1] - timestamp_begin) * INPUT_STRIDE
    else:
        duration = segment_size * 0.01
        stamps = [t for t in tokens if t >= timestamp_begin]
        if stamps and stamps[-1] != timestamp_begin:
            duration = (stamps[-1] - timestamp_begin) * TIME_PRECISION
        segments.append({"start": time_offset, "end": time_offset + duration, "tokens": tokens})
        seek += segment_size
    return segments, seek


def transcribe_tokens(engine, n_mels: int, n_vocab: int, initial_tokens, rules: dict, sample_len: int, max_windows: int = 64):
    """Greedy transcription of every clip of the engine's resident 16 kHz batch, window by window:
    log-mel window at each clip's seek position (``logmel_run_at``), encoder, ``greedy_decode``, ``segments_and_seek``.
    -> per clip the list of segments (token ids; absolute times).  The encoder and decoder weights must be loaded.
    Every window starts from the same ``initial_tokens`` (no ``condition_on_previous_text``, no temperature fallback): the
    token-level building block; ``Aligners.transcribe.transcribe_batch`` is the restatement of whisper.transcribe with those.
    Warns (``RuntimeWarning``) when ``max_windows`` is exhausted with audio left: never a silently truncated transcript."""
    lens = [int(n) for n in engine.clip_lengths]
    content = [n // 160 for n in lens]
    seeks = [0] * len(lens)
    out = [[] for _ in lens]
    for _ in range(max_windows):
        active = [i for i in range(len(lens)) if seeks[i] < content[i]]
        if not active:
            break
        engine.logmel_run_at(n_mels, [min(seeks[i], content[i]) for i in range(len(lens))])
        engine.whisper_encode_run()
        seqs = greedy_decode(engine, n_vocab, initial_tokens, rules, sample_len)
        for i in active:
            new = [t for t in seqs[i][len(initial_tokens):] if t != rules["eot"]]
            seg_size = min(N_FRAMES, content[i] - seeks[i])
            if not new:
                seeks[i] += seg_size
                continue
            segs, seeks[i] = segments_and_seek(new, rules["timestamp_begin"], seeks[i], seg_size)
            out[i].extend(segs)
    left = [i for i in range(len(lens)) if seeks[i] < content[i]]
    if left:
        import warnings
        warnings.warn(f"transcribe_tokens: stopped after {max_windows} windows with audio left in clips {left[:8]} (raise max_windows)", RuntimeWarning)
    return out
