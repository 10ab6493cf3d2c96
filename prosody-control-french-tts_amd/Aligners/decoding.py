"""Free-running Whisper decoding at temperature 0: the host side of ``whisper.decoding.DecodingTask`` with a
``GreedyDecoder`` (what ``whisper_timestamped.transcribe`` runs first, Code/Aligners/use_whisper_timestamped.py:150-163).

The array work of a step -- text decoder, output projection, the logit filters SuppressBlank / SuppressTokens /
ApplyTimestampRules and the arg-max -- is ``ProsodyEngine.whisper_decode_step`` (libpce.so).  Here: the prompt, the
vocabulary mask the filters read, the loop and its stopping rule.  Token ids in and out: turning text into ids and back
needs the checkpoint's tiktoken vocabulary, which is not reachable offline."""
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
