"""Whisper's byte-level BPE vocabulary on the host: ids -> text, word grouping, the special-token table.

The reference gets its tokenizer from the ``whisper`` package behind ``whisper_timestamped``
(Code/Aligners/use_whisper_timestamped.py:7,97: ``whisper.load_model`` -> ``whisper.tokenizer.get_tokenizer``), which
reads ``assets/multilingual.tiktoken`` (one ``base64(token bytes) rank`` pair per line) and appends the special tokens
in a fixed order (openai-whisper==20240930 ``tokenizer.py:get_encoding``).  That file cannot be fetched here; this
module reads the same format from a path the caller names (``PCE_WHISPER_DIR`` / the ``whisper_vocab`` config key) and
the tests build a miniature rank table in memory.  Third-party behaviour restated from the published source:
**parity unpinned**, hand-made cases in ``tests/test_aligner_host.py``.

No array math: the engine works on token ids; everything here is string / integer bookkeeping per decoded window.
"""
from __future__ import annotations

import base64
import string
from typing import Dict, Iterable, List, Sequence, Tuple

# openai-whisper tokenizer.py LANGUAGES, in its order (the language tokens follow <|startoftranscript|> in this order)
LANGUAGES = ("en zh de es ru ko fr ja pt tr pl ca nl ar sv it id hi fi vi he uk el ms cs ro da hu ta no th ur hr bg lt la mi ml cy sk te fa lv "
             "bn sr az sl kn et mk br eu is hy ne mn bs kk sq sw gl mr pa si km sn yo so af oc ka be tg sd gu am yi lo uz fo ht ps tk nn mt sa "
             "lb my bo tl mg as tt haw ln ha ba jw su yue").split()

_GPT2_SPLIT = r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""


def special_token_names(num_languages: int = 99) -> List[str]:
    """The specials appended after the mergeable ranks (tokenizer.py:get_encoding), in id order."""
    return (["<|endoftext|>", "<|startoftranscript|>"] + [f"<|{lang}|>" for lang in LANGUAGES[:num_languages]]
            + ["<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>", "<|nospeech|>", "<|notimestamps|>"]
            + [f"<|{i * 0.02:.2f}|>" for i in range(1501)])


class WhisperTokenizer:
    """``ranks``: {token bytes: id} of the mergeable vocabulary (ids 0 .. n-1, every single byte present)."""

    def __init__(self, ranks: Dict[bytes, int], num_languages: int = 99, language: str = "fr", task: str = "transcribe"):
        self.ranks = dict(ranks)
        n = len(self.ranks)
        if sorted(self.ranks.values()) != list(range(n)):
            raise ValueError("mergeable ranks must be the ids 0 .. n-1")
        self.id_to_bytes: List[bytes] = [b""] * n
        for tok, i in self.ranks.items():
            self.id_to_bytes[i] = tok
        self.special: Dict[str, int] = {name: n + k for k, name in enumerate(special_token_names(num_languages))}
        self.special_names = {v: k for k, v in self.special.items()}
        self.n_vocab = n + len(self.special)
        self.num_languages = num_languages
        self.language, self.task = language, task
        self.eot = self.special["<|endoftext|>"]
        self.sot = self.special["<|startoftranscript|>"]
        self.translate, self.transcribe = self.special["<|translate|>"], self.special["<|transcribe|>"]
        self.sot_lm, self.sot_prev = self.special["<|startoflm|>"], self.special["<|startofprev|>"]
        self.no_speech, self.no_timestamps = self.special["<|nospeech|>"], self.special["<|notimestamps|>"]
        self.timestamp_begin = self.special["<|0.00|>"]

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_tiktoken_file(cls, path, **kw) -> "WhisperTokenizer":
        ranks = {}
        with open(path, "rb") as f:
            for line in f:
                if line.strip():
                    tok, rank = line.split()
                    ranks[base64.b64decode(tok)] = int(rank)
        return cls(ranks, **kw)

    @classmethod
    def toy(cls, merges: Iterable[bytes] = (), **kw) -> "WhisperTokenizer":
        """256 single-byte tokens + the given multi-byte tokens (in merge order): enough structure for tests."""
        ranks = {bytes([b]): b for b in range(256)}
        for m in merges:
            if m not in ranks:
                ranks[m] = len(ranks)
        return cls(ranks, **kw)

    # ------------------------------------------------------------------ prompt pieces
    def language_token(self, language: str = None) -> int:
        return self.special[f"<|{language or self.language}|>"]

    def sot_sequence(self, language: str = None, task: str = None) -> Tuple[int, ...]:
        """<|startoftranscript|><|lang|><|task|> (tokenizer.py ``sot_sequence`` for a multilingual model)."""
        return (self.sot, self.language_token(language), self.transcribe if (task or self.task) == "transcribe" else self.translate)

    # ------------------------------------------------------------------ ids -> text
    def decode_bytes(self, tokens: Sequence[int]) -> bytes:
        return b"".join(self.id_to_bytes[t] for t in tokens if t < len(self.id_to_bytes))

    def decode(self, tokens: Sequence[int]) -> str:
        """Text of the ordinary tokens (timestamps and other specials dropped, as ``Tokenizer.decode`` does)."""
        return self.decode_bytes([t for t in tokens if t < self.eot]).decode("utf-8", errors="replace")

    def decode_with_timestamps(self, tokens: Sequence[int]) -> str:
        out, run = [], []
        for t in tokens:
            if t >= self.eot:
                if run:
                    out.append(self.decode_bytes(run).decode("utf-8", errors="replace")); run = []
                if t >= self.timestamp_begin:
                    out.append(f"<|{(t - self.timestamp_begin) * 0.02:.2f}|>")
                else:
                    out.append(self.special_names.get(t, ""))
            else:
                run.append(t)
        if run:
            out.append(self.decode_bytes(run).decode("utf-8", errors="replace"))
        return "".join(out)

    # ------------------------------------------------------------------ text -> ids (prompts, the suppress list)
    def _bpe(self, piece: bytes) -> List[int]:
        parts = [bytes([b]) for b in piece]
        while len(parts) > 1:
            best, at = None, -1
            for i in range(len(parts) - 1):
                r = self.ranks.get(parts[i] + parts[i + 1])
                if r is not None and (best is None or r < best):
                    best, at = r, i
            if best is None:
                break
            parts[at:at + 2] = [parts[at] + parts[at + 1]]
        return [self.ranks[p] for p in parts]

    def encode(self, text: str) -> List[int]:
        import regex
        out: List[int] = []
        for piece in regex.findall(_GPT2_SPLIT, text):
            out += self._bpe(piece.encode("utf-8"))
        return out

    def non_speech_tokens(self) -> Tuple[int, ...]:
        """tokenizer.py ``non_speech_tokens``: symbols the decoder must not emit for plain speech (the "-1" entry of
        ``suppress_tokens``): brackets, musical notes and the like, alone and after a space."""
        symbols = list('"#()*+/:;<=>@[\\]^_`{|}~「」『』')
        symbols += "<< >> <<< >>> -- --- -( -[ (' (\" (( )) ((( ))) [[ ]] {{ }} ♪♪ ♪♪♪".split()
        miscellaneous = set("♩♪♫♬♭♮♯")
        result = {self.encode(" -")[0], self.encode(" '")[0]}
        for symbol in symbols + list(miscellaneous):
            for tokens in (self.encode(symbol), self.encode(" " + symbol)):
                if len(tokens) == 1 or symbol in miscellaneous:
                    result.add(tokens[0])
        return tuple(sorted(result))

    def suppress_list(self) -> Tuple[int, ...]:
        """DecodingTask._get_suppress_tokens for ``suppress_tokens="-1"``: the non-speech symbols plus the task / prompt
        specials that may never be sampled."""
        return tuple(sorted(set(self.non_speech_tokens()) | {self.transcribe, self.translate, self.sot, self.sot_prev, self.sot_lm, self.no_speech}))

    def blank_tokens(self) -> Tuple[int, ...]:
        """SuppressBlank: a space, and end-of-text, at the first sampled position."""
        return tuple(self.encode(" ")) + (self.eot,)

    def decoding_rules(self, max_initial_timestamp: float = 1.0) -> dict:
        """The rule set ``Aligners.decoding.greedy_decode`` takes (DecodingOptions defaults of whisper.transcribe)."""
        return {"eot": self.eot, "no_timestamps": self.no_timestamps, "timestamp_begin": self.timestamp_begin,
                "suppress_tokens": self.suppress_list(), "blank_tokens": self.blank_tokens(),
                "max_initial_timestamp_index": None if max_initial_timestamp is None else round(max_initial_timestamp / 0.02)}

    # ------------------------------------------------------------------ word grouping (timing.py find_alignment)
    def split_tokens_on_unicode(self, tokens: Sequence[int]):
        """tokenizer.py ``split_tokens_on_unicode``: cut wherever the tokens so far decode to valid UTF-8."""
        decoded_full = self.decode_with_timestamps(tokens)
        rep = "�"
        words, word_tokens, current, offset = [], [], [], 0
        for t in tokens:
            current.append(t)
            decoded = self.decode_with_timestamps(current)
            if rep not in decoded or decoded_full[offset + decoded.index(rep)] == rep:
                words.append(decoded); word_tokens.append(current); current = []
                offset += len(decoded)
        return words, word_tokens

    def split_tokens_on_spaces(self, tokens: Sequence[int]):
        """tokenizer.py ``split_tokens_on_spaces``: a sub-word opens a word when it is special, starts with a space
        or is punctuation; otherwise it continues the previous word."""
        subwords, subword_tokens = self.split_tokens_on_unicode(tokens)
        words, word_tokens = [], []
        for sw, st in zip(subwords, subword_tokens):
            special = st[0] >= self.eot
            with_space = sw.startswith(" ")
            punctuation = sw.strip() in string.punctuation
            if special or with_space or punctuation or not words:
                words.append(sw); word_tokens.append(list(st))
            else:
                words[-1] = words[-1] + sw
                word_tokens[-1].extend(st)
        return words, word_tokens

    def split_to_word_tokens(self, tokens: Sequence[int]):
        if self.language in {"zh", "ja", "th", "lo", "my", "yue"}:
            return self.split_tokens_on_unicode(tokens)
        return self.split_tokens_on_spaces(tokens)
