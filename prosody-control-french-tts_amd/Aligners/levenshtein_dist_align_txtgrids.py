"""Word-level Levenshtein matching of two TextGrids (legacy aligner, ``Code/Aligners/levenshtein_dist_align_txtgrids.py``).

What the reference's module computes on strings is ``levenshtein_distance`` (:43-70): the two-row edit-distance DP over Python
characters with unit costs.  Here every distance is one lane-parallel integer DP on the GPU (``pce_levenshtein``: one wave per pair,
the whole batch in one launch), bit-identical to the reference function -- golden G9 (``tests/golden/levenshtein.json``) holds its
outputs on 350 pairs (accents, empty strings, the ``len(s1) < len(s2)`` swap, strings longer than a stripe).

**``main()`` of the reference (:98-158) is not mirrored, because it has no output to mirror**: its merge loop runs
``while i < n1 and j < n2`` but only ever advances the cursors to ``i_ = min(i + 1, n1 - 1)`` / ``j_ = min(j + 1, n2 - 1)`` (:113), so
``i < n1`` and ``j < n2`` hold forever -- on any pair of non-empty tiers the loop appends the last words (or grows ``w1`` / ``w2``) until
memory runs out, and the two ``write`` calls behind it (:151-152) are never reached (an empty tier raises IndexError at :108 instead).
:func:`merge_word_tiers` / :func:`main` below are the TERMINATING form of that loop -- the same three distances per step and the same
decisions (:112-136), with the cursors allowed to leave the tiers so that the tail loops (:138-146) run -- and are this package's own:
nothing in the reference pins them.  Several file pairs advance in lock-step, so a step of all of them is ONE ``pce_levenshtein`` launch.

``normalize_word`` (:34-41) needs ``unidecode`` (absent offline; the reference's own ``main`` never calls it): restated with NFKD
decomposition plus the Latin ligatures French text holds -- unpinned.
"""
from __future__ import annotations

import logging
import os
import re
import unicodedata
from typing import List, Sequence, Tuple

from ..textgrid_io import IntervalTier, TextGrid, read_textgrid, write_textgrid

_LIGATURES = {"œ": "oe", "Œ": "OE", "æ": "ae", "Æ": "AE", "ß": "ss", "’": "'", "‘": "'", "«": "<<", "»": ">>", "–": "-", "—": "--", "…": "..."}


def _engine(engine=None):
    if engine is None:
        from ..engine import get_default_engine
        engine = get_default_engine()
    return engine


def levenshtein_distances(pairs: Sequence[Tuple[str, str]], engine=None) -> List[int]:
    """Distances of a batch of (s1, s2) pairs in one launch."""
    return [int(d) for d in _engine(engine).levenshtein(list(pairs))]


def levenshtein_distance(s1: str, s2: str, engine=None) -> int:
    """``levenshtein_distance(s1, s2)`` (:43-70).  One pair = one launch: callers with many pairs use :func:`levenshtein_distances`."""
    return levenshtein_distances([(s1, s2)], engine)[0]


def normalize_word(word: str) -> str:
    """(:34-41) accents / punctuation / case removed.  ``unidecode`` is replaced by NFKD + a ligature table (unpinned, see above)."""
    s = "".join(_LIGATURES.get(ch, ch) for ch in word)
    s = "".join(ch for ch in unicodedata.normalize("NFKD", s) if not unicodedata.combining(ch))
    s = s.encode("ascii", "ignore").decode("ascii")
    s = s.replace(" ", "")
    for symbol in [".", ", ", "!", "?", ";", ":", "-"]:          # (", " can no longer occur once the blanks are gone: as in the reference)
        s = s.replace(symbol, "")
    return s.lower()


def list_to_textgrid(L, name="words") -> TextGrid:
    """(:11-32) (text, min, max) triples -> one tier: sorted by start, starts clamped to the previous end, empty intervals given 10 ms."""
    L = sorted(L, key=lambda x: x[1])
    tier = IntervalTier(name)
    last_max = 0.0
    for text, min_t, max_t in L:
        if min_t < last_max:
            min_t = last_max
        if max_t <= min_t:
            max_t = min_t + 0.01
        tier.add(min_t, max_t, text)
        last_max = max_t
    tier.tier_max = last_max if L else 0.0
    return TextGrid([tier])


def extract_transcription_from_textgrid(tg_path, output_txt_path) -> bool:
    """(:72-84)"""
    try:
        tg = read_textgrid(tg_path)
        words = [m for _, _, m in tg.tiers[0].intervals if m.strip()]
        transcription = re.sub(r"\s+", " ", " ".join(words)).strip()
        with open(output_txt_path, "w", encoding="utf-8") as f:
            f.write(transcription)
        logging.info(f"Transcription extracted to {output_txt_path}")
        return True
    except Exception as e:                                                  # noqa: BLE001  (the reference logs and goes on)
        logging.error(f"Error extracting transcription: {e}")
        return False


def update_transcription(textgrid_path, transcription_dir) -> None:
    """(:86-96)"""
    try:
        base_name = os.path.basename(textgrid_path).replace(".TextGrid", "")
        txt_path = os.path.join(transcription_dir, f"{base_name}.txt")
        if extract_transcription_from_textgrid(textgrid_path, txt_path):
            logging.info(f"Transcription updated: {base_name}")
        else:
            logging.warning(f"Update failed: {base_name}")
    except Exception as e:                                                  # noqa: BLE001
        logging.error(f"Error updating transcription: {e}")


class _Merge:
    """Cursor state of one tier pair (the locals of the reference's loop, :104-110)."""

    def __init__(self, I1, I2):
        self.I1, self.I2 = I1, I2
        self.n1, self.n2 = len(I1), len(I2)
        self.i = self.j = 0
        self.last1 = self.last2 = -1
        self.new1, self.new2 = [], []
        self.w1 = I1[0][2] if I1 else ""
        self.w2 = I2[0][2] if I2 else ""

    def live(self):
        return self.i < self.n1 and self.j < self.n2

    def _next(self, which):
        I, k = (self.I1, self.i) if which == 1 else (self.I2, self.j)
        return I[k + 1][2] if k + 1 < len(I) else ""

    def _start(self, which):
        I, last = (self.I1, self.last1) if which == 1 else (self.I2, self.last2)
        return I[last][1] if last != -1 else I[0][0]

    def skip_blanks(self):
        """The two ``continue`` branches (:115-122): blank marks pass through as " " intervals, no distance needed."""
        moved = True
        while moved and self.live():
            moved = False
            if self.w1.strip() == "":
                self.new1.append((" ", self._start(1), self.I1[self.i][1]))
                self.last1, self.i = self.i, self.i + 1
                self.w1 = self.I1[self.i][2] if self.i < self.n1 else ""
                moved = True
            elif self.w2.strip() == "":
                self.new2.append((" ", self._start(2), self.I2[self.j][1]))
                self.last2, self.j = self.j, self.j + 1
                self.w2 = self.I2[self.j][2] if self.j < self.n2 else ""
                moved = True

    def wanted(self):
        """The three string pairs of a step: d, di, dj (:112, :124-125)."""
        return [(self.w1, self.w2), (self.w1 + self._next(1), self.w2), (self.w1, self.w2 + self._next(2))]

    def decide(self, d, di, dj):
        """(:127-136)"""
        if d <= di and d <= dj:
            chosen = self.w2 if len(self.w2) > len(self.w1) else self.w1
            self.new1.append((chosen, self._start(1), self.I1[self.i][1]))
            self.new2.append((chosen, self._start(2), self.I2[self.j][1]))
            self.last1, self.last2, self.i, self.j = self.i, self.j, self.i + 1, self.j + 1
            self.w1 = self.I1[self.i][2] if self.i < self.n1 else ""
            self.w2 = self.I2[self.j][2] if self.j < self.n2 else ""
        elif di <= dj:        # (a next word exists here: past the end of a tier the merged string is w itself, di == d, and the first branch took it)
            self.i += 1; self.w1 = self.w1 + " " + self.I1[self.i][2]
        else:
            self.j += 1; self.w2 = self.w2 + " " + self.I2[self.j][2]

    def tails(self):
        """(:138-146)"""
        while self.i < self.n1:
            self.new1.append((self.I1[self.i][2], self._start(1), self.I1[self.i][1]))
            self.i, self.last1 = self.i + 1, self.i
        while self.j < self.n2:
            self.new2.append((self.I2[self.j][2], self._start(2), self.I2[self.j][1]))
            self.j, self.last2 = self.j + 1, self.j
        return self.new1, self.new2


def merge_word_tiers(tier_pairs, engine=None):
    """The terminating form of the reference's merge loop (see the module docstring) over several tier pairs at once.
    ``tier_pairs``: [(intervals1, intervals2), ...] with intervals = [(min, max, mark), ...]; -> [(new1, new2), ...] of
    (text, min, max) triples as :func:`list_to_textgrid` takes them.  All live pairs take a step together: their 3 distances each
    are one ``pce_levenshtein`` launch."""
    eng = _engine(engine)
    states = [_Merge(list(a), list(b)) for a, b in tier_pairs]
    while True:
        for s in states:
            s.skip_blanks()
        live = [s for s in states if s.live()]
        if not live:
            break
        want = [p for s in live for p in s.wanted()]
        dist = eng.levenshtein(want)
        for k, s in enumerate(live):
            s.decide(int(dist[3 * k]), int(dist[3 * k + 1]), int(dist[3 * k + 2]))
    return [s.tails() for s in states]


def main(textgrid1_input_path, textgrid2_input_path, transcription1_dir=None, transcription2_dir=None, engine=None):
    """Signature of the reference's ``main`` (:98); the terminating merge (module docstring), both TextGrids rewritten in place."""
    logging.info(f"Processing {textgrid1_input_path}, {textgrid2_input_path}")
    tg1, tg2 = read_textgrid(textgrid1_input_path), read_textgrid(textgrid2_input_path)
    (new1, new2), = merge_word_tiers([(tg1.tiers[0].intervals, tg2.tiers[0].intervals)], engine)
    write_textgrid(list_to_textgrid(new1), textgrid1_input_path)
    write_textgrid(list_to_textgrid(new2), textgrid2_input_path)
    if transcription1_dir:
        update_transcription(textgrid1_input_path, transcription1_dir)
    if transcription2_dir:
        update_transcription(textgrid2_input_path, transcription2_dir)
    logging.info("Alignment completed successfully.")
