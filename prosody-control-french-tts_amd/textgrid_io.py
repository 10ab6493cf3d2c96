"""Praat TextGrid text format: the wire format between alignment and prosody (R9).

Reader for the long and short text formats (what ``textgrid==1.6.1``'s ``TextGrid.read``
accepts and what its ``write`` emits) and a writer producing the long format with the same
field layout.  Only IntervalTiers are needed by the hot path
(Code/Preprocessing/gen_break_ssml.py:21-29, Code/Aligners/use_whisper_timestamped.py:330-395).
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import List, Optional, Tuple


@dataclass
class IntervalTier:
    """``textgrid.IntervalTier(name, minTime=0.0, maxTime=None)``: intervals kept sorted, no overlaps."""
    name: str = ""
    intervals: List[Tuple[float, float, str]] = field(default_factory=list)
    tier_min: float = 0.0
    tier_max: Optional[float] = None

    def add(self, t_min, t_max, mark: str):
        """``IntervalTier.add``: ValueError for an empty / inverted interval (``textgrid.Interval``), one that starts
        before the tier or ends after it, and one that overlaps an interval already there (the sorted insertion compares
        intervals, and comparing overlapping intervals raises) -- the errors that make the reference give a file up."""
        if t_min >= t_max:
            raise ValueError(t_min, t_max)
        if t_min < self.tier_min:
            raise ValueError(self.tier_min)
        if self.tier_max and t_max > self.tier_max:
            raise ValueError(self.tier_max)
        for a, b, _ in self.intervals:
            if a < t_max and t_min < b:
                raise ValueError((a, b), (t_min, t_max))
        k = 0
        while k < len(self.intervals) and self.intervals[k][0] < t_min:
            k += 1
        self.intervals.insert(k, (t_min, t_max, mark))

    @property
    def min_time(self):
        return self.intervals[0][0] if self.intervals else 0.0

    @property
    def max_time(self):
        return self.intervals[-1][1] if self.intervals else 0.0


@dataclass
class TextGrid:
    tiers: List[IntervalTier] = field(default_factory=list)
    min_time: float = 0.0
    max_time: float = 0.0


_STR = r'"((?:[^"]|"")*)"'


def _decode(raw: bytes) -> str:
    for enc in ("utf-8-sig", "utf-16"):
        try:
            return raw.decode(enc)
        except UnicodeError:
            continue
    return raw.decode("latin-1")


def read_textgrid(path) -> TextGrid:
    text = _decode(open(path, "rb").read())
    # tokenise into numbers and quoted strings after dropping the "key =" decorations
    body = text.split("\n", 2)[-1] if text.lstrip().startswith("File type") else text
    # "item [2]:" / "intervals [17]:" carry indices, not values: blank them outside quoted strings
    body = re.sub(_STR + r"|\[\d*\]", lambda m: m.group(0) if m.group(0).startswith('"') else " ", body)
    toks = re.findall(_STR + r"|(<exists>)|(-?\d+(?:\.\d+)?(?:[eE][-+]?\d+)?)", body)
    vals = []
    for s, ex, num in toks:
        if ex:
            vals.append(("flag", ex))
        elif num != "":
            vals.append(("num", float(num)))
        else:
            vals.append(("str", s.replace('""', '"')))
    # layout: xmin xmax <exists> n_tiers { "IntervalTier" name xmin xmax n { xmin xmax text } }
    i = 0
    def take(kind):
        nonlocal i
        while vals[i][0] != kind:
            i += 1
        v = vals[i][1]; i += 1
        return v
    tg = TextGrid()
    # the header string "TextGrid" (Object class) may or may not have been consumed
    if vals and vals[0] == ("str", "TextGrid"):
        i = 1
    tg.min_time = take("num"); tg.max_time = take("num")
    take("flag")
    n_tiers = int(take("num"))
    for _ in range(n_tiers):
        cls = take("str"); name = take("str")
        t0 = take("num"); t1 = take("num"); n = int(take("num"))
        tier = IntervalTier(name, tier_min=t0, tier_max=t1)
        for _ in range(n):
            if cls == "IntervalTier":
                a = take("num"); b = take("num"); m = take("str")
                tier.intervals.append((a, b, m))
            else:                                   # TextTier (points): time + mark, ignored by the hot path
                take("num"); take("str")
        if cls == "IntervalTier":
            tg.tiers.append(tier)
    return tg


def _q(s: str) -> str:
    return '"' + s.replace('"', '""') + '"'


def _gaps_filled(tier: IntervalTier, null: str = ""):
    """``IntervalTier._fillInTheGaps``: empty-mark intervals between the tier's start, the intervals and the TIER's own
    end (a tier created without ``maxTime`` gets no trailing filler)."""
    out, prev = [], tier.tier_min
    for a, b, m in tier.intervals:
        if prev < a:
            out.append((prev, a, null))
        out.append((a, b, m)); prev = b
    if tier.tier_max is not None and prev < tier.tier_max:
        out.append((prev, tier.tier_max, null))
    return out


def format_textgrid(tg: TextGrid) -> str:
    """The long text format exactly as ``textgrid==1.6.1``'s ``TextGrid.write`` prints it (tab indentation, numbers through
    ``str``, quotes doubled): what the reference's ``tg.write(path)`` leaves on disk
    (Code/Aligners/use_whisper_timestamped.py:612,638,656,692)."""
    max_t = tg.max_time
    if not max_t:
        max_t = max([(t.tier_max if t.tier_max else t.max_time) for t in tg.tiers])
    out = ['File type = "ooTextFile"', 'Object class = "TextGrid"\n', f"xmin = {tg.min_time}", f"xmax = {max_t}", "tiers? <exists>",
           f"size = {len(tg.tiers)}", "item []:"]
    for ti, tier in enumerate(tg.tiers, 1):
        ivs = _gaps_filled(tier)
        out += [f"\titem [{ti}]:", '\t\tclass = "IntervalTier"', f'\t\tname = "{tier.name}"', f"\t\txmin = {tier.tier_min}", f"\t\txmax = {max_t}",
                f"\t\tintervals: size = {len(ivs)}"]
        for k, (a, b, m) in enumerate(ivs, 1):
            out += [f"\t\t\tintervals [{k}]:", f"\t\t\t\txmin = {a}", f"\t\t\t\txmax = {b}", f'\t\t\t\ttext = {_q(m)}']
    return "\n".join(out) + "\n"


def write_textgrid(tg: TextGrid, path):
    with open(path, "w", encoding="utf-8") as f:
        f.write(format_textgrid(tg))


def words_to_textgrid(result: dict) -> TextGrid:
    """``json_to_textgrid`` (Code/Aligners/use_whisper_timestamped.py:330-395) on an in-memory
    transcription dict: one "words" tier, " " fillers for gaps, ``start >= end`` repaired by
    +10 ms, "[*]" -> " ", a single "..." interval when there is no word; ``maxTime`` of the grid = end of the last word.
    Raises ValueError where ``textgrid`` would (overlapping words)."""
    tier = IntervalTier("words")
    cur, total = 0.0, 0
    for seg in result["segments"]:
        for w in seg["words"]:
            total += 1
            if w["start"] >= w["end"]:
                w["end"] = w["start"] + 0.01
            if w["start"] > cur:
                tier.add(cur, w["start"], " ")
            tier.add(w["start"], w["end"], w["text"].replace("[*]", " "))
            cur = w["end"]
    if total == 0:
        xmax = 1.0
        if result["segments"] and "end" in result["segments"][-1]:
            xmax = result["segments"][-1]["end"]
        tier.add(0.0, xmax, "...")
        cur = xmax
    return TextGrid([tier], 0.0, cur)
