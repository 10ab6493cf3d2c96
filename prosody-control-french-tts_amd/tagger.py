"""SSML tagger: per-utterance / per-syntagme statistics -> prosody adjustments -> SSML tables.

Host-side consumer of the gathered measurements.  It restates the
decision logic of ``AudioPipeline.measure_prosody_and_build_ssml``
(Code/audioPipeline.py:261-711): word/pause sequence clean-up, syntagme construction on the
cumulative-millisecond timeline, sliding-median baselines, semitone / dB / rate clamps, EMA
smoothing with a jump limit, and the three CSV tables.  All audio arithmetic is *not* here: a
``MeasurementSource`` answers pitch / loudness / duration queries (in production from one
batched GPU pass, see :mod:`.audio_pipeline`; in the parity tests from values the reference
itself consumed, tests/golden/tagger.json).

Scalar arithmetic keeps the reference's operation order in float64 so the formatted
``{:+.2f}%`` strings come out identical.
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple
from xml.sax.saxutils import escape as xml_escape

import numpy as np
import pandas as pd

from .hostrules import CouldntDecodeError

FORBIDDEN_POS = {"DET", "ADP", "CCONJ", "SCONJ", "PART", "PRON"}          # Code/audioPipeline.py:27
PAUSE_MARKERS = {"[*]"}                                                   # Code/audioPipeline.py:65
INITIAL_PAUSE_THRESHOLD_MS = 150                                          # Code/Preprocessing/gen_break_ssml.py:9

Item = Tuple[str, Optional[str], int]          # ("word"|"pause", token, duration_ms)


# --------------------------------------------------------------------------- text side
class SimpleToken:
    __slots__ = ("text", "text_with_ws", "pos_")

    def __init__(self, text, ws, pos):
        self.text, self.text_with_ws, self.pos_ = text, text + ws, pos


class TablePosTagger:
    """spaCy-shaped callable (``nlp(text) -> tokens`` with ``text``, ``text_with_ws``, ``pos_``)
    backed by a closed-class word table.  Used when spaCy's ``fr_core_news_sm`` is not
    installed; pass ``spacy.load(...)`` to :class:`SsmlTagger` to get the reference's tagger."""

    DEFAULT = {
        **{w: "DET" for w in ("le", "la", "les", "l", "un", "une", "des", "du", "ce", "cet", "cette", "ces", "mon", "ma", "mes",
                              "ton", "ta", "tes", "son", "sa", "ses", "notre", "nos", "votre", "vos", "leur", "leurs")},
        **{w: "ADP" for w in ("de", "à", "dans", "en", "sur", "sous", "par", "pour", "avec", "sans", "chez", "vers", "entre", "d")},
        **{w: "CCONJ" for w in ("et", "ou", "mais", "donc", "or", "ni", "car")},
        **{w: "SCONJ" for w in ("que", "qu", "si", "comme", "quand", "lorsque", "puisque")},
        **{w: "PRON" for w in ("je", "tu", "il", "elle", "on", "nous", "vous", "ils", "elles", "qui", "me", "te", "se", "y")},
        **{w: "PART" for w in ("ne", "n")},
        **{p: "PUNCT" for p in (",", ".", "?", "!", ";", ":")},
    }
    _TOK = re.compile(r"(\[\*\]|\w+|[^\w\s])(\s*)")

    def __init__(self, table=None, default_pos="NOUN"):
        self.table = dict(self.DEFAULT if table is None else table)
        self.default_pos = default_pos

    def __call__(self, text: str):
        toks = [SimpleToken(m.group(1), m.group(2), self.table.get(m.group(1).lower(), self.default_pos))
                for m in self._TOK.finditer(text)]
        return toks or [SimpleToken("", "", self.default_pos)]


def remove_spurious_commas(text: str, nlp) -> str:
    """Drop a comma / pause marker that directly follows a DET/ADP/CCONJ/SCONJ/PART/PRON token
    (Code/audioPipeline.py:64-81)."""
    kept = []
    for tok in nlp(text):
        if (tok.text == "," or tok.text in PAUSE_MARKERS) and kept and kept[-1].pos_ in FORBIDDEN_POS:
            continue
        kept.append(tok)
    return "".join(t.text_with_ws for t in kept)


def words_and_pauses(intervals: Sequence[Tuple[float, float, str]]) -> List[Item]:
    """TextGrid tier -> [(kind, token, duration_ms)] (Code/Preprocessing/gen_break_ssml.py:12-42):
    ms by ``round(t*1000)``, pauses before the first word kept only from 150 ms up."""
    seq: List[Item] = []
    before_first_word = True
    for t_min, t_max, mark in intervals:
        text = mark.strip()
        dur = round(t_max * 1000) - round(t_min * 1000)
        if not text or text == " ":
            if (not before_first_word) or dur >= INITIAL_PAUSE_THRESHOLD_MS:
                seq.append(("pause", None, dur))
        else:
            seq.append(("word", text, dur))
            before_first_word = False
    return seq


def clean_sequence(raw_seq: Sequence[Item], nlp, end_pause_ms: int) -> List[Item]:
    """Comma clean-up, POS-based pause filtering and sentence-end pause clamp / injection
    (Code/audioPipeline.py:444-489)."""
    seq = [(k, remove_spurious_commas(t, nlp) if k == "word" else t, d) for k, t, d in raw_seq]
    filtered: List[Item] = []
    prev: Optional[Item] = None
    for item in seq:
        kind, tok, dur = item
        if kind == "pause" and prev is not None and prev[0] == "word" and nlp(prev[1].strip())[0].pos_ in FORBIDDEN_POS:
            prev = item
            continue
        filtered.append(item)
        prev = item
    out: List[Item] = []
    for i, (kind, tok, dur) in enumerate(filtered):
        if kind == "pause" and i > 0:
            pk, pt, _ = filtered[i - 1]
            if pk == "word" and pt.strip().endswith((".", "?", "!")):
                dur = max(dur, end_pause_ms)
        out.append((kind, tok, dur))
        if kind == "word" and tok.strip().endswith((".", "?", "!")):
            if not (i + 1 < len(filtered) and filtered[i + 1][0] == "pause"):
                out.append(("pause", "", end_pause_ms))
    return out


def build_syntagmes(seq: Sequence[Item]) -> List[dict]:
    """Word runs and pauses as syntagmes on a cumulative-ms cursor (Code/audioPipeline.py:265-311)."""
    out, words, cursor, start = [], [], 0, 0
    for kind, tok, dur in seq:
        if kind == "word":
            if not words:
                start = cursor
            words.append(tok.strip())
            cursor += dur
        else:
            if words:
                out.append({"words": " ".join(words), "start_ms": start, "end_ms": cursor, "pause_ms": 0})
                words = []
            out.append({"words": "", "start_ms": cursor, "end_ms": cursor + dur, "pause_ms": dur})
            cursor += dur
    if words:
        out.append({"words": " ".join(words), "start_ms": start, "end_ms": cursor, "pause_ms": 0})
    return out


# --------------------------------------------------------------------------- measurements
class MeasurementSource:
    """What the tagger asks about audio.  ``kind`` is "nat" (the recording) or "syn" (the raw
    synthesis of the same segment).  Semantics of the four reference closures
    (Code/audioPipeline.py:314-361), sentinels included; ``lufs`` / ``duration`` /
    ``part_duration`` raise :class:`CouldntDecodeError` for an undecodable file."""

    def median_pitch(self, segment: str, t0: float = 0.0, t1: Optional[float] = None) -> float: raise NotImplementedError
    def lufs(self, kind: str, segment: str, t0: float = 0.0, t1: Optional[float] = None) -> float: raise NotImplementedError
    def duration(self, kind: str, segment: str) -> float: raise NotImplementedError
    def part_duration(self, kind: str, segment: str, t0: float = 0.0, t1: Optional[float] = None) -> float: raise NotImplementedError


@dataclass
class ProsodySettings:
    """``prosody_settings`` of config.yaml with the code defaults of Code/audioPipeline.py:127-139."""
    pitch_semitones: float = 2.0
    pitch_lower_clip_factor: float = 0.7
    volume_pct: float = 7.0
    rate_percent: float = 15.0
    smoothing_alpha: float = 0.4
    max_jump_percent: float = 5.0
    end_punctuation_pause_ms: int = 150
    baseline_window: Optional[int] = None
    inter_syntagme_pause_factor: float = 1
    threshold_duration_before_slowing_down: float = 1.0
    slow_floor_per_sec: float = 2.0

    @classmethod
    def from_config(cls, cfg: dict) -> "ProsodySettings":
        known = {f for f in cls.__dataclass_fields__}
        return cls(**{k: v for k, v in (cfg or {}).items() if k in known})


@dataclass
class SegmentInput:
    name: str                                          # "segment_ph12"
    intervals: Sequence[Tuple[float, float, str]]      # first TextGrid tier


@dataclass
class TaggerResult:
    segment_stats: List[dict] = field(default_factory=list)
    baselines: List[dict] = field(default_factory=list)
    rows: List[dict] = field(default_factory=list)      # raw adjustments per syntagme
    smooth_pitch: List[float] = field(default_factory=list)
    smooth_rate: List[float] = field(default_factory=list)
    bdd_ssml: Optional[pd.DataFrame] = None
    bdd_syntagme_ssml: Optional[pd.DataFrame] = None
    bdd_syntagme_for_synth: Optional[pd.DataFrame] = None


def segment_sort_key(name: str) -> int:
    return int(re.search(r"segment_ph(\d+)", name).group(1))          # Code/audioPipeline.py:366


class SsmlTagger:
    def __init__(self, settings: ProsodySettings, azure_voice: str, nlp: Optional[Callable] = None):
        self.s = settings
        self.voice = azure_voice
        self.nlp = nlp if nlp is not None else TablePosTagger()

    # -- step 1: per-segment statistics (the records that are all-gathered across GPUs)
    def segment_statistics(self, segments: Sequence[SegmentInput], src: MeasurementSource) -> List[dict]:
        stats = []
        for seg in segments:
            seq = words_and_pauses(seg.intervals)
            wc = sum(1 for k, t, _ in seq if k == "word" and t.strip())
            p_nat = src.median_pitch(seg.name)
            l_nat = src.lufs("nat", seg.name)
            try:
                l_syn = src.lufs("syn", seg.name)
                d_syn = src.duration("syn", seg.name)
            except CouldntDecodeError:
                l_syn = l_nat
                d_syn = src.duration("nat", seg.name)
            d_nat = src.duration("nat", seg.name)
            rate_ratio = (wc / d_nat) / (wc / d_syn) if wc > 0 and d_syn > 0 else 1.0
            stats.append({"segment": seg.name, "p_nat": p_nat, "l_nat": l_nat, "l_syn": l_syn, "d_nat": d_nat, "d_syn": d_syn,
                          "wc": wc, "rate_ratio": rate_ratio})
        return stats

    # -- step 2: baselines over ALL segments (global or sliding median)
    def baselines(self, stats: Sequence[dict]) -> List[dict]:
        n, win = len(stats), self.s.baseline_window

        def med_f0(ws):
            return float(np.median([w["p_nat"] for w in ws if w["p_nat"] > 0])) or 1.0

        if win is None or win >= n:
            b = {"f0": med_f0(stats), "loud": float(np.median([w["l_nat"] for w in stats])),
                 "rate": float(np.median([w["rate_ratio"] for w in stats]))}
            return [dict(b) for _ in range(n)]
        half = win // 2
        out = []
        for i in range(n):
            ws = stats[max(0, i - half):min(n, i + half + 1)]
            out.append({"f0": med_f0(ws), "loud": float(np.median([w["l_nat"] for w in ws])),
                        "rate": float(np.median([w["rate_ratio"] for w in ws]))})
        return out

    # -- step 3: raw adjustments per syntagme
    def syntagmes_of(self, seg: SegmentInput) -> List[dict]:
        return build_syntagmes(clean_sequence(words_and_pauses(seg.intervals), self.nlp, self.s.end_punctuation_pause_ms))

    def syntagme_measurements(self, segments: Sequence[SegmentInput], src: MeasurementSource) -> List[dict]:
        """Everything Code/audioPipeline.py:499-523 MEASURES per syntagme (no baseline enters yet): these are the
        per-syntagme records of the multi-GPU exchange."""
        meas = []
        for seg in segments:
            for syn in self.syntagmes_of(seg):
                t0, t1 = syn["start_ms"] / 1000, syn["end_ms"] / 1000
                p_nat = src.median_pitch(seg.name, t0, t1)
                src.lufs("nat", seg.name, t0, t1)               # measured by the reference too (value unused)
                try:
                    l_syn = src.lufs("syn", seg.name, t0, t1)
                    syn_total = src.part_duration("syn", seg.name, t0, t1)
                except CouldntDecodeError:
                    l_syn = src.lufs("nat", seg.name, t0, t1)
                    syn_total = src.part_duration("nat", seg.name, t0, t1)
                nat_total = src.part_duration("nat", seg.name, t0, t1)
                meas.append({"segment": seg.name, "syntagme": syn["words"], "pause_ms": syn["pause_ms"], "p_nat": p_nat, "l_syn": l_syn,
                             "nat_total": nat_total, "syn_total": syn_total, "wc_syn": len(syn["words"].split())})
        return meas

    def rows_from_measurements(self, meas: Sequence[dict], base_of: dict) -> List[dict]:
        """The closed-form adjustment formulas of Code/audioPipeline.py:524-577 on measured syntagmes; ``base_of``:
        segment name -> its baselines."""
        s = self.s
        P_ST, R_PCT = s.pitch_semitones, s.rate_percent
        rows = []
        for m in meas:
            b = base_of[m["segment"]]
            p_nat, l_syn, wc_syn = m["p_nat"], m["l_syn"], m["wc_syn"]
            pause_s = m["pause_ms"] / 1000.0
            d_nat = max(m["nat_total"] - pause_s, 1e-4)
            d_syn = max(m["syn_total"] - pause_s, 1e-4)

            if p_nat > 0:
                st = 12 * np.log2(p_nat / b["f0"])
                st = np.clip(st, -P_ST * s.pitch_lower_clip_factor, P_ST)
                p_pct = (2 ** (st / 12) - 1) * 100
            else:
                p_pct = 0.0

            db_diff = b["loud"] - l_syn
            v_pct = (10 ** (db_diff / 20) - 1.0) * 100.0
            v_pct = np.clip(v_pct, -s.volume_pct, +s.volume_pct)

            if wc_syn > 0:
                nat_r, syn_r = wc_syn / d_nat, wc_syn / d_syn
                rp = (nat_r - syn_r) / syn_r * 100
            else:
                rp = 0.0
            length_s = d_nat
            if length_s <= 1.0:
                slow_factor = fast_factor = 1.0
            else:
                slow_factor, fast_factor = length_s ** 1.5, np.sqrt(length_s)
            rp = rp * slow_factor if rp < 0 else rp / fast_factor
            rp = rp - max(0.0, length_s - s.threshold_duration_before_slowing_down) * s.slow_floor_per_sec
            if length_s > 5.0:
                max_slow, max_fast = R_PCT * 1.5, R_PCT * 0.5
            else:
                max_slow, max_fast = R_PCT, R_PCT
            rp = np.clip(rp, -max_slow, +max_fast)
            rows.append({"segment": m["segment"], "syntagme": m["syntagme"], "pause": m["pause_ms"],
                         "raw_pitch": float(p_pct), "raw_volume": float(v_pct), "raw_rate": float(rp)})
        return rows

    def raw_rows(self, segments: Sequence[SegmentInput], base: Sequence[dict], src: MeasurementSource) -> List[dict]:
        return self.rows_from_measurements(self.syntagme_measurements(segments, src), {seg.name: b for seg, b in zip(segments, base)})

    # -- step 4: EMA + jump limit over all syntagmes, in order
    def smooth(self, rows: Sequence[dict]):
        a, mj = self.s.smoothing_alpha, self.s.max_jump_percent
        sm_p, sm_r = [rows[0]["raw_pitch"]], [rows[0]["raw_rate"]]
        for r in rows[1:]:
            sm_p.append(a * r["raw_pitch"] + (1 - a) * sm_p[-1])
            sm_r.append(a * r["raw_rate"] + (1 - a) * sm_r[-1])
        for i in range(1, len(sm_p)):
            if abs(sm_p[i] - sm_p[i - 1]) > mj:
                sm_p[i] = sm_p[i - 1] + np.sign(sm_p[i] - sm_p[i - 1]) * mj
            if abs(sm_r[i] - sm_r[i - 1]) > mj:
                sm_r[i] = sm_r[i - 1] + np.sign(sm_r[i] - sm_r[i - 1]) * mj
        return sm_p, sm_r

    # -- step 5: SSML strings and the three tables
    def _prosody_open(self, row, p_adj, r_adj) -> str:
        return (f'<prosody pitch="{p_adj:+.2f}%" rate="{r_adj:+.2f}%" volume="{row["raw_volume"]:+.2f}%">'
                f'{xml_escape(row["syntagme"])}')

    def _break(self, row) -> str:
        if row["pause"] < 50:
            return ""
        last = row["syntagme"][-1] if row["syntagme"] else None
        dur = row["pause"] if (last is not None and last in ".?!") else int(row["pause"] * self.s.inter_syntagme_pause_factor)
        return f'<break time="{dur}ms"/>'

    def tables(self, rows, sm_p, sm_r):
        ns_full = ('<speak xmlns="http://www.w3.org/2001/10/synthesis" xmlns:mstts="http://www.w3.org/2001/mstts" '
                   'version="1.0" xml:lang="fr-FR">')
        ns_plain = '<speak xmlns="http://www.w3.org/2001/10/synthesis" version="1.0" xml:lang="fr-FR">'
        voice = f'<voice name="{self.voice}">'
        lead, tail = '<mstts:silence type="Leading-exact" value="0"/>', '<mstts:silence type="Tailing-exact" value="0"/>'
        by_seg: dict = {}
        syn_rows, synth_rows = [], []
        for row, p, r in zip(rows, sm_p, sm_r):
            piece = self._prosody_open(row, p, r) + self._break(row) + "</prosody>"
            by_seg.setdefault(row["segment"], []).append(piece)
            common = {"segment": row["segment"], "syntagme": row["syntagme"], "pause": row["pause"]}
            syn_rows.append({**common, "ssml": ns_plain + voice + piece + "</voice></speak>"})
            synth_rows.append({**common, "ssml": ns_full + voice + lead + self._prosody_open(row, p, r) + "</prosody>" + tail
                               + "</voice></speak>"})
        seg_rows = [{"segment": seg, "ssml": ns_full + voice + lead + "".join(pieces) + tail + "</voice></speak>"}
                    for seg, pieces in by_seg.items()]
        return pd.DataFrame(seg_rows), pd.DataFrame(syn_rows), pd.DataFrame(synth_rows)

    # -- multi-GPU: utterances sharded by rank, exactly ONE exchange
    def run_sharded(self, segments: Sequence[SegmentInput], src: MeasurementSource, rank: int, world: int,
                    allgather: Callable[..., np.ndarray]) -> TaggerResult:
        """Rank ``rank`` of ``world`` measures only its contiguous block of the segment-sorted list
        (:func:`shard.shard_range`): the 7 per-segment statistics (Code/audioPipeline.py:391-400) and the per-syntagme
        measurements (:499-523).  Baselines only enter the closed-form formulas AFTER all measurements (:526, :535), so
        ONE all-gather of fp64 records (segment rows and syntagme rows in the same padded block, told apart by column
        0) moves everything; baselines, adjustments, the EMA over ALL syntagmes in segment order (:592-602) and the
        SSML strings are then computed by every rank.  Row counts per rank follow from ``shard_range`` and from the
        TextGrids (every rank reads all of them: texts are not exchanged), so the collective carries records only.
        ``allgather(local, counts)``: :func:`shard.allgather_records`.  Every rank returns the full tables."""
        from .shard import RECORD_WIDTH, SEGMENT_RECORD, SYNTAGME_RECORD, shard_range
        segments = sorted(segments, key=lambda s: segment_sort_key(s.name))
        syn_of = [self.syntagmes_of(seg) for seg in segments]                   # host only: TextGrids + POS table
        spans = [shard_range(len(segments), r, world) for r in range(world)]
        counts = [(b - a) + sum(len(syn_of[i]) for i in range(a, b)) for a, b in spans]
        lo, hi = spans[rank]
        mine = segments[lo:hi]
        res = TaggerResult()
        # the rank-local part may raise (an undecodable file, a slice Praat refuses, a device error): the rank still enters the ONE
        # collective, with its block flagged as failed, so that every rank leaves the step together (``shard.allgather_records``)
        local_error = None
        try:
            local = self.segment_statistics(mine, src)
            meas = self.syntagme_measurements(mine, src)
            index = {s.name: i for i, s in enumerate(segments)}
            rec = np.zeros((len(local) + len(meas), RECORD_WIDTH), dtype=np.float64)
            for k, st in enumerate(local):
                rec[k, 1:1 + len(SEGMENT_RECORD)] = [st[f] for f in SEGMENT_RECORD]
            for k, m in enumerate(meas, start=len(local)):
                rec[k, 0] = 1.0
                rec[k, 1:1 + len(SYNTAGME_RECORD)] = [index[m["segment"]]] + [m[f] for f in SYNTAGME_RECORD[1:]]
        except Exception as e:                                               # noqa: BLE001
            local_error = e
        if local_error is not None:
            try:
                allgather(np.zeros((0, RECORD_WIDTH)), counts, failed=True)
            except Exception:                                                # noqa: BLE001  (PeerFailure: what it is for)
                pass
            raise local_error
        allrec = allgather(rec, counts)
        seg_rec, syn_rec = allrec[allrec[:, 0] == 0.0], allrec[allrec[:, 0] == 1.0]
        texts = [(i, syn["words"]) for i, syns in enumerate(syn_of) for syn in syns]
        if len(seg_rec) != len(segments) or len(texts) != len(syn_rec) or any(int(r[1]) != i for r, (i, _) in zip(syn_rec, texts)):
            raise RuntimeError("gathered records do not match the TextGrids")
        res.segment_stats = [dict(zip(SEGMENT_RECORD, map(float, r[1:1 + len(SEGMENT_RECORD)])), segment=segments[i].name)
                             for i, r in enumerate(seg_rec)]
        for s in res.segment_stats:
            s["wc"] = int(s["wc"])
        res.baselines = self.baselines(res.segment_stats)
        all_meas = [dict(zip(SYNTAGME_RECORD[1:], map(float, r[2:1 + len(SYNTAGME_RECORD)])), segment=segments[i].name, syntagme=t)
                    for r, (i, t) in zip(syn_rec, texts)]
        for m in all_meas:
            m["pause_ms"], m["wc_syn"] = int(m["pause_ms"]), int(m["wc_syn"])
        res.rows = self.rows_from_measurements(all_meas, {seg.name: b for seg, b in zip(segments, res.baselines)})
        if res.rows:
            res.smooth_pitch, res.smooth_rate = self.smooth(res.rows)
            res.bdd_ssml, res.bdd_syntagme_ssml, res.bdd_syntagme_for_synth = self.tables(res.rows, res.smooth_pitch, res.smooth_rate)
        return res

    # -- everything
    def run(self, segments: Sequence[SegmentInput], src: MeasurementSource, gather: Optional[Callable] = None) -> TaggerResult:
        """``gather``: optional hook applied to the per-segment statistics (a list of dicts) that
        returns the statistics of ALL ranks in segment order (see :mod:`.shard`)."""
        segments = sorted(segments, key=lambda s: segment_sort_key(s.name))
        res = TaggerResult()
        if not segments:
            return res
        res.segment_stats = self.segment_statistics(segments, src)
        all_stats = gather(res.segment_stats) if gather else res.segment_stats
        all_base = self.baselines(all_stats)
        index = {s["segment"]: i for i, s in enumerate(all_stats)}
        res.baselines = [all_base[index[s.name]] for s in segments]
        res.rows = self.raw_rows(segments, res.baselines, src)
        if res.rows:
            res.smooth_pitch, res.smooth_rate = self.smooth(res.rows)
            res.bdd_ssml, res.bdd_syntagme_ssml, res.bdd_syntagme_for_synth = self.tables(res.rows, res.smooth_pitch, res.smooth_rate)
        return res
