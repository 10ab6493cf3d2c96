#!/usr/bin/env python3
"""Golden G8 (tests/golden/aligner.json): the reference's aligner module run on scripted transcriptions.

Runs only in the build container (needs /root/reference; the reference's Python never travels).  Imports
``Code/Aligners/use_whisper_timestamped.py`` ITSELF with stand-ins for the packages that are absent here and pins
everything that module owns:

  json_to_textgrid (:330-395)            tier contents + the file ``tg.write`` leaves, incl. the +10 ms repair, "[*]" -> " ",
                                         the "..." grid of a wordless result, and the ValueError on overlapping words
  WhisperTranscriber.clean_text (:263)   + remove_spurious_commas (:33-52) with the table POS tagger below
  _is_empty_result / _create_empty_result (:231-261)
  create_matching_textgrids (:425-498)   which files appear and what they hold
  main (:501-728)                        the complete directory / file contract on a directory of seeded WAVs: noise-gated
                                         files, a tiny file, a nearly empty transcription, a transcription that raises, normal
                                         files -- with ``whisper_timestamped.transcribe`` scripted (its results are part of the
                                         fixture) and the real ``scipy.io.wavfile`` for the gate

Stand-ins: ``whisper_timestamped`` (load_model / load_audio / transcribe: scripted), ``spacy`` (table POS tagger whose
table is stored in the fixture), ``textgrid``: a restatement FROM MEMORY of textgrid==1.6.1's Interval / IntervalTier /
TextGrid (sorted insertion that raises on overlaps, ``_fillInTheGaps``, ``write``'s long format with tab indentation).
The serialisation is therefore pinned to that recollection, the tier contents to the reference's own code.
"""
import bisect
import importlib
import json
import logging
import os
import re
import sys
import tempfile
import types
import wave
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(REF / "Code"))

POS = {"le": "DET", "la": "DET", "les": "DET", "un": "DET", "une": "DET", "des": "DET", "de": "ADP", "à": "ADP", "dans": "ADP",
       "et": "CCONJ", "ou": "CCONJ", "que": "SCONJ", "qui": "PRON", "il": "PRON", "elle": "PRON", "ne": "PART",
       ",": "PUNCT", ".": "PUNCT", "?": "PUNCT", "!": "PUNCT"}


# --------------------------------------------------------------------------- textgrid==1.6.1, restated from memory
class Interval:
    def __init__(self, minTime, maxTime, mark):
        if minTime >= maxTime:
            raise ValueError(minTime, maxTime)
        self.minTime, self.maxTime, self.mark = minTime, maxTime, mark

    def overlaps(self, other):
        return other.minTime < self.maxTime and self.minTime < other.maxTime

    def __lt__(self, other):
        if self.overlaps(other):
            raise ValueError(self, other)
        return self.minTime < other.minTime

    def __gt__(self, other):
        if self.overlaps(other):
            raise ValueError(self, other)
        return self.maxTime > other.maxTime

    def __eq__(self, other):
        return self.overlaps(other)


class IntervalTier:
    def __init__(self, name=None, minTime=0.0, maxTime=None):
        self.name, self.minTime, self.maxTime, self.intervals = name, minTime, maxTime, []

    def add(self, minTime, maxTime, mark):
        self.addInterval(Interval(minTime, maxTime, mark))

    def addInterval(self, interval):
        if interval.minTime < self.minTime:
            raise ValueError(self.minTime)
        if self.maxTime and interval.maxTime > self.maxTime:
            raise ValueError(self.maxTime)
        i = bisect.bisect_left(self.intervals, interval)
        if i != len(self.intervals) and self.intervals[i] == interval:
            raise ValueError(self.intervals[i])
        self.intervals.insert(i, interval)

    def __getitem__(self, i):
        return self.intervals[i]

    def _fillInTheGaps(self, null):
        prev_t, output = self.minTime, []
        for interval in self.intervals:
            if prev_t < interval.minTime:
                output.append(Interval(prev_t, interval.minTime, null))
            output.append(interval)
            prev_t = interval.maxTime
        if self.maxTime is not None and prev_t < self.maxTime:
            output.append(Interval(prev_t, self.maxTime, null))
        return output


class TextGrid:
    def __init__(self, name=None, minTime=0.0, maxTime=None, strict=True):
        self.name, self.minTime, self.maxTime, self.tiers, self.strict = name, minTime, maxTime, [], strict

    def __len__(self):
        return len(self.tiers)

    def append(self, tier):
        if self.maxTime is not None and tier.maxTime is not None and tier.maxTime > self.maxTime:
            raise ValueError(self.maxTime)
        self.tiers.append(tier)

    @classmethod
    def fromFile(cls, f, name=None):
        tg = cls(name=name)
        text = open(f, encoding="utf-8").read()
        tg.minTime = float(re.search(r"xmin = (\S+)", text).group(1))
        tg.maxTime = float(re.search(r"xmax = (\S+)", text).group(1))
        return tg

    def write(self, f, null=""):
        out = ['File type = "ooTextFile"', 'Object class = "TextGrid"\n', "xmin = {0}".format(self.minTime)]
        maxT = self.maxTime
        if not maxT:
            maxT = max([t.maxTime if t.maxTime else t[-1].maxTime for t in self.tiers])
        out += ["xmax = {0}".format(maxT), "tiers? <exists>", "size = {0}".format(len(self)), "item []:"]
        for i, tier in enumerate(self.tiers, 1):
            out += ["\titem [{0}]:".format(i), '\t\tclass = "IntervalTier"', '\t\tname = "{0}"'.format(tier.name),
                    "\t\txmin = {0}".format(tier.minTime), "\t\txmax = {0}".format(maxT)]
            output = tier._fillInTheGaps(null)
            out.append("\t\tintervals: size = {0}".format(len(output)))
            for j, interval in enumerate(output, 1):
                out += ["\t\t\tintervals [{0}]:".format(j), "\t\t\t\txmin = {0}".format(interval.minTime),
                        "\t\t\t\txmax = {0}".format(interval.maxTime), '\t\t\t\ttext = "{0}"'.format(interval.mark.replace('"', '""'))]
        with open(f, "w", encoding="utf-8") as sink:
            sink.write("\n".join(out) + "\n")


class Tok:
    def __init__(self, text, ws):
        self.text, self.text_with_ws, self.pos_ = text, text + ws, POS.get(text.lower(), "NOUN")


def nlp(text):
    toks = [Tok(m.group(1), m.group(2)) for m in re.finditer(r"(\[\*\]|\w+|[^\w\s])(\s*)", text)]
    return toks or [Tok("", "")]


SCRIPT = {"results": {}, "current": None}


def install_stubs():
    tg = types.ModuleType("textgrid")
    tg.Interval, tg.IntervalTier, tg.TextGrid = Interval, IntervalTier, TextGrid
    sys.modules["textgrid"] = tg
    spacy = types.ModuleType("spacy")
    spacy.load = lambda *a, **k: nlp
    sys.modules["spacy"] = spacy
    wt = types.ModuleType("whisper_timestamped")
    wt.load_model = lambda size, device=None: ("model", size)

    def load_audio(path):
        SCRIPT["current"] = os.path.basename(path)
        return np.zeros(16000, dtype=np.float32)

    def transcribe(model, audio, **cfg):
        r = SCRIPT["results"][SCRIPT["current"]]
        if isinstance(r, str):
            raise RuntimeError(r)
        return json.loads(json.dumps(r))
    wt.load_audio, wt.transcribe = load_audio, transcribe
    sys.modules["whisper_timestamped"] = wt


def word(text, a, b, conf=0.9):
    return {"text": text, "start": a, "end": b, "confidence": conf}


def result(words_per_segment, language="fr"):
    segs = []
    for k, ws in enumerate(words_per_segment):
        segs.append({"id": k, "seek": 0, "start": ws[0]["start"] if ws else 0.0, "end": ws[-1]["end"] if ws else 2.5,
                     "text": " " + " ".join(w["text"] for w in ws), "tokens": [50364 + k, 10 + k], "temperature": 0.0, "avg_logprob": -0.25,
                     "compression_ratio": 1.1, "no_speech_prob": 0.01, "confidence": 0.88, "words": ws})
    return {"text": "".join(s["text"] for s in segs), "segments": segs, "language": language}


def write_wav(path, pcm, rate=16000):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(rate); w.writeframes(np.asarray(pcm, dtype="<i2").tobytes())


def wav_inputs():
    """name -> (rate, int16 samples): what the test re-creates byte for byte (seeded)."""
    rng = np.random.default_rng(808)
    speech = lambda n: (rng.standard_normal(n) * 3000).astype(np.int16)
    return {
        "segment_ph1": (16000, speech(8000)),
        "segment_ph2": (16000, np.zeros(8000, dtype=np.int16)),                      # silence: gated inline
        "segment_ph3": (44100, (rng.standard_normal(9000) * 40).astype(np.int16)),   # RMS < 100: gated inline
        "segment_ph4": (16000, speech(6400)),                                        # transcription raises
        "segment_ph5": (16000, speech(7000)),                                        # two words only: "very little content"
        "segment_ph6": (16000, speech(400)),                                         # 844-byte file: gated by _check_audio_content (size)
        "segment_ph7": (22050, speech(9000)),                                        # overlapping words: textgrid raises
        "segment_ph10": (16000, speech(5000)),
    }


def scripted_results():
    return {
        "segment_ph1.wav": result([[word("Bonjour", 0.12, 0.48), word("le,", 0.48, 0.6), word("monde", 0.75, 1.2), word("[*]", 1.2, 1.5),
                                     word("et", 1.5, 1.5), word("voilà.", 1.62, 2.0, 0.5)],
                                    [word("Une", 2.4, 2.7), word("phrase;", 2.7, 3.3), word("de,", 3.3, 3.4), word("plus", 3.4, 3.9)]]),
        "segment_ph4.wav": "decoder exploded",
        "segment_ph5.wav": result([[word("Oui", 0.1, 0.4), word("non", 0.5, 0.8)]]),
        "segment_ph7.wav": result([[word("Un", 0.0, 0.5), word("deux", 0.4, 0.9), word("trois", 0.9, 1.4)]]),
        "segment_ph10.wav": result([[word("Que", 0.0, 0.3), word(",", 0.3, 0.35), word("dire", 0.35, 0.9), word("de", 0.9, 1.0), word(".", 1.0, 1.02),
                                     word("cela", 1.1, 1.6)]]),
    }


def snapshot(root: Path):
    out = {}
    for p in sorted(root.rglob("*")):
        if p.is_file() and p.suffix != ".wav":
            out[str(p.relative_to(root))] = p.read_text(encoding="utf-8")
    return out


def main():
    install_stubs()
    mod = importlib.import_module("Aligners.use_whisper_timestamped")
    log = logging.getLogger("g8"); log.addHandler(logging.NullHandler()); log.propagate = False
    fixture = {"pos_table": POS}

    # ---- json_to_textgrid
    cases = {
        "plain": result([[word("a", 0.5, 0.9), word("b", 0.9, 1.3)], [word("c", 2.0, 2.25)]]),
        "repair_and_marks": result([[word("un", 0.0, 0.4), word("[*]", 0.4, 0.4), word('de"ux', 0.8, 0.7), word("trois[*]", 1.5, 2)]]),
        "no_words": {"text": "", "segments": [{"id": 0, "start": 0.0, "end": 3.25, "text": "", "words": []}], "language": "fr"},
        "no_segments": {"text": "", "segments": [], "language": "fr"},
        "overlap": result([[word("un", 0.0, 0.5), word("deux", 0.4, 0.9)]]),
        "int_times": result([[word("x", 1, 2), word("y", 3, 4.5)]]),
    }
    jt = {}
    with tempfile.TemporaryDirectory() as td:
        for name, data in cases.items():
            jf = os.path.join(td, name + ".json")
            with open(jf, "w", encoding="utf-8") as f:
                json.dump(data, f, ensure_ascii=False)
            try:
                tg = mod.json_to_textgrid(jf, log)
                out = os.path.join(td, name + ".TextGrid")
                tg.write(out)
                jt[name] = {"input": data, "intervals": [[iv.minTime, iv.maxTime, iv.mark] for iv in tg.tiers[0].intervals], "maxTime": tg.maxTime,
                            "file": open(out, encoding="utf-8").read()}
            except ValueError:
                jt[name] = {"input": data, "raises": "ValueError"}
    fixture["json_to_textgrid"] = jt

    # ---- clean_text, empty-result helpers
    tr = mod.WhisperTranscriber(model_size="medium", device="cpu", language="fr", logger=log)
    texts = ["  Bonjour   le ,  monde .", "que, dire de. cela", "Il vient et [*] repart; vite", "à , la maison", "Que [*] faire", "rien à nettoyer",
             "de,, plus. et. fin", "OU, alors", "un point; virgule; ici", "la [*] pause et, le reste.", ""]
    fixture["clean_text"] = [[t, tr.clean_text(t)] for t in texts]
    fixture["remove_spurious_commas"] = [[t, mod.remove_spurious_commas(t)] for t in texts]
    empties = [result([]), result([[word("a", 0, 1), word("b", 1, 2)]]), result([[word("a", 0, 1), word("b", 1, 2), word("c", 2, 3)]]),
               result([[word("alpha", 0, 1), word("beta", 1, 2), word("gamma", 2, 3)]])]
    fixture["is_empty_result"] = [[e, bool(tr._is_empty_result(e))] for e in empties]
    fixture["empty_result"] = tr._create_empty_result()

    # ---- create_matching_textgrids
    with tempfile.TemporaryDirectory() as td:
        nat, syn = Path(td) / "nat", Path(td) / "syn"
        nat.mkdir(); syn.mkdir()
        g = TextGrid(); t = IntervalTier(name="words"); t.add(0.0, 2.75, "x"); g.append(t); g.maxTime = 2.75; g.write(str(nat / "a.TextGrid"))
        g = TextGrid(); t = IntervalTier(name="words"); t.add(0.0, 1.5, "y"); g.append(t); g.maxTime = 1.5; g.write(str(syn / "b.TextGrid"))
        (nat / "c.TextGrid").write_text("not a textgrid", encoding="utf-8")
        before = snapshot(Path(td))
        mod.create_matching_textgrids(str(nat), str(syn), log)
        fixture["matching"] = {"before": before, "after": snapshot(Path(td))}

    # ---- main(): the directory contract
    SCRIPT["results"] = scripted_results()
    wavs = wav_inputs()
    with tempfile.TemporaryDirectory() as td:
        voice = Path(td) / "Data" / "voice" / "V1"
        audio = voice / "audio"; audio.mkdir(parents=True)
        sib = Path(td) / "Data" / "voice" / "V1_microsoft" / "WhisperTS_textgrid_files"; sib.mkdir(parents=True)
        g = TextGrid(); t = IntervalTier(name="words"); t.add(0.0, 4.5, "seul"); g.append(t); g.maxTime = 4.5; g.write(str(sib / "segment_ph99.TextGrid"))
        for name, (rate, pcm) in wavs.items():
            write_wav(audio / f"{name}.wav", pcm, rate)
        (audio / "notes.txt").write_text("ignored", encoding="utf-8")
        out_path = voice / "WhisperTS_textgrid_files"
        try:
            mod.main(str(audio), str(out_path), whisper_model="medium", device="cpu", logger=log)
            code = None
        except SystemExit as e:
            code = e.code
        np.savez_compressed(OUT / "aligner_wavs.npz", **{f"{k}__{r}": p for k, (r, p) in wavs.items()})     # name__rate -> int16 samples
        fixture["main"] = {"exit": code, "wavs": "aligner_wavs.npz", "scripted": SCRIPT["results"],
                           "sibling_before": "segment_ph99.TextGrid", "files": snapshot(Path(td) / "Data" / "voice")}
    # exit codes of the degenerate calls
    codes = {}
    with tempfile.TemporaryDirectory() as td:
        for label, path in (("missing_dir", os.path.join(td, "nope")), ("no_wavs", td)):
            try:
                mod.main(path, os.path.join(td, "out_" + label), logger=log); codes[label] = None
            except SystemExit as e:
                codes[label] = e.code
    fixture["exit_codes"] = codes
    with open(OUT / "aligner.json", "w", encoding="utf-8") as f:
        json.dump(fixture, f, ensure_ascii=False, indent=1)
    print("wrote aligner.json:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in fixture.items()})
    print("main files:", sorted(fixture["main"]["files"]))


if __name__ == "__main__":
    main()
