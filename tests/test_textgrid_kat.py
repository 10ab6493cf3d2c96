"""Known-answer test of the TextGrid text formats against the example PRAAT'S OWN MANUAL publishes (page "TextGrid file formats":
a 2.3 s grid with the interval tiers "Mary" and "John" and the point tier "bell", shown in the long and in the short text format).
The golden G8 pins what the aligner module puts INTO its TextGrids; the serialisation on the golden side of G8 is a stand-in for
textgrid==1.6.1 written by the same author as ``textgrid_io`` (tests/golden/make_goldens_aligner.py) -- this file is the
independent anchor for the format itself.  (The manual's text is restated from the published page; Praat itself is absent here.)"""
import re

from prosody_control_french_tts_amd import textgrid_io as TG

MANUAL_LONG = '''File type = "ooTextFile"
Object class = "TextGrid"

xmin = 0
xmax = 2.3
tiers? <exists>
size = 3
item []:
    item [1]:
        class = "IntervalTier"
        name = "Mary"
        xmin = 0
        xmax = 2.3
        intervals: size = 1
        intervals [1]:
            xmin = 0
            xmax = 2.3
            text = ""
    item [2]:
        class = "IntervalTier"
        name = "John"
        xmin = 0
        xmax = 2.3
        intervals: size = 1
        intervals [1]:
            xmin = 0
            xmax = 2.3
            text = ""
    item [3]:
        class = "TextTier"
        name = "bell"
        xmin = 0
        xmax = 2.3
        points: size = 0
'''

MANUAL_SHORT = '''File type = "ooTextFile"
Object class = "TextGrid"

0
2.3
<exists>
3
"IntervalTier"
"Mary"
0
2.3
1
0
2.3
""
"IntervalTier"
"John"
0
2.3
1
0
2.3
""
"TextTier"
"bell"
0
2.3
0
'''

# the same grid after annotation, as the manual continues it: intervals with text, a quote doubled inside a mark, one point
ANNOTATED_LONG = '''File type = "ooTextFile"
Object class = "TextGrid"

xmin = 0
xmax = 2.3
tiers? <exists>
size = 3
item []:
    item [1]:
        class = "IntervalTier"
        name = "Mary"
        xmin = 0
        xmax = 2.3
        intervals: size = 3
        intervals [1]:
            xmin = 0
            xmax = 0.7
            text = ""
        intervals [2]:
            xmin = 0.7
            xmax = 1.6
            text = "I said ""hi"" to him"
        intervals [3]:
            xmin = 1.6
            xmax = 2.3
            text = ""
    item [2]:
        class = "IntervalTier"
        name = "John"
        xmin = 0
        xmax = 2.3
        intervals: size = 1
        intervals [1]:
            xmin = 0
            xmax = 2.3
            text = "item [3]: size = 7"
    item [3]:
        class = "TextTier"
        name = "bell"
        xmin = 0
        xmax = 2.3
        points: size = 1
        points [1]:
            number = 1.1
            mark = "ring"
'''


def _read(tmp_path, text, name):
    p = tmp_path / name
    p.write_text(text, encoding="utf-8")
    return TG.read_textgrid(p)


def test_reader_on_the_praat_manual_example_long_and_short(tmp_path):
    for k, text in enumerate((MANUAL_LONG, MANUAL_SHORT)):
        tg = _read(tmp_path, text, f"m{k}.TextGrid")
        assert (tg.min_time, tg.max_time) == (0.0, 2.3)
        assert [t.name for t in tg.tiers] == ["Mary", "John"]            # the point tier is not an IntervalTier: the hot path skips it
        for t in tg.tiers:
            assert t.intervals == [(0.0, 2.3, "")] and (t.tier_min, t.tier_max) == (0.0, 2.3)


def test_reader_handles_marks_that_look_like_syntax_and_points(tmp_path):
    tg = _read(tmp_path, ANNOTATED_LONG, "a.TextGrid")
    assert [t.name for t in tg.tiers] == ["Mary", "John"]
    assert tg.tiers[0].intervals == [(0.0, 0.7, ""), (0.7, 1.6, 'I said "hi" to him'), (1.6, 2.3, "")]
    assert tg.tiers[1].intervals == [(0.0, 2.3, "item [3]: size = 7")]   # brackets and numbers inside a mark stay text


def _tokens(text):
    """The values of a TextGrid text in order (numbers as floats, strings unescaped), whatever the decoration / indentation."""
    body = text.split("\n", 2)[-1]
    body = re.sub(r'"((?:[^"]|"")*)"|\[\d*\]', lambda m: m.group(0) if m.group(0).startswith('"') else " ", body)
    out = []
    for s, ex, num in re.findall(r'"((?:[^"]|"")*)"|(<exists>)|(-?\d+(?:\.\d+)?)', body):
        out.append(ex or (float(num) if num != "" else s.replace('""', '"')))
    return out


def test_writer_emits_the_manual_layout_field_for_field(tmp_path):
    """``format_textgrid`` of the manual's two interval tiers: the same sequence of fields and values as the manual's long format
    (the writer prints Python floats, "0.0" for the manual's "0", and tabs for its spaces; the point tier is outside its scope),
    every line carries the manual's key, and the text reads back to the same grid."""
    tg = _read(tmp_path, ANNOTATED_LONG, "a.TextGrid")
    text = TG.format_textgrid(tg)
    manual_two_tiers = ANNOTATED_LONG.split("    item [3]:")[0].replace("size = 3\nitem []", "size = 2\nitem []")
    assert _tokens(text) == _tokens(manual_two_tiers)
    keys = lambda t: [re.sub(r"\[\d+\]", "[]", ln.split("=")[0].strip()) for ln in t.splitlines() if ln.strip()]
    assert keys(text) == keys(manual_two_tiers)
    assert text.splitlines()[0] == 'File type = "ooTextFile"' and text.splitlines()[1] == 'Object class = "TextGrid"' and text.splitlines()[2] == ""
    assert 'text = "I said ""hi"" to him"' in text                        # quotes doubled, as Praat writes them
    p = tmp_path / "w.TextGrid"
    TG.write_textgrid(tg, p)
    back = TG.read_textgrid(p)
    assert [(t.name, t.intervals) for t in back.tiers] == [(t.name, t.intervals) for t in tg.tiers] and back.max_time == 2.3
