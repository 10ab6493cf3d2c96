"""Host logic of free-running transcription (Aligners/decoding.py): the vocabulary mask and whisper.transcribe's seek
rules on hand-made token sequences (timestamp_begin = 1000, one timestamp step = 20 ms = 2 mel frames)."""
import numpy as np

from prosody_control_french_tts_amd.Aligners import decoding as DEC

TS = 1000


def test_vocab_mask_bits():
    m = DEC.vocab_mask(20, [3, 4], [5, 19], 7)
    assert m.dtype == np.uint8 and m[3] == 1 and m[4] == 1 and m[7] == 1 and m[5] == 2 and m[19] == 2 and m[0] == 0


def test_segments_and_seek_rules():
    # <|0.00|> a b <|1.00|><|1.00|> c <|2.50|><|2.50|> d   -> two closed segments, seek moves to 2.50 s = 250 frames
    toks = [TS, 1, 2, TS + 50, TS + 50, 3, TS + 125, TS + 125, 4]
    segs, seek = DEC.segments_and_seek(toks, TS, seek=0)
    assert seek == 250 and [(s["start"], s["end"]) for s in segs] == [(0.0, 1.0), (1.0, 2.5)]
    assert segs[0]["tokens"] == [TS, 1, 2, TS + 50] and segs[1]["tokens"] == [TS + 50, 3, TS + 125]
    # ... ending on a single timestamp: the whole window is consumed, the open tail is a segment too
    toks = [TS, 1, TS + 50, TS + 50, 3, TS + 100]
    segs, seek = DEC.segments_and_seek(toks, TS, seek=3000)
    assert seek == 6000 and [(round(s["start"], 2), round(s["end"], 2)) for s in segs] == [(30.0, 31.0), (31.0, 32.0)]
    # no consecutive timestamps: one segment up to the last timestamp, window consumed
    segs, seek = DEC.segments_and_seek([TS, 1, 2, TS + 40], TS, seek=100, segment_size=720)
    assert seek == 820 and len(segs) == 1 and abs(segs[0]["start"] - 1.0) < 1e-12 and abs(segs[0]["end"] - 1.8) < 1e-12
    # no usable timestamp at all: the nominal window duration
    segs, seek = DEC.segments_and_seek([1, 2, 3], TS, seek=0, segment_size=500)
    assert seek == 500 and segs[0]["end"] == 5.0
