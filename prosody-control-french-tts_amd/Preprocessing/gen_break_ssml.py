"""``extract_words_and_pauses`` of ``Code/Preprocessing/gen_break_ssml.py:12-42``: the reader
side of the alignment -> prosody wire format.  (The rest of that module is legacy
break-only SSML generation and out of scope.)"""
from ..tagger import INITIAL_PAUSE_THRESHOLD_MS as INITIAL_PAUSE_THRESHOLD, words_and_pauses
from ..textgrid_io import read_textgrid

MIN_PAUSE_THRESHOLD = 150


def extract_words_and_pauses(textgrid_file):
    """[(kind, token, duration_ms)] from the first tier of a TextGrid file."""
    tg = read_textgrid(textgrid_file)
    return words_and_pauses(tg.tiers[0].intervals)
