"""SSML fragment formatting of the legacy pipeline (``Code/Pipeline/Get_Wav.py:8-66``).

Only the string-building half is mirrored: ``get_wav`` itself calls the Azure synthesiser, which
is out of scope.  ``create_ssml_fragment`` turns one syntagme's adjustment
percentages and its natural pause into either ``<break time='Nms'/>`` (empty text) or a
``<prosody ...>`` element, with the reference's compressions (|rate|^0.8 capped at +2, sqrt of
|pitch|), its pause rule (ms / 3, scaled by ``pause_coef``, clamped to [min_pause, max_pause],
``max_pause`` when missing or zero) and its breath hints after ``, ß ! ?``.

The three pause parameters are module globals which the reference's ``get_wav`` sets before
the first fragment is built (:91-94); here they carry those values from import time."""
import math
import re

pause_coef = 1.0
max_pause = 500
min_pause = 1

_CTRL = re.compile(r"[\x00-\x1F\x7F]")


def _signed_power(v, p):
    v = float(v)
    return math.copysign(abs(v) ** p, v) if v != 0 and not math.isnan(v) else (v if math.isnan(v) else 0.0 * v)


def _fmt(v):
    return "+0%" if (v == 0 or v == -math.inf) else f"{v:+.2f}%"


def create_ssml_fragment(text, pitch_adj, rate_adj, loudness_adj, duration_pause_syntagme_natural, voice, style, styledegree):
    is_pause = str(text).strip() == ""
    if not is_pause:
        rate = min(2, _signed_power(rate_adj, 0.80))
        pitch = _signed_power(pitch_adj, 0.5)
        pitch_mod, rate_mod, loud_mod = _fmt(pitch), _fmt(rate), _fmt(float(loudness_adj))
    ms = float(duration_pause_syntagme_natural) * 1000 / 3
    if math.isnan(ms) or ms == 0:
        pause = max_pause
    else:
        ms *= pause_coef
        pause = int(min(max(ms, min_pause), max_pause)) if not (ms > max_pause) else int(max_pause)
    if is_pause:
        return f"<break time='{pause}ms'/>"
    clean = _CTRL.sub("", str(text)).replace("&", "&amp;").replace("<", "&lt;").replace(">", "&gt;")
    if clean.endswith((",", "ß")):
        clean = clean[:-1] + ", h"
    elif clean.endswith("!"):
        clean = clean[:-1] + "! h"
    elif clean.endswith("?"):
        clean = clean[:-1] + "? h"
    body = f"<prosody pitch='{pitch_mod}' rate='{rate_mod}' volume='{loud_mod}'>{clean}</prosody>"
    return f"<mstts:express-as style='{style}' styledegree='{styledegree}'>{body}</mstts:express-as>" if style else body
