#!/usr/bin/env python3
"""Golden G9: ``levenshtein_distance`` of the reference's own module, run here.

Runs only in the build container (needs /root/reference).  ``Code/Aligners/levenshtein_dist_align_txtgrids.py`` imports ``unidecode`` and
``textgrid`` at module level (both absent offline): empty stand-in modules let the import pass; ``levenshtein_distance`` (:43-70) uses
neither, so every number below is the reference function's own output.  What is committed is data (the word pairs and the distances) and
this script; the reference's source never travels.

Pairs: French words with accents / ligatures / punctuation, the empty string on either side, equal strings, the ``len(s1) < len(s2)`` swap
(:54-55), merged multi-word strings as the merge loop builds them (``w1 + " " + words1[i_]``, :134), one-character strings, strings longer
than a 64-lane stripe, and seeded random strings over a small alphabet (many ties).
"""
import json
import sys
import types
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
for name in ("unidecode", "textgrid"):
    m = types.ModuleType(name)
    m.unidecode = lambda s: s
    sys.modules.setdefault(name, m)
sys.path.insert(0, str(REF / "Code" / "Aligners"))
import levenshtein_dist_align_txtgrids as L          # noqa: E402

WORDS = ["bonjour", "Bonjour,", "le", "la", "les", "monde", "monde.", "voilà", "voila", "une", "phrase", "phrases.", "très", "tres", "longue",
         "long", "ici?", "ici", "oui", "de", "du", "mer!", "mère", "encore", "un", "mot", "mots", "cœur", "coeur", "œuvre", "Noël", "noel", "garçon",
         "garcon", "aujourd'hui", "aujourd’hui", "où", "ou", "été", "était", "l'élève", "élèves", "façade", "naïve", "à", "a", "ça", "sa", "", " ",
         "[*]", "...", "hôpital", "hopital", "être", "etre", "sûr", "sur", "août", "aout", "peut-être", "peut", "quelqu'un", "quelque", "chose"]


def main():
    rng = np.random.default_rng(20261003)
    pairs = []
    for i, a in enumerate(WORDS):                                   # every word against a spread of the others (incl. itself)
        for k in (0, 1, 7, 13):
            pairs.append((a, WORDS[(i + k * 5) % len(WORDS)]))
    for _ in range(60):                                             # merged strings of the merge loop (:134-136)
        n1, n2 = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        pairs.append((" ".join(rng.choice(WORDS, n1)), " ".join(rng.choice(WORDS, n2))))
    for n1, n2 in [(0, 0), (0, 9), (9, 0), (1, 1), (1, 70), (63, 64), (64, 64), (65, 63), (64, 129), (130, 131), (200, 3), (3, 200), (257, 300), (700, 650)]:
        for alpha in ("ab", "abcdé "):
            a = "".join(rng.choice(list(alpha), n1)) if n1 else ""
            b = "".join(rng.choice(list(alpha), n2)) if n2 else ""
            pairs.append((a, b))
    big = "".join(rng.choice(list("abcdefghij"), 1500))             # longer than any workgroup stripe; b = a with edits
    edited = list(big)
    for p in sorted(rng.choice(len(edited), 90, replace=False), reverse=True):
        r = rng.random()
        if r < 0.34:
            del edited[p]
        elif r < 0.67:
            edited.insert(p, "z")
        else:
            edited[p] = "y"
    pairs.append((big, "".join(edited)))
    pairs.append(("\U0001F600abc", "abc\U0001F600"))                # code points above the BMP are ONE character in Python
    cases = [{"s1": a, "s2": b, "distance": int(L.levenshtein_distance(a, b))} for a, b in pairs]
    assert all(c["distance"] == L.levenshtein_distance(c["s2"], c["s1"]) for c in cases[:100])
    with open(OUT / "levenshtein.json", "w", encoding="utf-8") as f:
        json.dump({"source": "Code/Aligners/levenshtein_dist_align_txtgrids.py:43-70 (levenshtein_distance), imported and run",
                   "cases": cases}, f, ensure_ascii=False, indent=0)
    print("wrote levenshtein.json:", len(cases), "pairs; max distance", max(c["distance"] for c in cases))


if __name__ == "__main__":
    main()
