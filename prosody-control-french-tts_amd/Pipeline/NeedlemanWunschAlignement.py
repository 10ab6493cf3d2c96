"""Word-level global alignment natural <-> synthetic (legacy API of
``Code/Pipeline/NeedlemanWunschAlignement.py``).

Needleman-Wunsch with match +1 / mismatch -1 / gap -1 over the ``Text`` column, comparison
after the reference's token normalisation (the five characters ß ? . , ; are stripped and the
token is lower-cased *only when one of them was present*, :43-47), trace-back preference
diagonal > up > left (:69-80), output line format of :104.

``needleman_wunsch_alignement`` (the directory-level entry point of the legacy pipeline) aligns every
file pair of the run in ONE launch of the engine's batched integer kernel (``pce_nw_align``); the host
only normalises tokens to integer ids and formats the result.  ``needleman_wunsch`` is the host
restatement of a single alignment, kept for the golden-vector tests (integer DP, numpy row fill)."""
import csv
import os
import sys

import numpy as np

_STRIP = ("ß", "?", ".", ",", ";")
_GAP = ("-", "", 0, 0, 0)


def _read_segments_from_csv2(file_path):
    with open(file_path, newline="", encoding="utf-8") as f:
        return [(r["PhraseID"], r["Text"], float(r["Start"]), float(r["End"]), float(r["Duration"])) for r in csv.DictReader(f)]


def _norm(text):
    """Sequentially remove each special character; lower-case at the moment one is found."""
    for ch in _STRIP:
        if len(text) and ch in text:
            text = text.replace(ch, "").lower()
    return text


def needleman_wunsch(seq1, seq2, match_score=1, mismatch_score=-1, gap_penalty=-1):
    m, n = len(seq1), len(seq2)
    k1 = [_norm(w[1]) for w in seq1]
    k2 = [_norm(w[1]) for w in seq2]
    sub = np.where(np.array(k1, dtype=object)[:, None] == np.array(k2, dtype=object)[None, :], match_score, mismatch_score) \
        if m and n else np.zeros((m, n), dtype=np.int64)
    score = np.zeros((m + 1, n + 1), dtype=np.int64)
    score[:, 0] = np.arange(m + 1) * gap_penalty
    score[0, :] = np.arange(n + 1) * gap_penalty
    for i in range(1, m + 1):
        diag = score[i - 1, :-1] + sub[i - 1]
        up = score[i - 1, 1:] + gap_penalty
        best = np.maximum(diag, up)
        row = score[i]
        for j in range(1, n + 1):                    # the left dependency is sequential
            v = row[j - 1] + gap_penalty
            row[j] = best[j - 1] if best[j - 1] >= v else v
    a1, a2 = [], []
    i, j = m, n
    while i > 0 or j > 0:
        # the reference indexes seq[i-1] even at i == 0 (Python wraps to the last element)
        W1, W2 = seq1[i - 1], seq2[j - 1]
        if i > 0 and j > 0 and score[i, j] == score[i - 1, j - 1] + (match_score if _norm(W1[1]) == _norm(W2[1]) else mismatch_score):
            a1.append(W1); a2.append(W2); i -= 1; j -= 1
        elif i > 0 and score[i, j] == score[i - 1, j] + gap_penalty:
            a1.append(W1); a2.append(_GAP); i -= 1
        else:
            a1.append(_GAP); a2.append(W2); j -= 1
    return a1[::-1], a2[::-1]


def needleman_wunsch_batch(pairs, engine=None, match_score=1, mismatch_score=-1, gap_penalty=-1):
    """All (seq1, seq2) pairs in one ``pce_nw_align`` launch -> [(aligned1, aligned2), ...]."""
    if engine is None:
        from ..engine import get_default_engine
        engine = get_default_engine()
    vocab = {}

    def ids(seq):
        return [vocab.setdefault(_norm(w[1]), len(vocab)) for w in seq]

    res = engine.nw_align([(ids(a), ids(b)) for a, b in pairs], match_score, mismatch_score, gap_penalty)
    return [([a[i] if i >= 0 else _GAP for i in ii], [b[j] if j >= 0 else _GAP for j in jj]) for (a, b), (ii, jj) in zip(pairs, res)]


def format_alignment(aligned):
    return "".join(f"{d1[0]}: {d1[1]} ({d1[2]}-{d1[3]}, {d1[4]}) || {d2[0]}: {d2[1]} ({d2[2]}-{d2[3]}, {d2[4]})\n"
                   for d1, d2 in zip(*aligned))


def needleman_wunsch_alignement(in_needleman_wunsch_microsoft, in_needleman_wunsch, AligNeedlemanWhunch_out, engine=None):
    datatype = "Segments"
    d_syn = os.path.join(in_needleman_wunsch_microsoft, datatype)
    d_nat = os.path.join(in_needleman_wunsch, datatype)
    out = os.path.join(AligNeedlemanWhunch_out, datatype)
    os.makedirs(out, exist_ok=True)
    names = sorted(set(os.listdir(d_syn)).intersection(os.listdir(d_nat)))
    pairs = [(_read_segments_from_csv2(os.path.join(d_syn, name)), _read_segments_from_csv2(os.path.join(d_nat, name))) for name in names]
    for name, aligned in zip(names, needleman_wunsch_batch(pairs, engine) if pairs else []):
        with open(os.path.join(out, f"aligned_{name[:-4]}.txt"), "w", encoding="utf-8") as f:
            f.write(format_alignment(aligned))


def main():
    if len(sys.argv) != 4:
        print("Usage: python NeedlemanWunschAlignement.py", "<in_needleman_wunsch_microsoft>", "<in_needleman_wunsch>",
              "<AligNeedlemanWhunch_out>")
        sys.exit(1)
    needleman_wunsch_alignement(*sys.argv[1:4])


if __name__ == "__main__":
    main()
