"""Per-syntagme "RMS dB" and loudness adjustment (legacy API of
``Code/Pipeline/compute_loudness_adjustments.py``).

The reference squares pydub's int16 samples in int16 (numpy wraps modulo 2**16), averages,
takes sqrt(|.|) and 20 log10 (:17-24).  The engine returns the exact integer sum of the
wrapped squares for every slice (``k_energy``), so the float result is bit-identical; all rows of
a table are measured in ONE batched GPU pass instead of one full decode per row."""
import os
import sys

import numpy as np
import pandas as pd

from .. import hostrules as H
from ..engine import get_default_engine, make_slices


def loudness_batch(requests, engine=None):
    """``requests``: iterable of (path, start_s, end_s).  Returns the reference's value per request
    (0 for a missing / NaN path, Code/Pipeline/compute_loudness_adjustments.py:10-11)."""
    requests = list(requests)
    out = [0] * len(requests)
    files, by_rate = {}, {}
    for i, (path, a, b) in enumerate(requests):
        if pd.isna(path) or not os.path.isfile(path):
            continue
        if path not in files:
            files[path] = H.decode_wav(path)
        by_rate.setdefault(files[path][0], []).append(i)
    eng = engine or get_default_engine()
    for rate, idxs in by_rate.items():
        paths = list(dict.fromkeys(requests[i][0] for i in idxs))
        eng.upload([files[p][1] for p in paths], rate)
        cl, b, e = [], [], []
        for i in idxs:
            path, a, z = requests[i]
            lo, hi = H.pydub_slice_frames(len(files[path][1]), rate, a * 1000, z * 1000)     # audio[start*1000:end*1000]
            cl.append(paths.index(path)); b.append(lo); e.append(hi)
        res = eng.energy(make_slices(cl, b, e))
        for i, r in zip(idxs, res):
            out[i] = H.rms_db_from_wrapped(int(r["sum_sq_wrap16"]), int(r["n"]))
    return out


def _calculate_loudness(audio_file_path, start, end):
    return loudness_batch([(audio_file_path, start, end)])[0]


def _calculate_coeff_adjustment(df, engine=None):
    df["is_pause"] = df["syntagme"].apply(lambda x: not isinstance(x, str) or x.strip() == "")
    for side, col in (("natural", "natural_loudness"), ("synthesized", "synthesized_loudness")):
        req, where = [], []
        for idx, row in df.iterrows():
            ok = (isinstance(row["syntagme"], str) and row["syntagme"].strip() != ""
                  and not pd.isna(row[f"begin_syntagme_{side}"]) and not pd.isna(row[f"end_syntagme_{side}"]))
            if ok:
                req.append((row[f"{side}_syntagme_audio_path"], row[f"begin_syntagme_{side}"], row[f"end_syntagme_{side}"]))
                where.append(idx)
        vals = pd.Series(0.0, index=df.index)
        for idx, v in zip(where, loudness_batch(req, engine)):
            vals[idx] = v
        df[col] = vals
    eps = 1e-6
    nat, syn = df["natural_loudness"].to_numpy(float), df["synthesized_loudness"].to_numpy(float)
    adj = np.zeros(len(df))
    ok = (~df["is_pause"].to_numpy(bool)) & (np.abs(syn) > eps)
    with np.errstate(all="ignore"):
        adj[ok] = np.clip((nat[ok] - syn[ok]) / syn[ok] * 100, -20, 20)
    df["loudness_adjustment"] = adj


def calculate_loudness_adjustment(BDD2_dir, BDD3_dir):
    df = pd.read_csv(BDD2_dir)
    _calculate_coeff_adjustment(df)
    df["syntagme"] = df["syntagme"].replace(np.nan, "")
    df.to_csv(BDD3_dir, index=False)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        print("Usage: compute_BDD2_loudness.py", "<BDD2_dir>", "<BDD3_dir>")
        sys.exit(1)
    calculate_loudness_adjustment(sys.argv[1], sys.argv[2])
