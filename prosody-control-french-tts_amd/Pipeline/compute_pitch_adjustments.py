"""Per-syntagme pitch and pitch adjustment (legacy API of
``Code/Pipeline/compute_pitch_adjustments.py``).

``calculate_pitch_segment`` is the numeric part (:167-208): ``extract_part(start, end)`` then
``to_pitch(pitch_floor=f)`` for f in 75, 100, 150, 200 Hz (ceiling 600), the first floor that
yields voiced frames wins, result = geometric mean of the voiced F0.  Here all segments of a
table go through the GPU pitch pipeline together, one batched pass per floor that is still
needed.  The table bookkeeping around it keeps the reference's column names."""
import math
import os
import re
import sys

import numpy as np
import pandas as pd

from .. import hostrules as H
from ..engine import PitchParams, SLICE_OK, get_default_engine, make_slices

PITCH_FLOORS = (75, 100, 150, 200)


def pitch_segments_batch(requests, engine=None):
    """``requests``: iterable of (audio_path, start_s, end_s) -> list of mean pitches (0 where the
    reference returns 0: missing file, invalid times, no voiced frame for any floor)."""
    requests = list(requests)
    out = [0] * len(requests)
    files, by_rate = {}, {}
    for i, (path, a, b) in enumerate(requests):
        if not isinstance(path, str) or path.endswith("nan") or not os.path.exists(path):
            continue
        if path not in files:
            try:
                files[path] = H.decode_wav(path)
            except H.CouldntDecodeError:
                continue
        rate, pcm = files[path]
        if a >= b or a < 0 or b > len(pcm) / rate:           # snd.get_total_duration() = n / rate
            continue
        by_rate.setdefault(rate, []).append(i)
    eng = engine or get_default_engine()
    for rate, idxs in by_rate.items():
        paths = list(dict.fromkeys(requests[i][0] for i in idxs))
        eng.upload([files[p][1] for p in paths], rate)
        todo = {}
        for i in idxs:
            path, a, b = requests[i]
            try:
                lo, hi, x1 = H.praat_part_frames(len(files[path][1]), rate, a, b, preserve_times=False)
            except H.PraatError:
                continue
            todo[i] = (paths.index(path), lo, hi, x1)
        for floor in PITCH_FLOORS:
            if not todo:
                break
            keys = list(todo)
            sl = make_slices([todo[k][0] for k in keys], [todo[k][1] for k in keys], [todo[k][2] for k in keys], [todo[k][3] for k in keys])
            summ = eng.pitch(sl, PitchParams.praat(float(floor), 600.0), want_f0=False)["summary"]
            for k, s in zip(keys, summ):
                if s["status"] == SLICE_OK and s["n_voiced"] > 0:
                    out[k] = math.exp(float(s["mean_log_f0"]))       # statistics.geometric_mean
                    del todo[k]
    return out


def calculate_pitch_segment(audio_path, start_time, end_time):
    return pitch_segments_batch([(audio_path, start_time, end_time)])[0]


def _calculate_pitch_means(df):
    nat = df[df["average_natural_pitch_per_sentence"] != 0]["average_natural_pitch_per_sentence"].head(50)
    syn = df[df["average_synthesized_pitch_per_sentence"] != 0]["average_synthesized_pitch_per_sentence"].head(50)
    return (nat.mean() if len(nat) else None), (syn.mean() if len(syn) else None)


def _complete_audio_paths(df, base_path_natural, base_path_synthesized):
    df["natural_audio_path"] = [os.path.join(base_path_natural, p) for p in df["natural_audio_path"].astype("str")]
    df["synthesized_audio_path"] = [os.path.join(base_path_synthesized, p) for p in df["synthesized_audio_path"].astype("str")]
    return df


_ROW = r"(.+?):\s*(.*)\s*\((\d+\.\d+)-(\d+\.\d+),\s*(\d+\.\d+)\)"


def _extract_time_info(df):
    df["Text"] = df["synthesized"].str.extract(_ROW)[1].fillna(" ")
    for col in ("synthesized", "natural"):
        ex = df[col].str.extract(_ROW)
        df[f"begin_{col}"], df[f"end_{col}"], df[f"duration_{col}"] = ex[2].astype(float), ex[3].astype(float), ex[4].astype(float)
    is_pause = lambda seg: not re.search(r":\s*\w", seg)
    for col in ("synthesized", "natural"):
        p = df[col].apply(is_pause)
        df[f"duree_pause_{col}"] = np.where(p, df[f"duration_{col}"], 0)
        df[f"duration_{col}"] = np.where(p, 0, df[f"duration_{col}"])
    return df


def construct_syntagmes(df):
    cols = ["syntagme", "begin_syntagme_synthesized", "end_syntagme_synthesized", "duration_syntagme_synthesized",
            "begin_syntagme_natural", "end_syntagme_natural", "duration_syntagme_natural",
            "duration_pause_syntagme_synthesized", "duration_pause_syntagme_natural",
            "natural_syntagme_audio_path", "synthesized_syntagme_audio_path"]
    rows = []
    pauses = df[df["Text"].isna() | (df["Text"].str.strip() == "")].index.tolist()
    start = 0
    for end in pauses + [len(df)]:
        if start < end:
            part = df.iloc[start:end]
            rows.append([" ".join(part["Text"].dropna().str.strip().tolist()),
                         part.iloc[0]["begin_synthesized"], part.iloc[-1]["end_synthesized"], part["duration_synthesized"].sum(),
                         part.iloc[0]["begin_natural"], part.iloc[-1]["end_natural"], part["duration_natural"].sum(), 0, 0,
                         part.iloc[0]["natural_audio_path"], part.iloc[0]["synthesized_audio_path"]])
        if end < len(df):
            r = df.iloc[end]
            rows.append(["", r["begin_synthesized"], r["end_synthesized"], 0, r["begin_natural"], r["end_natural"], 0,
                         r["duree_pause_synthesized"], r["duree_pause_natural"], r["natural_audio_path"], r["synthesized_audio_path"]])
        start = end + 1
    return pd.DataFrame(rows, columns=cols)


def calculate_pitch_adjustment(df):
    df["is_pause"] = df["syntagme"].apply(lambda x: not isinstance(x, str) or x.strip() == "")
    nat, syn = df["natural_pitch_syntagme"].to_numpy(float), df["synthesized_pitch_syntagme"].to_numpy(float)
    adj = np.zeros(len(df))
    ok = (~df["is_pause"].to_numpy(bool)) & (syn != 0)
    adj[ok] = (nat[ok] - syn[ok]) / syn[ok] * 100
    adj[np.isinf(adj)] = 0
    df["pitch_adjustment"] = np.clip(adj, -100, 100)
    return df


def calculate_average_pitch(df):
    avg = {"natural": {}, "synthesized": {}}
    for path in df["natural_syntagme_audio_path"].dropna().unique():
        sel = df["natural_syntagme_audio_path"] == path
        nz = df[sel & (df["natural_pitch_syntagme"] != 0)]["natural_pitch_syntagme"]
        v = nz.mean() if not nz.empty else 0
        avg["natural"][path] = v
        df.loc[sel, ["average_natural_pitch", "average_natural_pitch_per_sentence"]] = v
    for path in df["synthesized_syntagme_audio_path"].dropna().unique():
        sel = df["synthesized_syntagme_audio_path"] == path
        nz = df[sel & (df["synthesized_pitch_syntagme"] != 0)]["synthesized_pitch_syntagme"]
        v = nz.mean() if not nz.empty else 0
        avg["synthesized"][path] = v
        df.loc[sel, "average_synthesized_pitch_per_sentence"] = v
    return df, avg


def compute_pitch_adjustments(BDD1_dir, audio_dir, audio_dir_microsoft, transcription_dir, transcription_dir_microsoft, BDD2_dir):
    df = pd.read_csv(BDD1_dir)
    pat = r"(.*?)(?:_segment)"
    df["natural_audio_path"] = df["natural"].str.extract(pat) + ".wav"
    df["synthesized_audio_path"] = df["synthesized"].str.extract(pat) + ".wav"
    df = _complete_audio_paths(df, audio_dir, audio_dir_microsoft)
    df = _extract_time_info(df)
    df["duree_pause_natural"] = df["duree_pause_natural"].fillna(0.01)
    df["duree_pause_synthesized"] = df["duree_pause_synthesized"].fillna(0.01)
    df = construct_syntagmes(df)
    for side in ("natural", "synthesized"):
        df[f"{side}_pitch_syntagme"] = pitch_segments_batch(
            zip(df[f"{side}_syntagme_audio_path"], df[f"begin_syntagme_{side}"], df[f"end_syntagme_{side}"]))
    df.loc[df["syntagme"].str.strip() == "", ["natural_pitch_syntagme", "synthesized_pitch_syntagme"]] = 0
    calculate_pitch_adjustment(df)
    df, _ = calculate_average_pitch(df)
    mean_nat, mean_syn = _calculate_pitch_means(df)
    pause = df["is_pause"].to_numpy(bool)
    syn, nat = df["synthesized_pitch_syntagme"].to_numpy(float), df["natural_pitch_syntagme"].to_numpy(float)
    a_syn = np.where(~pause & (syn != 0) & (mean_syn != 0), syn / mean_syn, 0.0) if mean_syn is not None else syn / mean_syn
    a_nat = np.where(~pause & (nat != 0) & (mean_nat != 0), nat / mean_nat, 0.0) if mean_nat is not None else nat / mean_nat
    df["adjustment_synthesized"], df["adjustment_natural"] = a_syn, a_nat
    with np.errstate(all="ignore"):
        rel = np.where(~pause & (a_nat != 0), a_syn / np.where(a_nat != 0, a_nat, 1.0), 0.0)
    df["relative_pitch_modification"] = rel
    df["pourcentage_relative_pitch_modification"] = np.where(rel != 0, (rel - 1) * 100, 0)
    df.to_csv(BDD2_dir, index=False)


if __name__ == "__main__":
    if len(sys.argv) != 7:
        print("Usage: compute_pitch_adjustments.py <BDD1_dir> <audio_dir> <audio_dir_microsoft> <transcription_dir> "
              "<transcription_dir_microsoft> <BDD2_dir>")
        sys.exit(1)
    compute_pitch_adjustments(*sys.argv[1:7])
