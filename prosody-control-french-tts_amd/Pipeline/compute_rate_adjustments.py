"""Speech-rate adjustment table (legacy API of ``Code/Pipeline/compute_rate_adjustments.py``).

Pure scalar bookkeeping on a CSV (words per minute natural vs synthesized, percentage
difference clipped to +-100): host logic, no GPU work.  Same function names, columns and CLI
as the reference (``calculer_metrics`` :27-63, ``calculate_rate`` :65-106)."""
import logging
import os
import sys

import numpy as np
import pandas as pd

log = logging.getLogger(__name__)


def compter_mots(texte):
    return len(texte.split()) if isinstance(texte, str) else 0


def calculate_adjustment_percentage(rate_naturel, rate_synthesized):
    return ((rate_naturel - rate_synthesized) / rate_synthesized) * 100 if rate_synthesized != 0 else 0


def calculer_metrics(df):
    syn = df["syntagme"]
    is_pause = np.array([not isinstance(x, str) or x.strip() == "" for x in syn], dtype=bool)
    df["is_pause"] = is_pause
    df["nombre_de_mots"] = [0 if p else compter_mots(x) for p, x in zip(is_pause, syn)]
    df["duree_natural_minutes"] = df["duration_syntagme_natural"] / 60
    df["duree_synthesized_minutes"] = df["duration_syntagme_synthesized"] / 60
    n = df["nombre_de_mots"].to_numpy(dtype=float)

    def per_minute(minutes):
        m = minutes.to_numpy(dtype=float)
        out = np.zeros(len(df), dtype=float)
        ok = (~is_pause) & (m > 0)
        out[ok] = n[ok] / m[ok]
        return out

    df["rate_natural"] = per_minute(df["duree_natural_minutes"])
    df["rate_synthesized"] = per_minute(df["duree_synthesized_minutes"])
    rn, rs = df["rate_natural"].to_numpy(), df["rate_synthesized"].to_numpy()
    adj = np.zeros(len(df), dtype=float)
    ok = (~is_pause) & (rs != 0)
    adj[ok] = ((rn[ok] - rs[ok]) / rs[ok]) * 100
    adj[~np.isfinite(adj) & ~np.isnan(adj)] = 0
    df["rate_adjustment"] = np.clip(adj, -100, 100)
    return df


def calculate_rate(BDD3_dir, BDD4_dir):
    try:
        df = pd.read_csv(BDD3_dir)
        if df.empty:
            log.error("Le DataFrame est vide")
            sys.exit(1)
        if "syntagme" in df.columns:
            df["syntagme"] = df["syntagme"].fillna("")
        for col in ("duration_syntagme_natural", "duration_syntagme_synthesized"):
            if col in df.columns:
                df[col] = df[col].fillna(0)
        df = calculer_metrics(df)
        df["rate_ajusté"] = df["rate_adjustment"]
        df.to_csv(BDD4_dir, index=False)
    except Exception as e:           # same contract as the reference: any failure ends the step
        log.error(f"Erreur lors du traitement: {e}")
        sys.exit(1)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        print("Usage: compute_BDD3_loudness_rate.py", "<BDD3_dir>", "<BDD4_dir>")
        sys.exit(1)
    if not os.path.exists(sys.argv[1]):
        log.error(f"Le fichier BDD3 n'existe pas: {sys.argv[1]}")
        sys.exit(1)
    out_dir = os.path.dirname(sys.argv[2])
    if out_dir and not os.path.exists(out_dir):
        os.makedirs(out_dir, exist_ok=True)
    calculate_rate(sys.argv[1], sys.argv[2])
