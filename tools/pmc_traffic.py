#!/usr/bin/env python3
"""Reduce rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of `python3 bench.py ...`)
to HBM bytes per launch per kernel, as MI355X_MICROARCH.md "HBM" prescribes for gfx950:
both counters are in KiB; FETCH_SIZE under-counts a wide coalesced read by exactly 2x
(128-B requests tallied at 64 B) so it is doubled; WRITE_SIZE is exact for 16-B stores.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name.split("(")[0]


def reduce(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


fetch, nf = reduce(sys.argv[1], "FETCH_SIZE")
write, nw = reduce(sys.argv[2], "WRITE_SIZE")
out = {"unit": "bytes per launch", "method": "mean over launches; (2*FETCH_SIZE + WRITE_SIZE)*1024, MI355X_MICROARCH.md HBM section",
       "fetch_KiB_raw": fetch, "write_KiB_raw": write, "launches": nf,
       "bytes_per_launch": {k: (2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0 for k in sorted(set(fetch) | set(write)) if k.startswith("k_")}}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out["bytes_per_launch"].items(), key=lambda kv: -kv[1]):
    print(f"{k:18s} fetch {fetch.get(k, 0) / 1024:10.2f} MiB(raw)  write {write.get(k, 0) / 1024:10.2f} MiB   -> {v / 1e6:10.2f} MB/launch")
