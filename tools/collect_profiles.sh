#!/bin/bash
# Copy what is judged from a tools/profile_round.sh output directory into profiles/<round> with a letter prefix, and REDUCE the PMC passes
# (the raw counter tables stay in gpurun_out: only the per-kernel summaries are committed).
# usage: tools/collect_profiles.sh gpurun_out/r05 profiles/r05 a
set -e
D=$1; P=$2; L=$3
mkdir -p $P
cp $D/bench_c3.json $P/${L}_bench_c3.json; cp $D/bench_c2.json $P/${L}_bench_c2.json
cp $D/stats_c3/*/*_kernel_stats.csv $P/${L}_c3_kernel_stats.csv; cp $D/stats_c2/*/*_kernel_stats.csv $P/${L}_c2_kernel_stats.csv
python3 tools/pmc_traffic.py $D/pmc_fetch_c3/*/*_counter_collection.csv $D/pmc_write_c3/*/*_counter_collection.csv $P/pmc_traffic_c3.json | head -6
python3 tools/pmc_traffic.py $D/pmc_fetch_c2/*/*_counter_collection.csv $D/pmc_write_c2/*/*_counter_collection.csv $P/pmc_traffic_c2.json > /dev/null
python3 - $D/pmc_mfma_c3/*/*_counter_collection.csv > $P/${L}_pmc_mfma_c3_summary.txt <<'PY'
import csv, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_[a-z0-9_]+(<\d>)?)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:30]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
print("# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -- python3 bench.py (C3, 2 steps)")
print("# MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), summed over the kernel's launches")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    g = v.get("GRBM_GUI_ACTIVE", 0)
    if g > 0 and v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
        print(f"{k:24s} launches {n[k]:4d}  mfma pipe busy {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * g / 8) * 100:5.1f} %")
PY
cat $P/${L}_pmc_mfma_c3_summary.txt | head -12
cp $D/decode_trace_summary.txt $P/decode_trace_summary.txt 2>/dev/null || true
cp $D/energy_rate.txt $P/energy_rate.txt 2>/dev/null || true
