#!/bin/bash
# Copy what is judged from a tools/profile_round.sh output directory into profiles/<round> with a letter prefix, and reduce the PMC passes.
# usage: tools/collect_profiles.sh gpurun_out/r02f profiles/r02 c
set -e
D=$1; P=$2; L=$3
cp $D/bench_c3.json $P/${L}_bench_c3.json; cp $D/bench_c2.json $P/${L}_bench_c2.json
cp $D/stats_c3/*/*_kernel_stats.csv $P/${L}_c3_kernel_stats.csv; cp $D/stats_c2/*/*_kernel_stats.csv $P/${L}_c2_kernel_stats.csv
for k in fetch_c3 write_c3 fetch_c2 write_c2 mfma_c3; do cp $D/pmc_$k/*/*_counter_collection.csv $P/${L}_pmc_${k}_counter_collection.csv; done
python3 tools/pmc_traffic.py $P/${L}_pmc_fetch_c3_counter_collection.csv $P/${L}_pmc_write_c3_counter_collection.csv $P/pmc_traffic_c3.json | head -6
python3 tools/pmc_traffic.py $P/${L}_pmc_fetch_c2_counter_collection.csv $P/${L}_pmc_write_c2_counter_collection.csv $P/pmc_traffic_c2.json > /dev/null
