#!/bin/bash
# Round profile on the GPU box: bench lines, rocprofv3 kernel statistics of the SAME commands, and the PMC passes
# (each counter group in its own run, only --kernel-trace beside it).  usage: tools/profile_round.sh <out dir under gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r05}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python3 $R/bench.py --workload c2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
C3="--steps 3 --warmup 1 --cpu-clips 0 --streamed-steps 0 --transcribe-steps 0 --medium-steps 0 --framing-clips 0"
C2="--workload c2 --steps 5 --warmup 2 --cpu-clips 0 --streamed-steps 0 --framing-clips 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -- python3 $R/bench.py $C3 > $OUT/stats_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c2 -- python3 $R/bench.py $C2 > $OUT/stats_c2.log 2>&1
P3="--steps 2 --warmup 1 --cpu-clips 0 --streamed-steps 0 --transcribe-steps 0 --medium-steps 0 --framing-clips 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c3 -- python3 $R/bench.py $P3 > $OUT/pmc_fetch_c3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c3 -- python3 $R/bench.py $P3 > $OUT/pmc_write_c3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma_c3 -- python3 $R/bench.py $P3 > $OUT/pmc_mfma_c3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c2 -- python3 $R/bench.py $C2 > $OUT/pmc_fetch_c2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c2 -- python3 $R/bench.py $C2 > $OUT/pmc_write_c2.log 2>&1
# the device-resident decoding loop: kernel + memory-copy trace of three 32-step loops, summarised (no --pmc beside it)
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/decode_trace -- python3 $R/tools/decode_trace.py > $OUT/decode_trace.log 2>&1
python3 $R/tools/decode_trace.py --summarise $OUT/decode_trace > $OUT/decode_trace_summary.txt 2>&1
python3 $R/tools/energy_rate.py 256 1250 > $OUT/energy_rate.txt 2>&1
# keep what is judged small: statistics + counter tables (the traces themselves stay in gpurun_out)
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
ls -R $OUT | head -60
