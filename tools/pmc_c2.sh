#!/bin/bash
# Instruction-mix counters of the C2 (prosody only) step, one group per pass.  usage: tools/pmc_c2.sh <out dir under gpurun_out>
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/${1:-pmc_c2}; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o -E "SQ_INSTS_VALU[A-Z0-9_]*|SQ_INSTS_[A-Z]*LDS[A-Z_]*|SQ_WAIT_INST_LDS|SQ_ACTIVE_INST_[A-Z]*|SQ_INSTS_SALU|SQ_INST_CYCLES_[A-Z]*|SQ_LDS_[A-Z_]*" | sort -u > $OUT/avail.txt
ARGS="--workload c2 --steps 3 --warmup 1 --cpu-clips 0 --streamed-steps 0 --framing-clips 0"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/bench.py $ARGS > $OUT/g$i.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:30]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, "  ".join(f"{n}={sum(x)/len(x):.4g}" for n, x in sorted(v.items())))
PY
find $OUT -name "*kernel_trace.csv" -delete
