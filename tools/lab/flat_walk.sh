#!/bin/bash
# Same-box sweep of k_gemm_flat's tile walk (PCE_FLAT_SN / PCE_FLAT_SM / PCE_FLAT_STAGGER, read by launch_gemm_flat): per configuration the
# C3 step without the extras (per-shape launch times from the bench line) and ONE rocprofv3 --pmc FETCH_SIZE pass of the same command
# (L2 -> fabric read bytes per launch and shape).  usage: gpurun -- 'bash tools/lab/flat_walk.sh "sn sm stagger" "sn sm stagger" ...'
# (sn: 0 = default choice, -1 = the whole width; sm: 0 = default 16; stagger: -1 = default)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; OUT=$R/gpurun_out/flat_walk; mkdir -p $OUT
ARGS="--cpu-clips 0 --streamed-steps 0 --transcribe-steps 0 --medium-steps 0 --framing-clips 0 --steps 3 --warmup 1"
export TMPDIR=/tmp
for cfg in "$@"; do
  set -- $cfg; sn=$1; sm=$2; st=$3; tag="sn${sn}_sm${sm}_st${st}"
  export PCE_FLAT_SN=$sn PCE_FLAT_SM=$sm PCE_FLAT_STAGGER=$st
  [ "$sn" = "0" ] && unset PCE_FLAT_SN; [ "$sm" = "0" ] && unset PCE_FLAT_SM; [ "$st" = "-1" ] && unset PCE_FLAT_STAGGER
  for rep in 1 2; do timeout 600 python3 bench.py $ARGS > $OUT/$tag.$rep.json 2> $OUT/$tag.$rep.err; done
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$tag -- python3 $R/bench.py $ARGS --steps 1 > $OUT/pmc_$tag.log 2>&1)
  python3 - "$tag" $OUT <<'PY'
import csv, glob, json, sys
tag, out = sys.argv[1], sys.argv[2]
for rep in (1, 2):
    try:
        j = json.loads(open(f"{out}/{tag}.{rep}.json").read().strip().splitlines()[-1])
        print(f"{tag:>20} rep{rep} step {j['ms_per_step']:7.2f}  flat {j['roofline']['achieved']:7.1f} TF/s  " + "  ".join(f"{g['shape']} {g['avg_ms']*1e3:.0f}us" for g in j["gemm_shapes"]))
    except Exception as e:
        print(tag, rep, "bench failed", e)
rows = []
for f in glob.glob(f"{out}/pmc_{tag}/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "k_gemm_flat" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
v = [2.0 * float(r["Counter_Value"]) * 1024 / 1e9 for r in rows]
if v:
    # encoder layers launch qkv, out, fc1, fc2 in turn (48 launches per pass), then 12 cross K|V projections
    per = {n: [] for n in ("qkv", "out", "fc1", "fc2", "xkv")}
    for p0 in range(0, len(v), 60):
        blk = v[p0:p0 + 60]
        for i, x in enumerate(blk[:48]): per[("qkv", "out", "fc1", "fc2")[i % 4]].append(x)
        for x in blk[48:]: per["xkv"].append(x)
    print(f"{tag:>20} fetch GB per launch (2 x FETCH_SIZE): " + "  ".join(f"{k} {sum(x)/len(x):.2f}" for k, x in per.items() if x) + f"   mean {sum(v)/len(v):.2f}")
PY
  find $OUT/pmc_$tag -name "*kernel_trace.csv" -delete
done
