#!/bin/bash
# Same-box A/B of one environment switch: usage  ab_env.sh VAR value_a value_b [reps]   (C3 step without the extras, alternating)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; V=$1; A=$2; B=$3; N=${4:-3}; mkdir -p gpurun_out/ab_env
for r in $(seq 1 $N); do for val in $A $B; do
  env $V=$val timeout 600 python3 bench.py --cpu-clips 0 --streamed-steps 0 --transcribe-steps 0 --medium-steps 0 --framing-clips 0 --steps 5 --warmup 2 > gpurun_out/ab_env/$V.$val.$r.json 2>/dev/null
  python3 - $V $val $r <<'PY'
import json, sys
V, val, r = sys.argv[1:4]
d = json.loads(open(f"gpurun_out/ab_env/{V}.{val}.{r}.json").read().strip().splitlines()[-1]); k = {x["kernel"]: x for x in d["kernels"]}
print(f"{V}={val} rep {r}: step {d['ms_per_step']:.2f}  flat {k['k_gemm_flat']['ms_per_step']:.2f}  attention {k['k_attention_lean']['ms_per_step']:.2f}  align {k['whisper_align']['ms_per_step']:.2f}  " +
      "  ".join(f"{g['shape']} {g['avg_ms']*1e3:.0f}us" for g in d["gemm_shapes"]))
PY
done; done
