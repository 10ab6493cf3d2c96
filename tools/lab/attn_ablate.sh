#!/bin/bash
# Same-box ablation of k_attention_lean: tools/lab/bin/libpce_<tag>.so variants (built by hand, results of the ablated kernels are WRONG: timing only)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/abl
set -u
for round in 1 2; do for tag in "$@"; do
  [ -f tools/lab/bin/libpce_$tag.so ] || { echo "no tools/lab/bin/libpce_$tag.so"; exit 1; }
  PCE_LIBRARY=$PWD/tools/lab/bin/libpce_$tag.so timeout 600 python3 bench.py --cpu-clips 0 --streamed-steps 0 --transcribe-steps 0 --medium-steps 0 --steps 2 --warmup 1 > gpurun_out/abl/$tag.$round.json 2>/dev/null
  python3 - "$tag" gpurun_out/abl/$tag.$round.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); k = {x["kernel"]: x for x in j["kernels"]}
a = k["k_attention_lean"]; print(f"{sys.argv[1]:>10}  step {j['ms_per_step']:7.2f}  attention {a['ms_per_step']:6.2f} ms/step  ({a['avg_ms']*1e3:7.1f} us per launch, 36 launches)")
PY
done; done
