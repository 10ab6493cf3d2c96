#!/bin/bash
# Same-box A/B of libpce.so builds (tools/lab/bin/libpce_<tag>.so): the persistent GEMM per encoder shape
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/abg
set -u
for round in 1 2; do for tag in "$@"; do
  [ -f tools/lab/bin/libpce_$tag.so ] || { echo "no tools/lab/bin/libpce_$tag.so"; exit 1; }
  PCE_LIBRARY=$PWD/tools/lab/bin/libpce_$tag.so timeout 600 python3 bench.py --cpu-clips 0 --streamed-steps 0 --transcribe-steps 0 --medium-steps 0 --steps 4 --warmup 1 > gpurun_out/abg/$tag.$round.json 2>/dev/null
  python3 - "$tag" gpurun_out/abg/$tag.$round.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"{sys.argv[1]:>12}  step {j['ms_per_step']:7.2f}  flat {j['roofline']['achieved']:7.1f} TF/s  " + "  ".join(f"{g['shape']} {g['avg_ms']*1e3:.0f}us" for g in j["gemm_shapes"]))
PY
done; done
