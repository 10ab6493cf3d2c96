#!/bin/bash
# What the driver runs at round end, in one call: smoke(), the default bench line (wall time printed), a summary of the line.
# usage (GPU box): bash tools/driver_like.sh  -> gpurun_out/bench_default.json
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SECONDS=0
python bench.py "$@" > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench.py $* : rc $? in $SECONDS s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print(d["metric"], "|", round(d["value"]), d["unit"], "| n_gpus", d["n_gpus"], "steps", d["steps"], "warmup", d["warmup"], "| ms_per_step", round(d["ms_per_step"], 2), "|", d["dtype"], d["scaling"], d["vs_baseline"])
r = d["roofline"]
print("roofline", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic")})
print("roofline.framing_hbm", r.get("framing_hbm"))
c = d.get("cpu_baseline")
print("cpu_baseline", {k: c[k] for k in ("value", "unit", "cores", "kind")} if c else None)
t = d.get("transcribe")
if t: print("transcribe", round(t["window_ms"], 1), "ms per window,", round(t["ms_per_incremental_step"], 3), "ms per incremental step, frac", round(t["roofline"]["frac"], 3))
PY
