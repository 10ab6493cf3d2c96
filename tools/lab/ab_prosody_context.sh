#!/bin/bash
# Same-box A/B: the prosody leg of the C3 step behind the Whisper leg on one context (0) or beside it in a second context (1).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/pctx
for r in 1 2; do for m in 0 1; do
  timeout 600 python3 bench.py --prosody-context $m --cpu-clips 0 --streamed-steps 0 --transcribe-steps 0 --medium-steps 0 --framing-clips 0 --steps 5 --warmup 2 > gpurun_out/pctx/$m.$r.json 2> gpurun_out/pctx/$m.$r.err
  python3 - $m $r <<'PY'
import json, sys
m, r = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f"gpurun_out/pctx/{m}.{r}.json").read().strip().splitlines()[-1]); k = {x["kernel"]: x for x in d["kernels"]}
    print(f"prosody_context={m} rep {r}: step {d['ms_per_step']:.2f} ms (median interval {d['step_ms_spread']['median']:.2f})  encoder {k['whisper_encoder']['ms_per_step']:.2f}  align {k['whisper_align']['ms_per_step']:.2f}  "
          f"k_pitch_frames {k['k_pitch_frames']['ms_per_step']:.2f}  k_pitch_refine {k['k_pitch_refine']['ms_per_step']:.2f}  k_stft_raw {k['k_stft_raw']['ms_per_step']:.2f}")
except Exception as e:
    print("failed", m, r, e, open(f"gpurun_out/pctx/{m}.{r}.err").read()[-600:])
PY
done; done
