#!/bin/bash
# same-box A/B of bench.py's `transcribe` window: the round-5 decoding kernels against the round-3 / 4 ones (PCE_XATTN_ABSORB=0 PCE_SELF_ROWS=0)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do for form in new old; do
  if [ $form = old ]; then export PCE_XATTN_ABSORB=0 PCE_SELF_ROWS=0; else unset PCE_XATTN_ABSORB PCE_SELF_ROWS; fi
  python3 bench.py --steps 2 --warmup 1 --cpu-clips 0 --streamed-steps 0 --medium-steps 0 --framing-clips 0 > gpurun_out/abt_$form.json 2>/dev/null
  python3 - $form gpurun_out/abt_$form.json <<'PY'
import json, sys
t = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])["transcribe"]
print(f"{sys.argv[1]:>4}  window {t['window_ms']:7.2f} ms  decode loop {t['decode_loop_ms']:7.2f} ms  incremental step {t['ms_per_incremental_step']:.3f} ms  first step {t['first_step_ms_incl_cross_kv_projection']:.2f} ms  alignment after {t['alignment_ms_after_loop']:.2f} ms")
PY
done; done
