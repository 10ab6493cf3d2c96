#!/bin/bash
# Same-box A/B of libpce builds (tools/lab/bin/libpce_<tag>.so; "product" = the tree's own library) over the C3 bench line: step, log-mel, STFT,
# the free-running decoding step and the alignment.  usage: tools/lab/ab_r06.sh tagA tagB ...   (each twice, interleaved)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/ab6
ARGS=${AB_BENCH_ARGS:---cpu-clips 0 --streamed-steps 0 --medium-steps 0 --framing-clips 0 --steps 4}
for round in 1 2; do for tag in "$@"; do
  lib=$PWD/tools/lab/bin/libpce_$tag.so; [ "$tag" = product ] && lib=$PWD/prosody-control-french-tts_amd/libpce.so
  [ -f $lib ] || { echo "no $lib"; exit 1; }
  PCE_LIBRARY=$lib timeout 900 python3 bench.py $ARGS > gpurun_out/ab6/$tag.$round.json 2> gpurun_out/ab6/$tag.$round.err
  python3 - "$tag" gpurun_out/ab6/$tag.$round.json <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
d = json.loads(open(path).read().strip().splitlines()[-1])
k = {x["kernel"]: x for x in d["kernels"]}
g = lambda n: (k[n]["ms_per_step"] if n in k else float("nan"))
t = d.get("transcribe") or {}
print(f"{tag:>10}  step {d['ms_per_step']:8.3f} ms  gemm_flat {d['roofline']['achieved']:.0f} TF/s  logmel {g('k_logmel_frames'):.3f} norm {g('k_logmel_norm'):.3f}  stft {g('k_stft_raw'):.3f}+{g('k_stft_norm'):.3f}  "
      f"align {g('whisper_align'):.2f}  attn {g('k_attention_lean'):.2f}  ln {g('k_add_layernorm'):.2f}  |  window {t.get('window_ms', float('nan')):.1f} ms  inc step {t.get('ms_per_incremental_step', float('nan')):.3f} ms", flush=True)
PY
done; done
