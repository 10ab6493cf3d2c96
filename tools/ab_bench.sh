#!/bin/bash
# Same-box A/B of two (or more) builds of libpce.so: the boxes of the pool differ by up to 6 % on one binary, so two builds are only
# comparable when they run on ONE box within ONE gpurun call, alternating.
#   1. build each variant here (cross-compiled) and keep it:   make -C prosody-control-french-tts_amd/csrc && cp prosody-control-french-tts_amd/libpce.so tools/lab/bin/libpce_<tag>.so
#      (tools/lab/bin is git-ignored but travels with the gpurun snapshot)
#   2. gpurun --timeout 1500 -- 'bash tools/ab_bench.sh <tagA> <tagB> [<tagC> ...]'
# Every tag is benched twice, interleaved (A B ... A B ...); the variant is selected with PCE_LIBRARY (engine.native_library_path): the product's libpce.so is never overwritten.  Extra arguments for bench.py
# go in AB_BENCH_ARGS (default: the C3 step without the CPU baseline, the streamed pass and the transcribe object).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
ARGS=${AB_BENCH_ARGS:---cpu-clips 0 --streamed-steps 0 --transcribe-steps 0 --medium-steps 0 --steps 5}
mkdir -p gpurun_out/ab
for round in 1 2; do
  for tag in "$@"; do
    [ -f tools/lab/bin/libpce_$tag.so ] || { echo "no tools/lab/bin/libpce_$tag.so"; exit 1; }
    PCE_LIBRARY=$PWD/tools/lab/bin/libpce_$tag.so timeout 900 python3 bench.py $ARGS > gpurun_out/ab/$tag.$round.json 2> gpurun_out/ab/$tag.$round.err
    python3 - "$tag" gpurun_out/ab/$tag.$round.json <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
d = json.loads(open(path).read().strip().splitlines()[-1])
shapes = [(g["shape"], round(g["achieved_tflops"])) for g in d.get("gemm_shapes", [])]
print(f"{tag:>8}  {d['ms_per_step']:8.3f} ms/step   {d['roofline']['kernel']} {d['roofline']['achieved']:.1f} {d['roofline']['unit']}   {shapes}")
PY
  done
done
