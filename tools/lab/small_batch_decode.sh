#!/bin/bash
# free-running decoding step at SMALL batch sizes (what the aligner runs: the windows of a few files) per library build: the price of fewer leaves
# (fewer workgroups per clip below 128 clips).  usage: tools/lab/small_batch_decode.sh tagA tagB ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for n in 1 4 16 64; do for tag in "$@"; do
  lib=$PWD/tools/lab/bin/libpce_$tag.so; [ "$tag" = product ] && lib=$PWD/prosody-control-french-tts_amd/libpce.so
  echo -n "$tag n=$n: "; PCE_LIBRARY=$lib timeout 300 python3 tools/decode_rate.py $n 32 2>&1 | grep "device loop"
done; done
