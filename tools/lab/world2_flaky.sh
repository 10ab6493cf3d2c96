#!/bin/bash
# How often does tests/test_gpu_world2.py::run_all fail under a given environment switch?  usage: world2_flaky.sh N "VAR=val" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; N=$1; shift
for kv in "none" "$@"; do
  fails=0
  for i in $(seq 1 $N); do
    if [ "$kv" = "none" ]; then out=$(python3 -m pytest tests/test_gpu_world2.py -q -k run_all 2>&1 | tail -1); else out=$(env $kv python3 -m pytest tests/test_gpu_world2.py -q -k run_all 2>&1 | tail -1); fi
    case "$out" in *failed*) fails=$((fails+1));; esac
  done
  echo "$kv: $fails failures of $N"
done
