#!/bin/bash
# same-box A/B of one alignment (tools/align_trace.py under rocprofv3 --kernel-trace) with an environment variable set to each of the given values:
#   tools/lab/ab_align_env.sh VAR v1 v2 ...
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; shift
for round in 1 2; do for v in "$@"; do
  export $VAR=$v
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/at_$v -- python3 tools/align_trace.py > /dev/null 2>&1
  echo "$VAR=$v"; python3 tools/align_trace.py --summarise gpurun_out/at_$v --brief | grep -E "last alignment|k_gemm_bf16|k_gemm_wide|k_attention"
  rm -rf gpurun_out/at_$v
done; done
