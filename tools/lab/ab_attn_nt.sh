cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for nt in 1 0 1 0; do
  export PCE_ATTN_NT=$nt
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/at_$nt -- python3 tools/align_trace.py > /dev/null 2>&1
  echo "PCE_ATTN_NT=$nt"; python3 tools/align_trace.py --summarise gpurun_out/at_$nt --brief | grep -E "last alignment|k_attention|k_gemm_flat<3>"
  rm -rf gpurun_out/at_$nt
done
