cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_dec; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 tools/decode_trace.py > $O/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 tools/decode_trace.py > $O/w.log 2>&1
F=$(find $O/f -name "*counter_collection.csv" | head -1); W=$(find $O/w -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py $F $W gpurun_out/pmc_traffic_decode.json; cat gpurun_out/pmc_traffic_decode.json | head -40
rm -rf $O
