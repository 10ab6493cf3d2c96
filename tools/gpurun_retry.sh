#!/bin/bash
# gpurun with retries while the pool is busy (exit code 3 = no box or slot free: nothing charged).  usage: tools/gpurun_retry.sh <timeout s> '<command>' [log]
T=$1; CMD=$2; LOG=${3:-/root/repo/gpurun_out/last_retry.log}
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$CMD" > $LOG 2>&1; rc=$?
  if [ $rc -ne 3 ] && ! grep -q "status=transient" $LOG; then exit $rc; fi
  sleep 120
done
exit 3
