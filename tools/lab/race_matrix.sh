#!/bin/bash
# Several processes on ONE GPU: which condition makes the log-mel / decoding kernels differ from their own first iteration?
# (profiles/r05/multiprocess_glitch.txt left it at "a platform effect"; this is the matrix that separates wave save / restore under
# oversubscription from a missing dependency inside libpce: single stream, serialised kernels, DISJOINT compute-unit masks per process.)
# usage: tools/lab/race_matrix.sh [iterations = 200] [configs...]   -> gpurun_out/race_matrix.txt
cd "$(dirname "$0")/../.." || exit 1
IT=${1:-200}; shift
OUT=gpurun_out/race_matrix.txt
mkdir -p gpurun_out
THIRDS="0:0-79;0:80-159;0:160-255"
HALVES="0:0-127;0:128-255"
run() {   # name, then VAR=value ... pairs
    local name=$1; shift
    echo "=== $name: $*" | tee -a $OUT
    local t0=$(date +%s)
    env PROBE_DETAIL=1 "$@" timeout 420 python3 tools/lab/race_probe.py "$IT" 2>&1 | grep -v "^$" | grep -E "iterations|differing|   frame|Error|error|Traceback" | cut -c1-700 | tee -a $OUT
    echo "    ($(( $(date +%s) - t0 )) s)" | tee -a $OUT
}
ALL="${*:-solo base3 noaux3 serial3 cumask3 mel3 dec3 two_idle1 cumask2_idle1 nosdma3 base3b}"
for c in $ALL; do
    case $c in
        solo)          run solo          PROBE_PAR=1 ;;
        base3|base3b)  run $c            PROBE_PAR=3 ;;
        noaux3)        run noaux3        PROBE_PAR=3 PCE_NO_AUX=1 ;;
        serial3)       run serial3       PROBE_PAR=3 AMD_SERIALIZE_KERNEL=3 ;;
        cumask3)       run cumask3       PROBE_PAR=3 "PROBE_CU_MASKS=$THIRDS" ;;
        mel3)          run mel3          PROBE_PAR=3 PROBE_STAGES=mel ;;
        dec3)          run dec3          PROBE_PAR=3 PROBE_STAGES=dec ;;
        two_idle1)     run two_idle1     PROBE_PAR=2 PROBE_IDLE=1 ;;
        cumask2_idle1) run cumask2_idle1 PROBE_PAR=2 PROBE_IDLE=1 "PROBE_CU_MASKS=$HALVES" ;;
        nosdma3)       run nosdma3       PROBE_PAR=3 HSA_ENABLE_SDMA=0 ;;
        base4)         run base4         PROBE_PAR=4 ;;
        cumask4)       run cumask4       PROBE_PAR=4 "PROBE_CU_MASKS=0:0-63;0:64-127;0:128-191;0:192-255" ;;
        v_mel_c_*)     r=${c#v_mel_c_}; run $c PROBE_PAR=3 "PROBE_ROLES=mel;$r;$r" ;;      # victim: log-mel only; two neighbours run only <r> (enc, dec, align, torch:gemm ...)
        shift_mel_c_*) r=${c#shift_mel_c_}; run $c PROBE_PAR=3 "PROBE_ROLES=mel;$r;$r" PROBE_SHIFT_MB=1537 ;;
        v_c2_c_*)      r=${c#v_c2_c_}; run $c PROBE_PAR=3 "PROBE_ROLES=c2;$r;$r" ;;
        v_dec_c_*)     r=${c#v_dec_c_}; run $c PROBE_PAR=3 "PROBE_ROLES=dec;$r;$r" ;;
        culprit_*)     r=${c#culprit_}; echo "=== $c: the log-mel repeat test beside two processes that run only <$r> (tools/lab/lds_culprit.hip)" | tee -a $OUT
                       tools/lab/bin/lds_culprit $r 25 > gpurun_out/culprit_a.txt 2>&1 &
                       p1=$!
                       tools/lab/bin/lds_culprit $r 25 > gpurun_out/culprit_b.txt 2>&1 &
                       p2=$!
                       sleep 3
                       env PROBE_STAGES=mel timeout 300 python3 tools/lab/race_probe.py "$IT" p0 2>&1 | grep -E "iterations|rror" | cut -c1-300 | tee -a $OUT
                       wait $p1 $p2; cat gpurun_out/culprit_a.txt | tee -a $OUT ;;
        pkv_*)         r=${c#pkv_}; echo "=== $c: tools/lab/pk_victim.hip class <$r>: alone; beside an MFMA kernel of the SAME process; beside two MFMA PROCESSES" | tee -a $OUT
                       tools/lab/bin/pk_victim $r 5 0 2>&1 | tail -3 | tee -a $OUT
                       tools/lab/bin/pk_victim $r 8 1 2>&1 | tail -8 | tee -a $OUT
                       tools/lab/bin/lds_culprit mfma 14 > gpurun_out/culprit_a.txt 2>&1 &
                       p1=$!
                       tools/lab/bin/lds_culprit mfma 14 > gpurun_out/culprit_b.txt 2>&1 &
                       p2=$!
                       sleep 3
                       tools/lab/bin/pk_victim $r 8 0 2>&1 | tail -8 | tee -a $OUT
                       wait $p1 $p2 ;;
        pks_*)         r=${c#pks_}; echo "=== $c: pk_victim class <$r> beside an MFMA kernel of the SAME process (second stream): default build, then -fno-slp-vectorize" | tee -a $OUT
                       tools/lab/bin/pk_victim $r 6 1 2>&1 | tail -4 | tee -a $OUT
                       tools/lab/bin/pk_victim_noslp $r 6 1 2>&1 | tail -4 | tee -a $OUT ;;
        canary_solo)   echo "=== canary_solo" | tee -a $OUT; tools/lab/bin/lds_canary 6 48 512 2>&1 | tail -30 | tee -a $OUT ;;
        canary_*)      r=${c#canary_}; echo "=== $c: the LDS / VGPR canary (48 KiB per workgroup) beside two libpce processes running only <$r>" | tee -a $OUT
                       tools/lab/bin/lds_canary 40 48 512 > gpurun_out/canary_$r.txt 2>&1 &
                       cpid=$!
                       env PROBE_PAR=2 PROBE_STAGES=$r timeout 300 python3 tools/lab/race_probe.py "$IT" 2>&1 | grep -E "iterations|rror" | cut -c1-300 | tee -a $OUT
                       wait $cpid; head -60 gpurun_out/canary_$r.txt | tee -a $OUT; tail -1 gpurun_out/canary_$r.txt | tee -a $OUT ;;
        *) echo "unknown configuration $c" ;;
    esac
done
echo "done" | tee -a $OUT
