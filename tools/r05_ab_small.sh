#!/bin/bash
# A/B of a lock-step switch on octane BE2 / BE3 sweeps: bash tools/r05_ab_small.sh VAR (runs VAR=0 and default)
OUT=gpurun_out/ab
mkdir -p $OUT
V=${1:-QEMB_GROUP_XREMAP}
OFF=${2:-0}
for key in be2 be3; do
  env $V=$OFF QEMB_BATCH_TRACE=1 timeout -k 10 120 python tools/octane_lockstep.py test_autogen_octane_$key 12 2>&1 | grep "RESULT\|qemb batch" | tail -3 > $OUT/${key}_off.log || exit 1
  QEMB_BATCH_TRACE=1 timeout -k 10 120 python tools/octane_lockstep.py test_autogen_octane_$key 12 2>&1 | grep "RESULT\|qemb batch" | tail -3 > $OUT/${key}_on.log || exit 1
done
tail -n 3 $OUT/*.log
