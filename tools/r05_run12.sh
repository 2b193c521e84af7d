#!/bin/bash
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r05_run12
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_fragment.py tests/test_gpu_be.py -q -x -k "lockstep or batch or octane or h8 or c5 or kbe" > $OUT/pytest_lockstep.log 2>&1 || { tail -30 $OUT/pytest_lockstep.log; exit 1; }
tail -2 $OUT/pytest_lockstep.log
for pre in 1 0; do
  QEMB_TAPE_PREPHASE=$pre QEMB_BATCH_TRACE=1 timeout -k 10 200 python tools/octane_lockstep.py test_autogen_octane_be2 8 > $OUT/octane_be2_pre$pre.log 2>&1
  QEMB_TAPE_PREPHASE=$pre QEMB_BATCH_TRACE=1 timeout -k 10 200 python tools/octane_lockstep.py test_autogen_octane_be3 8 > $OUT/octane_be3_pre$pre.log 2>&1
done
grep RESULT $OUT/octane_be*_pre*.log | cut -c1-140
grep "qemb batch" $OUT/octane_be2_pre1.log | tail -2
grep "qemb batch" $OUT/octane_be2_pre0.log | tail -2
echo done
