#!/bin/bash
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r05_run7
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm" > $OUT/pytest_gemm.log 2>&1 || { tail -30 $OUT/pytest_gemm.log; exit 1; }
tail -2 $OUT/pytest_gemm.log
timeout -k 10 900 python -m pytest tests/test_gpu_fragment.py tests/test_gpu_be.py -q -x -k "lockstep or batch or octane or randomised or bench_tiles" > $OUT/pytest_lockstep.log 2>&1 || { tail -30 $OUT/pytest_lockstep.log; exit 1; }
tail -2 $OUT/pytest_lockstep.log
for reg in 1 0; do
  QEMB_TAPE_REGIONS=$reg QEMB_BATCH_TRACE=1 timeout -k 10 200 python tools/octane_lockstep.py test_autogen_octane_be2 8 > $OUT/octane_be2_regions$reg.log 2>&1
  QEMB_TAPE_REGIONS=$reg QEMB_BATCH_TRACE=1 timeout -k 10 200 python tools/octane_lockstep.py test_autogen_octane_be3 8 > $OUT/octane_be3_regions$reg.log 2>&1
done
grep RESULT $OUT/octane_be*_regions*.log | cut -c1-260
grep "qemb batch" $OUT/octane_be2_regions1.log | tail -2
rm -rf gpurun_out/kt
QEMB_GRAPH=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python tools/octane_lockstep.py test_autogen_octane_be2 3 > $OUT/octane_be2_lockstep.log 2>&1
python tools/trace_lockstep_iteration.py gpurun_out/kt > $OUT/octane_be2_lockstep_iteration.txt || true
rm -rf gpurun_out/kt
timeout -k 10 400 python tools/size_sweep.py 96:9 132:12 176:16 > $OUT/size_sweep_mid.jsonl 2> $OUT/size_sweep_mid.err || true
QEMB_RING96=1 timeout -k 10 400 python tools/size_sweep.py 132:12 > $OUT/size_sweep_ring96.jsonl 2> $OUT/size_sweep_ring96.err || true
QEMB_SWEEP_LOCKSTEP_UPTO=160 QEMB_SWEEP_NBEST=8 timeout -k 10 400 python tools/size_sweep.py 96:9 132:12 > $OUT/size_sweep_lockstep8.jsonl 2> $OUT/size_sweep_lockstep8.err || true
echo done
