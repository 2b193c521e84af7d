#!/bin/bash
# round 5, second pass: new small eigensolver (tests + A/B), octane BE2 / BE3 phases and lock-step traces
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r05_gather2
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -x -k "jacobi or schmidt" > $OUT/pytest_jacobi.log 2>&1 || { tail -30 $OUT/pytest_jacobi.log; exit 1; }
tail -2 $OUT/pytest_jacobi.log
for mode in 1 0; do
  QEMB_JACOBI_TWOSIDED=$mode QEMB_BATCH_TRACE=1 timeout -k 10 200 python tools/octane_lockstep.py test_autogen_octane_be2 8 > $OUT/octane_be2_lockstep_phases_twosided$mode.log 2>&1
  QEMB_JACOBI_TWOSIDED=$mode QEMB_BATCH_TRACE=1 timeout -k 10 200 python tools/octane_lockstep.py test_autogen_octane_be3 8 > $OUT/octane_be3_lockstep_phases_twosided$mode.log 2>&1
done
grep RESULT $OUT/octane_be*_phases_*.log | cut -c1-200
timeout -k 10 200 python tools/jacobi_bench.py 24 42 57 64 96 > $OUT/jacobi_bench_twosided.jsonl 2>&1 || true
QEMB_JACOBI_TWOSIDED=0 timeout -k 10 200 python tools/jacobi_bench.py 24 42 57 64 96 > $OUT/jacobi_bench_onesided.jsonl 2>&1 || true
rm -rf gpurun_out/kt
QEMB_GRAPH=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python tools/octane_lockstep.py test_autogen_octane_be3 3 > $OUT/octane_be3_lockstep.log 2>&1
python tools/trace_lockstep_iteration.py gpurun_out/kt > $OUT/octane_be3_lockstep_iteration.txt || true
python tools/trace_lockstep.py gpurun_out/kt > $OUT/octane_be3_lockstep_trace.txt || true
cp gpurun_out/kt/*/*kernel_trace.csv $OUT/octane_be3_kernel_trace.csv
rm -rf gpurun_out/kt
QEMB_GRAPH=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python tools/octane_lockstep.py test_autogen_octane_be2 3 > $OUT/octane_be2_lockstep.log 2>&1
python tools/trace_lockstep_iteration.py gpurun_out/kt > $OUT/octane_be2_lockstep_iteration.txt || true
cp gpurun_out/kt/*/*kernel_trace.csv $OUT/octane_be2_kernel_trace.csv
rm -rf gpurun_out/kt
timeout -k 10 300 python tools/size_sweep.py 96:9 132:12 > $OUT/size_sweep_96_132.jsonl 2> $OUT/size_sweep_96_132.err || true
echo done
