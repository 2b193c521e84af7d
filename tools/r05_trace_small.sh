#!/bin/bash
# lock-step sweeps of octane BE2 / BE3: phase times (QEMB_BATCH_TRACE) and the kernels of one lock-step iteration (rocprofv3 kernel trace)
export TMPDIR=/tmp
OUT=gpurun_out/small
mkdir -p $OUT
for key in be2 be3; do
  true
  rm -rf gpurun_out/kt
  QEMB_GRAPH=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python tools/octane_lockstep.py test_autogen_octane_$key 3 > $OUT/${key}_trace.log 2>&1 || exit 1
  python tools/trace_lockstep_iteration.py gpurun_out/kt > $OUT/${key}_iteration.txt
  python tools/trace_lockstep.py gpurun_out/kt > $OUT/${key}_trace_summary.txt
  python tools/trace_sweep_phases.py gpurun_out/kt > $OUT/${key}_sweep_kernels.txt 2>&1
done
rm -rf gpurun_out/kt
echo done
