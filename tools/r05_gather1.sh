#!/bin/bash
# round 5, first data-gathering pass: size sweep (before any change), kernel stats at n = 96 / 132, BE3 lock-step trace
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r05_gather1
mkdir -p $OUT
timeout -k 10 400 python tools/size_sweep.py > $OUT/size_sweep_before.jsonl 2> $OUT/size_sweep_before.err
echo "size sweep done"
for sz in "96 9" "132 12"; do
  tag=$(echo $sz | tr ' ' '_')
  rm -rf gpurun_out/kt
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python tools/frag_bench.py $sz eeval > $OUT/frag_bench_$tag.log 2>&1
  cp gpurun_out/kt/*/*kernel_stats.csv $OUT/kernel_stats_n$tag.csv
  cp gpurun_out/kt/*/*kernel_trace.csv $OUT/kernel_trace_n$tag.csv
  rm -rf gpurun_out/kt
  echo "trace $sz done"
done
timeout -k 10 200 python tools/octane_be3_sweeps.py > $OUT/octane_be3_sweeps.log 2>&1
rm -rf gpurun_out/kt
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python tools/octane_lockstep.py test_autogen_octane_be3 3 > $OUT/octane_be3_lockstep.log 2>&1
python tools/trace_lockstep_iteration.py gpurun_out/kt > $OUT/octane_be3_lockstep_iteration.txt
python tools/trace_lockstep.py gpurun_out/kt > $OUT/octane_be3_lockstep_trace.txt
rm -rf gpurun_out/kt
QEMB_BATCH_TRACE=1 timeout -k 10 200 python tools/octane_lockstep.py test_autogen_octane_be2 5 > $OUT/octane_be2_lockstep_phases.log 2>&1
QEMB_BATCH_TRACE=1 timeout -k 10 200 python tools/octane_lockstep.py test_autogen_octane_be3 5 > $OUT/octane_be3_lockstep_phases.log 2>&1
echo done
