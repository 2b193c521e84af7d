#!/bin/bash
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r05_gather3
mkdir -p $OUT
for args in "3 16 600 200" "3 16 100 50" "4 12 1200 400" "2 24 600 200" "6 8 300 100"; do echo "== W D blocks reps = $args"; timeout -k 5 60 ./tools/probes/graph_dag $args; done > $OUT/graph_dag.txt 2>&1
cat $OUT/graph_dag.txt
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -x -k "jacobi or schmidt" > $OUT/pytest_jacobi.log 2>&1 || { tail -30 $OUT/pytest_jacobi.log; exit 1; }
tail -2 $OUT/pytest_jacobi.log
timeout -k 10 200 python tools/jacobi_bench.py 7 24 42 57 64 80 96 > $OUT/jacobi_bench_db.jsonl 2>&1 || true
cut -c1-200 $OUT/jacobi_bench_db.jsonl
for peers in 0 1; do
  QEMB_LOCKSTEP_PEERS=$peers QEMB_BATCH_TRACE=1 timeout -k 10 200 python tools/octane_lockstep.py test_autogen_octane_be3 8 > $OUT/octane_be3_peers$peers.log 2>&1
  QEMB_LOCKSTEP_PEERS=$peers QEMB_BATCH_TRACE=1 timeout -k 10 200 python tools/octane_lockstep.py test_autogen_octane_be2 8 > $OUT/octane_be2_peers$peers.log 2>&1
done
grep RESULT $OUT/octane_be*_peers*.log | cut -c1-160
QEMB_SWEEP_LOCKSTEP_UPTO=160 QEMB_SWEEP_NBEST=8 timeout -k 10 400 python tools/size_sweep.py 96:9 132:12 > $OUT/size_sweep_lockstep8.jsonl 2> $OUT/size_sweep_lockstep8.err || true
QEMB_SWEEP_NBEST=8 timeout -k 10 400 python tools/size_sweep.py 96:9 132:12 > $OUT/size_sweep_streams8.jsonl 2> $OUT/size_sweep_streams8.err || true
timeout -k 10 400 python tools/size_sweep.py 132:12 > $OUT/size_sweep_streams4.jsonl 2> $OUT/size_sweep_streams4.err || true
echo done
