#!/bin/bash
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r05_run6
mkdir -p $OUT
timeout -k 10 900 python bench.py --gpus 1 --scaling strong --frags-total 64 --steps 1 --warmup 1 --no-cpu-baseline --no-octane --no-size-sweep > $OUT/bench_strong_n1.json 2> $OUT/bench_strong_n1.err || { tail -20 $OUT/bench_strong_n1.err; exit 1; }
python -c "
import json; d = json.load(open('gpurun_out/r05_run6/bench_strong_n1.json')); print('strong N=1:', d['value'], d['ms_per_step'], d['config']['fragments_per_gpu'], d['config'].get('resident_bytes_per_fragment'), d['config'].get('redrawn_fragments_rank0'), d.get('four_index_route'))"
QEMB_BATCH_TRACE=1 QEMB_SWEEP_LOCKSTEP_UPTO=160 QEMB_SWEEP_NBEST=8 timeout -k 10 400 python tools/size_sweep.py 96:9 132:12 > $OUT/size_sweep_lockstep8.jsonl 2> $OUT/size_sweep_lockstep8.err || true
grep "qemb batch" $OUT/size_sweep_lockstep8.err | tail -8
echo done
