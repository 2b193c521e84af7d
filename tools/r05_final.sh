#!/bin/bash
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_r05
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/gputest.log 2>&1 || { tail -40 $OUT/gputest.log; exit 1; }
tail -2 $OUT/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout -k 10 900 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -3
grep "\[bench" $OUT/bench.err | tail -5
timeout -k 10 400 python bench.py --gpus 1 --scaling strong --frags-total 64 --steps 1 --warmup 1 --no-cpu-baseline --no-octane --no-size-sweep > $OUT/bench_strong_n1.json 2> $OUT/bench_strong_n1.err
echo done
