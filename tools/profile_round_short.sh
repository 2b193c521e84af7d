#!/bin/bash
# The parts of tools/profile_round.sh that follow a change of the solve outside the CCSD iteration (bench kernel stats, iteration trace, solve phases,
# per-kernel roofline, transforms): bash tools/profile_round_short.sh r04
set -e
TAG=${1:-rXX}
OUT=gpurun_out/prof_${TAG}_short
export TMPDIR=/tmp
mkdir -p $OUT
rm -rf gpurun_out/kt1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt1 -- python bench.py --nstreams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-octane > $OUT/bench_nstreams1.json 2> $OUT/bench_nstreams1.err || echo "rocprofv3 (nstreams 1) left with status $?"
cp gpurun_out/kt1/*/*kernel_stats.csv $OUT/bench_nstreams1_kernel_stats.csv
rm -rf gpurun_out/kt1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt3 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-octane > $OUT/bench_default.json 2> $OUT/bench_default.err || echo "rocprofv3 (default) left with status $?"
cp gpurun_out/kt3/*/*kernel_stats.csv $OUT/bench_default_kernel_stats.csv
rm -rf gpurun_out/kt3
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python tools/frag_bench.py 220 20 eeval > $OUT/frag_bench.log 2>&1
python tools/trace_iteration.py gpurun_out/kt > $OUT/iteration_kernel_trace.txt
python tools/trace_solve.py gpurun_out/kt > $OUT/solve_phases.txt
python tools/kernel_roofline.py "gpurun_out/kt/*/*kernel_trace.csv" > $OUT/kernel_roofline.jsonl
rm -rf gpurun_out/kt
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python tools/frag_bench.py 220 20 eeval four-index > $OUT/frag_bench_four_index.log 2>&1
python tools/trace_solve.py gpurun_out/kt > $OUT/solve_phases_four_index.txt
rm -rf gpurun_out/kt
python tools/transform_bench.py > $OUT/transform_bench.jsonl 2>&1
echo done
