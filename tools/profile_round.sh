#!/bin/bash
# The round's judged profiles in one gpurun call (run from the repo root on the GPU box): writes everything under gpurun_out/prof_<tag>/.
#   bash tools/profile_round.sh r02
set -e
TAG=${1:-rXX}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
# 1. the bench command, one fragment in flight: per-kernel durations of kernels that own the device (agreement with the HIP-event timers)
rm -rf gpurun_out/kt1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt1 -- python bench.py --nstreams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-octane --no-size-sweep > $OUT/bench_nstreams1.json 2> $OUT/bench_nstreams1.err || echo "rocprofv3 (nstreams 1) left with status $?"
cp gpurun_out/kt1/*/*kernel_stats.csv $OUT/bench_nstreams1_kernel_stats.csv
rm -rf gpurun_out/kt1
# 2. the default bench command (four fragments in flight: kernels of different streams overlap, durations are contended; under the profiler the contexts keep plain streams)
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt3 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-octane --no-size-sweep > $OUT/bench_default.json 2> $OUT/bench_default.err || echo "rocprofv3 (default) left with status $? (the profiler's own exit handlers; its CSV files are written before)"
cp gpurun_out/kt3/*/*kernel_stats.csv $OUT/bench_default_kernel_stats.csv
rm -rf gpurun_out/kt3
# 3. one CCSD iteration, kernel by kernel
QEMB_GEMM_SHAPELOG=gpurun_out/shapes.txt timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python tools/frag_bench.py 220 20 eeval > $OUT/frag_bench.log 2>&1
python tools/trace_iteration.py gpurun_out/kt > $OUT/iteration_kernel_trace.txt
python tools/trace_solve.py gpurun_out/kt > $OUT/solve_phases.txt
python tools/kernel_roofline.py "gpurun_out/kt/*/*kernel_trace.csv" gpurun_out/shapes.txt 220 20 660 > $OUT/kernel_roofline.jsonl
rm -rf gpurun_out/kt
QEMB_GEMM_SHAPELOG=gpurun_out/shapes4.txt timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python tools/frag_bench.py 220 20 eeval four-index > $OUT/frag_bench_four_index.log 2>&1
python tools/trace_solve.py gpurun_out/kt > $OUT/solve_phases_four_index_route.txt
python tools/kernel_roofline.py "gpurun_out/kt/*/*kernel_trace.csv" gpurun_out/shapes4.txt 220 20 660 > $OUT/kernel_roofline_four_index_route.jsonl
rm -rf gpurun_out/kt
# 3a. round 5: kernel stats of the mid-size points of the size sweep
for sz in "96 9" "132 12"; do
  tag=$(echo $sz | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python tools/frag_bench.py $sz eeval > $OUT/frag_bench_n$tag.log 2>&1
  cp gpurun_out/kt/*/*kernel_stats.csv $OUT/kernel_stats_n$tag.csv
  rm -rf gpurun_out/kt
done
# 3b. the HBM-bound kernels and the eigensolver on their own; AO -> fragment transforms; the small-fragment regime
python tools/hbm_kernels.py > $OUT/hbm_kernels.jsonl 2>&1
python tools/jacobi_bench.py 24 42 57 80 96 130 220 300 512 > $OUT/jacobi_bench.jsonl 2>&1
QEMB_JACOBI_TWOSIDED=0 python tools/jacobi_bench.py 24 42 57 80 96 > $OUT/jacobi_bench_one_sided_small.jsonl 2>&1
QEMB_JACOBI_BLOCK=0 python tools/jacobi_bench.py 220 > $OUT/jacobi_bench_per_pair_rounds.jsonl 2>&1
python tools/transform_bench.py > $OUT/transform_bench.jsonl 2>&1
QEMB_BATCH_TRACE=1 python tools/octane_quick.py 2>&1 | grep "RESULT\|qemb batch" > $OUT/octane_streams_lockstep.log
python tools/octane_be3_sweeps.py 2>&1 | grep RESULT > $OUT/octane_be3_sweeps.log
QEMB_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktl -- python tools/octane_lockstep.py > $OUT/octane_lockstep.log 2>&1
python tools/trace_lockstep.py gpurun_out/ktl > $OUT/octane_lockstep_trace.txt
python tools/trace_lockstep_iteration.py gpurun_out/ktl > $OUT/octane_lockstep_iteration.txt
rm -rf gpurun_out/ktl
QEMB_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktl -- python tools/octane_lockstep.py test_autogen_octane_be3 3 > $OUT/octane_be3_lockstep.log 2>&1
python tools/trace_lockstep_iteration.py gpurun_out/ktl > $OUT/octane_be3_lockstep_iteration.txt
rm -rf gpurun_out/ktl
QEMB_TAPE_REGIONS=0 QEMB_BATCH_TRACE=1 python tools/octane_lockstep.py test_autogen_octane_be2 8 2>&1 | grep "RESULT\|qemb batch" | tail -4 > $OUT/octane_lockstep_regions_off.log
QEMB_BATCH_TRACE=1 python tools/octane_lockstep.py test_autogen_octane_be2 8 2>&1 | grep "RESULT\|qemb batch" | tail -4 > $OUT/octane_lockstep_regions_on.log
python tools/transform_products.py > $OUT/transform_products.jsonl 2>&1
./tools/probes/graph_dag 3 16 600 200 > $OUT/probe_graph_dag.txt 2>&1 || true
./tools/probes/graph_dag 6 8 300 100 >> $OUT/probe_graph_dag.txt 2>&1 || true
echo done
