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
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt1 -- python bench.py --nstreams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-octane > $OUT/bench_nstreams1.json 2> $OUT/bench_nstreams1.err || echo "rocprofv3 (nstreams 1) left with status $?"
cp gpurun_out/kt1/*/*kernel_stats.csv $OUT/bench_nstreams1_kernel_stats.csv
rm -rf gpurun_out/kt1
# 2. the default bench command (four fragments in flight: kernels of different streams overlap, durations are contended; under the profiler the contexts keep plain streams)
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt3 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-octane > $OUT/bench_default.json 2> $OUT/bench_default.err || echo "rocprofv3 (default) left with status $? (the profiler's own exit handlers; its CSV files are written before)"
cp gpurun_out/kt3/*/*kernel_stats.csv $OUT/bench_default_kernel_stats.csv
rm -rf gpurun_out/kt3
# 3. one CCSD iteration, kernel by kernel
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python tools/frag_bench.py 220 20 > $OUT/frag_bench.log 2>&1
python tools/trace_iteration.py gpurun_out/kt > $OUT/iteration_kernel_trace.txt
python tools/trace_solve.py gpurun_out/kt > $OUT/solve_phases.txt
python tools/kernel_roofline.py "gpurun_out/kt/*/*kernel_trace.csv" > $OUT/kernel_roofline.jsonl
rm -rf gpurun_out/kt
# 3b. the HBM-bound kernels and the eigensolver on their own; AO -> fragment transforms; the small-fragment regime
python tools/hbm_kernels.py > $OUT/hbm_kernels.jsonl 2>&1
python tools/jacobi_bench.py 130 220 300 512 > $OUT/jacobi_bench.jsonl 2>&1
QEMB_JACOBI_BLOCK=0 python tools/jacobi_bench.py 220 > $OUT/jacobi_bench_per_pair_rounds.jsonl 2>&1
python tools/transform_bench.py > $OUT/transform_bench.jsonl 2>&1
QEMB_BATCH_TRACE=1 python tools/octane_quick.py 2>&1 | grep "RESULT\|qemb batch" > $OUT/octane_streams_lockstep.log
python tools/octane_be3_sweeps.py 2>&1 | grep RESULT > $OUT/octane_be3_sweeps.log
QEMB_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktl -- python tools/octane_lockstep.py > $OUT/octane_lockstep.log 2>&1
python tools/trace_lockstep.py gpurun_out/ktl > $OUT/octane_lockstep_trace.txt
python tools/trace_lockstep_iteration.py gpurun_out/ktl > $OUT/octane_lockstep_iteration.txt
rm -rf gpurun_out/ktl
# 3c. round 4: memory-side counters per HBM-bound kernel of a solve, and of one lock-step iteration of the octane sweep; the K = 220 products
bash tools/hbm_pmc.sh > $OUT/hbm_pmc.log 2>&1
cp gpurun_out/hbm_pmc.json $OUT/hbm_pmc.json
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/lpmc_$c
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/lpmc_$c -- python tools/octane_lockstep.py > gpurun_out/lpmc_$c.log 2>&1 || echo "rocprofv3 $c (lockstep) left with status $?"
done
python tools/pmc_lockstep_iteration.py gpurun_out/lpmc_FETCH_SIZE gpurun_out/lpmc_WRITE_SIZE > $OUT/octane_lockstep_iteration_pmc.json 2>&1 || echo "pmc_lockstep_iteration failed"
rm -rf gpurun_out/lpmc_FETCH_SIZE gpurun_out/lpmc_WRITE_SIZE
python tools/gemm_stamps.py > $OUT/gemm_stamps.jsonl 2>&1
python tools/transform_products.py > $OUT/transform_products.jsonl 2>&1

# 4. HBM traffic of the ladder dispatches (FETCH_SIZE / WRITE_SIZE, separate passes)
bash tools/pmc_ladder.sh > $OUT/pmc_ladder.log 2>&1
cp gpurun_out/pmc_ladder.json $OUT/pmc_ladder.json
# 5. SQ / GRBM counters of the GEMM main-loop variants
bash tools/gemm_pmc.sh > $OUT/gemm_pmc.txt 2>&1
cp gpurun_out/gemm_pmc.json $OUT/gemm_pmc.json
python tools/gemm_modes.py 5 > $OUT/gemm_modes.jsonl 2>&1
echo done
