#!/bin/bash
# The counter passes of a round (run from the repo root on the GPU box, its own gpurun call): PMC in runs of their own, as the MI355X guide prescribes.
#   bash tools/profile_round_pmc.sh r05
set -e
TAG=${1:-rXX}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
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

# 4. HBM traffic of the ladder dispatches (FETCH_SIZE / WRITE_SIZE, separate passes)
bash tools/pmc_ladder.sh > $OUT/pmc_ladder.log 2>&1
cp gpurun_out/pmc_ladder.json $OUT/pmc_ladder.json
# 5. SQ / GRBM counters of the GEMM main-loop variants
bash tools/gemm_pmc.sh > $OUT/gemm_pmc.txt 2>&1
cp gpurun_out/gemm_pmc.json $OUT/gemm_pmc.json
python tools/gemm_modes.py 5 > $OUT/gemm_modes.jsonl 2>&1
echo done
