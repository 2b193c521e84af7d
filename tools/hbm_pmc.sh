#!/bin/bash
# HBM traffic of the HBM-bound kernels of one n = 220 fragment solve from the memory-side counters (run on the GPU box through gpurun):
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (TCC slot budget; MI355X guide, HBM section), kernel trace only, the program
# directly after `--`.  Writes gpurun_out/hbm_pmc.json (copied to profiles/r04_hbm_pmc.json).
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/hpmc_$c
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/hpmc_$c -- python tools/frag_bench.py 220 20 block > gpurun_out/hpmc_$c.log 2>&1 || echo "rocprofv3 $c left with status $?"
done
python tools/hbm_pmc.py gpurun_out/hpmc_FETCH_SIZE gpurun_out/hpmc_WRITE_SIZE > gpurun_out/hbm_pmc.json
rm -rf gpurun_out/hpmc_FETCH_SIZE gpurun_out/hpmc_WRITE_SIZE
