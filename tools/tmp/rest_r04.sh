set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_r04
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/lpmc_$c
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/lpmc_$c -- python tools/octane_lockstep.py > gpurun_out/lpmc_$c.log 2>&1 || echo "rocprofv3 $c (lockstep) left with status $?"
done
python tools/pmc_lockstep_iteration.py gpurun_out/lpmc_FETCH_SIZE gpurun_out/lpmc_WRITE_SIZE > $OUT/octane_lockstep_iteration_pmc.json 2>&1 || echo "pmc_lockstep_iteration failed"
rm -rf gpurun_out/lpmc_FETCH_SIZE gpurun_out/lpmc_WRITE_SIZE
python tools/gemm_stamps.py > $OUT/gemm_stamps.jsonl 2>&1
python tools/transform_stagger.py > $OUT/transform_products.jsonl 2>&1
python tools/octane_sweep_series.py 2> $OUT/octane_sweep_series.log > /dev/null
bash tools/pmc_ladder.sh > $OUT/pmc_ladder.log 2>&1
cp gpurun_out/pmc_ladder.json $OUT/pmc_ladder.json
bash tools/gemm_pmc.sh > $OUT/gemm_pmc.txt 2>&1
cp gpurun_out/gemm_pmc.json $OUT/gemm_pmc.json
python tools/gemm_modes.py 5 > $OUT/gemm_modes.jsonl 2>&1
echo done
