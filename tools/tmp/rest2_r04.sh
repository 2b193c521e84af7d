export TMPDIR=/tmp
OUT=gpurun_out/prof_r04
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/lpmc_$c
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/lpmc_$c -- python tools/octane_lockstep.py > gpurun_out/lpmc_$c.log 2>&1 || echo "rocprofv3 $c (lockstep) left with status $?"
done
python tools/pmc_lockstep_iteration.py gpurun_out/lpmc_FETCH_SIZE gpurun_out/lpmc_WRITE_SIZE > $OUT/octane_lockstep_iteration_pmc.json 2>&1 || echo "pmc_lockstep_iteration failed"
rm -rf gpurun_out/lpmc_FETCH_SIZE gpurun_out/lpmc_WRITE_SIZE
python tools/octane_sweep_series.py 12 2> $OUT/octane_sweep_series.log > /dev/null
python tools/octane_sweep_series.py 12 --ballast 2> $OUT/octane_sweep_series_ballast.log > /dev/null
echo done
