#!/bin/bash
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r05_run4
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/gputest.log 2>&1 || { tail -40 $OUT/gputest.log; exit 1; }
tail -3 $OUT/gputest.log
timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_quick.json 2> $OUT/bench_quick.err || { tail -20 $OUT/bench_quick.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_run4/bench_quick.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "resident", d["config"].get("resident_bytes_per_fragment"), "four_index", d.get("four_index_route"))
print("octane", {k: d["octane_be2"].get(k) for k in ("lockstep_ms", "streams6_ms")} if isinstance(d.get("octane_be2"), dict) else d.get("octane_be2_sweep_ms"))
print("df_c4", d.get("df_c4"))
for r in d.get("size_sweep", {}).get("rows", []):
    print(r.get("n"), r.get("single_stream", {}).get("frac_of_peak"), r.get("best_mode", {}).get("frac_of_peak"), r.get("failed"))
PY
timeout -k 10 900 python bench.py --gpus 1 --scaling strong --frags-total 64 --steps 1 --warmup 1 --no-cpu-baseline --no-octane --no-size-sweep > $OUT/bench_strong_n1.json 2> $OUT/bench_strong_n1.err || { tail -20 $OUT/bench_strong_n1.err; exit 1; }
python -c "
import json; d = json.load(open('gpurun_out/r05_run4/bench_strong_n1.json')); print('strong N=1:', d['value'], d['ms_per_step'], d['config']['fragments_per_gpu'], d['config'].get('resident_bytes_per_fragment'))"
echo done
