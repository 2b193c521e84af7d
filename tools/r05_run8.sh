#!/bin/bash
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r05_run8
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -q -x -k "slab_rows or mo_transform or gemm_tall" > $OUT/pytest_slab.log 2>&1 || { tail -30 $OUT/pytest_slab.log; exit 1; }
tail -2 $OUT/pytest_slab.log
timeout -k 10 900 python -m pytest tests/test_gpu_fragment.py -q -x -k "n132 or n220 or mo_transform or rotation or larger" > $OUT/pytest_frag.log 2>&1 || { tail -30 $OUT/pytest_frag.log; exit 1; }
tail -2 $OUT/pytest_frag.log
timeout -k 10 300 python tools/transform_products.py > $OUT/transform_products.jsonl 2>&1
cut -c1-220 $OUT/transform_products.jsonl
timeout -k 10 300 python tools/frag_bench.py 220 20 eeval four-index > $OUT/frag_bench_four_index.log 2>&1
tail -1 $OUT/frag_bench_four_index.log | cut -c1-400
echo done
