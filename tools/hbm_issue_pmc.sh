#!/bin/bash
# Issue-side counters of the HBM-bound kernels of one n = 220 fragment solve (run on the GPU box through gpurun): are the passes short of the 8 TB/s roofline
# because too few loads are in flight (waves waiting on memory most of their life, few VMEM instructions per wave) or because address arithmetic keeps the
# waves busy (VALU / SALU instructions per VMEM instruction)?  SQ counters in two passes, the derived VMEM latency, the vector-cache (TCP) stall counters and GRBM_GUI_ACTIVE in passes of
# their own; kernel trace only, the program directly after `--`.  Writes gpurun_out/hbm_issue_pmc.json.
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
n=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU" \
           "SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY" "VmemLatency" "TCP_PENDING_STALL_CYCLES TCP_GATE_EN1" "TCP_TCR_TCP_STALL_CYCLES TCP_GATE_EN1" "GRBM_GUI_ACTIVE"; do
  n=$((n + 1))
  rm -rf gpurun_out/ipmc_$n
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/ipmc_$n -- python tools/frag_bench.py 220 20 block > gpurun_out/ipmc_$n.log 2>&1 || echo "rocprofv3 pass $n ($set) left with status $?"
done
python tools/hbm_issue_pmc.py gpurun_out/ipmc_1 gpurun_out/ipmc_2 gpurun_out/ipmc_3 gpurun_out/ipmc_4 gpurun_out/ipmc_5 gpurun_out/ipmc_6 > gpurun_out/hbm_issue_pmc.json
rm -rf gpurun_out/ipmc_[1-6]
