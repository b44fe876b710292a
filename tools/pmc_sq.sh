#!/bin/bash
# usage (GPU box): tools/pmc_sq.sh <tag>  -- one SQ counter pass + one kernel-trace pass over a short bench run
tag=${1:-x}
R=/root/repo
rm -rf $R/gpurun_out/pmc_${tag}_sq $R/gpurun_out/${tag}_kt
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-extra"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY \
    --kernel-include-regex 'k_trace' --output-format csv -d $R/gpurun_out/pmc_${tag}_sq -- $BENCH > $R/gpurun_out/pmc_${tag}_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT \
    --kernel-include-regex 'k_trace' --output-format csv -d $R/gpurun_out/pmc_${tag}_sq2 -- $BENCH > $R/gpurun_out/pmc_${tag}_sq2.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${tag}_sq $R/gpurun_out/pmc_${tag}_sq2
