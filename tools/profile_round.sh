#!/bin/bash
# usage (on the GPU box, via gpurun): tools/profile_round.sh <tag>
# 1. default bench.py run (JSON line incl. cpu_baseline)      -> gpurun_out/<tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same bench command -> gpurun_out/<tag>_stats/
# 3. two PMC passes (FETCH_SIZE, WRITE_SIZE), one run each      -> gpurun_out/pmc_<tag>_fetch, pmc_<tag>_write
# 4. one SQ pass (issue/occupancy counters)                     -> gpurun_out/pmc_<tag>_sq
# 5. one cache pass (L2 hit/miss, L1 accesses / L1->L2 reads)     -> gpurun_out/pmc_<tag>_cache
set -e
tag=${1:-r1}
R=/root/repo
rm -rf $R/gpurun_out/pmc_${tag}_* $R/gpurun_out/${tag}_stats
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
tail -c 3000 $R/gpurun_out/${tag}_bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- \
    python3 $R/bench.py --cpu-seconds 0 > $R/gpurun_out/${tag}_stats.log 2>&1
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-extra"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'k_trace|k_atrous' --output-format csv \
    -d $R/gpurun_out/pmc_${tag}_fetch -- $BENCH > $R/gpurun_out/pmc_${tag}_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex 'k_trace|k_atrous' --output-format csv \
    -d $R/gpurun_out/pmc_${tag}_write -- $BENCH > $R/gpurun_out/pmc_${tag}_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY \
    --kernel-include-regex 'k_trace|k_atrous' --output-format csv \
    -d $R/gpurun_out/pmc_${tag}_sq -- $BENCH > $R/gpurun_out/pmc_${tag}_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum \
    --kernel-include-regex 'k_trace|k_atrous' --output-format csv \
    -d $R/gpurun_out/pmc_${tag}_cache -- $BENCH > $R/gpurun_out/pmc_${tag}_cache.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${tag}_fetch $R/gpurun_out/pmc_${tag}_write $R/gpurun_out/pmc_${tag}_sq $R/gpurun_out/pmc_${tag}_cache \
    > $R/gpurun_out/pmc_${tag}_summary.txt
cat $R/gpurun_out/pmc_${tag}_summary.txt
