#!/bin/bash
# usage (GPU box): tools/dbg/k2_fetch.sh <tag>   -- K2 time (bench kernels) and FETCH_SIZE / WRITE_SIZE of k_atrous_tile at c3 and c4
set -e
tag=${1:-k2}
R=/root/repo; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for wl in c3 c4; do
  timeout -k 10 200 python3 $R/bench.py --workload $wl --steps 5 --warmup 2 --cpu-seconds 0 2>/dev/null | tail -1 | grep -oE '"atrous_total_ms": [0-9.]+|"ms_per_step": [0-9.]+' | tr '\n' ' '; echo " <- $wl"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/${tag}_${wl}_$ctr
    timeout -s KILL 150 rocprofv3 --pmc $ctr --kernel-include-regex 'k_atrous' --output-format csv -d $O/${tag}_${wl}_$ctr -- \
        python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --cpu-seconds 0 --no-extra > $O/${tag}_${wl}_$ctr.log 2>&1
  done
  python3 $R/tools/pmc_summary.py $O/${tag}_${wl}_FETCH_SIZE $O/${tag}_${wl}_WRITE_SIZE | grep -A3 k_atrous
done
