#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counters...>   -- one rocprofv3 --pmc pass over a short bench run (run on the GPU box)
# Output: gpurun_out/pmc_<tag>/  (csv).  One pass per invocation: rocprofv3 does not split counters over passes.
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-include-regex 'k_trace|k_atrous|k_shadow' --output-format csv \
    -d /root/repo/gpurun_out/pmc_$tag -- python3 /root/repo/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-extra \
    ${BENCH_ARGS} > /root/repo/gpurun_out/pmc_$tag.log 2>&1
