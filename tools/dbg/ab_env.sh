#!/bin/bash
# usage (GPU box): tools/dbg/ab_env.sh "<env assignments A>" "<env assignments B>" ...   -- bench.py c3 (20 steps) once per environment, interleaved twice
R=/root/repo
run() { env $1 timeout -k 10 200 python3 $R/bench.py --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['kernels']['trace_ms'],4), round(d['kernels']['atrous_total_ms'],4))"; }
for rep in 1 2; do for e in "$@"; do printf "%-28s " "$e"; run "$e" || exit 1; done; done
