#!/bin/bash
# development: K1 occupancy (workgroups per CU) through VKR_K1_EXTRA_LDS on the default bench workload
R=/root/repo
run() { timeout -k 10 200 python3 $R/bench.py --steps 10 --warmup 2 --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['kernels']['trace_ms'],4))"; }
for x in ${OCC_LDS:-0 1024 6000}; do printf "extra_lds %s: " $x; VKR_K1_EXTRA_LDS=$x run || exit 1; done
