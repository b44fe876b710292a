#!/bin/bash
# usage (GPU box): tools/dbg/ab_env_c4.sh "<env A>" "<env B>" ...   -- bench.py c4 (10 steps) once per environment, twice, interleaved
R=/root/repo
run() { env $1 timeout -k 10 200 python3 $R/bench.py --workload c4 --steps 10 --warmup 3 --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print(round(d['ms_per_step'],3), round(k['trace_ms'],3), round(k['sun_shadow_ms_full_frame'],3), round(k['point_shadow_ms_full_frame_per_light'],3))"; }
for rep in 1 2; do for e in "$@"; do printf "%-28s " "$e"; run "$e" || exit 1; done; done
