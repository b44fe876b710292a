#!/bin/bash
# usage (GPU box): tools/ab_env.sh "ENV1=.. ENV2=.." ["ENV..."] ...  -- bench.py once per environment setting, short summary each
# BENCH_ARGS (env) is appended to the bench command line
for envs in "$@"; do
  echo "== $envs"
  env $envs timeout -k 10 300 python -u bench.py --steps 10 --warmup 2 --cpu-seconds 0 ${BENCH_ARGS} 2>/dev/null | tail -1 | grep -oE '"value": [0-9.]+|"ms_per_step": [0-9.]+|"trace_ms": [0-9.]+|"atrous_total_ms": [0-9.]+|"lane_util[a-z_]+": [0-9.]+|"nodes_per_ray": [0-9.]+|"tris_per_ray": [0-9.]+|"lbvh_build_ms": [0-9.]+|"sun_shadow[a-z_]+": [0-9.]+' | tr '\n' ' '
  echo
done
