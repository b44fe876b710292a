#!/bin/bash
# usage: tools/variant.sh "<extra -D flags>" [bench args]   -- build the HIP library with extra defines, bench on the GPU box
set -e
cd /root/repo/vulkanrenderer_amd/csrc
touch rtao_kernels.hip shadow_kernels.hip bvh_build.hip vkr_api.hip
make -j4 FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function $1" 2>&1 | grep -E "rror" || true
cd /root/repo
shift || true
/usr/local/graft/bin/gpurun --timeout 400 -- "timeout -k 10 300 python -u bench.py --steps 10 --warmup 2 --cpu-seconds 0 $* > gpurun_out/variant.log 2>&1; echo exit=\$?" 2>&1 | grep -E "exit=|refused|status=" | tr '\n' ' '
echo
tail -1 gpurun_out/variant.log | grep -oE '"value": [0-9.]+|"ms_per_step": [0-9.]+|"trace_ms": [0-9.]+|"atrous_total_ms": [0-9.]+|"lane_util[a-z_]+": [0-9.]+|"nodes_per_ray": [0-9.]+|"tris_per_ray": [0-9.]+|"lbvh_build_ms": [0-9.]+|"sun_shadow[a-z_]+": [0-9.]+' | tr '\n' ' '
echo
