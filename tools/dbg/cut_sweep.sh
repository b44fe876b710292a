#!/bin/bash
# development: K1 entry-cut parameters (VKR_CUT_MAX x VKR_CUT_SMALL) on the default bench workload; prints ms_per_step, trace_ms, visits per ray
R=/root/repo
run() { timeout -k 10 200 python3 $R/bench.py --steps 10 --warmup 2 --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['kernels']['trace_ms'],4), round(d['roofline']['nodes_per_ray'],2), round(d['roofline']['tris_per_ray'],2))"; }
if [ -f $R/vulkanrenderer_amd/lib/variants/libvkr_nocut.so ]; then printf "nocut: "; VKR_LIB_PATH=$R/vulkanrenderer_amd/lib/variants/libvkr_nocut.so run || exit 1; fi
CFGS=${CUT_CFGS:-1:2 8:2 16:2 16:1 16:0.5 16:0.25}
for cfg in $CFGS; do
  printf "cut_max %s small %s: " ${cfg%%:*} ${cfg##*:}
  VKR_CUT_MAX=${cfg%%:*} VKR_CUT_SMALL=${cfg##*:} run || exit 1
done
