#!/bin/bash
# usage: tools/variants.sh name1:"-DFLAG=1 ..." name2:"..." ...
# Builds one copy of libvkr_rt.so per variant (extra -D flags) into vulkanrenderer_amd/lib/variants/, then runs
# bench.py once per variant in ONE gpurun call and prints a one-line summary each.
set -e
ROOT=/root/repo
SRC=$ROOT/vulkanrenderer_amd/csrc
OUTD=$ROOT/vulkanrenderer_amd/lib/variants
mkdir -p $OUTD
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function"
names=()
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  names+=($name)
  [ -n "$SKIP_BUILD" ] && continue
  ( bd=/tmp/vkr_variant_$name; mkdir -p $bd
    for f in vkr_api bvh_build rtao_kernels shadow_kernels indirect_build ibl_kernels comm; do
      /opt/rocm/bin/hipcc $BASE $flags -I$SRC -c $SRC/$f.hip -o $bd/$f.o &
    done; wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUTD/libvkr_$name.so $bd/*.o -ldl ) 2>&1 | grep -E "error" || true
done
cmd=""
for n in "${names[@]}"; do
  cmd+="echo == $n; ${BENCH_ENV} VKR_LIB_PATH=\$PWD/vulkanrenderer_amd/lib/variants/libvkr_$n.so timeout -k 10 200 python -u bench.py --steps 10 --warmup 2 --cpu-seconds 0 ${BENCH_ARGS} 2>/dev/null | tail -1 | grep -oE '\"value\": [0-9.]+|\"trace_ms\": [0-9.]+|\"lane_util[a-z_]+\": [0-9.]+|\"nodes_per_ray\": [0-9.]+|\"tris_per_ray\": [0-9.]+|\"sun_shadow[a-z_]+\": [0-9.]+|\"lbvh_build_ms\": [0-9.]+|\"bvh_refit_ms\": [0-9.]+' | tr '\n' ' '; echo; "
done
/usr/local/graft/bin/gpurun --timeout 900 -- "$cmd" 2>&1 | grep -vE "^\[gpurun\] (sending|merged)"
