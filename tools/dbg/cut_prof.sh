cd /tmp && export TMPDIR=/tmp
R=/root/repo
for v in nocut cut; do
  L=$R/vulkanrenderer_amd/lib/libvkr_rt.so; [ $v = nocut ] && L=$R/vulkanrenderer_amd/lib/variants/libvkr_nocut.so
  echo "== $v phase dump"
  VKR_LIB_PATH=$L VKR_PHASE_DUMP=1 timeout -k 10 200 python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 2>&1 >/dev/null | grep k1_phase | tail -1
  echo "== $v kernel stats"
  rm -rf $R/gpurun_out/cutprof_$v
  VKR_LIB_PATH=$L timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cutprof_$v -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-extra > $R/gpurun_out/cutprof_$v.log 2>&1
  f=$(ls $R/gpurun_out/cutprof_$v/*/*kernel_stats.csv | head -1); head -8 $f | cut -c1-160
done
