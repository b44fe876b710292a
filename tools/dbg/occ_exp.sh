for x in 0 1800 3700 6600 10000; do
  echo "EXTRA_LDS=$x"; VKR_K1_EXTRA_LDS=$x timeout -k 10 200 python -u bench.py --steps 10 --warmup 3 --cpu-seconds 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernels']['trace_ms'])" || exit 1
done
