#!/bin/bash
# usage: tools/grun.sh <timeout-seconds> '<command>'   -- gpurun, waiting (up to ~20 min) while the pool has no free box (exit code 3 only:
# nothing ran and nothing was charged; any other outcome is returned as it is, never retried)
T=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 60
done
exit 3
