#!/bin/bash
# usage (GPU box): tools/dbg/frame_timeline.sh <tag> [env assignments]   -- kernel start/end times of the last frame of a short c3 bench run
tag=$1; shift
R=/root/repo; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/tl_$tag
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl_$tag -- python3 $R/bench.py --steps 4 --warmup 2 --cpu-seconds 0 --no-extra > $O/tl_$tag.log 2>&1 || exit 1
f=$(ls $O/tl_$tag/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_pix", "k_trace_q", "k_atrous", "k_tile_cut"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first_stats = next((i for i, r in enumerate(rows) if "k_trace_q<true" in r["Kernel_Name"]), len(rows))
rows = rows[:max(0, first_stats - 1)]          # drop the bench's ray-counting pass (STATS build of the kernel) and its setup
# the last frame: from the last-but-one k_pix_setup pair on
idx = [i for i, r in enumerate(rows) if "k_pix_setup" in r["Kernel_Name"]]
start = idx[-2] if len(idx) >= 2 and int(rows[idx[-1]]["Start_Timestamp"]) - int(rows[idx[-2]]["Start_Timestamp"]) < 1_000_000 else idx[-1]
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:9.1f} us  q{r.get("Queue_Id", "?")}  {r["Kernel_Name"][:50]}')
PY
