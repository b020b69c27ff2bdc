#!/bin/bash
# kernel timeline of the last sweeps of a short bench run: start offset, duration and the gap to the previous kernel (us)
# usage: bash scripts/trace_sweep.sh <tag> [bench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/trace_$tag
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --no-extra-configs --steps 5 --warmup 2 "$@" > $out/run.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-70:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f  dur %8.1f  gap %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
    prev_end = max(prev_end, e)
PY
