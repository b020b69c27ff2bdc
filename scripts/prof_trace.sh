#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python script; prints per-launch durations in order
# usage: bash scripts/prof_trace.sh <tag> <script.py> [args...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/trace_$tag
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 "$@" > $out/run.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    print("%-60s %10.3f us" % (r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
tail -12 $out/run.log
