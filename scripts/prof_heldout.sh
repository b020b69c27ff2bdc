#!/bin/bash
# rocprofv3 kernel-trace + stats of scripts/bench_heldout.py; summary under gpurun_out/prof_heldout_<tag>
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_heldout_$tag
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 scripts/bench_heldout.py "$@" > $out/bench.log 2>&1
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
python3 - $out/kernel_stats.csv <<PY
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1]))):
    if "heldout" in r["Name"]:
        print("%-66s calls %5s avg %9.1f us total %8.2f ms" % (r["Name"][:66], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
tail -1 $out/bench.log | cut -c1-600
