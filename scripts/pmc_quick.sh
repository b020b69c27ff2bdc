#!/bin/bash
# one PMC pass over a short bench run, per-kernel averages: bash scripts/pmc_quick.sh <tag> "<counters>" [bench args...]
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmcq_$tag
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/run -- python3 bench.py --no-cpu-baseline --no-extra-configs --steps 3 --warmup 1 "$@" > $out/run.log 2>&1
f=$(find $out/run -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted(agg):
    if "ggs::" in k and "debug" not in k:
        print(k, {c: round(v / n[(k, c)], 1) for c, v in sorted(agg[k].items())}, "launches=%d" % max(n[(k, c)] for c in agg[k]))
PY
rm -rf $out/run
