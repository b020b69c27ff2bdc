#!/bin/bash
# rocprofv3 PMC passes on a short bench run (counters in their own runs, kernel-trace only).
# usage: bash scripts/prof_pmc.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
i=0
for ctrs in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p$i -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-40:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k in agg:
    if "z_kernel" in k or "chain" in k or "theta" in k or "count_sorted" in k:
        print(k, {c: round(v / n[(k, c)], 1) for c, v in agg[k].items()})
PY
done
