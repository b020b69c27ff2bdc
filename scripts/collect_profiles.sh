#!/bin/bash
# Collects the round's profile evidence on the GPU box into gpurun_out/profiles_<tag>/ :
#   kernel_stats.csv   rocprofv3 --kernel-trace --stats of the default bench command
#   pmc_*.txt          PMC passes (each in its own run, kernel-trace only -- never with other trace domains)
#   bench.json         the bench line of the un-profiled run
# usage: bash scripts/collect_profiles.sh <tag>
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_$tag
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench.json 2> $out/bench.err
tail -c 600 $out/bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline > $out/bench_profiled.log 2>&1
find $out/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pmc$i -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $out/pmc$i.log 2>&1
  f=$(find $out/pmc$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $out/pmc_counters.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted(agg):
    if "ggs::" in k and "debug" not in k:
        print(k, {c: round(v / n[(k, c)], 1) for c, v in sorted(agg[k].items())}, "launches=%d" % max(n[(k, c)] for c in agg[k]))
PY
done
rm -rf $out/trace/*/*.db $out/pmc*/ 2>/dev/null
cat $out/pmc_counters.txt | grep "z_sliced\|z_kernel" | cut -c1-400
