#!/bin/bash
# Collects one configuration's profile evidence on the GPU box into gpurun_out/profiles_<tag>/ :
#   bench.json         the bench line of the un-profiled run (carries config.workload and build.csrc_sha16)
#   kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command
#   pmc_counters.txt   PMC passes (each in its own run, kernel-trace only -- never with other trace domains); its
#                      first line stamps the workload string and the csrc hash of the run, and bench.py only reports
#                      roofline.traffic from a file whose stamp matches the run it is in
# usage: bash scripts/collect_profiles.sh <tag> [bench args...]      e.g.  r02_k1024 --topics 1024
tag=${1:-r02}; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_$tag
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 bench.py --no-extra-configs "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
tail -c 900 $out/bench.json; echo
python3 - $out/bench.json > $out/pmc_counters.txt <<'PY'
import json, sys
line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("# workload: %s | csrc_sha16: %s" % (line["config"]["workload"], line.get("build", {}).get("csrc_sha16", "?")))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --no-extra-configs "$@" > $out/bench_profiled.log 2>&1
find $out/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pmc$i -- python3 bench.py --no-cpu-baseline --no-extra-configs --steps 3 --warmup 1 "$@" > $out/pmc$i.log 2>&1
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
rm -rf $out/trace/*/*.db $out/pmc*/ $out/trace 2>/dev/null
head -12 $out/kernel_stats.csv | cut -c1-160
grep "z_sliced\|z_hot\|z_stream\|z_kernel" $out/pmc_counters.txt | cut -c1-400
