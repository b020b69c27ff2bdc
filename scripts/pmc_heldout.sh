#!/bin/bash
# SQ counter passes of the held-out kernel (each pass its own run, kernel-trace only)
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_heldout_$tag
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pmc$i -- python3 scripts/bench_heldout.py --cpu-docs 50 > $out/pmc$i.log 2>&1
  f=$(find $out/pmc$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $out/pmc_counters.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted(agg):
    if "heldout_particles" in k:
        print(k, {c: round(v / n[(k, c)], 1) for c, v in sorted(agg[k].items())}, "launches=%d" % max(n[(k, c)] for c in agg[k]))
PY
done
rm -rf $out/pmc*/ 2>/dev/null
cat $out/pmc_counters.txt
