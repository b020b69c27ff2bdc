#!/bin/bash
# rocprofv3 kernel-trace + stats of a short bench run; summary copied under gpurun_out/
# usage: bash scripts/prof_stats.sh <tag> [bench args...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --no-cpu-baseline "$@" > $out/bench.log 2>&1
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
column -s, -t < $out/kernel_stats.csv | cut -c1-200 | head -30
tail -1 $out/bench.log | cut -c1-400
