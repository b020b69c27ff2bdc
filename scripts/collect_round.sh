#!/bin/bash
# Everything the round's profiles/ and DESIGN.md numbers come from, in one GPU call:
#   profiles of configs 2 (the default command), 3 (K=1024) and the config-4 stand-in; the one-GPU rehearsals of the N>1 paths
tag=${1:-r04}
bash scripts/collect_profiles.sh ${tag} && \
bash scripts/collect_profiles.sh ${tag}_k1024 --topics 1024 --steps 10 --warmup 2 --no-cpu-baseline && \
bash scripts/collect_profiles.sh ${tag}_c4 --docs 18846 --types 60000 --mean-len 150 --topics 200 --no-cpu-baseline && \
bash scripts/rehearse_exchange.sh && \
REHEARSE_RANKS="4" bash scripts/rehearse_bench_n.sh > gpurun_out/rehearse_bench_n.log 2>&1 && tail -12 gpurun_out/rehearse_bench_n.log && \
python3 scripts/time_config5_exchange.py 4 > gpurun_out/config5_exchange.log 2>&1 && tail -1 gpurun_out/config5_exchange.log | cut -c1-400 && \
for s in pcgs collapsed; do python3 bench.py --scheme $s --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs > gpurun_out/bench_$s.json 2>/dev/null; cut -c1-260 gpurun_out/bench_$s.json; done && \
for k in 256 500 1024 2048; do python3 bench.py --scheme pcgs --topics $k --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs > gpurun_out/bench_pcgs_k$k.json 2>/dev/null; cut -c1-260 gpurun_out/bench_pcgs_k$k.json; done && \
python3 bench.py --scheme collapsed --topics 1024 --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs > gpurun_out/bench_collapsed_k1024.json 2>/dev/null && \
bash scripts/trace_sweep.sh ${tag}_c2 > gpurun_out/timeline_c2.txt 2>&1 && bash scripts/trace_sweep.sh ${tag}_sim8 --simulate-world 8 > gpurun_out/timeline_sim8.txt 2>&1 && \
{ [ -x scripts/bin/walk_probe ] && scripts/bin/walk_probe 50000 13 > gpurun_out/walk_probe.txt && scripts/bin/walk_probe 50000 100 >> gpurun_out/walk_probe.txt; true; }
