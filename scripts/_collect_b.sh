cd $GRAFT_REPO_ROOT
tag=r04
bash scripts/rehearse_exchange.sh && \
REHEARSE_RANKS="4" bash scripts/rehearse_bench_n.sh > gpurun_out/rehearse_bench_n.log 2>&1; tail -6 gpurun_out/rehearse_bench_n.log
python3 scripts/time_config5_exchange.py 4 > gpurun_out/config5_exchange.log 2>&1; tail -1 gpurun_out/config5_exchange.log | cut -c1-300
for s in pcgs collapsed; do python3 bench.py --scheme $s --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs > gpurun_out/bench_$s.json 2>/dev/null; cut -c1-200 gpurun_out/bench_$s.json; done
for k in 256 500 1024 2048; do python3 bench.py --scheme pcgs --topics $k --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs > gpurun_out/bench_pcgs_k$k.json 2>/dev/null; cut -c1-200 gpurun_out/bench_pcgs_k$k.json; done
python3 bench.py --scheme collapsed --topics 1024 --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs > gpurun_out/bench_collapsed_k1024.json 2>/dev/null
bash scripts/trace_sweep.sh ${tag}_c2 > gpurun_out/timeline_c2.txt 2>&1; bash scripts/trace_sweep.sh ${tag}_sim8 --simulate-world 8 > gpurun_out/timeline_sim8.txt 2>&1
