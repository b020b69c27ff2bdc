cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_native_exchange_gpu.py tests/test_configs_gpu.py tests/test_collapsed_gpu.py tests/test_distributed_gpu.py -x -q -k "not config5 and not config3" 2>&1 | tail -6 > gpurun_out/t_x.log; cat gpurun_out/t_x.log
for w in 8 4 2; do python3 bench.py --simulate-world $w --no-cpu-baseline --no-extra-configs --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sim $w', l['ms_per_step'], l['phase_ms_per_sweep'])"; done
bash scripts/trace_sweep.sh sim8 --simulate-world 8 > gpurun_out/timeline_sim8.txt 2>&1; tail -36 gpurun_out/timeline_sim8.txt
