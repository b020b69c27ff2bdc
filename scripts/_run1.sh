cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/t_all.log; cat gpurun_out/t_all.log
python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; python3 -c "
import json
l=json.loads(open('gpurun_out/bench_n1.json').read().strip().splitlines()[-1]); print('N=1', l['value'], l['ms_per_step'], l['phase_ms_per_sweep'], l['z_step']); print({k:(v['ms_per_step'], v['phase_ms_per_sweep']['z_ms']) for k,v in l['extra_configs'].items()})"
for w in 8; do python3 bench.py --simulate-world $w --no-cpu-baseline --no-extra-configs --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sim $w', l['ms_per_step'], l['phase_ms_per_sweep'])"; done
