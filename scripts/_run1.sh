cd $GRAFT_REPO_ROOT
export GGS_DEBUG=1
GGS_DEBUG_NO_OVERLAP=1 bash scripts/trace_sweep.sh sim8_noov --simulate-world 8 > gpurun_out/timeline_sim8_noov.txt 2>&1; tail -32 gpurun_out/timeline_sim8_noov.txt
for pc in 1 3 6; do GGS_DEBUG_PHICOLS=$pc python3 bench.py --simulate-world 8 --no-cpu-baseline --no-extra-configs --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('phicols $pc', l['ms_per_step'], l['phase_ms_per_sweep'])"; done
