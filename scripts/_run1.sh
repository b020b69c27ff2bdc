cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_collapsed_gpu.py -x -q -k "pcgs or parallel_schedule or wave" 2>&1 | tail -6 > gpurun_out/t_pcgs.log; cat gpurun_out/t_pcgs.log
export GGS_DEBUG=1
for cfg in "pcgs 100 1" "pcgs 200 -" "pcgs 256 -" "collapsed 100 -" "collapsed 128 -" "collapsed 200 -" "collapsed 100 1" "pcgs 1024 -" "collapsed 1024 -"; do
set -- $cfg
if [ "$3" = "-" ]; then unset GGS_DEBUG_PCGS_WAVE; else export GGS_DEBUG_PCGS_WAVE=$3; fi
python3 bench.py --no-cpu-baseline --no-extra-configs --scheme $1 --topics $2 --steps 5 --warmup 1 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', l['ms_per_step'], l['phase_ms_per_sweep']['z_ms'], l['z_step']['z_kernel'][:20])"
done
