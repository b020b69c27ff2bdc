#!/bin/bash
# timing-only experiment GGS_DEBUG_THETA_TAIL_PCT (ggs_api.hip): the next theta of the first p % of the documents on the table
# kernels' stream behind z_warm_kernel, beside the cold kernel's tail; results wrong on purpose
cd $GRAFT_REPO_ROOT
export GGS_DEBUG=1
for p in 0 10 20 30 40 0; do
  for own in 0 1; do
    export GGS_DEBUG_THETA_TAIL_PCT=$p GGS_DEBUG_THETA_TAIL_STREAM=$own
    echo "tail_pct=$p own_stream=$own"
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extra-configs --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['ms_per_step'], l['phase_ms_per_sweep'])"
  done
done
export GGS_DEBUG_THETA_TAIL_PCT=20 GGS_DEBUG_THETA_TAIL_STREAM=1
timeout -k 10 200 bash scripts/trace_sweep.sh tail30 2>&1 | tail -24 | cut -c1-150
