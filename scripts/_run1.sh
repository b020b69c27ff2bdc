cd $GRAFT_REPO_ROOT
export GGS_DEBUG=1
run() { python3 bench.py --no-cpu-baseline --no-extra-configs --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['ms_per_step'], l['phase_ms_per_sweep'], l['z_step']['z_form'], l['z_step']['num_hot'])"; }
run "baseline"
GGS_DEBUG_SPLIT=0 run "fused"
GGS_DEBUG_HOT=36 run "split hot36"
GGS_DEBUG_SPLIT=0 GGS_DEBUG_HOT=36 run "fused hot36"
GGS_DEBUG_THETA_EARLY=1 GGS_DEBUG_HOT=36 GGS_DEBUG_SPLIT=0 run "EARLY16 fused hot36"
GGS_DEBUG_THETA_EARLY=1 GGS_DEBUG_HOT=36 GGS_DEBUG_SPLIT=2 run "EARLY16 split hot36"
GGS_DEBUG_THETA_EARLY=8 GGS_DEBUG_HOT=46 GGS_DEBUG_SPLIT=0 run "EARLY8 fused hot46"
GGS_DEBUG_THETA_EARLY=8 GGS_DEBUG_HOT=46 GGS_DEBUG_SPLIT=2 run "EARLY8 split hot46"
GGS_DEBUG_THETA_EARLY=32 GGS_DEBUG_HOT=20 GGS_DEBUG_SPLIT=0 run "EARLY32 fused hot20"
GGS_DEBUG_THETA_EARLY=1 run "EARLY16 default hot (no LDS beside z)"
