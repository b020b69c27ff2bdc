cd $GRAFT_REPO_ROOT
export GGS_DEBUG=1
run() { python3 bench.py --no-cpu-baseline --no-extra-configs --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=l['phase_ms_per_sweep']; print('$1', l['ms_per_step'], 'z', p['z_ms'], 'theta', p['theta_ms'], 'phi', p['phi_ms'], 'merge', p['merge_ms'], l['z_step']['num_hot'])"; }
run base
run base2
for b in 8 16 32; do for w in 4 5 6 8; do GGS_DEBUG_THETA_B=$b GGS_DEBUG_THETA_WGS=$w run "thetaB=$b wgs=$w"; done; done
for hrows in 64 80; do GGS_DEBUG_HOT=$hrows run "hot=$hrows"; done
GGS_DEBUG_THETA_MAIN=0 run "theta on side"
