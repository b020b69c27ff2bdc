cd $GRAFT_REPO_ROOT
export GGS_DEBUG=1
for scheme in pcgs collapsed; do
for K in 100 128 144 160 168 176 192; do
for wave in 0 1; do
GGS_DEBUG_PCGS_WAVE=$wave python3 bench.py --no-cpu-baseline --no-extra-configs --scheme $scheme --topics $K --steps 5 --warmup 1 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$scheme K=$K wave=$wave', l['ms_per_step'], l['phase_ms_per_sweep']['z_ms'], l['z_step']['z_kernel'][:20])"
done; done; done
