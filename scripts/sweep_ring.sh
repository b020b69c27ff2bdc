#!/bin/bash
export GGS_DEBUG=1   # the library reads GGS_DEBUG_* only with this opt-in
# z_stream1_kernel ring depth (scripts/bin/libggs_ring<R>.so built with -DGGS_STREAM1_RING=R) at K=1024 / 500 / 200
cd $GRAFT_REPO_ROOT
for r in ${RINGS:-2 3 4}; do
  for k in ${KS:-1024 200}; do
    echo -n "ring=$r K=$k "; GGS_HIP_LIB=$PWD/scripts/bin/libggs_ring$r.so timeout -k 10 300 python3 bench.py --topics $k --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['ms_per_step'], l['phase_ms_per_sweep'])"
  done
done
