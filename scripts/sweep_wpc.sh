#!/bin/bash
export GGS_DEBUG=1   # the library reads GGS_DEBUG_* only with this opt-in
# waves per CU x ring depth of z_stream1_kernel
cd $GRAFT_REPO_ROOT
while read r k w; do
    echo -n "ring=$r K=$k wpc=$w "; GGS_DEBUG_WPC=$w GGS_HIP_LIB=$PWD/scripts/bin/libggs_ring$r.so timeout -k 10 300 python3 bench.py --topics $k --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['ms_per_step'], l['phase_ms_per_sweep']['z_ms'])"
done <<LIST
2 1024 4
2 1024 5
3 1024 3
3 1024 4
4 1024 3
3 500 4
3 500 5
2 500 6
2 500 5
LIST
