#!/bin/bash
export GGS_DEBUG=1   # the library reads GGS_DEBUG_* only with this opt-in
# How the phases of a one-rank RCCL run (bench.py --force-sharded) react to the stream / hardware-queue settings
cd $GRAFT_REPO_ROOT; out=gpurun_out/probe_queues; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 300 env "$@" python3 bench.py --force-sharded --no-cpu-baseline --steps 10 $EXTRA > $out/$tag.json 2> $out/$tag.err || { tail -3 $out/$tag.err; return; }
  python3 -c "
import json
l=json.loads(open('$out/$tag.json').read().strip().splitlines()[-1]); print('$tag', l['ms_per_step'], l['phase_ms_per_sweep'])"; }
run default X=1
run null_stream GGS_DEBUG_OWN_STREAM=0
run own_high GGS_DEBUG_OWN_STREAM=2
run hwq2 GPU_MAX_HW_QUEUES=2
run hwq3 GPU_MAX_HW_QUEUES=3
run hwq1 GPU_MAX_HW_QUEUES=1
run no_overlap GGS_DEBUG_NO_OVERLAP=1
run no_split GGS_DEBUG_SPLIT=0
run null_nosplit GGS_DEBUG_OWN_STREAM=0 GGS_DEBUG_SPLIT=0
