#!/bin/bash
# Rehearsals of `bench.py --gpus N` on the ONE GPU a builder has (the 8-GPU run is the driver's):
#   force_sharded   one rank through the real RCCL provider (ncclCommInitRank, reduce-scatter, all-gathers)
#   gloo_N          N ranks on one device over the callback provider (RCCL refuses two ranks per device)
# Each writes its JSON line (with parity_vs_one_gpu) and its stage lines under gpurun_out/rehearse/.
out=$GRAFT_REPO_ROOT/gpurun_out/rehearse
mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 bench.py --force-sharded --steps 10 --warmup 3 > $out/force_sharded.json 2> $out/force_sharded.err || { tail -20 $out/force_sharded.err; exit 1; }
for n in ${REHEARSE_RANKS:-2 4}; do
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) bench.py --gpus $n --backend gloo --single-device \
      --steps 10 --warmup 3 --no-weak-leg --no-large-leg > $out/gloo_$n.json 2> $out/gloo_$n.err || { tail -30 $out/gloo_$n.err; exit 1; }
done
for f in $out/*.json; do echo "== $f"; python3 - $f <<'PY'
import json, sys
l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: l.get(k) for k in ("value", "ms_per_step", "n_gpus", "parity_vs_one_gpu", "exchange", "z_step")})
print(l.get("verification"))
PY
done
grep -h "bench " $out/*.err | tail -40
