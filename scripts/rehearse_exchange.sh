#!/bin/bash
# One-GPU rehearsals of the N>1 bench paths (none of them is a multi-GPU number):
#   1  one rank through the real RCCL provider (--force-sharded)
#   2  two ranks on the one GPU over the callback provider (gloo, host staged): the whole `bench.py --gpus 2` flow
#   3  per-rank phase times of an N-way strong split, peers missing (--simulate-world N): what DESIGN.md section 6 budgets with
out=$GRAFT_REPO_ROOT/gpurun_out/rehearse; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 bench.py --force-sharded --no-cpu-baseline --steps 10 > $out/rccl_one_rank.json 2> $out/rccl_one_rank.err || { tail -5 $out/rccl_one_rank.err; exit 1; }
cut -c1-700 $out/rccl_one_rank.json
timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --backend gloo --single-device --steps 5 --warmup 1 > $out/gloo_two_ranks.json 2> $out/gloo_two_ranks.err || { tail -5 $out/gloo_two_ranks.err; exit 1; }
tail -1 $out/gloo_two_ranks.json | cut -c1-900
for n in 2 4 8; do
  timeout -k 10 300 python3 bench.py --simulate-world $n --simulate-rank 0 --no-cpu-baseline --steps 20 > $out/simulate_$n.json 2> $out/simulate_$n.err || { tail -5 $out/simulate_$n.err; exit 1; }
  python3 -c "
import json,sys
l=json.loads(open('$out/simulate_$n.json').read().strip().splitlines()[-1])
print('simulated rank 0 of $n:', l['ms_per_step'], 'ms/sweep', l['phase_ms_per_sweep'])"
done
