#!/bin/bash
# Copies what scripts/_collect_{a,b,c}.sh (= scripts/collect_round.sh in three calls) left under gpurun_out/ into profiles/<tag>_*.
# usage: bash scripts/install_profiles.sh [tag] [a|b|c ...]      (default: r04, all three parts that are present)
tag=${1:-r04}; shift
parts=${*:-a b c}
g=gpurun_out
cd "$(dirname "$0")/.."
for part in $parts; do
  case $part in
  a) for t in "" _k1024 _c4; do for f in bench.json kernel_stats.csv pmc_counters.txt; do cp $g/profiles_${tag}$t/$f profiles/${tag}${t}_$f; done; done
     cp ldagroupedgibbssampler_amd/csrc/ggs_resource_summary.txt profiles/${tag}_resource_summary.txt ;;
  b) for f in bench_pcgs bench_collapsed bench_pcgs_k256 bench_pcgs_k500 bench_pcgs_k1024 bench_pcgs_k2048 bench_collapsed_k1024; do cp $g/$f.json profiles/${tag}_$f.json; done
     for r in rccl_one_rank force_sharded gloo_two_ranks gloo_4 simulate_2 simulate_4 simulate_8; do cp $g/rehearse/$r.json profiles/${tag}_rehearse_$r.json; done
     for r in force_sharded gloo_two_ranks gloo_4; do cp $g/rehearse/$r.err profiles/${tag}_rehearse_$r.stages.txt; done
     cp $g/timeline_c2.txt profiles/${tag}_timeline_c2.txt; cp $g/timeline_sim8.txt profiles/${tag}_timeline_simulate_8.txt
     cp $g/config5_exchange.log profiles/${tag}_config5_exchange_dense_vs_sparse.log ;;
  c) cp $g/bench_default_full.json profiles/${tag}_bench_default_full.json
     for s in k100 k1024 pcgs_k100 pcgs_k200 pcgs_k1024; do cp $g/soak_$s.log profiles/${tag}_soak_$s.log; done ;;
  esac
done
python3 - $tag <<'PY'
import json, sys
tag = sys.argv[1]
def last(f):
    return json.loads(open(f).read().strip().splitlines()[-1])
l = last("profiles/%s_bench.json" % tag)
print("config 2:", l["value"], l["ms_per_step"], l["phase_ms_per_sweep"], l["build"])
for r in ["gloo_two_ranks", "gloo_4", "force_sharded", "rccl_one_rank", "simulate_2", "simulate_4", "simulate_8"]:
    try:
        l = last("profiles/%s_rehearse_%s.json" % (tag, r))
        print(r, l["ms_per_step"], l.get("parity_vs_one_gpu"), l.get("exchange", {}).get("rccl_nranks"), l.get("z_step", {}).get("z_form"))
    except Exception as e:      # noqa: BLE001
        print(r, "missing:", e)
print(open("profiles/%s_pmc_counters.txt" % tag).readline().strip()[-40:])
PY
