cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/bench_default_full.json 2> gpurun_out/bench_default_full.err; python3 -c "
import json
l=json.loads(open('gpurun_out/bench_default_full.json').read().strip().splitlines()[-1]); print('N=1', l['value'], l['ms_per_step'], l['roofline']['frac'], l['roofline']['traffic'], l['roofline']['traffic_source']); print({k:(v['ms_per_step'], v['roofline']['frac'], v['roofline']['traffic_source']) for k,v in l['extra_configs'].items()}); print(l['cpu_baseline']['value'], l['cpu_baseline']['cores'], l['cpu_baseline']['variants']['cpu_ref_1t'])"
python3 scripts/soak_parity.py --sweeps 10 --topics 100 > gpurun_out/soak_k100.log 2>&1; tail -3 gpurun_out/soak_k100.log
python3 scripts/soak_parity.py --sweeps 3 --topics 1024 > gpurun_out/soak_k1024.log 2>&1; tail -2 gpurun_out/soak_k1024.log
python3 scripts/soak_parity.py --sweeps 2 --topics 1024 --scheme pcgs > gpurun_out/soak_pcgs_k1024.log 2>&1; tail -2 gpurun_out/soak_pcgs_k1024.log
python3 scripts/soak_parity.py --sweeps 3 --topics 200 --scheme pcgs > gpurun_out/soak_pcgs_k200.log 2>&1; tail -2 gpurun_out/soak_pcgs_k200.log
python3 scripts/soak_parity.py --sweeps 4 --topics 100 --scheme pcgs > gpurun_out/soak_pcgs_k100.log 2>&1; tail -2 gpurun_out/soak_pcgs_k100.log
