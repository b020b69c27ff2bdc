cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_native_exchange_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/t_exchange.log; cat gpurun_out/t_exchange.log
bash scripts/trace_sweep.sh sim8 --simulate-world 8 > gpurun_out/timeline_sim8.txt 2>&1; tail -45 gpurun_out/timeline_sim8.txt
