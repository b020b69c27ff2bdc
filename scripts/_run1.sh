cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 scripts/time_config5_exchange.py 4 > gpurun_out/config5_exchange.log 2>&1; tail -4 gpurun_out/config5_exchange.log | cut -c1-600
