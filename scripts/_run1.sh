cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_collapsed_gpu.py -x -q -k "pcgs or parallel_schedule or wave" 2>&1 | tail -6 > gpurun_out/t_pcgs.log; cat gpurun_out/t_pcgs.log
for lib in "" "$GRAFT_REPO_ROOT/scripts/bin/libggs_d16_0.so"; do
for args in "--scheme pcgs --topics 2048 --steps 3 --warmup 1" "--scheme collapsed --topics 2048 --steps 3 --warmup 1" "--scheme pcgs --topics 4096 --steps 2 --warmup 1" "--scheme collapsed --topics 4096 --steps 2 --warmup 1"; do
GGS_HIP_LIB=$lib python3 bench.py --no-cpu-baseline --no-extra-configs $args 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$lib $args', l['ms_per_step'], l['phase_ms_per_sweep']['z_ms'])"
done; done
