for a in 0 1 3 7 15 2 4; do echo "ablate=$a"; GGS_DEBUG_ABLATE=$a timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep -o '"phase_ms_per_sweep.*' ; done
