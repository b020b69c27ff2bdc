export GGS_DEBUG=1   # the library reads GGS_DEBUG_* only with this opt-in
# z-kernel tile-size / residency sweep (timing only)
for cfg in "31 6" "31 5" "24 8" "24 6" "40 4" "48 4" "63 3" "16 8" "31 12"; do set -- $cfg; echo -n "T=$1 WPC=$2 "; GGS_DEBUG_TILE=$1 GGS_DEBUG_WPC=$2 timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep -o '"phase_ms_per_sweep.*' ; done
