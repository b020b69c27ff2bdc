#!/bin/bash
export GGS_DEBUG=1 GGS_DEBUG_WARM=3
for a in 0 1024 2048 3072; do
  export GGS_DEBUG_ABLATE=$a
  echo "ablate=$a"
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-configs --steps 10 2>&1 | grep -a "warm trace\|ms_per_step" | sed 's/.*"ms_per_step": \([0-9.]*\).*"z_ms": \([0-9.]*\).*/ms_per_step \1 z_ms \2/' | cut -c1-220
done
