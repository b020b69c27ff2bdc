#!/bin/bash
# timing-only ablations of z_warm_kernel (GGS_DEBUG_ABLATE 16 / 32 / 64): kernel durations from a trace, three tiers
export GGS_DEBUG=1 GGS_DEBUG_WARM=3
for a in 0 256 512; do
  export GGS_DEBUG_ABLATE=$a
  timeout -k 10 200 bash scripts/trace_sweep.sh abl$a > gpurun_out/warm_abl_$a.txt 2>&1
  echo "ablate=$a"; grep "z_" gpurun_out/warm_abl_$a.txt | tail -3
done
