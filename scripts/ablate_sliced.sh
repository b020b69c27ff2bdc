export GGS_DEBUG=1   # the library reads GGS_DEBUG_* only with this opt-in
# timing-only compile-time ablations of z_sliced_kernel (scripts/bin/libggs_abl<N>.so built with -DGGS_ABL=N:
# 1 no DMA, 2 no walk, 4 no score pass); results are wrong on purpose
mkdir -p gpurun_out
for a in ${ABL:-0 1 2 4 8}; do
  lib=scripts/bin/libggs_abl$a.so; [ $a = 0 ] && lib=ldagroupedgibbssampler_amd/csrc/libggs_hip.so
  echo -n "abl=$a "; GGS_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep -o '"avg_launch_ms": [0-9.]*' || echo failed
done
