export GGS_DEBUG=1   # the library reads GGS_DEBUG_* only with this opt-in
# timing-only ablations of the z kernel (results are wrong on purpose): GGS_DEBUG_ABLATE bits 2 no walk, 4 no staging, 8 no sum pass
for a in ${ABLATE_SET:-0 2 4 8 6 10 12 14}; do echo -n "ablate=$a "; GGS_DEBUG_ABLATE=$a timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep -o '"phase_ms_per_sweep.*' ; done
