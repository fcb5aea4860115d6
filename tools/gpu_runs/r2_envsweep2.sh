B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 10"
run() { name=$1; shift; env "$@" timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %.3f ms/step' % ('$name', d['ms_per_step']))"; }
for r in 1 2; do
run default X=1
run halo_min_wg_512 GDRN_HALO_MIN_WG=512
run halo_min_wg_300 GDRN_HALO_MIN_WG=300
run wgrad_blocks_1280 GDRN_WGRAD_BLOCKS=1280
run wgrad_blocks_1792 GDRN_WGRAD_BLOCKS=1792
done
