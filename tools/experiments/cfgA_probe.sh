QUICK="--cpu-images 0 --torch-cpu-images 0 --per-frame-surface 0 --host-probe 0 --fit-optim-warp 0 --repeats 0 --steps 3 --warmup 2 --fit-densify-steps 0"
for CFG in "6 1.25" "0 2.0" "6 2.0" "0 1.25" "6 1.25" "0 2.0"; do set -- $CFG
  echo -n "cfgA pair=$1 split threshold=$2: "
  VIDU4D_SURFEL_PAIR=$1 VIDU4D_SURFEL_SPLIT_AUTO_TILES_PER_CU=$2 timeout 600 python bench.py --surfels 50000 --res 256 --frames 32 $QUICK --fit-steps 200 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('op', round(d['value']), 'fit', round(d['fit_step']['images_per_s']), 'geometry', round(d['fit_step_geometry']['images_per_s']), 'captured', round(d.get('fit_step_captured',{}).get('images_per_s',0)))"
done
