QUICK="--cpu-images 0 --torch-cpu-images 0 --per-frame-surface 0 --host-probe 0 --fit-optim-warp 0 --repeats 0 --steps 3 --warmup 2"
for CFG in "6 1.25" "6 2.0" "0 2.0" "6 1.25" "6 2.0" "0 2.0"; do set -- $CFG
  echo -n "pair=$1 split threshold=$2: "
  VIDU4D_SURFEL_PAIR=$1 VIDU4D_SURFEL_SPLIT_AUTO_TILES_PER_CU=$2 timeout 600 python bench.py $QUICK --fit-steps 60 --fit-densify-steps 320 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fit', round(d['fit_step']['images_per_s']), 'geometry', round(d['fit_step_geometry']['images_per_s']), 'densify', round(d['fit_step_densify']['images_per_s']), {k:v for k,v in d['fit_step_densify'].items() if k not in ('images_per_s','ms_per_step','what')})"
done
