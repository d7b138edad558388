QUICK="--cpu-images 0 --torch-cpu-images 0 --fit-densify-steps 0 --per-frame-surface 0 --host-probe 0 --fit-optim-warp 0"
for K in 0 6 0 6; do
  echo -n "PAIR=$K: "
  VIDU4D_SURFEL_PAIR=$K timeout 600 python bench.py $QUICK --fit-steps ${FS:-100} --repeats 3 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['repeats'], {k: round(v,4) for k,v in d['stage_ms_avg'].items()}, d['fit_step']['images_per_s'], d['fit_step_geometry']['images_per_s'])"
done
