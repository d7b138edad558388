for pad in 0 12000 20000 32000 60000 140000; do
  echo -n "pad $pad: "
  VIDU4D_BWD_PAD_LDS=$pad python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 0 --per-frame-surface 0 --steps 20 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), {k:round(v,4) for k,v in d['stage_ms_avg'].items() if 'blend' in k})"
done
