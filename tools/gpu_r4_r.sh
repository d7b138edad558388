#!/bin/bash
# SQ counters of blend_bwd: product (quadrant walk) vs variants/half.so, plus the walk statistics of both
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/r; mkdir -p $O; export TMPDIR=/tmp
cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so
SUM='import csv,glob,sys,collections
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "blend_bwd_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
print(sys.argv[2], {k: round(v/n[k]) for k,v in sorted(acc.items())}, "launches", max(n.values()) if n else 0)'
for v in product half; do
  if [ $v = product ]; then cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so; else cp variants/$v.so vidu4d_amd/csrc/libvidu4d_surfel.so; fi
  cd /tmp
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --kernel-trace -d $O/sq1_$v -o pmc --output-format csv -- \
    python $R/bench.py --steps 4 --warmup 2 --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 0 --per-frame-surface 0 --no-stage-timers > $O/sq1_$v.log 2>&1
  python -c "$SUM" $O/sq1_$v $v
  rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE \
    --kernel-trace -d $O/sq2_$v -o pmc --output-format csv -- \
    python $R/bench.py --steps 4 --warmup 2 --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 0 --per-frame-surface 0 --no-stage-timers > $O/sq2_$v.log 2>&1
  python -c "$SUM" $O/sq2_$v $v; tail -2 $O/sq2_$v.log | cut -c1-300
  cd $R
  timeout 300 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 1 --per-frame-surface 0 2>/dev/null | python -c '
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["value"]), d["stage_ms_avg"]["blend_bwd"], json.dumps(d["roofline"]["limiter"].get("tile_walk")))' $v
  rm -rf $O/sq1_$v $O/sq2_$v
done
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
