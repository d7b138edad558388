#!/bin/bash
# SQ counters of the fit step's blend kernels (tools/fit_profile.py, regime 0)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/x; mkdir -p $O; export TMPDIR=/tmp
SUM='import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        for key in ("blend_bwd_kernel","blend_fwd_kernel","blend_combine_kernel","blend_seg_T"):
            if key in k:
                acc[key][r["Counter_Name"]]+=float(r["Counter_Value"]); n[key][r["Counter_Name"]]+=1
for key in acc: print(key, {c: round(v/n[key][c]) for c,v in sorted(acc[key].items())}, "launches", max(n[key].values()))'
cd /tmp
FIT_K=10 FIT_NO_TORCH_PROF=1 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --kernel-trace -d $O/sq1 -o pmc --output-format csv -- python $R/tools/fit_profile.py > $O/sq1.log 2>&1
python -c "$SUM" $O/sq1
FIT_K=10 FIT_NO_TORCH_PROF=1 rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE \
    --kernel-trace -d $O/sq2 -o pmc --output-format csv -- python $R/tools/fit_profile.py > $O/sq2.log 2>&1
python -c "$SUM" $O/sq2; tail -3 $O/sq2.log | cut -c1-200
rm -rf $O/sq1 $O/sq2
