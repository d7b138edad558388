#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/aa; mkdir -p $O; export TMPDIR=/tmp
bash tools/profile_fit.sh r04 > /dev/null 2>&1; cp gpurun_out/fit_r04/r04_*.csv gpurun_out/fit_r04/r04_fit_ab.txt $O/; head -5 $O/r04_fit_ab.txt
head -24 $O/r04_fit_step_kernel_stats.csv | cut -c1-130; head -16 $O/r04_fit_step_geometry_kernel_stats.csv | cut -c1-130
