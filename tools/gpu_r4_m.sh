#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/profile_round.sh r04 2>&1 | tail -3
cp gpurun_out/prof_r04/summary/* gpurun_out/ 2>/dev/null
ls gpurun_out/prof_r04/summary
