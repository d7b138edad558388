# the fitting step on balls of radius 0.5 .. 1.0: segment-parallel forward (the rule's choice at small radii) against whole
# tiles with and without paired workgroups
for R in 0.5 0.7 0.85 1.0; do for S0 in 0 8001; do
  for CFG in "auto 0" "0 0" "0 6" "1 0"; do set -- $CFG
    echo -n "radius $R step0=$S0 split=$1 pair=$2: "
    VIDU4D_SURFEL_SPLIT=$1 VIDU4D_SURFEL_PAIR=$2 FIT_STEP0=$S0 FIT_K=100 FIT_NO_TORCH_PROF=1 timeout 300 python tools/fit_profile.py $R 2>&1 | grep FIT_STEP | sed "s/.*step: //"
  done
done; done
