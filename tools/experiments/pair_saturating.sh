# pairs on a cloud whose pixels saturate early (opacity logits shifted up): long lists, short walks
for SH in 0 2 4; do for S0 in 0 8001; do for K in 0 6; do
  echo -n "opacity shift $SH step0=$S0 pair=$K: "
  FIT_OPACITY_SHIFT=$SH VIDU4D_SURFEL_PAIR=$K FIT_STEP0=$S0 FIT_K=100 FIT_NO_TORCH_PROF=1 timeout 300 python tools/fit_profile.py 2>&1 | grep FIT_STEP | sed "s/.*step: //"
done; done; done
