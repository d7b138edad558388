for K in 0 6 0 6; do for C in false true; do
echo -n "PAIR=$K captured=$C step0=8001: "
VIDU4D_SURFEL_PAIR=$K FIT_OPTS="{\"captured_step\":$C}" FIT_STEP0=8001 FIT_K=150 FIT_NO_TORCH_PROF=1 timeout 300 python tools/fit_profile.py 2>&1 | grep FIT_STEP | sed "s/.*step: //"
done; done
