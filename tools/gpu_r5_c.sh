#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5c; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python tools/ref_parity_report.py --configs small,cfgA,cfgB,cfgE_slice --out $O/ref_parity_probe.json > $O/refparity.log 2>&1; tail -3 $O/refparity.log | cut -c1-300
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5c/ref_parity_probe.json"))
for cfg in ("small","cfgA","cfgB","cfgE_slice"):
    for pair,fl in (("product_vs_oracle", d[cfg]["product_vs_oracle"]["floats"]), ("product_vs_strict", d[cfg]["strict"]["product_vs_ref"]["floats"]), ("oracle_vs_strict", d[cfg]["strict"]["oracle_vs_ref"]["floats"]), ("product_vs_default", d[cfg]["default"]["product_vs_ref"]["floats"])):
        print(cfg, pair, {k:(float("%.2e"%v["rel_l2"]), float("%.2e"%v["rel_l2_without_outliers"])) for k,v in fl.items()})
PY
