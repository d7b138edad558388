#!/bin/bash
# FETCH_SIZE calibration for the blend kernels' gather (tools/ubench/fetch_calib.hip); GPU box.  -> gpurun_out/fetch_calib.txt
cd "$(dirname "$0")/.."; R=$(pwd); export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/fetch_calib tools/ubench/fetch_calib.hip 2>/dev/null || exit 1
cd /tmp; rm -rf /tmp/fc
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/fc -o pmc --output-format csv -- /tmp/fetch_calib > /tmp/fc.log 2>&1
cat /tmp/fc.log | grep known
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/fc/**/*counter_collection.csv", recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        acc.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
known = {"k_gather128": (512 << 20) + (512 << 20) // 128 * 4, "k_gather112": (512 << 20) // 112 * 112 + (512 << 20) // 112 * 4, "k_stream": 512 << 20}
for k, v in acc.items():
    kib = sum(v) / len(v)
    name = next((n for n in known if n in k), None)
    if name is None:
        continue
    kn = known[name]
    print(f"{name}: FETCH_SIZE {kib:.0f} KiB per launch ({len(v)} launches), known {kn / 1024:.0f} KiB -> bytes per counted byte {kn / 1024 / kib:.3f}")
PY
