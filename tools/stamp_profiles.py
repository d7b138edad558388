#!/usr/bin/env python
"""Stamps profiles/pmc_traffic.json with the tree and the day its counters were collected on (bench.py replays its
`roofline.traffic` / `roofline.limiter` from this file and says so: `roofline.replayed_from`).  Run in the build container right
after the summaries of a `tools/gpu_run.sh profile <tag>` call have been copied into profiles/ (the GPU box has no .git):
    python tools/stamp_profiles.py <tag> [commit]      (commit: the tree that was measured, default HEAD)"""
import datetime, json, os, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=root, text=True).strip()
path = os.path.join(root, "profiles", "pmc_traffic.json")
d = json.load(open(path))
d["_meta"] = {"commit": commit, "date": datetime.date.today().isoformat(), "tag": tag,
              "command": "tools/profile_round.sh %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* / TCC_* passes of bench.py's headline step, "
                         "separate runs; FETCH x 2 per MI355X_MICROARCH.md's calibration)" % tag}
json.dump(d, open(path, "w"), indent=1)
print("stamped", path, d["_meta"])
