"""Builds libvidu4d_surfel.so (hand-written HIP for gfx950) in-tree with hipcc.

`python -m vidu4d_amd.build` or `vidu4d_amd.build.build()`.  The shared object lands next to the
sources (vidu4d_amd/csrc/libvidu4d_surfel.so) so that it travels with the tree; it is git-ignored.
Translation units are compiled in parallel and only when their sources changed.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "libvidu4d_surfel.so")
SOURCES = ["preprocess.hip", "binning.hip", "blend.hip", "quaternion.hip", "lbs.hip", "bone_tables.hip", "dense_stack.hip", "knn.hip", "post.hip", "optim.hip", "skin_field.hip", "loss.hip", "contract.hip", "capi.hip"]
HEADERS = ["surfel_math.h", "surfel_state.h", "post_math.h", "wave_utils.h", "wave_reduce.h", "bone_tables_math.h",
           os.path.join(INCLUDE, "vidu4d_surfel.h"),
           os.path.join(INCLUDE, "vidu4d_surfel_diag.h")]
ARCH = "gfx950"
# -munsafe-fp-atomics: hardware global_atomic_add_f32 / ds_add_f32 instead of CAS loops.
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fgpu-rdc" if False else "",
         "-Wall", "-Wno-unused-function", "-I", INCLUDE]
FLAGS = [f for f in FLAGS if f]
# Per-file extras.  blend.hip: clang's SLP vectoriser turns pairs of scalar fp32 ops into v_pk_* but
# pays for it with v_mov_b32 to assemble the register pairs -- 266 vs 248 VALU instructions in the
# backward inner loop (and 125 vs 110 VGPRs); the kernels are VALU-issue bound, so it is switched off.
# lbs.hip: the 64-bone instances of the bone loops cannot be fully unrolled (a warning per loop; the 32-bone ones are).
EXTRA_FLAGS = {"blend.hip": ["-fno-slp-vectorize"], "lbs.hip": ["-Wno-pass-failed"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    headers = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for src in SOURCES:
        spath = os.path.join(CSRC, src)
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, [spath, os.path.abspath(__file__)] + headers):
            jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", spath, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
