"""Compiles the reference's own rasterizer sources for gfx950 (TEST INFRASTRUCTURE).

The three .cu files are read IN PLACE from /root/reference (nothing is copied into the repo) and
compiled by hipcc as HIP through the thin header shim in oracle/ref_build/shim (cuda_runtime.h ->
hip_runtime.h, cub -> hipcub, cooperative_groups -> hip_cooperative_groups); the vendored glm has
native HIP support.  Output: oracle/_ref/libref_surfel.so (git-ignored, travels to the GPU box).
It exists so that (a) the CPU oracle and (b) the product can be compared with the real reference
executing on an MI355X, and so that golden vectors produced by the reference itself can be
committed (tests/golden/ref_*.npz, oracle/ref_build/make_ref_golden.py).  It is a checker, never a
code path of the product, and it is only built where /root/reference exists.
"""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/gs/submodules/diff-surfel-rasterization"
OUT_DIR = os.path.join(os.path.dirname(HERE), "_ref")
OUT = os.path.join(OUT_DIR, "libref_surfel.so")
# Second variant, "strict": the same sources compiled with floating-point contraction OFF and rsqrtf spelled
# 1/sqrtf -- i.e. every fp32 operation is the IEEE operation the source names, in source order.  How a
# compiler contracts a*b+c into FMAs (nvcc and hipcc both do, each its own way) is unspecified, so the
# default build is ONE possible rounding of the reference; the strict build is the reference's arithmetic
# itself, and it is what the CPU oracle (which fixes source order too) must match bit for bit on every
# integer output.  tests/test_gpu_reference.py compares oracle / product with both.
OUT_STRICT = os.path.join(OUT_DIR, "libref_surfel_strict.so")
VARIANTS = {"default": (OUT, []), "strict": (OUT_STRICT, ["-ffp-contract=off", "-DREF_STRICT_RSQRT"])}


def build(force=False):
    outs = [_build_variant(out, extra, force) for out, extra in VARIANTS.values()]
    return outs[0]


def _build_variant(OUT, extra_flags, force=False):
    if not os.path.isdir(REF):
        print("reference sources not present: skipping oracle/_ref build")
        return None
    srcs = [os.path.join(REF, "cuda_rasterizer", f) for f in ("forward.cu", "backward.cu", "rasterizer_impl.cu")]
    srcs.append(os.path.join(HERE, "ref_wrapper.cpp"))
    deps = srcs + [os.path.join(HERE, "shim", "cuda_runtime.h"), os.path.abspath(__file__)]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-munsafe-fp-atomics",
             "-I", os.path.join(HERE, "shim"), "-I", os.path.join(REF, "cuda_rasterizer"),
             "-I", os.path.join(REF, "third_party", "glm")] + list(extra_flags)
    tag = os.path.basename(OUT).rsplit(".", 1)[0]
    objs = []
    for src in srcs:
        obj = os.path.join(OUT_DIR, tag + "_" + os.path.basename(src).rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        text = open(src).read()
        # nvcc accepts the kernel-launch chevrons written with inner spaces ("<< <grid, block >> >");
        # clang does not.  The source passes through this one token-spacing normalisation on its way
        # into the compiler via a temporary file in the system temp directory (hipcc compiles a HIP
        # source twice, host and device, so it cannot read it from a pipe); nothing is written back
        # to the reference and nothing is copied into this repository.
        text = text.replace("<< <", "<<<").replace(">> >", ">>>")
        with tempfile.TemporaryDirectory() as td:
            tmp = os.path.join(td, os.path.basename(src).rsplit(".", 1)[0] + ".hip")
            with open(tmp, "w") as f:
                f.write(text)
            r = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-c", tmp, "-o", obj], capture_output=True,
                               text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError(f"compiling {src} for gfx950 failed")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("linking the reference for gfx950 failed")
    for o in objs:
        os.remove(o)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
