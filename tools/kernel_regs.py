#!/usr/bin/env python
"""Register / LDS budget of every kernel of one translation unit (hipcc -S of the file, gfx950).
Usage: python tools/kernel_regs.py vidu4d_amd/csrc/blend.hip [extra hipcc flags]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
    extra = ["-fno-slp-vectorize"] if src.endswith("blend.hip") else []
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-I",
                           os.path.join(ROOT, "include"), "-S", "--cuda-device-only", src, "-o", out] + extra + sys.argv[2:],
                          stderr=subprocess.DEVNULL)
    s = open(out).read()
    for b in s.split("  - .agpr_count:")[1:]:
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", b).group(1)  # noqa: E731
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("surfel::", "")
        print(f"{name:58s} vgpr {g('vgpr_count'):>3} sgpr {g('sgpr_count'):>3} lds {g('group_segment_fixed_size'):>6} "
              f"spill {g('vgpr_spill_count')}")
    print("asm:", out)


if __name__ == "__main__":
    main()
