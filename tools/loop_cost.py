#!/usr/bin/env python
"""Static issue-cost estimate of a kernel's innermost hot loop (hipcc -S of blend.hip, gfx950).

Usage: python tools/loop_cost.py <asm.s> <mangled-kernel-substring> <marker-instruction>
Finds the basic-block range of the kernel that holds <marker-instruction> (e.g. v_permlane16_swap), walks back to the loop
header label that branches reach and forward to the loop's last back-edge, and counts the VALU instructions by issue class
with the costs measured in round 3 (plain 2.5 cycles, SGPR-mask / VOPC 4.7, DPP 6, lane swap / transcendental 8).
"""
import re, sys

COST = {"plain": 2.5, "mask": 4.7, "cmp": 4.7, "dpp": 6.0, "swap": 8.0, "trans": 8.0}


def classify(line):
    op = line.split()[0]
    if not op.startswith("v_"):
        return None
    if "permlane" in op:
        return "swap"
    if "_dpp" in op or " row_" in line or "quad_perm" in line:
        return "dpp"
    if op.startswith(("v_rcp", "v_exp", "v_log", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_cmp"):
        return "cmp"
    if op.startswith("v_cndmask") or re.search(r"\bs\[\d+:\d+\]|\bvcc\b", line):
        return "mask"
    if op.startswith("v_readfirstlane") or op.startswith("v_readlane"):
        return "mask"
    return "plain"


def main():
    path, kern, marker = sys.argv[1], sys.argv[2], sys.argv[3]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kern in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    m = next(i for i, l in enumerate(body) if marker in l)
    a = max(i for i in range(m) if "s_ff1_i32_b64" in body[i])          # the loop header: next set bit of the cull mask
    a = max(i for i in range(a) if body[i].startswith(".LBB"))
    b = next(i for i in range(m, len(body)) if "s_barrier" in body[i])  # the batch's closing barrier
    counts, n_s, n_ds = {}, 0, 0
    for l in body[a:b]:
        l = l.strip()
        if not l or l.startswith(";") or l.startswith("."):
            continue
        l = l.split(";")[0].strip()
        if not l:
            continue
        c = classify(l)
        if c:
            counts[c] = counts.get(c, 0) + 1
        elif l.startswith("s_"):
            n_s += 1
        elif l.startswith("ds_"):
            n_ds += 1
    tot = sum(counts.values())
    cyc = sum(COST[k] * v for k, v in counts.items())
    print(f"{kern}: VALU {tot} {counts} -> ~{cyc:.0f} issue cycles; SALU {n_s}; LDS {n_ds}")


if __name__ == "__main__":
    main()
