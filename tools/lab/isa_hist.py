#!/usr/bin/env python3
"""Instruction histogram of one kernel of a hipcc -S listing, by basic block (label to label) so that the steady-state loop can be
read off:  python tools/lab/isa_hist.py kernels_bf16.s 'xyt32_bf16_kernel<true, 3, 0, true, 64'  (CPU only; feeds DESIGN's
issue-cycle estimates: VALU 4 clk per wave64 instruction, transcendentals 16, MFMA 32x32x16 bf16 8 issue + 32 pipe)."""
import collections
import re
import subprocess
import sys

TRANS = ("v_rcp", "v_log", "v_exp", "v_sqrt", "v_rsq", "v_sin", "v_cos")


def cat(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_pk_"):
        return "vpk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, want = sys.argv[1], sys.argv[2]
    detail = len(sys.argv) > 3 and sys.argv[3]
    lines = open(path).read().split("\n")
    start = None
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout
            if want in name:
                start = i
                break
    assert start is not None, "kernel not found"
    blocks, cur, label = [], collections.Counter(), "entry"
    ops = collections.Counter()
    for ln in lines[start + 1:]:
        if ln.startswith(".Lfunc_end") or ln.strip().startswith("s_endpgm"):
            pass
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            blocks.append((label, cur))
            cur, label = collections.Counter(), m.group(1)
            continue
        if ln.startswith(".Lfunc_end"):
            break
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        if not re.match(r"^[a-z]", op):
            continue
        cur[cat(op)] += 1
        if detail and label == detail:
            ops[op] += 1
    blocks.append((label, cur))
    keys = ["mfma", "trans", "vpk", "valu", "lds", "vmem", "salu", "wait", "barrier", "other"]
    print(f"{'block':>12} " + " ".join(f"{k:>7}" for k in keys) + "   total")
    for label, c in blocks:
        tot = sum(c.values())
        if tot < 40:
            continue
        print(f"{label:>12} " + " ".join(f"{c[k]:7d}" for k in keys) + f"   {tot}")
    if detail:
        for op, n in ops.most_common():
            print(f"  {n:5d} {op}")


if __name__ == "__main__":
    main()
