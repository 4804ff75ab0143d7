#!/usr/bin/env python3
"""Instruction mix of a kernel's basic blocks from hipcc -S output.

    python tools/isa_loop_stats.py launch_quad.s <mangled-kernel-name-substring> [--blocks]      (hipcc -S --cuda-device-only of the translation unit that instantiates the kernel)

Prints, per basic block of the kernel (label .LBBn_m), the instruction count by class
(fp64 FMA/MUL/ADD, other VALU, DPP moves, permlane, DS, VMEM, SALU, waitcnt/barrier), so the
inner-loop issue budget can be read off without a GPU."""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64")):
        return "f64"
    if op.startswith(("v_rcp_f64", "v_ldexp_f64", "v_rndne_f64", "v_min_f64", "v_max_f64", "v_cvt", "v_frexp", "v_log", "v_exp",
                      "v_cmp_", "v_cmpx", "v_trig", "v_fract", "v_div", "v_sqrt", "v_rsq")) and "f64" in op:
        return "f64_other"
    if "permlane" in op:
        return "permlane"
    if "dpp" in op:
        return "dpp"
    if op.startswith("v_"):
        return "valu32"
    if op.startswith("ds_"):
        return "ds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_sleep")):
        return op.split()[0]
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.rstrip().endswith(("E:", ":")) and "@" in l or
                 (l.startswith("_Z") and name in l and ":" in l))
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = Counter()
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end") or s.startswith("s_endpgm") and False:
            break
        if s.startswith(".LBB") and s.endswith(":"):
            cur = s[:-1]
            blocks[cur] = Counter()
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        if "dpp" in s and not "dpp" in op:
            op = op + "_dpp"
        blocks[cur][classify(op) if "dpp" not in s or op.startswith("v_fmac_f64") else ("f64" if op.startswith("v_fmac_f64") else "dpp")] += 1
    tot = Counter()
    for b, c in blocks.items():
        n = sum(c.values())
        if n >= 20:
            print("%-12s %4d  %s" % (b, n, dict(sorted(c.items()))))
        tot.update(c)
    print("total", sum(tot.values()), dict(sorted(tot.items())))


if __name__ == "__main__":
    main()
