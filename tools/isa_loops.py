"""Instruction census of the loops of a kernel in a hipcc -S dump.

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only x.hip -o x.s
    python tools/isa_loops.py x.s [kernel-name-substring] [min-instructions]

Prints, per kernel and per loop (a backward branch to a label), the instruction count and the classes that matter on
the fp64 chain: FMA-class, cross-lane moves, scalar loads, scratch traffic, waits.
"""
import re
import sys


def kernels(lines):
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):\s', l + ' ')
        if m and (i + 1 < len(lines)):
            end = next((j for j in range(i, len(lines)) if lines[j].startswith('.Lfunc_end')), len(lines))
            yield m.group(1), lines[i:end]


def loops(body):
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    for i, l in enumerate(body):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            yield labels[m.group(1)], i


def census(seg):
    ins = [l.strip() for l in seg if l.strip() and not l.strip().startswith(('.', ';'))]

    def cnt(pat):
        return sum(1 for l in ins if re.match(pat, l))
    return {"n": len(ins), "f64": cnt(r'v_(fma|mul|add|fmac|min|max|rcp|rndne|ldexp|cmp\w*)_f64'),
            "readlane": cnt(r'v_readlane'), "writelane": cnt(r'v_writelane'), "permlane": cnt(r'v_permlane'),
            "dpp": sum(1 for l in ins if 'dpp' in l or 'row_' in l or 'quad_perm' in l), "cndmask": cnt(r'v_cndmask'),
            "v_mov": cnt(r'v_mov'), "accvgpr": cnt(r'v_accvgpr'), "s_load": cnt(r's_load'), "scratch": cnt(r'scratch_'),
            "waitcnt": cnt(r's_waitcnt'), "ds": cnt(r'ds_'), "global": cnt(r'global_'), "s_nop": cnt(r's_nop'),
            "barrier": cnt(r's_barrier')}


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    least = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    lines = open(path).read().split('\n')
    for name, body in kernels(lines):
        if want not in name:
            continue
        print(name, "(%d lines)" % len(body))
        for a, b in sorted(set(loops(body))):
            c = census(body[a:b + 1])
            if c["n"] >= least:
                print("  loop @%d-%d: %s" % (a, b, " ".join("%s=%d" % kv for kv in c.items() if kv[1])))


if __name__ == "__main__":
    main()
