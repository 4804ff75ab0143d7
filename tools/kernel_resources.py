#!/usr/bin/env python3
"""Registers / scratch per kernel from `hipcc -S --cuda-device-only` output, and the scratch traffic on the hot path.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -S --cuda-device-only -o launch_quad.s pylda_amd/csrc/launch_quad.hip
    python tools/kernel_resources.py launch_quad.s [filter]
    python tools/kernel_resources.py --compile pylda_amd/csrc/launch_quad.hip [filter]

tests/test_kernel_resources.py holds the document kernels to what this prints (HOT BLOCK lines)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function",
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-I" + os.path.join(ROOT, "include")]      # pylda_amd/build.py


def compile_to_asm(source, out=None):
    """Device assembly of one translation unit with the library's own flags; returns the path."""
    from pylda_amd.build import _hipcc
    out = out or os.path.join(tempfile.mkdtemp(prefix="pylda_isa_"), os.path.basename(source) + ".s")
    subprocess.check_call([_hipcc()] + FLAGS + ["-S", "--cuda-device-only", source, "-o", out], stderr=subprocess.DEVNULL)
    return out


def demangle(names):
    out = subprocess.run(["c++filt"] + list(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def resources(lines, flt="estep"):
    """{kernel: {TotalNumSgprs, NumVgprs, ScratchSize, Occupancy}}"""
    name, info = None, {}
    for l in lines:
        m = re.match(r'^(_Z\w+):', l)
        if m:
            name = m.group(1)
        for key in ('TotalNumSgprs', 'NumVgprs', 'ScratchSize', 'Occupancy'):
            m2 = re.match(r'^; %s: (\d+)' % key, l)
            if m2 and name:
                info.setdefault(name, {})[key] = int(m2.group(1))
    return {k: v for k, v in info.items() if flt in k}


def in_loop_spills(lines, flt="estep"):
    """{(kernel, loop depth, opcode): count} of scratch_* / v_readlane / v_writelane in blocks marked as inside a loop
    (static: a cold block inside the loop - the hand-over exit of estep_quad.h - counts too)."""
    name, depth, spills = None, 0, {}
    for l in lines:
        m = re.match(r'^(_Z\w+):', l)
        if m:
            name, depth = m.group(1), 0
            continue
        s = l.strip()
        if s.startswith('.LBB') or s.startswith('; %bb.'):
            m = re.search(r'(?:in Loop: Header=\S+|Loop Header:) Depth=(\d+)', l)
            depth = int(m.group(1)) if m else 0
            continue
        if name and depth > 0 and flt in name and re.match(r'(scratch_|v_readlane|v_writelane)', s):
            key = (name, depth, s.split()[0])
            spills[key] = spills.get(key, 0) + 1
    return spills


def hot_scratch(lines, flt="estep", min_fma=16):
    """Scratch traffic on the HOT path: {kernel: [(block, fp64 multiply-adds, scratch instructions)]} for the basic
    blocks inside a loop that carry the FMA work and touch scratch.  (A cold block inside the loop may spill; the
    iterations may not.)"""
    out, name, cur = {}, None, None

    def close():
        if name and cur and cur["depth"] > 0 and cur["fma"] >= min_fma and cur["scratch"] and flt in name:
            out.setdefault(name, []).append((cur["label"], cur["fma"], cur["scratch"]))
    for l in lines:
        m = re.match(r'^(_Z\w+):', l)
        if m:
            close()
            name, cur = m.group(1), None
            continue
        s = l.strip()
        if s.startswith('.LBB') or s.startswith('; %bb.'):
            close()
            m = re.search(r'(?:in Loop: Header=\S+|Loop Header:) Depth=(\d+)', l)
            cur = {"label": s.split()[0], "depth": int(m.group(1)) if m else 0, "fma": 0, "scratch": 0}
        elif cur is not None:
            if re.match(r'v_(fma|fmac|mul)_f64', s):
                cur["fma"] += 1
            elif s.startswith('scratch_'):
                cur["scratch"] += 1
    close()
    return out


def main(argv):
    if argv and argv[0] == "--compile":
        sys.path.insert(0, ROOT)
        path = compile_to_asm(argv[1])
        argv = [path] + argv[2:]
    lines = open(argv[0]).read().splitlines()
    flt = argv[1] if len(argv) > 1 else "estep"
    res = resources(lines, flt)
    names = demangle(list(res))
    for k, v in res.items():
        print("%-64s sgpr %3d vgpr %3d scratch %4d occupancy %d" % (names[k][:64], v.get('TotalNumSgprs', -1), v.get('NumVgprs', -1),
                                                                    v.get('ScratchSize', -1), v.get('Occupancy', -1)))
    print()
    for (k, depth, op), n in sorted(in_loop_spills(lines, flt).items()):
        print("IN LOOP depth %d: %-56s %-22s x %d" % (depth, names.get(k, k)[:56], op, n))
    print()
    for k, blocks in hot_scratch(lines, flt).items():
        for label, fma, n in blocks:
            print("HOT BLOCK %-56s %-12s fma %3d scratch x %d" % (names.get(k, k)[:56], label, fma, n))


if __name__ == "__main__":
    main(sys.argv[1:])
