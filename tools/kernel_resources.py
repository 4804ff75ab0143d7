#!/usr/bin/env python3
"""Registers / scratch per kernel from `hipcc -S --cuda-device-only` output: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -S --cuda-device-only -o launch_quad.s pylda_amd/csrc/launch_quad.hip; python tools/kernel_resources.py launch_quad.s [filter]"""
import re, subprocess, sys
lines = open(sys.argv[1]).read().splitlines()
flt = sys.argv[2] if len(sys.argv) > 2 else "estep"
name, info = None, {}
for l in lines:
    m = re.match(r'^(_Z\w+):', l)
    if m:
        name = m.group(1)
    for key in ('TotalNumSgprs', 'NumVgprs', 'ScratchSize', 'Occupancy'):
        m2 = re.match(r'^; %s: (\d+)' % key, l)
        if m2 and name:
            info.setdefault(name, {})[key] = int(m2.group(1))
for k, v in info.items():
    if flt in k:
        d = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
        print("%-64s sgpr %3d vgpr %3d scratch %4d occupancy %d" % (d[:64], v.get('TotalNumSgprs', -1), v.get('NumVgprs', -1), v.get('ScratchSize', -1), v.get('Occupancy', -1)))

# spill traffic inside loops: scratch_* / v_readlane / v_writelane in blocks the assembler comments mark as
# "in Loop" or "Loop Header", per kernel and loop depth
print()
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
for (k, depth, op), n in sorted(spills.items()):
    d = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
    print("IN LOOP depth %d: %-56s %-22s x %d" % (depth, d[:56], op, n))

# scratch traffic on the HOT path: basic blocks inside a loop that carry the FMA work (>= 16 fp64 multiply-adds) and
# touch scratch.  (A cold block inside the loop - the hand-over exit of estep_quad.h - may spill; the iterations may not.)
def hot_scratch(lines, flt="estep"):
    out, name, cur = {}, None, None
    def close():
        if name and cur and cur["depth"] > 0 and cur["fma"] >= 16 and cur["scratch"] and flt in name:
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

print()
for k, blocks in hot_scratch(lines, flt).items():
    d = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
    for label, fma, n in blocks:
        print("HOT BLOCK %-56s %-12s fma %3d scratch x %d" % (d[:56], label, fma, n))
