#!/usr/bin/env python3
"""Build a COPY of the package under .scratch/<tag> with extra compiler flags (A/B experiments on the
GPU box without touching the working tree's library):

    python tools/ab_build.py prio -DPYLDA_QUAD_CPRIO=3
    gpurun -- 'cd .ab/prio && python tools/class_ab.py cfg3 193 208'
"""
import os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, flags = sys.argv[1], sys.argv[2:]
dst = os.path.join(root, ".ab", tag)
shutil.rmtree(dst, ignore_errors=True)
os.makedirs(dst)
for d in ("pylda_amd", "include", "tools"):
    shutil.copytree(os.path.join(root, d), os.path.join(dst, d),
                    ignore=shutil.ignore_patterns("lib", "__pycache__", "valu_bench*", "atomic_bench"))
subprocess.check_call([sys.executable, "-c",
                       "import sys; sys.path.insert(0, %r); from pylda_amd import build; "
                       "build.build(force=True, verbose=False, extra_flags=%r)" % (dst, flags)])
print("built", dst, flags)
