"""Build libpylda_hip.so (gfx950) in-tree with hipcc.

    python -m pylda_amd.build [--force]

The shared library is written to pylda_amd/lib/libpylda_hip.so.  It is
git-ignored (history stays source-only) but travels to the GPU box with the
working tree.  hipcc cross-compiles for gfx950 without a GPU present.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpylda_hip.so")
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cpp"))]


def _inputs():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(ROOT, "include", "pylda_hip.h"))
    deps.append(os.path.abspath(__file__))
    return deps


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > built for p in _inputs())


def build(force=False, verbose=True, extra_flags=(), jobs=None):
    """Compile every translation unit of csrc/ (in parallel) and link the shared library."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    common = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC",
              "-ffp-contract=off", "-Wall", "-Wno-unused-function",
              # same-address LDS atomics stay single ds_add instructions (the optimizer's
              # 64-bit wave scan is a 64-iteration scalar loop, 10x slower than the LDS unit)
              "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
              "-I" + os.path.join(ROOT, "include")] + list(extra_flags)
    srcs = sources()
    objs = [os.path.join(obj_dir, os.path.splitext(os.path.basename(s))[0] + ".o") for s in srcs]
    cmds = [common + ["-c", s, "-o", o] for s, o in zip(srcs, objs)]
    if verbose:
        print(" ".join(common + ["-c", "<%d sources of csrc/>" % len(srcs)]), flush=True)
    jobs = jobs or min(len(cmds), max(1, (os.cpu_count() or 2)))
    running, failed = [], []
    pending = list(cmds)
    while pending or running:
        while pending and len(running) < jobs:
            running.append((pending[0], subprocess.Popen(pending.pop(0))))
        cmd, proc = running.pop(0)
        if proc.wait() != 0:
            failed.append(cmd[-3])
    if failed:
        raise RuntimeError("hipcc failed on: " + ", ".join(os.path.basename(f) for f in failed))
    link = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH + ".tmp"] + objs
    if verbose:
        print(" ".join(link[:6] + ["<objects>"]), flush=True)
    subprocess.check_call(link)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
