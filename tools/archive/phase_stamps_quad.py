#!/usr/bin/env python3
"""Phase timing of the quad kernel's inner loop with s_memtime stamps (development tool).

    python tools/phase_stamps_quad.py [-DNAME=value ...]      # builds an instrumented COPY under .ab/dbgq
    gpurun -- 'cd .ab/dbgq && python run_dbg.py [cfg3|cfg4] [nmin nmax] [name=value ...]'

The copy is compiled with -DPYLDA_QUAD_STAMPS=1 (estep_quad.h: QUAD_STAMP); the working tree's library
is not touched.  Stamps drain the LDS queue, so absolute times are ~10-15 % high, the split between the
phases is what counts.  Every wavefront writes its per-phase sums, its HW_ID and LDS_ALLOC registers over
the document's gamma row; run_dbg.py prints cycles per inner iteration and phase, prologue / epilogue
cycles per document, and where the hardware placed the wavefronts (SIMD of wavefront i, LDS base of the
workgroup)."""
import os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, ".ab", "dbgq")
shutil.rmtree(dst, ignore_errors=True)
os.makedirs(dst)
for d in ("pylda_amd", "include"):
    shutil.copytree(os.path.join(root, d), os.path.join(dst, d), ignore=shutil.ignore_patterns("lib", "__pycache__"))
os.makedirs(os.path.join(dst, "tools"))
shutil.copy(os.path.join(root, "tools", "quad_stamps.h"), os.path.join(dst, "tools", "quad_stamps.h"))   # estep_quad.h includes it under -DPYLDA_QUAD_STAMPS=1
open(os.path.join(dst, "run_dbg.py"), "w").write('''
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pylda_amd import _capi
from pylda_amd.corpus import synthetic_lda_shard
args = sys.argv[1:]
cfg = args.pop(0) if args and args[0] in ("cfg3", "cfg4") else "cfg3"
nmin, nmax = (int(args.pop(0)), int(args.pop(0))) if len(args) >= 2 and args[0].isdigit() else (177, 192)
opts = [kv.split("=") for kv in args]
D, V, K, seed = (100000, 50000, 128, 1234) if cfg == "cfg3" else (100000, 100000, 256, 5678)
ptr, ids, cts = synthetic_lda_shard(D if cfg == "cfg3" else 1000000, V, 0, D, 128, 200, seed, chunk=25000, device="cuda", workers=8)
n = np.diff(ptr)
sel = np.nonzero((n >= nmin) & (n <= nmax))[0]
newptr = np.concatenate([[0], np.cumsum(n[sel])]).astype(np.int64)
idx = np.concatenate([np.arange(ptr[d], ptr[d + 1]) for d in sel])
np.random.seed(0)
eta = np.random.gamma(100., 0.01, (K, V))
ctx = _capi.Context(K, V)
for name, value in opts:
    ctx.set_option(name, int(value))
ctx.set_option("doc_values", 0)
corpus = ctx.corpus(newptr, ids[idx], cts[idx])
ctx.set_alpha(np.full(K, 1.0 / K)); ctx.set_eta(eta)
ctx.estep(corpus); ctx.estep(corpus)
g = ctx.get_gamma(corpus)
W = 4 if K <= 128 else 8
print(cfg, "documents", len(sel), "classes", [(c["kernel"], c["geometry"], c["documents"]) for c in corpus.plan()], "opts", opts)
names = ["t wait + A slots 0-7 (+rows)", "A slots 8.. + transpose 1", "transpose 2, reciprocals", "B FMA + swaps + write", "barrier 1",
         "C: partials read", "C: t stored, LDS drained", "barrier 2"]
sub = [(14, "C: gamma update + atomic done"), (15, "C: exp(psi(gamma) - c)")]
for w in (0, W - 1):
    base = 16 * w
    its = g[:, base + 11]
    ok = its > 0
    print("wavefront %d: mean iterations %.2f" % (w, its[ok].mean()))
    tot = 0
    for j, nm in enumerate(names):
        if j == 6:
            for k, nm2 in sub:
                v = (g[ok, base + k] / its[ok]).mean()
                tot += v
                print("   %-30s %8.1f" % (nm2, v))
        v = (g[ok, base + j] / its[ok]).mean()
        tot += v
        print("   %-30s %8.1f" % (nm, v))
    print("   %-30s %8.1f" % ("total / iteration", tot))
    print("   %-30s %8.1f per document" % ("prologue (gather, first t)", g[ok, base + 8].mean()))
    print("   %-30s %8.1f per document" % ("last half iteration", g[ok, base + 9].mean()))
    print("   %-30s %8.1f per document" % ("epilogue to the reductions", g[ok, base + 10].mean()))
    print("   %-30s %8.1f per document" % ("inner loop", sum(g[ok, base + j] for j in range(8)).mean()))
hw = np.stack([g[:, 16 * w + 12] for w in range(W)], 1).astype(np.int64)
lds = g[:, 13].astype(np.int64)
simd = (hw >> 4) & 3
print("SIMD of wavefront 0..%d: most common patterns" % (W - 1))
pat, cnt = np.unique(simd, axis=0, return_counts=True)
for i in np.argsort(-cnt)[:8]:
    print("   ", pat[i].tolist(), cnt[i])
print("wave slot (HW_ID[3:0]) patterns")
pat, cnt = np.unique(hw & 15, axis=0, return_counts=True)
for i in np.argsort(-cnt)[:8]:
    print("   ", pat[i].tolist(), cnt[i])
base, cnt = np.unique(lds & 0x1ff, return_counts=True)
print("LDS_ALLOC base field values:", dict(zip(base.tolist(), cnt.tolist())))
size, cnt = np.unique((lds >> 12) & 0x1ff, return_counts=True)
print("LDS_ALLOC size field values:", dict(zip(size.tolist(), cnt.tolist())))
''')
sys.path.insert(0, dst)
subprocess.check_call([sys.executable, "-c",
                       "import sys; sys.path.insert(0, %r); from pylda_amd import build; build.build(force=True, verbose=False, extra_flags=['-DPYLDA_QUAD_STAMPS=1'] + %r)" % (dst, sys.argv[1:])])
print("built", dst)
