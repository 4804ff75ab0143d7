#!/usr/bin/env python3
"""Phase timing of the quad kernel's inner loop with s_memtime stamps (development tool).

    python tools/phase_stamps_quad.py       # builds an instrumented COPY under .scratch/dbgq
    gpurun -- 'cd .scratch/dbgq && python run_dbg.py [lds_pad]'

Same method as tools/phase_stamps.py (the working tree is not touched; stamps drain the LDS queue, so
absolute times are ~10-15 % high, the split between phases is what counts).  Per-phase sums of
wavefront 0 (a gamma-phase wavefront) and wavefront 3 are written into the gamma output; run_dbg.py
prints cycles per inner iteration and phase for the N in [177, 192] class (quad<8,10,2>) of cfg 3,
with two workgroups per CU (default) or one (argument: LDS padding bytes, e.g. 40000)."""
import os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, ".scratch", "dbgq")
shutil.rmtree(dst, ignore_errors=True)
os.makedirs(dst)
for d in ("pylda_amd", "include"):
    shutil.copytree(os.path.join(root, d), os.path.join(dst, d), ignore=shutil.ignore_patterns("lib", "__pycache__"))
p = os.path.join(dst, "pylda_amd/csrc/estep_quad.h")
s = open(p).read()
def rep(a, b, count=1):
    global s
    assert a in s, a
    s = s.replace(a, b, count)
rep('''    long long moved = 0x7fffffffffffffffll;''','''    long long stamp_acc[10] = {0,0,0,0,0,0,0,0,0,0};
    long long stamp_prev = 0;
#define STAMP(j) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); long long now_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp_acc[j] += now_ - stamp_prev; stamp_prev = now_; } while (0)
    long long moved = 0x7fffffffffffffffll;''')
rep('''    for (;;) {                                                            // :174''','''    { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp_prev = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    for (;;) {                                                            // :174''')
rep('''#pragma unroll
        for (int i = 0; i < C0; ++i) myred[i * RS + c] = a[i];''','''#pragma unroll
        for (int i = 0; i < C0; ++i) myred[i * RS + c] = a[i];
        STAMP(0);   // t wait + pass A over the first eight slots (+ rows 0, 1) + writes''')
rep('''            wave_lds_exchange();                                          // the writes below stay behind the reads above''','''            STAMP(1);   // register slots 8.., row 2, first transpose landed and summed
            wave_lds_exchange();                                          // the writes below stay behind the reads above''')
rep('''        // B. q[k] over this lane's words (registers and LDS rows interleaved), then over the 4 word groups''','''        STAMP(2);       // second transpose, reciprocals
        // B. q[k] over this lane's words (registers and LDS rows interleaved), then over the 4 word groups''')
rep('''        __syncthreads();

        // C. gamma update by the topic threads''','''        STAMP(3);       // B FMAs + swaps + sp write
        __syncthreads();
        STAMP(4);       // barrier 1

        // C. gamma update by the topic threads''')
rep('''            keep_together(part);''','''            keep_together(part);
            STAMP(5);   // partial sums arrived''')
rep('''        ++it;
        --left;
        __syncthreads();''','''        ++it;
        --left;
        STAMP(6);       // gamma phase compute
        __syncthreads();
        STAMP(7);       // barrier 2''')
rep('''        p.gamma[(size_t)doc * K + tid] = gam;''','''        if (false) p.gamma[(size_t)doc * K + tid] = gam;''')
rep('''    term1 = wave_sum(term1);''','''    if (lane == 0 && (wave == 0 || wave == 3)) {
        const int base = wave == 0 ? 0 : 16;
        for (int j = 0; j < 10; ++j) p.gamma[(size_t)doc * K + base + j] = (double)stamp_acc[j];
        p.gamma[(size_t)doc * K + base + 10] = (double)it;
    }
    term1 = wave_sum(term1);''')
open(p, "w").write(s)
open(os.path.join(dst, "run_dbg.py"), "w").write('''
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pylda_amd import _capi
from pylda_amd.corpus import synthetic_lda_shard
pad = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nmin, nmax = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (177, 192)
ptr, ids, cts = synthetic_lda_shard(100000, 50000, 0, 50000, 128, 200, 1234, chunk=25000, device="cuda", workers=8)
n = np.diff(ptr)
sel = np.nonzero((n >= nmin) & (n <= nmax))[0]
newptr = np.concatenate([[0], np.cumsum(n[sel])]).astype(np.int64)
idx = np.concatenate([np.arange(ptr[d], ptr[d + 1]) for d in sel])
K, V = 128, 50000
np.random.seed(0)
eta = np.random.gamma(100., 0.01, (K, V))
ctx = _capi.Context(K, V)
ctx.set_option("lds_pad", pad)
ctx.set_option("doc_values", 0)
corpus = ctx.corpus(newptr, ids[idx], cts[idx])
ctx.set_alpha(np.full(K, 1.0 / K)); ctx.set_eta(eta)
ctx.estep(corpus); ctx.estep(corpus)
g = ctx.get_gamma(corpus)
print("documents", len(sel), "classes", [(c["kernel"], c["geometry"], c["documents"]) for c in corpus.plan()], "lds_pad", pad)
names = ["t wait + A slots 0-7 (+rows)", "A slots 8.. + transpose 1", "transpose 2, reciprocals", "B FMA+swaps+wr", "barrier1", "C: sp read", "C: compute", "barrier2"]
for base, w in ((0, "wave0 (topic wave)"), (16, "wave3")):
    its = g[:, base + 10]
    ok = its > 0
    print(w, "mean iterations", its[ok].mean())
    tot = 0
    for j, nm in enumerate(names):
        v = (g[ok, base + j] / its[ok]).mean()
        tot += v
        print("   %-26s %8.1f" % (nm, v))
    print("   %-26s %8.1f" % ("total/iter", tot))
''')
subprocess.check_call([sys.executable, "-m", "pylda_amd.build"], cwd=dst, stdout=subprocess.DEVNULL)
print("built", dst)
