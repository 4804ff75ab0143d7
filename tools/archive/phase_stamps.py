#!/usr/bin/env python3
"""Phase timing of the quilt kernel's inner loop with s_memtime stamps (development tool).

    python tools/phase_stamps.py            # builds an instrumented COPY under .ab/dbg
    gpurun -- 'cd .ab/dbg && python run_dbg.py'

The working tree is not touched: pylda_amd/ is copied, estep_quilt.h of the copy gets a stamp
(s_waitcnt lgkmcnt(0); s_memtime) at each phase boundary, the per-phase sums of wavefront 0 (a
gamma-phase wavefront) and wavefront 5 are written into the gamma output instead of gamma, and
run_dbg.py prints cycles per inner iteration and phase on a 25k-document cfg-3 corpus.  The stamps
drain the LDS queue, so the instrumented iteration is ~15 % slower than the real one; the split
between phases is what DESIGN.md quotes.  The anchors below are source lines of the kernel: when
the kernel changes, the assertions say which anchor to update.
"""
import os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, ".ab", "dbg")
shutil.rmtree(dst, ignore_errors=True)
os.makedirs(dst)
for d in ("pylda_amd", "include"):
    shutil.copytree(os.path.join(root, d), os.path.join(dst, d), ignore=shutil.ignore_patterns("lib", "__pycache__"))
shutil.copy(os.path.join(root, "bench.py"), dst)
p = os.path.join(dst, "pylda_amd/csrc/estep_quilt.h")
s = open(p).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b, 1)
rep('''    long long moved = 0x7fffffffffffffffll;''','''    long long stamp_acc[10] = {0,0,0,0,0,0,0,0,0,0};
    long long stamp_prev = 0;
#define STAMP(j) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); long long now_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp_acc[j] += now_ - stamp_prev; stamp_prev = now_; } while (0)
    long long moved = 0x7fffffffffffffffll;''')
rep('''    for (;;) {                                                            // :174''','''    { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp_prev = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    for (;;) {                                                            // :174''')
rep('''        if (moved <= thresh || left <= 0) break;''','''        STAMP(0);   // tq wait + A fma + writes
        if (moved <= thresh || left <= 0) break;''')
rep('''            const double s = lane_group_sum<LPW>(s0 + s1);''','''            STAMP(1);   // partial reads arrived
            const double s = lane_group_sum<LPW>(s0 + s1);''')
rep('''        // B. q[k] over this lane's words''','''        STAMP(2);   // r done
        // B. q[k] over this lane's words''')
rep('''        __syncthreads();

        // C. gamma update by the topic threads''','''        STAMP(3);   // B fma + swaps + sp write
        __syncthreads();
        STAMP(4);   // barrier 1

        // C. gamma update by the topic threads''')
rep('''            keep_together(part);''','''            keep_together(part);
            STAMP(5);   // partial sums arrived''')
rep('''        ++it;
        --left;
        __syncthreads();''','''        ++it;
        --left;
        STAMP(6);   // C compute
        __syncthreads();
        STAMP(7);   // barrier 2''')
rep('''        p.gamma[(size_t)doc * K + tid] = gam;''','''        if (false) p.gamma[(size_t)doc * K + tid] = gam;''')
rep('''    term1 = wave_sum(term1);''','''    if (lane == 0 && (wave == 0 || wave == 5)) {
        const int base = wave == 0 ? 0 : 16;
        for (int j = 0; j < 10; ++j) p.gamma[(size_t)doc * K + base + j] = (double)stamp_acc[j];
        p.gamma[(size_t)doc * K + base + 10] = (double)it;
    }
    term1 = wave_sum(term1);''')
open(p, "w").write(s)
open(os.path.join(dst, "run_dbg.py"), "w").write('''
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench
from pylda_amd.variational_bayes import VariationalBayes
dev = torch.device("cuda", 0)
wl = bench.build_workload("synth100k", 0, 1, dev, 25000)
ptr, ids, cts, V, K = wl["ptr"], wl["ids"], wl["cts"], wl["V"], wl["K"]
np.random.seed(0)
eta0 = np.random.gamma(100., 1. / 100., (K, V))
vb = VariationalBayes(hyper_parameter_optimize_interval=1, device=0)
vb._verbose = False
vb._initialize_parsed(ptr, ids, cts, V, K, 1.0 / K, 1.0 / V, eta=eta0)
import warnings; warnings.simplefilter("ignore")
for it in range(2):
    vb.learning()
g = np.asarray(vb._gamma)
names = ["tq wait+A fma+wr", "partials read", "rcp/r", "B fma+swaps+wr", "barrier1", "C: sp read", "C: compute", "barrier2"]
for base, w in ((0, "wave0 (topic wave)"), (16, "wave5")):
    its = g[:, base + 10]
    print(w, "mean iterations", its.mean())
    tot = 0
    for j, n in enumerate(names):
        v = (g[:, base + j] / its).mean()
        tot += v
        print("   %-18s %8.1f" % (n, v))
    print("   %-18s %8.1f" % ("total/iter", tot))
''')
subprocess.check_call([sys.executable, "-m", "pylda_amd.build"], cwd=dst, stdout=subprocess.DEVNULL)
print("built", dst)
