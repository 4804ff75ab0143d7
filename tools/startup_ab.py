"""Start-up of `launch_train --gpus 2 --share_gpu=1` on a 200k-line synthetic TEXT corpus: wall time
of one training iteration end to end and the start-up phases (PYLDA_TIMING=1), for this tree and - if given - for a
copy of an older tree:   python tools/startup_ab.py [/path/to/old/tree]"""
import os, subprocess, sys, tempfile, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from pylda_amd.corpus import synthetic_lda_shard

D, V = 200000, 50000
work = tempfile.mkdtemp(prefix="pylda_startup_")
corpus = os.path.join(work, "synth200k")
os.makedirs(corpus)
t0 = time.perf_counter()
ptr, ids, cts = synthetic_lda_shard(D, V, 0, D, 128, 200, 1234, chunk=25000, device="cuda", workers=8)
words = np.array(["w%05d" % v for v in range(V)])
with open(os.path.join(corpus, "train.dat"), "w") as out:
    for d in range(D):
        lo, hi = ptr[d], ptr[d + 1]
        out.write(" ".join(np.repeat(words[ids[lo:hi]], cts[lo:hi])) + "\n")
with open(os.path.join(corpus, "voc.dat"), "w") as out:
    out.writelines("%s\t1\t1\n" % w for w in words)
print("corpus: %d lines, %.0f MB of text, written in %.1f s" % (D, os.path.getsize(os.path.join(corpus, "train.dat")) / 1e6, time.perf_counter() - t0))
for label, tree in [("this tree", root)] + ([("older tree", os.path.abspath(sys.argv[1]))] if len(sys.argv) > 1 else []):
    for gpus in (1, 2):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
        env.update(PYLDA_SEED="1", PYLDA_TIMING="1", PYTHONPATH=tree)
        cmd = [sys.executable, "-m", "pylda_amd.launch_train", "--input_directory=%s/" % corpus, "--output_directory=%s" % os.path.join(work, "out"),
               "--number_of_topics=64", "--training_iterations=1", "--snapshot_interval=1000"]
        if gpus > 1:
            cmd += ["--gpus=%d" % gpus, "--share_gpu=1"]
        t0 = time.perf_counter()
        done = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=tree)
        wall = time.perf_counter() - t0
        phases = [l for l in done.stderr.splitlines() if "parse + initial eta" in l]
        print("%-10s --gpus %d: %.1f s end to end (rc %d)%s" % (label, gpus, wall, done.returncode, "".join("\n    " + l for l in phases)))
        if done.returncode != 0:
            print(done.stderr[-1500:])
