#!/usr/bin/env python3
"""bench.py - documents/sec per VB iteration of the MI355X E-step path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload synth100k|synth1m|ap|nips]

A "step" is one outer VB iteration over the rank's resident corpus: the hot
path (device compute_dirichlet_expectation + per-document phi/gamma kernels +
sufficient-statistics accumulation, variational_bayes.py:132-216), the RCCL
all-reduce of the K*V sufficient statistics when N > 1, and the device M-step /
alpha update that make the next iteration start from a new model (nothing is
cached between steps).  Inputs are resident in HBM before the timed region.
`python bench.py --gpus N` (N > 1) launches its own ranks through
torch.distributed.run; under an existing launcher (RANK/WORLD_SIZE set) it is a rank.

Workloads (BASELINE.json configs):
  synth1m    cfg 4: 1,000,000 docs TOTAL, V=100k, K=256, sharded over N GPUs -
             strong scaling (default primary line)
  synth100k  cfg 3: synthetic LDA corpus, 100,000 docs PER GPU, V=50k, K=128,
             mean 200 tokens/doc - weak scaling
  ap         cfg 2: associated-press train split (committed parsed fixture),
             K=10, replicated per GPU (latency-bound, 2000 documents)
  nips       cfg 5: parsed/nips.88-05 (committed parsed fixture), K=500,
             train = first 2,235 documents

The default run prints ONE JSON line on rank 0: the primary record - cfg 4, the
1M-document corpus north_star quotes 1/2/4/8-GPU throughput on, sharded over the
N ranks (strong scaling), the same workload at every N - with `roofline` and
`cpu_baseline`, and as sub-records the other configurations timed in the same
processes at every N: `synth100k` (cfg 3, the top-level record of rounds 1-4),
`ap_k10` (cfg 2) and `nips_k500` (cfg 5: the 50-iteration joint log-likelihood
trace asserted against the reference's own, held-out per-token log-likelihood
after 50 iterations).  At N = 1 `shard_proxy` / `nips_k500.shard_proxy` time rank
0's shard for N = 2 / 4 / 8 on the one GPU (a model of the curve, labelled so).

Protocol: 3 warm-up outer iterations from the seeded start, then
outer iterations 4-8 are THE timed window, whatever --steps / --warmup say: timed
step i is iteration 4 + (i mod 5); after iteration 8 the model returns to its
state after iteration 3 (a device checkpoint restored inside the timed region).
`--steps 5 --warmup 3` and `--steps 20 --warmup 5` therefore time the same work.

roofline: HBM bound, algorithmic bytes B = nnz*(8+16K) + D*(8K+8) per E-step
 over the kernels that move them - the document kernels of the E-step
(one launch class per words-per-lane instantiation, run concurrently) PLUS the
sufficient-statistics pass (gather + finalize) - timed with HIP events on the
launch streams inside the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X fp64 vector peak (spec), at ...
PEAK_CLOCK_MHZ = 2400.0         # ... the peak engine clock (MI355X_MICROARCH.md)
CHUNK = 25000
PROTOCOL_WARMUP = 3             # three warm-up outer iterations ...
PROTOCOL_WINDOW = 5             # ... then outer iterations 4-8 are the timed window

# (documents, nnz, sum of term ids, sum of counts) of the generated corpora: numpy PCG64 draws
# (pylda_amd/corpus.py::synthetic_lda_shard), independent of torch version and device.
EXPECTED_CHECKSUM = {
    "synth100k": [100000, 19651258, 493018266666, 19997266],       # one GPU's 100k documents (N = 1)
    "synth1m": [1000000, 198210795, 9901022603303, 199984652],
}


def algorithmic_bytes(nnz, D, K):
    return nnz * (8 + 16 * K) + D * (8 * K + 8)


def kernel_source_hash(csrc=None):
    """sha256 over the DEVICE sources whose HBM traffic profiles/traffic_*.json describes: the kernel headers of the
    E-step and of the statistics pass.  Host translation units (*.hip, *.cpp) are not part of it: an edit to the
    launcher or to a test hook does not change what the kernels move (tests/test_host_logic.py pins that)."""
    import hashlib
    h = hashlib.sha256()
    csrc = csrc or os.path.join(ROOT, "pylda_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith(".h") and (f.startswith("estep_") or f.startswith("sstats_")
                                 or f in ("doc_terms.h", "special_device.h", "prepare_kernels.h")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def traffic_record(name):
    """roofline.traffic: HBM bytes per E-step from the committed rocprofv3 --pmc passes of this workload
    (tools/profile_bench.sh writes profiles/traffic_<workload>.json with the hash of the kernel sources it
    profiled).  A file made with other kernel sources is reported as stale, not as a measurement."""
    if name is None:
        return None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % name)
    if not os.path.exists(tpath):
        return None, None
    try:
        rec = json.load(open(tpath))
    except Exception:
        return None, None
    have, want = rec.get("kernel_source_hash"), kernel_source_hash()
    if have != want:
        return None, "profiles/traffic_%s.json is STALE (kernel sources %s, profiled %s): %s bytes per E-step then" \
            % (name, want, have, rec.get("hbm_bytes_per_launch"))
    return rec.get("hbm_bytes_per_launch"), \
        "profiles/traffic_%s.json (rocprofv3 --pmc passes of this workload and these kernel sources [%s], corrected " \
        "per MI355X_MICROARCH.md; not re-measured in this run)" % (name, want)


def build_workload(name, rank, world, device, docs_override=None):
    from pylda_amd.corpus import synthetic_lda_shard
    workers = max(1, min(8, (os.cpu_count() or 1) // max(1, world)))
    if name == "synth100k":
        per_gpu = docs_override or 100000
        V, K, seed = 50000, 128, 1234
        total = per_gpu * world
        ptr, ids, cts = synthetic_lda_shard(total, V, rank * per_gpu, (rank + 1) * per_gpu, 128, 200, seed,
                                            chunk=min(CHUNK, per_gpu), device=device, workers=workers)
        return dict(ptr=ptr, ids=ids, cts=cts, V=V, K=K, scaling="weak", cfg="cfg3",
                    label="synthetic LDA corpus cfg3: %d docs/GPU, V=50000, K=128, mean 200 tokens/doc" % per_gpu)
    if name == "synth1m":
        total = docs_override or 1000000
        V, K, seed = 100000, 256, 5678
        chunk = min(CHUNK, max(1, total // max(1, world)))
        n_chunks = (total + chunk - 1) // chunk
        # contiguous runs of whole chunks per rank: every chunk holds `chunk` documents of the same
        # distribution, so the ranks' nnz differ by well under 1 % (reported as nnz_imbalance)
        first = (n_chunks * rank) // world
        last = (n_chunks * (rank + 1)) // world
        ptr, ids, cts = synthetic_lda_shard(total, V, first * chunk, min(total, last * chunk), 128, 200, seed,
                                            chunk=chunk, device=device, workers=workers)
        return dict(ptr=ptr, ids=ids, cts=cts, V=V, K=K, scaling="strong", cfg="cfg4",
                    label="synthetic LDA corpus cfg4: %d docs total, V=100000, K=256, sharded by document" % total)
    if name == "ap":
        g = np.load(os.path.join(ROOT, "tests", "golden", "ap_train_k10.npz"))
        return dict(ptr=g["doc_ptr"].astype(np.int64), ids=g["term_id"].astype(np.int32),
                    cts=g["term_ct"].astype(np.int32), V=int(g["eta"].shape[1]), K=10,
                    scaling="weak", eta=g["eta"], alpha=g["alpha"], cfg="cfg2",
                    label="associated-press train split (2000 docs, V=6806), K=10, replicated per GPU")
    if name == "nips":
        g = np.load(os.path.join(ROOT, "tests", "golden", "nips_trace_k500.npz"))
        return dict(ptr=g["doc_ptr"].astype(np.int64), ids=g["term_id"].astype(np.int32),
                    cts=g["term_ct"].astype(np.int32), V=len(g["words"]), K=int(g["K"]), scaling="weak", cfg="cfg5",
                    label="parsed/nips.88-05 train split (2235 docs, V=3209), K=500, replicated per GPU")
    raise SystemExit("unknown workload %r" % name)


def cpu_baseline(alpha, eta, ptr, ids, cts, budget_s, max_docs, repeats=3):
    """The reference's algorithm on the host CPU: numpy restatement (what the reference itself executes), single
    thread, on a FIXED prefix of the corpus timed `repeats` times - the figure is the MEDIAN (a time-boxed prefix timed
    once swung 1.8x between runs on the GPU box's shared host).  The prefix is sized by one probe of 20 documents so
    that the leg stays within ~budget_s of CPU time: documents = budget / repeats x the probe's rate, at most max_docs."""
    from oracle import vb_numpy
    E_log_eta = vb_numpy.compute_dirichlet_expectation(eta)
    D = len(ptr) - 1

    def run(n):
        out = []
        t0 = time.perf_counter()
        for d in range(n):
            lo, hi = int(ptr[d]), int(ptr[d + 1])
            _, ll, _, _, _ = vb_numpy.e_step_document(alpha, E_log_eta, ids[lo:hi].astype(np.int64), cts[lo:hi])
            out.append(ll)
        return time.perf_counter() - t0, np.array(out)

    probe = min(20, D)
    t_probe, _ = run(probe)
    n = int(max(probe, min(max_docs, D, budget_s / repeats / (t_probe / probe))))
    n = min(n, 200) if n >= 200 else n          # (200 documents when the budget allows: the same sample from run to run)
    times, doc_ll = [], None
    for _ in range(repeats):
        t, doc_ll = run(n)
        times.append(t)
    return n / float(np.median(times)), n, doc_ll


def cpu_baseline_all_cores(alpha, eta, ptr, ids, cts, budget_s, workers, docs_per_worker=600):
    """The same restatement in `workers` single-threaded processes side by side (SURVEY 8d's optional all-cores
    figure): every process (oracle/cpu_pool_worker.py, its own interpreter - this one holds a HIP context) runs
    its own slice of the corpus for `budget_s` seconds; the figure is the sum of the per-process rates.
    Stragglers are killed: the leg can delay the bench by budget_s + 90 s at most."""
    import subprocess
    import tempfile
    D = len(ptr) - 1
    workers = max(1, min(workers, D // 20))
    docs_per_worker = max(20, min(docs_per_worker, D // workers))
    last = workers * docs_per_worker
    here = os.path.dirname(os.path.abspath(__file__))
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "problem.npz")
        np.savez(path, alpha=alpha, eta=eta, ptr=ptr[:last + 1], ids=ids[:ptr[last]], cts=cts[:ptr[last]])
        procs = [subprocess.Popen([sys.executable, os.path.join(here, "oracle", "cpu_pool_worker.py"), path,
                                   str(w * docs_per_worker), str((w + 1) * docs_per_worker), "%g" % budget_s],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                 for w in range(workers)]
        deadline = time.perf_counter() + budget_s + 90.0
        rate, done, ok = 0.0, 0, 0
        for pr in procs:
            try:
                out, _ = pr.communicate(timeout=max(1.0, deadline - time.perf_counter()))
                n, secs = out.split()
                rate += int(n) / float(secs)
                done += int(n)
                ok += 1
            except Exception:
                pr.kill()
    if ok == 0:
        raise RuntimeError("no worker finished")
    return rate, done, ok


def c_oracle_rate(alpha, eta, ptr, ids, cts, n, repeats=3):
    from oracle import c_oracle
    c_oracle.load()
    times = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        c_oracle.e_step(alpha, eta, ptr[:n + 1], ids[:ptr[n]], cts[:ptr[n]])
        times.append(time.perf_counter() - t0)
    return n / float(np.median(times))


class Job(object):
    """Process-wide state of one bench run: rank, device, process group."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, self.world))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no HIP device is visible (there is no CPU fallback)")
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device("cuda", self.local_rank)
        self.group = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if args.share_gpu:      # test mode: every rank on GPU 0 (RCCL refuses that), exchange over gloo
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.device)
            self.group = dist.group.WORLD

    def barrier(self):
        if self.group is not None:
            import torch.distributed as dist
            dist.barrier()
        self.torch.cuda.synchronize()

    def reduce(self, values, op="sum"):
        """all-reduce a short list of floats over the ranks (identity at N = 1)."""
        if self.group is None:
            return list(values)
        import torch.distributed as dist
        backend_dev = self.device if dist.get_backend(self.group) == "nccl" else "cpu"
        t = self.torch.tensor(values, dtype=self.torch.float64, device=backend_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
        return [float(x) for x in t.cpu()]


def measure(job, args, name, steps, warmup, docs=None):
    """Build the workload, run `warmup` + `steps` learning() iterations, return the record pieces."""
    from pylda_amd import distributed
    from pylda_amd.corpus import corpus_checksum
    from pylda_amd.variational_bayes import VariationalBayes
    t_gen = time.perf_counter()
    wl = build_workload(name, job.rank, job.world, job.device, docs)
    t_gen = time.perf_counter() - t_gen
    ptr, ids, cts, V, K = wl["ptr"], wl["ids"], wl["cts"], wl["V"], wl["K"]
    D_local, nnz_local, tokens_local = len(ptr) - 1, int(ptr[-1]), int(cts.sum())

    np.random.seed(0)
    eta0 = wl.get("eta")
    if eta0 is None:
        eta0 = np.random.gamma(100., 1. / 100., (K, V))          # variational_bayes.py:95
    vb = VariationalBayes(hyper_parameter_optimize_interval=1, device=job.local_rank, process_group=job.group)
    vb._verbose = False
    t_init = time.perf_counter()
    vb._initialize_parsed(ptr, ids, cts, V, K, 1.0 / K, 1.0 / V, eta=eta0)      # upload + launch schedule
    t_init = time.perf_counter() - t_init
    if "alpha" in wl:
        vb._alpha_alpha = wl["alpha"].copy()
    ctx = vb._context()
    if args.variant >= 0:
        ctx.set_option("force_variant", args.variant)
    for kv in args.option:
        oname, value = kv.split("=")
        ctx.set_option(oname, int(value))
    if job.group is None:
        distributed.bind_to_torch_stream(ctx)

    # ---- SURVEY 8d protocol: 3 warm-up outer iterations from the seeded start, then outer iterations 4-8 timed.
    # The window is the same whatever --steps / --warmup say: a timed step is iteration 4 + (i mod 5), and after
    # iteration 8 the model goes back to its state after iteration 3 (device checkpoint of eta, pylda_model_checkpoint;
    # alpha and the counter on the host) INSIDE the timed region - every step runs the whole E-step, exchange and
    # M-step, nothing is cached.  --warmup beyond 3 runs further window steps untimed.
    t_first = time.perf_counter()
    t_step1 = None
    steady = name.startswith("synth")
    for _ in range(PROTOCOL_WARMUP):
        vb.learning()
        if t_step1 is None:
            t_step1 = time.perf_counter() - t_first          # includes the one-off postings (CSC) build of the first E-step
    job.torch.cuda.synchronize()
    t_first = time.perf_counter() - t_first
    ctx.model_checkpoint()
    alpha_ckpt, counter_ckpt = vb._alpha_alpha.copy(), vb._counter

    state = {"pos": 0}

    def window_step():
        if state["pos"] == PROTOCOL_WINDOW:
            ctx.model_checkpoint(restore=True)
            vb._alpha_alpha = alpha_ckpt.copy()
            vb._counter = counter_ckpt
            state["pos"] = 0
        state["pos"] += 1
        return vb.learning()

    for _ in range(max(0, warmup - PROTOCOL_WARMUP)):
        window_step()
    job.torch.cuda.synchronize()
    # ... and untimed steps of the same window until the process has run >= 2 s of them (large corpora: none needed): the
    # timed steps - and the shard proxies, which do the same - run at the clock the chip SUSTAINS under this load
    if steady:
        def one_window():
            t0 = time.perf_counter()
            for _ in range(PROTOCOL_WINDOW):
                window_step()
            job.torch.cuda.synchronize()
            return time.perf_counter() - t0
        # (the same count on every rank - a step holds collectives: the slowest rank's first window sizes the rest)
        t_window = job.reduce([one_window()], "max")[0]
        for _ in range(int(np.ceil(2.0 / max(t_window, 1e-3))) - 1):
            one_window()
    state["pos"] = PROTOCOL_WINDOW              # (the timed region starts at the window's first iteration)
    ctx.set_profiling(True)
    ctx.kernel_time()
    ctx.work_counters()
    vb._train_corpus.plan()
    job.barrier()
    stamps = [time.perf_counter()]
    positions = []
    for _ in range(steps):
        joint = window_step()
        positions.append(state["pos"])
        stamps.append(time.perf_counter())
    job.barrier()
    elapsed = time.perf_counter() - stamps[0]
    doc_ms, ss_ms, calls = ctx.kernel_time()
    sum_iters, sum_iter_terms = ctx.work_counters()
    tile_entries, handed_over = ctx.executed_work()          # (the same read of the device counters)
    clock_mhz = ctx.shader_clock_mhz()
    classes = vb._train_corpus.plan()
    ctx.set_profiling(False)
    step_ms = np.diff(np.array(stamps)) * 1e3
    per_position = [float(np.mean([m for m, q in zip(step_ms, positions) if q == pos + 1] or [np.nan]))
                    for pos in range(PROTOCOL_WINDOW)]

    elapsed = job.reduce([elapsed], "max")[0]
    sums = job.reduce([float(D_local), float(nnz_local), sum_iters] + [float(x) for x in corpus_checksum(ptr, ids, cts)])
    nnz_max = job.reduce([float(nnz_local)], "max")[0]
    D_total, nnz_total, sum_iters_total = int(sums[0]), int(sums[1]), sums[2]
    sums = sums[:2] + sums[3:]
    calls = max(1, calls)
    doc_ms, ss_ms = doc_ms / calls, ss_ms / calls
    for c in classes:
        c["kernel_ms"] = c["kernel_ms"] / calls
    B = algorithmic_bytes(nnz_local, D_local, K)
    kernel_ms = doc_ms + ss_ms
    achieved = B / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    # (the committed passes profiled the whole corpus on ONE GPU: they describe this rank's launch only if it holds all of it)
    whole = docs is None and (job.world == 1 or wl["scaling"] == "weak")
    traffic, traffic_source = traffic_record(name if whole else None)
    if not whole and docs is None:
        traffic_source = "profiles/traffic_%s.json describes the whole corpus on one GPU; rank 0 holds 1 / %d of it here" % (name, job.world)
    flops = 4.0 * K * sum_iter_terms / calls          # two mat-vecs per inner iteration actually executed, rank 0
    tflops = flops / (doc_ms * 1e-3) / 1e12 if doc_ms > 0 else 0.0
    # ... of which the kernels ran only the live part through the FMA pipes: the dense kernels K columns per term and
    # iteration, the live-topic kernel (estep_compact.h) the columns of its tile; the rest are the dead topics' exact zeros
    executed = 4.0 * tile_entries / calls
    peak_at_clock = FP64_VECTOR_PEAK_TFLOPS * clock_mhz / PEAK_CLOCK_MHZ if clock_mhz else None
    rec = {
        "value": D_total * steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "scaling": wl["scaling"],
        "doc_iterations_per_s": sum_iters_total / elapsed,
        "protocol": {"warmup_outer_iterations": PROTOCOL_WARMUP,
                     "window": "outer iterations %d-%d from the seeded start (SURVEY 8d); step i of the timed region is "
                               "iteration %d + (i mod %d), the model returns to its state after iteration %d through a "
                               "device checkpoint inside the timed region" % (PROTOCOL_WARMUP + 1, PROTOCOL_WARMUP + PROTOCOL_WINDOW,
                                                                             PROTOCOL_WARMUP + 1, PROTOCOL_WINDOW, PROTOCOL_WARMUP),
                     "median_ms_per_step": float(np.median(step_ms)),
                     "ms_per_window_iteration": per_position,
                     "mean_inner_iterations": sum_iters / calls / max(1, D_local)},
        "config": {"workload": wl["label"], "docs_total": D_total, "nnz_total": nnz_total,
                   "docs_per_gpu": D_local, "nnz_per_gpu": nnz_local, "tokens_per_gpu": tokens_local,
                   "nnz_imbalance": nnz_max * job.world / max(1, nnz_total) - 1.0,
                   "K": K, "V": V, "inner_iterations_cap": 50, "parallelism": "dp%d" % job.world,
                   "step": "e_step + sstats all-reduce + device m_step + alpha update",
                   "corpus_checksum": [int(x) for x in sums[2:]]},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "E-step document kernels (concurrent launch classes) + sufficient-statistics pass "
                               "(gather + finalize), rank 0, HIP events on the launch streams",
                     "kernel_ms": kernel_ms, "kernel_ms_documents": doc_ms, "kernel_ms_sstats": ss_ms,
                     "algorithmic_bytes": B, "launch_classes": classes,
                     "statistics_gather": {"document_blocks": vb._train_corpus.layout("gather_blocks"),
                                           "segments": vb._train_corpus.layout("gather_segments"),
                                           "rounds": vb._train_corpus.layout("gather_rounds"),
                                           "sweep_passes": vb._train_corpus.layout("gather_sweep_passes"),
                                           "partial_row_bytes": vb._train_corpus.layout("gather_partial_rows") * 8 *
                                           ctx_table_stride(ctx)}},
        # the honest companion: the document kernels are fp64-VALU / latency bound, not HBM bound;
        # flops = 4 K sum_d I_d N_d of the inner iterations executed IN THE TIMED WINDOW (device counters)
        "roofline_fp64": {"bound": "fp64 vector FMA", "achieved": tflops, "peak": FP64_VECTOR_PEAK_TFLOPS,
                          "unit": "TFLOP/s", "frac": tflops / FP64_VECTOR_PEAK_TFLOPS, "flops_per_launch": flops,
                          "mean_inner_iterations": sum_iters / calls / max(1, D_local),
                          # algorithmic flops (4 K sum_d I_d N_d) above, so that the fraction stays comparable
                          # with rounds 1-5; what was EXECUTED, and the clock the chip sustained, beside it
                          "executed_flops": executed, "live_fraction": executed / flops if flops > 0 else None,
                          "executed_achieved": executed / (doc_ms * 1e-3) / 1e12 if doc_ms > 0 else 0.0,
                          "documents_handed_to_live_topic_kernel": handed_over / calls,
                          "clock_mhz": clock_mhz, "peak_clock_mhz": PEAK_CLOCK_MHZ,
                          "peak_at_clock": peak_at_clock,
                          "frac_at_clock": tflops / peak_at_clock if peak_at_clock else None,
                          "clock_source": "resident kernels of the timed E-steps time themselves in shader cycles "
                                          "(s_memtime) and in ticks of the constant-rate counter (s_memrealtime): "
                                          "pylda_clock_counters"},
        "joint_log_likelihood": joint,
        "startup": {"generate_corpus_s": t_gen, "upload_and_schedule_s": t_init,
                    "first_step_s": t_step1, "first_%d_steps_s" % PROTOCOL_WARMUP: t_first,
                    "note": "the first E-step also builds the corpus' postings (CSC) for the statistics gather"},
    }
    expected = EXPECTED_CHECKSUM.get(name)
    if expected is not None and docs is None and (name == "synth1m" or job.world == 1):
        assert rec["config"]["corpus_checksum"] == expected, \
            "generated corpus differs from the recorded one: %r vs %r" % (rec["config"]["corpus_checksum"], expected)
    return rec, vb, ctx, wl


def cpu_leg(ctx, vb, wl, args, budget_s, max_docs, value, all_cores=0):
    """CPU baseline + per-document log-likelihood delta on a bounded sample (rank 0, N = 1)."""
    ptr, ids, cts = wl["ptr"], wl["ids"], wl["cts"]
    alpha = vb._alpha_alpha.copy()
    eta = vb._eta.copy()
    rate, n, cpu_ll = cpu_baseline(alpha, eta, ptr, ids, cts, budget_s, max_docs)
    sample = ctx.corpus(ptr[:n + 1], ids[:ptr[n]], cts[:ptr[n]])
    gpu = ctx.estep_host(sample, alpha, eta)
    ctx.set_option("doc_values", 0)
    sample.close()
    delta = np.abs(gpu["doc_ll"] - cpu_ll) / np.abs(cpu_ll)
    rec = {
        "cpu_baseline": {
            "value": rate, "unit": "docs/s", "cores": 1, "kind": "port",
            "sample": "first %d documents of rank 0's corpus, numpy/scipy restatement of "
                      "variational_bayes.py:132-216 (oracle/vb_numpy.py), single thread, median of 3 timings of the "
                      "same documents; host has %d cores" % (n, os.cpu_count()),
            "c_port_docs_per_s": c_oracle_rate(alpha, eta, ptr, ids, cts, n)},
        "ll_delta": {"max_rel": float(delta.max()), "median_rel": float(np.median(delta)),
                     "docs": int(n), "bar": 1e-5},
        "speedup_vs_cpu": value / rate,
    }
    if all_cores:
        try:
            workers = min(int(all_cores), os.cpu_count() or 1)
            t0 = time.perf_counter()
            rate_all, n_all, workers = cpu_baseline_all_cores(alpha, eta, ptr, ids, cts, min(budget_s, 8.0), workers)
            rec["cpu_baseline"]["all_cores"] = {
                "value": rate_all, "unit": "docs/s", "cores": workers, "kind": "port",
                "sample": "%d documents in %d single-threaded processes side by side (the same numpy/scipy "
                          "restatement, sum of the per-process rates; %.1f s wall incl. process start-up)"
                          % (n_all, workers, time.perf_counter() - t0)}
            rec["speedup_vs_cpu_all_cores"] = value / rate_all
        except Exception as e:      # the leg is a report, not a dependency of the measurement
            rec["cpu_baseline"]["all_cores"] = {"value": None, "error": repr(e)[:200]}
    return rec


def host_contract_leg(vb):
    """The reference's public seam at speed: e_step() -> (float, ndarray (K, V)) and m_step(ndarray) with the
    arrays crossing PCIe (page-locked buffers, pylda_host_alloc), timed OUTSIDE the bench's timed region; the
    device-resident learning() is what `value` measures."""
    import gc
    best_e, best_m = float("inf"), float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        ll, sstats = vb.e_step()
        t1 = time.perf_counter()
        vb.m_step(sstats)
        t2 = time.perf_counter()
        best_e, best_m = min(best_e, (t1 - t0) * 1e3), min(best_m, (t2 - t1) * 1e3)
        del sstats
        gc.collect()
    return {"e_step_ms": best_e, "m_step_ms": best_m,
            "note": "public e_step() / m_step() with host ndarrays (sufficient statistics K x V down and up again), best of 3"}


XGMI_LINK_GBPS = 153.0           # one of a GPU's 7 peer links; a ring all-reduce is bound by one link


def shard_proxy(job, args, step_ms_full, clock_full=None):
    """Single-GPU PROXY for the strong-scaling curve of cfg 4 (no multi-GPU node is available to this run): rank 0's
    step at the shard sizes N = 2 / 4 / 8 would produce - 1M / N documents with the FULL K x V tables - measured on
    this one GPU, plus a MODEL of the exchange (bytes / xGMI link figure, not overlapped).  Shows what does not
    shrink with the shard: table preparation, the statistics pass over all V terms, the M-step, the all-reduce."""
    import contextlib
    import io
    from pylda_amd import distributed
    from pylda_amd.variational_bayes import VariationalBayes
    rows = []
    for n in (2, 4, 8):
        wl = build_workload("synth1m", 0, n, job.device, args.docs)
        ptr, ids, cts, V, K = wl["ptr"], wl["ids"], wl["cts"], wl["V"], wl["K"]
        np.random.seed(0)
        eta0 = np.random.gamma(100., 1. / 100., (K, V))
        vb = VariationalBayes(hyper_parameter_optimize_interval=1, device=job.local_rank)
        vb._verbose = False
        vb._initialize_parsed(ptr, ids, cts, V, K, 1.0 / K, 1.0 / V, eta=eta0)
        ctx = vb._context()
        distributed.bind_to_torch_stream(ctx)
        for _ in range(PROTOCOL_WARMUP):
            vb.learning()
        job.torch.cuda.synchronize()
        ctx.model_checkpoint()
        alpha_ckpt, counter_ckpt = vb._alpha_alpha.copy(), vb._counter

        def restore():
            ctx.model_checkpoint(restore=True)
            vb._alpha_alpha = alpha_ckpt.copy()
            vb._counter = counter_ckpt

        # >= 2 s of the SAME window, untimed, before the timed one: every leg is measured at the clock the chip sustains
        # under this load, not at what a short run catches on the way up or down (round 5's N = 8 leg timed 0.27 s after
        # 0.16 s of warm-up and came out 5 % "super-linear"); the sustained clock of each leg is in the record
        t_steady = time.perf_counter()
        while time.perf_counter() - t_steady < 2.0:
            for _ in range(PROTOCOL_WINDOW):
                vb.learning()
            job.torch.cuda.synchronize()
            restore()
        ctx.set_profiling(True)
        ctx.kernel_time()
        ctx.work_counters()
        vb._verbose = True                      # (stream marks around the E-step and M-step spans; the line itself is swallowed)
        walls, e_span, m_span = [], [], []
        for _ in range(PROTOCOL_WINDOW):
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                vb.learning()
            walls.append((time.perf_counter() - t0) * 1e3)
            e_span.append(ctx.elapsed_ms(0, 1))
            m_span.append(ctx.elapsed_ms(1, 2))
        ctx.work_counters()
        clock_mhz = ctx.shader_clock_mhz()
        doc_ms, ss_ms, calls = ctx.kernel_time()
        ctx.set_profiling(False)
        calls = max(1, calls)
        doc_ms, ss_ms = doc_ms / calls, ss_ms / calls
        ldk = ctx_table_stride(ctx)
        nbytes = V * ldk * 8                    # what crosses the links: the V x ldk statistics pylda_allreduce_sstats reduces (csrc/comm.hip)
        ring = 2.0 * (n - 1) / n * nbytes / (XGMI_LINK_GBPS * 1e9) * 1e3
        direct = 2.0 * nbytes / n / (XGMI_LINK_GBPS * 1e9) * 1e3
        step = float(np.mean(walls))
        predicted = step + ring
        total_docs = args.docs or 1000000
        rows.append({"n_gpus_modelled": n, "docs_rank0": len(ptr) - 1, "nnz_rank0": int(ptr[-1]),
                     "ms_per_step_measured": step, "kernel_ms_documents": doc_ms, "kernel_ms_sstats": ss_ms,
                     "estep_span_ms": float(np.mean(e_span)), "mstep_span_ms": float(np.mean(m_span)),
                     "table_prep_ms": float(np.mean(e_span)) - doc_ms - ss_ms,
                     "host_and_launch_ms": step - float(np.mean(e_span)) - float(np.mean(m_span)),
                     "allreduce_bytes": nbytes, "table_stride": ldk, "clock_mhz": clock_mhz,
                     "allreduce_ms_model_ring": ring, "allreduce_ms_model_direct": direct,
                     "predicted_ms_per_step": predicted, "predicted_docs_per_s": total_docs / (predicted * 1e-3),
                     "predicted_strong_scaling_efficiency": step_ms_full / (n * predicted)})
        release(vb)
        del vb, ctx, wl
    return {"model": True,
            "note": "NOT a multi-GPU measurement: rank 0's shard of cfg 4 for N = 2 / 4 / 8 timed on ONE GPU (full K x V "
                    "tables, nnz-balanced document shard; every leg after >= 2 s of the same steps, its sustained shader "
                    "clock recorded) + a modelled, non-overlapped ring all-reduce of the V x ldk statistics at %.0f GB/s "
                    "per xGMI link; efficiency = T(1) / (N * (T_shard(N) + T_allreduce(N)))" % XGMI_LINK_GBPS,
            "ms_per_step_n1": step_ms_full, "ms_per_step_n1_clock_mhz": clock_full, "per_n": rows}


def ctx_table_stride(ctx):
    return int(ctx._lib.pylda_table_stride(ctx._h))


def release(vb):
    if vb._train_corpus is not None:
        vb._train_corpus.close()
        vb._train_corpus = None
    if vb._ctx is not None:
        vb._ctx.close()
        vb._ctx = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="synth1m", choices=["synth100k", "synth1m", "ap", "nips"])
    ap.add_argument("--docs", type=int, default=None, help="override the corpus size (smoke runs)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg")
    ap.add_argument("--cpu-workers", type=int, default=32,
                    help="processes of the all-cores CPU figure (0: skip; capped at the host's core count). "
                         "The 256-core host of the GPU box peaks near 32: 1184 docs/s at cfg 3, 938 with 128 "
                         "(the gathered K x V table does not fit its caches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", "--no-ap-extra", dest="no_extras", action="store_true",
                    help="primary record only (no cfg 2 / cfg 3 / cfg 5 sub-records)")
    ap.add_argument("--no-shard-proxy", action="store_true",
                    help="skip the single-GPU proxy of the cfg 4 strong-scaling curve (rank 0's shard for N = 2 / 4 / 8 + modelled exchange)")
    ap.add_argument("--extra-docs", type=int, default=None, help="corpus size (per GPU) of the cfg 3 sub-record (smoke runs)")
    ap.add_argument("--variant", type=int, default=-1, help="force a kernel variant (A/B runs)")
    ap.add_argument("--option", action="append", default=[], help="name=value library option (A/B runs)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="test mode: all ranks on GPU 0, exchange over gloo (exercises the N > 1 path on a 1-GPU box)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # invoked as plain `python bench.py --gpus N`: become the launcher, one rank per GPU
        os.execv(sys.executable, launcher_argv(args.gpus, sys.argv[1:]))

    job = Job(args)
    rec, vb, ctx, wl = measure(job, args, args.workload, args.steps, args.warmup, args.docs)
    K = wl["K"]
    out = None
    if job.rank == 0:
        out = {
            "metric": "documents/sec per VB iteration",
            "value": rec["value"], "unit": "docs/s", "n_gpus": job.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": rec["ms_per_step"], "higher_is_better": True,
            "scaling": rec["scaling"], "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if args.workload.startswith("synth") else "parsed corpus fixture (tests/golden)",
            "config": rec["config"], "roofline": rec["roofline"],
            "roofline_fp64": rec["roofline_fp64"], "doc_iterations_per_s": rec["doc_iterations_per_s"],
            "protocol": rec["protocol"],
            "joint_log_likelihood": rec["joint_log_likelihood"], "startup": rec["startup"],
            "primary_workload_note": "since round 5 the top-level record is cfg 4 (the 1M-document corpus north_star "
                                     "quotes 1/2/4/8-GPU throughput on, strong scaling) at EVERY N; rounds 1-4 had cfg 3 "
                                     "there, which is now the sub-record `synth100k` (same protocol, comparable with "
                                     "BENCH_r01-r04's top level)",
        }
        if not args.no_cpu_baseline and job.world == 1:
            out.update(cpu_leg(ctx, vb, wl, args, args.cpu_seconds, 2000 if args.workload != "synth1m" else 600, rec["value"],
                               all_cores=args.cpu_workers))
        if job.world == 1:
            out["host_array_contract"] = host_contract_leg(vb)
    release(vb)
    del vb, ctx, wl

    extras = not args.no_extras and args.workload == "synth1m"
    if extras and job.world == 1 and not args.no_shard_proxy:
        # ---- single-GPU proxy of the strong-scaling curve: rank 0's shard of cfg 4 for N = 2 / 4 / 8 ----
        out["shard_proxy"] = shard_proxy(job, args, rec["ms_per_step"], rec["roofline_fp64"].get("clock_mhz"))
    if extras:
        # ---- cfg 3 (100k documents per GPU, K=128) alongside, every N: weak scaling + its own roofline ----
        try:
            rec3, vb3, ctx3, wl3 = measure(job, args, "synth100k", PROTOCOL_WINDOW, PROTOCOL_WARMUP, args.extra_docs)
            if job.rank == 0:
                sub = {"value": rec3["value"], "unit": "docs/s", "n_gpus": job.world, "steps": PROTOCOL_WINDOW,
                       "warmup": PROTOCOL_WARMUP, "ms_per_step": rec3["ms_per_step"], "scaling": "weak",
                       "config": rec3["config"], "roofline": rec3["roofline"], "roofline_fp64": rec3["roofline_fp64"],
                       "doc_iterations_per_s": rec3["doc_iterations_per_s"], "protocol": rec3["protocol"],
                       "startup": rec3["startup"]}
                if not args.no_cpu_baseline and job.world == 1:
                    sub.update(cpu_leg(ctx3, vb3, wl3, args, min(args.cpu_seconds, 10.0), 1000, rec3["value"]))
                out["synth100k"] = sub
            release(vb3)
            del vb3, ctx3, wl3
        except AssertionError:
            raise
        except Exception as exc:
            if job.group is not None:
                raise
            out["synth100k"] = {"error": repr(exc)}
    if extras:
        # ---- cfg 2 (associated-press K=10) and cfg 5 (nips.88-05 K=500), every N: documents sharded over the ranks ----
        for key, fn in (("ap_k10", ap_extra), ("nips_k500", nips_extra)):
            try:
                sub = fn(job, args)
                if job.rank == 0:
                    out[key] = sub
            except AssertionError:
                raise
            except Exception as exc:                # the fixture may be absent in a stripped tree
                if job.group is not None:
                    raise
                out[key] = {"error": repr(exc)}
    if job.rank == 0:
        print(json.dumps(out), flush=True)
    if job.group is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def launcher_argv(gpus, argv):
    """Command line that runs this script as `gpus` ranks of one node (rendezvous on 127.0.0.1)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def ap_extra(job, args):
    """BASELINE.json cfg 2: AP K=10 E-step vs the CPU path and the goldens; at N > 1 the 2000 documents are
    sharded (nnz-balanced contiguous ranges), every rank runs the E-step on its shard."""
    from pylda_amd import _capi
    from pylda_amd.corpus import shard_bounds
    g = np.load(os.path.join(ROOT, "tests", "golden", "ap_train_k10.npz"))
    ptr, ids, cts = g["doc_ptr"].astype(np.int64), g["term_id"].astype(np.int32), g["term_ct"].astype(np.int32)
    alpha, eta = g["alpha"], g["eta"]
    D_total = len(ptr) - 1
    bounds = shard_bounds(ptr, job.world)
    lo, hi = int(bounds[job.rank]), int(bounds[job.rank + 1])
    sptr = ptr[lo:hi + 1] - ptr[lo]
    sids, scts = ids[ptr[lo]:ptr[hi]], cts[ptr[lo]:ptr[hi]]
    ctx = _capi.Context(10, eta.shape[1], device=job.local_rank)
    corpus = ctx.corpus(sptr, sids, scts)
    ctx.set_alpha(alpha)
    ctx.set_eta(eta)
    for _ in range(3):
        ctx.estep(corpus)
    ctx.synchronize()
    reps = 20
    ctx.set_profiling(True)
    ctx.kernel_time()
    job.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.estep(corpus)
        ctx.estep_results(corpus)
    job.barrier()
    elapsed = job.reduce([time.perf_counter() - t0], "max")[0]
    gpu_rate = D_total * reps / elapsed
    doc_ms, ss_ms, calls = ctx.kernel_time()
    ctx.set_profiling(False)
    doc_ll, _, iters = ctx.get_doc_values(corpus)
    delta = np.abs(doc_ll - g["doc_ll"][lo:hi]) / np.abs(g["doc_ll"][lo:hi])
    worst, same = job.reduce([float(delta.max())], "max")[0], job.reduce([float(np.sum(iters == g["iters"][lo:hi]))])[0]
    out = {"gpu_docs_per_s": gpu_rate, "n_gpus": job.world, "estep_ms": elapsed / reps * 1e3,
           "kernel_ms_documents": doc_ms / max(1, calls), "kernel_ms_sstats": ss_ms / max(1, calls),
           "launch_classes": len(corpus.plan()),
           "max_rel_ll_delta_vs_reference": worst, "iters_equal_fraction": same / D_total}
    if not args.no_cpu_baseline and job.rank == 0 and job.world == 1:
        rate, n, _ = cpu_baseline(alpha, eta, ptr, ids, cts, min(args.cpu_seconds, 6.0), 2000)
        out.update({"cpu_docs_per_s": rate, "cpu_sample_docs": n, "speedup": gpu_rate / rate})
    corpus.close()
    ctx.close()
    return out


def nips_extra(job, args):
    """BASELINE.json cfg 5: parsed/nips.88-05, K=500, 50 learning() iterations from the reference's seeded start,
    documents sharded over the N ranks (one all-reduce of the K x V statistics per iteration); E-step time / docs/s,
    the joint log-likelihood of EVERY iteration against the reference's own trace, and the held-out per-token
    log-likelihood after 50 iterations (rank 0) against the reference's value (tests/golden/nips_trace_k500.npz)."""
    from pylda_amd.corpus import shard_bounds
    from pylda_amd.variational_bayes import VariationalBayes
    g = np.load(os.path.join(ROOT, "tests", "golden", "nips_trace_k500.npz"))
    K, V = int(g["K"]), len(g["words"])
    ptr, ids, cts = g["doc_ptr"].astype(np.int64), g["term_id"].astype(np.int32), g["term_ct"].astype(np.int32)
    test = (g["test_doc_ptr"].astype(np.int64), g["test_term_id"].astype(np.int32), g["test_term_ct"].astype(np.int32))
    D, nnz = len(ptr) - 1, int(ptr[-1])
    bounds = shard_bounds(ptr, job.world)
    lo, hi = int(bounds[job.rank]), int(bounds[job.rank + 1])
    np.random.seed(int(g["seed"]))
    m = VariationalBayes(device=job.local_rank, process_group=job.group)
    m._verbose = False
    # eta: the seeded draw of variational_bayes.py:95 (the same on every rank)
    m._initialize_parsed(ptr[lo:hi + 1] - ptr[lo], ids[ptr[lo]:ptr[hi]], cts[ptr[lo]:ptr[hi]], V, K, 1.0 / K, 1.0 / V)
    ctx = m._context()
    n_iter = min(50, len(g["joint_ll"]))
    trace = [m.learning()]
    ctx.synchronize()
    ctx.set_profiling(True)
    ctx.kernel_time()
    job.barrier()
    t0 = time.perf_counter()
    for _ in range(n_iter - 1):
        trace.append(m.learning())
    job.barrier()
    elapsed = job.reduce([time.perf_counter() - t0], "max")[0]
    doc_ms, ss_ms, calls = ctx.kernel_time()
    classes = m._train_corpus.plan()
    ctx.set_profiling(False)
    calls = max(1, calls)
    joint = trace[-1]
    ref_trace = np.asarray(g["joint_ll"][:n_iter], dtype=np.float64)
    trace_delta = float(np.max(np.abs(np.array(trace) - ref_trace) / np.abs(ref_trace)))
    assert trace_delta < 1e-8, "nips K=500 joint log-likelihood trace differs from the reference's: %g" % trace_delta
    heldout = {int(it): v for it, v in g["heldout"]}
    tokens = int(g["test_tokens"])
    B = algorithmic_bytes(int(ptr[hi] - ptr[lo]), hi - lo, K)
    kernel_ms = (doc_ms + ss_ms) / calls
    for c in classes:
        c["kernel_ms"] /= calls
    out = {"docs_per_s": D * (n_iter - 1) / elapsed, "n_gpus": job.world, "ms_per_step": elapsed / (n_iter - 1) * 1e3,
           "iterations": n_iter,
           "config": {"workload": "parsed/nips.88-05, K=500, train = first 2235 documents (sharded by document), test = last 248",
                      "docs": D, "nnz": nnz, "K": K, "V": V, "docs_rank0": hi - lo},
           "roofline": {"bound": "hbm", "achieved": B / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": B / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "kernel_ms": kernel_ms,
                        "kernel_ms_documents": doc_ms / calls, "kernel_ms_sstats": ss_ms / calls,
                        "algorithmic_bytes": B, "launch_classes": classes,
                        "note": "rank 0's share of 2235 workgroup-documents on 256 CUs: latency-bound, not a bandwidth measurement"},
           "joint_log_likelihood": joint, "reference_joint_log_likelihood": float(ref_trace[-1]),
           "joint_rel_delta": abs(joint - ref_trace[-1]) / abs(ref_trace[-1]),
           "joint_trace_max_rel_delta": trace_delta}
    if job.rank == 0:
        wll, _ = m.e_step(test)
        out["heldout_per_token_log_likelihood"] = wll / tokens
        if n_iter in heldout:
            out["reference_heldout_per_token_log_likelihood"] = heldout[n_iter] / tokens
            out["heldout_rel_delta"] = abs(wll - heldout[n_iter]) / abs(heldout[n_iter])
        if job.world == 1 and not args.no_shard_proxy:
            out["shard_proxy"] = nips_shard_proxy(job, g, out["ms_per_step"])
        if not args.no_cpu_baseline and job.world == 1:
            rate, n, _ = cpu_baseline(m._alpha_alpha.copy(), m._eta.copy(), ptr, ids, cts, min(args.cpu_seconds, 10.0), 200)
            out["cpu_baseline"] = {"value": rate, "unit": "docs/s", "cores": 1, "kind": "port",
                                   "sample": "first %d training documents, numpy/scipy restatement, single thread" % n}
            out["speedup_vs_cpu"] = out["docs_per_s"] / rate
    job.barrier()
    release(m)
    return out


def nips_shard_proxy(job, g, step_ms_full):
    """Single-GPU PROXY of cfg 5 on N = 2 / 4 / 8 GPUs (BASELINE.json puts it on 8): rank 0's nnz-balanced shard of the
    2235 training documents (1118 / 559 / 280) with the full K x V model, timed alone on this GPU, + the modelled ring
    all-reduce of the K x V statistics.  A document is one workgroup on one CU and its 50 inner iterations are a
    serial chain, so the step cannot shrink below ceil(documents / CUs) chains: `residency_rounds`."""
    from pylda_amd.corpus import shard_bounds
    from pylda_amd.variational_bayes import VariationalBayes
    K, V = int(g["K"]), len(g["words"])
    ptr, ids, cts = g["doc_ptr"].astype(np.int64), g["term_id"].astype(np.int32), g["term_ct"].astype(np.int32)
    rows = []
    for n in (2, 4, 8):
        hi = int(shard_bounds(ptr, n)[1])
        np.random.seed(int(g["seed"]))
        m = VariationalBayes(device=job.local_rank)
        m._verbose = False
        m._initialize_parsed(ptr[:hi + 1], ids[:ptr[hi]], cts[:ptr[hi]], V, K, 1.0 / K, 1.0 / V)
        ctx = m._context()
        for _ in range(5):
            m.learning()
        ctx.synchronize()
        ctx.set_profiling(True)
        ctx.kernel_time()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            m.learning()
        ctx.synchronize()
        step = (time.perf_counter() - t0) / reps * 1e3
        doc_ms, ss_ms, calls = ctx.kernel_time()
        classes = m._train_corpus.plan()
        ctx.set_profiling(False)
        calls = max(1, calls)
        num_cu = job.torch.cuda.get_device_properties(job.device).multi_processor_count
        ring = 2.0 * (n - 1) / n * V * ctx_table_stride(ctx) * 8 / (XGMI_LINK_GBPS * 1e9) * 1e3      # (V x ldk: what pylda_allreduce_sstats reduces)
        rows.append({"n_gpus_modelled": n, "docs_rank0": hi, "nnz_rank0": int(ptr[hi]), "ms_per_step_measured": step,
                     "kernel_ms_documents": doc_ms / calls, "kernel_ms_sstats": ss_ms / calls,
                     "launch_classes": [(c["kernel"], c["geometry"], c["documents"]) for c in classes],
                     "residency_rounds": -(-hi // num_cu), "allreduce_ms_model_ring": ring,
                     "predicted_ms_per_step": step + ring,
                     "predicted_speedup_vs_n1": step_ms_full / (step + ring)})
        release(m)
    return {"model": True, "ms_per_step_n1": step_ms_full, "per_n": rows,
            "note": "NOT a multi-GPU measurement: rank 0's shard of the nips.88-05 training split for N = 2 / 4 / 8 timed on "
                    "ONE GPU (iterations 6-25 from the seeded start of the SHARD's own model) + a modelled, non-overlapped ring "
                    "all-reduce of the K x V statistics at %.0f GB/s per xGMI link" % XGMI_LINK_GBPS}


if __name__ == "__main__":
    main()
