#!/usr/bin/env python3
"""bench.py - documents/sec per VB iteration of the MI355X E-step path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload synth100k|synth1m|ap]

A "step" is one outer VB iteration over the rank's resident corpus: the hot
path (device compute_dirichlet_expectation + per-document phi/gamma kernel +
sufficient-statistics accumulation, variational_bayes.py:132-216), the RCCL
all-reduce of the K*V sufficient statistics when N > 1, and the device M-step /
alpha update that make the next iteration start from a new model (nothing is
cached between steps).  Inputs are resident in HBM before the timed region.

Workloads (BASELINE.json configs):
  synth100k  cfg 3: synthetic LDA corpus, 100,000 docs PER GPU, V=50k, K=128,
             mean 200 tokens/doc - weak scaling (default; the single-GPU
             roofline configuration)
  synth1m    cfg 4: 1,000,000 docs TOTAL, V=100k, K=256, sharded over N GPUs -
             strong scaling
  ap         cfg 2: associated-press train split (committed parsed fixture),
             K=10, replicated per GPU (latency-bound, 2000 documents)

Prints ONE JSON line on rank 0 (contract in the task statement) carrying
`roofline` (HBM bound, algorithmic bytes B = nnz*(8+16K) + D*(8K+8) per launch,
SURVEY 8d) and `cpu_baseline` (the numpy restatement of the reference timed on
this host, single thread, bounded sample).
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
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X fp64 vector peak (spec)
CHUNK = 25000


def algorithmic_bytes(nnz, D, K):
    return nnz * (8 + 16 * K) + D * (8 * K + 8)


def build_workload(name, rank, world, device, docs_override=None):
    from pylda_amd.corpus import synthetic_lda_corpus_torch
    if name == "synth100k":
        per_gpu = docs_override or 100000
        V, K, seed = 50000, 128, 1234
        chunks = (per_gpu + CHUNK - 1) // CHUNK
        ptr, ids, cts = synthetic_lda_corpus_torch(per_gpu * world, V, 128, 200, seed, device=device,
                                                   chunk=CHUNK, first_chunk=rank * chunks,
                                                   shard_chunks=chunks)
        return dict(ptr=ptr, ids=ids, cts=cts, V=V, K=K, scaling="weak",
                    label="synthetic LDA corpus cfg3: %d docs/GPU, V=50000, K=128, mean 200 tokens/doc" % per_gpu)
    if name == "synth1m":
        total = docs_override or 1000000
        V, K, seed = 100000, 256, 5678
        n_chunks = (total + CHUNK - 1) // CHUNK
        per = (n_chunks + world - 1) // world
        ptr, ids, cts = synthetic_lda_corpus_torch(total, V, 128, 200, seed, device=device, chunk=CHUNK,
                                                   first_chunk=rank * per, shard_chunks=per)
        return dict(ptr=ptr, ids=ids, cts=cts, V=V, K=K, scaling="strong",
                    label="synthetic LDA corpus cfg4: %d docs total, V=100000, K=256, sharded" % total)
    if name == "ap":
        g = np.load(os.path.join(ROOT, "tests", "golden", "ap_train_k10.npz"))
        return dict(ptr=g["doc_ptr"].astype(np.int64), ids=g["term_id"].astype(np.int32),
                    cts=g["term_ct"].astype(np.int32), V=int(g["eta"].shape[1]), K=10,
                    scaling="weak", eta=g["eta"], alpha=g["alpha"],
                    label="associated-press train split (2000 docs, V=6806), K=10, replicated per GPU")
    raise SystemExit("unknown workload %r" % name)


def cpu_baseline(alpha, eta, ptr, ids, cts, budget_s, max_docs):
    """The reference's algorithm on the host CPU: numpy restatement (what the
    reference itself executes), single thread, on a bounded prefix of the corpus."""
    from oracle import vb_numpy
    E_log_eta = vb_numpy.compute_dirichlet_expectation(eta)
    doc_ll = []
    t0 = time.perf_counter()
    n = 0
    while n < max_docs and n < len(ptr) - 1:
        lo, hi = int(ptr[n]), int(ptr[n + 1])
        _, ll, _, _, _ = vb_numpy.e_step_document(alpha, E_log_eta, ids[lo:hi].astype(np.int64),
                                                  cts[lo:hi])
        doc_ll.append(ll)
        n += 1
        if time.perf_counter() - t0 > budget_s and n >= 20:
            break
    elapsed = time.perf_counter() - t0
    return n / elapsed, n, np.array(doc_ll)


def c_oracle_rate(alpha, eta, ptr, ids, cts, n):
    from oracle import c_oracle
    c_oracle.load()
    t0 = time.perf_counter()
    c_oracle.e_step(alpha, eta, ptr[:n + 1], ids[:ptr[n]], cts[:ptr[n]])
    return n / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="synth100k", choices=["synth100k", "synth1m", "ap"])
    ap.add_argument("--docs", type=int, default=None, help="override the corpus size (smoke runs)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ap-extra", action="store_true")
    ap.add_argument("--variant", type=int, default=-1, help="force a kernel variant (A/B runs)")
    ap.add_argument("--option", action="append", default=[], help="name=value library option (A/B runs)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        group = dist.group.WORLD

    from pylda_amd import _capi, distributed
    from pylda_amd.variational_bayes import VariationalBayes

    wl = build_workload(args.workload, rank, world, device, args.docs)
    ptr, ids, cts, V, K = wl["ptr"], wl["ids"], wl["cts"], wl["V"], wl["K"]
    D_local, nnz_local, tokens_local = len(ptr) - 1, int(ptr[-1]), int(cts.sum())

    np.random.seed(0)
    eta0 = wl.get("eta")
    if eta0 is None:
        eta0 = np.random.gamma(100., 1. / 100., (K, V))          # variational_bayes.py:95
    vb = VariationalBayes(hyper_parameter_optimize_interval=1, device=local_rank, process_group=group)
    vb._verbose = False
    vb._initialize_parsed(ptr, ids, cts, V, K, 1.0 / K, 1.0 / V, eta=eta0)
    if "alpha" in wl:
        vb._alpha_alpha = wl["alpha"].copy()
    ctx = vb._context()
    if args.variant >= 0:
        ctx.set_option("force_variant", args.variant)
    for kv in args.option:
        name, value = kv.split("=")
        ctx.set_option(name, int(value))
    if group is None:
        distributed.bind_to_torch_stream(ctx)

    def barrier():
        if group is not None:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        vb.learning()
    ctx.set_profiling(True)
    ctx.kernel_time()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        joint = vb.learning()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, kernel_calls = ctx.kernel_time()
    ctx.set_profiling(False)

    totals = torch.tensor([elapsed, float(D_local), float(nnz_local)], dtype=torch.float64, device=device)
    if group is not None:
        import torch.distributed as dist
        tmax = totals.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
    D_total, nnz_total = int(totals[1]), int(totals[2])

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = D_total * args.steps / elapsed
        kernel_avg_ms = kernel_ms / max(1, kernel_calls)
        B = algorithmic_bytes(nnz_local, D_local, K)
        achieved = B / (kernel_avg_ms * 1e-3) / 1e9 if kernel_avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "documents/sec per VB iteration",
            "value": value, "unit": "docs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": wl["scaling"], "vs_baseline": None, "dtype": "f64", "data": "synthetic"
            if args.workload != "ap" else "associated-press (parsed fixture)",
            "config": {"workload": wl["label"], "docs_total": D_total, "nnz_total": nnz_total,
                       "docs_per_gpu": D_local, "nnz_per_gpu": nnz_local, "tokens_per_gpu": tokens_local,
                       "K": K, "V": V, "inner_iterations_cap": 50, "parallelism": "dp%d" % world,
                       "step": "e_step + sstats all-reduce + device m_step + alpha update"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "estep document kernels (one E-step's launches)",
                         "kernel_ms": kernel_avg_ms, "algorithmic_bytes": B},
            "joint_log_likelihood": joint,
        }
        # the honest companion: the kernel is fp64-VALU / latency bound, not HBM bound.
        # flops of the inner loops actually executed = sum_d I_d * 4 * N_d * K (two mat-vecs per iteration).
        try:
            import ctypes
            iters = np.empty(D_local, dtype=np.int32)
            ctx._check(ctx._lib.pylda_get_doc_values(ctx._h, vb._train_corpus._h, None, None,
                                                     iters.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
            work = float(np.dot(iters.astype(np.float64), np.diff(ptr).astype(np.float64))) * 4.0 * K
            tflops = work / (kernel_avg_ms * 1e-3) / 1e12 if kernel_avg_ms > 0 else 0.0
            out["roofline_fp64"] = {"bound": "fp64 vector FMA", "achieved": tflops, "peak": FP64_VECTOR_PEAK_TFLOPS,
                                    "unit": "TFLOP/s", "frac": tflops / FP64_VECTOR_PEAK_TFLOPS,
                                    "flops_per_launch": work, "mean_inner_iterations": float(iters.mean())}
        except Exception as exc:
            out["roofline_fp64"] = {"error": str(exc)}
        # ---- CPU baseline + per-document log-likelihood delta on a bounded sample ----
        if not args.no_cpu_baseline:
            alpha = vb._alpha_alpha.copy()
            eta = vb._eta.copy()
            rate, n, cpu_ll = cpu_baseline(alpha, eta, ptr, ids, cts, args.cpu_seconds, 2000)
            sample = ctx.corpus(ptr[:n + 1], ids[:ptr[n]], cts[:ptr[n]])
            gpu = ctx.estep_host(sample, alpha, eta)
            sample.close()
            delta = np.abs(gpu["doc_ll"] - cpu_ll) / np.abs(cpu_ll)
            out["cpu_baseline"] = {
                "value": rate, "unit": "docs/s", "cores": 1, "kind": "port",
                "sample": "first %d documents of rank 0's corpus, numpy/scipy restatement of "
                          "variational_bayes.py:132-216 (oracle/vb_numpy.py), single thread; "
                          "host has %d cores" % (n, os.cpu_count()),
                "c_port_docs_per_s": c_oracle_rate(alpha, eta, ptr, ids, cts, n),
            }
            out["ll_delta"] = {"max_rel": float(delta.max()), "median_rel": float(np.median(delta)),
                               "docs": int(n), "bar": 1e-5}
            out["speedup_vs_cpu"] = value / rate
        # ---- cfg 2 (associated-press K=10) alongside: speed-up target and parity ----
        if args.workload != "ap" and not args.no_ap_extra and world == 1:
            try:
                out["ap_k10"] = ap_extra(_capi, args)
            except Exception as exc:            # the fixture may be absent in a stripped tree
                out["ap_k10"] = {"error": str(exc)}
        print(json.dumps(out), flush=True)
    if group is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def ap_extra(_capi, args):
    """BASELINE.json cfg 2: AP K=10 E-step on one GPU vs the CPU path and the goldens."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "ap_train_k10.npz"))
    ptr, ids, cts = g["doc_ptr"].astype(np.int64), g["term_id"].astype(np.int32), g["term_ct"].astype(np.int32)
    alpha, eta = g["alpha"], g["eta"]
    ctx = _capi.Context(10, eta.shape[1])
    corpus = ctx.corpus(ptr, ids, cts)
    ctx.set_alpha(alpha)
    ctx.set_eta(eta)
    for _ in range(3):
        ctx.estep(corpus)
    ctx.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.estep(corpus)
        ctx.estep_results(corpus)
    gpu_rate = 2000 * reps / (time.perf_counter() - t0)
    doc_ll, _, iters = ctx.get_doc_values(corpus)
    delta = np.abs(doc_ll - g["doc_ll"]) / np.abs(g["doc_ll"])
    out = {"gpu_docs_per_s": gpu_rate, "estep_ms": 2000.0 / gpu_rate * 1e3,
           "max_rel_ll_delta_vs_reference": float(delta.max()),
           "iters_equal_fraction": float(np.mean(iters == g["iters"]))}
    if not args.no_cpu_baseline:
        rate, n, _ = cpu_baseline(alpha, eta, ptr, ids, cts, min(args.cpu_seconds, 6.0), 2000)
        out.update({"cpu_docs_per_s": rate, "cpu_sample_docs": n, "speedup": gpu_rate / rate})
    corpus.close()
    ctx.close()
    return out


if __name__ == "__main__":
    main()
