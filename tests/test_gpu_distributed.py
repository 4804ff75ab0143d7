"""The RCCL path on one GPU: a world-size-1 NCCL process group drives exactly the code
bench.py --gpus N runs per rank (zero-copy wrap of the library's device buffer,
stream binding, in-place all-reduce, small all-reduce)."""
import os
import socket

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.fixture(scope="module")
def nccl_group():
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist.group.WORLD
    dist.destroy_process_group()


def test_device_buffer_wrap_and_allreduce(nccl_group, ap_train):
    import torch
    from pylda_amd import _capi, distributed
    g = ap_train
    ctx = _capi.Context(10, 6806)
    distributed.bind_to_torch_stream(ctx)
    corpus = ctx.corpus(g["doc_ptr"][:201], g["term_id"][:g["doc_ptr"][200]], g["term_ct"][:g["doc_ptr"][200]])
    ctx.set_alpha(g["alpha"])
    ctx.set_eta(g["eta"])
    ctx.estep(corpus)
    before = ctx.get_sstats()
    t = distributed.allreduce_sstats(ctx, nccl_group)          # sum over 1 rank = identity, in place
    assert t.is_cuda and t.dtype == torch.float64 and t.numel() == ctx.sstats_elements()
    assert t.data_ptr() == ctx.sstats_device_ptr()             # zero-copy view of the library's buffer
    torch.cuda.synchronize()
    assert np.array_equal(ctx.get_sstats(), before)
    t.mul_(2.0)                                                # writes through to the library's memory
    torch.cuda.synchronize()
    assert np.array_equal(ctx.get_sstats(), 2.0 * before)
    ll, D, ass = distributed.allreduce_small(nccl_group, -12.5, 200, np.arange(10.0))
    assert ll == -12.5 and D == 200 and np.array_equal(ass, np.arange(10.0))
    corpus.close()
    ctx.close()


def test_sharded_learning_matches_single_process(nccl_group, ap_train):
    """VariationalBayes with a process group (world size 1) follows the same trajectory as without."""
    from pylda_amd.variational_bayes import VariationalBayes
    g = ap_train
    ptr = g["doc_ptr"][:401]
    ids, cts = g["term_id"][:ptr[-1]], g["term_ct"][:ptr[-1]]
    traces = []
    for group in (None, nccl_group):
        m = VariationalBayes(process_group=group)
        m._verbose = False
        m._initialize_parsed(ptr, ids, cts, 6806, 10, 0.1, 1.0 / 6806, eta=g["eta"].copy())
        traces.append([m.learning() for _ in range(3)] + [m._alpha_alpha.copy(), m._eta.copy()])
    a, b = traces
    assert a[:3] == b[:3]
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])


# ---- world size 2 on ONE GPU: the real exchange (allreduce_sstats on the library's device buffer,
# ordered on the context's stream) inside VariationalBayes(process_group=...).learning() ----
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _two_rank_worker(rank, world, port, out_dir, backend, one_gpu_each=False):
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from pylda_amd import corpus as C
    from pylda_amd.variational_bayes import VariationalBayes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    device = rank if one_gpu_each else 0
    torch.cuda.set_device(device)
    if backend == "nccl":
        dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "ap_train_k10.npz"))
    ptr = g["doc_ptr"][:401].astype(np.int64)
    ids, cts = g["term_id"][:ptr[-1]].astype(np.int32), g["term_ct"][:ptr[-1]].astype(np.int32)
    sp, si, sc, (lo, hi) = C.shard_csr(ptr, ids, cts, world, rank)
    m = VariationalBayes(process_group=dist.group.WORLD, device=device)
    m._verbose = False
    m._initialize_parsed(sp, si, sc, 6806, 10, 0.1, 1.0 / 6806, eta=g["eta"].copy())
    assert getattr(m._context(), "_torch_stream", None) is not None       # runs on a stream torch knows
    trace = []
    for _ in range(3):
        # a sentinel ahead of every iteration: ~25 ms of device time on the context's stream, so that the E-step's
        # kernels START late.  A collective issued on any other stream would run while they are still queued and sum
        # the statistics of the iteration before (zeros, the first time): the trajectory below would not match.
        with torch.cuda.stream(m._context()._torch_stream):
            torch.cuda._sleep(50_000_000)
        trace.append(m.learning())
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), trace=np.array(trace), alpha=m._alpha_alpha,
             eta=m._eta, gamma=m._gamma, lo=lo, hi=hi)
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_single_process(ap_train, tmp_path):
    """Two processes share GPU 0 (gloo; RCCL refuses two ranks on one device): each runs the E-step on
    its shard, the sufficient statistics are all-reduced between e_step and m_step on the context's
    stream, and the 3-iteration trajectory equals the unsharded one."""
    import torch.multiprocessing as mp
    from pylda_amd.variational_bayes import VariationalBayes
    g = ap_train
    ptr = g["doc_ptr"][:401]
    ids, cts = g["term_id"][:ptr[-1]], g["term_ct"][:ptr[-1]]
    m = VariationalBayes()
    m._verbose = False
    m._initialize_parsed(ptr, ids, cts, 6806, 10, 0.1, 1.0 / 6806, eta=g["eta"].copy())
    single = [m.learning() for _ in range(3)]
    alpha, eta, gamma = m._alpha_alpha.copy(), m._eta.copy(), m._gamma.copy()
    mp.spawn(_two_rank_worker, args=(2, _free_port(), str(tmp_path), "gloo"), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert r0["lo"] == 0 and r0["hi"] == r1["lo"] and r1["hi"] == 400
    for r in (r0, r1):
        assert rel_err(r["trace"], np.array(single)) < 1e-12
        assert rel_err(r["alpha"], alpha) < 1e-11
        assert rel_err(r["eta"], eta) < 1e-11
    assert np.array_equal(r0["eta"], r1["eta"]) and np.array_equal(r0["alpha"], r1["alpha"])
    assert rel_err(np.concatenate([r0["gamma"], r1["gamma"]]), gamma) < 1e-10


def _single_process_reference(ap_train):
    from pylda_amd.variational_bayes import VariationalBayes
    g = ap_train
    ptr = g["doc_ptr"][:401]
    m = VariationalBayes()
    m._verbose = False
    m._initialize_parsed(ptr, g["term_id"][:ptr[-1]], g["term_ct"][:ptr[-1]], 6806, 10, 0.1, 1.0 / 6806, eta=g["eta"].copy())
    single = [m.learning() for _ in range(3)]
    return single, m._alpha_alpha.copy(), m._eta.copy(), m._gamma.copy()


def _gpu_count():
    try:
        from pylda_amd import _capi
        return _capi.device_count()
    except Exception:
        return 0


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_two_gpus_over_rccl_match_single_process(ap_train, tmp_path):
    """The real thing, for a box with >= 2 GPUs: one rank per GPU, backend nccl (= RCCL over xGMI), zero-copy
    all-reduce of the library's statistics buffer and of the packed outer-iteration values on the context's stream.
    Sharded 3-iteration trajectory == the single-process one."""
    import torch.multiprocessing as mp
    single, alpha, eta, gamma = _single_process_reference(ap_train)
    mp.spawn(_two_rank_worker, args=(2, _free_port(), str(tmp_path), "nccl", True), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for r in (r0, r1):
        assert rel_err(r["trace"], np.array(single)) < 1e-12
        assert rel_err(r["alpha"], alpha) < 1e-11 and rel_err(r["eta"], eta) < 1e-11
    assert np.array_equal(r0["eta"], r1["eta"]) and np.array_equal(r0["alpha"], r1["alpha"])
    assert rel_err(np.concatenate([r0["gamma"], r1["gamma"]]), gamma) < 1e-10


def _c_abi_two_rank_worker(rank, world, uid_path, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    from pylda_amd import _capi
    from pylda_amd import corpus as C
    g = np.load(os.path.join(ROOT, "tests", "golden", "ap_train_k10.npz"))
    ptr = g["doc_ptr"][:401].astype(np.int64)
    ids, cts = g["term_id"][:ptr[-1]].astype(np.int32), g["term_ct"][:ptr[-1]].astype(np.int32)
    sp, si, sc, _ = C.shard_csr(ptr, ids, cts, world, rank)
    ctx = _capi.Context(10, 6806, device=rank)
    import time
    if rank == 0:
        open(uid_path + ".tmp", "wb").write(_capi.comm_unique_id())
        os.replace(uid_path + ".tmp", uid_path)
    while not os.path.exists(uid_path):
        time.sleep(0.05)
    ctx.comm_init(open(uid_path, "rb").read(), rank, world)
    corpus = ctx.corpus(sp, si, sc)
    ctx.set_option("doc_values", 0)
    ctx.set_alpha(g["alpha"])
    ctx.set_eta(g["eta"])
    ctx.estep(corpus)
    ctx.allreduce_sstats()
    ctx.mstep_enqueue(corpus, g["beta"])
    ctx.allreduce_outer()
    ll, nd, _, tll, ass, _ = ctx.outer_fetch()
    np.savez(os.path.join(out_dir, "c%d.npz" % rank), ll=ll, nd=nd, tll=tll, ass=ass, eta=ctx.get_eta())
    ctx.comm_destroy()
    corpus.close()
    ctx.close()


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_two_gpus_through_the_c_abi_communicator(ap_train, tmp_path):
    """The same exchange without torch: pylda_comm_init / pylda_allreduce_sstats / pylda_allreduce_outer (RCCL
    bound at run time) on two GPUs give the single-process outer iteration."""
    import torch.multiprocessing as mp
    from pylda_amd import _capi
    g = ap_train
    ptr = g["doc_ptr"][:401]
    ctx = _capi.Context(10, 6806)
    corpus = ctx.corpus(ptr, g["term_id"][:ptr[-1]], g["term_ct"][:ptr[-1]])
    ctx.set_option("doc_values", 0)
    ctx.set_alpha(g["alpha"])
    ctx.set_eta(g["eta"])
    ctx.estep(corpus)
    ctx.mstep_enqueue(corpus, g["beta"])
    ll, nd, _, tll, ass, _ = ctx.outer_fetch()
    eta = ctx.get_eta()
    corpus.close()
    ctx.close()
    mp.spawn(_c_abi_two_rank_worker, args=(2, str(tmp_path / "uid"), str(tmp_path)), nprocs=2, join=True)
    for r in (np.load(tmp_path / "c0.npz"), np.load(tmp_path / "c1.npz")):
        assert int(r["nd"]) == nd == 400
        assert abs(float(r["ll"]) - ll) < 1e-12 * abs(ll) and abs(float(r["tll"]) - tll) < 1e-12 * abs(tll)
        assert rel_err(r["ass"], ass) < 1e-12 and rel_err(r["eta"], eta) < 1e-12


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` invoked directly (no launcher, as the driver does for N = 1) becomes the
    torch.distributed.run launcher; --share-gpu puts both ranks on GPU 0 with gloo so the whole N > 1 bench path
    (sharded corpus generation, all-reduce inside learning(), max-over-ranks timing, rank-0 JSON) runs on this box."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--docs", "4000",
                          "--extra-docs", "3000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line, from rank 0
    rec = json.loads(lines[0])
    # top level: cfg 4, the corpus of the scaling curve, sharded over the ranks (strong scaling)
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "strong"
    assert rec["config"]["docs_total"] == 4000 and rec["config"]["docs_per_gpu"] == 2000 and rec["config"]["K"] == 256
    assert rec["value"] > 0 and rec["roofline"]["kernel_ms_documents"] > 0 and rec["roofline"]["kernel_ms_sstats"] > 0
    assert abs(rec["value"] - 4000 / (rec["ms_per_step"] * 1e-3)) < 1e-6 * rec["value"]
    assert sum(c["documents"] for c in rec["roofline"]["launch_classes"]) == rec["config"]["docs_per_gpu"]
    sub = rec["synth100k"]
    assert sub["n_gpus"] == 2 and sub["scaling"] == "weak" and sub["config"]["docs_total"] == 6000
    assert sum(c["documents"] for c in sub["roofline"]["launch_classes"]) == sub["config"]["docs_per_gpu"] == 3000
    # cfg 2 and cfg 5 ride along at every N, documents sharded over the ranks; the K = 500 trace is asserted inside
    assert rec["ap_k10"]["n_gpus"] == 2 and rec["ap_k10"]["iters_equal_fraction"] == 1.0
    assert rec["ap_k10"]["max_rel_ll_delta_vs_reference"] < 1e-9
    nips = rec["nips_k500"]
    assert nips["n_gpus"] == 2 and nips["iterations"] == 50 and nips["joint_trace_max_rel_delta"] < 1e-8
    assert nips["heldout_rel_delta"] < 1e-8


def test_bench_rehearsal_of_the_eight_rank_run(tmp_path):
    """The command the driver's 8-GPU run executes, end to end on this box's one GPU (`--share-gpu`: gloo instead of
    RCCL, everything else as at N = 8): eight chunk-aligned shards of the cfg 4 generator, eight shards of cfg 3,
    associated-press in eight nnz-balanced ranges of ~250 documents, nips.88-05 K = 500 in eight ranges of ~280 with
    the reference's 50-iteration trace asserted, max-over-ranks timing and the checksum reductions."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-gpu", "--docs", "200000",
                          "--extra-docs", "5000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=1500, env=env, cwd=str(tmp_path))
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["scaling"] == "strong" and rec["config"]["parallelism"] == "dp8"
    assert rec["config"]["docs_total"] == 200000 and rec["config"]["docs_per_gpu"] == 25000      # whole 25k-document chunks
    assert abs(rec["config"]["nnz_imbalance"]) < 0.01
    assert abs(rec["value"] - 200000 / (rec["ms_per_step"] * 1e-3)) < 1e-6 * rec["value"]
    assert "shard_proxy" not in rec                                  # (a single-GPU model: N = 1 only)
    sub = rec["synth100k"]
    assert sub["n_gpus"] == 8 and sub["config"]["docs_total"] == 40000 and sub["config"]["docs_per_gpu"] == 5000
    assert rec["ap_k10"]["n_gpus"] == 8 and rec["ap_k10"]["iters_equal_fraction"] == 1.0
    assert rec["ap_k10"]["max_rel_ll_delta_vs_reference"] < 1e-9
    nips = rec["nips_k500"]
    assert nips["n_gpus"] == 8 and nips["iterations"] == 50 and nips["joint_trace_max_rel_delta"] < 1e-8
    assert 270 <= nips["config"]["docs_rank0"] <= 290 and nips["heldout_rel_delta"] < 1e-8


def test_c_abi_allreduce_world_of_one(ap_train):
    """The multi-GPU entry points of the C ABI (RCCL bound at run time, no torch.distributed): a one-rank
    communicator reduces the library's own device buffer in place - an identity - on the context's stream,
    between E-step and M-step, and the short host vector likewise; call-order errors are reported."""
    from pylda_amd import _capi
    g = ap_train
    ptr = g["doc_ptr"][:301]
    ctx = _capi.Context(10, 6806)
    corpus = ctx.corpus(ptr, g["term_id"][:ptr[-1]], g["term_ct"][:ptr[-1]])
    ctx.set_alpha(g["alpha"])
    ctx.set_eta(g["eta"])
    with pytest.raises(_capi.PyldaError) as e:
        ctx.allreduce_sstats()                              # no communicator yet
    assert e.value.status == -4
    uid = _capi.comm_unique_id()
    assert len(uid) == 128
    ctx.comm_init(uid, 0, 1)
    with pytest.raises(_capi.PyldaError):
        ctx.allreduce_sstats()                              # no training E-step yet
    ctx.estep(corpus)
    before = ctx.get_sstats()
    ctx.allreduce_sstats()
    assert np.array_equal(ctx.get_sstats(), before)
    ll = ctx.estep_results(corpus)[0]
    vec = ctx.allreduce_doubles(np.concatenate([[ll, 300.0], np.arange(10.0)]))
    assert vec[0] == ll and vec[1] == 300.0 and np.array_equal(vec[2:], np.arange(10.0))
    tll, ass = ctx.mstep(corpus, g["beta"])                 # the M-step after the exchange, same stream
    assert np.isfinite(tll) and ass.shape == (10,)
    with pytest.raises(_capi.PyldaError):
        ctx.comm_init(uid, 0, 1)                            # one communicator per context
    ctx.comm_destroy()
    corpus.close()
    ctx.close()


# ---- the drop-in driver and cfg 5 over several ranks (all ranks on GPU 0, exchange over gloo) ----
def _nips_worker(rank, world, port, out_dir, iterations):
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from pylda_amd import corpus as C
    from pylda_amd.variational_bayes import VariationalBayes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "nips_trace_k500.npz"))
    K, V = int(g["K"]), len(g["words"])
    sp, si, sc, _ = C.shard_csr(g["doc_ptr"].astype(np.int64), g["term_id"].astype(np.int32), g["term_ct"].astype(np.int32),
                                world, rank)
    np.random.seed(int(g["seed"]))                   # every rank draws the reference's eta (:95)
    m = VariationalBayes(process_group=dist.group.WORLD, device=0)
    m._verbose = False
    m._initialize_parsed(sp, si, sc, V, K, 1.0 / K, 1.0 / V)
    trace = [m.learning() for _ in range(iterations)]
    np.savez(os.path.join(out_dir, "nips%d.npz" % rank), trace=np.array(trace), alpha=m._alpha_alpha)
    dist.destroy_process_group()


def test_two_rank_nips_k500_trace_equals_the_reference(tmp_path):
    """BASELINE cfg 5 over two ranks: the first five outer iterations at K = 500 (fused streaming kernel, one
    all-reduce of the 3209 x 512 statistics + one of the packed outer-iteration values per iteration) reproduce
    the reference's own joint log-likelihood trace."""
    import torch.multiprocessing as mp
    g = np.load(os.path.join(ROOT, "tests", "golden", "nips_trace_k500.npz"))
    mp.spawn(_nips_worker, args=(2, _free_port(), str(tmp_path), 5), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "nips0.npz"), np.load(tmp_path / "nips1.npz")
    assert np.array_equal(r0["trace"], r1["trace"]) and np.array_equal(r0["alpha"], r1["alpha"])
    assert rel_err(r0["trace"], g["joint_ll"][:5]) < 1e-9


def _write_mini_press(ap_train, tmp_path, n_docs=150):
    from test_gpu_variational_bayes import documents_from_csr
    g = ap_train
    words = [str(w) for w in g["words"]]
    corpus_dir = tmp_path / "mini-press"
    corpus_dir.mkdir()
    docs = documents_from_csr(words, g["doc_ptr"][:n_docs + 1], g["term_id"], g["term_ct"])
    (corpus_dir / "train.dat").write_text("\n".join(docs) + "\n")
    (corpus_dir / "voc.dat").write_text("".join("%s\t1\t1\n" % w for w in words))
    return corpus_dir


def test_launch_train_over_two_ranks_writes_the_same_files(ap_train, tmp_path):
    """`launch_train --gpus 2` (the reference's command line, launch_train.py:187-204, re-executing itself under
    torch.distributed.run; --share_gpu 1 puts both ranks on this box's one GPU): the run directory of rank 0 holds
    byte-identical exp_beta-N / exp_gamma-N files and the same model as the one-process run from the same seed."""
    import pickle
    import subprocess
    import sys
    corpus_dir = _write_mini_press(ap_train, tmp_path)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["PYLDA_SEED"] = "11"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    runs = {}
    for gpus in (1, 2):
        out_dir = tmp_path / ("out%d" % gpus)
        cmd = [sys.executable, "-m", "pylda_amd.launch_train", "--input_directory=%s/" % corpus_dir,
               "--output_directory=%s" % out_dir, "--number_of_topics=5", "--training_iterations=4", "--snapshot_interval=2"]
        if gpus > 1:
            cmd += ["--gpus=%d" % gpus, "--share_gpu=1"]
        done = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
        assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-4000:]
        found = list((out_dir / "mini-press").iterdir())
        assert len(found) == 1                                   # ONE run directory, rank 0's
        runs[gpus] = found[0]
        assert done.stdout.count("successfully load all training docs") == 1
    names = sorted(p.name for p in runs[1].iterdir())
    assert names == ["exp_beta-2", "exp_beta-4", "exp_gamma-2", "exp_gamma-4", "model-4", "option.txt"]
    assert sorted(p.name for p in runs[2].iterdir()) == names
    for name in ("exp_gamma-2", "exp_gamma-4"):
        assert (runs[1] / name).read_bytes() == (runs[2] / name).read_bytes(), name
    for name in ("exp_beta-2", "exp_beta-4"):
        # the same lines topic by topic; the ORDER of words whose probabilities tie to all six printed digits (most of
        # the 6806 types never occur in 150 documents) follows 1e-16 differences of the summation order over ranks
        a, b = ((runs[k] / name).read_text().split("==========\t") for k in (1, 2))
        assert len(a) == len(b) == 1 + 5
        for block_a, block_b in zip(a, b):
            assert sorted(block_a.splitlines()) == sorted(block_b.splitlines()), name
            probs = [float(l.split("\t")[1]) for l in block_b.splitlines()[1:]]
            assert probs == sorted(probs, reverse=True)
    one, two = (pickle.load(open(runs[k] / "model-4", "rb")) for k in (1, 2))
    assert two._number_of_documents == 150 and two._gamma.shape == (150, 5) and two._counter == 4
    assert rel_err(two._eta, one._eta) < 1e-12 and rel_err(two._gamma, one._gamma) < 1e-10
    assert rel_err(two._alpha_alpha, one._alpha_alpha) < 1e-12
    words_ll, gamma = two.inference(["%s %s" % (ap_train["words"][3], ap_train["words"][5])])     # the snapshot works
    assert np.isfinite(words_ll) and gamma.shape == (1, 5)


def test_launch_train_with_more_ranks_than_documents(ap_train, tmp_path):
    """Three ranks, two documents (and an empty line between them): every rank parses ITS line range only, so one rank
    holds no document at all - it still joins both all-reduces of every iteration, and rank 0 gathers gamma and the
    corpus from shards of 1, 0 and 1 documents."""
    import pickle
    import subprocess
    import sys
    corpus_dir = _write_mini_press(ap_train, tmp_path, n_docs=2)
    lines = (corpus_dir / "train.dat").read_text().splitlines()
    (corpus_dir / "train.dat").write_text(lines[0] + "\n\n" + lines[1] + "\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["PYLDA_SEED"] = "5"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    runs = {}
    for gpus in (1, 3):
        out_dir = tmp_path / ("out%d" % gpus)
        cmd = [sys.executable, "-m", "pylda_amd.launch_train", "--input_directory=%s/" % corpus_dir,
               "--output_directory=%s" % out_dir, "--number_of_topics=4", "--training_iterations=3", "--snapshot_interval=3"]
        if gpus > 1:
            cmd += ["--gpus=%d" % gpus, "--share_gpu=1"]
        done = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
        assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-4000:]
        runs[gpus] = next((out_dir / "mini-press").iterdir())
    assert (runs[1] / "exp_gamma-3").read_bytes() == (runs[3] / "exp_gamma-3").read_bytes()
    one, three = (pickle.load(open(runs[k] / "model-3", "rb")) for k in (1, 3))
    assert three._number_of_documents == 2 and three._gamma.shape == (2, 4)
    assert rel_err(three._eta, one._eta) < 1e-12 and rel_err(three._gamma, one._gamma) < 1e-10
    assert np.array_equal(three._train_csr[0], one._train_csr[0]) and np.array_equal(three._train_csr[1], one._train_csr[1])


def test_collectives_are_issued_under_the_context_stream(nccl_group, ap_train, monkeypatch):
    """What orders E-step -> all-reduce -> M-step on the device is that BOTH collectives of an outer iteration are
    issued while torch's current stream is the stream the library's kernels run on.  At world size 1 the reduction
    is an identity, so the data cannot tell; this fails if either all-reduce is issued under any other stream."""
    import torch
    import torch.distributed as dist
    from pylda_amd.variational_bayes import VariationalBayes
    g = ap_train
    ptr = g["doc_ptr"][:301]
    m = VariationalBayes(process_group=nccl_group)
    m._verbose = False
    m._initialize_parsed(ptr, g["term_id"][:ptr[-1]], g["term_ct"][:ptr[-1]], 6806, 10, 0.1, 1.0 / 6806, eta=g["eta"].copy())
    ctx = m._context()
    mine = ctx._torch_stream.cuda_stream
    assert mine != torch.cuda.default_stream().cuda_stream
    seen = []
    real = dist.all_reduce

    def spy(tensor, *args, **kwargs):
        seen.append((torch.cuda.current_stream().cuda_stream, tensor.data_ptr(), tensor.numel()))
        return real(tensor, *args, **kwargs)

    monkeypatch.setattr(dist, "all_reduce", spy)
    m.learning()
    assert [s for s, _, _ in seen] == [mine, mine], seen
    assert seen[0][1] == ctx.sstats_device_ptr() and seen[0][2] == ctx.sstats_elements()     # zero-copy, in place
    assert seen[1][1] == ctx.outer_device()[0] and seen[1][2] == 10 + 4
