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


def _two_rank_worker(rank, world, port, out_dir, backend):
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from pylda_amd import corpus as C
    from pylda_amd.variational_bayes import VariationalBayes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "ap_train_k10.npz"))
    ptr = g["doc_ptr"][:401].astype(np.int64)
    ids, cts = g["term_id"][:ptr[-1]].astype(np.int32), g["term_ct"][:ptr[-1]].astype(np.int32)
    sp, si, sc, (lo, hi) = C.shard_csr(ptr, ids, cts, world, rank)
    m = VariationalBayes(process_group=dist.group.WORLD, device=0)
    m._verbose = False
    m._initialize_parsed(sp, si, sc, 6806, 10, 0.1, 1.0 / 6806, eta=g["eta"].copy())
    assert getattr(m._context(), "_torch_stream", None) is not None       # runs on a stream torch knows
    trace = [m.learning() for _ in range(3)]
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


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` invoked directly (no launcher, as the driver does for N = 1) becomes the
    torch.distributed.run launcher; --share-gpu puts both ranks on GPU 0 with gloo so the whole N > 1 bench path
    (sharded corpus generation, all-reduce inside learning(), max-over-ranks timing, rank-0 JSON) runs on this box."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--docs", "3000",
                          "--extra-docs", "4000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "weak"
    assert rec["config"]["docs_total"] == 6000 and rec["config"]["docs_per_gpu"] == 3000
    assert rec["value"] > 0 and rec["roofline"]["kernel_ms_documents"] > 0 and rec["roofline"]["kernel_ms_sstats"] > 0
    assert abs(rec["value"] - 6000 / (rec["ms_per_step"] * 1e-3)) < 1e-6 * rec["value"]
    sub = rec["synth1m"]
    assert sub["n_gpus"] == 2 and sub["scaling"] == "strong" and sub["config"]["docs_total"] == 4000
    assert sum(c["documents"] for c in sub["roofline"]["launch_classes"]) == sub["config"]["docs_per_gpu"]


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
