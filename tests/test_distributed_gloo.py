"""N > 1 path on CPU: world_size-2 gloo run of the sharding + all-reduce
plumbing.  The per-shard E-step stands in as the oracle (this test checks the
exchange, not the kernel): sharded E-steps + all-reduce(sum) of the sufficient
statistics and of (LL, D, alpha statistics) must equal the unsharded run."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from oracle import c_oracle, vb_numpy
    from pylda_amd import corpus as C
    from pylda_amd import distributed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "ap_train_k10.npz"))
    ptr, ids, cts = g["doc_ptr"][:401].astype(np.int64), g["term_id"].astype(np.int32), g["term_ct"].astype(np.int32)
    ids, cts = ids[:ptr[-1]], cts[:ptr[-1]]
    sp, si, sc, (lo, hi) = C.shard_csr(ptr, ids, cts, world, rank)
    out = c_oracle.e_step(g["alpha"], g["eta"], sp, si, sc)
    sstats = out["sstats"].copy()
    distributed.allreduce_array_(sstats)
    gamma = out["gamma"]
    alpha_ss = np.sum(vb_numpy.psi(gamma) - vb_numpy.psi(gamma.sum(axis=1))[:, None], axis=0)
    ll, D, ass = distributed.allreduce_small(None, out["document_log_likelihood"], hi - lo, alpha_ss)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), sstats=sstats, ll=ll, D=D, alpha_ss=ass, lo=lo, hi=hi)
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle import c_oracle, vb_numpy
    g = np.load(os.path.join(ROOT, "tests", "golden", "ap_train_k10.npz"))
    ptr = g["doc_ptr"][:401].astype(np.int64)
    ids, cts = g["term_id"][:ptr[-1]].astype(np.int32), g["term_ct"][:ptr[-1]].astype(np.int32)
    full = c_oracle.e_step(g["alpha"], g["eta"], ptr, ids, cts)
    alpha_ss = np.sum(vb_numpy.psi(full["gamma"]) - vb_numpy.psi(full["gamma"].sum(axis=1))[:, None], axis=0)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert r0["lo"] == 0 and r0["hi"] == r1["lo"] and r1["hi"] == 400
    for r in (r0, r1):
        assert np.max(np.abs(r["sstats"] - full["sstats"])) < 1e-10
        assert abs(float(r["ll"]) - full["document_log_likelihood"]) < 1e-9 * abs(full["document_log_likelihood"])
        assert int(r["D"]) == 400
        assert np.max(np.abs(r["alpha_ss"] - alpha_ss)) < 1e-9
    assert np.array_equal(r0["sstats"], r1["sstats"])      # every rank holds the identical reduced buffer


def _gather_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pylda_amd import cli
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(rank)
    rows = [5, 0, 3][rank]                                   # (a rank without documents sends nothing)
    gamma = rng.random((rows, 4)) + rank
    whole = cli._gather_rows(gamma, rank, world)
    ids = cli._gather_rows(np.arange(rows * 2, dtype=np.int32) + 100 * rank, rank, world)
    np.save(os.path.join(out_dir, "gamma%d.npy" % rank), gamma)
    if rank == 0:
        np.save(os.path.join(out_dir, "whole.npy"), whole)
        np.save(os.path.join(out_dir, "ids.npy"), ids)
    else:
        assert whole is None and ids is None
    dist.destroy_process_group()


def test_rank_zero_gathers_rows_without_pickling(tmp_path):
    """launch_train --gpus N: gamma and the corpus shards reach rank 0 as tensor receives into slices of ONE
    pre-allocated array (gather_object pickled 2 GB at cfg 4), in rank order, empty shards included."""
    import torch.multiprocessing as mp
    world = 3
    mp.spawn(_gather_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / ("gamma%d.npy" % r)) for r in range(world)]
    assert np.array_equal(np.load(tmp_path / "whole.npy"), np.concatenate(parts, axis=0))
    assert np.array_equal(np.load(tmp_path / "ids.npy"),
                          np.concatenate([np.arange(10, dtype=np.int32), np.arange(6, dtype=np.int32) + 200]))


def test_line_ranges_balance_tokens_and_cover():
    """launch_train --gpus N: contiguous line ranges balanced by TOKENS (the proxy for distinct terms known before parsing;
    ADVICE r5: bytes also weigh long words)."""
    sys.path.insert(0, ROOT)
    from pylda_amd import cli
    rng = np.random.default_rng(0)
    docs = [" ".join("w" * int(rng.integers(1, 12)) for _ in range(int(n))) for n in rng.integers(1, 400, 1000)]
    for world in (1, 2, 3, 8):
        b = cli._line_ranges(docs, world)
        assert b[0] == 0 and b[-1] == 1000 and len(b) == world + 1 and all(x <= y for x, y in zip(b, b[1:]))
        weights = [sum(d.count(" ") + 1 for d in docs[lo:hi]) for lo, hi in zip(b, b[1:])]
        assert max(weights) - min(weights) <= 2 * 401                         # within a line or two of each other
    assert cli._line_ranges([], 4) == [0, 0, 0, 0, 0]
    assert cli._line_ranges(["a", "b"], 8)[-1] == 2                           # more ranks than lines: empty shards


def test_launch_train_refuses_a_world_it_was_not_asked_for(monkeypatch):
    """`--gpus 4` under a launcher with WORLD_SIZE=1 ran as one rank without a word; WORLD_SIZE without RANK
    raised a KeyError."""
    sys.path.insert(0, ROOT)
    from pylda_amd import cli
    opt = cli._parse(cli.TRAIN_FLAGS, ["--gpus=4"], "launch_train")
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(SystemExit, match="does not match"):
        cli._join_ranks(opt)
    opt = cli._parse(cli.TRAIN_FLAGS, ["--gpus=2"], "launch_train")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.delenv("RANK", raising=False)
    with pytest.raises(SystemExit, match="RANK is not"):
        cli._join_ranks(opt)
    monkeypatch.delenv("WORLD_SIZE")
    opt = cli._parse(cli.TRAIN_FLAGS, [], "launch_train")
    assert cli._join_ranks(opt) == (0, 1, 0, None)


def _init_shard_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pylda_amd import cli
    from pylda_amd.variational_bayes import VariationalBayes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vocabulary = ["w%d" % i for i in range(30)]
    rng = np.random.default_rng(7)
    documents = [" ".join("w%d" % t for t in rng.integers(0, 34, rng.integers(0, 25))) for _ in range(41)]   # (some empty, some OOV tokens)
    np.random.seed(100 + rank)                       # only rank 0's stream may matter
    engine = VariationalBayes(device=0, process_group=None)
    engine._verbose = False
    cli._initialize_shard(engine, documents, vocabulary, 4, 0.25, 1.0 / 30, rank, world)
    ptr, ids, cts = engine._train_csr
    np.savez(os.path.join(out_dir, "shard%d.npz" % rank), ptr=ptr, ids=ids, cts=cts, eta=engine._eta, D=engine._number_of_documents,
             bounds=np.array(cli._line_ranges(documents, world)))
    dist.destroy_process_group()


def test_ranks_parse_their_own_lines_and_share_rank_zeros_eta(tmp_path):
    """launch_train --gpus N start-up on CPU: every rank parses its byte-balanced line range only, the shards put
    together are the one-process parse, and eta is rank 0's seeded draw (variational_bayes.py:95) on every rank."""
    import torch.multiprocessing as mp
    world = 3
    mp.spawn(_init_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    shards = [np.load(tmp_path / ("shard%d.npz" % r)) for r in range(world)]
    sys.path.insert(0, ROOT)
    from pylda_amd import _capi
    vocabulary = ["w%d" % i for i in range(30)]
    rng = np.random.default_rng(7)
    documents = [" ".join("w%d" % t for t in rng.integers(0, 34, rng.integers(0, 25))) for _ in range(41)]
    ptr, ids, cts, dropped = _capi.parse_corpus(documents, vocabulary)
    assert sum(int(s["D"]) for s in shards) == len(ptr) - 1
    assert np.array_equal(np.concatenate([s["ids"] for s in shards]), ids)
    assert np.array_equal(np.concatenate([s["cts"] for s in shards]), cts)
    assert np.array_equal(np.concatenate([np.diff(s["ptr"]) for s in shards]), np.diff(ptr))
    np.random.seed(100)
    eta0 = np.random.gamma(100., 1. / 100., (4, 30))
    for s in shards:
        assert np.array_equal(s["eta"], eta0)
        assert np.array_equal(s["bounds"], shards[0]["bounds"])
