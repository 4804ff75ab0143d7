"""Command-line drivers that speak the reference's launch_train / launch_test
interface (flags, output directory layout, snapshot files) on top of the
MI355X variational-Bayes engine.

Behavioural contract taken from the reference:
  training   launch_train.py:31-62 (flags), :102-115 (corpus files), :119-124 (default
             priors), :127-141 (run directory name), :148-162 (option.txt), :196-204
             (snapshot cadence, exp_beta-N / exp_gamma-N / model-N)
  held-out   launch_test.py:18-23 (flags), :49-57 (corpus-name check), :62-66 (test.dat),
             :90-97 (per-snapshot evaluation, test-N via numpy.savetxt)
Only inference mode 2 (variational Bayes) exists in this package; modes 0 / 1 (hybrid,
Monte-Carlo) are outside its scope and are refused instead of silently substituted.
"""
import argparse
import datetime
import os
import pickle
import sys

import numpy

TRAIN_FLAGS = (
    # name, type, default, help
    ("input_directory", str, None, "input directory [None]"),
    ("output_directory", str, None, "output directory [None]"),
    ("number_of_topics", int, -1, "total number of topics [-1]"),
    ("training_iterations", int, -1, "total number of iterations [-1]"),
    ("snapshot_interval", int, 10, "snapshot interval [10]"),
    ("alpha_alpha", float, -1, "hyper-parameter for Dirichlet distribution of topics [1.0/number_of_topics]"),
    ("alpha_beta", float, -1, "hyper-parameter for Dirichlet distribution of vocabulary [1.0/number_of_types]"),
    ("inference_mode", int, 2, "inference mode [2: variational bayes - the only engine here]"),
    ("device", int, 0, "GPU index [0] (one process; with --gpus N rank r runs on GPU r)"),
    ("gpus", int, 1, "GPUs of this node to shard the documents over [1]: re-executes itself under "
                     "torch.distributed.run, one rank per GPU, one RCCL all-reduce of the K x V statistics per iteration"),
    ("share_gpu", int, 0, "test mode [0]: 1 = all ranks on GPU 0, exchange over gloo (RCCL refuses two ranks per device)"),
)
TEST_FLAGS = (
    ("input_directory", str, None, "input directory [None]"),
    ("model_directory", str, None, "model directory [None]"),
    ("snapshot_index", int, -1, "snapshot index [-: evaluate on all available snapshots]"),
)
RULE = "========== ========== ========== ========== =========="


def _parse(flags, argv, prog):
    parser = argparse.ArgumentParser(prog=prog, allow_abbrev=False)
    for name, kind, default, text in flags:
        parser.add_argument("--" + name, type=kind, default=default, help=text)
    return parser.parse_args(argv)


def _lines(path):
    with open(path, "r") as stream:
        return [line.strip().lower() for line in stream]


def _phase(label, since):
    """PYLDA_TIMING=1: wall time of the start-up phases on stderr (every rank)."""
    import time
    now = time.perf_counter()
    if os.environ.get("PYLDA_TIMING"):
        sys.stderr.write("[pylda timing] rank %s %-40s %8.1f ms\n" % (os.environ.get("RANK", "0"), label, (now - since) * 1e3))
    return now


def _banner(pairs):
    print(RULE)
    for key, value in pairs:
        print("%s=%s" % (key, value))
    print(RULE)


def train_main(argv=None):
    opt = _parse(TRAIN_FLAGS, argv, "launch_train")
    for required in ("number_of_topics", "training_iterations", "snapshot_interval"):
        if getattr(opt, required) <= 0:
            raise SystemExit("--%s must be positive" % required)
    if opt.input_directory is None or opt.output_directory is None:
        raise SystemExit("--input_directory and --output_directory are required")
    if opt.inference_mode != 2:
        sys.stderr.write("error: pylda_amd implements inference mode 2 (variational bayes) only, got %d...\n"
                         % opt.inference_mode)
        return 2
    if opt.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # invoked as the reference's one-process command: become the launcher, one rank per GPU
        os.execv(sys.executable, _launcher_argv(opt.gpus, sys.argv[1:] if argv is None else list(argv)))
    rank, world, device, group = _join_ranks(opt)
    seed = os.environ.get("PYLDA_SEED")      # the reference has no seed flag (numpy's global RNG, unseeded): reproducible runs
    if rank != 0:
        sys.stdout = open(os.devnull, "w")    # one copy of the reference's console output
    source = opt.input_directory.rstrip("/")
    corpus_name = os.path.basename(source)
    documents = _lines(os.path.join(source, "train.dat"))
    print("successfully load all training docs from %s..." % os.path.abspath(os.path.join(source, "train.dat")))
    vocabulary = list(dict.fromkeys(entry.split()[0] for entry in _lines(os.path.join(source, "voc.dat")) if entry))
    print("successfully load all the words from %s..." % os.path.abspath(os.path.join(source, "voc.dat")))
    topics = opt.number_of_topics
    prior_topics = opt.alpha_alpha if opt.alpha_alpha > 0 else 1.0 / topics
    prior_words = opt.alpha_beta if opt.alpha_beta > 0 else 1.0 / len(vocabulary)

    stamp = datetime.datetime.now().strftime("%y%m%d-%H%M%S")
    run = "%s-lda-I%d-S%d-K%d-aa%f-ab%f-im%d/" % (stamp, opt.training_iterations, opt.snapshot_interval,
                                                  topics, prior_topics, prior_words, opt.inference_mode)
    run_dir = os.path.join(opt.output_directory, corpus_name, run)
    if rank == 0:
        os.makedirs(os.path.abspath(run_dir))
    settings = (("input_directory", source), ("corpus_name", corpus_name),
                ("training_iterations", "%d" % opt.training_iterations),
                ("snapshot_interval", str(opt.snapshot_interval)), ("number_of_topics", str(topics)),
                ("alpha_alpha", str(prior_topics)), ("alpha_beta", str(prior_words)),
                ("inference_mode", "%d" % opt.inference_mode))
    if rank == 0:
        with open(run_dir + "option.txt", "w") as out:
            out.writelines("%s=%s\n" % pair for pair in settings)
    _banner((("output_directory", run_dir),) + settings[:1] + settings[1:])

    from pylda_amd.variational_bayes import VariationalBayes
    import time
    engine = VariationalBayes(device=device, process_group=group)
    if seed is not None:
        numpy.random.seed(int(seed))
    started = time.perf_counter()
    if group is None:
        engine._initialize(documents, vocabulary, topics, prior_topics, prior_words)
    else:
        _initialize_shard(engine, documents, vocabulary, topics, prior_topics, prior_words, rank, world)
    _phase("parse + initial eta", started)
    whole, whole_at = None, -1
    for _ in range(opt.training_iterations):
        engine.learning()
        if engine._counter % opt.snapshot_interval == 0:
            whole, whole_at = _whole_model(engine, group, rank, world), engine._counter
            if rank == 0:
                whole.export_beta("%sexp_beta-%d" % (run_dir, engine._counter))
                whole.export_gamma("%sexp_gamma-%d" % (run_dir, engine._counter))
    # (the last iteration's exports already gathered gamma - 2 GB at cfg 4: the snapshot adds the corpus to that copy)
    whole = _whole_model(engine, group, rank, world, with_corpus=True, reuse_gamma=whole_at == engine._counter, gathered=whole)
    if rank == 0:
        with open(os.path.join(run_dir, "model-%d" % engine._counter), "wb") as out:
            pickle.dump(whole, out)
    if group is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


def _launcher_argv(gpus, argv):
    """`python -m pylda_amd.launch_train ... --gpus N` as N ranks of this node (rendezvous on 127.0.0.1)."""
    import socket
    probe = socket.socket()
    probe.bind(("127.0.0.1", 0))
    port = probe.getsockname()[1]
    probe.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "pylda_amd.launch_train"] + list(argv)


def _join_ranks(opt):
    """(rank, world, device, process group) of a --gpus N run; (0, 1, --device, None) otherwise."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and opt.gpus <= 1:
        # started under torchrun (or another launcher) without --gpus: WORLD_SIZE ranks that each trained the whole
        # corpus on --device and wrote the same run directory would be silently wrong - the launcher's size it is
        opt.gpus = world
    if "WORLD_SIZE" in os.environ and opt.gpus != world:
        # (also --gpus 4 under a launcher of ONE rank: a quarter of the machine silently doing all of the work)
        raise SystemExit("--gpus %d does not match the launcher's WORLD_SIZE %d" % (opt.gpus, world))
    if opt.gpus <= 1 or world <= 1:
        return 0, 1, opt.device, None
    import torch
    import torch.distributed as dist
    if "RANK" not in os.environ:
        raise SystemExit("WORLD_SIZE=%d is set but RANK is not: start the ranks with torch.distributed.run (or pass "
                         "--gpus N and let launch_train start them)" % world)
    rank = int(os.environ["RANK"])
    device = 0 if opt.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(device)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if opt.share_gpu:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    return rank, world, device, dist.group.WORLD


def _host_group():
    """A gloo group next to the RCCL one, for the host-side object exchanges (eta of the start, gamma for exports)."""
    import torch.distributed as dist
    if dist.get_backend() == "gloo":
        return dist.group.WORLD
    if not hasattr(_host_group, "group"):
        _host_group.group = dist.new_group(backend="gloo")
    return _host_group.group


def _line_ranges(documents, world):
    """Contiguous ranges of the corpus' lines, one per rank, balanced by TOKENS (blank-separated fields: the proxy for
    distinct terms that is known before parsing - SURVEY 8e balances by nnz, which needs the vocabulary look-up; bytes,
    the round-5 proxy, also weigh long words): world + 1 line offsets, the same on every rank."""
    weight = numpy.fromiter((line.count(" ") + 1 for line in documents), dtype=numpy.int64, count=len(documents))
    ends = numpy.cumsum(weight)
    total = int(ends[-1]) if len(ends) else 0
    bounds = [0]
    for r in range(1, world):
        at = int(numpy.searchsorted(ends, total * r / world, side="left"))
        bounds.append(min(max(at, bounds[-1]), len(documents)))
    bounds.append(len(documents))
    return bounds


def _initialize_shard(engine, documents, vocabulary, topics, prior_topics, prior_words, rank, world):
    """variational_bayes.py:82-96 on one rank of several: this rank parses ITS lines only (round 4 parsed the whole
    corpus on every rank and threw 1 - 1/N of it away), rank 0 alone draws eta (:95, the process' first draw from
    numpy's global stream - the reference's own initial state under a seed) and every rank receives it."""
    import torch
    import torch.distributed as dist
    from pylda_amd.inferencer import Inferencer
    lo, hi = _line_ranges(documents, world)[rank:rank + 2]
    Inferencer._initialize(engine, vocabulary, topics, prior_topics, prior_words)
    engine._parsed_corpus = None
    verbose, engine._verbose = engine._verbose, False
    engine._train_csr = engine.parse_to_csr(documents[lo:hi])
    engine._verbose = verbose
    engine._number_of_documents = len(engine._train_csr[0]) - 1
    parsed = torch.tensor([engine._number_of_documents], dtype=torch.int64)
    dist.all_reduce(parsed, group=_host_group())
    if verbose and rank == 0:       # the reference's line (variational_bayes.py:128) with the CORPUS' count, once
        print("successfully parse %d documents..." % int(parsed))
    engine._gamma = None
    engine._gamma_init_pending = True
    shape = (engine._number_of_topics, engine._number_of_types)
    eta = torch.from_numpy(numpy.random.gamma(100., 1. / 100., shape)) if rank == 0 else torch.empty(shape, dtype=torch.float64)
    dist.broadcast(eta, src=0, group=_host_group())
    engine._eta = eta.numpy()
    engine._ctx = None
    engine._train_corpus = None


def _gather_rows(local, rank, world):
    """The ranks' arrays (equal trailing shape) one after the other in a pre-allocated array on rank 0 - tensor
    receives straight into its slices, no pickling (gamma is 2 GB at cfg 4); None elsewhere."""
    import torch
    import torch.distributed as dist
    group = _host_group()
    local = numpy.ascontiguousarray(local)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local.shape[0]], dtype=torch.int64), group=group)
    sizes = [int(t) for t in sizes]
    if rank != 0:
        if local.shape[0]:
            dist.send(torch.from_numpy(local), dst=0, group=group)
        return None
    whole = numpy.empty((sum(sizes),) + local.shape[1:], dtype=local.dtype)
    whole[:sizes[0]] = local
    at = sizes[0]
    for src in range(1, world):
        if sizes[src]:
            dist.recv(torch.from_numpy(whole[at:at + sizes[src]]), src=src, group=group)
        at += sizes[src]
    return whole


def _whole_model(engine, group, rank, world, with_corpus=False, reuse_gamma=False, gathered=None):
    """The engine the exporters and the snapshot pickle see: at one rank the engine itself; at several, on rank 0, a
    copy that holds the gamma rows of every rank in document order (the shards are contiguous) and - for the snapshot,
    which like the reference's pickle carries the parsed corpus - the whole corpus, gathered once.  reuse_gamma (the
    same on every rank): `gathered` is what this function returned for an export of the SAME iteration - rank 0 takes
    its gamma from there, only the corpus travels."""
    if group is None:
        return engine
    import copy
    if reuse_gamma:
        gamma = gathered._gamma_host if rank == 0 else None
    else:
        gamma = _gather_rows(numpy.asarray(engine._gamma), rank, world)
    corpus = None
    if with_corpus:
        doc_ptr, term_id, term_ct = engine._train_csr
        lengths = _gather_rows(numpy.diff(numpy.asarray(doc_ptr, dtype=numpy.int64)), rank, world)
        ids = _gather_rows(numpy.asarray(term_id, dtype=numpy.int32), rank, world)
        cts = _gather_rows(numpy.asarray(term_ct, dtype=numpy.int32), rank, world)
        if rank == 0:
            corpus = (numpy.concatenate([numpy.zeros(1, numpy.int64), numpy.cumsum(lengths)]), ids, cts)
    if rank != 0:
        return None
    whole = copy.copy(engine)
    whole.__dict__.update(engine.__getstate__())          # host copies only, no device handles
    whole._gamma_host = gamma
    whole._train_csr = corpus
    whole._parsed_lists = None
    whole._number_of_documents = gamma.shape[0]
    return whole


def evaluate_snapshot(snapshot_path, test_documents, gamma_path):
    with open(snapshot_path, "rb") as stream:
        engine = pickle.load(stream)
    log_likelihood, gamma = engine.inference(test_documents)
    print("held-out likelihood of snapshot %s is %g" % (os.path.abspath(snapshot_path), log_likelihood))
    numpy.savetxt(gamma_path, gamma)
    return log_likelihood


def test_main(argv=None):
    opt = _parse(TEST_FLAGS, argv, "launch_test")
    if opt.input_directory is None or opt.model_directory is None:
        raise SystemExit("--input_directory and --model_directory are required")
    source = opt.input_directory.rstrip("/")
    models = opt.model_directory.rstrip("/")
    if not os.path.exists(models):
        sys.stderr.write("error: model directory %s does not exist...\n" % os.path.abspath(models))
        return 1
    trained_on = os.path.basename(os.path.dirname(os.path.abspath(models)))
    if os.path.basename(source) != trained_on:
        sys.stderr.write("error: corpus name does not match for input (%s) and model (%s)...\n"
                         % (os.path.basename(source), trained_on))
        return 1
    _banner((("model_directory", models), ("input_directory", source),
             ("corpus_name", os.path.basename(source)), ("snapshot_index", opt.snapshot_index)))
    held_out = _lines(os.path.join(source, "test.dat"))
    print("successfully load all testing docs from %s..." % os.path.abspath(os.path.join(source, "test.dat")))
    if opt.snapshot_index >= 0:
        wanted = ["model-%d" % opt.snapshot_index]
        if not os.path.exists(os.path.join(models, wanted[0])):
            sys.stderr.write("error: model snapshot %s does not exist...\n"
                             % os.path.abspath(os.path.join(models, wanted[0])))
            return 1
    else:
        wanted = sorted(name for name in os.listdir(models) if name.startswith("model-"))
    for name in wanted:
        evaluate_snapshot(os.path.join(models, name), held_out,
                          os.path.join(models, "test-" + name.split("-")[-1]))
    return 0
