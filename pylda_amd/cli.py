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
    engine = VariationalBayes(device=device, process_group=group)
    if seed is not None:
        numpy.random.seed(int(seed))
    engine._initialize(documents, vocabulary, topics, prior_topics, prior_words)
    if group is not None:
        _shard_engine(engine, group, rank, world)
    for _ in range(opt.training_iterations):
        engine.learning()
        if engine._counter % opt.snapshot_interval == 0:
            whole = _whole_model(engine, group, rank, world)
            if rank == 0:
                whole.export_beta("%sexp_beta-%d" % (run_dir, engine._counter))
                whole.export_gamma("%sexp_gamma-%d" % (run_dir, engine._counter))
    whole = _whole_model(engine, group, rank, world)
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
    if opt.gpus <= 1 or world <= 1:
        return 0, 1, opt.device, None
    if opt.gpus != world:
        raise SystemExit("--gpus %d does not match the launcher's WORLD_SIZE %d" % (opt.gpus, world))
    import torch
    import torch.distributed as dist
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


def _shard_engine(engine, group, rank, world):
    """Every rank parsed the whole corpus and drew its own eta (:95, unseeded in the reference): keep rank 0's
    draw everywhere, and this rank's contiguous nnz-balanced range of the documents."""
    import torch
    import torch.distributed as dist
    from pylda_amd.corpus import shard_csr
    eta = torch.from_numpy(numpy.ascontiguousarray(engine._eta))
    dist.broadcast(eta, src=0, group=_host_group())
    engine._eta = eta.numpy()
    engine._whole_csr = engine._train_csr
    doc_ptr, term_id, term_ct, (lo, hi) = shard_csr(*engine._train_csr, world, rank)
    engine._train_csr = (doc_ptr, term_id, term_ct)
    engine._parsed_lists = None
    engine._number_of_documents = hi - lo
    engine._train_corpus = None


def _whole_model(engine, group, rank, world):
    """The engine the exporters and the snapshot pickle see: at one rank the engine itself; at several, on rank 0, a
    copy that holds the WHOLE corpus and the gamma rows of every rank in document order (the shards are contiguous)."""
    if group is None:
        return engine
    import copy
    import torch.distributed as dist
    rows = [None] * world if rank == 0 else None
    dist.gather_object(numpy.asarray(engine._gamma), rows, dst=0, group=_host_group())
    if rank != 0:
        return None
    whole = copy.copy(engine)
    whole.__dict__.update(engine.__getstate__())          # host copies only, no device handles
    whole._gamma_host = numpy.concatenate(rows, axis=0)
    whole._train_csr = engine._whole_csr
    whole._parsed_lists = None
    whole._number_of_documents = len(engine._whole_csr[0]) - 1
    whole.__dict__.pop("_whole_csr", None)
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
