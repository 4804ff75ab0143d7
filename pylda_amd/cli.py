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
    ("device", int, 0, "GPU index [0]"),
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
    os.makedirs(os.path.abspath(run_dir))
    settings = (("input_directory", source), ("corpus_name", corpus_name),
                ("training_iterations", "%d" % opt.training_iterations),
                ("snapshot_interval", str(opt.snapshot_interval)), ("number_of_topics", str(topics)),
                ("alpha_alpha", str(prior_topics)), ("alpha_beta", str(prior_words)),
                ("inference_mode", "%d" % opt.inference_mode))
    with open(run_dir + "option.txt", "w") as out:
        out.writelines("%s=%s\n" % pair for pair in settings)
    _banner((("output_directory", run_dir),) + settings[:1] + settings[1:])

    from pylda_amd.variational_bayes import VariationalBayes
    engine = VariationalBayes(device=opt.device)
    engine._initialize(documents, vocabulary, topics, prior_topics, prior_words)
    for _ in range(opt.training_iterations):
        engine.learning()
        if engine._counter % opt.snapshot_interval == 0:
            engine.export_beta("%sexp_beta-%d" % (run_dir, engine._counter))
            engine.export_gamma("%sexp_gamma-%d" % (run_dir, engine._counter))
    with open(os.path.join(run_dir, "model-%d" % engine._counter), "wb") as out:
        pickle.dump(engine, out)
    return 0


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
