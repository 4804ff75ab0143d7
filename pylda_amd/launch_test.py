#!/usr/bin/env python3
"""Held-out evaluation driver with the reference's command line (launch_test.py:10-97).

    python -m pylda_amd.launch_test --input_directory=./associated-press \\
        --model_directory=./associated-press/<run>/ [--snapshot_index=N]

Loads model-N snapshot(s) written by pylda_amd.launch_train, runs inference() on
test.dat (held-out mode of the E-step on the GPU), prints the held-out
log-likelihood and writes the gamma matrix to test-N with numpy.savetxt.
"""
import optparse
import os
import pickle
import sys

import numpy


def parse_args(argv=None):
    parser = optparse.OptionParser()
    parser.set_defaults(input_directory=None, model_directory=None, snapshot_index=-1)
    parser.add_option("--input_directory", type="string", dest="input_directory", help="input directory [None]")
    parser.add_option("--model_directory", type="string", dest="model_directory", help="model directory [None]")
    parser.add_option("--snapshot_index", type="int", dest="snapshot_index",
                      help="snapshot index [-: evaluate on all available snapshots]")
    options, _ = parser.parse_args(argv)
    return options


def evaluate_snapshot(input_snapshot_path, test_docs, output_gamma_path):
    """launch_test.py:90-97."""
    with open(input_snapshot_path, "rb") as stream:
        lda_inferencer = pickle.load(stream)
    log_likelihood, gamma_values = lda_inferencer.inference(test_docs)
    print("held-out likelihood of snapshot %s is %g" % (os.path.abspath(input_snapshot_path), log_likelihood))
    numpy.savetxt(output_gamma_path, gamma_values)
    return log_likelihood


def main(argv=None):
    options = parse_args(argv)
    assert options.input_directory is not None and options.model_directory is not None        # :41-42
    input_directory = options.input_directory.rstrip("/")
    input_corpus_name = os.path.basename(input_directory)
    model_directory = options.model_directory.rstrip("/")
    if not os.path.exists(model_directory):
        sys.stderr.write("error: model directory %s does not exist...\n" % os.path.abspath(model_directory))
        return 1
    corpus_directory = os.path.split(os.path.abspath(model_directory))[0]
    model_corpus_name = os.path.split(os.path.abspath(corpus_directory))[1]
    if input_corpus_name != model_corpus_name:                                               # :55-57
        sys.stderr.write("error: corpus name does not match for input (%s) and model (%s)...\n"
                         % (input_corpus_name, model_corpus_name))
        return 1
    print("========== ========== ========== ========== ==========")
    print("model_directory=" + model_directory)
    print("input_directory=" + input_directory)
    print("corpus_name=" + input_corpus_name)
    print("snapshot_index=" + str(options.snapshot_index))
    print("========== ========== ========== ========== ==========")
    with open(os.path.join(input_directory, "test.dat"), "r") as stream:                      # :62-66
        test_docs = [line.strip().lower() for line in stream]
    print("successfully load all testing docs from %s..." % os.path.abspath(os.path.join(input_directory, "test.dat")))

    if options.snapshot_index >= 0:
        snapshot = os.path.join(model_directory, "model-%d" % options.snapshot_index)
        if not os.path.exists(snapshot):
            sys.stderr.write("error: model snapshot %s does not exist...\n" % os.path.abspath(snapshot))
            return 1
        evaluate_snapshot(snapshot, test_docs, os.path.join(model_directory, "test-%d" % options.snapshot_index))
    else:
        for name in sorted(os.listdir(model_directory)):
            if name.startswith("model-"):
                index = int(name.split("-")[-1])
                evaluate_snapshot(os.path.join(model_directory, name), test_docs,
                                  os.path.join(model_directory, "test-%d" % index))
    return 0


if __name__ == "__main__":
    sys.exit(main())
