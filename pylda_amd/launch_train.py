#!/usr/bin/env python3
"""Training driver with the reference's command line (launch_train.py:10-204),
running the variational-Bayes engine on an MI355X.

    python -m pylda_amd.launch_train --input_directory=./associated-press --output_directory=./ \\
        --number_of_topics=10 --training_iterations=100 --inference_mode=2

Same flags, same output layout (<output>/<corpus>/<timestamp>-lda-I..-S..-K..-aa..-ab..-im../
with option.txt, exp_beta-N, exp_gamma-N, model-N).  Only --inference_mode=2
(variational Bayes) exists here: the hybrid and Monte-Carlo engines are outside
this repository's scope and are refused loudly instead of silently substituted.
"""
import datetime
import optparse
import os
import pickle
import sys


def parse_args(argv=None):
    parser = optparse.OptionParser()
    parser.set_defaults(input_directory=None, output_directory=None, training_iterations=-1,
                        snapshot_interval=10, number_of_topics=-1, alpha_alpha=-1, alpha_beta=-1,
                        inference_mode=2, device=0)
    parser.add_option("--input_directory", type="string", dest="input_directory", help="input directory [None]")
    parser.add_option("--output_directory", type="string", dest="output_directory", help="output directory [None]")
    parser.add_option("--number_of_topics", type="int", dest="number_of_topics", help="total number of topics [-1]")
    parser.add_option("--training_iterations", type="int", dest="training_iterations",
                      help="total number of iterations [-1]")
    parser.add_option("--snapshot_interval", type="int", dest="snapshot_interval", help="snapshot interval [10]")
    parser.add_option("--alpha_alpha", type="float", dest="alpha_alpha",
                      help="hyper-parameter for Dirichlet distribution of topics [1.0/number_of_topics]")
    parser.add_option("--alpha_beta", type="float", dest="alpha_beta",
                      help="hyper-parameter for Dirichlet distribution of vocabulary [1.0/number_of_types]")
    parser.add_option("--inference_mode", type="int", dest="inference_mode",
                      help="inference mode [2: variational bayes (the only engine of pylda_amd)]")
    parser.add_option("--device", type="int", dest="device", help="GPU index [0]")
    options, _ = parser.parse_args(argv)
    return options


def read_corpus(input_directory):
    """launch_train.py:102-115: one document per line, lower-cased; vocabulary = first column of voc.dat."""
    with open(os.path.join(input_directory, "train.dat"), "r") as stream:
        train_docs = [line.strip().lower() for line in stream]
    with open(os.path.join(input_directory, "voc.dat"), "r") as stream:
        vocab = [line.strip().lower().split()[0] for line in stream if line.strip()]
    vocab = list(dict.fromkeys(vocab))          # unique, first-occurrence order (deterministic)
    return train_docs, vocab


def main(argv=None):
    options = parse_args(argv)
    assert options.number_of_topics > 0                                          # launch_train.py:73
    assert options.training_iterations > 0                                       # :75
    assert options.snapshot_interval > 0                                         # :77
    assert options.input_directory is not None and options.output_directory is not None   # :87-88
    if options.inference_mode != 2:
        sys.stderr.write("error: pylda_amd implements inference mode 2 (variational bayes) only, "
                         "got %d...\n" % options.inference_mode)
        return 2
    number_of_topics = options.number_of_topics
    training_iterations = options.training_iterations
    snapshot_interval = options.snapshot_interval
    input_directory = options.input_directory.rstrip("/")
    corpus_name = os.path.basename(input_directory)
    output_directory = os.path.join(options.output_directory, corpus_name)
    os.makedirs(output_directory, exist_ok=True)

    train_docs, vocab = read_corpus(input_directory)
    print("successfully load all training docs from %s..." % os.path.abspath(os.path.join(input_directory, "train.dat")))
    print("successfully load all the words from %s..." % os.path.abspath(os.path.join(input_directory, "voc.dat")))

    alpha_alpha = options.alpha_alpha if options.alpha_alpha > 0 else 1.0 / number_of_topics      # :119-121
    alpha_beta = options.alpha_beta if options.alpha_beta > 0 else 1.0 / len(vocab)               # :122-124

    suffix = datetime.datetime.now().strftime("%y%m%d-%H%M%S")                                    # :127-138
    suffix += "-lda-I%d-S%d-K%d-aa%f-ab%f-im%d/" % (training_iterations, snapshot_interval, number_of_topics,
                                                    alpha_alpha, alpha_beta, options.inference_mode)
    output_directory = os.path.join(output_directory, suffix)
    os.mkdir(os.path.abspath(output_directory))

    with open(output_directory + "option.txt", "w") as out:                                       # :148-162
        out.write("input_directory=" + input_directory + "\n")
        out.write("corpus_name=" + corpus_name + "\n")
        out.write("training_iterations=%d\n" % training_iterations)
        out.write("snapshot_interval=" + str(snapshot_interval) + "\n")
        out.write("number_of_topics=" + str(number_of_topics) + "\n")
        out.write("alpha_alpha=" + str(alpha_alpha) + "\n")
        out.write("alpha_beta=" + str(alpha_beta) + "\n")
        out.write("inference_mode=%d\n" % options.inference_mode)

    print("========== ========== ========== ========== ==========")
    print("output_directory=" + output_directory)
    print("input_directory=" + input_directory)
    print("corpus_name=" + corpus_name)
    print("training_iterations=%d" % training_iterations)
    print("snapshot_interval=" + str(snapshot_interval))
    print("number_of_topics=" + str(number_of_topics))
    print("alpha_alpha=" + str(alpha_alpha))
    print("alpha_beta=" + str(alpha_beta))
    print("inference_mode=%d" % options.inference_mode)
    print("========== ========== ========== ========== ==========")

    from pylda_amd.variational_bayes import VariationalBayes
    lda_inferencer = VariationalBayes(device=options.device)
    lda_inferencer._initialize(train_docs, vocab, number_of_topics, alpha_alpha, alpha_beta)      # :194
    for _ in range(training_iterations):                                                          # :196-201
        lda_inferencer.learning()
        if lda_inferencer._counter % snapshot_interval == 0:
            lda_inferencer.export_beta(output_directory + "exp_beta-" + str(lda_inferencer._counter))
            lda_inferencer.export_gamma(output_directory + "exp_gamma-" + str(lda_inferencer._counter))
    model_snapshot_path = os.path.join(output_directory, "model-" + str(lda_inferencer._counter))
    with open(model_snapshot_path, "wb") as out:                                                  # :203-204
        pickle.dump(lda_inferencer, out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
