"""The engine-independent layer of the reference (inferencer.py): the Dirichlet
expectation helper and the abstract `Inferencer` whose method names the drivers
call.  Kept so that an engine built on pylda_amd looks like the reference's to
launch_train / launch_test."""
import numpy
import scipy.special


def compute_dirichlet_expectation(dirichlet_parameter):
    """E[log x] for x ~ Dirichlet(parameter), row-wise for a matrix (inferencer.py:15-18).
    Host helper for the text exporters; the hot path computes it on the device
    (csrc/prepare_kernels.h)."""
    parameter = numpy.asarray(dirichlet_parameter, dtype=numpy.float64)
    total = parameter.sum(axis=-1, keepdims=parameter.ndim > 1)
    return scipy.special.psi(parameter) - scipy.special.psi(total)


def _engine_must_provide(name):
    def missing(self, *args, **kwargs):
        raise NotImplementedError("%s.%s" % (type(self).__name__, name))
    missing.__name__ = name
    return missing


class Inferencer(object):
    """Base of the inference engines (inferencer.py:29-87): owns the vocabulary maps,
    the prior vectors and the iteration counter; engines supply the rest."""

    def __init__(self, hyper_parameter_optimize_interval=10):
        if hyper_parameter_optimize_interval <= 0:                 # the reference asserts (:38)
            raise AssertionError("hyper_parameter_optimize_interval must be positive")
        self._hyper_parameter_optimize_interval = hyper_parameter_optimize_interval

    def _initialize(self, vocab, number_of_topics, alpha_alpha, alpha_beta):
        """inferencer.py:45-58: symmetric priors as vectors, counter at zero."""
        self.parse_vocabulary(vocab)
        self._number_of_types = len(self._type_to_index)
        self._number_of_topics = number_of_topics
        self._counter = 0
        self._alpha_alpha = numpy.full(number_of_topics, float(alpha_alpha))
        self._alpha_beta = numpy.full(self._number_of_types, float(alpha_beta))

    def parse_vocabulary(self, vocab):
        """Word type <-> id maps (inferencer.py:60-67).

        The reference walks `set(vocab)`, so its ids follow Python's string hashing and
        change from process to process.  Here ids follow first occurrence
        in `vocab`: deterministic, and a caller that passes the reference's own index
        order gets the reference's ids."""
        ordered = list(dict.fromkeys(vocab))
        self._index_to_type = dict(enumerate(ordered))
        self._type_to_index = {word: index for index, word in enumerate(ordered)}
        self._vocab = list(ordered)

    parse_data = _engine_must_provide("parse_data")
    learning = _engine_must_provide("learning")
    inference = _engine_must_provide("inference")
    export_beta = _engine_must_provide("export_beta")
    export_gamma = _engine_must_provide("export_gamma")
