"""Inferencer base class: the L3 interface of the reference
(/root/reference/inferencer.py:29-87), kept so that engines built on
pylda_amd expose the same method names to launch_train / launch_test."""
import numpy
import scipy.special


def compute_dirichlet_expectation(dirichlet_parameter):
    """E[log x], x ~ Dirichlet (inferencer.py:15-18).  Host helper used by the
    text exporters only; on the hot path the same quantity is computed on the
    device (csrc/prepare_kernels.h)."""
    parameter = numpy.asarray(dirichlet_parameter, dtype=numpy.float64)
    total = parameter.sum(axis=-1, keepdims=parameter.ndim > 1)
    return scipy.special.psi(parameter) - scipy.special.psi(total)


class Inferencer(object):
    def __init__(self, hyper_parameter_optimize_interval=10):
        assert hyper_parameter_optimize_interval > 0                       # inferencer.py:38
        self._hyper_parameter_optimize_interval = hyper_parameter_optimize_interval

    def _initialize(self, vocab, number_of_topics, alpha_alpha, alpha_beta):
        """inferencer.py:45-58."""
        self.parse_vocabulary(vocab)
        self._number_of_types = len(self._type_to_index)
        self._counter = 0
        self._number_of_topics = number_of_topics
        self._alpha_alpha = numpy.zeros(self._number_of_topics) + alpha_alpha
        self._alpha_beta = numpy.zeros(self._number_of_types) + alpha_beta

    def parse_vocabulary(self, vocab):
        """Word type <-> id maps (inferencer.py:60-67).

        The reference iterates `set(vocab)`, so its id order follows Python's
        string hashing and changes from process to process.
        Here ids follow first occurrence in `vocab`: deterministic, and a
        caller that passes the reference's own index order gets its ids.
        """
        self._type_to_index = {}
        self._index_to_type = {}
        for word in vocab:
            if word not in self._type_to_index:
                index = len(self._type_to_index)
                self._type_to_index[word] = index
                self._index_to_type[index] = word
        self._vocab = list(self._type_to_index.keys())

    def parse_data(self):
        raise NotImplementedError

    def learning(self):
        raise NotImplementedError

    def inference(self):
        raise NotImplementedError

    def export_beta(self, exp_beta_path, top_display=-1):
        raise NotImplementedError

    def export_gamma(self, exp_gamma_path, top_display=-1):
        raise NotImplementedError
