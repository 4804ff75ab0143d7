"""VariationalBayes: PyLDA's variational-Bayes LDA engine with the E-step on an
MI355X.

Host-side mirror of the reference's plugin seam: same class name, method
names, keyword defaults, return orders and attributes as
/root/reference/variational_bayes.py:55-356, so that a launch_train-style
driver is a drop-in.  The arithmetic of the hot path (e_step, :132-216) runs
in libpylda_hip.so through the C ABI (pylda_amd/_capi.py); there is no CPU
implementation of it in this package.

Device residency: eta, the sufficient statistics and gamma live on the GPU
between calls.  `_eta` and `_gamma` are lazy properties: reading them pulls
the device copy once, assigning them marks the host copy as the newer one
(it is uploaded at the next E-step).  learning() therefore moves only
scalars and K-vectors across PCIe per outer iteration, while the public
e_step()/m_step() keep the reference's host-array contract exactly.
"""
import sys
import time

import numpy
import scipy.special

from pylda_amd import _capi
from pylda_amd.corpus import csr_to_lists, lists_to_csr
from pylda_amd.inferencer import Inferencer, compute_dirichlet_expectation


class VariationalBayes(Inferencer):
    def __init__(self, hyper_parameter_optimize_interval=1, device=0, process_group=None):
        Inferencer.__init__(self, hyper_parameter_optimize_interval)
        self._device = device
        self._process_group = process_group        # torch.distributed group, or None
        self._ctx = None
        self._train_corpus = None
        self._eta_host = None
        self._gamma_host = None
        self._eta_device_newer = False      # device eta is ahead of the host copy
        self._eta_host_newer = False        # host eta must be uploaded before the next E-step
        self._gamma_host_stale = False      # device gamma is ahead of the host copy
        self._gamma_on_device = False       # the training corpus holds the gamma of an E-step
        self._verbose = True
        # Side effects of the reference's E-step loop that have no counterpart on the device, reproduced
        # on request (both off by default):
        #   _reference_rng_stream   draw (and discard) numpy.random.permutation(D) per E-step (:159) so that
        #                           numpy's global RNG stays in step with a reference run;
        #   _progress_lines         the "successfully processed %d documents..." lines (:209-210).
        self._reference_rng_stream = False
        self._progress_lines = False

    # ------------------------------------------------------------------ state
    @property
    def _eta(self):
        if self._eta_device_newer:
            self._eta_host = self._ctx.get_eta()
            self._eta_device_newer = False
        return self._eta_host

    @_eta.setter
    def _eta(self, value):
        self._eta_host = value
        self._eta_device_newer = False
        self._eta_host_newer = True

    @property
    def _gamma(self):
        if self._gamma_host_stale:
            self._gamma_host = self._ctx.get_gamma(self._train_corpus)
            self._gamma_host_stale = False
        elif self._gamma_host is None and self.__dict__.get("_gamma_init_pending"):
            # variational_bayes.py:92 (D x K doubles nothing on the hot path reads: built on demand)
            self._gamma_host = (numpy.zeros((self._number_of_documents, self._number_of_topics))
                                + self._alpha_alpha[numpy.newaxis, :]
                                + 1.0 * self._number_of_types / self._number_of_topics)
        return self._gamma_host

    @_gamma.setter
    def _gamma(self, value):
        self._gamma_host = value
        self._gamma_host_stale = False
        self._gamma_on_device = False
        self._gamma_init_pending = False

    def __getstate__(self):
        """Snapshots are pickles of the whole object (launch_train.py:203-204):
        materialise the host copies, drop the live device handles."""
        state = dict(self.__dict__)
        state["_eta_host"] = self._eta
        state["_gamma_host"] = self._gamma
        for key in ("_ctx", "_train_corpus", "_process_group"):
            state[key] = None
        state["_eta_device_newer"] = state["_gamma_host_stale"] = False
        state["_gamma_on_device"] = False
        state["_eta_host_newer"] = True
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)

    def _context(self):
        if self._ctx is None:
            self._ctx = _capi.Context(self._number_of_topics, self._number_of_types, self._device)
            self._eta_host_newer = True
            # the reference's API exposes corpus-level likelihoods only (:214,:216)
            self._ctx.set_option("doc_values", 0)
            if self._process_group is not None:
                # run on torch's current stream so the RCCL all-reduce issued through
                # torch.distributed is stream-ordered with the kernels, no host sync
                from pylda_amd import distributed
                distributed.bind_to_torch_stream(self._ctx)
        return self._ctx

    def _training_corpus(self):
        if self._train_corpus is None:
            csr = self.__dict__.get("_train_csr")
            if csr is None:
                csr = lists_to_csr(*self._parsed_corpus)
            self._train_corpus = self._context().corpus(*csr)
        return self._train_corpus

    @property
    def _parsed_corpus(self):
        """The reference's container (:120-121), built on demand from the CSR the native parser made."""
        if self.__dict__.get("_parsed_lists") is None and self.__dict__.get("_train_csr") is not None:
            self._parsed_lists = csr_to_lists(*self._train_csr)
        return self.__dict__.get("_parsed_lists")

    @_parsed_corpus.setter
    def _parsed_corpus(self, value):
        self._parsed_lists = value
        self._train_csr = None
        self._train_corpus = None        # the device copy is rebuilt from the new container

    def _reference_side_effects(self, number_of_documents):
        if self.__dict__.get("_reference_rng_stream"):
            numpy.random.permutation(number_of_documents)                      # :159 (visiting order; discarded)
        if self.__dict__.get("_progress_lines"):
            for done in range(1000, number_of_documents + 1, 1000):            # :209-210 (the reference prints
                print("successfully processed %d documents..." % done)         #  doc_id + 1 in visiting order)

    def _push_model(self):
        ctx = self._context()
        ctx.set_alpha(self._alpha_alpha)
        if self._eta_host_newer:
            ctx.set_eta(self._eta_host)
            self._eta_host_newer = False

    # ------------------------------------------------------------ initialise
    def _initialize(self, corpus, vocab, number_of_topics, alpha_alpha, alpha_beta):
        """variational_bayes.py:82-96 (same RNG draw for eta, so a seeded run
        starts from the reference's own initial state)."""
        Inferencer._initialize(self, vocab, number_of_topics, alpha_alpha, alpha_beta)
        self._parsed_corpus = None
        self._train_csr = self.parse_to_csr(corpus)
        self._number_of_documents = len(self._train_csr[0]) - 1
        self._gamma = None
        self._gamma_init_pending = True      # :92's value, materialised when first read
        self._eta = numpy.random.gamma(100., 1. / 100.,
                                       (self._number_of_topics, self._number_of_types))  # :95
        self._ctx = None
        self._train_corpus = None

    def _initialize_parsed(self, doc_ptr, term_id, term_ct, number_of_types, number_of_topics,
                           alpha_alpha, alpha_beta, eta=None):
        """Same state as _initialize for a corpus that is already parsed to CSR
        (synthetic corpora, shards of a distributed run): no vocabulary, no text."""
        self._type_to_index = {}
        self._index_to_type = {}
        self._number_of_types = int(number_of_types)
        self._counter = 0
        self._number_of_topics = int(number_of_topics)
        self._alpha_alpha = numpy.zeros(self._number_of_topics) + alpha_alpha
        self._alpha_beta = numpy.zeros(self._number_of_types) + alpha_beta
        self._parsed_corpus = None
        self._train_csr = (numpy.asarray(doc_ptr), numpy.asarray(term_id), numpy.asarray(term_ct))
        self._number_of_documents = len(doc_ptr) - 1
        self._gamma = None
        self._gamma_init_pending = True      # :92's value, materialised when first read
        if eta is None:
            eta = numpy.random.gamma(100., 1. / 100., (self._number_of_topics, self._number_of_types))
        self._eta = eta
        self._ctx = None
        self._train_corpus = self._context().corpus(doc_ptr, term_id, term_ct)

    def parse_to_csr(self, corpus):
        """Text lines -> CSR (doc_ptr, term_id, term_ct) with the rules of parse_data
        (variational_bayes.py:98-130): tokens outside the vocabulary are skipped (:108-109),
        documents left empty are dropped with a warning (:116-118).  Runs in the native
        library (pylda_parse_corpus): Python dict parsing dominates start-up at 1M documents."""
        vocabulary = [self._index_to_type[i] for i in range(len(self._index_to_type))]
        doc_ptr, term_id, term_ct, dropped = _capi.parse_corpus(list(corpus), vocabulary)
        for _ in range(dropped):
            sys.stderr.write("warning: document collapsed during parsing")
        if self._verbose:
            print("successfully parse %d documents..." % (len(doc_ptr) - 1))
        return doc_ptr, term_id, term_ct

    def parse_data(self, corpus):
        """The reference's signature and container: ([ids (N_d,)], [counts (1, N_d)])."""
        return csr_to_lists(*self.parse_to_csr(corpus))

    # ---------------------------------------------------------------- E-step
    def e_step(self, parsed_corpus=None, local_parameter_iteration=50,
               local_parameter_converge_threshold=1e-6):
        """variational_bayes.py:132-216, on the GPU.

        Training mode (parsed_corpus is None): returns (document_log_likelihood,
        phi_sufficient_statistics (K, V) ndarray) and sets self._gamma.
        Held-out mode: returns (words_log_likelihood, gamma_values (D, K));
        self._gamma is left untouched (:212-216).
        """
        ctx = self._context()
        self._push_model()
        if parsed_corpus is None:
            corpus = self._training_corpus()
            ctx.estep(corpus, local_parameter_iteration, local_parameter_converge_threshold, False)
            self._reference_side_effects(corpus.D)
            document_log_likelihood, _, _ = ctx.estep_results(corpus)
            self._gamma_host_stale = self._gamma_on_device = True
            return document_log_likelihood, ctx.get_sstats()
        if len(parsed_corpus) == 3:                       # already CSR (inference fast path)
            corpus = ctx.corpus(*parsed_corpus)
        else:
            word_ids, word_cts = parsed_corpus
            assert len(word_ids) == len(word_cts)                               # :140
            corpus = ctx.corpus(*lists_to_csr(word_ids, word_cts))
        try:
            ctx.estep(corpus, local_parameter_iteration, local_parameter_converge_threshold, True)
            self._reference_side_effects(corpus.D)
            _, words_log_likelihood, _ = ctx.estep_results(corpus)
            gamma_values = ctx.get_gamma(corpus)
        finally:
            corpus.close()
        return words_log_likelihood, gamma_values

    # ---------------------------------------------------------------- M-step
    def m_step(self, phi_sufficient_statistics):
        """variational_bayes.py:218-235 on the device: topic log-likelihood from
        the pre-update eta, eta <- sstats + beta, alpha sufficient statistics
        from the gamma of the last training E-step."""
        ctx = self._context()
        self._push_model()
        phi_sufficient_statistics = numpy.asarray(phi_sufficient_statistics, dtype=numpy.float64)
        assert phi_sufficient_statistics.shape == (self._number_of_topics, self._number_of_types)
        ctx.set_sstats(phi_sufficient_statistics)
        if self._gamma_on_device:
            topic_log_likelihood, alpha_sufficient_statistics = ctx.mstep(self._training_corpus(),
                                                                          self._alpha_beta)
        else:
            # self._gamma was assigned by the caller (or comes from _initialize / a snapshot): the
            # K-vector of :232-233 from that host array, as the reference does
            topic_log_likelihood, _ = ctx.mstep(None, self._alpha_beta, want_alpha_ss=False)
            alpha_sufficient_statistics = numpy.sum(
                compute_dirichlet_expectation(numpy.asarray(self._gamma, dtype=numpy.float64)), axis=0)
        self._eta_device_newer = True
        self._eta_host_newer = False
        return topic_log_likelihood, alpha_sufficient_statistics

    # -------------------------------------------------------------- learning
    def learning(self):
        """One outer VB iteration (variational_bayes.py:239-261), device-resident:
        E-step -> [RCCL all-reduce of the sufficient statistics] -> M-step ->
        alpha Newton update.  Only scalars and K-vectors cross PCIe.

        The reference's learning() is a template method: it calls self.e_step(), self.m_step() and
        self.optimize_hyperparameters() (:243-249), and hybrid.py:23,85 overrides e_step alone.  A subclass (or an
        instance attribute) that replaces e_step or m_step is therefore honoured: the iteration then runs through
        the public seam with its host arrays, exactly as the reference does; the fused device path is taken only
        while both are this class' own."""
        if self._seam_is_overridden():
            return self._learning_through_seam()
        self._counter += 1
        ctx = self._context()
        self._push_model()
        corpus = self._training_corpus()

        # Everything is enqueued before anything is read back: E-step -> [all-reduce] -> device M-step -> pack ->
        # [all-reduce of the packed rank-local values] -> ONE copy + wait (pylda_outer_fetch).  The reference's two
        # wall-clock spans (:241-250) are therefore device time between stream marks, not host time.
        timed = self._verbose
        if timed:
            ctx.mark_time(0)
        ctx.estep(corpus, 50, 1e-6, False)
        self._reference_side_effects(corpus.D)
        group = self._process_group
        if group is not None:
            from pylda_amd import distributed
            distributed.allreduce_sstats(ctx, group)
        self._gamma_host_stale = self._gamma_on_device = True
        if timed:
            ctx.mark_time(1)
        update_alpha = self._hyper_parameter_optimize_interval > 0 and \
            self._counter % self._hyper_parameter_optimize_interval == 0
        # the alpha update (optimize_hyperparameters, :277-324, with the reference's defaults) runs on the device
        # behind the sum over the ranks, unless a subclass replaced the method or _device_alpha_update is cleared
        # (a method patched on the INSTANCE counts as a replacement too; and the device update divides by the corpus'
        # document count, so a caller who changed _number_of_documents gets the host form, which honours it)
        on_device = update_alpha and self.__dict__.get("_device_alpha_update", True) and \
            type(self).optimize_hyperparameters is VariationalBayes.optimize_hyperparameters and \
            "optimize_hyperparameters" not in self.__dict__ and \
            (group is not None or self._number_of_documents == corpus.D)
        ctx.mstep_enqueue(corpus, self._alpha_beta, hyper_parameter_iteration=100 if on_device else 0)
        if group is not None:
            distributed.allreduce_outer(ctx, group)
        if timed:
            ctx.mark_time(2)
        document_log_likelihood, number_of_documents, _, topic_log_likelihood, alpha_sufficient_statistics, alpha = \
            ctx.outer_fetch()
        self._eta_device_newer = True
        clock_e_step = ctx.elapsed_ms(0, 1) * 1e-3 if timed else 0.0
        clock_m_step = time.time()
        if on_device:
            self._alpha_alpha = alpha
        elif update_alpha:
            # (one rank: the reference's own count, self._number_of_documents, :279; several: the sum over the ranks)
            self.optimize_hyperparameters(alpha_sufficient_statistics,
                                          number_of_documents=number_of_documents if group is not None else None)
        clock_m_step = time.time() - clock_m_step + (ctx.elapsed_ms(1, 2) * 1e-3 if timed else 0.0)

        joint_log_likelihood = document_log_likelihood + topic_log_likelihood
        if self._verbose:
            print("e_step and m_step of iteration %d finished in %d and %d seconds respectively "
                  "with log likelihood %g" % (self._counter, clock_e_step, clock_m_step,
                                              joint_log_likelihood))
        return joint_log_likelihood

    def _seam_is_overridden(self):
        cls = type(self)
        return cls.e_step is not VariationalBayes.e_step or cls.m_step is not VariationalBayes.m_step or \
            "e_step" in self.__dict__ or "m_step" in self.__dict__

    def _learning_through_seam(self):
        """learning() as the reference writes it (:239-261): every step through the overridable methods."""
        self._counter += 1
        group = self._process_group
        clock_e_step = time.time()
        document_log_likelihood, phi_sufficient_statistics = self.e_step()
        if group is not None:       # document shards: the statistics and the likelihood are sums over the ranks
            from pylda_amd import distributed
            phi_sufficient_statistics = distributed.allreduce_host_array(phi_sufficient_statistics, group, self._device)
        clock_e_step = time.time() - clock_e_step
        clock_m_step = time.time()
        topic_log_likelihood, alpha_sufficient_statistics = self.m_step(phi_sufficient_statistics)
        number_of_documents = None
        if group is not None:
            document_log_likelihood, number_of_documents, alpha_sufficient_statistics = distributed.allreduce_small(
                group, document_log_likelihood, self._number_of_documents, alpha_sufficient_statistics)
        if self._hyper_parameter_optimize_interval > 0 and \
                self._counter % self._hyper_parameter_optimize_interval == 0:
            if number_of_documents is None:
                self.optimize_hyperparameters(alpha_sufficient_statistics)
            else:
                self.optimize_hyperparameters(alpha_sufficient_statistics, number_of_documents=number_of_documents)
        clock_m_step = time.time() - clock_m_step
        joint_log_likelihood = document_log_likelihood + topic_log_likelihood
        if self._verbose:
            print("e_step and m_step of iteration %d finished in %d and %d seconds respectively "
                  "with log likelihood %g" % (self._counter, clock_e_step, clock_m_step,
                                              joint_log_likelihood))
        return joint_log_likelihood

    def inference(self, corpus):
        """variational_bayes.py:263-271."""
        parsed_corpus = self.parse_to_csr(corpus)
        words_log_likelihood, corpus_gamma_values = self.e_step(parsed_corpus)
        return words_log_likelihood, corpus_gamma_values

    # --------------------------------------------------------- alpha update
    def optimize_hyperparameters(self, alpha_sufficient_statistics, hyper_parameter_iteration=100,
                                 hyper_parameter_decay_factor=0.9, hyper_parameter_maximum_decay=10,
                                 hyper_parameter_converge_threshold=1e-6, number_of_documents=None):
        """Newton update of alpha with a decaying step (variational_bayes.py:277-324).

        K-sized host arithmetic.  Keeps the reference's formula, including its
        element-wise 1/hessian where Blei's closed form has a sum (:292-295);
        end-to-end likelihood traces only match with it.
        """
        assert alpha_sufficient_statistics.shape == (self._number_of_topics,)
        docs = self._number_of_documents if number_of_documents is None else number_of_documents
        psi, trigamma = scipy.special.psi, lambda x: scipy.special.polygamma(1, x)
        accepted = self._alpha_alpha
        decay = 0
        for _ in range(hyper_parameter_iteration):
            alpha = self._alpha_alpha
            total = numpy.sum(alpha)
            gradient = docs * (psi(total) - psi(alpha)) + alpha_sufficient_statistics
            hessian = -docs * trigamma(alpha)
            if not numpy.all(numpy.isfinite(gradient)):
                print("illegal alpha gradient vector", gradient)
            ratio_sum = numpy.sum(gradient / hessian)
            inverse_hessian = 1.0 / hessian                      # vector, as in the reference
            z = docs * trigamma(total)
            correction = ratio_sum / (1.0 / z + inverse_hessian)
            while True:
                step = numpy.power(hyper_parameter_decay_factor, decay) * (gradient - correction) / hessian
                if numpy.any(alpha <= step):
                    decay += 1
                    if decay > hyper_parameter_maximum_decay:
                        break
                    continue
                accepted = alpha - step
                break
            mean_change = numpy.mean(abs(accepted - alpha))
            self._alpha_alpha = accepted
            if mean_change <= hyper_parameter_converge_threshold:
                break

    # -------------------------------------------------------------- exports
    def export_beta(self, exp_beta_path, top_display=-1):
        """Per-topic word distribution, most probable first (variational_bayes.py:326-341)."""
        E_log_eta = compute_dirichlet_expectation(self._eta)
        with open(exp_beta_path, 'w') as output:
            for topic_index in range(self._number_of_topics):
                output.write("==========\t%d\t==========\n" % (topic_index))
                row = E_log_eta[topic_index, :]
                beta_probability = numpy.exp(row - scipy.special.logsumexp(row))
                ranked = numpy.argsort(beta_probability)[::-1]
                if top_display > 0:
                    ranked = ranked[:top_display]
                for type_index in ranked:
                    output.write("%s\t%g\n" % (self._index_to_type[type_index],
                                               beta_probability[type_index]))

    def export_gamma(self, exp_gamma_path, top_display=-1):
        """Per-document topic proportions, largest first (variational_bayes.py:343-356)."""
        gamma = self._gamma
        exp_gamma = gamma / numpy.sum(gamma, axis=1)[:, numpy.newaxis]
        with open(exp_gamma_path, 'w') as output:
            for document_index in range(self._number_of_documents):
                ranked = numpy.argsort(exp_gamma[document_index, :])[::-1]
                if top_display > 0:
                    ranked = ranked[:top_display]
                output.write("%s\n" % "\t".join(
                    "%d:%g" % (topic_index, exp_gamma[document_index, topic_index])
                    for topic_index in ranked))


if __name__ == "__main__":
    print("not implemented...")
