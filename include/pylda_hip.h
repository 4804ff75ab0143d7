/*
 * pylda_hip.h - C ABI of libpylda_hip.so: the MI355X (gfx950) implementation
 * of PyLDA's variational-Bayes E-step hot path.
 *
 * The reference (kzhai/PyLDA) has no FFI: its seam is two Python methods,
 * VariationalBayes.e_step (variational_bayes.py:132-216) and m_step
 * (:218-235), driven by learning() (:239-261) and inference() (:263-271).
 * Each entry point below names the reference lines it replaces.  The Python
 * binding that calls these (ctypes) is pylda_amd/_capi.py; the stub a
 * maintainer of the reference would add is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C types only; the caller owns every host buffer; the library
 *    never keeps a host pointer after the call returns;
 *  - matrices handed over by the host use numpy's C order: eta and sstats
 *    are (K, V) row-major, gamma is (D, K) row-major;
 *  - every function returns 0 on success or a negative pylda_status; the
 *    text of the last failure is pylda_last_error(ctx) (or (NULL) for
 *    failures of pylda_create); no C++ exception crosses the ABI;
 *  - a context is used from one thread at a time (ctypes drops the GIL for
 *    the duration of a call); work is enqueued on the context's HIP stream
 *    and the *_get_* / *_results functions synchronise it;
 *  - there is no CPU fallback: without a usable HIP device pylda_create
 *    fails with PYLDA_ERR_HIP.
 */
#ifndef PYLDA_HIP_H
#define PYLDA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pylda_ctx pylda_ctx;
typedef struct pylda_corpus pylda_corpus;

typedef enum pylda_status {
    PYLDA_OK = 0,
    PYLDA_ERR_INVALID = -1, /* bad argument (shape, range, NULL, unsorted CSR ...) */
    PYLDA_ERR_HIP = -2,     /* HIP runtime error (no device, launch failure ...)   */
    PYLDA_ERR_OOM = -3,     /* host or device allocation failed                    */
    PYLDA_ERR_STATE = -4    /* call sequence error (e.g. results before an e-step) */
} pylda_status;

/* ABI version: bumped whenever an existing entry point changes its signature or meaning.
 *   1  round-1 interface
 *   2  pylda_parse_corpus gained doc_separator, pylda_kernel_time its third output,
 *      pylda_set_stream(NULL) = HIP's null stream (pylda_use_own_stream restores the private one)
 *   3  additions only: pylda_abi_version, pylda_mstep_enqueue / pylda_outer_device / pylda_allreduce_outer /
 *      pylda_outer_fetch (one host wait per outer iteration), pylda_model_checkpoint, pylda_mark_time /
 *      pylda_elapsed_ms, pylda_work_counters, pylda_executed_work, pylda_clock_counters, pylda_host_alloc / pylda_host_free, pylda_test_alpha_update;
 *      pylda_set_alpha no longer waits for the stream (and is a no-op when handed the values the device holds)
 * A host compiled against another version must refuse to run: compare PYLDA_ABI_VERSION with
 * pylda_abi_version() right after loading the library. */
#define PYLDA_ABI_VERSION 3
int pylda_abi_version(void);

/* Library version string, e.g. "pylda_hip 0.3 (gfx950, abi 3)". */
const char* pylda_version(void);

/* Number of visible HIP devices (0 is a valid answer, not an error). */
int pylda_device_count(int* count);

/* Create a context for a model with K topics over V word types on `device`.
 * Allocates the device-resident tables (eta, exp(E_log_eta), sstats: K*V
 * doubles each).  Replaces the per-call numpy allocations of
 * variational_bayes.py:147-152. */
int pylda_create(int device, int K, int V, pylda_ctx** out);
void pylda_destroy(pylda_ctx* ctx);
const char* pylda_last_error(const pylda_ctx* ctx);

/* Run the context's work on an existing HIP stream (e.g. a torch stream, so
 * that an RCCL all-reduce issued through torch.distributed on the same stream
 * is ordered after the E-step without a host sync).  As everywhere in HIP, a
 * NULL handle names the device's default (null) stream - which is what
 * torch.cuda.current_stream().cuda_stream is for torch's default stream.
 * pylda_use_own_stream returns to the context's private (non-blocking)
 * stream.  Both drain the stream in use before switching; the caller keeps a
 * stream it handed over alive. */
int pylda_set_stream(pylda_ctx* ctx, void* hip_stream);
int pylda_use_own_stream(pylda_ctx* ctx);
int pylda_synchronize(pylda_ctx* ctx);

/* Upload a parsed corpus: the (word_ids, word_cts) lists parse_data builds
 * (variational_bayes.py:98-130) flattened to CSR.  doc_ptr has D+1 entries,
 * term ids are in [0, V), counts are >= 1, ids are unique within a document
 * (parse_data builds them from a dict).  Copies to the device once and
 * builds the launch schedule (documents bucketed by distinct-term count). */
int pylda_corpus_create(pylda_ctx* ctx, int64_t D, const int64_t* doc_ptr,
                        const int32_t* term_id, const int32_t* term_ct, pylda_corpus** out);
void pylda_corpus_destroy(pylda_corpus* corpus);
/* D, nnz, token total, largest distinct-term count. */
int pylda_corpus_info(const pylda_corpus* corpus, int64_t* D, int64_t* nnz, int64_t* tokens,
                      int32_t* max_terms);

/* Model state.  eta is self._eta (K, V) (variational_bayes.py:95,226);
 * alpha is self._alpha_alpha (K,) (inferencer.py:57). */
int pylda_set_eta(pylda_ctx* ctx, const double* eta_kv);
int pylda_get_eta(pylda_ctx* ctx, double* eta_kv);
int pylda_set_alpha(pylda_ctx* ctx, const double* alpha_k);

/* THE HOT PATH: one E-step over `corpus` (variational_bayes.py:132-216).
 *   max_iter / tol  local_parameter_iteration / _converge_threshold (:132)
 *   heldout         0: training mode (parsed_corpus == None): accumulates the
 *                      sufficient statistics (:207);
 *                   1: held-out mode (:154-155, :202-204): word log-likelihood,
 *                      no sufficient statistics.
 * Asynchronous: enqueues compute_dirichlet_expectation (inferencer.py:15-18),
 * the per-document phi/gamma loop and the reductions on the context's
 * stream.  Results stay on the device until fetched. */
int pylda_estep(pylda_ctx* ctx, pylda_corpus* corpus, int max_iter, double tol, int heldout);

/* Corpus-level scalars of the last E-step (synchronises):
 * document_log_likelihood (:143,:195-199), words_log_likelihood (:144,:204)
 * and how many documents were finished by the log-space safety-net kernel. */
int pylda_estep_results(pylda_ctx* ctx, pylda_corpus* corpus, double* document_log_likelihood,
                        double* words_log_likelihood, int64_t* logspace_documents);

/* phi_sufficient_statistics (K, V) of the last training-mode E-step (:214). */
int pylda_get_sstats(pylda_ctx* ctx, double* sstats_kv);
/* Upload sufficient statistics (K, V) handed to m_step by the caller (:218). */
int pylda_set_sstats(pylda_ctx* ctx, const double* sstats_kv);
/* gamma_values (D, K) of the last E-step over `corpus` (:213,:216). */
int pylda_get_gamma(pylda_ctx* ctx, pylda_corpus* corpus, double* gamma_dk);
/* Per-document values (any pointer may be NULL): the document's own terms of
 * :195-199, of :204, and the number of inner iterations it ran. */
int pylda_get_doc_values(pylda_ctx* ctx, pylda_corpus* corpus, double* doc_ll,
                         double* doc_words_ll, int32_t* iters);

/* One-shot host-buffer form of the hot path (the signature SURVEY.md 8b
 * proposes): set_alpha + set_eta + estep + fetch.  Output pointers may be
 * NULL.  scalars_out[0] = document_log_likelihood, [1] = words_log_likelihood. */
int pylda_estep_host(pylda_ctx* ctx, pylda_corpus* corpus, const double* alpha_k,
                     const double* eta_kv, int max_iter, double tol, int heldout,
                     double* gamma_dk, double* sstats_kv, double* doc_ll, double* doc_words_ll,
                     int32_t* iters, double* scalars_out);

/* Device-resident views for multi-GPU data parallelism: the sufficient
 * statistics live word-major, (V, ldk) doubles with ldk = pylda_table_stride
 * (K rounded up to 16 / 32 / a multiple of 64; padding columns are zero).
 * The Python side wraps this pointer (zero-copy) and all-reduces its V*ldk
 * elements over RCCL between e_step and m_step.  Also the gamma buffer of a
 * corpus, (D, K). */
int pylda_table_stride(const pylda_ctx* ctx);
void* pylda_sstats_device(pylda_ctx* ctx);
void* pylda_eta_device(pylda_ctx* ctx);      /* (K, V) doubles, numpy layout */
void* pylda_gamma_device(pylda_corpus* corpus);

/* Tell the library that the caller wrote eta (have_eta = 1) and/or the
 * sufficient statistics (have_sstats = 1) through the device pointers above
 * (e.g. after an all-reduce); -1 leaves a flag unchanged. */
int pylda_mark_device_state(pylda_ctx* ctx, int have_eta, int have_sstats);

/* Multi-GPU exchange without Python (documents shard across GPUs, ONE all-reduce of the
 * K x V sufficient statistics per outer iteration).  RCCL is bound at run time (dlopen of librccl.so;
 * PYLDA_RCCL_PATH overrides the search), so the library has no link-time dependency on it.
 *   rank 0:      pylda_comm_unique_id(id)        128 opaque bytes; the host hands them to every rank
 *                                                (MPI, a socket, a file)
 *   every rank:  pylda_comm_init(ctx, id, rank, world_size)      collective
 *   per outer iteration, between pylda_estep and pylda_mstep:
 *                pylda_allreduce_sstats(ctx)     in-place sum of the V x ldk device buffer, enqueued on the
 *                                                context's stream (ordered with the kernels, no host sync)
 *                pylda_allreduce_doubles(ctx, v, n)   sum of a short host vector (document log-likelihood,
 *                                                #documents, alpha statistics: variational_bayes.py:232-233)
 * Every rank then runs the identical pylda_mstep.  The Python class does the same through torch.distributed. */
#define PYLDA_COMM_ID_BYTES 128
int pylda_comm_unique_id(void* id_out);
int pylda_comm_init(pylda_ctx* ctx, const void* id, int rank, int world_size);
int pylda_comm_destroy(pylda_ctx* ctx);
int pylda_allreduce_sstats(pylda_ctx* ctx);
int pylda_allreduce_doubles(pylda_ctx* ctx, double* values, int64_t n);

/* Device M-step (variational_bayes.py:218-235) on the resident buffers:
 * topic log-likelihood from the PRE-update eta (:222-224), eta <- sstats +
 * beta (:226), alpha sufficient statistics from the gamma of `corpus`
 * (:232-233).  beta_v is self._alpha_beta (V,).  alpha_ss_k may be NULL. */
int pylda_mstep(pylda_ctx* ctx, pylda_corpus* corpus, const double* beta_v,
                double* topic_log_likelihood, double* alpha_ss_k);

/* learning() (variational_bayes.py:239-261) with ONE host wait per outer iteration.  After pylda_estep (and the
 * all-reduce of the sufficient statistics when there are several ranks):
 *   pylda_mstep_enqueue   the device half of m_step (:218-235) - topic log-likelihood terms of the PRE-update eta,
 *                         eta <- sstats + beta, alpha sufficient statistics from the gamma of `corpus` - plus a
 *                         pack of every value the host half needs into one device vector; nothing is waited for;
 *                         hyper_parameter_iteration > 0 also asks for the alpha update of the iteration
 *                         (optimize_hyperparameters, :277-324, the reference's defaults: 100, 0.9, 10, 1e-6) ON THE
 *                         DEVICE - it runs inside pylda_outer_fetch, behind the sum over the ranks, and leaves the new
 *                         alpha where the next E-step reads it; 0: alpha stays (hyper_parameter_optimize_interval);
 *   pylda_outer_device    that vector (3K + 4 doubles) and how many of its LEADING elements are rank-local sums:
 *                         [document log-likelihood (:214), #documents, documents redone in log space, 0,
 *                          alpha sufficient statistics (K) (:232-233) | per-topic likelihood terms (K) | alpha (K),
 *                          both replicated];
 *                         a multi-rank host sums the leading *n_reduce elements over the ranks in place, on the
 *                         context's stream (the Python class: torch.distributed; a C host: pylda_allreduce_outer);
 *   pylda_outer_fetch     [the alpha update] + one device-to-host copy + the wait; any output pointer may be NULL;
 *                         alpha_k receives the alpha the device now holds (updated or not). */
int pylda_mstep_enqueue(pylda_ctx* ctx, pylda_corpus* corpus, const double* beta_v, int hyper_parameter_iteration,
                        double hyper_parameter_decay_factor, int hyper_parameter_maximum_decay,
                        double hyper_parameter_converge_threshold);
void* pylda_outer_device(pylda_ctx* ctx, int64_t* n_reduce);
int pylda_allreduce_outer(pylda_ctx* ctx);
int pylda_outer_fetch(pylda_ctx* ctx, double* document_log_likelihood, double* number_of_documents,
                      int64_t* logspace_documents, double* topic_log_likelihood, double* alpha_ss_k, double* alpha_k);

/* Page-locked host memory for the arrays of the public e_step() / m_step() contract (eta, the sufficient
 * statistics and gamma as host ndarrays, variational_bayes.py:212-216): buffers from here move at the PCIe rate
 * (~50 GB/s) instead of through the runtime's staging copy of pageable memory (cfg 3: e_step() 52 -> about 25 ms).
 * Any host pointer is still accepted everywhere; this is an allocator, not a requirement. */
int pylda_host_alloc(int64_t bytes, void** out);
int pylda_host_free(void* p);

/* Device-side model checkpoint: restore = 0 saves eta (the model; the counterpart of the reference's
 * snapshot pickles, launch_train.py:203-204, without the trip through the host), restore = 1 puts the saved
 * eta back.  Asynchronous on the context's stream.  bench.py uses it to time the same outer iterations
 * repeatedly. */
int pylda_model_checkpoint(pylda_ctx* ctx, int restore);

/* Time stamps on the context's stream (4 slots): pylda_mark_time records an event NOW in stream order,
 * pylda_elapsed_ms waits for slot_to and returns the device time between the two.  learning() uses them
 * for the reference's "e_step and m_step ... finished in" line (variational_bayes.py:254) now that the
 * host no longer waits between the two steps. */
int pylda_mark_time(pylda_ctx* ctx, int slot);
int pylda_elapsed_ms(pylda_ctx* ctx, int slot_from, int slot_to, double* ms);

/* Profiling: when enabled, every pylda_estep brackets (HIP events on the launch
 * streams) its document kernels as a group, each launch class on its own, and
 * the sufficient-statistics pass (gather + finalize).  pylda_kernel_time
 * returns the accumulated group times (ms) and the number of E-steps since the
 * last reset, and resets them. */
int pylda_set_profiling(pylda_ctx* ctx, int enabled);
int pylda_kernel_time(pylda_ctx* ctx, double* doc_kernel_ms, double* sstats_kernel_ms,
                      int64_t* estep_calls);
/* With profiling enabled every pylda_estep also accumulates, on the device, the inner iterations its documents
 * actually ran (sum_d I_d) and their terms (sum_d I_d N_d: 4 K flops each, :177-185).  Returns and resets them
 * (synchronises). */
int pylda_work_counters(pylda_ctx* ctx, double* inner_iterations, double* inner_iteration_terms);
/* ... and the rest of the SAME read (no device access: the values of the last pylda_work_counters call):
 * what the kernels really ran - tile_entries = sum_d N_d (K x iterations of the dense kernel + tile columns x iterations
 * of the live-topic kernel), two FMAs each, and the documents handed to the live-topic kernel (estep_compact.h: the
 * iterations of :174-190 on the topics whose gamma still differs from alpha); tile_entries / (K x inner_iteration_terms)
 * is the fraction of the dense N_d x K work that was executed - */
int pylda_executed_work(pylda_ctx* ctx, double* tile_entries, double* handed_over);
/* ... and the shader clock under that load: spans of resident kernels of the profiled E-steps (one in 64 live-topic
 * wavefronts, the work counter's own kernel) measured twice - in shader cycles (s_memtime) and in ticks of the constant
 * wall_hz counter (s_memrealtime).  Sustained clock in Hz = shader_ticks / wall_ticks * wall_hz. */
int pylda_clock_counters(pylda_ctx* ctx, double* shader_ticks, double* wall_ticks, double* wall_hz);

/* The launch schedule of a corpus: documents are bucketed by distinct-term
 * count into launch classes (one kernel instantiation each; the classes of one
 * E-step run concurrently on separate streams).  Fills up to `capacity`
 * entries per array (any may be NULL): kernel variant index (see
 * "force_variant"), its geometry code, documents and distinct (doc, term)
 * pairs in the class, and the profiled kernel time accumulated for the class
 * since the last call (ms; reset by the call).  Returns the number of classes
 * (>= 0) or a negative status. */
int pylda_corpus_plan(pylda_corpus* corpus, int32_t capacity, int32_t* variant, int32_t* geometry,
                      int64_t* documents, int64_t* terms, double* kernel_ms);

/* Layout facts of an uploaded corpus (measurement / test hook): "gather_blocks" - document blocks of the
 * statistics gather (1: unblocked; set when the first training E-step builds the postings, 0 before),
 * "gather_segments" - its posting segments, "gather_rounds" - the term ranges the gather is run in so that their
 * segments share one set of partial rows, "gather_partial_rows" - those rows, "gather_sweep_passes" - passes of the
 * persistent sweep that replaces rows and rounds (0: not in use), "gather_live" - 1: the pass reads the documents' lists of
 * live topics (sstats_live.h), "live_off_by_alpha" - 1: alpha keeps more topics from ever dying than a tile of the
 * live-topic kernel has columns, so the corpus runs on the dense kernels alone (decided before every E-step; when it
 * changes the postings are rebuilt once in the layout that fits).  Returns the value, or a negative pylda_status. */
int64_t pylda_corpus_layout(pylda_corpus* corpus, const char* name);

/* Tuning / test options:
 *   "doc_values"     1 (default): pylda_get_doc_values returns complete per-document
 *                    log-likelihoods.  0: training fast path - the corpus-level
 *                    document_log_likelihood is identical, but its log-B entropy term is
 *                    taken once per corpus from the sufficient statistics instead of a
 *                    second table gather per document; doc_ll[] is then unavailable;
 *   "force_logspace" 0|1  run every document through the log-space
 *                         safety-net kernel (the reference's formulation);
 *   "force_variant"  -1 (automatic) or a kernel variant index: 0..2 generic LDS 64/256/512
 *                    threads, 3 generic global, 4 slab, 6 quilt, 9 group-fused streaming (more than 256 terms at
 *                    table stride 64 / 128 / 256), 10 quad, 11 fused streaming, 12 generic with the per-term scalars
 *                    in global memory (documents of any length), 13 fused streaming for 512 < K <= 1024 (5, 7, 8 were
 *                    the column, two-pass streaming and hybrid kernels of rounds 1-2, removed);
 *                    a variant that cannot take a document falls back to the automatic choice;
 *   "gather_rows"    statistics gather (variational_bayes.py:207): 0 64-topic chunks, 1 whole rows,
 *                    2 (default) whole rows with the postings fetched in bulk (table stride 128 / 256);
 *   "gather_blocks"  document blocks of that gather, for corpora created afterwards: -1 (default)
 *                    automatic - blocks of about one L2 while a (term, block) pair keeps >= 8
 *                    postings, else unblocked; 0 / 1 off; n > 1 forced (a multiple of 8);
 *   "gather_sweep"   the document-blocked gather as ONE persistent kernel in which every wavefront owns a few terms
 *                    and all workgroups sweep the document blocks together (no partial rows; table stride 128 / 256):
 *                    0 never, 1 (default) where the partial rows of the dispatch-paced gather would exceed their
 *                    budget, 2 whenever the gather is blocked;
 *   "gather_round_mb" budget of the gather's partial rows (one per (term, document block) pair) in MiB, 0 = 4 GiB:
 *                    beyond it the statistics pass is the sweep (gather_sweep = 1) or the gather runs in rounds over term
 *                    ranges that reuse the rows.  The sweep-or-gather decision is taken against this figure only; the rounds
 *                    are additionally capped by a quarter of the device memory free at the time (more, smaller rounds on
 *                    a busy device) - which changes no bit of the result: partial rows are summed in segment order and the
 *                    likelihood's entropy partials are laid out by blocks of the whole table, whatever the rounds;
 *   "wide_postings"  1: 64-bit CSR positions in the postings whatever the corpus size (automatic from 2^31 pairs);
 *   "sweep_xcd"      1 (default): the sweep's rendezvous per XCD (the 32 workgroups that share an L2), 0: chip-wide;
 *   "sweep_spin"     polls of a rendezvous before a workgroup goes on alone (pacing only, never correctness);
 *   "sweep_sub"      sub-steps the sweep walks a document block's range in (default 4; results are bitwise the same);
 *   "terms_overlap"  1 (default): the document-terms pass of the training fast path runs on an auxiliary stream beside
 *                    the dispatch-paced statistics gather, 0: in front of it;
 *   "launch_order"   1 (default): launch classes with the fewest documents go out first, 0: longest documents first
 *                    (scheduling options never change a bit of the results);
 *   "slab_uber"      1 (default): the slab launch classes of a small corpus go out as one dispatch;
 *   "gather_live"    1 (default): where a corpus hands documents to the live-topic kernel ("compact") the statistics pass
 *                    reads the lists of live topics those documents leave (10 bytes per live topic instead of a row of K
 *                    doubles per posting; one plain walk of the postings: no document blocks, sweep or rounds); 0: the
 *                    row gathers above.  Takes effect for corpora whose postings are built afterwards;
 *   "compact_phase"  1 (default): the live-topic kernels of an E-step run behind ALL its dense kernels, 0: behind their
 *                    launch class on its stream (scheduling only);
 *   "compact"        1 (default): at 64 < K <= 512 a document leaves the dense kernel once few enough topics still move
 *                    (gamma_k != alpha_k bitwise; 8-64 by document length and table stride) and finishes on the live-topic
 *                    kernel; 0: dense kernels only.  Same iteration counts, results equal to rounding (another summation order);
 *   "compact_pair"   -1 (default): from table stride 256 on a document is handed over at twice the columns one wavefront
 *                    holds and starts on two wavefronts that split the columns; 0: never, 1: always;
 *   "compact_stream" 1 (default): the fused streaming kernel (table stride 384 / 512) hands over too - without a tile, the
 *                    live-topic kernel gathers its entries from the table; 0: the quad kernel only;
 *   "compact_cap"    test hook: hand over at this many live topics at most (0: the kernel's capacity; <= 64);
 *   "compact_guard_fail" test hook: the live-topic kernel's exactness guard fails for every document (they are redone by
 *                    the log-space kernel);
 *   "quad" (1: documents of <= 224 distinct terms at 64 < K <= 256 run on the quad kernel), "quad_stream" (1: ... and
 *   those of 225-256 terms too, with word slots streamed from the table), "quilt_odd",
 *   "quilt12", "lds_pad"  A/B switches of kernel geometry ("Tried and measured"). */
int pylda_set_option(pylda_ctx* ctx, const char* name, int64_t value);

/* Test hook: evaluate the device special functions on n host values. */
int pylda_test_special(pylda_ctx* ctx, int64_t n, const double* x, double* digamma_out,
                       double* lgamma_out);

/* Native corpus ingest: parse_data (variational_bayes.py:98-130) without the Python
 * interpreter.  `text` holds the documents, each ended by the byte `doc_separator`
 * ('\n' for the contents of a train.dat file; the Python binding joins the caller's strings
 * with a byte that occurs in none of them) except possibly the last; tokens are separated by
 * the white space Python's str.split() splits on (the Unicode set, UTF-8 encoded - a '\n'
 * inside a document is just white space when it is not the separator).  `vocab` holds the
 * word types, one per line ('\n'), in id order (id = index among the distinct lines).
 * Rules kept from the reference: tokens not in the vocabulary are skipped (:108-109),
 * documents left without tokens are dropped (:116-118); the caller lower-cases/strips
 * lines beforehand if it wants launch_train.py:106 semantics (ASCII lower-casing can be
 * requested with lowercase=1).  Within a document, term ids appear in first-occurrence
 * order (what the reference's dict gives on CPython >= 3.7).
 * Two-call protocol: call with term_id == NULL to get the sizes (*n_docs, *nnz), then
 * with buffers of n_docs+1 / nnz / nnz elements.  Pure host code: no GPU needed. */
int pylda_parse_corpus(const char* text, int64_t text_bytes, int doc_separator, const char* vocab,
                       int64_t vocab_bytes, int lowercase, int64_t* n_docs, int64_t* nnz,
                       int64_t* doc_ptr, int32_t* term_id, int32_t* term_ct, int64_t* dropped_docs);

/* Test hook: the device alpha update (optimize_hyperparameters, variational_bayes.py:277-324) on host vectors:
 * alpha_k, alpha_ss_k (K each), #documents -> alpha_out_k. */
int pylda_test_alpha_update(pylda_ctx* ctx, const double* alpha_k, const double* alpha_ss_k, double number_of_documents,
                            int hyper_parameter_iteration, double hyper_parameter_decay_factor,
                            int hyper_parameter_maximum_decay, double hyper_parameter_converge_threshold,
                            double* alpha_out_k);

/* Test hook: out[i] = exp(digamma(x[i]) - c), the fused form the inner loop uses. */
int pylda_test_expdigamma(pylda_ctx* ctx, int64_t n, const double* x, double c, double* out);

#ifdef __cplusplus
}
#endif
#endif /* PYLDA_HIP_H */
