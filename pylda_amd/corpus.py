"""Corpus containers for the hot path.

The reference's parsed corpus is a pair of Python lists
(variational_bayes.py:120-121): word_ids[d] = int array (N_d,), word_cts[d] =
int array (1, N_d).  The device wants CSR; these helpers convert both ways,
shard a corpus across ranks, and generate the synthetic LDA corpora that
BASELINE.json's configs 3 and 4 name.
"""
import numpy as np


def lists_to_csr(word_ids, word_cts):
    D = len(word_ids)
    lengths = np.fromiter((len(w) for w in word_ids), dtype=np.int64, count=D)
    doc_ptr = np.zeros(D + 1, dtype=np.int64)
    np.cumsum(lengths, out=doc_ptr[1:])
    if D and doc_ptr[-1]:
        term_id = np.concatenate([np.asarray(w).ravel() for w in word_ids]).astype(np.int32)
        term_ct = np.concatenate([np.asarray(c).ravel() for c in word_cts]).astype(np.int32)
    else:
        term_id = np.zeros(0, np.int32)
        term_ct = np.zeros(0, np.int32)
    return doc_ptr, term_id, term_ct


def csr_to_lists(doc_ptr, term_id, term_ct):
    ids, cts = [], []
    for d in range(len(doc_ptr) - 1):
        lo, hi = int(doc_ptr[d]), int(doc_ptr[d + 1])
        ids.append(np.asarray(term_id[lo:hi], dtype=np.int64))
        cts.append(np.asarray(term_ct[lo:hi], dtype=np.int64)[np.newaxis, :])
    return ids, cts


def shard_bounds(doc_ptr, world_size):
    """Contiguous document ranges balanced by nnz: returns
    world_size+1 document offsets."""
    doc_ptr = np.asarray(doc_ptr, dtype=np.int64)
    D = doc_ptr.size - 1
    nnz = int(doc_ptr[-1])
    bounds = [0]
    for r in range(1, world_size):
        target = nnz * r / world_size
        d = int(np.searchsorted(doc_ptr, target, side="left"))
        bounds.append(min(max(d, bounds[-1]), D))
    bounds.append(D)
    return bounds


def shard_csr(doc_ptr, term_id, term_ct, world_size, rank):
    """This rank's slice of the corpus as its own CSR, plus (first_doc, last_doc)."""
    b = shard_bounds(doc_ptr, world_size)
    lo, hi = b[rank], b[rank + 1]
    a, z = int(doc_ptr[lo]), int(doc_ptr[hi])
    return (np.asarray(doc_ptr[lo:hi + 1], dtype=np.int64) - a, np.asarray(term_id[a:z]),
            np.asarray(term_ct[a:z]), (lo, hi))


def _dirichlet_rows(rng, concentration, rows, cols):
    """Dirichlet(concentration * 1) rows via normalised gamma draws (numpy's own
    dirichlet() switches to a slow stick-breaking path for small concentrations)."""
    g = rng.standard_gamma(concentration, size=(rows, cols))
    g /= g.sum(axis=1, keepdims=True)
    return g


def synthetic_lda_corpus(num_docs, vocab_size, true_topics=128, mean_len=200, seed=1234,
                         topic_concentration=0.01, doc_concentration=0.1, chunk=25000):
    """LDA generative corpus (cfg 3/4): beta*_k ~ Dir(0.01), theta_d ~
    Dir(0.1), L_d ~ Poisson(mean_len) >= 1, z ~ theta_d, w ~ beta*_z, collapsed to
    (distinct id, count) per document.  Returns CSR (doc_ptr, term_id, term_ct).

    Seeding is chunk-wise through SeedSequence.spawn so that no process has to
    materialise more than `chunk` documents of tokens at a time, and a shard
    [a, b) of a larger corpus can be generated alone (see `first_doc`)."""
    return synthetic_lda_shard(num_docs, vocab_size, 0, num_docs, true_topics, mean_len, seed,
                               topic_concentration, doc_concentration, chunk)


def _topic_word_cdf(topic_seed, topic_concentration, true_topics, vocab_size):
    rng = np.random.Generator(np.random.PCG64(topic_seed))
    beta = _dirichlet_rows(rng, topic_concentration, true_topics, vocab_size)
    word_cdf = np.cumsum(beta, axis=1)
    word_cdf /= word_cdf[:, -1:]
    word_cdf += np.arange(true_topics)[:, None]          # row k lives in [k, k+1]
    return word_cdf.ravel()


def _draw_chunk(chunk_seed, n, true_topics, mean_len, doc_concentration):
    """Every RANDOM draw of one chunk of documents, in a fixed order, from numpy's PCG64:
    topic mixtures, lengths, and the two uniforms per token.  What follows (two sorted searches and
    a unique-count) is deterministic integer / comparison work that may run anywhere."""
    rng = np.random.Generator(np.random.PCG64(chunk_seed))
    theta = _dirichlet_rows(rng, doc_concentration, n, true_topics)
    lengths = np.maximum(rng.poisson(mean_len, size=n), 1)
    topic_cdf = np.cumsum(theta, axis=1)
    topic_cdf /= topic_cdf[:, -1:]
    topic_cdf += np.arange(n)[:, None]
    tokens = int(lengths.sum())
    return topic_cdf.ravel(), lengths, rng.random(tokens), rng.random(tokens)


def _collapse_chunk_numpy(drawn, word_cdf, n, true_topics, vocab_size):
    topic_cdf, lengths, u_topic, u_word = drawn
    doc_of_token = np.repeat(np.arange(n, dtype=np.int64), lengths)
    z = np.searchsorted(topic_cdf, u_topic + doc_of_token, side="right") - doc_of_token * true_topics
    np.clip(z, 0, true_topics - 1, out=z)
    w = np.searchsorted(word_cdf, u_word + z, side="right") - z * vocab_size
    np.clip(w, 0, vocab_size - 1, out=w)
    uniq, counts = np.unique(doc_of_token * vocab_size + w, return_counts=True)
    per_doc = np.bincount(uniq // vocab_size, minlength=n).astype(np.int64)
    return (uniq % vocab_size).astype(np.int32), counts.astype(np.int32), per_doc


def _collapse_chunk_torch(drawn, word_cdf_dev, n, true_topics, vocab_size, dev):
    """The same arithmetic as _collapse_chunk_numpy on a torch device (IEEE fp64 adds, comparisons and
    integer ops only: bit-identical results), so a 1M-document corpus collapses in seconds."""
    import torch
    topic_cdf, lengths, u_topic, u_word = drawn
    f64 = torch.float64
    topic_cdf = torch.from_numpy(topic_cdf).to(dev)
    doc = torch.repeat_interleave(torch.arange(n, device=dev), torch.from_numpy(lengths).to(dev))
    z = torch.searchsorted(topic_cdf, torch.from_numpy(u_topic).to(dev) + doc.to(f64), right=True) - doc * true_topics
    z.clamp_(0, true_topics - 1)
    w = torch.searchsorted(word_cdf_dev, torch.from_numpy(u_word).to(dev) + z.to(f64), right=True) - z * vocab_size
    w.clamp_(0, vocab_size - 1)
    uniq, counts = torch.unique(doc * vocab_size + w, return_counts=True)
    per_doc = torch.bincount(uniq // vocab_size, minlength=n)
    return ((uniq % vocab_size).to(torch.int32).cpu().numpy(), counts.to(torch.int32).cpu().numpy(),
            per_doc.cpu().numpy().astype(np.int64))


def synthetic_lda_shard(num_docs, vocab_size, first_doc, last_doc, true_topics=128, mean_len=200,
                        seed=1234, topic_concentration=0.01, doc_concentration=0.1, chunk=25000,
                        device=None, workers=1):
    """Documents [first_doc, last_doc) of synthetic_lda_corpus(num_docs, ...):
    identical to slicing the full corpus, without generating the rest.
    first_doc / last_doc must be multiples of `chunk` (or num_docs).

    `device`: a torch device for the deterministic half of the work (searches, unique-count); the
    corpus is the same, bit for bit, as with device=None (all numpy).  `workers`: threads that draw
    chunks ahead (numpy's generators release the GIL)."""
    assert first_doc % chunk == 0 and (last_doc % chunk == 0 or last_doc == num_docs)
    root = np.random.SeedSequence(seed)
    n_chunks = (num_docs + chunk - 1) // chunk
    topic_seed, *chunk_seeds = root.spawn(1 + n_chunks)
    word_cdf = _topic_word_cdf(topic_seed, topic_concentration, true_topics, vocab_size)
    dev = None
    if device is not None:
        import torch
        dev = torch.device(device)
        word_cdf_dev = torch.from_numpy(word_cdf).to(dev)
    todo = list(range(first_doc // chunk, (last_doc + chunk - 1) // chunk))
    sizes = {ci: min(chunk, num_docs - ci * chunk) for ci in todo}

    def draw(ci):
        return _draw_chunk(chunk_seeds[ci], sizes[ci], true_topics, mean_len, doc_concentration)

    ptr_parts, id_parts, ct_parts = [np.zeros(1, np.int64)], [], []
    base = 0
    pool = None
    if workers > 1 and len(todo) > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=workers)
    pending = {}
    try:
        for pos, ci in enumerate(todo):
            if pool is not None:                             # keep `workers` chunks in flight, bounded memory
                for ahead in todo[pos:pos + workers]:
                    if ahead not in pending:
                        pending[ahead] = pool.submit(draw, ahead)
                drawn = pending.pop(ci).result()
            else:
                drawn = draw(ci)
            if dev is None:
                ids, cts, per_doc = _collapse_chunk_numpy(drawn, word_cdf, sizes[ci], true_topics, vocab_size)
            else:
                ids, cts, per_doc = _collapse_chunk_torch(drawn, word_cdf_dev, sizes[ci], true_topics, vocab_size, dev)
            id_parts.append(ids)
            ct_parts.append(cts)
            ptr_parts.append(base + np.cumsum(per_doc))
            base += int(per_doc.sum())
    finally:
        if pool is not None:
            pool.shutdown(wait=True)
    cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dt)
    return np.concatenate(ptr_parts), cat(id_parts, np.int32), cat(ct_parts, np.int32)


def corpus_checksum(doc_ptr, term_id, term_ct):
    """(documents, nnz, sum of term ids, sum of counts): identifies a generated corpus in bench records."""
    return [int(len(doc_ptr) - 1), int(doc_ptr[-1]), int(np.asarray(term_id, dtype=np.int64).sum()),
            int(np.asarray(term_ct, dtype=np.int64).sum())]
