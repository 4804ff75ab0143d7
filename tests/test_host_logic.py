"""CPU tests of the host-side logic: corpus containers, sharding, parse_data,
vocabulary order, the synthetic generator, the alpha Newton update."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from pylda_amd import corpus as C


def test_lists_csr_round_trip():
    ids = [np.array([3, 1, 2]), np.array([5]), np.array([0, 4])]
    cts = [np.array([[1, 2, 3]]), np.array([[9]]), np.array([[1, 1]])]
    ptr, tid, tct = C.lists_to_csr(ids, cts)
    assert ptr.tolist() == [0, 3, 4, 6] and tid.dtype == np.int32 and tct.tolist() == [1, 2, 3, 9, 1, 1]
    ids2, cts2 = C.csr_to_lists(ptr, tid, tct)
    assert all(np.array_equal(a, b) for a, b in zip(ids, ids2))
    assert cts2[0].shape == (1, 3)                       # variational_bayes.py:121 layout
    ptr0, tid0, tct0 = C.lists_to_csr([], [])
    assert ptr0.tolist() == [0] and tid0.size == 0


def test_shard_bounds_balance_nnz_and_cover():
    rng = np.random.default_rng(0)
    lens = rng.integers(1, 300, size=1000)
    ptr = np.concatenate([[0], np.cumsum(lens)])
    for world in (1, 2, 3, 8):
        b = C.shard_bounds(ptr, world)
        assert b[0] == 0 and b[-1] == 1000 and all(x <= y for x, y in zip(b, b[1:]))
        loads = [ptr[b[r + 1]] - ptr[b[r]] for r in range(world)]
        assert max(loads) - min(loads) <= 2 * lens.max()
    ids = np.arange(ptr[-1], dtype=np.int32)
    cts = np.ones(ptr[-1], np.int32)
    parts = [C.shard_csr(ptr, ids, cts, 4, r) for r in range(4)]
    assert np.array_equal(np.concatenate([p[1] for p in parts]), ids)
    assert all(p[0][0] == 0 and p[0][-1] == p[1].size for p in parts)
    assert C.shard_bounds(np.array([0]), 4) == [0, 0, 0, 0, 0]          # empty corpus


def test_parse_data_matches_reference_rules(tiny):
    from pylda_amd.variational_bayes import VariationalBayes
    m = VariationalBayes()
    m._verbose = False
    m.parse_vocabulary([str(w) for w in tiny["words"]])
    docs = [str(d) for d in tiny["docs"]] + ["zzz qqq", ""]           # OOV-only and empty lines are dropped
    ids, cts = m.parse_data(docs)
    assert len(ids) == 3
    ptr, tid, tct = C.lists_to_csr(ids, cts)
    assert np.array_equal(ptr, tiny["doc_ptr"])
    for d in range(3):                                                  # same multiset {(id, count)}
        lo, hi = ptr[d], ptr[d + 1]
        assert dict(zip(tid[lo:hi], tct[lo:hi])) == dict(zip(tiny["term_id"][lo:hi], tiny["term_ct"][lo:hi]))
    assert cts[0].shape == (1, len(ids[0])) and ids[0].dtype == np.int64


def test_vocabulary_order_is_first_occurrence():
    from pylda_amd.inferencer import Inferencer
    inf = Inferencer()
    inf._initialize(["b", "a", "b", "c"], 3, 0.5, 0.25)
    assert inf._type_to_index == {"b": 0, "a": 1, "c": 2} and inf._index_to_type[2] == "c"
    assert inf._number_of_types == 3 and inf._alpha_beta.tolist() == [0.25] * 3
    assert inf._alpha_alpha.tolist() == [0.5] * 3 and inf._counter == 0


def test_optimize_hyperparameters_matches_reference_golden(ap_train):
    from pylda_amd.variational_bayes import VariationalBayes
    g = ap_train
    m = VariationalBayes()
    m._number_of_topics, m._number_of_documents = 10, 2000
    m._alpha_alpha = g["alpha"].copy()
    m.optimize_hyperparameters(g["alpha_ss"])
    assert rel_err(m._alpha_alpha, g["alpha_after"]) < 1e-12          # includes the vector-c quirk (:292-295)


def test_dirichlet_expectation_host_helper():
    from pylda_amd.inferencer import compute_dirichlet_expectation
    from oracle import vb_numpy
    x = np.random.default_rng(1).gamma(2.0, 1.0, (4, 9))
    assert np.allclose(compute_dirichlet_expectation(x), vb_numpy.compute_dirichlet_expectation(x), atol=1e-14)
    assert np.allclose(compute_dirichlet_expectation(x[0]), vb_numpy.compute_dirichlet_expectation(x[0]), atol=1e-14)


def test_synthetic_generators_small():
    ptr, ids, cts = C.synthetic_lda_corpus(300, 200, true_topics=5, mean_len=30, seed=3, chunk=100)
    assert len(ptr) == 301 and ptr[-1] == ids.size == cts.size and cts.min() >= 1
    assert ids.min() >= 0 and ids.max() < 200
    for d in (0, 17, 299):                                              # ids unique and sorted within a doc
        w = ids[ptr[d]:ptr[d + 1]]
        assert np.all(np.diff(w) > 0)
    a = C.synthetic_lda_shard(300, 200, 100, 200, true_topics=5, mean_len=30, seed=3, chunk=100)
    assert np.array_equal(a[1], ids[ptr[100]:ptr[200]])                 # shard == slice of the whole
    assert abs(cts.sum() / 300.0 - 30) < 3                                # mean length as requested


def reference_rules_parse(lines, type_to_index):
    """The parse_data rules (variational_bayes.py:104-121) spelled out in plain Python."""
    ids, cts = [], []
    for line in lines:
        tally = {}
        for token in line.split():
            if token in type_to_index:
                tally[type_to_index[token]] = tally.get(type_to_index[token], 0) + 1
        if tally:
            ids.append(list(tally.keys()))
            cts.append(list(tally.values()))
    return ids, cts


def test_native_ingest_matches_parse_data_rules(ap_train):
    from pylda_amd import _capi
    rng = np.random.default_rng(4)
    words = [str(w) for w in ap_train["words"][:500]]
    lines = []
    for _ in range(300):
        n = int(rng.integers(0, 40))
        toks = [words[i] for i in rng.integers(0, 500, n)] + ["zzzoov"] * int(rng.integers(0, 3))
        rng.shuffle(toks)
        lines.append(("  " if rng.random() < 0.2 else "") + " ".join(toks) + ("\t " if rng.random() < 0.2 else ""))
    lines += ["", "zzzoov qqqoov", words[3], " ".join([words[7]] * 1000)]
    ptr, tid, tct, dropped = _capi.parse_corpus(lines, words)
    ref_ids, ref_cts = reference_rules_parse(lines, {w: i for i, w in enumerate(words)})
    assert len(ptr) - 1 == len(ref_ids) and dropped == len(lines) - len(ref_ids)
    for d, (ri, rc) in enumerate(zip(ref_ids, ref_cts)):
        assert tid[ptr[d]:ptr[d + 1]].tolist() == ri            # same ids, same (first-occurrence) order
        assert tct[ptr[d]:ptr[d + 1]].tolist() == rc
    assert tct[-1] == 1000 and tid.dtype == np.int32 and ptr.dtype == np.int64
    # upper-case input is only matched when lower-casing is requested (launch_train.py:106 does it beforehand)
    up = [l.upper() for l in lines[:50]]
    p2, t2, c2, _ = _capi.parse_corpus(up, words, lowercase=True)
    p1, t1, c1, _ = _capi.parse_corpus(lines[:50], words)
    assert np.array_equal(p1, p2) and np.array_equal(t1, t2) and np.array_equal(c1, c2)
    p3, t3, _, d3 = _capi.parse_corpus(up, words)
    assert len(p3) == 1 and t3.size == 0 and d3 == sum(1 for l in up if l.strip() or True)
    # duplicate vocabulary entries keep their first id; empty corpus
    p4, t4, c4, _ = _capi.parse_corpus(["b a b"], ["a", "b", "a"])
    assert t4.tolist() == [1, 0] and c4.tolist() == [2, 1]
    p5, t5, c5, d5 = _capi.parse_corpus([], ["a"])
    assert p5.tolist() == [0] and t5.size == 0 and d5 == 0


def test_exports_match_reference_bytes(tmp_path):
    """export_beta / export_gamma (variational_bayes.py:326-356) against the bytes the reference
    itself wrote for the tiny corpus (tests/golden/make_golden.py::make_tiny_exports): header lines,
    descending order, `%g` formatting, top_display cut."""
    from pylda_amd.variational_bayes import VariationalBayes
    g = load_golden("tiny_exports.npz")
    m = VariationalBayes()
    m._verbose = False
    m.parse_vocabulary([str(w) for w in g["words"]])
    m._number_of_topics, m._number_of_documents = g["eta"].shape[0], g["gamma"].shape[0]
    m._eta = g["eta"].copy()
    m._gamma = g["gamma"].copy()
    for name, fn, top in (("exp_beta", m.export_beta, -1), ("exp_beta_top2", m.export_beta, 2),
                          ("exp_gamma", m.export_gamma, -1), ("exp_gamma_top2", m.export_gamma, 2)):
        path = tmp_path / name
        fn(str(path), top)
        assert path.read_bytes() == bytes(g[name]), name


def test_native_ingest_python_whitespace_and_awkward_vocabulary():
    """The tokeniser splits exactly where str.split() does (Unicode white space, control separators), a
    document that contains line breaks stays one document, and vocabulary entries no token can match
    (empty, containing or surrounded by white space) keep their ids without shifting the others."""
    from pylda_amd import _capi
    vocab = ["a", "", "b c", "d", " e", "f ", "g", "　", "h\x00i", "été"]
    index = {w: i for i, w in enumerate(vocab)}
    lines = ["a d g\x1cg\x85a", "d\ng\r\na", "a b c d", "　  ", "e f g", "g\x00a", "h\x00i g\x01",
             "a\x0bd\x0cg\x1d\x1e\x1fg", " a  d ", "été a d g g a d g",
             "a​d"]                                   # U+200B is NOT white space for str.split()
    ptr, tid, tct, dropped = _capi.parse_corpus(lines, vocab)
    ref_ids, ref_cts = reference_rules_parse(lines, index)
    assert len(ref_ids) == 8                               # the test data exercises both kept and dropped lines
    assert len(ptr) - 1 == len(ref_ids) and dropped == len(lines) - len(ref_ids)
    for d, (ri, rc) in enumerate(zip(ref_ids, ref_cts)):
        assert tid[ptr[d]:ptr[d + 1]].tolist() == ri and tct[ptr[d]:ptr[d + 1]].tolist() == rc
    # the separator search falls through to 0xFF when every control candidate occurs in the text
    hard = ["a\x00d", "g\x01a", "d\x02g", "a\x03g", "d"]
    p2, t2, c2, _ = _capi.parse_corpus(hard, vocab)
    r2, _ = reference_rules_parse(hard, index)
    assert len(p2) - 1 == len(r2) and [t2[p2[d]:p2[d + 1]].tolist() for d in range(len(r2))] == r2


def test_bench_launcher_command_line():
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run."""
    import bench
    argv = bench.launcher_argv(4, ["--gpus", "4", "--steps", "7"])
    assert argv[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert argv[argv.index("--nproc-per-node") + 1] == "4" and argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert argv[-5].endswith("bench.py") and argv[-4:] == ["--gpus", "4", "--steps", "7"]
    assert bench.algorithmic_bytes(10, 2, 4) == 10 * (8 + 64) + 2 * (32 + 8)


def test_hybrid_corpus_generator_is_the_numpy_generator():
    """The generator bench.py uses (numpy PCG64 draws, collapse on a torch device, chunks drawn ahead by
    threads) gives the all-numpy corpus bit for bit; shards are slices of the whole."""
    a = C.synthetic_lda_shard(900, 300, 0, 900, 7, 40, seed=11, chunk=300)
    b = C.synthetic_lda_shard(900, 300, 0, 900, 7, 40, seed=11, chunk=300, device="cpu", workers=3)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    s = C.synthetic_lda_shard(900, 300, 300, 900, 7, 40, seed=11, chunk=300, device="cpu", workers=2)
    assert np.array_equal(s[1], a[1][a[0][300]:]) and np.array_equal(s[0], a[0][300:] - a[0][300])
    assert C.corpus_checksum(*a) == [900, int(a[0][-1]), int(a[1].astype(np.int64).sum()), int(a[2].sum())]


def test_bench_all_cores_cpu_leg_runs_workers_and_adds_rates():
    """bench.py's optional all-cores CPU figure: single-threaded oracle processes side by side, stragglers killed."""
    import bench
    rng = np.random.default_rng(0)
    K, V, D = 8, 120, 80
    ptr, ids, cts = [0], [], []
    for _ in range(D):
        u, c = np.unique(rng.choice(V, size=20), return_counts=True)
        ids.append(u)
        cts.append(c)
        ptr.append(ptr[-1] + u.size)
    rate, done, workers = bench.cpu_baseline_all_cores(
        np.full(K, 0.1), rng.gamma(100.0, 0.01, (K, V)), np.array(ptr, np.int64),
        np.concatenate(ids).astype(np.int32), np.concatenate(cts).astype(np.int32), 0.2, 3, docs_per_worker=20)
    assert workers == 3 and 15 <= done <= 60 and rate > 0


def test_traffic_record_is_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    """bench.py reports roofline.traffic from profiles/traffic_<workload>.json only while the sha256 of the kernel
    sources equals the one the rocprofv3 passes were made with; otherwise null and a STALE note."""
    import json
    import bench
    want = bench.kernel_source_hash()
    assert len(want) == 16 and want == bench.kernel_source_hash()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    (tmp_path / "pylda_amd" / "csrc").mkdir(parents=True)
    (tmp_path / "pylda_amd" / "csrc" / "estep_x.h").write_text("// kernel\n")
    now = bench.kernel_source_hash()
    assert now != want
    path = tmp_path / "profiles" / "traffic_synth100k.json"
    path.write_text(json.dumps({"hbm_bytes_per_launch": 123, "kernel_source_hash": now}))
    assert bench.traffic_record("synth100k")[0] == 123
    (tmp_path / "pylda_amd" / "csrc" / "estep_x.h").write_text("// kernel, edited\n")
    value, note = bench.traffic_record("synth100k")
    assert value is None and "STALE" in note
    assert bench.traffic_record("nips") == (None, None) and bench.traffic_record(None) == (None, None)
    # host translation units are not kernel sources: an edit to the launcher (round 4: a test hook in estep_api.hip)
    # must not stale the committed traffic figure; an edit to any device header must
    (tmp_path / "pylda_amd" / "csrc" / "estep_x.h").write_text("// kernel\n")
    assert bench.traffic_record("synth100k")[0] == 123
    for host_unit in ("estep_api.hip", "sstats_gather.hip", "plan.hip", "sstats_plan.cpp", "ingest.cpp"):
        (tmp_path / "pylda_amd" / "csrc" / host_unit).write_text("// host code, edited\n")
        assert bench.traffic_record("synth100k")[0] == 123, host_unit
    for device_header in ("sstats_sweep.h", "doc_terms.h", "special_device.h", "estep_common.h"):
        (tmp_path / "pylda_amd" / "csrc" / device_header).write_text("// device code\n")
        assert bench.traffic_record("synth100k")[0] is None, device_header
        (tmp_path / "pylda_amd" / "csrc" / device_header).unlink()
    # ... and the real tree: every file the hash covers is a header, none a translation unit
    import os
    real = os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "pylda_amd", "csrc")
    touched = tmp_path / "copy"
    import shutil
    shutil.copytree(real, touched)
    base = bench.kernel_source_hash(str(touched))
    with open(touched / "estep_api.hip", "a") as fh:
        fh.write("// a test hook\n")
    assert bench.kernel_source_hash(str(touched)) == base
    with open(touched / "estep_quad.h", "a") as fh:
        fh.write("// a kernel edit\n")
    assert bench.kernel_source_hash(str(touched)) != base


def test_launch_train_flags_and_launcher_command():
    """The reference's flags (launch_train.py:31-62) plus --gpus / --share_gpu; `--gpus N` turns into a
    torch.distributed.run command line on 127.0.0.1 that re-runs the same module with the same arguments."""
    from pylda_amd import cli
    names = [f[0] for f in cli.TRAIN_FLAGS]
    for flag in ("input_directory", "output_directory", "number_of_topics", "training_iterations", "snapshot_interval",
                 "alpha_alpha", "alpha_beta", "inference_mode", "gpus"):
        assert flag in names
    argv = cli._launcher_argv(4, ["--input_directory=x/", "--gpus=4"])
    assert argv[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node" in argv
    assert argv[argv.index("--nproc-per-node") + 1] == "4" and argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert argv[-4:] == ["-m", "pylda_amd.launch_train", "--input_directory=x/", "--gpus=4"]


def test_shards_of_a_corpus_smaller_than_the_world():
    """More ranks than documents: trailing ranks get empty, well-formed shards."""
    from pylda_amd.corpus import shard_bounds, shard_csr
    ptr = np.array([0, 3, 5], np.int64)
    ids, cts = np.arange(5, dtype=np.int32), np.ones(5, np.int32)
    bounds = shard_bounds(ptr, 4)
    assert bounds[0] == 0 and bounds[-1] == 2 and all(a <= b for a, b in zip(bounds, bounds[1:]))
    seen = 0
    for rank in range(4):
        sp, si, sc, (lo, hi) = shard_csr(ptr, ids, cts, 4, rank)
        assert sp[0] == 0 and sp[-1] == si.size == sc.size and len(sp) == hi - lo + 1
        seen += hi - lo
    assert seen == 2


def test_traffic_hash_covers_every_device_header_of_the_estep():
    """The other direction of the guard: every csrc header that defines a kernel (or device code a kernel inlines) and is
    included - directly or through other headers - by the translation units that launch the E-step's kernels must be part
    of bench.kernel_source_hash(); a new kernel header with a name the hash does not match would otherwise let a stale
    profiles/traffic_*.json pass as current."""
    import os
    import re
    import bench
    csrc = os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "pylda_amd", "csrc")
    hashed = {f for f in os.listdir(csrc)
              if f.endswith(".h") and (f.startswith("estep_") or f.startswith("sstats_") or f in ("doc_terms.h", "special_device.h", "prepare_kernels.h"))}

    def includes(name, seen):
        for inc in re.findall(r'#include "([^"/]+)"', open(os.path.join(csrc, name)).read()):
            if inc not in seen and os.path.exists(os.path.join(csrc, inc)):
                seen.add(inc)
                includes(inc, seen)
        return seen

    reached = set()
    for unit in ("estep_api.hip", "launch_small.hip", "launch_quad.hip", "launch_stream.hip", "sstats_gather.hip"):
        includes(unit, reached)
    device = {h for h in reached if re.search(r"__global__|__device__", open(os.path.join(csrc, h)).read())}
    host_side = {"host_internal.h", "host_plan.h", "comm.h", "postings.h"}          # declarations only / host code
    assert device - host_side <= hashed, sorted(device - host_side - hashed)
    # ... and the hash really reads them: touching a copy of each changes it
    import shutil, tempfile
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, "csrc")
        shutil.copytree(csrc, copy)
        base = bench.kernel_source_hash(copy)
        for h in sorted(device - host_side):
            with open(os.path.join(copy, h), "a") as fh:
                fh.write("// edit\n")
            now = bench.kernel_source_hash(copy)
            assert now != base, h
            base = now


def test_learning_detects_an_overridden_seam():
    """learning() takes the fused device path only while e_step / m_step are the class' own (the reference dispatches
    through self.e_step() / self.m_step(): variational_bayes.py:243-247, hybrid.py:23,85): the detection, and the
    reference's order of calls through the seam, without a GPU."""
    from pylda_amd.variational_bayes import VariationalBayes
    calls = []

    class Wrapped(VariationalBayes):
        def e_step(self, parsed_corpus=None, local_parameter_iteration=50, local_parameter_converge_threshold=1e-6):
            calls.append("e")
            return -10.0, np.ones((2, 3))

        def m_step(self, phi_sufficient_statistics):
            calls.append(("m", phi_sufficient_statistics.shape))
            return -5.0, np.array([1.0, 2.0])

        def optimize_hyperparameters(self, alpha_sufficient_statistics, **kwargs):
            calls.append(("a", tuple(alpha_sufficient_statistics)))

    m = Wrapped()
    m._verbose = False
    m._counter = 0
    assert m._seam_is_overridden()
    assert m.learning() == -15.0                       # :252 joint = document + topic log-likelihood
    assert calls == ["e", ("m", (2, 3)), ("a", (1.0, 2.0))] and m._counter == 1
    m._hyper_parameter_optimize_interval = 2           # :247: every second iteration only
    calls.clear()
    m.learning()
    m.learning()
    assert [c for c in calls if c[0] == "a"] == [("a", (1.0, 2.0))]
    plain = VariationalBayes()
    assert not plain._seam_is_overridden()
    plain.e_step = lambda *a, **k: None                # a method patched on the instance counts as well
    assert plain._seam_is_overridden()
