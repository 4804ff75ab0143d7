"""The RCCL path on one GPU: a world-size-1 NCCL process group drives exactly the code
bench.py --gpus N runs per rank (zero-copy wrap of the library's device buffer,
stream binding, in-place all-reduce, small all-reduce)."""
import os
import socket

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.fixture(scope="module")
def nccl_group():
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist.group.WORLD
    dist.destroy_process_group()


def test_device_buffer_wrap_and_allreduce(nccl_group, ap_train):
    import torch
    from pylda_amd import _capi, distributed
    g = ap_train
    ctx = _capi.Context(10, 6806)
    distributed.bind_to_torch_stream(ctx)
    corpus = ctx.corpus(g["doc_ptr"][:201], g["term_id"][:g["doc_ptr"][200]], g["term_ct"][:g["doc_ptr"][200]])
    ctx.set_alpha(g["alpha"])
    ctx.set_eta(g["eta"])
    ctx.estep(corpus)
    before = ctx.get_sstats()
    t = distributed.allreduce_sstats(ctx, nccl_group)          # sum over 1 rank = identity, in place
    assert t.is_cuda and t.dtype == torch.float64 and t.numel() == ctx.sstats_elements()
    assert t.data_ptr() == ctx.sstats_device_ptr()             # zero-copy view of the library's buffer
    torch.cuda.synchronize()
    assert np.array_equal(ctx.get_sstats(), before)
    t.mul_(2.0)                                                # writes through to the library's memory
    torch.cuda.synchronize()
    assert np.array_equal(ctx.get_sstats(), 2.0 * before)
    ll, D, ass = distributed.allreduce_small(nccl_group, -12.5, 200, np.arange(10.0))
    assert ll == -12.5 and D == 200 and np.array_equal(ass, np.arange(10.0))
    corpus.close()
    ctx.close()


def test_sharded_learning_matches_single_process(nccl_group, ap_train):
    """VariationalBayes with a process group (world size 1) follows the same trajectory as without."""
    from pylda_amd.variational_bayes import VariationalBayes
    g = ap_train
    ptr = g["doc_ptr"][:401]
    ids, cts = g["term_id"][:ptr[-1]], g["term_ct"][:ptr[-1]]
    traces = []
    for group in (None, nccl_group):
        m = VariationalBayes(process_group=group)
        m._verbose = False
        m._initialize_parsed(ptr, ids, cts, 6806, 10, 0.1, 1.0 / 6806, eta=g["eta"].copy())
        traces.append([m.learning() for _ in range(3)] + [m._alpha_alpha.copy(), m._eta.copy()])
    a, b = traces
    assert a[:3] == b[:3]
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
