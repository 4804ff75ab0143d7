"""Document-sharded data parallelism for the E-step: one process per GPU,
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

Documents are independent given (alpha, E_log_eta), so a
rank runs the E-step on its own contiguous, nnz-balanced shard and the only
exchange per outer iteration is ONE all-reduce(sum) of the K*V sufficient
statistics (device-resident, in place) plus one tiny all-reduce of
(document log-likelihood, #documents, alpha statistics).  Every rank then
performs the identical M-step.  Summation order differs from a single-GPU
run at the 1e-16 level only.
"""
import numpy as np


class _DevicePointer(object):
    """Zero-copy view of library-owned device memory for torch.as_tensor."""

    def __init__(self, ptr, shape, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


def device_tensor(ptr, shape, device):
    """Zero-copy torch view of `shape` doubles at device address `ptr`.  Fails loudly if torch made a copy instead
    (a pointer it attributes to another device): a collective on a copy would silently reduce nothing."""
    import torch
    t = torch.as_tensor(_DevicePointer(ptr, shape), device=device)
    if t.data_ptr() != int(ptr):
        raise RuntimeError("pylda_amd.distributed: torch copied the library's device buffer (pointer %#x on %s) instead of "
                           "wrapping it" % (int(ptr), device))
    return t


def bind_to_torch_stream(ctx):
    """Run the context on a dedicated torch stream (kept on the context), so that collectives
    issued through torch.distributed under `torch.cuda.stream(ctx._torch_stream)` are ordered
    with the library's kernels on the device, no host synchronisation in between.  (Torch's
    DEFAULT stream has handle 0, the legacy null stream: it would work - pylda_set_stream(NULL)
    means exactly that stream - but it synchronises implicitly with every blocking stream of
    the process; a stream of our own does not.)"""
    import torch
    torch.cuda.set_device(ctx.device)
    stream = torch.cuda.Stream(device=ctx.device)
    ctx.set_stream(stream.cuda_stream)
    ctx._torch_stream = stream          # keeps the stream alive as long as the context uses it
    return stream


def _stream_scope(ctx):
    import contextlib
    import torch
    stream = getattr(ctx, "_torch_stream", None)
    if stream is None:
        # the context runs on its private stream: order by draining it (host sync)
        ctx.synchronize()
        return contextlib.nullcontext()
    return torch.cuda.stream(stream)


def allreduce_sstats(ctx, group=None):
    """In-place all-reduce(sum) of the (V, ldk) sufficient statistics, ordered after the E-step's
    kernels and before whatever the context enqueues next (the M-step).

    RCCL (backend "nccl"): zero-copy on the library's device buffer.  Any other backend (gloo in
    the single-GPU multi-rank tests): staged through a host copy on the same stream."""
    import torch
    import torch.distributed as dist
    t = device_tensor(ctx.sstats_device_ptr(), (ctx.sstats_elements(),), torch.device("cuda", ctx.device))
    with _stream_scope(ctx):
        if dist.get_backend(group) == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        else:
            host = t.cpu()                                   # waits for the E-step on this stream
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            t.copy_(host)
            torch.cuda.current_stream(ctx.device).synchronize()   # `host` is freed on return
        if getattr(ctx, "_torch_stream", None) is None:
            torch.cuda.current_stream(ctx.device).synchronize()
    ctx.mark_device_state(have_sstats=1)
    return t


def allreduce_outer(ctx, group=None):
    """In-place sum over the ranks of the rank-local part of the packed outer-iteration values
    (pylda_outer_device: document log-likelihood, #documents, log-space documents, alpha statistics),
    on the context's stream between pylda_mstep_enqueue and pylda_outer_fetch: no host wait with RCCL."""
    import torch
    import torch.distributed as dist
    ptr, _, n_reduce = ctx.outer_device()
    t = device_tensor(ptr, (n_reduce,), torch.device("cuda", ctx.device))
    with _stream_scope(ctx):
        if dist.get_backend(group) == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        else:
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            t.copy_(host)
            torch.cuda.current_stream(ctx.device).synchronize()
        if getattr(ctx, "_torch_stream", None) is None:
            torch.cuda.current_stream(ctx.device).synchronize()
    return t


def allreduce_small(group, document_log_likelihood, number_of_documents, alpha_ss, device=None):
    """Sum (LL, D, alpha sufficient statistics) over ranks; returns python/numpy values."""
    import torch
    import torch.distributed as dist
    K = alpha_ss.shape[0]
    buf = torch.empty(K + 2, dtype=torch.float64)
    buf[0] = float(document_log_likelihood)
    buf[1] = float(number_of_documents)
    buf[2:] = torch.from_numpy(np.ascontiguousarray(alpha_ss, dtype=np.float64))
    if device is None and dist.get_backend(group) == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    if device is not None:
        buf = buf.to(device)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    buf = buf.cpu()
    return float(buf[0]), int(round(float(buf[1]))), buf[2:].numpy().copy()


def allreduce_array_(array, group=None):
    """In-place sum of a host numpy array over ranks (gloo path, used by tests)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(array)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return array


def allreduce_host_array(array, group=None, device=None):
    """Sum of a host numpy array over the ranks, returned as a host array (the public e_step()/m_step() seam keeps the
    reference's host-array contract: RCCL reduces a device copy, gloo the array itself)."""
    import torch
    import torch.distributed as dist
    array = np.ascontiguousarray(array, dtype=np.float64)
    if dist.get_backend(group) == "nccl":
        t = torch.from_numpy(array).to(torch.device("cuda", torch.cuda.current_device() if device is None else device))
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.cpu().numpy()
    return allreduce_array_(array.copy(), group)
