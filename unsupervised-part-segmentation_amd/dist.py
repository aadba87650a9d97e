"""Data parallelism over one 8xMI355X node: one process per GPU, torch.distributed backend "nccl"
(= RCCL over xGMI).  The reference has no parallelism at all; the design here is pure DP:

  * the batch axis is sharded, weights are replicated (identical per-name seeds on every rank);
  * gradients live in ONE flat fp32 buffer per optimizer key (7 buckets: 0.8 ... 54.6 MB), each bucket is
    all-reduced (sum) asynchronously as soon as its backward segment is complete and the 1/world factor is
    folded into the fused Adam kernel -- no per-tensor collectives, no extra scaling pass;
  * the ~6 batch-mean scalars that drive the Lagrangian / EMA state (model.py:801-866) are averaged with one
    tiny all-reduce so the replicas' state stays identical.
"""
import torch
import torch.distributed as dist

from . import switches as SW


# UPS_FORCE_COLLECTIVES=1: issue the gradient all-reduces even at world size 1 (identity) -- exercises the RCCL call pattern
# (asynchronous bucket all-reduces launched from inside the backward pass beside the side streams) on a single-GPU box
FORCE_COLLECTIVES = SW.flag("UPS_FORCE_COLLECTIVES")
# UPS_DP_STANDIN=1 (single-GPU measurement of the data-parallel stream budget, tools/probes/stream_dp.py): every bucket "all-reduce"
# is an out-of-place device copy of the bucket on a stream of its own, started where the real collective would start and waited
# for where the real one is -- the load an RCCL kernel puts on HBM and on a hardware queue, without a second GPU.  The values are
# untouched (world size 1: the sum over ranks is the bucket itself).
STANDIN = SW.flag("UPS_DP_STANDIN")
FORCE_COLLECTIVES = FORCE_COLLECTIVES or STANDIN
_standin = {}


class _StandinWork(object):
    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


def _standin_copy(flat_grad):
    dev = flat_grad.device
    st = _standin.get(("stream", dev))
    if st is None:
        st = _standin[("stream", dev)] = torch.cuda.Stream(device=dev)
    buf = _standin.get(("buf", dev))
    if buf is None or buf.numel() < flat_grad.numel():
        buf = _standin[("buf", dev)] = torch.empty(max(flat_grad.numel(), 16 << 20), dtype=flat_grad.dtype, device=dev)
    st.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(st):
        buf[:flat_grad.numel()].copy_(flat_grad)
        ev = st.record_event()
    return _StandinWork(ev)


def init_from_env(backend="nccl"):
    """torchrun / torch.distributed.run contract: RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, init_method="env://", world_size=world, rank=rank)
    return world, rank, local


def allreduce_bucket(flat_grad, world_size, group=None, async_op=True):
    """Sum-all-reduce one optimizer key's flat gradient; returns the work handle (or None)."""
    if world_size <= 1 and not FORCE_COLLECTIVES:
        return None
    if STANDIN and world_size <= 1 and flat_grad.is_cuda:
        return _standin_copy(flat_grad)
    return dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def wait_all(handles):
    for h in handles:
        if h is not None:
            h.wait()


def average_scalars(stats, world_size, group=None):
    """stats: 1-D tensor of batch-mean scalars -> mean over ranks (in place)."""
    if world_size > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
        stats /= world_size
    return stats


def sum_flag(flag, world_size, group=None):
    """flag: 1-element tensor -> sum over ranks (in place): a collective "did any rank fail" decision on log steps."""
    if world_size > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=group)
    return flag


def shard_seed(base_seed, rank):
    """Rank-offset seeds for data / noise (weights use the SAME seed on every rank)."""
    return int(base_seed) + 7919 * int(rank)
