"""Data-parallel plumbing: one process per GPU, batch shards, ONE flat-bucket all-reduce per iteration.

The reference has no distributed code at all; its loops shard naturally over the batch:
samples interact only through (i) the mean in the BCE loss, (ii) the gradient sum, (iii) the generators' BatchNorm
batch statistics (kept per rank: generators are never trained, their running statistics are rank-local state).
So each rank runs the fused step on its shard, the trainers put every discriminator gradient plus the local
discriminator-loss mean in one contiguous fp32 bucket, `allreduce_bucket_` SUMs it over RCCL (backend "nccl" on
ROCm; xGMI inside a node) and the fused Adam kernel reads the bucket scaled by 1/world.  With equal shard sizes the
mean of the per-rank means is the global mean, so the result equals a single process on the global batch up to
summation order.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, world, device_index).

    Rehearsal switches (a one-GPU box cannot run RCCL with two ranks): GDM_DIST_BACKEND=gloo selects the gloo backend
    (it moves device tensors through the host), GDM_SINGLE_DEVICE=1 places every rank on device 0.
    """
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1, 0
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("GDM_DIST_BACKEND", backend)
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    dev = 0 if os.environ.get("GDM_SINGLE_DEVICE") == "1" else local
    if torch.cuda.is_available():
        torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group(backend, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend)
    return dist.get_rank(), dist.get_world_size(), dev


def world_size(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group)
    return 1


def shard_bounds(global_batch, world, rank):
    """Contiguous equal shards (the global batch must divide evenly so that mean-of-means == global mean)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def allreduce_bucket_(bucket, n_reduce, group=None):
    """In-place SUM all-reduce of bucket[:n_reduce] (gradients | loss scalar); returns the factor (1/world) the
    consumer has to apply (folded into the Adam kernel's grad_scale and into the loss read-out)."""
    w = world_size(group)
    if w > 1:
        dist.all_reduce(bucket[:n_reduce], op=dist.ReduceOp.SUM, group=group)
    return 1.0 / w


def allreduce_async_(tensor, group=None):
    """Start an in-place SUM all-reduce of a contiguous tensor; returns the work handle (None on a single rank).
    The collective is ordered after the work already enqueued on the current stream; ``handle.wait()`` makes the
    then-current stream wait for it."""
    if world_size(group) <= 1:
        return None
    return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group, async_op=True)


def all_gather_cat(tensor, group=None):
    """Concatenation along dim 0 of every rank's ``tensor`` (same shape on all ranks), in rank order; the tensor itself
    on a single rank.  Used for the exact global-batch BatchNorm statistics (per-rank Welford partials)."""
    w = world_size(group)
    if w <= 1:
        return tensor
    parts = [torch.empty_like(tensor) for _ in range(w)]
    dist.all_gather(parts, tensor.contiguous(), group=group)
    return torch.cat(parts, dim=0)
