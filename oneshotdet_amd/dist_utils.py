"""Data-parallel gradient exchange: the ONE collective of the training step (tools/train_net.py:83-88 uses
DistributedDataParallel; here the gradients already live in one flat fp32 buffer, so the exchange is a few large
contiguous all-reduces — RCCL over xGMI with backend "nccl", gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def average_flat_(flat, group=None, n_buckets=4):
    """In-place average of a flat 1-D tensor over the ranks of `group`; no-op without an initialised process group or
    with a single rank.  Buckets are contiguous slices, so no packing copies are needed."""
    if not (dist.is_available() and dist.is_initialized()):
        return flat
    world = dist.get_world_size(group)
    if world == 1:
        return flat
    n = flat.numel()
    if n == 0:
        return flat
    chunk = max(1, (n + n_buckets - 1) // n_buckets)
    works = [dist.all_reduce(flat[i:i + chunk], group=group, async_op=True) for i in range(0, n, chunk)]
    for w in works:
        w.wait()
    flat.mul_(1.0 / world)
    return flat
