"""Multi-GPU plumbing: one process per GPU over RCCL/xGMI.

* Inference shards *scenes*: every depth map is independent, so ranks take scenes round-robin and
  there is no data-path collective (``shard_scenes``).
* Training (BASELINE config 4) is data parallel with one scene per GPU and per-replica BatchNorm, like
  the reference's ``nn.DataParallel`` (reference train.py:177).  Gradients are combined with ONE
  all-reduce of a single flat float32 bucket (698 936 parameters = 2.8 MB: latency-bound on xGMI, so
  bucketing finer would only add launches).  The reduce op is SUM, not mean: the reference loss sums
  the per-sample MAE over the batch (networks.py:176-179) and DataParallel reduce-adds replica
  gradients, so averaging would scale the step by 1/world_size.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); no-op for 1 process.
    Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size(group=None):
    """Ranks in ``group`` (1 without an initialised process group)."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group)
    return 1


def shard_scenes(num_scenes, rank, world_size):
    """Round-robin scene ownership; the union over ranks is exactly range(num_scenes)."""
    return list(range(rank, num_scenes, world_size))


def allreduce_gradients_sum(module, group=None):
    """Sum gradients across ranks through one flat bucket.  Parameters without a gradient contribute
    zeros so every rank reduces the same layout.  Returns the number of elements reduced."""
    params = [p for p in module.parameters() if p.requires_grad]
    if not params:
        return 0
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float()
                      for p in params])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    offset = 0
    for p in params:
        n = p.numel()
        if p.grad is None:
            p.grad = torch.empty_like(p)
        p.grad.copy_(flat[offset:offset + n].view_as(p))
        offset += n
    return int(flat.numel())


class GradBucket(object):
    """The single flat float32 gradient bucket of the data-parallel training step, kept for the life of the
    model: every parameter's ``.grad`` is a VIEW into it, so backward accumulates straight into the bucket and
    the step's gradient exchange is exactly one in-place ``all_reduce(SUM)`` (2.8 MB for PointMVSNet's 698 936
    parameters) with no gather / scatter copies around it."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise RuntimeError("GradBucket: the module has no trainable parameters")
        dev = self.params[0].device
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
        offset = 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise RuntimeError("GradBucket: parameters must be float32 on one device")
            n = p.numel()
            p.grad = self.flat[offset:offset + n].view_as(p)
            offset += n

    def numel(self):
        return int(self.flat.numel())

    def zero_(self):
        """Replaces optimizer.zero_grad(): the views stay attached to the bucket."""
        self.flat.zero_()

    def attached(self):
        base = self.flat.untyped_storage().data_ptr()
        return all(p.grad is not None and p.grad.untyped_storage().data_ptr() == base for p in self.params)

    def allreduce_sum(self, group=None):
        """One SUM all-reduce over RCCL/xGMI (gloo on CPU); a no-op for a single process."""
        if not self.attached():
            raise RuntimeError("GradBucket: a parameter's .grad was replaced (optimizer.zero_grad(set_to_none=True)?); "
                               "use GradBucket.zero_()")
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        return self.numel()


def broadcast_parameters(module, src=0, group=None):
    """Make every replica start from rank ``src``'s parameters and buffers."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def assert_replicas_equal(flat, group=None):
    """Raise unless every rank of ``group`` holds the same ``flat`` tensor: a float64 (sum, sum of squares) checksum is
    all-reduced with MIN and with MAX and the two must agree exactly (replicas that applied the same updates in the
    same order hold the same bits, hence the same sums).  Two small collectives, meant for every N-th step.  Returns the
    checksum as a tuple of floats; a no-op returning the local checksum for one process."""
    v = flat.detach().double()
    local = torch.stack([v.sum(), (v * v).sum()])
    if world_size(group) <= 1:
        return tuple(float(x) for x in local)
    lo, hi = local.clone(), local.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if not torch.equal(lo, hi):
        raise RuntimeError("replica parameters diverged: checksum range [%r, %r] over %d ranks (rank %d holds %r); every "
                           "rank must start from the same model (TrainStep broadcasts rank 0's) and apply every step"
                           % (lo.tolist(), hi.tolist(), world_size(group), dist.get_rank(group), local.tolist()))
    return tuple(float(x) for x in local)
